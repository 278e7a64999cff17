"""Drop-in for the reference's utils/bbox.py: decode_outputs (utils/bbox.py:13-79) and post_process
(utils/bbox.py:82-159) with the same signatures and return types, computed by libccnet_hip.so on the GPU.

Differences a caller can observe (documented in DESIGN.md):
  * decode_outputs returns a contiguous (B, A, 5+nc) tensor (the reference returns a permuted view of the same values);
  * post_process does not overwrite prediction[:, :, :4] with corner boxes (the reference does, as a side effect);
  * there is one device->host sync per batch (the candidate counts), not one per image.
NMS follows torchvision 0.9.0 `batched_nms` (coordinate trick) with a stable descending sort; keep order and
indices are bit-exact against the oracle's scalar restatement.
"""
import threading

import numpy as np
import torch

from .. import ops

# the two small caches below are shared by every request thread of the socket server
_cache_lock = threading.Lock()


def _require_cuda(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("ccnet_amd.%s needs CUDA (ROCm) tensors; this build has no CPU path" % what)


_hw_cache = {}


def _image_hw(hw, device):
    """int32 (B, 2) device tensor of (image_h, image_w); batches of a dataset mostly repeat the same sizes, so the small
    host->device copy is made once per distinct tuple (bounded cache)."""
    key = (hw, str(device))
    with _cache_lock:
        t = _hw_cache.get(key)
    if t is None:
        t = torch.tensor(hw, dtype=torch.int32).reshape(-1, 2).to(device)
        with _cache_lock:
            if len(_hw_cache) >= 64:
                _hw_cache.clear()
            _hw_cache[key] = t
    return t


def decode_outputs(outputs, input_shape):
    """outputs: list of (B, 5+nc, h_k, w_k) fp32 CUDA tensors (any b/c strides); input_shape = [in_h, in_w].
    Returns (B, sum h_k*w_k, 5+nc): [cx, cy, w, h] normalised, obj and class probabilities."""
    outputs = list(outputs)
    for o in outputs:
        _require_cuda(o, "decode_outputs")
    heads = []
    for o in outputs:
        o = o.float()
        if o.stride(2) != o.shape[3] * o.stride(3):
            o = o.contiguous()
        heads.append(o)
    B, nch = heads[0].shape[0], heads[0].shape[1]
    A = sum(int(h.shape[2]) * int(h.shape[3]) for h in heads)
    out = torch.empty(B, A, nch, dtype=torch.float32, device=heads[0].device)
    if B * A:
        # the same kernel leaves obj * max class probability per anchor beside the rows; post_process reads those 4 bytes per
        # anchor for its threshold test as long as the tensor it is given is this one, unmodified (_scores_of)
        scores = torch.empty(B, A, dtype=torch.float32, device=out.device) if 5 < nch <= ops.DECODE_SCORE_MAX_NCH else None
        with torch.cuda.device(out.device):  # launches go to the tensors' device, whatever the caller's current device is
            ops.decode(heads, nch - 5, input_shape[0], input_shape[1], out, scores)
        if scores is not None:
            out._ccnet_scores = (scores, out._version, out.data_ptr())
    return out


def _scores_of(prediction, pred, num_classes):
    """The per-anchor scores decode_outputs left beside `prediction`, or None: only when `pred` (what the kernels will read) IS
    that tensor, no in-place operation has touched it since (torch's version counter) and the class count is the decoded one."""
    side = getattr(prediction, "_ccnet_scores", None)
    if side is None or pred is not prediction:
        return None
    scores, version, ptr = side
    if prediction._version != version or prediction.data_ptr() != ptr or prediction.shape[2] != 5 + num_classes:
        return None
    if scores.shape != prediction.shape[:2] or scores.device != prediction.device:
        return None
    return scores


_nmax_guess = {}


# Buffers of finished calls, reused by later calls of the same (B, A, rows, device): candidates, result rows + counts, NMS
# workspace, keep indices and the pinned host copy.  Five allocations per call were a third of post_process' host time
# (scripts/post_host_profile.py); a buffer set goes back only after its call's event has been waited for, so nothing on the GPU
# still uses it, whichever stream the next call runs on.
_pool = {}


def _acquire(B, A, nmax, dev):
    key = (B, A, nmax, str(dev))
    with _cache_lock:
        free = _pool.get(key)
        if free:
            return key, free.pop()
    cand = torch.empty(B, A, 7, dtype=torch.float32, device=dev)
    both = _result_buffer(B, nmax, dev)
    ws = torch.empty(ops.nms_workspace_bytes(B, nmax) + 256, dtype=torch.uint8, device=dev)
    keep_idx = torch.empty(B, nmax, dtype=torch.int32, device=dev)
    host = torch.empty(both.numel(), dtype=torch.float32, pin_memory=True)
    return key, (cand, both, ws, keep_idx, host)


def _release(key, bufs):
    with _cache_lock:
        if key not in _pool and len(_pool) >= 16:
            _pool.clear()
        free = _pool.setdefault(key, [])
        if len(free) < 4:
            free.append(bufs)


class PendingDetections:
    """Detections of one batch whose kernels and device->host copy have been submitted (post_process_async).  `result()` waits
    for the copy and returns what post_process returns.  Holding the object keeps every buffer the kernels use alive."""

    def __init__(self, pred, num_classes, conf_thres, nms_thres, image_hw, scores=None):
        self._done = None
        self.scores = scores
        B, A, _ = pred.shape
        dev = pred.device
        self.B, self.A, self.key = B, A, (B, A, str(dev))
        self.pred, self.nms_thres, self.image_hw = pred, nms_thres, image_hw
        with _cache_lock:
            nmax = _nmax_guess.get(self.key)
        if nmax is None:
            # first call for this (B, A, device): wait for the counts, then run NMS with the exact maximum
            cand = torch.empty(B, A, 7, dtype=torch.float32, device=dev)
            counts = torch.empty(B, dtype=torch.int32, device=dev)
            ops.filter_candidates(pred, num_classes, conf_thres, cand, counts, scores)
            self._exact(cand, counts, min(max(int(counts.cpu().max()), 1), A))
        else:
            self.nmax = min(nmax, A)
            self.pool_key, self.bufs = _acquire(B, A, self.nmax, dev)
            self.cand, self.both = self.bufs[0], self.bufs[1]
            # the filter kernel writes its counts straight into the result buffer
            self.counts = self.both[B * self.nmax * 7 + B:].view(torch.int32)
            ops.filter_candidates(pred, num_classes, conf_thres, self.cand, self.counts, scores)
            self._submit()

    def _exact(self, cand, counts, nmax):
        """NMS over candidates that are already there, with rows for the known maximum."""
        B, dev = self.B, cand.device
        self.nmax = nmax
        self.pool_key, self.bufs = _acquire(B, self.A, nmax, dev)
        self.both = self.bufs[1]
        self.both[B * nmax * 7 + B:].view(torch.int32).copy_(counts)
        self.cand, self.counts = cand, counts
        self._submit()

    def _submit(self):
        _, both, ws, keep_idx, host = self.bufs
        _nms(self.cand, self.counts, self.nmax, self.nms_thres, self.image_hw, both, ws, keep_idx)
        self.host = host
        self.host.copy_(both, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def result(self):
        if self._done is not None:
            return self._done
        B, A = self.B, self.A
        while True:
            self.event.synchronize()
            nmax = self.nmax
            host = self.host.numpy()
            cmax = int(host[B * nmax * 7 + B:].view(np.int32).max())
            with _cache_lock:
                if len(_nmax_guess) >= 64:
                    _nmax_guess.clear()
                _nmax_guess[self.key] = min(A, max(64, -(-(cmax + cmax // 4) // 64) * 64))
            if cmax <= nmax:
                # ONE copy out of the pinned buffer (it goes back to the pool), the per-image arrays are views of that copy
                det = host[:B * nmax * 7].reshape(B, nmax, 7).copy()
                nk = host[B * nmax * 7:B * nmax * 7 + B].view(np.int32).tolist()
                self._done = [det[b, :nk[b]] for b in range(B)]
                _release(self.pool_key, self.bufs)   # idle since the event; (after an exact re-launch the first attempt's set is dropped)
                self.pred = self.cand = self.both = self.counts = self.host = self.scores = self.bufs = None
                return self._done
            # an image had more candidates than the rows NMS was given (the kernels cut, never overrun): once more, exact
            with torch.cuda.device(self.cand.device):
                cand, counts = self.cand, self.counts.clone()
                self._exact(cand, counts, min(cmax, A))


def _result_buffer(B, nmax, dev):
    """rows (B, nmax, 7) | keep counts (B) | candidate counts (B): one buffer, so that a single copy brings all three."""
    return torch.empty(B * nmax * 7 + 2 * B, dtype=torch.float32, device=dev)


def _nms(cand, counts, nmax, nms_thres, image_hw, both=None, ws=None, keep_idx=None):
    B = cand.shape[0]
    dev = cand.device
    if ws is None:
        ws = torch.empty(ops.nms_workspace_bytes(B, nmax) + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    ws = ws[off:]
    if both is None:
        both = _result_buffer(B, nmax, dev)
    det = both[:B * nmax * 7].view(B, nmax, 7)
    nkeep = both[B * nmax * 7:B * nmax * 7 + B].view(torch.int32)
    if keep_idx is None:
        keep_idx = torch.empty(B, nmax, dtype=torch.int32, device=dev)
    ops.nms(cand, counts, nmax, nms_thres, image_hw, det, keep_idx, nkeep, ws)
    return det, keep_idx, nkeep, both


def nms_keep_indices(cand, counts, nms_thres):
    """cand (B, A, 7) rows [x1,y1,x2,y2,obj,cls_conf,cls_id], counts (B,) int32: per-image kept indices (score order)."""
    _require_cuda(cand, "nms_keep_indices")
    B = cand.shape[0]
    cand = cand.float().contiguous()
    nmax = max(int(counts.cpu().max()), 1) if counts.numel() else 1
    hw = torch.ones(B, 2, dtype=torch.int32, device=cand.device)
    with torch.cuda.device(cand.device):
        det, keep_idx, nkeep, _ = _nms(cand, counts.to(torch.int32).contiguous(), nmax, nms_thres, hw)
    nk = nkeep.cpu().numpy()
    ki = keep_idx.cpu().numpy()
    return [ki[b, :nk[b]].astype(np.int64) for b in range(B)]


def post_process_async(prediction, num_classes, input_shape, image_info, conf_thres=0.5, nms_thres=0.4):
    """post_process with the wait taken out: submits filter, NMS and the device->host copy of the result on the current stream
    and returns a PendingDetections; `.result()` gives post_process's return value.  A serving loop that launches the next
    batch's forward before collecting this batch's detections keeps the GPU busy through the host round trip (bench.py).

    NMS needs the largest per-image candidate count to size its rows.  The first call for a (B, A, device) waits for the
    counts; later calls launch NMS at once with an estimate from the previous call (1.25x, rounded up to 64) and fetch rows,
    keep counts and candidate counts in ONE device->host copy.  If an image turned out to have more candidates than the
    estimate, NMS is launched again with the exact maximum when the result is collected -- results do not depend on the
    estimate, only the number of host round trips does."""
    _require_cuda(prediction, "post_process")
    B, A = int(prediction.shape[0]), int(prediction.shape[1])
    if A == 0 or B == 0:
        return _Ready([None for _ in range(B)])
    pred = prediction.float().contiguous()
    hw = _image_hw(tuple((int(image_info[i][2]), int(image_info[i][1])) for i in range(B)), pred.device)
    with torch.cuda.device(pred.device):
        return PendingDetections(pred, num_classes, conf_thres, nms_thres, hw, _scores_of(prediction, pred, num_classes))


class _Ready:
    def __init__(self, value):
        self._value = value

    def result(self):
        return self._value


def post_process(prediction, num_classes, input_shape, image_info, conf_thres=0.5, nms_thres=0.4):
    """Same contract as the reference: list (len B) of float32 arrays (n_i, 7) with columns
    [top, left, bottom, right, obj_conf, class_conf, class_id] in original-image pixels; (0,7) when nothing survives;
    None entries only when there are no anchors at all.  image_info[i] = [name, image_w, image_h] (tools/demo.py:52)."""
    return post_process_async(prediction, num_classes, input_shape, image_info, conf_thres, nms_thres).result()
