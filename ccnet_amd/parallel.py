"""Multi-GPU layer: one process per GPU, the image batch sharded by rank with replicated weights, no collective on
the forward path.  The only exchange is one all_gather of the finished detections (RCCL over xGMI
when the backend is "nccl"; "gloo" on CPU for tests).

network.net.Net is NOT shard-invariant (its Transformer attends across the image batch): sharding it
runs independent replicas on sub-batches, and results match the reference evaluated on the same sub-batches."""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous chunk [lo, hi) of `total` images owned by `rank` (first total % world ranks get one extra)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


ROWS_GUESS = 128  # rows per image carried by the first (usually only) collective


def _pack(local, bmax, k):
    """float32 [1 + bmax + bmax*k*7]: number of local images, per-image counts (-1 = None), rows zero-padded to k."""
    buf = np.zeros(1 + bmax + bmax * k * 7, np.float32)
    buf[0] = len(local)
    buf[1:1 + bmax] = -1.0
    rows = buf[1 + bmax:].reshape(bmax, k, 7)
    for i, d in enumerate(local):
        if d is not None:
            n = int(d.shape[0])
            buf[1 + i] = n
            if n:
                rows[i, :min(n, k)] = np.asarray(d, dtype=np.float32)[:k]
    return buf


def all_gather_detections(local, group=None, device=None, max_local=None):
    """local: list (this rank's images, in order) of (n_i, 7) float32 arrays (None allowed).  Returns the list for the
    whole batch in rank order on every rank.

    One collective in the common case: every rank sends [n_images, counts, rows padded to ROWS_GUESS per image] (counts
    travel as float32, exact below 2^24); only when some image anywhere has more rows is a second all_gather made with the
    true maximum -- every rank sees the same counts, so all take the same branch.  `max_local` = the largest number of
    images any rank holds (`shard_range` gives it: ceil(total / world)); without it one more small collective finds it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(local)
    world = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")

    def gather(host):
        t = torch.from_numpy(host).to(device)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        return [p_.cpu().numpy() for p_ in parts]

    if max_local is None:
        bmax = max(int(g[0]) for g in gather(np.array([len(local)], np.float32)))
    else:
        bmax = int(max_local)
        if len(local) > bmax:
            raise ValueError("all_gather_detections: %d local images but max_local=%d" % (len(local), bmax))
    bmax = max(bmax, 1)
    k = ROWS_GUESS
    got = gather(_pack(local, bmax, k))
    kmax = max(int(g[1:1 + bmax].max()) for g in got)
    if kmax > k:
        k = kmax
        got = gather(_pack(local, bmax, k))
    out = []
    for g in got:
        n = int(g[0])
        counts = g[1:1 + bmax]
        rows = g[1 + bmax:].reshape(bmax, k, 7)
        for i in range(n):
            c = int(counts[i])
            out.append(None if c < 0 else rows[i, :c].copy())
    return out


# ------------------------------------------------------------------------------------------------------------------
# Device-side exchange: filter + NMS on the local shard, ONE all_gather of the device result buffer (rows, keep counts and
# candidate counts as int32 bit patterns) BEFORE anything is copied to the host, then one device->host copy of the gathered
# buffer.  With the nccl backend (= RCCL over xGMI) the collective runs on the process group's own stream behind the NMS
# kernels, so no host round trip sits between the local NMS and the exchange (off the forward's critical path).
_rows_guess = {}
_buffers = {}   # (key, nmax, stream) -> the call's device buffers, kept between calls (a step allocates nothing in steady state)


def _exchange_buffers(key, nmax, B, A, bmax, world, dev, on_device):
    from . import ops
    stream = torch.cuda.current_stream(dev).cuda_stream
    k = (key, nmax, stream)
    got = _buffers.get(k)
    if got is None:
        if len(_buffers) > 16:
            _buffers.clear()
        n_rows = bmax * nmax * 7
        ws = torch.empty(ops.nms_workspace_bytes(max(B, 1), nmax) + 256, dtype=torch.uint8, device=dev)
        got = {"cand": torch.empty(max(B, 1), A, 7, dtype=torch.float32, device=dev),
               # rows (bmax, nmax, 7) | keep counts (bmax) | candidate counts (bmax); zeroed once: the rows and counts of the
               # padding images (uneven shards) are never written afterwards, everything else is rewritten by every call
               "both": torch.zeros(n_rows + 2 * bmax, dtype=torch.float32, device=dev),
               "gathered": torch.empty(world * (n_rows + 2 * bmax), dtype=torch.float32, device=dev) if on_device else None,
               "keep_idx": torch.empty(max(B, 1), nmax, dtype=torch.int32, device=dev),
               "ws": ws[(-ws.data_ptr()) % 256:]}
        _buffers[k] = got
    return got


def sharded_post_process(prediction, num_classes, input_shape, image_info, conf_thres=0.5, nms_thres=0.4, group=None,
                         total_images=None):
    """post_process (utils/bbox.py:82-159) of this rank's shard + the detections all_gather.  prediction: (B_local, A, 5+nc)
    CUDA tensor of the local images, image_info: their [name, iw, ih] rows.  Returns the list for the WHOLE batch in rank
    order on every rank (same row contents as ccnet_amd.utils.bbox.post_process on the unsharded batch for the
    shard-invariant nets).  total_images: size of the whole batch when the shards are uneven (shard_range); default: every
    rank holds B_local images.  Without an initialised process group this is plain post_process; with one, the exchange runs
    even at world size 1 (a one-rank RCCL group executes the same all_gather_into_tensor, tests/test_parallel_gpu.py).
    The device buffers are kept between calls per (shard layout, row estimate, stream): calls on ONE stream are serialised by
    the device->host copy at the end, callers on different streams get their own buffers."""
    from . import ops
    from .utils import bbox
    if not (dist.is_available() and dist.is_initialized()):
        return bbox.post_process(prediction, num_classes, input_shape, image_info, conf_thres, nms_thres)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B, A = int(prediction.shape[0]), int(prediction.shape[1])
    if total_images is None:
        locals_ = [B] * world
    else:
        locals_ = [hi - lo for lo, hi in (shard_range(total_images, r, world) for r in range(world))]
        if locals_[rank] != B:
            raise ValueError("sharded_post_process: rank %d holds %d images, shard_range says %d" % (rank, B, locals_[rank]))
    bmax = max(max(locals_), 1)
    dev = prediction.device
    on_device = dist.get_backend(group) == "nccl"
    pred = prediction.float().contiguous()
    hw = bbox._image_hw(tuple((int(image_info[i][2]), int(image_info[i][1])) for i in range(B)), dev) if B else None
    key = (tuple(locals_), A, str(dev))
    nmax = min(_rows_guess.get(key, ROWS_GUESS), max(A, 1))
    with torch.cuda.device(dev):
        first = True
        while True:
            buf = _exchange_buffers(key, nmax, B, A, bmax, world, dev, on_device)
            both, n_rows = buf["both"], bmax * nmax * 7
            cnt = both[n_rows + bmax:].view(torch.int32)   # candidate counts, int32 bit patterns in the fp32 buffer
            if B and A:
                if first:
                    ops.filter_candidates(pred, num_classes, conf_thres, buf["cand"], cnt[:B], bbox._scores_of(prediction, pred, num_classes))
                    cand, first = buf["cand"], False
                else:
                    cnt[:B].copy_(cnt0)
                ops.nms(cand, cnt[:B], nmax, nms_thres, hw, both[:B * nmax * 7].view(B, nmax, 7), buf["keep_idx"],
                        both[n_rows:n_rows + B].view(torch.int32), buf["ws"])
            if on_device:
                dist.all_gather_into_tensor(buf["gathered"], both, group=group)  # RCCL, ordered behind the NMS kernels
                host = buf["gathered"].cpu().numpy().reshape(world, -1)
            else:  # gloo (CPU tests, one-GPU rehearsal): the same buffer travels through the host
                mine = both.cpu()
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine, group=group)
                host = np.stack([p_.numpy() for p_ in parts])
            cmax = int(host[:, n_rows + bmax:].view(np.int32).max())
            # every rank sees every count, so all ranks size the next call alike and take the same branch below
            _rows_guess[key] = min(max(A, 1), max(64, -(-(cmax + cmax // 4) // 64) * 64))
            if len(_rows_guess) > 64:
                _rows_guess.clear()
            if cmax <= nmax:
                break
            cnt0 = cnt[:B].clone() if B and A else None   # the retry's buffers are other buffers: carry the counts over
            nmax = min(cmax, A)
    out = []
    for r in range(world):
        rows = host[r, :n_rows].reshape(bmax, nmax, 7)
        nk = host[r, n_rows:n_rows + bmax].view(np.int32).tolist()
        for i in range(locals_[r]):
            out.append(rows[i, :nk[i]] if A else None)   # views: `host` is this call's own array
    return out
