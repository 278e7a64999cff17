"""ccnet_nms_noalloc (include/ccnet_hip.h): the NMS entry point whose workspace is the library's static device array, for
callers that cannot allocate (HIP graph capture, tight serving loops).  Called through the C ABI and compared with ccnet_nms
(caller workspace) bit for bit -- rows, keep indices, keep counts -- on two calls in a row, the second with a smaller input;
sizes beyond the static maximum are rejected with a clear error."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cands(B, n, seed):
    rng = np.random.RandomState(seed)
    cand = np.zeros((B, n, 7), np.float32)
    c = rng.uniform(0, 1, (B, n, 2))
    wh = np.exp(rng.normal(math.log(0.08), 0.5, (B, n, 2)))
    cand[..., 0:2] = c - wh / 2
    cand[..., 2:4] = c + wh / 2
    cand[..., 4] = rng.uniform(0, 1, (B, n))
    cand[..., 5] = rng.uniform(0.5, 1, (B, n))
    cand[..., 6] = rng.randint(0, 6, (B, n))
    counts = rng.randint(n // 2, n + 1, B).astype(np.int32)
    counts[0] = n
    return torch.from_numpy(cand).cuda(), torch.from_numpy(counts).cuda()


def _run(lib, cand, counts, nmax, thr, noalloc):
    from ccnet_amd import _lib
    B, A, _ = cand.shape
    hw = torch.tensor([[480 + 3 * b, 640 - 5 * b] for b in range(B)], dtype=torch.int32, device="cuda")
    det = torch.full((B, nmax, 7), -1.0, dtype=torch.float32, device="cuda")
    keep = torch.full((B, nmax), -1, dtype=torch.int32, device="cuda")
    nk = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    st = _lib.stream_ptr()
    if noalloc:
        _lib.check(lib.ccnet_nms_noalloc(cand.data_ptr(), counts.data_ptr(), B, A, nmax, thr, hw.data_ptr(), det.data_ptr(), keep.data_ptr(),
                                         nk.data_ptr(), st))
    else:
        ws = torch.empty(int(lib.ccnet_nms_workspace_bytes(B, nmax)) + 256, dtype=torch.uint8, device="cuda")
        ws = ws[(-ws.data_ptr()) % 256:]
        _lib.check(lib.ccnet_nms(cand.data_ptr(), counts.data_ptr(), B, A, nmax, thr, hw.data_ptr(), det.data_ptr(), keep.data_ptr(), nk.data_ptr(),
                                 ws.data_ptr(), st))
    torch.cuda.synchronize()
    nk = nk.cpu().numpy()
    det, keep = det.cpu().numpy(), keep.cpu().numpy()
    return [(det[b, :nk[b]].copy(), keep[b, :nk[b]].copy()) for b in range(B)]


def test_nms_noalloc_matches_nms_on_two_calls_in_a_row():
    from ccnet_amd import _lib
    lib = _lib.lib()
    _lib.check(lib.ccnet_nms_noalloc_init())
    assert int(lib.ccnet_nms_noalloc_max_bytes()) == 32 << 20
    # first call: 8 images x 2000 candidates (tiled kernels, ~5 MB of the scratch); second: smaller in both B and n, so stale
    # words of the first call sit right behind (and between) the regions the second one uses
    for B, n, seed in ((8, 2000, 1), (3, 700, 2), (2, 100, 3)):
        cand, counts = _cands(B, n, seed)
        got = _run(lib, cand, counts, n, 0.6, True)
        want = _run(lib, cand, counts, n, 0.6, False)
        assert sum(len(k) for _, k in want) > B
        for b in range(B):
            assert np.array_equal(got[b][1], want[b][1]), "keep indices differ, image %d (B=%d n=%d)" % (b, B, n)
            assert np.array_equal(got[b][0], want[b][0]), "rows differ, image %d (B=%d n=%d)" % (b, B, n)


def test_nms_noalloc_rejects_sizes_beyond_its_scratch():
    from ccnet_amd import _lib
    lib = _lib.lib()
    B, n = 16, 20000  # 16 x (8 W^2 + 72 W) words of bitmask panel, W = 313: 100 MiB
    assert int(lib.ccnet_nms_workspace_bytes(B, n)) > int(lib.ccnet_nms_noalloc_max_bytes())
    cand = torch.zeros(B, n, 7, device="cuda")
    counts = torch.zeros(B, dtype=torch.int32, device="cuda")
    hw = torch.ones(B, 2, dtype=torch.int32, device="cuda")
    det = torch.zeros(B, n, 7, device="cuda")
    keep = torch.zeros(B, n, dtype=torch.int32, device="cuda")
    nk = torch.zeros(B, dtype=torch.int32, device="cuda")
    rc = lib.ccnet_nms_noalloc(cand.data_ptr(), counts.data_ptr(), B, n, n, 0.6, hw.data_ptr(), det.data_ptr(), keep.data_ptr(), nk.data_ptr(),
                               _lib.stream_ptr())
    assert rc == -1 and b"static scratch holds" in lib.ccnet_last_error()
