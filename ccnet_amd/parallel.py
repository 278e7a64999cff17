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
