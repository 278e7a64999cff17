"""Multi-GPU layer: one process per GPU, the image batch sharded by rank with replicated weights, no collective on
the forward path.  The only exchange is an all_gather of the finished detections (RCCL over xGMI
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


def all_gather_detections(local, group=None, device=None):
    """local: list (this rank's images, in order) of (n_i, 7) float32 arrays (None allowed).  Returns the list for the
    whole batch in rank order on every rank.  Two collectives: counts (int32) then zero-padded rows."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(local)
    world = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    n_local = torch.tensor([len(local)], dtype=torch.int32, device=device)
    n_all = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(n_all, n_local, group=group)
    n_all = [int(t.item()) for t in n_all]
    bmax = max(n_all) if n_all else 0
    counts = torch.full((max(bmax, 1),), -1, dtype=torch.int32)
    for i, d in enumerate(local):
        counts[i] = -1 if d is None else d.shape[0]
    counts = counts.to(device)
    c_all = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(c_all, counts, group=group)
    c_all = [t.cpu().numpy() for t in c_all]
    kmax = max(1, max(int(c.max()) for c in c_all))
    rows = torch.zeros(max(bmax, 1), kmax, 7, dtype=torch.float32)
    for i, d in enumerate(local):
        if d is not None and d.shape[0]:
            rows[i, :d.shape[0]] = torch.from_numpy(np.ascontiguousarray(d, dtype=np.float32))
    rows = rows.to(device)
    r_all = [torch.zeros_like(rows) for _ in range(world)]
    dist.all_gather(r_all, rows, group=group)
    out = []
    for r in range(world):
        rr = r_all[r].cpu().numpy()
        for i in range(n_all[r]):
            c = int(c_all[r][i])
            out.append(None if c < 0 else rr[i, :c].copy())
    return out
