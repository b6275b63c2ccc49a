"""Multi-GPU partition of the solver: images (and their hypotheses) are independent, so each rank
(one process per GPU, torch.distributed over RCCL) solves a contiguous block of the batch and the
[B_local, 7] results (azimuth, elevation, theta, distance, px, py, score) are collected with one
all-gather.  The only other coupling is a scalar: the loss normaliser 1 / (B * H * W) uses the size of
the whole batch (reference nemo/models/batch_solve_pose.py:31-34,332), which every rank knows from the
partition itself - no collective on the data path.  The reference has no multi-process code
(nemo/utils/distributed_utils.py is never called by its models), so this is the build's own layer.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n, rank=None, world_size=None):
    """Contiguous block [lo, hi) of n items owned by `rank`; the first n % world ranks get one extra."""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local, n_total=None):
    """Concatenate per-rank [n_r, k] tensors in rank order (shards may differ by one row)."""
    rank, ws = world()
    if ws == 1:
        return local
    n_local = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    counts = [torch.zeros_like(n_local) for _ in range(ws)]
    dist.all_gather(counts, n_local)
    counts = [int(c) for c in counts]
    pad = max(counts)
    buf = torch.zeros(pad, *local.shape[1:], device=local.device, dtype=local.dtype)
    buf[: local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(out, buf)
    res = torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)
    if n_total is not None and res.shape[0] != n_total:
        raise RuntimeError(f"gathered {res.shape[0]} rows, expected {n_total}")
    return res
