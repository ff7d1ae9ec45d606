"""Multi-GPU sharding of a Monte-Carlo run: one process per GPU, shots split by
global shot index, only the u64 counters are all-reduced (RCCL over xGMI when
the backend is "nccl", gloo on CPU).  Shots are independent
(examples/css_decoding.rs:14-21 carries no state between iterations except the
failure count), and the noise of a shot is a function of (seed, global shot
index) alone, so the reduced counters do not depend on the number of ranks."""
import numpy as np


def shard_range(total_shots, rank, world_size):
    """Contiguous slice [lo, hi) of the global shot range owned by `rank`."""
    lo = total_shots * rank // world_size
    hi = total_shots * (rank + 1) // world_size
    return lo, hi


def all_reduce_counters(counters, device=None):
    """Sum a (..., 8) uint64 counter array over all ranks.  One collective per
    call: pass a [points x 8] array to reduce a whole p-sweep at once."""
    import torch
    import torch.distributed as dist

    arr = np.ascontiguousarray(counters, dtype=np.uint64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return arr.copy()
    t = torch.from_numpy(arr.astype(np.int64))   # counters stay far below 2^63
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)


def run_sharded(run, total_shots, first_shot=0, rank=None, world_size=None):
    """Run `run(first_shot, n_shots) -> counters` on this rank's slice and
    all-reduce the counters."""
    import torch.distributed as dist

    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    lo, hi = shard_range(total_shots, rank, world_size)
    local = run(first_shot + lo, hi - lo)
    return all_reduce_counters(local)
