"""Read sharding across the GPUs of one node.

Reads are independent and the MAT is immutable on this path (--no-add,
src/usher_common.cpp:649), so rank g of W places the contiguous slice
[R*g/W, R*(g+1)/W) against its own replica of the MAT and the per-rank results
are concatenated in rank order on the host.  There is no data-path collective;
torch.distributed (RCCL or gloo) is used only to gather the small result arrays.
"""
import numpy as np


def shard_bounds(n_reads, rank, world):
    """Contiguous shard [lo, hi) of rank `rank` out of `world`."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    lo = (n_reads * rank) // world
    hi = (n_reads * (rank + 1)) // world
    return lo, hi


def shard_reads(reads, rank, world):
    lo, hi = shard_bounds(reads.n_reads, rank, world)
    return reads.slice(lo, hi)


RESULT_FIELDS = ("best_bfs_j", "score", "num_best", "flags")


def gather_results(local, dist=None, dst=0):
    """Concatenate per-rank PlacementResult-like objects in rank order on `dst`.
    `dist` is torch.distributed (initialised) or None for a single process.
    Returns a dict of numpy arrays on `dst`, None elsewhere."""
    payload = {f: np.asarray(getattr(local, f)) for f in RESULT_FIELDS}
    if dist is None or dist.get_world_size() == 1:
        return payload
    world = dist.get_world_size()
    gathered = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(payload, gathered, dst=dst)
    if dist.get_rank() != dst:
        return None
    return {f: np.concatenate([g[f] for g in gathered]) for f in RESULT_FIELDS}
