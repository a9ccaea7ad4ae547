"""Read sharding across the GPUs of one node.

Reads are independent and the MAT is immutable on this path (--no-add,
src/usher_common.cpp:649), so rank g of W places the contiguous slice
[R*g/W, R*(g+1)/W) against its own replica of the MAT and the per-rank results
are concatenated in rank order on the host.  There is no data-path collective;
torch.distributed (RCCL or gloo) is used only to gather the small result arrays.

WEPP's own placer (wepp_filter::cartesian_map, src/WEPP/initial_filter.cpp:140-239) is
different: reads still shard, but every read adds to the scores and per-bin read counts of
the haplotypes it maps to, so the per-haplotype arrays of the ranks must be SUMMED -- the one
real exchange step on these paths, an all-reduce (RCCL over xGMI on GPUs) of N doubles and
N x 50 ints (epp_allreduce below).
"""
import os

import numpy as np


def shared_flat_image(tree, dist=None, local_rank=0, tag="0", directory="/dev/shm"):
    """ONE flatten per node when the ranks are processes of their own: local rank 0 flattens `tree` and leaves the
    image in `directory` (wepp_flat_save), the other ranks of the node read it back (wepp_flat_load); the file is gone
    when this returns.  Every rank gets a FlatView to upload (Mat(tree, device, flat=...)).  The reference re-expands
    the tree per sample (src/usher_common.cpp:339)."""
    from .api import FlatView
    if dist is None or dist.get_world_size() == 1:
        return FlatView(tree)
    path = os.path.join(directory, f"wepp_flat_{tag}.bin")
    flat = None
    ok = [True]
    if local_rank == 0:
        flat = FlatView(tree)
        try:
            flat.save(path)
        except Exception:          # (no room in `directory`: every rank flattens for itself, as before)
            ok[0] = False
            try:
                os.remove(path)
            except OSError:
                pass
    # (a node's ranks agree on how they get the image; with several nodes the flag is the job's: one node short of
    # space sends every rank to its own flatten)
    flags = [None] * dist.get_world_size()
    dist.all_gather_object(flags, ok[0])
    shared = all(flags)
    if local_rank != 0:
        flat = FlatView.load(path) if shared else FlatView(tree)
    dist.barrier()
    if local_rank == 0 and shared:
        os.remove(path)
    return flat


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


def shard_epp_reads(reads, rank, world):
    """Contiguous shard of an EppReads batch."""
    from .api import EppReads
    lo, hi = shard_bounds(reads.n_reads, rank, world)
    base = reads.slice(lo, hi)
    return EppReads(base.read_off, base.read_word, reads.start[lo:hi], reads.end[lo:hi], reads.degree[lo:hi])


def epp_true_read_counts(reads, genome_size):
    """arena::build_range_trees, src/WEPP/arena.cpp:137-147: degrees per read-start bin."""
    bins = np.minimum(reads.start // (genome_size // 50), 49)
    return np.bincount(bins, weights=reads.degree, minlength=50).astype(np.int64)


def epp_divergence(counts, true_counts):
    """haplotype::dist_divergence, src/WEPP/initial_filter.cpp:224-233."""
    with np.errstate(divide="ignore", invalid="ignore"):
        prop = counts.astype(np.float64) / true_counts.astype(np.float64)[None, :]
        over = np.count_nonzero(prop > 0.5 / 100, axis=1)
        return over / np.float64(np.count_nonzero(true_counts))


def epp_allreduce(local, local_reads, genome_size, dist=None, device=None):
    """Combine the per-rank results of Mat.epp_map(shard, ..., want_counts=True) into the
    whole-batch haplotype arrays on every rank: score and mapped_read_counts are summed over
    the ranks (all-reduce; on `device` when the backend is RCCL), the divergence is recomputed
    from the summed counts.  Per-read arrays stay per shard (gather_results-style
    concatenation is the caller's choice).  `dist` = initialised torch.distributed or None."""
    true_counts = epp_true_read_counts(local_reads, genome_size)
    score, counts = np.array(local["score"], np.float64), np.array(local["counts"], np.int32)
    if dist is not None and dist.get_world_size() > 1:
        import torch
        dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
        ts = torch.from_numpy(score).to(dev)
        tc = torch.from_numpy(counts).to(dev)
        tt = torch.from_numpy(true_counts).to(dev)
        dist.all_reduce(ts)
        dist.all_reduce(tc)
        dist.all_reduce(tt)
        score, counts, true_counts = ts.cpu().numpy(), tc.cpu().numpy(), tt.cpu().numpy()
    return dict(score=score, counts=counts, divergence=epp_divergence(counts, true_counts), true_read_counts=true_counts)
