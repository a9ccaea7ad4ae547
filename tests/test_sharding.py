"""N>1 path on CPU: two gloo processes each place their contiguous shard of the
reads (with the Python model of the sweep standing in for the GPU, which this
container lacks) and rank 0 concatenates; the result must equal the unsharded
oracle run.  Covers wepp_amd/sharding.py, the only multi-GPU logic there is
(the path has no collective)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import wepp_amd as w
from wepp_amd.sharding import shard_bounds

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_bounds_partition():
    for n in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out_path):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    import sweep_model as sm
    from wepp_amd.sharding import gather_results, shard_reads
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = w.generate_tree(31, 1500, genome_len=800, p_ambiguous=0.01, root_mutations=1)
    reads = g.reads(32, 41, p_substitution=0.004, p_n=0.01)
    mine = shard_reads(reads, rank, world)
    # one flatten per node: rank 0 flattens, rank 1 reads the image back (wepp_flat_save / wepp_flat_load)
    from wepp_amd.sharding import shared_flat_image
    before = w.flatten_count()
    flat = shared_flat_image(g.tree, dist, rank, f"test_{port}", directory=os.path.dirname(out_path))
    assert w.flatten_count() - before == (1 if rank == 0 else 0)
    # no room (here: no such directory) for the image: every rank flattens for itself, nobody waits for a file
    before = w.flatten_count()
    own = shared_flat_image(g.tree, dist, rank, f"test_{port}", directory=os.path.join(os.path.dirname(out_path), "missing", "dir"))
    assert w.flatten_count() - before == 1
    own.close()
    fm = sm.FlatModel(flat)

    class Local:
        pass
    loc = Local()
    res = []
    for r in range(mine.n_reads):
        p, rf, a, ms = mine.entries(r)
        S = [(int(p[i]), int(rf[i]), int(a[i]), int(ms[i])) for i in range(len(p))]
        res.append(fm.place_full(S, nchunks=2))
    loc.best_bfs_j = np.array([x["best_j"] for x in res], np.uint32)
    loc.score = np.array([x["score"] for x in res], np.int32)
    loc.num_best = np.array([x["num_best"] for x in res], np.uint32)
    loc.flags = np.array([x["has_unique"] for x in res], np.uint32)
    full = gather_results(loc, dist)
    if rank == 0:
        np.savez(out_path, **full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_placement_equals_unsharded(tmp_path, oracle):
    out = str(tmp_path / "gathered.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    g = w.generate_tree(31, 1500, genome_len=800, p_ambiguous=0.01, root_mutations=1)
    reads = g.reads(32, 41, p_substitution=0.004, p_n=0.01)
    want = oracle.OracleTree(g.tree).place_batch(reads, 2)
    assert (got["score"] == want["score"]).all()
    assert (got["best_bfs_j"] == want["best_j"]).all()
    assert (got["num_best"] == want["num_best"]).all()
    assert (got["flags"] == want["has_unique"]).all()


def _epp_worker(rank, world, port, out_path):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    import epp_model
    from wepp_amd.sharding import epp_allreduce, shard_epp_reads
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = w.generate_tree(41, 600, genome_len=800)
    reads = g.reads(42, 90, read_len=100, amplicon_len=200, amplicon_step=150, p_substitution=0.01, p_n=0.02,
                    windows=True, max_degree=4)
    mine = shard_epp_reads(reads, rank, world)
    fv = w.FlatView(g.tree)
    local = epp_model.epp_map(fv.get("epp_word"), fv.get("epp_node"), g.tree.n_nodes, mine, 800)
    full = epp_allreduce(local, mine, 800, dist)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(dict(mp=local["max_parsimony"], mult=local["multiplicity"]), gathered, dst=0)
    if rank == 0:
        np.savez(out_path, score=full["score"], counts=full["counts"], divergence=full["divergence"],
                 mp=np.concatenate([x["mp"] for x in gathered]), mult=np.concatenate([x["mult"] for x in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_epp_allreduce_equals_unsharded_oracle(tmp_path, oracle):
    """cartesian_map sharded over reads: two gloo ranks map their halves (Python model of the
    sweep standing in for the GPU), all-reduce the per-haplotype scores / read counts, and the
    result equals the oracle's unsharded run."""
    out = str(tmp_path / "epp.npz")
    port = 29500 + (os.getpid() % 2000) + 1
    mp.spawn(_epp_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    g = w.generate_tree(41, 600, genome_len=800)
    reads = g.reads(42, 90, read_len=100, amplicon_len=200, amplicon_step=150, p_substitution=0.01, p_n=0.02,
                    windows=True, max_degree=4)
    want = oracle.OracleTree(g.tree).epp_map(reads, genome_size=800)
    assert (got["mp"] == want["max_parsimony"]).all() and (got["mult"] == want["multiplicity"]).all()
    assert np.allclose(got["score"], want["score"], rtol=1e-12, atol=1e-15)
    assert (got["counts"] == want["counts"]).all()
    assert np.array_equal(got["divergence"], want["divergence"], equal_nan=True)


def _gpu_worker(rank, world, port, out_path):
    """Like _worker, with the HIP library placing the shard: each rank is its own process with its own handle on the
    box's GPU (what `bench.py --gpus N` and the driver's scaling runs do on N GPUs)."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    from wepp_amd.sharding import gather_results, shard_reads
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = w.generate_tree(33, 60_000, p_ambiguous=0.01, p_masked_node=0.002, root_mutations=1)
    reads = g.reads(34, 70_001, p_substitution=0.004, p_n=0.03, p_iupac=0.1)
    from wepp_amd.sharding import shared_flat_image
    mat = w.Mat(g.tree, device=0, flat=shared_flat_image(g.tree, dist, rank, f"test_{port}", directory=os.path.dirname(out_path)))
    res = mat.place_batch(shard_reads(reads, rank, world))
    full = gather_results(res, dist)
    mat.close()
    if rank == 0:
        np.savez(out_path, **full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_processes_on_the_gpu_equal_the_unsharded_oracle(tmp_path, oracle):
    """The N>1 path with the HIP library in BOTH processes (the CPU tests above put a Python model in its place): two
    ranks sharing ONE flat image (rank 0 flattens, rank 1 reads it back), each holding its own handle on device 0 and placing its contiguous half of 70 001
    reads; rank 0 concatenates -- equal to the oracle's unsharded run."""
    out = str(tmp_path / "gathered_gpu.npz")
    port = 29500 + (os.getpid() % 2000) + 2
    mp.spawn(_gpu_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    g = w.generate_tree(33, 60_000, p_ambiguous=0.01, p_masked_node=0.002, root_mutations=1)
    reads = g.reads(34, 70_001, p_substitution=0.004, p_n=0.03, p_iupac=0.1)
    want = oracle.OracleTree(g.tree).incremental().place_batch(reads, nthreads=os.cpu_count() or 1)
    assert (got["score"] == want["score"]).all()
    assert (got["best_bfs_j"] == want["best_j"]).all()
    assert (got["num_best"] == want["num_best"]).all()
    assert ((got["flags"] & 1) == want["has_unique"]).all()
