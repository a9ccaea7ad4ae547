#!/usr/bin/env python3
"""bench.py -- reads placed / s on a SARS-CoV-2-like MAT (~16M nodes).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (for N > 1 launch with torch.distributed.run; ranks read
RANK / LOCAL_RANK / WORLD_SIZE).  The MAT is replicated on every GPU, each rank
places its own shard of reads (weak scaling: 1e6 reads per GPU), there is no
data-path collective; torch.distributed is used only for the timing barrier
and the max-over-ranks reduction.  A "step" = one pass of the hot path
(wepp_place_batch_device: both passes of the reference's per-sample loop,
src/usher_common.cpp:386-446) over one batch of reads already resident in HBM.

Prints ONE JSON line on rank 0 (see DESIGN.md section 7 for how each field is
obtained).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md: L2 (per XCD 4 MiB) ~34.5 TB/s aggregate
N_CUS = 256            # MI355X_MICROARCH.md chip-level parameters
MAX_CLOCK_GHZ = 2.4    # max clock; a CU issues at most one vector instruction per cycle (4 SIMDs, one per 4 cycles)
VALU_ISSUE_PEAK = N_CUS * MAX_CLOCK_GHZ   # G wave-instructions / s
KERNEL_SOURCES = ("place_dev.hpp", "route_kernels.hip", "sweep_kernels.hip", "walk_kernels.hip", "wave_kernels.hip", "seed_kernels.hip", "device_mat.hpp",
                  "flatmat.hpp", "flatmat.cpp", "capi.cpp", "sort_reads.hip")


def kernel_hash():
    """Identifies the build of the placement kernels a profile belongs to: bench.py refuses to quote
    counters (roofline.traffic, instruction counts) measured on other code."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "wepp_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    # workload knobs (defaults = BASELINE.json configs[2], the config the metric is quoted on)
    ap.add_argument("--nodes", type=int, default=16_000_000)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--tile", type=int, default=64, help="reads sharing one sweep of the event stream (T)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0,
                    help="target wall time of the CPU baseline sample (0 disables it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=2,
                    help="extra untimed-for-value steps with work skipping off (whole-tree streaming roofline)")
    ap.add_argument("--no-crowns", action="store_true",
                    help="disable work skipping: every read sweeps the whole-tree stream (roofline run)")
    ap.add_argument("--no-walk", action="store_true",
                    help="place every read by a sweep of its stream (no per-read walks of the position index)")
    ap.add_argument("--pcie-steps", type=int, default=5,
                    help="timed steps of the host-buffer leg (wepp_place_batch: H2D of the reads and D2H of the "
                         "results inside); 0 disables it")
    ap.add_argument("--no-sensitivity", action="store_true",
                    help="skip the sensitivity ladder (reads/s against the N rate and the entries per read)")
    ap.add_argument("--batches", type=int, default=8,
                    help="distinct read batches (all resident in HBM) the timed loop rotates over")
    ap.add_argument("--p-n", type=float, default=None, help="per-base N rate of the synthetic reads (default 0.005 "
                    "for 150 bp reads, SURVEY 8(d) config 2/3; 0.02 for long reads, config 5)")
    ap.add_argument("--workload", choices=("reads", "genome"), default="reads",
                    help="genome: the timed batches are whole-genome samples (what read_vcf makes of consensus genomes; "
                         "--reads of them per step) instead of amplicon reads")
    ap.add_argument("--no-legs", action="store_true", help="skip the legs of the other configs (configs[3] shard, configs[4] "
                    "long reads, whole-genome samples) and the tree-shape ladder")
    ap.add_argument("--ladder-nodes", type=int, default=4_000_000, help="nodes of the tree-shape ladder's trees (0 disables it)")
    return ap.parse_args()


def usable_cores():
    """Host threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:  # cgroup v2
            q, p = fh.read().split()
            if q != "max":
                n = min(n, max(1, int(float(q) / float(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(tree, reads, gpu_res, target_s, ot=None, inc=None):
    """Oracle (CPU restatement of the reference loop) on a bounded sample of the
    same reads; the NODE range is split over all usable host threads, which is
    how the reference parallelises a sample (tbb::parallel_for,
    src/usher_common.cpp:386)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bridge
    cores = usable_cores()
    if ot is None:
        ot = oracle_bridge.OracleTree(tree)
    t0 = time.perf_counter()
    ot.place_batch(reads.slice(0, 1), nthreads=cores, node_parallel=True)   # probe: sizes the sample
    t1 = time.perf_counter() - t0
    n = int(min(reads.n_reads, 4096, max(1, target_s / max(t1, 1e-6))))
    sample = reads.slice(0, n)
    t0 = time.perf_counter()
    want = ot.place_batch(sample, nthreads=cores, node_parallel=True)
    dt = time.perf_counter() - t0
    ok = bool((want["score"] == gpu_res["score"][:n]).all() and (want["best_j"] == gpu_res["best"][:n]).all()
              and (want["num_best"] == gpu_res["num_best"][:n]).all()
              and (want["has_unique"] == (gpu_res["flags"][:n] & 1)).all())
    # SURVEY 8(d) CPU baseline (ii): the incremental restatement (oracle/incremental_oracle.c: one tree walk per group
    # of reads instead of one evaluation per (read, node); proven equal to the faithful one by tests/test_incremental.py)
    # on a larger sample, reads split over the threads -- what a CPU can do once the per-node loop of the reference is
    # given up, a fairer yardstick for a throughput claim than the faithful loop
    if inc is None:
        inc = ot.incremental()
    n2 = int(min(reads.n_reads, 4096))
    t0 = time.perf_counter()
    want2 = inc.place_batch(reads.slice(0, n2), nthreads=cores)
    dt2 = time.perf_counter() - t0
    ok2 = bool((want2["score"] == gpu_res["score"][:n2]).all() and (want2["best_j"] == gpu_res["best"][:n2]).all()
               and (want2["num_best"] == gpu_res["num_best"][:n2]).all()
               and (want2["has_unique"] == (gpu_res["flags"][:n2] & 1)).all())
    inc.close()
    ot.close()
    return {
        "incremental": {"value": n2 / dt2, "unit": "reads/s", "cores": cores,
                        "what": "oracle/incremental_oracle.c, the restatement that walks the tree once per group of reads "
                                "(SURVEY 8(d) CPU baseline (ii)); test infrastructure like the faithful one",
                        "sample": f"first {n2} reads of the step batch, reads split over {cores} threads, {dt2:.1f} s",
                        "sample_matches_gpu": ok2},
        "value": n / dt,
        "unit": "reads/s",
        "cores": cores,
        "kind": "port",
        "kind_note": "the oracle's C restatement of mapper2_body / the usher_common loop (oracle/mapper2_oracle.c), NOT the "
                     "reference's TBB build: the reference needs TBB / Boost / protobuf headers this image lacks (DESIGN.md 6)",
        "sample": f"first {n} reads of the step batch on the same MAT, node range split over {cores} threads "
                  f"(like tbb::parallel_for over nodes), {dt:.1f} s; probe read {t1:.2f} s",
        "sample_matches_gpu": ok,
    }


def pmc_profile(mode, reads_per_step):
    """Counters per step of the placement kernels from the committed rocprofv3 --pmc profile of THIS workload
    (`mode`: short_reads / whole_tree / long_reads), THIS number of reads per step and THIS build of the kernels
    (profiles/pmc_counters.json, written by tools/summarize_profile.py from tools/profile.sh runs; key
    "<mode>:<reads per step>", kernel hash inside).  None when no such profile is committed: bench.py never prices
    one workload with another's counters."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_counters.json")) as fh:
            t = json.load(fh)[f"{mode}:{reads_per_step}"]
    except (OSError, KeyError, ValueError):
        return None
    if t.get("kernel_hash") != kernel_hash():
        return None
    return t


def sweep_roofline(mode, reads_per_step, sweep_ms, alg_bytes, passes, n_launch):
    """The roofline object of one measurement (contract shape: bound / achieved / peak / unit / frac / traffic) and the
    other utilisation views.
    achieved = ALGORITHMIC bytes per step -- what the kernels ask memory for, counted by the library: a sweep reads
    its stream once per 64-read tile (SURVEY 8(d) B_pass), a walk's lanes count their own requests (a 32-byte index
    entry per node event, sparse-table bytes, range-query aggregates, list heads, read words, results: DESIGN.md 4.2)
    -- over the kernel time of a step measured live with HIP events on the launch stream.
    traffic = HBM-side bytes per step from the committed rocprofv3 --pmc profile of the same workload and build:
    (FETCH_SIZE + WRITE_SIZE) x 1024 for the walks, whose 32-byte gathers the counter tallies as they are (cross-check:
    TCC_MISS x 64 B agrees); MI355X_MICROARCH.md's doubling of FETCH_SIZE applies to wide coalesced streams (the
    whole-tree sweep) and is reported beside it as traffic_x2.  Infinity-Cache hits are counted by both."""
    prof = pmc_profile(mode, reads_per_step)
    secs = sweep_ms * 1e-3
    alg = alg_bytes / secs / 1e9
    roof = {"bound": "hbm", "achieved": alg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / HBM_PEAK_GBS,
            "traffic": None,
            "kernel": "placement kernels of one step (k_walk plain + chunked, k_sweep, k_finalize*), concurrent on side streams",
            "algorithmic_bytes_per_step": alg_bytes, "passes_per_step": passes, "kernel_ms_per_step": sweep_ms,
            "steps_timed": n_launch,
            "achieved_note": "algorithmic bytes counted by the kernels' own lanes / tiles over the live kernel time; the index "
                             "and the streams are shared between reads, so part of it is served by L2 / Infinity Cache"}
    cands = []
    if prof is None:
        roof["provenance"] = (f"no rocprofv3 --pmc profile of {mode}:{reads_per_step} for kernel hash {kernel_hash()} is committed "
                              "(profiles/pmc_counters.json, tools/profile.sh + tools/summarize_profile.py): traffic null")
        return roof, cands
    roof["provenance"] = (f"traffic and the other counters per step from {prof['profile']} (rocprofv3 --pmc passes of this "
                          f"workload, {reads_per_step} reads per step, kernel hash {prof['kernel_hash']}; its kernel time was "
                          f"{prof.get('kernel_ms_per_step_trace')} ms per step); achieved and kernel_ms measured live")
    raw, x2 = prof.get("raw_bytes_per_step"), prof.get("traffic_bytes_per_step")
    streaming = mode == "whole_tree"
    if raw is not None:
        t = x2 if streaming else raw
        roof.update({"traffic": t, "traffic_gbs": t / secs / 1e9, "traffic_frac": t / secs / 1e9 / HBM_PEAK_GBS,
                     "traffic_raw": raw, "traffic_x2": x2, "tcc_miss_bytes": prof.get("tcc_miss_bytes_per_step"),
                     "l2_hit_rate": prof.get("l2_hit_rate"),
                     "traffic_note": ("(2*FETCH_SIZE + WRITE_SIZE)*1024: wide coalesced stream" if streaming else
                                      "(FETCH_SIZE + WRITE_SIZE)*1024: 32-byte gathers are tallied as they are (TCC_MISS x 64 B "
                                      "agrees); traffic_x2 is the figure with MI355X_MICROARCH.md's doubling") +
                                     "; Infinity-Cache hits are counted, true HBM traffic is lower"})
        cands.append({"bound": "hbm_traffic", "achieved": roof["traffic_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": roof["traffic_frac"]})
    if prof.get("valu_insts_per_step") is not None:
        a = prof["valu_insts_per_step"] / secs / 1e9
        cands.append({"bound": "valu_issue", "achieved": a, "peak": VALU_ISSUE_PEAK, "unit": "G wave-instructions/s",
                      "frac": a / VALU_ISSUE_PEAK, "valu_insts_per_step": prof["valu_insts_per_step"],
                      "salu_insts_per_step": prof.get("salu_insts_per_step")})
    if prof.get("ta_busy_cycles_per_step") is not None:
        a = prof["ta_busy_cycles_per_step"] / secs / 1e9
        cands.append({"bound": "vmem_address", "achieved": a, "peak": VALU_ISSUE_PEAK, "unit": "G busy cycles/s over the 256 address units",
                      "frac": a / VALU_ISSUE_PEAK, "vmem_read_insts_per_step": prof.get("vmem_rd_insts_per_step"),
                      "tcp_cache_accesses_per_step": prof.get("tcp_cache_accesses_per_step")})
    if prof.get("sq_wait_any_frac") is not None:
        roof["sq_wait_any_over_wave_cycles"] = prof["sq_wait_any_frac"]
    return roof, cands


class DeviceBatch:
    """A batch of reads resident in HBM plus its result buffers."""

    def __init__(self, torch, reads, dev):
        self.reads = reads
        self.R = reads.n_reads
        self.nw = int(reads.read_off[-1])
        self.d_off = torch.from_numpy(reads.read_off.astype(np.int32)).to(dev)
        self.d_word = torch.from_numpy((reads.read_word if self.nw else np.zeros(1, np.uint32)).astype(np.int32)).to(dev)
        self.out = [torch.zeros(self.R, dtype=torch.int32, device=dev) for _ in range(4)]   # best, score, num_best, flags

    def place(self, mat, stream):
        mat.place_batch_device(self.d_off.data_ptr(), self.d_word.data_ptr(), self.R, self.nw, self.out[0].data_ptr(),
                               self.out[1].data_ptr(), self.out[2].data_ptr(), self.out[3].data_ptr(), stream)


def truncate_reads(w, reads, k):
    """The reads of `reads` that list at least k positions, cut to their first k entries."""
    cnt = np.diff(reads.read_off.astype(np.int64))
    sel = np.nonzero(cnt >= k)[0]
    if k == 0:
        return w.Reads(np.zeros(reads.n_reads + 1, np.uint32), np.zeros(0, np.uint32))
    idx = (reads.read_off[sel].astype(np.int64)[:, None] + np.arange(k)[None, :]).ravel()
    return w.Reads((np.arange(len(sel) + 1) * k).astype(np.uint32), reads.read_word[idx])


GENOME_LEN = 29903
LADDER = (   # tree shapes of the ladder (wepp_place.h: wepp_gen_tree_params): what the default tree is a best case of
    ("default shape", {}),
    ("paths ~3x longer (parent = the deeper of 2 draws)", {"depth_choices": 2}),
    ("paths ~5x longer (deepest of 3 draws)", {"depth_choices": 3}),
    ("10 % back-mutations", {"p_back_mutation": 0.10}),
    ("star-like: half of the nodes hang off ~1000 hubs (polytomies of ~2000 children)", {"p_hub": 0.5}),
    ("deep and bushy: deepest of 3 draws, 30 % on hubs", {"depth_choices": 3, "p_hub": 0.3}),
)


def genome_samples(g, seed, n, p_sub=0.001, p_n=0.0005):
    """Whole-genome samples: a leaf's genotype over the whole genome with substitution errors and Ns -- what read_vcf
    (src/mutation_annotated_tree.cpp:2033-2130) makes of a consensus genome: ~20 true + ~30 erroneous alleles + ~15 Ns."""
    return g.reads(seed, n, read_len=GENOME_LEN, amplicon_len=GENOME_LEN, amplicon_step=GENOME_LEN, p_substitution=p_sub, p_n=p_n)


def long_reads_batch(g, seed, n):
    return g.reads(seed, n, read_len=1200, amplicon_len=1200, amplicon_step=1020, p_substitution=0.03, p_n=0.02)


def matches(want, out, n):
    """the checker's structured results against the first n entries of a DeviceBatch's output tensors"""
    return bool((want["score"] == out[1][:n].cpu().numpy()).all() and (want["best_j"] == out[0][:n].cpu().numpy().view(np.uint32)).all()
                and (want["num_best"] == out[2][:n].cpu().numpy().view(np.uint32)).all()
                and (want["has_unique"] == (out[3][:n].cpu().numpy().view(np.uint32) & 1)).all())


def run_leg(torch, mat, dev, stream, label, mode, host_batches, steps, pcie_steps, checker=None, n_check=256):
    """One workload on the resident MAT: device-resident rate over `steps` placements rotating over the batches,
    PCIe-inclusive rate (wepp_place_batch from host buffers), the contract-shaped roofline of its placement kernels,
    and the first reads of batch 0 against the incremental CPU checker."""
    batches = [DeviceBatch(torch, rd, dev) for rd in host_batches]
    for b in batches:                     # set-up: the handle's grow-only workspaces reach their size
        b.place(mat, stream)
    torch.cuda.synchronize()
    mat.timing_reset()
    t0 = time.perf_counter()
    for i in range(steps):
        batches[i % len(batches)].place(mat, stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    sweep_ms, n_launch, passes, alg_bytes = mat.last_timing()
    seeds = mat.last_seeds(detail=True)
    batches[0].place(mat, stream)
    torch.cuda.synchronize()
    R = batches[0].R
    pcls, _ = mat.last_plans(R)
    roof, cands = sweep_roofline(mode, R, sweep_ms, alg_bytes, passes, n_launch)
    out = {"leg": label, "reads_per_step": R, "mean_entries": batches[0].nw / max(1, R), "distinct_batches": len(batches), "steps": steps,
           "reads_per_s": R / dt, "ms_per_step": dt * 1e3,
           "reads_by_plan_class": {PLAN_NAMES[c]: int(k) for c, k in enumerate(np.bincount(pcls, minlength=7)) if k},
           "roofline": roof, "roofline_candidates": cands}
    if seeds[0]:
        out["seeded"] = {"samples_per_step": seeds[0] // steps, "chunks_evaluated_per_sample": seeds[1] / seeds[0],
                         "chunks_of_the_tree": seeds[2] // seeds[0], "most_chunks_one_sample": seeds[3],
                         "samples_by_chunks_le_1_4_16_64_256_1024_4096_more": [int(x) // steps for x in seeds[4]]}
    if pcie_steps:
        hres = mat.place_batch(host_batches[0])
        t0 = time.perf_counter()
        for i in range(pcie_steps):
            hres = mat.place_batch(host_batches[(i + 1) % len(host_batches)], out=hres)
        dtp = (time.perf_counter() - t0) / pcie_steps
        out["pcie_inclusive_reads_per_s"] = R / dtp
        out["pcie_inclusive_ms_per_step"] = dtp * 1e3
    if checker is not None:
        n = min(n_check, R)
        t0 = time.perf_counter()
        want = checker.place_batch(host_batches[0].slice(0, n), nthreads=usable_cores())
        out["sample_matches_gpu"] = matches(want, batches[0].out, n)
        out["sample"] = f"first {n} reads of the leg's first batch against oracle/incremental_oracle.c, {time.perf_counter() - t0:.1f} s"
    return out


PLAN_NAMES = ("walk8", "walk16", "sweep", "walkc8", "walkc16", "window", "seed")


def ladder_prepare(w, nodes, label, kw, want_checker):
    """Host side of one ladder entry (runs in a background thread: the ctypes calls release the GIL): the tree, its
    shape, its flat image, the two batches, and the checker's results for the first reads of each."""
    t0 = time.perf_counter()
    g = w.generate_tree(21, nodes, **kw)
    shape = g.shape()
    short = g.reads(22, 1_000_000, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.001, p_n=0.005)
    longr = long_reads_batch(g, 24, 50_000)
    t1 = time.perf_counter()
    flat = w.FlatView(g.tree)
    t_flat = time.perf_counter() - t1
    want = None
    if want_checker:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_bridge
        ot = oracle_bridge.OracleTree(g.tree)
        inc = ot.incremental()
        want = (inc.place_batch(short.slice(0, 256), nthreads=4), inc.place_batch(longr.slice(0, 64), nthreads=4))
        inc.close()
        ot.close()
    return {"label": label, "params": kw, "g": g, "shape": shape, "short": short, "long": longr, "flat": flat,
            "flatten_s": t_flat, "host_s": time.perf_counter() - t0, "want": want}


def ladder_measure(torch, w, dev, stream, prep):
    """Device side of one ladder entry: upload, the default batch and the long-read batch, crown / candidate sizes."""
    mat = w.Mat(prep["g"].tree, device=dev.index, flat=prep["flat"])
    prep["flat"].close()
    st = mat.stats
    out = {"tree": prep["label"], "generator": prep["params"], "shape": prep["shape"], "flatten_s": round(prep["flatten_s"], 1),
           "streams_nodes": [int(st.stream_nodes[i]) for i in range(st.n_streams)],
           "window_crowns": int(st.n_window_crowns), "window_crown_nodes": int(st.window_crown_nodes),
           "window_streams_that_are_candidate_crowns": int(st.n_window_streams_crown), "window_streams": int(st.n_window_streams),
           "window_stream_elements": int(st.window_stream_nodes), "seed_chunks": int(st.seed_chunks), "device_bytes": int(st.device_bytes)}
    for key, rd, nchk, k in (("default_batch", prep["short"], 256, 0), ("long_reads", prep["long"], 64, 1)):
        b = DeviceBatch(torch, rd, dev)
        b.place(mat, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            b.place(mat, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        pc, _ = mat.last_plans(rd.n_reads)
        out[key] = {"reads": rd.n_reads, "mean_entries": b.nw / rd.n_reads, "reads_per_s": rd.n_reads / dt, "ms_per_step": dt * 1e3,
                    "reads_by_plan_class": {PLAN_NAMES[c]: int(n) for c, n in enumerate(np.bincount(pc, minlength=7)) if n}}
        if prep["want"] is not None:
            out[key]["sample_matches_gpu"] = matches(prep["want"][k], b.out, nchk)
    mat.close()
    prep["g"].close()
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with python -m torch.distributed.run --nproc-per-node N")
        args.gpus = world

    import torch
    import wepp_amd as w

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the placement engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    # ---- synthetic workload (identical tree on every rank, per-rank reads) ----
    genome = args.workload == "genome"
    long_reads = args.read_len > 400 and not genome
    t0 = time.perf_counter()
    # the tree-shape ladder's host work (trees, flat images, batches, checker results) runs in background threads from
    # the start: it is ready when the main workload has been measured
    ladder_jobs = None
    if rank == 0 and world == 1 and not args.no_legs and not args.no_crowns and args.ladder_nodes > 0:
        from concurrent.futures import ThreadPoolExecutor
        ladder_pool = ThreadPoolExecutor(max_workers=3)
        want_chk = not args.no_cpu_baseline
        ladder_jobs = [ladder_pool.submit(ladder_prepare, w, args.ladder_nodes, label, kw, want_chk) for label, kw in LADDER]
    g = w.generate_tree(21, args.nodes)
    amp_len, amp_step = (args.read_len, int(args.read_len * 0.85)) if long_reads else (400, 300)
    p_sub = 0.03 if long_reads else 0.001
    p_n = args.p_n if args.p_n is not None else (0.0005 if genome else 0.02 if long_reads else 0.005)

    def gen_reads(seed, n, pn=p_n):
        if genome:
            return genome_samples(g, seed, n, p_sub, pn)
        return g.reads(seed, n, read_len=args.read_len, amplicon_len=amp_len, amplicon_step=amp_step,
                       p_substitution=p_sub, p_n=pn)

    # the timed loop rotates over N_BATCHES distinct batches, all resident in HBM (seeds 22 + 8 * rank + i: rank 0 places
    # seeds 22 .. 29): no step finds the index lines and the routing of the step before it in the caches
    seed0 = (900 if genome else 24 if long_reads else 22) + args.batches * rank
    batches_host = [gen_reads(seed0 + i, args.reads) for i in range(args.batches)]
    reads = batches_host[0]
    t_gen = time.perf_counter() - t0
    # one flatten (host), one upload (this rank's device): the C++ host flattens once per node and uploads from every
    # device thread (wepp_flat_create + wepp_mat_upload); a bench rank is its own process, so its flatten is rank-local
    # ... under torch.distributed.run every rank is a process: local rank 0 flattens and leaves the image in /dev/shm
    # (wepp_flat_save), the other ranks of the node read it back (wepp_flat_load): ONE flatten per node here too
    t0 = time.perf_counter()
    from wepp_amd.sharding import shared_flat_image
    flat = shared_flat_image(g.tree, dist if world > 1 else None, local_rank, f"{os.environ.get('MASTER_PORT', '0')}_{args.nodes}")
    t_flat = time.perf_counter() - t0
    t0 = time.perf_counter()
    mat = w.Mat(g.tree, device=local_rank, flat=flat)
    flat.close()
    mat.set_tile_reads(args.tile)
    mat.set_use_crowns(not args.no_crowns)
    mat.set_use_walk(not args.no_walk)
    t_up = time.perf_counter() - t0
    st = mat.stats
    batches = [DeviceBatch(torch, rd, dev) for rd in batches_host]
    batch = batches[0]
    R, nw = batch.R, batch.nw
    stream = torch.cuda.current_stream().cuda_stream

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # ---- the timed region: K steps, inputs resident in HBM ----------------------------------
    # Set-up, not warm-up: every distinct batch of the rotation is placed once, so that the handle's grow-only device
    # workspaces have the size the rotation needs before the W warm-up steps (a batch whose longest read changes the
    # tile size can need a workspace a quarter larger: one hipFree + hipMalloc of ~100 MB, 16 ms, once per handle -- with
    # 8 batches and W = 2 it landed in the timed steps of the long-read run: 3.7 instead of 2.9 ms per step)
    for b in batches:
        b.place(mat, stream)
    fence()
    for i in range(args.warmup):
        batches[i % len(batches)].place(mat, stream)
    fence()
    mat.timing_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        batches[i % len(batches)].place(mat, stream)
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    sweep_ms, n_launch, passes, alg_bytes = mat.last_timing()
    walk_reads, walk_iters = mat.last_walk()
    reads_timed = sum(batches[i % len(batches)].R for i in range(args.steps))
    batch.place(mat, stream)                  # (batch 0 once more: its routing is what `streams` reports)
    fence()
    tiers = mat.last_tiers(R)
    pcls, _ = mat.last_plans(R)
    ref_out = [t.clone() for t in batch.out]

    # ---- SURVEY 8(d)'s metric as defined: wall clock of wepp_place_batch with HOST buffers (validation of the
    # read words, H2D of the reads, kernels, D2H of the results inside) -----------------------
    pcie = None
    if args.pcie_steps > 0:
        hres = mat.place_batch(reads)                 # grows the handle's staging buffers once
        same = bool((hres.score == ref_out[1].cpu().numpy()).all() and
                    (hres.best_bfs_j == ref_out[0].cpu().numpy().view(np.uint32)).all() and
                    (hres.num_best == ref_out[2].cpu().numpy().view(np.uint32)).all())
        fence()
        t0 = time.perf_counter()
        per_call = []
        n_placed = 0
        for i in range(args.pcie_steps):
            rd = batches_host[(i + 1) % len(batches_host)]     # distinct host batches in rotation, like the timed loop
            t1 = time.perf_counter()
            hres = mat.place_batch(rd, out=hres)               # synchronous: the results are in host memory on return
            per_call.append(time.perf_counter() - t1)
            n_placed += rd.n_reads
        fence()
        el = max_over_ranks(time.perf_counter() - t0)
        pcie = {"value": n_placed * world / el, "ms_per_step": el / args.pcie_steps * 1e3,
                "ms_per_step_median": float(np.median(per_call)) * 1e3, "ms_per_step_min": float(np.min(per_call)) * 1e3,
                "steps": args.pcie_steps, "bytes_in_per_step": 4 * (R + 1) + 4 * nw, "bytes_out_per_step": 16 * R,
                "results_identical_to_timed_run": same}

    # ---- the "HBM-roofline run" of configs[2]: same batch, work skipping off, so that
    # every tile streams the whole-tree event stream once (not part of `value`) ----
    whole = None
    if not args.no_crowns and args.roofline_steps > 0:
        mat.set_use_crowns(False)
        mat.set_use_walk(False)
        batch.place(mat, stream)
        fence()
        mat.timing_reset()
        for _ in range(args.roofline_steps):
            batch.place(mat, stream)
        fence()
        w_ms, w_n, w_passes, w_bytes = mat.last_timing()
        same = all(bool((a == b).all()) for a, b in zip(ref_out, batch.out))
        mat.set_use_crowns(True)
        mat.set_use_walk(not args.no_walk)
        whole = {"kernel_ms_per_step": w_ms, "steps_timed": w_n, "stream_sweeps_per_step": w_passes,
                 "algorithmic_bytes_per_step": w_bytes, "results_identical_to_timed_run": same}

    # ---- sensitivity: how much of `value` is the generator's choice of |S| (rank 0 only) --------
    sens = None
    if rank == 0 and world == 1 and not args.no_sensitivity and not args.no_crowns and not genome:
        sens = []

        def leg(label, rd, steps=3):
            if rd.n_reads == 0:
                return
            b = DeviceBatch(torch, rd, dev)
            b.place(mat, stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                b.place(mat, stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            tr = mat.last_tiers(rd.n_reads)
            pc, _ = mat.last_plans(rd.n_reads)
            sens.append({"reads": label, "n_reads": rd.n_reads, "mean_entries": b.nw / rd.n_reads,
                         "reads_per_s": rd.n_reads / dt, "ms_per_step": dt * 1e3,
                         "share_on_whole_tree_stream": float((tr == st.n_streams - 1).mean()),
                         "share_on_window_crowns": float((tr == w.WINDOW_CROWN_SLOT).mean()),
                         "reads_by_plan_class": {w.PLAN_NAMES[c]: int(k) for c, k in enumerate(np.bincount(pc, minlength=7)) if k}})

        n_s = min(R, 1_000_000)
        for pn in ((0.005, 0.02, 0.05) if not long_reads else (0.02, 0.05)):
            leg(f"p_n = {pn}", gen_reads(122, n_s, pn))
        if not long_reads:
            pool = gen_reads(123, n_s, 0.06)     # mean ~9 entries: a pool with enough long lists to cut from
            for k in (0, 1, 2, 4, 8):
                leg(f"exactly {k} entries", truncate_reads(w, pool, k))

    if rank == 0:
        gpu_res = {"score": ref_out[1].cpu().numpy(), "best": ref_out[0].cpu().numpy().view(np.uint32),
                   "num_best": ref_out[2].cpu().numpy().view(np.uint32), "flags": ref_out[3].cpu().numpy().view(np.uint32)}
        total_reads = reads_timed * world          # (every rank places batches of the same sizes)
        value = total_reads / elapsed
        mode = "genome_samples" if genome else "long_reads" if long_reads else ("whole_tree" if (args.no_crowns and args.no_walk) else "short_reads")
        roof, cands = sweep_roofline(mode, R, sweep_ms, alg_bytes, passes, n_launch)
        if mode == "whole_tree":
            # every tile streams the whole-tree stream: the tiles of a launch sweep the same 1 MB chunk together, the bytes
            # come from the L2s -- the algorithmic rate can exceed the HBM peak and is an L2 figure
            roof.update({"served_from": "L2 (chunk-major 1 MB chunks shared by the tiles of a launch): frac > 1 is possible and says "
                                        "nothing about HBM; the binding unit of this run is vector issue (roofline_candidates)",
                         "l2_peak_gbs": L2_PEAK_GBS, "frac_of_l2_peak": roof["achieved"] / L2_PEAK_GBS})
        shape = (f"{R} synthetic whole-genome samples per GPU per step (a leaf genotype + 0.1 % substitutions + N rate {p_n}: what read_vcf "
                 f"makes of consensus genomes; {args.batches} batches in rotation from seed {seed0})" if genome else
                 f"{R} synthetic midnight-amplicon-like {args.read_len} bp reads per GPU per step (3 % substitutions, "
                 f"N rate {p_n}; {args.batches} batches in rotation from seed {seed0}); BASELINE.json configs[4] shape on one GPU" if long_reads else
                 f"{R} synthetic ARTIC-like {args.read_len} bp reads per GPU per step (0.1 % substitutions, N rate {p_n}; "
                 f"{args.batches} distinct batches in rotation, seeds {seed0}..{seed0 + args.batches - 1}); BASELINE.json configs[2]")
        counts = np.bincount(tiers, minlength=16)
        out = {
            "metric": "reads placed/sec on SARS-CoV-2 MAT (~16M nodes)",
            "value": value,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "value_pcie_inclusive": pcie["value"] if pcie else None,
            "value_note": "inputs and outputs resident in HBM (wepp_place_batch_device); value_pcie_inclusive (also under "
                          "config.pcie_inclusive_reads_per_s) is SURVEY 8(d)'s wall clock of wepp_place_batch with host buffers.  "
                          "Both are a best case of the synthetic generator (reads of reference-like genotypes with ~1 entry): "
                          "see `sensitivity`",
            "pcie_inclusive": pcie,
            "config": {
                "pcie_inclusive_reads_per_s": pcie["value"] if pcie else None,
                "pcie_inclusive_ms_per_step": pcie["ms_per_step"] if pcie else None,
                "distinct_batches_in_rotation": args.batches,
                "setup_places_every_batch_once": True,     # (device workspaces reach their size before warm-up; see the timed region)
                "workload": f"synthetic SARS-CoV-2-like MAT N={st.n_nodes} nodes M={st.n_mutations} mutations "
                            f"(seed 21, L=29903), {shape}",
                "reads_per_gpu": R,
                "read_words_per_gpu": nw,
                "tile_reads_T": args.tile,
                "work_skipping": not args.no_crowns,
                "parallelism": f"read-sharded x{world}, MAT replicated, no collective",
                "mat": {"nodes": int(st.n_nodes), "mutations": int(st.n_mutations), "events": int(st.n_events),
                        "blocks": int(st.n_blocks), "leaves": int(st.n_leaves), "max_depth": int(st.max_depth),
                        "device_bytes": int(st.device_bytes)},
                "setup_s": {"generate_tree_and_batches": round(t_gen, 1), "flatten_host": round(t_flat, 1), "upload_and_handle": round(t_up, 1)},
                "flatten": "one per node: local rank 0 flattens and shares the image through /dev/shm (wepp_flat_save / wepp_flat_load)" if world > 1
                           else "one flatten, one upload",
                "kernel_hash": kernel_hash(),
            },
            "roofline": roof,
            "roofline_candidates": cands,
            "walk": {"reads_walked_per_step": int(walk_reads), "wave_iterations_per_step": int(walk_iters // max(1, args.steps)),
                     "enabled": not args.no_walk,
                     "reads_by_plan_class": {w.PLAN_NAMES[c]: int(n) for c, n in enumerate(np.bincount(pcls, minlength=7)) if n}},
            "streams": [{"tau": int(st.stream_tau[i]), "nodes": int(st.stream_nodes[i]), "bytes": int(st.stream_bytes_of[i]),
                         "reads_routed": int(counts[i])} for i in range(st.n_streams)],
            "window_crowns": {"what": "per genome window (2560 positions every 1024) the nodes a read confined to the window can be "
                                      "placed on: out_w - in_w <= base(root) whatever it lists (the last crown of a window), and "
                                      "out_w <= its ROOT score (the smaller ones) -- no + |S| slack as for the tree-wide streams "
                                      "above (DESIGN.md 4.2c)",
                              "crowns": int(st.n_window_crowns), "nodes_in_all": int(st.window_crown_nodes),
                              "reads_routed": int(counts[w.WINDOW_CROWN_SLOT])},
            "window_streams": {"what": "what the tiles of reads with more than 32 entries inside one genome window sweep (plan class "
                                       "'window'; `streams` counts them under the whole tree): the window's candidate crown, or the "
                                       "whole tree as the window sees it where no crown was built",
                               "streams": int(st.n_window_streams), "of_which_candidate_crowns": int(st.n_window_streams_crown),
                               "elements_in_all": int(st.window_stream_nodes),
                               "reads_routed": int((pcls == w.PLAN_WIN).sum())},
        }
        if whole is not None:
            wr, wc = sweep_roofline("whole_tree", R, whole["kernel_ms_per_step"], whole["algorithmic_bytes_per_step"],
                                    whole["stream_sweeps_per_step"], whole["steps_timed"])
            valu = next((c for c in wc if c["bound"] == "valu_issue"), None)
            out["roofline_whole_tree"] = {
                "what": "same batch with work skipping and walks off: every 64-read tile streams the whole-tree event "
                        "stream once (BASELINE.json configs[2] 'HBM-roofline run'); not part of `value`",
                "reads_per_s": R / (whole["kernel_ms_per_step"] * 1e-3),
                "results_identical_to_timed_run": whole["results_identical_to_timed_run"],
                "served_from": "L2: the tiles of a launch sweep the same 1 MB chunk together (chunk-major order), so the "
                               "algorithmic rate is an L2 rate, not an HBM rate -- it is priced against the L2 peak, and the "
                               "kernel's binding unit is vector issue",
                "algorithmic_gbs": wr["achieved"], "l2_peak_gbs": L2_PEAK_GBS, "frac_of_l2_peak": wr["achieved"] / L2_PEAK_GBS,
                "algorithmic_bytes_per_step": wr["algorithmic_bytes_per_step"], "kernel_ms_per_step": wr["kernel_ms_per_step"],
                "hbm_side_traffic_gbs": wr.get("traffic_gbs"), "hbm_side_traffic_frac_of_8TBs": wr.get("traffic_frac"),
                "l2_hit_rate": wr.get("l2_hit_rate"), "valu_issue": valu, "provenance": wr.get("provenance")}
        out["sensitivity"] = sens
        # ---- the CPU checker (oracle/incremental_oracle.c) on the bench MAT: shared by the legs and the CPU baseline ----
        ot = inc = None
        if world == 1 and not args.no_cpu_baseline and args.cpu_baseline_seconds > 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_bridge
            ot = oracle_bridge.OracleTree(g.tree)
            inc = ot.incremental()
        # ---- the other configs of BASELINE.json on the same resident MAT (rank 0, one GPU): each leg is one GPU's share
        # of its config, device-resident and PCIe-inclusive, with its own contract-shaped roofline ----
        if world == 1 and not args.no_legs and not args.no_crowns and not genome and not long_reads:
            legs = []
            legs.append(run_leg(torch, mat, dev, stream, "configs[3]: 1.25 M 150 bp reads, one GPU's shard of the 10 M-read run", "short_reads",
                                [g.reads(52 + i, 1_250_000, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.001, p_n=0.005) for i in range(4)],
                                steps=8, pcie_steps=3, checker=inc))
            legs.append(run_leg(torch, mat, dev, stream, "configs[4]: 125 000 reads of 1.2 kb, one GPU's shard of the 1 M-read run", "long_reads",
                                [long_reads_batch(g, 24 + i, 125_000) for i in range(4)], steps=8, pcie_steps=3, checker=inc, n_check=128))
            legs.append(run_leg(torch, mat, dev, stream, "whole-genome samples (the input of usher_common: read_vcf's Missing_Samples), 20 000 per step", "genome_samples",
                                [genome_samples(g, 900 + i, 20_000) for i in range(4)], steps=8, pcie_steps=3, checker=inc, n_check=128))
            out["legs"] = legs
        # ---- tree-shape ladder: the default batch and 1.2 kb reads on trees of other shapes (what every number above is
        # a best case of: median root path ~20 mutations, no polytomies) ----
        if ladder_jobs is not None:
            out["tree_ladder"] = {"nodes": args.ladder_nodes,
                                  "what": "same generator, other shapes (wepp_gen_tree_params: depth_choices, p_hub, p_back_mutation); per tree "
                                          "the crown / candidate sizes, 1 M default reads and 50 000 reads of 1.2 kb per step, first reads "
                                          "checked against the incremental oracle",
                                  "trees": [ladder_measure(torch, w, dev, stream, j.result()) for j in ladder_jobs]}
            ladder_pool.shutdown()
        if ot is not None:
            out["cpu_baseline"] = cpu_baseline(g.tree, reads, gpu_res, args.cpu_baseline_seconds, ot, inc)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    mat.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
