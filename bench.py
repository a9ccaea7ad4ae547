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
KERNEL_SOURCES = ("place_kernels.hip", "device_mat.hpp", "flatmat.hpp", "flatmat.cpp", "capi.cpp", "sort_reads.hip")


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
    ap.add_argument("--p-n", type=float, default=None, help="per-base N rate of the synthetic reads (default 0.005 "
                    "for 150 bp reads, SURVEY 8(d) config 2/3; 0.02 for long reads, config 5)")
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


def cpu_baseline(tree, reads, gpu_res, target_s):
    """Oracle (CPU restatement of the reference loop) on a bounded sample of the
    same reads; the NODE range is split over all usable host threads, which is
    how the reference parallelises a sample (tbb::parallel_for,
    src/usher_common.cpp:386)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bridge
    cores = usable_cores()
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
    ot.close()
    return {
        "value": n / dt,
        "unit": "reads/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {n} reads of the step batch on the same MAT, node range split over {cores} threads "
                  f"(like tbb::parallel_for over nodes), {dt:.1f} s; probe read {t1:.2f} s",
        "sample_matches_gpu": ok,
    }


def pmc_profile(mode):
    """Counters per step of k_sweep from the committed rocprofv3 --pmc profile of THIS workload
    (`mode`: short_reads / whole_tree / long_reads) and THIS build of the kernels
    (profiles/pmc_counters.json, written by tools/summarize_profile.py from tools/profile.sh runs):
    HBM-side bytes (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE,
    x 1024) and vector / scalar instruction counts.  None unless the profile's kernel hash is the hash
    of the sources this process was built from."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_counters.json")) as fh:
            t = json.load(fh)[mode]
    except (OSError, KeyError, ValueError):
        return None
    if t.get("kernel_hash") != kernel_hash():
        return None
    return t


def sweep_roofline(mode, sweep_ms, alg_bytes, passes, n_launch):
    """The roofline objects of one measurement.  `hbm`: algorithmic and counter traffic against the 8 TB/s
    peak.  The other candidates come from the committed rocprofv3 --pmc profile of this workload and this
    build: vector instruction issue (256 CUs x one wave-instruction per cycle) and the vector memory address
    units (TA busy cycles: a gather whose 64 lanes touch 64 cache lines keeps its CU's unit busy for 64+
    cycles -- what bounds the per-read walks).  `binding` = the candidate with the largest utilisation."""
    prof = pmc_profile(mode)
    secs = sweep_ms * 1e-3
    alg = alg_bytes / secs / 1e9
    hbm = {"bound": "hbm", "kernel": "placement kernels (k_walk / k_sweep / k_finalize)", "achieved": alg, "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": alg / HBM_PEAK_GBS,
           "achieved_note": "ALGORITHMIC bytes over the live kernel time: a sweep reads its stream once per 64-read tile, "
                            "a walk 26 bytes per loop iteration (DESIGN.md 4.1, 4.2).  Streams and index are shared between "
                            "reads and largely served by the L2s / Infinity Cache: see traffic / traffic_gbs for what left the L2s",
           "algorithmic_bytes_per_step": alg_bytes, "passes_per_step": passes, "kernel_ms_per_step": sweep_ms,
           "steps_timed": n_launch, "traffic": None, "traffic_gbs": None, "traffic_frac": None}
    cands = []
    if prof is None:
        binding = {"bound": "unknown", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                   "kernel_ms_per_step": sweep_ms, "steps_timed": n_launch,
                   "provenance": "no rocprofv3 --pmc profile of this workload for this build of the kernels is committed "
                                 "(profiles/pmc_counters.json, tools/profile.sh + tools/summarize_profile.py)"}
        return binding, hbm, cands
    prov = (f"counters per step from {prof['profile']} (rocprofv3 --pmc runs of this workload, kernel hash "
            f"{prof['kernel_hash']}); duration measured live with HIP events; the profiled run's kernel time was "
            f"{prof.get('kernel_ms_per_step_trace')} ms per step")
    if prof.get("traffic_bytes_per_step") is not None:
        hbm.update({"traffic": prof["traffic_bytes_per_step"], "traffic_gbs": prof["traffic_bytes_per_step"] / secs / 1e9,
                    "traffic_frac": prof["traffic_bytes_per_step"] / secs / 1e9 / HBM_PEAK_GBS,
                    "l2_hit_rate": prof.get("l2_hit_rate"),
                    "traffic_note": "bytes per step leaving the L2s, (2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes; "
                                    "Infinity-Cache hits are counted, so true HBM traffic is lower still"})
        cands.append({"bound": "hbm", "achieved": hbm["traffic_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s (counter traffic)",
                      "frac": hbm["traffic_frac"]})
    if prof.get("valu_insts_per_step") is not None:
        a = prof["valu_insts_per_step"] / secs / 1e9
        cands.append({"bound": "valu_issue", "achieved": a, "peak": VALU_ISSUE_PEAK, "unit": "G wave-instructions/s",
                      "frac": a / VALU_ISSUE_PEAK, "valu_insts_per_step": prof["valu_insts_per_step"],
                      "salu_insts_per_step": prof.get("salu_insts_per_step"),
                      "peak_note": "256 CUs x one vector instruction per cycle x 2.4 GHz max clock (the clock held under load is "
                                   "lower: frac is a lower bound)"})
    if prof.get("ta_busy_cycles_per_step") is not None:
        a = prof["ta_busy_cycles_per_step"] / secs / 1e9
        cands.append({"bound": "vmem_address", "achieved": a, "peak": VALU_ISSUE_PEAK, "unit": "G busy cycles/s over the 256 address units",
                      "frac": a / VALU_ISSUE_PEAK, "ta_busy_cycles_per_step": prof["ta_busy_cycles_per_step"],
                      "vmem_read_insts_per_step": prof.get("vmem_rd_insts_per_step"),
                      "tcp_cache_accesses_per_step": prof.get("tcp_cache_accesses_per_step"),
                      "peak_note": "TA_TA_BUSY summed over the 256 vector-memory address units / (256 x 2.4 GHz x kernel time): the "
                                   "share of the time the units were busy splitting gathers into cache-line requests"})
    best = max(cands, key=lambda c: c["frac"]) if cands else None
    binding = dict(best) if best else {"bound": "unknown", "achieved": None, "peak": None, "unit": None, "frac": None}
    binding.update({"kernel": "placement kernels (k_walk / k_sweep / k_finalize)", "traffic": prof.get("traffic_bytes_per_step"),
                    "kernel_ms_per_step": sweep_ms, "steps_timed": n_launch, "provenance": prov})
    return binding, hbm, cands


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
    long_reads = args.read_len > 400
    t0 = time.perf_counter()
    g = w.generate_tree(21, args.nodes)
    amp_len, amp_step = (args.read_len, int(args.read_len * 0.85)) if long_reads else (400, 300)
    p_sub = 0.03 if long_reads else 0.001
    p_n = args.p_n if args.p_n is not None else (0.02 if long_reads else 0.005)

    def gen_reads(seed, n, pn=p_n):
        return g.reads(seed, n, read_len=args.read_len, amplicon_len=amp_len, amplicon_step=amp_step,
                       p_substitution=p_sub, p_n=pn)

    reads = gen_reads((24 if long_reads else 22) + rank, args.reads)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    mat = w.Mat(g.tree, device=local_rank)
    mat.set_tile_reads(args.tile)
    mat.set_use_crowns(not args.no_crowns)
    mat.set_use_walk(not args.no_walk)
    t_flat = time.perf_counter() - t0
    st = mat.stats
    batch = DeviceBatch(torch, reads, dev)
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
    for _ in range(args.warmup):
        batch.place(mat, stream)
    fence()
    mat.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.place(mat, stream)
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    sweep_ms, n_launch, passes, alg_bytes = mat.last_timing()
    walk_reads, walk_iters = mat.last_walk()
    tiers = mat.last_tiers(R)
    ref_out = [t.clone() for t in batch.out]

    # ---- SURVEY 8(d)'s metric as defined: wall clock of wepp_place_batch with HOST buffers (validation of the
    # read words, H2D of the reads, kernels, D2H of the results inside) -----------------------
    pcie = None
    if args.pcie_steps > 0:
        hres = mat.place_batch(reads)                 # grows the handle's staging buffers once
        fence()
        t0 = time.perf_counter()
        per_call = []
        for _ in range(args.pcie_steps):
            t1 = time.perf_counter()
            hres = mat.place_batch(reads, out=hres)           # synchronous: the results are in host memory on return
            per_call.append(time.perf_counter() - t1)
        fence()
        el = max_over_ranks(time.perf_counter() - t0)
        same = bool((hres.score == ref_out[1].cpu().numpy()).all() and
                    (hres.best_bfs_j == ref_out[0].cpu().numpy().view(np.uint32)).all())
        pcie = {"value": R * world * args.pcie_steps / el, "ms_per_step": el / args.pcie_steps * 1e3,
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
    if rank == 0 and world == 1 and not args.no_sensitivity and not args.no_crowns:
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
            sens.append({"reads": label, "n_reads": rd.n_reads, "mean_entries": b.nw / rd.n_reads,
                         "reads_per_s": rd.n_reads / dt, "ms_per_step": dt * 1e3,
                         "share_on_whole_tree_stream": float((tr == st.n_streams - 1).mean()),
                         "median_stream_nodes": int(st.stream_nodes[int(np.median(tr))])})

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
        total_reads = R * world * args.steps
        value = total_reads / elapsed
        mode = "long_reads" if long_reads else ("whole_tree" if (args.no_crowns and args.no_walk) else "short_reads")
        issue, hbm, cands = sweep_roofline(mode, sweep_ms, alg_bytes, passes, n_launch)
        shape = (f"{R} synthetic midnight-amplicon-like {args.read_len} bp reads per GPU per step (3 % substitutions, "
                 f"N rate {p_n}; seed 24+rank); BASELINE.json configs[4] shape on one GPU" if long_reads else
                 f"{R} synthetic ARTIC-like {args.read_len} bp reads per GPU per step (0.1 % substitutions, N rate {p_n}; "
                 f"seed 22+rank); BASELINE.json configs[2]")
        counts = np.bincount(tiers, minlength=st.n_streams)
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
            "value_note": "inputs and outputs resident in HBM (wepp_place_batch_device); value_pcie_inclusive is SURVEY 8(d)'s "
                          "wall clock of wepp_place_batch with host buffers.  Both are a best case of the synthetic "
                          "generator (reads of reference-like genotypes with ~1 entry): see `sensitivity`",
            "value_pcie_inclusive": pcie["value"] if pcie else None,
            "pcie_inclusive": pcie,
            "config": {
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
                "setup_s": {"generate": round(t_gen, 1), "flatten_upload": round(t_flat, 1)},
                "kernel_hash": kernel_hash(),
            },
            "roofline": issue,
            "roofline_hbm": hbm,
            "roofline_candidates": cands,
            "walk": {"reads_walked_per_step": int(walk_reads), "wave_iterations_per_step": int(walk_iters // max(1, args.steps)),
                     "enabled": not args.no_walk},
            "streams": [{"tau": int(st.stream_tau[i]), "nodes": int(st.stream_nodes[i]), "bytes": int(st.stream_bytes_of[i]),
                         "reads_routed": int(counts[i])} for i in range(st.n_streams)],
        }
        if whole is not None:
            wi, wh, wc = sweep_roofline("whole_tree", whole["kernel_ms_per_step"], whole["algorithmic_bytes_per_step"],
                                        whole["stream_sweeps_per_step"], whole["steps_timed"])
            out["roofline_whole_tree"] = {
                "what": "same batch with work skipping off: every 64-read tile streams the whole-tree event "
                        "stream once (BASELINE.json configs[2] 'HBM-roofline run'); not part of `value`",
                "reads_per_s": R / (whole["kernel_ms_per_step"] * 1e-3),
                "results_identical_to_timed_run": whole["results_identical_to_timed_run"],
                "binding": wi, "hbm": wh, "candidates": wc,
                "served_from": "L2 (chunk-major 1 MB chunks)", "l2_peak": L2_PEAK_GBS,
                "algorithmic_frac_of_l2_peak": wh["achieved"] / L2_PEAK_GBS}
        out["sensitivity"] = sens
        if world == 1 and not args.no_cpu_baseline and args.cpu_baseline_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(g.tree, reads, gpu_res, args.cpu_baseline_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    mat.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
