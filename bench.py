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


def pmc_traffic(mode):
    """HBM-side bytes per step of k_sweep from the committed rocprofv3 PMC profile of
    this same command (profiles/pmc_traffic.json, written by tools/summarize_profile.py);
    None when no profile is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            t = json.load(fh)[mode]
        return t["traffic_bytes_per_step"]
    except (OSError, KeyError, ValueError):
        return None


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
    t0 = time.perf_counter()
    g = w.generate_tree(21, args.nodes)
    amp_len, amp_step = (400, 300) if args.read_len <= 400 else (args.read_len, int(args.read_len * 0.85))
    reads = g.reads(22 + rank, args.reads, read_len=args.read_len, amplicon_len=amp_len, amplicon_step=amp_step,
                    p_substitution=0.001 if args.read_len <= 400 else 0.03,
                    p_n=0.005 if args.read_len <= 400 else 0.02)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    mat = w.Mat(g.tree, device=local_rank)
    mat.set_tile_reads(args.tile)
    mat.set_use_crowns(not args.no_crowns)
    t_flat = time.perf_counter() - t0
    st = mat.stats

    R = reads.n_reads
    nw = int(reads.read_off[-1])
    d_off = torch.from_numpy(reads.read_off.astype(np.int32)).to(dev)
    d_word = torch.from_numpy((reads.read_word if nw else np.zeros(1, np.uint32)).astype(np.int32)).to(dev)
    d_best = torch.zeros(R, dtype=torch.int32, device=dev)
    d_score = torch.zeros(R, dtype=torch.int32, device=dev)
    d_nbest = torch.zeros(R, dtype=torch.int32, device=dev)
    d_flags = torch.zeros(R, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        mat.place_batch_device(d_off.data_ptr(), d_word.data_ptr(), R, nw, d_best.data_ptr(), d_score.data_ptr(),
                               d_nbest.data_ptr(), d_flags.data_ptr(), stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    mat.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    sweep_ms, n_launch, passes, alg_bytes = mat.last_timing()

    # ---- the "HBM-roofline run" of configs[2]: same batch, work skipping off, so that
    # every tile streams the whole-tree event stream once (not part of `value`) ----
    whole = None
    if not args.no_crowns and args.roofline_steps > 0:
        ref_out = [t.clone() for t in (d_best, d_score, d_nbest, d_flags)]
        mat.set_use_crowns(False)
        step()
        fence()
        mat.timing_reset()
        for _ in range(args.roofline_steps):
            step()
        fence()
        w_ms, w_n, w_passes, w_bytes = mat.last_timing()
        same = all(bool((a == b).all()) for a, b in zip(ref_out, (d_best, d_score, d_nbest, d_flags)))
        mat.set_use_crowns(True)
        whole = {"kernel_ms_per_step": w_ms, "steps_timed": w_n, "stream_sweeps_per_step": w_passes,
                 "algorithmic_bytes_per_step": w_bytes, "results_identical_to_timed_run": same}

    if rank == 0:
        gpu_res = {"score": d_score.cpu().numpy(), "best": d_best.cpu().numpy().view(np.uint32),
                   "num_best": d_nbest.cpu().numpy().view(np.uint32), "flags": d_flags.cpu().numpy().view(np.uint32)}
        total_reads = R * world * args.steps
        value = total_reads / elapsed
        achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9
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
            "config": {
                "workload": f"synthetic SARS-CoV-2-like MAT N={st.n_nodes} nodes M={st.n_mutations} mutations "
                            f"(seed 21, L=29903), {R} synthetic ARTIC-like {args.read_len} bp reads per GPU per step "
                            f"(seed 22+rank); BASELINE.json configs[2]",
                "reads_per_gpu": R,
                "read_words_per_gpu": nw,
                "tile_reads_T": args.tile,
                "work_skipping": not args.no_crowns,
                "parallelism": f"read-sharded x{world}, MAT replicated, no collective",
                "mat": {"nodes": int(st.n_nodes), "mutations": int(st.n_mutations), "events": int(st.n_events),
                        "blocks": int(st.n_blocks), "leaves": int(st.n_leaves), "max_depth": int(st.max_depth),
                        "device_bytes": int(st.device_bytes)},
                "setup_s": {"generate": round(t_gen, 1), "flatten_upload": round(t_flat, 1)},
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_sweep",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic("work_skipping_off" if args.no_crowns else "work_skipping_on"),
                "traffic_note": "bytes per step leaving the L2s, (2*FETCH_SIZE+WRITE_SIZE)*1024 from a separate "
                                "rocprofv3 --pmc run of this command (profiles/); Infinity-Cache hits included",
                "algorithmic_bytes_per_step": alg_bytes,
                "stream_sweeps_per_step": passes,
                "kernel_ms_per_step": sweep_ms,
                "steps_timed": n_launch,
                "whole_tree_stream_bytes": int(st.stream_bytes),
                "streams": [{"tau": int(st.stream_tau[i]), "nodes": int(st.stream_nodes[i]),
                             "bytes": int(st.stream_bytes_of[i])} for i in range(st.n_streams)],
            },
        }
        if whole is not None:
            a = whole["algorithmic_bytes_per_step"] / (whole["kernel_ms_per_step"] * 1e-3) / 1e9
            out["roofline_whole_tree"] = {
                "what": "same batch with work skipping off: every 64-read tile streams the whole-tree event "
                        "stream once (BASELINE.json configs[2] 'HBM-roofline run'); not part of `value`.  `achieved` "
                        "counts the bytes every tile sweeps (algorithmic); the stream is cut into ~1 MB chunks swept "
                        "chunk-major, so most of them are served by the L2s (`traffic` = bytes leaving the L2s) and "
                        "the kernel is vector-issue bound (DESIGN.md 4.1)",
                "bound": "hbm", "kernel": "k_sweep", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": a / HBM_PEAK_GBS, "traffic": pmc_traffic("work_skipping_off"),
                # the same rate against what actually serves the bytes (MI355X_MICROARCH.md: L2 ~34.5 TB/s aggregate);
                # frac can exceed 1 because the HBM peak is not the binding resource once the chunks are L2-resident
                "served_from": "L2 (chunk-major 1 MB chunks)", "l2_peak": L2_PEAK_GBS, "frac_of_l2_peak": a / L2_PEAK_GBS,
                "reads_per_s": R / (whole["kernel_ms_per_step"] * 1e-3), **whole}
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
