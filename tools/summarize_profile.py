#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory into profiles/<name>/summary.json (per-step counters of the
placement kernels + kernel-trace durations) and refreshes profiles/pmc_counters.json, which bench.py reads for
its roofline objects (keyed by workload mode and by the hash of the kernel sources the profile was taken on).

    python tools/summarize_profile.py gpurun_out/prof_<tag> <name> <mode: short_reads|whole_tree|long_reads|probe>

(mode probe: a tools/walk_probe.py leg profiled with PROFILE_PROG; PROFILE_STEPS = placements in the run; not entered into
pmc_counters.json)

profiles/pmc_counters.json is keyed "<mode>:<reads per step>" (+ the kernel hash inside the entry): bench.py quotes a
profile only for the workload, the batch size and the build it was taken on.

HBM-side bytes follow MI355X_MICROARCH.md (HBM section): bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, with
FETCH_SIZE doubled because on gfx950 it reports half of the bytes of a coalesced streaming read.
Infinity-Cache hits are counted by FETCH_SIZE, so this is the traffic leaving the L2s, an upper bound on
true HBM traffic.  Counters are summed over every kernel between the library's two timing events of a
placement call (sweeps, walks, their job tables and finalizes); routing is excluded, like in the timing."""
import collections, csv, glob, json, os, shutil, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PLACE = ("k_sweep", "k_walk", "k_seed", "k_finalize", "k_gather_jobs", "k_first_pos", "rocprim", "block_id_wrapper", "radix", "scan")


def is_place(name):
    return any(t in name for t in PLACE) and "k_route" not in name and "k_scatter" not in name


def main(src, name, mode):
    from bench import kernel_hash
    out_dir = os.path.join("profiles", name)
    os.makedirs(out_dir, exist_ok=True)
    summ = {"source": src, "mode": mode, "kernel_hash": kernel_hash()}
    steps = int(os.environ.get("PROFILE_STEPS", "3"))     # (a probe leg, tools/walk_probe.py: 1 + PROBE_STEPS placements)
    bt = os.path.join(src, "bench_trace.json")
    if mode.startswith("probe"):
        summ["probe_under_trace"] = open(bt).read().strip().splitlines()[-1] if os.path.exists(bt) else None
    elif os.path.exists(bt):
        try:
            b = json.load(open(bt))
            steps = b["steps"] + b["warmup"] + 1      # (+ the untimed step that reports batch 0's routing)
            if b["config"].get("setup_places_every_batch_once"):
                steps += b["config"]["distinct_batches_in_rotation"]     # (+ bench.py's set-up pass over the rotation)
            summ["reads_per_step"] = b["config"]["reads_per_gpu"]
            summ["bench_under_trace"] = {k: b[k] for k in ("value", "ms_per_step")}
            summ["bench_under_trace"]["kernel_ms_per_step"] = b["roofline"]["kernel_ms_per_step"]
            summ["bench_under_trace"]["algorithmic_bytes_per_step"] = b["roofline"].get("algorithmic_bytes_per_step")
            assert b["config"]["kernel_hash"] == summ["kernel_hash"], "profile taken on other kernel sources"
        except Exception as e:  # noqa: BLE001
            summ["bench_under_trace_error"] = repr(e)
    # (a directory merged back by several gpurun calls holds one file per run: the newest is this run's)
    newest = lambda fs: sorted(fs, key=os.path.getmtime, reverse=True)
    ks = newest(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")))
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        summ["kernel_stats"] = [{"name": r["Name"].split("(")[0][-48:], "calls": int(r["Calls"]),
                                 "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])} for r in rows[:10]]
        shutil.copy(ks[0], os.path.join(out_dir, "kernel_stats.csv"))
        tot = sum(float(r["TotalDurationNs"]) for r in rows if is_place(r["Name"]))
        summ["placement_kernels_ms_per_step_sum_of_durations"] = tot / 1e6 / steps
    counters = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        fs = newest(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")))
        if not fs:
            continue
        shutil.copy(fs[0], os.path.join(out_dir, os.path.basename(d) + "_counter_collection.csv"))
        agg = collections.defaultdict(float)
        for row in csv.DictReader(open(fs[0])):
            if is_place(row["Kernel_Name"]):
                agg[row["Counter_Name"]] += float(row["Counter_Value"])
        for k in agg:
            counters[k] = {"sum_over_run": agg[k], "per_step": agg[k] / steps}
    summ["counters_per_step"] = {k: v["per_step"] for k, v in counters.items()}
    summ["steps_in_run"] = steps
    entry = {"profile": out_dir, "kernel_hash": summ["kernel_hash"],
             "kernel_ms_per_step_trace": summ.get("bench_under_trace", {}).get("kernel_ms_per_step")}
    c = summ["counters_per_step"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        entry["traffic_bytes_per_step"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        entry["raw_bytes_per_step"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    for src_name, dst in (("SQ_INSTS_VALU", "valu_insts_per_step"), ("SQ_INSTS_SALU", "salu_insts_per_step"),
                          ("SQ_INSTS_VMEM_RD", "vmem_rd_insts_per_step"), ("SQ_INSTS_LDS", "lds_insts_per_step"),
                          ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_cycles_per_step"), ("SQ_WAVES", "waves_per_step"),
                          ("TA_TA_BUSY_sum", "ta_busy_cycles_per_step"), ("TA_FLAT_READ_WAVEFRONTS_sum", "ta_read_wavefronts_per_step"),
                          ("TCP_TOTAL_CACHE_ACCESSES_sum", "tcp_cache_accesses_per_step"), ("TCP_TCC_READ_REQ_sum", "tcp_tcc_read_req_per_step"),
                          ("GRBM_GUI_ACTIVE", "grbm_gui_active_per_step")):
        if src_name in c:
            entry[dst] = c[src_name]
    if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0) > 0:
        entry["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        entry["tcc_miss_bytes_per_step"] = c["TCC_MISS_sum"] * 64
    if c.get("SQ_WAVE_CYCLES"):
        entry["sq_wait_any_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_INSTS_VMEM_RD") and c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        entry["cache_lines_per_vmem_read_inst"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["SQ_INSTS_VMEM_RD"]
    summ["entry"] = entry
    json.dump(summ, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
    if mode.startswith("probe"):              # (a probe leg is not a bench workload: nothing for bench.py to quote)
        print(json.dumps(entry, indent=1))
        return
    tp = os.path.join("profiles", "pmc_counters.json")
    t = json.load(open(tp)) if os.path.exists(tp) else {}
    t[f"{mode}:{summ.get('reads_per_step', 0)}"] = entry
    json.dump(t, open(tp, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
