#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory into profiles/<name>/summary.json
(per-launch means of the k_sweep counters + kernel-trace durations) and refreshes
profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.

HBM-side bytes follow MI355X_MICROARCH.md (HBM section): bytes = (FETCH_SIZE +
WRITE_SIZE) * 1024, with FETCH_SIZE doubled because on gfx950 it reports half of
the bytes of a coalesced streaming read (TCC_EA0_RDREQ tallied at 64 B per 128-B
request).  Infinity-Cache hits are counted by FETCH_SIZE, so this is the traffic
leaving the L2s, an upper bound on true HBM traffic."""
import collections, csv, glob, json, os, shutil, sys


def main(src, name, mode):
    out_dir = os.path.join("profiles", name)
    os.makedirs(out_dir, exist_ok=True)
    summ = {"source": src, "mode": mode, "kernel": "k_sweep"}
    ks = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        summ["kernel_stats"] = [{"name": r["Name"].split("(")[0], "calls": int(r["Calls"]),
                                 "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])} for r in rows[:6]]
        shutil.copy(ks[0], os.path.join(out_dir, "kernel_stats.csv"))
    bt = os.path.join(src, "bench_trace.json")
    steps = 3
    if os.path.exists(bt):
        try:
            b = json.load(open(bt))
            steps = b["steps"] + b["warmup"]
            summ["bench_under_trace"] = {k: b[k] for k in ("value", "ms_per_step")}
            summ["bench_under_trace"]["roofline"] = b["roofline"]
        except Exception:
            pass
    counters = {}
    for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc"):
        fs = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
        if not fs:
            continue
        shutil.copy(fs[0], os.path.join(out_dir, d + "_counter_collection.csv"))
        agg = collections.defaultdict(float)
        n = collections.defaultdict(int)
        for row in csv.DictReader(open(fs[0])):
            if "k_sweep" in row["Kernel_Name"]:  # k_sweep, k_sweep_multi
                agg[row["Counter_Name"]] += float(row["Counter_Value"])
                n[row["Counter_Name"]] += 1
        for k in agg:
            counters[k] = {"sum_over_run": agg[k], "launches": n[k], "per_step": agg[k] / steps}
    summ["counters_k_sweep"] = counters
    summ["steps_in_run"] = steps
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        f, wr = counters["FETCH_SIZE"]["per_step"], counters["WRITE_SIZE"]["per_step"]
        summ["traffic_bytes_per_step"] = {"raw": (f + wr) * 1024, "fetch_doubled": (2 * f + wr) * 1024}
    if "TCC_HIT_sum" in counters:
        h, m = counters["TCC_HIT_sum"]["sum_over_run"], counters["TCC_MISS_sum"]["sum_over_run"]
        summ["l2_hit_rate"] = h / (h + m)
    json.dump(summ, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
    tp = os.path.join("profiles", "pmc_traffic.json")
    t = json.load(open(tp)) if os.path.exists(tp) else {}
    if "traffic_bytes_per_step" in summ:
        t[mode] = {"traffic_bytes_per_step": summ["traffic_bytes_per_step"]["fetch_doubled"],
                   "raw_bytes_per_step": summ["traffic_bytes_per_step"]["raw"], "profile": out_dir,
                   "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the k_sweep launches of one step; "
                           "counts Infinity-Cache hits (traffic leaving the L2s)"}
        json.dump(t, open(tp, "w"), indent=1)
    print(json.dumps({k: summ[k] for k in summ if k not in ("counters_k_sweep",)}, indent=1)[:1800])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
