#!/usr/bin/env python3
"""Run by tests/test_gpu_parity.py::test_full_size_nrich_and_long_shard in a CHILD process whose environment does not
carry the test suite's WEPP_IX_PRE_MIN_NODES=0: the library then runs with the product's defaults (per-entry
pre-test bytes only on streams of >= 8192 nodes).  Checks, at 16 M nodes, the batches the bench's sensitivity
ladder times -- N rate 5 % and "exactly 8 entries" (bench.py) -- and configs[4] at its per-GPU shard size
(125 000 reads of 1.2 kb) against the incremental CPU checker (oracle/incremental_oracle.c, proven equal to the
faithful restatement of mapper2_body by tests/test_incremental.py); reference semantics:
src/usher_mapper.cpp:168-506, src/usher_common.cpp:386-446.  Prints one JSON line; exits non-zero on a mismatch."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import oracle_bridge  # noqa: E402
import wepp_amd as w  # noqa: E402
from bench import truncate_reads  # noqa: E402
from wepp_amd import Reads  # noqa: E402


def gather(reads, idx):
    off = np.zeros(len(idx) + 1, np.uint32)
    words = []
    for i, q in enumerate(idx):
        a, b = int(reads.read_off[q]), int(reads.read_off[q + 1])
        words.append(reads.read_word[a:b])
        off[i + 1] = off[i] + (b - a)
    return Reads(off, np.concatenate(words) if words else np.zeros(0, np.uint32))


def same(got, idx, want, what):
    for f, k in (("score", "score"), ("best_bfs_j", "best_j"), ("num_best", "num_best"), ("has_unique", "has_unique")):
        a = getattr(got, f)[idx]
        if not (a == want[k]).all():
            bad = np.nonzero(a != want[k])[0][:5]
            raise SystemExit(f"{what}: {f} differs from the checker at reads {np.asarray(idx)[bad].tolist()}: "
                             f"{a[bad].tolist()} vs {want[k][bad].tolist()}")


def main():
    assert "WEPP_IX_PRE_MIN_NODES" not in os.environ, "this check is about the product's defaults"
    n_nodes = int(os.environ.get("NRICH_NODES", "16000000"))
    n_reads = int(os.environ.get("NRICH_READS", "1000000"))
    n_long = int(os.environ.get("NRICH_LONG_READS", "125000"))
    nthr = os.cpu_count() or 1
    t0 = time.perf_counter()
    g = w.generate_tree(21, n_nodes)
    mat = w.Mat(g.tree)
    ot = oracle_bridge.OracleTree(g.tree)
    inc = ot.incremental()
    out = {"nodes": n_nodes, "setup_s": round(time.perf_counter() - t0, 1), "batches": []}

    # ---- the two N-rich batches of bench.py's ladder, whole, as the bench places them ----
    pool = g.reads(123, n_reads, p_n=0.06)
    batches = [("p_n = 0.05", g.reads(122, n_reads, p_n=0.05)), ("exactly 8 entries", truncate_reads(w, pool, 8))]
    classes_seen = set()
    for label, rd in batches:
        res = mat.place_batch(rd)
        pcls, pst = mat.last_plans(rd.n_reads)
        _, wcrown = mat.last_crowns(rd.n_reads)
        # (reads on window crowns -- stream slot 15 -- are counted per crown level)
        key = pcls.astype(np.int32) * 1024 + pst.astype(np.int32) * 16 + np.where(wcrown == 255, 0, wcrown)
        pick = []
        for k in np.unique(key):
            pick.extend(np.nonzero(key == k)[0][:200].tolist())
        # ... and the chunked classes in strength: they are what these batches spend their time in
        for c in (w.PLAN_WALKC8, w.PLAN_WALKC16):
            pick.extend(np.nonzero(pcls == c)[0][:2500].tolist())
        step = max(1, rd.n_reads // 6000)
        pick = np.array(sorted(set(pick) | set(range(0, rd.n_reads, step))))
        assert len(pick) >= min(5000, rd.n_reads), len(pick)
        t1 = time.perf_counter()
        same(res, pick, inc.place_batch(gather(rd, pick), nthreads=nthr), label)
        checked = np.zeros(rd.n_reads, bool)
        checked[pick] = True
        plans = {}
        for k in np.unique(key):
            members = key == k
            assert (checked & members).sum() >= min(200, members.sum()), (label, int(k))
            plans[f"{w.PLAN_NAMES[k // 1024]}:{(k // 16) % 64}" + (f".{k % 16}" if (k // 16) % 64 == w.WINDOW_CROWN_SLOT else "")] = \
                [int(members.sum()), int((checked & members).sum())]
        classes_seen |= set(np.unique(pcls).tolist())
        # the whole batch against the placement without work skipping (whole-tree sweeps), all four result arrays
        mat.set_use_crowns(False)
        mat.set_use_walk(False)
        r0 = mat.place_batch(rd)
        mat.set_use_crowns(True)
        mat.set_use_walk(True)
        for f in ("score", "best_bfs_j", "num_best", "flags"):
            assert (getattr(r0, f) == getattr(res, f)).all(), (label, "work skipping on / off", f)
        out["batches"].append({"reads": label, "n_reads": rd.n_reads, "checked": int(len(pick)),
                               "checker_s": round(time.perf_counter() - t1, 1), "plans_total_checked": plans})
    if n_nodes >= 1_000_000:
        assert {w.PLAN_WALK8, w.PLAN_WALK16} <= classes_seen, sorted(classes_seen)
    out["classes_seen"] = [w.PLAN_NAMES[c] for c in sorted(classes_seen)]

    # ---- configs[4] at its per-GPU shard size: 1.2 kb reads (window plans cut into their product number of waves) ----
    lr = g.reads(24, n_long, read_len=1200, amplicon_len=1200, amplicon_step=1020, p_substitution=0.03, p_n=0.02)
    rl = mat.place_batch(lr)
    lcls, _ = mat.last_plans(lr.n_reads)
    every = np.arange(0, lr.n_reads, 25)          # (5 000 of the shard's 125 000: the incremental checker takes seconds)
    t1 = time.perf_counter()
    same(rl, every, inc.place_batch(gather(lr, every), nthreads=nthr), "1.2 kb shard")
    # ... and the WHOLE shard against the placement without any work skipping (every read sweeps the whole tree: no
    # crowns, no window candidates, no walks), all four result arrays
    mat.set_use_crowns(False)
    mat.set_use_walk(False)
    r0 = mat.place_batch(lr)
    mat.set_use_crowns(True)
    mat.set_use_walk(True)
    for f in ("score", "best_bfs_j", "num_best", "flags"):
        assert (getattr(r0, f) == getattr(rl, f)).all(), ("long shard, work skipping on / off", f)
    half = lr.n_reads // 2
    ra, rb = mat.place_batch(lr.slice(0, half)), mat.place_batch(lr.slice(half, lr.n_reads))
    for f in ("score", "best_bfs_j", "num_best", "flags"):
        assert (np.concatenate([getattr(ra, f), getattr(rb, f)]) == getattr(rl, f)).all(), ("two halves", f)
    sub = lr.slice(0, min(5000, lr.n_reads))
    mat.set_tile_reads(16)
    r16 = mat.place_batch(sub)
    mat.set_tile_reads(64)
    for f in ("score", "best_bfs_j", "num_best", "flags"):
        assert (getattr(r16, f) == getattr(rl, f)[: sub.n_reads]).all(), ("tile size", f)
    out["long_reads"] = {"n_reads": lr.n_reads, "whole_shard_equals_no_work_skipping": True, "checked": int(len(every)), "checker_s": round(time.perf_counter() - t1, 1),
                         "window_plan_share": float((lcls == w.PLAN_WIN).mean()), "mean_entries": float(lr.read_off[-1]) / lr.n_reads}
    # best_j_vec of long reads (wepp_best_nodes): listed on the window's candidates -- on the whole tree, the only
    # tree-wide stream a root score of ~36 admits, 20 000 reads x 16 M nodes would take minutes
    sub = lr.slice(0, min(20000, lr.n_reads))
    rs = mat.place_batch(sub)
    t1 = time.perf_counter()
    vec = mat.best_nodes(sub, rs)
    dt = time.perf_counter() - t1
    for q in range(0, sub.n_reads, 100):
        want = inc.place_sample(*sub.entries(q), want_best_vec=True)["best_j_vec"]
        if vec[q].tolist() != want.tolist():
            raise SystemExit(f"best_j_vec of long read {q}: {vec[q].tolist()[:8]} vs {want.tolist()[:8]}")
    out["long_reads"]["best_nodes"] = {"reads": sub.n_reads, "seconds": round(dt, 3), "checked": len(range(0, sub.n_reads, 100)),
                                       "nodes_listed": int(sum(len(v) for v in vec))}
    assert dt < 10.0 or n_nodes < 1_000_000, dt
    inc.close()
    ot.close()
    mat.close()
    out["total_s"] = round(time.perf_counter() - t0, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
