#!/usr/bin/env python3
"""tools/host_proxy.py [--nodes N] [--handles 8] [--procs 5] [--reads 1250000] [--iters 6] [--cores 16]

What an 8-GPU node asks of the HOST, measured on the one GPU of the box (the driver's 8-GPU scaling run cannot be
launched from here): PCIe-inclusive placement (wepp_place_batch: check + stage + H2D + kernels + D2H + copy-out) of
configs[3]'s per-GPU shard, 1.25 M reads, by
  (a) ONE handle (the single-rank rate),
  (b) H handles on H host threads of one process, all on device 0 -- the C++ host's model (usher_place_samples: one
      flatten, one handle and one host thread per device; the handles split the host threads the process may use),
  (c) P processes (one handle each, the flat image shared through a file: wepp_flat_save / wepp_flat_load) -- bench.py's
      model, one rank per GPU; the GPU box admits at most 6 processes on its card (this one included), so P <= 5,
each with the process free to use every core it is given and again restricted to --cores cores (sched_setaffinity).
The GPU is shared by all handles here, so the aggregate also contains the kernels' queueing on ONE device: the number
to read is how far the aggregate is from H x the single rate once the kernels' share is taken out, and the host CPU
seconds per million reads.  Prints one JSON line."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_seconds():
    t = os.times()
    return t.user + t.system + t.children_user + t.children_system


def loop(mat, batches, iters):
    res = mat.place_batch(batches[0])
    for i in range(iters):
        res = mat.place_batch(batches[(i + 1) % len(batches)], out=res)
    return res


def child(args_json):
    import wepp_amd as w
    a = json.loads(args_json)
    flat = w.FlatView.load(a["image"])
    mat = w.Mat(None, device=0, flat=flat)
    flat.close()
    z = np.load(a["reads"])
    batches = [w.Reads(z[f"off{i}"], z[f"word{i}"]) for i in range(a["n_batches"])]
    loop(mat, batches, 1)                  # warm-up: staging buffers, workspaces
    print("READY", flush=True)
    sys.stdin.readline()                   # the parent releases all children together
    t0 = time.perf_counter()
    c0 = cpu_seconds()
    loop(mat, batches, a["iters"])
    print(json.dumps({"wall_s": time.perf_counter() - t0, "cpu_s": cpu_seconds() - c0}), flush=True)
    mat.close()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=16_000_000)
    ap.add_argument("--handles", type=int, default=8)
    ap.add_argument("--procs", type=int, default=5)
    ap.add_argument("--reads", type=int, default=1_250_000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cores", type=int, default=16)
    args = ap.parse_args()
    import subprocess
    import wepp_amd as w
    all_cores = sorted(os.sched_getaffinity(0))
    g = w.generate_tree(21, args.nodes)
    n_batches = 3
    batches = [g.reads(52 + i, args.reads, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.001, p_n=0.005) for i in range(n_batches)]
    t0 = time.perf_counter()
    flat = w.FlatView(g.tree)
    t_flat = time.perf_counter() - t0
    image = "/dev/shm/wepp_host_proxy_image.bin"
    reads_file = "/dev/shm/wepp_host_proxy_reads.npz"
    t0 = time.perf_counter()
    flat.save(image)
    t_save = time.perf_counter() - t0
    np.savez(reads_file, **{f"off{i}": b.read_off for i, b in enumerate(batches)}, **{f"word{i}": b.read_word for i, b in enumerate(batches)})
    out = {"nodes": args.nodes, "reads_per_call": args.reads, "calls_timed_per_handle": args.iters, "flatten_s": round(t_flat, 1),
           "image_save_s": round(t_save, 1), "image_bytes": os.path.getsize(image), "cores_available": len(all_cores), "legs": []}
    t0 = time.perf_counter()
    probe = w.FlatView.load(image)
    out["image_load_s"] = round(time.perf_counter() - t0, 1)
    probe.close()

    def threads_leg(n_handles, cores):
        os.sched_setaffinity(0, set(all_cores[:cores]) if cores else set(all_cores))
        mats = [w.Mat(None, device=0, flat=flat) for _ in range(n_handles)]      # (the pools size themselves at the first big call)
        print(f"[proxy] {n_handles} handle(s) uploaded", file=sys.stderr, flush=True)
        for i, m in enumerate(mats):
            loop(m, batches, 1)
            print(f"[proxy] handle {i} warmed up", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        c0 = cpu_seconds()
        th = [threading.Thread(target=loop, args=(m, batches, args.iters)) for m in mats]
        for t in th:
            t.start()
        for t in th:
            t.join()
        wall, cpu = time.perf_counter() - t0, cpu_seconds() - c0
        print(f"[proxy] {n_handles} thread(s) done", file=sys.stderr, flush=True)
        for m in mats:
            m.close()
        n = n_handles * (args.iters + 1) * args.reads
        return {"model": f"{n_handles} handle(s) on {n_handles} thread(s) of one process", "cores": cores or len(all_cores), "wall_s": wall,
                "aggregate_reads_per_s": n / wall, "ms_per_call_per_handle": wall / (args.iters + 1) * 1e3, "host_cpu_s_per_million_reads": cpu / n * 1e6,
                "host_cores_busy": cpu / wall}

    def procs_leg(n_procs, cores):
        aff = set(all_cores[:cores]) if cores else set(all_cores)
        os.sched_setaffinity(0, aff)
        payload = json.dumps({"image": image, "reads": reads_file, "n_batches": n_batches, "iters": args.iters})
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", payload], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
              for _ in range(n_procs)]
        for p in ps:
            line = p.stdout.readline()
            while line and not line.startswith("READY"):
                line = p.stdout.readline()
        t0 = time.perf_counter()
        for p in ps:
            p.stdin.write("go\n")
            p.stdin.flush()
        reps = [json.loads(p.stdout.readline()) for p in ps]
        wall = time.perf_counter() - t0
        for p in ps:
            p.wait()
        n = n_procs * (args.iters + 1) * args.reads
        cpu = sum(r["cpu_s"] for r in reps)
        return {"model": f"{n_procs} process(es), one handle each, image shared through /dev/shm", "cores": cores or len(all_cores), "wall_s": wall,
                "aggregate_reads_per_s": n / wall, "ms_per_call_per_handle": max(r["wall_s"] for r in reps) / (args.iters + 1) * 1e3,
                "host_cpu_s_per_million_reads": cpu / n * 1e6, "host_cores_busy": cpu / wall}

    for cores in (0, args.cores):
        for leg in ([lambda c=cores: threads_leg(1, c), lambda c=cores: threads_leg(args.handles, c)] +
                    ([lambda c=cores: procs_leg(min(5, args.procs), c)] if args.procs else [])):
            out["legs"].append(leg())
            print(json.dumps(out["legs"][-1]), file=sys.stderr, flush=True)
    os.sched_setaffinity(0, set(all_cores))
    flat.close()
    os.remove(image)
    os.remove(reads_file)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
