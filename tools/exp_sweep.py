#!/usr/bin/env python3
"""Experiments on the sweep kernel: times one placement step for variants of the bench batch.
usage: exp_sweep.py [nodes] [reads] -- runs: standard batch, reads with no entry, reads with exactly one entry;
each with work skipping on and off.  Prints kernel ms (HIP events on the launch stream)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wepp_amd as w


def subset(reads, mask):
    k = np.diff(reads.read_off)
    keep = np.nonzero(mask)[0]
    off = np.zeros(len(keep) + 1, np.uint32)
    off[1:] = np.cumsum(k[keep])
    idx = np.concatenate([np.arange(reads.read_off[r], reads.read_off[r + 1]) for r in keep]) if off[-1] else np.zeros(0, np.int64)
    return w.Reads(off, reads.read_word[idx.astype(np.int64)] if off[-1] else np.zeros(0, np.uint32))


def main():
    nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    g = w.generate_tree(21, nodes)
    reads = g.reads(22, R * 3, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.001, p_n=0.005)
    k = np.diff(reads.read_off)
    mat = w.Mat(g.tree, device=0)
    dev = torch.device("cuda", 0)
    variants = {"standard": np.arange(len(k)) < R, "empty": k == 0, "one_entry": k == 1, "two_plus": k >= 2}
    for name, mask in variants.items():
        sub = subset(reads, mask)
        n = min(sub.n_reads, R)
        sub = sub.slice(0, n)
        nw = int(sub.read_off[-1])
        d_off = torch.from_numpy(sub.read_off.astype(np.int32)).to(dev)
        d_word = torch.from_numpy((sub.read_word if nw else np.zeros(1, np.uint32)).astype(np.int32)).to(dev)
        outs = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4)]
        for crowns in (True, False):
            mat.set_use_crowns(crowns)
            reps = 3 if crowns else 1
            for i in range(1 + reps):
                if i == 1:
                    torch.cuda.synchronize(); mat.timing_reset()
                mat.place_batch_device(d_off.data_ptr(), d_word.data_ptr(), n, nw, *[o.data_ptr() for o in outs],
                                       torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ms, nl, passes, ab = mat.last_timing()
            print(json.dumps({"variant": name, "reads": n, "words": nw, "crowns": crowns, "kernel_ms": round(ms, 4),
                              "passes": passes, "alg_GBps": round(ab / ms / 1e6, 1)}), flush=True)
    mat.close()


if __name__ == "__main__":
    main()
