#!/usr/bin/env python3
"""Event counters of the sweep per stream (needs a -DWEPP_SWEEP_STATS build:
tools/build_variant.sh stats -DWEPP_SWEEP_STATS; WEPP_PLACE_LIB=variants/stats/libwepp_place.so).
Runs one step of the bench workload with work skipping on, then off."""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wepp_amd as w
from wepp_amd._lib import lib

NAMES = ["block_visits", "blocks_with_hit", "hit_events", "hit_read_matches", "heavy_evals", "heavy_reduced",
         "blocks_with_summary_update", "waves", "cyc_setup", "cyc_nohit_blocks", "cyc_hit_blocks_light_only",
         "cyc_hit_blocks_before_eval", "cyc_evals", "cyc_setup_clear", "cyc_setup_stage", "cyc_setup_checkpoint",
         "t_list", "t_read_off", "t_first_words", "t_prefix_clear", "-", "-", "-", "-"]


def stats(reset=True):
    buf = (ctypes.c_ulonglong * (16 * 24))()
    assert lib.wepp_debug_sweep_stats(buf, 1 if reset else 0) == 0
    return np.array(buf[:]).reshape(16, 24)


def main():
    nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    read_len = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    g = w.generate_tree(21, nodes)
    amp_len, amp_step = (400, 300) if read_len <= 400 else (read_len, int(read_len * 0.85))
    reads = g.reads(22, R, read_len=read_len, amplicon_len=amp_len, amplicon_step=amp_step,
                    p_substitution=0.001 if read_len <= 400 else 0.03, p_n=0.005 if read_len <= 400 else 0.02)
    k = np.diff(reads.read_off)
    print(json.dumps({"entries_per_read_hist": np.bincount(k)[:12].tolist()}))
    mat = w.Mat(g.tree, device=0)
    dev = torch.device("cuda", 0)
    nw = int(reads.read_off[-1])
    d_off = torch.from_numpy(reads.read_off.astype(np.int32)).to(dev)
    d_word = torch.from_numpy(reads.read_word.astype(np.int32)).to(dev)
    outs = [torch.zeros(R, dtype=torch.int32, device=dev) for _ in range(4)]
    st = mat.stats
    for crowns in ((True,) if os.environ.get("STATS_CROWNS_ONLY") else (True, False)):
        mat.set_use_crowns(crowns)
        stats(True)
        mat.place_batch_device(d_off.data_ptr(), d_word.data_ptr(), R, nw, *[o.data_ptr() for o in outs],
                               torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        s = stats(True)
        for t in range(st.n_streams):
            if s[t, 7]:
                print(json.dumps({"crowns": crowns, "tau": int(st.stream_tau[t]), "stream_nodes": int(st.stream_nodes[t]),
                                  **{n: int(v) for n, v in zip(NAMES, s[t]) if n != "-"}}))
    mat.close()


if __name__ == "__main__":
    main()
