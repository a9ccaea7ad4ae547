"""First-contact GPU check: C-ABI placement vs oracle on fuzz trees and a
synthetic config; prints a short report.  (The pytest -m gpu suite is the real
gate; this is a quick diagnostic.)"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import wepp_amd as w
import oracle_bridge as ob, fuzz_trees as ft


def compare(tree, reads, tile=64, tag=""):
    mat = w.Mat(tree)
    mat.set_tile_reads(tile)
    t0 = time.time()
    res = mat.place_batch(reads)
    dt = time.time() - t0
    want = ob.OracleTree(tree).place_batch(reads, nthreads=os.cpu_count())
    bad = np.flatnonzero((res.score != want["score"]) | (res.best_bfs_j != want["best_j"]) |
                         (res.num_best != want["num_best"]) | (res.has_unique != want["has_unique"]))
    print(f"{tag}: N={tree.n_nodes} R={reads.n_reads} tile={tile} gpu={dt*1e3:.1f}ms mismatches={len(bad)}")
    for r in bad[:5]:
        print("   read", r, "gpu", res.score[r], res.best_bfs_j[r], res.num_best[r], res.has_unique[r],
              "oracle", want[r], "S", [tuple(int(x[i]) for x in reads.entries(r)) for i in range(len(reads.entries(r)[0]))])
    mat.close()
    return len(bad)


def main():
    rng = np.random.default_rng(5)
    total = 0
    for it in range(40):
        tree, ref = ft.random_tree(rng)
        samples = [ft.random_sample(rng, ref) for _ in range(int(rng.integers(1, 100)))]
        total += compare(tree, ft.reads_from_samples(samples), tile=int(rng.choice([1, 3, 64])), tag=f"fuzz{it}")
    g = w.generate_tree(1, 50000, genome_len=15225, p_ambiguous=0.002, p_masked_node=0.0005, root_mutations=1)
    total += compare(g.tree, g.reads(2, 2000, p_iupac=0.05), tag="rsv-like")
    g = w.generate_tree(11, 100000)
    total += compare(g.tree, g.reads(12, 1000), tag="config2-sample")
    total += compare(g.tree, g.reads(13, 300, read_len=1200, amplicon_len=1200, amplicon_step=1000,
                                     p_substitution=0.03, p_n=0.02), tag="config5-like")
    print("TOTAL MISMATCHES", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
