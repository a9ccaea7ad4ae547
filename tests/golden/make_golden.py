"""Generates tests/golden/mapper2_cases.json.

The reference has no tests or fixtures for this path and cannot
be built in this image without stand-in headers, so these vectors are produced
by oracle/mapper2_oracle.c (the line-by-line C restatement of mapper2_body and
the usher_common per-sample loop).  They pin the oracle against regressions
and give the GPU path fixed inputs; they are NOT reference-generated.  The one
reference-derived known answer is the SURVEY.md Appendix B case, stored under
"survey_appendix_b" with the values printed in SURVEY.md.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import fuzz_trees as ft          # noqa: E402
import oracle_bridge as ob       # noqa: E402
from wepp_amd import Tree        # noqa: E402

A, C, G, T, N = 1, 2, 4, 8, 15


def tree_to_json(tree):
    return dict(parent=tree.parent.tolist(), mut_off=tree.mut_off.tolist(), mut_pos=tree.mut_pos.tolist(),
                mut_ref=tree.mut_ref.tolist(), mut_par=tree.mut_par.tolist(), mut_mut=tree.mut_mut.tolist())


def run_case(name, tree, samples, note=""):
    ot = ob.OracleTree(tree)
    outs = []
    for S in samples:
        pos = [s[0] for s in S]; ref = [s[1] for s in S]; mut = [s[2] for s in S]; ms = [s[3] for s in S]
        o = ot.place_sample(pos, ref, mut, ms, want_best_vec=True)
        p = ot.place_sample(pos, ref, mut, ms, per_node_scores=True)
        outs.append(dict(sample=[list(map(int, s)) for s in S], score=int(o["score"]), num_best=int(o["num_best"]),
                         best_j=int(o["best_j"]), has_unique=int(o["has_unique"]),
                         best_j_vec=[int(x) for x in o["best_j_vec"]],
                         node_scores=[int(x) for x in p["node_scores"]]))
    return dict(name=name, note=note, tree=tree_to_json(tree), bfs_ids=[int(x) for x in ot.bfs_ids()], results=outs)


def hand_cases():
    cases = []
    # ((A,B),(C,D)) from SURVEY.md Appendix B
    t = Tree.from_lists([-1, 0, 0, 1, 1, 2, 2],
                        [[], [(100, A, A, G)], [(400, T, T, C)], [(200, C, C, T)], [(300, G, G, A)], [(100, A, A, G)], []])
    cases.append(run_case("survey_appendix_b", t, [[(100, A, G, 0), (200, C, T, 0), (500, A, C, 0)]],
                          "expected -p scores 3,2,4,1,3,3,5; best leaf A (bfs j=3), score 1, num_best 1"))
    samples = [
        [],                                                    # empty S
        [(100, A, N, 1), (200, C, N, 1)],                      # all-N
        [(100, A, G | A, 0)],                                  # IUPAC containing ref
        [(100, A, G | C, 0), (300, G, A | T, 0)],              # IUPAC without ref
        [(100, A, G, 0)],
        [(400, T, C, 0), (100, A, G, 0)][::-1],
        [(100, A, C, 0)],                                      # other allele at a mutated site
        [(150, C, T, 0)],                                      # site the tree never mutates
    ]
    cases.append(run_case("quartet_edge_samples", t, samples))
    # root with mutations (one masked), back-mutation on a branch, repeated position, zero-mutation internal node
    t2 = Tree.from_lists(
        [-1, 0, 0, 1, 1, 2, 2, 3, 3],
        [[(-1, 0, 0, 0), (10, A, A, C), (20, G, G, T)],       # root: masked + two real
         [(10, A, C, A)],                                      # back to ref
         [],                                                   # zero-mutation internal
         [(10, A, A, G), (30, T, T, C)],                      # same position again deeper
         [(40, C, C, G | A)],                                  # ambiguous mut_nuc on a leaf
         [(-1, 0, 0, 0), (50, A, A, T)],                      # masked first, leaf
         [(20, G, T, G)],                                      # leaf back-mutation
         [(30, T, C, T)],
         []])                                                  # zero-mutation leaf
    s2 = [[], [(10, A, C, 0)], [(10, A, G, 0), (30, T, C, 0)], [(20, G, T, 0)], [(20, G, N, 1)],
          [(10, A, G, 0), (20, G, T, 0), (30, T, C, 0), (40, C, G, 0)], [(40, C, A, 0)], [(50, A, T, 0)],
          [(10, A, C | G, 0), (60, T, A, 0)]]
    cases.append(run_case("root_muts_masked_backmut", t2, s2))
    # ties: equal scores broken by num_leaves then by larger bfs j
    t3 = Tree.from_lists([-1, 0, 0, 0, 1, 1, 2, 2, 3],
                         [[], [(5, A, A, C)], [(5, A, A, C)], [(5, A, A, C)], [], [], [], [], []])
    cases.append(run_case("ties_num_leaves_then_bfs", t3, [[(5, A, C, 0)], [(5, A, C, 0), (9, G, T, 0)], []]))
    # single node
    t4 = Tree.from_lists([-1], [[(7, C, C, T)]])
    cases.append(run_case("single_node", t4, [[], [(7, C, T, 0)], [(7, C, G, 0)], [(8, A, G, 0)]]))
    return cases


def main():
    rng = np.random.default_rng(20260101)
    cases = hand_cases()
    for i in range(40):
        tree, ref = ft.random_tree(rng)
        samples = [ft.random_sample(rng, ref) for _ in range(6)]
        cases.append(run_case(f"fuzz_{i}", tree, samples))
    out = os.path.join(HERE, "mapper2_cases.json")
    with open(out, "w") as fh:
        json.dump(dict(generator="tests/golden/make_golden.py (oracle/mapper2_oracle.c)", cases=cases), fh)
    print("wrote", out, os.path.getsize(out), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
