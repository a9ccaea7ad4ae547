"""Random small trees and samples for fuzzing (SURVEY.md Appendix B recipe:
tiny genome so that positions collide and back-mutations occur, true parent
alleles, a few masked mutations, IUPAC / N sample entries)."""
import numpy as np

from wepp_amd import Reads, Tree


def random_tree(rng, n_nodes=None, genome=60, p_masked=0.025, p_ambig=0.05, max_muts=4, root_muts=True,
                p_root_masked=0.1):
    n = int(n_nodes if n_nodes is not None else rng.integers(1, 65))
    ref = {p: 1 << int(rng.integers(0, 4)) for p in range(1, genome + 1)}
    parent = [-1]
    for i in range(1, n):
        parent.append(int(rng.integers(0, i)))
    geno = [dict() for _ in range(n)]  # position -> allele mask along the path
    muts = []
    for i in range(n):
        g = dict(geno[parent[i]]) if i else {}
        nm = int(rng.integers(0, max_muts + 1)) if (i or root_muts) else 0
        ml = []
        if i and rng.random() < p_masked:
            ml.append((-1, 0, 0, 0))
        if i == 0 and rng.random() < p_root_masked:
            # a masked root mutation; nucs as the loader leaves them (all zero) or,
            # rarely, as an in-memory tree could hold them
            if rng.random() < 0.5:
                ml.append((-1, 0, 0, 0))
            else:
                ml.append((-1, 1 << int(rng.integers(0, 4)), 0, 1 << int(rng.integers(0, 4))))
        poss = sorted(set(int(x) for x in rng.integers(1, genome + 1, size=nm)))
        for p in poss:
            cur = g.get(p, ref[p])
            while True:
                m = 1 << int(rng.integers(0, 4))
                if rng.random() < p_ambig:
                    m |= 1 << int(rng.integers(0, 4))
                if m != cur:
                    break
            ml.append((p, ref[p], cur, m))
            g[p] = m
        geno[i] = g
        muts.append(ml)
    return Tree.from_lists(parent, muts), ref


def random_sample(rng, ref, genome=60, max_k=7):
    k = int(rng.integers(0, max_k + 1))
    poss = sorted(set(int(x) for x in rng.integers(1, genome + 1, size=k)))
    ents = []
    for p in poss:
        u = rng.random()
        if u < 0.6:
            a = 1 << int(rng.integers(0, 4))
            ents.append((p, ref[p], a, 0))
        elif u < 0.8:
            a = int(rng.integers(1, 16))
            ents.append((p, ref[p], a, 1 if (a == 15 and rng.random() < 0.5) else 0))
        else:
            ents.append((p, ref[p], 15, 1))
    return ents


def reads_from_samples(samples):
    return Reads.from_lists(samples)
