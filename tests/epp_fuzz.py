"""Random windowed reads (WEPP raw_read, src/WEPP/read.hpp:8-14) for a fuzz tree: a window
[start, end] of the genome, the genotype of a random node inside it with a few substitution
errors and N's -- listed wherever the read differs from the reference (sam2pb.cpp:521-535)."""
import numpy as np

from wepp_amd import EppReads


def genotypes(tree, ref):
    """position -> allele along the root path, for every node (caller ids)."""
    n = tree.n_nodes
    geno = [None] * n
    order = np.argsort(tree.parent, kind="stable")
    done = np.zeros(n, bool)
    # parents have smaller ids in the fuzz trees; fall back to a worklist otherwise
    for i in range(n):
        p = int(tree.parent[i])
        g = dict(geno[p]) if p >= 0 else {}
        assert p < i
        for k in range(int(tree.mut_off[i]), int(tree.mut_off[i + 1])):
            if tree.mut_pos[k] >= 0:
                g[int(tree.mut_pos[k])] = int(tree.mut_mut[k])
        geno[i] = g
    return geno


def random_epp_reads(rng, tree, ref, genome, n_reads, max_len=25, p_err=0.05, p_n=0.05, geno=None, max_degree=4):
    geno = geno if geno is not None else genotypes(tree, ref)
    reads, start, end, degree = [], [], [], []
    for _ in range(n_reads):
        ln = int(rng.integers(1, max_len + 1))
        s = int(rng.integers(1, genome + 1))
        e = min(genome, s + ln - 1)
        g = geno[int(rng.integers(0, tree.n_nodes))]
        ents = []
        for p in range(s, e + 1):
            a = g.get(p, ref[p])
            if a not in (1, 2, 4, 8):                 # ambiguous tree allele: the read shows one base of it
                bits = [b for b in (1, 2, 4, 8) if a & b]
                a = bits[int(rng.integers(0, len(bits)))]
            u = rng.random()
            if u < p_n:
                a = 15
            elif u < p_n + p_err:
                a = 1 << int(rng.integers(0, 4))
            if a != ref[p]:
                ents.append((p, ref[p], a, 1 if a == 15 else 0))
        reads.append(ents)
        start.append(s); end.append(e); degree.append(int(rng.integers(1, max_degree + 1)))
    return EppReads.from_lists(reads, start, end, degree)
