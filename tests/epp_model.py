"""Pure-Python model of the EPP sweep (wepp_amd/csrc/epp_kernels.hip) over the flat MAT's EPP
event stream: checks the stream and the range logic on the CPU, without a GPU.  One read at a
time, no tiles / windows / chunks (those only change which events a lane looks at)."""
import numpy as np


def read_distances(epp_word, epp_node, n_nodes, pos, mut, start, end):
    """Distance of one read to every node, by arena index."""
    listed = {int(p): int(m) for p, m in zip(pos, mut)}
    D = sum(1 for m in listed.values() if m != 15)
    out = np.zeros(n_nodes, np.int64)
    prev = 0
    for w, nd in zip(epp_word.tolist(), epp_node.tolist()):
        out[prev:nd] = D
        prev = nd
        p = w & 0xFFFFF
        if not (start <= p <= end):
            continue
        refm = 1 << ((w >> 20) & 3)
        par = (w >> 22) & 15
        m = (w >> 26) & 15
        pare = par if par else refm
        if p in listed:
            al = listed[p]
            d = 0 if al == 15 else int(m != al) - int(pare != al)
        else:
            d = int(m != refm) - int(pare != refm)
        D += -d if (w >> 30) & 1 else d
    out[prev:n_nodes] = D
    return out


def epp_map(epp_word, epp_node, n_nodes, reads, genome_size, max_cached=2048):
    from wepp_amd import unpack_read_word
    R = reads.n_reads
    mp = np.zeros(R, np.int32); mult = np.zeros(R, np.uint32)
    score = np.zeros(n_nodes); counts = np.zeros((n_nodes, 50), np.int32)
    lists = []
    bin_size = genome_size // 50
    for r in range(R):
        pos, _, mu, _ = reads.entries(r)
        d = read_distances(epp_word, epp_node, n_nodes, pos, mu, int(reads.start[r]), int(reads.end[r]))
        m = int(d.min())
        epp = np.flatnonzero(d == m)
        mp[r] = m; mult[r] = len(epp)
        score[epp] += reads.degree[r] / ((1 + m) * len(epp))
        counts[epp, min(int(reads.start[r]) // bin_size, 49)] += reads.degree[r]
        lists.append(epp if len(epp) <= max_cached else np.zeros(0, np.int64))
    return dict(max_parsimony=mp, multiplicity=mult, score=score, counts=counts, lists=lists)
