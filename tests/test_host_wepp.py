"""C++ host mirror of WEPP's data path (wepp_amd/host/wepp_filter.*, wepp-epp): reads .pb
loader (sam.proto), read masking, site_read_map, create_condensed_tree, and -- on the GPU --
the whole `tree.pb + reads.pb -> haplotype scores` run against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import epp_fuzz
import fuzz_trees as ft
import pb_fixture as pbf
from test_host_cpp import _names, _tree_lists, _expected_dump
import wepp_amd as w

CLI = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wepp_amd", "wepp-epp")
CH = {1: "A", 2: "C", 4: "G", 8: "T"}


def _reference(ref, genome):
    return "".join(CH[ref[p]] for p in range(1, genome + 1))


def _random_read_records(rng, tree, ref, genome, n_reads, geno):
    """(name, start, content, degree): the genotype of a random node over a window, with a few
    errors, N's and '_' (a deleted base: skipped by the loader)."""
    recs = []
    for q in range(n_reads):
        ln = int(rng.integers(1, 40))
        s = int(rng.integers(1, genome + 1))
        e = min(genome, s + ln - 1)
        g = geno[int(rng.integers(0, tree.n_nodes))]
        content = []
        for p in range(s, e + 1):
            a = g.get(p, ref[p])
            if a not in CH:
                a = [b for b in (1, 2, 4, 8) if a & b][0]
            u = rng.random()
            c = CH[a]
            if u < 0.05:
                c = "N"
            elif u < 0.08:
                c = "_"
            elif u < 0.12:
                c = CH[1 << int(rng.integers(0, 4))]
            content.append(c)
        recs.append((f"read_{q}", s, "".join(content), int(rng.integers(1, 6))))
    return recs


def _as_reads(recs, reference, mask=()):
    """What load_reads_from_proto + mask_reads leave (sam2pb.cpp:489-549, arena.hpp:60-72)."""
    ents, start, end, degree = [], [], [], []
    code = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15}
    for (_, s, content, d) in recs:
        e = []
        for i, c in enumerate(content):
            if c != reference[s + i - 1] and c != "_" and (s + i) not in mask:
                e.append((s + i, code[reference[s + i - 1]], code[c], 1 if c == "N" else 0))
        ents.append(e); start.append(s); end.append(s + len(content) - 1); degree.append(d)
    return ents, start, end, degree


def _condense(parent, muts, sites):
    """create_condensed_tree (src/WEPP/util.cpp:79-133) restated: returns (parent, muts, sources)
    of the condensed tree in creation (BFS queue) order, ids of the original nodes."""
    n = len(parent)
    kids = [[] for _ in range(n)]
    root = parent.index(-1)
    for i, p in enumerate(parent):
        if p >= 0:
            kids[p].append(i)
    cpar, cmuts, csrc, corig = [-1], [[m for m in muts[root] if m[0] in sites]], [[root]], [root]
    queue = [(c, 0) for c in kids[root]]
    while queue:
        cur, cp = queue.pop(0)
        covered = [m for m in muts[cur] if m[0] in sites]
        if covered:
            cpar.append(cp); cmuts.append(covered); csrc.append([cur]); corig.append(cur)
            me = len(cpar) - 1
            queue.extend((c, me) for c in kids[cur])
        else:
            csrc[cp].append(cur)
            queue.extend((c, cp) for c in kids[cur])
    return cpar, cmuts, csrc, corig


def _sites(ents, start, end, mask=()):
    sites = set()
    for e, s, t in zip(ents, start, end):
        amb = {x[0] for x in e if x[2] == 15}
        sites |= {j for j in range(s, t + 1) if j not in amb and j not in mask}
    return sites


def _setup(tmp_path, rng, n_nodes, n_reads, genome=120, mask=()):
    tree, ref = ft.random_tree(rng, n_nodes=n_nodes, genome=genome, p_masked=0.0, p_root_masked=0.0, p_ambig=0.0)
    parent, muts = _tree_lists(tree)
    names = _names(parent)
    reference = _reference(ref, genome)
    recs = _random_read_records(rng, tree, ref, genome, n_reads, epp_fuzz.genotypes(tree, ref))
    pb, rpb, fa, bed = (str(tmp_path / x) for x in ("t.pb", "r.pb", "ref.fa", "mask.bed"))
    dfs = pbf.write_pb(pb, parent, names, muts)
    pbf.write_reads_pb(rpb, recs, {"merged_1": ["a", "b"]})
    open(fa, "w").write(">ref some description\n" + "\n".join(reference[i:i + 50].lower() for i in range(0, genome, 50)) + "\n")
    open(bed, "w").write("".join(f"ref\t{m - 1}\t{m}\n" for m in mask))
    newname = _expected_dump(parent, names, muts, dfs)
    return tree, parent, muts, newname, reference, recs, pb, rpb, fa, bed


def test_reads_pb_loader_and_condensed_tree(tmp_path):
    rng = np.random.default_rng(2025)
    for it in range(8):
        mask = (7, 33, 90) if it % 2 else ()
        d = tmp_path / f"c{it}"
        d.mkdir()
        tree, parent, muts, newname, reference, recs, pb, rpb, fa, bed = _setup(d, rng, int(rng.integers(2, 60)), 30, mask=mask)
        out = subprocess.run([CLI, "-i", pb, "-r", rpb, "-f", fa, "-m", bed, "--dump"], check=True, capture_output=True,
                             text=True).stdout.splitlines()
        ents, start, end, degree = _as_reads(recs, reference, mask)
        got_reads = [l.split() for l in out if l.startswith("read ")]
        assert len(got_reads) == len(recs)
        for g, rec, e, s, t, dg in zip(got_reads, recs, ents, start, end, degree):
            assert g[1] == rec[0] and (int(g[2]), int(g[3]), int(g[4])) == (s, t, dg)
            assert [tuple(int(x) for x in f.split(":")) for f in g[5:]] == [(p, r, a) for (p, r, a, _) in e]
        cpar, cmuts, csrc, corig = _condense(parent, [[(m[0], m[1], m[3]) for m in ml] for ml in muts], _sites(ents, start, end, mask))
        haps = [l.split() for l in out if l.startswith("hap ")]
        # arena order = pre-order of the condensed tree; compare as sets keyed by name
        got = {h[1]: (h[2], int(h[3]), [tuple(int(x) for x in f.split(":")) for f in h[4:]]) for h in haps}
        assert len(got) == len(cpar)
        for k in range(len(cpar)):
            name = newname[corig[k]]
            pname = newname[corig[cpar[k]]] if cpar[k] >= 0 else "-"
            assert got[name] == (pname, len(csrc[k]), sorted(cmuts[k])), (it, k)


def test_uncondense_leaves(tmp_path):
    # leaf "c" stands for two identical samples and carries a mutation: it becomes an internal node
    # with the samples as children (mutation_annotated_tree.cpp:1232-1247)
    parent, names = [-1, 0, 0], ["r", "c", "B"]
    d = pbf.Data()
    d.newick, dfs = pbf.newick_and_dfs(parent, names)
    muts = [[], [(10, 1, 1, 2)], []]
    for i in dfs:
        lst = d.node_mutations.add()
        for (p, r, pa, mu) in muts[i]:
            m = lst.mutation.add()
            m.position, m.ref_nuc, m.par_nuc = p, r.bit_length() - 1, pa.bit_length() - 1
            m.mut_nuc.append(mu.bit_length() - 1)
    cn = d.condensed_nodes.add()
    cn.node_name = "c"
    cn.condensed_leaves.extend(["x", "y"])
    pb, rpb, fa = (str(tmp_path / x) for x in ("t.pb", "r.pb", "ref.fa"))
    open(pb, "wb").write(d.SerializeToString())
    pbf.write_reads_pb(rpb, [("q", 1, "ACGT", 1)])
    open(fa, "w").write(">r\nACGTACGTACGTACGT\n")
    out = subprocess.run([CLI, "-i", pb, "-r", rpb, "-f", fa, "--dump"], check=True, capture_output=True, text=True).stdout
    nodes = dict(l.split()[1:3] for l in out.splitlines() if l.startswith("node "))
    assert set(nodes) == {"node_1", "node_2", "B", "x", "y"} and nodes["x"] == nodes["y"] == "node_2"


@pytest.mark.gpu
def test_wepp_epp_end_to_end(tmp_path, oracle):
    rng = np.random.default_rng(77)
    mask = (15, 64)
    tree, parent, muts, newname, reference, recs, pb, rpb, fa, bed = _setup(tmp_path, rng, 250, 400, genome=200, mask=mask)
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([CLI, "-i", pb, "-r", rpb, "-f", fa, "-m", bed, "-d", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ents, start, end, degree = _as_reads(recs, reference, mask)
    cpar, cmuts, csrc, corig = _condense(parent, [[(m[0], m[1], m[3]) for m in ml] for ml in muts], _sites(ents, start, end, mask))
    ctree = w.Tree.from_lists(cpar, cmuts)
    reads = w.EppReads.from_lists(ents, start, end, degree)
    ot = oracle.OracleTree(ctree)
    want = ot.epp_map(reads, genome_size=200)
    dfs_ids = ot.dfs_ids()
    rows = [l.split("\t") for l in open(out / "haplotype_scores.tsv").read().splitlines()[1:]]
    assert [x[0] for x in rows] == [newname[corig[i]] for i in dfs_ids]
    assert np.allclose([float(x[1]) for x in rows], want["score"], rtol=1e-9, atol=1e-9)
    assert np.allclose([float(x[2]) for x in rows], want["divergence"], rtol=1e-9, equal_nan=True)
    assert [int(x[3]) for x in rows] == [len(csrc[i]) for i in dfs_ids]
    rr = [l.split("\t") for l in open(out / "read_placements.tsv").read().splitlines()[1:]]
    assert [int(x[4]) for x in rr] == want["max_parsimony"].tolist()
    assert [int(x[5]) for x in rr] == want["multiplicity"].tolist()
