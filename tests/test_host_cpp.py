"""The C++ host mirror of the reference interface (wepp_amd/host): .pb / VCF
loaders on the CPU, and the usher-compatible driver `wepp-usher` end to end on
the GPU against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import fuzz_trees as ft
import pb_fixture as pbf
import wepp_amd as w

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "wepp_amd", "wepp-usher")


def _names(parent):
    n = len(parent)
    has_child = [False] * n
    for p in parent:
        if p >= 0:
            has_child[p] = True
    return [f"label{i}" if has_child[i] else f"leaf_{i}|x/2021" for i in range(n)]


def _tree_lists(tree):
    muts = []
    for i in range(tree.n_nodes):
        a, b = int(tree.mut_off[i]), int(tree.mut_off[i + 1])
        muts.append([(int(tree.mut_pos[k]), int(tree.mut_ref[k]), int(tree.mut_par[k]), int(tree.mut_mut[k]))
                     for k in range(a, b)])
    return tree.parent.tolist(), muts


def _expected_dump(parent, names, muts, dfs):
    """What the loader must produce: internal nodes renamed node_<k> in order of
    appearance in the Newick (= DFS order), mutations with mut == par dropped,
    masked mutations zeroed."""
    newname = {}
    k = 0
    has_child = [False] * len(parent)
    for p in parent:
        if p >= 0:
            has_child[p] = True
    for i in dfs:
        if has_child[i]:
            k += 1
            newname[i] = f"node_{k}"
        else:
            newname[i] = names[i]
    return newname


def _as_vcf_reader_sees(e):
    """An allele that decodes to N -- 'N' itself, or 'V' through upstream's
    fall-through (mutation_annotated_tree.cpp:65-71) -- is read back as missing (:2105-2111)."""
    p, r, a, ms = e
    if a in (15, 7):
        return (p, r, 15, 1)
    return (p, r, a, 1 if ms else 0)


def _run_dump(pb, vcf):
    out = subprocess.run([CLI, "-i", pb, "-v", vcf, "--dump"], check=True, capture_output=True, text=True).stdout
    nodes, samples = {}, {}
    for line in out.splitlines():
        f = line.split()
        if f[0] == "node":
            nodes[f[1]] = (f[3], [tuple(int(x) for x in t.split(":")) for t in f[5:]])
        elif f[0] == "sample":
            samples[f[1]] = [tuple(int(x) for x in t.split(":")) for t in f[2:]]
    return nodes, samples


@pytest.mark.parametrize("compress", [False, True])
def test_pb_and_vcf_loaders(tmp_path, compress):
    rng = np.random.default_rng(5 + compress)
    for it in range(12):
        tree, ref = ft.random_tree(rng, n_nodes=int(rng.integers(2, 80)), p_root_masked=0.2)
        parent, muts = _tree_lists(tree)
        names = _names(parent)
        # add a no-op mutation (mut == par) that the loader must drop (mutation_annotated_tree.cpp:580-582)
        muts[-1] = muts[-1] + [(61, 1, 2, 2)]
        samples = [ft.random_sample(rng, ref) for _ in range(5)]
        snames = [f"sample_{i}" for i in range(5)]
        pb = str(tmp_path / f"t{it}.pb") + (".gz" if compress else "")
        vcf = str(tmp_path / f"s{it}.vcf") + (".gz" if compress else "")
        dfs = pbf.write_pb(pb, parent, names, muts, compress=compress)
        pbf.write_vcf(vcf, snames, samples, compress=compress)
        nodes, got_samples = _run_dump(pb, vcf)
        newname = _expected_dump(parent, names, muts, dfs)
        assert len(nodes) == len(parent)
        for i in range(len(parent)):
            par_name, got = nodes[newname[i]]
            assert par_name == (newname[parent[i]] if parent[i] >= 0 else "-")
            # the .pb stores par_nuc as ONE nucleotide index, so an ambiguous parent allele is
            # read back as its highest base (the scorer never reads par_nuc anyway)
            lp = lambda pa: 1 << (pa.bit_length() - 1)
            want = [(p, r, lp(pa), mu) if p >= 0 else (p, 0, 0, 0) for (p, r, pa, mu) in muts[i]
                    if p < 0 or mu != lp(pa)]
            assert got == sorted(want, key=lambda t: t[0]), (it, i)
        for s, ents in zip(snames, samples):
            # an ambiguity code that happens to be N is read back as missing (mutation_annotated_tree.cpp:2108-2111)
            want = [_as_vcf_reader_sees(e) for e in ents]
            assert got_samples[s] == want, (it, s)


def test_samples_already_in_tree_are_ignored(tmp_path):
    parent = [-1, 0, 0]
    names = ["r", "A", "B"]
    pb, vcf = str(tmp_path / "t.pb"), str(tmp_path / "s.vcf")
    pbf.write_pb(pb, parent, names, [[], [(10, 1, 1, 2)], []])
    pbf.write_vcf(vcf, ["A", "new1"], [[(10, 1, 2, 0)], [(10, 1, 2, 0)]])
    nodes, samples = _run_dump(pb, vcf)
    assert list(samples) == ["new1"] and set(nodes) == {"node_1", "A", "B"}


def test_loader_errors_do_not_crash(tmp_path):
    r = subprocess.run([CLI, "-i", str(tmp_path / "missing.pb"), "-v", str(tmp_path / "x.vcf"), "--dump"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "Could not open" in r.stderr


@pytest.mark.gpu
def test_wepp_usher_end_to_end(tmp_path, oracle):
    """usher -i tree.pb -v samples.vcf -n -d out [-p]: files and values as the reference writes them."""
    rng = np.random.default_rng(99)
    tree, ref = ft.random_tree(rng, n_nodes=300, genome=200, p_root_masked=0.0)
    parent, muts = _tree_lists(tree)
    names = _names(parent)
    samples = [ft.random_sample(rng, ref, genome=200) for _ in range(40)]
    snames = [f"s{i}" for i in range(40)]
    pb, vcf = str(tmp_path / "t.pb.gz"), str(tmp_path / "s.vcf")
    dfs = pbf.write_pb(pb, parent, names, muts, compress=True)
    pbf.write_vcf(vcf, snames, samples)
    out1, out2 = tmp_path / "o1", tmp_path / "o2"
    out1.mkdir(); out2.mkdir()
    r = subprocess.run([CLI, "-i", pb, "-v", vcf, "-n", "-d", str(out1)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ot = oracle.OracleTree(tree)
    newname = _expected_dump(parent, names, muts, dfs)
    bfs = ot.bfs_ids()
    rows = [l.split("\t") for l in open(out1 / "placement_stats.tsv").read().splitlines()]
    assert len(rows) == 40
    n_imputed = 0
    for q, S in enumerate(samples):
        S = [_as_vcf_reader_sees(e) for e in S]
        o = ot.place_sample(*(list(zip(*S)) if S else ([], [], [], [])))
        assert rows[q][0] == snames[q] and int(rows[q][1]) == o["score"] and int(rows[q][2]) == o["num_best"]
        imp = ot.imputed_at_node(*(list(zip(*S)) if S else ([], [], [], [])), o["best_j"])
        n_imputed += len(imp)
        want4 = ";".join(f"{p}:{pbf.NUC.get(n, 'N') if n != 7 else 'V'}" for p, n in imp)
        assert (rows[q][3] if len(rows[q]) > 3 else "") == want4, (q, rows[q], want4)
        assert f"Sample name: {snames[q]}\tParsimony score: {o['score']}\tNumber of parsimony-optimal placements: {o['num_best']}" in r.stderr
    assert n_imputed > 5
    # -s: rows ordered by (score, number of optimal placements), stable (usher_common.cpp:276-283)
    out3 = tmp_path / "o3"
    out3.mkdir()
    r3 = subprocess.run([CLI, "-i", pb, "-v", vcf, "-n", "-s", "-d", str(out3)], capture_output=True, text=True)
    assert r3.returncode == 0, r3.stderr
    rows3 = [l.split("\t") for l in open(out3 / "placement_stats.tsv").read().splitlines()]
    want3 = sorted(rows, key=lambda x: (int(x[1]), int(x[2])))
    assert [x[:3] for x in rows3] == [x[:3] for x in want3]
    r = subprocess.run([CLI, "-i", pb, "-v", vcf, "-n", "-p", "-d", str(out2)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = open(out2 / "parsimony-scores.tsv").read().splitlines()
    assert lines[0].startswith("#Sample\tTree node\tParsimony score")
    assert len(lines) == 1 + 40 * 300
    S = [_as_vcf_reader_sees(e) for e in samples[3]]
    o = ot.place_sample(*(list(zip(*S)) if S else ([], [], [], [])), per_node_scores=True)
    block = [l.split("\t") for l in lines[1 + 3 * 300: 1 + 4 * 300]]
    assert [int(b[2]) for b in block] == o["node_scores"].tolist()
    assert [b[1] for b in block] == [newname[int(i)] for i in bfs]
    assert all((b[4] == "y") == (int(b[2]) == o["score"]) for b in block)
    # last column: "N/A" for non-optimal nodes, "*" for a zero score, else the first `score` entries of
    # node_excess_mutations as par+position+mut (usher_common.cpp:555-574, Mutation::get_string)
    n_lists = 0
    for q in range(40):
        Sq = [_as_vcf_reader_sees(e) for e in samples[q]]
        cols = list(zip(*Sq)) if Sq else ([], [], [], [])
        oq = ot.place_sample(*cols, per_node_scores=True)
        blk = [l.split("\t") for l in lines[1 + q * 300: 1 + (q + 1) * 300]]
        for k, b in enumerate(blk):
            sc = int(oq["node_scores"][k])
            if sc != oq["score"]:
                assert b[5] == "N/A"
            elif sc == 0:
                assert b[5] == "*"
            else:
                want = ot.excess_at_node(*cols, k)[:sc]
                assert b[5] == ",".join(f"{pbf.NUC.get(pa, 'N')}{p}{pbf.NUC.get(mu, 'N')}" for (p, _, pa, mu) in want), (q, k)
                n_lists += 1
    assert n_lists > 40


@pytest.mark.gpu
def test_wepp_usher_devices_and_chunked_scores(tmp_path):
    """The C++ multi-GPU driver (usher_place_samples with a device list: one host thread and one handle
    per entry, contiguous sample ranges -- the per-sample loop of usher_common.cpp:386-446 sharded) with
    `--devices 0,0` (two handles, two host threads, one GPU), and the -p mode cut into chunks of three
    samples: every file byte-identical to the single-handle, single-chunk run.  Without -n the tool
    refuses (it never adds samples to the tree)."""
    rng = np.random.default_rng(101)
    tree, ref = ft.random_tree(rng, n_nodes=400, genome=150, p_root_masked=0.0)
    parent, muts = _tree_lists(tree)
    names = _names(parent)
    samples = [ft.random_sample(rng, ref, genome=150) for _ in range(53)]
    snames = [f"s{i}" for i in range(53)]
    pb, vcf = str(tmp_path / "t.pb"), str(tmp_path / "s.vcf")
    pbf.write_pb(pb, parent, names, muts, compress=False)
    pbf.write_vcf(vcf, snames, samples)

    def run(name, *extra, env=None):
        out = tmp_path / name
        out.mkdir()
        e = dict(os.environ)
        e.update(env or {})
        r = subprocess.run([CLI, "-i", pb, "-v", vcf, "-n", "-d", str(out)] + list(extra), capture_output=True, text=True, env=e)
        assert r.returncode == 0, r.stderr
        return out, r.stderr

    one, err1 = run("one")
    two, err2 = run("two", "--devices", "0,0")
    three, _ = run("three", "--devices", "0,0,0", "-S", "-r")
    one_s, _ = run("one_s", "-S", "-r")
    assert open(one / "placement_stats.tsv").read() == open(two / "placement_stats.tsv").read()
    assert open(one_s / "placement_stats.tsv").read() == open(three / "placement_stats.tsv").read()
    assert err1 == err2
    p1, _ = run("p1", "-p")
    p2, _ = run("p2", "-p", "--devices", "0,0", env={"WEPP_USHER_CHUNK_ROWS": "3"})
    p3, _ = run("p3", "-p", "-s", env={"WEPP_USHER_CHUNK_ROWS": "7"})
    p4, _ = run("p4", "-p", "-s", "--devices", "0,0,0")
    a = open(p1 / "parsimony-scores.tsv").read()
    assert len(a.splitlines()) == 1 + 53 * 400
    assert a == open(p2 / "parsimony-scores.tsv").read()
    assert open(p3 / "parsimony-scores.tsv").read() == open(p4 / "parsimony-scores.tsv").read()
    r = subprocess.run([CLI, "-i", pb, "-v", vcf, "-d", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 1 and "--no-add" in r.stderr


def test_synthetic_pb_gz_round_trip(tmp_path):
    """wepp-synth writes the generator's tree with the repo's own save_mutation_annotated_tree (gzipped like UCSC's
    public MATs) and the samples as a VCF; the loaders read both back (--load-only: no GPU) and the tree they build is
    the generated one: same node count, and -- node by node through --dump -- the same parents and mutations as
    wepp_amd.generate_tree with the same seed."""
    import json
    synth = os.path.join(ROOT, "wepp_amd", "wepp-synth")
    pb, vcf = str(tmp_path / "t.pb.gz"), str(tmp_path / "s.vcf.gz")
    r = subprocess.run([synth, "--nodes", "3000", "--seed", "9", "--pb", pb, "--samples", "40", "--vcf", vcf, "--read-len", "300"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(pb, "rb").read(2) == b"\x1f\x8b"                      # gzip magic
    rep = json.loads(subprocess.run([CLI, "-i", pb, "-v", vcf, "--load-only"], check=True, capture_output=True, text=True).stdout)
    assert rep["nodes"] == 3000 and rep["samples"] == 40
    nodes, samples = _run_dump(pb, vcf)
    g = w.generate_tree(9, 3000)
    parent, muts = _tree_lists(g.tree)
    has_child = [False] * 3000
    for p_ in parent:
        if p_ >= 0:
            has_child[p_] = True
    leaves = {f"s{i}": i for i in range(3000) if not has_child[i]}
    assert len(samples) == 40
    n_checked = 0
    for name, i in leaves.items():
        par_name, got = nodes[name]
        want = sorted([(p, r, pa, mu) for (p, r, pa, mu) in muts[i] if mu != pa], key=lambda t: t[0])
        assert got == want, name
        n_checked += 1
    assert n_checked > 1000


@pytest.mark.gpu
def test_one_flatten_for_three_devices(tmp_path):
    """The multi-GPU host loop flattens the tree ONCE and uploads the image from every device thread
    (wepp_flat_create + wepp_mat_upload; the reference re-expands the tree per sample, usher_common.cpp:339): with
    `--devices 0,0,0` on a 300 K-node tree written by the repo's own .pb.gz writer (wepp-synth) the run reports one
    full flatten, writes the same placement_stats.tsv as the single-device run, and its peak resident memory stays
    within 1.3x of it."""
    import json
    synth = os.path.join(ROOT, "wepp_amd", "wepp-synth")
    pb, vcf = str(tmp_path / "t.pb.gz"), str(tmp_path / "s.vcf")
    subprocess.run([synth, "--nodes", "300000", "--seed", "5", "--pb", pb, "--samples", "600", "--vcf", vcf, "--read-len", "400",
                    "--p-n", "0.01"], check=True, capture_output=True)

    def run(name, *extra):
        out = tmp_path / name
        out.mkdir()
        r = subprocess.run([CLI, "-i", pb, "-v", vcf, "-n", "-d", str(out), "--report"] + list(extra), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return out, json.loads(r.stdout.strip().splitlines()[-1])

    one, rep1 = run("one")
    three, rep3 = run("three", "--devices", "0,0,0")
    assert rep1["flattens"] == 1 and rep3["flattens"] == 1 and rep3["devices"] == 3
    assert rep1["nodes"] == 300000 and rep1["samples"] == 600
    assert open(one / "placement_stats.tsv").read() == open(three / "placement_stats.tsv").read()
    assert rep3["peak_rss_mb"] <= 1.3 * rep1["peak_rss_mb"], (rep1, rep3)
