"""Builds parsimony.proto messages with the Python protobuf runtime (no protoc in
this image): the schema of /root/reference/parsimony.proto is declared through a
FileDescriptorProto.  Used to write .pb fixtures for the C++ loader tests."""
import gzip

import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto


def _schema():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "parsimony.proto"
    fd.package = "Parsimony"
    fd.syntax = "proto3"

    def msg(name, fields):
        m = fd.message_type.add()
        m.name = name
        for (fname, num, ftype, label, tname) in fields:
            f = m.field.add()
            f.name, f.number, f.type, f.label = fname, num, ftype, label
            if tname:
                f.type_name = tname
    O, R = _F.LABEL_OPTIONAL, _F.LABEL_REPEATED
    msg("mut", [("position", 1, _F.TYPE_INT32, O, ""), ("ref_nuc", 2, _F.TYPE_INT32, O, ""),
                ("par_nuc", 3, _F.TYPE_INT32, O, ""), ("mut_nuc", 4, _F.TYPE_INT32, R, ""),
                ("chromosome", 5, _F.TYPE_STRING, O, "")])
    msg("mutation_list", [("mutation", 1, _F.TYPE_MESSAGE, R, ".Parsimony.mut")])
    msg("condensed_node", [("node_name", 1, _F.TYPE_STRING, O, ""), ("condensed_leaves", 2, _F.TYPE_STRING, R, "")])
    msg("node_metadata", [("clade_annotations", 1, _F.TYPE_STRING, R, "")])
    msg("data", [("newick", 1, _F.TYPE_STRING, O, ""), ("node_mutations", 2, _F.TYPE_MESSAGE, R, ".Parsimony.mutation_list"),
                 ("condensed_nodes", 3, _F.TYPE_MESSAGE, R, ".Parsimony.condensed_node"),
                 ("metadata", 4, _F.TYPE_MESSAGE, R, ".Parsimony.node_metadata")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("Parsimony.data"))


Data = _schema()


def newick_and_dfs(parent, names):
    """Newick string (children in ascending id order) and the DFS pre-order of ids."""
    n = len(parent)
    kids = [[] for _ in range(n)]
    root = 0
    for i, p in enumerate(parent):
        if p < 0:
            root = i
        else:
            kids[p].append(i)
    dfs = []

    def rec(i):
        dfs.append(i)
        if not kids[i]:
            return names[i]
        return "(" + ",".join(rec(c) for c in kids[i]) + ")" + names[i] + ":0.5"
    import sys
    sys.setrecursionlimit(100000)
    return rec(root) + ";", dfs


def write_pb(path, parent, names, muts, metadata=True, compress=False, unpacked_mut_nuc=False):
    """muts[i] = list of (position, ref_mask, par_mask, mut_mask); masks one-hot
    except mut (may be ambiguous); position < 0 = masked."""
    d = Data()
    d.newick, dfs = newick_and_dfs(parent, names)
    for i in dfs:
        lst = d.node_mutations.add()
        for (p, r, pa, mu) in muts[i]:
            m = lst.mutation.add()
            m.position = int(p)
            if p >= 0:
                m.ref_nuc = int(r).bit_length() - 1
                m.par_nuc = int(pa).bit_length() - 1
                m.mut_nuc.extend(b for b in range(4) if mu & (1 << b))
            m.chromosome = "NC_045512v2"
        if metadata:
            d.metadata.add().clade_annotations.append("")
    raw = d.SerializeToString()
    (gzip.open if compress else open)(path, "wb").write(raw)
    return dfs


NUC = {1: "A", 2: "C", 4: "G", 8: "T", 15: "N", 5: "R", 10: "Y", 6: "S", 9: "W", 12: "K", 3: "M", 14: "B", 13: "D", 11: "H",
       7: "V"}   # upstream reads V back as N (its switch falls through, mutation_annotated_tree.cpp:65-71)


def write_vcf(path, sample_names, samples, compress=False):
    """samples[s] = list of (position, ref_mask, allele_mask, is_missing).  One VCF
    row per distinct position (sorted), ALT alleles collected per row."""
    rows = {}
    for s, ents in enumerate(samples):
        for (p, r, a, ms) in ents:
            rows.setdefault(p, {"ref": r, "alts": [], "gt": {}})
            row = rows[p]
            if ms and a == 15:
                row["gt"][s] = "."
            else:
                ch = NUC[a]
                if ch not in row["alts"]:
                    row["alts"].append(ch)
                row["gt"][s] = str(row["alts"].index(ch) + 1)
    lines = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(sample_names)]
    for p in sorted(rows):
        row = rows[p]
        alts = ",".join(row["alts"]) if row["alts"] else "."
        gts = "\t".join(row["gt"].get(s, "0") for s in range(len(sample_names)))
        lines.append(f"NC_045512v2\t{p}\t.\t{NUC[row['ref']]}\t{alts}\t.\t.\t.\tGT\t{gts}")
    data = ("\n".join(lines) + "\n").encode()
    (gzip.open if compress else open)(path, "wb").write(data)


def _sam_schema():
    """/root/reference/sam.proto (package Sam) declared at run time."""
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "sam.proto"
    fd.package = "Sam"
    fd.syntax = "proto3"
    O, R = _F.LABEL_OPTIONAL, _F.LABEL_REPEATED

    def msg(name, fields):
        m = fd.message_type.add()
        m.name = name
        for (fname, num, ftype, label, tname) in fields:
            f = m.field.add()
            f.name, f.number, f.type, f.label = fname, num, ftype, label
            if tname:
                f.type_name = tname
    msg("read_info", [("read", 1, _F.TYPE_STRING, O, ""), ("start_idx", 3, _F.TYPE_INT32, O, ""),
                      ("content", 6, _F.TYPE_STRING, O, ""), ("degree", 5, _F.TYPE_INT32, O, "")])
    msg("column_info", [("column_name", 1, _F.TYPE_STRING, O, ""), ("input_columns", 2, _F.TYPE_STRING, R, "")])
    msg("sam", [("reads", 1, _F.TYPE_MESSAGE, R, ".Sam.read_info"), ("reverse_columns", 2, _F.TYPE_MESSAGE, R, ".Sam.column_info")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("Sam.sam"))


Sam = _sam_schema()


def write_reads_pb(path, reads, reverse_columns=None):
    """reads = list of (name, start_idx (1-based), content over ACGTN_, degree), the message
    `wepp sam2PB` writes (sam::dump_proto, src/WEPP/sam2pb.cpp:111-147)."""
    d = Sam()
    for (name, start, content, degree) in reads:
        r = d.reads.add()
        r.read, r.start_idx, r.content, r.degree = name, int(start), content, int(degree)
    for col, inputs in (reverse_columns or {}).items():
        c = d.reverse_columns.add()
        c.column_name = col
        c.input_columns.extend(inputs)
    open(path, "wb").write(d.SerializeToString())
