#!/usr/bin/env python3
"""Kernels and memory copies of the LAST wepp_place_batch of a rocprofv3 --kernel-trace --memory-copy-trace run, in start order:
python tools/diag/timeline.py <output dir> [rows]"""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-50:]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
rows = rows[-n:]
t0 = rows[0][0]
for s, e, name in rows:
    print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name))
