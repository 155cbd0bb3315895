#!/usr/bin/env python3
"""Tuning only: per-context averages of the address-translation counters of the sweep kernel, from a rocprofv3 --pmc run of tools/mode_probe.py
(PROBE_CONTIG=1: odd contexts have a physically contiguous arena).  usage: utcl_summary.py <rocprofv3 output dir> <dispatches per context>"""
import csv, glob, sys, collections
d, per = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "rowimg" in r["Kernel_Name"]]
ids = sorted({int(r["Dispatch_Id"]) for r in rows})
rank = {i: n for n, i in enumerate(ids)}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[rank[int(r["Dispatch_Id"])] // per][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({r["Counter_Name"] for r in rows})
print("%-6s " % "ctx" + " ".join("%32s" % n for n in names))
for c in sorted(acc):
    print("%-6d " % c + " ".join("%32.0f" % (sum(acc[c][n]) / max(len(acc[c][n]), 1)) for n in names))
