"""Summarises rocprofv3 --pmc csv output per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:70]
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
import re
filt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else re.compile("tiled|ref_kernel|tail|rowimg|finalize|border")
for k, cs in rows.items():
    if not filt.search(k): continue
    print("kernel:", k)
    for c in sorted(cs):
        v = cs[c]
        print("  %-32s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
