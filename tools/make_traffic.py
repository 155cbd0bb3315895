#!/usr/bin/env python3
"""Writes profiles/traffic_<workload>.json from two PMC summaries (tools/pmc_cv.sh: fused pair step, unfused build), tagged with the
cost-volume kernel revision of the library that was measured.   usage: make_traffic.py WORKLOAD fused_summary.txt build_summary.txt REV"""
import hashlib, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import WORKLOADS, algorithmic_bytes

wl, fused, build, rev = sys.argv[1:5]


def counters(path, kernel_substr):
    cur, out = None, {}
    for line in open(path):
        if line.startswith("kernel:"):
            cur = line[7:].strip()
        elif cur and kernel_substr in cur:
            m = re.match(r"\s+(\S+)\s+mean (\S+)", line)
            if m:
                out[m.group(1)] = float(m.group(2))
    return out


H, W, C, k, hW, wW = WORKLOADS[wl]
f, b = counters(fused, "rowimg"), counters(build, "rowimg")
# gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; both in KB
hbm = lambda c: int(round((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024))
sys.path.insert(0, os.path.join(root, "tools"))
from kernel_hash import kernel_source_hash
src_hash = kernel_source_hash()
out = {
    "workload": wl, "kernel_rev": rev,
    # the counters belong to exactly this source of the cost-volume kernels: tests/test_abi_cpu.py and bench.py compare the hash, so
    # a kernel change without new PMC passes cannot keep reporting old traffic (a forgotten DFE_CV_KERNEL_REV bump included)
    "source_sha256": src_hash,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, mean of 4 dispatches (tools/pmc_cv.sh; summaries %s, %s)" % (os.path.basename(fused), os.path.basename(build)),
    "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled; WRITE_SIZE exact; bytes = KB * 1024",
    "FETCH_SIZE_KB_raw": f["FETCH_SIZE"], "WRITE_SIZE_KB_raw": f["WRITE_SIZE"],
    "hbm_bytes_per_launch": hbm(f),
    "algorithmic_bytes_per_launch": algorithmic_bytes(H, W, C, k, hW, wW),
    "build_only": {"FETCH_SIZE_KB_raw": b["FETCH_SIZE"], "WRITE_SIZE_KB_raw": b["WRITE_SIZE"], "hbm_bytes_per_launch": hbm(b)},
}
json.dump(out, open(os.path.join(root, "profiles", "traffic_%s.json" % wl), "w"), indent=1)
print(json.dumps(out))
