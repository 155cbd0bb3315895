#!/usr/bin/env python3
"""Tuning only: VGPRs / scratch / SGPR spills of every kernel in a .hip file (default: the cost-volume kernels).
usage: kres.py [file.hip] [substring] [-Dmacro=value ...]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("-D")]
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
src = args[0] if args else os.path.join(root, "depth-estimation_amd/csrc/ssd_cost_volume.hip")
pat = args[1] if len(args) > 1 else "rowimg"
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", *defs,
                      "-I" + os.path.dirname(src), "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                     capture_output=True, text=True).stderr
cur = None
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        if "error" in line: print(line)
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
    elif cur is not None:
        cur[k] = v
        if k.startswith("LDS") and pat in cur["name"]:
            n = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            n = n.replace("(anonymous namespace)::", "")
            short = re.sub(r"^void ", "", n)
            short = short[: short.rindex("(")] if "(" in short else short
            print("%-70s VGPR %s scratch %s sgpr-spill %s" % (short[-70:], cur.get("VGPRs"), cur.get("ScratchSize [bytes/lane]"), cur.get("SGPRs Spill")))
