"""Hash of everything that decides what ONE STEP of the headline workload reads from and writes to HBM:
  * depth-estimation_amd/csrc/ssd_cost_volume.hip up to its `// C ABI` marker (the kernels and their launchers; the host pipelines
    behind the marker do not change what a launch reads or writes),
  * the device part of csrc/dfe_internal.h (from the `// ---- shared by the fused cost-volume epilogue` marker on: the tile-row
    record layout DFE_REC*, CvFuseArgs, fine_epilogue, ...),
  * csrc/postops.hip from its `// ---- fused single-scale tail` marker to the `// ---- A12` marker (flow_finalize_kernel and the
    tail it shares code with),
with // comments, blank lines and leading / trailing whitespace removed.  profiles/traffic_*.json carries it; tests/test_abi_cpu.py
and bench.py compare it, so PMC traffic measured on another version of the kernels cannot be reported (a forgotten
DFE_CV_KERNEL_REV bump; ADVICE r3: the record layout had moved into dfe_internal.h, outside the hash)."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "depth-estimation_amd", "csrc")
SRC = os.path.join(CSRC, "ssd_cost_volume.hip")
PARTS = (
    ("ssd_cost_volume.hip", None, "\n// C ABI"),
    ("dfe_internal.h", "\n// ---- shared by the fused cost-volume epilogue", None),
    ("postops.hip", "\n// ---- fused single-scale tail", "\n// ---- A12"),
)


def _slice(text, name, begin, end):
    a = 0
    if begin is not None:
        a = text.find(begin)
        if a < 0:
            raise ValueError("%s has no `%s` marker" % (name, begin.strip()))
    b = len(text)
    if end is not None:
        b = text.find(end, a)
        if b < 0:
            raise ValueError("%s has no `%s` marker" % (name, end.strip()))
    return text[a:b]


def kernel_source_hash(csrc=CSRC):
    lines = []
    for name, begin, end in PARTS:
        text = _slice(open(os.path.join(csrc, name)).read(), name, begin, end)
        for line in text.splitlines():
            line = re.sub(r"//.*$", "", line).strip()
            if line:
                lines.append(line)
    return hashlib.sha256("\n".join(lines).encode()).hexdigest()


if __name__ == "__main__":
    print(kernel_source_hash())
