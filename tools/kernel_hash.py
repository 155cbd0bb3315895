"""Hash of the cost-volume KERNEL source: depth-estimation_amd/csrc/ssd_cost_volume.hip up to its `// C ABI` marker (the kernels and
their launchers; the host pipelines behind the marker do not change what a launch reads or writes), with // comments, blank lines
and leading / trailing whitespace removed.  profiles/traffic_*.json carries it; tests/test_abi_cpu.py and bench.py compare it, so PMC
traffic measured on another version of the kernels cannot be reported (a forgotten DFE_CV_KERNEL_REV bump)."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "depth-estimation_amd", "csrc", "ssd_cost_volume.hip")


def kernel_source_hash(path=SRC):
    text = open(path).read()
    cut = text.find("\n// C ABI")
    if cut < 0:
        raise ValueError("%s has no `// C ABI` marker" % path)
    lines = []
    for line in text[:cut].splitlines():
        line = re.sub(r"//.*$", "", line).strip()
        if line:
            lines.append(line)
    return hashlib.sha256("\n".join(lines).encode()).hexdigest()


if __name__ == "__main__":
    print(kernel_source_hash())
