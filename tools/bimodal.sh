#!/bin/bash
# 1080p bimodality evidence: N fresh processes of the 1080p bench step, each printing its arena address and its
# fused-kernel time; then counter passes (memory-side stalls, TLB) for whichever mode a process lands in.   usage: tools/bimodal.sh <tag> [n]
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=${1:-bimodal}; n=${2:-8}
out=gpurun_out/$tag; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TCC_[A-Z0-9_]*(STALL|BUSY|TAG_STALL|NORMAL_WRITEBACK|EA0_WRREQ_LEVEL)[A-Z0-9_]*|TCP_UTCL[0-9][A-Z0-9_]*|TCP_TCC_[A-Z_]*STALL[A-Z_]*|UTCL2[A-Z0-9_]*|TCP_PENDING_STALL_CYCLES[A-Z_]*)\b" | sort -u > $out/counters_avail.txt
wc -l $out/counters_avail.txt
for i in $(seq 1 $n); do
  DFE_DEBUG_ARENA=1 timeout -k 10 200 python bench.py --workload 1080p --no-cpu-baseline --steps 60 --warmup 10 > $out/run$i.log 2> $out/run$i.err || exit $?
  python3 - $out/run$i.log $out/run$i.err <<'PY'
import json, sys, re
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ar = re.findall(r"scratch arena (\S+) \.\. \S+ \((\d+) bytes, (not )?2 MiB", open(sys.argv[2]).read())
print("step %.4f ms  fused kernel %.4f ms  build %.4f ms   arena %s" % (j["ms_per_step"], j["roofline"]["kernel_ms"], j.get("roofline_build_only", {}).get("kernel_ms", 0), ar[-1] if ar else None))
PY
done
