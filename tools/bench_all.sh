#!/bin/bash
# every bench workload, one JSON line each, into gpurun_out/<tag>_bench_<workload>.log     usage: tools/bench_all.sh <tag> [workloads...]
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=${1:-bench}; shift
wls=${@:-vga 720p 1080p vga-luma vga-pyramid 720p-pyramid 1080p-pyramid 4k-pyramid 4k-pyramid-f16 1080p-pyramid-f16 vga-pyramid-f16 vga-pyramid-learned 1080p-pyramid-learned 720p-radial vga-f16 1080p-f16 4k-f16 version2-vga version2-vga-mfma version2-180p time-matching vga-learned vga-learned-thr 720p-learned}
mkdir -p gpurun_out
for w in $wls; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/${tag}_bench_$w.log 2>&1
  rc=$?
  if [ $rc -ge 124 ]; then echo "$w killed"; exit $rc; fi
  python3 - "$w" gpurun_out/${tag}_bench_$w.log <<'PY'
import json, sys
w, p = sys.argv[1:3]
try:
    j = json.loads(open(p).read().strip().splitlines()[-1])
    r = j["roofline"]
    b = j.get("roofline_build_only", {})
    print("%-22s %8.4f ms  %9.1f Mpix/s  frac %.4f  kernel %s %s  build %s" % (w, j["ms_per_step"], j["value"], r["frac"] or 0, r.get("kernel_ms"), (r.get("kernel") or "")[:40], b.get("frac")))
except Exception as e:
    print(w, "FAILED", e, open(p).read()[-300:])
PY
done
