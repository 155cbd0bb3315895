#!/bin/bash
# fp16-volume evidence: rocprofv3 kernel stats + SQ / LDS / HBM counter passes of the F16 row-image kernel
# usage: tools/prof16.sh <tag>
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=${1:-r03_a}
mkdir -p gpurun_out
for w in vga-f16 4k-f16; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/${tag}_bench_$w.log 2>&1 || exit $?
  tail -1 gpurun_out/${tag}_bench_$w.log | cut -c1-300
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_$w -- python3 tools/prof_cv.py $w 12 pair16 > gpurun_out/${tag}_prof_$w.log 2>&1 || exit $?
  find gpurun_out/${tag}_prof_$w -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} gpurun_out/${tag}_${w}_kernel_stats.csv
  cat gpurun_out/${tag}_${w}_kernel_stats.csv | cut -c1-200
done
bash tools/pmc_cv.sh ${tag}_pmc_vga-f16 vga-f16 pair16 > gpurun_out/${tag}_pmc_vga-f16.log 2>&1 || exit $?
tail -40 gpurun_out/${tag}_pmc_vga-f16/summary.txt
