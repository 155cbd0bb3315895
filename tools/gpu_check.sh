#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench, rocprofv3 kernel-trace of the bench, PMC passes, other workloads.
# Stops at the first step that is killed / times out (rc >= 124); ordinary test failures do not stop it.
# usage: tools/gpu_check.sh [tag]     (tag names the files under gpurun_out/, e.g. r02_f)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=${1:-chk}
mkdir -p gpurun_out
export TMPDIR=/tmp
step() {  # step <name> <timeout> <cmd...>
  local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/${tag}_steps.log
  timeout -k 10 "$to" "$@" > "gpurun_out/${tag}_$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/${tag}_steps.log
  tail -n 6 "gpurun_out/${tag}_$name.log" | cut -c1-400
  if [ $rc -ge 124 ]; then echo "step $name killed (rc=$rc): stopping"; exit $rc; fi
  return 0
}
: > gpurun_out/${tag}_steps.log
rocminfo 2>/dev/null | grep -m3 -E "gfx950|Compute Unit|Marketing" > gpurun_out/${tag}_device.log
nproc >> gpurun_out/${tag}_device.log
step pytest_gpu 900 python -m pytest tests -m gpu -q -x
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step bench_vga 600 python bench.py
step prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline
find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} gpurun_out/${tag}_bench_vga_kernel_stats.csv
for w in 720p 1080p vga-luma vga-pyramid 720p-pyramid 1080p-pyramid 720p-radial vga-f16 1080p-f16 4k-f16; do
  step bench_$w 300 python bench.py --workload $w --no-cpu-baseline
done
bash tools/pmc_cv.sh ${tag}_pmc_pair vga pair > gpurun_out/${tag}_pmc_pair.log 2>&1 || exit $?
bash tools/pmc_cv.sh ${tag}_pmc_build vga 0 > gpurun_out/${tag}_pmc_build.log 2>&1 || exit $?
tail -3 gpurun_out/${tag}_pmc_build.log
