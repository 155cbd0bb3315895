#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench, and a rocprofv3 kernel-trace of the bench.
# Stops at the first step that is killed / times out (rc >= 124); ordinary test failures do not stop it.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
step() {  # step <name> <timeout> <cmd...>
  local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/steps.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/steps.log
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -ge 124 ]; then echo "step $name killed (rc=$rc): stopping"; exit $rc; fi
  return 0
}
: > gpurun_out/steps.log
rocminfo 2>/dev/null | grep -m3 -E "gfx950|Compute Unit|Marketing" > gpurun_out/device.log
nproc >> gpurun_out/device.log
which luajit lua th >> gpurun_out/device.log 2>&1
step pytest_gpu 900 python -m pytest tests -m gpu -q -x
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step bench 600 python bench.py
step prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -20
