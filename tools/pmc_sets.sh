#!/bin/bash
# PMC passes with CALLER-chosen counter sets (one rocprofv3 run per quoted set, kernel trace only) over a python command.
# usage: tools/pmc_sets.sh <tag> <kernel-name regex> "<set 1>" ["<set 2>" ...] -- <python args...>
#   e.g. tools/pmc_sets.sh r05_y_mfma "conv_mfma|fmm_kernel" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY" -- bench.py --workload version2-vga-mfma --steps 20 --warmup 5 --no-cpu-baseline
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=$1; filt=$2; shift 2
sets=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do sets+=("$1"); shift; done
shift
out=gpurun_out/$tag; mkdir -p $out
n=0
for s in "${sets[@]}"; do
  n=$((n + 1))
  timeout -k 10 280 rocprofv3 --pmc $s --kernel-trace --output-format csv -d $out/set$n -- python3 "$@" > $out/set$n.log 2>&1
  rc=$?; echo "set $n ($s) rc=$rc"
  if [ $rc -ge 124 ]; then exit $rc; fi
done
python3 tools/pmc_summary.py $out "$filt" > $out/summary.txt
cat $out/summary.txt
find $out -name "*.csv" -size +2M -delete
