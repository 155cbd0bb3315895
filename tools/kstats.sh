#!/bin/bash
# rocprofv3 kernel stats of one bench workload -> gpurun_out/<tag>_<workload>_kernel_stats.csv      usage: tools/kstats.sh <tag> <workload>...
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=$1; shift
for w in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_$w -- python3 bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_prof_$w.log 2>&1 || exit $?
  find gpurun_out/${tag}_prof_$w -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} gpurun_out/${tag}_${w}_kernel_stats.csv
  echo "== $w"; python3 - gpurun_out/${tag}_${w}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-90s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
