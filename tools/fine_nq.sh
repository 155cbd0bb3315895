#!/bin/bash
# the fused finest-scale kernel's tile height: pyramid workloads at DFE_FINE_NQ = 3 / 4 / 5     usage: tools/fine_nq.sh <tag> [workloads...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=${1:-nq}; shift
wls=${@:-720p-pyramid 1080p-pyramid 4k-pyramid-f16}
for nq in 4 5 3; do
  echo "== DFE_FINE_NQ=$nq"
  DFE_FINE_NQ=$nq bash tools/bench_all.sh ${tag}_nq$nq $wls || exit $?
done
