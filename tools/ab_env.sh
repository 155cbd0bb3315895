#!/bin/bash
# Tuning only: bench workloads with and without an environment setting, interleaved, in ONE gpurun call
# usage: tools/ab_env.sh "<workloads>" VAR=VALUE [passes]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
wls=$1; kv=$2; passes=${3:-2}
for pass in $(seq $passes); do
  for mode in default "$kv"; do
    for w in $wls; do
      if [ "$mode" = default ]; then timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline > /tmp/ab_$w.log 2>&1
      else timeout -k 10 200 env "$kv" python bench.py --workload $w --no-cpu-baseline > /tmp/ab_$w.log 2>&1; fi
      [ $? -ne 0 ] && { echo "$mode $w failed"; tail -3 /tmp/ab_$w.log; exit 1; }
      python3 -c "
import json
j=json.loads(open('/tmp/ab_$w.log').read().strip().splitlines()[-1]); r=j['roofline']; b=j.get('roofline_build_only') or {}
print('[%-20s] %-22s step %.4f ms  kernel %s (%s)  build %s' % ('$mode','$w',j['ms_per_step'],r.get('kernel_ms'),r.get('frac'),b.get('kernel_ms')))"
    done
  done
done
