#!/bin/bash
# Tuning only: bench workloads with the product library and with side libraries, interleaved, in ONE gpurun call (boxes differ by a few %)
# usage: tools/ab_bench.sh "<workloads>" lib [lib...]      (lib = path of a libdfe build, "-" = the product library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
wls=$1; shift
for pass in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset DFE_LIB; tag=product; else export DFE_LIB=$PWD/$lib; tag=$(basename $lib .so); fi
    for w in $wls; do
      timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline > /tmp/ab_$w.log 2>&1 || { echo "$tag $w failed"; tail -3 /tmp/ab_$w.log; exit 1; }
      python3 -c "
import json
j=json.loads(open('/tmp/ab_$w.log').read().strip().splitlines()[-1]); r=j['roofline']; b=j.get('roofline_build_only',{})
print('[%-16s] %-8s step %.4f ms  kernel %.4f (%.4f)  build %s' % ('$tag','$w',j['ms_per_step'],r.get('kernel_ms') or 0,r['frac'] or 0,b.get('kernel_ms')))"
    done
  done
done
