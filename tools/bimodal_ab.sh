#!/bin/bash
# 1080p spread, lever test: N fresh processes per library, interleaved.  usage: tools/bimodal_ab.sh <tag> <n> lib [lib...]   ("-" = the product library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=$1; n=$2; shift 2
mkdir -p gpurun_out/$tag
for i in $(seq 1 $n); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset DFE_LIB; name=product; else export DFE_LIB=$PWD/$lib; name=$(basename $lib .so); fi
    timeout -k 10 200 python bench.py --workload ${WL:-1080p} --no-cpu-baseline --steps 60 --warmup 10 > gpurun_out/$tag/${name}_$i.log 2>/dev/null
    python3 -c "
import json,sys
j=json.loads(open('gpurun_out/$tag/${name}_$i.log').read().strip().splitlines()[-1])
print('%-18s run %2d  step %.4f ms  fused %.4f  build %.4f' % ('$name', $i, j['ms_per_step'], j['roofline']['kernel_ms'], j.get('roofline_build_only',{}).get('kernel_ms',0)))"
  done
done
