#!/bin/bash
# Tuning only: the learned-filter workloads with the batched convolution's two tile shapes (option conv_narrow), interleaved in one call.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for pass in 1 2; do
  for nv in 0 1 auto; do
    if [ $nv = auto ]; then unset DFE_CONV_NARROW; else export DFE_CONV_NARROW=$nv; fi
    for w in ${WORKLOADS:-vga-pyramid-learned 1080p-pyramid-learned version2-vga time-matching}; do
      printf "[narrow %4s] %-24s " $nv $w
      timeout -k 10 200 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('stage_ms'))" || exit 1
    done
  done
done
