#!/bin/bash
# Tuning only: the learned-filter workloads with variants of the batched convolution, interleaved in one call.
#   usage: tools/ab_conv.sh            (VAR = option's environment name, VALS = values; "auto" = unset)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
VAR=${VAR:-DFE_CONV_TPB}
for pass in 1 2; do
  for nv in ${VALS:-1 2 4 auto}; do
    if [ $nv = auto ]; then unset $VAR; else export $VAR=$nv; fi
    for w in ${WORKLOADS:-vga-pyramid-learned 1080p-pyramid-learned version2-vga time-matching}; do
      printf "[%s %4s] %-24s " $VAR $nv $w
      timeout -k 10 200 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('stage_ms'))" || exit 1
    done
  done
done
