#!/bin/bash
# Rehearsal of bench.py's multi-rank path on a ONE-GPU box: N ranks (default 2, at most 4: the box's process guard) share cuda:0, the
# process group is gloo (RCCL refuses two ranks on one GPU).  What it proves: the launch line the driver uses, barriers, the max-over-
# ranks reduction, the scale_1080p leg and the JSON line run end to end with real kernels; what it does not: RCCL itself, xGMI, scaling.
# usage: tools/rehearse_ranks.sh [N] [workload...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
n=${1:-2}; shift
export DFE_BENCH_DEVICE=0 DFE_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for w in ${@:-vga vga-pyramid 4k-f16-bands}; do
  extra="--no-gather"; [ $w = 4k-f16-bands ] && extra="--no-gather --bands $n"
  echo "== $w, $n ranks on one GPU (gloo)"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 200)) \
    bench.py --gpus $n --steps 10 --warmup 3 --workload $w --no-cpu-baseline $extra 2>&1 | grep -v "amdgpu.ids\|^W0\|^\*\*\*\|OMP_NUM_THREADS" | cut -c1-1500 || exit 1
done
