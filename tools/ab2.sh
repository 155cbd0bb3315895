#!/bin/bash
# Tuning only: A/B of side libraries in ONE gpurun call (boxes differ by +-5 %):  tools/ab2.sh "<tune_sweep args>" lib [lib...]
# (lib = path of a libdfe build, "-" = the product library); two interleaved passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
args=$1; shift
for pass in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset DFE_LIB; tag=product; else export DFE_LIB=$lib; tag=$(basename $lib .so); fi
    ROUNDS=${ROUNDS:-3} timeout -k 10 150 python tools/tune_sweep.py $args 2>&1 | grep -v amdgpu.ids | sed "s/^/[$tag] /"
  done
done
