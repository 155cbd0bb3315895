#!/bin/bash
# Tuning only: the matrix-core convolution (tools/cm_probe.py) with the product library and side builds, one after the other
# usage: tools/abl_cm.sh lib...      ("-" = product)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset DFE_LIB; else export DFE_LIB=$PWD/$lib; fi
  echo "== $lib"
  timeout -k 10 200 python tools/cm_probe.py 2>&1 | grep -v amdgpu
done
