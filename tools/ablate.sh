#!/bin/bash
# Tuning only: times side libraries of the cost-volume kernels built with parts removed (-DDFE_ABLATE=n):
#   tools/ablate.sh MODE n [n...]     (MODE = 2 tiled / 3 row-image)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mode=$1; shift
for n in "$@"; do
  CV_MODE=$mode DFE_LIB=tools/ubench/libdfe_abl$n.so timeout -k 10 120 python tools/tune_cv.py vga 2>&1 | grep "tyq=5" | sed "s/^/ABL=$n /"
done
