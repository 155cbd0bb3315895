#!/bin/bash
# Tuning only: builds side libraries of the cost-volume kernel with parts removed (-DDFE_ABLATE=n) and times them.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for n in "$@"; do
  DFE_LIB=tools/ubench/libdfe_abl$n.so timeout -k 10 120 python tools/tune_cv.py vga 2>&1 | grep "tyq=[045]" | sed "s/^/ABL=$n /"
done
