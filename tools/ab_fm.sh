#!/bin/bash
# Tuning only: tools/prof_fm.py shapes with the product library and side libraries, interleaved in one call.
# usage: tools/ab_fm.sh "<shapes>" lib [lib...]     ("-" = the product library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
shapes=$1; shift
for pass in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset DFE_LIB; tag=product; else export DFE_LIB=$PWD/$lib; tag=$(basename $lib .so); fi
    for s in $shapes; do
      printf "[%-14s] " $tag; timeout -k 10 120 python tools/prof_fm.py $s 20 | cut -d'|' -f1 || exit 1
    done
  done
done
