#!/bin/bash
# Tuning only: the flat matcher with and without the half-tile split (option fm_split: 0 off, 1 on; the r04_ag sweep also had a start stagger of v - 2 sleeps
# for v > 1, since removed), interleaved in one call, then the parity tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for pass in 1 2; do
  for sp in ${SPLITS:-auto 0 1}; do
    for s in ${SHAPES:-k32 k10 tm k32-720p}; do
      if [ $sp = auto ]; then unset DFE_FM_SPLIT; else export DFE_FM_SPLIT=$sp; fi
      printf "[split %4s] " $sp; timeout -k 10 120 python tools/prof_fm.py $s 20 2>/dev/null | cut -d'|' -f1 || exit 1
    done
  done
done
unset DFE_FM_SPLIT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "flat or matching or version2 or time_matching" 2>&1 | tail -5
