#!/bin/bash
# a few counter passes of the fused pair step for several libraries.  usage: tools/pmc_ab.sh <tag> <workload> lib [lib...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=$1; wl=$2; shift 2
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset DFE_LIB; name=product; else export DFE_LIB=$PWD/$lib; name=$(basename $lib .so); fi
  out=gpurun_out/${tag}_$name; mkdir -p $out
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 tools/prof_cv.py $wl 4 pair > $out/p$i.log 2>&1
  done
  echo "== $name"; python3 tools/pmc_summary.py $out | grep -A40 "rowimg" | grep -v "finalize" | head -30
done
