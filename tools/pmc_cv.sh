#!/bin/bash
# PMC passes over the cost-volume kernel (counters in their own runs, kernel-trace only).
# usage: tools/pmc_cv.sh <tag> [workload] [pair|0..3]   (see tools/prof_cv.py)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=${1:-pmc}; wl=${2:-vga}; what=${3:-pair}
out=gpurun_out/$tag; mkdir -p $out
pass() { # pass <name> <counters...>
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 tools/prof_cv.py $wl 4 $what > $out/$name.log 2>&1
  local rc=$?
  echo "pass $name rc=$rc"
  if [ $rc -ge 124 ]; then exit $rc; fi
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass sq2 SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT
pass sq3 SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_IFETCH
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
pass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
python3 tools/pmc_summary.py $out > $out/summary.txt; cat $out/summary.txt
