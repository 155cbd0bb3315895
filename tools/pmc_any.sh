#!/bin/bash
# PMC passes (counters in their own runs, kernel-trace only) + kernel stats over ANY python command.
# usage: tools/pmc_any.sh <tag> <kernel-name filter (regex)> <python args...>      e.g. tools/pmc_any.sh r04_fm_k32 feat_matching tools/prof_fm.py k32 6
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=$1; filt=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 "$@" > $out/stats.log 2>&1 || { echo "stats rc=$?"; tail -5 $out/stats.log; exit 1; }
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $out/kernel_stats.csv
pass() { # pass <name> <counters...>
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 "${CMD[@]}" > $out/$name.log 2>&1
  local rc=$?
  echo "pass $name rc=$rc"
  if [ $rc -ge 124 ]; then exit $rc; fi
}
CMD=("$@")
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass sq2 SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT
pass sq3 SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_IFETCH
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
pass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
python3 tools/pmc_summary.py $out "$filt" > $out/summary.txt
{ echo "# $tag: python3 $*"; grep -h "win " $out/stats.log; echo "## kernel stats (rocprofv3 --kernel-trace --stats)"; python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-100s calls %5s avg %9.1f us min %9.1f max %9.1f  %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
echo "## PMC (mean per dispatch)"; cat $out/summary.txt; } > gpurun_out/${tag}.txt
cat gpurun_out/${tag}.txt
