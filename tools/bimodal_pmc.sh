#!/bin/bash
# counters of the fused 1080p kernel in several fresh processes (one rocprofv3 pass = one process = one "mode"): the kernel duration of
# each pass next to its memory-side stall / TLB counters, so that the slow mode can be read off the counters.   usage: tools/bimodal_pmc.sh <tag> [n]
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=${1:-bimodal_pmc}; n=${2:-5}
out=gpurun_out/$tag; mkdir -p $out
pass() { # pass <name> <counters...>
  local name=$1; shift
  DFE_DEBUG_ARENA=1 timeout -k 10 90 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 tools/prof_cv.py 1080p 6 pair > $out/$name.log 2>&1
  local rc=$?
  if [ $rc -ge 124 ]; then echo "pass $name killed (rc $rc)"; fi
}
for i in $(seq 1 $n); do
  # (two counters per pass: more TCC counters than that "exceed the capabilities of the hardware to collect" on this part)
  pass tcca$i TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
  pass tccb$i TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
  pass tlba$i TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum
  pass tlbb$i TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
done
python3 - $out <<'PY'
import csv, glob, os, sys, collections, re
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(d): continue
    dur, cnt = [], collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "rowimg" in r.get("Kernel_Name", ""):
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "rowimg" in r.get("Kernel_Name", ""):
                cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
    ar = re.findall(r"scratch arena (\S+)", open(d + ".log").read()) if os.path.exists(d + ".log") else []
    print("%-6s kernel us %s  arena %s" % (os.path.basename(d), " ".join("%.0f" % x for x in dur[-4:]), ar[-1] if ar else "?"))
    for k in sorted(cnt):
        print("         %-44s %.4g" % (k, sum(cnt[k][-4:]) / max(1, len(cnt[k][-4:]))))
PY
