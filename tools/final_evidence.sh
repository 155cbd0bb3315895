#!/bin/bash
# The round's evidence in one call on the GPU box: device, GPU tests, smoke, the default bench line (with the CPU baseline), its
# rocprofv3 kernel stats, every other workload's bench line, kernel stats of the workloads DESIGN quotes.  Everything lands under
# gpurun_out/<tag>_*; copy what is to be judged into profiles/.   usage: gpurun --timeout 1200 -- 'bash tools/final_evidence.sh r03_z'
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=${1:-final}
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m3 -E "gfx950|Compute Unit|Marketing" > gpurun_out/${tag}_device.log
nproc >> gpurun_out/${tag}_device.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_pytest_gpu.log 2>&1; rc=$?
tail -3 gpurun_out/${tag}_pytest_gpu.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1 || { tail -5 gpurun_out/${tag}_smoke.log; exit 1; }
tail -1 gpurun_out/${tag}_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/${tag}_bench_default.log 2>&1 || { tail -5 gpurun_out/${tag}_bench_default.log; exit 1; }
tail -1 gpurun_out/${tag}_bench_default.log | cut -c1-600
bash tools/bench_all.sh $tag || exit $?
bash tools/kstats.sh $tag vga vga-f16 720p-radial vga-pyramid vga-pyramid-learned 1080p-pyramid-learned 1080p-pyramid-f16 version2-vga version2-vga-mfma vga-learned vga-luma time-matching > gpurun_out/${tag}_kstats.log 2>&1 || { tail -5 gpurun_out/${tag}_kstats.log; exit 1; }
cat gpurun_out/${tag}_kstats.log
timeout -k 10 300 python tools/time_version2.py > gpurun_out/${tag}_time_version2.log 2>&1 || { tail -5 gpurun_out/${tag}_time_version2.log; exit 1; }
grep -v amdgpu.ids gpurun_out/${tag}_time_version2.log
# the feature matcher (verdict r3 item 2): kernel stats + PMC passes at the three shapes, and the headline kernel's traffic file
for s in v2 k32 k10; do bash tools/pmc_any.sh ${tag}_fm_$s feat_matching tools/prof_fm.py $s 6 > gpurun_out/${tag}_fm_$s.log 2>&1 || { tail -3 gpurun_out/${tag}_fm_$s.log; exit 1; }; grep -E "K=|avg" gpurun_out/${tag}_fm_$s.txt; done
bash tools/refresh_traffic.sh ${tag} > gpurun_out/${tag}_refresh.log 2>&1 || { tail -3 gpurun_out/${tag}_refresh.log; exit 1; }
tail -1 gpurun_out/${tag}_refresh.log | cut -c1-200
