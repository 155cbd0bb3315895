#!/bin/bash
# Regenerates profiles/traffic_vga.json on the GPU box: the PMC passes of the fused pair step and of the build alone (tools/pmc_cv.sh),
# then tools/make_traffic.py, which tags the file with the kernel revision AND the hash of ssd_cost_volume.hip.  The refreshed file and
# the two summaries are left under gpurun_out/<tag>_traffic/ (gpurun merges them back; copy them into profiles/).
# usage (from the build container):  gpurun -- 'bash tools/refresh_traffic.sh r03_x'
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=${1:-traffic}
bash tools/pmc_cv.sh ${tag}_pmc_pair vga pair > gpurun_out/${tag}_pmc_pair.log 2>&1 || exit $?
bash tools/pmc_cv.sh ${tag}_pmc_build vga 0 > gpurun_out/${tag}_pmc_build.log 2>&1 || exit $?
rev=$(python3 -c "import depth_estimation_amd as d; print(d.lib().dfe_kernel_revision().decode())")
python3 tools/make_traffic.py vga gpurun_out/${tag}_pmc_pair/summary.txt gpurun_out/${tag}_pmc_build/summary.txt "$rev" || exit $?
mkdir -p gpurun_out/${tag}_traffic
cp profiles/traffic_vga.json gpurun_out/${tag}_traffic/traffic_vga.json
cp gpurun_out/${tag}_pmc_pair/summary.txt gpurun_out/${tag}_traffic/${tag}_pmc_vga_fused_pipeline.txt
cp gpurun_out/${tag}_pmc_build/summary.txt gpurun_out/${tag}_traffic/${tag}_pmc_vga_build_only.txt
