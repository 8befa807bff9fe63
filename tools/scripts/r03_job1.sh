: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 120 ./tools/ubench/issue_model > gpurun_out/r03_issue_model.txt 2>&1
echo ubench done
timeout -k 10 300 python tools/data_dependence.py > gpurun_out/r03_data_dependence.txt 2>&1
echo datadep done
FLAN_AMD_LIB=tools/ubench/libflanhip_clock.so timeout -k 10 200 python tools/stamp_report.py --ana 4 --fused > gpurun_out/r03_clock.txt 2>&1
echo clock done
timeout -k 10 900 bash tools/scripts/r03_attribution.sh
echo pmc done
