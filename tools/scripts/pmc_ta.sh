#!/bin/bash
# TA / TCP / TD counters of one kernel variant:  tools/scripts/pmc_ta.sh <tag> <run_variant.py args...>  -> gpurun_out/pmcta_<tag>.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
  "TA_TA_BUSY_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
  "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmcta_${TAG}_$i -- python3 $R/tools/run_variant.py "$@" > /dev/null 2> $R/gpurun_out/pmcta_${TAG}_$i.err || echo "pass $i failed"
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmcta_${TAG}_1 gpurun_out/pmcta_${TAG}_2 gpurun_out/pmcta_${TAG}_3 > gpurun_out/pmcta_${TAG}.txt
rm -rf gpurun_out/pmcta_${TAG}_[1-3]
grep -A1 "analyze\|synthesize" gpurun_out/pmcta_${TAG}.txt
