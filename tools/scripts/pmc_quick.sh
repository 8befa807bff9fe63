#!/bin/bash
# two SQ passes over one kernel variant (wave-time breakdown + LDS): tools/scripts/pmc_quick.sh <tag> <run_variant.py args...>
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INST_LEVEL_LDS" ; do
  i=$((i+1))
  echo "pass $i" >> $R/gpurun_out/pmcq_${TAG}.progress
  timeout -k 10 240 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmcq_${TAG}_$i -- python3 $R/tools/run_variant.py "$@" > /dev/null 2> $R/gpurun_out/pmcq_${TAG}_$i.err || echo "pass $i failed"
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmcq_${TAG}_1 gpurun_out/pmcq_${TAG}_2 > gpurun_out/pmcq_${TAG}.txt
rm -rf gpurun_out/pmcq_${TAG}_[1-2]
grep -A1 "analyze\|synthesize" gpurun_out/pmcq_${TAG}.txt
