#!/bin/bash
# PMC passes over one kernel variant:  tools/scripts/pmc_variants.sh <tag> <run_variant.py args...>   -> gpurun_out/pmc_<tag>.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_BUSY_CYCLES" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
  "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL" \
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/tools/run_variant.py "$@" > /dev/null 2> $R/gpurun_out/pmc_${TAG}_$i.err || echo "pass $i failed"
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_1 gpurun_out/pmc_${TAG}_2 gpurun_out/pmc_${TAG}_3 gpurun_out/pmc_${TAG}_4 gpurun_out/pmc_${TAG}_5 > gpurun_out/pmc_${TAG}.txt
rm -rf gpurun_out/pmc_${TAG}_[1-5]
grep -A1 "analyze\|synthesize" gpurun_out/pmc_${TAG}.txt
