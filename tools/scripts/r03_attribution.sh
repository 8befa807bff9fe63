#!/bin/bash
# Round 3, stall attribution of the dft 2048 pair on the bench shape:
#   PMC passes (instruction fetch / I-cache, LDS, instruction classes, wave-time split, clock) -> gpurun_out/r03_attr_pmc.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
  "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_STALL" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
  "GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" ; do
  i=$((i+1))
  echo "pass $i" >> $R/gpurun_out/r03_attr.progress
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/r03_attr_$i -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu --no-configs > /dev/null 2> $R/gpurun_out/r03_attr_$i.err || echo "pass $i failed" >> $R/gpurun_out/r03_attr.progress
done
cd $R
python3 tools/pmc_summary.py gpurun_out/r03_attr_1 gpurun_out/r03_attr_2 gpurun_out/r03_attr_3 gpurun_out/r03_attr_4 gpurun_out/r03_attr_5 gpurun_out/r03_attr_6 > gpurun_out/r03_attr_pmc.txt
rm -rf gpurun_out/r03_attr_[1-6]
