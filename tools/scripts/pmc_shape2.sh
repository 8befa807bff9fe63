# SQ counters of the conversion kernels at a given shape and kernel configuration: bash tools/scripts/pmc_shape2.sh WINDOW HOP DFT VARIANT TAG
# (8 ch x 60 s; summary in gpurun_out/sq_TAG.txt)
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --window $1 --hop $2 --dft $3 --no-cpu --no-configs --steps 5 --warmup 2 --preroll-ms 0 --kernel-variant 4=$4"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/prof_sh1 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/prof_sh2 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/prof_sh3 -- $CMD > /dev/null 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/prof_sh1 gpurun_out/prof_sh2 gpurun_out/prof_sh3 > gpurun_out/sq_$5.txt
rm -rf gpurun_out/prof_sh1 gpurun_out/prof_sh2 gpurun_out/prof_sh3
grep -A1 "k_analyze\|k_synthesize" gpurun_out/sq_$5.txt | cut -c1-400
