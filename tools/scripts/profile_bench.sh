: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python $R/bench.py --steps 100 --warmup 10 --no-cpu --no-configs --no-pcie > $R/gpurun_out/prof_kt.json 2> $R/gpurun_out/prof_kt.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/prof_sq1 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-configs --no-pcie > /dev/null 2> $R/gpurun_out/prof_sq1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $R/gpurun_out/prof_sq2 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-configs --no-pcie > /dev/null 2> $R/gpurun_out/prof_sq2.err
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $R/gpurun_out/prof_sq3 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-configs --no-pcie > /dev/null 2> $R/gpurun_out/prof_sq3.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_BUSY_CU_CYCLES --output-format csv -d $R/gpurun_out/prof_sq4 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-configs --no-pcie > /dev/null 2> $R/gpurun_out/prof_sq4.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-configs --no-pcie > /dev/null 2> $R/gpurun_out/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-configs --no-pcie > /dev/null 2> $R/gpurun_out/prof_write.err
cd $R
python tools/pmc_summary.py gpurun_out/prof_sq1 gpurun_out/prof_sq2 > gpurun_out/sq_summary.txt
python tools/pmc_summary.py gpurun_out/prof_sq3 gpurun_out/prof_sq4 > gpurun_out/inst_classes.txt
python tools/pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write > gpurun_out/hbm_summary.txt
find gpurun_out/prof_kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats.csv \;
rm -rf gpurun_out/prof_sq1 gpurun_out/prof_sq2 gpurun_out/prof_sq3 gpurun_out/prof_sq4 gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_kt
cat gpurun_out/prof_kt.json
