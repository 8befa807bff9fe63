# SQ and HBM counters of the above-16384 kernels at (4096, 1024, 32768), 8 ch x 60 s; summary in gpurun_out/big_counters.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --window 4096 --hop 1024 --dft 32768 --no-cpu --no-configs --steps 3 --warmup 1 --preroll-ms 0"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/prof_b1 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/prof_b2 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_b3 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_b4 -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b5 -- $CMD > /dev/null 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/prof_b1 gpurun_out/prof_b2 gpurun_out/prof_b3 gpurun_out/prof_b4 > gpurun_out/big_counters.txt
find gpurun_out/prof_b5 -name "*kernel_stats.csv" -exec cp {} gpurun_out/big_kernel_stats.csv \;
rm -rf gpurun_out/prof_b1 gpurun_out/prof_b2 gpurun_out/prof_b3 gpurun_out/prof_b4 gpurun_out/prof_b5
grep -A1 "_big\|k_phase" gpurun_out/big_counters.txt | cut -c1-400
head -8 gpurun_out/big_kernel_stats.csv
