# Audio::resample 96 -> 48 kHz (config 5's stage, k_resample_ols3): kernel time, SQ counters (what a wavefront's time goes into) and HBM traffic,
# separate PMC passes; summaries in gpurun_out/resample_sq_counters.txt / resample_hbm_counters.txt / resample_kernel_stats.csv
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_resample.py 96000:48000"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rs -- $CMD > $R/gpurun_out/resample_prof.log 2>/dev/null
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
  "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/prof_rs_sq$i -- $CMD > /dev/null 2> $R/gpurun_out/prof_rs_sq$i.err || echo "pass $i failed"
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_rs_fetch -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_rs_write -- $CMD > /dev/null 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/prof_rs_sq1 gpurun_out/prof_rs_sq2 gpurun_out/prof_rs_sq3 gpurun_out/prof_rs_sq4 > gpurun_out/resample_sq_counters.txt
python tools/pmc_summary.py gpurun_out/prof_rs_fetch gpurun_out/prof_rs_write > gpurun_out/resample_hbm_counters.txt
find gpurun_out/prof_rs -name "*kernel_stats.csv" -exec cp {} gpurun_out/resample_kernel_stats.csv \;
rm -rf gpurun_out/prof_rs gpurun_out/prof_rs_sq? gpurun_out/prof_rs_fetch gpurun_out/prof_rs_write
grep -A3 "k_resample_ols" gpurun_out/resample_sq_counters.txt | cut -c1-400
grep -A1 "k_resample_ols" gpurun_out/resample_hbm_counters.txt
cut -c1-140 gpurun_out/resample_kernel_stats.csv | head -5
