# SQ and HBM counters + kernel stats of the conversion kernels at a shape: bash tools/scripts/pmc_shape_full.sh WINDOW HOP DFT TAG   (8 ch x 60 s; gpurun_out/counters_TAG.txt, kernel_stats_TAG.csv)
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
TAG=$4
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --window $1 --hop $2 --dft $3 --no-cpu --no-configs --no-pcie --steps 5 --warmup 2 --preroll-ms 0"
D=$R/gpurun_out/prof_$TAG
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d ${D}_1 -- $CMD > /dev/null 2>&1
echo pass1 > $R/gpurun_out/progress_$TAG.txt
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT --output-format csv -d ${D}_2 -- $CMD > /dev/null 2>&1
echo pass2 >> $R/gpurun_out/progress_$TAG.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${D}_3 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d ${D}_4 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d ${D}_5 -- $CMD > /dev/null 2>&1
echo pass5 >> $R/gpurun_out/progress_$TAG.txt
rocprofv3 --kernel-trace --stats --output-format csv -d ${D}_6 -- $CMD > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py ${D}_1 ${D}_2 ${D}_3 ${D}_4 ${D}_5 > gpurun_out/counters_$TAG.txt
find ${D}_6 -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_$TAG.csv \;
rm -rf ${D}_1 ${D}_2 ${D}_3 ${D}_4 ${D}_5 ${D}_6
grep -A1 "k_analyze\|k_synthesize" gpurun_out/counters_$TAG.txt | cut -c1-420
head -8 gpurun_out/kernel_stats_$TAG.csv
