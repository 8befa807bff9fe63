# HBM traffic (PMC, separate passes) of every frame processor at the config 3 size; summary per kernel in gpurun_out/config3_hbm_summary.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_c3_fetch -- python $R/tools/bench_config3.py > /dev/null 2> $R/gpurun_out/prof_c3_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_c3_write -- python $R/tools/bench_config3.py > /dev/null 2> $R/gpurun_out/prof_c3_write.err
cd $R
python tools/pmc_summary.py gpurun_out/prof_c3_fetch gpurun_out/prof_c3_write > gpurun_out/config3_hbm_summary.txt
rm -rf gpurun_out/prof_c3_fetch gpurun_out/prof_c3_write
head -c 6000 gpurun_out/config3_hbm_summary.txt
