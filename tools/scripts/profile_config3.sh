: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python $R/tools/bench_config3.py > $R/gpurun_out/prof_c3.json 2> $R/gpurun_out/prof_c3.err
cd $R
find gpurun_out/prof_c3 -name "*kernel_stats.csv" -exec cp {} gpurun_out/config3_kernel_stats.csv \;
rm -rf gpurun_out/prof_c3
cut -c1-150 gpurun_out/config3_kernel_stats.csv
