# per-kernel times of the stretch stage (tools/bench_stretch.py) under rocprofv3; summary in gpurun_out/mtc_kernel_stats.csv
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT

cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mtc -- python $R/tools/bench_stretch.py --rounds 3 > $R/gpurun_out/prof_mtc.json 2> $R/gpurun_out/prof_mtc.err
cd $R
find gpurun_out/prof_mtc -name "*kernel_stats.csv" -exec cp {} gpurun_out/mtc_kernel_stats.csv \;
rm -rf gpurun_out/prof_mtc
cut -c1-160 gpurun_out/mtc_kernel_stats.csv | head -20
