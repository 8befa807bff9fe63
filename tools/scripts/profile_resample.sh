# per-kernel times of Audio::resample for the rate pairs given (default: the slow ones)
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
PAIRS=${@:-"192000:48000 192000:44100 48000:44100 44100:48001"}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rs -- python $R/tools/bench_resample.py $PAIRS > $R/gpurun_out/resample_prof.log 2>/dev/null
cd $R
find gpurun_out/prof_rs -name "*kernel_stats.csv" -exec cp {} gpurun_out/resample_kernel_stats.csv \;
rm -rf gpurun_out/prof_rs
cut -c1-160 gpurun_out/resample_kernel_stats.csv | head -14
cat gpurun_out/resample_prof.log
