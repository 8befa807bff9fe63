# kernel-trace summaries of the secondary shapes (dft 4096 = the API default size; config 4 whole on one GPU)
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s1 -- python $R/bench.py --no-cpu --no-configs --dft 4096 > $R/gpurun_out/shape_dft4096.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s2 -- python $R/bench.py --no-cpu --no-configs --channels 64 --seconds 600 --steps 5 --warmup 1 > $R/gpurun_out/shape_config4.json 2>/dev/null
cd $R
find gpurun_out/prof_s1 -name "*kernel_stats.csv" -exec cp {} gpurun_out/shape_dft4096_kernel_stats.csv \;
find gpurun_out/prof_s2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/shape_config4_kernel_stats.csv \;
rm -rf gpurun_out/prof_s1 gpurun_out/prof_s2
head -6 gpurun_out/shape_dft4096_kernel_stats.csv | cut -c1-150
head -6 gpurun_out/shape_config4_kernel_stats.csv | cut -c1-150
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s3 -- python $R/bench.py --no-cpu --no-configs --dft 4096 --hop 128 > $R/gpurun_out/shape_api_default.json 2>/dev/null
cd $R
find gpurun_out/prof_s3 -name "*kernel_stats.csv" -exec cp {} gpurun_out/shape_api_default_kernel_stats.csv \;
rm -rf gpurun_out/prof_s3
head -6 gpurun_out/shape_api_default_kernel_stats.csv | cut -c1-150
