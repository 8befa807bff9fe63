# round 5: the dft 1024 / 512 kernels (pv_kernels_v3.h): configurations A/B, kernel stats and SQ counters of the product configuration
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd $R
V3_CFGS="1024 256 1024 0 1 2;512 128 512 0 1 2" bash tools/scripts/v3_variants.sh > /dev/null
cp gpurun_out/v3_variants.txt gpurun_out/r05_v3_variants.txt
cd /tmp && export TMPDIR=/tmp
for cfg in "1024 256 1024" "512 128 512"; do
	set -- $cfg
	rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python $R/bench.py --window $1 --hop $2 --dft $3 --steps 100 --warmup 10 --no-cpu --no-configs > $R/gpurun_out/r05_dft$3_bench_profiled.json 2> /dev/null
	find $R/gpurun_out/prof_kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r05_dft$3_kernel_stats.csv \;
	rm -rf $R/gpurun_out/prof_kt
	cd $R && bash tools/scripts/pmc_shape2.sh $1 $2 $3 0 r05_dft$3 > /dev/null && mv gpurun_out/sq_r05_dft$3.txt gpurun_out/r05_dft$3_sq_counters.txt; cd /tmp
	python $R/bench.py --window $1 --hop $2 --dft $3 --steps 20 --warmup 5 --no-cpu --no-configs > $R/gpurun_out/r05_dft$3_bench.json 2>/dev/null
done
cat $R/gpurun_out/r05_v3_variants.txt; head -5 $R/gpurun_out/r05_dft1024_kernel_stats.csv
