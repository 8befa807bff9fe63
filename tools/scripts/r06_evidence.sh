# Round 6 evidence on one box (about eight minutes): the headline's kernel trace, SQ / instruction-class / HBM counters (tools/scripts/profile_bench.sh), the
# driver-style bench line, config 3 dispatch by dispatch, kernel stats and counters of dft 4096 (API default and hop 512), 8192, 16384, 512, 256, 32768.
# Everything lands in gpurun_out/r06/ ; tools/collect_r06.py copies and stamps it into profiles/.
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R

bash tools/scripts/profile_bench.sh > $O/profile_bench.log 2>&1
for f in prof_kt.json sq_summary.txt inst_classes.txt hbm_summary.txt kernel_stats.csv; do cp gpurun_out/$f $O/ 2>/dev/null; done
echo "headline profile done" > $O/progress.txt
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench done" >> $O/progress.txt
bash tools/scripts/timeline_config3.sh > $O/config3_timeline.txt 2>&1
echo "config 3 done" >> $O/progress.txt
for cfg in "2048 128 4096 api_default" "2048 512 4096 dft4096_hop512" "8192 2048 8192 dft8192" "4096 1024 16384 dft16384" "512 128 512 dft512" "256 64 256 dft256" "4096 1024 32768 big"; do
  set -- $cfg
  bash tools/scripts/pmc_shape_full.sh $1 $2 $3 r06_$4 > $O/pmc_$4.log 2>&1
  cp gpurun_out/counters_r06_$4.txt gpurun_out/kernel_stats_r06_$4.csv $O/ 2>/dev/null
  echo "$4 done" >> $O/progress.txt
done
SHAPES="2048 512 2048;2048 512 4096;2048 128 4096;4096 1024 4096;1024 256 1024;512 128 512;256 64 256;8192 2048 8192;4096 1024 8192;2048 512 8192;4096 1024 16384;16384 4096 16384;8192 2048 16384;2048 512 2998;4096 1024 32768;32768 8192 32768;4096 1024 20000;4096 1024 44100" TAG=r06_all bash tools/scripts/time_shapes.sh > $O/time_shapes.log 2>&1
cp gpurun_out/shapes_r06_all.txt $O/
echo "all done" >> $O/progress.txt
tail -3 $O/bench.err; tail -c 400 $O/bench.json; cat $O/config3_timeline.txt | tail -12
