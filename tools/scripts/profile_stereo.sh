: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_2ch -- python $R/bench.py --no-cpu --no-configs --channels 2 --steps 100 --warmup 10 > $R/gpurun_out/bench_2ch.json 2>/dev/null
cd $R
find gpurun_out/prof_2ch -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_2ch.csv \;
rm -rf gpurun_out/prof_2ch
head -8 gpurun_out/kernel_stats_2ch.csv | cut -c1-170
python3 -c "
import json
d=json.loads(open('gpurun_out/bench_2ch.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'])"
