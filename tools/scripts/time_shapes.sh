# 8 ch x 60 s round trips at the ( window hop dft ) shapes in $SHAPES (";"-separated), optional $VARIANT for --kernel-variant; result lines in gpurun_out/shapes_$TAG.txt
: ${GRAFT_REPO_ROOT:?} ${SHAPES:?}
TAG=${TAG:-run}
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && : > gpurun_out/shapes_$TAG.txt
IFS=';' read -ra LIST <<< "$SHAPES"
for cfg in "${LIST[@]}"; do set -- $cfg
  timeout -k 10 180 python bench.py --window $1 --hop $2 --dft $3 --no-cpu --no-configs --steps ${STEPS:-10} --warmup 3 ${VARIANT:+--kernel-variant $VARIANT} > gpurun_out/bench_shape_$1_$2_$3.json 2> gpurun_out/bench_shape_$1_$2_$3.err || { echo "$1 $2 $3 FAILED" >> gpurun_out/shapes_$TAG.txt; tail -3 gpurun_out/bench_shape_$1_$2_$3.err >> gpurun_out/shapes_$TAG.txt; continue; }
  python -c "
import json
d=json.loads(open('gpurun_out/bench_shape_$1_$2_$3.json').read().strip().splitlines()[-1]); print($1, $2, $3, d['ms_per_step'], round(d['value']/1e6,2), 'M frames/s', d['roundtrip_hbm']['frac_of_8TBs'], {k: v for k, v in d.get('kernel_ms', {}).items() if k != 'note'})" >> gpurun_out/shapes_$TAG.txt
done
cat gpurun_out/shapes_$TAG.txt
