# stage times of small inputs (CH channels x 60 s) at a few shapes, optionally with a kernel variant: CH=2 VARIANT=12=1 bash tools/scripts/stereo_shapes.sh "W H D;..."
: ${GRAFT_REPO_ROOT:?}
cd $GRAFT_REPO_ROOT
IFS=';' read -ra LIST <<< "${1:-2048 512 4096;512 128 512;256 64 256}"
for cfg in "${LIST[@]}"; do set -- $cfg
python bench.py --channels ${CH:-2} --window $1 --hop $2 --dft $3 --no-cpu --no-configs --no-pcie --steps 20 --warmup 5 ${VARIANT:+--kernel-variant $VARIANT} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 $3 ${CH:-2} ch ${VARIANT:-}', d['ms_per_step'], d['ms_per_step_median'], round(d['value']/1e6,1), {k:v for k,v in d['kernel_ms'].items() if k!='note'})"
done
