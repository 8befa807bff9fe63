# the headline round trip at several chain lengths (seconds of audio, 8 channels: L = 2.75 frames per 7.5 s): the slope is a frame slot, the intercept the launch's fixed part
: ${GRAFT_REPO_ROOT:?}
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && : > gpurun_out/chain_sweep.txt
for sec in ${SECS:-15 30 45 60 90 120}; do
python bench.py --seconds $sec --no-cpu --no-configs --no-pcie --steps 30 --warmup 5 ${VARIANT:+--kernel-variant $VARIANT} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($sec, d['ms_per_step'], d['ms_per_step_median'], {k:v for k,v in d['kernel_ms'].items() if k!='note'})" >> gpurun_out/chain_sweep.txt
done
cat gpurun_out/chain_sweep.txt
