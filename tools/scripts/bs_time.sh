# 8 ch x 60 s round trips at dft sizes with a large prime factor (the chirp-z kernels, pv_kernels_bs.h) beside a mixed-radix neighbour;
# variant 1 = the ping-pong kernels that read their tables every frame (A/B against the ones that keep them in registers)
for v in 0 1; do
for cfg in "2048 512 2998" "2048 512 3000" "1024 256 2018" "2048 512 4094" "512 128 1006" "2048 512 5998" "4096 1024 8186"; do set -- $cfg; timeout -k 10 180 python bench.py --window $1 --hop $2 --dft $3 --no-cpu --no-configs --steps 10 --warmup 3 --kernel-variant 4=$v > gpurun_out/bench_bs_$3.json 2>gpurun_out/bench_bs_$3.err; python -c "
import json,sys
d=json.loads(open(\"gpurun_out/bench_bs_$3.json\").read().strip().splitlines()[-1]); print('variant $v:', $1, $2, $3, d[\"ms_per_step\"], d[\"value\"], d.get(\"kernel_ms\"))"; done; done
