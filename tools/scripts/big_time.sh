# 8 ch x 60 s round trips at dft sizes above 16384 (the residue-pair kernels, pv_kernels_big.h)
for cfg in "4096 1024 32768" "4096 1024 65536" "4096 1024 24576" "8192 2048 32768"; do set -- $cfg; timeout -k 10 300 python bench.py --window $1 --hop $2 --dft $3 --no-cpu --no-configs --steps 5 --warmup 2 > gpurun_out/bench_big_$3_$1.json 2>gpurun_out/bench_big_$3_$1.err; python -c "
import json,sys
d=json.loads(open(\"gpurun_out/bench_big_$3_$1.json\").read().strip().splitlines()[-1]); print($1, $2, $3, d[\"ms_per_step\"], d[\"value\"], d.get(\"kernel_ms\"))"; done
