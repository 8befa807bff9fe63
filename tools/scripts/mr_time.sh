# 8 ch x 60 s round trips at dft sizes served by the mixed-radix kernels (and one by the direct sums)
for cfg in "2048 512 3000" "4096 1024 16384" "2048 512 6000" "2048 512 12000" "1024 256 2002"; do set -- $cfg; timeout -k 10 120 python bench.py --window $1 --hop $2 --dft $3 --no-cpu --no-configs --steps 10 --warmup 3 > gpurun_out/bench_any_$3.json 2>gpurun_out/bench_any_$3.err; python -c "
import json,sys
d=json.loads(open(\"gpurun_out/bench_any_$3.json\").read().strip().splitlines()[-1]); print($1, $2, $3, d[\"ms_per_step\"], d[\"value\"], d.get(\"kernel_ms\"))"; done
