#!/bin/bash
# A/B of the dft 1024 / 512 kernel configurations (conversions.hip: FLANHIP_V3_CFGS_9 / _8) on one box: bench.py per configuration index
mkdir -p gpurun_out
out=gpurun_out/v3_variants.txt
: > $out
IFS=';' read -ra V3_LIST <<< "${V3_CFGS:-1024 256 1024 0 1 2;512 128 512 0 1 2}"      # "W H D variant..." separated by ;
for cfg in "${V3_LIST[@]}"; do
	set -- $cfg; W=$1; H=$2; D=$3; shift 3
	for v in "$@"; do
		python bench.py --window $W --hop $H --dft $D --no-cpu --no-configs --steps 20 --warmup 5 --kernel-variant 4=$v > gpurun_out/v3_tmp.json 2> gpurun_out/v3_tmp.err
		python - "$D" "$v" >> $out <<'PY'
import json,sys
d=json.loads(open("gpurun_out/v3_tmp.json").read().strip().splitlines()[-1])
k=d.get("kernel_ms",{})
print("dft %s variant %s: %.1f M frames/s  step %.4f ms  analyze %.4f synth %.4f fixup %.4f prepass %.4f" % (sys.argv[1], sys.argv[2], d["value"]/1e6, d["ms_per_step"], k.get("k_analyze",0), k.get("k_synthesize",0), k.get("k_ola_fixup",0), k.get("prepass",0)))
PY
	done
done
cat $out
