# bench.py's headline numbers for every library built by build_flag_variants.sh -> gpurun_out/flag_variants.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/flag_variants.txt
for so in $R/tools/ubench/variants/libflanhip_*.so; do
  name=$(basename $so .so)
  FLAN_AMD_LIB=$so timeout -k 10 120 python $R/bench.py --no-cpu --no-configs --steps 300 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$name', round(d['value']/1e6,1), d['ms_per_step'], k['k_analyze'], k['k_synthesize'])" >> $R/gpurun_out/flag_variants.txt || echo "$name failed" >> $R/gpurun_out/flag_variants.txt
done
cat $R/gpurun_out/flag_variants.txt
