: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
R=$GRAFT_REPO_ROOT
for name in base nopk nopk_topdown; do
  so=$R/tools/ubench/variants/libflanhip_$name.so
  for args in "--dft 4096" "--dft 4096 --hop 128" "--dft 1024" "--dft 8192" "--hop 128" "--hop 256"; do
    FLAN_AMD_LIB=$so timeout -k 10 120 python $R/bench.py --no-cpu --no-configs $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$name', '$args', round(d['value']/1e6,1), d['ms_per_step'], k.get('k_analyze'), k.get('k_synthesize'))"
  done
done
