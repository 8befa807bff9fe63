: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for so in $R/tools/ubench/variants/libflanhip_*.so; do name=$(basename $so .so); for a in "--dft 4096" "--dft 4096 --hop 128"; do
  FLAN_AMD_LIB=$so timeout -k 10 120 python $R/bench.py --no-cpu --no-configs --steps 200 --warmup 20 $a 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$name', '$a', round(d['value']/1e6,1), d['ms_per_step'], k['k_analyze'], k['k_synthesize'])"; done; done; done
