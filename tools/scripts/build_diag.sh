#!/bin/bash
# Diagnostic variants of the library (never the product, never the timed one in bench.py):
#   tools/scripts/build_diag.sh stamps      -> tools/ubench/libflanhip_stamps.so   (s_memtime stamps in the v2 kernels: tools/stamp_report.py)
#   tools/scripts/build_diag.sh ablations   -> tools/ubench/libflanhip_ablations.so (phase-ablated kernel variants 101..: tools/ab_kernels.py)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
KIND=${1:-stamps}
DEF=-DFLANHIP_STAMPS
[ "$KIND" = ablations ] && DEF=-DFLANHIP_ABLATIONS
[ "$KIND" = clock ] && DEF="-DFLANHIP_STAMPS -DFLANHIP_STAMPS_CLOCK_ONLY"   # the product's frame loop, only the wavefront's life stamped in both clocks (tools/stamp_report.py)
[ "$KIND" = packed ] && DEF="-DFLANHIP_DIAG_PACKED"     # the product's conversions.hip is built WITHOUT packed fp32 (flan_amd/build.py); this variant with
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
[ "$KIND" = packed ] && NOPK=""
[ "$KIND" = maxilp ] && DEF="-mllvm -amdgpu-sched-strategy=max-ilp"
[ "$KIND" = maxmem ] && DEF="-mllvm -amdgpu-sched-strategy=max-memory-clause"
[ "$KIND" = iterilp ] && DEF="-mllvm -amdgpu-sched-strategy=iterative-ilp"
python3 $R/flan_amd/build.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc $DEF $NOPK \
  -I$R/include -I$R/flan_amd/csrc -c $R/flan_amd/csrc/conversions.hip -o /tmp/conversions_$KIND.o
OBJS=$(ls $R/flan_amd/csrc/*.o | grep -v conversions.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ubench/libflanhip_$KIND.so /tmp/conversions_$KIND.o $OBJS -ldl
echo built $R/tools/ubench/libflanhip_$KIND.so
