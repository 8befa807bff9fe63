# Build libflanhip variants whose conversions.hip was compiled with one extra compiler option each (a scheduling / vectoriser sweep):
#   bash tools/scripts/build_flag_variants.sh          -> tools/ubench/variants/libflanhip_<name>.so
# then on the GPU:  bash tools/scripts/run_flag_variants.sh
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/flan_amd/csrc
V=$R/tools/ubench/variants
# the product build of conversions.hip (flan_amd/build.py: no packed fp32) is the baseline every variant adds one option to
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc -Xclang -target-feature -Xclang -packed-fp32-ops -I$R/include -I$C"
OTHERS="$C/core.o $C/processors.o $C/processors_ext.o $C/processors_arrange.o $C/resample.o $C/utility.o $C/collective.o $C/transfer.o"
build() { # name, flags...
  name=$1; shift
  /opt/rocm/bin/hipcc $BASE "$@" -c $C/conversions.hip -o $V/conv_$name.o 2> $V/$name.err && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libflanhip_$name.so $V/conv_$name.o $OTHERS -ldl && rm -f $V/conv_$name.o && echo "built $name" || echo "FAILED $name"
}
build base &
build noslp -fno-slp-vectorize &
build bias0 -mllvm -amdgpu-schedule-metric-bias=0 &
build bias100 -mllvm -amdgpu-schedule-metric-bias=100 &
wait
build nohighrp -mllvm -amdgpu-disable-unclustered-high-rp-reschedule &
build trackers -mllvm -amdgpu-use-amdgpu-trackers &
build nopost -mllvm -enable-post-misched=0 &
build topdown -mllvm -misched-prera-direction=topdown &
wait
build bottomup -mllvm -misched-prera-direction=bottomup &
build nocluster -mllvm -misched-cluster=0 &
build o2 -O2 &
build slpvf2 -mllvm -slp-max-vf=2 &
wait
build relaxed -mllvm -amdgpu-schedule-relaxed-occupancy &
build prealloc -mllvm -amdgpu-prealloc-sgpr-spill-vgprs &
build nolowocc -mllvm -amdgpu-disable-clustered-low-occupancy-reschedule &
build slpthr -mllvm -slp-threshold=-8 &
wait
ls -la $V/*.so | awk '{print $5, $9}'
