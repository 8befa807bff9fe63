// valu_issue.hip -- how many cycles does a wave64 fp32 VALU instruction occupy a gfx950 SIMD?
// Independent and dependent v_fma_f32 / v_pk_fma_f32 chains at 1, 2, 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template<int ILP, bool PACKED>
__global__ void k(float* out, int iters, float a, float b) {
  float x[ILP]; v2f y[ILP];
  for (int i = 0; i < ILP; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = v2f{x[i], x[i] + 1.f}; }
  for (int it = 0; it < iters; ++it) {
    #pragma unroll
    for (int r = 0; r < 16; ++r) {
      #pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (PACKED) y[i] = __builtin_elementwise_fma(y[i], v2f{a, a}, v2f{b, b});
        else x[i] = __builtin_fmaf(x[i], a, b);
      }
    }
  }
  float s = 0; for (int i = 0; i < ILP; ++i) s += x[i] + y[i].x + y[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template<int ILP, bool PACKED>
void run(int waves_per_simd) {
  const int threads = 64 * 4 * waves_per_simd;   // one block per CU, waves_per_simd waves on each of the 4 SIMDs
  const int blocks = 256, iters = 4000;
  float* d; hipMalloc(&d, sizeof(float) * threads * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<ILP, PACKED><<<blocks, threads>>>(d, 10, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<ILP, PACKED><<<blocks, threads>>>(d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = double(iters) * 16 * ILP * waves_per_simd;   // wave-instructions issued on one SIMD
  printf("%s ILP=%d waves/SIMD=%d : %.3f ms -> %.2f ns per wave-instruction per SIMD (x GHz = cycles)\n",
         PACKED ? "v_pk_fma_f32" : "v_fma_f32   ", ILP, waves_per_simd, ms, ms * 1e6 / instr_per_simd);
  hipFree(d);
}

int main() {
  // clocks: an idle MI355X needs tens of milliseconds of load before they settle -- spin ~150 ms first
  { float* d; hipMalloc(&d, sizeof(float) * 256 * 1024); for (int i = 0; i < 60; ++i) k<4, false><<<256, 1024>>>(d, 4000, 1.0001f, 0.5f); hipDeviceSynchronize(); hipFree(d); }
  for (int w : {1, 2, 4}) { run<1, false>(w); run<4, false>(w); run<1, true>(w); run<4, true>(w); }
  return 0;
}
