// Microbenchmark: issue rate of plain vs packed fp32 VALU instructions on gfx950 (cycles per wave64 instruction per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[16];
  f32x2 p[16];
  for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 1.f}; }
  const float c = seed * 0.5f;
  const f32x2 c2 = {c, c};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) a[i] = fmaf(a[i], c, 0.25f);
      if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], c2, c2);
      if (MODE == 2) a[i] = __builtin_amdgcn_fmed3f(a[i], -c, c);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
  float* d; hipMalloc(&d, 256 * 1024 * 64 * 4);
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * 16;
  const double waves_per_simd_total = waves_per_simd;  // per SIMD
  const double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd_total);
  printf("%-14s waves/SIMD=%d: %.3f ms -> %.3f ns per wave64 instr per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
         ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4}) { run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_med3_f32", w); }
  return 0;
}
