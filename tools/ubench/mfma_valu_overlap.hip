// Microbenchmark: can ONE wave overlap its own VALU work with its own in-flight MFMAs (same basic block)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NM, int NV, bool INTER>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x;
  const float c = seed * 0.5f;
  for (int it = 0; it < iters; ++it) {
    if (INTER && NM > 0) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NV / NM; ++n) v[(m * (NV / NM) + n) & 15] = fmaf(v[(m * (NV / NM) + n) & 15], c, 0.25f);
      }
    } else {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NV; ++n) v[n & 15] = fmaf(v[n & 15], c, 0.25f);
    }
  }
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NM, int NV, bool INTER = false> void run(int wps) {
  float* d; (void)hipMalloc(&d, 256 * 1024 * 64 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NM, NV, INTER>), dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NM, NV, INTER>), dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%s MFMA=%2d VALU=%3d waves/SIMD=%d: %.3f ms  -> %.1f ns per iteration per SIMD-wave-slot\n", INTER ? "interleaved" : "blocked    ", NM, NV, wps, ms, ms * 1e6 / iters / wps);
  (void)hipFree(d);
}
int main() {
  for (int w : {1, 2}) { run<8, 0>(w); run<0, 96>(w); run<8, 96>(w); run<8, 48>(w); run<8, 192>(w); run<8, 96, true>(w); run<8, 48, true>(w); run<8, 192, true>(w); }
  return 0;
}
