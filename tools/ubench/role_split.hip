// Microbenchmark behind the producer/consumer convolution: inside ONE 8-wave workgroup, waves 0-3
// (one per SIMD) issue only MFMAs (+ LDS fragment reads), waves 4-7 (their SIMD partners) only VALU work (+ LDS writes),
// with one s_barrier per iteration as in the convolution's K loop.  How long is an iteration compared with the two
// roles alone, i.e. do the matrix pipe and the VALU of one SIMD overlap across waves?
//   hipcc -O3 --offload-arch=gfx950 role_split.hip -o role_split && ./role_split
#include <hip/hip_runtime.h>

#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// NM MFMAs per consumer wave and iteration, NV packed-fp32 FMAs per producer wave and iteration (+ NV/8 cvt),
// LDS: consumers read NM fragments (ds_read_b128), producers write NV/16 x 16 bytes.  ROLE: 0 both, 1 consumers only
// (producers idle at the barrier), 2 producers only, 3 every wave does BOTH (half the MFMAs and half the VALU each).
template <int NM, int NV, int ROLE, int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool consumer = wave < 4;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{seed + i + tid, seed - i};
  const f32x2 c = f32x2{seed * 0.5f, seed * 0.25f}, d = f32x2{0.25f, 0.125f};
  for (int i = tid; i < 64 * 1024 / 16; i += 512) reinterpret_cast<f16x8*>(lds)[i] = a;
  __syncthreads();
  auto mfma_part = [&](int n) {
#pragma unroll
    for (int m = 0; m < n; ++m) {
      const f16x8 fa = *reinterpret_cast<const f16x8*>(lds + ((m * 64 + lane) * 80) % (60 * 1024));
      acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, b, acc[m & 7], 0, 0, 0);
    }
  };
  // KIND 0: n packed-fp32 FMAs (v_pk_fma_f32, 128 flops per lane pair);  KIND 1: the same arithmetic as 2n scalar v_fma_f32;
  // KIND 2: 2n scalar FMAs with literal constants (v_fmaak_f32 / v_fmac_f32: VOP2 encodings)
  auto valu_part = [&](int n) {
    if constexpr (KIND == 0) {
#pragma unroll
      for (int q = 0; q < n; ++q) v[q & 7] = __builtin_elementwise_fma(v[q & 7], c, d);
    } else if constexpr (KIND == 1) {
#pragma unroll
      for (int q = 0; q < n; ++q) {
        float x = v[q & 7][0], y = v[q & 7][1];
        asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(c[0]), "v"(d[0]));
        v[q & 7][0] = x;
        v[q & 7][1] = y;
      }
    } else {
#pragma unroll
      for (int q = 0; q < n; ++q) {
        float x = v[q & 7][0], y = v[q & 7][1];
        asm volatile("v_fmaak_f32 %0, %0, %2, 0x3e800000\n\tv_fmaak_f32 %1, %1, %2, 0x3e800000" : "+v"(x), "+v"(y) : "v"(c[0]));
        v[q & 7][0] = x;
        v[q & 7][1] = y;
      }
    }
#pragma unroll
    for (int q = 0; q < n / 16; ++q) {
      f16x8 h;
      for (int e = 0; e < 8; ++e) h[e] = (_Float16)v[e][q & 1];
      *reinterpret_cast<f16x8*>(lds + 32 * 1024 + ((q * 512 + tid) * 16) % (28 * 1024)) = h;
    }
  };
  for (int it = 0; it < iters; ++it) {
    if (ROLE == 3) {
      mfma_part(NM / 2);
      valu_part(NV / 2);
    } else if (consumer) {
      if (ROLE != 2) mfma_part(NM);
    } else {
      if (ROLE != 1) valu_part(NV);
    }
    __builtin_amdgcn_s_barrier();
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  out[blockIdx.x * 512 + tid] = s;
}

template <int NM, int NV, int ROLE, int KIND>
float run() {
  float* d;
  (void)hipMalloc(&d, 256 * 512 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NM, NV, ROLE, KIND>), dim3(256), dim3(512), 0, 0, d, 10, 1.0f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NM, NV, ROLE, KIND>), dim3(256), dim3(512), 0, 0, d, iters, 1.0f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(d);
  return ms * 1e6f / iters;  // ns per iteration
}

template <int NM, int NV, int KIND>
void row() {
  const float both = run<NM, NV, 0, KIND>(), cons = run<NM, NV, 1, KIND>(), prod = run<NM, NV, 2, KIND>(), mixed = run<NM, NV, 3, KIND>();
  printf("%s MFMA/iter=%3d  pkFMA-equivalents/iter=%3d : consumers alone %7.1f ns  producers alone %7.1f ns  split roles %7.1f ns  (max %7.1f, sum %7.1f)"
         "  every wave both %7.1f ns\n", KIND == 0 ? "v_pk_fma_f32" : KIND == 1 ? "v_fma_f32   " : "v_fmaak_f32 ", NM, NV, cons, prod, both, cons > prod ? cons : prod, cons + prod, mixed);
}

int main() {
  row<24, 64, 0>();
  row<24, 128, 0>();
  row<48, 128, 0>();
  row<48, 256, 0>();
  row<24, 64, 1>();
  row<24, 128, 1>();
  row<48, 128, 1>();
  row<48, 256, 1>();
  row<24, 128, 2>();
  row<48, 128, 2>();
  row<48, 256, 2>();
  return 0;
}
