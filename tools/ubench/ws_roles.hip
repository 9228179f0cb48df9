// Micro-benchmark behind csrc/conv_ws.hip: one 16-wave workgroup per CU; waves 8-15 run the producers' prologue arithmetic
// (the exact fp16 affine + degree-7 GELU of xform8, 2 x 8 elements per thread, + 2 ds_write_b128), waves 0-7 the consumers'
// chunk (3 taps x 2 k-steps x (2 A + 2 B ds_read_b128, 4 MFMAs of 32x32x16 f16)).  Modes: producers alone, consumers alone,
// both, with / without one s_barrier per iteration.  Question: do the two overlap on one SIMD, and what does an iteration cost?
//   hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 ws_roles.hip -o ws_roles && ./ws_roles
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu7(float v) {
  const float vc = __builtin_amdgcn_fmed3f(v, -4.0f, 4.0f);
  const float w = vc * vc;
  float p = fmaf(-1.301278171e-09f, w, 1.041951057e-07f);
  p = fmaf(p, w, -3.657111166e-06f);
  p = fmaf(p, w, 7.485478930e-05f);
  p = fmaf(p, w, -1.006488756e-03f);
  p = fmaf(p, w, 9.505392772e-03f);
  p = fmaf(p, w, -6.588783436e-02f);
  p = fmaf(p, w, 3.986733897e-01f);
  return v * fmaf(vc, p, 0.5f);
}
__device__ __forceinline__ u32x4 xform8(u32x4 raw, f32x4 s0, f32x4 s1, f32x4 s2, f32x4 s3) {
  const f16x8 h = __builtin_bit_cast(f16x8, raw);
  f16x8 o;
  o[0] = (_Float16)gelu7(fmaf((float)h[0], s0[0], s0[1]));
  o[1] = (_Float16)gelu7(fmaf((float)h[1], s0[2], s0[3]));
  o[2] = (_Float16)gelu7(fmaf((float)h[2], s1[0], s1[1]));
  o[3] = (_Float16)gelu7(fmaf((float)h[3], s1[2], s1[3]));
  o[4] = (_Float16)gelu7(fmaf((float)h[4], s2[0], s2[1]));
  o[5] = (_Float16)gelu7(fmaf((float)h[5], s2[2], s2[3]));
  o[6] = (_Float16)gelu7(fmaf((float)h[6], s3[0], s3[1]));
  o[7] = (_Float16)gelu7(fmaf((float)h[7], s3[2], s3[3]));
  return __builtin_bit_cast(u32x4, o);
}

// MODE bit 0: producers work, bit 1: consumers work, bit 2: barrier per iteration.  NP = producer items per iteration (2 = one chunk)
// CM (consumer variant): 0 = as in the kernel, 1 = accumulators in AGPRs, 2 = MFMAs on register operands (no fragment reads),
// 3 = fragment reads only (no MFMAs), 4 = AGPR accumulators on register operands
template <int MODE, int NP, int CM = 0>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 80 * 1024 / 16; i += 1024) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3c003c00u, 0x38003800u, 0x3c003c00u, 0x34003400u};
  __syncthreads();
  float res = 0.f;
  if (wave >= 8) {
    const int pt = tid - 512;
    const int oct = pt & 3, r0 = pt >> 2;
    const int dst0 = r0 * 64 + ((oct ^ ((r0 >> 2) & 3)) << 4);
    u32x4 a0 = {0x3c003800u + (unsigned)tid, 0x34003a00u, 0xb800bc00u, 0x3e003900u}, a1 = {0x38003c00u, 0xb4003a00u + (unsigned)tid, 0x3800bc00u, 0x3a003900u};
    f32x4 s0 = {seed, 0.1f, seed * 0.9f, -0.1f}, s1 = {seed * 1.1f, 0.2f, seed, -0.2f}, s2 = s0 * 0.7f, s3 = s1 * 1.3f;
    for (int it = 0; it < iters; ++it) {
      if (MODE & 1) {
#pragma unroll
        for (int j = 0; j < NP / 2; ++j) {
          asm volatile("" : "+v"(a0), "+v"(a1));  // fresh inputs every iteration
          const u32x4 o0 = xform8(a0, s0, s1, s2, s3);
          const u32x4 o1 = xform8(a1, s0, s1, s2, s3);
          *reinterpret_cast<u32x4*>(smem + (it & 1) * 40960 + dst0) = o0;
          *reinterpret_cast<u32x4*>(smem + (it & 1) * 40960 + dst0 + 8192) = o1;
        }
      }
      if (MODE & 4) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    res = (float)a0[0];
  } else {
    const int wt = wave & 3, wc = wave >> 2, l31 = lane & 31, hh = lane >> 5;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int boff[2];
    const int wr = wc * 64 + l31;
    for (int ks = 0; ks < 2; ++ks) boff[ks] = 16384 + wr * 64 + (((ks * 2 + hh) ^ ((wr >> 2) & 3)) << 4);
    const int d = (int)seed + 1;
    f32x4v acc4[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc4[i][j][r] = 0.f;
    const int l15 = lane & 15, kg = lane >> 4;
    for (int it = 0; it < iters; ++it) {
      if ((MODE & 2) && CM >= 5) {
        // 16x16x32 MFMAs (4 passes each): per tap one k-step of 32 = 4 A fragments (16 rows each) + 4 B fragments (16 channels each),
        // 16 MFMAs -- the same fragment bytes and the same FLOPs as 2 k-steps x (2 A + 2 B, 4 MFMAs of 32x32x16)
        const char* const sb = smem + (it & 1) * 40960;
        int row = wt * 64 + l15;
        for (int kk = 0; kk < 3; ++kk, row += d) {
          const char* const wk = sb + 16384 + kk * (128 * 64);
          f16x8 xa[4], xb[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int r = row + m * 16, c = wc * 64 + m * 16 + l15;
            if (CM == 5) {
              xa[m] = __builtin_bit_cast(f16x8, u32x4{(unsigned)r, 0x3c003c00u, 0x38003800u, 0x34003400u});
              xb[m] = __builtin_bit_cast(f16x8, u32x4{0x3c003c00u, (unsigned)c, 0x38003800u, 0x34003400u});
              asm volatile("" : "+v"(xa[m]), "+v"(xb[m]));
            } else {
              xa[m] = *reinterpret_cast<const f16x8*>(sb + r * 64 + ((kg ^ ((r >> 2) & 3)) << 4));
              xb[m] = *reinterpret_cast<const f16x8*>(wk + c * 64 + ((kg ^ ((c >> 2) & 3)) << 4));
            }
          }
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc4[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[m], xb[n], acc4[m][n], 0, 0, 0);
        }
      }
      if ((MODE & 2) && CM < 5) {
        const char* const sb = smem + (it & 1) * 40960;
        int row = wt * 64 + l31;
        for (int kk = 0; kk < 3; ++kk, row += d) {
          const int swz = (row >> 2) & 3;
          const char* const wk = sb + kk * (128 * 64);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int ao = row * 64 + (((ks * 2 + hh) ^ swz) << 4);
            f16x8 x0, x1;
            if (CM == 2 || CM == 4) {
              x0 = __builtin_bit_cast(f16x8, u32x4{(unsigned)ao, 0x3c003c00u, 0x38003800u, 0x34003400u});
              x1 = __builtin_bit_cast(f16x8, u32x4{0x3c003c00u, (unsigned)ao, 0x38003800u, 0x34003400u});
              asm volatile("" : "+v"(x0), "+v"(x1));
            } else {
              x0 = *reinterpret_cast<const f16x8*>(sb + ao);
              x1 = *reinterpret_cast<const f16x8*>(sb + ao + 2048);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              f16x8 bf;
              if (CM == 2 || CM == 4) {
                bf = x1;
              } else {
                bf = *reinterpret_cast<const f16x8*>(wk + boff[ks] + nt * 2048);
              }
              if (CM == 3) {
                asm volatile("" ::"v"(x0), "v"(x1), "v"(bf));
              } else if (CM == 1 || CM == 4) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[0][nt]) : "v"(x0), "v"(bf));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[1][nt]) : "v"(x1), "v"(bf));
              } else {
                acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0, bf, acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1, bf, acc[1][nt], 0, 0, 0);
              }
            }
          }
        }
      }
      if (MODE & 4) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) res += acc[i][j][r];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) res += acc4[i][j][r];
  }
  out[blockIdx.x * 1024 + tid] = res;
}

template <int MODE, int NP, int CM = 0>
float run() {
  float* d;
  (void)hipMalloc(&d, 256 * 1024 * 4);
  const int iters = 2000, LDS = 90 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, NP, CM>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NP, CM>), dim3(256), dim3(1024), LDS, 0, d, 10, 1.0f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NP, CM>), dim3(256), dim3(1024), LDS, 0, d, iters, 1.0f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(d);
  return ms * 1e6f / iters;
}

int main() {
  printf("per iteration (one K chunk of a 256 x 128 tile per CU: 48 MFMAs + 48 ds_read_b128 per SIMD; 32 prologue elements per SIMD lane):\n");
  printf("  no barrier : producers alone %7.1f ns   consumers alone %7.1f ns   both %7.1f ns\n", run<1, 2>(), run<2, 2>(), run<3, 2>());
  printf("  s_barrier  : producers alone %7.1f ns   consumers alone %7.1f ns   both %7.1f ns\n", run<5, 2>(), run<6, 2>(), run<7, 2>());
  printf("  producers with twice the arithmetic: alone %7.1f ns   both (barrier) %7.1f ns\n", run<5, 4>(), run<7, 4>());
  printf("consumer variants (s_barrier): consumers alone / both\n");
  printf("  as in the kernel            %7.1f %7.1f\n", run<6, 2, 0>(), run<7, 2, 0>());
  printf("  accumulators in AGPRs       %7.1f %7.1f\n", run<6, 2, 1>(), run<7, 2, 1>());
  printf("  MFMAs on register operands  %7.1f %7.1f\n", run<6, 2, 2>(), run<7, 2, 2>());
  printf("  fragment reads only         %7.1f %7.1f\n", run<6, 2, 3>(), run<7, 2, 3>());
  printf("  AGPRs + register operands   %7.1f %7.1f\n", run<6, 2, 4>(), run<7, 2, 4>());
  printf("  16x16x32 MFMAs, register operands %7.1f %7.1f\n", run<6, 2, 5>(), run<7, 2, 5>());
  printf("  16x16x32 MFMAs, fragment reads    %7.1f %7.1f\n", run<6, 2, 6>(), run<7, 2, 6>());
  return 0;
}
