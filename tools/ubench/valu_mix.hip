// Microbenchmark: issue cost of the individual VALU instructions of the convolution prologue (conv_ws.hip producers) on gfx950:
// ns and cycles per wave64 instruction per SIMD, at 2 and 4 resident waves per SIMD.  16 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;
  float c = seed * 0.5f;
  float sc = __builtin_amdgcn_readfirstlane(seed * 0.25f);
  float c3 = seed * 0.125f + threadIdx.x;
  unsigned long long msk;
  asm volatile("s_mov_b64 %0, 0x55555555" : "=s"(msk));
  for (int it = 0; it < iters; ++it) {
#define A0(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
#define A1(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3e800001" : "+v"(a[i]) : "v"(c));
#define A2(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "s"(sc));
#define A3(i) asm volatile("v_fma_mix_f32 %0, %0, %1, %1 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(c));
#define A4(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define A5(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define A6(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
#define A7(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
#define A8(i) asm volatile("v_fmamk_f32 %0, %0, 0x3e800001, %1" : "+v"(a[i]) : "v"(c));
#define A9(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(c));
#define B0(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "s"(msk));
#define B1(i) asm volatile("v_med3_f32 %0, %0, -4.0, 4.0" : "+v"(a[i]));
#define B2(i) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(c), "s"(sc));
#define B3(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define B4(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(a[i]));
#define B5(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(c3));
#define B6(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
#define B7(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
#define B8(i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define B9(i) asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(a[i]) : "v"(c));
    if (MODE == 10) { REP16(B0) }
    if (MODE == 11) { REP16(B1) }
    if (MODE == 12) { REP16(B2) }
    if (MODE == 13) { REP16(B3) }
    if (MODE == 14) { REP16(B4) }
    if (MODE == 15) { REP16(B5) }
    if (MODE == 16) { asm volatile("s_mov_b64 vcc, %0" :: "s"(msk) : "vcc"); REP16(B6) }
    if (MODE == 17) { REP16(B7) }
    if (MODE == 18) { REP16(B8) }
    if (MODE == 19) { REP16(B9) }
    if (MODE == 0) { REP16(A0) }
    if (MODE == 1) { REP16(A1) }
    if (MODE == 2) { REP16(A2) }
    if (MODE == 3) { REP16(A3) }
    if (MODE == 4) { REP16(A4) }
    if (MODE == 5) { REP16(A5) }
    if (MODE == 6) { REP16(A6) }
    if (MODE == 7) { REP16(A7) }
    if (MODE == 8) { REP16(A8) }
    if (MODE == 9) { REP16(A9) }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
  float* d; hipMalloc(&d, 256 * 1024 * 64 * 4);
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / ((double)iters * 16 * waves_per_simd);
  printf("%-28s waves/SIMD=%d: %.3f ns per wave64 instr per SIMD\n", name, waves_per_simd, ns);
  hipFree(d);
}
int main() {
  for (int w : {2, 4}) {
    run<0>("v_fma_f32 (vgprs)", w); run<1>("v_fmaak_f32 (literal)", w); run<8>("v_fmamk_f32 (literal)", w); run<2>("v_fma_f32 (sgpr addend)", w);
    run<15>("v_fma_f32 (3 distinct vgprs)", w); run<17>("v_fma_f32 (same vgpr x3)", w); run<10>("v_cndmask_b32_e64 (sgpr mask)", w); run<16>("v_cndmask_b32 (vcc set)", w);
    run<11>("v_med3_f32 (inline consts)", w); run<12>("v_fma_mix_f32 (sgpr addend)", w); run<13>("v_and_b32", w); run<14>("v_cvt_pk_f16_f32 (1 vgpr)", w);
    run<18>("v_pk_mul_f16", w); run<19>("v_dot2c_f32_f16", w);
    run<9>("v_fmac_f32 (VOP2)", w); run<3>("v_fma_mix_f32", w); run<4>("v_cvt_pk_f16_f32", w); run<5>("v_mul_f32", w); run<6>("v_med3_f32", w); run<7>("v_cndmask_b32", w);
  }
  return 0;
}
