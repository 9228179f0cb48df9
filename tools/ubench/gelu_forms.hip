// Microbenchmark: cost per element of the GELU forms a convolution prologue could use, as the producers of conv_ws.hip run them
// (eight independent elements per step, scheduling barriers between the steps), plus the issue cost of the transcendental
// instructions on gfx950.  ns per wave64 ELEMENT per SIMD at 2 and 4 resident waves per SIMD.
//   poly7   : Phi = clamp01(0.5 + v p(min(v^2, 16))), p of degree 7 (the fp16 prologue)
//   poly6   : the same with degree 6 (the bf16 prologue)
//   sigm3   : Phi = 1 / (1 + exp2(-v q(min(v^2, 36)))), q of degree 2 in v^2 (max |Phi error| 3.1e-5 on the whole line)
//   table   : Phi by linear interpolation in a 512-entry LDS table of (value, slope) pairs over [-4, 4] (|error| 7e-6):
//             v_fma + v_med3 + v_floor + v_sub + v_cvt + ds_read_b64 + v_fma + v_mul -- fewer VALU, one LDS read per element
//   hipcc -O3 --offload-arch=gfx950 gelu_forms.hip -o gelu_forms && ./gelu_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#define SB() __builtin_amdgcn_sched_barrier(0)
#define G8(expr) _Pragma("unroll") for (int i = 0; i < 8; ++i) { expr; } SB();

__shared__ float2 g_tab[512];
template <int FORM>
__device__ __forceinline__ void gelu8(float (&v)[8]) {
  float w[8], p[8];
  if constexpr (FORM == 7) {
    int idx[8];
    float2 e[8];
    G8(w[i] = __builtin_amdgcn_fmed3f(fmaf(v[i], 64.0f, 256.0f), 0.0f, 511.0f))
    G8(p[i] = __builtin_floorf(w[i]))
    G8(w[i] = w[i] - p[i])
    G8(idx[i] = (int)p[i])
    G8(e[i] = g_tab[idx[i]])
    G8(p[i] = fmaf(e[i].y, w[i], e[i].x))
    G8(v[i] = v[i] * p[i])
    return;
  }
  if constexpr (FORM == 0 || FORM == 1) {
    G8(w[i] = v[i] * v[i])
    G8(w[i] = __builtin_fminf(w[i], 16.0f))
    if constexpr (FORM == 0) {
      G8(p[i] = fmaf(-1.301278171e-09f, w[i], 1.041951057e-07f))
      G8(p[i] = fmaf(p[i], w[i], -3.657111166e-06f))
      G8(p[i] = fmaf(p[i], w[i], 7.485478930e-05f))
      G8(p[i] = fmaf(p[i], w[i], -1.006488756e-03f))
      G8(p[i] = fmaf(p[i], w[i], 9.505392772e-03f))
      G8(p[i] = fmaf(p[i], w[i], -6.588783436e-02f))
      G8(p[i] = fmaf(p[i], w[i], 3.986733897e-01f))
    } else {
      G8(p[i] = fmaf(2.81608722e-08f, w[i], -1.89188380e-06f))
      G8(p[i] = fmaf(p[i], w[i], 5.41903041e-05f))
      G8(p[i] = fmaf(p[i], w[i], -8.78980255e-04f))
      G8(p[i] = fmaf(p[i], w[i], 9.11294959e-03f))
      G8(p[i] = fmaf(p[i], w[i], -6.53883549e-02f))
      G8(p[i] = fmaf(p[i], w[i], 3.98526915e-01f))
    }
    G8(p[i] = __builtin_amdgcn_fmed3f(fmaf(v[i], p[i], 0.5f), 0.0f, 1.0f))
    G8(v[i] = v[i] * p[i])
  } else if constexpr (FORM == 2) {
    // exp2 argument: -log2(e) * v * (c1 + c3 w + c5 w^2)
    G8(w[i] = v[i] * v[i])
    G8(w[i] = __builtin_fminf(w[i], 36.0f))
    G8(p[i] = fmaf(9.844227e-04f, w[i], -1.0654461e-01f))
    G8(p[i] = fmaf(p[i], w[i], -2.3014676f))
    G8(p[i] = p[i] * v[i])
    G8(p[i] = __builtin_amdgcn_exp2f(p[i]))
    G8(p[i] = p[i] + 1.0f)
    G8(p[i] = __builtin_amdgcn_rcpf(p[i]))
    G8(v[i] = v[i] * p[i])
  } else if constexpr (FORM == 3) {  // exp2 only
    G8(v[i] = __builtin_amdgcn_exp2f(v[i]))
  } else if constexpr (FORM == 4) {  // rcp only
    G8(v[i] = __builtin_amdgcn_rcpf(v[i]))
  } else if constexpr (FORM == 5) {  // one fmaak
    G8(v[i] = fmaf(v[i], 0.99f, 0.25f))
  } else if constexpr (FORM == 6) {  // rsq
    G8(v[i] = __builtin_amdgcn_rsqf(v[i]))
  }
}

template <int FORM>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[8];
  if (FORM == 7) {
    for (int i = threadIdx.x; i < 512; i += 256) {
      const float x0 = (i - 256) / 64.0f, x1 = (i - 255) / 64.0f;
      const float f0 = 0.5f * erfcf(-x0 * 0.70710678f), f1 = 0.5f * erfcf(-x1 * 0.70710678f);
      g_tab[i] = float2{f0, f1 - f0};
    }
    __syncthreads();
  }
  for (int i = 0; i < 8; ++i) a[i] = seed * 0.01f * (i + 1) + threadIdx.x * 0.003f - 0.4f + ((threadIdx.x * 37) & 63) * 0.03f;
  for (int it = 0; it < iters; ++it) {
    gelu8<FORM>(a);
    // (keep the values in a sane range without adding more than one cheap instruction per element)
    if (FORM <= 2 || FORM == 7) { G8(a[i] = fmaf(a[i], 0.5f, 0.3f)) }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int FORM>
double run(const char* name, int waves_per_simd, double sub) {
  float* d;
  (void)hipMalloc(&d, 256 * 1024 * 64 * 4);
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / ((double)iters * 8 * waves_per_simd) - sub;
  printf("%-34s waves/SIMD=%d: %7.3f ns per wave64 element per SIMD\n", name, waves_per_simd, ns);
  (void)hipFree(d);
  return ns;
}

int main() {
  for (int w : {2, 4}) {
    const double f = run<5>("v_fmaak_f32 (one instruction)", w, 0);
    run<3>("v_exp_f32", w, 0);
    run<4>("v_rcp_f32", w, 0);
    run<6>("v_rsq_f32", w, 0);
    run<0>("gelu poly7 (fp16 prologue)", w, f);
    run<1>("gelu poly6 (bf16 prologue)", w, f);
    run<2>("gelu sigm3 (exp2 + rcp)", w, f);
    run<7>("gelu table (LDS, linear interp.)", w, f);
  }
  return 0;
}
