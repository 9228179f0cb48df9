// Shared definitions for the gfx950 sampler library.  Device code here is written
// for CDNA4 only (wave64, MFMA, 160 KiB LDS); there is no other backend.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <string>

namespace vqvs {

// ---- error plumbing --------------------------------------------------------------
void set_error(const std::string& msg);
#define VQVS_FAIL(code, ...)                        \
  do {                                              \
    char _buf[512];                                 \
    snprintf(_buf, sizeof(_buf), __VA_ARGS__);      \
    ::vqvs::set_error(_buf);                        \
    return (code);                                  \
  } while (0)
#define VQVS_HIP(expr)                                                                       \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) VQVS_FAIL(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// ---- vector types ---------------------------------------------------------------
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Statistics (sum, sum of squares) are emitted per (clip, time tile of STAT_TILE rows, channel).
constexpr int STAT_TILE = 256;

// ---- device math ------------------------------------------------------------------
// Exact-erf GELU (reference unet.py:341-342, nn.GELU() default) evaluated with the
// Abramowitz-Stegun 7.1.28 rational form of erf: |erf error| <= 3e-7, no exp, one rcp.
// Measured against float64: max |gelu error| = 8.7e-7 over [-8, 8] (tests/test_kernels_gpu.py).
__device__ __forceinline__ float gelu_f(float v) {
  const float z = fabsf(v) * 0.70710678118654752440f;
  float p = fmaf(0.0000430638f, z, 0.0002765672f);
  p = fmaf(p, z, 0.0001520143f);
  p = fmaf(p, z, 0.0092705272f);
  p = fmaf(p, z, 0.0422820123f);
  p = fmaf(p, z, 0.0705230784f);
  p = fmaf(p, z, 1.0f);
  p = p * p;
  p = p * p;
  p = p * p;
  p = p * p;
  const float e = 1.0f - __builtin_amdgcn_rcpf(p);  // erf(|v|/sqrt2)
  return 0.5f * v * (1.0f + copysignf(e, v));
}

// d/dv gelu(v) = Phi(v) + v*phi(v), same erf evaluation as gelu_f (backward of the classifier, guidance)
__device__ __forceinline__ float gelu_grad_f(float v) {
  const float z = fabsf(v) * 0.70710678118654752440f;
  float p = fmaf(0.0000430638f, z, 0.0002765672f);
  p = fmaf(p, z, 0.0001520143f);
  p = fmaf(p, z, 0.0092705272f);
  p = fmaf(p, z, 0.0422820123f);
  p = fmaf(p, z, 0.0705230784f);
  p = fmaf(p, z, 1.0f);
  p = p * p;
  p = p * p;
  p = p * p;
  p = p * p;
  const float e = 1.0f - __builtin_amdgcn_rcpf(p);
  const float Phi = 0.5f * (1.0f + copysignf(e, v));
  return fmaf(v * 0.39894228040143267794f, __expf(-0.5f * v * v), Phi);
}

// Packed-math variants (v_pk_fma_f32 / v_pk_mul_f32 process two floats per lane per issue; a plain
// wave64 VALU instruction occupies the SIMD for 4 cycles on gfx950, so the prologue is written on
// float2 throughout).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float x) { return f32x2{x, x}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// GELU quality levels of the fused prologues (Q):
//   GELU_EXACT : the A&S 7.1.28 form above, packed (|gelu error| <= 8.7e-7)                       -- VQVS_PREC_F32
//   GELU_POLY6 : v * (0.5 + vc*P(vc^2)), vc = clamp(v, -4, 4), P = degree-6 minimax fit of (Phi(v) - 0.5)/v;
//                max |gelu error| = 5.7e-4 on [-8, 8], below the bf16 rounding of the stored result -- VQVS_PREC_BF16
//   GELU_POLY7 : the same form with a degree-7 fit: |gelu error| <= 4.4e-5 on [-4, 4] (<= 3.4e-4 on [-8, 8], i.e.
//                3.2e-5 relative on the positive tail), below the fp16 rounding of O(1) results (2^-12) -- VQVS_PREC_F16
// No transcendental in the polynomial forms.
enum { GELU_EXACT = 0, GELU_POLY6 = 1, GELU_POLY7 = 2 };
template <int Q>
__device__ __forceinline__ f32x2 gelu2(f32x2 v) {
  if constexpr (Q == GELU_EXACT) {
    f32x2 z;
    z[0] = fabsf(v[0]);
    z[1] = fabsf(v[1]);
    z = z * splat2(0.70710678118654752440f);
    f32x2 p = fma2(splat2(0.0000430638f), z, splat2(0.0002765672f));
    p = fma2(p, z, splat2(0.0001520143f));
    p = fma2(p, z, splat2(0.0092705272f));
    p = fma2(p, z, splat2(0.0422820123f));
    p = fma2(p, z, splat2(0.0705230784f));
    p = fma2(p, z, splat2(1.0f));
    p = p * p;
    p = p * p;
    p = p * p;
    p = p * p;
    f32x2 e;
    e[0] = copysignf(1.0f - __builtin_amdgcn_rcpf(p[0]), v[0]);
    e[1] = copysignf(1.0f - __builtin_amdgcn_rcpf(p[1]), v[1]);
    return (v * splat2(0.5f)) * (e + splat2(1.0f));
  } else {
    f32x2 vc;
    vc[0] = __builtin_amdgcn_fmed3f(v[0], -4.0f, 4.0f);
    vc[1] = __builtin_amdgcn_fmed3f(v[1], -4.0f, 4.0f);
    const f32x2 w = vc * vc;
    f32x2 p;
    if constexpr (Q == GELU_POLY6) {
      p = fma2(splat2(2.81608722e-08f), w, splat2(-1.89188380e-06f));
      p = fma2(p, w, splat2(5.41903041e-05f));
      p = fma2(p, w, splat2(-8.78980255e-04f));
      p = fma2(p, w, splat2(9.11294959e-03f));
      p = fma2(p, w, splat2(-6.53883549e-02f));
      p = fma2(p, w, splat2(3.98526915e-01f));
    } else {
      p = fma2(splat2(-1.301278171e-09f), w, splat2(1.041951057e-07f));
      p = fma2(p, w, splat2(-3.657111166e-06f));
      p = fma2(p, w, splat2(7.485478930e-05f));
      p = fma2(p, w, splat2(-1.006488756e-03f));
      p = fma2(p, w, splat2(9.505392772e-03f));
      p = fma2(p, w, splat2(-6.588783436e-02f));
      p = fma2(p, w, splat2(3.986733897e-01f));
    }
    return v * fma2(vc, p, splat2(0.5f));  // Phi~(-4) = -7e-5 is not clamped
  }
}
template <typename T> struct GeluQ;  // quality level that goes with an activation storage type
template <> struct GeluQ<float> { static constexpr int q = GELU_EXACT; };
template <> struct GeluQ<bf16_t> { static constexpr int q = GELU_POLY6; };
#ifdef VQVS_F16_GELU6
template <> struct GeluQ<half_t> { static constexpr int q = GELU_POLY6; };
#else
template <> struct GeluQ<half_t> { static constexpr int q = GELU_POLY7; };
#endif

// element load/store of 8 consecutive channels as fp32, for both activation storage types
template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int BYTES = 4;
  __device__ static __forceinline__ f32x8 load8(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  }
  __device__ static __forceinline__ void store8(float* p, f32x8 v) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
};
template <>
struct Elem<bf16_t> {
  static constexpr int BYTES = 2;
  __device__ static __forceinline__ f32x8 load8(const bf16_t* p) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
    return __builtin_convertvector(a, f32x8);
  }
  __device__ static __forceinline__ void store8(bf16_t* p, f32x8 v) {
    *reinterpret_cast<bf16x8*>(p) = __builtin_convertvector(v, bf16x8);
  }
};

template <>
struct Elem<half_t> {
  static constexpr int BYTES = 2;
  __device__ static __forceinline__ f32x8 load8(const half_t* p) {
    const f16x8 a = *reinterpret_cast<const f16x8*>(p);
    return __builtin_convertvector(a, f32x8);
  }
  __device__ static __forceinline__ void store8(half_t* p, f32x8 v) {
    *reinterpret_cast<f16x8*>(p) = __builtin_convertvector(v, f16x8);
  }
};

// one of three kernel instantiations by the model's precision (VQVS_PREC_*): F32 -> float, BF16 -> bf16_t, F16 -> half_t
#define VQVS_BY_PRECISION(precision, CALL) \
  do {                                      \
    if ((precision) == 0) {                 \
      using T = float;                      \
      CALL;                                 \
    } else if ((precision) == 1) {          \
      using T = ::vqvs::bf16_t;             \
      CALL;                                 \
    } else {                                \
      using T = ::vqvs::half_t;             \
      CALL;                                 \
    }                                       \
  } while (0)

__device__ __forceinline__ f32x8 f32x8_zero() { return f32x8{0, 0, 0, 0, 0, 0, 0, 0}; }

// split an fp32 octet into bf16 hi + bf16 lo (x ~= hi + lo to ~2^-17 relative)
__device__ __forceinline__ void split_bf16(f32x8 v, bf16x8& hi, bf16x8& lo) {
  hi = __builtin_convertvector(v, bf16x8);
  const f32x8 r = v - __builtin_convertvector(hi, f32x8);
  lo = __builtin_convertvector(r, bf16x8);
}

}  // namespace vqvs
