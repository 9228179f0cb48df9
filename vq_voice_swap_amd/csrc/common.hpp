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

__device__ __forceinline__ f32x8 f32x8_zero() { return f32x8{0, 0, 0, 0, 0, 0, 0, 0}; }

// split an fp32 octet into bf16 hi + bf16 lo (x ~= hi + lo to ~2^-17 relative)
__device__ __forceinline__ void split_bf16(f32x8 v, bf16x8& hi, bf16x8& lo) {
  hi = __builtin_convertvector(v, bf16x8);
  const f32x8 r = v - __builtin_convertvector(hi, f32x8);
  lo = __builtin_convertvector(r, bf16x8);
}

}  // namespace vqvs
