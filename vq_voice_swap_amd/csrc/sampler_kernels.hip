// DDPM reverse-step kernels, the counter-based normal generator, and the VQ codebook search.
// All of these are HBM-bound elementwise / small-reduction kernels.
#include "kernels.hpp"
#include "sampler_kernels.hpp"

namespace vqvs {

namespace {

// ---------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), keyed by the sampler seed; the counter carries
// (quad index within the clip, GLOBAL clip index, step index, stream id) so a clip's noise
// does not depend on which GPU or batch slot it is sampled in.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[0] = n0;
  c[1] = (uint32_t)p1;
  c[2] = n2;
  c[3] = (uint32_t)p0;
}

__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint32_t quad, uint64_t clip, uint32_t step, uint32_t stream_id) {
  uint32_t c[4] = {quad, (uint32_t)clip, step, stream_id ^ ((uint32_t)(clip >> 32) << 8)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  // Box-Muller on (0,1] x [0,1)
  const float u0 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u1 = (float)(c[1] >> 8) * (1.0f / 16777216.0f);
  const float u2 = ((float)(c[2] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u3 = (float)(c[3] >> 8) * (1.0f / 16777216.0f);
  const float r0 = sqrtf(-2.0f * logf(u0));
  const float r1 = sqrtf(-2.0f * logf(u2));
  float s0, c0, s1, c1;
  sincosf(6.28318530717958647692f * u1, &s0, &c0);
  sincosf(6.28318530717958647692f * u3, &s1, &c1);
  return f32x4{r0 * c0, r0 * s0, r1 * c1, r1 * s1};
}

__global__ __launch_bounds__(256) void randn_kernel(float* out, int T, uint64_t seed, uint64_t clip_offset, uint32_t stream_id) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q * 4 >= T) return;
  const f32x4 n = philox_normal4(seed, (uint32_t)q, clip_offset + b, 0u, stream_id);
  float* o = out + (size_t)b * T + q * 4;
  if (q * 4 + 3 < T) {
    *reinterpret_cast<f32x4*>(o) = n;
  } else {
    for (int j = 0; q * 4 + j < T; ++j) o[j] = n[j];
  }
}

struct StepCoef {
  float c1, c2, sig, sq1mat, rsat, sqat, rs1mat, c3;
};

// per-clip scalars, evaluated in the reference's operation order (diffusion.py:64-78)
__device__ __forceinline__ StepCoef step_coef(float a_t, float a_prev, bool sigma_large) {
  StepCoef k;
  const float alphas = a_t / a_prev;
  const float betas = 1.0f - alphas;
  const float om = 1.0f - a_t;
  k.c1 = 1.0f / sqrtf(alphas);
  k.c2 = betas * (1.0f / sqrtf(om));
  const float sig2 = sigma_large ? betas : betas * (1.0f - a_prev) / om;
  k.sig = sqrtf(sig2);
  k.sq1mat = sqrtf(om);
  k.rsat = 1.0f / sqrtf(a_t);
  k.sqat = sqrtf(a_t);
  k.rs1mat = 1.0f / sqrtf(om);
  k.c3 = sig2;
  return k;
}

constexpr int SUM_CHUNK = 4096;

// sum over time of x0 = (x_t - sqrt(1-a_t) eps) rsqrt(a_t), per (clip, chunk); fp64 partials
__global__ __launch_bounds__(256) void ddpm_x0sum_kernel(const float* x_t, const float* eps, const float* a_t, double* partial, int T, int nchunk) {
  __shared__ double red[256];
  const int b = blockIdx.y;
  const float at = a_t[b];
  const float sq = sqrtf(1.0f - at), rs = 1.0f / sqrtf(at);
  const int beg = blockIdx.x * SUM_CHUNK, end = min(T, beg + SUM_CHUNK);
  double s = 0.0;
  for (int t = beg + threadIdx.x; t < end; t += 256) {
    const size_t i = (size_t)b * T + t;
    s += (double)((x_t[i] - sq * eps[i]) * rs);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(size_t)b * nchunk + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void ddpm_step_kernel(const float* x_t, const float* eps, const float* noise, const float* a_t,
                                                        const float* a_prev, const double* partial, int nchunk, float* out, int T,
                                                        uint32_t flags, float noise_scale, uint64_t seed, uint64_t clip_offset,
                                                        uint32_t step_index) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q * 4 >= T) return;
  const StepCoef k = step_coef(a_t[b], a_prev[b], flags & 1u);
  float mean = 0.f;
  if (flags & 2u) {
    double s = 0.0;
    for (int i = 0; i < nchunk; ++i) s += partial[(size_t)b * nchunk + i];
    mean = (float)(s / (double)T);
  }
  const size_t base = (size_t)b * T + q * 4;
  const int n = min(4, T - q * 4);
  float xv[4], ev[4], nv[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < n; ++j) {
    xv[j] = x_t[base + j];
    ev[j] = eps[base + j];
  }
  if (noise_scale != 0.f) {
    if (noise) {
      for (int j = 0; j < n; ++j) nv[j] = noise[base + j] * noise_scale;
    } else {
      const f32x4 z = philox_normal4(seed, (uint32_t)q, clip_offset + b, step_index, 0u);
      for (int j = 0; j < 4; ++j) nv[j] = z[j] * noise_scale;
    }
  }
  for (int j = 0; j < n; ++j) {
    float e = ev[j];
    if (flags & 2u) {
      float x0 = (xv[j] - k.sq1mat * e) * k.rsat;
      x0 = fminf(fmaxf(x0 - mean, -1.0f), 1.0f);
      e = (xv[j] - x0 * k.sqat) * k.rs1mat;
    }
    out[base + j] = k.c1 * (xv[j] - k.c2 * e) + k.sig * nv[j];
  }
}

// mean = eps_to_prev(eps)
__global__ __launch_bounds__(256) void ddpm_mean_kernel(const float* x_t, const float* eps, const float* a_t, const float* a_prev, float* out, int T) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const StepCoef k = step_coef(a_t[b], a_prev[b], false);
  const size_t i = (size_t)b * T + t;
  out[i] = k.c1 * (x_t[i] - k.c2 * eps[i]);
}

// eps' = prev_to_eps(mean + sigma^2 * grad) = (-(mean + s2 g) * sqrt(alpha) + x_t) * sqrt(1-a_t) / beta
__global__ __launch_bounds__(256) void ddpm_guided_eps_kernel(const float* x_t, const float* mean, const float* grad, const float* a_t,
                                                              const float* a_prev, float* out, int T, uint32_t flags) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float at = a_t[b], ap = a_prev[b];
  const StepCoef k = step_coef(at, ap, flags & 1u);
  const float alphas = at / ap;
  const float betas = 1.0f - alphas;
  const size_t i = (size_t)b * T + t;
  const float m = mean[i] + k.c3 * grad[i];
  out[i] = (-m * sqrtf(alphas) + x_t[i]) * k.sq1mat / betas;
}

// ---------------------------------------------------------------------------------
// VQ nearest codeword (reference vq.py:127-131, 199-221).
//   dist[k] = ((-2 * <x, e_k>) + |e_k|^2) + |x|^2   in fp32, dot as an fmaf chain in channel
//   order; argmin with the FIRST minimal index (torch.argmin semantics).
// Workgroup = 32 time positions of one clip; thread (pos = tid&31, cg = tid>>5) scores 16
// codes of every 128-code tile.  z is read in its NCT layout, coalesced along time.
// ---------------------------------------------------------------------------------
constexpr int VQ_POS = 32, VQ_TILE = 128, VQ_KC = 64, VQ_DS = VQ_KC + 4;

__global__ void vq_norms_kernel(const float* dict, float* en, int K, int Cd) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  float s = 0.f;
  for (int c = 0; c < Cd; ++c) s = fmaf(dict[(size_t)k * Cd + c], dict[(size_t)k * Cd + c], s);
  en[k] = s;
}

__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* z, const float* dict, const float* en, int64_t* idx_out, int Cd,
                                                        int T1, int K) {
  __shared__ __attribute__((aligned(16))) float xs[VQ_KC][VQ_POS];
  __shared__ __attribute__((aligned(16))) float ds[VQ_TILE][VQ_DS];
  __shared__ float bd[8][VQ_POS];
  __shared__ int bi[8][VQ_POS];
  const int tid = threadIdx.x;
  const int pos = tid & 31, cg = tid >> 5;
  const int b = blockIdx.y, t0 = blockIdx.x * VQ_POS;
  const float* zb = z + (size_t)b * Cd * T1;
  float best = INFINITY;
  int best_i = 0;
  float xn = 0.f;
  for (int k0 = 0; k0 < K; k0 += VQ_TILE) {
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    xn = 0.f;
    for (int c0 = 0; c0 < Cd; c0 += VQ_KC) {
      __syncthreads();
      for (int i = tid; i < VQ_KC * VQ_POS; i += 256) {
        const int c = i >> 5, p = i & 31;
        xs[c][p] = (c0 + c < Cd && t0 + p < T1) ? zb[(size_t)(c0 + c) * T1 + t0 + p] : 0.f;
      }
      for (int i = tid; i < VQ_TILE * (VQ_KC / 4); i += 256) {
        const int r = i / (VQ_KC / 4), c4 = (i % (VQ_KC / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k0 + r < K) {
          if (c0 + c4 + 3 < Cd) {
            v = *reinterpret_cast<const f32x4*>(dict + (size_t)(k0 + r) * Cd + c0 + c4);
          } else {
            for (int j = 0; j < 4; ++j)
              if (c0 + c4 + j < Cd) v[j] = dict[(size_t)(k0 + r) * Cd + c0 + c4 + j];
          }
        }
        *reinterpret_cast<f32x4*>(&ds[r][c4]) = v;
      }
      __syncthreads();
#pragma unroll 2
      for (int c = 0; c < VQ_KC; c += 4) {
        const float x0 = xs[c][pos], x1 = xs[c + 1][pos], x2 = xs[c + 2][pos], x3 = xs[c + 3][pos];
        xn = fmaf(x0, x0, xn);
        xn = fmaf(x1, x1, xn);
        xn = fmaf(x2, x2, xn);
        xn = fmaf(x3, x3, xn);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const f32x4 e = *reinterpret_cast<const f32x4*>(&ds[cg * 16 + j][c]);
          acc[j] = fmaf(x0, e[0], acc[j]);
          acc[j] = fmaf(x1, e[1], acc[j]);
          acc[j] = fmaf(x2, e[2], acc[j]);
          acc[j] = fmaf(x3, e[3], acc[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int code = k0 + cg * 16 + j;
      if (code < K) {
        const float d = (-2.0f * acc[j] + en[code]) + xn;
        if (d < best) {
          best = d;
          best_i = code;
        }
      }
    }
  }
  bd[cg][pos] = best;
  bi[cg][pos] = best_i;
  __syncthreads();
  if (tid < VQ_POS && t0 + tid < T1) {
    float d = bd[0][tid];
    int i = bi[0][tid];
    for (int g = 1; g < 8; ++g) {
      const float dg = bd[g][tid];
      const int ig = bi[g][tid];
      if (dg < d || (dg == d && ig < i)) {
        d = dg;
        i = ig;
      }
    }
    idx_out[(size_t)b * T1 + t0 + tid] = i;
  }
}

__global__ __launch_bounds__(256) void vq_embed_kernel(const int64_t* idx, const float* dict, float* out, int Cd, int T1, int K) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T1) return;
  long long k = idx[(size_t)b * T1 + t];
  k = k < 0 ? 0 : (k >= K ? K - 1 : k);
  out[((size_t)b * Cd + c) * T1 + t] = dict[(size_t)k * Cd + c];
}

}  // namespace

int run_randn(float* out, int B, int T, uint64_t seed, uint64_t clip_offset, uint32_t stream_id, hipStream_t st) {
  dim3 grid(((T + 3) / 4 + 255) / 256, B);
  hipLaunchKernelGGL(randn_kernel, grid, dim3(256), 0, st, out, T, seed, clip_offset, stream_id);
  VQVS_HIP(hipGetLastError());
  return 0;
}

int ddpm_scratch_doubles(int B, int T) { return B * ((T + SUM_CHUNK - 1) / SUM_CHUNK); }

int run_ddpm_step(const float* x_t, const float* eps, const float* noise, const float* a_t, const float* a_prev, float* out,
                  double* scratch, int B, int T, uint32_t flags, float noise_scale, uint64_t seed, uint64_t clip_offset,
                  uint32_t step_index, hipStream_t st) {
  const int nchunk = (T + SUM_CHUNK - 1) / SUM_CHUNK;
  if (flags & 2u) {
    hipLaunchKernelGGL(ddpm_x0sum_kernel, dim3(nchunk, B), dim3(256), 0, st, x_t, eps, a_t, scratch, T, nchunk);
  }
  dim3 grid(((T + 3) / 4 + 255) / 256, B);
  hipLaunchKernelGGL(ddpm_step_kernel, grid, dim3(256), 0, st, x_t, eps, noise, a_t, a_prev, scratch, nchunk, out, T, flags,
                     noise_scale, seed, clip_offset, step_index);
  VQVS_HIP(hipGetLastError());
  return 0;
}

int run_ddpm_mean(const float* x_t, const float* eps, const float* a_t, const float* a_prev, float* out, int B, int T, hipStream_t st) {
  hipLaunchKernelGGL(ddpm_mean_kernel, dim3((T + 255) / 256, B), dim3(256), 0, st, x_t, eps, a_t, a_prev, out, T);
  VQVS_HIP(hipGetLastError());
  return 0;
}

int run_ddpm_guided_eps(const float* x_t, const float* mean, const float* grad, const float* a_t, const float* a_prev, float* out,
                        int B, int T, uint32_t flags, hipStream_t st) {
  hipLaunchKernelGGL(ddpm_guided_eps_kernel, dim3((T + 255) / 256, B), dim3(256), 0, st, x_t, mean, grad, a_t, a_prev, out, T, flags);
  VQVS_HIP(hipGetLastError());
  return 0;
}

int run_vq_argmin(const float* z, const float* dict, float* en_scratch, int64_t* idx, int B, int Cd, int T1, int K, hipStream_t st) {
  hipLaunchKernelGGL(vq_norms_kernel, dim3((K + 255) / 256), dim3(256), 0, st, dict, en_scratch, K, Cd);
  hipLaunchKernelGGL(vq_argmin_kernel, dim3((T1 + VQ_POS - 1) / VQ_POS, B), dim3(256), 0, st, z, dict, en_scratch, idx, Cd, T1, K);
  VQVS_HIP(hipGetLastError());
  return 0;
}

int run_vq_embed(const int64_t* idx, const float* dict, float* out, int B, int Cd, int T1, int K, hipStream_t st) {
  hipLaunchKernelGGL(vq_embed_kernel, dim3((T1 + 255) / 256, Cd, B), dim3(256), 0, st, idx, dict, out, Cd, T1, K);
  VQVS_HIP(hipGetLastError());
  return 0;
}

}  // namespace vqvs
