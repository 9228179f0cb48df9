// Bandwidth-bound and tiny kernels around the MFMA convolutions: the 1->C input conv, the
// C->1 output conv, GroupNorm/FiLM coefficient preparation, the timestep embedding, and the
// NCT<->NTC layout changes at the library boundary.
#include <atomic>

#include "kernels.hpp"

namespace vqvs {

namespace {

// ------------------------------------------------------------------------------------
// in_conv: out[b][t][c] = bias[c] + sum_k w[c][k] x[b][t+k-1]  (+ condp[b][t/rate][c])
// One thread = one time row x 8 channels; a workgroup = 256 rows = one statistics tile.
// Pure write-bound: 4 B read, C*sizeof(T) B written per row.
// ------------------------------------------------------------------------------------
// MC: more than one input channel (a template flag: the mono kernel, which every caller of the reference runs, stays as it was)
template <typename T, bool MC>
__global__ __launch_bounds__(256) void in_conv_kernel(const InConvArgs a) {
  __shared__ float red[256 * 8 * 2];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * STAT_TILE;
  const int opr = a.C >> 3;      // octets per row
  const int rpp = 256 / opr;     // rows per pass
  const int oct = tid % opr;
  const int r0 = tid / opr;
  const int c = oct * 8;
  float w0[8], w1[8], w2[8], bs[8];
  const int Cin = MC ? a.Cin : 1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {  // input channel 0 in registers (the only one, for every caller of the reference)
    w0[j] = a.w[(c + j) * Cin * 3 + 0];
    w1[j] = a.w[(c + j) * Cin * 3 + 1];
    w2[j] = a.w[(c + j) * Cin * 3 + 2];
    bs[j] = a.bias[c + j];
  }
  const float* xb = a.x + (size_t)b * Cin * a.T;
  f32x8 s1 = f32x8_zero(), s2 = f32x8_zero();
  const bool active = r0 < rpp && oct < opr;
  // nearest-neighbour source row exactly as PyTorch's upsample_nearest1d computes it: min(floor(dst * (float)in / out), in - 1)
  const float cond_scale = a.condp ? (float)a.cond_len / (float)a.T : 0.f;
  if (active) {
    for (int r = r0; r < STAT_TILE; r += rpp) {
      const int t = t0 + r;
      if (t >= a.T) break;
      const float xm = t > 0 ? xb[t - 1] : 0.f;
      const float x0 = xb[t];
      const float xp = t + 1 < a.T ? xb[t + 1] : 0.f;
      f32x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaf(w2[j], xp, fmaf(w1[j], x0, fmaf(w0[j], xm, bs[j])));
      if constexpr (MC)
      for (int ci = 1; ci < Cin; ++ci) {  // further input channels (unet.py:25 allows them): weights from L1, in channel order
        const float* xc = xb + (size_t)ci * a.T;
        const float ym = t > 0 ? xc[t - 1] : 0.f, y0 = xc[t], yp = t + 1 < a.T ? xc[t + 1] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float* wc = a.w + ((c + j) * Cin + ci) * 3;
          v[j] = fmaf(wc[2], yp, fmaf(wc[1], y0, fmaf(wc[0], ym, v[j])));
        }
      }
      if (a.condp) {
        const int cr = min((int)floorf((float)t * cond_scale), a.cond_len - 1);
        const T* cp = reinterpret_cast<const T*>(a.condp) + ((size_t)b * a.cond_len + cr) * a.C + c;
        v += Elem<T>::load8(cp);
      }
      s1 += v;
      s2 += v * v;
      Elem<T>::store8(reinterpret_cast<T*>(a.out) + ((size_t)b * a.T + t) * a.C + c, v);
    }
  }
  // deterministic tile statistics
  if (active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(r0 * a.C + c + j) * 2 + 0] = s1[j];
      red[(r0 * a.C + c + j) * 2 + 1] = s2[j];
    }
  }
  __syncthreads();
  if (tid < a.C) {
    float t1 = 0.f, t2 = 0.f;
    for (int g = 0; g < rpp; ++g) {
      t1 += red[(g * a.C + tid) * 2 + 0];
      t2 += red[(g * a.C + tid) * 2 + 1];
    }
    float* o = a.stats + (((size_t)b * a.ntiles + blockIdx.x) * a.C + tid) * 2;
    o[0] = t1;
    o[1] = t2;
  }
}

// ------------------------------------------------------------------------------------
// out_conv: y[b][t] = bias + sum_k sum_c w[k][c] * gelu(x[b][t+k-1][c]*scale+shift)
// Pure read-bound.  Each row's three tap dot-products are formed once (8 channels per lane,
// shuffle-reduced over the row's lanes), parked in LDS, then combined across neighbours.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void out_conv_kernel(const OutConvArgs a) {
  constexpr int GQ = GeluQ<T>::q;
  __shared__ float y[3][STAT_TILE + 2];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * STAT_TILE;
  const int opr = a.C >> 3;  // lanes per row (power of two: 4, 8, 16, ...)
  const int rpp = 256 / opr;
  const int oct = tid % opr;
  const int r0 = tid / opr;
  const int c = oct * 8;
  f32x8 sc, sh, w0, w1, w2;
  {
    const float2* p = a.ss + (size_t)b * a.C + c;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float2 q = p[j];
      sc[j] = q.x;
      sh[j] = q.y;
      w0[j] = a.w[0 * a.C + c + j];
      w1[j] = a.w[1 * a.C + c + j];
      w2[j] = a.w[2 * a.C + c + j];
    }
  }
  const T* xb = reinterpret_cast<const T*>(a.in) + (size_t)b * a.L * a.C + c;
  for (int r = r0; r < STAT_TILE + 2; r += rpp) {
    const int t = t0 - 1 + r;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if (t >= 0 && t < a.L) {
      const f32x8 v = Elem<T>::load8(xb + (size_t)t * a.C);
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const f32x2 g = gelu2<GQ>(fma2(f32x2{v[j], v[j + 1]}, f32x2{sc[j], sc[j + 1]}, f32x2{sh[j], sh[j + 1]}));
        p0 = fmaf(w0[j + 1], g[1], fmaf(w0[j], g[0], p0));
        p1 = fmaf(w1[j + 1], g[1], fmaf(w1[j], g[0], p1));
        p2 = fmaf(w2[j + 1], g[1], fmaf(w2[j], g[0], p2));
      }
    }
    for (int m = 1; m < opr; m <<= 1) {
      p0 += __shfl_xor(p0, m);
      p1 += __shfl_xor(p1, m);
      p2 += __shfl_xor(p2, m);
    }
    if (oct == 0) {
      y[0][r] = p0;
      y[1][r] = p1;
      y[2][r] = p2;
    }
  }
  __syncthreads();
  const int t = t0 + tid;
  if (t < a.L) a.out[(size_t)b * a.L + t] = a.bias + y[0][tid] + y[1][tid + 1] + y[2][tid + 2];
}

// out_conv for widths whose octet count is not a power of two (base_channels 96, 160, ...: the shuffle reduction above needs a row's
// lanes to be a power-of-two group inside one wave): one thread = one row, all channels; same arithmetic per element, the channel
// sum in channel order.  A fallback for unusual widths, not a tuned kernel.
template <typename T>
__global__ __launch_bounds__(256) void out_conv_rows_kernel(const OutConvArgs a) {
  constexpr int GQ = GeluQ<T>::q;
  __shared__ float y[3][STAT_TILE + 2];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * STAT_TILE;
  const float2* ss = a.ss + (size_t)b * a.C;
  for (int r = tid; r < STAT_TILE + 2; r += 256) {
    const int t = t0 - 1 + r;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if (t >= 0 && t < a.L) {
      const T* xr = reinterpret_cast<const T*>(a.in) + ((size_t)b * a.L + t) * a.C;
      for (int c = 0; c < a.C; c += 8) {
        const f32x8 v = Elem<T>::load8(xr + c);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const float2 q0 = ss[c + j], q1 = ss[c + j + 1];
          const f32x2 g = gelu2<GQ>(fma2(f32x2{v[j], v[j + 1]}, f32x2{q0.x, q1.x}, f32x2{q0.y, q1.y}));
          p0 = fmaf(a.w[0 * a.C + c + j + 1], g[1], fmaf(a.w[0 * a.C + c + j], g[0], p0));
          p1 = fmaf(a.w[1 * a.C + c + j + 1], g[1], fmaf(a.w[1 * a.C + c + j], g[0], p1));
          p2 = fmaf(a.w[2 * a.C + c + j + 1], g[1], fmaf(a.w[2 * a.C + c + j], g[0], p2));
        }
      }
    }
    y[0][r] = p0;
    y[1][r] = p1;
    y[2][r] = p2;
  }
  __syncthreads();
  const int t = t0 + tid;
  if (t < a.L) a.out[(size_t)b * a.L + t] = a.bias + y[0][tid] + y[1][tid + 1] + y[2][tid + 2];
}

// ------------------------------------------------------------------------------------
// GroupNorm (+FiLM) coefficients.  One workgroup per clip.  Tile partials are summed in a
// fixed order in fp64, so the result does not depend on scheduling.
//   scale = rstd*gamma*(a+1),  shift = (beta - mean*rstd*gamma)*(a+1) + b
// ------------------------------------------------------------------------------------
constexpr int GN_SPLIT = 8;  // workgroups per clip (each owns groups/GN_SPLIT whole groups)

__global__ __launch_bounds__(256) void gn_prepare_kernel(const GnArgs a) {
  __shared__ double part[256 * 2];
  __shared__ double chs[256], chq[256];
  __shared__ double gmean[32], grstd[32];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int gs = a.Ctot / a.groups;                 // channels per group
  const int ng = a.groups / gridDim.y;              // groups owned by this workgroup
  const int c_lo = blockIdx.y * ng * gs;            // first channel owned
  const int cw_tot = ng * gs;                       // channels owned
  const int C0 = a.src[0].C;
  // this thread's affine parameters and FiLM row entries (used at the very end) are requested now, so that their latency passes
  // behind the reduction instead of adding a second round trip to global memory after it (cw_tot <= 256: one channel per thread)
  float pg = 0.f, pb = 0.f, pfa = 0.f, pfb = 0.f;
  if (tid < cw_tot) {
    pg = a.gamma[c_lo + tid];
    pb = a.beta[c_lo + tid];
    if (a.film) {
      const float* f = a.film + (size_t)b * a.film_stride + a.film_off;
      pfa = f[c_lo + tid];
      pfb = f[a.Ctot + c_lo + tid];
    }
  }
  for (int c0 = 0; c0 < cw_tot; c0 += 256) {
    const int cw = min(256, cw_tot - c0);
    const int nsl = 256 / cw;  // tile slices per channel
    const int c = c_lo + c0 + tid % cw, sl = tid / cw;
    double s1 = 0.0, s2 = 0.0;
    unsigned bad = 0;
    if (sl < nsl) {
      const bool second = c >= C0;
      const GnSrc g = second ? a.src[1] : a.src[0];
      const int cl = second ? c - C0 : c;
      const float* p = g.partials + (size_t)b * g.ntiles * g.C * 2;
      // eight independent loads in flight per thread (the partials of a long clip are 250 tiles deep: one load per iteration
      // made this kernel a chain of memory latencies); the additions keep their order, so the result is unchanged
      for (int t0 = sl; t0 < g.ntiles; t0 += 8 * nsl) {
        float2 q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int t = t0 + k * nsl;
          q[k] = t < g.ntiles ? *reinterpret_cast<const float2*>(p + ((size_t)t * g.C + cl) * 2) : float2{0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s1 += (double)q[k].x;
          s2 += (double)q[k].y;
          // range guard (the statistics are fp32 sums of the stored values): a non-finite partial means an activation overflowed
          // the storage type; in the fp16 mode a tile whose sum of squares reaches 9e8 may hold an element beyond 3e4 (of 65504)
          if (!(fabsf(q[k].x) <= 3.0e38f) || !(q[k].y <= 3.0e38f)) bad |= 1u;
          if (a.guard && q[k].y >= 9.0e8f) bad |= 2u;
        }
      }
      part[tid * 2] = s1;
      part[tid * 2 + 1] = s2;
    }
    if (bad && a.status) atomicOr(a.status, bad);
    __syncthreads();
    // slices of a channel: pairwise tree of fixed shape (deterministic), log2(nsl) steps instead of nsl serial LDS reads
    int top = 1;
    while (top < nsl) top <<= 1;
    for (int st = top >> 1; st >= 1; st >>= 1) {
      if (sl < st && sl + st < nsl) {
        part[tid * 2] += part[(tid + st * cw) * 2];
        part[tid * 2 + 1] += part[(tid + st * cw) * 2 + 1];
      }
      __syncthreads();
    }
    if (tid < cw) {
      chs[c0 + tid] = part[tid * 2];
      chq[c0 + tid] = part[tid * 2 + 1];
    }
    __syncthreads();
  }
  if (tid < ng) {
    double t1 = 0.0, t2 = 0.0;
    for (int j = 0; j < gs; ++j) {
      t1 += chs[tid * gs + j];
      t2 += chq[tid * gs + j];
    }
    const double mean = t1 * a.inv_count;
    double var = t2 * a.inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    gmean[tid] = mean;
    grstd[tid] = 1.0 / sqrt(var + 1e-5);
  }
  __syncthreads();
  for (int i = tid; i < cw_tot; i += 256) {
    const int c = c_lo + i;
    const int g = i / gs;
    const bool pre = i == tid;  // (always, while cw_tot <= 256; a wider share falls back to loading here)
    double scale = grstd[g] * (double)(pre ? pg : a.gamma[c]);
    double shift = (double)(pre ? pb : a.beta[c]) - gmean[g] * scale;
    if (a.film) {
      const float* f = a.film + (size_t)b * a.film_stride + a.film_off;
      const double fa = (double)(pre ? pfa : f[c]) + 1.0;
      const double fb = (double)(pre ? pfb : f[a.Ctot + c]);
      scale *= fa;
      shift = shift * fa + fb;
    }
    a.ss[(size_t)b * a.Ctot + c] = make_float2((float)scale, (float)shift);
    if (a.mr) a.mr[(size_t)b * a.Ctot + c] = make_float2((float)gmean[g], (float)grstd[g]);
  }
}

// ------------------------------------------------------------------------------------
// Timestep embedding: one workgroup per clip, wave-per-output-row mat-vecs.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// w1 / w2 are stored TRANSPOSED ([in][out]) so that thread r's reads of column r coalesce across the wave.
__global__ __launch_bounds__(256) void time_embed_kernel(const TimeEmbedArgs a) {
  __shared__ float e0[1024], e1[1024];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int E = a.E, half = E >> 1;
  const float t = a.ts[b];
  for (int j = tid; j < half; j += 256) {
    const float arg = t * a.freqs[j];
    e0[j] = cosf(arg);
    e0[half + j] = sinf(arg);
  }
  __syncthreads();
  for (int r = tid; r < E; r += 256) {  // time_embed.proj, then time_embed_extra.0 = GELU
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int j = 0; j < E; j += 4) {
      acc0 = fmaf(a.w1[(size_t)(j + 0) * E + r], e0[j + 0], acc0);
      acc1 = fmaf(a.w1[(size_t)(j + 1) * E + r], e0[j + 1], acc1);
      acc2 = fmaf(a.w1[(size_t)(j + 2) * E + r], e0[j + 2], acc2);
      acc3 = fmaf(a.w1[(size_t)(j + 3) * E + r], e0[j + 3], acc3);
    }
    e1[r] = gelu_f((acc0 + acc1) + (acc2 + acc3) + a.b1[r]);
  }
  __syncthreads();
  const float* ce = nullptr;
  if (a.class_embed) {
    long long lab = a.labels[b];
    if (lab < 0) lab = 0;
    if (lab >= a.num_labels) lab = a.num_labels - 1;
    ce = a.class_embed + (size_t)lab * E;
  }
  for (int r = tid; r < E; r += 256) {  // time_embed_extra.1 (+ class embedding)
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int j = 0; j < E; j += 4) {
      acc0 = fmaf(a.w2[(size_t)(j + 0) * E + r], e1[j + 0], acc0);
      acc1 = fmaf(a.w2[(size_t)(j + 1) * E + r], e1[j + 1], acc1);
      acc2 = fmaf(a.w2[(size_t)(j + 2) * E + r], e1[j + 2], acc2);
      acc3 = fmaf(a.w2[(size_t)(j + 3) * E + r], e1[j + 3], acc3);
    }
    float v = (acc0 + acc1) + (acc2 + acc3) + a.b2[r];
    if (ce) v += ce[r];
    a.emb[(size_t)b * E + r] = v;
    a.gemb[(size_t)b * E + r] = gelu_f(v);
  }
}

__global__ void gelu_rows_kernel(const float* in, float* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = gelu_f(in[i]);
}

// film[b][r] = bias[r] + W[r] . gemb[b].  W is stored TRANSPOSED ([E][R]): a workgroup owns 64 output rows r (lane = row: coalesced
// weight reads) of 32 clips; its four waves split E, each accumulating 32 clips in registers against gemb staged in LDS as
// [e][32 clips], and the four partial sums are added in wave order through LDS (deterministic).  Grid: (R / 64, B / 32).
constexpr int FILM_NB = 32;
__global__ __launch_bounds__(256) void film_kernel(const FilmArgs a, int B) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  float* const g = fsm;                   // [E][FILM_NB]
  float* const red = fsm + a.E * FILM_NB;  // [3][64][FILM_NB + 1]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = blockIdx.x * 64 + lane;
  const int E = a.E, b0 = blockIdx.y * FILM_NB;
  const int nb = min(FILM_NB, B - b0);
  for (int i = threadIdx.x; i < E * FILM_NB; i += 256) {
    const int e = i / FILM_NB, bb = i % FILM_NB;
    g[i] = bb < nb ? a.gemb[(size_t)(b0 + bb) * E + e] : 0.f;
  }
  __syncthreads();
  float acc[FILM_NB];
#pragma unroll
  for (int k = 0; k < FILM_NB; ++k) acc[k] = 0.f;
  const int eq = E / 4;  // (host: E is a multiple of 32)
  if (r < a.R) {
    for (int e0 = wv * eq; e0 < (wv + 1) * eq; e0 += 8) {
      float w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = a.w[(size_t)(e0 + u) * a.R + r];  // 8 independent loads in flight
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4* gp = reinterpret_cast<const f32x4*>(&g[(e0 + u) * FILM_NB]);
#pragma unroll
        for (int q = 0; q < FILM_NB / 4; ++q) {
          const f32x4 gv = gp[q];
          acc[4 * q + 0] = fmaf(w[u], gv[0], acc[4 * q + 0]);
          acc[4 * q + 1] = fmaf(w[u], gv[1], acc[4 * q + 1]);
          acc[4 * q + 2] = fmaf(w[u], gv[2], acc[4 * q + 2]);
          acc[4 * q + 3] = fmaf(w[u], gv[3], acc[4 * q + 3]);
        }
      }
    }
  }
  if (wv > 0) {
#pragma unroll
    for (int k = 0; k < FILM_NB; ++k) red[((wv - 1) * 64 + lane) * (FILM_NB + 1) + k] = acc[k];
  }
  __syncthreads();
  if (wv == 0 && r < a.R) {
    const float bias = a.bias[r];
#pragma unroll
    for (int k = 0; k < FILM_NB; ++k) {
      float v = acc[k];
#pragma unroll
      for (int w2 = 0; w2 < 3; ++w2) v += red[(w2 * 64 + lane) * (FILM_NB + 1) + k];  // fixed order
      if (k < nb) a.film[(size_t)(b0 + k) * a.R + r] = v + bias;
    }
  }
}

// ------------------------------------------------------------------------------------
// Stand-alone prologue for the deep levels: g = gelu(x*scale+shift) (optionally avg-pooled by 2).
// One thread = 8 channels of one output row.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void xform_kernel(const XformArgs a) {
  constexpr int GQ = GeluQ<T>::q;
  const int b = blockIdx.y;
  const int opr = a.C >> 3;
  const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= (long long)a.Lout * opr) return;
  const int t = (int)(item / opr), c = (int)(item % opr) * 8;
  f32x8 sc, sh;
  const float2* p = a.ss + (size_t)b * a.ss_stride + a.ss_c0 + c;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float2 q = p[j];
    sc[j] = q.x;
    sh[j] = q.y;
  }
  const T* in = reinterpret_cast<const T*>(a.in) + (size_t)b * a.Lin * a.C + c;
  auto tr = [&](f32x8 v) {
    f32x8 r;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const f32x2 g = gelu2<GQ>(fma2(f32x2{v[j], v[j + 1]}, f32x2{sc[j], sc[j + 1]}, f32x2{sh[j], sh[j + 1]}));
      r[j] = g[0];
      r[j + 1] = g[1];
    }
    return r;
  };
  f32x8 v;
  if (a.avg) {
    v = (tr(Elem<T>::load8(in + (size_t)(2 * t) * a.C)) + tr(Elem<T>::load8(in + (size_t)(2 * t + 1) * a.C))) * 0.5f;
  } else {
    v = tr(Elem<T>::load8(in + (size_t)t * a.C));
  }
  Elem<T>::store8(reinterpret_cast<T*>(a.out) + ((size_t)b * a.Lout + t) * a.C + c, v);
}

// ------------------------------------------------------------------------------------
// NCT float32 <-> NTC T, 32x32 tiles through LDS (used only at the library boundary:
// conditioning input, encoder output, unit-test handles, debug taps).
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nct_to_ntc_kernel(const float* in, T* out, int C, int L) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    tile[i][tx] = (c < C && t < L) ? in[((size_t)b * C + c) * L + t] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    if (t < L && c < C) out[((size_t)b * L + t) * C + c] = (T)tile[tx][i];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ntc_to_nct_kernel(const T* in, float* out, int C, int L) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    tile[i][tx] = (c < C && t < L) ? (float)in[((size_t)b * L + t) * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    if (t < L && c < C) out[((size_t)b * C + c) * L + t] = tile[tx][i];
  }
}

// statistics of an NTC tensor (only for tensors that enter the library from outside)
template <typename T>
__global__ __launch_bounds__(256) void ntc_stats_kernel(const T* in, float* stats, int C, int L, int ntiles) {
  const int b = blockIdx.y, tile = blockIdx.x;
  const T* p = in + (size_t)b * L * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s1 = 0.f, s2 = 0.f;
    const int t1 = min(L, (tile + 1) * STAT_TILE);
    for (int t = tile * STAT_TILE; t < t1; ++t) {
      const float v = (float)p[(size_t)t * C + c];
      s1 += v;
      s2 += v * v;
    }
    float* o = stats + (((size_t)b * ntiles + tile) * C + c) * 2;
    o[0] = s1;
    o[1] = s2;
  }
}

}  // namespace

int launch_in_conv(const InConvArgs& a, int B, int precision, hipStream_t st) {
  if (a.C % 8 || a.C > 256) VQVS_FAIL(-1, "in_conv: unsupported C=%d", a.C);  // (an octet count that does not divide 256 leaves the last threads idle)
  dim3 grid((a.T + STAT_TILE - 1) / STAT_TILE, B);
  if (a.Cin > 1) {
    VQVS_BY_PRECISION(precision, hipLaunchKernelGGL((in_conv_kernel<T, true>), grid, dim3(256), 0, st, a));
  } else {
    VQVS_BY_PRECISION(precision, hipLaunchKernelGGL((in_conv_kernel<T, false>), grid, dim3(256), 0, st, a));
  }
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_out_conv(const OutConvArgs& a, int B, int precision, hipStream_t st) {
  const int opr = a.C / 8;
  if (a.C % 8 || opr > 64) VQVS_FAIL(-1, "out_conv: unsupported C=%d", a.C);
  dim3 grid((a.L + STAT_TILE - 1) / STAT_TILE, B);
  if (opr & (opr - 1)) {
    VQVS_BY_PRECISION(precision, hipLaunchKernelGGL(out_conv_rows_kernel<T>, grid, dim3(256), 0, st, a));
  } else {
    VQVS_BY_PRECISION(precision, hipLaunchKernelGGL(out_conv_kernel<T>, grid, dim3(256), 0, st, a));
  }
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_gn_prepare(const GnArgs& a, int B, hipStream_t st) {
  if (a.Ctot > 2048 || a.groups > 32 || a.Ctot % a.groups) VQVS_FAIL(-1, "gn: unsupported Ctot=%d groups=%d", a.Ctot, a.groups);
  int split = GN_SPLIT;
  while (split > 1 && (a.groups % split || (a.groups / split) * (a.Ctot / a.groups) > 256)) split >>= 1;
  if ((a.groups / split) * (a.Ctot / a.groups) > 256 && a.Ctot > 256) split = a.groups;  // one group per workgroup
  hipLaunchKernelGGL(gn_prepare_kernel, dim3(B, split), dim3(256), 0, st, a);
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_time_embed(const TimeEmbedArgs& a, int B, hipStream_t st) {
  if (a.E > 1024 || a.E % 64) VQVS_FAIL(-1, "time_embed: unsupported E=%d", a.E);
  hipLaunchKernelGGL(time_embed_kernel, dim3(B), dim3(256), 0, st, a);
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_gelu_rows(const float* in, float* out, int n, hipStream_t st) {
  hipLaunchKernelGGL(gelu_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, in, out, n);
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_film(const FilmArgs& a, int B, hipStream_t st) {
  if (a.E > 1024 || a.E < 32 || a.E % 32) VQVS_FAIL(-1, "film: unsupported E=%d", a.E);
  const int lds = (a.E * FILM_NB + 3 * 64 * (FILM_NB + 1)) * 4;  // <= 153 KiB at E = 1024 (the attribute below covers it)
  static std::atomic<bool> attr_done[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_done[dev].load(std::memory_order_acquire)) {
    VQVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&film_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(film_kernel, dim3((a.R + 63) / 64, (B + FILM_NB - 1) / FILM_NB), dim3(256), lds, st, a, B);
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_xform(const XformArgs& a, int B, int precision, hipStream_t st) {
  if (a.C % 8) VQVS_FAIL(-1, "xform: unsupported C=%d", a.C);
  const long long items = (long long)a.Lout * (a.C / 8);
  dim3 grid((unsigned)((items + 255) / 256), B);
  VQVS_BY_PRECISION(precision, hipLaunchKernelGGL(xform_kernel<T>, grid, dim3(256), 0, st, a));
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_nct_to_ntc(const float* in, void* out, float* stats, int B, int C, int L, int ntiles, int precision, hipStream_t st) {
  dim3 grid((L + 31) / 32, (C + 31) / 32, B);
  VQVS_BY_PRECISION(precision, {
    hipLaunchKernelGGL(nct_to_ntc_kernel<T>, grid, dim3(256), 0, st, in, (T*)out, C, L);
    if (stats) hipLaunchKernelGGL(ntc_stats_kernel<T>, dim3(ntiles, B), dim3(256), 0, st, (const T*)out, stats, C, L, ntiles);
  });
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_ntc_to_nct(const void* in, float* out, int B, int C, int L, int in_precision, hipStream_t st) {
  dim3 grid((L + 31) / 32, (C + 31) / 32, B);
  VQVS_BY_PRECISION(in_precision, hipLaunchKernelGGL(ntc_to_nct_kernel<T>, grid, dim3(256), 0, st, (const T*)in, out, C, L));
  VQVS_HIP(hipGetLastError());
  return 0;
}

}  // namespace vqvs
