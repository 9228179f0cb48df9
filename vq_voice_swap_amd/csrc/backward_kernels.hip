// Input-gradient kernels for classifier guidance (reference sample_diffusion.py:34-42 calls
// torch.autograd.grad through models/classifier.py; here the backward pass with respect to the
// waveform is an explicit schedule, see net.cpp "classifier").  The transposed convolutions reuse the
// forward MFMA kernel; this file holds the bandwidth-bound element-wise pieces and the per-clip head.
//
// GroupNorm backward (per clip, group of n = channels_per_group * L elements), with the forward written as
// u = scale_c * x + shift_c  (scale_c = rstd * gamma'_c, gamma' includes FiLM):
//   dx = scale_c * du  -  rstd * A / n  -  rstd^2 * (x - mean) * Bsum / n
//   A    = sum over the group of gamma'_c * du          = sum_c gamma'_c * S1_c
//   Bsum = sum over the group of gamma'_c * du * xhat   = sum_c (S2_c - beta'_c * S1_c)
//   S1_c = sum_t du,   S2_c = sum_t du * u              (what bw_act emits per tile)
// i.e. dx = P_c * du + Q_c * x + R_c with per-(clip, channel) coefficients.
#include "kernels.hpp"

namespace vqvs {

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// ------------------------------------------------------------------------------------
// bw_act: one workgroup = one (clip, tile of STAT_TILE rows); a thread owns 8 channels of every rpp-th row.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bw_act_kernel(const BwActArgs a) {
  __shared__ float red[256 * 8 * 2];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * STAT_TILE;
  const int opr = a.C >> 3;   // octets per row (<= 128)
  const int rpp = 256 / opr;  // rows per pass
  const int oct = tid % opr;
  const int r0 = tid / opr;
  const int c = oct * 8;
  f32x8 sc, sh;
  {
    const float2* p = a.ss + (size_t)b * a.ss_stride + a.ss_c0 + c;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float2 q = p[j];
      sc[j] = q.x;
      sh[j] = q.y;
    }
  }
  const int Lt = a.resize == BW_FROM_HALF ? (a.L >> 1) : (a.resize == BW_FROM_DOUBLE ? 2 * a.L : a.L);
  const T* tb = reinterpret_cast<const T*>(a.t) + (size_t)b * Lt * a.t_C + a.t_c0 + c;
  const T* xb = reinterpret_cast<const T*>(a.xf) + (size_t)b * a.L * a.C + c;
  T* ob = reinterpret_cast<T*>(a.du) + (size_t)b * a.L * a.du_C + a.du_c0 + c;
  f32x8 s1 = f32x8_zero(), s2 = f32x8_zero();
  if (r0 < rpp) {
    for (int r = r0; r < STAT_TILE; r += rpp) {
      const int t = t0 + r;
      if (t >= a.L) break;
      f32x8 g;
      if (a.resize == BW_FROM_HALF) g = Elem<T>::load8(tb + (size_t)(t >> 1) * a.t_C) * 0.5f;
      else if (a.resize == BW_FROM_DOUBLE) g = Elem<T>::load8(tb + (size_t)(2 * t) * a.t_C) + Elem<T>::load8(tb + (size_t)(2 * t + 1) * a.t_C);
      else g = Elem<T>::load8(tb + (size_t)t * a.t_C);
      const f32x8 x = Elem<T>::load8(xb + (size_t)t * a.C);
      f32x8 du;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float u = fmaf(x[j], sc[j], sh[j]);
        du[j] = g[j] * gelu_grad_f(u);
        s1[j] += du[j];
        s2[j] = fmaf(du[j], u, s2[j]);
      }
      Elem<T>::store8(ob + (size_t)t * a.du_C, du);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(r0 * a.C + c + j) * 2 + 0] = s1[j];
      red[(r0 * a.C + c + j) * 2 + 1] = s2[j];
    }
  }
  __syncthreads();
  for (int cc = tid; cc < a.C; cc += 256) {
    float q1 = 0.f, q2 = 0.f;
    for (int g = 0; g < rpp; ++g) {  // fixed order: deterministic
      q1 += red[(g * a.C + cc) * 2 + 0];
      q2 += red[(g * a.C + cc) * 2 + 1];
    }
    float* o = a.partials + (((size_t)b * gridDim.x + blockIdx.x) * a.part_C + a.part_c0 + cc) * 2;
    o[0] = q1;
    o[1] = q2;
  }
}

// ------------------------------------------------------------------------------------
// gn_bw: one workgroup per clip; tile partials summed in a fixed order in fp64.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_bw_kernel(const GnBwArgs a) {
  __shared__ double part[256 * 2];
  __shared__ double chs[2048], chq[2048];  // (up to 2 x 1024 channels: the concatenated input of the widest up block)
  __shared__ double gA[32], gB[32];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int gs = a.C / a.groups;
  const float* p = a.partials + (size_t)b * a.ntiles * a.C * 2;
  // (this thread's first forward coefficients, used at the very end, are requested before the reduction)
  const float2 ss_pre = tid < a.C ? a.ss[(size_t)b * a.C + tid] : float2{0.f, 0.f};
  const float2 mr_pre = tid < a.C ? a.mr[(size_t)b * a.C + tid] : float2{0.f, 0.f};
  for (int c0 = 0; c0 < a.C; c0 += 256) {
    const int cw = min(256, a.C - c0);
    const int nsl = 256 / cw;
    const int c = c0 + tid % cw, sl = tid / cw;
    if (sl < nsl) {
      double s1 = 0.0, s2 = 0.0;
      // eight independent loads in flight per thread (as gn_prepare_kernel: one load per iteration made this a chain of memory
      // latencies, 32 deep at the top level); the additions keep their order
      for (int t0 = sl; t0 < a.ntiles; t0 += 8 * nsl) {
        float2 q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int t = t0 + k * nsl;
          q[k] = t < a.ntiles ? *reinterpret_cast<const float2*>(p + ((size_t)t * a.C + c) * 2) : float2{0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s1 += (double)q[k].x;
          s2 += (double)q[k].y;
        }
      }
      part[tid * 2] = s1;
      part[tid * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < cw) {
      double t1 = 0.0, t2 = 0.0;
      for (int k = 0; k < nsl; ++k) {
        t1 += part[(k * cw + tid) * 2];
        t2 += part[(k * cw + tid) * 2 + 1];
      }
      chs[c0 + tid] = t1;
      chq[c0 + tid] = t2;
    }
    __syncthreads();
  }
  const float2* ss = a.ss + (size_t)b * a.C;
  const float2* mr = a.mr + (size_t)b * a.C;
  if (tid < a.groups) {
    const double mean = (double)mr[tid * gs].x, rstd = (double)mr[tid * gs].y;
    double A = 0.0, Bs = 0.0;
    for (int j = 0; j < gs; ++j) {
      const int c = tid * gs + j;
      const double scale = (double)ss[c].x, shift = (double)ss[c].y;
      const double gam = scale / rstd;          // gamma'
      const double bet = shift + mean * scale;  // beta'
      A += gam * chs[c];
      Bs += chq[c] - bet * chs[c];
    }
    gA[tid] = A * a.inv_count;
    gB[tid] = Bs * a.inv_count;
  }
  __syncthreads();
  for (int c = tid; c < a.C; c += 256) {
    const int g = c / gs;
    const float2 mrc = c == tid ? mr_pre : mr[c], ssc = c == tid ? ss_pre : ss[c];
    const double mean = (double)mrc.x, rstd = (double)mrc.y;
    const double P = (double)ssc.x;
    const double Q = -rstd * rstd * gB[g];
    const double R = -rstd * gA[g] - Q * mean;
    a.coef[(size_t)b * a.C + c] = make_float4((float)P, (float)Q, (float)R, 0.f);
  }
}

// ------------------------------------------------------------------------------------
// bw_affine: one thread = 8 channels of one row.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bw_affine_kernel(const BwAffineArgs a) {
  const int b = blockIdx.y;
  const int opr = a.C >> 3;
  const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= (long long)a.L * opr) return;
  const int t = (int)(item / opr), c = (int)(item % opr) * 8;
  const float4* cf = a.coef + (size_t)b * a.coef_stride + a.coef_c0 + c;
  const size_t idx = ((size_t)b * a.L + t) * a.C + c;
  const f32x8 du = Elem<T>::load8(reinterpret_cast<const T*>(a.du) + ((size_t)b * a.L + t) * a.du_C + a.du_c0 + c);
  const f32x8 x = Elem<T>::load8(reinterpret_cast<const T*>(a.xf) + idx);
  f32x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float4 q = cf[j];
    v[j] = fmaf(q.x, du[j], fmaf(q.y, x[j], q.z));
  }
  if (a.skip) {
    const T* sk = reinterpret_cast<const T*>(a.skip);
    if (a.skip_mode == BW_FROM_HALF)
      v += Elem<T>::load8(sk + ((size_t)b * (a.L >> 1) + (t >> 1)) * a.C + c) * 0.5f;
    else if (a.skip_mode == BW_FROM_DOUBLE)
      v += Elem<T>::load8(sk + ((size_t)b * (2 * a.L) + 2 * t) * a.C + c) + Elem<T>::load8(sk + ((size_t)b * (2 * a.L) + 2 * t + 1) * a.C + c);
    else
      v += Elem<T>::load8(sk + idx);
  }
  if (a.extra) v += Elem<T>::load8(reinterpret_cast<const T*>(a.extra) + ((size_t)b * a.L + t) * a.extra_C + a.extra_c0 + c);
  Elem<T>::store8(reinterpret_cast<T*>(a.out) + idx, v);
}

template <typename T>
__global__ __launch_bounds__(256) void bw_add_kernel(const T* x, const T* y, T* out, long long n8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  Elem<T>::store8(out + i * 8, Elem<T>::load8(x + i * 8) + Elem<T>::load8(y + i * 8));
}

// ------------------------------------------------------------------------------------
// EncoderPredictor head: one workgroup = 64 latent positions of one clip.  lane = position, wave = logit / channel slice.
// ------------------------------------------------------------------------------------
constexpr int EH_POS = 64;
template <typename T>
__global__ __launch_bounds__(256) void enc_head_kernel(const EncHeadArgs a) {
  extern __shared__ float ehs[];
  float* rows = ehs;                      // [EH_POS][Cb + 1]
  float* lg = rows + EH_POS * (a.Cb + 1);  // [D][EH_POS]
  float* red = lg + (size_t)a.D * EH_POS;  // [4][EH_POS]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.y, i0 = blockIdx.x * EH_POS;
  const int npos = min(EH_POS, a.T1 - i0);
  for (int k = tid; k < EH_POS * a.Cb; k += 256) {
    const int p = k / a.Cb, c = k - p * a.Cb;
    rows[p * (a.Cb + 1) + c] = p < npos ? a.o[((size_t)b * a.T + (size_t)(i0 + p) * a.rate) * a.Cb + c] : 0.f;
  }
  __syncthreads();
  const float* row = rows + lane * (a.Cb + 1);
  for (int d = wv; d < a.D; d += 4) {
    const float* w = a.w + (size_t)d * a.Cb;  // wave-uniform
    float acc = a.bias[d];
    for (int c = 0; c < a.Cb; ++c) acc = fmaf(w[c], row[c], acc);
    lg[d * EH_POS + lane] = acc;
    if (a.logits && lane < npos) a.logits[((size_t)b * a.D + d) * a.T1 + i0 + lane] = acc;
  }
  if (!a.targets) return;
  __syncthreads();
  // softmax over the D logits of each position: every wave reduces its slice, then the four slices are combined
  float m = -3.0e38f;
  for (int d = wv; d < a.D; d += 4) m = fmaxf(m, lg[d * EH_POS + lane]);
  red[wv * EH_POS + lane] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[lane], red[EH_POS + lane]), fmaxf(red[2 * EH_POS + lane], red[3 * EH_POS + lane]));
  __syncthreads();
  float sum = 0.f;
  for (int d = wv; d < a.D; d += 4) sum += __expf(lg[d * EH_POS + lane] - m);
  red[wv * EH_POS + lane] = sum;
  __syncthreads();
  sum = (red[lane] + red[EH_POS + lane]) + (red[2 * EH_POS + lane] + red[3 * EH_POS + lane]);
  const float inv = 1.0f / sum;
  long long y = lane < npos ? a.targets[(size_t)b * a.T1 + i0 + lane] : 0;
  if (y < 0) y = 0;
  if (y >= a.D) y = a.D - 1;
  // d(-gscale * CE)/dlogit = gscale * (onehot - softmax)
  for (int d = wv; d < a.D; d += 4) lg[d * EH_POS + lane] = a.gscale * ((d == (int)y ? 1.0f : 0.0f) - __expf(lg[d * EH_POS + lane] - m) * inv);
  __syncthreads();
  if (lane < npos) {
    T* out = reinterpret_cast<T*>(a.dO) + ((size_t)b * a.T + (size_t)(i0 + lane) * a.rate) * a.Cb;
    for (int c = wv; c < a.Cb; c += 4) {
      float acc = 0.f;
      for (int d = 0; d < a.D; ++d) acc = fmaf(a.w[(size_t)d * a.Cb + c], lg[d * EH_POS + lane], acc);
      out[c] = (T)acc;
    }
  }
}

// ------------------------------------------------------------------------------------
// in_conv_bw: per row the three tap dot-products over channels, then combined across neighbours.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void in_conv_bw_kernel(const InConvBwArgs a) {
  __shared__ float y[3][STAT_TILE + 2];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * STAT_TILE;
  const int opr = a.C >> 3;  // lanes per row (power of two <= 64)
  const int rpp = 256 / opr;
  const int oct = tid % opr;
  const int r0 = tid / opr;
  const int c = oct * 8;
  float w0[8], w1[8], w2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    w0[j] = a.w[(c + j) * 3 + 0];
    w1[j] = a.w[(c + j) * 3 + 1];
    w2[j] = a.w[(c + j) * 3 + 2];
  }
  const T* db = reinterpret_cast<const T*>(a.dh) + (size_t)b * a.T * a.C + c;
  for (int r = r0; r < STAT_TILE + 2; r += rpp) {
    const int t = t0 - 1 + r;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if (t >= 0 && t < a.T) {
      const f32x8 v = Elem<T>::load8(db + (size_t)t * a.C);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        p0 = fmaf(w0[j], v[j], p0);
        p1 = fmaf(w1[j], v[j], p1);
        p2 = fmaf(w2[j], v[j], p2);
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
  // forward: h[t][c] += w[c][k] * x[t+k-1]  =>  dx[t] = sum_k p_k[t-k+1];  y[.][r] holds row t0-1+r
  const int t = t0 + tid;
  if (t < a.T) a.out[(size_t)b * a.T + t] = (y[0][tid + 2] + y[1][tid + 1] + y[2][tid]) * a.out_scale;
}

// in_conv_bw for widths whose octet count is not a power of two (base_channels 96, 160, ...: the shuffle reduction above needs a
// row's lanes to be a power-of-two group inside one wave): one thread = one row, all channels, the channel sum in channel order --
// the counterpart of out_conv_rows_kernel (misc_kernels.hip).  A fallback for unusual widths, not a tuned kernel.
template <typename T>
__global__ __launch_bounds__(256) void in_conv_bw_rows_kernel(const InConvBwArgs a) {
  __shared__ float y[3][STAT_TILE + 2];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * STAT_TILE;
  for (int r = tid; r < STAT_TILE + 2; r += 256) {
    const int t = t0 - 1 + r;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if (t >= 0 && t < a.T) {
      const T* row = reinterpret_cast<const T*>(a.dh) + ((size_t)b * a.T + t) * a.C;
      for (int c = 0; c < a.C; c += 8) {
        const f32x8 v = Elem<T>::load8(row + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          p0 = fmaf(a.w[(c + j) * 3 + 0], v[j], p0);
          p1 = fmaf(a.w[(c + j) * 3 + 1], v[j], p1);
          p2 = fmaf(a.w[(c + j) * 3 + 2], v[j], p2);
        }
      }
    }
    y[0][r] = p0;
    y[1][r] = p1;
    y[2][r] = p2;
  }
  __syncthreads();
  const int t = t0 + tid;
  if (t < a.T) a.out[(size_t)b * a.T + t] = (y[0][tid + 2] + y[1][tid + 1] + y[2][tid]) * a.out_scale;
}

// ------------------------------------------------------------------------------------
// classifier head, forward and (optionally) backward.  256 threads per clip.
// ------------------------------------------------------------------------------------
constexpr int HEAD_NT = 1024;  // 16 waves: the head is one workgroup per clip and latency-bound, so breadth is what it needs
constexpr int HEAD_MAXH = 16;

template <typename T>
__global__ __launch_bounds__(HEAD_NT) void cls_head_kernel(const HeadArgs a) {
  extern __shared__ float hs[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NW = HEAD_NT / 64;
  const int b = blockIdx.x;
  const int C = a.C, L = a.L, H = a.heads, F = a.F, NL = a.NL;
  const int ch = C / H;
  const int L1 = L + 1;
  // LDS carve-up (floats)
  float* sc = hs;              // [C]
  float* sh = sc + C;          // [C]
  float* wgt = sh + C;         // [H][L1] attention weights of query token 0
  float* dsc = wgt + H * L1;   // [H][L1] scores, later d(score)
  float* pooled = dsc + H * L1;  // [H][C]  sum_s w_s xhat_s, later d(pooled)
  float* av = pooled + H * C;  // [C]     attention output, later its gradient
  float* feat = av + C;        // [F]     c_proj output, later its gradient
  float* gl = feat + F;        // [F]     gelu(feat)
  float* lg = gl + F;          // [NL]    logits, later their gradient
  float* s1 = lg + NL;         // [C]
  float* s2 = s1 + C;          // [C]
  float* cP = s2 + C;          // [C] x3 (P, Q, R)
  __shared__ float redm[NW], reds[NW];

  const T* hb = reinterpret_cast<const T*>(a.h) + (size_t)b * L * C;
  for (int c = tid; c < C; c += HEAD_NT) {
    const float2 q = a.ss[(size_t)b * C + c];
    sc[c] = q.x;
    sh[c] = q.y;
  }
  __syncthreads();
  auto xhat = [&](int s, int c) { return gelu_f(fmaf((float)hb[(size_t)s * C + c], sc[c], sh[c])); };

  // scores of query token 0 against every key: token 0 (zero input) scores c0[h]
  for (int s = wv; s < L; s += NW) {
    float acc[HEAD_MAXH];
#pragma unroll
    for (int h = 0; h < HEAD_MAXH; ++h) acc[h] = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float x = xhat(s, c);
#pragma unroll
      for (int h = 0; h < HEAD_MAXH; ++h)
        if (h < H) acc[h] = fmaf(a.r[h * C + c], x, acc[h]);
    }
#pragma unroll
    for (int h = 0; h < HEAD_MAXH; ++h)
      if (h < H) {
        const float v = wave_sum_f(acc[h]);
        if (lane == 0) dsc[h * L1 + s + 1] = v + a.c0[h];
      }
  }
  if (tid < H) dsc[tid * L1] = a.c0[tid];
  __syncthreads();
  // softmax over the L+1 keys, one wave per head
  for (int h = wv; h < H; h += NW) {
    float m = -3.0e38f;
    for (int s = lane; s < L1; s += 64) m = fmaxf(m, dsc[h * L1 + s]);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    float sum = 0.f;
    for (int s = lane; s < L1; s += 64) {
      const float e = __expf(dsc[h * L1 + s] - m);
      wgt[h * L1 + s] = e;
      sum += e;
    }
    sum = wave_sum_f(sum);
    const float inv = 1.0f / sum;
    for (int s = lane; s < L1; s += 64) wgt[h * L1 + s] *= inv;
  }
  __syncthreads();
  // pooled[h][c] = sum_{s>=1} w[h][s] * xhat_{s-1}[c]
  for (int c = tid; c < C; c += HEAD_NT) {
    float acc[HEAD_MAXH];
#pragma unroll
    for (int h = 0; h < HEAD_MAXH; ++h) acc[h] = 0.f;
    for (int s = 0; s < L; ++s) {
      const float x = xhat(s, c);
#pragma unroll
      for (int h = 0; h < HEAD_MAXH; ++h)
        if (h < H) acc[h] = fmaf(wgt[h * L1 + s + 1], x, acc[h]);
    }
#pragma unroll
    for (int h = 0; h < HEAD_MAXH; ++h)
      if (h < H) pooled[h * C + c] = acc[h];
  }
  __syncthreads();
  // a[j] = bv[j] + Wv[j] . pooled[head(j)]     (the weights sum to one, so bv enters once)
  // forward mat-vecs: thread = output, walking a TRANSPOSED copy of the weights ([in][out]) so that a wave's loads
  // coalesce; four independent partial sums keep several loads in flight
  for (int j = tid; j < C; j += HEAD_NT) {
    const float* pl = pooled + (j / ch) * C;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
    for (int c = 0; c < C; c += 4) {
      a0 = fmaf(a.wvT[(size_t)(c + 0) * C + j], pl[c + 0], a0);
      a1 = fmaf(a.wvT[(size_t)(c + 1) * C + j], pl[c + 1], a1);
      a2 = fmaf(a.wvT[(size_t)(c + 2) * C + j], pl[c + 2], a2);
      a3 = fmaf(a.wvT[(size_t)(c + 3) * C + j], pl[c + 3], a3);
    }
    av[j] = ((a0 + a1) + (a2 + a3)) + a.bv[j];
  }
  __syncthreads();
  for (int f = tid; f < F; f += HEAD_NT) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
    for (int c = 0; c < C; c += 4) {
      a0 = fmaf(a.wcT[(size_t)(c + 0) * F + f], av[c + 0], a0);
      a1 = fmaf(a.wcT[(size_t)(c + 1) * F + f], av[c + 1], a1);
      a2 = fmaf(a.wcT[(size_t)(c + 2) * F + f], av[c + 2], a2);
      a3 = fmaf(a.wcT[(size_t)(c + 3) * F + f], av[c + 3], a3);
    }
    const float v = ((a0 + a1) + (a2 + a3)) + a.bc[f];
    feat[f] = v;
    gl[f] = gelu_f(v);
  }
  __syncthreads();
  for (int n = tid; n < NL; n += HEAD_NT) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
    for (int f = 0; f < F; f += 4) {
      a0 = fmaf(a.wlT[(size_t)(f + 0) * NL + n], gl[f + 0], a0);
      a1 = fmaf(a.wlT[(size_t)(f + 1) * NL + n], gl[f + 1], a1);
      a2 = fmaf(a.wlT[(size_t)(f + 2) * NL + n], gl[f + 2], a2);
      a3 = fmaf(a.wlT[(size_t)(f + 3) * NL + n], gl[f + 3], a3);
    }
    const float v = ((a0 + a1) + (a2 + a3)) + a.bl[n];
    lg[n] = v;
    a.logits[(size_t)b * NL + n] = v;
  }
  if (!a.labels) return;
  __syncthreads();

  // ---- backward of gscale * log_softmax(logits)[y] ----
  {
    float m = -3.0e38f;
    for (int n = tid; n < NL; n += HEAD_NT) m = fmaxf(m, lg[n]);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    if (lane == 0) redm[wv] = m;
    __syncthreads();
    m = redm[0];
    for (int k = 1; k < NW; ++k) m = fmaxf(m, redm[k]);
    float sum = 0.f;
    for (int n = tid; n < NL; n += HEAD_NT) sum += __expf(lg[n] - m);
    sum = wave_sum_f(sum);
    if (lane == 0) reds[wv] = sum;
    __syncthreads();
    sum = 0.f;
    for (int k = 0; k < NW; ++k) sum += reds[k];
    long long y = a.labels[b];
    if (y < 0) y = 0;
    if (y >= NL) y = NL - 1;
    const float inv = 1.0f / sum;
    __syncthreads();
    for (int n = tid; n < NL; n += HEAD_NT) lg[n] = a.gscale * ((n == (int)y ? 1.0f : 0.0f) - __expf(lg[n] - m) * inv);
  }
  __syncthreads();
  for (int f = tid; f < F; f += HEAD_NT) {  // d feat
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int n = 0;
#pragma unroll 2
    for (; n + 4 <= NL; n += 4) {
      a0 = fmaf(a.wl[(size_t)(n + 0) * F + f], lg[n + 0], a0);
      a1 = fmaf(a.wl[(size_t)(n + 1) * F + f], lg[n + 1], a1);
      a2 = fmaf(a.wl[(size_t)(n + 2) * F + f], lg[n + 2], a2);
      a3 = fmaf(a.wl[(size_t)(n + 3) * F + f], lg[n + 3], a3);
    }
    for (; n < NL; ++n) a0 = fmaf(a.wl[(size_t)n * F + f], lg[n], a0);
    feat[f] = ((a0 + a1) + (a2 + a3)) * gelu_grad_f(feat[f]);
  }
  __syncthreads();
  for (int j = tid; j < C; j += HEAD_NT) {  // d a
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
    for (int f = 0; f < F; f += 4) {
      a0 = fmaf(a.wc[(size_t)(f + 0) * C + j], feat[f + 0], a0);
      a1 = fmaf(a.wc[(size_t)(f + 1) * C + j], feat[f + 1], a1);
      a2 = fmaf(a.wc[(size_t)(f + 2) * C + j], feat[f + 2], a2);
      a3 = fmaf(a.wc[(size_t)(f + 3) * C + j], feat[f + 3], a3);
    }
    av[j] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  for (int c = tid; c < C; c += HEAD_NT) {  // d pooled[h][c] = sum_{j in head h} Wv[j][c] * da[j]
    for (int h = 0; h < H; ++h) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
      for (int j = h * ch; j < (h + 1) * ch; j += 4) {
        a0 = fmaf(a.wv[(size_t)(j + 0) * C + c], av[j + 0], a0);
        a1 = fmaf(a.wv[(size_t)(j + 1) * C + c], av[j + 1], a1);
        a2 = fmaf(a.wv[(size_t)(j + 2) * C + c], av[j + 2], a2);
        a3 = fmaf(a.wv[(size_t)(j + 3) * C + c], av[j + 3], a3);
      }
      pooled[h * C + c] = (a0 + a1) + (a2 + a3);
    }
  }
  __syncthreads();
  // d w[h][s] = dpooled[h] . xhat_{s-1}   (a constant over s drops out of the softmax backward: token 0 -> 0)
  for (int s = wv; s < L; s += NW) {
    float acc[HEAD_MAXH];
#pragma unroll
    for (int h = 0; h < HEAD_MAXH; ++h) acc[h] = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float x = xhat(s, c);
#pragma unroll
      for (int h = 0; h < HEAD_MAXH; ++h)
        if (h < H) acc[h] = fmaf(pooled[h * C + c], x, acc[h]);
    }
#pragma unroll
    for (int h = 0; h < HEAD_MAXH; ++h)
      if (h < H) {
        const float v = wave_sum_f(acc[h]);
        if (lane == 0) dsc[h * L1 + s + 1] = v;
      }
  }
  if (tid < H) dsc[tid * L1] = 0.f;
  __syncthreads();
  for (int h = wv; h < H; h += NW) {  // softmax backward
    float dot = 0.f;
    for (int s = lane; s < L1; s += 64) dot = fmaf(wgt[h * L1 + s], dsc[h * L1 + s], dot);
    dot = wave_sum_f(dot);
    for (int s = lane; s < L1; s += 64) dsc[h * L1 + s] = wgt[h * L1 + s] * (dsc[h * L1 + s] - dot);
  }
  __syncthreads();
  // d xhat_s[c] = sum_h w[h][s+1]*dpooled[h][c] + dscore[h][s+1]*r[h][c];  du = d xhat * gelu'(u)
  auto du_of = [&](int s, int c, float& u) {
    float dx = 0.f;
    for (int h = 0; h < H; ++h) dx = fmaf(wgt[h * L1 + s + 1], pooled[h * C + c], fmaf(dsc[h * L1 + s + 1], a.r[h * C + c], dx));
    u = fmaf((float)hb[(size_t)s * C + c], sc[c], sh[c]);
    return dx * gelu_grad_f(u);
  };
  for (int c = tid; c < C; c += HEAD_NT) {
    float q1 = 0.f, q2 = 0.f;
    for (int s = 0; s < L; ++s) {
      float u;
      const float du = du_of(s, c, u);
      q1 += du;
      q2 = fmaf(du, u, q2);
    }
    s1[c] = q1;
    s2[c] = q2;
  }
  __syncthreads();
  const int gs = C / a.groups;
  if (tid < a.groups) {
    const float2 q = a.mr[(size_t)b * C + tid * gs];
    const double mean = (double)q.x, rstd = (double)q.y;
    double A = 0.0, Bs = 0.0;
    for (int j = 0; j < gs; ++j) {
      const int c = tid * gs + j;
      const double scale = (double)sc[c], shift = (double)sh[c];
      A += (scale / rstd) * (double)s1[c];
      Bs += (double)s2[c] - (shift + mean * scale) * (double)s1[c];
    }
    A *= a.inv_count;
    Bs *= a.inv_count;
    const double Q = -rstd * rstd * Bs;
    const double R = -rstd * A - Q * mean;
    for (int j = 0; j < gs; ++j) {
      const int c = tid * gs + j;
      cP[c] = sc[c];
      cP[C + c] = (float)Q;
      cP[2 * C + c] = (float)R;
    }
  }
  __syncthreads();
  T* ob = reinterpret_cast<T*>(a.dh) + (size_t)b * L * C;
  for (int i = tid; i < L * C; i += HEAD_NT) {
    const int s = i / C, c = i - s * C;
    float u;
    const float du = du_of(s, c, u);
    ob[i] = (T)fmaf(cP[c], du, fmaf(cP[C + c], (float)hb[i], cP[2 * C + c]));
  }
}

size_t head_lds_floats(const HeadArgs& a) {
  const size_t C = a.C, L1 = a.L + 1, H = a.heads, F = a.F, NL = a.NL;
  return 2 * C + 2 * H * L1 + H * C + C + 2 * F + NL + 2 * C + 3 * C;
}

}  // namespace

int launch_bw_act(const BwActArgs& a, int B, int precision, hipStream_t st) {
  const int opr = a.C / 8;
  if (a.C % 8 || opr > 256) VQVS_FAIL(-1, "bw_act: unsupported C=%d", a.C);  // (256 % opr != 0: the last 256 - rpp * opr threads idle)
  if (a.resize == BW_FROM_HALF && (a.L & 1)) VQVS_FAIL(-1, "bw_act: avg-pool backward needs an even length");
  dim3 grid((a.L + STAT_TILE - 1) / STAT_TILE, B);
  VQVS_BY_PRECISION(precision, hipLaunchKernelGGL(bw_act_kernel<T>, grid, dim3(256), 0, st, a));
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_gn_bw(const GnBwArgs& a, int B, hipStream_t st) {
  if (a.C > 2048 || a.groups > 32 || a.C % a.groups) VQVS_FAIL(-1, "gn_bw: unsupported C=%d groups=%d", a.C, a.groups);
  hipLaunchKernelGGL(gn_bw_kernel, dim3(B), dim3(256), 0, st, a);
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_bw_affine(const BwAffineArgs& a, int B, int precision, hipStream_t st) {
  if (a.C % 8) VQVS_FAIL(-1, "bw_affine: unsupported C=%d", a.C);
  const long long items = (long long)a.L * (a.C / 8);
  dim3 grid((unsigned)((items + 255) / 256), B);
  VQVS_BY_PRECISION(precision, hipLaunchKernelGGL(bw_affine_kernel<T>, grid, dim3(256), 0, st, a));
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_in_conv_bw(const InConvBwArgs& a, int B, int precision, hipStream_t st) {
  const int opr = a.C / 8;
  if (a.C % 8 || opr > 64) VQVS_FAIL(-1, "in_conv_bw: unsupported C=%d", a.C);
  dim3 grid((a.T + STAT_TILE - 1) / STAT_TILE, B);
  if (opr & (opr - 1)) {
    VQVS_BY_PRECISION(precision, hipLaunchKernelGGL(in_conv_bw_rows_kernel<T>, grid, dim3(256), 0, st, a));
  } else {
    VQVS_BY_PRECISION(precision, hipLaunchKernelGGL(in_conv_bw_kernel<T>, grid, dim3(256), 0, st, a));
  }
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_bw_add(const void* x, const void* y, void* out, long long n, int precision, hipStream_t st) {
  if (n % 8) VQVS_FAIL(-1, "bw_add: element count must be a multiple of 8");
  const long long n8 = n / 8;
  dim3 grid((unsigned)((n8 + 255) / 256));
  VQVS_BY_PRECISION(precision, hipLaunchKernelGGL(bw_add_kernel<T>, grid, dim3(256), 0, st, (const T*)x, (const T*)y, (T*)out, n8));
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_enc_head(const EncHeadArgs& a, int B, int precision, hipStream_t st) {
  const size_t lds = ((size_t)EH_POS * (a.Cb + 1) + (size_t)a.D * EH_POS + 4 * EH_POS) * 4;
  if (lds > 150 * 1024) VQVS_FAIL(-1, "encoder-predictor head: num_latents=%d needs %zu bytes of LDS (max 150 KB)", a.D, lds);
  if (a.T1 * a.rate != a.T) VQVS_FAIL(-1, "encoder-predictor head: T=%d is not %d x %d", a.T, a.T1, a.rate);
  dim3 grid((a.T1 + EH_POS - 1) / EH_POS, B);
  static bool attr_done[3] = {false, false, false};
  VQVS_BY_PRECISION(precision, {
    if (!attr_done[precision]) {
      VQVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_head_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      attr_done[precision] = true;
    }
    hipLaunchKernelGGL(enc_head_kernel<T>, grid, dim3(256), lds, st, a);
  });
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_cls_head(const HeadArgs& a, int B, int precision, hipStream_t st) {
  if (a.heads < 1 || a.heads > HEAD_MAXH || a.C % a.heads) VQVS_FAIL(-1, "classifier head: unsupported heads=%d C=%d", a.heads, a.C);
  if (a.groups > HEAD_NT) VQVS_FAIL(-1, "classifier head: too many groups");
  const size_t lds = head_lds_floats(a) * 4;
  if (lds > 150 * 1024) VQVS_FAIL(-1, "classifier head: %zu bytes of LDS needed (sequence too long: L=%d)", lds, a.L);
  static bool attr_done[3] = {false, false, false};
  VQVS_BY_PRECISION(precision, {
    if (!attr_done[precision]) {
      VQVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&cls_head_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      attr_done[precision] = true;
    }
    hipLaunchKernelGGL(cls_head_kernel<T>, dim3(B), dim3(HEAD_NT), lds, st, a);
  });
  VQVS_HIP(hipGetLastError());
  return 0;
}

}  // namespace vqvs
