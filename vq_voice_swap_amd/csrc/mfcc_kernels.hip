// MFCC front end of the reference's ConvMFCCEncoder (reference vq_voice_swap/models/conv_encoder.py:42-58, 96-104):
// mu-law expansion -> torchaudio.transforms.MFCC (reflect-padded Hann STFT power spectrum -> mel filter bank -> log or
// dB -> DCT-II) -> first and second order `deltas` (conv_encoder.py:123-129) -> channels-last feature rows for the
// convolution stack.  torchaudio is a third-party dependency absent from the reference tree: the arithmetic below restates
// its published algorithm (see oracle/ref_cpu.py `mfcc_transform` for the citation); the three constant tensors (window,
// filter bank, DCT matrix) are buffers of the reference module and arrive from the checkpoint.
//
// The transform is a few GFLOP per batch: the kernels are written for exactness (a direct DFT accumulated in fp64 against
// an fp64 twiddle table -- closer to the exact transform than the reference's fp32 FFT), not for the last microsecond.
#include "kernels.hpp"

namespace vqvs {

namespace {

constexpr int MF_FR = 8;       // frames per workgroup
constexpr int MF_MAXN = 512;   // max n_fft
constexpr int MF_MAXK = 256;   // max n_fft/2 + 1 (one thread per frequency bin)

__device__ __forceinline__ float ulaw_expand(float x) {  // conv_encoder.py:132-133, mu = 255
  const float m = (exp2f(8.0f * fabsf(x)) - 1.0f) * (1.0f / 255.0f);
  return x > 0.f ? m : (x < 0.f ? -m : 0.f);
}

__global__ __launch_bounds__(256) void mfcc_logmel_kernel(const MfccArgs a) {
  __shared__ double tw[MF_MAXN * 2];
  __shared__ float xs[MF_FR][MF_MAXN];
  __shared__ float pw[MF_FR][MF_MAXK];
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * MF_FR;
  const int N = a.n_fft, half = N >> 1;
  for (int i = tid; i < 2 * N; i += 256) tw[i] = a.twiddle[i];
  // windowed frames: torch.stft(center=True, pad_mode="reflect") pads n_fft/2 samples on both sides
  const float* xb = a.x + (size_t)b * a.T;
  for (int i = tid; i < MF_FR * N; i += 256) {
    const int f = i / N, n = i - f * N;
    float v = 0.f;
    if (f0 + f < a.frames) {
      int s = (f0 + f) * a.hop + n - half;
      if (s < 0) s = -s;
      if (s >= a.T) s = 2 * (a.T - 1) - s;
      v = xb[s];
      if (a.ulaw) v = ulaw_expand(v);
      v *= a.window[n];
    }
    xs[f][n] = v;
  }
  __syncthreads();
  // power spectrum: thread k owns frequency bin k of all MF_FR frames
  if (tid < a.n_freqs) {
    double re[MF_FR], im[MF_FR];
#pragma unroll
    for (int f = 0; f < MF_FR; ++f) re[f] = im[f] = 0.0;
    int idx = 0;
    for (int n = 0; n < N; ++n) {
      const double c = tw[2 * idx], s = tw[2 * idx + 1];
#pragma unroll
      for (int f = 0; f < MF_FR; ++f) {
        const double xv = (double)xs[f][n];
        re[f] = fma(xv, c, re[f]);
        im[f] = fma(xv, s, im[f]);
      }
      idx += tid;
      if (idx >= N) idx -= N;
    }
#pragma unroll
    for (int f = 0; f < MF_FR; ++f) pw[f][tid] = (float)((re[f] * re[f] + im[f] * im[f]) * a.power_scale);
  }
  __syncthreads();
  // mel filter bank + log / dB
  float lmax = -INFINITY;
  for (int i = tid; i < MF_FR * a.n_mels; i += 256) {
    const int f = i / a.n_mels, m = i - f * a.n_mels;
    if (f0 + f >= a.frames) continue;
    double acc = 0.0;
    for (int k = 0; k < a.n_freqs; ++k) acc = fma((double)pw[f][k], (double)a.fb[k * a.n_mels + m], acc);
    const float mel = (float)acc;
    const float v = a.log_mels ? logf(mel + 1e-6f) : 10.0f * log10f(fmaxf(mel, 1e-10f));
    a.logmel[((size_t)b * a.frames + f0 + f) * a.n_mels + m] = v;
    lmax = fmaxf(lmax, v);
  }
  if (a.wgmax) {
    red[tid] = lmax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
      __syncthreads();
    }
    if (tid == 0) a.wgmax[(size_t)b * gridDim.x + blockIdx.x] = red[0];
  }
}

// AmplitudeToDB(top_db = 80) of a 3-D input takes ONE maximum over the whole batch (oracle/ref_cpu.py mfcc_transform)
__global__ __launch_bounds__(256) void mfcc_batch_max_kernel(const float* wgmax, int n, float* out) {
  __shared__ float red[256];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, wgmax[i]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// DCT + deltas + delta-deltas -> feat[b][row][64]: channels 0..12 MFCC, 13..25 deltas, 26..38 delta-deltas, 39..63 zero;
// rows frames .. rows_alloc-1 are zero (the padding row read by the stride-2 convolution).
constexpr int FT_OUT = 60, FT_WIN = 64, FT_C = 13;
__global__ __launch_bounds__(256) void mfcc_features_kernel(const MfccFeatArgs a) {
  __shared__ float m[FT_WIN][FT_C + 1], d[FT_WIN][FT_C + 1];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int g0 = blockIdx.x * FT_OUT;  // first output frame of this workgroup; window row i <-> frame g0 - 2 + i
  const float floor_db = a.batch_max ? a.batch_max[0] - 80.0f : -INFINITY;
  for (int i = tid; i < FT_WIN * FT_C; i += 256) {
    const int r = i / FT_C, c = i - r * FT_C;
    const int g = g0 - 2 + r;
    float v = 0.f;
    if (g >= 0 && g < a.frames) {
      const float* lm = a.logmel + ((size_t)b * a.frames + g) * a.n_mels;
      double acc = 0.0;
      for (int j = 0; j < a.n_mels; ++j) acc = fma((double)fmaxf(lm[j], floor_db), (double)a.dct[j * FT_C + c], acc);
      v = (float)acc;
    }
    m[r][c] = v;
  }
  __syncthreads();
  auto clampf = [&](int g) { return g < 0 ? 0 : (g >= a.frames ? a.frames - 1 : g); };
  for (int i = tid; i < FT_WIN * FT_C; i += 256) {
    const int r = i / FT_C, c = i - r * FT_C;
    const int g = g0 - 2 + r;
    float v = 0.f;
    if (r >= 1 && r <= FT_WIN - 2 && g >= 0 && g < a.frames) {
      const float right = m[clampf(g - 1) - (g0 - 2)][c], left = m[clampf(g + 1) - (g0 - 2)][c], cur = m[r][c];
      v = ((right - cur) + (cur - left)) / 2;  // conv_encoder.py:127-129, same order of operations
    }
    d[r][c] = v;
  }
  __syncthreads();
  for (int i = tid; i < FT_OUT * 64; i += 256) {
    const int ro = i >> 6, ch = i & 63;
    const int g = g0 + ro, r = ro + 2;
    if (g >= a.rows_alloc) continue;
    float v = 0.f;
    if (g < a.frames) {
      if (ch < FT_C) v = m[r][ch];
      else if (ch < 2 * FT_C) v = d[r][ch - FT_C];
      else if (ch < 3 * FT_C) {
        const int c = ch - 2 * FT_C;
        const float right = d[clampf(g - 1) - (g0 - 2)][c], left = d[clampf(g + 1) - (g0 - 2)][c], cur = d[r][c];
        v = ((right - cur) + (cur - left)) / 2;
      }
    }
    a.feat[((size_t)b * a.rows_alloc + g) * 64 + ch] = v;
  }
}

}  // namespace

int launch_mfcc_logmel(const MfccArgs& a, int B, hipStream_t st) {
  if (a.n_fft > MF_MAXN || a.n_freqs > MF_MAXK || a.n_freqs != a.n_fft / 2 + 1) VQVS_FAIL(-1, "mfcc: unsupported n_fft=%d", a.n_fft);
  if (a.T <= a.n_fft / 2) VQVS_FAIL(-1, "mfcc: clip of %d samples is shorter than the reflect padding (%d)", a.T, a.n_fft / 2);
  dim3 grid((a.frames + MF_FR - 1) / MF_FR, B);
  hipLaunchKernelGGL(mfcc_logmel_kernel, grid, dim3(256), 0, st, a);
  VQVS_HIP(hipGetLastError());
  return 0;
}

int mfcc_groups(int frames) { return (frames + MF_FR - 1) / MF_FR; }

int launch_mfcc_batch_max(const float* wgmax, int n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(mfcc_batch_max_kernel, dim3(1), dim3(256), 0, st, wgmax, n, out);
  VQVS_HIP(hipGetLastError());
  return 0;
}

int launch_mfcc_features(const MfccFeatArgs& a, int B, hipStream_t st) {
  dim3 grid((a.rows_alloc + FT_OUT - 1) / FT_OUT, B);
  hipLaunchKernelGGL(mfcc_features_kernel, grid, dim3(256), 0, st, a);
  VQVS_HIP(hipGetLastError());
  return 0;
}

}  // namespace vqvs
