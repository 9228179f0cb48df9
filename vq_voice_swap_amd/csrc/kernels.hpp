// Kernel argument blocks + host launchers (definitions in *.hip).
//
// Activation layout inside the library is channels-last: [clip][time][channel]
// ("NTC"), element type T = float (VQVS_PREC_F32), fp16 (VQVS_PREC_F16) or bf16 (VQVS_PREC_BF16).  A time
// row of C channels is contiguous, which is what an MFMA operand fragment wants
// (8 consecutive k = 8 consecutive channels = one 16-byte LDS read) and what makes
// every HBM access of a wave a run of whole rows.
#pragma once

#include "common.hpp"

namespace vqvs {

// ----------------------------------------------------------------------------------
// Fused 1-D convolution as an implicit GEMM on MFMA.
//
//   out[b][t][co] = bias[co] + sum over segments s, taps k, channels c of
//                   W_s[co][c][k] * in_s[b][t + (k-1)*dil_s][c]            (zero padded)
//                   (+ identity skip: resize(skip)[b][t][co])
//
// A *segment* is one K-range of the GEMM: a source tensor read through an optional
// fused prologue.  This covers, in one kernel, everything a reference ResBlock does
// around its two convolutions (reference vq_voice_swap/models/unet.py:280-316):
//   xform   : y = gelu(x*scale + shift) with per-(clip,channel) scale/shift, i.e.
//             GroupNorm (unet.py:345-349) [+ FiLM h*(a+1)+b, unet.py:311-314] + GELU
//   resize  : avg_pool1d(.,2) or nearest x2 applied between the activation and the conv
//             (unet.py:280-285, 324-334), evaluated while staging rows into LDS
//   concat  : torch.cat([h, skip], 1) (unet.py:156) = two 3-tap segments
//   1x1 skip: skip conv (unet.py:265-267) = extra 1-tap raw segments of the same GEMM
// Epilogue: bias, identity skip (with the same resize), per-(clip,tile,channel) sum and
// sum-of-squares of the result (feeds the next GroupNorm), store.
// ----------------------------------------------------------------------------------
enum { RESIZE_NONE = 0, RESIZE_AVG2 = 1, RESIZE_UP2 = 2 };

struct SegDesc {
  const void* src;   // [B][Lsrc][Csrc] of T
  const float2* ss;  // [B][ss_stride] (scale, shift) or nullptr = raw segment (no affine, no GELU)
  long long w_off;   // element offset of this segment in the packed weights
  int Csrc;          // channels per row of the source tensor
  int c0;            // first source channel used
  int C;             // channels used (multiple of 32)
  int Lsrc;          // rows per clip in the source
  int ntaps;         // 3 (dilated) or 1
  int dil;
  int resize;        // RESIZE_*
  int ss_stride, ss_c0;
  // Two-tensor affine prologue (guidance backward only, conv_mfma_kernel's AFF2 form): when `src2` is set the staged row is
  //   coef.x * src + coef.y * src2 + coef.z      per (clip, channel),
  // i.e. GroupNorm's backward  d h = P * du + Q * h + R  (backward_kernels.hip gn_bw / bw_affine) evaluated while the NEXT transposed
  // convolution stages its input, instead of a streaming pass that writes d h and a convolution that reads it back.  `src2` has the
  // shape and channel window of `src`; no (scale, shift), no resize.
  const void* src2;
  const float4* coef;  // [B][coef_stride] (P, Q, R, 0), this segment's channels start at coef_c0
  int coef_stride, coef_c0;
};

// Fused GELU backward in the epilogue of a transposed convolution (the guidance backward schedule, backward_kernels.hip
// bw_act): out = acc * gelu'(u), u = xf * scale + shift, and the tile statistics become (sum out, sum out*u) -- the partial
// sums GroupNorm's backward needs.  Output channels [c_begin, c_begin + C) read forward tensor `xf` ([B][Lout][C] of T); two
// sources cover the concatenated input of an up block.  (scale, shift) rows are indexed by the OUTPUT channel.
struct BwActFuse {
  const void* xf;
  int C, c_begin;
};

struct GnArgs;
struct ConvArgs {
  SegDesc seg[3];
  int nseg;
  const bf16_t* w_hi;  // packed [segment][chunk of 32 ci][tap][Cout][32] (hi part of the bf16 split)
  const bf16_t* w_lo;  // lo part (VQVS_PREC_F32 only)
  long long w_bytes;   // bytes of one packed plane (bounds of the weight buffer descriptor)
  const float* bias;   // [Cout]
  int Cout, Lout;
  const void* skip;  // identity-skip source [B][skip_L][skip_C] of T, or nullptr
  int skip_C, skip_L, skip_resize;
  void* out;       // [B][Lout][Cout] of T (or float if out_f32)
  int out_f32;
  float* stats;   // [B][ntiles][Cout][2] or nullptr
  int ntiles;     // ceil(Lout / tile_rows)
  int tile_rows;  // output rows per workgroup = conv_tile_rows(max dilation of the 3-tap segments)
  int out_rows;   // rows per clip of the output allocation (0 = Lout); > Lout leaves padding rows untouched
  int epi_gelu;   // 1: out = skip + gelu(acc) -- the ResConv / Conv+GELU blocks of ConvMFCCEncoder (conv_encoder.py:60-84, 113-120)
  BwActFuse bw[2];  // fused GELU backward (nbw = 0: off)
  int nbw;
  const float2* bw_ss;  // [B][bw_ss_stride] forward (scale, shift), column = output channel
  int bw_ss_stride;
  // GroupNorm fused into the consumer (conv_ws.hip refresh_ss): when set, the prologue segments' (scale, shift) rows are NOT read
  // from seg[].ss but built by the convolution's producers from the tile partials described here (every prologue segment is one
  // source of *gn, in order, whole).  Only launch_conv_ws honours it, and only where ws_fuses_gn() says so.
  const GnArgs* gn;
  int ws_f32;  // VQVS_PREC_F32 only: 1 = the launch may run on conv_ws_kernel's fp32 form (never set for the encoders, whose output is
               // vector-quantised: their convolutions keep one summation order, conv_mfma_kernel's, so codes stay bit-stable)
  int rev;  // 1: walk the tiles from the last clip to the first (consecutive launches alternate: a launch starts on what its
            // predecessor wrote last, which is what the Infinity Cache still holds); results do not depend on it
};

int launch_conv(const ConvArgs& a, int B, int precision, hipStream_t st);
// wave-specialised persistent kernel (conv_ws.hip): 1 = launched, 0 = shape not covered (launch_conv falls back to conv_mfma_kernel), < 0 = error
int launch_conv_ws(const ConvArgs& a, int B, int precision, hipStream_t st);
// true when launch_conv_ws would take `a` AND build a.gn's (scale, shift) rows itself (then no gn_prepare launch is needed)
bool ws_fuses_gn(const ConvArgs& a, int B, int precision);
// false when the run-time switches (VQVS_WS=0, VQVS_WS_F32=0) keep this precision mode off conv_ws_kernel altogether
bool ws_available(int precision);
int conv_tile_rows(int dmax, int Cout, int precision, bool ws_ok = false);

// ----------------------------------------------------------------------------------
// small kernels
// ----------------------------------------------------------------------------------
struct InConvArgs {  // Conv1d(Cin -> C, k=3, pad=1) (+ nearest-upsampled cond projection), unet.py:49, 137-139
  const float* x;    // [B][Cin][T] (the boundary's NCT layout; Cin = 1 for every caller of the reference)
  const float* w;    // [C][Cin][3]
  int Cin;           // 0 is read as 1
  const float* bias; // [C]
  const void* condp; // [B][cond_len][C] of T or nullptr: added as F.interpolate(cond, T) (nearest), unet.py:139
  int cond_len;      // rows of condp per clip (T/256 behind a UNet encoder, T/320 behind the MFCC encoder)
  void* out;         // [B][T][C] of T
  float* stats;      // [B][ntiles][C][2]
  int C, T, ntiles;
};
int launch_in_conv(const InConvArgs& a, int B, int precision, hipStream_t st);

struct OutConvArgs {  // GroupNorm+GELU then Conv1d(C -> 1, k=3, pad=1), unet.py:113-116, 162
  const void* in;     // [B][L][C] of T
  const float2* ss;   // [B][C]
  const float* w;     // [3][C] (tap-major)
  float bias;
  float* out;         // [B][L] f32
  int C, L;
};
int launch_out_conv(const OutConvArgs& a, int B, int precision, hipStream_t st);

struct GnSrc {
  const float* partials;  // [B][ntiles][C][2]
  int ntiles, C;
};
struct GnArgs {  // per-(clip,channel) scale/shift of GroupNorm [+FiLM]; nn.GroupNorm eps=1e-5, unet.py:345-349
  GnSrc src[2];
  int nsrc, Ctot, groups;
  double inv_count;    // 1 / (channels_per_group * L)
  const float* gamma;  // [Ctot]
  const float* beta;   // [Ctot]
  const float* film;   // [B][film_stride] rows (a | b) at film_off, or nullptr   (unet.py:311-314)
  int film_stride, film_off;
  float2* ss;          // out [B][Ctot]
  float2* mr;          // optional out [B][Ctot]: (mean, rstd) of the channel's group (kept for the backward pass)
  unsigned* status;    // device status word of the handle: bit 0 is set when a partial is not finite, bit 1 (fp16 mode, `guard`)
  int guard;           // when a 256-row tile's sum of squares reaches 9e8, i.e. an activation may have passed 3e4 of fp16's 65504
};
int launch_gn_prepare(const GnArgs& a, int B, hipStream_t st);

struct TimeEmbedArgs {  // wavegrad.py:359-373 + unet.py:41-45, 133-135
  const float* ts;      // [B]
  const float* freqs;   // [E/2]
  const float *w1, *b1; // [E in][E out] (TRANSPOSED), [E]   time_embed.proj
  const float *w2, *b2; // time_embed_extra.1
  const float* class_embed;  // [num_labels][E] or nullptr
  const int64_t* labels;     // [B] or nullptr
  int num_labels;
  float* emb;   // [B][E]
  float* gemb;  // [B][E] = gelu(emb), input of every block's cond_layers (unet.py:273-278)
  int E;
};
int launch_time_embed(const TimeEmbedArgs& a, int B, hipStream_t st);
int launch_gelu_rows(const float* in, float* out, int n, hipStream_t st);

struct FilmArgs {  // all blocks' cond_layers Linear at once: film[b][r] = bias[r] + W[r] . gemb[b]
  const float* gemb;  // [B][E]
  const float* w;     // [E][R] (TRANSPOSED)
  const float* bias;  // [R]
  float* film;        // [B][R]
  int E, R;
};
int launch_film(const FilmArgs& a, int B, hipStream_t st);

struct XformArgs {  // g = gelu(x*scale + shift) [+ avg_pool1d(.,2)], written once for convolutions whose
  const void* in;    // output-channel tiling would otherwise repeat the prologue >= 4 times (Cout >= 256)
  const float2* ss;  // [B][ss_stride], this tensor's channels start at ss_c0
  void* out;         // [B][Lout][C] of T
  int C, Lin, Lout, avg, ss_stride, ss_c0;
};
int launch_xform(const XformArgs& a, int B, int precision, hipStream_t st);

// ----------------------------------------------------------------------------------
// Input-gradient kernels of the noised-audio classifier (guidance, reference sample_diffusion.py:34-42):
// the backward pass of ResBlock (unet.py:307-316) with respect to its input only (no weight gradients).
// The two transposed convolutions of a block run on the forward MFMA kernel with transposed, tap-flipped
// weights; the kernels below are the element-wise pieces in between.
// ----------------------------------------------------------------------------------
enum { BW_SAME = 0, BW_FROM_HALF = 1, BW_FROM_DOUBLE = 2 };  // resolution of an incoming gradient relative to the output rows
struct BwActArgs {  // du = resize^T(t) * gelu'(u), u = xf*scale + shift;  partial sums (sum du, sum du*u) per tile
  const void* t;     // gradient w.r.t. the activation output; rows of t_C channels, this tensor's slice starts at t_c0.
                     // BW_SAME: [B][L]; BW_FROM_HALF: [B][L/2], enters as 0.5*t[r>>1] (avg-pool backward);
                     // BW_FROM_DOUBLE: [B][2L], enters as t[2r] + t[2r+1] (nearest-upsampling backward)
  const void* xf;    // forward tensor the GroupNorm read: [B][L][C]
  const float2* ss;  // forward (scale, shift): row b at ss + b*ss_stride + ss_c0
  void* du;          // out rows of du_C channels, slice at du_c0 (may alias t when BW_SAME)
  float* partials;   // [B][ntiles][part_C][2] at column part_c0, tiles of STAT_TILE rows
  int C, L, resize;
  int t_C, t_c0, du_C, du_c0, ss_stride, ss_c0, part_C, part_c0;
};
int launch_bw_act(const BwActArgs& a, int B, int precision, hipStream_t st);

struct GnBwArgs {  // GroupNorm backward coefficients: dx = P*du + Q*x + R  per (clip, channel)
  const float* partials;  // [B][ntiles][C][2] from bw_act
  int ntiles, C, groups;
  double inv_count;       // 1 / (channels_per_group * L)
  const float2* ss;       // forward (scale, shift) = (rstd*gamma', beta' - mean*rstd*gamma')
  const float2* mr;       // forward (mean, rstd)
  float4* coef;           // out [B][C] (P, Q, R, 0)
};
int launch_gn_bw(const GnBwArgs& a, int B, hipStream_t st);

struct BwAffineArgs {  // out = P*du + Q*xf + R (+ skip gradient) (+ extra)
  const void* du;     // rows of du_C channels, slice at du_c0
  const void* xf;     // [B][L][C]
  const float4* coef; // row b at coef + b*coef_stride + coef_c0
  const void* skip;   // gradient arriving over the identity skip ([B][.][C]), or nullptr
  int skip_mode;      // BW_SAME / BW_FROM_HALF (0.5*skip[r>>1]) / BW_FROM_DOUBLE (skip[2r] + skip[2r+1])
  const void* extra;  // second addend (gradient through the 1x1 skip convolution): rows of extra_C channels, slice at extra_c0
  void* out;          // [B][L][C] (may alias du when du_C == C)
  int C, L;
  int du_C, du_c0, coef_stride, coef_c0, extra_C, extra_c0;
};
int launch_bw_affine(const BwAffineArgs& a, int B, int precision, hipStream_t st);
int launch_bw_add(const void* a, const void* b, void* out, long long n, int precision, hipStream_t st);  // out = a + b (n multiple of 8)

struct InConvBwArgs {  // dx[b][t] = sum_k sum_c w[c][k] * dh[b][t-k+1][c]   (backward of unet.py:137 in_conv, Cin = 1)
  const void* dh;   // [B][T][C]
  const float* w;   // [C][3]
  float* out;       // [B][T] f32
  int C, T;
  float out_scale;  // the result is multiplied by this (undoes the gradient scaling of the fp16 mode)
};
int launch_in_conv_bw(const InConvBwArgs& a, int B, int precision, hipStream_t st);

// Classifier head (classifier.py:98-104, 153-191, 26-28): GroupNorm+GELU, attention pool whose only consumed output
// is the prepended zero token, c_proj, GELU, Linear -- and, when labels are given, the gradient of
// gscale * log_softmax(logits)[label] with respect to the head's input.  One workgroup per clip.
struct HeadArgs {
  const void* h;       // [B][L][C] output of the last block
  const float2* ss;    // [B][C] GroupNorm (scale, shift) of stem.out.0.0
  const float2* mr;    // [B][C] (mean, rstd)
  int C, L, heads, F, NL, groups;
  double inv_count;
  const float* r;      // [heads][C]  ch^-1/2 * Wk_h^T bq_h   (query token is constant: its input is zero)
  const float* c0;     // [heads]     ch^-1/2 * bq_h . bk_h
  const float* wv;     // [C][C] value rows of qkv_proj   (backward walks the row-major matrices by column,
  const float* wvT;    // [C][C] transposed                 the forward the transposed copies: both coalesce)
  const float* bv;     // [C]
  const float* wc;     // [F][C] c_proj
  const float* wcT;    // [C][F]
  const float* bc;     // [F]
  const float* wl;     // [NL][F] out.1
  const float* wlT;    // [F][NL]
  const float* bl;     // [NL]
  float* logits;       // [B][NL] out
  const int64_t* labels;  // [B] or nullptr = forward only
  float gscale;
  void* dh;            // [B][L][C] out (gradient), when labels
};
int launch_cls_head(const HeadArgs& a, int B, int precision, hipStream_t st);

// EncoderPredictor head (encoder_predictor.py:53-58, 60-64; vq_vae.py:125-130): nearest down-sampling of the UNet
// output by `rate`, 1x1 convolution to `D` logits per latent position, and -- with targets -- the gradient of
// -gscale * sum of cross-entropies with respect to the UNet output (non-zero only at the sampled rows).
struct EncHeadArgs {
  const float* o;      // [B][T][Cb] UNet output (float32, channels-last)
  const float* w;      // [D][Cb]
  const float* bias;   // [D]
  float* logits;       // [B][D][T1] (the reference's NCT layout) or nullptr
  const int64_t* targets;  // [B][T1] or nullptr = forward only
  float gscale;
  void* dO;            // [B][T][Cb] of T: rows i*rate receive the gradient (the tensor is zeroed by the caller)
  int Cb, D, T, T1, rate;
};
int launch_enc_head(const EncHeadArgs& a, int B, int precision, hipStream_t st);

// ----------------------------------------------------------------------------------
// MFCC front end of ConvMFCCEncoder (conv_encoder.py:42-58, 96-104; torchaudio.transforms.MFCC restated, see mfcc_kernels.hip)
// ----------------------------------------------------------------------------------
struct MfccArgs {
  const float* x;         // [B][T] waveform (mu-law companded when ulaw)
  const double* twiddle;  // [n_fft][2]: cos, -sin of 2 pi n / n_fft
  const float* window;    // [n_fft]            buffer mfcc.MelSpectrogram.spectrogram.window
  const float* fb;        // [n_freqs][n_mels]  buffer mfcc.MelSpectrogram.mel_scale.fb
  float* logmel;          // out [B][frames][n_mels]: log(mel + 1e-6) or 10 log10(max(mel, 1e-10))
  float* wgmax;           // out per-workgroup maximum [B][groups] (dB variant) or nullptr
  int T, n_fft, hop, n_freqs, n_mels, frames, ulaw, log_mels;
  double power_scale;     // 1 / sum(window^2) when the spectrogram is normalized, else 1
};
int launch_mfcc_logmel(const MfccArgs& a, int B, hipStream_t st);
int mfcc_groups(int frames);
int launch_mfcc_batch_max(const float* wgmax, int n, float* out, hipStream_t st);
struct MfccFeatArgs {
  const float* logmel;     // [B][frames][n_mels]
  const float* dct;        // [n_mels][13]       buffer mfcc.dct_mat
  const float* batch_max;  // [1] maximum of logmel over the batch (dB variant: values are floored at max - 80) or nullptr
  float* feat;             // out [B][rows_alloc][64]
  int frames, rows_alloc, n_mels;
};
int launch_mfcc_features(const MfccFeatArgs& a, int B, hipStream_t st);

// layout changes at the library boundary (reference tensors are NCT float32)
int launch_nct_to_ntc(const float* in, void* out, float* stats, int B, int C, int L, int ntiles, int precision, hipStream_t st);
int launch_ntc_to_nct(const void* in, float* out, int B, int C, int L, int in_precision, hipStream_t st);

}  // namespace vqvs
