/*
 * vqvs.h -- C ABI of the MI355X (gfx950) native sampler library `libvqvs_hip.so`.
 *
 * The reference (unixpickle/vq-voice-swap) is pure Python: it has no C / FFI /
 * operator boundary of its own.  Its boundary for the DDPM sampling hot path is
 * the Python object surface listed in SURVEY.md section 8(b).  Every entry point
 * below states the reference interface (file:line under the reference repo) it
 * stands behind; the Python classes in `vq_voice_swap_amd/` bind them with ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types.  All pointers named `d_*` are
 *     DEVICE pointers on the model's device; `h_*` are HOST pointers.
 *   - tensors at the boundary use the reference's layout: float32, NCT
 *     ([batch][channels][time], time contiguous); indices are int64.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream).  The caller owns all boundary buffers; the
 *     library owns only the packed weights and a scratch arena per model handle.
 *   - return value 0 = success, negative = error (vqvs_last_error() describes
 *     it, thread-local).  Nothing here falls back to a CPU path.
 *   - a handle is not thread-safe (one scratch arena); different handles are
 *     independent.  The handle-less entry points (vqvs_ddpm_step with CONSTRAIN,
 *     vqvs_vq_argmin) keep one small scratch buffer per (device, stream): calls on
 *     different streams never share it and may run concurrently; calls on one
 *     stream are ordered by the stream.  One process per GPU.
 */
#ifndef VQVS_H
#define VQVS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQVS_OK 0
#define VQVS_ERR_ARG (-1)     /* bad shape / null pointer / unsupported configuration */
#define VQVS_ERR_HIP (-2)     /* a HIP runtime call failed */
#define VQVS_ERR_STATE (-3)   /* handle misuse */

/* network kinds */
#define VQVS_KIND_PREDICTOR 0 /* UNetPredictor  (reference vq_voice_swap/models/unet.py:16-184) */
#define VQVS_KIND_ENCODER 1   /* UNetEncoder    (reference unet.py:187-245) */
#define VQVS_KIND_RESBLOCK 2  /* one ResBlock   (reference unet.py:248-316); unit-test granularity */
#define VQVS_KIND_CLASSIFIER 3 /* Classifier   (reference vq_voice_swap/models/classifier.py:18-191), base_channels + num_labels;
                                  reserved[1] = output_mult (0 = 16); topology_set = 1: channel_mult / depth_mult (n_dilations = 0) */
#define VQVS_KIND_ENCPRED 4    /* EncoderPredictor (reference models/encoder_predictor.py:14-75): base_channels, out_channels =
                                  bottleneck_dim, reserved[1] = downsample_rate, reserved[2] = num_latents */

#define VQVS_KIND_MFCC_ENCODER 5 /* ConvMFCCEncoder (reference models/conv_encoder.py:14-133): base_channels, out_channels,
                                  reserved[1] = version (1: n_fft 320, 40 mels, log; 2: n_fft 400, 80 mels, dB, normalized),
                                  reserved[2] = input_ulaw.  fp32 precision only.  Its parameter table starts with the three
                                  buffers of torchaudio.transforms.MFCC that a reference checkpoint carries (mfcc.dct_mat, ...) */

/* activation storage / arithmetic */
#define VQVS_PREC_F32 0  /* fp32 activations; convs as 3-term bf16-split MFMA, fp32 accumulate (~2^-17 rel.) */
#define VQVS_PREC_BF16 1 /* bf16 activations; bf16 MFMA, fp32 accumulate */
#define VQVS_PREC_F16 2  /* fp16 (IEEE binary16) activations and weights; f16 MFMA, fp32 accumulate; GroupNorm statistics,
                            FiLM, GELU and the residual add are evaluated in fp32.  Meets the 1e-3 waveform-RMS parity
                            bar at 2-byte storage; activations must stay below 65504 in magnitude */

typedef struct vqvs_model vqvs_model;

typedef struct vqvs_cfg {
  int32_t kind;          /* VQVS_KIND_* */
  int32_t base_channels; /* multiple of 32 in 32..256 (reference configs: 32, 64; tuned: 32, 64, 128); MFCC encoder: a power of two;
                            other widths of a predictor / encoder: the padded physical width, see reserved[4] */
  int32_t in_channels;   /* 1 (the reference's default and every caller's value, unet.py:25) .. 64; predictor / encoder handles only */
  int32_t out_channels;  /* predictor: 1 or a multiple of 32; encoder: multiple of 32 */
  int32_t cond_channels; /* 0 = unconditional (unet.py:46-47) */
  int32_t num_labels;    /* 0 = no class embedding (unet.py:44-45) */
  int32_t precision;     /* VQVS_PREC_* */
  int32_t max_batch;     /* scratch arena is sized for this many clips ... */
  int32_t max_T;         /* ... of this many samples (a multiple of the UNet's downsample rate: 256 for the default topology) */
  int32_t debug_taps;    /* 1 = keep every block output resident for vqvs_debug_read_tap */
  /* VQVS_KIND_RESBLOCK only: */
  int32_t rb_cin, rb_cout, rb_resize /*0 none, 1 avg-pool/2, 2 nearest x2*/, rb_dilation, rb_emb_channels /*0 = no FiLM*/;
  int32_t reserved[5];   /* [0] dropout flag (key names), [1] / [2] per kind (above), [3] predictor: conditioning-length code (below),
                            [4] predictor / encoder: the REAL base_channels of a width that is not a multiple of 32, 0 = base_channels.
                            The reference takes any width (unet.py:17-30).  Such a model is built at the PHYSICAL width base_channels =
                            2^k * q_p, where real = 2^k * q (q odd) and q_p = the next power of two >= q, widened until the product is
                            a multiple of 32 (48 -> 64, 40 -> 64, 24 -> 32, 100 -> 128): channels come in blocks of q_p whose first q
                            are the real ones, and the caller passes every parameter in the physical shapes of vqvs_param_info with
                            entry c of a channel axis at (c / q) * q_p + c % q and zeros elsewhere (concatenated inputs and FiLM's
                            (a | b) rows are whole numbers of blocks: one rule for every axis).  GroupNorm groups and counts follow the
                            real width (unet.py:345-349); pad channels are exactly zero in every tensor.
                            vq_voice_swap_amd/unet.py pad_state does this for the Python classes. */
  /* Topology of VQVS_KIND_PREDICTOR / VQVS_KIND_ENCODER / VQVS_KIND_CLASSIFIER (reference UNetPredictor.__init__ unet.py:17-30,
   * UNetEncoder.__init__ unet.py:188-196, ClassifierStem.__init__ classifier.py:52-58).  topology_set = 0: the reference's defaults -- channel_mult (1,1,2,2,2,4,4,8,8), depth_mult 2,
   * middle_dilations (4,8,16,32) / out_dilations () -- and the fields below are ignored.  topology_set = 1: they describe the
   * network: n_levels = len(channel_mult) in 1..VQVS_MAX_LEVELS, every channel_mult[i] * base_channels a multiple of 32 and at most
   * 1024; depth_mult in 1..8; n_dilations = len(middle_dilations) (predictor) or len(out_dilations) (encoder), 0 allowed, each
   * dilation in 1..32.  T must be a multiple of 2^(n_levels - 1).
   * Limits of the builder, checked by every entry point that takes a vqvs_cfg (VQVS_ERR_ARG): predictor: channel_mult[0] == 1 (the
   * output head normalises base_channels, unet.py:113-116 -- the reference builds such a model and fails in forward); classifier:
   * the final width channel_mult[n_levels - 1] * base_channels at most 64 or a multiple of 64 (attention heads of 64 channels,
   * classifier.py:131-150); every kind: the widest per-clip tensor -- max over levels of (max_T / 2^i) rows x 2 * channel_mult[i] *
   * base_channels -- below 2 GiB at 4 bytes per element (rows of a clip are addressed by 32-bit byte offsets). */
  int32_t topology_set;
  int32_t n_levels;
  int32_t channel_mult[12];
  int32_t depth_mult;
  int32_t n_dilations;
  int32_t dilations[12];
} vqvs_cfg;
#define VQVS_MAX_LEVELS 12

/* ---- parameters -----------------------------------------------------------
 * The library owns the topology.  It enumerates the parameters it needs under
 * the reference's own state-dict key names (checkpoint format = reference
 * vq_voice_swap/models/base.py:74-104; key list in SURVEY.md 8(b)), relative to
 * the module (e.g. "down_blocks.3.pre_cond.2.weight"), so a caller can feed it
 * from any {"kwargs","state_dict"} checkpoint. */
int vqvs_param_count(const vqvs_cfg* cfg);
int vqvs_param_info(const vqvs_cfg* cfg, int index, char* name_out, int name_cap, int64_t shape_out[4], int* ndim_out);

/* Build a model: packs `h_params[i]` (host float32, contiguous, in vqvs_param_info
 * order) into device-resident MFMA operand layout and allocates the scratch arena.
 * Replaces nn.Module construction + .to(device): reference diffusion_model.py:14-40,
 * vq_vae.py:15-32, models/base.py:83-104. */
int vqvs_model_create(const vqvs_cfg* cfg, const float* const* h_params, int n_params, int device, vqvs_model** out);
void vqvs_model_destroy(vqvs_model* m);
/* bytes of device memory held by the handle (weights + arena) */
int64_t vqvs_model_device_bytes(const vqvs_model* m);

/* ---- UNet forward -----------------------------------------------------------
 * eps = UNetPredictor.forward(x, ts, cond=, labels=)   reference unet.py:118-163
 *   d_x     [B,in_channels,T] f32      d_ts [B] f32
 *   d_cond  [B,cond_channels,T1] f32 or NULL (must match cfg, unet.py:126-131); T1 = T/256 (cfg.reserved[3] = 0: cond from a
 *           UNet encoder), (T/160 + 1 - 2)/2 + 1 = T/320 (reserved[3] = 1: cond from the MFCC encoder), or ANY length L
 *           (reserved[3] = 1000 + L: the handle then expects exactly L rows per clip, for every T); it is added to the
 *           in_conv output through nearest-neighbour up-sampling to T, as F.interpolate(cond, T) does (unet.py:138-139),
 *           with PyTorch's own source index min(floor(t * (float)L / T), L - 1)
 *   d_labels[B] int64 or NULL (must match cfg)
 *   d_out   [B,out_channels,T] f32 */
int vqvs_unet_forward(vqvs_model* m, const float* d_x, const float* d_ts, const float* d_cond,
                      const int64_t* d_labels, float* d_out, int B, int T, void* stream);

/* z = UNetEncoder.forward(x)   reference unet.py:229-241
 *   d_x [B,in_channels,T] f32 -> d_z [B,out_channels,T/downsample_rate] f32 (NCT) */
int vqvs_encoder_forward(vqvs_model* m, const float* d_x, float* d_z, int B, int T, void* stream);

/* z = ConvMFCCEncoder.forward(x)   reference models/conv_encoder.py:90-110 (VQVS_KIND_MFCC_ENCODER handles):
 * mu-law expansion, MFCC (13 coefficients at 100 frames/s) with first and second order deltas, convolution stack.
 *   d_x [B,1,T] f32 -> d_z [B,out_channels,(T/160 + 1 - 2)/2 + 1] f32 (NCT): 200 positions for 4 s at 16 kHz.
 * The dB variant (version 2) floors the log-mel spectrogram at (maximum over the WHOLE batch) - 80, as
 * torchaudio.functional.amplitude_to_DB does for a 3-D input: its results depend on the batch composition. */
int vqvs_mfcc_encoder_forward(vqvs_model* m, const float* d_x, float* d_z, int B, int T, void* stream);

/* Testing entry for the same handles: the front end's log-mel rows are INJECTED (d_logmel [B][T/160 + 1][n_mels] f32, version-1
 * / log_mels front end only) and everything behind them runs as above: DCT to 13 coefficients, `deltas` twice, concatenation
 * and the convolution stack -- the part of conv_encoder.py:96-110 that is the reference's own code.  With logmel = dct_mat . c the
 * encoder sees the MFCC tensor c (dct_mat has orthonormal columns), which is how tests/test_conv_mfcc.py holds this path to
 * fixture F12 (made from the reference with torchaudio.transforms.MFCC stubbed out). */
int vqvs_mfcc_encoder_forward_logmel(vqvs_model* m, const float* d_logmel, float* d_z, int B, int T, void* stream);

/* y = ResBlock.forward(x, emb)   reference unet.py:307-316 (VQVS_KIND_RESBLOCK handles)
 *   d_x [B,rb_cin,L] f32, d_emb [B,rb_emb_channels] f32 or NULL -> d_y [B,rb_cout,L'] f32 */
int vqvs_resblock_forward(vqvs_model* m, const float* d_x, const float* d_emb, float* d_y, int B, int L, void* stream);

/* Range guard: the device status word of a handle, read and cleared (synchronises the device).  Bit 0: a GroupNorm partial sum
 * was not finite -- an activation overflowed the storage type (fp16: 65504) or the input held NaN; bit 1 (VQVS_PREC_F16 only):
 * a 256-row tile's sum of squares reached 9e8, i.e. an activation may have passed 3e4 -- or the tile's RMS ~1.9e3, which fp16
 * still holds: advisory.  The reference (fp32 throughout, unet.py:337-349) has no such limit, so a caller that sees bit 0 must
 * re-run in VQVS_PREC_F32.  Only tensors that feed a GroupNorm are observed.  The call waits for the whole device
 * (hipDeviceSynchronize), whatever stream the forwards ran on. */
int vqvs_model_status(vqvs_model* m, unsigned* h_status);

/* ---- classifier guidance (BASELINE config 5) -------------------------------------
 * logits = Classifier.forward(x, ts)   reference models/classifier.py:31-36 (stem :111-121, attention pool
 * :153-191).  VQVS_KIND_CLASSIFIER handles; parameters are named relative to the Classifier module
 * ("stem.blocks.3.pre_cond.2.weight", "out.1.weight", ...).  T must be a multiple of 512.
 *   d_x [B,1,T] f32, d_ts [B] f32 -> d_logits [B,num_labels] f32 */
int vqvs_classifier_forward(vqvs_model* m, const float* d_x, const float* d_ts, float* d_logits, int B, int T, void* stream);
/* grad = scale * d/dx log_softmax(Classifier(x, ts))[labels]   -- what the reference's cond_fn obtains with
 * torch.autograd.grad (sample_diffusion.py:34-42).  Runs the forward pass, then an explicit input-gradient
 * schedule (transposed convolutions on the MFMA kernel, GroupNorm / GELU / attention-pool backward).
 *   d_labels [B] int64, d_grad [B,1,T] f32 out, d_logits [B,num_labels] f32 out or NULL */
int vqvs_classifier_guidance(vqvs_model* m, const float* d_x, const float* d_ts, const int64_t* d_labels, float scale,
                             float* d_grad, float* d_logits, int B, int T, void* stream);

/* ---- encoder-predictor guidance ---------------------------------------------------
 * logits = EncoderPredictor.forward(x, ts)   reference models/encoder_predictor.py:43-58: UNetPredictor with a
 * bottleneck output, nearest down-sampling by `downsample_rate`, 1x1 convolution to num_latents logits.
 * VQVS_KIND_ENCPRED handles; parameters are named relative to the module ("unet.in_conv.weight", "out.weight").
 *   d_x [B,1,T] f32, d_ts [B] f32 -> d_logits [B,num_latents,T/rate] f32 */
int vqvs_encpred_forward(vqvs_model* m, const float* d_x, const float* d_ts, float* d_logits, int B, int T, void* stream);
/* grad = -scale * d/dx sum_{b,i} cross_entropy(logits[b,:,i], targets[b,i]) -- the cond_fn of VQVAE.decode(enc_pred=...)
 * (reference vq_vae.py:125-130, encoder_predictor.py:60-64), computed by an explicit backward schedule through the
 * whole UNet (concatenating, up- and down-sampling blocks).
 *   d_targets [B,T/rate] int64, d_grad [B,1,T] f32 out, d_logits [B,num_latents,T/rate] f32 out or NULL */
int vqvs_encpred_guidance(vqvs_model* m, const float* d_x, const float* d_ts, const int64_t* d_targets, float scale,
                          float* d_grad, float* d_logits, int B, int T, void* stream);

/* ---- DDPM reverse step --------------------------------------------------------
 * x_prev = Diffusion.ddpm_previous(x_t, ts, step, eps, noise, sigma_large, constrain)
 * reference diffusion/diffusion.py:48-90 (without cond_fn; with cond_fn the caller
 * uses the two half-steps below around its own cond_fn, diffusion.py:80-83).
 *   d_alpha_t, d_alpha_prev [B] f32: schedule(ts), schedule(ts-step) (schedule.py:30-41)
 *   d_noise [B,1,T] f32, or NULL to draw N(0,1) in-kernel from Philox4x32-10 keyed by
 *     (seed, clip_offset + row, step_index) -- independent of how clips are sharded;
 *     noise_scale 0 reproduces the reference's zero-noise last iteration (diffusion.py:127) */
#define VQVS_DDPM_SIGMA_LARGE 1u
#define VQVS_DDPM_CONSTRAIN 2u
int vqvs_ddpm_step(const float* d_x_t, const float* d_eps, const float* d_noise, const float* d_alpha_t,
                   const float* d_alpha_prev, float* d_x_prev, int B, int T, uint32_t flags, float noise_scale,
                   uint64_t seed, uint64_t clip_offset, uint32_t step_index, void* stream);
/* mean = eps_to_prev(eps)  and  eps' = prev_to_eps(mean + sigma^2 * grad)   (diffusion.py:69-83) */
int vqvs_ddpm_mean(const float* d_x_t, const float* d_eps, const float* d_alpha_t, const float* d_alpha_prev,
                   float* d_mean, int B, int T, void* stream);
int vqvs_ddpm_guided_eps(const float* d_x_t, const float* d_mean, const float* d_grad, const float* d_alpha_t,
                         const float* d_alpha_prev, float* d_eps_out, int B, int T, uint32_t flags, void* stream);
/* x_T ~ N(0,1) from the same counter-based generator (replaces torch.randn, sample_diffusion.py:86) */
int vqvs_randn(float* d_out, int B, int T, uint64_t seed, uint64_t clip_offset, uint32_t stream_id, void* stream);

/* ---- vector quantisation -------------------------------------------------------
 * idx = argmin_k ((-2 z.e_k) + |e_k|^2) + |z|^2, first index on ties
 * reference vq.py:112-143, 199-243.   d_z [B,Cd,T1] f32 NCT, d_dict [K,Cd] f32 -> d_idx [B,T1] int64 */
int vqvs_vq_argmin(const float* d_z, const float* d_dict, int64_t* d_idx, int B, int Cd, int T1, int K, void* stream);
/* out[b,:,t] = dict[idx[b,t],:]   reference vq.py:98-110 */
int vqvs_vq_embed(const int64_t* d_idx, const float* d_dict, float* d_out, int B, int Cd, int T1, int K, void* stream);

/* ---- test / profiling hooks ------------------------------------------------------ */
int vqvs_debug_tap_count(const vqvs_model* m);
int vqvs_debug_tap_info(const vqvs_model* m, int i, char* name_out, int name_cap, int* channels, int* length_shift);
/* rows per clip of tap i at clip length T (UNet taps: T >> length_shift; MFCC encoder taps: frame counts) */
int vqvs_debug_tap_rows(const vqvs_model* m, int i, int T);
/* copies tap i of the LAST forward to host as float32 NCT [B][C][L]; synchronises the device */
int vqvs_debug_read_tap(vqvs_model* m, int i, int B, int T, float* h_out);
/* copies the conditioning vector of the LAST forward -- time_embed_extra(time_embed(ts)) [+ class_embed(labels)], reference
 * unet.py:133-135 with wavegrad.py:359-373 -- to host as float32 [B][4*base_channels]; returns its width; synchronises */
int vqvs_debug_read_embedding(vqvs_model* m, int B, float* h_out);
/* number of kernels one forward enqueues, and the algorithmic activation bytes it moves (SURVEY 8d Model A) */
int vqvs_forward_kernel_count(const vqvs_model* m);
int64_t vqvs_forward_model_bytes(const vqvs_model* m, int B, int T);
int64_t vqvs_forward_flops(const vqvs_model* m, int B, int T);

/* live per-kernel timing: when on, every forward brackets each enqueued kernel with hipEvents on
 * the launch stream; vqvs_profile_read returns the elapsed ms of each op of the last forward and
 * vqvs_op_info its kind ("conv", "gn_prepare", ...) and algorithmic bytes / flops for (B, T). */
int vqvs_set_profiling(vqvs_model* m, int on);
int vqvs_op_info(const vqvs_model* m, int i, char* kind_out, int kind_cap, int64_t* bytes_out, int64_t* flops_out, int B, int T);
int vqvs_op_desc(const vqvs_model* m, int i, char* out, int cap);
int vqvs_profile_read(vqvs_model* m, float* h_ms, int cap);

const char* vqvs_last_error(void);
const char* vqvs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VQVS_H */
