// Topology of the reference's UNets (reference vq_voice_swap/models/unet.py:17-116, 188-227),
// restated as a static schedule of fused kernels.  Per ResBlock (unet.py:248-316):
//
//   gn_prepare(x stats, pre_cond.0.0)                 -> scale/shift of GroupNorm 1
//   conv  [GN1+GELU -> resize -> Conv3]               -> h1   (+ statistics of h1)
//   gn_prepare(h1 stats, pre_cond.3, FiLM a|b)        -> scale/shift of GroupNorm 2 (+FiLM)
//   conv  [GN2*FiLM+GELU -> dilated Conv3] + skip     -> out  (+ statistics of out)
//
// i.e. two streaming passes over the activations per block (SURVEY.md 8d "Model A").
#include "net.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>

namespace vqvs {

namespace {

// The UNets' topology (unet.py:17-30, 188-196): the reference's defaults unless the configuration carries its own
// (vqvs_cfg.topology_set; predictor and encoder handles -- the classifier stem and the encoder predictor are built with the defaults).
struct Topo {
  std::vector<int> mult;  // channel_mult
  int depth = 2;          // depth_mult
  std::vector<int> dil;   // middle_dilations (predictor) / out_dilations (encoder)
  int levels() const { return (int)mult.size(); }
};
Topo topo_of(const vqvs_cfg& c) {
  Topo t;
  const bool custom = c.topology_set != 0 && (c.kind == VQVS_KIND_PREDICTOR || c.kind == VQVS_KIND_ENCODER || c.kind == VQVS_KIND_CLASSIFIER);
  if (custom) {
    t.mult.assign(c.channel_mult, c.channel_mult + std::max(0, std::min(c.n_levels, (int)VQVS_MAX_LEVELS)));
    t.depth = c.depth_mult;
    t.dil.assign(c.dilations, c.dilations + std::max(0, std::min(c.n_dilations, 12)));
  } else {
    t.mult = {1, 1, 2, 2, 2, 4, 4, 8, 8};  // unet.py:20
    t.depth = 2;                            // unet.py:22
    if (c.kind != VQVS_KIND_ENCODER && c.kind != VQVS_KIND_CLASSIFIER) t.dil = {4, 8, 16, 32};  // unet.py:21 (UNetEncoder: out_dilations = (), unet.py:192)
  }
  return t;
}

struct BlockSpec {
  std::string prefix;
  int cin, cout, resize, dil;
  bool cat;
};

void predictor_blocks(int base, const Topo& tp, std::vector<BlockSpec>& down, std::vector<BlockSpec>& mid, std::vector<BlockSpec>& up) {
  const int NLEVEL = tp.levels(), DEPTH_MULT = tp.depth;
  std::vector<int> stack{base};
  int cur = base;
  for (int depth = 0; depth < NLEVEL; ++depth) {
    const int mult = tp.mult[depth];
    for (int i = 0; i < DEPTH_MULT; ++i) {
      down.push_back({"down_blocks." + std::to_string(down.size()), cur, mult * base, RESIZE_NONE, 2, false});
      cur = mult * base;
      stack.push_back(cur);
    }
    if (depth != NLEVEL - 1) {
      down.push_back({"down_blocks." + std::to_string(down.size()), cur, cur, RESIZE_AVG2, 2, false});
      stack.push_back(cur);
    }
  }
  for (size_t i = 0; i < tp.dil.size(); ++i) mid.push_back({"middle_blocks." + std::to_string(i), cur, cur, RESIZE_NONE, tp.dil[i], false});
  for (int depth = NLEVEL - 1; depth >= 0; --depth) {
    const int mult = tp.mult[depth];
    for (int i = 0; i < DEPTH_MULT + 1; ++i) {
      const int sk = stack.back();
      stack.pop_back();
      up.push_back({"up_blocks." + std::to_string(up.size()), cur + sk, mult * base, RESIZE_NONE, 2, true});
      cur = mult * base;
    }
    if (depth) up.push_back({"up_blocks." + std::to_string(up.size()), cur, cur, RESIZE_UP2, 2, false});
  }
}

int encoder_blocks(int base, const Topo& tp, std::vector<BlockSpec>& blocks) {  // returns the width of the last block
  const int NLEVEL = tp.levels(), DEPTH_MULT = tp.depth;
  int cur = base;
  for (int depth = 0; depth < NLEVEL; ++depth) {
    const int mult = tp.mult[depth];
    for (int i = 0; i < DEPTH_MULT; ++i) {
      blocks.push_back({"blocks." + std::to_string(blocks.size()), cur, mult * base, RESIZE_NONE, 2, false});
      cur = mult * base;
    }
    if (depth != NLEVEL - 1) blocks.push_back({"blocks." + std::to_string(blocks.size()), cur, cur, RESIZE_AVG2, 2, false});
  }
  for (int d : tp.dil) blocks.push_back({"blocks." + std::to_string(blocks.size()), cur, cur, RESIZE_NONE, d, false});  // out_dilations, unet.py:219-220
  return cur;
}

// ClassifierStem (classifier.py:79-96): like the encoder, but FiLM-conditioned and with a x0.5 block after EVERY level
// (classifier.py:52-58: channel_mult and depth_mult with the UNets' defaults; output_mult = vqvs_cfg.reserved[1], 0 = 16.)  Returns the
// width of the last block.
int classifier_blocks(int base, const Topo& tp, std::vector<BlockSpec>& blocks) {
  const int NLEVEL = tp.levels(), DEPTH_MULT = tp.depth;
  int cur = base;
  for (int depth = 0; depth < NLEVEL; ++depth) {
    const int mult = tp.mult[depth];
    for (int i = 0; i < DEPTH_MULT; ++i) {
      blocks.push_back({"stem.blocks." + std::to_string(blocks.size()), cur, mult * base, RESIZE_NONE, 2, false});
      cur = mult * base;
    }
    blocks.push_back({"stem.blocks." + std::to_string(blocks.size()), cur, cur, RESIZE_AVG2, 2, false});
  }
  return cur;
}
int classifier_output_mult(const vqvs_cfg& c) { return c.reserved[1] > 0 ? c.reserved[1] : 16; }

void block_params(std::vector<ParamDef>& out, const std::string& p, const BlockSpec& s, int emb, bool dropout) {
  const std::string pre = p.empty() ? "" : p + ".";
  out.push_back({pre + "pre_cond.0.0.weight", {s.cin}});
  out.push_back({pre + "pre_cond.0.0.bias", {s.cin}});
  out.push_back({pre + "pre_cond.2.weight", {s.cout, s.cin, 3}});
  out.push_back({pre + "pre_cond.2.bias", {s.cout}});
  out.push_back({pre + "pre_cond.3.weight", {s.cout}});
  out.push_back({pre + "pre_cond.3.bias", {s.cout}});
  if (emb) {
    out.push_back({pre + "cond_layers.1.weight", {2 * s.cout, emb}});
    out.push_back({pre + "cond_layers.1.bias", {2 * s.cout}});
  }
  const std::string c2 = pre + (dropout ? "post_cond.2" : "post_cond.1");  // unet.py:295-305
  out.push_back({c2 + ".weight", {s.cout, s.cout, 3}});
  out.push_back({c2 + ".bias", {s.cout}});
  if (s.cin != s.cout) {
    out.push_back({pre + "skip.1.weight", {s.cout, s.cin, 1}});
    out.push_back({pre + "skip.1.bias", {s.cout}});
  }
}

bool cfg_dropout(const vqvs_cfg& c) { return c.reserved[0] != 0; }

uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// float32 -> IEEE binary16, round to nearest even (VQVS_PREC_F16 packs the convolution weights in the operand type)
uint16_t f2h(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const uint32_t a = u & 0x7fffffffu;
  if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (a > 0x7f800000u ? 0x200u : 0u));
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // >= 65520 rounds to infinity
  if (a < 0x33000001u) return (uint16_t)sign;                // < 2^-25 rounds to zero
  int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7fffffu) | 0x800000u;
  int shift = e < -14 ? (13 + (-14 - e)) : 13;               // subnormal results lose further bits
  uint32_t r = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (r & 1u))) ++r;
  if (e < -14) return (uint16_t)(sign | r);                  // r may carry into the smallest normal: still correct bits
  return (uint16_t)(sign | (((uint32_t)(e + 15) << 10) + (r - 0x400u)));
}
float bf2f(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

struct Blob {
  std::vector<char> data;
  size_t add(const void* p, size_t bytes) {
    size_t off = (data.size() + 255) & ~(size_t)255;
    data.resize(off + bytes);
    if (p) memcpy(data.data() + off, p, bytes);
    return off;
  }
};

// conv weights in MFMA B-operand order: [segment][chunk of 32 ci][tap][Cout][32 ci]
struct PackedConv {
  std::vector<uint16_t> hi, lo;
  bool f16 = false;  // VQVS_PREC_F16: `hi` holds binary16 bits, `lo` is unused
  explicit PackedConv(int precision = VQVS_PREC_F32) : f16(precision == VQVS_PREC_F16) {}
  long long append(const float* W, int Cout, int Cin_total, int ktaps, int cb, int C) {
    const long long off = (long long)hi.size();
    for (int ch = 0; ch < C / 32; ++ch)
      for (int k = 0; k < ktaps; ++k)
        for (int co = 0; co < Cout; ++co)
          for (int j = 0; j < 32; ++j) {
            const float w = W[((size_t)co * Cin_total + cb + ch * 32 + j) * ktaps + k];
            if (f16) {
              hi.push_back(f2h(w));
              continue;
            }
            const uint16_t h = f2bf(w);
            hi.push_back(h);
            lo.push_back(f2bf(w - bf2f(h)));
          }
    return off;
  }
};

// weights of the transposed convolution (gradient w.r.t. the input): Wt[ci][co][k] = W[co][ci][K-1-k]
std::vector<float> transpose_flip(const float* W, int Cout, int Cin, int ktaps) {
  std::vector<float> t((size_t)Cout * Cin * ktaps);
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int k = 0; k < ktaps; ++k) t[((size_t)ci * Cout + co) * ktaps + k] = W[((size_t)co * Cin + ci) * ktaps + (ktaps - 1 - k)];
  return t;
}

// A tensor's length is a function of the base length of the run.  UNets: a power-of-two shift.  ConvMFCCEncoder
// (conv_encoder.py:44-58, 65-71): lengths derived from the MFCC frame count T / hop + 1, encoded as lshift >= LEN_FRAMES.
constexpr int MFCC_HOP = 160;     // input_rate / mfcc_rate = 16000 / 100 (conv_encoder.py:29-30, never changed by make_encoder)
constexpr int LEN_FRAMES = 100;   // T / hop + 1                      (torch.stft, center=True)
constexpr int LEN_FRAMES_PAD = 101;  // frames rounded up to even: allocation of the tensor the stride-2 convolution reads
constexpr int LEN_PAIRS = 102;    // LEN_FRAMES_PAD / 2: rows of that tensor viewed as [pairs][2 * channels]
constexpr int LEN_HALF = 103;     // (frames + 2 - 4) / 2 + 1: output length of Conv1d(k = 4, stride = 2, padding = 1)
constexpr int LEN_ABS = 1000;     // lshift >= LEN_ABS: a fixed number of rows, lshift - LEN_ABS, whatever the base length (a conditioning
                                  // sequence of arbitrary length: the reference up-samples ANY length to T, unet.py:138-139)
double lscale(int lshift) {
  if (lshift >= LEN_ABS) return 0.0;  // (not proportional to the base length: left out of the per-unit-length accounting)
  if (lshift >= LEN_FRAMES) return (lshift == LEN_PAIRS || lshift == LEN_HALF ? 0.5 : 1.0) / MFCC_HOP;
  return lshift >= 0 ? 1.0 / (double)(1 << lshift) : (double)(1 << -lshift);
}
int shiftL(int L, int lshift) {
  if (lshift >= LEN_ABS) return lshift - LEN_ABS;
  if (lshift >= LEN_FRAMES) {
    const int frames = L / MFCC_HOP + 1;
    if (lshift == LEN_FRAMES) return frames;
    if (lshift == LEN_FRAMES_PAD) return (frames + 1) & ~1;
    if (lshift == LEN_PAIRS) return (frames + 1) >> 1;
    return (frames - 2) / 2 + 1;
  }
  return lshift >= 0 ? (L >> lshift) : (L << -lshift);
}
int ntiles_of(int L, int rows = STAT_TILE) { return (L + rows - 1) / rows; }
constexpr int MIN_TILE_ROWS = 124;  // the smallest statistics tile any producer uses (128-row tiles, dilation 2)

class Builder {
 public:
  Builder(vqvs_model* m, const float* const* hp) : m_(m), hp_(hp) {
    for (size_t i = 0; i < m->params.size(); ++i) pidx_[m->params[i].name] = (int)i;
    es_ = m->cfg.precision == VQVS_PREC_F32 ? 4 : 2;
    maxB_ = m->cfg.max_batch;
    maxL_ = m->cfg.max_T;
  }

  const float* P(const std::string& name) const {
    auto it = pidx_.find(name);
    if (it == pidx_.end()) return nullptr;
    return hp_[it->second];
  }
  // [rows][cols] host matrix uploaded as [cols][rows]
  size_t blob_f32_transposed(const float* w, int rows, int cols) {
    std::vector<float> t((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = w[(size_t)r * cols + c];
    return blob.add(t.data(), t.size() * 4);
  }
  size_t blob_f32(const std::string& name) {
    auto it = pidx_.find(name);
    return blob.add(hp_[it->second], m_->params[it->second].numel() * 4);
  }

  // ---- arena planning -------------------------------------------------------------
  TensorH new_tensor(int C, int lshift, bool f32, bool stats) {
    TensorH t;
    t.id = next_id_++;
    t.C = C;
    t.lshift = lshift;
    t.f32 = f32;
    const size_t bytes = ((size_t)maxB_ * shiftL(maxL_, lshift) * C * (f32 ? 4 : es_) + 255) & ~(size_t)255;
    t.off = arena_alloc(bytes);
    sizes_[t.id] = bytes;
    offs_[t.id] = t.off;
    refs_[t.id] = 1;
    if (stats) {
      t.has_stats = true;
      t.stats_off = stats_floats;
      stats_floats += (size_t)maxB_ * ntiles_of(shiftL(maxL_, lshift), MIN_TILE_ROWS) * C * 2;
      tile_rows_[t.id] = STAT_TILE;
    }
    return t;
  }
  int stat_rows(const TensorH& t) const { return tile_rows_.at(t.id); }
  void retain(const TensorH& t) { refs_[t.id]++; }
  void release(const TensorH& t) {
    if (persist) return;  // forward tensors of a model with a backward pass stay resident
    if (--refs_[t.id] == 0 && !m_->cfg.debug_taps) arena_free(offs_[t.id], sizes_[t.id]);
  }
  bool persist = false;
  size_t alloc_stats(int C, int lshift) {
    const size_t off = stats_floats;
    stats_floats += (size_t)maxB_ * ntiles_of(shiftL(maxL_, lshift), MIN_TILE_ROWS) * C * 2;
    return off;
  }
  // what the backward pass needs to know about one forward ResBlock
  struct BlockRec {
    BlockSpec s;
    TensorH x, x2, h1, out;  // x2: the skip-stack tensor of a concatenating block (id < 0 otherwise)
    size_t ss1 = 0, ss2 = 0, mr1 = 0, mr2 = 0;
  };
  size_t alloc_ss(int C) {
    const size_t off = ss_floats;
    ss_floats += (size_t)maxB_ * C * 2;
    return off;
  }
  size_t alloc_misc(size_t floats) {
    const size_t off = misc_floats;
    misc_floats += (floats + 63) & ~(size_t)63;
    return off;
  }
  // the handle's device status word (range guard of gn_prepare; read and cleared by vqvs_model_status)
  size_t status_off() {
    if (status_off_ == (size_t)-1) {
      status_off_ = alloc_misc(64);
      m_->status_misc_off = status_off_;
    }
    return status_off_;
  }
  size_t status_off_ = (size_t)-1;

  // ---- pointer resolution at run time ------------------------------------------------
  // (regions are laid out after planning: [activations | stats | ss | misc])
  char* act(size_t off) const { return m_->d_arena + off; }
  float* statp(size_t foff) const { return reinterpret_cast<float*>(m_->d_arena + m_->stats_off) + foff; }
  float* ssp(size_t foff) const { return reinterpret_cast<float*>(m_->d_arena + m_->ss_off) + foff; }
  float* miscp(size_t foff) const { return reinterpret_cast<float*>(m_->d_arena + m_->misc_off) + foff; }
  char* wp(size_t off) const { return m_->d_weights + off; }

  // ---- ops ---------------------------------------------------------------------------
  // A GroupNorm whose only consumer is the prologue of ONE convolution can be built by that convolution's producers (conv_ws.hip
  // WsGn): the link lets the two schedule entries agree at run time -- `fused` (set by add_conv) says whether the convolution
  // takes it for this batch / length, in which case the gn_prepare entry does nothing.
  struct GnLink {
    std::function<bool(const RunCtx&)> fused;
    std::function<void(const RunCtx&, GnArgs&)> fill;
  };
  std::shared_ptr<GnLink> add_gn(const std::vector<TensorH>& srcs, const std::string& gn_name, bool film, int film_off, int film_stride,
              size_t film_misc_off, size_t ss_off, bool want_mr = false, size_t mr_off = 0) {
    int Ctot = 0;
    for (auto& s : srcs) Ctot += s.C;
    const size_t g_off = blob_f32(gn_name + ".weight");
    const size_t b_off = blob_f32(gn_name + ".bias");
    // (a padded handle: the reference's groups and counts follow the REAL width, see pad_map)
    const int Creal = (int)((long long)Ctot * real_base(m_->cfg) / m_->cfg.base_channels);
    const int groups = gn_groups(Creal);
    const int lshift = srcs[0].lshift;
    Builder* self = this;
    std::vector<TensorH> S = srcs;
    std::vector<int> SR;
    for (auto& t : srcs) SR.push_back(stat_rows(t));
    const size_t st_off = status_off();
    const int guard = m_->cfg.precision == VQVS_PREC_F16 ? 1 : 0;
    m_->meta.push_back({"gn_prepare", gn_name + " C=" + std::to_string(Ctot) + " L>>" + std::to_string(lshift), 0, 0, 0});
    auto link = std::make_shared<GnLink>();
    link->fill = [=](const RunCtx& c, GnArgs& a) {
      const int L = shiftL(c.Lbase, lshift);
      a.nsrc = (int)S.size();
      for (int i = 0; i < a.nsrc; ++i) a.src[i] = GnSrc{self->statp(S[i].stats_off), ntiles_of(L, SR[i]), S[i].C};
      a.Ctot = Ctot;
      a.groups = groups;
      a.inv_count = 1.0 / ((double)(Creal / groups) * (double)L);
      a.gamma = reinterpret_cast<const float*>(self->wp(g_off));
      a.beta = reinterpret_cast<const float*>(self->wp(b_off));
      a.film = film ? self->miscp(film_misc_off) : nullptr;
      a.film_stride = film_stride;
      a.film_off = film_off;
      a.ss = reinterpret_cast<float2*>(self->ssp(ss_off));
      a.mr = want_mr ? reinterpret_cast<float2*>(self->ssp(mr_off)) : nullptr;
      a.status = reinterpret_cast<unsigned*>(self->miscp(st_off));
      a.guard = guard;
    };
    m_->add_op([=](const RunCtx& c) -> int {
      if (link->fused && link->fused(c)) return 0;
      GnArgs a{};
      link->fill(c, a);
      return launch_gn_prepare(a, c.B, c.st);
    });
    return link;
  }

  // g = gelu(GN(x)) [avg-pooled] written once (deep levels, see XformArgs)
  TensorH add_xform(const TensorH& src, size_t ss_off, int ss_stride, int ss_c0, bool avg) {
    TensorH g = new_tensor(src.C, src.lshift + (avg ? 1 : 0), false, false);
    Builder* self = this;
    const int prec = m_->cfg.precision;
    const TensorH S = src;
    m_->meta.push_back({"xform", "C=" + std::to_string(src.C) + " L>>" + std::to_string(g.lshift) + (avg ? " avg" : ""),
                        src.C * (lscale(src.lshift) + lscale(g.lshift)), 0, 0});
    m_->add_op([=](const RunCtx& c) -> int {
      XformArgs a{};
      a.in = self->act(S.off);
      a.ss = reinterpret_cast<const float2*>(self->ssp(ss_off));
      a.out = self->act(g.off);
      a.C = S.C;
      a.Lin = shiftL(c.Lbase, S.lshift);
      a.Lout = shiftL(c.Lbase, g.lshift);
      a.avg = avg ? 1 : 0;
      a.ss_stride = ss_stride;
      a.ss_c0 = ss_c0;
      return launch_xform(a, c.B, prec, c.st);
    });
    return g;
  }

  struct SegSpec {
    TensorH t;
    int c0, C, ntaps, dil, resize;
    bool xform;
    size_t ss_off;
    int ss_stride, ss_c0;
    long long w_off;
    // two-tensor affine prologue (kernels.hpp SegDesc.src2): staged row = P * t + Q * t2 + R with (P, Q, R, 0) rows at coef_off (misc region)
    bool aff2 = false;
    TensorH t2{};
    size_t coef_off = 0;
  };

  // GELU backward fused into a transposed convolution's epilogue (kernels.hpp BwActFuse): forward tensors covering the output
  // channels in order, the forward (scale, shift) table and the partial-sum block the GroupNorm backward reads
  struct BwFuse {
    std::vector<TensorH> xf;
    size_t ss_off;
    int ss_stride;
    size_t part_off;
  };
  static bool bw_fuse_enabled() {
    static int v = -1;
    if (v < 0) {
      const char* e = getenv("VQVS_BW_FUSE");  // 0: keep the separate bw_act kernel (A/B measurements)
      v = e ? atoi(e) : 1;
    }
    return v != 0;
  }

  // out_lshift: logical output length when it differs from the allocation of `out` (padding rows); epi_gelu: out = skip + gelu(acc).
  // Returns the rows per workgroup tile (= rows per statistics / partial-sum tile of the output).
  int add_conv(const std::vector<SegSpec>& segs, const PackedConv& pk, const std::vector<float>& bias, int Cout, const TensorH& out,
               const TensorH* skip, int skip_resize, int out_lshift = -1000, bool epi_gelu = false, const BwFuse* fuse = nullptr,
               std::shared_ptr<GnLink> gn = nullptr) {
    const size_t hi_off = blob.add(pk.hi.data(), pk.hi.size() * 2);
    const size_t lo_off = m_->cfg.precision == VQVS_PREC_F32 ? blob.add(pk.lo.data(), pk.lo.size() * 2) : 0;
    const size_t bias_off = blob.add(bias.data(), bias.size() * 4);
    const long long w_bytes = (long long)pk.hi.size() * 2;
    const bool x3 = m_->cfg.precision == VQVS_PREC_F32;
    const int prec = m_->cfg.precision;
    Builder* self = this;
    std::vector<SegSpec> S = segs;
    const TensorH O = out;
    const bool has_skip = skip != nullptr;
    const TensorH K = skip ? *skip : TensorH{};
    int dmax = 0;
    for (auto& g : segs)
      if (g.ntaps == 3 && g.dil > dmax) dmax = g.dil;
    int kchunks = skip ? 2 : 0;  // (K chunks of 32 channels per tile: a one-chunk tile is only covered at 32 output channels)
    for (auto& g : segs) kchunks += g.C / 32;
    bool ws_ok = !out.f32 && !epi_gelu && fuse == nullptr && out_lshift == -1000 && kchunks >= 2;
    for (auto& g : segs) ws_ok = ws_ok && !g.aff2;  // (conv_mfma_kernel's two-tensor prologue)
    if (m_->cfg.precision == VQVS_PREC_F32) {
      // fp32 storage: a tile geometry other than the default one may only be chosen where conv_ws_kernel is certain to take the
      // launch (conv_mfma_kernel has one geometry): its fp32 form has no avg-pooled sources, wants 32-channel chunks, and
      // addresses a clip's rows through 32-bit descriptors -- checked here against the model's largest clip
      const bool kind_ok = m_->cfg.kind == VQVS_KIND_PREDICTOR || m_->cfg.kind == VQVS_KIND_RESBLOCK;
      bool ok = kind_ok && Cout % 64 == 0 && !(skip && skip_resize == RESIZE_AVG2);
      auto clip_bytes = [&](const TensorH& t) { return (long long)shiftL(m_->cfg.max_T, t.lshift) * t.C * 4; };
      for (auto& g : segs) ok = ok && g.resize != RESIZE_AVG2 && g.C % 32 == 0 && (g.ntaps == 1 || g.ntaps == 3) && clip_bytes(g.t) < (1LL << 29);
      if (skip) ok = ok && skip->C == Cout && clip_bytes(*skip) < (1LL << 29);
      ok = ok && clip_bytes(out) < (1LL << 31);
      ws_ok = ws_ok && ok;
    }
    const int tile_rows = conv_tile_rows(dmax, Cout, m_->cfg.precision, ws_ok);
    if (out.has_stats) tile_rows_[out.id] = tile_rows;
    // cost
    double ktot = 0;
    double conv_elems = 0, conv_f32 = 0;
    for (auto& s : segs) {
      conv_elems += s.C * lscale(s.t.lshift) * (s.aff2 ? 2 : 1);
      ktot += (double)s.C * s.ntaps;
    }
    if (skip) conv_elems += K.C * lscale(K.lshift);
    if (fuse) conv_elems += Cout * lscale(out.lshift);  // the forward tensor(s) read by the fused GELU backward
    const bool has_fuse = fuse != nullptr;
    const BwFuse F = fuse ? *fuse : BwFuse{};
    if (out.f32) conv_f32 += 4.0 * Cout * lscale(out.lshift);
    else conv_elems += Cout * lscale(out.lshift);
    const double conv_flops = 2.0 * Cout * ktot * lscale(out.lshift);
    std::string desc;
    for (auto& s : segs) desc += (desc.empty() ? "" : "+") + std::to_string(s.C) + "x" + std::to_string(s.ntaps) + (s.resize == RESIZE_AVG2 ? "v" : s.resize == RESIZE_UP2 ? "^" : "") + (s.ntaps == 3 && s.dil > 1 ? "d" + std::to_string(s.dil) : "");
    desc += "->" + std::to_string(Cout) + " L>>" + std::to_string(out.lshift) + (skip ? " +id" : "");
    m_->meta.push_back({"conv", desc, conv_elems, conv_f32, conv_flops});
    const int ws_f32 = (m_->cfg.kind == VQVS_KIND_PREDICTOR || m_->cfg.kind == VQVS_KIND_RESBLOCK) ? 1 : 0;
    const int rev = (conv_seq_++) & 1;  // consecutive convolutions walk their tiles in opposite directions (ConvArgs.rev)
    auto build = [=](const RunCtx& c, ConvArgs& a) {
      a.nseg = (int)S.size();
      for (int i = 0; i < a.nseg; ++i) {
        SegDesc& g = a.seg[i];
        g.src = self->act(S[i].t.off);
        g.ss = S[i].xform ? reinterpret_cast<const float2*>(self->ssp(S[i].ss_off)) : nullptr;
        g.w_off = S[i].w_off;
        g.Csrc = S[i].t.C;
        g.c0 = S[i].c0;
        g.C = S[i].C;
        g.Lsrc = shiftL(c.Lbase, S[i].t.lshift);
        g.ntaps = S[i].ntaps;
        g.dil = S[i].dil;
        g.resize = S[i].resize;
        g.ss_stride = S[i].ss_stride;
        g.ss_c0 = S[i].ss_c0;
        if (S[i].aff2) {
          g.src2 = self->act(S[i].t2.off);
          g.coef = reinterpret_cast<const float4*>(self->miscp(S[i].coef_off));
          g.coef_stride = S[i].C;
          g.coef_c0 = 0;
        }
      }
      a.w_hi = reinterpret_cast<const bf16_t*>(self->wp(hi_off));
      a.w_lo = x3 ? reinterpret_cast<const bf16_t*>(self->wp(lo_off)) : nullptr;
      a.w_bytes = w_bytes;
      a.bias = reinterpret_cast<const float*>(self->wp(bias_off));
      a.Cout = Cout;
      a.Lout = shiftL(c.Lbase, out_lshift == -1000 ? O.lshift : out_lshift);
      a.out_rows = shiftL(c.Lbase, O.lshift);
      a.epi_gelu = epi_gelu ? 1 : 0;
      if (has_skip) {
        a.skip = self->act(K.off);
        a.skip_C = K.C;
        a.skip_L = shiftL(c.Lbase, K.lshift);
        a.skip_resize = skip_resize;
      }
      a.out = self->act(O.off);
      a.out_f32 = O.f32 ? 1 : 0;
      a.stats = O.has_stats ? self->statp(O.stats_off) : nullptr;
      a.ntiles = ntiles_of(a.Lout, tile_rows);
      a.tile_rows = tile_rows;
      a.rev = rev;
      a.ws_f32 = ws_f32;
      if (has_fuse) {
        a.nbw = (int)F.xf.size();
        int c0 = 0;
        for (int i = 0; i < a.nbw; ++i) {
          a.bw[i] = BwActFuse{self->act(F.xf[i].off), F.xf[i].C, c0};
          c0 += F.xf[i].C;
        }
        a.bw_ss = reinterpret_cast<const float2*>(self->ssp(F.ss_off));
        a.bw_ss_stride = F.ss_stride;
        a.stats = self->statp(F.part_off);
      }
    };
    if (gn) {
      GnLink* const gl = gn.get();
      gl->fused = [=](const RunCtx& c) -> bool {
        ConvArgs a{};
        build(c, a);
        GnArgs g{};
        gl->fill(c, g);
        a.gn = &g;
        return ws_fuses_gn(a, c.B, prec);
      };
    }
    m_->add_op([=](const RunCtx& c) -> int {
      ConvArgs a{};
      build(c, a);
      GnArgs g{};
      if (gn) {
        gn->fill(c, g);
        a.gn = &g;
        if (!ws_fuses_gn(a, c.B, prec)) a.gn = nullptr;  // (then the gn_prepare entry before this one has run)
      }
      return launch_conv(a, c.B, prec, c.st);
    });
    return tile_rows;
  }

  // One reference ResBlock (unet.py:248-316).  `ins` = 1 tensor, or 2 for torch.cat([h, skip], 1).
  // film_misc_off/film_off/film_stride locate this block's (a|b) rows; emb = false for the encoder.
  TensorH resblock(const std::string& p, const BlockSpec& s, const std::vector<TensorH>& ins, bool emb, int film_off, int film_stride,
                   size_t film_misc_off, BlockRec* rec = nullptr) {
    const std::string pre = p.empty() ? "" : p + ".";
    const int cin = s.cin, cout = s.cout;
    const int in_shift = ins[0].lshift;
    const int out_shift = in_shift + (s.resize == RESIZE_AVG2 ? 1 : (s.resize == RESIZE_UP2 ? -1 : 0));
    // Prologue hoisted into its own kernel (the convolution then reads transformed rows raw):
    //  - fp32 mode (conv_mfma_kernel: every channel-tile workgroup repeats the prologue of the same input rows): from 4 tiles up;
    //  - 2-byte modes (conv_ws_kernel: the producers' prologue runs beside the MFMAs, measured cheaper than the extra pass from
    //    512 channels too): only where the per-clip (scale, shift) table of the launch does not fit the CU's LDS next to the
    //    tiles -- conv 1 over a concatenated input of more than 640 channels (768 and 1024 in the UNets).
    static const int pre_xform_min = getenv("VQVS_PRE_XFORM_MIN") ? atoi(getenv("VQVS_PRE_XFORM_MIN")) : 0;  // (A/B: > 0 = hoist from this Cout up, both convs)
    const bool two_byte = m_->cfg.precision != VQVS_PREC_F32;
    const bool pre_xform1 = pre_xform_min > 0 ? cout >= pre_xform_min : (two_byte ? cin > 640 : cout >= 512);  // (2 x 8 B x cin of table beside the tiles)
    const bool pre_xform2 = pre_xform_min > 0 ? cout >= pre_xform_min : (two_byte ? false : cout >= 512);
    // GroupNorm 1 coefficients over the (virtually concatenated) input
    const size_t ss1 = alloc_ss(cin);
    const size_t mr1 = rec ? alloc_ss(cin) : 0;
    // (a model with a backward pass keeps every table in memory: no fusion there)
    auto gn1 = add_gn(ins, pre + "pre_cond.0.0", false, 0, 0, 0, ss1, rec != nullptr, mr1);
    const bool fuse_gn = rec == nullptr && two_byte;
    // conv 1
    TensorH h1 = new_tensor(cout, out_shift, false, true);
    {
      PackedConv pk(m_->cfg.precision);
      std::vector<SegSpec> segs;
      const float* W = P(pre + "pre_cond.2.weight");
      int cb = 0;
      std::vector<TensorH> tmp;
      for (auto& t : ins) {
        SegSpec g{t, 0, t.C, 3, 1, s.resize, true, ss1, cin, cb, 0};
        if (pre_xform1) {  // prologue hoisted out of the GEMM: the conv reads g raw
          g.t = add_xform(t, ss1, cin, cb, s.resize == RESIZE_AVG2);
          g.xform = false;
          if (s.resize == RESIZE_AVG2) g.resize = RESIZE_NONE;
          tmp.push_back(g.t);
        }
        g.w_off = pk.append(W, cout, cin, 3, cb, t.C);
        segs.push_back(g);
        cb += t.C;
      }
      const float* b = P(pre + "pre_cond.2.bias");
      add_conv(segs, pk, std::vector<float>(b, b + cout), cout, h1, nullptr, 0, -1000, false, nullptr, fuse_gn && !pre_xform1 ? gn1 : nullptr);
      for (auto& t : tmp) release(t);
    }
    // GroupNorm 2 (+FiLM) coefficients
    const size_t ss2 = alloc_ss(cout);
    const size_t mr2 = rec ? alloc_ss(cout) : 0;
    auto gn2 = add_gn({h1}, pre + "pre_cond.3", emb, film_off, film_stride, film_misc_off, ss2, rec != nullptr, mr2);
    const std::shared_ptr<GnLink> gl2 = fuse_gn && !pre_xform2 ? gn2 : nullptr;
    // conv 2 + skip
    TensorH out = new_tensor(cout, out_shift, false, true);
    {
      PackedConv pk(m_->cfg.precision);
      std::vector<SegSpec> segs;
      const std::string c2 = pre + (cfg_dropout(m_->cfg) ? "post_cond.2" : "post_cond.1");
      const float* W = P(c2 + ".weight");
      const float* b = P(c2 + ".bias");
      std::vector<float> bias(b, b + cout);
      SegSpec g{h1, 0, cout, 3, s.dil, RESIZE_NONE, true, ss2, cout, 0, 0};
      TensorH g2{};
      if (pre_xform2) {
        g2 = add_xform(h1, ss2, cout, 0, false);
        g.t = g2;
        g.xform = false;
      }
      g.w_off = pk.append(W, cout, cout, 3, 0, cout);
      segs.push_back(g);
      if (cin != cout) {  // 1x1 skip conv folded into the same GEMM as raw 1-tap segments
        const float* Ws = P(pre + "skip.1.weight");
        const float* bs = P(pre + "skip.1.bias");
        for (int i = 0; i < cout; ++i) bias[i] += bs[i];
        int cb = 0;
        for (auto& t : ins) {
          SegSpec r{t, 0, t.C, 1, 1, RESIZE_NONE, false, 0, 0, 0, 0};
          r.w_off = pk.append(Ws, cout, cin, 1, cb, t.C);
          segs.push_back(r);
          cb += t.C;
        }
        add_conv(segs, pk, bias, cout, out, nullptr, 0, -1000, false, nullptr, gl2);
      } else {  // identity skip (never together with a concatenated input in this topology)
        add_conv(segs, pk, bias, cout, out, &ins[0], s.resize, -1000, false, nullptr, gl2);
      }
      if (pre_xform2) release(g2);
    }
    release(h1);
    if (rec) *rec = BlockRec{s, ins[0], ins.size() > 1 ? ins[1] : TensorH{}, h1, out, ss1, ss2, mr1, mr2};
    return out;
  }

  // ---- backward pieces (classifier guidance) ------------------------------------------------------
  // slice [c0, c0 + xf.C) of the (possibly wider) gradient tensor `t` / output `du`; statistics into columns
  // [c0, c0 + xf.C) of a partials block that is Ctot channels wide
  void add_bw_act(const TensorH& t, int resize, const TensorH& xf, size_t ss_off, const TensorH& du, size_t part_off, int c0 = 0,
                  int Ctot = 0) {
    Builder* self = this;
    const int prec = m_->cfg.precision;
    if (Ctot == 0) Ctot = xf.C;
    m_->meta.push_back({"bw_act", "C=" + std::to_string(xf.C) + " L>>" + std::to_string(xf.lshift) +
                                      (resize == BW_FROM_HALF ? " from L/2" : resize == BW_FROM_DOUBLE ? " from 2L" : ""),
                        xf.C * (2 * lscale(xf.lshift) + lscale(t.lshift)), 0, 0});
    m_->add_op([=](const RunCtx& c) -> int {
      BwActArgs a{};
      a.t = self->act(t.off);
      a.xf = self->act(xf.off);
      a.ss = reinterpret_cast<const float2*>(self->ssp(ss_off));
      a.du = self->act(du.off);
      a.partials = self->statp(part_off);
      a.C = xf.C;
      a.L = shiftL(c.Lbase, xf.lshift);
      a.resize = resize;
      a.t_C = t.C;
      a.t_c0 = t.C == xf.C ? 0 : c0;
      a.du_C = du.C;
      a.du_c0 = du.C == xf.C ? 0 : c0;
      a.ss_stride = Ctot;
      a.ss_c0 = c0;
      a.part_C = Ctot;
      a.part_c0 = c0;
      return launch_bw_act(a, c.B, prec, c.st);
    });
  }
  size_t add_gn_bw(int C, int lshift, size_t part_off, size_t ss_off, size_t mr_off, int tile_rows = STAT_TILE) {
    const size_t coef = alloc_misc((size_t)maxB_ * C * 4);
    Builder* self = this;
    const int groups = gn_groups(C);
    m_->meta.push_back({"gn_bw", "C=" + std::to_string(C) + " L>>" + std::to_string(lshift), 0, 0, 0});
    m_->add_op([=](const RunCtx& c) -> int {
      GnBwArgs a{};
      const int L = shiftL(c.Lbase, lshift);
      a.partials = self->statp(part_off);
      a.ntiles = ntiles_of(L, tile_rows);
      a.C = C;
      a.groups = groups;
      a.inv_count = 1.0 / ((double)(C / groups) * (double)L);
      a.ss = reinterpret_cast<const float2*>(self->ssp(ss_off));
      a.mr = reinterpret_cast<const float2*>(self->ssp(mr_off));
      a.coef = reinterpret_cast<float4*>(self->miscp(coef));
      return launch_gn_bw(a, c.B, c.st);
    });
    return coef;
  }
  // out = P*du + Q*xf + R (+ skip) (+ extra); du / extra / coef may be slices [c0, c0 + xf.C) of Ctot-wide buffers
  void add_bw_affine(const TensorH& du, const TensorH& xf, size_t coef, const TensorH* skip, int skip_mode, const TensorH* extra,
                     const TensorH& out, int c0 = 0, int Ctot = 0) {
    Builder* self = this;
    const int prec = m_->cfg.precision;
    if (Ctot == 0) Ctot = xf.C;
    const bool has_skip = skip != nullptr, has_extra = extra != nullptr;
    const TensorH K = skip ? *skip : TensorH{}, E = extra ? *extra : TensorH{};
    m_->meta.push_back({"bw_affine", "C=" + std::to_string(xf.C) + " L>>" + std::to_string(xf.lshift),
                        xf.C * lscale(xf.lshift) * (3 + (has_extra ? 1 : 0)) + (has_skip ? K.C * lscale(K.lshift) : 0.0), 0, 0});
    m_->add_op([=](const RunCtx& c) -> int {
      BwAffineArgs a{};
      a.du = self->act(du.off);
      a.xf = self->act(xf.off);
      a.coef = reinterpret_cast<const float4*>(self->miscp(coef));
      a.skip = has_skip ? self->act(K.off) : nullptr;
      a.skip_mode = skip_mode;
      a.extra = has_extra ? self->act(E.off) : nullptr;
      a.out = self->act(out.off);
      a.C = xf.C;
      a.L = shiftL(c.Lbase, xf.lshift);
      a.du_C = du.C;
      a.du_c0 = du.C == xf.C ? 0 : c0;
      a.coef_stride = Ctot;
      a.coef_c0 = c0;
      a.extra_C = has_extra ? E.C : 0;
      a.extra_c0 = has_extra && E.C != xf.C ? c0 : 0;
      return launch_bw_affine(a, c.B, prec, c.st);
    });
  }
  void add_bw_add(const TensorH& x, const TensorH& y, const TensorH& out) {  // out = x + y (same shape)
    Builder* self = this;
    const int prec = m_->cfg.precision;
    m_->meta.push_back({"bw_add", "C=" + std::to_string(x.C) + " L>>" + std::to_string(x.lshift), 3 * x.C * lscale(x.lshift), 0, 0});
    m_->add_op([=](const RunCtx& c) -> int {
      const long long n = (long long)c.B * shiftL(c.Lbase, x.lshift) * x.C;
      return launch_bw_add(self->act(x.off), self->act(y.off), self->act(out.off), n, prec, c.st);
    });
  }
  // plain convolution of a raw tensor with transposed weights (no prologue, no statistics, zero bias)
  // aff_xf / aff_coef: the source is read through the two-tensor affine prologue P * src + Q * aff_xf + R (GroupNorm's backward folded
  // into this convolution's staging instead of a bw_affine pass in front of it)
  int add_conv_t(const TensorH& src, const float* W, int Cout_fwd, int Cin_fwd, int ktaps, int dil, const TensorH& out,
                 const BwFuse* fuse = nullptr, const TensorH* aff_xf = nullptr, size_t aff_coef = 0) {
    const std::vector<float> Wt = transpose_flip(W, Cout_fwd, Cin_fwd, ktaps);
    PackedConv pk(m_->cfg.precision);
    SegSpec g{src, 0, Cout_fwd, ktaps, dil, RESIZE_NONE, false, 0, 0, 0, 0};
    if (aff_xf) {
      g.aff2 = true;
      g.t2 = *aff_xf;
      g.coef_off = aff_coef;
    }
    g.w_off = pk.append(Wt.data(), Cin_fwd, Cout_fwd, ktaps, 0, Cout_fwd);
    return add_conv({g}, pk, std::vector<float>(Cin_fwd, 0.f), Cin_fwd, out, nullptr, 0, -1000, false, fuse);
  }

  // Gradient of one ResBlock with respect to its input(s):
  //   out = skip(resize(x)) + conv2(gelu(GN2(conv1(resize(gelu(GN1(x)))))))       (unet.py:307-316)
  // x may be the concatenation [x, x2] (up path, unet.py:156).  Consumes `dout` (released); returns dx and sets
  // *dx2 to the gradient of the second input when the block concatenates.
  TensorH resblock_backward(const BlockRec& r, const TensorH& dout, TensorH* dx2 = nullptr) {
    const BlockSpec& s = r.s;
    const std::string pre = s.prefix.empty() ? "" : s.prefix + ".";
    const int cin = s.cin, cout = s.cout;
    const int in_shift = r.x.lshift, out_shift = r.out.lshift;
    const bool cat = r.x2.id >= 0;
    // gradient at the input resolution from a tensor at the output resolution
    const int rs = s.resize == RESIZE_AVG2 ? BW_FROM_HALF : (s.resize == RESIZE_UP2 ? BW_FROM_DOUBLE : BW_SAME);
    // d gelu2 = conv2^T(dout);  du2 = . * gelu'(u2);  d h1 = GN2 backward
    TensorH t1 = new_tensor(cout, out_shift, false, false);
    const std::string c2 = pre + (cfg_dropout(m_->cfg) ? "post_cond.2" : "post_cond.1");
    const size_t pa = alloc_stats(cout, out_shift);
    const bool fuse = bw_fuse_enabled();  // the multiplication by gelu'(u) and its partial sums ride in the convolution's epilogue
    const BwFuse f2{{r.h1}, r.ss2, cout, pa};
    const int tr2 = add_conv_t(dout, P(c2 + ".weight"), cout, cout, 3, s.dil, t1, fuse ? &f2 : nullptr);
    if (!fuse) add_bw_act(t1, BW_SAME, r.h1, r.ss2, t1, pa);
    const size_t cf2 = add_gn_bw(cout, out_shift, pa, r.ss2, r.mr2, fuse ? tr2 : STAT_TILE);
    // d h1 = P * du2 + Q * h1 + R: evaluated by conv1^T while it stages its input (kernels.hpp SegDesc.src2) instead of a streaming
    // pass that writes d h1 and a convolution that reads it back.  VQVS_BW_AFF2=0: the separate bw_affine launch (A/B measurements).
    static const bool aff2 = getenv("VQVS_BW_AFF2") ? atoi(getenv("VQVS_BW_AFF2")) != 0 : true;
    if (!aff2) add_bw_affine(t1, r.h1, cf2, nullptr, BW_SAME, nullptr, t1);
    // d resize(gelu1) = conv1^T(d h1);  du1 = resize^T(.) * gelu'(u1);  dx = GN1 backward + skip path
    TensorH t2 = new_tensor(cin, out_shift, false, false);
    const size_t pb = alloc_stats(cin, in_shift);
    const bool fuse1 = fuse && rs == BW_SAME;  // (resizing blocks keep the separate kernel: their gradient changes resolution first)
    BwFuse f1{{r.x}, r.ss1, cin, pb};
    if (cat) f1.xf.push_back(r.x2);
    const int tr1 = add_conv_t(t1, P(pre + "pre_cond.2.weight"), cout, cin, 3, 1, t2, fuse1 ? &f1 : nullptr, aff2 ? &r.h1 : nullptr, cf2);
    release(t1);
    if (!cat) {
      TensorH du1 = rs != BW_SAME ? new_tensor(cin, in_shift, false, false) : t2;
      if (!fuse1) add_bw_act(t2, rs, r.x, r.ss1, du1, pb);
      if (rs != BW_SAME) release(t2);
      const size_t cf1 = add_gn_bw(cin, in_shift, pb, r.ss1, r.mr1, fuse1 ? tr1 : STAT_TILE);
      if (cin != cout) {
        TensorH t3 = new_tensor(cin, out_shift, false, false);
        add_conv_t(dout, P(pre + "skip.1.weight"), cout, cin, 1, 1, t3);
        add_bw_affine(du1, r.x, cf1, nullptr, BW_SAME, &t3, du1);
        release(t3);
      } else {
        add_bw_affine(du1, r.x, cf1, &dout, rs, nullptr, du1);
      }
      release(dout);
      return du1;
    }
    // concatenated input (never resized, always a 1x1 skip convolution): t2 and t3 are cin wide, the two sources
    // take their column slices; GroupNorm 1 runs over the concatenation
    const TensorH srcs[2] = {r.x, r.x2};
    int c0 = 0;
    for (int i = 0; i < 2 && !fuse1; ++i) {
      add_bw_act(t2, BW_SAME, srcs[i], r.ss1, t2, pb, c0, cin);
      c0 += srcs[i].C;
    }
    const size_t cf1 = add_gn_bw(cin, in_shift, pb, r.ss1, r.mr1, fuse1 ? tr1 : STAT_TILE);
    TensorH t3 = new_tensor(cin, out_shift, false, false);
    add_conv_t(dout, P(pre + "skip.1.weight"), cout, cin, 1, 1, t3);
    release(dout);
    TensorH outs[2];
    c0 = 0;
    for (int i = 0; i < 2; ++i) {
      outs[i] = new_tensor(srcs[i].C, in_shift, false, false);
      add_bw_affine(t2, srcs[i], cf1, nullptr, BW_SAME, &t3, outs[i], c0, cin);
      c0 += srcs[i].C;
    }
    release(t2);
    release(t3);
    if (dx2) *dx2 = outs[1];
    return outs[0];
  }

  void tap(const std::string& name, const TensorH& t) { m_->taps.push_back({name, t}); }

  // region sizes after planning
  size_t act_high = 0, stats_floats = 0, ss_floats = 0, misc_floats = 0;
  Blob blob;
  int es() const { return es_; }

 private:
  size_t arena_alloc(size_t bytes) {
    for (size_t i = 0; i < free_.size(); ++i) {
      if (free_[i].second >= bytes) {
        const size_t off = free_[i].first;
        if (free_[i].second == bytes) free_.erase(free_.begin() + i);
        else { free_[i].first += bytes; free_[i].second -= bytes; }
        return off;
      }
    }
    const size_t off = act_high;
    act_high += bytes;
    return off;
  }
  void arena_free(size_t off, size_t bytes) {
    free_.push_back({off, bytes});
    // coalesce
    std::sort(free_.begin(), free_.end());
    std::vector<std::pair<size_t, size_t>> merged;
    for (auto& f : free_) {
      if (!merged.empty() && merged.back().first + merged.back().second == f.first) merged.back().second += f.second;
      else merged.push_back(f);
    }
    // a free block that ends at the high-water mark is given back
    if (!merged.empty() && merged.back().first + merged.back().second == act_high) {
      // keep the high-water mark (arena size is the maximum ever needed)
    }
    free_ = merged;
  }

  vqvs_model* m_;
  const float* const* hp_;
  std::map<std::string, int> pidx_;
  int es_, maxB_, maxL_;
  int next_id_ = 0;
  int conv_seq_ = 0;
  std::map<int, size_t> sizes_, offs_;
  std::map<int, int> refs_;
  std::map<int, int> tile_rows_;  // rows per statistics tile of each tensor (set by its producer)
  std::vector<std::pair<size_t, size_t>> free_;
};

}  // namespace

int tensor_rows(int Lbase, int lshift) { return shiftL(Lbase, lshift); }

// The guidance gradients d log p / d activation are O(1e-6 ... 1e-3): in fp16 storage they would sit in or below the
// subnormal range (6e-5).  The backward schedule is linear in the gradient, so the fp16 mode carries it multiplied by 2^10
// from the head (gscale) to the last kernel (in_conv_bw), which divides it out again in fp32.  bf16 / fp32 have the range.
// The caller's guidance scale is NOT part of what the backward schedule carries (it is linear in it): in_conv_bw applies it in fp32,
// so neither a very large --classifier-scale can overflow fp16 nor a very small one push the gradients into its subnormals.
float grad_scale(int precision) { return precision == VQVS_PREC_F16 ? 1024.0f : 1.0f; }

int unet_rate(const vqvs_cfg& c) {  // UNetPredictor / UNetEncoder.downsample_rate (unet.py:182-184, 243-245): 2^(len(channel_mult) - 1)
  return 1 << (topo_of(c).levels() - 1);
}

int gn_groups(int ch) {  // unet.py:345-349
  int g = 32;
  while (ch % g) g /= 2;
  return g;
}

// Widths that are not multiples of 32 (vqvs_cfg.reserved[4] = the REAL base_channels of a predictor / encoder, 0 = base_channels itself):
// the handle is built at the PHYSICAL width base_channels = padded_base(real), whose channels come in blocks of q_p of which the
// first q carry the real channels and the rest are zero (weights, biases, affine parameters and FiLM rows of the pad channels are
// zero: a pad channel is exactly 0 in every tensor).  real = 2^k * q with q odd; q_p = the next power of two >= q, widened until
// 2^k * q_p is a multiple of 32.  Every real width m * real is a multiple of q and every GroupNorm group of the reference
// (unet.py:345-349: 32 groups halved until they divide the width -- a power of two) is a whole number of blocks, so the reference's
// groups are runs of physical channels too: the zero channels add nothing to a group's sums, and only the group COUNT and the number
// of values per group (inv_count) have to come from the real width.
struct PadMap {
  int q = 1, qp = 1;
  int phys(int c) const { return (c / q) * qp + c % q; }
};
PadMap pad_map(int real) {
  PadMap m;
  int k = 0;
  while (((real >> k) & 1) == 0 && k < 5) ++k;
  m.q = real >> k;
  m.qp = 1;
  while (m.qp < m.q) m.qp <<= 1;
  while (((1 << k) * m.qp) % 32) m.qp <<= 1;
  if (real % 32 == 0) m.qp = m.q;  // (nothing to pad)
  return m;
}
int padded_base(int real) {
  const PadMap m = pad_map(real);
  return real / m.q * m.qp;
}
int real_base(const vqvs_cfg& c) {
  const bool unet = c.kind == VQVS_KIND_PREDICTOR || c.kind == VQVS_KIND_ENCODER;
  return unet && c.reserved[4] > 0 ? c.reserved[4] : c.base_channels;
}

static int check_topology(const vqvs_cfg& c) {
  if (c.topology_set && c.kind != VQVS_KIND_PREDICTOR && c.kind != VQVS_KIND_ENCODER && c.kind != VQVS_KIND_CLASSIFIER)
    VQVS_FAIL(VQVS_ERR_ARG, "a custom topology is built for predictor, encoder and classifier handles only (kind %d)", c.kind);
  if (c.topology_set) {
    if (c.n_levels < 1 || c.n_levels > VQVS_MAX_LEVELS) VQVS_FAIL(VQVS_ERR_ARG, "n_levels must be in 1..%d (got %d)", VQVS_MAX_LEVELS, c.n_levels);
    for (int i = 0; i < c.n_levels; ++i)
      if (c.channel_mult[i] < 1 || c.channel_mult[i] * c.base_channels > 1024)
        VQVS_FAIL(VQVS_ERR_ARG, "channel_mult[%d] = %d: widths must be in base_channels..1024", i, c.channel_mult[i]);
    if (c.depth_mult < 1 || c.depth_mult > 8) VQVS_FAIL(VQVS_ERR_ARG, "depth_mult must be in 1..8 (got %d)", c.depth_mult);
    if (c.n_dilations < 0 || c.n_dilations > 12) VQVS_FAIL(VQVS_ERR_ARG, "at most 12 middle / output dilations (got %d)", c.n_dilations);
    for (int i = 0; i < c.n_dilations; ++i)
      if (c.dilations[i] < 1 || c.dilations[i] > 32) VQVS_FAIL(VQVS_ERR_ARG, "dilation %d is outside 1..32", c.dilations[i]);
    // Limits of the BUILDER, enforced here for every caller of the C ABI (the Python wrappers raise the same errors earlier):
    // * predictor: the output head (out.0.0 GroupNorm + out.1 conv, unet.py:113-116) is built for base_channels, the last up block
    //   returns channel_mult[0] * base_channels -- the reference constructs such a model and fails in forward;
    // * classifier: the attention pool splits the final width into heads of 64 channels (classifier.py:131-150 asserts the same).
    if (c.kind == VQVS_KIND_PREDICTOR && c.channel_mult[0] != 1)
      VQVS_FAIL(VQVS_ERR_ARG, "predictor channel_mult[0] must be 1 (got %d): the output head is built for base_channels", c.channel_mult[0]);
    if (c.kind == VQVS_KIND_CLASSIFIER) {
      const int cur = c.channel_mult[c.n_levels - 1] * c.base_channels;
      if (cur > 64 && cur % 64)
        VQVS_FAIL(VQVS_ERR_ARG, "classifier: the final width %d must be at most 64 or a multiple of 64 (attention heads of 64 channels)", cur);
    }
  }
  return 0;
}

// elements of the widest per-clip tensor a launch may address: level i holds max_T / 2^i rows of channel_mult[i] * base channels, and a
// concatenated input (the predictor's up path) is twice that wide.  (The default topology: max_T rows of 2 * base channels.)
static long long widest_clip_elems(const vqvs_cfg& c) {
  const Topo t = topo_of(c);
  long long best = 0;
  for (int i = 0; i < t.levels(); ++i) {
    const long long rows = ((long long)c.max_T >> i) + 1;
    best = std::max(best, rows * t.mult[i] * c.base_channels * 2);
  }
  return best;
}

int enumerate_params(const vqvs_cfg& c, std::vector<ParamDef>& out) {
  out.clear();
  if (int e = check_topology(c)) return e;
  const int base = c.base_channels;
  const bool drop = cfg_dropout(c);
  if (c.kind == VQVS_KIND_PREDICTOR || c.kind == VQVS_KIND_ENCPRED) {
    const int E = 4 * base;
    out.push_back({"time_embed.proj.weight", {E, E}});
    out.push_back({"time_embed.proj.bias", {E}});
    out.push_back({"time_embed_extra.1.weight", {E, E}});
    out.push_back({"time_embed_extra.1.bias", {E}});
    if (c.num_labels > 0) out.push_back({"class_embed.weight", {c.num_labels, E}});
    if (c.cond_channels > 0) {
      out.push_back({"cond_proj.weight", {base, c.cond_channels, 3}});
      out.push_back({"cond_proj.bias", {base}});
    }
    out.push_back({"in_conv.weight", {base, c.in_channels, 3}});
    out.push_back({"in_conv.bias", {base}});
    std::vector<BlockSpec> d, mdl, u;
    predictor_blocks(base, topo_of(c), d, mdl, u);
    for (auto& s : d) block_params(out, s.prefix, s, E, drop);
    for (auto& s : mdl) block_params(out, s.prefix, s, E, drop);
    for (auto& s : u) block_params(out, s.prefix, s, E, drop);
    out.push_back({"out.0.0.weight", {base}});
    out.push_back({"out.0.0.bias", {base}});
    out.push_back({"out.1.weight", {c.out_channels, base, 3}});
    out.push_back({"out.1.bias", {c.out_channels}});
    if (c.kind == VQVS_KIND_ENCPRED) {  // encoder_predictor.py:40-41: self.unet = UNetPredictor(...), self.out = Conv1d(bottleneck, latents, 1)
      for (auto& d : out) d.name = "unet." + d.name;
      out.push_back({"out.weight", {c.reserved[2], c.out_channels, 1}});
      out.push_back({"out.bias", {c.reserved[2]}});
    }
  } else if (c.kind == VQVS_KIND_ENCODER) {
    out.push_back({"in_conv.weight", {base, c.in_channels, 3}});
    out.push_back({"in_conv.bias", {base}});
    std::vector<BlockSpec> blocks;
    const int cur = encoder_blocks(base, topo_of(c), blocks);
    for (auto& s : blocks) block_params(out, s.prefix, s, 0, false);
    out.push_back({"out.0.0.weight", {cur}});
    out.push_back({"out.0.0.bias", {cur}});
    out.push_back({"out.1.weight", {c.out_channels, cur, 3}});
    out.push_back({"out.1.bias", {c.out_channels}});
  } else if (c.kind == VQVS_KIND_RESBLOCK) {
    BlockSpec s{"", c.rb_cin, c.rb_cout, c.rb_resize, c.rb_dilation, false};
    block_params(out, "", s, c.rb_emb_channels, drop);
  } else if (c.kind == VQVS_KIND_CLASSIFIER) {  // classifier.py:18-104, 131-150
    const int E = 4 * base, F = classifier_output_mult(c) * base;
    std::vector<BlockSpec> blocks;
    const int cur = classifier_blocks(base, topo_of(c), blocks);
    out.push_back({"stem.time_embed.proj.weight", {E, E}});
    out.push_back({"stem.time_embed.proj.bias", {E}});
    out.push_back({"stem.time_embed_extra.1.weight", {E, E}});
    out.push_back({"stem.time_embed_extra.1.bias", {E}});
    out.push_back({"stem.in_conv.weight", {base, 1, 3}});
    out.push_back({"stem.in_conv.bias", {base}});
    for (auto& s : blocks) block_params(out, s.prefix, s, E, false);
    out.push_back({"stem.out.0.0.weight", {cur}});
    out.push_back({"stem.out.0.0.bias", {cur}});
    out.push_back({"stem.out.1.qkv_proj.weight", {3 * cur, cur, 1}});
    out.push_back({"stem.out.1.qkv_proj.bias", {3 * cur}});
    out.push_back({"stem.out.1.c_proj.weight", {F, cur, 1}});
    out.push_back({"stem.out.1.c_proj.bias", {F}});
    out.push_back({"out.1.weight", {c.num_labels, F}});
    out.push_back({"out.1.bias", {c.num_labels}});
  } else if (c.kind == VQVS_KIND_MFCC_ENCODER) {  // conv_encoder.py:42-84; buffers of torchaudio.transforms.MFCC first
    const int version = c.reserved[1];
    const int n_fft = version == 2 ? 400 : 2 * MFCC_HOP, n_mels = version == 2 ? 80 : 40, mid = 12 * base;
    out.push_back({"mfcc.dct_mat", {n_mels, 13}});
    out.push_back({"mfcc.MelSpectrogram.spectrogram.window", {n_fft}});
    out.push_back({"mfcc.MelSpectrogram.mel_scale.fb", {n_fft / 2 + 1, n_mels}});
    out.push_back({"blocks.0.0.weight", {mid, 39, 3}});
    out.push_back({"blocks.0.0.bias", {mid}});
    out.push_back({"blocks.1.conv.weight", {mid, mid, 3}});
    out.push_back({"blocks.1.conv.bias", {mid}});
    out.push_back({"blocks.2.0.weight", {mid, mid, 4}});
    out.push_back({"blocks.2.0.bias", {mid}});
    for (int i = 3; i <= 8; ++i) {
      out.push_back({"blocks." + std::to_string(i) + ".conv.weight", {mid, mid, i <= 4 ? 3 : 1}});
      out.push_back({"blocks." + std::to_string(i) + ".conv.bias", {mid}});
    }
    out.push_back({"blocks.9.weight", {c.out_channels, mid, 1}});
    out.push_back({"blocks.9.bias", {c.out_channels}});
  } else {
    VQVS_FAIL(VQVS_ERR_ARG, "unknown model kind %d", c.kind);
  }
  return 0;
}

static int check_cfg(const vqvs_cfg& c) {
  if (c.precision != VQVS_PREC_F32 && c.precision != VQVS_PREC_BF16 && c.precision != VQVS_PREC_F16) VQVS_FAIL(VQVS_ERR_ARG, "bad precision %d", c.precision);
  if (c.max_batch < 1 || c.max_T < 1) VQVS_FAIL(VQVS_ERR_ARG, "max_batch/max_T must be positive");
  if (c.kind == VQVS_KIND_RESBLOCK) {
    if (c.rb_cin % 32 || c.rb_cout % 32 || c.rb_cin < 32 || c.rb_cout < 32) VQVS_FAIL(VQVS_ERR_ARG, "resblock channels must be multiples of 32");
    if (c.rb_dilation < 1 || c.rb_dilation > 32) VQVS_FAIL(VQVS_ERR_ARG, "resblock dilation must be in 1..32");
    if (c.rb_resize != RESIZE_NONE && c.rb_cin != c.rb_cout) VQVS_FAIL(VQVS_ERR_ARG, "resizing resblock cannot change channels");
    if (c.rb_emb_channels % 64) VQVS_FAIL(VQVS_ERR_ARG, "resblock emb channels must be a multiple of 64");
    return 0;
  }
  // the reference accepts any width (models/unet.py:17-30); here: powers of two from 32 to 128 (in_conv / out_conv distribute
  // C / 8 row pieces over a 256-thread workgroup; every convolution works on 32-channel K chunks)
  // predictor / encoder: any multiple of 32 up to 256 (unusual widths run the generic forms: 32-channel tiles, the stand-alone GroupNorm
  // pass, the row-per-thread output convolution); the guidance models' backward kernels want a power of two
  if (c.base_channels % 32 || c.base_channels < 32 || c.base_channels > 256)
    VQVS_FAIL(VQVS_ERR_ARG, "base_channels must be a multiple of 32 in 32..256 (got %d)", c.base_channels);
  // (round 6: the guidance models take any multiple of 32 too -- in_conv_bw and bw_act have row-per-thread / partial-pass forms for widths
  //  whose octet count is not a power of two; the MFCC encoder's 12 * base wide stack stays on powers of two)
  if (c.kind == VQVS_KIND_MFCC_ENCODER && (c.base_channels & (c.base_channels - 1)))
    VQVS_FAIL(VQVS_ERR_ARG, "base_channels must be a power of two for this kind of model (got %d)", c.base_channels);
  if ((c.kind == VQVS_KIND_PREDICTOR || c.kind == VQVS_KIND_ENCODER) && c.reserved[4] != 0) {
    // a padded handle (pad_map): reserved[4] = the real base_channels, base_channels = its physical width
    if (c.reserved[4] < 1 || c.reserved[4] > c.base_channels || padded_base(c.reserved[4]) != c.base_channels)
      VQVS_FAIL(VQVS_ERR_ARG, "real base_channels %d (reserved[4]) does not pad to base_channels %d (expected %d)", c.reserved[4], c.base_channels,
                c.reserved[4] >= 1 ? padded_base(c.reserved[4]) : 0);
  }
  if (c.in_channels < 1 || c.in_channels > 64) VQVS_FAIL(VQVS_ERR_ARG, "in_channels must be in 1..64 (got %d)", c.in_channels);
  if (c.in_channels != 1 && c.kind != VQVS_KIND_PREDICTOR && c.kind != VQVS_KIND_ENCODER)
    VQVS_FAIL(VQVS_ERR_ARG, "in_channels = %d: only predictor and encoder handles take more than one input channel (the classifier stem and the "
                            "encoder predictor are mono in the reference too, classifier.py:73, encoder_predictor.py:40)", c.in_channels);
  if (c.kind == VQVS_KIND_MFCC_ENCODER) {
    if (c.precision != VQVS_PREC_F32) VQVS_FAIL(VQVS_ERR_ARG, "the MFCC encoder feeds the VQ layer (bit-exact indices): fp32 precision only");
    if (c.reserved[1] != 1 && c.reserved[1] != 2) VQVS_FAIL(VQVS_ERR_ARG, "ConvMFCCEncoder version must be 1 or 2 (got %d)", c.reserved[1]);
    if (c.out_channels % 32 || c.out_channels < 32) VQVS_FAIL(VQVS_ERR_ARG, "encoder out_channels must be a multiple of 32");
    if (c.max_T < 2 * 400) VQVS_FAIL(VQVS_ERR_ARG, "max_T must be at least 800 samples");
    return 0;
  }
  if (int e = check_topology(c)) return e;
  {
    const int rate = unet_rate(c);
    if (c.max_T % rate) VQVS_FAIL(VQVS_ERR_ARG, "max_T must be a multiple of the downsample rate %d (got %d)", rate, c.max_T);
  }
  // the kernels address rows of one clip with 32-bit byte offsets (buffer loads): the widest per-clip tensor must stay below 2 GiB
  // (4 bytes per element in the fp32 mode; the widest tensor a launch addresses is a level's concatenated input, 2 x its width)
  if (widest_clip_elems(c) * 4 > 0x7fffffffLL)
    VQVS_FAIL(VQVS_ERR_ARG, "max_T=%d is too long for base_channels=%d with this topology (a clip's widest tensor would exceed 2 GiB)", c.max_T, c.base_channels);
  if (c.kind == VQVS_KIND_PREDICTOR) {
    if (c.out_channels != 1 && (c.out_channels % 32)) VQVS_FAIL(VQVS_ERR_ARG, "out_channels must be 1 or a multiple of 32");
    if (c.cond_channels % 32) VQVS_FAIL(VQVS_ERR_ARG, "cond_channels must be a multiple of 32");
    if (c.reserved[3] != 0 && c.reserved[3] != 1 && (c.reserved[3] <= LEN_ABS || c.reserved[3] > LEN_ABS + (1 << 20)))
      VQVS_FAIL(VQVS_ERR_ARG, "cond length code (reserved[3]) must be 0 (T/256), 1 (T/320) or 1000 + rows per clip (1 .. 2^20)");
  } else if (c.kind == VQVS_KIND_ENCPRED) {
    if (c.out_channels % 32 || c.out_channels < 32 || c.out_channels > 256) VQVS_FAIL(VQVS_ERR_ARG, "bottleneck_dim must be a multiple of 32 in 32..256");
    if (c.cond_channels) VQVS_FAIL(VQVS_ERR_ARG, "the encoder predictor's UNet is unconditional");
    if (c.num_labels) VQVS_FAIL(VQVS_ERR_ARG, "the encoder predictor's UNet has no class embedding");
    if (c.reserved[2] < 1) VQVS_FAIL(VQVS_ERR_ARG, "num_latents (reserved[2]) must be positive");
    if (c.reserved[1] < 1 || c.max_T % c.reserved[1]) VQVS_FAIL(VQVS_ERR_ARG, "downsample_rate %d must divide max_T", c.reserved[1]);
  } else if (c.kind == VQVS_KIND_CLASSIFIER) {
    if (c.num_labels < 1 || c.num_labels > 8192) VQVS_FAIL(VQVS_ERR_ARG, "classifier num_labels must be in 1..8192 (got %d)", c.num_labels);
    if (c.max_T % (2 * unet_rate(c))) VQVS_FAIL(VQVS_ERR_ARG, "classifier max_T must be a multiple of %d (got %d)", 2 * unet_rate(c), c.max_T);
    if (c.reserved[1] < 0 || c.reserved[1] * c.base_channels > 4096) VQVS_FAIL(VQVS_ERR_ARG, "classifier output_mult (reserved[1]) out of range (got %d)", c.reserved[1]);
    if (c.topology_set && c.n_dilations != 0) VQVS_FAIL(VQVS_ERR_ARG, "the classifier stem has no dilation list");
  } else {
    if (c.out_channels % 32 || c.out_channels < 32) VQVS_FAIL(VQVS_ERR_ARG, "encoder out_channels must be a multiple of 32");
  }
  return 0;
}

int build_model(vqvs_model* m, const float* const* hp) {
  const vqvs_cfg& c = m->cfg;
  if (int e = check_cfg(c)) return e;
  if (int e = enumerate_params(c, m->params)) return e;
  const int prec = c.precision;
  const int base = c.base_channels;
  const int in_ch = c.in_channels;
  // NOTE: the builder object must outlive the ops (lambdas capture it for pointer resolution).
  auto keep = std::make_shared<Builder>(m, hp);
  m->keepalive = keep;
  Builder* bp = keep.get();
  Builder& b = *bp;

  if (c.kind == VQVS_KIND_RESBLOCK) {
    BlockSpec s{"", c.rb_cin, c.rb_cout, c.rb_resize, c.rb_dilation, false};
    const int E = c.rb_emb_channels;
    TensorH x = b.new_tensor(c.rb_cin, 0, false, true);
    const int Cin = c.rb_cin;
    m->meta.push_back({"nct_to_ntc", "", 0, 0, 0});
    m->add_op([=](const RunCtx& r) -> int {
      return launch_nct_to_ntc(r.x, bp->act(x.off), bp->statp(x.stats_off), r.B, Cin, r.Lbase, ntiles_of(r.Lbase), prec, r.st);
    });
    size_t film_misc = 0;
    if (E) {
      const size_t gemb = b.alloc_misc((size_t)c.max_batch * E);
      film_misc = b.alloc_misc((size_t)c.max_batch * 2 * c.rb_cout);
      const size_t w_off = b.blob_f32_transposed(b.P("cond_layers.1.weight"), 2 * c.rb_cout, E);
      const size_t bias_off = b.blob_f32("cond_layers.1.bias");
      const int R = 2 * c.rb_cout;
      m->meta.push_back({"film", "", 0, 0, 0});
      m->add_op([=](const RunCtx& r) -> int {
        if (!r.emb) VQVS_FAIL(VQVS_ERR_ARG, "resblock handle was built with an embedding; d_emb is NULL");
        if (int e = launch_gelu_rows(r.emb, bp->miscp(gemb), r.B * E, r.st)) return e;
        FilmArgs f{bp->miscp(gemb), reinterpret_cast<const float*>(bp->wp(w_off)), reinterpret_cast<const float*>(bp->wp(bias_off)),
                   bp->miscp(film_misc), E, R};
        return launch_film(f, r.B, r.st);
      });
    }
    TensorH y = b.resblock("", s, {x}, E != 0, 0, 2 * c.rb_cout, film_misc);
    const int Cout = c.rb_cout;
    m->meta.push_back({"ntc_to_nct", "", 0, 0, 0});
    m->add_op([=](const RunCtx& r) -> int {
      return launch_ntc_to_nct(bp->act(y.off), r.out, r.B, Cout, shiftL(r.Lbase, y.lshift), prec, r.st);
    });
    b.tap("out", y);
  } else if (c.kind == VQVS_KIND_PREDICTOR || c.kind == VQVS_KIND_ENCPRED) {
    // ENCPRED (encoder_predictor.py:25-58): the same UNet under the "unet." prefix with a bottleneck output, every
    // intermediate kept resident for the backward schedule appended below
    const bool encpred = c.kind == VQVS_KIND_ENCPRED;
    const std::string px = encpred ? "unet." : "";
    b.persist = encpred;
    const int E = 4 * base;
    std::vector<BlockSpec> d, mdl, u;
    predictor_blocks(base, topo_of(c), d, mdl, u);
    for (auto* v : {&d, &mdl, &u})
      for (auto& sp : *v) sp.prefix = px + sp.prefix;
    // ---- embedding + all blocks' FiLM rows
    std::vector<float> freqs(E / 2, 0.0f);
    {
      // wavegrad.py:361-369 (float32 tensor math) at the REAL embedding width 4 * real base; a padded handle keeps the real entries at
      // their physical positions (the halves of the embedding -- cos | sin -- are whole numbers of blocks), pad entries are 0 and meet
      // zero weight rows
      const int Er = 4 * real_base(c);
      const PadMap pm = pad_map(real_base(c));
      for (int i = 0; i < Er / 2; ++i)
        freqs[pm.phys(i)] = (float)(std::exp((double)(float)(-std::log(100.0 / 0.1)) * (double)i / (double)(Er / 2 - 1))) * 100.0f;
    }
    const size_t freq_off = b.blob.add(freqs.data(), freqs.size() * 4);
    const size_t w1 = b.blob_f32_transposed(b.P(px + "time_embed.proj.weight"), E, E), b1 = b.blob_f32(px + "time_embed.proj.bias");
    const size_t w2 = b.blob_f32_transposed(b.P(px + "time_embed_extra.1.weight"), E, E), b2 = b.blob_f32(px + "time_embed_extra.1.bias");
    const size_t ce = c.num_labels > 0 ? b.blob_f32(px + "class_embed.weight") : 0;
    const size_t emb_off = b.alloc_misc((size_t)c.max_batch * E);
    const size_t gemb_off = b.alloc_misc((size_t)c.max_batch * E);
    m->emb_misc_off = emb_off;
    m->emb_E = E;
    const int NL = c.num_labels;
    m->meta.push_back({"time_embed", "", 0, 0, 0});
    m->add_op([=](const RunCtx& r) -> int {
      TimeEmbedArgs a{};
      a.ts = r.ts;
      a.freqs = reinterpret_cast<const float*>(bp->wp(freq_off));
      a.w1 = reinterpret_cast<const float*>(bp->wp(w1));
      a.b1 = reinterpret_cast<const float*>(bp->wp(b1));
      a.w2 = reinterpret_cast<const float*>(bp->wp(w2));
      a.b2 = reinterpret_cast<const float*>(bp->wp(b2));
      a.class_embed = NL > 0 ? reinterpret_cast<const float*>(bp->wp(ce)) : nullptr;
      a.labels = r.labels;
      a.num_labels = NL;
      a.emb = bp->miscp(emb_off);
      a.gemb = bp->miscp(gemb_off);
      a.E = E;
      return launch_time_embed(a, r.B, r.st);
    });
    std::vector<const BlockSpec*> all;
    for (auto& s : d) all.push_back(&s);
    for (auto& s : mdl) all.push_back(&s);
    for (auto& s : u) all.push_back(&s);
    std::vector<int> film_row(all.size());
    int R = 0;
    std::vector<float> Wall, ball;
    for (size_t i = 0; i < all.size(); ++i) {
      film_row[i] = R;
      const float* W = b.P(all[i]->prefix + ".cond_layers.1.weight");
      const float* bb = b.P(all[i]->prefix + ".cond_layers.1.bias");
      const int rows = 2 * all[i]->cout;
      Wall.insert(Wall.end(), W, W + (size_t)rows * E);
      ball.insert(ball.end(), bb, bb + rows);
      R += rows;
    }
    const size_t wall_off = b.blob_f32_transposed(Wall.data(), R, E);
    const size_t ball_off = b.blob.add(ball.data(), ball.size() * 4);
    const size_t film_off = b.alloc_misc((size_t)c.max_batch * R);
    m->meta.push_back({"film", "", 0, 0, 0});
    m->add_op([=](const RunCtx& r) -> int {
      FilmArgs f{bp->miscp(gemb_off), reinterpret_cast<const float*>(bp->wp(wall_off)), reinterpret_cast<const float*>(bp->wp(ball_off)),
                 bp->miscp(film_off), E, R};
      return launch_film(f, r.B, r.st);
    });
    // ---- conditioning projection (unet.py:46-47, 138-139)
    TensorH condp{};
    const bool has_cond = c.cond_channels > 0;
    if (has_cond) {
      // conditioning rows per clip: T / 256 behind a UNet encoder, (T / 160 + 1 - 2) / 2 + 1 = T / 320 behind the MFCC encoder
      // or, reserved[3] = 1000 + rows, exactly that many rows for any T (nearest up-sampling to T with PyTorch's index formula, in_conv)
      const int cond_ls = c.reserved[3] >= LEN_ABS ? c.reserved[3] : (c.reserved[3] == 1 ? LEN_HALF : 8);
      TensorH ct = b.new_tensor(c.cond_channels, cond_ls, false, false);
      const int CC = c.cond_channels;
      m->meta.push_back({"nct_to_ntc", "", 0, 0, 0});
      m->add_op([=](const RunCtx& r) -> int {
        return launch_nct_to_ntc(r.cond, bp->act(ct.off), nullptr, r.B, CC, shiftL(r.Lbase, cond_ls), 0, prec, r.st);
      });
      condp = b.new_tensor(base, cond_ls, false, false);
      PackedConv pk(prec);
      Builder::SegSpec g{ct, 0, CC, 3, 1, RESIZE_NONE, false, 0, 0, 0, 0};
      g.w_off = pk.append(b.P(px + "cond_proj.weight"), base, CC, 3, 0, CC);
      const float* bb = b.P(px + "cond_proj.bias");
      b.add_conv({g}, pk, std::vector<float>(bb, bb + base), base, condp, nullptr, 0);
      b.release(ct);
    }
    // ---- in_conv
    TensorH h = b.new_tensor(base, 0, false, true);
    {
      const size_t w = b.blob_f32(px + "in_conv.weight"), bi = b.blob_f32(px + "in_conv.bias");
      const TensorH cp = condp;
      m->meta.push_back({"in_conv", std::to_string(in_ch) + "->" + std::to_string(base), base + (has_cond ? base * lscale(condp.lshift) : 0.0), 4.0 * in_ch, 0});
      m->add_op([=](const RunCtx& r) -> int {
        InConvArgs a{};
        a.x = r.x;
        a.Cin = in_ch;
        a.w = reinterpret_cast<const float*>(bp->wp(w));
        a.bias = reinterpret_cast<const float*>(bp->wp(bi));
        a.condp = has_cond ? bp->act(cp.off) : nullptr;
        a.cond_len = has_cond ? shiftL(r.Lbase, cp.lshift) : 0;
        a.out = bp->act(h.off);
        a.stats = bp->statp(h.stats_off);
        a.C = base;
        a.T = r.Lbase;
        a.ntiles = ntiles_of(r.Lbase);
        return launch_in_conv(a, r.B, prec, r.st);
      });
    }
    if (has_cond) b.release(condp);
    b.tap("in_conv", h);
    // ---- down / middle / up with the skip stack (unet.py:141-160)
    std::vector<TensorH> skips{h};
    b.retain(h);  // one reference held by `h`, one by the stack
    size_t bi = 0;
    std::vector<Builder::BlockRec> recs(all.size());
    auto recp = [&](size_t i) { return encpred ? &recs[i] : nullptr; };
    for (auto& s : d) {
      TensorH o = b.resblock(s.prefix, s, {h}, true, film_row[bi], R, film_off, recp(bi));
      ++bi;
      b.release(h);
      h = o;
      skips.push_back(h);
      b.retain(h);
      b.tap(s.prefix, h);
    }
    for (auto& s : mdl) {
      TensorH o = b.resblock(s.prefix, s, {h}, true, film_row[bi], R, film_off, recp(bi));
      ++bi;
      b.release(h);
      h = o;
      b.tap(s.prefix, h);
    }
    for (auto& s : u) {
      TensorH o;
      if (s.cat) {
        TensorH sk = skips.back();
        skips.pop_back();
        o = b.resblock(s.prefix, s, {h, sk}, true, film_row[bi], R, film_off, recp(bi));
        b.release(sk);
      } else {
        o = b.resblock(s.prefix, s, {h}, true, film_row[bi], R, film_off, recp(bi));
      }
      ++bi;
      b.release(h);
      h = o;
      b.tap(s.prefix, h);
    }
    // ---- output head (unet.py:113-116, 162)
    const size_t ss = b.alloc_ss(base);
    const size_t mr_out = encpred ? b.alloc_ss(base) : 0;
    b.add_gn({h}, px + "out.0.0", false, 0, 0, 0, ss, encpred, mr_out);
    if (c.out_channels == 1) {
      const float* W = b.P(px + "out.1.weight");  // [1][base][3] -> [3][base]
      std::vector<float> wt(3 * base);
      for (int k = 0; k < 3; ++k)
        for (int ci = 0; ci < base; ++ci) wt[k * base + ci] = W[ci * 3 + k];
      const size_t w = b.blob.add(wt.data(), wt.size() * 4);
      const float bias = b.P(px + "out.1.bias")[0];
      m->meta.push_back({"out_conv", std::to_string(base) + "->1", (double)base, 4.0, 0});
      m->add_op([=](const RunCtx& r) -> int {
        OutConvArgs a{};
        a.in = bp->act(h.off);
        a.ss = reinterpret_cast<const float2*>(bp->ssp(ss));
        a.w = reinterpret_cast<const float*>(bp->wp(w));
        a.bias = bias;
        a.out = r.out;
        a.C = base;
        a.L = r.Lbase;
        return launch_out_conv(a, r.B, prec, r.st);
      });
    } else {
      TensorH o = b.new_tensor(c.out_channels, 0, true, false);
      PackedConv pk(prec);
      Builder::SegSpec g{h, 0, base, 3, 1, RESIZE_NONE, true, ss, base, 0, 0};
      g.w_off = pk.append(b.P(px + "out.1.weight"), c.out_channels, base, 3, 0, base);
      const float* bb = b.P(px + "out.1.bias");
      b.add_conv({g}, pk, std::vector<float>(bb, bb + c.out_channels), c.out_channels, o, nullptr, 0);
      const int OC = c.out_channels;
      if (!encpred) {
        m->meta.push_back({"ntc_to_nct", "", 0, 0, 0});
        m->add_op([=](const RunCtx& r) -> int { return launch_ntc_to_nct(bp->act(o.off), r.out, r.B, OC, r.Lbase, 0, r.st); });
      } else {
        // ---- EncoderPredictor head + backward schedule (vq_vae.py:125-130: grads of the summed cross-entropy)
        const int D = c.reserved[2], rate = c.reserved[1];
        const size_t hw = b.blob_f32("out.weight"), hb = b.blob_f32("out.bias");
        b.persist = false;
        TensorH dO = b.new_tensor(OC, 0, false, false);
        const int es = b.es();
        m->meta.push_back({"enc_head", "D=" + std::to_string(D), 0, 0, 0});
        m->add_op([=](const RunCtx& r) -> int {
          if (r.backward) VQVS_HIP(hipMemsetAsync(bp->act(dO.off), 0, (size_t)r.B * r.Lbase * OC * es, r.st));
          EncHeadArgs a{};
          a.o = reinterpret_cast<const float*>(bp->act(o.off));
          a.w = reinterpret_cast<const float*>(bp->wp(hw));
          a.bias = reinterpret_cast<const float*>(bp->wp(hb));
          a.logits = r.out;
          a.targets = r.backward ? r.labels : nullptr;
          a.gscale = grad_scale(prec);  // (the caller's scale multiplies the finished gradient in fp32, in_conv_bw: the 2-byte gradient tensors never see it)
          a.dO = bp->act(dO.off);
          a.Cb = OC;
          a.D = D;
          a.T = r.Lbase;
          a.T1 = r.Lbase / rate;
          a.rate = rate;
          return launch_enc_head(a, r.B, prec, r.st);
        });
        m->cur_phase = 1;
        // out.1 (3-tap conv base -> bottleneck over gelu(GN(h))) backwards: conv^T, gelu', GroupNorm backward
        TensorH t = b.new_tensor(base, 0, false, false);
        const size_t po = b.alloc_stats(base, 0);
        const bool fuse = Builder::bw_fuse_enabled();
        const Builder::BwFuse fo{{h}, ss, base, po};
        const int tro = b.add_conv_t(dO, b.P(px + "out.1.weight"), OC, base, 3, 1, t, fuse ? &fo : nullptr);
        b.release(dO);
        if (!fuse) b.add_bw_act(t, BW_SAME, h, ss, t, po);
        const size_t cfo = b.add_gn_bw(base, 0, po, ss, mr_out, fuse ? tro : STAT_TILE);
        b.add_bw_affine(t, h, cfo, nullptr, BW_SAME, nullptr, t);
        // blocks in reverse; gradients of skip-stack tensors wait in `pending` until the down path reaches their producer
        std::map<int, TensorH> pending;
        TensorH g2 = t;
        auto add_pending = [&](const TensorH& produced) {  // produced = forward tensor whose gradient g2 currently is
          auto it = pending.find(produced.id);
          if (it == pending.end()) return;
          b.add_bw_add(g2, it->second, g2);
          b.release(it->second);
          pending.erase(it);
        };
        for (size_t i = all.size(); i-- > 0;) {
          add_pending(recs[i].out);
          TensorH dsk{};
          g2 = b.resblock_backward(recs[i], g2, &dsk);
          if (recs[i].x2.id >= 0) pending[recs[i].x2.id] = dsk;
        }
        add_pending(recs[0].x);  // in_conv output: also the first skip-stack entry
        const TensorH gi = g2;
        const size_t inw = b.blob_f32(px + "in_conv.weight");
        m->meta.push_back({"in_conv_bw", std::to_string(base) + "->1", (double)base, 4.0, 0});
        m->add_op([=](const RunCtx& r) -> int {
          InConvBwArgs a{};
          a.dh = bp->act(gi.off);
          a.w = reinterpret_cast<const float*>(bp->wp(inw));
          a.out = r.grad_out;
          a.C = base;
          a.T = r.Lbase;
          a.out_scale = r.gscale / grad_scale(prec);
          return launch_in_conv_bw(a, r.B, prec, r.st);
        });
        m->cur_phase = 0;
      }
    }
  } else if (c.kind == VQVS_KIND_CLASSIFIER) {
    // Classifier.forward (classifier.py:31-36, 111-121) + the input gradient used by cond_fn (sample_diffusion.py:34-42)
    const int E = 4 * base, F = classifier_output_mult(c) * base, NL = c.num_labels;
    std::vector<BlockSpec> blocks;
    const int cur = classifier_blocks(base, topo_of(c), blocks);
    b.persist = true;  // the backward pass re-reads every block input, h1 and their GroupNorm coefficients
    std::vector<float> freqs(E / 2);
    for (int i = 0; i < E / 2; ++i)
      freqs[i] = (float)(std::exp((double)(float)(-std::log(100.0 / 0.1)) * (double)i / (double)(E / 2 - 1))) * 100.0f;
    const size_t freq_off = b.blob.add(freqs.data(), freqs.size() * 4);
    const size_t w1 = b.blob_f32_transposed(b.P("stem.time_embed.proj.weight"), E, E), b1 = b.blob_f32("stem.time_embed.proj.bias");
    const size_t w2 = b.blob_f32_transposed(b.P("stem.time_embed_extra.1.weight"), E, E), b2 = b.blob_f32("stem.time_embed_extra.1.bias");
    const size_t emb_off = b.alloc_misc((size_t)c.max_batch * E);
    const size_t gemb_off = b.alloc_misc((size_t)c.max_batch * E);
    m->emb_misc_off = emb_off;
    m->emb_E = E;
    m->meta.push_back({"time_embed", "", 0, 0, 0});
    m->add_op([=](const RunCtx& r) -> int {
      TimeEmbedArgs a{};
      a.ts = r.ts;
      a.freqs = reinterpret_cast<const float*>(bp->wp(freq_off));
      a.w1 = reinterpret_cast<const float*>(bp->wp(w1));
      a.b1 = reinterpret_cast<const float*>(bp->wp(b1));
      a.w2 = reinterpret_cast<const float*>(bp->wp(w2));
      a.b2 = reinterpret_cast<const float*>(bp->wp(b2));
      a.emb = bp->miscp(emb_off);
      a.gemb = bp->miscp(gemb_off);
      a.E = E;
      return launch_time_embed(a, r.B, r.st);
    });
    std::vector<int> film_row(blocks.size());
    int R = 0;
    std::vector<float> Wall, ball;
    for (size_t i = 0; i < blocks.size(); ++i) {
      film_row[i] = R;
      const float* W = b.P(blocks[i].prefix + ".cond_layers.1.weight");
      const float* bb = b.P(blocks[i].prefix + ".cond_layers.1.bias");
      const int rows = 2 * blocks[i].cout;
      Wall.insert(Wall.end(), W, W + (size_t)rows * E);
      ball.insert(ball.end(), bb, bb + rows);
      R += rows;
    }
    const size_t wall_off = b.blob_f32_transposed(Wall.data(), R, E);
    const size_t ball_off = b.blob.add(ball.data(), ball.size() * 4);
    const size_t film_off = b.alloc_misc((size_t)c.max_batch * R);
    m->meta.push_back({"film", "", 0, 0, 0});
    m->add_op([=](const RunCtx& r) -> int {
      FilmArgs f{bp->miscp(gemb_off), reinterpret_cast<const float*>(bp->wp(wall_off)), reinterpret_cast<const float*>(bp->wp(ball_off)),
                 bp->miscp(film_off), E, R};
      return launch_film(f, r.B, r.st);
    });
    TensorH h = b.new_tensor(base, 0, false, true);
    const size_t inw = b.blob_f32("stem.in_conv.weight"), inb = b.blob_f32("stem.in_conv.bias");
    m->meta.push_back({"in_conv", "1->" + std::to_string(base), (double)base, 4.0, 0});
    m->add_op([=](const RunCtx& r) -> int {
      InConvArgs a{};
      a.x = r.x;
      a.w = reinterpret_cast<const float*>(bp->wp(inw));
      a.bias = reinterpret_cast<const float*>(bp->wp(inb));
      a.condp = nullptr;
      a.cond_len = 0;
      a.out = bp->act(h.off);
      a.stats = bp->statp(h.stats_off);
      a.C = base;
      a.T = r.Lbase;
      a.ntiles = ntiles_of(r.Lbase);
      return launch_in_conv(a, r.B, prec, r.st);
    });
    b.tap("stem.in_conv", h);
    std::vector<Builder::BlockRec> recs(blocks.size());
    for (size_t i = 0; i < blocks.size(); ++i) {
      h = b.resblock(blocks[i].prefix, blocks[i], {h}, true, film_row[i], R, film_off, &recs[i]);
      b.tap(blocks[i].prefix, h);
    }
    // ---- head: GroupNorm + GELU + attention pool (query token only) + c_proj + GELU + Linear, and its backward
    const size_t ssh = b.alloc_ss(cur), mrh = b.alloc_ss(cur);
    b.add_gn({h}, "stem.out.0.0", false, 0, 0, 0, ssh, true, mrh);
    const int ch = std::min(cur, 64), heads = cur / ch;
    const float* Wqkv = b.P("stem.out.1.qkv_proj.weight");  // [3*cur][cur][1]: q rows, k rows, v rows
    const float* bqkv = b.P("stem.out.1.qkv_proj.bias");
    const double sc2 = 1.0 / std::sqrt((double)ch);  // both q and k carry ch^-1/4 (classifier.py:181-186)
    std::vector<float> rvec((size_t)heads * cur), c0(heads);
    for (int hd = 0; hd < heads; ++hd) {
      double cc = 0.0;
      for (int j = 0; j < ch; ++j) cc += (double)bqkv[hd * ch + j] * (double)bqkv[cur + hd * ch + j];
      c0[hd] = (float)(cc * sc2);
      for (int ci = 0; ci < cur; ++ci) {
        double acc = 0.0;
        for (int j = 0; j < ch; ++j) acc += (double)bqkv[hd * ch + j] * (double)Wqkv[(size_t)(cur + hd * ch + j) * cur + ci];
        rvec[(size_t)hd * cur + ci] = (float)(acc * sc2);
      }
    }
    const size_t r_off = b.blob.add(rvec.data(), rvec.size() * 4), c0_off = b.blob.add(c0.data(), c0.size() * 4);
    const size_t wv_off = b.blob.add(Wqkv + (size_t)2 * cur * cur, (size_t)cur * cur * 4);
    const size_t wvT_off = b.blob_f32_transposed(Wqkv + (size_t)2 * cur * cur, cur, cur);
    const size_t wcT_off = b.blob_f32_transposed(b.P("stem.out.1.c_proj.weight"), F, cur);
    const size_t wlT_off = b.blob_f32_transposed(b.P("out.1.weight"), NL, F);
    const size_t bv_off = b.blob.add(bqkv + 2 * cur, (size_t)cur * 4);
    const size_t wc_off = b.blob_f32("stem.out.1.c_proj.weight"), bc_off = b.blob_f32("stem.out.1.c_proj.bias");
    const size_t wl_off = b.blob_f32("out.1.weight"), bl_off = b.blob_f32("out.1.bias");
    b.persist = false;
    TensorH dh = b.new_tensor(cur, h.lshift, false, false);
    const int groups_h = gn_groups(cur);
    const TensorH hl = h;
    m->meta.push_back({"cls_head", "C=" + std::to_string(cur), 0, 0, 0});
    m->add_op([=](const RunCtx& r) -> int {
      HeadArgs a{};
      a.h = bp->act(hl.off);
      a.ss = reinterpret_cast<const float2*>(bp->ssp(ssh));
      a.mr = reinterpret_cast<const float2*>(bp->ssp(mrh));
      a.C = cur;
      a.L = shiftL(r.Lbase, hl.lshift);
      a.heads = heads;
      a.F = F;
      a.NL = NL;
      a.groups = groups_h;
      a.inv_count = 1.0 / ((double)(cur / groups_h) * (double)a.L);
      a.r = reinterpret_cast<const float*>(bp->wp(r_off));
      a.c0 = reinterpret_cast<const float*>(bp->wp(c0_off));
      a.wv = reinterpret_cast<const float*>(bp->wp(wv_off));
      a.wvT = reinterpret_cast<const float*>(bp->wp(wvT_off));
      a.wcT = reinterpret_cast<const float*>(bp->wp(wcT_off));
      a.wlT = reinterpret_cast<const float*>(bp->wp(wlT_off));
      a.bv = reinterpret_cast<const float*>(bp->wp(bv_off));
      a.wc = reinterpret_cast<const float*>(bp->wp(wc_off));
      a.bc = reinterpret_cast<const float*>(bp->wp(bc_off));
      a.wl = reinterpret_cast<const float*>(bp->wp(wl_off));
      a.bl = reinterpret_cast<const float*>(bp->wp(bl_off));
      a.logits = r.out;
      a.labels = r.backward ? r.labels : nullptr;
      a.gscale = grad_scale(prec);  // (the caller's scale multiplies the finished gradient in fp32, in_conv_bw: the 2-byte gradient tensors never see it)
      a.dh = bp->act(dh.off);
      return launch_cls_head(a, r.B, prec, r.st);
    });
    // ---- backward schedule (phase 1): blocks in reverse, then the 1 -> base input convolution
    m->cur_phase = 1;
    TensorH g = dh;
    for (size_t i = blocks.size(); i-- > 0;) g = b.resblock_backward(recs[i], g);
    {
      const TensorH gi = g;
      m->meta.push_back({"in_conv_bw", std::to_string(base) + "->1", (double)base, 4.0, 0});
      m->add_op([=](const RunCtx& r) -> int {
        InConvBwArgs a{};
        a.dh = bp->act(gi.off);
        a.w = reinterpret_cast<const float*>(bp->wp(inw));
        a.out = r.grad_out;
        a.C = base;
        a.T = r.Lbase;
        a.out_scale = r.gscale / grad_scale(prec);
        return launch_in_conv_bw(a, r.B, prec, r.st);
      });
    }
    m->cur_phase = 0;
  } else if (c.kind == VQVS_KIND_MFCC_ENCODER) {
    // ConvMFCCEncoder.forward (conv_encoder.py:90-110).  The convolution stack runs on the fused MFMA kernel in the fp32
    // mode: every layer is a raw segment with the GELU in the EPILOGUE (out = skip + gelu(conv(x)), conv_encoder.py:113-120);
    // Conv1d(k = 4, stride = 2, padding = 1) is a 3-tap stride-1 convolution over the input viewed as [pairs][2 * mid]:
    //   out[t] = W0 x[2t-1] + W1 x[2t] + W2 x[2t+1] + W3 x[2t+2]  =  sum over taps j = -1, 0, +1 of pair rows t + j with
    //   even half: (0, W1, W3), odd half: (W0, W2, 0).
    const int version = c.reserved[1], ulaw = c.reserved[2] ? 1 : 0;
    const int n_fft = version == 2 ? 400 : 2 * MFCC_HOP, n_mels = version == 2 ? 80 : 40, n_freqs = n_fft / 2 + 1, mid = 12 * base;
    const int OC = c.out_channels;
    std::vector<double> tw(2 * (size_t)n_fft);
    for (int n = 0; n < n_fft; ++n) {
      tw[2 * n] = std::cos(2.0 * M_PI * n / n_fft);
      tw[2 * n + 1] = -std::sin(2.0 * M_PI * n / n_fft);
    }
    const size_t tw_off = b.blob.add(tw.data(), tw.size() * 8);
    const size_t win_off = b.blob_f32("mfcc.MelSpectrogram.spectrogram.window"), fb_off = b.blob_f32("mfcc.MelSpectrogram.mel_scale.fb");
    const size_t dct_off = b.blob_f32("mfcc.dct_mat");
    double wss = 0.0;
    for (int n = 0; n < n_fft; ++n) wss += (double)b.P("mfcc.MelSpectrogram.spectrogram.window")[n] * (double)b.P("mfcc.MelSpectrogram.spectrogram.window")[n];
    const double power_scale = version == 2 ? 1.0 / wss : 1.0;  // Spectrogram(normalized=True): spec / sqrt(sum w^2), then |.|^2
    const int max_frames = c.max_T / MFCC_HOP + 1;
    const size_t logmel_off = b.alloc_misc((size_t)c.max_batch * max_frames * n_mels);
    const size_t wgmax_off = b.alloc_misc((size_t)c.max_batch * mfcc_groups(max_frames) + 64);
    const bool db = version == 2;
    TensorH feat = b.new_tensor(64, LEN_FRAMES, false, false);  // fp32 handle: the activation type is float
    m->meta.push_back({"mfcc_logmel", "n_fft=" + std::to_string(n_fft) + " mels=" + std::to_string(n_mels), 0, 4.0, 0});
    m->add_op([=](const RunCtx& r) -> int {
      if (r.logmel != nullptr) {  // testing entry (vqvs_mfcc_encoder_forward_logmel): injected log-mel rows, the transform is skipped
        if (db) VQVS_FAIL(-1, "the injected log-mel entry exists for the log_mels (version 1) front end only");
        VQVS_HIP(hipMemcpyAsync(bp->miscp(logmel_off), r.logmel, (size_t)r.B * shiftL(r.Lbase, LEN_FRAMES) * n_mels * sizeof(float),
                                hipMemcpyDeviceToDevice, r.st));
        return 0;
      }
      MfccArgs a{};
      a.x = r.x;
      a.twiddle = reinterpret_cast<const double*>(bp->wp(tw_off));
      a.window = reinterpret_cast<const float*>(bp->wp(win_off));
      a.fb = reinterpret_cast<const float*>(bp->wp(fb_off));
      a.logmel = bp->miscp(logmel_off);
      a.wgmax = db ? bp->miscp(wgmax_off) + 64 : nullptr;
      a.T = r.Lbase;
      a.n_fft = n_fft;
      a.hop = MFCC_HOP;
      a.n_freqs = n_freqs;
      a.n_mels = n_mels;
      a.frames = shiftL(r.Lbase, LEN_FRAMES);
      a.ulaw = ulaw;
      a.log_mels = db ? 0 : 1;
      a.power_scale = power_scale;
      if (int e = launch_mfcc_logmel(a, r.B, r.st)) return e;
      if (db) return launch_mfcc_batch_max(bp->miscp(wgmax_off) + 64, r.B * mfcc_groups(a.frames), bp->miscp(wgmax_off), r.st);
      return 0;
    });
    m->meta.push_back({"mfcc_features", "13 x 3 -> 64 channels", 64.0 / MFCC_HOP, 0, 0});
    m->add_op([=](const RunCtx& r) -> int {
      MfccFeatArgs a{};
      a.logmel = bp->miscp(logmel_off);
      a.dct = reinterpret_cast<const float*>(bp->wp(dct_off));
      a.batch_max = db ? bp->miscp(wgmax_off) : nullptr;
      a.feat = reinterpret_cast<float*>(bp->act(feat.off));
      a.frames = shiftL(r.Lbase, LEN_FRAMES);
      a.rows_alloc = a.frames;
      a.n_mels = n_mels;
      return launch_mfcc_features(a, r.B, r.st);
    });
    b.tap("features", feat);
    auto conv_layer = [&](const TensorH& in, int in_C, int in_lshift, const std::vector<float>& W, int cout, int cin, int ktaps,
                          const float* bias, const TensorH& out, int out_lshift, const TensorH* skip, bool gelu) {
      TensorH view = in;
      view.C = in_C;
      view.lshift = in_lshift;
      PackedConv pk(prec);
      Builder::SegSpec g{view, 0, in_C, ktaps, 1, RESIZE_NONE, false, 0, 0, 0, 0};
      g.w_off = pk.append(W.data(), cout, cin, ktaps, 0, in_C);
      b.add_conv({g}, pk, std::vector<float>(bias, bias + cout), cout, out, skip, RESIZE_NONE, out_lshift, gelu);
    };
    auto wvec = [&](const std::string& name, size_t n) { return std::vector<float>(b.P(name), b.P(name) + n); };
    // blocks.0: Conv1d(39 -> mid, 3) + GELU over the zero-padded 64-channel feature rows
    TensorH h0 = b.new_tensor(mid, LEN_FRAMES, false, false);
    {
      std::vector<float> W((size_t)mid * 64 * 3, 0.f);
      const float* w = b.P("blocks.0.0.weight");
      for (int co = 0; co < mid; ++co)
        for (int ci = 0; ci < 39; ++ci)
          for (int k = 0; k < 3; ++k) W[((size_t)co * 64 + ci) * 3 + k] = w[((size_t)co * 39 + ci) * 3 + k];
      conv_layer(feat, 64, LEN_FRAMES, W, mid, 64, 3, b.P("blocks.0.0.bias"), h0, LEN_FRAMES, nullptr, true);
    }
    b.release(feat);
    b.tap("blocks.0", h0);
    // blocks.1: ResConv(3), written into an allocation with an even number of rows (last row zero)
    TensorH h1 = b.new_tensor(mid, LEN_FRAMES_PAD, false, false);
    {
      const TensorH hp = h1;
      m->meta.push_back({"pad_row", "", 0, 0, 0});
      m->add_op([=](const RunCtx& r) -> int {
        const int frames = shiftL(r.Lbase, LEN_FRAMES), rows = shiftL(r.Lbase, LEN_FRAMES_PAD);
        if (rows == frames) return 0;
        VQVS_HIP(hipMemset2DAsync(bp->act(hp.off) + (size_t)frames * mid * 4, (size_t)rows * mid * 4, 0, (size_t)mid * 4, r.B, r.st));
        return 0;
      });
    }
    conv_layer(h0, mid, LEN_FRAMES, wvec("blocks.1.conv.weight", (size_t)mid * mid * 3), mid, mid, 3, b.P("blocks.1.conv.bias"), h1,
               LEN_FRAMES, &h0, true);
    b.release(h0);
    // blocks.2: Conv1d(mid -> mid, 4, stride 2, padding 1) + GELU as a 3-tap convolution over [pairs][2 * mid]
    TensorH h2 = b.new_tensor(mid, LEN_HALF, false, false);
    {
      std::vector<float> W((size_t)mid * 2 * mid * 3, 0.f);
      const float* w = b.P("blocks.2.0.weight");  // [mid][mid][4]
      for (int co = 0; co < mid; ++co)
        for (int ci = 0; ci < mid; ++ci) {
          const float* wk = w + ((size_t)co * mid + ci) * 4;
          float* even = &W[((size_t)co * 2 * mid + ci) * 3];
          float* odd = &W[((size_t)co * 2 * mid + mid + ci) * 3];
          even[1] = wk[1];  // x[2t]
          even[2] = wk[3];  // x[2t+2] = even half of pair row t+1
          odd[0] = wk[0];   // x[2t-1] = odd half of pair row t-1
          odd[1] = wk[2];   // x[2t+1]
        }
      conv_layer(h1, 2 * mid, LEN_PAIRS, W, mid, 2 * mid, 3, b.P("blocks.2.0.bias"), h2, LEN_HALF, nullptr, true);
    }
    b.release(h1);
    b.tap("blocks.2", h2);
    TensorH h = h2;
    for (int i = 3; i <= 8; ++i) {  // ResConv(3) x 2, ResConv(1) x 4
      const int k = i <= 4 ? 3 : 1;
      TensorH o = b.new_tensor(mid, LEN_HALF, false, false);
      const std::string nm = "blocks." + std::to_string(i) + ".conv";
      conv_layer(h, mid, LEN_HALF, wvec(nm + ".weight", (size_t)mid * mid * k), mid, mid, k, b.P(nm + ".bias"), o, LEN_HALF, &h, true);
      b.release(h);
      h = o;
    }
    b.tap("blocks.8", h);
    TensorH o = b.new_tensor(OC, LEN_HALF, true, false);
    conv_layer(h, mid, LEN_HALF, wvec("blocks.9.weight", (size_t)OC * mid), OC, mid, 1, b.P("blocks.9.bias"), o, LEN_HALF, nullptr, false);
    b.release(h);
    m->meta.push_back({"ntc_to_nct", "", 0, 0, 0});
    m->add_op([=](const RunCtx& r) -> int { return launch_ntc_to_nct(bp->act(o.off), r.out, r.B, OC, shiftL(r.Lbase, LEN_HALF), 0, r.st); });
  } else {  // encoder (unet.py:229-241)
    std::vector<BlockSpec> blocks;
    const int cur = encoder_blocks(base, topo_of(c), blocks);
    const int out_ls = topo_of(c).levels() - 1;  // the output has T / downsample_rate rows
    TensorH h = b.new_tensor(base, 0, false, true);
    {
      const size_t w = b.blob_f32("in_conv.weight"), bi = b.blob_f32("in_conv.bias");
      m->meta.push_back({"in_conv", "1->" + std::to_string(base), (double)base, 4.0, 0});
      m->add_op([=](const RunCtx& r) -> int {
        InConvArgs a{};
        a.x = r.x;
        a.Cin = in_ch;
        a.w = reinterpret_cast<const float*>(bp->wp(w));
        a.bias = reinterpret_cast<const float*>(bp->wp(bi));
        a.condp = nullptr;
        a.cond_len = 0;
        a.out = bp->act(h.off);
        a.stats = bp->statp(h.stats_off);
        a.C = base;
        a.T = r.Lbase;
        a.ntiles = ntiles_of(r.Lbase);
        return launch_in_conv(a, r.B, prec, r.st);
      });
    }
    b.tap("in_conv", h);
    for (auto& s : blocks) {
      TensorH o = b.resblock(s.prefix, s, {h}, false, 0, 0, 0);
      b.release(h);
      h = o;
      b.tap(s.prefix, h);
    }
    const size_t ss = b.alloc_ss(cur);
    b.add_gn({h}, "out.0.0", false, 0, 0, 0, ss);
    TensorH o = b.new_tensor(c.out_channels, out_ls, true, false);
    PackedConv pk(prec);
    Builder::SegSpec g{h, 0, cur, 3, 1, RESIZE_NONE, true, ss, cur, 0, 0};
    g.w_off = pk.append(b.P("out.1.weight"), c.out_channels, cur, 3, 0, cur);
    const float* bb = b.P("out.1.bias");
    b.add_conv({g}, pk, std::vector<float>(bb, bb + c.out_channels), c.out_channels, o, nullptr, 0);
    const int OC = c.out_channels;
    m->meta.push_back({"ntc_to_nct", "", 0, 0, 0});
    m->add_op([=](const RunCtx& r) -> int { return launch_ntc_to_nct(bp->act(o.off), r.out, r.B, OC, r.Lbase >> out_ls, 0, r.st); });
  }

  for (auto& mt : m->meta) {
    if (mt.kind == "xform") continue;  // overhead traffic, not part of the algorithmic (Model A) bytes
    m->cost.elems_T += mt.elems_T;
    m->cost.bytes_f32 += mt.bytes_f32;
    m->cost.flops += mt.flops;
  }
  // ---- lay out the arena and upload the weights ------------------------------------------
  m->act_bytes = (b.act_high + 255) & ~(size_t)255;
  m->stats_off = m->act_bytes;
  m->stats_floats = b.stats_floats;
  m->ss_off = (m->stats_off + b.stats_floats * 4 + 255) & ~(size_t)255;
  m->ss_floats = b.ss_floats;
  m->misc_off = (m->ss_off + b.ss_floats * 4 + 255) & ~(size_t)255;
  m->misc_floats = b.misc_floats;
  m->arena_bytes = m->misc_off + b.misc_floats * 4 + 256;
  m->weights_bytes = b.blob.data.size() + 256;
  VQVS_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_weights), m->weights_bytes));
  VQVS_HIP(hipMemcpy(m->d_weights, b.blob.data.data(), b.blob.data.size(), hipMemcpyHostToDevice));
  VQVS_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_arena), m->arena_bytes));
  VQVS_HIP(hipMemset(m->d_arena, 0, m->arena_bytes));
  b.blob.data.clear();
  b.blob.data.shrink_to_fit();
  return 0;
}

int run_model(vqvs_model* m, const RunCtx& ctx) {
  if (m->profiling) {
    if (m->events.size() != m->ops.size() + 1) {
      for (auto e : m->events) (void)hipEventDestroy(e);
      m->events.assign(m->ops.size() + 1, nullptr);
      for (auto& e : m->events) VQVS_HIP(hipEventCreate(&e));
    }
    VQVS_HIP(hipEventRecord(m->events[0], ctx.st));
    for (size_t i = 0; i < m->ops.size(); ++i) {
      if (m->op_phase[i] == 0 || ctx.backward)
        if (int e = m->ops[i](ctx)) return e;
      VQVS_HIP(hipEventRecord(m->events[i + 1], ctx.st));
    }
    m->last_B = ctx.B;
    m->last_L = ctx.Lbase;
    return 0;
  }
  for (size_t i = 0; i < m->ops.size(); ++i)
    if (m->op_phase[i] == 0 || ctx.backward)
      if (int e = m->ops[i](ctx)) return e;
  m->last_B = ctx.B;
  m->last_L = ctx.Lbase;
  return 0;
}

}  // namespace vqvs
