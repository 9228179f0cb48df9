// Host-side network description: topology -> parameter table -> packed device weights ->
// static launch schedule over a liveness-planned scratch arena.  Nothing here runs per
// element; it decides what the kernels in conv_mfma.hip / misc_kernels.hip are pointed at.
#pragma once

#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/vqvs.h"
#include "kernels.hpp"

namespace vqvs {

struct ParamDef {
  std::string name;
  std::vector<int64_t> shape;
  size_t numel() const {
    size_t n = 1;
    for (auto s : shape) n *= (size_t)s;
    return n;
  }
};

// Activation tensor in the arena: [max_batch][L][C], L = base length shifted by lshift.
struct TensorH {
  int id = -1;
  size_t off = 0;        // byte offset in the arena
  int C = 0;
  int lshift = 0;        // L = lshift >= 0 ? Lbase >> lshift : Lbase << -lshift
  bool f32 = false;      // forced float32 storage (boundary tensors)
  size_t stats_off = 0;  // float offset in the statistics region, valid if has_stats
  bool has_stats = false;
};

struct RunCtx {
  int B = 0;
  int Lbase = 0;  // T for UNets, L of the input for a resblock handle
  const float* x = nullptr;
  const float* ts = nullptr;
  const float* cond = nullptr;
  const int64_t* labels = nullptr;
  const float* emb = nullptr;  // resblock handle: external embedding
  const float* logmel = nullptr;  // MFCC encoder handle, testing entry: [B][frames][n_mels] log-mel rows replacing the front end's
  float* out = nullptr;
  // classifier handles: logits [B][num_labels]; with backward, grad_out [B][T] = gscale * d log p(labels) / dx
  bool backward = false;
  float gscale = 1.0f;
  float* grad_out = nullptr;
  hipStream_t st = nullptr;
};

struct TapDef {
  std::string name;
  TensorH t;
};

}  // namespace vqvs

struct vqvs_model {
  vqvs_cfg cfg{};
  int device = 0;
  std::vector<vqvs::ParamDef> params;
  // device memory
  char* d_weights = nullptr;
  size_t weights_bytes = 0;
  char* d_arena = nullptr;
  size_t arena_bytes = 0;
  // arena regions
  size_t act_bytes = 0, stats_off = 0, stats_floats = 0, ss_off = 0, ss_floats = 0, misc_off = 0, misc_floats = 0;
  std::vector<std::function<int(const vqvs::RunCtx&)>> ops;
  struct OpMeta {
    std::string kind;      // "conv", "gn_prepare", "in_conv", ...
    std::string desc;      // human-readable shape (profiling tables)
    double elems_T = 0;    // algorithmic activation elements (storage type) per clip per unit of base length
    double bytes_f32 = 0;  // float32 boundary bytes per clip per unit of base length
    double flops = 0;      // per clip per unit of base length
  };
  std::vector<OpMeta> meta;  // parallel to ops
  // ops of phase 0 always run; phase 1 ops (the backward schedule of a classifier handle) only when asked for
  std::vector<uint8_t> op_phase;
  int cur_phase = 0;
  void add_op(std::function<int(const vqvs::RunCtx&)> fn) {
    ops.push_back(std::move(fn));
    op_phase.push_back((uint8_t)cur_phase);
  }
  bool profiling = false;
  std::vector<hipEvent_t> events;  // ops.size() + 1 when profiling
  std::vector<vqvs::TapDef> taps;
  // accounting (per clip, per unit of base length): elements moved / flops, as (coefficient, lshift) lists
  struct Cost {
    double elems_T = 0;    // activation elements of type T moved per clip at Lbase = 1 (scaled by L)
    double bytes_f32 = 0;  // boundary float32 bytes per clip per unit length
    double flops = 0;      // per clip per unit length
  } cost;
  int last_B = 0, last_L = 0;
  size_t status_misc_off = (size_t)-1;  // device status word (misc region, floats) or -1: the handle has no GroupNorm
  size_t emb_misc_off = 0;  // conditioning vector [max_batch][emb_E] of the last forward (misc region, floats); emb_E = 0: none
  int emb_E = 0;
  std::shared_ptr<void> keepalive;  // schedule builder (resolves arena/weight offsets for the ops)
};

namespace vqvs {
int enumerate_params(const vqvs_cfg& cfg, std::vector<ParamDef>& out);
int build_model(vqvs_model* m, const float* const* h_params);
int run_model(vqvs_model* m, const RunCtx& ctx);
int gn_groups(int ch);
int real_base(const vqvs_cfg& c);  // the reference width of a padded predictor / encoder handle (vqvs_cfg.reserved[4]), else base_channels
int padded_base(int real);
int unet_rate(const vqvs_cfg& cfg);  // downsample rate of a predictor / encoder handle's topology (256 for the reference's default)
int tensor_rows(int Lbase, int lshift);  // rows per clip of a tensor with length code `lshift` at base length Lbase
}  // namespace vqvs
