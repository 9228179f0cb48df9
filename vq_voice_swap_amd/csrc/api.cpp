// extern "C" surface of libvqvs_hip.so (declared and documented in include/vqvs.h).
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "net.hpp"
#include "sampler_kernels.hpp"

namespace vqvs {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

// scratch of the handle-less entry points (DDPM step partial sums, VQ code norms, discarded logits): one buffer per
// (device, stream), so calls issued on different streams never share it -- work on ONE stream is ordered by the stream itself.
// A buffer only ever grows.  An entry point holds a LEASE on its buffer (ScratchLease, an in-flight count under the mutex) from
// scratch_get until its kernels are ENQUEUED: a buffer is only freed -- by a grow on the same stream or by the eviction of the
// least recently used buffer once a device has more than MAX_STREAMS of them -- when no lease is out on it and after the owning
// device (the current one: the map is keyed by hipGetDevice) was synchronised, so neither a thread that has the pointer but has
// not launched yet nor a queued kernel can still be using it.  A leased buffer that has to grow is retired instead and freed by
// the first later call that finds the entry without leases.
struct DeviceScratch {
  void* p = nullptr;
  size_t bytes = 0;
  unsigned long long used = 0;
  int leases = 0;
  std::vector<void*> retired;
};
static std::map<std::pair<int, void*>, DeviceScratch> g_scratch;
static std::mutex g_scratch_mu;
static unsigned long long g_scratch_tick = 0;
struct ScratchLease {
  DeviceScratch* sc = nullptr;
  void* p = nullptr;
  ScratchLease() = default;
  ScratchLease(const ScratchLease&) = delete;
  ScratchLease& operator=(const ScratchLease&) = delete;
  ~ScratchLease() {
    if (!sc) return;
    std::lock_guard<std::mutex> lk(g_scratch_mu);  // (std::map nodes are stable: the entry cannot have moved, and a leased one is never erased)
    --sc->leases;
  }
};
static int scratch_get(size_t bytes, void* stream, ScratchLease& lease) {
  constexpr size_t MIN_BYTES = (size_t)1 << 20;
  constexpr size_t MAX_STREAMS = 32;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  int dev = 0;
  VQVS_HIP(hipGetDevice(&dev));
  const auto key = std::make_pair(dev, stream);
  if (g_scratch.find(key) == g_scratch.end()) {
    size_t n_dev = 0;
    auto lru = g_scratch.end();
    for (auto it = g_scratch.begin(); it != g_scratch.end(); ++it)
      if (it->first.first == dev) {
        ++n_dev;
        if (it->second.leases == 0 && (lru == g_scratch.end() || it->second.used < lru->second.used)) lru = it;
      }
    if (n_dev >= MAX_STREAMS && lru != g_scratch.end()) {  // (streams come and go: their buffers are reclaimed here, oldest idle one first;
      VQVS_HIP(hipDeviceSynchronize());                     //  with every buffer leased the cap is exceeded for the moment)
      if (lru->second.p) VQVS_HIP(hipFree(lru->second.p));
      for (void* q : lru->second.retired) VQVS_HIP(hipFree(q));
      g_scratch.erase(lru);
    }
  }
  DeviceScratch& sc = g_scratch[key];
  if (sc.leases == 0 && (!sc.retired.empty() || (sc.p && sc.bytes < bytes))) {
    VQVS_HIP(hipDeviceSynchronize());
    for (void* q : sc.retired) VQVS_HIP(hipFree(q));
    sc.retired.clear();
    if (sc.p && sc.bytes < bytes) {
      VQVS_HIP(hipFree(sc.p));
      sc.p = nullptr;
      sc.bytes = 0;
    }
  } else if (sc.p && sc.bytes < bytes) {  // (another thread on the same stream still holds the smaller buffer: retire it, free it later)
    sc.retired.push_back(sc.p);
    sc.p = nullptr;
    sc.bytes = 0;
  }
  if (!sc.p) {
    size_t n = bytes < MIN_BYTES ? MIN_BYTES : bytes;
    VQVS_HIP(hipMalloc(&sc.p, n));
    sc.bytes = n;
  }
  sc.used = ++g_scratch_tick;
  ++sc.leases;
  lease.sc = &sc;
  lease.p = sc.p;
  return 0;
}
}  // namespace vqvs

using namespace vqvs;

extern "C" {

const char* vqvs_last_error(void) { return g_err.c_str(); }
#ifndef VQVS_BUILD_ID
#define VQVS_BUILD_ID "unstamped"
#endif
const char* vqvs_version(void) { return "vqvs-hip 0.1 (gfx950) build " VQVS_BUILD_ID; }  // (build id: hash of the sources, csrc/Makefile)

int vqvs_param_count(const vqvs_cfg* cfg) {
  if (!cfg) VQVS_FAIL(VQVS_ERR_ARG, "cfg is NULL");
  std::vector<ParamDef> p;
  if (int e = enumerate_params(*cfg, p)) return e;
  return (int)p.size();
}

int vqvs_param_info(const vqvs_cfg* cfg, int index, char* name_out, int name_cap, int64_t shape_out[4], int* ndim_out) {
  if (!cfg) VQVS_FAIL(VQVS_ERR_ARG, "cfg is NULL");
  std::vector<ParamDef> p;
  if (int e = enumerate_params(*cfg, p)) return e;
  if (index < 0 || index >= (int)p.size()) VQVS_FAIL(VQVS_ERR_ARG, "param index %d out of range", index);
  const ParamDef& d = p[index];
  if (name_out && name_cap > 0) {
    strncpy(name_out, d.name.c_str(), name_cap - 1);
    name_out[name_cap - 1] = 0;
  }
  if (shape_out)
    for (int i = 0; i < 4; ++i) shape_out[i] = i < (int)d.shape.size() ? d.shape[i] : 1;
  if (ndim_out) *ndim_out = (int)d.shape.size();
  return 0;
}

int vqvs_model_create(const vqvs_cfg* cfg, const float* const* h_params, int n_params, int device, vqvs_model** out) {
  if (!cfg || !h_params || !out) VQVS_FAIL(VQVS_ERR_ARG, "NULL argument");
  *out = nullptr;
  std::vector<ParamDef> p;
  if (int e = enumerate_params(*cfg, p)) return e;
  if (n_params != (int)p.size()) VQVS_FAIL(VQVS_ERR_ARG, "expected %d parameters, got %d", (int)p.size(), n_params);
  for (int i = 0; i < n_params; ++i)
    if (!h_params[i]) VQVS_FAIL(VQVS_ERR_ARG, "parameter %d (%s) is NULL", i, p[i].name.c_str());
  int ndev = 0;
  VQVS_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) VQVS_FAIL(VQVS_ERR_ARG, "device %d not present (%d visible)", device, ndev);
  VQVS_HIP(hipSetDevice(device));
  vqvs_model* m = new vqvs_model();
  m->cfg = *cfg;
  m->device = device;
  if (int e = build_model(m, h_params)) {
    vqvs_model_destroy(m);
    return e;
  }
  *out = m;
  return 0;
}

void vqvs_model_destroy(vqvs_model* m) {
  if (!m) return;
  for (auto e : m->events) (void)hipEventDestroy(e);
  if (m->d_weights) (void)hipFree(m->d_weights);
  if (m->d_arena) (void)hipFree(m->d_arena);
  delete m;
}

int64_t vqvs_model_device_bytes(const vqvs_model* m) { return m ? (int64_t)(m->weights_bytes + m->arena_bytes) : 0; }

static int check_run(vqvs_model* m, int kind, int B, int L) {
  if (!m) VQVS_FAIL(VQVS_ERR_ARG, "model is NULL");
  if (m->cfg.kind != kind) VQVS_FAIL(VQVS_ERR_STATE, "handle kind %d used as kind %d", m->cfg.kind, kind);
  if (B < 1 || B > m->cfg.max_batch) VQVS_FAIL(VQVS_ERR_ARG, "batch %d outside 1..%d", B, m->cfg.max_batch);
  if (L < 1 || L > m->cfg.max_T) VQVS_FAIL(VQVS_ERR_ARG, "length %d outside 1..%d", L, m->cfg.max_T);
  if (kind == VQVS_KIND_CLASSIFIER && (L % (2 * unet_rate(m->cfg))))  // (the stem halves the length after EVERY level: 2^levels, classifier.py:79-96)
    VQVS_FAIL(VQVS_ERR_ARG, "T=%d is not a multiple of the classifier downsample rate %d", L, 2 * unet_rate(m->cfg));
  if (kind != VQVS_KIND_RESBLOCK && kind != VQVS_KIND_MFCC_ENCODER && (L % unet_rate(m->cfg)))
    VQVS_FAIL(VQVS_ERR_ARG, "T=%d is not a multiple of the UNet downsample rate %d", L, unet_rate(m->cfg));
  if (kind == VQVS_KIND_RESBLOCK && m->cfg.rb_resize == RESIZE_AVG2 && (L % 2)) VQVS_FAIL(VQVS_ERR_ARG, "avg-pool resblock needs even L");
  return 0;
}

int vqvs_unet_forward(vqvs_model* m, const float* d_x, const float* d_ts, const float* d_cond, const int64_t* d_labels, float* d_out,
                      int B, int T, void* stream) {
  if (int e = check_run(m, VQVS_KIND_PREDICTOR, B, T)) return e;
  if (!d_x || !d_ts || !d_out) VQVS_FAIL(VQVS_ERR_ARG, "x, ts and out must be non-NULL");
  // reference unet.py:126-131
  if ((d_labels == nullptr) != (m->cfg.num_labels == 0)) VQVS_FAIL(VQVS_ERR_ARG, "must provide labels if and only if model is class conditional");
  if ((d_cond == nullptr) != (m->cfg.cond_channels == 0)) VQVS_FAIL(VQVS_ERR_ARG, "must provide cond sequence if and only if model is conditional");
  RunCtx c;
  c.B = B;
  c.Lbase = T;
  c.x = d_x;
  c.ts = d_ts;
  c.cond = d_cond;
  c.labels = d_labels;
  c.out = d_out;
  c.st = reinterpret_cast<hipStream_t>(stream);
  return run_model(m, c);
}

int vqvs_encoder_forward(vqvs_model* m, const float* d_x, float* d_z, int B, int T, void* stream) {
  if (int e = check_run(m, VQVS_KIND_ENCODER, B, T)) return e;
  if (!d_x || !d_z) VQVS_FAIL(VQVS_ERR_ARG, "x and z must be non-NULL");
  RunCtx c;
  c.B = B;
  c.Lbase = T;
  c.x = d_x;
  c.out = d_z;
  c.st = reinterpret_cast<hipStream_t>(stream);
  return run_model(m, c);
}

int vqvs_mfcc_encoder_forward(vqvs_model* m, const float* d_x, float* d_z, int B, int T, void* stream) {
  if (int e = check_run(m, VQVS_KIND_MFCC_ENCODER, B, T)) return e;
  if (!d_x || !d_z) VQVS_FAIL(VQVS_ERR_ARG, "x and z must be non-NULL");
  if (T < 800) VQVS_FAIL(VQVS_ERR_ARG, "T=%d is too short for the MFCC front end (needs at least 800 samples)", T);
  RunCtx c;
  c.B = B;
  c.Lbase = T;
  c.x = d_x;
  c.out = d_z;
  c.st = reinterpret_cast<hipStream_t>(stream);
  return run_model(m, c);
}

int vqvs_mfcc_encoder_forward_logmel(vqvs_model* m, const float* d_logmel, float* d_z, int B, int T, void* stream) {
  if (int e = check_run(m, VQVS_KIND_MFCC_ENCODER, B, T)) return e;
  if (!d_logmel || !d_z) VQVS_FAIL(VQVS_ERR_ARG, "logmel and z must be non-NULL");
  if (T < 800) VQVS_FAIL(VQVS_ERR_ARG, "T=%d is too short for the MFCC front end (needs at least 800 samples)", T);
  RunCtx c;
  c.B = B;
  c.Lbase = T;
  c.logmel = d_logmel;
  c.out = d_z;
  c.st = reinterpret_cast<hipStream_t>(stream);
  return run_model(m, c);
}

int vqvs_resblock_forward(vqvs_model* m, const float* d_x, const float* d_emb, float* d_y, int B, int L, void* stream) {
  if (int e = check_run(m, VQVS_KIND_RESBLOCK, B, L)) return e;
  if (!d_x || !d_y) VQVS_FAIL(VQVS_ERR_ARG, "x and y must be non-NULL");
  if ((d_emb == nullptr) != (m->cfg.rb_emb_channels == 0)) VQVS_FAIL(VQVS_ERR_ARG, "emb must be given iff the block has cond_layers");
  RunCtx c;
  c.B = B;
  c.Lbase = L;
  c.x = d_x;
  c.emb = d_emb;
  c.out = d_y;
  c.st = reinterpret_cast<hipStream_t>(stream);
  return run_model(m, c);
}

int vqvs_classifier_forward(vqvs_model* m, const float* d_x, const float* d_ts, float* d_logits, int B, int T, void* stream) {
  if (int e = check_run(m, VQVS_KIND_CLASSIFIER, B, T)) return e;
  if (!d_x || !d_ts || !d_logits) VQVS_FAIL(VQVS_ERR_ARG, "x, ts and logits must be non-NULL");
  RunCtx c;
  c.B = B;
  c.Lbase = T;
  c.x = d_x;
  c.ts = d_ts;
  c.out = d_logits;
  c.st = reinterpret_cast<hipStream_t>(stream);
  return run_model(m, c);
}

int vqvs_classifier_guidance(vqvs_model* m, const float* d_x, const float* d_ts, const int64_t* d_labels, float scale, float* d_grad,
                             float* d_logits, int B, int T, void* stream) {
  if (int e = check_run(m, VQVS_KIND_CLASSIFIER, B, T)) return e;
  if (!d_x || !d_ts || !d_labels || !d_grad) VQVS_FAIL(VQVS_ERR_ARG, "x, ts, labels and grad must be non-NULL");
  float* logits = d_logits;
  ScratchLease lease;  // (held until run_model has enqueued every kernel)
  if (!logits) {
    if (int e = scratch_get((size_t)B * m->cfg.num_labels * 4, stream, lease)) return e;
    logits = reinterpret_cast<float*>(lease.p);
  }
  RunCtx c;
  c.B = B;
  c.Lbase = T;
  c.x = d_x;
  c.ts = d_ts;
  c.labels = d_labels;
  c.out = logits;
  c.backward = true;
  c.gscale = scale;
  c.grad_out = d_grad;
  c.st = reinterpret_cast<hipStream_t>(stream);
  return run_model(m, c);
}

int vqvs_encpred_forward(vqvs_model* m, const float* d_x, const float* d_ts, float* d_logits, int B, int T, void* stream) {
  if (int e = check_run(m, VQVS_KIND_ENCPRED, B, T)) return e;
  if (!d_x || !d_ts || !d_logits) VQVS_FAIL(VQVS_ERR_ARG, "x, ts and logits must be non-NULL");
  if (T % m->cfg.reserved[1]) VQVS_FAIL(VQVS_ERR_ARG, "T=%d is not a multiple of the downsample rate %d", T, m->cfg.reserved[1]);
  RunCtx c;
  c.B = B;
  c.Lbase = T;
  c.x = d_x;
  c.ts = d_ts;
  c.out = d_logits;
  c.st = reinterpret_cast<hipStream_t>(stream);
  return run_model(m, c);
}

int vqvs_encpred_guidance(vqvs_model* m, const float* d_x, const float* d_ts, const int64_t* d_targets, float scale, float* d_grad,
                          float* d_logits, int B, int T, void* stream) {
  if (int e = check_run(m, VQVS_KIND_ENCPRED, B, T)) return e;
  if (!d_x || !d_ts || !d_targets || !d_grad) VQVS_FAIL(VQVS_ERR_ARG, "x, ts, targets and grad must be non-NULL");
  if (T % m->cfg.reserved[1]) VQVS_FAIL(VQVS_ERR_ARG, "T=%d is not a multiple of the downsample rate %d", T, m->cfg.reserved[1]);
  RunCtx c;
  c.B = B;
  c.Lbase = T;
  c.x = d_x;
  c.ts = d_ts;
  c.labels = d_targets;  // the head reads them as targets; the UNet itself is label-free
  c.out = d_logits;
  c.backward = true;
  c.gscale = scale;
  c.grad_out = d_grad;
  c.st = reinterpret_cast<hipStream_t>(stream);
  return run_model(m, c);
}

int vqvs_ddpm_step(const float* d_x_t, const float* d_eps, const float* d_noise, const float* d_alpha_t, const float* d_alpha_prev,
                   float* d_x_prev, int B, int T, uint32_t flags, float noise_scale, uint64_t seed, uint64_t clip_offset,
                   uint32_t step_index, void* stream) {
  if (!d_x_t || !d_eps || !d_alpha_t || !d_alpha_prev || !d_x_prev) VQVS_FAIL(VQVS_ERR_ARG, "NULL argument");
  if (B < 1 || T < 1) VQVS_FAIL(VQVS_ERR_ARG, "bad shape B=%d T=%d", B, T);
  ScratchLease lease;
  if (flags & VQVS_DDPM_CONSTRAIN)
    if (int e = scratch_get((size_t)ddpm_scratch_doubles(B, T) * 8, stream, lease)) return e;
  return run_ddpm_step(d_x_t, d_eps, d_noise, d_alpha_t, d_alpha_prev, d_x_prev, reinterpret_cast<double*>(lease.p), B, T, flags,
                       noise_scale, seed, clip_offset, step_index, reinterpret_cast<hipStream_t>(stream));
}

int vqvs_ddpm_mean(const float* d_x_t, const float* d_eps, const float* d_alpha_t, const float* d_alpha_prev, float* d_mean, int B, int T,
                   void* stream) {
  if (!d_x_t || !d_eps || !d_alpha_t || !d_alpha_prev || !d_mean) VQVS_FAIL(VQVS_ERR_ARG, "NULL argument");
  return run_ddpm_mean(d_x_t, d_eps, d_alpha_t, d_alpha_prev, d_mean, B, T, reinterpret_cast<hipStream_t>(stream));
}

int vqvs_ddpm_guided_eps(const float* d_x_t, const float* d_mean, const float* d_grad, const float* d_alpha_t, const float* d_alpha_prev,
                         float* d_eps_out, int B, int T, uint32_t flags, void* stream) {
  if (!d_x_t || !d_mean || !d_grad || !d_alpha_t || !d_alpha_prev || !d_eps_out) VQVS_FAIL(VQVS_ERR_ARG, "NULL argument");
  return run_ddpm_guided_eps(d_x_t, d_mean, d_grad, d_alpha_t, d_alpha_prev, d_eps_out, B, T, flags, reinterpret_cast<hipStream_t>(stream));
}

int vqvs_randn(float* d_out, int B, int T, uint64_t seed, uint64_t clip_offset, uint32_t stream_id, void* stream) {
  if (!d_out || B < 1 || T < 1) VQVS_FAIL(VQVS_ERR_ARG, "bad argument");
  return run_randn(d_out, B, T, seed, clip_offset, stream_id, reinterpret_cast<hipStream_t>(stream));
}

int vqvs_vq_argmin(const float* d_z, const float* d_dict, int64_t* d_idx, int B, int Cd, int T1, int K, void* stream) {
  if (!d_z || !d_dict || !d_idx) VQVS_FAIL(VQVS_ERR_ARG, "NULL argument");
  if (B < 1 || Cd < 1 || T1 < 1 || K < 1) VQVS_FAIL(VQVS_ERR_ARG, "bad shape");
  if (Cd % 4) VQVS_FAIL(VQVS_ERR_ARG, "Cd must be a multiple of 4 (got %d)", Cd);
  ScratchLease lease;
  if (int e = scratch_get((size_t)K * 4, stream, lease)) return e;
  return run_vq_argmin(d_z, d_dict, reinterpret_cast<float*>(lease.p), d_idx, B, Cd, T1, K, reinterpret_cast<hipStream_t>(stream));
}

int vqvs_vq_embed(const int64_t* d_idx, const float* d_dict, float* d_out, int B, int Cd, int T1, int K, void* stream) {
  if (!d_idx || !d_dict || !d_out) VQVS_FAIL(VQVS_ERR_ARG, "NULL argument");
  return run_vq_embed(d_idx, d_dict, d_out, B, Cd, T1, K, reinterpret_cast<hipStream_t>(stream));
}

int vqvs_debug_tap_count(const vqvs_model* m) { return m ? (int)m->taps.size() : 0; }

int vqvs_debug_tap_info(const vqvs_model* m, int i, char* name_out, int name_cap, int* channels, int* length_shift) {
  if (!m || i < 0 || i >= (int)m->taps.size()) VQVS_FAIL(VQVS_ERR_ARG, "bad tap index");
  if (name_out && name_cap > 0) {
    strncpy(name_out, m->taps[i].name.c_str(), name_cap - 1);
    name_out[name_cap - 1] = 0;
  }
  if (channels) *channels = m->taps[i].t.C;
  if (length_shift) *length_shift = m->taps[i].t.lshift;
  return 0;
}

int vqvs_debug_tap_rows(const vqvs_model* m, int i, int T) {
  if (!m || i < 0 || i >= (int)m->taps.size()) VQVS_FAIL(VQVS_ERR_ARG, "bad tap index");
  return tensor_rows(T, m->taps[i].t.lshift);
}

int vqvs_debug_read_tap(vqvs_model* m, int i, int B, int T, float* h_out) {
  if (!m || i < 0 || i >= (int)m->taps.size() || !h_out) VQVS_FAIL(VQVS_ERR_ARG, "bad argument");
  if (!m->cfg.debug_taps) VQVS_FAIL(VQVS_ERR_STATE, "model was not created with debug_taps=1");
  const TensorH& t = m->taps[i].t;
  const int L = tensor_rows(T, t.lshift);
  const size_t n = (size_t)B * t.C * L;
  float* d_tmp = nullptr;
  VQVS_HIP(hipSetDevice(m->device));
  VQVS_HIP(hipMalloc(reinterpret_cast<void**>(&d_tmp), n * 4));
  int e = launch_ntc_to_nct(m->d_arena + t.off, d_tmp, B, t.C, L, t.f32 ? 0 : m->cfg.precision, nullptr);
  if (!e) {
    hipError_t he = hipMemcpy(h_out, d_tmp, n * 4, hipMemcpyDeviceToHost);
    if (he != hipSuccess) {
      set_error(std::string("hipMemcpy failed: ") + hipGetErrorString(he));
      e = VQVS_ERR_HIP;
    }
  }
  (void)hipFree(d_tmp);
  return e;
}

int vqvs_debug_read_embedding(vqvs_model* m, int B, float* h_out) {
  if (!m || !h_out || B < 1 || B > m->cfg.max_batch) VQVS_FAIL(VQVS_ERR_ARG, "bad argument");
  if (!m->emb_E) VQVS_FAIL(VQVS_ERR_STATE, "this model kind has no timestep embedding");
  VQVS_HIP(hipSetDevice(m->device));
  VQVS_HIP(hipDeviceSynchronize());
  const float* src = reinterpret_cast<const float*>(m->d_arena + m->misc_off) + m->emb_misc_off;
  VQVS_HIP(hipMemcpy(h_out, src, (size_t)B * m->emb_E * sizeof(float), hipMemcpyDeviceToHost));
  return m->emb_E;
}

int vqvs_forward_kernel_count(const vqvs_model* m) { return m ? (int)m->ops.size() : 0; }

int64_t vqvs_forward_model_bytes(const vqvs_model* m, int B, int T) {
  if (!m) return 0;
  const double es = m->cfg.precision == VQVS_PREC_F32 ? 4.0 : 2.0;  // BF16 and F16 store 2 bytes
  return (int64_t)((m->cost.elems_T * es + m->cost.bytes_f32) * (double)T * (double)B);
}

int64_t vqvs_forward_flops(const vqvs_model* m, int B, int T) {
  if (!m) return 0;
  return (int64_t)(m->cost.flops * (double)T * (double)B);
}

}  // extern "C"

// ---- profiling hooks (bench.py: live per-kernel timing with HIP events on the launch stream) ----
extern "C" {

int vqvs_set_profiling(vqvs_model* m, int on) {
  if (!m) VQVS_FAIL(VQVS_ERR_ARG, "model is NULL");
  m->profiling = on != 0;
  return 0;
}

// kind / algorithmic bytes / flops of op i for the given problem size
int vqvs_op_info(const vqvs_model* m, int i, char* kind_out, int kind_cap, int64_t* bytes_out, int64_t* flops_out, int B, int T) {
  if (!m || i < 0 || i >= (int)m->meta.size()) VQVS_FAIL(VQVS_ERR_ARG, "bad op index");
  const auto& mt = m->meta[i];
  if (kind_out && kind_cap > 0) {
    strncpy(kind_out, mt.kind.c_str(), kind_cap - 1);
    kind_out[kind_cap - 1] = 0;
  }
  const double es = m->cfg.precision == VQVS_PREC_F32 ? 4.0 : 2.0;  // BF16 and F16 store 2 bytes
  if (bytes_out) *bytes_out = (int64_t)((mt.elems_T * es + mt.bytes_f32) * (double)T * (double)B);
  if (flops_out) *flops_out = (int64_t)(mt.flops * (double)T * (double)B);
  return 0;
}

int vqvs_op_desc(const vqvs_model* m, int i, char* out, int cap) {
  if (!m || i < 0 || i >= (int)m->meta.size() || !out || cap < 1) VQVS_FAIL(VQVS_ERR_ARG, "bad argument");
  strncpy(out, m->meta[i].desc.c_str(), cap - 1);
  out[cap - 1] = 0;
  return 0;
}

// Device status word of the handle (synchronises the device; the word is cleared): bit 0 = a GroupNorm partial was not finite
// (an activation overflowed the storage type, or NaN input), bit 1 = fp16 mode, a tile's sum of squares reached 9e8 (an
// activation may be beyond 3e4 of fp16's 65504).  Callers check it once per sample, not per step.
int vqvs_model_status(vqvs_model* m, unsigned* h_status) {
  if (!m || !h_status) VQVS_FAIL(VQVS_ERR_ARG, "NULL argument");
  *h_status = 0;
  if (m->status_misc_off == (size_t)-1) return 0;
  unsigned* d = reinterpret_cast<unsigned*>(reinterpret_cast<float*>(m->d_arena + m->misc_off) + m->status_misc_off);
  VQVS_HIP(hipDeviceSynchronize());  // (forwards run on the caller's stream; a blocking copy on the null stream alone does not wait for a non-blocking one)
  VQVS_HIP(hipMemcpy(h_status, d, sizeof(unsigned), hipMemcpyDeviceToHost));
  if (*h_status) VQVS_HIP(hipMemset(d, 0, sizeof(unsigned)));
  return 0;
}

// elapsed milliseconds of every op of the LAST profiled forward (synchronises the device)
int vqvs_profile_read(vqvs_model* m, float* h_ms, int cap) {
  if (!m || !h_ms) VQVS_FAIL(VQVS_ERR_ARG, "NULL argument");
  if (m->events.size() != m->ops.size() + 1) VQVS_FAIL(VQVS_ERR_STATE, "no profiled forward has run");
  VQVS_HIP(hipEventSynchronize(m->events.back()));
  const int n = (int)m->ops.size() < cap ? (int)m->ops.size() : cap;
  for (int i = 0; i < n; ++i) VQVS_HIP(hipEventElapsedTime(&h_ms[i], m->events[i], m->events[i + 1]));
  return n;
}

}  // extern "C"

#ifdef VQVS_TIMING
namespace vqvs { int conv_timing_read(unsigned long long* out16, int reset); }
extern "C" int vqvs_debug_conv_timing(unsigned long long* h_out16, int reset) { return vqvs::conv_timing_read(h_out16, reset); }
namespace vqvs { int ws_timing_read(unsigned long long* out32, int reset); }
extern "C" int vqvs_debug_ws_timing(unsigned long long* h_out32, int reset) { return vqvs::ws_timing_read(h_out32, reset); }
#endif
