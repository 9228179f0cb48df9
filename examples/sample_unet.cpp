// Unconditional DDPM sampling from C++ through the ABI of include/vqvs.h alone (no Python, no torch): the loop of
// Diffusion.ddpm_sample (reference vq_voice_swap/diffusion/diffusion.py:92-133, ExpSchedule schedule.py:15-31) around
// vqvs_randn / vqvs_unet_forward / vqvs_ddpm_step.  Weights come from tools/export_weights.py.
//
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/sample_unet.cpp
//       -Lvq_voice_swap_amd -lvqvs_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/vq_voice_swap_amd -o /tmp/sample_unet
//   /tmp/sample_unet weights.bin out.f32 [clips=2] [T=64000] [steps=50] [seed=1234] [precision: 0 fp32 | 1 bf16 | 2 fp16]
//
// Writes the clips as raw float32 [clips][T] (and clip 0 as 16-bit mono WAV next to it).
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vqvs.h"

#define HIPCHECK(e)                                                                   \
  do {                                                                                \
    hipError_t _e = (e);                                                              \
    if (_e != hipSuccess) {                                                           \
      fprintf(stderr, "%s failed: %s\n", #e, hipGetErrorString(_e));                  \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)
#define VQCHECK(e)                                                                    \
  do {                                                                                \
    if ((e) != 0) {                                                                   \
      fprintf(stderr, "%s failed: %s\n", #e, vqvs_last_error());                      \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

static void write_wav(const std::string& path, const float* x, int n, int rate) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return;
  const uint32_t bytes = (uint32_t)n * 2, riff = 36 + bytes, fmt_len = 16, r = (uint32_t)rate, br = r * 2;
  const uint16_t pcm = 1, ch = 1, align = 2, bits = 16;
  fwrite("RIFF", 1, 4, f); fwrite(&riff, 4, 1, f); fwrite("WAVEfmt ", 1, 8, f); fwrite(&fmt_len, 4, 1, f);
  fwrite(&pcm, 2, 1, f); fwrite(&ch, 2, 1, f); fwrite(&r, 4, 1, f); fwrite(&br, 4, 1, f); fwrite(&align, 2, 1, f); fwrite(&bits, 2, 1, f);
  fwrite("data", 1, 4, f); fwrite(&bytes, 4, 1, f);
  for (int i = 0; i < n; ++i) {
    float v = x[i] < -1.f ? -1.f : (x[i] > 1.f ? 1.f : x[i]);
    const int16_t s = (int16_t)lrintf(v * 32767.f);
    fwrite(&s, 2, 1, f);
  }
  fclose(f);
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s weights.bin out.f32 [clips] [T] [steps] [seed] [precision]\n", argv[0]);
    return 2;
  }
  const int B = argc > 3 ? atoi(argv[3]) : 2, T = argc > 4 ? atoi(argv[4]) : 64000, steps = argc > 5 ? atoi(argv[5]) : 50;
  const uint64_t seed = argc > 6 ? strtoull(argv[6], nullptr, 10) : 1234;
  const int precision = argc > 7 ? atoi(argv[7]) : VQVS_PREC_F32;

  // ---- weights, in vqvs_param_info order
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  char magic[8];
  int32_t base = 0, n = 0;
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "VQVSW1", 6) || fread(&base, 4, 1, f) != 1 || fread(&n, 4, 1, f) != 1) {
    fprintf(stderr, "not a VQVSW1 file\n");
    return 1;
  }
  vqvs_cfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.kind = VQVS_KIND_PREDICTOR;
  cfg.base_channels = base;
  cfg.in_channels = 1;
  cfg.out_channels = 1;
  cfg.precision = precision;
  cfg.max_batch = B;
  cfg.max_T = T;
  if (vqvs_param_count(&cfg) != n) { fprintf(stderr, "parameter count mismatch: file %d, library %d\n", n, vqvs_param_count(&cfg)); return 1; }
  std::vector<std::vector<float>> tensors(n);
  std::vector<const float*> ptrs(n);
  for (int i = 0; i < n; ++i) {
    int32_t len = 0;
    int64_t numel = 0, shape[4];
    int nd = 0;
    char want[256];
    if (fread(&len, 4, 1, f) != 1 || len <= 0 || len > 255) return 1;
    std::string name(len, 0);
    if (fread(&name[0], 1, len, f) != (size_t)len || fread(&numel, 8, 1, f) != 1) return 1;
    VQCHECK(vqvs_param_info(&cfg, i, want, sizeof(want), shape, &nd));
    int64_t expect = 1;
    for (int k = 0; k < nd; ++k) expect *= shape[k];
    if (name != want || numel != expect) { fprintf(stderr, "tensor %d: file has %s[%lld], library wants %s[%lld]\n", i, name.c_str(), (long long)numel, want, (long long)expect); return 1; }
    tensors[i].resize(numel);
    if (fread(tensors[i].data(), 4, numel, f) != (size_t)numel) return 1;
    ptrs[i] = tensors[i].data();
  }
  fclose(f);

  HIPCHECK(hipSetDevice(0));
  vqvs_model* model = nullptr;
  VQCHECK(vqvs_model_create(&cfg, ptrs.data(), n, 0, &model));
  tensors.clear();

  hipStream_t st;
  HIPCHECK(hipStreamCreate(&st));
  float *x = nullptr, *eps = nullptr, *xn = nullptr, *d_ts = nullptr, *d_at = nullptr, *d_ap = nullptr;
  const size_t nbytes = (size_t)B * T * sizeof(float);
  HIPCHECK(hipMalloc((void**)&x, nbytes)); HIPCHECK(hipMalloc((void**)&eps, nbytes)); HIPCHECK(hipMalloc((void**)&xn, nbytes));
  HIPCHECK(hipMalloc((void**)&d_ts, B * sizeof(float))); HIPCHECK(hipMalloc((void**)&d_at, B * sizeof(float))); HIPCHECK(hipMalloc((void**)&d_ap, B * sizeof(float)));

  VQCHECK(vqvs_randn(x, B, T, seed, 0, 1, st));  // x_T ~ N(0,1), keyed by (seed, clip index)
  const float k = -logf(1e-5f);                   // ExpSchedule: alpha_bar(t) = exp(-k t^2)
  std::vector<float> h_ts(B), h_at(B), h_ap(B);
  for (int i = 0; i < steps; ++i) {
    const float t = (float)((double)(steps - i) / (double)steps), tp = t - (float)(1.0 / (double)steps);
    for (int b = 0; b < B; ++b) { h_ts[b] = t; h_at[b] = expf(-k * (t * t)); h_ap[b] = expf(-k * (tp * tp)); }
    HIPCHECK(hipMemcpyAsync(d_ts, h_ts.data(), B * 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_at, h_at.data(), B * 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_ap, h_ap.data(), B * 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));  // the host vectors are reused next iteration
    VQCHECK(vqvs_unet_forward(model, x, d_ts, nullptr, nullptr, eps, B, T, st));
    const bool last = i + 1 == steps;  // zero noise on the last iteration (diffusion.py:127)
    VQCHECK(vqvs_ddpm_step(x, eps, nullptr, d_at, d_ap, xn, B, T, VQVS_DDPM_CONSTRAIN, last ? 0.f : 1.f, seed, 0, (uint32_t)i, st));
    float* tmp = x; x = xn; xn = tmp;
  }
  std::vector<float> out((size_t)B * T);
  HIPCHECK(hipMemcpyAsync(out.data(), x, nbytes, hipMemcpyDeviceToHost, st));
  HIPCHECK(hipStreamSynchronize(st));
  FILE* o = fopen(argv[2], "wb");
  if (!o) { perror(argv[2]); return 1; }
  fwrite(out.data(), 4, out.size(), o);
  fclose(o);
  write_wav(std::string(argv[2]) + ".wav", out.data(), T, 16000);
  printf("%s: %d clips x %d samples, %d steps, base_channels %d, %s\n", vqvs_version(), B, T, steps, base, precision == 0 ? "fp32" : (precision == 2 ? "fp16" : "bf16"));
  vqvs_model_destroy(model);
  return 0;
}
