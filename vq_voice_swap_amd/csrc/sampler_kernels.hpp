// Host launchers of the model-independent kernels (DDPM step, RNG, VQ).
#pragma once
#include "common.hpp"

namespace vqvs {
int run_randn(float* out, int B, int T, uint64_t seed, uint64_t clip_offset, uint32_t stream_id, hipStream_t st);
int ddpm_scratch_doubles(int B, int T);
int run_ddpm_step(const float* x_t, const float* eps, const float* noise, const float* a_t, const float* a_prev, float* out,
                  double* scratch, int B, int T, uint32_t flags, float noise_scale, uint64_t seed, uint64_t clip_offset,
                  uint32_t step_index, hipStream_t st);
int run_ddpm_mean(const float* x_t, const float* eps, const float* a_t, const float* a_prev, float* out, int B, int T, hipStream_t st);
int run_ddpm_guided_eps(const float* x_t, const float* mean, const float* grad, const float* a_t, const float* a_prev, float* out,
                        int B, int T, uint32_t flags, hipStream_t st);
int run_vq_argmin(const float* z, const float* dict, float* en_scratch, int64_t* idx, int B, int Cd, int T1, int K, hipStream_t st);
int run_vq_embed(const int64_t* idx, const float* dict, float* out, int B, int Cd, int T1, int K, hipStream_t st);
}  // namespace vqvs
