// HBM read rate by access pattern (tools for DESIGN.md section 6): every byte of a [rows][C] fp16 tensor is read once, 16 B per lane.
//   pattern 0: a wave-load covers 16 rows x 64 B (one 32-channel chunk; the row's other chunks are read by later loads of the
//              same workgroup, one "step" apart) -- what a K-chunked convolution does
//   pattern 1: a wave-load covers 1 KiB of consecutive bytes (whole rows)
//   hipcc -O3 --offload-arch=gfx950 read_pattern.hip -o read_pattern && ./read_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN, int CB /* bytes per row */, int DEPTH>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, unsigned* out, long long rows_total) {
  const int tid = threadIdx.x;
  const long long tiles = rows_total / 256;
  const long long per = (tiles + gridDim.x - 1) / gridDim.x;
  const int k8 = blockIdx.x & 7, i8 = blockIdx.x >> 3;           // contiguous run of tiles per XCD, then per workgroup
  const long long xs = tiles * k8 / 8, xe = tiles * (k8 + 1) / 8;
  const int nk = gridDim.x / 8;
  const long long tb = xs + (xe - xs) * i8 / nk, te = xs + (xe - xs) * (i8 + 1) / nk;
  (void)per;
  u32x4 acc = {0, 0, 0, 0};
  constexpr int NCH = CB / 64;  // 64-byte chunks per row
  for (long long t = tb; t < te; ++t) {
    const char* base = src + t * 256 * CB;
    if (PATTERN == 0) {
      for (int c = 0; c < NCH; ++c) {
        u32x4 v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) v[i] = *reinterpret_cast<const u32x4*>(base + (long long)((tid >> 2) + 128 * i) * CB + c * 64 + (tid & 3) * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc += v[i];
      }
    } else {
      for (int c = 0; c < NCH; ++c) {
        u32x4 v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) v[i] = *reinterpret_cast<const u32x4*>(base + (long long)(c * 2 + i) * 8192 + tid * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc += v[i];
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) out[blockIdx.x * 512 + tid] = acc[0];
}

template <int PATTERN, int CB, int DEPTH>
void run(const char* d, unsigned* o, long long bytes, int wgs_per_cu) {
  const long long rows = bytes / CB;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PATTERN, CB, DEPTH>), dim3(256 * wgs_per_cu), dim3(512), 0, 0, d, o, rows);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<PATTERN, CB, DEPTH>), dim3(256 * wgs_per_cu), dim3(512), 0, 0, d, o, rows);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("  pattern %d, %3d B rows, %d workgroups of 512 per CU: %6.2f TB/s\n", PATTERN, CB, wgs_per_cu, bytes * 5 / (ms * 1e-3) / 1e12);
}

int main() {
  const long long bytes = 2048ll << 20;  // 2 GiB: far beyond the 256 MiB last-level cache
  char* d;
  unsigned* o;
  (void)hipMalloc(&d, bytes);
  (void)hipMalloc(&o, 1024 * 512 * 4);
  (void)hipMemset(d, 1, bytes);
  for (int w = 1; w <= 4; w *= 2) {
    run<0, 128, 2>(d, o, bytes, w);
    run<1, 128, 2>(d, o, bytes, w);
    run<0, 256, 2>(d, o, bytes, w);
    run<1, 256, 2>(d, o, bytes, w);
  }
  return 0;
}
