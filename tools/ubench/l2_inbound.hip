// Inbound rate of ONE CU (and of the chip) by where the bytes come from and how they are fetched -- the number the deep
// (Cout >= 256) convolution levels are priced against: their tiles re-read weights and rows that sit in L2.
//   source : "shared"  every workgroup reads the same region (weights: L2 hits after the first touch)
//            "private" every workgroup re-reads its own small region (rows shared by a few channel tiles: L2 hits)
//            "stream"  every workgroup walks its own part of a 2 GiB buffer once (HBM)
//   fetch  : LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction) or register loads (buffer_load_dwordx4)
//   hipcc -O3 --offload-arch=gfx950 l2_inbound.hip -o l2_inbound && ./l2_inbound
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

// region = bytes one workgroup cycles through (power of two); stride = byte distance between workgroups' regions (0 = shared)
template <bool DMA, int DEPTH>
__global__ __launch_bounds__(1024) void k(const char* __restrict__ src, unsigned region, unsigned long long stride, int iters, int nwaves, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave >= nwaves) return;
  const char* base = src + stride * blockIdx.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)region, 0x00020000);
  u32x4 acc = {0, 0, 0, 0};
  unsigned off = (unsigned)wave * DEPTH * 1024u;
  const unsigned step = (unsigned)nwaves * DEPTH * 1024u;
  for (int it = 0; it < iters; ++it) {
    const unsigned o = off & (region - 1);
    if (DMA) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + (wave * DEPTH + j) * 1024), 16, (int)(o + j * 1024 + lane * 16), 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 v[DEPTH];
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) v[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(o + j * 1024 + lane * 16), 0, 0));
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) acc += v[j];
    }
    off += step;
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) sink[blockIdx.x * 1024 + tid] = acc[0];
}

template <bool DMA, int DEPTH>
void run(const char* name, const char* d, unsigned* sink, unsigned region, unsigned long long stride, int nwaves, int grid, double bytes_per_wg) {
  const int iters = (int)(bytes_per_wg / ((double)nwaves * DEPTH * 1024));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<DMA, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int lds = DMA ? 16 * DEPTH * 1024 : 0;
  hipLaunchKernelGGL((k<DMA, DEPTH>), dim3(grid), dim3(1024), lds, 0, d, region, stride, iters, nwaves, sink);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<DMA, DEPTH>), dim3(grid), dim3(1024), lds, 0, d, region, stride, iters, nwaves, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_wg = (double)iters * nwaves * DEPTH * 1024 * 3 / (ms * 1e-3);
  printf("  %-8s %s depth %d, %2d waves, %3d workgroups: %7.1f GB/s per CU, %6.2f TB/s chip\n", name, DMA ? "LDS-DMA " : "register", DEPTH, nwaves, grid,
         per_wg / 1e9, per_wg * grid / 1e12);
}

// reference point of DESIGN.md section 6: a plain 16-byte-per-lane copy (1 GiB -> 1 GiB), the "6.29 TB/s" of MI355X_MICROARCH.md
__global__ __launch_bounds__(256) void copy16(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
void run_copy(char* d, int blocks_per_cu) {
  const size_t n = (1ull << 30) / 16;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(copy16, dim3(256 * blocks_per_cu), dim3(256), 0, 0, (const u32x4*)d, (u32x4*)(d + (1ull << 30)), n);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(copy16, dim3(256 * blocks_per_cu), dim3(256), 0, 0, (const u32x4*)d, (u32x4*)(d + (1ull << 30)), n);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("  plain 16-byte copy, %2d blocks of 256 per CU: %6.2f TB/s (read + written)\n", blocks_per_cu, 2.0 * (double)(1ull << 30) * 5 / (ms * 1e-3) / 1e12);
}

int main() {
  char* d;
  unsigned* sink;
  const unsigned long long total = 2ull << 30;
  (void)hipMalloc(&d, total);
  (void)hipMalloc(&sink, 256 * 1024 * 4);
  (void)hipMemset(d, 1, total);
  printf("# plain copy on this box (the calibration point for every TB/s figure of this round)\n");
  for (int bpc : {4, 8, 16, 32}) run_copy(d, bpc);
  printf("# tools/ubench/l2_inbound: inbound bytes per CU by source and fetch path (one 16-wave workgroup per CU)\n");
  for (int grid : {1, 32, 256}) {
    for (int nw : {4, 8, 16}) {
      run<true, 4>("shared", d, sink, 1u << 20, 0, nw, grid, 64e6);
      run<true, 8>("shared", d, sink, 1u << 20, 0, nw, grid, 64e6);
      run<false, 4>("shared", d, sink, 1u << 20, 0, nw, grid, 64e6);
      run<false, 8>("shared", d, sink, 1u << 20, 0, nw, grid, 64e6);
    }
  }
  for (int nw : {8, 16}) {
    run<true, 8>("private", d, sink, 1u << 16, 1ull << 16, nw, 256, 64e6);
    run<false, 8>("private", d, sink, 1u << 16, 1ull << 16, nw, 256, 64e6);
    run<true, 8>("private", d, sink, 1u << 18, 1ull << 18, nw, 256, 64e6);
    run<false, 8>("private", d, sink, 1u << 18, 1ull << 18, nw, 256, 64e6);
  }
  for (int nw : {4, 8, 16}) {
    run<true, 4>("stream", d, sink, 1u << 23, 1ull << 23, nw, 256, 8e6);
    run<true, 8>("stream", d, sink, 1u << 23, 1ull << 23, nw, 256, 8e6);
    run<false, 4>("stream", d, sink, 1u << 23, 1ull << 23, nw, 256, 8e6);
    run<false, 8>("stream", d, sink, 1u << 23, 1ull << 23, nw, 256, 8e6);
  }
  return 0;
}
