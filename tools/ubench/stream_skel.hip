// Memory skeleton of csrc/conv_ws.hip without any arithmetic: 16-wave workgroups, one per CU, persistent over 256-row tiles of a
// [rows][C] fp16 tensor.  Waves 8-15 load 32-channel chunks (16 rows x 64 B per wave-load, DEPTH chunks in flight, counted vmcnt)
// and ds_write them; waves 0-7 store the previous tile's rows (from LDS) to a second tensor; one s_barrier per chunk.
// Variants: loads only / stores only / both; stores issued by the consumer or by the loader waves.
//   hipcc -O3 --offload-arch=gfx950 stream_skel.hip -o stream_skel && ./stream_skel
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE /*1 loads, 2 stores, 3 both*/, int DEPTH, int C>
__global__ __launch_bounds__(1024) void k(const char* __restrict__ src, char* __restrict__ dst, int tiles_total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NCH = C / 32;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k8 = blockIdx.x & 7, i8 = blockIdx.x >> 3, nk = gridDim.x / 8;
  const int xs = (int)((long long)tiles_total * k8 / 8), xe = (int)((long long)tiles_total * (k8 + 1) / 8);
  const int tb = xs + (xe - xs) * i8 / nk, te = xs + (xe - xs) * (i8 + 1) / nk;
  const int Q = (te - tb) * NCH;
  if (Q <= 0) return;
  if (wave >= 8) {
    const int pt = tid - 512, oct = pt & 3, r0 = pt >> 2;
    const int dst0 = r0 * 64 + ((oct ^ ((r0 >> 2) & 3)) << 4);
    u32x4 a[DEPTH][2];
    int q_issue = 0;
    auto issue = [&](u32x4& x0, u32x4& x1) {
      const int qq = q_issue < Q ? q_issue : Q - 1;
      const int t = tb + qq / NCH, c = qq % NCH;
      const char* base = src + (long long)t * 256 * C * 2;
      i32x4 rs;
      const unsigned long long bp = reinterpret_cast<unsigned long long>(base);
      rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)bp);
      rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(bp >> 32) & 0xffffu));
      rs[2] = 256 * C * 2;
      rs[3] = 0x00020000;
      const int off0 = (r0 * C + c * 32 + oct * 8) * 2, off1 = off0 + 256 * C;
      if (MODE & 1)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %4, 0 offen\n\tbuffer_load_dwordx4 %1, %3, %4, 0 offen" : "=&v"(x0), "=&v"(x1) : "v"(off0), "v"(off1), "s"(rs));
      else
        asm volatile("" : "=v"(x0), "=v"(x1));
      ++q_issue;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(a[d][0], a[d][1]);
    int q = 0;
    for (;;) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        if (MODE & 1) {
          if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(4)" : "+v"(a[d][0]), "+v"(a[d][1]));
          if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(a[d][0]), "+v"(a[d][1]));
          if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(6)" : "+v"(a[d][0]), "+v"(a[d][1]));
        }
        *reinterpret_cast<u32x4*>(smem + (q & 1) * 16384 + dst0) = a[d][0];
        *reinterpret_cast<u32x4*>(smem + (q & 1) * 16384 + dst0 + 8192) = a[d][1];
        issue(a[d][0], a[d][1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (++q == Q) goto done;
      }
    }
  done:
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
    __builtin_amdgcn_s_barrier();
    int tile = tb;
    for (int g = 0; g < Q; ++g) {
      const int c = g % NCH;
      if ((MODE & 2) && c == NCH - 1) {  // last chunk of a tile: write the tile (rows from LDS, any bytes will do)
        char* o = dst + (long long)tile * 256 * C * 2;
        constexpr int PIECES = 256 * C * 2 / 16 / 512;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(smem + 32768 + ((i * 512 + tid) * 16) % 32768);
          *reinterpret_cast<u32x4*>(o + (long long)(i * 512 + tid) * 16) = v;
        }
        ++tile;
      } else if (c == NCH - 1) ++tile;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

template <int MODE, int DEPTH, int C>
void run(const char* s, char* d, long long bytes) {
  const int tiles = (int)(bytes / (256 * C * 2));
  const int LDS = 80 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, DEPTH, C>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, DEPTH, C>), dim3(256), dim3(1024), LDS, 0, s, d, tiles);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, DEPTH, C>), dim3(256), dim3(1024), LDS, 0, s, d, tiles);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double moved = (double)bytes * ((MODE & 1 ? 1 : 0) + (MODE & 2 ? 1 : 0)) * 5;
  printf("  C=%3d depth %d %s: %7.3f ms per pass, %5.2f TB/s\n", C, DEPTH, MODE == 1 ? "loads only " : MODE == 2 ? "stores only" : "loads+stores", ms / 5, moved / (ms * 1e-3) / 1e12);
}

int main() {
  const long long bytes = 512ll << 20;
  char *s, *d;
  (void)hipMalloc(&s, bytes);
  (void)hipMalloc(&d, bytes);
  (void)hipMemset(s, 1, bytes);
  run<1, 3, 64>(s, d, bytes);
  run<2, 3, 64>(s, d, bytes);
  run<3, 3, 64>(s, d, bytes);
  run<1, 2, 64>(s, d, bytes);
  run<1, 4, 64>(s, d, bytes);
  run<3, 4, 64>(s, d, bytes);
  run<1, 3, 128>(s, d, bytes);
  run<2, 3, 128>(s, d, bytes);
  run<3, 3, 128>(s, d, bytes);
  return 0;
}
