// Fused GroupNorm/FiLM/GELU/resize -> Conv1d (k=3 dilated | k=1) -> skip/bias/statistics,
// as an implicit GEMM on the gfx950 matrix cores.  See kernels.hpp for the operator contract
// and DESIGN.md section 3 for the tiling.
//
//   workgroup  = 256 threads = 4 waves (x WGN = 2 for 128-channel tiles), output tile 256 time rows x CT channels
//   wave       = 64 rows x 32*WN channels = 2 x WN tiles of v_mfma_f32_32x32x16_f16 (VQVS_PREC_F16, operands = the stored
//                fp16 values) or v_mfma_f32_32x32x16_bf16 (VQVS_PREC_BF16; VQVS_PREC_F32 through the split below)
//   GEMM roles : A = activations (M = time), B = weights (N = output channel), so an
//                accumulator lane owns ONE output channel.
//   K loop     : segment -> chunk of 32 input channels -> tap -> 2 k-steps of 16.
//                One staged activation chunk (rows t0-d .. t0+255+d, 32 channels, prologue
//                applied once) serves all three taps: the taps differ only in the LDS row a
//                lane reads.
//   pipeline   : the K loop is software-pipelined.  While the matrix cores work on chunk i
//                (LDS buffer i&1) the global loads of chunk i+1 (raw activations, scale/shift,
//                weights) are already in flight into registers; they are transformed
//                (affine + GELU + bf16 split) and written to LDS buffer (i+1)&1 after the
//                MFMAs.  One __syncthreads() per chunk.
//   LDS rows are 64 B of 2-byte operands padded to 80 B: a ds_read_b128 lane group then touches 16
//                distinct 16-byte slots (5*r mod 16 is a bijection) -> conflict free.
//   VQVS_PREC_F32: operands are split x = hi + lo in bf16 and the product is evaluated as
//                hi*hi + lo*hi + hi*lo with fp32 accumulation (drops only lo*lo ~ 2^-18).
//   epilogue   : accumulators -> LDS -> whole rows: [GELU (ConvMFCCEncoder)] + identity skip | x gelu'(u) (guidance backward),
//                tile statistics for the next GroupNorm (or its backward), one rounding to the storage type, coalesced stores.
#include <cstdio>
#include <cstdlib>

#include <atomic>

#include "kernels.hpp"

// Optional in-kernel phase timing (tools/conv_phases.py; build with -DVQVS_TIMING into a separate library):
// every wave accumulates s_memtime deltas per phase and adds them to g_conv_timing at exit.
#ifdef VQVS_TIMING
__device__ unsigned long long g_conv_timing[24];
__device__ unsigned long long g_conv_span[4] = {~0ull, 0ull, ~0ull, 0ull};  // min/max workgroup start, min/max end (wall clock, 100 MHz)
#define TMARK(i)                                                    \
  {                                                                 \
    const unsigned long long _t = __builtin_amdgcn_s_memtime();     \
    tacc[i] += _t - tlast;                                          \
    tlast = _t;                                                     \
  }
#else
#define TMARK(i)
#endif

namespace vqvs {

namespace {

constexpr int TT_MAX = 256;  // staged time rows per workgroup (WM = 2); WM = 1 workgroups stage 128
constexpr int ROWB = 80;  // LDS bytes per 32-channel row (64 data + 16 pad)

template <int HALO, int TT>
constexpr int act_bytes() { return (TT + HALO) * ROWB; }

// MFMA operand element type that goes with an activation storage type: fp32 storage is split into two bf16 planes,
// 2-byte storage types are their own operand type (raw segments reach LDS as untouched bits).
template <typename T> struct Op { typedef bf16x8 v8; };
template <> struct Op<half_t> { typedef f16x8 v8; };
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <bool X3, typename V8>
__device__ __forceinline__ void put_row(char* act_hi, char* act_lo, int off, f32x8 v) {
  if constexpr (X3) {
    bf16x8 hi, lo;
    split_bf16(v, hi, lo);
    *reinterpret_cast<bf16x8*>(act_hi + off) = hi;
    *reinterpret_cast<bf16x8*>(act_lo + off) = lo;
  } else {
    *reinterpret_cast<V8*>(act_hi + off) = __builtin_convertvector(v, V8);
  }
}

template <int GQ>
__device__ __forceinline__ f32x8 affine_gelu(f32x8 v, const f32x8& sc, const f32x8& sh) {
  f32x8 r;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const f32x2 g = gelu2<GQ>(fma2(f32x2{v[j], v[j + 1]}, f32x2{sc[j], sc[j + 1]}, f32x2{sh[j], sh[j + 1]}));
    r[j] = g[0];
    r[j + 1] = g[1];
  }
  return r;
}

// raw (untransformed) octet as it sits in HBM, fetched with a buffer load: the address is
// (wave-uniform descriptor, per-lane 32-bit byte offset) -- no 64-bit per-lane arithmetic -- and an offset
// outside [0, num_records) (rows before the start / after the end of the clip) simply returns zeros.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct Raw8;
template <> struct Raw8<float> {
  u32x4 a, b;
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int off) {
    a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 0);
  }
  __device__ __forceinline__ f32x8 get() const {
    const f32x4 x = __builtin_bit_cast(f32x4, a), y = __builtin_bit_cast(f32x4, b);
    return f32x8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
  }
};
template <> struct Raw8<bf16_t> {
  u32x4 a;
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int off) { a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
  __device__ __forceinline__ f32x8 get() const { return __builtin_convertvector(__builtin_bit_cast(bf16x8, a), f32x8); }
};
template <> struct Raw8<half_t> {
  u32x4 a;
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int off) { a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
  __device__ __forceinline__ f32x8 get() const { return __builtin_convertvector(__builtin_bit_cast(f16x8, a), f32x8); }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long long bytes) {
  const int n = bytes > 0x7fffffffLL ? 0x7fffffff : (int)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}

// Everything a thread needs to know about the segment it is staging.  It is rebuilt only when the K loop crosses
// into the next segment (<= 3 times per tile): the per-chunk work is then `ch`-relative address arithmetic, with no
// scalar loads of the argument block inside the loop.
struct IterGeom {
  int s, ch, nch;   // segment, chunk, chunks in this segment
  int ntaps, d;     // taps and dilation (0 for 1-tap)
  int nrows;        // LDS rows to stage
  int base_time;    // time index of LDS row 0 (in the staged resolution)
  int row_bound;    // valid time range [0, row_bound)
  int up, avg, xform;  // (ints, not bools: keeps the struct free of sub-dword tails, so copies stay in registers)
  const void* clip;           // this clip's rows of the source tensor ...
  int clip_bytes;             // ... and their size (the buffer descriptor is rebuilt from these two at each use)
  int base_off, step;         // byte offset of the thread's first (row, octet) item in chunk 0; bytes between its items
  const float2* ssp;          // (scale, shift) of the thread's octet in chunk 0
  const void* avg_src;        // avg-pool path: the thread's octet of row 0, chunk 0
  int Csrc, c0;
  int wbase, wstep;           // byte offset of chunk 0 in the packed weights, bytes per chunk
  int aff2;                   // two-tensor affine prologue (SegDesc.src2): rows of `clip2` at the same offsets, coefficients at `cfp`
  const void* clip2;
  const float4* cfp;          // (P, Q, R, 0) of the thread's octet in chunk 0
};

// DMA (all segments raw bf16, i.e. inputs already transformed by xform_kernel or untransformed skip-conv inputs): the
// chunk goes from memory straight into LDS with `buffer_load_dwordx4 ... lds` -- no staging registers, no VALU, no
// ds_write.  A wave-wide DMA writes 64 x 16 B contiguously, so the LDS rows are unpadded (64 B); bank conflicts of the
// fragment reads are avoided by an XOR swizzle of the 16-byte column with bits 2..3 of the row, applied on the global
// address side when loading and on the LDS address side when reading.
// BWF: the epilogue can multiply by gelu'(u) (fused GELU backward of the guidance schedule).  A template flag, not a run-time
// branch: carried by every forward convolution the extra epilogue code measured +1.5...2 % (21.5 -> 21.9 ms per forward).
// AFF2: a segment may carry the two-tensor affine prologue of the guidance backward (kernels.hpp SegDesc.src2): template flag for the
// same reason as BWF -- its second set of prefetch registers and coefficient loads must not weigh on any other launch.
template <typename T, bool X3, int WN, int HALO, bool SKIPV, int WGN, int WM, bool DMA, bool BWF, bool AFF2 = false>
__global__ __launch_bounds__(256 * WGN, 2) void conv_mfma_kernel(const ConvArgs a) {
  static_assert(!AFF2 || (!DMA && !SKIPV), "the two-tensor prologue stages through registers and has no identity skip");
  constexpr int TT = 4 * WM * 32;  // staged rows: 4 waves along time x WM MFMA tiles of 32 rows
  constexpr int NTH = 256 * WGN;  // 4 waves along time x WGN waves along output channels
  constexpr int NPF = (TT * 4) / NTH;  // prefetched (row, octet) items per thread: exactly the 256 staged rows x 4 octets
  constexpr int CT = WGN * WN * 32;
  constexpr int ACT_BYTES = act_bytes<HALO, TT>();
  constexpr int W_BYTES = 3 * CT * ROWB;
  constexpr int NWV = (3 * CT * 4 + NTH - 1) / NTH;  // 16-byte weight pieces per thread per chunk
  constexpr int PLANES = X3 ? 2 : 1;
  constexpr int BUF_BYTES = PLANES * (ACT_BYTES + W_BYTES);
  constexpr int GQ = GeluQ<T>::q;
  typedef typename Op<T>::v8 V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef VQVS_TIMING
  unsigned long long tacc[18] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long wall0 = wall_clock64();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = (tid >> 6) & 3;   // position along time
  const int wvn = tid >> 8;          // position along output channels (0 .. WGN-1)
  // XCD-aware tile order.  Workgroup `lin` (x fastest) is dispatched to XCD lin % 8, each XCD with its own L2: give every
  // XCD one CONTIGUOUS run of tiles, ordered output-channel tile fastest (co-tiles re-read the same input rows), then
  // time (neighbours share their halo rows), then clip -- so those re-reads hit the XCD's L2 instead of crossing dies.
  int tile_x, tile_y, tile_b;
  {
    const int nx = (int)gridDim.x, ny = (int)gridDim.y;
    const int total = nx * ny * (int)gridDim.z;
    const int lin = (int)blockIdx.x + nx * ((int)blockIdx.y + ny * (int)blockIdx.z);
    const int k = lin & 7, j = lin >> 3;           // XCD, position within the XCD's run
    const int q = total >> 3, r = total & 7;        // XCD k owns q + (k < r) tiles, starting at k*q + min(k, r)
    const int id = k * q + (k < r ? k : r) + j;
    tile_y = id % ny;
    tile_x = (id / ny) % nx;
    tile_b = id / (ny * nx);
  }
  const int b = tile_b;
  const int co0 = tile_y * CT;
  // A tile produces tile_rows = 256 - 2*dmax output rows, so that rows + halo = 256 staged rows = a whole number
  // of (row, octet) items per thread: no wave carries an extra, mostly empty, halo item to every barrier.
  const int TTO = a.tile_rows;
  const int t0 = tile_x * TTO;
  const int oct = tid & 3;
  const int l31 = lane & 31;


  // ---- per-thread invariants of the staging loops (hoisted out of the K loop) ----
  int act_lds[NPF];   // LDS byte offset of (row r_i, octet)
#pragma unroll
  for (int i = 0; i < NPF; ++i) act_lds[i] = ((tid >> 2) + (NTH / 4) * i) * ROWB + oct * 16;
  int w_lds[NWV], w_goff[NWV];  // LDS byte offset / global byte offset (within one chunk block) of weight piece i
#pragma unroll
  for (int i = 0; i < NWV; ++i) {
    const int idx = tid + NTH * i;
    const int row = idx >> 2, q = idx & 3;
    const int tap = row / CT, col = row - tap * CT;
    w_lds[i] = row * ROWB + q * 16;
    w_goff[i] = ((tap * a.Cout + co0 + col) * 32 + q * 8) * 2;
  }
  const __amdgpu_buffer_rsrc_t rs_wh = make_rsrc(a.w_hi, a.w_bytes);
  const __amdgpu_buffer_rsrc_t rs_wl = make_rsrc(X3 ? a.w_lo : a.w_hi, a.w_bytes);
  (void)rs_wl;

  // accumulators start at the bias of the lane's output channel (an accumulator lane owns one channel)
  f32x16 acc[WM][WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const float bj = a.bias[co0 + (wvn * WN + j) * 32 + (lane & 31)];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bj;
  }

  int niter = 0;
  for (int s = 0; s < a.nseg; ++s) niter += a.seg[s].C >> 5;

  auto geom = [&](int s) {
    IterGeom g;
    // field-wise selects over the three argument-block entries: one batch of scalar loads, no dynamically indexed copy
    struct { const void* src; const float2* ss; long long w_off; int Csrc, c0, C, Lsrc, ntaps, dil, resize, ss_stride, ss_c0; } sg;
#define VQVS_SEGF(f) sg.f = s == 0 ? a.seg[0].f : (s == 1 ? a.seg[1].f : a.seg[2].f)
    VQVS_SEGF(src); VQVS_SEGF(ss); VQVS_SEGF(w_off); VQVS_SEGF(Csrc); VQVS_SEGF(c0); VQVS_SEGF(C); VQVS_SEGF(Lsrc);
    VQVS_SEGF(ntaps); VQVS_SEGF(dil); VQVS_SEGF(resize); VQVS_SEGF(ss_stride); VQVS_SEGF(ss_c0);
#undef VQVS_SEGF
    g.s = s;
    g.ch = 0;
    g.nch = sg.C >> 5;
    g.ntaps = sg.ntaps;
    g.d = (sg.ntaps == 3) ? sg.dil : 0;
    g.up = sg.resize == RESIZE_UP2;
    g.avg = sg.resize == RESIZE_AVG2;
    g.xform = sg.ss != nullptr;
    g.nrows = g.up ? (TTO / 2 + 2) : (TTO + 2 * g.d);
    g.base_time = g.up ? ((t0 >> 1) - 1) : (t0 - g.d);
    g.row_bound = g.avg ? a.Lout : sg.Lsrc;
    const T* const clip = reinterpret_cast<const T*>(sg.src) + (size_t)b * sg.Lsrc * sg.Csrc;
    // byte offset of item i = ((base_time + r_i) * Csrc + c0 + ch*32 + oct*8) * sizeof(T); negative or past-the-end rows
    // fall outside the descriptor and read as zero (their LDS rows are re-zeroed after the prologue anyway)
    g.clip = clip;
    {
      const long long nb = (long long)sg.Lsrc * sg.Csrc * (int)sizeof(T);
      g.clip_bytes = nb > 0x7fffffffLL ? 0x7fffffff : (int)nb;
    }
    const int row_bytes = sg.Csrc * (int)sizeof(T);
    g.base_off = (g.base_time * sg.Csrc + sg.c0 + oct * 8) * (int)sizeof(T) + (tid >> 2) * row_bytes;
    g.step = (NTH / 4) * row_bytes;
    g.ssp = g.xform ? sg.ss + (size_t)b * sg.ss_stride + sg.ss_c0 + oct * 8 : nullptr;
    g.avg_src = clip + sg.c0 + oct * 8;
    g.Csrc = sg.Csrc;
    g.c0 = sg.c0;
    g.wbase = (int)(sg.w_off * 2);
    g.wstep = g.ntaps * a.Cout * 32 * 2;
    g.aff2 = 0;
    g.clip2 = nullptr;
    g.cfp = nullptr;
    if constexpr (AFF2) {
      const void* s2 = s == 0 ? a.seg[0].src2 : (s == 1 ? a.seg[1].src2 : a.seg[2].src2);
      if (s2 != nullptr) {
        const float4* cf = s == 0 ? a.seg[0].coef : (s == 1 ? a.seg[1].coef : a.seg[2].coef);
        const int cst = s == 0 ? a.seg[0].coef_stride : (s == 1 ? a.seg[1].coef_stride : a.seg[2].coef_stride);
        const int cc0 = s == 0 ? a.seg[0].coef_c0 : (s == 1 ? a.seg[1].coef_c0 : a.seg[2].coef_c0);
        g.aff2 = 1;
        g.clip2 = reinterpret_cast<const T*>(s2) + (size_t)b * sg.Lsrc * sg.Csrc;
        g.cfp = cf + (size_t)b * cst + cc0 + oct * 8;
      }
    }
    return g;
  };

  // ---- prefetch registers -------------------------------------------------------------
  Raw8<T> ra[NPF];
  Raw8<T> ra2[AFF2 ? NPF : 1];  // AFF2: the second tensor's rows
  f32x4 rcf[AFF2 ? 8 : 1];      // AFF2: (P, Q, R, 0) of the thread's eight channels
  (void)ra2;
  (void)rcf;
  f32x4 rss[4];
  u32x4 rwh[NWV], rwl[NWV];
  (void)rwl;

  auto issue_loads = [&](const IterGeom& g) {
    if (g.xform) {
      const float2* p = g.ssp + g.ch * 32;
#pragma unroll
      for (int j = 0; j < 4; ++j) rss[j] = *reinterpret_cast<const f32x4*>(p + 2 * j);
    }
    TMARK(11)
    if (!g.avg) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.clip), 0, g.clip_bytes, 0x00020000);
      const int base = g.base_off + g.ch * (32 * (int)sizeof(T));
#pragma unroll
      for (int i = 0; i < NPF; ++i) ra[i].load(rs, base + i * g.step);
      if constexpr (AFF2) {
        if (g.aff2) {
          const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.clip2), 0, g.clip_bytes, 0x00020000);
#pragma unroll
          for (int i = 0; i < NPF; ++i) ra2[i].load(rs2, base + i * g.step);
          const float4* cp = g.cfp + g.ch * 32;
#pragma unroll
          for (int j = 0; j < 8; ++j) rcf[j] = *reinterpret_cast<const f32x4*>(cp + j);
        }
      }
    }
    TMARK(12)
    const int wbase = g.wbase + g.ch * g.wstep;
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
      rwh[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_wh, w_goff[i] + wbase, 0, 0);
      if constexpr (X3) rwl[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_wl, w_goff[i] + wbase, 0, 0);
    }
    TMARK(13)
  };

  // ---- DMA staging (see the template comment): 1 KiB pieces = 16 LDS rows of 64 B, dealt round-robin to the waves ----
  constexpr int DROW = 64;
  constexpr int D_ACT_BYTES = (TT + HALO) * DROW;
  constexpr int D_BUF_BYTES = D_ACT_BYTES + 3 * CT * DROW;
  auto dma_stage = [&](const IterGeom& g, int buf) {
    if constexpr (DMA) {
      typedef __attribute__((address_space(3))) void* lds_ptr;
      const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
      constexpr int NWAVES = NTH / 64;
      const int q = lane & 3, rl = lane >> 2;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.clip), 0, g.clip_bytes, 0x00020000);
      char* const abuf = smem + buf * D_BUF_BYTES;
      const int npa = (g.nrows + 15) >> 4;
      for (int p = w8; p < npa; p += NWAVES) {
        const int r = p * 16 + rl;
        const int o = q ^ ((r >> 2) & 3);
        const int voff = ((g.base_time + r) * g.Csrc + g.c0 + g.ch * 32 + o * 8) * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(abuf + p * 1024), 16, voff, 0, 0, 0);
      }
      char* const wbuf = abuf + D_ACT_BYTES;
      const int npw = (g.ntaps * CT) >> 4;
      const int wbase = g.wbase + g.ch * g.wstep;
      for (int p = w8; p < npw; p += NWAVES) {
        const int r = p * 16 + rl;
        const int tap = r / CT, col = r - tap * CT;
        const int o = q ^ ((r >> 2) & 3);
        const int voff = wbase + ((tap * a.Cout + co0 + col) * 32 + o * 8) * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wh, (lds_ptr)(wbuf + p * 1024), 16, voff, 0, 0, 0);
      }
    }
  };

  auto store_stage = [&](const IterGeom& g, int buf) {
    char* const act_hi = smem + buf * BUF_BYTES;
    char* const act_lo = act_hi + ACT_BYTES;  // X3 only
    char* const w_hi = act_hi + PLANES * ACT_BYTES;
    char* const w_lo = w_hi + W_BYTES;  // X3 only
    f32x8 sc, sh;
    if (g.xform) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc[2 * j] = rss[j][0]; sh[2 * j] = rss[j][1]; sc[2 * j + 1] = rss[j][2]; sh[2 * j + 1] = rss[j][3];
      }
    }
    if (AFF2 && g.aff2) {
      // d h = P * du + Q * h + R per (clip, channel) (GroupNorm backward), then the rows outside the clip back to the convolution's
      // zero padding (their loads returned zeros, the affine made them R).  Host side: 3-tap segment, whole 256-row window.
      if constexpr (AFF2) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
          f32x8 v = ra[i].get();
          const f32x8 x = ra2[i].get();
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaf(rcf[j][0], v[j], fmaf(rcf[j][1], x[j], rcf[j][2]));
          put_row<X3, V8>(act_hi, act_lo, act_lds[i], v);
        }
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
          const int r = (tid >> 2) + (NTH / 4) * i;
          const int tm = g.base_time + r;
          if (r >= g.nrows || tm < 0 || tm >= g.row_bound) put_row<X3, V8>(act_hi, act_lo, act_lds[i], f32x8_zero());
        }
      }
    } else if (!g.avg && g.nrows >= TT && (g.xform || !X3)) {
      // fast path (the usual 3-tap segment): every staged item is a row of this segment and the whole wave takes the same
      // branch, so the four items run as straight-line code -- no per-item exec masks, one uniform branch instead of four
      // (measured -0.7 % convolution time)
      if (g.xform) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) put_row<X3, V8>(act_hi, act_lo, act_lds[i], affine_gelu<GQ>(ra[i].get(), sc, sh));
      } else {
        if constexpr (!X3) {
#pragma unroll
          for (int i = 0; i < NPF; ++i) *reinterpret_cast<u32x4*>(act_hi + act_lds[i]) = ra[i].a;
        }
      }
      if (g.base_time < 0 || g.base_time + g.nrows > g.row_bound) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
          const int r = (tid >> 2) + (NTH / 4) * i;
          const int tm = g.base_time + r;
          if (tm < 0 || tm >= g.row_bound) put_row<X3, V8>(act_hi, act_lo, act_lds[i], f32x8_zero());
        }
      }
    } else if (!g.avg) {
#pragma unroll
      for (int i = 0; i < NPF; ++i) {
        const int r = (tid >> 2) + (NTH / 4) * i;
        if (r < g.nrows) {
          if (!X3 && !g.xform) {  // raw bf16 segment: the bits go to LDS untouched
            *reinterpret_cast<u32x4*>(act_hi + act_lds[i]) = ra[i].a;
          } else {
            f32x8 v = ra[i].get();
            if (g.xform) v = affine_gelu<GQ>(v, sc, sh);
            put_row<X3, V8>(act_hi, act_lo, act_lds[i], v);
          }
        }
      }
      // zero padding of the convolution: only tiles that touch a sequence end have rows outside [0, row_bound)
      if (g.base_time < 0 || g.base_time + g.nrows > g.row_bound) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
          const int r = (tid >> 2) + (NTH / 4) * i;
          const int tm = g.base_time + r;
          if (r < g.nrows && (tm < 0 || tm >= g.row_bound)) put_row<X3, V8>(act_hi, act_lo, act_lds[i], f32x8_zero());
        }
      }
    } else {  // avg-pool segments (8 of 130 convs): staged synchronously, two source rows per LDS row
      const T* const src_c = reinterpret_cast<const T*>(g.avg_src) + g.ch * 32;
      for (int r = tid >> 2; r < g.nrows; r += NTH / 4) {
        const int tm = g.base_time + r;
        f32x8 v = f32x8_zero();
        if (tm >= 0 && tm < g.row_bound) {
          const T* p = src_c + (size_t)(2 * tm) * g.Csrc;
          f32x8 v0 = Elem<T>::load8(p);
          f32x8 v1 = Elem<T>::load8(p + g.Csrc);
          if (g.xform) { v0 = affine_gelu<GQ>(v0, sc, sh); v1 = affine_gelu<GQ>(v1, sc, sh); }
          v = (v0 + v1) * 0.5f;
        }
        put_row<X3, V8>(act_hi, act_lo, r * ROWB + oct * 16, v);
      }
    }
    // weights: every piece is written (1-tap segments leave garbage in the unused tap rows)
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
      if (NWV * NTH == 3 * CT * 4 || tid + NTH * i < 3 * CT * 4) {
        *reinterpret_cast<u32x4*>(w_hi + w_lds[i]) = rwh[i];
        if constexpr (X3) *reinterpret_cast<u32x4*>(w_lo + w_lds[i]) = rwl[i];
      }
    }
  };

  auto mfma_stage = [&](const IterGeom& g, int buf) {
    const char* const act_hi = DMA ? smem + buf * D_BUF_BYTES : smem + buf * BUF_BYTES;
    const char* const act_lo = act_hi + ACT_BYTES;
    const char* const w_hi = DMA ? act_hi + D_ACT_BYTES : act_hi + PLANES * ACT_BYTES;
    const char* const w_lo = w_hi + W_BYTES;
    constexpr int RB = DMA ? DROW : ROWB;  // LDS row pitch
    // byte address of k-octet `o` (0..3) of LDS row `r`: padded rows, or unpadded rows with the column swizzle
    auto frag_addr = [&](int r, int o) { return DMA ? r * DROW + ((o ^ ((r >> 2) & 3)) << 4) : r * ROWB + (o << 4); };
    const int ohalf = lane >> 5;
    for (int k = 0; k < g.ntaps; ++k) {
      const int toff = (g.ntaps == 3) ? (k - 1) * g.d : 0;
      int ar[WM];  // LDS row of the A fragment (activation) for this tap
#pragma unroll
      for (int mt = 0; mt < WM; ++mt) {
        const int tl = wave * (WM * 32) + mt * 32 + l31;
        ar[mt] = g.up ? (((tl + toff) >> 1) + 1) : (tl + toff + g.d);
      }
      const int wrow = k * CT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        V8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) {
          const int ad = frag_addr(ar[mt], ks * 2 + ohalf);
          ah[mt] = *reinterpret_cast<const V8*>(act_hi + ad);
          if constexpr (X3) al[mt] = *reinterpret_cast<const V8*>(act_lo + ad);
        }
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
          const int ad = frag_addr(wrow + (wvn * WN + nt) * 32 + l31, ks * 2 + ohalf);
          bh[nt] = *reinterpret_cast<const V8*>(w_hi + ad);
          if constexpr (X3) bl[nt] = *reinterpret_cast<const V8*>(w_lo + ad);
        }
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) {
            if constexpr (X3) {
              acc[mt][nt] = mfma32(al[mt], bh[nt], acc[mt][nt]);
              acc[mt][nt] = mfma32(ah[mt], bl[nt], acc[mt][nt]);
            }
            acc[mt][nt] = mfma32(ah[mt], bh[nt], acc[mt][nt]);
          }
      }
    }
    (void)RB;
  };

  // epilogue geometry (needed early: the identity-skip rows are prefetched before the last MFMA phase)
  constexpr int OS = CT + 4;
  constexpr int OPR = CT / 8;     // 8-channel octets per row
  constexpr int RPP = NTH / OPR;  // rows per pass
  constexpr int NEP = TT / RPP;   // passes
  constexpr bool SKIP_PF = SKIPV && !X3;  // identity-skip variant; (fp32 mode: register budget)
  const int eoct = tid % OPR;
  const int r0 = tid / OPR;
  const int cg = co0 + eoct * 8;
  const T* const skip_b = a.skip ? reinterpret_cast<const T*>(a.skip) + (size_t)b * a.skip_L * a.skip_C + cg : nullptr;
  const bool skip_pf = SKIP_PF && skip_b != nullptr && a.skip_resize != RESIZE_AVG2;
  Raw8<T> rsk[SKIP_PF ? NEP : 1];
  const __amdgpu_buffer_rsrc_t rs_skip = make_rsrc(
      a.skip ? reinterpret_cast<const T*>(a.skip) + (size_t)b * a.skip_L * a.skip_C : reinterpret_cast<const T*>(a.bias),
      a.skip ? (long long)a.skip_L * a.skip_C * (int)sizeof(T) : 0);

  // ------------------------------ pipelined K loop ------------------------------
  {
    IterGeom cur = geom(0);
    TMARK(0)  // setup
    if constexpr (DMA) {
      // chunk it+1 streams into the other LDS buffer while the matrix cores work on chunk it.  The DMA is issued AFTER
      // the barrier: every wave has then finished reading that buffer (chunk it-1).
      dma_stage(cur, 0);
      for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces of chunk `it` have landed
        TMARK(2)
        __syncthreads();
        TMARK(4)
        IterGeom nxt = cur;
        if (it + 1 < niter) {
          if (++nxt.ch == nxt.nch) nxt = geom(cur.s + 1);
          dma_stage(nxt, buf ^ 1);
        } else if (skip_pf) {
#pragma unroll
          for (int i = 0; i < (SKIP_PF ? NEP : 1); ++i) {
            const int tm = t0 + r0 + RPP * i;
            const int ts = a.skip_resize == RESIZE_UP2 ? (tm >> 1) : tm;
            rsk[i].load(rs_skip, (ts * a.skip_C + cg) * (int)sizeof(T));
          }
        }
        TMARK(1)
        mfma_stage(cur, buf);
        TMARK(5)
        cur = nxt;
      }
    } else {
    issue_loads(cur);
    for (int it = 0; it < niter; ++it) {
      const int buf = it & 1;
#ifdef VQVS_TIMING
      TMARK(16)  // loop back-edge
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): isolate the wait for the global loads
      TMARK(2)
#endif
      store_stage(cur, buf);  // consumes the registers loaded one iteration ago
      TMARK(3)  // prologue arithmetic + LDS writes
      IterGeom nxt = cur;
      const bool more = it + 1 < niter;
      if (more) {
        if (++nxt.ch == nxt.nch) nxt = geom(cur.s + 1);
        TMARK(14)
        issue_loads(nxt);  // in flight across the barrier and the MFMAs below
      } else if (skip_pf) {
#pragma unroll
        for (int i = 0; i < (SKIP_PF ? NEP : 1); ++i) {
          const int tm = t0 + r0 + RPP * i;
          const int ts = a.skip_resize == RESIZE_UP2 ? (tm >> 1) : tm;
          rsk[i].load(rs_skip, (ts * a.skip_C + cg) * (int)sizeof(T));  // rows past the end read as zero and are never stored
        }
      }
      TMARK(1)
      __syncthreads();
      TMARK(4)  // barrier
      mfma_stage(cur, buf);
      TMARK(5)  // LDS fragment reads + MFMA
#ifdef VQVS_TIMING
      {
        float probe;
        asm volatile("v_mov_b32 %0, %1" : "=v"(probe) : "v"(acc[WM - 1][WN - 1][15]));  // waits for the last MFMA result
        asm volatile("" ::"v"(probe));
      }
      TMARK(17)  // matrix-pipe drain
#endif
      cur = nxt;
    }
    }
  }

  // ------------------------------ epilogue ------------------------------
  // accumulators -> LDS tile [256][CT+4] f32 -> whole-row reads: bias, skip, statistics, store.
  float* const ost = reinterpret_cast<float*>(smem);
  __syncthreads();
  TMARK(6)  // barrier before the epilogue
#pragma unroll
  for (int mt = 0; mt < WM; ++mt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave * (WM * 32) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        ost[row * OS + (wvn * WN + nt) * 32 + l31] = acc[mt][nt][r];
      }
  TMARK(7)  // accumulators -> LDS
  __syncthreads();
  TMARK(8)

  f32x8 s1 = f32x8_zero(), s2 = f32x8_zero();
  const int out_rows = a.out_rows ? a.out_rows : a.Lout;
  // fused GELU backward (guidance): this thread's eight output channels read one forward tensor and their (scale, shift)
  f32x8 bsc = f32x8_zero(), bsh = f32x8_zero();
  const T* bxf = nullptr;
  int bxf_C = 0;
  if ((BWF && a.nbw)) {
    const int k = (a.nbw == 2 && cg >= a.bw[1].c_begin) ? 1 : 0;
    bxf_C = a.bw[k].C;
    bxf = reinterpret_cast<const T*>(a.bw[k].xf) + (size_t)b * a.Lout * bxf_C + (cg - a.bw[k].c_begin);
    const float2* p = a.bw_ss + (size_t)b * a.bw_ss_stride + cg;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float2 q = p[j];
      bsc[j] = q.x;
      bsh[j] = q.y;
    }
  }
#pragma unroll
  for (int i = 0; i < NEP; ++i) {
    const int r = r0 + RPP * i;
    const int tm = t0 + r;
    if (r < TTO && tm < a.Lout) {
      f32x8 v = Elem<float>::load8(ost + r * OS + eoct * 8);
      if (a.epi_gelu) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const f32x2 g = gelu2<GQ>(f32x2{v[j], v[j + 1]});
          v[j] = g[0];
          v[j + 1] = g[1];
        }
      }
      if (skip_pf) {
        v += rsk[SKIP_PF ? i : 0].get();
      } else if (skip_b) {
        if (a.skip_resize == RESIZE_AVG2) {  // avg-pooled identity skip (down-sampling blocks)
          const T* p = skip_b + (size_t)(2 * tm) * a.skip_C;
          v += (Elem<T>::load8(p) + Elem<T>::load8(p + a.skip_C)) * 0.5f;
        } else {
          v += Elem<T>::load8(skip_b + (size_t)(a.skip_resize == RESIZE_UP2 ? (tm >> 1) : tm) * a.skip_C);
        }
      }
      if ((BWF && a.nbw)) {  // v <- v * gelu'(u); statistics (sum v, sum v*u)
        const f32x8 x = Elem<T>::load8(bxf + (size_t)tm * bxf_C);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float u = fmaf(x[j], bsc[j], bsh[j]);
          v[j] *= gelu_grad_f(u);
          s1[j] += v[j];
          s2[j] = fmaf(v[j], u, s2[j]);
        }
      } else {
        s1 += v;
        s2 += v * v;
      }
      const size_t oidx = ((size_t)b * out_rows + tm) * a.Cout + cg;
      if (a.out_f32)
        Elem<float>::store8(reinterpret_cast<float*>(a.out) + oidx, v);
      else
        Elem<T>::store8(reinterpret_cast<T*>(a.out) + oidx, v);
    }
  }
  TMARK(9)  // row phase: skip, statistics, stores
  if (a.stats) {
    __syncthreads();
    float* const red = reinterpret_cast<float*>(smem);  // [RPP][CT][2]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(r0 * CT + eoct * 8 + j) * 2 + 0] = s1[j];
      red[(r0 * CT + eoct * 8 + j) * 2 + 1] = s2[j];
    }
    __syncthreads();
    if (tid < CT) {
      float t1 = 0.f, t2 = 0.f;
      for (int g = 0; g < RPP; ++g) {  // fixed order: deterministic
        t1 += red[(g * CT + tid) * 2 + 0];
        t2 += red[(g * CT + tid) * 2 + 1];
      }
      float* o = a.stats + (((size_t)b * a.ntiles + tile_x) * a.Cout + co0 + tid) * 2;
      o[0] = t1;
      o[1] = t2;
    }
  }
#ifdef VQVS_TIMING
  TMARK(10)  // statistics reduce
  if (tid == 0) {
    const unsigned long long wall1 = wall_clock64();
    atomicMin(&g_conv_span[0], wall0);
    atomicMax(&g_conv_span[1], wall0);
    atomicMin(&g_conv_span[2], wall1);
    atomicMax(&g_conv_span[3], wall1);
  }
  if (lane == 0 && wave == 1 && ((tile_x + tile_b) & 15) == 3) {  // a 1/64 sample: the atomics must not become the workload
    for (int i = 0; i < 18; ++i) atomicAdd(&g_conv_timing[i], tacc[i]);
    atomicAdd(&g_conv_timing[23], 1ull);
  }
#endif
}

bool dma_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VQVS_DMA");  // 0 = stage raw segments through registers (A/B measurements)
    v = e ? atoi(e) : 1;
  }
  return v != 0;
}

template <bool X3, int WN, int HALO, int WGN = 1, int WM = 2>
constexpr int lds_bytes() {
  constexpr int CT = WGN * WN * 32;
  constexpr int TT = 4 * WM * 32;
  constexpr int stage = 2 * (X3 ? 2 : 1) * (act_bytes<HALO, TT>() + 3 * CT * ROWB);
  constexpr int ost = TT * (CT + 4) * 4;
  return stage > ost ? stage : ost;
}

template <typename T, bool X3, int WN, int HALO, bool SKIPV, int WGN, int WM, bool DMA, bool BWF, bool AFF2 = false>
int launch_o(const ConvArgs& a, int B, hipStream_t st) {
  constexpr int LDS = lds_bytes<X3, WN, HALO, WGN, WM>();
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static std::atomic<bool> attr_done[64];  // (function attributes are per device: keyed by the current one)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_done[dev].load(std::memory_order_acquire)) {
    VQVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<T, X3, WN, HALO, SKIPV, WGN, WM, DMA, BWF, AFF2>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done[dev].store(true, std::memory_order_release);
  }
  dim3 grid((a.Lout + a.tile_rows - 1) / a.tile_rows, a.Cout / (WGN * WN * 32), B);
  hipLaunchKernelGGL((conv_mfma_kernel<T, X3, WN, HALO, SKIPV, WGN, WM, DMA, BWF, AFF2>), grid, dim3(256 * WGN), LDS, st, a);
  VQVS_HIP(hipGetLastError());
  return 0;
}

template <typename T, bool X3, int WN, int HALO, bool SKIPV, int WGN = 1, int WM = 2, bool DMA = false>
int launch_t(const ConvArgs& a, int B, hipStream_t st) {
  if constexpr (!SKIPV) {  // (transposed convolutions have no identity skip)
    bool aff2 = false;
    for (int s = 0; s < a.nseg; ++s) aff2 = aff2 || a.seg[s].src2 != nullptr;
    if (aff2) {  // two-tensor affine prologue (guidance backward, conv1^T of a block): register staging, small halo
      if constexpr (!DMA && HALO == 4) {
        if (a.nbw) return launch_o<T, X3, WN, HALO, SKIPV, WGN, WM, DMA, true, true>(a, B, st);
        return launch_o<T, X3, WN, HALO, SKIPV, WGN, WM, DMA, false, true>(a, B, st);
      } else {
        VQVS_FAIL(-1, "conv: the two-tensor prologue is built for register-staged launches of dilation <= 2");
      }
    }
    if (a.nbw) return launch_o<T, X3, WN, HALO, SKIPV, WGN, WM, DMA, true>(a, B, st);
  } else {
    if (a.nbw) VQVS_FAIL(-1, "conv: the fused GELU backward is not built for identity-skip launches");
  }
  return launch_o<T, X3, WN, HALO, SKIPV, WGN, WM, DMA, false>(a, B, st);
}

template <typename T, bool X3>
int launch_p(const ConvArgs& a, int B, hipStream_t st, bool wide, bool big_halo, int dmax) {
  if constexpr (!X3) {
    // 128-channel output tiles (8 waves): the prologue and the activation reads are shared by twice as many channels
    if (a.Cout % 128 == 0) {
      bool raw = true;  // every segment untransformed and unresized-or-upsampled: stage by LDS-DMA
      for (int s = 0; s < a.nseg; ++s) raw = raw && a.seg[s].ss == nullptr && a.seg[s].resize != RESIZE_AVG2 && a.seg[s].src2 == nullptr;
      // (measured on the 512-channel launches: -3...-5 % for plain, -10...-14 % with 1x1 skip-conv segments,
      //  +4...+7 % with an identity skip, which therefore keeps the register path)
      if (raw && a.skip == nullptr && dma_enabled()) {
        return big_halo ? launch_t<T, X3, 2, 64, false, 2, 2, true>(a, B, st) : launch_t<T, X3, 2, 4, false, 2, 2, true>(a, B, st);
      }
      if (a.skip != nullptr) return big_halo ? launch_t<T, X3, 2, 64, true, 2>(a, B, st) : launch_t<T, X3, 2, 4, true, 2>(a, B, st);
      return big_halo ? launch_t<T, X3, 2, 64, false, 2>(a, B, st) : launch_t<T, X3, 2, 4, false, 2>(a, B, st);
    }
  }
  if constexpr (X3) {
    // fp32 mode stages two bf16 planes, so LDS allows one workgroup per CU: make it 8 waves (4 x 64 rows, 2 x 32 channels)
    if (wide) return big_halo ? launch_t<T, X3, 1, 64, false, 2>(a, B, st) : launch_t<T, X3, 1, 4, false, 2>(a, B, st);
  }
  if (a.skip != nullptr && !X3) {
    if (wide) return big_halo ? launch_t<T, X3, 2, 64, true>(a, B, st) : launch_t<T, X3, 2, 4, true>(a, B, st);
    return big_halo ? launch_t<T, X3, 1, 64, true>(a, B, st) : launch_t<T, X3, 1, 4, true>(a, B, st);
  }
  if (wide) return big_halo ? launch_t<T, X3, 2, 64, false>(a, B, st) : launch_t<T, X3, 2, 4, false>(a, B, st);
  return big_halo ? launch_t<T, X3, 1, 64, false>(a, B, st) : launch_t<T, X3, 1, 4, false>(a, B, st);
}

}  // namespace

// (128-row tiles with three workgroups per CU were measured 10-25 % slower than 256-row tiles: the per-tile
// fixed costs double while the SIMDs are already ~70 % busy.)
// ws_ok: the launch is one conv_ws_kernel covers for every length (2-byte storage, no fp32 / padded output, no epilogue GELU, no
// fused backward).  Only such launches may use the 128-row geometry (two 8-wave workgroups per CU), and only at Cout = 64: from
// 128 output channels on a workgroup's LDS (weights or out-tile) no longer fits twice into a CU; conv_mfma_kernel has no 128-row
// form at all.  VQVS_WS_ROWS128 = 0 / 1 (A/B measurements).
int conv_tile_rows(int dmax, int Cout, int precision, bool ws_ok) {
  ws_ok = ws_ok && ws_available(precision);  // (the same switches ws_plan obeys: a geometry conv_mfma_kernel lacks is never chosen with them off)
  static const int on = getenv("VQVS_WS_ROWS128") ? atoi(getenv("VQVS_WS_ROWS128")) : 0;
  if (on && ws_ok && precision != 0 && Cout == 64 && 128 - 2 * dmax >= 64) return 128 - 2 * dmax;
  // fp32 storage, 128 output channels and up: the 128-row x 128-channel tile of conv_ws_kernel (every row transformed once per 128
  // output channels); not where the dilation's halo or the rounding of short clips to 124-row tiles would eat the gain
  static const int tall = getenv("VQVS_WS_TALL") ? atoi(getenv("VQVS_WS_TALL")) : 1;  // 0: off; 2: at 512 channels too (A/B)
  if (tall && ws_ok && precision == 0 && Cout % 128 == 0 && (Cout <= 256 || tall >= 2) && dmax <= 2) return 128 - 2 * dmax;
  return TT_MAX - 2 * dmax;
}

#ifdef VQVS_TIMING
int conv_timing_read(unsigned long long* out16, int reset) {
  VQVS_HIP(hipDeviceSynchronize());
  VQVS_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_conv_timing), 24 * sizeof(unsigned long long)));
  {  // workgroup start / end spread of the launches since the last reset, in slots 18..21 (100 MHz ticks)
    unsigned long long sp[4];
    VQVS_HIP(hipMemcpyFromSymbol(sp, HIP_SYMBOL(g_conv_span), sizeof(sp)));
    out16[18] = sp[1] - sp[0];  // latest start - earliest start
    out16[19] = sp[3] - sp[2];  // latest end - earliest end
    out16[20] = sp[3] - sp[0];  // whole span
    out16[21] = sp[2] - sp[0];  // earliest end - earliest start
    if (reset) {
      const unsigned long long init[4] = {~0ull, 0ull, ~0ull, 0ull};
      VQVS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_span), init, sizeof(init)));
    }
  }
  if (reset) {
    unsigned long long z[24] = {};
    VQVS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_timing), z, sizeof(z)));
  }
  return 0;
}
#endif

int launch_conv(const ConvArgs& a, int B, int precision, hipStream_t st) {
  if (a.Cout % 32 != 0 || a.nseg < 1 || a.nseg > 3) VQVS_FAIL(-1, "conv: unsupported shape Cout=%d nseg=%d", a.Cout, a.nseg);
  int dmax = 0;
  for (int s = 0; s < a.nseg; ++s) {
    const SegDesc& g = a.seg[s];
    if (g.C % 32 != 0 || (g.ntaps != 1 && g.ntaps != 3) || g.dil > 32 || g.dil < 1)
      VQVS_FAIL(-1, "conv: unsupported segment C=%d taps=%d dil=%d", g.C, g.ntaps, g.dil);
    if (g.resize == RESIZE_UP2 && g.ntaps == 3 && g.dil != 1) VQVS_FAIL(-1, "conv: upsample needs dilation 1");
    if (g.src2 != nullptr && (g.ss != nullptr || g.resize != RESIZE_NONE || g.ntaps != 3 || g.dil > 2 || a.skip != nullptr || g.coef == nullptr))
      VQVS_FAIL(-1, "conv: the two-tensor prologue needs a plain 3-tap segment of dilation <= 2 without identity skip");
    if (g.ntaps == 3 && g.dil > dmax) dmax = g.dil;
  }
  const bool wide = (a.Cout % 64) == 0;
  const bool big_halo = dmax > 2;
  {
    const int r = launch_conv_ws(a, B, precision, st);  // the 2-byte modes' common shapes run on the wave-specialised kernel
    if (r != 0) return r < 0 ? r : 0;
  }
  if (a.gn != nullptr) VQVS_FAIL(-1, "conv: a fused GroupNorm is only taken where ws_fuses_gn() says so (Cout=%d)", a.Cout);
  static const int trace = getenv("VQVS_WS_TRACE") ? atoi(getenv("VQVS_WS_TRACE")) : 0;  // (which launches conv_ws_kernel declined)
  if (trace)
    fprintf(stderr, "conv_mfma: Cout=%d Lout=%d nseg=%d seg0(C=%d taps=%d dil=%d rsz=%d xf=%d) skip=%d epi_gelu=%d nbw=%d out_f32=%d prec=%d\n", a.Cout, a.Lout,
            a.nseg, a.seg[0].C, a.seg[0].ntaps, a.seg[0].dil, a.seg[0].resize, a.seg[0].ss != nullptr, a.skip != nullptr, a.epi_gelu, a.nbw, a.out_f32, precision);
  if (a.tile_rows != TT_MAX - 2 * dmax) VQVS_FAIL(-1, "conv: tile_rows %d does not match dilation %d (only conv_ws_kernel has a 128-row form)", a.tile_rows, dmax);
  if (precision == 0) return launch_p<float, true>(a, B, st, wide, big_halo, dmax);
  if (precision == 2) return launch_p<half_t, false>(a, B, st, wide, big_halo, dmax);
  return launch_p<bf16_t, false>(a, B, st, wide, big_halo, dmax);
}

}  // namespace vqvs
