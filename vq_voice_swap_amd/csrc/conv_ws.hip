// Wave-specialised, persistent form of the fused convolution (kernels.hpp ConvArgs) for the 2-byte
// storage modes.  Same operator contract and the same GEMM roles as conv_mfma.hip -- A = activations (M = time), B = weights
// (N = output channel), K = (segment, chunk of 32 input channels, tap) -- but the work of a tile is split between two kinds of
// waves that never change role, so the SIMD's matrix pipe and its vector ALU are busy at the same time instead of in turns:
//
//   workgroup = 16 waves = one per CU (LDS-bound), persistent: it walks a contiguous run of (clip, time tile, channel tile)s.
//   waves 8..15  PRODUCERS: global loads of the raw activation rows (three chunks in flight in registers), the fused prologue
//                (GroupNorm/FiLM affine + GELU, reference unet.py:280-285, 311-315) as scalar fp32 VALU, ds_write of MFMA operands.
//   waves 0..7   CONSUMERS: weight chunks by LDS-DMA (buffer_load ... lds, no registers), ds_read fragments, MFMAs; at the
//                end of a tile the statistics of the next GroupNorm and ONE rounding to the storage type into an LDS out-tile,
//                which they store as whole rows one step later (behind that step's weight wait).
//   One s_barrier per K chunk ("step"): during step g the consumers multiply chunk g (stage g & 1) while the producers stage
//   chunk g + 1 (stage (g + 1) & 1) -- across tile boundaries too, so there is no per-tile pipeline fill.
//   The identity skip (unet.py:316) is one more K segment whose B operand is the identity matrix: x * 1.0 accumulates exactly in
//   fp32, so the accumulator holds bias + conv + skip before its only rounding, and the epilogue needs no row-layout pass.
//
// LDS rows are 64 B (32 channels) with the 16-byte column XOR-swizzled by bits 2..3 of the row, for activations (written by
// ds_write_b128) and weights (lane-linear DMA image, swizzle applied to the source address) alike: ds_read_b128 conflict-free.
#include <atomic>
#include <cstddef>
#include <cstdlib>

#include "kernels.hpp"

namespace vqvs {

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

template <typename T> struct WsOp;
template <> struct WsOp<half_t> {
  typedef f16x8 v8;
  static constexpr unsigned short one = 0x3C00;
#ifdef VQVS_F16_GELU6
  static constexpr int gq = GELU_POLY6;  // (A/B: the bf16 mode's polynomial in the fp16 mode)
#else
  static constexpr int gq = GELU_POLY7;
#endif
};
template <> struct WsOp<bf16_t> {
  typedef bf16x8 v8;
  static constexpr unsigned short one = 0x3F80;
  static constexpr int gq = GELU_POLY6;
};
// fp32 storage (the parity mode): operands are split x = hi + lo in bf16 and every product is hi*hi + lo*hi + hi*lo in fp32
// accumulators (drops lo*lo, ~2^-18 relative); the prologue is the exact-erf GELU of common.hpp
template <> struct WsOp<float> {
  typedef bf16x8 v8;
  static constexpr unsigned short one = 0x3F80;
  static constexpr int gq = GELU_EXACT;
};
__device__ __forceinline__ f32x16 ws_mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 ws_mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// The prologue arithmetic is written on scalars on purpose: a packed fp32 instruction (v_pk_fma_f32) does not run beside
// another wave's MFMAs on the same SIMD, a plain one does (profiles/r02_ubench_role_split.txt).
// Eight elements at a time, Horner step by Horner step: left to itself the scheduler emits each element's dependent chain
// back to back (seven v_fmaak in a row, every one waiting for its predecessor's result); the scheduling barriers keep eight
// independent instructions between an instruction and its consumer, which is what a wave needs to issue VALU at full rate.
#ifndef VQVS_WS_GELU_IL
#define VQVS_WS_GELU_IL 1  // 0: the per-element form (A/B measurements)
#endif
template <int Q, int N, bool IL>
__device__ __forceinline__ void ws_gelu_n(float (&v)[N]) {
  // gelu(v) = v * Phi(v), Phi(v) = clamp01(0.5 + v * p(min(v^2, 16))): inside [-4, 4] this is the polynomial itself; outside, v * p(16)
  // = v / 8 carries the sum past [0, 1] and the clamp (a free output modifier of the v_fma) saturates Phi at 0 / 1 -- one v_min in
  // place of the v_med3 on v (VOP3 with two inline constants: twice the issue cost, tools/ubench/valu_mix.hip).
  float w[N], p[N];
#define WS_G8(expr)                    \
  _Pragma("unroll") for (int i = 0; i < N; ++i) { expr; } \
  if (VQVS_WS_GELU_IL && IL) __builtin_amdgcn_sched_barrier(0);
  WS_G8(w[i] = v[i] * v[i])
  WS_G8(w[i] = __builtin_fminf(w[i], 16.0f))
  if constexpr (Q == GELU_POLY6) {
    WS_G8(p[i] = fmaf(2.81608722e-08f, w[i], -1.89188380e-06f))
    WS_G8(p[i] = fmaf(p[i], w[i], 5.41903041e-05f))
    WS_G8(p[i] = fmaf(p[i], w[i], -8.78980255e-04f))
    WS_G8(p[i] = fmaf(p[i], w[i], 9.11294959e-03f))
    WS_G8(p[i] = fmaf(p[i], w[i], -6.53883549e-02f))
    WS_G8(p[i] = fmaf(p[i], w[i], 3.98526915e-01f))
  } else {
    WS_G8(p[i] = fmaf(-1.301278171e-09f, w[i], 1.041951057e-07f))
    WS_G8(p[i] = fmaf(p[i], w[i], -3.657111166e-06f))
    WS_G8(p[i] = fmaf(p[i], w[i], 7.485478930e-05f))
    WS_G8(p[i] = fmaf(p[i], w[i], -1.006488756e-03f))
    WS_G8(p[i] = fmaf(p[i], w[i], 9.505392772e-03f))
    WS_G8(p[i] = fmaf(p[i], w[i], -6.588783436e-02f))
    WS_G8(p[i] = fmaf(p[i], w[i], 3.986733897e-01f))
  }
  WS_G8(p[i] = __builtin_amdgcn_fmed3f(fmaf(v[i], p[i], 0.5f), 0.0f, 1.0f))
  WS_G8(v[i] = v[i] * p[i])
#undef WS_G8
}

// (per-element form, scheduling left to the compiler: the AVG instantiations, whose four raw rows per chunk set leave no
//  registers for eight pinned chains)
template <int Q>
__device__ __forceinline__ float ws_gelu(float v) {
  const float w = __builtin_fminf(v * v, 16.0f);
  float p;
  if constexpr (Q == GELU_POLY6) {
    p = fmaf(2.81608722e-08f, w, -1.89188380e-06f);
    p = fmaf(p, w, 5.41903041e-05f);
    p = fmaf(p, w, -8.78980255e-04f);
    p = fmaf(p, w, 9.11294959e-03f);
    p = fmaf(p, w, -6.53883549e-02f);
    p = fmaf(p, w, 3.98526915e-01f);
  } else {
    p = fmaf(-1.301278171e-09f, w, 1.041951057e-07f);
    p = fmaf(p, w, -3.657111166e-06f);
    p = fmaf(p, w, 7.485478930e-05f);
    p = fmaf(p, w, -1.006488756e-03f);
    p = fmaf(p, w, 9.505392772e-03f);
    p = fmaf(p, w, -6.588783436e-02f);
    p = fmaf(p, w, 3.986733897e-01f);
  }
  return v * __builtin_amdgcn_fmed3f(fmaf(v, p, 0.5f), 0.0f, 1.0f);
}

// A K segment of the launch (kernels.hpp SegDesc, plus the identity-skip pseudo-segment), as the kernel wants it.  The waves keep
// the segment they are in -- and, prefetched, the one that follows -- in scalar registers; the per-chunk quantities are those
// plus a multiple of the chunk index, so a step costs a handful of scalar adds and no argument loads.
struct WsSeg {
  const void* src;   // [B][Lsrc][Csrc] of T
  const float2* ss;  // (scale, shift) rows [B][ss_stride], or nullptr = raw (no prologue)
  int Csrc;
  int clip_bytes;    // Lsrc * Csrc * sizeof(T): bytes of one clip's rows (< 2^29)
  int c0;            // first source channel (identity segment: relative to the output channel tile)
  int nch;           // chunks of 32 channels
  int ntaps;         // 3, 1, or 0 = identity segment (weights = the identity block, written by the producers)
  int dil;           // dilation of a 3-tap segment, else 0: LDS row 0 holds time t0 - dil
  int rsz;           // RESIZE_*: staged row t is source row t (NONE), t >> 1 (UP2: nearest, unet.py:304-305), or the mean of source
                     // rows 2t and 2t + 1 AFTER the prologue (AVG2: avg_pool1d(2), unet.py:300-303)
  int ss_lds;        // byte offset of this segment's (scale, shift) pairs in the per-clip LDS table
  // (everything above is what a producer wave prefetches per segment: 48 contiguous bytes = one s_load_dwordx8 + one s_load_dwordx4)
  int ss_stride, ss_c0;
  int wbase, wstep;  // byte offset of chunk 0's packed weights [tap][Cout][32], bytes per chunk
  int lds_off;       // resident-weights form: byte offset of chunk 0's image in the resident block
};
// GroupNorm (+FiLM) coefficients built inside the convolution that reads them (kernels.hpp ConvArgs.gn, GnArgs) -- by the CONSUMER
// waves, at the first step of a clip (their accumulators are dead there) for a clip that the producers will reach ss_ring / 2 clips
// later: every thread takes one channel of the concatenated prologue input, adds its tile partials in tile order (fp64), the
// channels of a group are combined by an xor butterfly over `cpg` lanes (a power of two, groups never straddle a wave), and the
// (scale, shift) pair goes straight into the LDS table the prologue reads.  Same formulas as gn_prepare_kernel (misc_kernels.hip; reference unet.py:311-314, 345-349).
struct WsGn {
  const float *part0, *part1;  // [B][ntiles][C][2] of the (one or two) sources
  int ntiles0, ntiles1, C0, C1;
  int nsrc, Ctot, cpg, nclips;
  int tpc;  // lanes per channel (1, 2, 4 or 8): the tile partials of a channel are summed in `tpc` interleaved slices
  double inv_count;
  const float *gamma, *beta, *film;
  int film_stride, film_off;
  unsigned* status;
  int guard;
};
struct WsArgs {
  WsGn gn;  // gn.nsrc == 0: the (scale, shift) rows come from seg[].ss
  WsSeg seg[4];
  int nseg, nchunks;  // segments (incl. the identity one), chunks per tile (>= 2)
  const void* w;
  const void* w_lo;  // fp32 storage: the lo plane of the split weights (same packing as w)
  int w_bytes;
  const float* bias;
  const void* skip;  // fp32 storage: identity-skip source, added in the epilogue (exact) instead of as a K segment
  int skip_L, skip_rsz;
  void* out;
  float* stats;
  int Cout, Lout, TTO, ntx, nty, ntiles, ntiles_stat;
  int wres_bytes;  // resident-weights form: bytes of all segments' weights
  int ss_bytes;    // bytes of one clip's (scale, shift) table (all prologue segments), ss_ring copies of it live in LDS
  int ss_ring;     // 2, 4 or 8 (power of two): clips whose chunks can be in flight at once
  int rev;         // 1: the launch walks its tiles from the last to the first (ConvArgs.rev)
  int zp;          // 1: whole-clip tiles (template flag ZP): a clip of at most 255 rows is ONE tile whatever the dilation
};
static_assert(offsetof(WsArgs, gn) == 0, "gn_table reads WsGn through the kernarg segment pointer: it must stay the first member");
#define WS_SEGF(s, f) ((s) == 0 ? a.seg[0].f : ((s) == 1 ? a.seg[1].f : ((s) == 2 ? a.seg[2].f : a.seg[3].f)))

struct TileCo {
  int b, tx, ty;
};

#ifndef VQVS_WS_EXP
#define VQVS_WS_EXP 0  // ablation bits for tools/experiments (results are WRONG when non-zero): 1 no activation loads, 2 no weight DMA,
#endif                // 4 no tile store, 8 no prologue arithmetic, 16 no MFMA loop, 32 no tile-end statistics / rounding, 64 constant
                      // (scale, shift), 128 no zero masks, 256 store at the step's start, 1024 idle consumers, 2048 trivial load cursor, 4096 fragment reads without MFMAs, 32768 no barriers inside the step loop,
                      // 131072 every second step barrier skipped (what would two K chunks per barrier be worth?)

#ifdef VQVS_TIMING
__device__ unsigned long long g_ws_timing[32];
#define WS_TMARK(i)                                                \
  {                                                                \
    const unsigned long long _t = __builtin_amdgcn_s_memtime();    \
    tacc[i] += _t - tlast;                                         \
    tlast = _t;                                                    \
  }
#else
#define WS_TMARK(i)
#endif

// RES: the launch's weights fit into LDS next to everything else (one output channel tile per launch, K small): they are fetched
// once per workgroup instead of once per tile and step -- the per-CU path from L2 is the scarce resource (~11 B / cycle / CU,
// whether the bytes come from HBM or from L2), and a re-streamed 32-channel weight chunk (3 * CT * 64 B) weighs more on it than
// the activation chunk it multiplies (16 KiB).
// AVG: the launch has an avg-pooled segment (down-sampling blocks, unet.py:297-303): EVERY chunk then issues four loads per
// thread (two source rows per staged row; the other segments' second pair is an out-of-range dummy that never reaches memory),
// so that "the oldest chunk has landed" stays one counted wait.
// Geometry (template): ROWS staged rows per tile (256: one workgroup of 16 waves per CU; 128: 8 waves, two workgroups per CU whose
// barriers, bookkeeping and tile epilogues interleave), CT output channels per tile (32 / 64 / 128).  ROWS / 32 producer waves (a
// thread stages two rows x eight channels of every chunk) and as many consumer waves, each owning MT x 32 rows by WN x 32 channels.
// fp32 storage comes in two geometries, both with 16 waves: 256 rows x 64 channels, and TALL = 128 rows x 128 channels for the
// levels from 128 output channels up -- the same 32 accumulator registers per consumer wave, but every row is loaded, transformed
// and split ONCE per 128 output channels instead of once per 64 (a producer thread then stages one row of a chunk, not two).
template <typename T, int ROWS>
constexpr int ws_threads() { return sizeof(T) == 4 ? 1024 : ROWS * 4; }
// ZP ("zero padded"): the whole clip is ONE tile.  A tile normally stages 256 rows of which 256 - 2 * dilation are output rows, so the
// middle blocks' dilations 4 .. 32 (unet.py:21) turn a 250-row clip into TWO tiles of twice the work.  Where the clip has at most 255
// rows, every halo row is the convolution's zero padding: LDS row r then holds time r for every segment, rows from Lout on are staged
// as zeros anyway, and a tap whose row falls outside [0, 256) reads row 255 -- a row of zeros -- instead (per-lane A addresses for
// both 32-row blocks, rebuilt only when the dilation changes).  One tile per (clip, channel tile): the statistics of the tiles the
// static schedule expects beyond the first are written as zeros.
template <typename T, int ROWS, int CT, bool RES, bool AVG, bool ZP = false>
__global__ __launch_bounds__((ws_threads<T, ROWS>())) void conv_ws_kernel(const WsArgs a) {
  constexpr bool X3 = sizeof(T) == 4;     // fp32 storage: hi / lo bf16 planes of activations and weights, three MFMAs per product
  constexpr bool TALL = X3 && ROWS == 128;
  constexpr int WN = (CT == 128 && !TALL) ? 2 : 1;  // 32-channel MFMA tiles per consumer wave
  constexpr int MT = CT == 32 ? 1 : 2;    // 32-row MFMA tiles per consumer wave
  constexpr int RW = 32 * MT;             // rows per consumer wave
  constexpr int NWT = ROWS / RW;          // consumer waves along time ...
  constexpr int NWC = CT / (WN * 32);     // ... and along channels
  constexpr int NCW = NWT * NWC;          // consumer waves = producer waves (ROWS / 32; 8 in the TALL geometry)
  constexpr int NPT = NCW * 64;           // producer threads (= consumer threads)
  constexpr int PR = ROWS * 4 / NPT;      // rows a producer thread stages per chunk (2; TALL: 1)
  constexpr int HR = ROWS / 2;            // PR == 2: distance between the two rows a producer thread stages
  constexpr bool DB = CT == 32;           // a tile may be ONE chunk (32 x 3 -> 32): out-tile and statistics are double-buffered
  constexpr int ES = (int)sizeof(T);
  constexpr bool L4 = AVG || (X3 && !TALL);  // four loads per thread and chunk (two source rows per staged row, or two rows of 32-byte octets)
  static_assert(!(AVG && X3), "avg-pooled segments are not built for fp32 storage");
  static_assert(!X3 || (CT == 64 && ROWS == 256) || (CT == 128 && ROWS == 128), "fp32 storage: 256 x 64 or 128 x 128 tiles");
  static_assert(PR == 1 || PR == 2, "producer rows per chunk");
  static_assert(!TALL || !RES, "the 128 x 128 fp32 tile streams its weights (two planes of 3 x 128 x 64 B per chunk)");
  static_assert(!ZP || (!X3 && !RES && !AVG && ROWS == 256 && CT >= 64), "whole-clip tiles: 2-byte storage, streaming 256-row tiles of 64 / 128 channels");
  constexpr int ACT_BYTES = ROWS * 64;    // staged rows x 32 channels (one bf16 / fp16 plane)
  constexpr int W_BYTES = 3 * CT * 64;    // ... of weights
  constexpr int ACT2 = (X3 ? 2 : 1) * ACT_BYTES, W2 = (X3 ? 2 : 1) * W_BYTES;  // [hi plane][lo plane]
  constexpr int STAGE = ACT2 + W2;
  constexpr int OP = CT * 4 + 16;         // out-tile pitch in bytes: one LDS row = one PAIR of output rows, a dword per channel (lo = even row)
  // WPE: the 128-channel tile with resident weights has no room for a whole out-tile (66 KiB next to 96 KiB of weights): every
  // consumer wave rounds, transposes and stores its own 64 x 64 part through a private 2 KiB region, 16 rows at a time.
  constexpr bool WPE = RES && WN == 2 && !X3;
  // streaming form: [stage 0: activations | weights][stage 1][out-tile][statistics]
  // resident form : [activations 0][activations 1][out-tile, or 8 private regions][statistics][all weights]
  constexpr int ACT_STRIDE = RES ? ACT2 : STAGE;  // activation slot s at s * ACT_STRIDE
  constexpr int WS_OFF = ACT2;                    // streaming form: weight slot s at WS_OFF + s * WS_STRIDE
  constexpr int WS_STRIDE = STAGE;
  constexpr int O_OFF = RES ? 2 * ACT2 : 2 * STAGE;
  constexpr int O_BYTES = X3 ? 0 : (WPE ? NCW * 2048 : (ROWS / 2) * OP);  // (fp32 storage: rows are stored from the accumulators)
  constexpr int R_BYTES = NWT * CT * 8;                  // [NWT time slices][CT][2] partial statistics
  constexpr int R_OFF = O_OFF + (DB ? 2 : 1) * O_BYTES;
  constexpr int C_OFF = R_OFF + (DB ? 2 : 1) * R_BYTES;  // resident form: the 32 x 32 identity block (2 KiB), then 2 KiB of zeros (identity-skip chunks)
  constexpr int WRES_OFF = C_OFF + (RES ? 4096 : 0);
  const int SS_OFF = WRES_OFF + (RES ? (X3 ? 2 : 1) * a.wres_bytes : 0);  // per-clip (scale, shift) tables, a.ss_ring of them
  constexpr int GQ = WsOp<T>::gq;
  typedef typename WsOp<T>::v8 V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef VQVS_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long t_sh0 = tlast, t_rt0 = __builtin_amdgcn_s_memrealtime();
#endif

  // Tile range of this workgroup.  Workgroup w runs on XCD w % 8 (observed, used for speed only): every XCD gets one contiguous
  // run of tiles (channel tile fastest, then time, then clip), every workgroup a contiguous piece of it -- halo rows and the
  // input rows shared by channel tiles are re-read through the same L2.
  int tb, te;
  {
    const int nwg = (int)gridDim.x, w = (int)blockIdx.x;
    const int nx = nwg < 8 ? nwg : 8;
    const int k = w % nx, i = w / nx;
    const int nk = (nwg - k + nx - 1) / nx;
    const unsigned tot = (unsigned)a.ntiles;  // (host side: ntiles < 2^24, so the products below fit 32 bits)
    const unsigned s = tot * (unsigned)k / (unsigned)nx, e = tot * (unsigned)(k + 1) / (unsigned)nx;
    const unsigned len = e - s;
    tb = (int)(s + len * (unsigned)i / (unsigned)nk);
    te = (int)(s + len * (unsigned)(i + 1) / (unsigned)nk);
  }
  const int n = a.nchunks;
  const int Q = (te - tb) * n;  // steps of this workgroup
  if (Q <= 0) return;
  // A reversed launch (a.rev) walks the same ranges from the far end: consecutive launches alternate direction, so each one
  // starts on the rows its predecessor touched last -- the part of the tensor that is still in the 256 MB Infinity Cache.
  TileCo first;
  {
    const int ft = a.rev ? a.ntiles - 1 - tb : tb;
    const int rest = ft / a.nty;
    first.ty = ft - rest * a.nty;
    first.b = rest / a.ntx;
    first.tx = rest - first.b * a.ntx;
  }
  auto next_tile = [&](TileCo& t) {
    if (a.rev) {
      if (--t.ty < 0) {
        t.ty = a.nty - 1;
        if (--t.tx < 0) {
          t.tx = a.ntx - 1;
          --t.b;
        }
      }
    } else if (++t.ty == a.nty) {
      t.ty = 0;
      if (++t.tx == a.ntx) {
        t.tx = 0;
        ++t.b;
      }
    }
  };
  bool skipb = false;  // (ablation bit 131072: the next step barrier is left out)
  auto sync_lds = [&]() {  // LDS writes / reads of this wave are done; global loads stay in flight across the barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(VQVS_WS_EXP & 32768) && !((VQVS_WS_EXP & 131072) && skipb)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (wave >= NCW) {
    // =============================== producers ===============================
    const int pt = tid - NPT;
    const int oct = pt & 3, r0 = pt >> 2;  // this thread stages rows r0 and r0 + HR, channel octet `oct`, of every chunk
    const int dst0 = r0 * 64 + ((oct ^ ((r0 >> 2) & 3)) << 4);
    struct Raw {
      u32x4 a0, a1;
      u32x4 b0, b1;   // AVG: the second source row of each staged row; fp32 storage: channels 4..7 of the octet
      unsigned meta;  // bit 0: prologue, bit 1 / 2: row 0 / 1 inside the clip (outside: the convolution's zero padding),
                      // bit 4: avg-pooled segment, bit 5: some staged row of this tile lies outside the clip
                      // (wave-uniform: only then are the zero masks applied)
      int ssaddr;     // LDS byte address of this thread's eight (scale, shift) pairs
    };
    Raw R0, R1, R2;
    struct Prep {  // everything the two loads of a chunk need
      i32x4 rs;
      int off0, off1, ssaddr;
      int db;  // AVG: byte distance to the second source row (avg-pooled segment), or 2^30 = out of range (any other segment)
      unsigned meta;
    };
    // load cursor: tile `lt`, segment `lseg`, chunk `lch` of the next loads; `cur` = their parameters (updated incrementally inside
    // a segment, rebuilt from `nx` = the prefetched fields of the segment that follows when one is entered)
    TileCo lt = first;
    int lseg = 0, lch = 0, lnch = 0, issued = 0;
    struct SegF {
      const void* src;
      const float2* ss;
      int Csrc, clip_bytes, c0, nch, ntaps, dil, rsz, ss_lds;
    };
    auto fetch = [&](int sg) -> SegF {  // (one batch of scalar loads from the argument block, issued a segment ahead of its use)
      const WsSeg& g = a.seg[sg];
      return SegF{g.src, g.ss, g.Csrc, g.clip_bytes, g.c0, g.nch, g.ntaps, g.dil, g.rsz, g.ss_lds};
    };
    SegF nx = fetch(0);
    Prep cur;
    int cur_xf = 0, cur_avg = 0, cur_edge = 0;  // (wave-uniform parts of meta, kept scalar)
    int ss_clip = -1;  // clip whose (scale, shift) table was written last
    unsigned cur_valid = 0;
    bool ss_defer = true;  // start-up: the first clip's table is copied AFTER the first three chunk loads are on their way (below)
    auto refresh_ss = [&](int clip) {
      // first chunk of a new clip: its (scale, shift) rows (every prologue segment's) go to LDS once -- the producers then read
      // them with ds_read instead of four more global loads per chunk and thread.  Slot b % ring: chunks of at most `ring`
      // clips are in flight (host: ring = 4 when a clip can take fewer than four steps).
      char* const tab = smem + SS_OFF + (clip & (a.ss_ring - 1)) * a.ss_bytes;
      if (a.gn.nsrc > 0) return;  // (the consumers build the tables of a launch with a fused GroupNorm)
      for (int sg = 0; sg < a.nseg; ++sg) {
        const WsSeg& g = a.seg[sg];
        if (g.ss == nullptr) continue;
        const char* const row = reinterpret_cast<const char*>(g.ss + (size_t)((unsigned)clip * (unsigned)g.ss_stride + (unsigned)g.ss_c0));
        for (int o = pt * 16; o < g.nch * 256; o += NPT * 16) *reinterpret_cast<f32x4*>(tab + g.ss_lds + o) = *reinterpret_cast<const f32x4*>(row + o);
      }
    };
    auto enter = [&]() {  // the cursor has just moved to (lt, lseg, chunk 0)
      const SegF f = nx;
      const int L = a.Lout;
      const unsigned long long clip = reinterpret_cast<unsigned long long>(f.src) + (unsigned long long)(unsigned)lt.b * (unsigned)f.clip_bytes;
      // raw buffer descriptor of this clip's rows: out-of-range rows (before / after the clip) read as zero
      cur.rs[0] = (int)(unsigned)clip;
      cur.rs[1] = (int)((unsigned)(clip >> 32) & 0xffffu);
      cur.rs[2] = (VQVS_WS_EXP & 2048) ? 0x7fffffff : f.clip_bytes;
      cur.rs[3] = 0x00020000;
      const int sdil = ZP ? 0 : f.dil;  // (ZP: LDS row r holds time r)
      const int tm0 = lt.tx * a.TTO - sdil + r0, tm1 = tm0 + HR;  // time of this thread's two rows
      const int sr0 = f.rsz == RESIZE_UP2 ? (tm0 >> 1) : (f.rsz == RESIZE_AVG2 ? 2 * tm0 : tm0);  // (first) source row of row 0
      cur.off0 = (sr0 * f.Csrc + f.c0 + (f.ntaps == 0 ? lt.ty * CT : 0) + oct * 8) * ES;
      cur.off1 = cur.off0 + (f.rsz == RESIZE_UP2 ? HR : (f.rsz == RESIZE_AVG2 ? 4 * HR : 2 * HR)) * f.Csrc * (ES / 2);
      cur.db = X3 ? 16 : (f.rsz == RESIZE_AVG2 ? 2 * f.Csrc : 0x40000000);
      cur_avg = f.rsz == RESIZE_AVG2 ? 1 : 0;
      if (__builtin_expect(lseg == 0 && lt.b != ss_clip, 0)) {
        ss_clip = lt.b;
        if (!ss_defer) refresh_ss(lt.b);
      }
      cur.ssaddr = SS_OFF + (lt.b & (a.ss_ring - 1)) * a.ss_bytes + f.ss_lds + oct * 64;
      cur_xf = f.ss != nullptr ? 1 : 0;
      cur_valid = ((tm0 >= 0 && tm0 < L) ? 2u : 0u) | ((tm1 >= 0 && tm1 < L) ? 4u : 0u);
      cur_edge = (lt.tx * a.TTO - sdil < 0 || lt.tx * a.TTO - sdil + ROWS > L) ? 1 : 0;
      lnch = f.nch;
      nx = fetch(lseg + 1 == a.nseg ? 0 : lseg + 1);
    };
    enter();
    ss_defer = false;
    auto prepare = [&]() -> Prep {
      Prep pr = cur;
      pr.meta = cur_valid | (unsigned)(cur_xf | (cur_avg << 4) | (cur_edge << 5));
      if (VQVS_WS_EXP & 2048) {  // ablation (valid for a single 64-channel segment only): the cheapest possible cursor
        ++issued;
        const int adv = (issued & 1) ? 64 : a.TTO * 128 - 64;
        cur.off0 += adv;
        cur.off1 += adv;
        cur.ssaddr ^= 256;
        return pr;
      }
      if (++issued < Q) {  // (past the end the last chunk is simply loaded again and never staged)
        if (++lch == lnch) {
          lch = 0;
          if (++lseg == a.nseg) {
            lseg = 0;
            next_tile(lt);
          }
          enter();
        } else {
          cur.off0 += 32 * ES;
          cur.off1 += 32 * ES;
          cur.ssaddr += 256;
        }
      }
      return pr;
    };
    // The loads are inline assembly on purpose: hipcc's own wait insertion drains the whole queue (vmcnt(0)) at the top of the
    // rotating loop, which would serialise every third step behind loads issued a moment earlier.  Each chunk issues exactly two
    // loads (this thread's two activation rows; four with AVG), so "this chunk has landed, the two younger ones may still be in
    // flight" is vmcnt(4) (vmcnt(8)), stated in acquire().
    auto issue = [&](Raw& r, const Prep& pr) {
      r.meta = pr.meta;
      r.ssaddr = pr.ssaddr;
      if (VQVS_WS_EXP & 1) {
        asm volatile("" : "=v"(r.a0), "=v"(r.a1));  // (opaque garbage, so that nothing downstream folds away)
        if constexpr (L4 || TALL) asm volatile("" : "=v"(r.b0), "=v"(r.b1));
        return;
      }
      // (the descriptor is wave-uniform by construction; say so, or the compiler may hand the assembly a VGPR copy of it when it
      //  runs short of SGPRs.  The s_nop covers the VALU-write -> VMEM-read wait states.)
      i32x4 rs;
#pragma unroll
      for (int i = 0; i < 4; ++i) rs[i] = __builtin_amdgcn_readfirstlane(pr.rs[i]);
      if constexpr (TALL) {
        const int ob0 = pr.off0 + 16;
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dwordx4 %0, %2, %4, 0 offen\n\t"
            "buffer_load_dwordx4 %1, %3, %4, 0 offen"
            : "=&v"(r.a0), "=&v"(r.b0)
            : "v"(pr.off0), "v"(ob0), "s"(rs));
      } else if constexpr (L4) {
        const int ob0 = pr.off0 + pr.db, ob1 = pr.off1 + pr.db;
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dwordx4 %0, %4, %8, 0 offen\n\t"
            "buffer_load_dwordx4 %1, %5, %8, 0 offen\n\t"
            "buffer_load_dwordx4 %2, %6, %8, 0 offen\n\t"
            "buffer_load_dwordx4 %3, %7, %8, 0 offen"
            : "=&v"(r.a0), "=&v"(r.b0), "=&v"(r.a1), "=&v"(r.b1)
            : "v"(pr.off0), "v"(ob0), "v"(pr.off1), "v"(ob1), "s"(rs));
      } else {
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dwordx4 %0, %2, %4, 0 offen\n\t"
            "buffer_load_dwordx4 %1, %3, %4, 0 offen"
            : "=&v"(r.a0), "=&v"(r.a1)
            : "v"(pr.off0), "v"(pr.off1), "s"(rs));
      }
    };
    auto acquire = [&](Raw& r) {  // the oldest chunk in flight has landed; nothing that reads it may be scheduled above this
      if (VQVS_WS_EXP & 1) return;
      if constexpr (L4)
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(r.a0), "+v"(r.a1), "+v"(r.b0), "+v"(r.b1));
      else if constexpr (TALL)
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(r.a0), "+v"(r.b0));
      else
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(r.a0), "+v"(r.a1));
    };
    auto affine8 = [&](u32x4 raw, const f32x4& s0, const f32x4& s1, const f32x4& s2, const f32x4& s3, float (&u)[8]) {
      const V8 h = __builtin_bit_cast(V8, raw);
      u[0] = fmaf((float)h[0], s0[0], s0[1]);
      u[1] = fmaf((float)h[1], s0[2], s0[3]);
      u[2] = fmaf((float)h[2], s1[0], s1[1]);
      u[3] = fmaf((float)h[3], s1[2], s1[3]);
      u[4] = fmaf((float)h[4], s2[0], s2[1]);
      u[5] = fmaf((float)h[5], s2[2], s2[3]);
      u[6] = fmaf((float)h[6], s3[0], s3[1]);
      u[7] = fmaf((float)h[7], s3[2], s3[3]);
      if (VQVS_WS_GELU_IL) __builtin_amdgcn_sched_barrier(0);
    };
    auto xform8_plain = [&](u32x4 raw, const f32x4& s0, const f32x4& s1, const f32x4& s2, const f32x4& s3) -> u32x4 {
      const V8 h = __builtin_bit_cast(V8, raw);
      V8 o;
      o[0] = (T)ws_gelu<GQ>(fmaf((float)h[0], s0[0], s0[1]));
      o[1] = (T)ws_gelu<GQ>(fmaf((float)h[1], s0[2], s0[3]));
      o[2] = (T)ws_gelu<GQ>(fmaf((float)h[2], s1[0], s1[1]));
      o[3] = (T)ws_gelu<GQ>(fmaf((float)h[3], s1[2], s1[3]));
      o[4] = (T)ws_gelu<GQ>(fmaf((float)h[4], s2[0], s2[1]));
      o[5] = (T)ws_gelu<GQ>(fmaf((float)h[5], s2[2], s2[3]));
      o[6] = (T)ws_gelu<GQ>(fmaf((float)h[6], s3[0], s3[1]));
      o[7] = (T)ws_gelu<GQ>(fmaf((float)h[7], s3[2], s3[3]));
      return __builtin_bit_cast(u32x4, o);
    };
    // avg-pooled rows: mean of the two TRANSFORMED source rows, one rounding (the order of conv_mfma.hip's avg path)
    auto xform8_avg = [&](u32x4 ra, u32x4 rb, const f32x4& s0, const f32x4& s1, const f32x4& s2, const f32x4& s3) -> u32x4 {
      const V8 h = __builtin_bit_cast(V8, ra), k = __builtin_bit_cast(V8, rb);
      V8 o;
#define WS_AVG1(i, sc, sh) o[i] = (T)((ws_gelu<GQ>(fmaf((float)h[i], sc, sh)) + ws_gelu<GQ>(fmaf((float)k[i], sc, sh))) * 0.5f);
      WS_AVG1(0, s0[0], s0[1]) WS_AVG1(1, s0[2], s0[3]) WS_AVG1(2, s1[0], s1[1]) WS_AVG1(3, s1[2], s1[3])
      WS_AVG1(4, s2[0], s2[1]) WS_AVG1(5, s2[2], s2[3]) WS_AVG1(6, s3[0], s3[1]) WS_AVG1(7, s3[2], s3[3])
#undef WS_AVG1
      return __builtin_bit_cast(u32x4, o);
    };
    auto xform8 = [&](u32x4 raw, const f32x4& s0, const f32x4& s1, const f32x4& s2, const f32x4& s3) -> u32x4 {
      if constexpr (AVG) {
        return xform8_plain(raw, s0, s1, s2, s3);
      } else {
        float u[8];
        affine8(raw, s0, s1, s2, s3, u);
        ws_gelu_n<GQ, 8, true>(u);
        V8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (T)u[i];
        return __builtin_bit_cast(u32x4, o);
      }
    };
    auto raw8_avg = [&](u32x4 ra, u32x4 rb) -> u32x4 {
      const V8 h = __builtin_bit_cast(V8, ra), k = __builtin_bit_cast(V8, rb);
      V8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (T)(((float)h[i] + (float)k[i]) * 0.5f);
      return __builtin_bit_cast(u32x4, o);
    };
    auto stage = [&](const Raw& r, int slot) {
      char* const sb = smem + slot * ACT_STRIDE;
      const int um = __builtin_amdgcn_readfirstlane((int)r.meta);
      if constexpr (X3) {
        // fp32 rows: exact-erf GELU of the affine (common.hpp gelu_f), then the bf16 hi / lo planes of the result
        const f32x4 fa0 = __builtin_bit_cast(f32x4, r.a0), fb0 = __builtin_bit_cast(f32x4, r.b0);
        f32x8 v0 = {fa0[0], fa0[1], fa0[2], fa0[3], fb0[0], fb0[1], fb0[2], fb0[3]};
        f32x8 v1 = v0;  // (PR == 1: unused)
        if constexpr (PR == 2) {
          const f32x4 fa1 = __builtin_bit_cast(f32x4, r.a1), fb1 = __builtin_bit_cast(f32x4, r.b1);
          v1 = f32x8{fa1[0], fa1[1], fa1[2], fa1[3], fb1[0], fb1[1], fb1[2], fb1[3]};
        }
        if ((um & 1) && !(VQVS_WS_EXP & 8)) {
          const f32x4* const sp = reinterpret_cast<const f32x4*>(smem + r.ssaddr);
          const f32x4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
          const float sc[8] = {s0[0], s0[2], s1[0], s1[2], s2[0], s2[2], s3[0], s3[2]};
          const float sh[8] = {s0[1], s0[3], s1[1], s1[3], s2[1], s2[3], s3[1], s3[3]};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v0[e] = gelu_f(fmaf(v0[e], sc[e], sh[e]));
            if constexpr (PR == 2) v1[e] = gelu_f(fmaf(v1[e], sc[e], sh[e]));
          }
        }
        if (um & 32) {  // (edge tiles: rows outside the clip are the convolution's zero padding)
          const float m0 = (r.meta & 2u) ? 1.f : 0.f, m1 = (r.meta & 4u) ? 1.f : 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v0[e] *= m0;
            if constexpr (PR == 2) v1[e] *= m1;
          }
        }
        bf16x8 h0, l0;
        split_bf16(v0, h0, l0);
        *reinterpret_cast<bf16x8*>(sb + dst0) = h0;
        *reinterpret_cast<bf16x8*>(sb + ACT_BYTES + dst0) = l0;
        if constexpr (PR == 2) {
          bf16x8 h1, l1;
          split_bf16(v1, h1, l1);
          *reinterpret_cast<bf16x8*>(sb + dst0 + HR * 64) = h1;
          *reinterpret_cast<bf16x8*>(sb + ACT_BYTES + dst0 + HR * 64) = l1;
        }
      } else {
      u32x4 o0 = r.a0, o1 = r.a1;
      if ((um & 1) && !(VQVS_WS_EXP & 8)) {
        const f32x4* const sp = reinterpret_cast<const f32x4*>(smem + ((VQVS_WS_EXP & 64) ? 0 : r.ssaddr));
        f32x4 s0, s1, s2, s3;
        if (VQVS_WS_EXP & 64) {
          s0 = f32x4{1.0f, 0.1f, 0.9f, -0.1f}; s1 = f32x4{1.1f, 0.2f, 1.0f, -0.2f}; s2 = f32x4{0.7f, 0.07f, 0.63f, -0.07f}; s3 = f32x4{1.43f, 0.26f, 1.3f, -0.26f};
        } else {
          s0 = sp[0]; s1 = sp[1]; s2 = sp[2]; s3 = sp[3];
        }
        if (AVG && (um & 16)) {
          o0 = xform8_avg(r.a0, r.b0, s0, s1, s2, s3);
          o1 = xform8_avg(r.a1, r.b1, s0, s1, s2, s3);
        } else {
          o0 = xform8(r.a0, s0, s1, s2, s3);
          o1 = xform8(r.a1, s0, s1, s2, s3);
        }
      } else if (AVG && (um & 16)) {
        o0 = raw8_avg(r.a0, r.b0);
        o1 = raw8_avg(r.a1, r.b1);
      }
      if ((um & 32) && !(VQVS_WS_EXP & 128)) {
        // rows outside the clip are the convolution's zero padding (edge tiles only): AND with a 0 / ~0 lane mask (v_bfe_i32 + v_and_b32; a
        // v_cndmask_b32 on VCC measured ~7x the issue cost of a v_and_b32, tools/ubench/valu_mix.hip)
        const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe((int)r.meta, 1, 1), m1 = (unsigned)__builtin_amdgcn_sbfe((int)r.meta, 2, 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] &= m0;
          o1[e] &= m1;
        }
      }
      *reinterpret_cast<u32x4*>(sb + dst0) = o0;
      *reinterpret_cast<u32x4*>(sb + dst0 + HR * 64) = o1;
      }
    };

    {
      const Prep p0 = prepare();
      issue(R0, p0);
      const Prep p1 = prepare();
      issue(R1, p1);
      const Prep p2 = prepare();
      issue(R2, p2);
      // The first clip's (scale, shift) rows: a round trip to global memory that used to complete (load -> LDS) before the first
      // activation load was even issued; behind the three chunk loads both latencies pass together.  (The cursor may have entered
      // the next clip during the three prepare() calls: that table was copied there, ss_defer being off after the first enter.)
      refresh_ss(first.b);
    }
    // The first clip's (scale, shift) table -- copied above, or built by the consumer waves (fused GroupNorm) -- is visible from
    // here; the consumers match this barrier.  The first three chunk loads are already in flight behind it.
    sync_lds();
    int q = 0;
    WS_TMARK(2)
#ifndef VQVS_WS_PREP_EARLY
#define VQVS_WS_PREP_EARLY 1  // (0: the cursor runs behind the barrier, as through round 5 -- A/B)
#endif
#if VQVS_WS_PREP_EARLY
    // the parameters of the loads a step issues are computed at the END of the step before it, in front of the barrier -- scalar work
    // that would otherwise run right behind the barrier, beside the consumers' MFMA burst, now fills the wait for the other waves
    Prep prn = prepare();
#define WS_PBODY(R)              \
  {                              \
    WS_TMARK(2)                  \
    acquire(R);                  \
    WS_TMARK(0)                  \
    stage(R, q & 1);             \
    WS_TMARK(1)                  \
    issue(R, prn);               \
    prn = prepare();             \
    WS_TMARK(4)                  \
    skipb = (q & 1) != 0;        \
    sync_lds();                  \
    skipb = false;               \
    WS_TMARK(3)                  \
  }                              \
  if (++q == Q) break;
#else
#define WS_PBODY(R)              \
  {                              \
    const Prep pr = prepare();   \
    WS_TMARK(2)                  \
    acquire(R);                  \
    WS_TMARK(0)                  \
    stage(R, q & 1);             \
    WS_TMARK(1)                  \
    issue(R, pr);                \
    WS_TMARK(4)                  \
    sync_lds();                  \
    WS_TMARK(3)                  \
  }                              \
  if (++q == Q) break;
#endif
    for (;;) {
      WS_PBODY(R0)
      WS_PBODY(R1)
      WS_PBODY(R2)
    }
#undef WS_PBODY
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the over-issued loads of the tail
    sync_lds();  // the consumers' last step
#ifdef VQVS_TIMING
    if (lane == 0 && ((int)blockIdx.x & 15) == 3) {
      for (int i = 0; i < 5; ++i) atomicAdd(&g_ws_timing[i], tacc[i]);
      atomicAdd(&g_ws_timing[16], 1ull);
      atomicAdd(&g_ws_timing[18], (unsigned long long)Q);
    }
#endif
  } else {
    // =============================== consumers ===============================
    const int wt = wave % NWT, wc = wave / NWT;  // time slice (RW rows), channel half (WN x 32 channels)
    const int l31 = lane & 31, hh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(X3 ? a.w_lo : a.w), 0, a.w_bytes, 0x00020000);
    T* const outp = reinterpret_cast<T*>(a.out);

    f32x16 acc[MT][WN];
    // LDS byte offset of this lane's B (weight) fragment for k-step 0: rows tap * CT + wc * WN * 32 + nt * 32 + l31 -- the
    // swizzle depends on l31 only, taps and channel tiles are immediate offsets
    const int boff = (wc * (WN * 32) + l31) * 64 + ((hh ^ ((l31 >> 2) & 3)) << 4);
    // weight DMA: piece p = 16 rows of 64 B; this lane's row within a piece and its (source-side) swizzled octet
    const int dma_lane = ((lane >> 2) * 32 + (((lane & 3) ^ ((lane >> 4) & 3)) * 8)) * 2;

    // B operand of an identity-segment chunk (unet.py:316: out += skip): the 32 x 32 identity block (k = skip channel, n = output
    // channel, both relative to the chunk's 32 channels) for the MFMA tile that owns those output channels, a block of zeros for
    // every other one -- two constant 2 KiB images (rows of 32 k-values, swizzled like the weight rows).  Resident form: written
    // once, here; streaming form: into the chunk's (otherwise unused) weight slot, one step ahead like a weight DMA.
    auto write_consts = [&](int off) {
      if (tid < 128) {
        const int j = tid >> 2, oct = tid & 3, dlt = j - oct * 8;  // row j: element e of this octet is 1 iff e == dlt
        u32x4 bi;
#pragma unroll
        for (int e = 0; e < 4; ++e) bi[e] = (dlt == 2 * e ? (unsigned)WsOp<T>::one : 0u) | (dlt == 2 * e + 1 ? ((unsigned)WsOp<T>::one << 16) : 0u);
        *reinterpret_cast<u32x4*>(smem + off + j * 64 + ((oct ^ ((j >> 2) & 3)) << 4)) = bi;
      } else if (tid < 256) {
        *reinterpret_cast<u32x4*>(smem + off + 2048 + (tid - 128) * 16) = u32x4{0u, 0u, 0u, 0u};
      }
    };
    if constexpr (RES) write_consts(C_OFF);
    auto gn_table = [&](int b) {
      // (the descriptor is read through an opaque pointer: its scalar loads then stay inside this rare block instead of being
      //  hoisted out of the K loop, where they cost the kernel its last SGPRs)
      typedef const __attribute__((address_space(4))) WsGn* GnPtr;  // (WsArgs is the kernel's only parameter, WsGn its first member)
      GnPtr G = (GnPtr)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(G));
      if (b < 0 || b >= G->nclips) return;
      char* const tab = smem + SS_OFF + (b & (a.ss_ring - 1)) * a.ss_bytes;
      // a wave takes 64 / tpc channels, a lane one slice (tiles j, j + tpc, ...) of one channel; eight loads in flight per lane
      const int tpc = G->tpc, chw = 64 / tpc;
      const int cl = lane & (chw - 1), j = lane / chw;
      for (int cb = wave * chw; cb < G->Ctot; cb += NCW * chw) {  // (C0 is a multiple of 64: a wave stays inside one source)
        const int c = cb + cl;
        const bool second = c >= G->C0;
        const int ccl = second ? c - G->C0 : c;
        const int nt = second ? G->ntiles1 : G->ntiles0, Cs = second ? G->C1 : G->C0;
        const float* p = (second ? G->part1 : G->part0) + ((size_t)(unsigned)b * (unsigned)nt * (unsigned)Cs + (unsigned)ccl) * 2;
        // (the channel's affine parameters and FiLM entries are requested first: their latency passes behind the partial sums)
        const float gam = G->gamma[c], bet = G->beta[c];
        float ffa = 0.f, ffb = 0.f;
        if (G->film) {
          const float* f = G->film + (size_t)(unsigned)b * G->film_stride + G->film_off;
          ffa = f[c];
          ffb = f[G->Ctot + c];
        }
        double s1 = 0.0, s2 = 0.0;
        unsigned bad = 0;
        for (int t0 = j; t0 < nt; t0 += 8 * tpc) {
          float2 q[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int t = t0 + k * tpc;
            q[k] = t < nt ? *reinterpret_cast<const float2*>(p + (size_t)t * Cs * 2) : float2{0.f, 0.f};
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            s1 += (double)q[k].x;
            s2 += (double)q[k].y;
            if (!(fabsf(q[k].x) <= 3.0e38f) || !(q[k].y <= 3.0e38f)) bad |= 1u;  // (the range guard of gn_prepare_kernel)
            if (G->guard && q[k].y >= 9.0e8f) bad |= 2u;
          }
        }
        if (bad && G->status) atomicOr(G->status, bad);
        for (int off = chw; off < 64; off <<= 1) {  // the slices of a channel ...
          s1 += __shfl_xor(s1, off);
          s2 += __shfl_xor(s2, off);
        }
        for (int off = 1; off < G->cpg; off <<= 1) {  // ... then the channels of a group (cpg <= chw)
          s1 += __shfl_xor(s1, off);
          s2 += __shfl_xor(s2, off);
        }
        const double mean = s1 * G->inv_count;
        double var = s2 * G->inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + 1e-5);
        double scale = rstd * (double)gam;
        double shift = (double)bet - mean * scale;
        if (G->film) {
          const double fa = (double)ffa + 1.0;
          scale *= fa;
          shift = shift * fa + (double)ffb;
        }
        if (j == 0) *reinterpret_cast<float2*>(tab + c * 8) = float2{(float)scale, (float)shift};
      }
    };
    const int gn_ahead = a.gn.nsrc > 0 ? (a.ss_ring >> 1) * (a.rev ? -1 : 1) : 0;  // clips between a table's construction and its clip
    int gn_clip = -1;  // clip at whose first step the last table was built

    auto dma = [&](int ntaps, int woff, const TileCo& t, int slot) {  // weights of a chunk -> stage `slot`, 1 KiB (16 rows) per wave-instruction
      if (!RES && ntaps == 0) write_consts(WS_OFF + slot * WS_STRIDE);
      if (RES || ntaps == 0 || (VQVS_WS_EXP & 2)) return;
      const int wb = woff + t.ty * (CT * 64);
      const int np = ntaps * (CT / 16);
      for (int p = wave; p < np; p += NCW) {
        const int tap = (p * 16) / CT, colb = p * 16 - tap * CT;
        const int voff = wb + (tap * a.Cout + colb) * 64 + dma_lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(smem + WS_OFF + slot * WS_STRIDE + p * 1024), 16, voff, 0, 0, 0);
        if constexpr (X3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wl, (lds_ptr)(smem + WS_OFF + slot * WS_STRIDE + W_BYTES + p * 1024), 16, voff, 0, 0, 0);
      }
    };
    auto sync_all = [&]() {  // + this wave's weight DMA has landed (streaming form; with resident weights nothing waits for VMEM:
      if constexpr (RES)     //   the tile stores drain on their own, overlapping the producers' loads)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (!(VQVS_WS_EXP & 32768) && !((VQVS_WS_EXP & 131072) && skipb)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };

    // coordinates of the tile whose out-tile sits in LDS (stored at the start of the next step)
    TileCo pt_{0, 0, 0};
    bool pending = false;
    int obuf = 0, pbuf = 0;  // (DB) buffer the next epilogue writes / the pending tile sits in
    // store_tile(i0, i1): pieces [i0, i1) of the pending out-tile's NIT row-pair pieces leave for global memory; piece 0 carries the
    // tile statistics.  All pieces go at the end of the successor's FIRST step.  -DVQVS_WS_SPREAD=1 spreads them over the successor's
    // steps 0 .. n - 2 (all before the barrier in front of its last chunk, whose epilogue rewrites the out-tile), `st_per` pieces per
    // step: built and parity-clean in round 5 on the hypothesis that the store (~1-2 us of consumer work in one step) makes that
    // step wait -- measured 67.75 against 67.9 clips/s (same-box A/B): it does not; what the stores cost is not their burstiness
    // (DESIGN.md section 7, round 5).
#ifndef VQVS_WS_SPREAD
#define VQVS_WS_SPREAD 0
#endif
    constexpr int NIT = (X3 || WPE) ? 1 : (ROWS / 2) / (NPT / (CT / 8));
    const int st_per = (VQVS_WS_SPREAD && n > 1) ? (NIT + (n - 1) - 1) / (n - 1) : NIT;
    auto store_tile = [&](int i0, int i1) {
      int zl = 0;
      asm volatile("" : "+v"(zl));  // (keeps the per-lane address arithmetic of this rare block out of the loop's live registers)
      const int ltid = tid + zl;
      const int t0 = pt_.tx * a.TTO;
      const int nvalid = min(a.TTO, a.Lout - t0);
      const int co0 = pt_.ty * CT;
      (void)nvalid;
      if (i0 == 0 && a.stats != nullptr && ltid < CT) {
        const float2* const red = reinterpret_cast<const float2*>(smem + R_OFF + pbuf * R_BYTES);
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int g = 0; g < NWT; ++g) {  // fixed order: deterministic
          const float2 v = red[g * CT + ltid];
          t1 += v.x;
          t2 += v.y;
        }
        float2* o = reinterpret_cast<float2*>(a.stats) + ((size_t)pt_.b * a.ntiles_stat + pt_.tx) * a.Cout + co0 + ltid;
        *o = float2{t1, t2};
        if constexpr (ZP)  // (the clip's only tile: the further statistics tiles the schedule's GroupNorm sums hold nothing)
          for (int t = 1; t < a.ntiles_stat; ++t) o[(size_t)t * a.Cout] = float2{0.f, 0.f};
      }
      if constexpr (WPE || X3) return;  // (the rows left with their wave, at the tile's last step)
      // out-tile -> global: a thread takes 8 channels of a row pair (2 x 16 B of LDS), separates the two rows (lo / hi halves of
      // the dwords) and stores 16 B of each; a wave writes whole 2 x CT-byte rows
      constexpr int PPR = CT / 8;  // 8-channel pieces per row
      constexpr int PSTEP = NPT / PPR;  // row pairs between this thread's pieces
      const int p0 = ltid / PPR, col = ltid - p0 * PPR;
      const char* const lsrc = smem + O_OFF + pbuf * O_BYTES + p0 * OP + col * 32;
      // buffer stores through a descriptor that ends behind the tile's last valid row (the clip's end, or where the next tile's
      // rows begin): rows past it are dropped by the address check -- no per-row branches
      // (ablation bit 16384: every tile is stored over the clip's first rows -- the same store instructions, but the bytes stay in L2)
      const int t0s = (VQVS_WS_EXP & 16384) ? 0 : t0;
      const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(outp + (size_t)((VQVS_WS_EXP & 16384) ? 0 : pt_.b) * a.Lout * a.Cout, 0,
                                                                            (t0s + nvalid) * a.Cout * 2, 0x00020000);
      const int vo = ((t0s + 2 * p0) * a.Cout + co0 + col * 8) * 2;
      static_assert(X3 || WPE || NIT == (ROWS / 2) / PSTEP, "pieces per tile");
#pragma unroll
      for (int i = 0; i < (ROWS / 2) / PSTEP; ++i) {
        if (i < i0 || i >= i1) continue;  // (wave-uniform)
        const u32x4 lo = *reinterpret_cast<const u32x4*>(lsrc + i * PSTEP * OP);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(lsrc + i * PSTEP * OP + 16);
        u32x4 e, o;
        e[0] = __builtin_amdgcn_perm(lo[1], lo[0], 0x05040100u);
        e[1] = __builtin_amdgcn_perm(lo[3], lo[2], 0x05040100u);
        e[2] = __builtin_amdgcn_perm(hi[1], hi[0], 0x05040100u);
        e[3] = __builtin_amdgcn_perm(hi[3], hi[2], 0x05040100u);
        o[0] = __builtin_amdgcn_perm(lo[1], lo[0], 0x07060302u);
        o[1] = __builtin_amdgcn_perm(lo[3], lo[2], 0x07060302u);
        o[2] = __builtin_amdgcn_perm(hi[1], hi[0], 0x07060302u);
        o[3] = __builtin_amdgcn_perm(hi[3], hi[2], 0x07060302u);
        const int v = vo + (2 * i * PSTEP) * a.Cout * 2;
        __builtin_amdgcn_raw_buffer_store_b128(e, rs_o, v, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o, rs_o, v + a.Cout * 2, 0, 0);
      }
    };

    // chunk cursors: `ct` / `cci` = tile and chunk of the current step; the DMA cursor (tile `nt_`, segment `dseg`, chunk `dch`) runs
    // one chunk ahead; its segment's fields are fetched when the segment is entered, the following segment's are prefetched
    TileCo ct = first, nt_ = first;
    int cci = 0;
    struct SegW {
      int nch, ntaps, dil, wbase, wstep, lds_off;
    };
    auto fetchw = [&](int sg) -> SegW {
      const WsSeg& g = a.seg[sg];
      return SegW{g.nch, g.ntaps, g.dil, g.wbase, g.wstep, g.lds_off};
    };
    int dseg = 0, dch = 0;
    SegW dw = fetchw(0), dnx = fetchw(a.nseg > 1 ? 1 : 0);
    int d_woff = dw.wbase;
    // LDS byte offset of the weights of the chunk the DMA cursor points at, when it is consumed in a step of parity `par`
    // (an identity chunk has no weights: the value is its chunk index within the channel tile)
    auto wlds = [&](int par) { return dw.ntaps == 0 ? dch : (RES ? WRES_OFF + dw.lds_off + dch * dw.wstep : WS_OFF + par * WS_STRIDE); };
    if constexpr (RES) {
      // all weights, once: segment images are contiguous in the packed weights ([chunk][tap][Cout][32], Cout == CT here)
      const int nsg = a.nseg;
      for (int sg = 0; sg < nsg; ++sg) {
        const int np = (WS_SEGF(sg, nch) * WS_SEGF(sg, wstep)) >> 10;
        const int gb = WS_SEGF(sg, wbase), lb = WS_SEGF(sg, lds_off);
        for (int p = wave; p < np; p += NCW) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(smem + WRES_OFF + lb + p * 1024), 16, gb + p * 1024 + dma_lane, 0, 0, 0);
          if constexpr (X3)  // (the lo plane sits behind all hi images)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wl, (lds_ptr)(smem + WRES_OFF + a.wres_bytes + lb + p * 1024), 16, gb + p * 1024 + dma_lane, 0, 0, 0);
        }
      }
    }
    auto dma_advance = [&]() {
      if (++dch == dw.nch) {
        dch = 0;
        if (++dseg == a.nseg) {
          dseg = 0;
          next_tile(nt_);
        }
        dw = dnx;
        d_woff = dw.wbase;
        dnx = fetchw(dseg + 1 == a.nseg ? 0 : dseg + 1);
      } else {
        d_woff += dw.wstep;
      }
    };
    // the first chunk's weights are requested before anything this workgroup has to wait for (the tables of a fused GroupNorm,
    // the producers' first barrier): their latency passes behind both
    dma(dw.ntaps, d_woff, nt_, 0);
    int ntaps = dw.ntaps, d = dw.dil, wb = wlds(0);  // current chunk
    dma_advance();
    // the first tile's bias: requested here, behind the weights, so that its latency passes behind the tables and the first barrier
    // (it used to be loaded -- and waited for -- at the first step, ~1.5 us on the critical path of a one-tile workgroup)
    float bj[WN], bjn[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) bjn[nt] = bj[nt] = a.bias[first.ty * CT + wc * (WN * 32) + nt * 32 + l31];
    int bias_ty = first.ty;
    // Tables only for the clips this workgroup reaches: [first.b, last_b] in walk order.  At the deep levels a workgroup owns ONE tile
    // (64 clips x 4 channel tiles = 256 tiles), and the table of the clip `ss_ring / 2` on -- two dependent round trips to global
    // memory plus fp64 arithmetic, ~4-5 us at the first step -- was built for a clip the workgroup never touches.
    const int last_b = (a.rev ? a.ntiles - te : te - 1) / (a.ntx * a.nty);
    auto reaches = [&](int c) { return a.rev ? (c >= last_b && c <= first.b) : (c >= first.b && c <= last_b); };
    if (a.gn.nsrc > 0)
      for (int i = 0; i < (a.ss_ring >> 1); ++i) {
        const int c = first.b + (a.rev ? -i : i);
        if (reaches(c)) gn_table(c);
      }
#ifdef VQVS_TIMING
    const unsigned long long t_su1 = __builtin_amdgcn_s_memtime();  // (startup marks: requests out, first tables built)
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the first tables of a fused GroupNorm are in LDS: the producers read them next)
    __builtin_amdgcn_s_barrier();  // (pairs with the producers' barrier behind their first (scale, shift) table)
#ifdef VQVS_TIMING
    const unsigned long long t_su2 = __builtin_amdgcn_s_memtime();  // (first barrier passed)
#endif
    // (k-step 1 = the same address with bit 5 flipped: the swizzle XORs the 16-byte column index.  The WPE instantiation has no
    //  registers to spare and flips the bit at every use; the others keep both addresses.)
    constexpr int NKS = (WPE || ZP) ? 1 : 2;
    int aoff[3][NKS], aoff_d = -1;
    int aoff1[ZP ? 3 : 1];  // ZP: the second 32-row block's own addresses (its rows leave [0, 256) at other lanes than the first's)
    (void)aoff1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first chunk's (or all resident) weights have landed
#ifdef VQVS_TIMING
    const unsigned long long t_su3 = __builtin_amdgcn_s_memtime();  // (weights landed)
#endif
    sync_all();
#ifdef VQVS_TIMING
    const unsigned long long t_su4 = __builtin_amdgcn_s_memtime();  // (loop starts)
#endif
    WS_TMARK(4)
    if (VQVS_WS_EXP & 1024) {  // ablation: the consumers only keep the barrier count
      for (int g = 0; g < Q; ++g) sync_all();
      return;
    }
    for (int g = 0; g < Q; ++g) {
      if (cci == 0) {  // first chunk of a tile
        if (gn_ahead != 0 && ct.b != gn_clip) {  // ... of a clip: the table of the clip `ss_ring / 2` clips on (its slot is free by now)
          gn_clip = ct.b;
          if (reaches(ct.b + gn_ahead)) gn_table(ct.b + gn_ahead);
        }
        WS_TMARK(6)
        if ((n == 1 || (VQVS_WS_EXP & 256)) && pending) {  // (one-chunk tiles: the previous tile leaves here, from the other buffer)
          store_tile(0, NIT);
          pending = false;
        }
        if (ct.ty != bias_ty) {  // (one channel tile per launch at Cout <= 128: loaded once, at start-up)
          bias_ty = ct.ty;
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) bj[nt] = bjn[nt];  // requested a tile ago
        }
        if (a.nty > 1) {  // several channel tiles (channel tile fastest): the NEXT tile's bias is requested now and has a whole tile to arrive
          TileCo nxt = ct;
          next_tile(nxt);
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) bjn[nt] = a.bias[nxt.ty * CT + wc * (WN * 32) + nt * 32 + l31];
        }
        // (the bias as the C operand of the tile's first MFMAs, from a 16-register tile, instead of these v_mov: measured +0.4 %
        //  convolution time -- the second code copy of the first tap costs more than the moves)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt][r] = bj[nt];
      }
      WS_TMARK(0)
      const int nx_ntaps = dw.ntaps, nx_dil = dw.dil, nx_wb = wlds((g + 1) & 1);  // the chunk the DMA cursor points at = the next step's
      if (g + 1 < Q) {
        dma(dw.ntaps, d_woff, nt_, (g + 1) & 1);
        dma_advance();
      }
      WS_TMARK(1)
      const char* const sb = smem + (g & 1) * ACT_STRIDE;
      // fp32 storage: the identity-skip rows of this tile (accumulator layout) are requested before the tile's LAST chunk is
      // multiplied, so that their latency passes behind its MFMAs instead of in front of the epilogue
      float skv[MT][16];
      if constexpr (X3) {
        if (cci == n - 1 && a.skip != nullptr) {
          // (buffer loads through a descriptor that ends behind the tile's last valid row of the SOURCE: rows past it read as zero)
          const int t0 = ct.tx * a.TTO;
          const int lastrow = t0 + min(a.TTO, a.Lout - t0);
          const int srows = a.skip_rsz == RESIZE_UP2 ? (lastrow + 1) >> 1 : (a.skip_rsz == RESIZE_AVG2 ? 2 * lastrow : lastrow);
          const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
              const_cast<float*>(reinterpret_cast<const float*>(a.skip)) + (size_t)ct.b * a.skip_L * a.Cout, 0, srows * a.Cout * 4, 0x00020000);
          const int row0 = t0 + wt * RW + 4 * hh;
          const int ch4 = (ct.ty * CT + wc * (WN * 32) + l31) * 4;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = row0 + mt * 32 + (r & 3) + 8 * (r >> 2);
              if (a.skip_rsz == RESIZE_AVG2) {
                const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_s, (2 * row) * a.Cout * 4 + ch4, 0, 0));
                const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_s, (2 * row + 1) * a.Cout * 4 + ch4, 0, 0));
                skv[mt][r] = (x0 + x1) * 0.5f;
              } else {
                const int sr = a.skip_rsz == RESIZE_UP2 ? (row >> 1) : row;
                skv[mt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_s, sr * a.Cout * 4 + ch4, 0, 0));
              }
            }
        }
      }
      {
        if (d != aoff_d) {  // A-fragment offsets of the three taps (swizzled rows: not additive in the tap), rebuilt when the dilation changes
          aoff_d = d;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            if constexpr (ZP) {
              // tap k of output row r reads time r + (k - 1) d (1-tap and identity chunks: d = 0); outside the staged 256 rows lies
              // only zero padding (host: Lout <= 255), and row 255 is a row of zeros
              int row = wt * RW + l31 + (k - 1) * d, row1 = row + 32;
              row = (unsigned)row < 256u ? row : 255;
              row1 = (unsigned)row1 < 256u ? row1 : 255;
              aoff[k][0] = row * 64 + ((hh ^ ((row >> 2) & 3)) << 4);
              aoff1[k] = row1 * 64 + ((hh ^ ((row1 >> 2) & 3)) << 4);
            } else {
            const int row = wt * RW + l31 + k * d;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) aoff[k][ks] = (row * 64 + ((hh ^ ((row >> 2) & 3)) << 4)) ^ (ks * 32);
            }
          }
        }
        // LDS address of this lane's B fragments, per MFMA tile and k-step: the chunk's weights, or for an identity chunk j (= wb)
        // the identity block where the tile owns output channels [32 j, 32 j + 32) and the zero block elsewhere -- one code path
        int bad[WN][NKS];  // (k-step 1: bit 5 flipped, as above; every base is a multiple of 64)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks)
          bad[nt][ks] = ((ntaps != 0 ? wb + nt * (32 * 64) : ((RES ? C_OFF : WS_OFF + (g & 1) * WS_STRIDE) + (wc * WN + nt == wb ? 0 : 2048) - wc * (WN * 32 * 64))) + boff) ^ (ks * 32);
        auto tap = [&](int k) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int ao = NKS == 2 ? aoff[k][ks % NKS] : (aoff[k][0] ^ (ks * 32));
            const V8 a0 = *reinterpret_cast<const V8*>(sb + ao);
            V8 a1 = a0;
            if constexpr (ZP) a1 = *reinterpret_cast<const V8*>(sb + (aoff1[k] ^ (ks * 32)));
            else if constexpr (MT == 2) a1 = *reinterpret_cast<const V8*>(sb + ao + 32 * 64);
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) {
              const V8 bf = *reinterpret_cast<const V8*>(smem + (NKS == 2 ? bad[nt][ks % NKS] : (bad[nt][0] ^ (ks * 32))) + k * (CT * 64));
              if constexpr (X3) {
                // hi * hi + lo * hi + hi * lo (activation lo plane ACT_BYTES behind the hi plane, weight lo plane W_BYTES / all hi
                // images behind)
                const V8 bl = *reinterpret_cast<const V8*>(smem + bad[nt][ks % NKS] + k * (CT * 64) + (RES ? a.wres_bytes : W_BYTES));
                const V8 a0l = *reinterpret_cast<const V8*>(sb + ACT_BYTES + ao);
                const V8 a1l = *reinterpret_cast<const V8*>(sb + ACT_BYTES + ao + 32 * 64);
                acc[0][nt] = ws_mfma(a0, bf, acc[0][nt]);
                acc[MT - 1][nt] = ws_mfma(a1, bf, acc[MT - 1][nt]);
                acc[0][nt] = ws_mfma(a0l, bf, acc[0][nt]);
                acc[MT - 1][nt] = ws_mfma(a1l, bf, acc[MT - 1][nt]);
                acc[0][nt] = ws_mfma(a0, bl, acc[0][nt]);
                acc[MT - 1][nt] = ws_mfma(a1, bl, acc[MT - 1][nt]);
              } else if (VQVS_WS_EXP & 4096) {  // ablation: the fragment reads stay, the MFMAs go
                asm volatile("" ::"v"(a0), "v"(a1), "v"(bf));
              } else {
                acc[0][nt] = ws_mfma(a0, bf, acc[0][nt]);
                if constexpr (MT == 2) acc[MT - 1][nt] = ws_mfma(a1, bf, acc[MT - 1][nt]);
              }
            }
          }
        };
        if (!(VQVS_WS_EXP & 16)) {
          tap(0);
          if (ntaps == 3) {
            tap(1);
            tap(2);
          }
        }
      }
      WS_TMARK(2)
      // last chunk of the tile: statistics + rounding + out-tile
      if (cci == n - 1 && (VQVS_WS_EXP & 32)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) asm volatile("" ::"v"(acc[mt][nt][0]), "v"(acc[mt][nt][15]));  // keep the MFMAs alive
        pt_ = ct;
        pbuf = obuf;
        pending = true;
      }
      if (cci == n - 1 && !(VQVS_WS_EXP & 32)) {
        int zl = 0;
        asm volatile("" : "+v"(zl));
        const int t0 = ct.tx * a.TTO;
        const int nvalid = min(a.TTO, a.Lout - t0);
        const int lim = nvalid - (wt * RW + 4 * hh) + zl;  // rows of this lane: mt * 32 + (r & 3) + 8 * (r >> 2) < lim are real
        if (wt * RW + RW > nvalid) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
              for (int r = 0; r < 16; ++r)
                if (mt * 32 + (r & 3) + 8 * (r >> 2) >= lim) acc[mt][nt][r] = 0.f;
        }
        if constexpr (X3) {
          // fp32 storage: the identity skip (unet.py:316) is added HERE, exactly, from global memory in the accumulators' own
          // layout (a lane = one output channel, 32 lanes = 128 contiguous bytes of a row); statistics of the stored values; rows
          // leave as dwords through a descriptor that ends behind the tile's last valid row.
          const int ch = ct.ty * CT + wc * (WN * 32) + l31 + zl;
          float* const outf = reinterpret_cast<float*>(a.out);
          const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
              outf + (size_t)ct.b * a.Lout * a.Cout, 0, (t0 + nvalid) * a.Cout * 4, 0x00020000);
          const int row0 = t0 + wt * RW + 4 * hh;  // this lane's first row (of the clip); + mt * 32 + (r & 3) + 8 * (r >> 2)
          if (a.skip != nullptr) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mt][0][r] += skv[mt][r];
          }
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = acc[mt][0][r];  // (rows past the tile's end were zeroed above)
              s1 += v;
              s2 = fmaf(v, v, s2);
              if (!(VQVS_WS_EXP & 4))
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o, ((row0 + mt * 32 + (r & 3) + 8 * (r >> 2)) * a.Cout + ch) * 4, 0, 0);
            }
          s1 += __shfl_xor(s1, 32);
          s2 += __shfl_xor(s2, 32);
          if (hh == 0) reinterpret_cast<float2*>(smem + R_OFF)[wt * CT + wc * (WN * 32) + l31 + zl] = float2{s1, s2};
        } else if constexpr (WPE) {
          // Wave-private epilogue: 16 rows at a time (accumulator registers 4 j .. 4 j + 3 of both row blocks) are rounded in PAIRS
          // of rows into this wave's 2 KiB region ([8 row pairs][64 channels] dwords), read back as 8-channel pieces of a pair,
          // un-zipped (v_perm) and stored: a lane writes 16 B of each of two rows, 8 lanes one 128-byte row segment.  LDS
          // operations of one wave execute in order, so the region needs no barrier.  Statistics: those of the rounded values.
          char* const priv = smem + O_OFF + wave * 2048 + zl;
          char* const wr0 = priv + (2 * hh) * 256 + l31 * 4;
          const int zlane = lane + zl;   // (lane-derived addresses are rebuilt here, not carried through the K loop)
          const int prw = zlane >> 3;                                                // pair this lane reads back
          const int rrow = (prw >> 2) * 32 + ((prw >> 1) & 1) * 4 + (prw & 1) * 2;  // its first row within the wave's 64 (+ 8 j)
          const char* const rd0 = priv + prw * 256 + (zlane & 7) * 32;
          // buffer stores through a descriptor that ends behind the tile's last valid row: the rows past it (the clip's end, or the
          // rows the next tile owns) are out of range and dropped by the address check -- no per-lane masks, no branches.  (RES:
          // one channel tile per launch, so a row is CT * 2 = 256 bytes.)
          const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
              outp + (size_t)ct.b * a.Lout * CT, 0, (t0 + nvalid) * (CT * 2), 0x00020000);
          const int vo0 = ((t0 + wt * 64 + rrow) * CT + wc * 64 + (zlane & 7) * 8) * 2;
          float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                  const int r = 4 * j + 2 * rr;
                  char* const o0 = wr0 + (mt * 4 + rr) * 256 + nt * 128;
                  if constexpr (WsOp<T>::one == 0x3C00) {
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    const h2 ones = {(_Float16)1.0f, (_Float16)1.0f};
                    const h2 pk = {(_Float16)acc[mt][nt][r], (_Float16)acc[mt][nt][r + 1]};
                    s1[nt] = __builtin_amdgcn_fdot2(pk, ones, s1[nt], false);
                    s2[nt] = __builtin_amdgcn_fdot2(pk, pk, s2[nt], false);
                    *reinterpret_cast<h2*>(o0) = pk;
                  } else {
                    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                    const b2 pk = {(__bf16)acc[mt][nt][r], (__bf16)acc[mt][nt][r + 1]};
                    const float v0 = (float)pk[0], v1 = (float)pk[1];
                    s1[nt] += v0 + v1;
                    s2[nt] = fmaf(v0, v0, fmaf(v1, v1, s2[nt]));
                    *reinterpret_cast<b2*>(o0) = pk;
                  }
                }
            const u32x4 lo = *reinterpret_cast<const u32x4*>(rd0);
            const u32x4 hi = *reinterpret_cast<const u32x4*>(rd0 + 16);
            if (!(VQVS_WS_EXP & 4)) {
              u32x4 e, o;
              e[0] = __builtin_amdgcn_perm(lo[1], lo[0], 0x05040100u);
              e[1] = __builtin_amdgcn_perm(lo[3], lo[2], 0x05040100u);
              e[2] = __builtin_amdgcn_perm(hi[1], hi[0], 0x05040100u);
              e[3] = __builtin_amdgcn_perm(hi[3], hi[2], 0x05040100u);
              __builtin_amdgcn_raw_buffer_store_b128(e, rs_o, vo0 + j * (8 * CT * 2), 0, 0);
              o[0] = __builtin_amdgcn_perm(lo[1], lo[0], 0x07060302u);
              o[1] = __builtin_amdgcn_perm(lo[3], lo[2], 0x07060302u);
              o[2] = __builtin_amdgcn_perm(hi[1], hi[0], 0x07060302u);
              o[3] = __builtin_amdgcn_perm(hi[3], hi[2], 0x07060302u);
              __builtin_amdgcn_raw_buffer_store_b128(o, rs_o, vo0 + j * (8 * CT * 2) + CT * 2, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);  // (one piece at a time: overlapped pieces cost more registers than the kernel has)
          }
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            s1[nt] += __shfl_xor(s1[nt], 32);
            s2[nt] += __shfl_xor(s2[nt], 32);
            if (hh == 0) reinterpret_cast<float2*>(smem + R_OFF)[wt * CT + wc * 64 + nt * 32 + l31 + zl] = float2{s1[nt], s2[nt]};
          }
        } else {
        // round PAIRS of rows (same channel) into one dword of the out-tile; the statistics are those of the rounded values
        char* const ob = smem + O_OFF + obuf * O_BYTES + (wt * (RW / 2) + 2 * hh) * OP + (wc * (WN * 32) + l31 + zl) * 4;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              char* const o0 = ob + (mt * 16 + ((r & 3) >> 1) + 4 * (r >> 2)) * OP + nt * 128;
              if constexpr (WsOp<T>::one == 0x3C00) {
                // fp16: one v_cvt_pk per pair; sum = dot((a, b), (1, 1)), sum of squares = dot((a, b), (a, b)), fp32 accumulation
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 ones = {(_Float16)1.0f, (_Float16)1.0f};
                const h2 pk = {(_Float16)acc[mt][nt][r], (_Float16)acc[mt][nt][r + 1]};
                s1 = __builtin_amdgcn_fdot2(pk, ones, s1, false);
                s2 = __builtin_amdgcn_fdot2(pk, pk, s2, false);
                *reinterpret_cast<h2*>(o0) = pk;
              } else {
                typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                const b2 pk = {(__bf16)acc[mt][nt][r], (__bf16)acc[mt][nt][r + 1]};
                const float v0 = (float)pk[0], v1 = (float)pk[1];
                s1 += v0 + v1;
                s2 = fmaf(v0, v0, fmaf(v1, v1, s2));
                *reinterpret_cast<b2*>(o0) = pk;
              }
            }
          s1 += __shfl_xor(s1, 32);
          s2 += __shfl_xor(s2, 32);
          if (hh == 0) reinterpret_cast<float2*>(smem + R_OFF + obuf * R_BYTES)[wt * CT + wc * (WN * 32) + nt * 32 + l31 + zl] = float2{s1, s2};
        }
        }
        pt_ = ct;
        pbuf = obuf;
        if constexpr (DB) obuf ^= 1;
        pending = true;
      }
      WS_TMARK(3)
      if (++cci == n) {
        cci = 0;
        next_tile(ct);
      }
      ntaps = nx_ntaps;
      d = nx_dil;
      wb = nx_wb;
      // The previous tile's rows leave at the END of its successor's first step, behind the wait for this step's weight DMA: the
      // stores then have a whole step to be acknowledged before the next vmcnt(0) (CDNA counts stores in vmcnt too, and a wait
      // placed right after them exposes the full write latency once per tile).
      if (n > 1 && pending && cci >= 1) {  // (cci - 1 = the successor's step that just ended: 0 .. n - 2)
        if constexpr (!RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int i0 = (cci - 1) * st_per, i1 = i0 + st_per;
        if (!(VQVS_WS_EXP & 4) && i0 < NIT) store_tile(i0, i1 < NIT ? i1 : NIT);
        if (i1 >= NIT) pending = false;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(VQVS_WS_EXP & 32768) && !((VQVS_WS_EXP & 131072) && (g & 1) == 0 && g < Q - 1)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      } else {
        skipb = (g & 1) == 0 && g < Q - 1;
        sync_all();
        skipb = false;
      }
      WS_TMARK(4)
    }
    if (pending) store_tile(0, NIT);
#ifdef VQVS_TIMING
    WS_TMARK(5)
    if (lane == 0 && ((int)blockIdx.x & 15) == 3) {
      for (int i = 0; i < 7; ++i) atomicAdd(&g_ws_timing[8 + i], tacc[i]);  // (5: the last tile's store, 6: look-ahead GroupNorm tables)
      atomicAdd(&g_ws_timing[22], t_su1 - t_sh0);  // startup of the sampled consumer wave: entry -> requests out and tables built,
      atomicAdd(&g_ws_timing[23], t_su2 - t_su1);  // -> first barrier passed,
      atomicAdd(&g_ws_timing[24], t_su3 - t_su2);  // -> first weights landed,
      atomicAdd(&g_ws_timing[25], t_su4 - t_su3);  // -> second barrier passed (the step loop starts)
      atomicAdd(&g_ws_timing[20], __builtin_amdgcn_s_memtime() - t_sh0);      // shader-clock ticks of this workgroup ...
      atomicAdd(&g_ws_timing[21], __builtin_amdgcn_s_memrealtime() - t_rt0);  // ... and 100 MHz ticks: their ratio is the clock it ran at
      atomicAdd(&g_ws_timing[17], 1ull);
      atomicAdd(&g_ws_timing[19], (unsigned long long)(te - tb));
    }
#endif
  }
}

int ws_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VQVS_WS");  // 0: every convolution on conv_mfma_kernel (A/B measurements)
    v = e ? atoi(e) : 1;
  }
  return v;
}

// (function attributes and compute-unit counts are per DEVICE: both caches are keyed by the current device, so a process that
//  drives several GPUs -- not the one-process-per-GPU layout of sampler.py, but allowed by the ABI -- sets them on each)
constexpr int WS_MAX_DEV = 64;
int ws_cur_dev() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= WS_MAX_DEV) return 0;
  return dev;
}
int ws_num_cus() {
  static std::atomic<int> n[WS_MAX_DEV];
  const int dev = ws_cur_dev();
  int v = n[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    hipDeviceProp_t p;
    v = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    n[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}

// LDS bytes besides resident weights and the (scale, shift) ring (the kernel's layout constants, restated for the planner)
constexpr int ws_fixed_lds(int rows, int ct, bool res, bool x3 = false) {
  const int ncw = rows / 32, nwt = ct == 32 ? rows / 32 : rows / 64, db = ct == 32 ? 2 : 1, pl = x3 ? 2 : 1;  // (x3: ncw is unused)
  const int obytes = x3 ? 0 : ((res && ct == 128) ? ncw * 2048 : (rows / 2) * (ct * 4 + 16));
  return (res ? 2 * pl * rows * 64 : 2 * pl * (rows * 64 + 3 * ct * 64)) + db * obytes + db * nwt * ct * 8 + (res ? 4096 : 0);
}
constexpr int WS_LDS_MAX = 160 * 1024;
// LDS one workgroup may use: two workgroups of the 128-row geometry share a CU
constexpr int ws_lds_cap(int rows, bool x3 = false) { return (rows == 256 || x3) ? WS_LDS_MAX : WS_LDS_MAX / 2; }

template <typename T, int ROWS, int CT, bool RES, bool AVG, bool ZP = false>
int ws_launch(const WsArgs& w, hipStream_t st) {
  constexpr bool X3 = sizeof(T) == 4;
  const int lds = ws_fixed_lds(ROWS, CT, RES, X3) + (RES ? (X3 ? 2 : 1) * w.wres_bytes : 0) + w.ss_ring * w.ss_bytes;
  static std::atomic<bool> attr_done[WS_MAX_DEV];  // (per instantiation and device; setting it twice from two threads is harmless)
  const int dev = ws_cur_dev();
  if (!attr_done[dev].load(std::memory_order_acquire)) {
    VQVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<T, ROWS, CT, RES, AVG, ZP>), hipFuncAttributeMaxDynamicSharedMemorySize, ws_lds_cap(ROWS, X3)));
    attr_done[dev].store(true, std::memory_order_release);
  }
  static const int grid_env = getenv("VQVS_WS_GRID") ? atoi(getenv("VQVS_WS_GRID")) : 0;  // (A/B measurements: workgroups per launch)
  constexpr int NT = ws_threads<T, ROWS>();
  const int nwg = (grid_env > 0 ? grid_env : ws_num_cus()) * (1024 / NT);  // persistent workgroups: one (two) per CU
  const int grid = w.ntiles < nwg ? w.ntiles : nwg;
  hipLaunchKernelGGL((conv_ws_kernel<T, ROWS, CT, RES, AVG, ZP>), dim3(grid), dim3(NT), lds, st, w);
  VQVS_HIP(hipGetLastError());
  return 0;
}
template <typename T, int ROWS, int CT>
int ws_launch_f(const WsArgs& w, bool res, bool avg, hipStream_t st) {
  // (no RES && AVG instantiation: resident weights + four loads per chunk do not fit 128 VGPRs, and a spill has no place beside
  //  the producers' counted waits -- launch_conv_ws never asks for it, tools/check_no_scratch.py gates the rest at build time)
  if (avg && res) return -1;
  if (avg) return ws_launch<T, ROWS, CT, false, true>(w, st);
  return res ? ws_launch<T, ROWS, CT, true, false>(w, st) : ws_launch<T, ROWS, CT, false, false>(w, st);
}
int ws_launch_f32(const WsArgs& w, int rows, bool res, hipStream_t st) {  // fp32 storage: 256 x 64 or 128 x 128 tiles, no avg-pooled segments
  if (rows == 128) return res ? -1 : ws_launch<float, 128, 128, false, false>(w, st);
  return res ? ws_launch<float, 256, 64, true, false>(w, st) : ws_launch<float, 256, 64, false, false>(w, st);
}
template <typename T>
int ws_launch_t(const WsArgs& w, int rows, int CT, bool res, bool avg, hipStream_t st) {
  if (w.zp) {
    if (rows != 256 || res || avg || CT < 64) return -1;
    return CT == 128 ? ws_launch<T, 256, 128, false, false, true>(w, st) : ws_launch<T, 256, 64, false, false, true>(w, st);
  }
  if (rows == 128) {  // (two workgroups per CU; 32-channel tiles only come in the 256-row geometry)
    if (CT == 128) return ws_launch_f<T, 128, 128>(w, res, avg, st);
    if (CT == 64) return ws_launch_f<T, 128, 64>(w, res, avg, st);
    return -1;
  }
  if (CT == 128) return ws_launch_f<T, 256, 128>(w, res, avg, st);
  if (CT == 64) return ws_launch_f<T, 256, 64>(w, res, avg, st);
  return ws_launch_f<T, 256, 32>(w, res, avg, st);
}

}  // namespace

#ifdef VQVS_TIMING
int ws_timing_read(unsigned long long* out32, int reset) {
  VQVS_HIP(hipDeviceSynchronize());
  VQVS_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_ws_timing), 32 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[32] = {};
    VQVS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ws_timing), z, sizeof(z)));
  }
  return 0;
}
#endif

// Returns 1 when the launch was taken by the wave-specialised kernel, 0 when the shape is not covered (caller falls back to
// conv_mfma_kernel), < 0 on error.
namespace {
struct WsPlan {
  WsArgs w;
  int CT, rows;
  bool res, avg, gn;
};
// The launch as the kernel wants it; false = shape not covered.  plan.gn: a.gn is honoured (the producers build the rows).
bool ws_plan(const ConvArgs& a, int B, int precision, WsPlan& plan);
}  // namespace

// The one predicate behind "may the schedule builder pick a geometry only conv_ws_kernel has" (conv_tile_rows) and ws_plan's own
// switches: with VQVS_WS=0 or (fp32 storage) VQVS_WS_F32=0 every launch must keep conv_mfma_kernel's tile geometry.
int ws_f32_enabled() {
  static const int v = getenv("VQVS_WS_F32") ? atoi(getenv("VQVS_WS_F32")) : 1;  // 0: the fp32 mode stays on conv_mfma_kernel (A/B)
  return v;
}
bool ws_available(int precision) { return ws_enabled() != 0 && (precision != 0 || ws_f32_enabled() != 0); }

bool ws_fuses_gn(const ConvArgs& a, int B, int precision) {
  WsPlan plan;
  return a.gn != nullptr && ws_plan(a, B, precision, plan) && plan.gn;
}

int launch_conv_ws(const ConvArgs& a, int B, int precision, hipStream_t st) {
  WsPlan plan;
  if (!ws_plan(a, B, precision, plan)) return 0;
  if (a.gn != nullptr && !plan.gn) return 0;  // (the caller launches gn_prepare and comes back without a.gn)
  const int rc = precision == 0 ? ws_launch_f32(plan.w, plan.rows, plan.res, st)
                 : precision == 2 ? ws_launch_t<half_t>(plan.w, plan.rows, plan.CT, plan.res, plan.avg, st)
                                  : ws_launch_t<bf16_t>(plan.w, plan.rows, plan.CT, plan.res, plan.avg, st);
  return rc < 0 ? rc : 1;
}

namespace {
bool ws_plan(const ConvArgs& a, int B, int precision, WsPlan& plan) {
  const bool x3 = precision == 0;  // fp32 storage: only where the caller allows it (ConvArgs.ws_f32) and the lo weights exist
  if (!ws_available(precision) || (x3 && (!a.ws_f32 || a.w_lo == nullptr || a.Cout % 64 != 0))) return 0;
  const int es = x3 ? 4 : 2;
  if (a.Cout % 32 != 0 || a.out_f32 || a.epi_gelu || a.nbw || (a.out_rows != 0 && a.out_rows != a.Lout)) return 0;
  // rows per clip a source must have for a given resize (kernels.hpp RESIZE_*)
  auto len_ok = [&](int rsz, int Lsrc) { return rsz == RESIZE_UP2 ? Lsrc * 2 == a.Lout : (rsz == RESIZE_AVG2 ? Lsrc / 2 == a.Lout : Lsrc == a.Lout); };
  if ((long long)a.Lout * a.Cout * es >= (1LL << 31)) return 0;  // (the tile stores go through a 32-bit descriptor of one clip's rows)
  if (a.skip != nullptr && (a.skip_C != a.Cout || !len_ok(a.skip_resize, a.skip_L) || (long long)a.skip_L * a.skip_C * es >= (1LL << 29))) return 0;
  bool avg = a.skip != nullptr && a.skip_resize == RESIZE_AVG2;
  int dmax0 = 0;
  for (int s = 0; s < a.nseg; ++s)
    if (a.seg[s].ntaps == 3 && a.seg[s].dil > dmax0) dmax0 = a.seg[s].dil;
  // fp32 storage: the schedule builder asks for the 128-row x 128-channel tile through tile_rows (conv_tile_rows)
  const bool tall = x3 && a.Cout % 128 == 0 && a.tile_rows > 0 && a.tile_rows == 128 - 2 * dmax0;
  int CT = x3 ? (tall ? 128 : 64) : (a.Cout % 128 == 0 ? 128 : (a.Cout % 64 == 0 ? 64 : 32));
  // Short launches (the deep levels at half a node's batch, 32 clips: 128 tiles of 128 channels for 256 CUs): when 128-channel tiles
  // would leave half of the chip idle, 64-channel tiles put a tile on every CU -- half the MFMAs per step for the same number of
  // dependent steps.  Bitwise the same results: a tile's width enters neither its elements' sums nor the order of its statistics.
  static const int narrow_env = getenv("VQVS_WS_NARROW") ? atoi(getenv("VQVS_WS_NARROW")) : 1;  // 0: off (A/B measurements)
  // (only from 256 output channels: a 128-channel launch may keep its weights resident, and that form's wave-private epilogue sums
  //  the tile statistics in another order than the out-tile epilogue -- the batch size must not change a clip's bits)
  if (narrow_env && !x3 && CT == 128 && a.Cout >= 256 && a.tile_rows > 0) {
    const long long rt = a.Lout <= 255 ? 1 : (a.Lout + a.tile_rows - 1) / a.tile_rows;  // (a clip of <= 255 rows is one tile: ZP, below)
    if (rt * (a.Cout / 128) * B * 2 <= ws_num_cus()) CT = 64;
  }
  plan.w = WsArgs{};
  WsArgs& w = plan.w;
  int dmax = 0, n = 0;
  for (int s = 0; s < a.nseg; ++s) {
    const SegDesc& g = a.seg[s];
    if (g.C % 32 != 0 || (g.ntaps != 1 && g.ntaps != 3) || !len_ok(g.resize, g.Lsrc) || g.src2 != nullptr) return 0;  // (src2: conv_mfma_kernel's two-tensor prologue)
    if ((long long)g.Lsrc * g.Csrc * es >= (1LL << 29)) return 0;  // (the dummy loads of an AVG launch rely on offset + 2^30 being out of range)
    avg = avg || g.resize == RESIZE_AVG2;
    WsSeg& q = w.seg[s];
    q.src = g.src;
    q.ss = g.ss;
    q.Csrc = g.Csrc;
    q.clip_bytes = g.Lsrc * g.Csrc * es;
    q.c0 = g.c0;
    q.nch = g.C / 32;
    q.ntaps = g.ntaps;
    q.dil = g.ntaps == 3 ? g.dil : 0;
    q.rsz = g.resize;
    q.ss_stride = g.ss_stride;
    q.ss_c0 = g.ss_c0;
    q.ss_lds = w.ss_bytes;
    if (g.ss != nullptr) w.ss_bytes += g.C * 8;
    q.wbase = (int)(g.w_off * 2);
    q.wstep = g.ntaps * a.Cout * 64;
    q.lds_off = w.wres_bytes;
    w.wres_bytes += q.nch * q.wstep;
    if (q.dil > dmax) dmax = q.dil;
    n += q.nch;
  }
  w.nseg = a.nseg;
  if (a.skip != nullptr && x3) {  // (fp32 storage adds the skip in the epilogue, exactly)
    w.skip = a.skip;
    w.skip_L = a.skip_L;
    w.skip_rsz = a.skip_resize;
  } else if (a.skip != nullptr) {
    WsSeg& q = w.seg[w.nseg++];
    q.src = a.skip;
    q.ss = nullptr;
    q.Csrc = a.skip_C;
    q.clip_bytes = a.skip_L * a.skip_C * 2;
    q.c0 = 0;
    q.nch = CT / 32;
    q.ntaps = 0;
    q.dil = 0;
    q.rsz = a.skip_resize;
    n += q.nch;
  }
  // the staged window: 256 rows, or 128 (tile_rows = 128 - 2 dmax, chosen by the schedule builder: conv_tile_rows)
  if (x3 && (avg || (!tall && a.tile_rows != 256 - 2 * dmax))) return 0;
  const int rows = a.tile_rows == 256 - 2 * dmax ? 256 : (a.tile_rows == 128 - 2 * dmax && CT >= 64 && a.tile_rows > 0 ? 128 : 0);
  if (rows == 0 || a.w_bytes > 0x7fffffffLL || n < (CT == 32 ? 1 : 2)) return 0;
  plan.rows = rows;
  w.nchunks = n;
  w.w = a.w_hi;
  w.w_lo = a.w_lo;
  w.w_bytes = (int)a.w_bytes;
  w.bias = a.bias;
  w.out = a.out;
  w.stats = a.stats;
  w.Cout = a.Cout;
  w.Lout = a.Lout;
  w.TTO = a.tile_rows;
  w.ntx = (a.Lout + a.tile_rows - 1) / a.tile_rows;
  w.nty = a.Cout / CT;
  // whole-clip tiles (template flag ZP): the clip fits one staged window of zero-padded rows although its dilation would split it
  static const int zp_env = getenv("VQVS_WS_ZP") ? atoi(getenv("VQVS_WS_ZP")) : 1;  // 0: off (A/B measurements)
  if (zp_env && !x3 && CT >= 64 && !avg && rows == 256 && a.Lout <= 255 && w.ntx > 1 && w.nty > 1) {  // (nty > 1: never a resident-weights launch)
    w.zp = 1;
    w.TTO = 256;
    w.ntx = 1;
  }
  w.ntiles = w.ntx * w.nty * B;
  w.ntiles_stat = a.ntiles;
  static const int rev_env = getenv("VQVS_WS_REV") ? atoi(getenv("VQVS_WS_REV")) : 1;  // 0: every launch walks forward (A/B measurements)
  w.rev = rev_env ? a.rev : 0;
  if (w.ntiles <= 0) return 0;
  // resident weights: one channel tile per launch and everything fits the CU's LDS
  static const int res_env = getenv("VQVS_WS_RES") ? atoi(getenv("VQVS_WS_RES")) : 1;  // 0: always stream the weights (A/B measurements)
  if (w.ss_bytes > 8192) return 0;  // (one 16-byte piece per producer thread)
  // (scale, shift) tables in LDS: the producers' load cursor runs up to four chunks ahead of the chunk being staged, and a table is
  // written when the cursor ENTERS its clip -- so the ring must hold every clip between the staged chunk and the cursor:
  // 2 clips when a clip takes at least four steps, 3 at two or three steps, 5 when a clip is a single step (32 x 3 -> 32, L <= 254)
  const long long spc = (long long)w.ntx * w.nty * n;  // steps per clip
  w.ss_ring = spc >= 4 ? 2 : (spc >= 2 ? 4 : 8);
  const int ss_total = w.ss_ring * w.ss_bytes;
  if (ws_fixed_lds(rows, CT, false, x3) + ss_total > ws_lds_cap(rows, x3)) return 0;
  // (avg-pooled launches stream their weights: resident weights + four loads per chunk do not fit 128 VGPRs without spills, and
  //  compiler-generated scratch traffic has no place beside the producers' counted waits)
  if (w.ntiles >= (1 << 24)) return 0;  // (the kernel's tile-range arithmetic is 32-bit)
  plan.CT = CT;
  plan.avg = avg;
  plan.res = res_env && !avg && !tall && w.nty == 1 && ws_fixed_lds(rows, CT, true, x3) + (x3 ? 2 : 1) * w.wres_bytes + ss_total <= ws_lds_cap(rows, x3);
  if (plan.res && CT == 128 && (long long)a.Lout * CT * 2 >= (1LL << 31)) return 0;  // (the wave-private epilogue's store descriptor covers one clip)
  // GroupNorm built by the producers (WsGn): every prologue segment is one whole source of *a.gn, in order; a group is a power of
  // two of lanes; few enough tile partials per channel that the serial sum at a clip change stays short
  plan.gn = false;
  static const int gn_env = getenv("VQVS_WS_GN") ? atoi(getenv("VQVS_WS_GN")) : 1;  // 0: always the gn_prepare launch (A/B measurements)
  // (clips of up to 8 tiles: measured +0.6 % clips/s; from 16 tiles on the table's construction costs what its gn_prepare launch
  //  did -- 66.19 / 66.08 / 66.01 clips/s at 16 / 32 / 64 against 66.16 at 8, 67.1 against 67.5 with every launch fused)
  // (round 5: with the table's parameter loads hoisted in front of its partial sums, 16 / 32 tiles measure 67.80 / 67.88 clips/s against
  //  67.66 at 8 and 67.73 at 64, two alternating rounds on one box: 32, i.e. clips of up to 8064 rows)
  static const int gn_max_tiles = getenv("VQVS_WS_GN_TILES") ? atoi(getenv("VQVS_WS_GN_TILES")) : 32;
  if (a.gn != nullptr && gn_env) {
    const GnArgs& g = *a.gn;
    const int cpg = g.groups > 0 ? g.Ctot / g.groups : 0;
    bool ok = g.nsrc >= 1 && g.nsrc <= 2 && g.mr == nullptr && cpg >= 1 && cpg <= 64 && (cpg & (cpg - 1)) == 0 && g.Ctot % 64 == 0 &&
              g.src[0].C % 64 == 0 && g.Ctot * 8 == w.ss_bytes;
    int si = 0;
    for (int s = 0; s < a.nseg && ok; ++s) {
      if (a.seg[s].ss == nullptr) continue;
      ok = si < g.nsrc && a.seg[s].c0 == 0 && a.seg[s].C == g.src[si].C && a.seg[s].Csrc == g.src[si].C && g.src[si].ntiles <= gn_max_tiles;
      ++si;
    }
    ok = ok && si == g.nsrc;
    if (ok) {
      WsGn& G = w.gn;
      G.nsrc = g.nsrc;
      G.part0 = g.src[0].partials;
      G.ntiles0 = g.src[0].ntiles;
      G.C0 = g.src[0].C;
      G.part1 = g.nsrc > 1 ? g.src[1].partials : nullptr;
      G.ntiles1 = g.nsrc > 1 ? g.src[1].ntiles : 0;
      G.C1 = g.nsrc > 1 ? g.src[1].C : 0;
      G.nclips = B;
      G.Ctot = g.Ctot;
      G.cpg = cpg;
      {  // slices per channel: enough that a lane adds at most ~32 partials, as long as a group still fits the wave's channels
        int mt = 0;
        for (int i = 0; i < g.nsrc; ++i) mt = g.src[i].ntiles > mt ? g.src[i].ntiles : mt;
        int tpc = 1;
        while (tpc < 8 && mt > 8 * tpc && cpg * 2 * tpc <= 64) tpc *= 2;
        G.tpc = tpc;
      }
      G.inv_count = g.inv_count;
      G.gamma = g.gamma;
      G.beta = g.beta;
      G.film = g.film;
      G.film_stride = g.film_stride;
      G.film_off = g.film_off;
      G.status = g.status;
      G.guard = g.guard;
      plan.gn = true;
    }
  }
  return true;
}
}  // namespace

}  // namespace vqvs
