// bf16 specialisations of the two convolution kernels (the bench path): 8-wave workgroups (2 waves per SIMD),
// register-prefetch staging, LDS rows padded to 48 bytes instead of swizzled (the HBM tile image stays compact and
// unswizzled for bf16), LDS addresses precomputed per lane so the tap loops are MFMA + ds_read only.
//   k_conv3x3_bf16<MT,NT>   forward / input-gradient 3x3 conv (same contract as k_conv3x3 in conv.hip)
//   k_conv_wgrad_bf16<NTT>  weight gradient (same contract as k_conv_wgrad in conv.hip)
#include <type_traits>
#include "kernels.h"
#include "xchg_dev.h"

namespace dta {

constexpr int RB = 48;   // LDS bytes per 16-channel bf16 row: 32 data + 16 pad -> conflict-free b128 / tr_b16 reads
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

// two floats -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hw_bf16x2));
}

// n / d for the small non-negative operands of the staging plans: magic = ceil(2^32 / d), exact for n * d < 2^32
__device__ __forceinline__ int fastdiv(int n, unsigned magic) { return (int)__umulhi((unsigned)n, magic); }
static inline unsigned fastdiv_magic(int d) { return d > 1 ? (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d) : 0u; }
__device__ __forceinline__ int fastdiv1(int n, int d, unsigned magic) { return d > 1 ? fastdiv(n, magic) : n; }

__device__ __forceinline__ bf16x8 lds_b128(const unsigned char* base, int off) {
  return *reinterpret_cast<const bf16x8*>(base + off);
}
// Transposing fragment read: 8 k-values (rows r..r+7 in two groups of 4) of one 16-channel column block.
__device__ __forceinline__ bf16x8 lds_tr8(const unsigned char* base, int off) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(base + off));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(base + off + 4 * RB));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 lds_tr8w(const unsigned char* base, int off) {   // same, 32-byte rows
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(base + off));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(base + off + 4 * 32));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad conv
// ------------------------------------------------------------------------------------------------
// developer instrumentation (-DDTA_TICKS): cycle stamps of one workgroup of the few-chunk convs (conv2 forward: N == 64,
// NC == 2, statistics on), wave 0: [0] entry, [1] tables + halo zero, [2] first chunk landed, [3] chunk loop done,
// [4] output stored, [5] statistics done
#ifdef DTA_TICKS
// developer instrumentation: start / end time (100 MHz wall clock, common to all XCDs) of every workgroup of selected kernels
__device__ long long g_wgstamp_conv[4][8192][2];
extern "C" int dta_debug_wgstamps_conv(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgstamp_conv), sizeof(long long) * 4 * 8192 * 2); }
struct WgStamp {
  int id; long long t0;
  __device__ WgStamp(int id_) : id(id_), t0(wall_clock64()) {}
  __device__ ~WgStamp() {
    const int w = blockIdx.x + gridDim.x * blockIdx.y;
    if (threadIdx.x == 0 && id >= 0 && w < 8192) { g_wgstamp_conv[id][w][0] = t0; g_wgstamp_conv[id][w][1] = wall_clock64(); }
  }
};
#define WGSTAMP(id) WgStamp _wgstamp(id)
#else
#define WGSTAMP(id)
#endif
#ifdef DTA_TICKS
// fused-input first conv (XN): per chunk half-interval, waves 0 (stages first) and NW/2 (multiplies first) of workgroup 100:
// [0] tile-out, [1] multiply phase when first, [2] staging (wait for the input loads, convert, LDS writes, next fetches),
// [3] multiply phase when second, [4] barrier wait; summed over the chunks -> g_xticks[wave class][5], + [5] whole loop
__device__ long long g_xticks[2][8];
__device__ long long g_xticks2[2][4];      // staging split: [0] input wait + convert + LDS writes, [1] weight LDS writes, [2] fetch issue
extern "C" int dta_debug_xticks(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xticks), sizeof(long long) * 16); }
extern "C" int dta_debug_xticks2(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xticks2), sizeof(long long) * 8); }
#define XS(i) xs_[i] = clock64();
#define XS_ACC { xsacc_[0] += xs_[1] - xs_[0]; xsacc_[1] += xs_[2] - xs_[1]; xsacc_[2] += xs_[3] - xs_[2]; }
#define XT(i) xt_[i] = clock64();
#define XT_ACC { _Pragma("unroll") for (int k_ = 0; k_ < 5; ++k_) xacc_[k_] += xt_[k_ + 1] - xt_[k_]; }
#else
#define XT(i)
#define XT_ACC
#define XS(i)
#define XS_ACC
#endif
#ifdef DTA_TICKS
__device__ long long g_cticks[16];
extern "C" int dta_debug_cticks(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cticks), sizeof(long long) * 16); }
#define CTICK(i) do { if (!XN && a.N == 64 && a.NC == 2 && a.stats && blockIdx.x == 100 && blockIdx.y == 0 && threadIdx.x == 0) g_cticks[i] = clock64(); } while (0)
#else
#define CTICK(i)
#endif
template <int MT, int NT, bool XN, int NWV = 8, int MINW = 1>
__global__ __launch_bounds__(NWV * 64, MINW) void k_conv3x3_bf16(ConvArgs a) {
  WGSTAMP(XN ? 0 : (a.stats ? (a.N == 64 ? 1 : 2) : -1));      // first conv, second conv, third conv (forward launches)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef DTA_TICKS
  const long long xentry_ = clock64();
#endif
  CTICK(0);
  constexpr int NW = NWV, NTHR = NW * 64;       // eight waves, or four for maps so small that 256-row tiles leave CUs idle
  constexpr int MWG = NW * MT * 32;
  constexpr int N = NT * 32;
  constexpr bool TAP_PIPE = true;
  static_assert(!(XN && MT * NT >= 6), "six accumulator tiles + the fp32 staging sets spill (see launch_conv_bf16_t)");
  int* rowtab = (int*)smem;                 // [MWG] global output row or -1
  int* plq = rowtab + MWG;                  // [MWG] (pl << 16) | q_topleft
  float* red = (float*)(plq + MWG);         // [NW][N]
  float* cmean = red + NW * N;              // [N], then a second [NW][N] reduction array (statistics epilogue)
  unsigned char* sbuf = (unsigned char*)(cmean + N + NW * N);
  const int Q = a.Q, HW = a.HW, W2 = a.W + 2;
  const int xbytes = a.ppw * Q * RB, wbytes = 9 * N * RB, stage = xbytes + wbytes;
  const bool dbuf = a.dbuf != 0;            // two LDS stages: staging of chunk k+1 overlaps the MFMAs of chunk k

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.y;
  const int pg = a.spp == 1 ? (int)blockIdx.x : (int)blockIdx.x / a.spp, split = blockIdx.x - pg * a.spp;      // (whole patches per workgroup: no division)
  const int b0 = pg * a.ppw;
  const int npatch = min(a.ppw, a.B - b0);

  // the launch's row tables (ConvArgs::tabs: [MWG] row relative to the workgroup's first patch or -1, then [MWG] plq), requested
  // at kernel entry so that the round trip runs under the first chunk's loads; partial workgroups build their own
  int tab0_[2] = {-1, -1}, tab1_[2] = {0, 0};
  static_assert(MWG <= 2 * NTHR, "two table entries per thread at most");
  if (a.tabs && a.spp == 1 && npatch == a.ppw) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + k * NTHR < MWG) { tab0_[k] = a.tabs[tid + k * NTHR]; tab1_[k] = a.tabs[MWG + tid + k * NTHR]; }
  }
  const int khalf16 = (lane >> 5) * 16;
  int abase[MT];
  const int bbase = (lane & 31) * RB + khalf16;
  const bool xc = a.x_compact != 0;
  // row tables + the zero halo of halo-free inputs: LDS-only work, done while the first chunk's loads are in flight
  // (cycle stamps of a two-chunk workgroup: tables 4.0 k cycles, then 2.3 k waiting for the first chunk)
#define DTA_TABLES \
  { \
    if (a.tabs && a.spp == 1 && npatch == a.ppw) {   /* a full workgroup: the launch's common tables, built once by the prep launch */ \
      _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_) {                                              \
        const int lr_ = tid + k_ * NTHR;                                                              \
        if (lr_ < MWG) { rowtab[lr_] = tab0_[k_] < 0 ? -1 : b0 * HW + tab0_[k_]; plq[lr_] = tab1_[k_]; } \
      }                                                                                               \
    } else conv_row_tables<MWG, NTHR>(a, rowtab, plq, reinterpret_cast<int*>(red), reinterpret_cast<int*>(cmean), b0, npatch, split); \
    if (xc || XN) {   /* (the fused-input kernel writes interior pixels only, whatever the HBM tile format) */ \
      u32x4* z = reinterpret_cast<u32x4*>(sbuf); \
      const u32x4 zero = {0u, 0u, 0u, 0u}; \
      for (int i = tid; i < xbytes / 16; i += NTHR) { z[i] = zero; if (dbuf) z[i + stage / 16] = zero; } \
    } \
    __syncthreads(); \
_Pragma("unroll") \
    for (int mt = 0; mt < MT; ++mt) { \
      int v = plq[(wave * MT + mt) * 32 + (lane & 31)]; \
      abase[mt] = ((v >> 16) * Q + (v & 0xFFFF)) * RB + khalf16; \
    } \
  }
  // (measured, round 6: building the tables UNDER the first chunk's loads instead -- issued first, 4 k of the 19 k cycles in
  //  front of the loop -- leaves this kernel at 79.9 us and makes the STEP 4-5 us slower in 4 of 4 same-box alternations)
  if (XN) DTA_TABLES

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  // the epilogue's bias: fetched here, so that its global round trip (~0.7 us) runs under the chunk loop instead of
  // standing at the head of the epilogue of a few-chunk workgroup
  float bias[NT];
  if (!XN) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 32 + (lane & 31);
      float bv = 0.f;
      if (a.bias[0]) {
        if (a.bias_mode == 1) bv = n < a.bias_split ? a.bias[0][n] : a.bias[1][n - a.bias_split];
        else bv = a.bias[g][(a.ncg > 1 ? (int)blockIdx.z : 0) * N + n];
      }
      bias[nt] = bv;
    }
  }

  // ---- staging plan: 16-byte vectors; thread t owns vectors t, t+512, ... (fixed per thread for all chunks) ----
  constexpr int XV = 4, WV = (9 * N * 2 + NTHR - 1) / NTHR;
  // input tiles in HBM: haloed [q][16] rows, or (x_compact, the network input) halo-free [pixel][16] rows whose zero
  // halo exists only in LDS (zeroed once below, never overwritten)
  const int trows = xc ? HW : Q;              // tile rows in HBM
  const int vpp = trows * 2;                  // vectors per patch tile
  const int nxv = npatch * vpp, wvec = 9 * N * 2;
  const bf16_t* xg = (const bf16_t*)a.x_tl + (size_t)g * a.x_gs + ((size_t)b0 * a.NC) * ((size_t)(a.x_compact ? HW : Q) * 16);
  // column groups (ConvArgs::ncg > 1): this workgroup computes columns [cgi * N, cgi * N + N) of NF = ncg * N
  const int ncg = a.ncg > 1 ? a.ncg : 1, cgi = ncg > 1 ? (int)blockIdx.z : 0, NF = N * ncg;
  const bf16_t* wg = (const bf16_t*)a.wp + (size_t)g * a.NC * 9 * NF * 16;
  const size_t xchunk = (size_t)trows * 16;   // elements between consecutive chunks of one patch
  // (offsets are 32-bit and relative to the workgroup's first patch: the 64-bit part is a scalar base, not a register pair per vector)
  unsigned xsrc[XV];
  int xdst[XV], wdst[WV];
#pragma unroll
  for (int u = 0; u < XV; ++u) {
    int v = min(tid + u * NTHR, max(nxv, 1) - 1);
    int pl = fastdiv(v, a.m_vpp), o = v - pl * vpp;
    int row = o >> 1;
    if (xc) { const int hh = fastdiv(row, a.m_W); row = (hh + 1) * W2 + (row - hh * a.W) + 1; }   // pixel -> haloed-grid row
    xsrc[u] = (unsigned)(((size_t)pl * a.NC) * xchunk + (size_t)o * 8);
    xdst[u] = (pl * Q + row) * RB + (o & 1) * 16;
  }
  // ---- network input read directly as fp32 NCHW (no separate pack pass): thread t owns (patch, 4-channel group,
  // pixel) quads t, t+512, ...; the 16 channel planes of a chunk are one contiguous run per patch, so consecutive lanes
  // read consecutive floats.  Four channels of a pixel are converted and written to the LDS row as one 8-byte store.
  constexpr int QV = 2 * MT;                 // quads per thread: a workgroup's MWG rows x 4 channel groups over NTHR threads
  constexpr bool xn = XN;
  const float* xf = xn ? a.x_nchw[g] + (size_t)b0 * a.Cx * HW : nullptr;
  const int nquad = xn ? npatch * 4 * HW : 0;
  unsigned qsrc[QV];
  int qdst[QV], qch[QV];
#pragma unroll
  for (int u = 0; u < QV; ++u) {
    int q = min(tid + u * NTHR, max(nquad, 1) - 1);
    int pl = fastdiv(q, a.m_4HW), rem = q - pl * 4 * HW;
    int cq = fastdiv(rem, a.m_HW), px = rem - cq * HW;
    const int hh = fastdiv(px, a.m_W);
    const int row = (hh + 1) * W2 + (px - hh * a.W) + 1;
    qsrc[u] = (unsigned)((size_t)pl * a.Cx * HW + px);    // channel 0 of the patch, relative to the workgroup's first
    qdst[u] = (pl * Q + row) * RB + cq * 8;
    qch[u] = cq * 4;
  }
  float rf[QV][4];
  bf16_t* xo = (XN && a.x_tl_out) ? (bf16_t*)a.x_tl_out + (size_t)g * a.x_gs + ((size_t)b0 * a.NC) * ((size_t)(a.x_compact ? HW : Q) * 16) : nullptr;
#pragma unroll
  for (int u = 0; u < WV; ++u) {
    int v = min(tid + u * NTHR, wvec - 1);
    wdst[u] = (v >> 1) * RB + (v & 1) * 16;
  }
  // 16-byte vector index inside a chunk's [9][NF][16] weight slab (recomputed at each fetch: shifts and masks, no registers held)
  auto wsrc = [&](int u) -> int {
    const int v = min(tid + u * NTHR, wvec - 1);
    return ncg > 1 ? (v / (2 * N)) * (2 * NF) + cgi * 2 * N + v % (2 * N) : v;
  };
  u32x4 rx[XV], rw[WV];
  CTICK(1);
#define DTA_FETCH(chunk_)                                                                             \
  {                                                                                                   \
    if (xn) {                                                                                         \
      _Pragma("unroll") for (int u = 0; u < QV; ++u)                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
          const int ch_ = min((chunk_) * 16 + qch[u] + j, a.Cx - 1);   /* padded channels: clamped, zeroed at the store */ \
          rf[u][j] = xf[qsrc[u] + (size_t)ch_ * HW];                                                  \
        }                                                                                             \
    } else {                                                                                          \
      _Pragma("unroll") for (int u = 0; u < XV; ++u)                                                  \
          rx[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xg + xsrc[u] + (size_t)(chunk_) * xchunk)); /* next reader is far (weight gradient) or none */ \
    }                                                                                                 \
    const u32x4* swp_ = reinterpret_cast<const u32x4*>(wg + (size_t)(chunk_) * 9 * NF * 16);          \
    _Pragma("unroll") for (int u = 0; u < WV; ++u) rw[u] = swp_[wsrc(u)];                             \
  }
#define DTA_STORE(sx_, sw_, chunk_)                                                                   \
  {                                                                                                   \
    if (xn) {                                                                                         \
      _Pragma("unroll") for (int u = 0; u < QV; ++u) {                                                \
        /* channels past Cx (last chunk) hold a clamped copy of channel Cx-1: their packed weights are zero, and the */ \
        /* weight-gradient reduction never reads their rows, so they need no masking */                \
        u32x2 pk_;                                                                                    \
        pk_.x = cvt_pk_bf16(rf[u][0], rf[u][1]);                                                      \
        pk_.y = cvt_pk_bf16(rf[u][2], rf[u][3]);                                                      \
        if (tid + u * NTHR < nquad) *reinterpret_cast<u32x2*>((sx_) + qdst[u]) = pk_;                 \
      }                                                                                               \
    } else {                                                                                          \
      _Pragma("unroll") for (int u = 0; u < XV; ++u)                                                  \
          if (tid + u * NTHR < nxv) *reinterpret_cast<u32x4*>((sx_) + xdst[u]) = rx[u];               \
    }                                                                                                 \
    _Pragma("unroll") for (int u = 0; u < WV; ++u)                                                    \
        if (tid + u * NTHR < wvec) *reinterpret_cast<u32x4*>((sw_) + wdst[u]) = rw[u];                \
  }
// The nine taps of a chunk, software-pipelined two taps deep: the fragment reads of tap t + 2 are issued right behind the
// MFMAs of tap t (two fragment sets, alternating), so a read has two tap times (~256 matrix-pipe cycles) to land instead of
// being requested when its MFMAs are next in line -- at any moment only ONE of a SIMD's two waves multiplies (the other
// stages), so nothing else covers the LDS latency.  The order is pinned with sched_group_barrier (DS reads 0x100, MFMA
// 0x008); the wait counts stay the compiler's.
#define DTA_TAP_LOAD(set_, tap_, sx_, sw_)                                                            \
  {                                                                                                   \
    const int toffB_ = (((tap_) / 3) * W2 + ((tap_) % 3)) * RB;                                       \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) af[set_][mt] = lds_b128((sx_), abase[mt] + toffB_); \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                 \
        bf[set_][nt] = lds_b128((sw_), bbase + ((tap_) * N + nt * 32) * RB);                          \
  }
#define DTA_COMPUTE(sx_, sw_)                                                                         \
  if constexpr (TAP_PIPE) {                                                                           \
    bf16x8 af[2][MT], bf[2][NT];                                                                      \
    DTA_TAP_LOAD(0, 0, sx_, sw_)                                                                      \
    DTA_TAP_LOAD(1, 1, sx_, sw_)                                                                      \
    _Pragma("unroll") for (int tap = 0; tap < 9; ++tap) {                                             \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                             \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tap & 1][mt], bf[tap & 1][nt], acc[mt][nt], 0, 0, 0); \
      if (tap + 2 < 9) DTA_TAP_LOAD(tap & 1, tap + 2, sx_, sw_)                                       \
    }                                                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MT + NT), 0);                                    \
    _Pragma("unroll") for (int tap = 0; tap < 9; ++tap) {                                             \
      __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                        \
      if (tap + 2 < 9) __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);                       \
    }                                                                                                 \
  } else {                                                                                            \
    _Pragma("unroll") for (int tap = 0; tap < 9; ++tap) {                                             \
      bf16x8 af[1][MT], bf[1][NT];                                                                    \
      DTA_TAP_LOAD(0, tap, sx_, sw_)                                                                  \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                             \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][mt], bf[0][nt], acc[mt][nt], 0, 0, 0); \
    }                                                                                                 \
  }

  if (XN && dbuf) {
    // fp32 input, two LDS stages: the input of chunk k+2 AND k+3 is in flight (two register sets, alternating), because
    // 16 dword loads per lane need more than one multiply phase to land; weights stay one chunk ahead
    float rg[QV][4];
#define DTA_FETCH_XF(dst_, chunk_)                                                                    \
    _Pragma("unroll") for (int u = 0; u < QV; ++u)                                                    \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
        const int ch_ = min(min((chunk_), a.NC - 1) * 16 + qch[u] + j, a.Cx - 1);                     \
        dst_[u][j] = __builtin_nontemporal_load(xf + qsrc[u] + (size_t)ch_ * HW);   /* read once per step */ \
      }
#define DTA_STORE_XF(src_, sx_)                                                                       \
    _Pragma("unroll") for (int u = 0; u < QV; ++u) {                                                  \
      u32x2 pk_;                                                                                      \
      pk_.x = cvt_pk_bf16(src_[u][0], src_[u][1]);                                                    \
      pk_.y = cvt_pk_bf16(src_[u][2], src_[u][3]);                                                    \
      if (tid + u * NTHR < nquad) *reinterpret_cast<u32x2*>((sx_) + qdst[u]) = pk_;                   \
    }
#define DTA_FETCH_W(chunk_)                                                                           \
    {                                                                                                 \
      const u32x4* swp_ = reinterpret_cast<const u32x4*>(wg + (size_t)min((chunk_), a.NC - 1) * 9 * NF * 16); \
      _Pragma("unroll") for (int u = 0; u < WV; ++u) rw[u] = swp_[wsrc(u)];                           \
    }
#define DTA_STORE_W(sw_)                                                                              \
    _Pragma("unroll") for (int u = 0; u < WV; ++u)                                                    \
        if (tid + u * NTHR < wvec) *reinterpret_cast<u32x4*>((sw_) + wdst[u]) = rw[u];
#define DTA_TILE_OUT(cx_, chunk_)                                                                     \
    if (xo) {   /* the chunk image just completed: its rows leave as the halo-free bf16 tile of (patch, chunk); plain */ \
                /* stores: nontemporal ones made this kernel 3.5 us slower (same-box A/B), the reader is far either way */ \
      _Pragma("unroll") for (int u = 0; u < XV; ++u)                                                  \
        if (tid + u * NTHR < nxv)                                                                     \
          *reinterpret_cast<u32x4*>(xo + xsrc[u] + (size_t)(chunk_) * xchunk) = *reinterpret_cast<const u32x4*>((cx_) + xdst[u]); \
    }
    DTA_FETCH_XF(rf, 0)
    DTA_FETCH_W(0)
    DTA_STORE_XF(rf, sbuf)
    DTA_STORE_W(sbuf + xbytes)
    if (a.NC > 1) {
      DTA_FETCH_XF(rf, 1)
      DTA_FETCH_W(1)
    }
    if (a.NC > 2) DTA_FETCH_XF(rg, 2)
    __syncthreads();
    unsigned char* s0 = sbuf;
    unsigned char* s1 = sbuf + stage;
    // The two waves of a SIMD (w and w + NW/2) take the two halves of a chunk interval in opposite order: the first stages
    // the next chunk (waits for its loads, converts, writes LDS) and then multiplies, the second multiplies first and
    // stages afterwards -- nobody reads the stage being written before the barrier, so the order inside the interval is
    // free, and the matrix pipe has one wave's MFMAs to run while the other wave's staging waits on memory.
    const bool late = !(a.pixel_order & 2) && __builtin_amdgcn_readfirstlane(wave) >= NW / 2;
#ifdef DTA_TICKS
    long long xt_[6] = {0, 0, 0, 0, 0, 0}, xacc_[5] = {0, 0, 0, 0, 0}, xs_[4] = {0, 0, 0, 0}, xsacc_[3] = {0, 0, 0};
    const long long xt0_ = clock64();
    if (blockIdx.x == 100 && blockIdx.y == 0 && tid == 0) g_xticks[0][6] = xt0_ - xentry_;
#endif
    for (int chunk = 0; chunk < a.NC; chunk += 2) {
      // even chunk in stage 0; chunk+1 (set rf) goes to stage 1, then rf refills with chunk+3
      XT(0)
      DTA_TILE_OUT(s0, chunk)
      XT(1)
      if (late) { DTA_COMPUTE(s0, s0 + xbytes) }
      XT(2)
      if (chunk + 1 < a.NC) {
        XS(0)
        DTA_STORE_XF(rf, s1)
        XS(1)
        DTA_STORE_W(s1 + xbytes)
        XS(2)
        if (chunk + 2 < a.NC) DTA_FETCH_W(chunk + 2)
        if (chunk + 3 < a.NC) DTA_FETCH_XF(rf, chunk + 3)
        XS(3)
        XS_ACC
      }
      XT(3)
      if (!late) { DTA_COMPUTE(s0, s0 + xbytes) }
      XT(4)
      __syncthreads();
      XT(5)
      XT_ACC
      if (chunk + 1 >= a.NC) break;
      // odd chunk in stage 1; chunk+2 (set rg) goes to stage 0, then rg refills with chunk+4
      XT(0)
      DTA_TILE_OUT(s1, chunk + 1)
      XT(1)
      if (late) { DTA_COMPUTE(s1, s1 + xbytes) }
      XT(2)
      if (chunk + 2 < a.NC) {
        XS(0)
        DTA_STORE_XF(rg, s0)
        XS(1)
        DTA_STORE_W(s0 + xbytes)
        XS(2)
        if (chunk + 3 < a.NC) DTA_FETCH_W(chunk + 3)
        if (chunk + 4 < a.NC) DTA_FETCH_XF(rg, chunk + 4)
        XS(3)
        XS_ACC
      }
      XT(3)
      if (!late) { DTA_COMPUTE(s1, s1 + xbytes) }
      XT(4)
      __syncthreads();
      XT(5)
      XT_ACC
    }
#ifdef DTA_TICKS
    if (blockIdx.x == 100 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == NW / 2)) {
      long long* o_ = g_xticks[wave ? 1 : 0];
      for (int k_ = 0; k_ < 5; ++k_) o_[k_] = xacc_[k_];
      o_[5] = clock64() - xt0_;
      for (int k_ = 0; k_ < 3; ++k_) g_xticks2[wave ? 1 : 0][k_] = xsacc_[k_];
    }
#endif
#undef DTA_TILE_OUT
#undef DTA_STORE_W
#undef DTA_FETCH_W
#undef DTA_STORE_XF
#undef DTA_FETCH_XF
  } else {
  // chunk k+1 sits in registers while chunk k is multiplied; with two LDS stages its ds_writes also overlap
  DTA_FETCH(0)
  if (!XN) DTA_TABLES
  DTA_STORE(sbuf, sbuf + xbytes, 0)
  if (a.NC > 1) DTA_FETCH(1)
  __syncthreads();
  CTICK(2);
  // (two LDS stages: the two waves of a SIMD take staging and multiplying in opposite order, as in the fused-input loop)
  const bool late2 = dbuf && !(a.pixel_order & 2) && __builtin_amdgcn_readfirstlane(wave) >= NW / 2;
  for (int chunk = 0; chunk < a.NC; ++chunk) {
    unsigned char* cx = sbuf + ((dbuf && (chunk & 1)) ? stage : 0);
    unsigned char* nx = sbuf + ((dbuf && !(chunk & 1)) ? stage : 0);
    const bool more = chunk + 1 < a.NC;
    if (xo) {   // the chunk image just completed: its rows leave as the halo-free bf16 tile of (patch, chunk)
#pragma unroll
      for (int u = 0; u < XV; ++u)
        if (tid + u * NTHR < nxv)
          *reinterpret_cast<u32x4*>(xo + xsrc[u] + (size_t)chunk * xchunk) = *reinterpret_cast<const u32x4*>(cx + xdst[u]);
    }
    if (late2) { DTA_COMPUTE(cx, cx + xbytes) }
    if (dbuf && more) {
      DTA_STORE(nx, nx + xbytes, chunk + 1)
      if (chunk + 2 < a.NC) DTA_FETCH(chunk + 2)
    }
    if (!late2) { DTA_COMPUTE(cx, cx + xbytes) }
    __syncthreads();
    if (!dbuf && more) {
      DTA_STORE(nx, nx + xbytes, chunk + 1)
      if (chunk + 2 < a.NC) DTA_FETCH(chunk + 2)
      __syncthreads();
    }
  }
  }
#undef DTA_COMPUTE
#undef DTA_TAP_LOAD
#undef DTA_TABLES
#undef DTA_STORE
#undef DTA_FETCH

  CTICK(3);
#ifdef DTA_TICKS
  const long long xloopend_ = clock64();
#endif
  // ---- epilogue: bias, store, per-workgroup (mean, M2) per column ----
  if (XN) {      // (the fused-input kernel had no register to spare for the bias during its loop)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      int n = nt * 32 + (lane & 31);
      float bv = 0.f;
      if (a.bias[0]) {
        if (a.bias_mode == 1) bv = n < a.bias_split ? a.bias[0][n] : a.bias[1][n - a.bias_split];
        else bv = a.bias[g][cgi * N + n];
      }
      bias[nt] = bv;
    }
  }
  // BatchNorm partials of this workgroup's rows, per column: (mean, M2) from ONE pass over the accumulators -- sums of
  // d = v - K and d^2 with the shift K = the column's value in the tile's first row (a sample of the distribution: the
  // cancellation in S2 - S1^2 / n then costs ~eps (1 + (mean - K)^2 / var), a few ulp; a constant column gives d = 0
  // exactly).  The two-pass form (mean first, then squared deviations: a second sweep over the accumulators behind two
  // more barriers) was 11 % of a few-chunk workgroup.
  float s1[NT], s2[NT], shiftK[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { s1[nt] = 0.f; s2[nt] = 0.f; shiftK[nt] = 0.f; }
  float* red2 = cmean + N;                  // [NW][N]
  if (a.stats) {
    if (wave == 0 && lane < 32) {
      const bool v0 = rowtab[0] >= 0;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) cmean[nt * 32 + lane] = bias[nt] + (v0 ? acc[0][nt][0] : 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) shiftK[nt] = cmean[nt * 32 + (lane & 31)];
  }
  float* yg = a.y + (size_t)g * a.y_gs + cgi * N;
  if (a.y_fmt == FMT_F32) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int lr = (wave * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int orow = rowtab[lr];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float v = acc[mt][nt][r] + bias[nt];
          if (orow >= 0) {
            yg[(size_t)orow * a.y_rs + nt * 32 + (lane & 31)] = v;
            const float d = v - shiftK[nt];
            s1[nt] += d; s2[nt] += d * d;
          }
        }
      }
    }
  } else {
    // 16-bit output rows: a lane holds ONE column of 16 rows, so neighbouring lanes trade values (rows r and r + 1 are
    // consecutive pixels): the even lane packs columns (n, n + 1) of row r, the odd lane the same pair of row r + 1.
    // The packed tile then goes through LDS (the staging buffers are free: the chunk loop ended on a barrier) and leaves
    // as 16-byte vectors, 8 per thread, contiguous in HBM -- four times fewer, four times wider stores than 4-byte ones
    // scattered from the accumulator layout (cycle stamps: 10.7 k -> the epilogue was 37 % of a few-chunk workgroup).
    // Statistics stay on the fp32 values.
    unsigned short* y16 = reinterpret_cast<unsigned short*>(a.y) + (size_t)g * a.y_gs + cgi * N;
    const bool odd = lane & 1;
    const int ncol = (lane & 31) & ~1;
    constexpr int EP = N * 2 + 16;                  // LDS row pitch in bytes (16-byte aligned, rows 4 apart miss banks)
    unsigned char* E = sbuf;
    // (the format is a compile-time constant inside the unrolled packing loop: one uniform branch out here instead of
    // one per packed pair)
    auto pack_tile = [&](auto fmt_tag) {
    constexpr int YFMT = decltype(fmt_tag)::value;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int lr0 = (wave * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int orow0 = rowtab[lr0], orow1 = rowtab[lr0 + 1];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float v0 = acc[mt][nt][r] + bias[nt], v1 = acc[mt][nt][r + 1] + bias[nt];
          const float d0 = v0 - shiftK[nt], d1 = v1 - shiftK[nt];
          if (orow0 >= 0) { s1[nt] += d0; s2[nt] += d0 * d0; }
          if (orow1 >= 0) { s1[nt] += d1; s2[nt] += d1 * d1; }
          const float got = lane_xor1(odd ? v0 : v1);      // even lane <- neighbour's row r, odd lane <- neighbour's row r + 1
          const unsigned pk = odd ? pack2_fmt(got, v1, YFMT) : pack2_fmt(v0, got, YFMT);
          *reinterpret_cast<unsigned*>(E + (lr0 + (odd ? 1 : 0)) * EP + (nt * 32 + ncol) * 2) = pk;
        }
      }
    }
    };
    CTICK(6);
    if (a.y_fmt == FMT_F16) pack_tile(std::integral_constant<int, FMT_F16>{});
    else pack_tile(std::integral_constant<int, FMT_BF16>{});
    CTICK(7);
    __syncthreads();
    CTICK(8);
    constexpr int VPR = N / 8;                      // 16-byte vectors per output row
#pragma unroll
    for (int u = 0; u < MWG * VPR / NTHR; ++u) {
      const int v = tid + u * NTHR, row = v / VPR, part = v % VPR;
      const int orow = rowtab[row];
      if (orow >= 0)
        *reinterpret_cast<u32x4*>(y16 + (size_t)orow * a.y_rs + part * 8) = *reinterpret_cast<const u32x4*>(E + row * EP + part * 16);
    }
  }
  CTICK(4);
#ifdef DTA_TICKS
  if (XN && blockIdx.x == 100 && blockIdx.y == 0 && tid == 0) g_xticks[0][7] = clock64() - xloopend_;
#endif
  if (a.stats == nullptr) return;
  const int cnt = (a.spp == 1) ? npatch * HW : min(MWG, HW - split * MWG);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float u1 = s1[nt] + __shfl_xor(s1[nt], 32), u2 = s2[nt] + __shfl_xor(s2[nt], 32);
    if (lane < 32) { red[wave * N + nt * 32 + lane] = u1; red2[wave * N + nt * 32 + lane] = u2; }
  }
  __syncthreads();
  if (tid < N) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { t1 += red[w * N + tid]; t2 += red2[w * N + tid]; }
    const float dm = t1 / (float)cnt;
    const float mean = cmean[tid] + dm;
    float m2 = t2 - t1 * dm;
    m2 = m2 > 0.f ? m2 : 0.f;
    cmean[tid] = mean;
    float* o = a.stats + (((size_t)g * gridDim.x + blockIdx.x) * NF + cgi * N + tid) * 2;
    if (a.fan_count) fan_store2(o, cmean[tid], m2);
    else { o[0] = cmean[tid]; o[1] = m2; }
  }
  CTICK(5);
#ifdef DTA_TICKS
  if (XN && blockIdx.x == 100 && blockIdx.y == 0 && tid == 0) g_xticks[1][7] = clock64() - xloopend_;
#endif
  // no finalize launch: the last workgroup of each logical group folds the group's rows (kernels.h); the staging area is free
  if (a.fan_count) conv_stats_fanin<NTHR>(a, g, N, HW, MWG, reinterpret_cast<double*>(sbuf), reinterpret_cast<int*>(cmean));
}

template <int MT, int NT, int NW = 8, int MINW = 1>
static int launch_conv_bf16_t(ConvArgs a, int G, hipStream_t st) {
  constexpr int MWG = NW * MT * 32, N = NT * 32;
  int nwg;
  conv_geometry(a.HW, MWG, a.B, &a.ppw, &a.spp, &nwg);
  if (a.tabs_rows != MWG) a.tabs = nullptr;      // (tables of another tile: every workgroup builds its own)
  a.m_vpp = fastdiv_magic((a.x_compact ? a.HW : a.Q) * 2); a.m_W = fastdiv_magic(a.W); a.m_HW = fastdiv_magic(a.HW); a.m_4HW = fastdiv_magic(4 * a.HW);
  size_t tab = (size_t)MWG * 8 + (size_t)17 * N * 4, stage = ((size_t)a.ppw * a.Q + (size_t)9 * N) * RB;
  // few chunks (the 32- and 64-channel layers): one LDS stage, so that two or three workgroups share a CU and overlap
  // each other's prologue / epilogue instead of double-buffering a two-iteration loop
  a.dbuf = tab + 2 * stage <= 160 * 1024 && a.NC > 4;   // measured: conv2's input-gradient conv 28 -> 22 us
  size_t lds = tab + (a.dbuf ? 2 : 1) * stage;
  if (a.y_fmt != FMT_F32) {   // the 16-bit epilogue transposes the output tile through the staging area
    const size_t epi = tab + (size_t)MWG * (N * 2 + 16);
    if (epi > lds) lds = epi;
  }
  if (lds > 160 * 1024) { dta_set_error("conv3x3(bf16): LDS need %zu B exceeds 160 KiB (H=%d W=%d)", lds, a.H, a.W); return 1; }
  if (a.ppw * a.Q * 2 > 4 * NW * 64) { dta_set_error("conv3x3(bf16): %dx%d tile exceeds the staging plan", a.H, a.W); return 1; }
  static DevOnce attr_once;      // (function attributes are per device)
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)k_conv3x3_bf16<MT, NT, false, NW, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (MT * NT < 6) hipFuncSetAttribute((const void*)k_conv3x3_bf16<MT, NT, true, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if (a.x_nchw[0]) {
    if (a.spp != 1) { dta_set_error("conv3x3(bf16): the fused-input first conv needs whole patches per workgroup"); return 1; }
    // (six accumulator tiles + two fp32 staging sets do not fit 256 registers -- the instantiation spilled 26..71 of them --
    //  so a 576-row x 64-column first conv is not offered with the fused input: capi.hip's fused_input() plans the pack job)
    if constexpr (MT * NT < 6) hipLaunchKernelGGL((k_conv3x3_bf16<MT, NT, true, NW>), dim3(nwg, G), dim3(NW * 64), lds, st, a);
    else { dta_set_error("conv3x3(bf16): no fused-input kernel for %d-row x %d-column workgroups", MWG, N); return 1; }
  } else {
    hipLaunchKernelGGL((k_conv3x3_bf16<MT, NT, false, NW, MINW>), dim3(nwg, G, a.ncg > 1 ? a.ncg : 1), dim3(NW * 64), lds, st, a);
  }
  DTA_CHECK_LAUNCH("k_conv3x3_bf16");
  return 0;
}

int conv_mwg_bf16(int N, int HW) {
  if (N == 32 && HW > 512) {
    const int r512 = (HW + 511) / 512 * 512, r576 = (HW + 575) / 576 * 576;
    if (r576 < r512) return 576;
  }
  return conv_mwg(N);
}

// Output rows per workgroup the launcher below will use for `a` (geometry fields only: N, HW, B, mwg, stats, ncg): the
// prep launch builds the row tables of exactly that tile (capi.hip).  Keep in step with the switch below.
int conv_bf16_rows(const ConvArgs& a, int G) {
  switch (a.N) {
    case 32: return (a.mwg == 576 || (a.stats == nullptr && conv_mwg_bf16(32, a.HW) == 576)) ? 576 : 512;
    case 64: {
      int ppw, spp, nwg;
      conv_geometry(a.HW, 512, a.B, &ppw, &spp, &nwg);
      if (a.stats == nullptr && nwg * G <= 128) return 256;
      return a.mwg == 256 ? 256 : a.mwg == 576 ? 576 : 512;
    }
    case 128: return (a.ncg == 2 && a.mwg == 576) ? 576 : 256;
  }
  return 0;
}

template <>
int launch_conv3x3<bf16_t>(const ConvArgs& a, int G, hipStream_t st) {
  switch (a.N) {     // workgroup rows must match the plan's choice (conv_mwg_bf16): 512 or 576 for N<=64, 256 for N=128
    case 32:
      // (an input-gradient conv has no BN partials, so the plan's geometry is not binding: pick the tighter tile here)
      if (a.mwg == 576 || (a.stats == nullptr && conv_mwg_bf16(32, a.HW) == 576)) return launch_conv_bf16_t<3, 1, 6>(a, G, st);
      return launch_conv_bf16_t<2, 1>(a, G, st);
    case 64: {
      // an input-gradient conv (no BN partials, so the plan's workgroup geometry is not binding) over few rows:
      // 256-row workgroups double the number of busy CUs
      int ppw, spp, nwg;
      conv_geometry(a.HW, 512, a.B, &ppw, &spp, &nwg);
      if (a.stats == nullptr && nwg * G <= 128) return launch_conv_bf16_t<1, 2>(a, G, st);
      // 256-row workgroups need 114 VGPRs against 184: two workgroups share a CU instead of one (the few-chunk layers are
      // all prologue / epilogue, so the overlap of two workgroups is worth more than the larger tile)
      // (three 256-row workgroups per CU -- __launch_bounds__(512, 6), 80 registers -- spill 72 registers: not an option)
      if (a.mwg == 256) return launch_conv_bf16_t<1, 2>(a, G, st);
      // (a 24x24 map as one 576-row workgroup; two 32-column groups of <3,1> tiles instead were measured slower: 1.046 -> 1.063 ms)
      if (a.mwg == 576) return launch_conv_bf16_t<3, 2, 6>(a, G, st);
      return launch_conv_bf16_t<2, 2>(a, G, st);
    }
    case 128:
      // (four-wave, 128-row workgroups for the 5x5 maps were measured: every workgroup stages the full weight set, so
      // halving the rows doubles that traffic -- third conv 18.9 -> 18.0 us, its input-gradient conv 17 -> 24 us; not used)
      // ncg == 2 (the plan's choice): two 64-column groups per row tile -- twice the workgroups, each staging HALF the
      // weight slab per chunk (the layer is all weight staging: 36 KiB per chunk against a 6 KiB input tile)
      if (a.ncg == 2) { ConvArgs h = a; h.N = 64; return h.mwg == 576 ? launch_conv_bf16_t<3, 2, 6>(h, G, st) : launch_conv_bf16_t<1, 2>(h, G, st); }
      return launch_conv_bf16_t<1, 4>(a, G, st);
  }
  dta_set_error("conv3x3: unsupported output width %d", a.N);
  return 1;
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
constexpr int RW = 32;    // weight-gradient LDS rows stay compact (32 B): with chunk tiles 128 B (mod 256 B) apart the two
                          // 16-lane groups of a ds_read_b64_tr_b16 half-wave hit disjoint banks for any row offset


// One k-step (16 window rows) of all nine taps of a (32 input channels x 32 output columns) tile.
// The three taps of one kernel row read the same X rows shifted by one: their A fragments (8 consecutive rows per
// lane) are cut out of ONE 12-row transposing read (3 x ds_read_b64_tr_b16: rows r-1 .. r+10) with 16-bit funnel
// shifts, and the dY fragment is shared by all nine MFMAs -> 11 LDS reads per 9 MFMAs (the per-tap scheme needed 20).
__device__ __forceinline__ u32x2 lds_tr4w(const unsigned char* base, int off) {
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(base + off));
  return __builtin_bit_cast(u32x2, v);
}
struct WgradFrags { bf16x8 b; u32x2 r[3][3]; };
__device__ __forceinline__ void wgrad_kstep9_load(WgradFrags& f, const unsigned char* bx, const unsigned char* by,
                                                  const int (&a_dy)[3], int b_off, int koff) {
  f.b = lds_tr8w(by, b_off + koff);
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int h = 0; h < 3; ++h) f.r[d][h] = lds_tr4w(bx, a_dy[d] + koff + h * 4 * RW);
}
__device__ __forceinline__ void wgrad_kstep9_mma(const WgradFrags& f, f32x16 (&acc)[9]) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const unsigned R0 = f.r[d][0][0], R1 = f.r[d][0][1], R2 = f.r[d][1][0], R3 = f.r[d][1][1], R4 = f.r[d][2][0];
    const u32x4 fm = {R0, R1, R2, R3};                                     // rows r-1 .. r+6   (dx = -1)
    const u32x4 fz = {__builtin_amdgcn_alignbit(R1, R0, 16), __builtin_amdgcn_alignbit(R2, R1, 16),
                      __builtin_amdgcn_alignbit(R3, R2, 16), __builtin_amdgcn_alignbit(R4, R3, 16)};   // r .. r+7
    const u32x4 fp = {R1, R2, R3, R4};                                     // rows r+1 .. r+8   (dx = +1)
    acc[d * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fm), f.b, acc[d * 3 + 0], 0, 0, 0);
    acc[d * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fz), f.b, acc[d * 3 + 1], 0, 0, 0);
    acc[d * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp), f.b, acc[d * 3 + 2], 0, 0, 0);
  }
}

// developer instrumentation (-DDTA_TICKS): per-wave cycle totals of the staging / k-step / barrier phases
#ifdef DTA_TICKS
__device__ long long g_wticks[64];
extern "C" int dta_debug_wticks(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wticks), sizeof(long long) * 64); }
#define WTICK_DECL long long wt_[5] = {0, 0, 0, 0, 0}, wacc_[4] = {0, 0, 0, 0}; const long long wt0_ = clock64();
#define WTICK(i) wt_[i] = clock64(); if (i == 4) { wacc_[0] += wt_[1] - wt_[0]; wacc_[1] += wt_[2] - wt_[1]; wacc_[2] += wt_[3] - wt_[2]; wacc_[3] += wt_[4] - wt_[3]; }
#define WTICK_DUMP if (blockIdx.x == 17 && lane == 0 && a.N == 64 && a.NCx > 8) { long long* o_ = g_wticks + wave * 8; \
    o_[0] = wacc_[0]; o_[1] = wacc_[1]; o_[2] = wacc_[2]; o_[3] = clock64() - wt0_; o_[4] = niter; o_[5] = wacc_[3]; }
#else
#define WTICK_DECL
#define WTICK(i)
#define WTICK_DUMP
#endif
// The workgroup program; bx = index of this workgroup among the job's workgroups (a launch may carry two jobs: below).
template <int CT, int NTT, bool BIGW, bool STACK = false, bool D2 = false>
__device__ __forceinline__ void wgrad_bf16_body(const WgradArgs& a, const int bx, unsigned char* smem) {
  WGSTAMP(CT == 2 ? 3 : -1);      // first conv's weight gradient
#ifdef DTA_TICKS
  const long long wentry_ = clock64();
#endif
  constexpr int NTHR = 512;
  // CT 32-channel input tiles x NTT 32-column tiles per workgroup = PAIRS wave tiles; the 8 waves are PAIRS tiles x
  // KS k-slices (KS = 2 for the usual 4 tiles; 4 when the layer has only 32 input channels: CT = 1, NTT = 2)
  constexpr int PAIRS = CT * NTT, KS = 8 / PAIRS;
  static_assert(PAIRS == 2 || PAIRS == 4, "8 waves = tiles x k-slices");
  constexpr int N = NTT * 32;
  constexpr int XCH = CT * 2, YCH = NTT * 2;
  // The K dimension (haloed-grid rows q0..q1 of a patch) is walked in bands of a.bl rows; a band's LDS window holds
  // tile rows [band*bl, band*bl + WR) of the X and dY chunk tiles (WR = bl + 2*(W+3) covers the +-(W+3) tap shifts),
  // so k-step addresses are the same for every band.  11x11 patches are a single band.
  const int Q = a.Q, WR = a.wr, W2 = a.W + 2;
  const int xbytes = XCH * WR * RW, stage = (XCH + YCH) * WR * RW;
  const bool dbuf = a.dbuf != 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware placement: workgroup id b runs on XCD b % 8 (observed dispatch rule, speed only); consecutive LOGICAL
  // indices (the channel groups of one batch split, which all re-read the same dY tiles) are mapped to one XCD so
  // that those re-reads hit its L2.
  const int ngr = a.ngroups > 0 ? a.ngroups : 1;      // column groups: this workgroup's N columns start at ng * N
  const int total = a.cgroups * ngr * a.S * a.G;
  const int per_xcd = (total + 7) / 8;
  const int logical = (bx % 8) * per_xcd + bx / 8;
  if (logical >= total) return;
  const int cg = logical % a.cgroups, ng = (logical / a.cgroups) % ngr, s = (logical / (a.cgroups * ngr)) % a.S,
            g = logical / (a.cgroups * ngr * a.S);
  // wave = ((c-tile, n-tile) pair, k slice): the KS waves of a pair split the k-steps (ks = slice, slice + KS, ...) and
  // hold partial sums of the same nine tap tiles; waves w and w+4 share a SIMD and belong to different slices
  const int pair = wave % PAIRS, khalf = wave / PAIRS;   // khalf: k-slice index 0 .. KS-1
  const int ct = pair / NTT, nt = pair % NTT;
  const int chunk0 = cg * XCH;
  const int nxch = max(0, min(XCH, a.NCx - chunk0));

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int q0 = a.W + 3, q1 = Q - a.W - 3;
  const int gq = lane >> 4, li = lane & 15;
  const int lane_off = (8 * (gq >> 1) + (li >> 2)) * RW + (li & 3) * 8;
  const int b_off = ((nt * 2 + (gq & 1)) * WR + q0) * RW + lane_off;
  int a_dy[3];   // first row of the 12-row read of kernel row dy: q0 + dy*W2 - 1  (>= 0: q0 = W2 + 1)
#pragma unroll
  for (int d = 0; d < 3; ++d) a_dy[d] = ((ct * 2 + (gq & 1)) * WR + q0 + (d - 1) * W2 - 1) * RW + lane_off;

  // ---- staging plan (band independent): thread t owns window vectors t, t+512, ... of the X part and of the dY part
  // vectors per thread: a chunk window has 2*WR vectors; WR <= 192 (11x11 patches: 172) or <= 256 (BIGW)
  constexpr int XV = BIGW ? XCH : (3 * XCH + 3) / 4, YV = BIGW ? YCH : (3 * YCH + 3) / 4;
  const int vpc = WR * 2;                      // 16-byte vectors per chunk window
  // X tiles may be halo-free in HBM (x_compact: the network input, single band): HW rows per chunk, scattered to their
  // haloed-grid rows of the LDS window; the halo rows stay zero from the initial fill
  const bool xc = a.x_compact != 0;
  const int HWp = a.H * a.W;
  const int vpcx = xc ? HWp * 2 : vpc, xtrows = xc ? HWp : Q;
  const bool yc = a.y_compact != 0;            // likewise the dY tiles
  const int vpcy = yc ? HWp * 2 : vpc, ytrows = yc ? HWp : Q;
  // small maps (5x5): ppi patches of the workgroup's list are stacked Q rows apart in ONE window and swept by one run
  // of k-steps -- a patch of 33 centre rows is two or three k-steps, i.e. less than one per wave and barrier (halo-free
  // tiles only: the rows between the stacked patches are their zero halos, so the sweep may run across them)
  // (a compile-time variant: the plain kernels are at their register limit and must not carry any of this)
  const int ppi = STACK ? a.ppi : 1;
  const int nxv1 = nxch * vpcx, nyv1 = YCH * vpcy;        // vectors of one patch
  const int nxv = ppi * nxv1, nyv = ppi * nyv1;
  const bf16_t* xg = (const bf16_t*)a.x_tl + (size_t)g * a.x_gs;
  const bf16_t* yg = (const bf16_t*)a.dy_tl + (size_t)g * a.dy_gs;
  const size_t xpatch = (size_t)a.NCx * xtrows * 16, ypatch = (size_t)a.NCy * ytrows * 16;
  // per vector: element offset inside the patch's tile and window row.  (The LDS image is linear: vector v of a part
  // lives at byte 16*v.)  Vectors past the end of the part are not stored; their loads are pointed at a valid row.
  int xsrc[XV], ysrc[YV], xrow[XV], yrow[YV], xdst[XV], ydst[YV];
  int xpp[STACK ? XV : 1], ypp[STACK ? YV : 1];   // which of the window's stacked patches a vector belongs to
#pragma unroll
  for (int u = 0; u < XV; ++u) {
    int v = min(tid + u * NTHR, max(nxv, 1) - 1);
    const int pp = (STACK && nxv1 > 0) ? v / nxv1 : 0;
    v -= pp * nxv1;
    int ch = v / vpcx, o = v - ch * vpcx;
    int row = o >> 1;
    xsrc[u] = ((nxch > 0 ? chunk0 : 0) + ch) * xtrows * 16 + o * 8;
    if (xc) { const int hh = row / a.W; row = (hh + 1) * W2 + (row - hh * a.W) + 1; }   // pixel -> haloed-grid row
    xrow[u] = xc ? 0 : row;                    // compact rows always exist
    xdst[u] = (ch * WR + pp * Q + row) * RW + (o & 1) * 16;
    if constexpr (STACK) { xpp[u] = pp; xsrc[u] += pp * a.S * (int)xpatch; }     // (small maps: fits 32 bits by far)
  }
#pragma unroll
  for (int u = 0; u < YV; ++u) {
    int v = min(tid + u * NTHR, nyv - 1);
    const int pp = STACK ? v / nyv1 : 0;
    v -= pp * nyv1;
    int ch = v / vpcy, o = v - ch * vpcy;
    int row = o >> 1;
    ysrc[u] = (a.ych0 + ng * YCH + ch) * ytrows * 16 + o * 8;
    if (yc) { const int hh = row / a.W; row = (hh + 1) * W2 + (row - hh * a.W) + 1; }   // pixel -> haloed-grid row
    yrow[u] = yc ? 0 : row;
    ydst[u] = (ch * WR + pp * Q + row) * RW + (o & 1) * 16;
    if constexpr (STACK) { ypp[u] = pp; ysrc[u] += pp * a.S * (int)ypatch; }
  }
  // D2: TWO windows in flight in registers (sets rx/ry and rx2/ry2).  With one set a load has one iteration (one patch:
  // 1.1 us of MFMA work for the first conv, less for the others) to land before it is written to LDS -- shorter than the
  // loaded-memory latency, so the waves stalled on vmcnt at their store phase; with two it has two iterations.
  u32x4 rx[XV], ry[YV];
  u32x4 rx2[D2 ? XV : 1], ry2[D2 ? YV : 1];
  // Branch-free fetch: wave-uniform base (patch, band) + per-lane 32-bit offset.  A window row that falls beyond the
  // tile (last band; rows Q..WR-1 of an 11x11 patch) is redirected to row 0 of its chunk, a halo row, i.e. zeros.
  // (LDS-DMA staging was measured slower here: its LDS writes stall the transposing fragment reads.)
  // A stacked patch beyond the workgroup's list (last window) is stored as zeros, not loaded.
#define DTA_FETCH(RX_, RY_, b_, band_)                                                                            \
  {                                                                                                               \
    const int r0_ = (band_) * a.bl;                                                                               \
    const bf16_t* xb_ = xg + (size_t)(b_) * xpatch + (size_t)r0_ * 16;                                            \
    const bf16_t* yb_ = yg + (size_t)(b_) * ypatch + (size_t)r0_ * 16;                                            \
    const int npi_ = STACK ? min(ppi, (a.B - (b_) + a.S - 1) / a.S) : 1;                                          \
    const u32x4 zero_ = {0u, 0u, 0u, 0u};                                                                         \
    _Pragma("unroll") for (int u = 0; u < XV; ++u) {                                                              \
      const int off_ = (xrow[u] + r0_ < Q) ? xsrc[u] : xsrc[u] - (xrow[u] + r0_) * 16;                            \
      if (!STACK || xpp[STACK ? u : 0] < npi_) RX_[u] = *reinterpret_cast<const u32x4*>(xb_ + off_);              \
      else RX_[u] = zero_;                                                                                        \
    }                                                                                                             \
    _Pragma("unroll") for (int u = 0; u < YV; ++u) {                                                              \
      const int off_ = (yrow[u] + r0_ < Q) ? ysrc[u] : ysrc[u] - (yrow[u] + r0_) * 16;                            \
      if (!STACK || ypp[STACK ? u : 0] < npi_) RY_[u] = *reinterpret_cast<const u32x4*>(yb_ + off_);              \
      else RY_[u] = zero_;                                                                                        \
    }                                                                                                             \
  }
#define DTA_STORE(RX_, RY_, base_)                                                                                \
  {                                                                                                               \
    _Pragma("unroll") for (int u = 0; u < XV; ++u)                                                                \
        if (tid + u * NTHR < nxv) *reinterpret_cast<u32x4*>((base_) + xdst[u]) = RX_[u];                          \
    _Pragma("unroll") for (int u = 0; u < YV; ++u)                                                                \
        if (tid + u * NTHR < nyv) *reinterpret_cast<u32x4*>((base_) + xbytes + ydst[u]) = RY_[u];                 \
  }
  // flattened (patch, band) iteration space of this workgroup
  const int npb = (a.B - s + a.S - 1) / a.S;
  const int niter = ppi > 1 ? (npb + ppi - 1) / ppi : npb * a.nbands;      // (stacked windows: single-band patches only)
#define DTA_ITER_B(it_) (ppi > 1 ? s + (it_) * ppi * a.S : s + ((it_) / a.nbands) * a.S)
#define DTA_ITER_BAND(it_) (ppi > 1 ? 0 : (it_) % a.nbands)
  // window it + 1 is written to LDS and the fetch of window it + AHEAD issued during iteration it; the register set
  // that carries window w is set (w & 1) when D2 (rx2/ry2 for odd windows), rx/ry otherwise
  constexpr int AHEAD = D2 ? 3 : 2;
  if (niter > 0) DTA_FETCH(rx, ry, DTA_ITER_B(0), DTA_ITER_BAND(0))
  {  // zero everything once (absent chunks and halo rows stay zero for the whole kernel) -- under the first loads
    u32x4 z = {0, 0, 0, 0};
    u32x4* d = reinterpret_cast<u32x4*>(smem);
    int tot = (dbuf ? 2 : 1) * stage / 16;
    for (int v = tid; v < tot; v += NTHR) d[v] = z;
  }
  if (niter > 0) {
    __syncthreads();                       // zero fill complete before the first window lands
    DTA_STORE(rx, ry, smem)
    if (D2) {
      if (niter > 1) DTA_FETCH(rx2, ry2, DTA_ITER_B(1), DTA_ITER_BAND(1))
      if (niter > 2) DTA_FETCH(rx, ry, DTA_ITER_B(2), DTA_ITER_BAND(2))
    } else {
      if (niter > 1) DTA_FETCH(rx, ry, DTA_ITER_B(1), DTA_ITER_BAND(1))
    }
  }
  __syncthreads();
  WTICK_DECL
#ifdef DTA_TICKS
  if (blockIdx.x == 17 && threadIdx.x == 0 && a.N == 64 && a.NCx > 8) g_wticks[6] = wt0_ - wentry_;      // entry -> loop
#endif
  // one iteration; RX_/RY_ = the register set that holds window it + 1 (and then receives window it + AHEAD)
#define DTA_WGRAD_ITER(RX_, RY_, it)                                                                              \
  {                                                                                                               \
    unsigned char* cur = smem + ((dbuf && ((it) & 1)) ? stage : 0);                                               \
    unsigned char* nxt = smem + ((dbuf && !((it) & 1)) ? stage : 0);                                              \
    const bool more = (it) + 1 < niter;                                                                           \
    WTICK(0)                                                                                                      \
    const int rem = (q1 - q0) - DTA_ITER_BAND(it) * a.bl;                                                         \
    /* stacked window: the sweep runs from the first patch's first centre row to the last present patch's last */ \
    const int nks = ppi > 1 ? ((min(ppi, npb - (it) * ppi) - 1) * Q + (q1 - q0) + 15) / 16 : (min(a.bl, rem) + 15) / 16; \
    /* The two waves of a SIMD (k halves 0 and 1) run their phases in opposite order: while one issues its LDS */  \
    /* writes and global loads for the next window, the other keeps the matrix core busy. */                      \
    const bool stage_now = dbuf && more;                                                                          \
    if ((khalf & 1) == 1 && stage_now) {                                                                          \
      DTA_STORE(RX_, RY_, nxt)                                                                                    \
      if ((it) + AHEAD < niter) DTA_FETCH(RX_, RY_, DTA_ITER_B((it) + AHEAD), DTA_ITER_BAND((it) + AHEAD))        \
    }                                                                                                             \
    WTICK(1)                                                                                                      \
    /* (requesting k-step j + 1's fragments before k-step j's MFMAs -- straight-line, two fragment sets -- was */   \
    /* measured: as plain code the scheduler / wait-count pass re-serialise it and it spills, 66 -> 77 us; with the */ \
    /* order pinned by empty asm blocks (next set's reads, then this set's MFMAs without waits) still 71 -> 75 us) */ \
    _Pragma("unroll 1") for (int ks = khalf; ks < nks; ks += KS) {                                                \
      WgradFrags f;                                                                                               \
      wgrad_kstep9_load(f, cur, cur + xbytes, a_dy, b_off, ks * 16 * RW);                                         \
      wgrad_kstep9_mma(f, acc);                                                                                   \
    }                                                                                                             \
    WTICK(2)                                                                                                      \
    if ((khalf & 1) == 0 && stage_now) {                                                                          \
      DTA_STORE(RX_, RY_, nxt)                                                                                    \
      if ((it) + AHEAD < niter) DTA_FETCH(RX_, RY_, DTA_ITER_B((it) + AHEAD), DTA_ITER_BAND((it) + AHEAD))        \
    }                                                                                                             \
    WTICK(3)                                                                                                      \
    __syncthreads();                                                                                              \
    WTICK(4)                                                                                                      \
    if (!dbuf && more) {                                                                                          \
      DTA_STORE(RX_, RY_, nxt)                                                                                    \
      if ((it) + AHEAD < niter) DTA_FETCH(RX_, RY_, DTA_ITER_B((it) + AHEAD), DTA_ITER_BAND((it) + AHEAD))        \
      __syncthreads();                                                                                            \
    }                                                                                                             \
  }
  if constexpr (D2) {
    for (int it = 0; it < niter; it += 2) {
      DTA_WGRAD_ITER(rx2, ry2, it)                       // window it + 1 is odd
      if (it + 1 < niter) DTA_WGRAD_ITER(rx, ry, it + 1)
    }
  } else {
    for (int it = 0; it < niter; ++it) DTA_WGRAD_ITER(rx, ry, it)
  }
#undef DTA_WGRAD_ITER
#undef DTA_ITER_BAND
#undef DTA_ITER_B
#undef DTA_STORE
#undef DTA_FETCH
  WTICK_DUMP
#ifdef DTA_TICKS
  const long long wloopend_ = clock64();
#endif
  // the odd-k waves hand their partial sums to their even-k partners through LDS (the staging buffers are free now:
  // the loop ended on a barrier), two tap tiles per pass
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j0 = 0; j0 < 9; j0 += 2) {
    if (khalf != 0) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        if (j0 + jj < 9) {
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(((pair * (KS - 1) + khalf - 1) * 2 + jj) * 16 + r) * 64 + lane] = acc[j0 + jj][r];
        }
    }
    __syncthreads();
    if (khalf == 0) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        if (j0 + jj < 9) {
#pragma unroll
          for (int q = 0; q < KS - 1; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j0 + jj][r] += red[(((pair * (KS - 1) + q) * 2 + jj) * 16 + r) * 64 + lane];
        }
    }
    __syncthreads();
  }
#ifdef DTA_TICKS
  if (blockIdx.x == 17 && threadIdx.x == 0 && a.N == 64 && a.NCx > 8) g_wticks[7] = clock64() - wloopend_;      // k-slice hand-over
#endif
  if (khalf != 0) return;
  // partial[g][tap][c][s][n]: the S partial sums of one output row are contiguous for the reduction
  float* out = a.partial + (size_t)g * 9 * a.Cpad * a.S * a.N + (size_t)s * a.N + ng * N;     // (a.N = ngr * N columns per row)
#pragma unroll
  for (int j = 0; j < 9; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int c = cg * CT * 32 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (c < a.Cpad) __builtin_nontemporal_store(acc[j][r], &out[((size_t)j * a.Cpad + c) * a.S * a.N + nt * 32 + (lane & 31)]);
    }
  }
#ifdef DTA_TICKS
  if (blockIdx.x == 17 && threadIdx.x == 0 && a.N == 64 && a.NCx > 8) {
    g_wticks[14] = clock64() - wloopend_;      // ... + slab stores issued
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
    g_wticks[15] = clock64() - wloopend_;      // ... + slab stores performed
  }
#endif
}

template <int CT, int NTT, bool BIGW, bool STACK = false, bool D2 = false>
__global__ __launch_bounds__(512, 1) void k_conv_wgrad_bf16(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  wgrad_bf16_body<CT, NTT, BIGW, STACK, D2>(a, blockIdx.x, smem);
}

// The weight gradients of the SECOND and THIRD conv in one launch (both are <1,2> programs: plain windows for the 11x11
// maps, stacked patches for the 5x5 maps).  Alone, each is dominated by what a workgroup pays once -- LDS clear, first
// window, the k-slice hand-over and the slab epilogue, ~15 us of a 24 / 30 us launch -- and each needs the whole GPU
// to itself (one 512-thread workgroup of 254 registers per CU).  Side by side, each job gets a share of the CUs, its
// workgroups walk proportionally more patches, and the fixed part is paid once for both: measured model
// 15.3 + 8.7 (256 / n2) and 14.7 + 15.2 (256 / n3) us -> ~40 us together against 54 one after the other; the slabs
// (one per batch split) shrink with the workgroup counts.  Blocks [0, na) run job a, the rest job b; na is a multiple
// of 8, so both keep the XCD-aware placement of their batch splits.
// (BIG = false: plain windows + stacked patches, the 11x11 networks; BIG = true: both multi-band / wide windows, the
// 24x24 crops of the year models)
template <bool BIG>
__global__ __launch_bounds__(512, 1) void k_conv_wgrad_bf16_pair(WgradArgs a, WgradArgs b, int na) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((int)blockIdx.x < na) wgrad_bf16_body<1, 2, BIG, false, false>(a, blockIdx.x, smem);
  else wgrad_bf16_body<1, 2, BIG, !BIG, false>(b, blockIdx.x - na, smem);
}

// The first conv's weight gradient with the data-parallel exchange's head segment riding along: blocks [0, nmain) are the
// weight gradient (one 512-thread workgroup per CU: 240 of the 256 CUs for Hang2020, 216 for a three-year ensemble), the
// XCHG_SIDE_WGS blocks behind them pull and sum this rank's shard of every rank's head bucket -- everything but the first
// convs' weights, complete before this launch -- through peer memory while the matrix cores work (xchg_dev.h; reference
// train.py:89-98: DDP overlaps the gradient all-reduce with the backward).  The main workgroups never wait for the side ones.
// Instantiated for the programs a first conv resolves to: <2,2,D2> (Hang2020: both branches' 64 columns, 11x11),
// <4,1> plain windows (a spectral network / the years of an ensemble, 11x11) and <4,1> wide windows (24x24 crops).
template <int CT, int NTT, bool BIGW, bool STACK, bool D2>
__global__ __launch_bounds__(512, 1) void k_conv_wgrad_bf16_xchg(WgradArgs a, XchgArgs side, int nmain) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((int)blockIdx.x < nmain) { wgrad_bf16_body<CT, NTT, BIGW, STACK, D2>(a, blockIdx.x, smem); return; }
  xchg_side_job(side, (int)blockIdx.x - nmain, reinterpret_cast<int*>(smem));
}

// kernel variant a launch resolves to
enum { WV_PLAIN = 0, WV_BIGW = 1, WV_STACK = 2, WV_D2 = 3 };

template <int CT, int NTT>
static int resolve_wgrad_bf16(const WgradArgs& a, int G, int cgroups, WgradArgs& a2, size_t& lds, int& variant, int& total);

template <int CT, int NTT>
static int launch_wgrad_bf16_t(const WgradArgs& a, int G, int cgroups, hipStream_t st) {
  WgradArgs a2;
  size_t lds;
  int variant, total;
  if (resolve_wgrad_bf16<CT, NTT>(a, G, cgroups, a2, lds, variant, total)) return 1;
  static DevOnce attr_once;      // (function attributes are per device)
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)k_conv_wgrad_bf16<CT, NTT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_conv_wgrad_bf16<CT, NTT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (CT == 1) hipFuncSetAttribute((const void*)k_conv_wgrad_bf16<CT, NTT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (CT == 2 && NTT == 2) hipFuncSetAttribute((const void*)k_conv_wgrad_bf16<CT, NTT, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const dim3 grid(8 * ((total + 7) / 8));
  if (variant == WV_STACK) {
    if constexpr (CT == 1) hipLaunchKernelGGL((k_conv_wgrad_bf16<CT, NTT, false, true>), grid, dim3(512), lds, st, a2);
  } else if (variant == WV_D2) {
    if constexpr (CT == 2 && NTT == 2) hipLaunchKernelGGL((k_conv_wgrad_bf16<CT, NTT, false, false, true>), grid, dim3(512), lds, st, a2);
  } else if (variant == WV_PLAIN) hipLaunchKernelGGL((k_conv_wgrad_bf16<CT, NTT, false>), grid, dim3(512), lds, st, a2);
  else hipLaunchKernelGGL((k_conv_wgrad_bf16<CT, NTT, true>), grid, dim3(512), lds, st, a2);
  DTA_CHECK_LAUNCH("k_conv_wgrad_bf16");
  return 0;
}

template <int CT, int NTT>
static int resolve_wgrad_bf16(const WgradArgs& a, int G, int cgroups, WgradArgs& a2, size_t& lds, int& variant, int& total) {
  a2 = a;
  // band plan: window rows WR = bl + 2*(W+3), WR == 4 (mod 8) (bank-disjoint chunk tiles), WR <= 252 (staging plan)
  wgrad_band_plan(a.Q, a.W, 252, &a2.bl, &a2.wr, &a2.nbands);
  if (a2.bl < 16 || a2.wr > 256) { dta_set_error("conv_wgrad(bf16): %dx%d patch is too wide for the band plan", a.H, a.W); return 1; }
  size_t stage = (size_t)(CT * 2 + NTT * 2) * a2.wr * RW;
  a2.dbuf = 2 * stage <= 160 * 1024;
  lds = (a2.dbuf ? 2 : 1) * stage;
  if (lds < 48 * 1024) lds = 48 * 1024;   // the final reduction passes up to 6 x 2 tiles (8 KiB each) through LDS
  if (lds > 160 * 1024) { dta_set_error("conv_wgrad(bf16): LDS need %zu B exceeds 160 KiB", lds); return 1; }
  a2.ppi = 1;
  static const bool no_stack = dev_getenv("DTA_NO_WGRAD_STACK") != nullptr;      // development A/B switches, read once
  static const bool no_d2 = dev_getenv("DTA_NO_WGRAD_D2") != nullptr;
  if (CT == 1 && a2.nbands == 1 && a.x_compact && a.y_compact && !no_stack) {
    // stack patches while the window (rows: (p - 1) Q + first centre row + 16 k-step rows + tap reach, = 4 mod 8) stays
    // within the staging plan's 192 rows and the per-thread vector registers (X: 3 XCH / 4, dY: 3 YCH / 4 per 512 threads)
    constexpr int XV = (3 * CT * 2 + 3) / 4, YV = (3 * NTT * 2 + 3) / 4;
    const int HWp = a.H * a.W, q0 = a.W + 3, span = a.Q - 2 * (a.W + 3);
    for (int pp = 2; pp <= 8; ++pp) {
      int wr = q0 + ((pp - 1) * a.Q + span + 15) / 16 * 16 + (a.W + 3);
      while ((wr & 7) != 4) ++wr;
      if (wr > 192 || pp * CT * 2 * HWp * 2 > XV * 512 || pp * NTT * 2 * HWp * 2 > YV * 512) break;
      if ((size_t)2 * (CT * 2 + NTT * 2) * wr * RW > 160 * 1024) break;
      a2.ppi = pp; a2.wr = wr;
    }
  }
  stage = (size_t)(CT * 2 + NTT * 2) * a2.wr * RW;
  a2.dbuf = 2 * stage <= 160 * 1024;
  lds = (a2.dbuf ? 2 : 1) * stage;
  if (lds < 48 * 1024) lds = 48 * 1024;
  a2.cgroups = cgroups; a2.G = G;
  a2.ngroups = a.N / (NTT * 32);
  total = cgroups * a2.ngroups * a.S * G;
  // two windows in flight in registers: the first conv's <2,2> (cycle stamps: loop 119.9 k -> 116.7 k cycles; the
  // second conv's <1,2> measured slower with it, 23.8 -> 24.9 us, and stays on one set)
  if (a2.ppi > 1) variant = WV_STACK;      // (CT == 1 only: the layers with small maps, 64 -> 128 channels)
  else if (a2.wr <= 192 && a2.dbuf && CT == 2 && NTT == 2 && !no_d2) variant = WV_D2;
  else if (a2.wr <= 192) variant = WV_PLAIN;
  else variant = WV_BIGW;
  return 0;
}

// Weight gradient of the first conv + the exchange's side job in one launch.  Returns 0 = launched, 2 = this plan does not
// resolve to a program the combined kernel is instantiated for, or leaves no CU for the side workgroups (the caller launches
// the plain weight gradient and lets the exchange sum its head segment itself), 1 = error.
template <int CT, int NTT, bool BIGW, bool D2>
static int launch_wgrad_xchg_t(const WgradArgs& a2, size_t lds, int total, const XchgArgs& side, hipStream_t st) {
  static DevOnce attr_once;
  if (attr_once.first()) hipFuncSetAttribute((const void*)k_conv_wgrad_bf16_xchg<CT, NTT, BIGW, false, D2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int nmain = 8 * ((total + 7) / 8);
  hipLaunchKernelGGL((k_conv_wgrad_bf16_xchg<CT, NTT, BIGW, false, D2>), dim3(nmain + XCHG_SIDE_WGS), dim3(512), lds, st, a2, side, nmain);
  DTA_CHECK_LAUNCH("k_conv_wgrad_bf16_xchg");
  return 0;
}
int launch_conv_wgrad_bf16_xchg(const WgradArgs& a, int G, const XchgArgs& side, hipStream_t st) {
  const int cpw = wgrad_cpw(a.N), cgroups = (a.Cpad + cpw - 1) / cpw;
  WgradArgs a2;
  size_t lds;
  int variant, total;
  if (a.N == 64 && a.Cpad > 32) {
    if (resolve_wgrad_bf16<2, 2>(a, G, cgroups, a2, lds, variant, total)) return 1;
    if (variant != WV_D2 || 8 * ((total + 7) / 8) + XCHG_SIDE_WGS > 256) return 2;
    return launch_wgrad_xchg_t<2, 2, false, true>(a2, lds, total, side, st);
  }
  if (a.N == 32) {
    if (resolve_wgrad_bf16<4, 1>(a, G, cgroups, a2, lds, variant, total)) return 1;
    if (8 * ((total + 7) / 8) + XCHG_SIDE_WGS > 256) return 2;      // (one 512-thread workgroup per CU: the side ones need CUs of their own)
    if (variant == WV_PLAIN) return launch_wgrad_xchg_t<4, 1, false, false>(a2, lds, total, side, st);
    if (variant == WV_BIGW) return launch_wgrad_xchg_t<4, 1, true, false>(a2, lds, total, side, st);
  }
  return 2;
}

template <>
int launch_conv_wgrad<bf16_t>(const WgradArgs& a, int G, hipStream_t st) {
  int cpw = wgrad_cpw(a.N);
  int cgroups = (a.Cpad + cpw - 1) / cpw;
  switch (a.N) {
    case 32: return launch_wgrad_bf16_t<4, 1>(a, G, cgroups, st);
    case 64:
      // a 32-channel input (conv2) fills only one of the two c-tiles: run one tile pair-wise over four k-slices
      if (a.Cpad <= 32 && cgroups == 1) return launch_wgrad_bf16_t<1, 2>(a, G, cgroups, st);
      return launch_wgrad_bf16_t<2, 2>(a, G, cgroups, st);
    case 128:
      // two 64-column groups (wgrad_ngroups): the plan's slab count already accounts for them
      if (a.ngroups == 2) return launch_wgrad_bf16_t<1, 2>(a, G, (a.Cpad + 31) / 32, st);
      return launch_wgrad_bf16_t<1, 4>(a, G, cgroups, st);
  }
  dta_set_error("conv_wgrad: unsupported width %d", a.N);
  return 1;
}

// Second and third conv in one launch when both resolve to the pair kernel's programs; otherwise one after the other.
int launch_conv_wgrad_pair_bf16(const WgradArgs& conv2, const WgradArgs& conv3, int G, hipStream_t st) {
  const bool shapes = conv2.N == 64 && conv2.Cpad <= 32 && conv3.N == 128 && conv3.ngroups == 2;
  if (shapes) {
    WgradArgs a2, b2;
    size_t la, lb;
    int va, vb, ta, tb;
    if ((conv2.Cpad + wgrad_cpw(conv2.N) - 1) / wgrad_cpw(conv2.N) != 1) return launch_conv_wgrad<bf16_t>(conv3, G, st) || launch_conv_wgrad<bf16_t>(conv2, G, st);
    if (resolve_wgrad_bf16<1, 2>(conv2, G, 1, a2, la, va, ta)) return 1;
    if (resolve_wgrad_bf16<1, 2>(conv3, G, (conv3.Cpad + 31) / 32, b2, lb, vb, tb)) return 1;
    const int na = 8 * ((ta + 7) / 8), nb = 8 * ((tb + 7) / 8);
    static DevOnce attr_once;
    if (attr_once.first()) {
      hipFuncSetAttribute((const void*)k_conv_wgrad_bf16_pair<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipFuncSetAttribute((const void*)k_conv_wgrad_bf16_pair<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (va == WV_PLAIN && vb == WV_STACK && na + nb <= 256) {
      hipLaunchKernelGGL(k_conv_wgrad_bf16_pair<false>, dim3(na + nb), dim3(512), la > lb ? la : lb, st, a2, b2, na);
      DTA_CHECK_LAUNCH("k_conv_wgrad_bf16_pair");
      return 0;
    }
    if (va == WV_BIGW && vb == WV_BIGW && na + nb <= 256) {
      hipLaunchKernelGGL(k_conv_wgrad_bf16_pair<true>, dim3(na + nb), dim3(512), la > lb ? la : lb, st, a2, b2, na);
      DTA_CHECK_LAUNCH("k_conv_wgrad_bf16_pair");
      return 0;
    }
  }
  if (launch_conv_wgrad<bf16_t>(conv3, G, st)) return 1;
  return launch_conv_wgrad<bf16_t>(conv2, G, st);
}
// plan-time test: would the two layers' launches resolve to the pair kernel's programs (plain windows / stacked patches)?
int wgrad_pair_plan_bf16(const WgradArgs& conv2, const WgradArgs& conv3) {
  if (!(conv2.N == 64 && conv2.Cpad <= 32 && conv3.N == 128 && conv3.ngroups == 2)) return 0;
  if ((conv2.Cpad + wgrad_cpw(conv2.N) - 1) / wgrad_cpw(conv2.N) != 1) return 0;
  WgradArgs a2, b2;
  size_t la, lb;
  int va, vb, ta, tb;
  if (resolve_wgrad_bf16<1, 2>(conv2, 1, 1, a2, la, va, ta)) return 0;
  if (resolve_wgrad_bf16<1, 2>(conv3, 1, (conv3.Cpad + 31) / 32, b2, lb, vb, tb)) return 0;
  return (va == WV_PLAIN && vb == WV_STACK) ? 1 : (va == WV_BIGW && vb == WV_BIGW) ? 2 : 0;
}

}  // namespace dta
