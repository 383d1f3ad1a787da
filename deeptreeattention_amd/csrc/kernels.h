// Internal launch interface between the C-ABI orchestration (capi.hip) and the kernel files.
#pragma once
#include "common.h"

namespace dta {

enum { KIND_SPECTRAL = 0, KIND_SPATIAL = 1, KIND_PLAIN = 2 };
constexpr int MAXG = 16;  // groups per launch: Hang2020's two branches, the years of an ensemble, or the levels x years of a
                          // multi-stage step (reference multi_stage.py:41-66: 5 levels x the data's years); every per-group array of
                          // a kernel-argument struct is this long, and every such struct stays under the 4 KB kernarg segment

// ---- conv.hip ----------------------------------------------------------------------------------
struct PackWArgs {
  const float* src[MAXG];
  int G, NC, N, K;   // K = real size of the contraction-channel dim (Cin for fwd, Cout for dgrad)
  int mode, nsplit;
};
struct ConvArgs {
  const void* x_tl; size_t x_gs;      // input tiles, group stride in elements (0 = shared by groups)
  const void* wp;                     // [G][NC][9][N][16]
  const float* bias[MAXG]; int bias_mode, bias_split;
  float* y; size_t y_gs; int y_rs;    // output rows [row][y_rs], group offset y_gs (elements); storage format y_fmt
  int y_fmt;                          // FMT_F32 (default) / FMT_F16 / FMT_BF16: bf16 kernels only
  int mwg;                            // output rows per workgroup the plan chose (0 = conv_mwg(N)); bf16 kernels
  float* stats;                       // [G][nwg][N][2] (mean, M2) or null
  int B, H, W, NC, N, Q, HW, ppw, spp, dbuf;
  int x_compact;                      // bf16: input tiles are halo-free [patch][chunk][pixel][16] (network input only)
  // bf16 first conv fed straight from the caller's tensor: x_nchw[g] = float32 [B][Cx][H][W]; the kernel converts while
  // staging and (x_tl_out != null) writes the halo-free bf16 tiles it built as a by-product for the weight gradient
  const float* x_nchw[MAXG]; int Cx; void* x_tl_out;
  int pixel_order;                    // bf16 developer switches: bit 0 = tile rows in pixel order (conv_row_tables), bit 1 = all waves stage before they multiply
  // BatchNorm statistics without a finalize launch (see FanIn below): the last workgroup to arrive in each of FAN_R
  // logical groups folds that group's (mean, M2) rows into one row of raw sums; the stage kernels add the FAN_R rows up
  int ncg;                            // bf16 kernels: column groups (blockIdx.z) of N / ncg columns each (0 / 1 = one); the
                                      // weight slab, bias, outputs and statistics rows keep their full-width layouts
  // bf16 kernels: row tables of a FULL workgroup of this launch (conv_row_tables below), built once per forward by the prep
  // launch: [tabs_rows] output row relative to the workgroup's first patch (or -1), then [tabs_rows] plq; null = every workgroup
  // builds its own (a few-chunk workgroup spent 4 k of its 21 k cycles there)
  const int* tabs; int tabs_rows;
  // bf16 kernels: reciprocals of the per-thread staging plan's divisors (magic = ceil(2^32 / d); n / d = umulhi(n, magic) for
  // n * d < 2^32), filled by the launcher: the plan's eight to twenty runtime integer divisions were ~1.5 k of a few-chunk
  // workgroup's 19 k cycles
  unsigned m_vpp, m_W, m_HW, m_4HW;
  unsigned* fan_count;                // [gridDim.y][FAN_R] arrival counters (zero on entry, left zero) or null
  double* fan_sums;                   // [gridDim.y][FAN_R][N][3] = sum n m, sum n m^2, sum M2 per column
};
// ---- fan-in of per-workgroup partial rows inside one launch ---------------------------------------------------------
// Workgroups whose linear block id has the same residue mod FAN_R form a LOGICAL group (on this GPU they also share an
// XCD -- block b runs on XCD b % 8 -- which only makes the hand-off faster; nothing depends on the placement).  Every
// workgroup publishes its row with 8-byte agent-scope stores, drains them and bumps the group's counter; the one that
// sees the count complete (the last to arrive) reads the group's rows back with 8-byte agent-scope loads ("8-B agent
// atomics both sides", the cross-XCD-safe hand-off), folds them in a fixed order and leaves ONE row per group for the
// NEXT launch, which then adds FAN_R rows instead of waiting for a finalize launch over hundreds.
constexpr int FAN_R = 8;
struct WgradArgs {
  const void* x_tl; size_t x_gs; int NCx;
  const void* dy_tl; size_t dy_gs; int NCy, ych0;
  float* partial;                     // [G][9][Cpad][S][N]
  int B, H, W, Q, N, Cpad, S, dbuf, cgroups, G;
  int bl, wr, nbands;                 // K-band plan (rows per band, LDS window rows, bands per patch)
  int ppi;                            // bf16, small halo-free maps: patches stacked per LDS window (0 / 1 = one)
  int ngroups;                        // bf16: output-column groups per (channel group, slab): a workgroup covers N / ngroups columns (0 = 1)
  int x_compact;                      // bf16, single band: X tiles are halo-free [patch][chunk][pixel][16]
  int y_compact;                      // likewise the dY tiles
};
// bl = rows per band (multiple of 16), wr = bl + 2*(W+3) rounded up to 4 (mod 8), wr <= wr_max
void wgrad_band_plan(int Q, int W, int wr_max, int* bl, int* wr, int* nbands);
struct WgradReduceArgs {
  const float* partial; float* dst[MAXG];
  int G, S, N, C, Cpad, mode, nsplit;
};
template <typename T> int launch_pack_input(const float* x, void* out, int B, int C, int H, int W, hipStream_t st);
struct PackWGroup { PackWArgs job[6]; void* dst[6]; int n = 0; };
template <typename T> int launch_pack_conv_w(const PackWArgs& a, void* dst, hipStream_t st);
template <typename T> int launch_pack_conv_w_group(const PackWGroup& gr, hipStream_t st);
template <typename T> int launch_conv3x3(const ConvArgs& a, int G, hipStream_t st);
template <typename T> int launch_conv_wgrad(const WgradArgs& a, int G, hipStream_t st);
template <> int launch_conv3x3<bf16_t>(const ConvArgs& a, int G, hipStream_t st);       // conv_bf16.hip
template <> int launch_conv_wgrad<bf16_t>(const WgradArgs& a, int G, hipStream_t st);   // conv_bf16.hip
int launch_wgrad_reduce(const WgradReduceArgs& a, hipStream_t st);
// several layers' split-K reductions in one launch (block ranges -> jobs)
struct WgradReduceGroup {
  WgradReduceArgs job[3]; int start[4]; int n = 0;
  // optional rider: slot_dst[0] = (float)slot_src[0] -- alpha's float64 gradient (complete since the backward's first
  // launch, accumulated order-independently) into its fp32 exchange slot of the gradient buffer: ONE rounding of the
  // finished double instead of fp32 atomics in arrival order, so the data-parallel step is run-to-run reproducible too
  const double* slot_src = nullptr; float* slot_dst = nullptr;
};
int launch_wgrad_reduce_group(WgradReduceGroup& gr, hipStream_t st);
int wgrad_reduce_nblocks(const WgradReduceArgs& a);
// bf16: the weight gradients of the second and third conv in ONE launch when both resolve to the pair kernel's programs
// (conv_bf16.hip); otherwise third then second, each in its own launch
int launch_conv_wgrad_pair_bf16(const WgradArgs& conv2, const WgradArgs& conv3, int G, hipStream_t st);
// the first conv's weight gradient with the peer exchange's head-segment reduce-scatter as side workgroups (conv_bf16.hip,
// xchg_dev.h): 0 launched, 2 not applicable to this plan, 1 error
struct XchgArgs;
int launch_conv_wgrad_bf16_xchg(const WgradArgs& a, int G, const XchgArgs& side, hipStream_t st);
// plan-time test (geometry fields only: H, W, Q, N, Cpad, ngroups, x/y_compact): 0 = no pairing, 1 = plain + stacked
// windows (11x11 networks), 2 = both wide / multi-band windows (24x24 crops)
int wgrad_pair_plan_bf16(const WgradArgs& conv2, const WgradArgs& conv3);
#if defined(__HIPCC__)
// out_g[n][c][tap] (torch layout) = sum_s partial[g][tap][c][s][n]
__device__ __forceinline__ void wgrad_reduce_blocks(const WgradReduceArgs& a, int bx, int nblocks) {
  // item = (group, tap, channel, 4 consecutive n); FOUR lanes share an item and each sums every fourth slab, then the
  // four partial sums meet through two shuffles: this is a pure streaming read of S slabs, bound by the bytes in
  // flight, and four times as many threads keep four times as many loads outstanding
  const int N = a.N, N4 = N / 4;
  const size_t total = (size_t)a.G * 9 * a.C * N4;
  const size_t sstride = (size_t)N;              // the S partial rows of an item are adjacent
  const int part = threadIdx.x & 3;
  const size_t first = (bx * (size_t)blockDim.x + threadIdx.x) >> 2, step = ((size_t)nblocks * blockDim.x) >> 2;
  const size_t iters = (total + step - 1) / step;   // all lanes of a quad iterate together (shuffles below)
  for (size_t it = 0; it < iters; ++it) {
    const size_t i = first + it * step;
    const bool live = i < total;
    const size_t ii = live ? i : 0;
    int n4 = ii % N4;
    size_t r = ii / N4;
    int c = r % a.C; r /= a.C;
    int tap = r % 9;
    int g = r / 9;
    const float* p = a.partial + (((size_t)g * 9 + tap) * a.Cpad + c) * a.S * N + n4 * 4;
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = part;
    for (; s + 12 < a.S; s += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 q_ = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)(s + 4 * u) * sstride));   // slabs: one reader
        float4 v = make_float4(q_[0], q_[1], q_[2], q_[3]);
        acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
      }
    }
    for (; s < a.S; s += 4) {
      float4 v = *reinterpret_cast<const float4*>(p + (size_t)s * sstride);
      acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
    }
    float out[4] = {(acc[0].x + acc[1].x) + (acc[2].x + acc[3].x), (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y),
                    (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z), (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      out[k] += __shfl_xor(out[k], 1);
      out[k] += __shfl_xor(out[k], 2);
    }
    if (live) {
      // lane `part` of the quad writes output n4*4 + part
      const int n = n4 * 4 + part;
      int nn = n;
      float* dst;
      if (a.mode == 1) { dst = n < a.nsplit ? a.dst[0] : a.dst[1]; nn = n < a.nsplit ? n : n - a.nsplit; }
      else dst = a.dst[g];
      const float o = part == 0 ? out[0] : part == 1 ? out[1] : part == 2 ? out[2] : out[3];
      if (dst) dst[((size_t)nn * a.C + c) * 9 + tap] = o;
    }
  }
}

#endif
#if defined(__HIPCC__)
__device__ __forceinline__ int fan_group() { return (int)((blockIdx.x + blockIdx.y * gridDim.x) & (FAN_R - 1)); }
// first block x of row blockIdx.y whose linear id has residue q, and how many there are
__device__ __forceinline__ int fan_first(int q) { return (q - (int)((blockIdx.y * gridDim.x) & (FAN_R - 1))) & (FAN_R - 1); }
__device__ __forceinline__ int fan_members(int q) { const int x0 = fan_first(q); return (int)gridDim.x > x0 ? ((int)gridDim.x - x0 + FAN_R - 1) / FAN_R : 0; }
__device__ __forceinline__ void fan_store2(float* p, float a, float b) {      // one 8-byte agent-scope store
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 fan_load2(const float* p) {                  // one 8-byte agent-scope load
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
}
// Call with the whole workgroup after its row stores were ISSUED: drains them, arrives, and returns true in every thread of
// the group's last arriver (which also re-arms the counter for the next launch).  flag: one int of LDS.
__device__ __forceinline__ bool fan_arrive(unsigned* count, int* flag) {
  __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): this wave's row stores are performed
  __syncthreads();
  if (threadIdx.x == 0) {
    const int q = fan_group();
    unsigned* c = count + blockIdx.y * FAN_R + q;
    const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old + 1u == (unsigned)fan_members(q);
    if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}
// output rows a conv workgroup owns (its weight in the combination of the (mean, M2) partials)
__device__ __forceinline__ int conv_wg_count(int wg, int HW, int MWG, int B) {
  if (HW <= MWG) { int ppw = MWG / HW; return min(ppw, B - wg * ppw) * HW; }
  int spp = (HW + MWG - 1) / MWG;
  return min(MWG, HW - (wg % spp) * MWG);
}
// Tail of a conv kernel's BatchNorm-statistics epilogue when ConvArgs::fan_count is set.  Every workgroup has just
// published its (mean, M2) row with fan_store2; the last arriver of each logical group folds the group's rows into raw
// sums in double -- sum n m, sum n m^2, sum M2, the single-pass form k_bn_finalize uses -- in a fixed order (slices of rows,
// then slices in order), so the result does not depend on which workgroup arrived last.  dred: NTHR * 3 doubles of LDS.
template <int NTHR>
__device__ __forceinline__ void conv_stats_fanin(const ConvArgs& a, int g, int N, int HW, int MWG, double* dred, int* flag) {
  if (!fan_arrive(a.fan_count, flag)) return;
  const int tid = threadIdx.x, q = fan_group(), x0 = fan_first(q), nrows = fan_members(q);
  const int col = tid % N, sl = tid / N, T = NTHR / N;
  const float* st = a.stats + (size_t)g * gridDim.x * N * 2;
  double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll 4
  for (int r = sl; r < nrows; r += T) {
    const int wg = x0 + r * FAN_R;
    const float2 v = fan_load2(st + ((size_t)wg * N + col) * 2);
    const double nb = conv_wg_count(wg, HW, MWG, a.B), m = (double)v.x;
    s0 += nb * m; s1 += nb * m * m; s2 += (double)v.y;
  }
  dred[tid] = s0; dred[NTHR + tid] = s1; dred[2 * NTHR + tid] = s2;
  __syncthreads();
  if (tid < N) {
    double t0 = 0, t1 = 0, t2 = 0;
    for (int j = 0; j < T; ++j) { t0 += dred[j * N + tid]; t1 += dred[NTHR + j * N + tid]; t2 += dred[2 * NTHR + j * N + tid]; }
    double* o = a.fan_sums + (((size_t)blockIdx.y * FAN_R + q) * N + tid) * 3;
    o[0] = t0; o[1] = t1; o[2] = t2;
  }
}
#endif
#if defined(__HIPCC__)
// Row tables of a conv workgroup: which pixel each of its MWG tile rows computes (rowtab: global output row or -1;
// plq: (patch << 16) | haloed-grid row of the window's top-left tap).
//
// The A fragments are ds_read_b128 reads of 16 haloed-grid rows per lane group, RB = 48 bytes apart: a lane group is
// conflict-free exactly when its 16 rows differ mod 16 (48 B = 12 banks, 12 i mod 64 is a bijection of i mod 16 onto the
// sixteen 16-byte slots of the bank row).  Consecutive pixels do NOT have that property -- every image-row end skips two
// halo rows -- and the hardware's lane groups are not contiguous ({0-3,12-15,20-27}, {4-11,16-19,28-31}, same +32): in pixel
// order the A reads cost 2.25 LDS cycles per group instead of 1 (SQ_LDS_BANK_CONFLICT was 43 % of SQ_LDS_IDX_ACTIVE in
// the first conv).  So pixels are dealt to lane groups by residue: the k-th pixel (in pixel order) whose row is = c mod 16
// goes to lane group k, slot c.  Any assignment of pixels to tile rows is valid -- outputs and statistics go through
// rowtab -- and unused slots point at row `slot` (same residue class), so every group reads 16 distinct slots.
// If some residue class has more pixels than there are groups (possible for split maps), pixel order is kept.
template <int MWG, int NTHR>
__device__ __forceinline__ void conv_row_tables(const ConvArgs& a, int* rowtab, int* plq, int* hist, int* flag, int b0, int npatch, int split) {
  constexpr int PASSES = (MWG + NTHR - 1) / NTHR, NWAVE = NTHR / 64, NGROUP = MWG / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HW = a.HW, W2 = a.W + 2, Q = a.Q;
  int orow[PASSES], pq[PASSES], res[PASSES], rank[PASSES];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int lr = tid + p * NTHR;
    int pl, pix;
    bool valid;
    if (a.spp == 1) { pl = lr / HW; pix = lr - pl * HW; valid = lr < MWG && pl < npatch; }
    else { pl = 0; pix = split * MWG + lr; valid = lr < MWG && pix < HW; }
    const int h = valid ? pix / a.W : 0, w = valid ? pix - h * a.W : 0;
    orow[p] = valid ? (b0 + pl) * HW + pix : -1;
    pq[p] = valid ? ((pl << 16) | (h * W2 + w)) : 0;
    res[p] = valid ? (pl * Q + h * W2 + w) & 15 : -1;
    int mine = 0, cnt = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const unsigned long long m = __ballot(res[p] == c);
      if (res[p] == c) mine = __popcll(m & lt);
      if (lane == c) cnt = __popcll(m);
    }
    rank[p] = mine;
    if (lane < 16) hist[(p * NWAVE + wave) * 16 + lane] = cnt;
    // unused slots: no output row, and a window row of the slot's own residue class
    if (lr < MWG) {
      const int li = lr & 31;
      const int slot = li < 4 ? li : li < 12 ? li - 4 : li < 16 ? li - 8 : li < 20 ? li - 8 : li < 28 ? li - 12 : li - 16;
      rowtab[lr] = -1;
      plq[lr] = (Q * a.ppw > 17 + 2 * W2) ? slot : 0;
    }
  }
  if (tid == 0) *flag = a.pixel_order & 1;
  __syncthreads();
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    if (res[p] >= 0) {
      for (int j = 0; j < p * NWAVE + wave; ++j) rank[p] += hist[j * 16 + res[p]];
      if (rank[p] >= NGROUP) *flag = 1;
    }
  }
  __syncthreads();
  const int over = *flag;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int lr = tid + p * NTHR;
    if (over) {
      if (lr < MWG) { rowtab[lr] = orow[p]; plq[lr] = pq[p]; }
    } else if (res[p] >= 0) {
      const int k = rank[p], c = res[p];
      const int li = (k & 1) ? (c < 8 ? c + 4 : c < 12 ? c + 8 : c + 16) : (c < 4 ? c : c < 8 ? c + 8 : c + 12);
      const int dst = (k >> 1) * 32 + li;
      rowtab[dst] = orow[p];
      plq[dst] = pq[p];
    }
  }
}
#endif
void conv_geometry(int HW, int MWG, int B, int* ppw, int* spp, int* nwg);
int conv_bf16_rows(const ConvArgs& a, int G);      // conv_bf16.hip: rows per workgroup its launcher will choose
// row tables of up to six conv launches of a step, built by extra blocks of the prep launch (one block per job)
struct ConvTabJob { int HW, W, Q, rows, ppw, order; int* dst; };
struct ConvTabGroup { ConvTabJob job[6]; int n = 0; };
int conv_mwg(int N);
// bf16 kernels: output rows per workgroup for an HW-pixel map -- 576 (six waves x three 32-row tiles) when that wastes
// fewer rows than 512 (a 24x24 map is exactly one 576-row workgroup instead of 512 + 64 rows of two)
int conv_mwg_bf16(int N, int HW);
int wgrad_cpw(int N);
int wgrad_ngroups(int N, int bf16);     // column groups of the bf16 weight-gradient kernel (the slab count divides by it)

struct ColsumArgs;
struct BnBwdFanArgs;
// ---- stage.hip (BatchNorm + ReLU + pool + attention, forward and backward) ------------------------
struct BnFinalizeArgs {
  const double* fsum = nullptr;                        // FAN_R rows of raw sums per conv launch row (instead of `stats`)
  const float* stats; int nwg, N, HW, MWG, B;          // partials of one conv launch (per group)
  const float* gamma[MAXG]; const float* beta[MAXG];   // per group (bias_mode 1: concatenated columns)
  float* rmean[MAXG]; float* rvar[MAXG]; long long* nbt[MAXG];
  int cat_mode, nsplit;                                // 1: G==1 launch whose columns are [branch0|branch1]
  float* coef;                                         // [G][N][4] = scale, shift, mean, rstd
  int training; float momentum, eps;
  const float* gate = nullptr;                         // device, per group: <= 0 -> this group's running statistics / counter stay untouched
};
int launch_bn_finalize(const BnFinalizeArgs& a, int G, hipStream_t st);
// kernel-side form of the above (group offsets resolved)
struct BnFinK {
  // fsum != null: the conv launch already folded its partials into FAN_R rows of raw sums (ConvArgs::fan_sums):
  // fsum[(q * fsum_ld + c) * 3 + k] for group g at fsum + g * fsum_goff
  const double* fsum; size_t fsum_goff; int fsum_ld;
  const float* stats; size_t stats_goff; int stats_ld;
  int nwg, C, HW, MWG, B;
  const float* gamma[MAXG]; const float* beta[MAXG];
  float* rmean[MAXG]; float* rvar[MAXG]; long long* nbt[MAXG];
  float* coef; int training; float momentum, eps;
  const float* gate;
};
BnFinK bn_finalize_kargs(const BnFinalizeArgs& b);
// the stage kernels can combine the conv partials themselves (every workgroup, redundantly and in the same order, so
// all of them hold bit-identical coefficients) instead of waiting for a k_bn_finalize launch; worth it up to this many
// conv workgroups (each stage workgroup reads nwg x C x 8 bytes of partials from L2)
constexpr int BN_INKERNEL_MAX_NWG = 512;

struct AttParams {          // forward-side attention parameters of one branch/stage (device pointers)
  // spectral: a1t/a2t = dense transposed centre taps [C][C] (in-major), c1/c2 biases
  // spatial : wc [C], bc, k1 [k*k], b1, k2 [k*k], b2
  const float* p[6];
};
struct StageArgs {
  int kind[MAXG];                      // per group
  const float* y; size_t y_gs; int y_rs;        // conv output (or raw activations when !apply_bn); strides in elements
  int y_fmt;                           // its storage format (FMT_*): the pointer is only a base address
  const float* coef; int coef_gs;      // [..][C][4]
  int apply_bn, relu, pool;            // pool: 2x2 floor max-pool after ReLU
  int B, C, Hc, Wc;                    // conv-resolution dims
  AttParams att[MAXG];
  int att_k[MAXG], att_pool[MAXG];           // spatial stencil size / class-pool size
  int vslot;                           // LDS vector slot (floats); filled by the launcher
  void* a_tl; size_t a_gs; int a_nc, a_ch0;      // gated map as tiles for the next conv (or null)
  int a_compact;                       // bf16 + lean kernels: halo-free tiles [patch][chunk][pixel][16]
  float* a_nchw; size_t a_nchw_gs;     // gated map as fp32 NCHW (standalone modules) or null
  float* feat; size_t feat_gs; int F[MAXG];      // [B][F]
  // attention intermediates of every patch ([G][B][attsave_ld] floats, >= 3 * vslot): written by the forward,
  // read back by the backward instead of recomputing the attention; null = backward recomputes
  float* attsave; int attsave_ld;
  // bn_inkernel != 0: `coef` has not been computed yet -- every workgroup derives its group's scale/shift from `bnfin`
  // (conv partials in training mode, running statistics otherwise) in its prologue; workgroup 0 of each group also
  // writes them to `coef` for the backward and updates the running statistics
  int bn_inkernel; BnFinK bnfin;
  // lead > 0 (lean kernels, training): no finalize launch ran -- the first `lead` = G * ceil(C / 8) workgroups of row 0 compute
  // the coefficients from `bnfin` (bn_lead_block) and the others wait for lead_flag[0] >= lead_need (cleared by k_forward_prep)
  int lead; unsigned* lead_flag; unsigned lead_need;
  int lean;                            // allow the lean register-resident kernels for the 11x11 network stages
};
template <typename T> int launch_stage_fwd(const StageArgs& a, int G, hipStream_t st);

struct StageBwdArgs {
  StageArgs f;                         // forward description (recomputed per patch)
  const float* da; size_t da_gs;       // grad wrt gated map, fp32 [B][HWz][C] (or null)
  const float* da_nchw; size_t da_nchw_gs;       // same as NCHW (standalone modules) or null
  const float* dfeat; size_t dfeat_gs; // [B][F] or null
  float* dv; size_t dv_gs;             // out: grad wrt BN output (pre-ReLU), fp32 [B][HWc][C]
  float* bnpart; size_t bnpart_gs;     // out: [B][C][2] per-patch (sum dv, sum dv*xhat)
  float* vec; size_t vec_gs; int vec_ld;         // out: per-patch attention-gradient vectors [B][vec_ld]
  // pooled stages: dv is 3/4 zeros (only each 2x2 window's maximum gets a gradient).  dv_compact: write per patch
  // [HWz][C] gradient values followed by [HWz][C] window positions (1 byte) instead of the dense [HWc][C] map;
  // k_bn_bwd_apply_lds expands it on the fly (same per-patch stride)
  int dv_compact;
  // storage formats (FMT_F32 / FMT_BF16) of the incoming gradient map `da` and of `dv`; 16-bit only with the lean kernels
  // (stage_bwd_is_lean): per-patch strides stay the same element counts, the compact position bytes follow the values
  int da_fmt, dv_fmt;
  // lean kernels, no finalize launch: every workgroup folds the BatchNorm partial sums of ITS patches into one row,
  // the last arriver of each logical group (FanIn) folds the group's rows: bn_fan_rows [G][grid x][C][2] floats (hand-off
  // scratch), bn_fan_sums [G][FAN_R][C][2] doubles (sum dv, sum dv xhat) for the apply launch, bn_fan_count [G][FAN_R]
  float* bn_fan_rows; double* bn_fan_sums; unsigned* bn_fan_count;
};
int launch_stage_bwd(const StageBwdArgs& a, int G, hipStream_t st);
bool stage_bwd_is_lean(const StageBwdArgs& a, int G);
bool stage_fwd_is_lean(const StageArgs& a);
bool stage_fwd_will_be_lean(const StageArgs& a, int G);      // (after the vslot choice launch_stage_fwd makes)

struct BnBwdFinalizeArgs {
  const float* bnpart; size_t bnpart_gs; int B, C, HW;
  const float* coef; int coef_gs;
  const float* gamma[MAXG];
  float* dgamma[MAXG]; float* dbeta[MAXG]; float* dconvbias[MAXG];
  int cat_mode, nsplit;                // G==1 launch with concatenated branch columns
  float* bcoef; int bcoef_gs;          // out [..][C][4] = A (gamma*rstd), Bc (dbeta/n), Cc (dgamma/n), 0
  int training;
};
int launch_bn_bwd_finalize(const BnBwdFinalizeArgs& a, int G, hipStream_t st);

struct BnBwdApplyArgs {
  const float* dv; size_t dv_gs; const float* y; size_t y_gs; int y_rs; int y_fmt; int dv_fmt;
  const float* coef; int coef_gs; const float* bcoef; int bcoef_gs;
  int B, C, H, W;
  void* dy_tl; size_t dy_gs; int dy_nc, dy_ch0;
  int dy_compact;                      // halo-free output tiles [patch][chunk][pixel][16] (LDS-image kernel, bf16)
  int dv_compact, Hz, Wz;              // see StageBwdArgs::dv_compact
  int cslice;                          // channels per workgroup (filled by the launcher)
};
// DTA_FANIN experiment: no finalize launch ran -- every apply workgroup derives the coefficients of its channels from the
// FAN_R rows of batch sums the stage-backward launch left (StageBwdArgs::bn_fan_sums), and workgroup (0, g, *) also writes
// d(gamma), d(beta), d(conv bias)
struct BnBwdFanArgs {
  const double* fan; const float* gamma[MAXG]; float* dgamma[MAXG]; float* dbeta[MAXG]; float* dconvbias[MAXG]; int training;
};
bool bn_bwd_apply_uses_lds(int C, int H, int W, size_t elem_bytes);
// cs / ncs: up to two batch column-sum jobs riding as extra workgroups of the launch (LDS-image kernel)
template <typename T> int launch_bn_bwd_apply(const BnBwdApplyArgs& a, int G, hipStream_t st, const BnBwdFanArgs* fan = nullptr,
                                              const ColsumArgs* cs = nullptr, int ncs = 0);

// ---- heads.hip -----------------------------------------------------------------------------------
// C[m][n] (+)= sum_k A(m,k) * B(k,n) + bias[n]; arbitrary element strides; fp32 MFMA 32x32x2.
struct GemmArgs {
  const float* A; int sa_m, sa_k;         // (element strides: 32-bit -- 22 descriptors + the slab-reduction jobs share one 4 KB kernarg segment)
  const float* Bm; int sb_k, sb_n;
  float* C; int sc_m, sc_n;
  const float* bias;
  float* rowsum_out;                  // optional: rowsum_out[m] += sum_k A(m,k) (atomic; must be pre-zeroed)
  int M, N, K, ksplit, accumulate;    // ksplit > 1: atomic accumulation into a pre-zeroed C
  // Hang2020 blend folded into the GEMM: results (and row sums) are multiplied by sigmoid(alpha) (mode 1) or
  // 1 - sigmoid(alpha) (mode 2) -- d(joint)/d(branch score) -- so the branch gradients are never materialised
  const double* sig_alpha; int sig_mode;
  // year ensembles with the missing-year decision on the device: gate[0] <= 0 -> every output of this GEMM (and its row
  // sums) is an exact zero, whatever the operands hold (selected, not multiplied: a skipped year's operands may be NaN)
  const float* gate;
  // epilogue options (plain-store GEMMs only: ksplit == 1, no accumulate): relu != 0 -> C = max(C, 0) after the bias;
  // mask != null -> C(m, n) is kept where mask[m * mask_m + n] > 0 and zeroed elsewhere (a ReLU's backward, the mask being
  // the ReLU's output)
  int relu; const float* mask; int mask_m;
};
// (22: the 21 deferred parameter-gradient GEMMs of a three-year ensemble ride in the step's last launch instead of forcing a
//  flush in the middle of the backward; 22 x 160 B + the slab-reduction jobs stay inside the 4 KB kernel-argument segment)
constexpr int GEMM_GROUP_MAX = 22;
struct GemmGroup {
  GemmArgs g[GEMM_GROUP_MAX];
  int start[GEMM_GROUP_MAX + 1];
  int n = 0;
  bool add(const GemmArgs& a) { if (n >= GEMM_GROUP_MAX) return false; g[n++] = a; return true; }
};
int launch_gemm(const GemmArgs& a, hipStream_t st);
int launch_gemm_group(GemmGroup& gg, hipStream_t st);
// the parameter-gradient GEMMs nothing waits for and the split-K reductions of the conv weight gradients as ONE launch
// (both are block-range -> job kernels with 256-thread blocks; a dependent launch costs ~11 us on this GPU)
// wide: the GEMM workgroups are 16 waves (the bf16 step: -1.5 us); false: four waves (the fp32 step measured 0.3 % slower wide)
int launch_gemm_group_with_reduce(GemmGroup& gg, WgradReduceGroup& gr, hipStream_t st, bool wide = false);
int gemm_auto_ksplit(int M, int N, int K);
struct ColsumArgs {
  const float* A; int rows, cols; long lda;
  int nseg; int off[8], len[8]; float* dst[8]; long dst_stride[8];
};
#if defined(__HIPCC__)
// Batch reductions over per-patch partials are latency problems (0.5-2 MB read by a handful of blocks): a 1024-thread
// block owns 8 columns x 128 row slices, so a thread has at most rows/128 independent loads, all in flight at once,
// then three shuffles fold the 8 slices of a wave and 16 wave partials meet in LDS.
// colsum8<NV>: column j (< ncols <= 8) of A (row pitch lda floats, NV consecutive floats per item); on return threads
// t < 8 * NV hold the sums in double (item t / NV, component t % NV).  sc: 16 x 16 floats.
template <int NV, int NTHR = 1024>
__device__ __forceinline__ double colsum8(const float* A, size_t lda, int rows, int ncols, float (*sc)[16]) {
  constexpr int NSL = NTHR / 8, NWV = NTHR / 64;       // row slices, waves
  const int t = threadIdx.x, cl = t & 7, sl = t >> 3, lane = t & 63, wave = t >> 6;
  float acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = 0.f;
  if (cl < ncols) {
#pragma unroll 8
    for (int r = sl; r < rows; r += NSL) {
      const float* p = A + (size_t)r * lda + cl * NV;
      if (NV == 2) { const float2 v = *reinterpret_cast<const float2*>(p); acc[0] += v.x; acc[NV - 1] += v.y; }
      else acc[0] += p[0];
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float v = acc[k];
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    if (lane < 8) sc[wave][lane * NV + k] = v;
  }
  __syncthreads();
  double out = 0;
  if (t < 8 * NV) {
#pragma unroll
    for (int w = 0; w < NWV; ++w) out += (double)sc[w][t];
  }
  __syncthreads();
  return out;
}
template <int NTHR = 1024>
__device__ __forceinline__ void colsum_scatter_block(const ColsumArgs& a, int bx, float (*sc)[16]) {
  const int j0 = bx * 8, t = threadIdx.x;
  const double v = colsum8<1, NTHR>(a.A + j0, (size_t)a.lda, a.rows, min(8, a.cols - j0), sc);
  const int j = j0 + t;
  if (t < 8 && j < a.cols)
    for (int s = 0; s < a.nseg; ++s)
      if (j >= a.off[s] && j < a.off[s] + a.len[s]) {
        if (a.dst[s]) a.dst[s][(size_t)(j - a.off[s]) * a.dst_stride[s]] = (float)v;
        break;
      }
}
inline int colsum_nblocks(const ColsumArgs& a) { return (a.cols + 7) / 8; }
#endif
int launch_colsum_scatter(const ColsumArgs& a, hipStream_t st);
// BatchNorm-backward finalize and up to two batch column-sum jobs (spatial-attention parameter gradients) in one
// launch: both are [batch] reductions over per-patch partials that the same stage-backward kernel produced.
int launch_bn_bwd_finalize_colsum(const BnBwdFinalizeArgs& a, int G, const ColsumArgs* cs, int ncs, hipStream_t st);
// up to two batch column-sum jobs riding as extra workgroups of another launch (the BatchNorm-backward apply launch
// when no finalize launch exists)
struct ColsumPair { ColsumArgs cs[2]; int nblk[2]; };
int launch_pack_spectral_att(const float* w1, const float* w2, int C, int K, float* packed, hipStream_t st);
struct SpecPackGroup { const float* w1[3 * MAXG]; const float* w2[3 * MAXG]; float* packed[3 * MAXG]; int C[3 * MAXG], K[3 * MAXG]; int n = 0; };
int launch_pack_spectral_att_group(const SpecPackGroup& gr, hipStream_t st);
#if defined(__HIPCC__)
// packed = [a1t | a2t | a1 | a2], each [C][C]; *t is input-major (a_t[i][o] = W[o][i][K/2])
__device__ __forceinline__ void pack_spectral_att_job(const SpecPackGroup& gr, int j, size_t i0, size_t stride) {
  const int C = gr.C[j], K = gr.K[j];
  const float* w1 = gr.w1[j]; const float* w2 = gr.w2[j];
  float* packed = gr.packed[j];
  for (size_t i = i0; i < (size_t)C * C; i += stride) {
    int o = (int)(i / C), in = (int)(i - (size_t)o * C);
    float v1 = w1[i * K + K / 2], v2 = w2[i * K + K / 2];
    packed[in * C + o] = v1;
    packed[C * C + in * C + o] = v2;
    packed[2 * C * C + i] = v1;
    packed[3 * C * C + i] = v2;
  }
}
#endif
// dst[c * ld + r] = src[r * cols + c] (r < rows; columns rows..ld-1 of every dst row are zero-filled): the last heads'
// classifier weights [classes][F] as [F][classes padded to 4] for the fused forward tail (lanes read consecutive classes)
struct TransposeGroup { const float* src[4]; float* dst[4]; int rows[4], cols[4], ld[4]; int n = 0; };
#if defined(__HIPCC__)
__device__ __forceinline__ void transpose_job(const TransposeGroup& tg, int j, size_t i0, size_t stride) {
  const int rows = tg.rows[j], cols = tg.cols[j], ld = tg.ld[j];
  const float* src = tg.src[j]; float* dst = tg.dst[j];
  for (size_t i = i0; i < (size_t)cols * ld; i += stride) {
    const int c = (int)(i / ld), r = (int)(i - (size_t)c * ld);
    dst[i] = r < rows ? src[(size_t)r * cols + c] : 0.f;
  }
}
#endif
// one launch for everything the forward needs before its first conv (conv.hip)
struct PrepArgs {
  const float* x[MAXG]; int nx; size_t x_tl_gs;   // nx inputs (one per group with its own input), tile group stride in bytes
  void* x_tl; int B, C, H, W, NC, CG, ncg, x_compact;
  PackWGroup packs; SpecPackGroup spacks; TransposeGroup trans; ConvTabGroup tabs;
  float* zero; size_t zero_n4;         // float4 count to clear, or zero == null
};
template <typename T> int launch_forward_prep(PrepArgs a, hipStream_t st);

struct BlendArgs {
  const float* spec; const float* spat; const double* alpha; float* joint; int B, classes;
};
int launch_blend(const BlendArgs& a, hipStream_t st);
// gate (device, per source; may be null = all): sources with gate <= 0 are left out of the mean; kept (device, may be null)
// receives {number of sources kept, 1 / that number}; nothing kept -> NaN scores, as an empty mean is
struct MeanArgs { const float* src[MAXG]; int n; float* dst; size_t count; const float* gate = nullptr; float* kept = nullptr; };
int launch_year_flags(const float* const* x, int years, size_t n, float* flags, float* clear_next, hipStream_t st);
int launch_mean_scores(const MeanArgs& a, hipStream_t st);
struct BlendBwdArgs {
  const float* spec; const float* spat; const double* alpha; const float* djoint;
  double* dalpha; int B, classes;
};
// the blend's backward lives inside the head GEMMs (GemmArgs::sig_mode); its d(alpha) reduction rides as extra blocks
// of a grouped GEMM launch
int launch_gemm_group_with_blend_fin(GemmGroup& gg, const BlendBwdArgs& fin, hipStream_t st);
struct CeArgs {
  const float* logits; const long long* labels; const float* weight;  // weight may be null (= ones)
  float* dlogits; float* loss; float* rowtmp; int B, classes;
};
int launch_weighted_ce(const CeArgs& a, hipStream_t st);
// blend (optional) + weighted CE + loss in ONE launch: logits = spat ? w spec + (1 - w) spat : spec, w = sigmoid(alpha);
// rowtmp: B + 2 floats, word B + 1 is a block counter that must be zero on entry and is left zero
struct BlendCeArgs {
  const float* spec; const float* spat; const double* alpha; float* joint;   // joint may be null (or == spec when no blend)
  const long long* labels; const float* weight; float* dlogits; float* loss; float* rowtmp; int B, classes;
  float gscale = 1.f;      // factor on dlogits only (year ensemble: d(mean over kept years) / d(year score))
  const float* gscale_dev = nullptr;   // non-null: the factor is read from the device (decided there: kept years)
  int relu_mask = 0;                   // the scores are a ReLU's output: dlogits is the gradient w.r.t. the ReLU's INPUT
  // year ensemble (nsrc > 0): the scores are the MEAN over the kept sources (reference year.py:33), formed on the fly exactly as
  // k_mean_scores forms it (sum in source order, selected by src_gate > 0 -- NULL = all --, times 1 / kept); `joint` receives
  // the mean, kept_out (may be NULL) {kept, 1 / kept}, and dlogits = d(loss)/d(ONE source's scores) = d(loss)/d(mean) / kept
  // (gscale / gscale_dev are ignored); nothing kept: NaN scores and loss, exact-zero dlogits
  const float* src[MAXG] = {}; int nsrc = 0; const float* src_gate = nullptr; float* kept_out = nullptr;
};
int launch_blend_ce(const BlendCeArgs& a, hipStream_t st);
// several independent losses in ONE launch (blockIdx.y = loss): the levels of a multi-stage step, each with its own class
// count, labels, class weights, sources and outputs (reference multi_stage.py:277-288: one F.cross_entropy per level);
// every entry has the same batch size (the grid's x extent is the loss kernel's last-arriver count)
constexpr int BLEND_CE_MULTI_MAX = 8;
struct BlendCeMulti { BlendCeArgs a[BLEND_CE_MULTI_MAX]; int n = 0; };
int launch_blend_ce_multi(const BlendCeMulti& m, hipStream_t st);
// ---- stage.hip: fused forward tail of Hang2020 on 11x11 patches ------------------------------------------------------
// The third stage of BOTH branches (BatchNorm -> ReLU -> 2x2 pool -> spectral / spatial attention -> features), the two
// last-head classifiers, the sigmoid(alpha) blend and -- when ce.labels is set -- the class-weighted cross-entropy with
// its gradient and the loss, in ONE launch: a workgroup owns four patches x two branches, everything between the third
// conv's output and the scores stays in registers / LDS (reference Hang2020.py:24-31, :105-124, :149-168, :55-66, :256-261;
// src/main.py:78).  Replaces k_stage_fwd_lean<128,5,5> + k_gemm_group + k_blend (or k_blend_ce).
struct TailArgs {
  StageArgs st;                        // stage 3 (groups: 0 spectral, 1 spatial): y, coef, att, feat, attsave
  const float* wt; int ldw;            // [128 + 512][ldw] transposed last-head weights (TransposeGroup), ldw = classes padded to 4
  const float* bias[2];                // the two heads' biases
  float* scores[2];                    // out: branch scores [B][classes] (the backward's d(alpha) reads them)
  const double* alpha; int classes;
  BlendCeArgs ce;                      // ce.labels == null: blend only (ce.joint = the blended scores, required)
};
bool tail_fwd_supported(const StageArgs& st3, int G, int classes);
int launch_tail_fwd(const TailArgs& a, hipStream_t st);
struct AdamArgs {
  float* p; const float* g; float* m; float* v; size_t n;
  double* alpha_p; const double* alpha_g; double* alpha_m; double* alpha_v;
  const float* alpha_g32;              // non-null: alpha's gradient is read from this fp32 exchange slot instead
  float lr, beta1, beta2, eps, bc1, bc2; float grad_scale;
  float* gz; double* alpha_gz;         // non-null: gradients are cleared after use (step + zero_grad in one pass)
  // device-side gating (year ensembles under data parallelism): when `active` is non-null the step is applied only if
  // active[0] > 0 (otherwise the moments do not decay and the parameters stay, exactly as torch's Adam passes over a
  // parameter whose grad is None; the gradient buffer is still cleared), and the bias corrections come from the
  // DEVICE step counter (dev_step[0] = steps taken so far; this one is step dev_step[0] + 1) instead of bc1 / bc2
  const float* active; const int* dev_step; int* dev_step_out = nullptr;
  float* g_inactive = nullptr;         // gated: a group that is not stepped has no gradient -- its buffer is cleared whatever gz says
};
int launch_adam(const AdamArgs& a, hipStream_t st);
constexpr int ADAM_MAX_SEG = 16;
struct AdamMulti { AdamArgs seg[ADAM_MAX_SEG]; int n; };
int launch_adam_multi(const AdamMulti& m, hipStream_t st);      // blockIdx.y = segment
int launch_softmax_top2(const float* logits, int B, int classes, float* probs, long long* top_idx, float* top_score,
                        hipStream_t st);
// the same for up to 8 levels in ONE launch (blockIdx.y = level), each row's scores formed on the fly as the mean over the
// level's kept sources exactly as k_mean_scores forms it (sum in source order, selected by gate > 0, times 1 / kept):
// MultiStage.predict_step's per-level `softmax(mean over years)` (reference multi_stage.py:306-318, year.py:33)
struct SoftmaxLevel { const float* src[MAXG]; int nsrc; const float* gate; float* mean_out; int classes;
                      float* probs; long long* top_idx; float* top_score; };
struct SoftmaxMulti { SoftmaxLevel lv[BLEND_CE_MULTI_MAX]; int n, B; };
int launch_softmax_top2_multi(const SoftmaxMulti& m, hipStream_t st);

}  // namespace dta
