// Small dense pieces of the Hang2020 hot path on gfx950: classifier heads (nn.Linear forward/backward,
// reference src/models/Hang2020.py:55-66), batch reductions of attention/bias gradients, the
// sigmoid(alpha) blend (:260-261), class-weighted cross-entropy (src/main.py:78) and Adam (src/main.py:136).
#include <stdlib.h>
#include "kernels.h"
#include "ce_dev.h"

namespace dta {

// ------------------------------------------------------------------------------------------------
// Strided fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products/accumulation).
// 64x64 output tile per workgroup, wave w -> 32x32 quadrant, K chunk 32 staged in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int GK = 32;
constexpr int GP = 36;   // LDS row pitch (floats): 16-byte aligned rows, conflict-free b128 operand reads

// Both operands live in LDS as [row][kperm] (row = m for A, n for B) with kperm(k) = (k & 1) * 16 + (k >> 1): the
// 16 k values one lane feeds to the 16 MFMAs of a chunk (k = 2j + lane/32) are contiguous -> four ds_read_b128.
enum { LD_SCALAR = 0, LD_VEC_K = 1, LD_VEC_ROW = 2, LD_RUNTIME = 3,
       // wave-tile form only: unit stride along k / along the rows but NOT 16-byte loadable (a score matrix whose class count
       // is not a multiple of four): element-wise loads into the same LDS images as LD_VEC_K / LD_VEC_ROW
       LD_SC_K = 4, LD_SC_ROW = 5 };
// wave-tile load mode of one operand, or LD_SCALAR if neither stride is 1
__host__ __device__ __forceinline__ int gemm_load_mode(const float* p, int rows, int K, long s_row, long s_k);
__host__ __device__ __forceinline__ int gemm_wt_mode(const float* p, int rows, int K, long s_row, long s_k) {
  const int m = gemm_load_mode(p, rows, K, s_row, s_k);
  if (m != LD_SCALAR) return m;
  return s_k == 1 ? LD_SC_K : (s_row == 1 ? LD_SC_ROW : LD_SCALAR);
}

__host__ __device__ __forceinline__ int gemm_load_mode(const float* p, int rows, int K, long s_row, long s_k) {
  const bool al = (((size_t)p) & 15) == 0;
  if (s_k == 1 && al && (s_row & 3) == 0 && (K & 3) == 0) return LD_VEC_K;
  if (s_row == 1 && al && (s_k & 3) == 0 && (rows & 3) == 0) return LD_VEC_ROW;
  return LD_SCALAR;
}

// one 64 x GK operand tile -> 8 floats per thread
__device__ __forceinline__ void gemm_load(float (&r)[8], const float* p, int mode, int rows, int r0, long s_row, long s_k,
                                          int k0, int kend, int t) {
  if (mode == LD_VEC_K) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = t + u * 256, row = i >> 3, k = (i & 7) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r0 + row < rows && k0 + k < kend) v = *(const f32x4*)(p + (size_t)(r0 + row) * s_row + (k0 + k));
      r[4 * u] = v[0]; r[4 * u + 1] = v[1]; r[4 * u + 2] = v[2]; r[4 * u + 3] = v[3];
    }
  } else if (mode == LD_VEC_ROW) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = t + u * 256, k = i >> 4, row = (i & 15) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r0 + row < rows && k0 + k < kend) v = *(const f32x4*)(p + (size_t)(k0 + k) * s_k + (r0 + row));
      r[4 * u] = v[0]; r[4 * u + 1] = v[1]; r[4 * u + 2] = v[2]; r[4 * u + 3] = v[3];
    }
  } else {
    const bool rowfast = s_row == 1;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = t + u * 256;
      const int row = rowfast ? (i & 63) : (i >> 5), k = rowfast ? (i >> 6) : (i & 31);
      r[u] = (r0 + row < rows && k0 + k < kend) ? p[(size_t)(r0 + row) * s_row + (size_t)(k0 + k) * s_k] : 0.f;
    }
  }
}

__device__ __forceinline__ int kperm(int k) { return (k & 1) * 16 + (k >> 1); }

// LDS image of one operand tile.  k-fast operands: [row][GP] with permuted k (above).  row-fast operands (the
// transposed gradients GEMMs): [k][64] exactly as the float4 global loads deliver it; the MFMA feed is then 16
// conflict-free ds_read_b32 (32 consecutive rows per half-wave).
__device__ __forceinline__ void gemm_store(float* S, const float (&r)[8], int mode, bool rowfast, int t) {
  if (mode == LD_VEC_K) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = t + u * 256, row = i >> 3, k = (i & 7) * 4;   // k, k+2 -> even half; k+1, k+3 -> odd half
      *(f32x2*)&S[row * GP + (k >> 1)] = f32x2{r[4 * u], r[4 * u + 2]};
      *(f32x2*)&S[row * GP + 16 + (k >> 1)] = f32x2{r[4 * u + 1], r[4 * u + 3]};
    }
  } else if (mode == LD_VEC_ROW) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = t + u * 256, k = i >> 4, row = (i & 15) * 4;
      *(f32x4*)&S[k * 64 + row] = f32x4{r[4 * u], r[4 * u + 1], r[4 * u + 2], r[4 * u + 3]};
    }
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = t + u * 256;
      if (rowfast) S[(i >> 6) * 64 + (i & 63)] = r[u];
      else S[(i >> 5) * GP + kperm(i & 31)] = r[u];
    }
  }
}

// the 16 k-values lane (row, h = lane / 32) feeds to the chunk's MFMAs: k = 2j + h
__device__ __forceinline__ void gemm_fetch(float (&v)[16], const float* S, bool rowfast, int row, int h) {
  if (rowfast) {
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = S[(2 * j + h) * 64 + row];
  } else {
    const f32x4* p = (const f32x4*)&S[row * GP + h * 16];
#pragma unroll
    for (int j = 0; j < 4; ++j) { f32x4 q = p[j]; v[4 * j] = q[0]; v[4 * j + 1] = q[1]; v[4 * j + 2] = q[2]; v[4 * j + 3] = q[3]; }
  }
}

// AM / BM: compile-time load mode of each operand (straight-line code: the loads of both operands stay in flight
// together and under the MFMAs), or LD_RUNTIME for the generic any-stride path.
template <int AM, int BM>
__device__ __forceinline__ void gemm_block(const GemmArgs& a, int bx, int by, int bz, float* As, float* Bs) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m0 = bx * 64, n0 = by * 64;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int kper = ((a.K + a.ksplit - 1) / a.ksplit + GK - 1) / GK * GK;
  const int kbeg = bz * kper, kend = min(a.K, kbeg + kper);
  if (bz > 0 && kbeg >= kend) return;   // K rounded up to whole chunks can leave trailing slices empty
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float rsum = 0.f;   // row sums of A over this block's K range (bias gradients), first column-tile only
  const int amode = AM == LD_RUNTIME ? gemm_load_mode(a.A, a.M, a.K, a.sa_m, a.sa_k) : AM;
  const int bmode = BM == LD_RUNTIME ? gemm_load_mode(a.Bm, a.N, a.K, a.sb_n, a.sb_k) : BM;
  const bool arow = AM == LD_RUNTIME ? a.sa_m == 1 : AM == LD_VEC_ROW, brow = BM == LD_RUNTIME ? a.sb_n == 1 : BM == LD_VEC_ROW;
  float ra[8], rb[8];
  if (kbeg < kend) {
    gemm_load(ra, a.A, amode, a.M, m0, a.sa_m, a.sa_k, kbeg, kend, t);
    gemm_load(rb, a.Bm, bmode, a.N, n0, a.sb_n, a.sb_k, kbeg, kend, t);
  }
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    __syncthreads();
    gemm_store(As, ra, amode, arow, t);
    gemm_store(Bs, rb, bmode, brow, t);
    __syncthreads();
    if (k0 + GK < kend) {   // next chunk's global loads fly under this chunk's MFMAs
      gemm_load(ra, a.A, amode, a.M, m0, a.sa_m, a.sa_k, k0 + GK, kend, t);
      gemm_load(rb, a.Bm, bmode, a.N, n0, a.sb_n, a.sb_k, k0 + GK, kend, t);
    }
    if (a.rowsum_out && by == 0 && t < 64) {
#pragma unroll
      for (int k = 0; k < GK; ++k) rsum += arow ? As[k * 64 + t] : As[t * GP + k];
    }
    float av[16], bv[16];
    gemm_fetch(av, As, arow, wm + (lane & 31), lane >> 5);
    gemm_fetch(bv, Bs, brow, wn + (lane & 31), lane >> 5);
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
  }
  float osc = 1.f;
  if (a.sig_mode) {
    const double wd = 1.0 / (1.0 + exp(-a.sig_alpha[0]));
    osc = a.sig_mode == 1 ? (float)wd : (float)(1.0 - wd);
  }
  const bool off = a.gate && !(a.gate[0] > 0.f);
  if (a.rowsum_out && by == 0 && t < 64 && m0 + t < a.M) atomicAdd(a.rowsum_out + m0 + t, off ? 0.f : rsum * osc);
  const int n = n0 + wn + (lane & 31);
  if (n >= a.N) return;
  const float bias = (a.bias && bz == 0) ? a.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m >= a.M) continue;
    float* c = a.C + (size_t)m * a.sc_m + (size_t)n * a.sc_n;
    float v = off ? 0.f : acc[r] * osc + bias;
    if (a.relu) v = v < 0.f ? 0.f : v;      // (NaN stays NaN, as torch.relu; fmaxf would return 0)
    if (a.mask && !(a.mask[(size_t)m * a.mask_m + n] > 0.f)) v = 0.f;
    if (a.ksplit > 1) atomicAdd(c, v);
    else if (a.accumulate) *c += v;
    else *c = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Wave-tile form (both operands 16-byte loadable): a workgroup owns ONE 32x32 output tile and its four waves split the
// K range; every wave stages its own 32 x 32 operand chunks through a private LDS region (no workgroup barrier in the
// K loop), and the four partial tiles meet in LDS at the end -> plain coalesced stores, no atomics and no pre-zeroed
// output unless the caller also splits K across workgroups (ksplit > 1).  These GEMMs are latency problems (0.3 GFLOP
// over 200-500 workgroups): 4x as many, 4x shorter dependent chains than the 64x64 form below.
// ------------------------------------------------------------------------------------------------
constexpr int WT_REGION = 2 * 32 * GP;     // floats per wave: A chunk + B chunk ([32][GP] k-fast or [32][32] row-fast)
// (element-wise A with a row-fast vector B: the two GEMMs that read a score-gradient matrix [batch][classes] -- head input
//  gradient and head weight gradient -- when the class count is not a multiple of four)
__host__ __device__ __forceinline__ bool gemm_wave_tiles(const GemmArgs& a) {
  const int am = gemm_wt_mode(a.A, a.M, a.K, a.sa_m, a.sa_k), bm = gemm_load_mode(a.Bm, a.N, a.K, a.sb_n, a.sb_k);
  if (bm == LD_SCALAR || am == LD_SCALAR) return false;
  return am == LD_VEC_K || am == LD_VEC_ROW || bm == LD_VEC_ROW;
}
__host__ __device__ __forceinline__ int gemm_nblocks(const GemmArgs& a) {
  const int ks = a.ksplit < 1 ? 1 : a.ksplit;
  if (gemm_wave_tiles(a)) return ((a.M + 31) / 32) * ((a.N + 31) / 32) * ks;
  return ((a.M + 63) / 64) * ((a.N + 63) / 64) * ks;
}
// one 32 x GK operand chunk -> 16 floats per lane (four 16-byte loads)
template <int MODE>
__device__ __forceinline__ void wt_load(f32x4 (&r)[4], const float* p, int rows, int r0, long s_row, long s_k, int k0, int kend, int lane) {
  if (MODE == LD_SC_K || MODE == LD_SC_ROW) {
    // element e = lane + 64 j: consecutive lanes read consecutive addresses (k-fast: 32 k of one row; row-fast: 32 rows of one k)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int e = lane + 64 * (4 * u + c);
        const int row = MODE == LD_SC_K ? e >> 5 : e & 31, k = MODE == LD_SC_K ? e & 31 : e >> 5;
        r[u][c] = (r0 + row < rows && k0 + k < kend) ? p[(size_t)(r0 + row) * s_row + (size_t)(k0 + k) * s_k] : 0.f;
      }
    return;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = lane + u * 64;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (MODE == LD_VEC_K) {
      const int row = i >> 3, k = (i & 7) * 4;
      if (r0 + row < rows && k0 + k < kend) v = *(const f32x4*)(p + (size_t)(r0 + row) * s_row + (k0 + k));
    } else {
      const int k = i >> 3, row = (i & 7) * 4;
      if (r0 + row < rows && k0 + k < kend) v = *(const f32x4*)(p + (size_t)(k0 + k) * s_k + (r0 + row));
    }
    r[u] = v;
  }
}
template <int MODE>
__device__ __forceinline__ void wt_store(float* S, const f32x4 (&r)[4], int lane) {
  if (MODE == LD_SC_K || MODE == LD_SC_ROW) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int e = lane + 64 * (4 * u + c);
        if (MODE == LD_SC_K) S[(e >> 5) * GP + kperm(e & 31)] = r[u][c];      // the LD_VEC_K image
        else S[(e >> 5) * 32 + (e & 31)] = r[u][c];                           // the LD_VEC_ROW image
      }
    return;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = lane + u * 64;
    if (MODE == LD_VEC_K) {
      const int row = i >> 3, k = (i & 7) * 4;        // k, k+2 -> even half; k+1, k+3 -> odd half (kperm)
      *(f32x2*)&S[row * GP + (k >> 1)] = f32x2{r[u][0], r[u][2]};
      *(f32x2*)&S[row * GP + 16 + (k >> 1)] = f32x2{r[u][1], r[u][3]};
    } else {
      const int k = i >> 3, row = (i & 7) * 4;
      *(f32x4*)&S[k * 32 + row] = r[u];
    }
  }
}
template <int MODE>
__device__ __forceinline__ void wt_fetch(float (&v)[16], const float* S, int row, int h) {
  if (MODE == LD_VEC_ROW || MODE == LD_SC_ROW) {
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = S[(2 * j + h) * 32 + row];
  } else {
    const f32x4* p = (const f32x4*)&S[row * GP + h * 16];
#pragma unroll
    for (int j = 0; j < 4; ++j) { f32x4 q = p[j]; v[4 * j] = q[0]; v[4 * j + 1] = q[1]; v[4 * j + 2] = q[2]; v[4 * j + 3] = q[3]; }
  }
}
// NWT waves per workgroup split the K range (4: the 256-thread launches; 8 / 16: the step's last launch, whose long-K
// weight-gradient GEMMs run beside the split-K slab reduction -- under its HBM load a chunk's round trip is ~3 us, so
// the dependent chunks per wave, not the flops, set that launch's duration)
template <int AM, int BM, int NWT = 4>
__device__ __forceinline__ void gemm_block_wt(const GemmArgs& a, int bx, int by, int bz, float* smem) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m0 = bx * 32, n0 = by * 32;
  const int ks = a.ksplit < 1 ? 1 : a.ksplit;
  const int kblk = ((a.K + ks - 1) / ks + GK - 1) / GK * GK;            // this workgroup's K range
  const int kb0 = bz * kblk, kb1 = min(a.K, kb0 + kblk);
  const int kper = (((kb1 - kb0) + NWT - 1) / NWT + GK - 1) / GK * GK;  // this wave's share of it
  const int kbeg = kb0 + wave * kper, kend = min(kb1, kbeg + kper);
  float* As = smem + wave * WT_REGION;
  float* Bs = As + 32 * GP;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float rsum = 0.f;
  f32x4 ra[4], rb[4];
  if (kbeg < kend) {
    wt_load<AM>(ra, a.A, a.M, m0, a.sa_m, a.sa_k, kbeg, kend, lane);
    wt_load<BM>(rb, a.Bm, a.N, n0, a.sb_n, a.sb_k, kbeg, kend, lane);
  }
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    // the region is private to this wave: program order (plus the compiler's lgkmcnt waits) is all the ordering needed
    wt_store<AM>(As, ra, lane);
    wt_store<BM>(Bs, rb, lane);
    if (k0 + GK < kend) {
      wt_load<AM>(ra, a.A, a.M, m0, a.sa_m, a.sa_k, k0 + GK, kend, lane);
      wt_load<BM>(rb, a.Bm, a.N, n0, a.sb_n, a.sb_k, k0 + GK, kend, lane);
    }
    __builtin_amdgcn_wave_barrier();
    if (a.rowsum_out && by == 0 && lane < 32) {
#pragma unroll
      for (int k = 0; k < GK; ++k) rsum += (AM == LD_VEC_ROW || AM == LD_SC_ROW) ? As[k * 32 + lane] : As[lane * GP + k];
    }
    float av[16], bv[16];
    wt_fetch<AM>(av, As, lane & 31, lane >> 5);
    wt_fetch<BM>(bv, Bs, lane & 31, lane >> 5);
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
  }
  float osc = 1.f;
  if (a.sig_mode) {
    const double wd = 1.0 / (1.0 + exp(-a.sig_alpha[0]));
    osc = a.sig_mode == 1 ? (float)wd : (float)(1.0 - wd);
  }
  const bool off = a.gate && !(a.gate[0] > 0.f);
  // partial tiles -> LDS ([wave][32][33]), then all 256 threads sum the four and store rows of 32 consecutive columns
  float* P = smem + wave * WT_REGION;
#pragma unroll
  for (int r = 0; r < 16; ++r) P[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 33 + (lane & 31)] = acc[r];
  if (a.rowsum_out && by == 0 && lane < 32) P[32 * 33 + lane] = rsum;
  __syncthreads();
  if (a.rowsum_out && by == 0 && t < 32 && m0 + t < a.M) {
    float rs = 0.f;
#pragma unroll
    for (int w = 0; w < NWT; ++w) rs += smem[w * WT_REGION + 32 * 33 + t];
    atomicAdd(a.rowsum_out + m0 + t, off ? 0.f : rs * osc);
  }
  const int n = n0 + (t & 31);
#pragma unroll
  for (int j = 0; j < 16 / NWT; ++j) {
    const int ml = (t >> 5) + 2 * NWT * j, m = m0 + ml;
    if (m >= a.M || n >= a.N) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NWT; ++w) v += smem[w * WT_REGION + ml * 33 + (t & 31)];
    v = off ? 0.f : v * osc + ((a.bias && bz == 0) ? a.bias[n] : 0.f);
    if (a.relu) v = v < 0.f ? 0.f : v;      // (NaN stays NaN, as torch.relu; fmaxf would return 0)
    if (a.mask && !(a.mask[(size_t)m * a.mask_m + n] > 0.f)) v = 0.f;
    float* c = a.C + (size_t)m * a.sc_m + (size_t)n * a.sc_n;
    if (ks > 1) atomicAdd(c, v);
    else if (a.accumulate) *c += v;
    else *c = v;
  }
}

// Several independent small GEMMs in one launch: block -> (problem, m-tile, n-tile, k-slice).
// WT_ONLY: every problem of the group takes the wave-tile form (the host checked: gemm_wave_tiles); the 64x64 any-stride
// program is then not part of the kernel -- it is what a 16-wave workgroup's 128-register budget cannot hold (31 spills)
template <int NWT = 4, bool WT_ONLY = false>
__device__ __forceinline__ void gemm_group_block(const GemmGroup& gg, float* smem) {
  int pi = 0;
  while (pi + 1 < gg.n && (int)blockIdx.x >= gg.start[pi + 1]) ++pi;
  const GemmArgs& a = gg.g[pi];
  int local = blockIdx.x - gg.start[pi];
  const int am = gemm_wt_mode(a.A, a.M, a.K, a.sa_m, a.sa_k), bm = gemm_load_mode(a.Bm, a.N, a.K, a.sb_n, a.sb_k);
  if (WT_ONLY || gemm_wave_tiles(a)) {
    const int tm = (a.M + 31) / 32, tn = (a.N + 31) / 32;
    const int bx = local % tm; local /= tm;
    const int by = local % tn;
    const int bz = local / tn;
    if (am == LD_VEC_K && bm == LD_VEC_K) gemm_block_wt<LD_VEC_K, LD_VEC_K, NWT>(a, bx, by, bz, smem);              // x W^T
    else if (am == LD_VEC_K && bm == LD_VEC_ROW) gemm_block_wt<LD_VEC_K, LD_VEC_ROW, NWT>(a, bx, by, bz, smem);     // dy W
    else if (am == LD_VEC_ROW && bm == LD_VEC_ROW) gemm_block_wt<LD_VEC_ROW, LD_VEC_ROW, NWT>(a, bx, by, bz, smem); // dy^T x
    else if (am == LD_SC_K) gemm_block_wt<LD_SC_K, LD_VEC_ROW, NWT>(a, bx, by, bz, smem);       // dy W, classes % 4 != 0
    else if (am == LD_SC_ROW) gemm_block_wt<LD_SC_ROW, LD_VEC_ROW, NWT>(a, bx, by, bz, smem);   // dy^T x, classes % 4 != 0
    else gemm_block_wt<LD_VEC_ROW, LD_VEC_K, NWT>(a, bx, by, bz, smem);
    return;
  }
  if constexpr (!WT_ONLY) {
    if (NWT > 4 && threadIdx.x >= 256) return;     // (the 64x64 form is a 256-thread program; exited waves leave its barriers)
    const int tm = (a.M + 63) / 64, tn = (a.N + 63) / 64;
    const int bx = local % tm; local /= tm;
    const int by = local % tn;
    const int bz = local / tn;
    gemm_block<LD_RUNTIME, LD_RUNTIME>(a, bx, by, bz, smem, smem + 64 * GP);   // any strides, any alignment
  }
}
constexpr int GEMM_SMEM_FLOATS = 4 * WT_REGION;     // 36 KiB (the 64x64 form needs 2 * 64 * GP of it)
__global__ __launch_bounds__(256) void k_gemm_group(GemmGroup gg) {
  __shared__ __attribute__((aligned(16))) float smem[GEMM_SMEM_FLOATS];
  gemm_group_block<4>(gg, smem);
}

int gemm_auto_ksplit(int M, int N, int K) {
  int tiles = ((M + 63) / 64) * ((N + 63) / 64);
  // measured on MI355X: beyond ~128 workgroups or K slices under 128 the same-address atomic chains cost more
  // than the shorter K loop saves
  constexpr int target = 128, kgran = 128;
  int ks = (target + tiles - 1) / tiles;
  int kmax = (K + kgran - 1) / kgran;
  if (ks > kmax) ks = kmax;
  return ks < 1 ? 1 : ks;
}

// wave-tile GEMMs split K inside the workgroup: no cross-workgroup split (and so no atomics, no pre-zeroed output,
// run-to-run identical sums) unless K is huge
static void gemm_group_plan(GemmGroup& gg) {
  for (int i = 0; i < gg.n; ++i)
    if (gemm_wave_tiles(gg.g[i]) && gg.g[i].K <= 8192) gg.g[i].ksplit = 1;
}

int launch_gemm_group(GemmGroup& gg, hipStream_t st) {
  if (gg.n == 0) return 0;
  gemm_group_plan(gg);
  int total = 0;
  for (int i = 0; i < gg.n; ++i) {
    gg.start[i] = total;
    total += gemm_nblocks(gg.g[i]);
  }
  gg.start[gg.n] = total;
  hipLaunchKernelGGL(k_gemm_group, dim3(total), dim3(256), 0, st, gg);
  DTA_CHECK_LAUNCH("k_gemm_group");
  return 0;
}

template <int NWT>
__global__ __launch_bounds__(NWT * 64) void k_gemm_group_reduce(GemmGroup gg, WgradReduceGroup gr, int ngemm) {
  extern __shared__ __attribute__((aligned(16))) float smem_dyn[];      // NWT * WT_REGION floats
  if (gr.slot_dst && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) gr.slot_dst[0] = (float)gr.slot_src[0];
  if ((int)blockIdx.x < ngemm) { gemm_group_block<NWT, (NWT > 4)>(gg, smem_dyn); return; }
  const int bx = blockIdx.x - ngemm;
  int j = 0;
  while (j + 1 < gr.n && bx >= gr.start[j + 1]) ++j;
  wgrad_reduce_blocks(gr.job[j], bx - gr.start[j], gr.start[j + 1] - gr.start[j]);
}
template <int NWT>
static void launch_gemm_group_reduce_t(const GemmGroup& gg, WgradReduceGroup& gr, int ngemm, hipStream_t st) {
  static DevOnce attr_once;
  const size_t lds = (size_t)NWT * WT_REGION * sizeof(float);
  if (attr_once.first()) hipFuncSetAttribute((const void*)k_gemm_group_reduce<NWT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int total = 0;       // (the jobs' block counts are planned for 256 threads)
  for (int j = 0; j < gr.n; ++j) { gr.start[j] = total; total += (wgrad_reduce_nblocks(gr.job[j]) * 4 + NWT - 1) / NWT; }
  gr.start[gr.n] = total;
  hipLaunchKernelGGL(k_gemm_group_reduce<NWT>, dim3(ngemm + total), dim3(NWT * 64), lds, st, gg, gr, ngemm);
}

int launch_gemm_group_with_reduce(GemmGroup& gg, WgradReduceGroup& gr, hipStream_t st, bool wide) {
  if (gg.n == 0) return launch_wgrad_reduce_group(gr, st);
  if (gr.n == 0) return launch_gemm_group(gg, st) || launch_wgrad_reduce_group(gr, st);      // (the second: the slot rider, if any)
  gemm_group_plan(gg);
  int ngemm = 0;
  for (int i = 0; i < gg.n; ++i) {
    gg.start[i] = ngemm;
    ngemm += gemm_nblocks(gg.g[i]);
  }
  gg.start[gg.n] = ngemm;
  static const int nwt_env = dev_getenv("DTA_TAIL_NWT") ? atoi(dev_getenv("DTA_TAIL_NWT")) : 0;
  // (measured, same box: 4 waves 25.6 us, 8 waves 27.5, 16 waves 24.0 -- DTA_TAIL_NWT=4 / 16 force a form)
  bool all_wt = true;      // the 16-wave kernel holds the wave-tile programs only
  for (int i = 0; i < gg.n; ++i) all_wt = all_wt && gemm_wave_tiles(gg.g[i]);
  if (nwt_env == 4 || !all_wt || (!wide && nwt_env != 16)) launch_gemm_group_reduce_t<4>(gg, gr, ngemm, st);
  else launch_gemm_group_reduce_t<16>(gg, gr, ngemm, st);
  DTA_CHECK_LAUNCH("k_gemm_group_reduce");
  return 0;
}

int launch_gemm(const GemmArgs& a, hipStream_t st) {
  GemmGroup gg;
  gg.n = 1;
  gg.g[0] = a;
  return launch_gemm_group(gg, st);
}

// ------------------------------------------------------------------------------------------------
// Column sums over the batch with a segment table: column j of A[rows][lda] lands in dst[seg][j - off].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_colsum_scatter(ColsumArgs a) {
  __shared__ float sc[16][16];
  colsum_scatter_block(a, blockIdx.x, sc);
}

int launch_colsum_scatter(const ColsumArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_colsum_scatter, dim3(colsum_nblocks(a)), dim3(1024), 0, st, a);
  DTA_CHECK_LAUNCH("k_colsum_scatter");
  return 0;
}

// spectral attention Conv1d weights [C][C][K]: only tap K/2 is live on a length-1 sequence.
// packed = [a1t | a2t | a1 | a2], each [C][C]; *t is input-major (a_t[i][o] = W[o][i][K/2]).
__global__ void k_pack_spectral_att(SpecPackGroup gr) {
  pack_spectral_att_job(gr, blockIdx.y, blockIdx.x * (size_t)blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}
int launch_pack_spectral_att_group(const SpecPackGroup& gr, hipStream_t st) {
  if (gr.n == 0) return 0;
  hipLaunchKernelGGL(k_pack_spectral_att, dim3(64, gr.n), dim3(256), 0, st, gr);
  DTA_CHECK_LAUNCH("k_pack_spectral_att");
  return 0;
}
int launch_pack_spectral_att(const float* w1, const float* w2, int C, int K, float* packed, hipStream_t st) {
  SpecPackGroup gr;
  gr.n = 1; gr.w1[0] = w1; gr.w2[0] = w2; gr.packed[0] = packed; gr.C[0] = C; gr.K[0] = K;
  return launch_pack_spectral_att_group(gr, st);
}

// ------------------------------------------------------------------------------------------------
// Hang2020 blend: joint = spec * sigmoid(alpha) + spat * (1 - sigmoid(alpha)); alpha is float64.
// ------------------------------------------------------------------------------------------------
// (blend2: ce_dev.h -- ONE definition for every kernel that blends)
__global__ void k_blend(BlendArgs a) {
  const double wd = 1.0 / (1.0 + exp(-a.alpha[0]));
  const float w = (float)wd, w1 = (float)(1.0 - wd);
  size_t n = (size_t)a.B * a.classes;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    a.joint[i] = blend2(a.spec[i], a.spat[i], w, w1);
}
int launch_blend(const BlendArgs& a, hipStream_t st) {
  size_t n = (size_t)a.B * a.classes;
  hipLaunchKernelGGL(k_blend, dim3((unsigned)min((size_t)1024, (n + 255) / 256)), dim3(256), 0, st, a);
  DTA_CHECK_LAUNCH("k_blend");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Year ensemble: scores = mean over the years' last-head scores (reference src/models/year.py:33).
// ------------------------------------------------------------------------------------------------
__global__ void k_mean_scores(MeanArgs a) {
  float kept = 0.f;
  unsigned use = 0u;                                  // bit k: source k takes part (a bit mask: an indexed array would live in scratch)
#pragma unroll
  for (int k = 0; k < MAXG; ++k)
    if (k < a.n && (!a.gate || a.gate[k] > 0.f)) { use |= 1u << k; kept += 1.f; }
  const float inv = 1.f / kept;                       // nothing kept: inf, and 0 * inf = NaN below -- an empty mean
  if (a.kept && blockIdx.x == 0 && threadIdx.x == 0) { a.kept[0] = kept; a.kept[1] = inv; }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.count; i += (size_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < MAXG; ++k)
      if ((use >> k) & 1u) acc += a.src[k][i];         // (selected, not multiplied: a skipped year's scores may be anything)
    a.dst[i] = acc * inv;
  }
}
// flags[y] = 1 when year y's tensor has a non-zero element (NaN counts), else 0: the reference's `x.sum() == 0` test of a
// missing year (year.py:27) for the non-negative crops its loader produces (a sum of non-negative floats is zero exactly
// when all of them are), decided on the device.  flags arrive zeroed; a block stops at its first hit.
struct YearPtrs { const float* x[MAXG]; };
__global__ __launch_bounds__(256) void k_year_flags(YearPtrs px, size_t n, float* flags, float* clear_next) {
  const int y = blockIdx.y;
  const float* p = px.x[y];
  const size_t n4 = n / 4;
  if (clear_next && blockIdx.x == 0 && threadIdx.x == 0) clear_next[y] = 0.f;      // the bank the NEXT call sets
  bool hit = false;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(p)[i];
    hit = !(v[0] == 0.f && v[1] == 0.f && v[2] == 0.f && v[3] == 0.f);
    if (__any(hit)) break;                            // this wave has its answer
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) hit = hit || !(p[i] == 0.f);
  if (hit) flags[y] = 1.f;
}
int launch_year_flags(const float* const* x, int years, size_t n, float* flags, float* clear_next, hipStream_t st) {
  YearPtrs px = {};
  for (int y = 0; y < years; ++y) px.x[y] = x[y];
  // flags must be zero on entry: cleared here, unless the caller alternates two banks and lets each call clear the other
  if (!clear_next && hipMemsetAsync(flags, 0, sizeof(float) * years, st) != hipSuccess) { dta_set_error("year flags: memset failed"); return 1; }
  hipLaunchKernelGGL(k_year_flags, dim3(64, years), dim3(256), 0, st, px, n, flags, clear_next);
  DTA_CHECK_LAUNCH("k_year_flags");
  return 0;
}
int launch_mean_scores(const MeanArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_mean_scores, dim3((unsigned)min((size_t)1024, (a.count + 255) / 256)), dim3(256), 0, st, a);
  DTA_CHECK_LAUNCH("k_mean_scores");
  return 0;
}

constexpr int BLEND_FIN_BLOCKS = 32;
// d(alpha) = w (1 - w) * sum_{b,n} djoint * (spec - spat): BLEND_FIN_BLOCKS blocks each reduce a slice of the
// (B, classes) slab and add it to dalpha, which must arrive zeroed like every other gradient buffer.
__device__ __forceinline__ void blend_bwd_fin_block(const BlendBwdArgs& a, double* sd, int part, int nparts) {
  double acc = 0;
  {
    const size_t n4 = (size_t)a.B * a.classes / 4, per = (n4 + nparts - 1) / nparts;
    const size_t beg = part * per, end = min(n4, beg + per);
    const f32x4* dj = (const f32x4*)a.djoint; const f32x4* sp = (const f32x4*)a.spec; const f32x4* st = (const f32x4*)a.spat;
    float f0 = 0.f, f1 = 0.f;
#pragma unroll 4
    for (size_t i = beg + threadIdx.x; i < end; i += 256) {
      const f32x4 d = dj[i], x = sp[i], y = st[i];
      f0 += d[0] * (x[0] - y[0]) + d[1] * (x[1] - y[1]);
      f1 += d[2] * (x[2] - y[2]) + d[3] * (x[3] - y[3]);
    }
    acc = (double)f0 + (double)f1;
    if (part == 0)   // tail when B * classes is not a multiple of 4
      for (size_t i = n4 * 4 + threadIdx.x; i < (size_t)a.B * a.classes; i += 256) acc += (double)(a.djoint[i] * (a.spec[i] - a.spat[i]));
  }
  sd[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sd[threadIdx.x] += sd[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const double wd = 1.0 / (1.0 + exp(-a.alpha[0]));
    // The 32 partials meet in one double through atomics, in whatever order the blocks finish.  Each is first rounded
    // to a multiple of 2^-50: sums of such numbers below 4 in magnitude are exact in double, so the result does not
    // depend on the order and the whole train step is bit-reproducible (cost: 4e-16 absolute per partial).
    const double pq = rint(sd[0] * wd * (1.0 - wd) * 0x1p50) * 0x1p-50;
    atomicAdd(a.dalpha, pq);
  }
}
__global__ __launch_bounds__(256) void k_blend_bwd_fin(BlendBwdArgs a) {
  __shared__ double sd[256];
  blend_bwd_fin_block(a, sd, blockIdx.x, gridDim.x);
}
// grouped GEMMs + trailing blocks that reduce the blend's d(alpha) (independent of the GEMMs)
__global__ __launch_bounds__(256) void k_gemm_group_fin(GemmGroup gg, BlendBwdArgs fin, int ngemm) {
  __shared__ __attribute__((aligned(16))) float smem[GEMM_SMEM_FLOATS];
  if ((int)blockIdx.x >= ngemm) { blend_bwd_fin_block(fin, reinterpret_cast<double*>(smem), blockIdx.x - ngemm, gridDim.x - ngemm); return; }
  gemm_group_block<4>(gg, smem);
}
int launch_gemm_group_with_blend_fin(GemmGroup& gg, const BlendBwdArgs& fin, hipStream_t st) {
  const int nfin = BLEND_FIN_BLOCKS;
  if (gg.n == 0) {
    hipLaunchKernelGGL(k_blend_bwd_fin, dim3(nfin), dim3(256), 0, st, fin);
    DTA_CHECK_LAUNCH("k_blend_bwd_fin");
    return 0;
  }
  gemm_group_plan(gg);
  int total = 0;
  for (int i = 0; i < gg.n; ++i) {
    gg.start[i] = total;
    total += gemm_nblocks(gg.g[i]);
  }
  gg.start[gg.n] = total;
  hipLaunchKernelGGL(k_gemm_group_fin, dim3(total + nfin), dim3(256), 0, st, gg, fin, total);
  DTA_CHECK_LAUNCH("k_gemm_group_fin");
  return 0;
}
// ------------------------------------------------------------------------------------------------
// F.cross_entropy(logits, y, weight=w): loss = sum_i w[y_i] * nll_i / sum_i w[y_i], plus dlogits.
// Label -100 is ignored (torch's default ignore_index); any other label outside [0, classes) is a caller bug (torch
// raises / device-asserts): it poisons the loss and that row's gradient with NaN so that it cannot train silently.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ce_rows(CeArgs a) {
  __shared__ float sc[4];
  const int t = threadIdx.x, lane = t & 63, row = blockIdx.x * 4 + (t >> 6);
  float part = 0.f;
  for (int i = t; i < a.B; i += 256) {
    long long y = a.labels[i];
    if (y >= 0 && y < a.classes) part += a.weight ? a.weight[y] : 1.f;
  }
  const float den = block_sum256(part, sc);
  if (blockIdx.x == 0 && t == 0) a.rowtmp[a.B] = den;
  if (row >= a.B) return;
  const float* z = a.logits + (size_t)row * a.classes;
  const long long y = a.labels[row];
  const bool ok = y >= 0 && y < a.classes;
  const float wy = ok ? (a.weight ? a.weight[y] : 1.f) : 0.f;
  float mx = -3.4e38f;
  for (int n = lane; n < a.classes; n += 64) mx = fmaxf(mx, z[n]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int n = lane; n < a.classes; n += 64) se += __expf(z[n] - mx);
  se = wave_sum(se);
  const float lse = __logf(se);
  const float poison = (ok || y == -100) ? 0.f : __builtin_nanf("");
  if (lane == 0) a.rowtmp[row] = ok ? wy * (lse + mx - z[y]) : poison;
  if (a.dlogits) {
    const float sc2 = (den > 0.f ? wy / den : 0.f) + poison;
    float* d = a.dlogits + (size_t)row * a.classes;
    for (int n = lane; n < a.classes; n += 64) {
      float p = __expf(z[n] - mx - lse);
      d[n] = sc2 * (p - ((ok && n == (int)y) ? 1.f : 0.f));
    }
  }
}
__global__ __launch_bounds__(256) void k_ce_fin(CeArgs a) {
  __shared__ double sd[256];
  double acc = 0;
  for (int r = threadIdx.x; r < a.B; r += 256) acc += a.rowtmp[r];
  sd[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sd[threadIdx.x] += sd[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) a.loss[0] = (float)(sd[0] / (double)a.rowtmp[a.B]);
}
// The same cross-entropy with the Hang2020 blend folded in (the blended scores never make a round trip through HBM
// before the loss reads them) and the loss finalised by the LAST block to arrive (a counter in the scratch, reset by
// that block): one launch instead of k_blend + k_ce_rows + k_ce_fin.  The row terms cross workgroups / XCDs through
// device-scope atomics (the per-XCD L2s are not coherent for plain accesses); the summation order of the loss is fixed.
__global__ __launch_bounds__(256) void k_blend_ce(BlendCeArgs a) {
  __shared__ float sc[8];
  __shared__ double sd[256];
  __shared__ int is_last;
  blend_ce_body(a, sc, sd, &is_last);
}
int launch_blend_ce(const BlendCeArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_blend_ce, dim3((a.B + 3) / 4), dim3(256), 0, st, a);
  DTA_CHECK_LAUNCH("k_blend_ce");
  return 0;
}
__global__ __launch_bounds__(256) void k_blend_ce_multi(BlendCeMulti m) {
  __shared__ float sc[8];
  __shared__ double sd[256];
  __shared__ int is_last;
  blend_ce_body(m.a[blockIdx.y], sc, sd, &is_last);
}
int launch_blend_ce_multi(const BlendCeMulti& m, hipStream_t st) {
  if (m.n < 1 || m.n > BLEND_CE_MULTI_MAX) { dta_set_error("blend_ce_multi: 1..%d losses", BLEND_CE_MULTI_MAX); return 1; }
  for (int i = 1; i < m.n; ++i)
    if (m.a[i].B != m.a[0].B) { dta_set_error("blend_ce_multi: every loss of the launch needs the same batch size"); return 1; }
  hipLaunchKernelGGL(k_blend_ce_multi, dim3((m.a[0].B + 3) / 4, m.n), dim3(256), 0, st, m);
  DTA_CHECK_LAUNCH("k_blend_ce_multi");
  return 0;
}
int launch_weighted_ce(const CeArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_ce_rows, dim3((a.B + 3) / 4), dim3(256), 0, st, a);
  DTA_CHECK_LAUNCH("k_ce_rows");
  hipLaunchKernelGGL(k_ce_fin, dim3(1), dim3(256), 0, st, a);
  DTA_CHECK_LAUNCH("k_ce_fin");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Inference epilogue: softmax over classes (reference multi_stage.py:302,315; main.py:190) plus the top-2 labels and
// scores main.py:192-205 extracts on the host.  One wave per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_softmax_top2(const float* logits, int B, int classes, float* probs,
                                                      long long* top_idx, float* top_score) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* z = logits + (size_t)row * classes;
  float mx = -3.4e38f;
  for (int n = lane; n < classes; n += 64) mx = fmaxf(mx, z[n]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int n = lane; n < classes; n += 64) se += __expf(z[n] - mx);
  se = wave_sum(se);
  const float inv = 1.f / se;
  float b1 = -1.f, b2 = -1.f;
  int i1 = -1, i2 = -1;
  for (int n = lane; n < classes; n += 64) {
    float pr = __expf(z[n] - mx) * inv;
    if (probs) probs[(size_t)row * classes + n] = pr;
    if (pr > b1) { b2 = b1; i2 = i1; b1 = pr; i1 = n; }
    else if (pr > b2) { b2 = pr; i2 = n; }
  }
  // merge the per-lane (best, second) pairs across the wave; ties resolve to the lower class index (as torch.topk)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ob1 = __shfl_xor(b1, o), ob2 = __shfl_xor(b2, o);
    int oi1 = __shfl_xor(i1, o), oi2 = __shfl_xor(i2, o);
    auto better = [](float a, int ia, float b, int ib) { return a > b || (a == b && ia >= 0 && (ib < 0 || ia < ib)); };
    float n1, n2; int j1, j2;
    if (better(b1, i1, ob1, oi1)) {
      n1 = b1; j1 = i1;
      if (better(b2, i2, ob1, oi1)) { n2 = b2; j2 = i2; } else { n2 = ob1; j2 = oi1; }
    } else {
      n1 = ob1; j1 = oi1;
      if (better(b1, i1, ob2, oi2)) { n2 = b1; j2 = i1; } else { n2 = ob2; j2 = oi2; }
    }
    b1 = n1; i1 = j1; b2 = n2; i2 = j2;
  }
  if (lane == 0) {
    top_idx[row * 2] = i1; top_idx[row * 2 + 1] = i2;
    top_score[row * 2] = b1; top_score[row * 2 + 1] = b2;
  }
}
__global__ __launch_bounds__(256) void k_softmax_top2_multi(SoftmaxMulti m) {
  const SoftmaxLevel& a = m.lv[blockIdx.y];
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m.B) return;
  const int classes = a.classes;
  float kept = 0.f;
  unsigned use = 0u;
#pragma unroll
  for (int k = 0; k < MAXG; ++k)
    if (k < a.nsrc && (!a.gate || a.gate[k] > 0.f)) { use |= 1u << k; kept += 1.f; }
  const float kinv = 1.f / kept;                      // nothing kept: inf, 0 * inf = NaN -- an empty mean, as k_mean_scores
  auto zat = [&](int n) __attribute__((always_inline)) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < MAXG; ++k)
      if ((use >> k) & 1u) acc += a.src[k][(size_t)row * classes + n];
    return acc * kinv;
  };
  // the first 256 classes of the row live in registers (four per lane); wider rows re-form the rest
  float zc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int n = lane + 64 * k; zc[k] = n < classes ? zat(n) : -3.4e38f; }
  // visit this lane's classes in ascending order: the register-held ones statically indexed, then the re-formed tail
  auto each = [&](auto&& f) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int n = lane + 64 * k; if (n < classes) f(n, zc[k]); }
    for (int n = lane + 256; n < classes; n += 64) f(n, zat(n));
  };
  float mx = -3.4e38f;
  each([&](int, float z) __attribute__((always_inline)) { mx = fmaxf(mx, z); });
  mx = wave_max(mx);
  float se = 0.f;
  each([&](int, float z) __attribute__((always_inline)) { se += __expf(z - mx); });
  se = wave_sum(se);
  const float inv = 1.f / se;
  float b1 = -1.f, b2 = -1.f;
  int i1 = -1, i2 = -1;
#define DTA_ROW_OUT(n_, z_)                                                                      \
  {                                                                                              \
    const int nn = (n_);                                                                         \
    const float zz = (z_);                                                                       \
    if (a.mean_out) a.mean_out[(size_t)row * classes + nn] = zz;                                 \
    const float pr = __expf(zz - mx) * inv;                                                      \
    if (a.probs) a.probs[(size_t)row * classes + nn] = pr;                                       \
    if (pr > b1) { b2 = b1; i2 = i1; b1 = pr; i1 = nn; }                                         \
    else if (pr > b2) { b2 = pr; i2 = nn; }                                                      \
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int n = lane + 64 * k; if (n < classes) DTA_ROW_OUT(n, zc[k]) }
  for (int n = lane + 256; n < classes; n += 64) DTA_ROW_OUT(n, zat(n))
#undef DTA_ROW_OUT
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ob1 = __shfl_xor(b1, o), ob2 = __shfl_xor(b2, o);
    int oi1 = __shfl_xor(i1, o), oi2 = __shfl_xor(i2, o);
    auto better = [](float a_, int ia, float b_, int ib) { return a_ > b_ || (a_ == b_ && ia >= 0 && (ib < 0 || ia < ib)); };
    float n1, n2; int j1, j2;
    if (better(b1, i1, ob1, oi1)) {
      n1 = b1; j1 = i1;
      if (better(b2, i2, ob1, oi1)) { n2 = b2; j2 = i2; } else { n2 = ob1; j2 = oi1; }
    } else {
      n1 = ob1; j1 = oi1;
      if (better(b1, i1, ob2, oi2)) { n2 = b1; j2 = i1; } else { n2 = ob2; j2 = oi2; }
    }
    b1 = n1; i1 = j1; b2 = n2; i2 = j2;
  }
  if (lane == 0) {
    if (a.top_idx) { a.top_idx[row * 2] = i1; a.top_idx[row * 2 + 1] = i2; }
    if (a.top_score) { a.top_score[row * 2] = b1; a.top_score[row * 2 + 1] = b2; }
  }
}
int launch_softmax_top2_multi(const SoftmaxMulti& m, hipStream_t st) {
  if (m.n < 1 || m.n > BLEND_CE_MULTI_MAX) { dta_set_error("softmax_top2_multi: 1..%d levels", BLEND_CE_MULTI_MAX); return 1; }
  hipLaunchKernelGGL(k_softmax_top2_multi, dim3((m.B + 3) / 4, m.n), dim3(256), 0, st, m);
  DTA_CHECK_LAUNCH("k_softmax_top2_multi");
  return 0;
}
int launch_softmax_top2(const float* logits, int B, int classes, float* probs, long long* top_idx, float* top_score,
                        hipStream_t st) {
  hipLaunchKernelGGL(k_softmax_top2, dim3((B + 3) / 4), dim3(256), 0, st, logits, B, classes, probs, top_idx, top_score);
  DTA_CHECK_LAUNCH("k_softmax_top2");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// torch.optim.Adam (defaults: no weight decay, no amsgrad) over one flat fp32 buffer + the fp64 alpha.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void adam_block(const AdamArgs& a) {
  float bc1 = a.bc1, bc2 = a.bc2;
  if (a.active) {
    // the step count advances on the device too: dev_step = steps taken so far, dev_step_out (a DIFFERENT word, read
    // by nobody during this launch) receives the count after this step
    if (a.dev_step_out && blockIdx.x == 0 && threadIdx.x == 0) a.dev_step_out[0] = a.dev_step[0] + (a.active[0] > 0.f ? 1 : 0);
    if (!(a.active[0] > 0.f)) {      // skipped everywhere: only optimizer.zero_grad()'s part of the pass
      float* z = a.gz ? a.gz : a.g_inactive;
      if (z)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) z[i] = 0.f;
      return;
    }
    // (one thread per block takes the two double-precision powers: per thread they cost more than the update itself)
    __shared__ float s_bc[2];
    if (threadIdx.x == 0) {
      const double st = (double)(a.dev_step[0] + 1);
      s_bc[0] = (float)(1.0 - pow((double)a.beta1, st)); s_bc[1] = (float)(1.0 - pow((double)a.beta2, st));
    }
    __syncthreads();
    bc1 = s_bc[0]; bc2 = s_bc[1];
  }
  const float ss = a.lr / bc1, rbc2 = rsqrtf(bc2);
  auto alpha_update = [&](double graw) {
    double g = graw * (double)a.grad_scale;
    double m = (double)a.beta1 * a.alpha_m[0] + (1.0 - (double)a.beta1) * g;
    double v = (double)a.beta2 * a.alpha_v[0] + (1.0 - (double)a.beta2) * g * g;
    a.alpha_m[0] = m; a.alpha_v[0] = v;
    a.alpha_p[0] -= ((double)a.lr / (double)bc1) * (m / (sqrt(v) / sqrt((double)bc2) + (double)a.eps));
    if (a.alpha_gz) a.alpha_gz[0] = 0.0;
  };
  // alpha's gradient in an exchange slot of g: the thread that owns that element steps alpha (it reads the slot before
  // the pass clears it)
  const bool slot_mode = a.alpha_p && a.alpha_g32;
  // 16-byte form (the flat buffers of the trainers / DtaAdam: every segment starts on a 16-byte boundary and is padded to
  // a multiple of four floats): a quarter of the memory instructions of the scalar loop below, same arithmetic per element
  const bool vec = (a.n & 3) == 0 && ((((size_t)a.p | (size_t)a.g | (size_t)a.m | (size_t)a.v | (size_t)a.gz) & 15) == 0);
  if (vec) {
    const size_t nq = a.n >> 2;
    for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < nq; q += (size_t)gridDim.x * blockDim.x) {
      const size_t i = q << 2;
      const f32x4 gr = *(const f32x4*)(a.g + i);
      const f32x4 mo = __builtin_nontemporal_load((const f32x4*)(a.m + i));
      const f32x4 vo = __builtin_nontemporal_load((const f32x4*)(a.v + i));
      f32x4 po = *(const f32x4*)(a.p + i), mn, vn;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float g = gr[c] * a.grad_scale;
        mn[c] = a.beta1 * mo[c] + (1.f - a.beta1) * g;
        vn[c] = a.beta2 * vo[c] + (1.f - a.beta2) * g * g;
        po[c] -= ss * (mn[c] / (sqrtf(vn[c]) * rbc2 + a.eps));
      }
      __builtin_nontemporal_store(mn, (f32x4*)(a.m + i));
      __builtin_nontemporal_store(vn, (f32x4*)(a.v + i));
      *(f32x4*)(a.p + i) = po;
      if (slot_mode && a.alpha_g32 >= a.g + i && a.alpha_g32 < a.g + i + 4) alpha_update((double)gr[(int)(a.alpha_g32 - (a.g + i))]);
      if (a.gz) { const f32x4 z = {0.f, 0.f, 0.f, 0.f}; *(f32x4*)(a.gz + i) = z; }
    }
    if (a.alpha_p && !a.alpha_g32 && blockIdx.x == 0 && threadIdx.x == 0) alpha_update(a.alpha_g[0]);
    return;
  }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    // moments: touched once per step, by this kernel only -> nontemporal both ways (they would only evict useful lines)
    const float graw = a.g[i];
    float g = graw * a.grad_scale;
    float m = a.beta1 * __builtin_nontemporal_load(a.m + i) + (1.f - a.beta1) * g;
    float v = a.beta2 * __builtin_nontemporal_load(a.v + i) + (1.f - a.beta2) * g * g;
    __builtin_nontemporal_store(m, a.m + i); __builtin_nontemporal_store(v, a.v + i);
    a.p[i] -= ss * (m / (sqrtf(v) * rbc2 + a.eps));
    if (slot_mode && a.g + i == a.alpha_g32) alpha_update((double)graw);
    if (a.gz) a.gz[i] = 0.f;
  }
  if (a.alpha_p && !a.alpha_g32 && blockIdx.x == 0 && threadIdx.x == 0) alpha_update(a.alpha_g[0]);
}
__global__ void k_adam(AdamArgs a) { adam_block(a); }
// several parameter groups in one launch (a year ensemble's per-year optimizers): blockIdx.y picks the group
__global__ void k_adam_multi(AdamMulti m) { adam_block(m.seg[blockIdx.y]); }
int launch_adam_multi(const AdamMulti& m, hipStream_t st) {
  size_t most = 1;
  for (int i = 0; i < m.n; ++i) {
    const AdamArgs& a = m.seg[i];
    const bool vec = (a.n & 3) == 0 && ((((size_t)a.p | (size_t)a.g | (size_t)a.m | (size_t)a.v | (size_t)a.gz) & 15) == 0);
    const size_t items = vec ? a.n / 4 : a.n;
    most = max(most, min((size_t)2048, (items + 255) / 256));
  }
  hipLaunchKernelGGL(k_adam_multi, dim3((unsigned)most, m.n), dim3(256), 0, st, m);
  DTA_CHECK_LAUNCH("k_adam_multi");
  return 0;
}
int launch_adam(const AdamArgs& a, hipStream_t st) {
  // (workgroups for the 16-byte form: a quad per thread)
  const bool vec = (a.n & 3) == 0 && ((((size_t)a.p | (size_t)a.g | (size_t)a.m | (size_t)a.v | (size_t)a.gz) & 15) == 0);
  const size_t items = vec ? a.n / 4 : a.n;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)max((size_t)1, min((size_t)2048, (items + 255) / 256))), dim3(256), 0, st, a);
  DTA_CHECK_LAUNCH("k_adam");
  return 0;
}

}  // namespace dta
