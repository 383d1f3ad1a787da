// Per-patch fused stages of the Hang2020 sub-networks on gfx950:
//   BatchNorm (batch or running statistics) -> ReLU -> optional 2x2 max-pool -> spectral / spatial attention
//   -> gated map written as conv tiles for the next layer + pooled classifier features,
// and the matching backward (recompute-from-conv-output, no saved activations besides the conv output).
// Restates /root/reference/src/models/Hang2020.py:24-31 (conv_module after the conv), :105-124
// (spatial_attention.forward) and :149-168 (spectral_attention.forward).
#include "kernels.h"
#include "ce_dev.h"

namespace dta {
#ifdef DTA_TICKS
// developer instrumentation: start / end time (100 MHz wall clock, common to all XCDs) of every workgroup of selected kernels
__device__ long long g_wgstamp_stage[4][8192][2];
extern "C" int dta_debug_wgstamps_stage(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgstamp_stage), sizeof(long long) * 4 * 8192 * 2); }
struct WgStamp {
  int id; long long t0;
  __device__ WgStamp(int id_) : id(id_), t0(wall_clock64()) {}
  __device__ ~WgStamp() {
    const int w = blockIdx.x + gridDim.x * blockIdx.y;
    if (threadIdx.x == 0 && id >= 0 && w < 8192) { g_wgstamp_stage[id][w][0] = t0; g_wgstamp_stage[id][w][1] = wall_clock64(); }
  }
};
#define WGSTAMP(id) WgStamp _wgstamp(id)
#else
#define WGSTAMP(id)
#endif

// ------------------------------------------------------------------------------------------------
// BatchNorm statistics: combine the conv workgroups' (mean, M2) partials (Chan et al.) in double.
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(1024) void k_bn_finalize(BnFinK a) {
  // block = 8 channels x 128 slices of the conv workgroups' partials (grid = (C/8, G)): at most nwg/128 independent
  // loads per thread, all in flight at once -- this launch is pure latency.  Single pass in double: with N = sum n_i,
  //   mean = sum n_i m_i / N,   M2 = sum M2_i + sum n_i m_i^2 - N mean^2
  // (m_i^2 is exact in double; the cancellation costs ~1e-13 of the variance)
  __shared__ double red[16][24];
  const int g = blockIdx.y, t = threadIdx.x, C = a.C;
  const int cl = t & 7, sl = t >> 3, lane = t & 63, wave = t >> 6, c = blockIdx.x * 8 + cl;
  float* coef = a.coef + (size_t)g * C * 4;
  if (!a.training) {
    if (t < 8 && c < C) {
      float rstd = rsqrtf(a.rvar[g][c] + a.eps);
      float sc = a.gamma[g][c] * rstd;
      coef[c * 4 + 0] = sc; coef[c * 4 + 1] = a.beta[g][c] - a.rmean[g][c] * sc;
      coef[c * 4 + 2] = a.rmean[g][c]; coef[c * 4 + 3] = rstd;
    }
    return;
  }
  const float* st = a.stats + (size_t)g * a.stats_goff;
  // the finishing threads' parameter loads go out with the partials (behind the barrier they would be a second
  // dependent round trip of this all-latency launch)
  float pgam = 0.f, pbet = 0.f, prm = 0.f, prv = 0.f, pgate = 1.f;
  if (t < 8 && c < C) {
    pgam = a.gamma[g][c]; pbet = a.beta[g][c];
    if (a.rmean[g]) { prm = a.rmean[g][c]; prv = a.rvar[g][c]; }
    if (a.gate) pgate = a.gate[g];
  }
  double s[3] = {0, 0, 0};
  if (c < C) {
#pragma unroll 4
    for (int wg = sl; wg < a.nwg; wg += 128) {
      const float2 v = *reinterpret_cast<const float2*>(st + ((size_t)wg * a.stats_ld + c) * 2);
      const double nb = conv_wg_count(wg, a.HW, a.MWG, a.B), m = (double)v.x;
      s[0] += nb * m; s[1] += nb * m * m; s[2] += (double)v.y;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double v = s[k];
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    if (lane < 8) red[wave][k * 8 + lane] = v;
  }
  __syncthreads();
  if (t < 8 && c < C) {
    double tot[3] = {0, 0, 0};
#pragma unroll
    for (int w = 0; w < 16; ++w) { tot[0] += red[w][t]; tot[1] += red[w][8 + t]; tot[2] += red[w][16 + t]; }
    const double n = (double)a.B * a.HW, mean = tot[0] / n;
    double m2 = tot[2] + tot[1] - n * mean * mean;
    m2 = m2 > 0 ? m2 : 0;
    double var = m2 / n;
    float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    float sc = pgam * rstd;
    coef[c * 4 + 0] = sc; coef[c * 4 + 1] = pbet - (float)mean * sc;
    coef[c * 4 + 2] = (float)mean; coef[c * 4 + 3] = rstd;
    if (a.rmean[g] && pgate > 0.f) {      // (a year the step skips keeps its statistics, year.py:27)
      double unb = n > 1 ? m2 / (n - 1) : var;
      a.rmean[g][c] = (1.f - a.momentum) * prm + a.momentum * (float)mean;
      a.rvar[g][c] = (1.f - a.momentum) * prv + a.momentum * (float)unb;
      if (c == 0 && a.nbt[g]) a.nbt[g][0] += 1;
    }
  }
}

BnFinK bn_finalize_kargs(const BnFinalizeArgs& b) {
  BnFinK a;
  a.stats = b.stats; a.nwg = b.nwg; a.HW = b.HW; a.MWG = b.MWG; a.B = b.B;
  a.fsum = b.fsum; a.fsum_ld = b.N;
  if (b.cat_mode) { a.C = b.nsplit; a.stats_goff = (size_t)b.nsplit * 2; a.stats_ld = b.N; a.fsum_goff = (size_t)b.nsplit * 3; }
  else { a.C = b.N; a.stats_goff = (size_t)b.nwg * b.N * 2; a.stats_ld = b.N; a.fsum_goff = (size_t)FAN_R * b.N * 3; }
  for (int g = 0; g < MAXG; ++g) {
    a.gamma[g] = b.gamma[g]; a.beta[g] = b.beta[g]; a.rmean[g] = b.rmean[g]; a.rvar[g] = b.rvar[g]; a.nbt[g] = b.nbt[g];
  }
  a.coef = b.coef; a.training = b.training; a.momentum = b.momentum; a.eps = b.eps; a.gate = b.gate;
  return a;
}

int launch_bn_finalize(const BnFinalizeArgs& b, int G, hipStream_t st) {
  BnFinK a = bn_finalize_kargs(b);
  hipLaunchKernelGGL(k_bn_finalize, dim3((a.C + 7) / 8, G), dim3(1024), 0, st, a);
  DTA_CHECK_LAUNCH("k_bn_finalize");
  return 0;
}

// The same statistics computed by a 256-thread stage workgroup for ALL C channels of its group, into LDS: lc[c][4] =
// (scale, shift, mean, rstd).  Single pass over the partials in double: with N = sum n_i,
//   mean = sum n_i m_i / N,   M2 = sum M2_i + sum n_i m_i^2 - N mean^2
// (m_i^2 is exact in double and the sums have <= 512 terms, so the cancellation costs ~1e-13 of the variance).
// Every workgroup runs the same instruction sequence on the same data: the coefficients are bit-identical everywhere.
// `red` = 2 NTHR + 2 floats of scratch.  Workgroup `writer` also publishes the coefficients and the running statistics.
template <int NTHR = 256>
__device__ __forceinline__ void bn_coef_block(const BnFinK& a, int g, float* lc, float* red, bool writer) {
  const int t = threadIdx.x, C = a.C;
  const int c = t % C, sl = t / C, T = NTHR / C;    // C in {32, 64, 128}: T slices of the partials per channel
  // 256 doubles, 8-byte aligned inside the 514-float scratch
  double* dred = reinterpret_cast<double*>(red + ((reinterpret_cast<uintptr_t>(red) >> 2) & 1));
  if (!a.training) {
    if (t < C) {
      const float rstd = rsqrtf(a.rvar[g][t] + a.eps), sc = a.gamma[g][t] * rstd;
      lc[t * 4 + 0] = sc; lc[t * 4 + 1] = a.beta[g][t] - a.rmean[g][t] * sc;
      lc[t * 4 + 2] = a.rmean[g][t]; lc[t * 4 + 3] = rstd;
    }
  } else {
    double s1 = 0, s2 = 0, s3 = 0;
    double tot[3];
    if (a.fsum) {
      // the conv launch folded its partials into FAN_R rows of raw sums already: thread c adds its channel's FAN_R rows
      // itself (24 independent loads in flight, fixed order: every workgroup computes bit-identical coefficients) -- no
      // reduction rounds through LDS
      if (t < C) {
        const double* fs = a.fsum + (size_t)g * a.fsum_goff + (size_t)t * 3;
        double r[FAN_R][3];
#pragma unroll
        for (int q = 0; q < FAN_R; ++q) { r[q][0] = fs[(size_t)q * a.fsum_ld * 3]; r[q][1] = fs[(size_t)q * a.fsum_ld * 3 + 1]; r[q][2] = fs[(size_t)q * a.fsum_ld * 3 + 2]; }
#pragma unroll
        for (int q = 0; q < FAN_R; ++q) { s1 += r[q][0]; s2 += r[q][1]; s3 += r[q][2]; }
      }
      tot[0] = s1; tot[1] = s2; tot[2] = s3;
    } else {
      const float* st = a.stats + (size_t)g * a.stats_goff;
#pragma unroll 8
      for (int wg = sl; wg < a.nwg; wg += T) {       // unrolled: independent L2 loads in flight
        const float2 v = *reinterpret_cast<const float2*>(st + ((size_t)wg * a.stats_ld + c) * 2);
        const double nb = conv_wg_count(wg, a.HW, a.MWG, a.B), m = (double)v.x;
        s1 += nb * m; s2 += nb * m * m; s3 += (double)v.y;
      }
    }
    if (!a.fsum) {
    const double part[3] = {s1, s2, s3};
#pragma unroll
    for (int k = 0; k < 3; ++k) {                  // three rounds through the 256-double scratch
      __syncthreads();
      dred[t] = part[k];
      __syncthreads();
      double acc = 0;
      if (t < C)
        for (int j = 0; j < T; ++j) acc += dred[j * C + t];
      tot[k] = acc;
    }
    }
    if (t < C) {
      const double n = (double)a.B * a.HW, mean = tot[0] / n;
      double m2 = tot[2] + tot[1] - n * mean * mean;
      m2 = m2 > 0 ? m2 : 0;
      const double var = m2 / n;
      const float rstd = (float)(1.0 / sqrt(var + (double)a.eps)), sc = a.gamma[g][t] * rstd;
      lc[t * 4 + 0] = sc; lc[t * 4 + 1] = a.beta[g][t] - (float)mean * sc;
      lc[t * 4 + 2] = (float)mean; lc[t * 4 + 3] = rstd;
      if (writer && a.rmean[g] && (!a.gate || a.gate[g] > 0.f)) {      // (a year the step skips keeps its statistics, year.py:27)
        const double unb = n > 1 ? m2 / (n - 1) : var;
        a.rmean[g][t] = (1.f - a.momentum) * a.rmean[g][t] + a.momentum * (float)mean;
        a.rvar[g][t] = (1.f - a.momentum) * a.rvar[g][t] + a.momentum * (float)unb;
        if (t == 0 && a.nbt[g]) a.nbt[g][0] += 1;
      }
    }
  }
  __syncthreads();
  if (writer && a.coef) {
    float* coef = a.coef + (size_t)g * C * 4;
    for (int i = t; i < C * 4; i += NTHR) coef[i] = lc[i];
  }
}

// "Lead" workgroups: the BatchNorm statistics WITHOUT a launch of their own.  The first `lead` workgroups of the consuming
// (stage) launch do k_bn_finalize's work -- workgroup idx = (group, 8-channel tile), NTHR / 8 slices of the conv partials per
// channel -- publish the coefficients with agent-scope stores and count themselves into `flag`; every other workgroup of the
// launch has its own loads (conv output, attention weights) in flight meanwhile and reads the coefficients once the count is
// complete (bn_lead_wait).  Lead workgroups have the lowest block ids, so they are dispatched before any waiter; the counter is
// cleared by the step's first launch (k_forward_prep).  scratch: NTHR / 64 * 24 doubles of LDS.
template <int NTHR>
__device__ __forceinline__ void bn_lead_block(const BnFinK& a, int idx, unsigned* flag, double* red) {
  constexpr int SL = NTHR / 8, NW = NTHR / 64;
  const int t = threadIdx.x, C = a.C, nblk = (C + 7) / 8;
  const int g = idx / nblk, cl = t & 7, sl = t >> 3, lane = t & 63, wave = t >> 6, c = (idx % nblk) * 8 + cl;
  float* coef = a.coef + (size_t)g * C * 4;
  const float* st = a.stats + (size_t)g * a.stats_goff;
  float pgam = 0.f, pbet = 0.f, prm = 0.f, prv = 0.f, pgate = 1.f;
  if (t < 8 && c < C) {
    pgam = a.gamma[g][c]; pbet = a.beta[g][c];
    if (a.rmean[g]) { prm = a.rmean[g][c]; prv = a.rvar[g][c]; }
    if (a.gate) pgate = a.gate[g];
  }
  double s[3] = {0, 0, 0};
  if (c < C) {
#pragma unroll 8
    for (int wg = sl; wg < a.nwg; wg += SL) {
      const float2 v = *reinterpret_cast<const float2*>(st + ((size_t)wg * a.stats_ld + c) * 2);
      const double nb = conv_wg_count(wg, a.HW, a.MWG, a.B), m = (double)v.x;
      s[0] += nb * m; s[1] += nb * m * m; s[2] += (double)v.y;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double v = s[k];
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    if (lane < 8) red[wave * 24 + k * 8 + lane] = v;
  }
  __syncthreads();
  if (t < 8 && c < C) {
    double tot[3] = {0, 0, 0};
#pragma unroll
    for (int w = 0; w < NW; ++w) { tot[0] += red[w * 24 + t]; tot[1] += red[w * 24 + 8 + t]; tot[2] += red[w * 24 + 16 + t]; }
    const double n = (double)a.B * a.HW, mean = tot[0] / n;
    double m2 = tot[2] + tot[1] - n * mean * mean;
    m2 = m2 > 0 ? m2 : 0;
    const double var = m2 / n;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps)), sc = pgam * rstd;
    fan_store2(coef + c * 4, sc, pbet - (float)mean * sc);
    fan_store2(coef + c * 4 + 2, (float)mean, rstd);
    if (a.rmean[g] && pgate > 0.f) {      // (a year the step skips keeps its statistics, year.py:27)
      const double unb = n > 1 ? m2 / (n - 1) : var;
      a.rmean[g][c] = (1.f - a.momentum) * prm + a.momentum * (float)mean;
      a.rvar[g][c] = (1.f - a.momentum) * prv + a.momentum * (float)unb;
      if (c == 0 && a.nbt[g]) a.nbt[g][0] += 1;
    }
  }
  if (wave == 0) {
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): the coefficient stores are performed
    if (t == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// The waiters' side: thread 0 polls the count (bounded: a lost lead workgroup poisons the step with NaN instead of hanging the
// device), then every thread may read the coefficients with bn_lead_coef2.
__device__ __forceinline__ bool bn_lead_wait(const unsigned* flag, unsigned need, int* ok_lds) {
  if (threadIdx.x == 0) {
    int ok = 0;
    for (int spin = 0; spin < (1 << 22); ++spin) {
      if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    *ok_lds = ok;
  }
  __syncthreads();
  return *ok_lds != 0;
}

// ------------------------------------------------------------------------------------------------
// Per-patch stage: LDS plan (floats).  Rows padded to C+1 so per-pixel loops over channels and
// per-channel loops over pixels are both bank-conflict free.  The spatial branch keeps its single-channel
// maps zero-padded by the stencil radius so the k x k stencils and their transposes run without bounds checks.
// ------------------------------------------------------------------------------------------------
struct StageGeom {
  int C, ld, Hc, Wc, HWc, Hz, Wz, HWz, vslot;
};
__host__ __device__ inline int stage_vslot(int C, int Hz, int Wz) {
  int pm = (Hz + 6) * (Wz + 6);   // padded map for stencil radius <= 3
  int v = C > pm ? C : pm;
  return v;
}
// Compile-time specialisation of a stage: CT channels, HT x WT conv-resolution map, PT = 2x2 pool after ReLU.
// HT == 0 keeps the geometry (and the stencil sizes) as run-time values.
// YT = storage format of the conv output the stage reads (FMT_F32, or FMT_F16 in bf16 mode): compile-time, so that the
// loads of the unrolled fetch loops stay free of branches.
template <int CT, int HT, int WT, int PT, int YT = FMT_F32>
struct StageCfg {
  static constexpr int C = CT, H = HT, W = WT, P = PT, YF = YT;
  static constexpr bool fixed = HT > 0;
};
template <typename CFG>
__device__ __forceinline__ StageGeom stage_geom(const StageArgs& a) {
  StageGeom s;
  s.C = CFG::C ? CFG::C : a.C; s.ld = s.C + 1;
  s.Hc = CFG::fixed ? CFG::H : a.Hc; s.Wc = CFG::fixed ? CFG::W : a.Wc; s.HWc = s.Hc * s.Wc;
  const bool pool = CFG::fixed ? (CFG::P != 0) : (a.pool != 0);
  s.Hz = pool ? s.Hc / 2 : s.Hc; s.Wz = pool ? s.Wc / 2 : s.Wc; s.HWz = s.Hz * s.Wz;
  s.vslot = a.vslot;
  return s;
}
// spatial-attention stencil size / class-pool size are functions of the channel count in the reference
// (Hang2020.py:77-99): compile-time when the stage is specialised
template <typename CFG> __device__ __forceinline__ int cfg_att_k(const StageArgs& a, int g) {
  return CFG::fixed ? (CFG::C == 32 ? 7 : CFG::C == 64 ? 5 : 3) : a.att_k[g];
}
template <typename CFG> __device__ __forceinline__ int cfg_att_pool(const StageArgs& a, int g) {
  return CFG::fixed ? (CFG::C == 32 ? 4 : CFG::C == 64 ? 2 : 1) : a.att_pool[g];
}
template <typename CFG> __device__ __forceinline__ bool cfg_pool(const StageArgs& a) {
  return CFG::fixed ? (CFG::P != 0) : (a.pool != 0);
}
// LDS vector slot size: padded single-channel maps only when a spatial-attention group is present
int stage_vslot_for(const StageArgs& a, int G) {
  int Hz = a.pool ? a.Hc / 2 : a.Hc, Wz = a.pool ? a.Wc / 2 : a.Wc;
  bool spatial = false;
  for (int g = 0; g < G; ++g) spatial |= a.kind[g] == KIND_SPATIAL;
  return spatial ? stage_vslot(a.C, Hz, Wz) : a.C;
}
static size_t stage_lds_floats(const StageArgs& a, bool bwd) {
  int Hz = a.pool ? a.Hc / 2 : a.Hc, Wz = a.pool ? a.Wc / 2 : a.Wc;
  int HWz = Hz * Wz, ld = a.C + 1;
  size_t n = (size_t)HWz * ld;                 // Z
  if (bwd) n += (size_t)HWz * ld;               // D
  // vectors / padded maps + reduction scratch.  backward: v0..v4 full slots, v5 per-pixel-or-channel, v6 per-channel
  n += bwd ? (size_t)5 * a.vslot + (HWz > a.C ? HWz : a.C) + a.C + 512 : (size_t)4 * a.vslot + 512 + 2 + 4 * a.C;
  return n;
}

// Partitioned reduction helper: 256 threads = (256/C) slices x C columns.  f(c, i) summed over i in [0, n).
template <typename F>
__device__ __forceinline__ void colreduce(int C, int n, float* scratch, float* out, float scale, F f) {
  const int t = threadIdx.x, c = t % C, sl = t / C, nsl = 256 / C;
  float acc = 0.f;
  if (sl < nsl) {
#pragma unroll 8
    for (int i = sl; i < n; i += nsl) acc += f(c, i);   // unrolled: independent (global) loads overlap
  }
  __syncthreads();
  scratch[t] = acc;
  __syncthreads();
  if (t < C) {
    float s = 0.f;
    for (int k = 0; k < nsl; ++k) s += scratch[k * C + t];
    out[t] = s * scale;
  }
  __syncthreads();
}

// Per-pixel reduction over channels, all 256 threads busy: LP = 2^k lanes (LP <= 64, LP * npix <= 256) share a
// pixel, each sums channels c = l, l+LP, ..., then a butterfly over the LP lanes.  done(p, sum) runs on one lane.
template <typename F, typename G>
__device__ __forceinline__ void pixreduce(int npix, int C, F f, G done) {
  int LP = 64;
  while (LP > 1 && LP * npix > 256) LP >>= 1;
  if (LP == 1) {
    for (int p = threadIdx.x; p < npix; p += 256) {
      float acc = 0.f;
      for (int c = 0; c < C; ++c) acc += f(p, c);
      done(p, acc);
    }
    return;
  }
  const int p = threadIdx.x / LP, l = threadIdx.x % LP;
  float acc = 0.f;
  if (p < npix)
    for (int c = l; c < C; c += LP) acc += f(p, c);
  for (int o = LP >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (p < npix && l == 0) done(p, acc);
}

// k x k cross-correlation of a zero-padded single-channel map (radius r = k/2, row pitch Wp = Wz + 2r) at pixel
// (h, w): no bounds checks.  flip = true gives the transposed stencil (gradient wrt the input map).
__device__ __forceinline__ float stencil_at(const float* mp, const float* kw, int k, int Wp, int h, int w, bool flip) {
  float acc = 0.f;
  const float* row = mp + h * Wp + w;
  if (!flip) {
    for (int ky = 0; ky < k; ++ky, row += Wp)
      for (int kx = 0; kx < k; ++kx) acc += kw[ky * k + kx] * row[kx];
  } else {
    for (int ky = 0; ky < k; ++ky, row += Wp)
      for (int kx = 0; kx < k; ++kx) acc += kw[(k - 1 - ky) * k + (k - 1 - kx)] * row[kx];
  }
  return acc;
}

// Forward recompute shared by both kernels.  On return (all threads synced):
//   Z [HWz][ld]  post BN/ReLU/pool activations (pooled straight from global memory: no pre-pool tile in LDS)
//   spectral: v0 = pooled, v1 = h (post ReLU), v2 = gate
//   spatial : v0 = m (post ReLU, zero-padded map), v1 = t1 (post ReLU, zero-padded map), v2 = s (gate, per pixel)
template <typename CFG>
__device__ __forceinline__ void stage_forward(const StageArgs& a, const StageGeom& s, int g, int b, int kind,
                                              float* Z, float* v0, float* v1, float* v2, float* scratch,
                                              bool use_saved = false, const float* da = nullptr, float* D = nullptr,
                                              bool z_ready = false, const float* lcoef = nullptr) {
  constexpr int CT = CFG::C;
  const bool pool = cfg_pool<CFG>(a);
  const int t = threadIdx.x, C = s.C, ld = s.ld;
  const size_t ybase = (size_t)g * a.y_gs + (size_t)b * s.HWc * a.y_rs;   // element index: the conv output may be 16-bit
  constexpr int yf = CFG::YF;
  const float* coef = lcoef ? lcoef : (a.coef ? a.coef + (size_t)g * a.coef_gs : nullptr);
  if (use_saved) {   // v0 | v1 | v2 as the forward kernel left them (padded maps include their zero borders)
    const float* src = a.attsave + ((size_t)g * a.B + b) * a.attsave_ld;
    if (kind == KIND_SPECTRAL) {   // three C-vectors, stored packed
      for (int i = t; i < 3 * C; i += 256) { const int k = i / C; v0[k * s.vslot + (i - k * C)] = __builtin_nontemporal_load(src + i); }
    } else if (kind == KIND_SPATIAL) {
      for (int i = t; i < 3 * s.vslot; i += 256) v0[i] = __builtin_nontemporal_load(src + i);
    }
  } else if (kind == KIND_SPATIAL) {   // zero the padded maps' borders (interiors are overwritten below)
    for (int i = t; i < 2 * s.vslot; i += 256) v0[i] = 0.f;   // v0 and v1 are adjacent
  }
  // backward of a specialised stage: the incoming gradient ([HWz][C], contiguous) is fetched into registers ahead
  // of the activations so both sets of global loads are in flight together; it lands in D once Z is built
  constexpr int HZT = CFG::fixed ? (CFG::P ? CFG::H / 2 : CFG::H) * (CFG::P ? CFG::W / 2 : CFG::W) : 0;
  constexpr int ND = CFG::fixed ? (HZT * CT + 255) / 256 : 1;
  float rd[ND];
  if (CFG::fixed && da) {
#pragma unroll
    for (int u = 0; u < ND; ++u) { const int i = t + u * 256; rd[u] = i < HZT * CT ? __builtin_nontemporal_load(da + i) : 0.f; }
  }
  // the forward's next reader of the conv output is the backward, far away: load it nontemporally there
  auto ldy = [&](size_t i_) { return use_saved ? ld_fmt<false>(a.y, ybase + i_, yf) : ld_fmt<true>(a.y, ybase + i_, yf); };
  if (z_ready) {
    // the caller already built Z from prefetched registers
  } else if (CT > 0 && CT % 8 == 0 && yf != FMT_F32 && (a.y_rs & 7) == 0 && (ybase & 7) == 0) {
    // 16-bit conv output with 16-byte aligned rows: a thread owns one 8-channel octet (256 % (C / 8) == 0, so the octet
    // and its BN coefficients are fixed per thread) and fetches it as ONE 16-byte load per pixel -- the scalar form below
    // issues eight 2-byte loads and eight address computations for the same bytes, and these generic kernels are
    // instruction-issue bound (24x24 crops: 72 -> 9 fetch iterations per thread)
    constexpr int NO = (CT > 0 ? CT : 8) / 8;
    const int o = t % NO, p0 = t / NO, pstep = 256 / NO;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = a.apply_bn ? coef[(o * 8 + e) * 4 + 0] : 1.f;
      sh[e] = a.apply_bn ? coef[(o * 8 + e) * 4 + 1] : 0.f;
    }
    const unsigned short* y16 = (const unsigned short*)a.y + ybase + o * 8;
    auto ld8 = [&](size_t i_, float (&v)[8]) {
      const u32x4* pq = (const u32x4*)(y16 + i_);
      const u32x4 q = use_saved ? *pq : __builtin_nontemporal_load(pq);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] = unpack_lo(q[e], yf) * sc[2 * e] + sh[2 * e]; v[2 * e + 1] = unpack_hi(q[e], yf) * sc[2 * e + 1] + sh[2 * e + 1]; }
    };
    if (!pool) {
#pragma unroll 4
      for (int p = p0; p < s.HWc; p += pstep) {
        float v[8];
        ld8((size_t)p * a.y_rs, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) Z[p * ld + o * 8 + e] = a.relu ? relu_nan(v[e]) : v[e];
      }
    } else {
#pragma unroll 2
      for (int pz = p0; pz < s.HWz; pz += pstep) {
        const int hz = pz / s.Wz, wz = pz - hz * s.Wz;
        const size_t y0 = (size_t)((2 * hz) * s.Wc + 2 * wz) * a.y_rs;
        float v0_[8], v1_[8], v2_[8], v3_[8];
        ld8(y0, v0_); ld8(y0 + a.y_rs, v1_); ld8(y0 + (size_t)s.Wc * a.y_rs, v2_); ld8(y0 + (size_t)(s.Wc + 1) * a.y_rs, v3_);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float m = max_nan(max_nan(v0_[e], v1_[e]), max_nan(v2_[e], v3_[e]));
          Z[pz * ld + o * 8 + e] = a.relu ? relu_nan(m) : m;
        }
      }
    }
  } else if (CT > 0) {
    // 256 % C == 0: every thread keeps one channel, its BN coefficients live in registers
    const int c = t % C, p0 = t / C, pstep = 256 / C;
    const float sc = a.apply_bn ? coef[c * 4 + 0] : 1.f, sh = a.apply_bn ? coef[c * 4 + 1] : 0.f;
    if (!pool) {
#pragma unroll 8
      for (int p = p0; p < s.HWc; p += pstep) {      // unrolled: independent global loads in flight per thread
        float v = ldy((size_t)p * a.y_rs + c) * sc + sh;
        if (a.relu) v = relu_nan(v);
        Z[p * ld + c] = v;
      }
    } else {
#pragma unroll 2
      for (int pz = p0; pz < s.HWz; pz += pstep) {   // 2x2 max-pool straight from global memory (floor: last row/col dropped)
        int hz = pz / s.Wz, wz = pz - hz * s.Wz;
        const size_t y0 = (size_t)((2 * hz) * s.Wc + 2 * wz) * a.y_rs + c;
        float v0_ = ldy(y0) * sc + sh, v1_ = ldy(y0 + a.y_rs) * sc + sh;
        float v2_ = ldy(y0 + (size_t)s.Wc * a.y_rs) * sc + sh, v3_ = ldy(y0 + (size_t)(s.Wc + 1) * a.y_rs) * sc + sh;
        float m = max_nan(max_nan(v0_, v1_), max_nan(v2_, v3_));
        if (a.relu) m = relu_nan(m);
        Z[pz * ld + c] = m;
      }
    }
  } else {
    const int n = pool ? s.HWz : s.HWc;
    for (int i = t; i < n * C; i += 256) {
      int p = i / C, c = i - p * C;
      const float sc = a.apply_bn ? coef[c * 4 + 0] : 1.f, sh = a.apply_bn ? coef[c * 4 + 1] : 0.f;
      float v;
      if (!pool) v = ldy((size_t)p * a.y_rs + c) * sc + sh;
      else {
        int hz = p / s.Wz, wz = p - hz * s.Wz;
        const size_t y0 = (size_t)((2 * hz) * s.Wc + 2 * wz) * a.y_rs + c;
        v = max_nan(max_nan(ldy(y0) * sc + sh, ldy(y0 + a.y_rs) * sc + sh),
                  max_nan(ldy(y0 + (size_t)s.Wc * a.y_rs) * sc + sh, ldy(y0 + (size_t)(s.Wc + 1) * a.y_rs) * sc + sh));
      }
      if (a.relu) v = relu_nan(v);
      Z[p * ld + c] = v;
    }
  }
  if (CFG::fixed && da) {
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = t + u * 256;
      if (i < HZT * CT) { const int p = i / (CT > 0 ? CT : 1), c = i - p * CT; D[p * ld + c] = rd[u]; }
    }
  }
  __syncthreads();
  if (use_saved) return;
  if (kind == KIND_SPECTRAL) {
    const float* a1t = a.att[g].p[0]; const float* c1 = a.att[g].p[1];
    const float* a2t = a.att[g].p[2]; const float* c2 = a.att[g].p[3];
    colreduce(C, s.HWz, scratch, v0, 1.f / (float)s.HWz, [&](int c, int i) { return Z[i * ld + c]; });
    colreduce(C, C, scratch, v1, 1.f, [&](int o, int i) { return a1t[i * C + o] * v0[i]; });
    if (t < C) v1[t] = relu_nan(v1[t] + c1[t]);
    __syncthreads();
    colreduce(C, C, scratch, v2, 1.f, [&](int o, int i) { return a2t[i * C + o] * v1[i]; });
    if (t < C) v2[t] = sigmoidf_(v2[t] + c2[t]);
    __syncthreads();
  } else if (kind == KIND_SPATIAL) {
    const float* wc = a.att[g].p[0]; const float bc = a.att[g].p[1][0];
    const float* k1 = a.att[g].p[2]; const float b1 = a.att[g].p[3][0];
    const float* k2 = a.att[g].p[4]; const float b2 = a.att[g].p[5][0];
    const int k = cfg_att_k<CFG>(a, g), r = k / 2, Wp = s.Wz + 2 * r;
    pixreduce(s.HWz, C, [&](int p, int c) { return wc[c] * Z[p * ld + c]; },
              [&](int p, float acc) {
                int h = p / s.Wz, w = p - h * s.Wz;
                v0[(h + r) * Wp + w + r] = relu_nan(acc + bc);
              });
    __syncthreads();
    for (int p = t; p < s.HWz; p += 256) {
      int h = p / s.Wz, w = p - h * s.Wz;
      v1[(h + r) * Wp + w + r] = relu_nan(b1 + stencil_at(v0, k1, k, Wp, h, w, false));
    }
    __syncthreads();
    for (int p = t; p < s.HWz; p += 256) {
      int h = p / s.Wz, w = p - h * s.Wz;
      v2[p] = sigmoidf_(b2 + stencil_at(v1, k2, k, Wp, h, w, false));
    }
    __syncthreads();
  }
}

__device__ __forceinline__ float gate_of(int kind, const float* v2, int p, int c) {
  return kind == KIND_SPECTRAL ? v2[c] : (kind == KIND_SPATIAL ? v2[p] : 1.f);
}

// Forward of one patch (b, g).  z_ready: the caller already built Z (BN + ReLU of the conv output) in LDS.
template <typename T, typename CFG>
__device__ __forceinline__ void stage_fwd_patch(const StageArgs& a, const StageGeom& s, int b, int g, bool z_ready,
                                                float* Z, float* v0, float* v1, float* v2, float* scratch,
                                                const float* lcoef = nullptr) {
  const int t = threadIdx.x, C = s.C, ld = s.ld;
  const int kind = a.kind[g];
  stage_forward<CFG>(a, s, g, b, kind, Z, v0, v1, v2, scratch, false, nullptr, nullptr, z_ready, lcoef);
  if (a.attsave) {
    float* dst = a.attsave + ((size_t)g * a.B + b) * a.attsave_ld;
    if (kind == KIND_SPECTRAL) {
      for (int i = t; i < 3 * C; i += 256) { const int k = i / C; __builtin_nontemporal_store(v0[k * s.vslot + (i - k * C)], dst + i); }
    } else if (kind == KIND_SPATIAL) {
      for (int i = t; i < 3 * s.vslot; i += 256) __builtin_nontemporal_store(v0[i], dst + i);   // read back by the backward only
    }
  }

  // classifier features
  if (a.feat) {
    float* f = a.feat + (size_t)g * a.feat_gs + (size_t)b * a.F[g];
    if (kind == KIND_SPECTRAL) {
      if (t < C) f[t] = v2[t] * v0[t];  // mean_p(z*gate) == gate * mean_p(z)
    } else if (kind == KIND_SPATIAL) {
      const int ps = cfg_att_pool<CFG>(a, g), hp = s.Hz / ps, wp = s.Wz / ps;
      for (int i = t; i < C * hp * wp; i += 256) {
        int c = i / (hp * wp), rem = i - c * hp * wp, ph = rem / wp, pw = rem - ph * wp;
        float m = -3.4e38f;
        for (int dy = 0; dy < ps; ++dy)
          for (int dx = 0; dx < ps; ++dx) {
            int p = (ph * ps + dy) * s.Wz + pw * ps + dx;
            m = max_nan(m, Z[p * ld + c] * v2[p]);
          }
        f[i] = m;
      }
    } else if (a.F[g] > 0) {
      for (int i = t; i < C * s.HWz; i += 256) { int c = i / s.HWz, p = i - c * s.HWz; f[i] = Z[p * ld + c]; }
    }
  }
  // gated map as conv tiles for the next layer (zero halo written here: the workspace is borrowed, not ours)
  if (a.a_tl) {
    const int W2 = s.Wz + 2, Qz = (s.Hz + 2) * W2, nch = C / 16;
    T* dst = (T*)a.a_tl + (size_t)g * a.a_gs + ((size_t)b * a.a_nc + a.a_ch0) * Qz * 16;
    constexpr int VW = TlVec<T>::VW, PARTS = 16 / VW;
    for (int i = t; i < nch * Qz * PARTS; i += 256) {
      int ch = i / (Qz * PARTS), rem = i - ch * Qz * PARTS, q = rem / PARTS, part = rem % PARTS;
      int hh = q / W2 - 1, ww = q % W2 - 1;
      const bool in = hh >= 0 && hh < s.Hz && ww >= 0 && ww < s.Wz;
      const int p = in ? hh * s.Wz + ww : 0;
      float v[VW];
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        int c = ch * 16 + tl_pos<T>(q, part * VW + j);
        v[j] = in ? Z[p * ld + c] * gate_of(kind, v2, p, c) : 0.f;
      }
      tl_store_vec(dst + ((size_t)ch * Qz + q) * 16, part, v);
    }
  }
  if (a.a_nchw) {
    float* dst = a.a_nchw + (size_t)g * a.a_nchw_gs + (size_t)b * C * s.HWz;
    for (int i = t; i < C * s.HWz; i += 256) {
      int c = i / s.HWz, p = i - c * s.HWz;
      dst[i] = Z[p * ld + c] * gate_of(kind, v2, p, c);
    }
  }
}

template <typename T, typename CFG>
__global__ __launch_bounds__(256) void k_stage_fwd(StageArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const StageGeom s = stage_geom<CFG>(a);
  const int g = blockIdx.y, t = threadIdx.x, ld = s.ld;
  float* Z = sm;
  float* v0 = Z + (size_t)s.HWz * ld;
  float* v1 = v0 + s.vslot; float* v2 = v1 + s.vslot; float* v3 = v2 + s.vslot;
  float* scratch = v3 + s.vslot;
  float* lcoef = nullptr;
  // Un-pooled network stage launched with half a grid: two patches per workgroup, the second one's conv output is
  // fetched into registers while the first is processed (see k_stage_bwd).
  constexpr bool PIPE = CFG::fixed && CFG::P == 0;
  if (PIPE && (int)gridDim.x < a.B) {
    constexpr int NQ = PIPE ? (CFG::H * CFG::W * CFG::C + 255) / 256 : 1;
    constexpr int CQ = PIPE ? CFG::C : 1, NEL = PIPE ? CFG::H * CFG::W * CFG::C : 0;
    float ry[NQ];
#define DTA_STAGE_ISSUE(b_)                                                                          \
    {                                                                                                \
      const size_t y_ = (size_t)g * a.y_gs + (size_t)(b_) * s.HWc * a.y_rs;                          \
      _Pragma("unroll") for (int u = 0; u < NQ; ++u) {                                               \
        const int i = t + u * 256;                                                                   \
        if (i < NEL) ry[u] = ld_fmt<true>(a.y, y_ + (size_t)(i / CQ) * a.y_rs + (i % CQ), CFG::YF);  /* next reader: the backward */ \
      }                                                                                              \
    }
#define DTA_STAGE_LAND()                                                                             \
    _Pragma("unroll") for (int u = 0; u < NQ; ++u) {                                                 \
      const int i = t + u * 256;                                                                     \
      if (i < NEL) Z[(i / CQ) * ld + (i % CQ)] = relu_nan(ry[u] * psc + psh);                      \
    }
    const int b0 = blockIdx.x, b1 = blockIdx.x + gridDim.x;
    DTA_STAGE_ISSUE(b0)          // the first patch's conv output is in flight while the statistics are combined
    const float* coef = a.coef + (size_t)g * a.coef_gs;
    if (a.bn_inkernel) {
      lcoef = scratch + 514;
      bn_coef_block(a.bnfin, g, lcoef, scratch, blockIdx.x == 0);
      coef = lcoef;
    }
    const float psc = coef[(t % CQ) * 4 + 0], psh = coef[(t % CQ) * 4 + 1];
    DTA_STAGE_LAND()
    if (b1 < a.B) DTA_STAGE_ISSUE(b1)
    stage_fwd_patch<T, CFG>(a, s, b0, g, true, Z, v0, v1, v2, scratch, lcoef);
    if (b1 < a.B) {
      __syncthreads();   // the second patch reuses the LDS tiles
      DTA_STAGE_LAND()
      stage_fwd_patch<T, CFG>(a, s, b1, g, true, Z, v0, v1, v2, scratch, lcoef);
    }
#undef DTA_STAGE_LAND
#undef DTA_STAGE_ISSUE
    return;
  }
  if (a.bn_inkernel) {
    lcoef = scratch + 514;
    bn_coef_block(a.bnfin, g, lcoef, scratch, blockIdx.x == 0);
  }
  stage_fwd_patch<T, CFG>(a, s, blockIdx.x, g, false, Z, v0, v1, v2, scratch, lcoef);
}

template <typename T, typename CFG>
static int launch_stage_fwd_c(const StageArgs& a, int G, size_t lds, hipStream_t st) {
  static DevOnce attr_once;      // (function attributes are per device)
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)k_stage_fwd<T, CFG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const bool pipe = CFG::fixed && CFG::P == 0 && a.apply_bn && a.relu && a.B >= 512;   // two patches per workgroup
  hipLaunchKernelGGL((k_stage_fwd<T, CFG>), dim3(pipe ? (a.B + 1) / 2 : a.B, G), dim3(256), lds, st, a);
  DTA_CHECK_LAUNCH("k_stage_fwd");
  return 0;
}

// true when (C, Hc, Wc, pool, stencil/pool sizes) are those of one of the 11x11 network's three stages
static bool stage_net_cfg(const StageArgs& a) {
  int lvl = a.C == 32 ? 0 : a.C == 64 ? 1 : a.C == 128 ? 2 : -1;
  if (lvl < 0) return false;
  const int H[3] = {11, 11, 5}, P[3] = {0, 1, 1}, K[3] = {7, 5, 3}, CP[3] = {4, 2, 1};
  if (a.Hc != H[lvl] || a.Wc != H[lvl] || (a.pool != 0) != (P[lvl] != 0)) return false;
  for (int g = 0; g < MAXG; ++g)
    if (a.kind[g] == KIND_SPATIAL && (a.att_k[g] != K[lvl] || a.att_pool[g] != CP[lvl])) return false;
  return true;
}

// true when (C, Hc, Wc, pool) are those of one of the three stages of a spectral network on 24x24 crops (BASELINE
// configs[4]: year ensembles) and no group is a spatial network (the reference's spatial branch is 11x11-only)
static bool stage_net24_cfg(const StageArgs& a) {
  int lvl = a.C == 32 ? 0 : a.C == 64 ? 1 : a.C == 128 ? 2 : -1;
  if (lvl < 0) return false;
  const int H[3] = {24, 24, 12}, P[3] = {0, 1, 1};
  if (a.Hc != H[lvl] || a.Wc != H[lvl] || (a.pool != 0) != (P[lvl] != 0)) return false;
  for (int g = 0; g < MAXG; ++g)
    if (a.kind[g] != KIND_SPECTRAL) return false;      // (unused entries are zero = KIND_SPECTRAL)
  return true;
}

template <typename T> int launch_stage_fwd_lean(const StageArgs& a, int G, hipStream_t st);
bool stage_fwd_is_lean(const StageArgs& a) {
  if (!((a.lean & 1) && a.apply_bn && a.relu && !a.a_nchw)) return false;
  if (stage_net_cfg(a)) return a.y_fmt == FMT_F32 || a.y_fmt == FMT_F16;
  return stage_net24_cfg(a) && a.y_fmt == FMT_F16;      // (bf16 mode only: the configuration BASELINE names)
}

bool stage_fwd_will_be_lean(const StageArgs& a_in, int G) {
  StageArgs a = a_in;
  a.vslot = stage_vslot_for(a, G);
  return stage_fwd_is_lean(a);
}

template <typename T>
int launch_stage_fwd(const StageArgs& a_in, int G, hipStream_t st) {
  StageArgs a = a_in;
  a.vslot = stage_vslot_for(a, G);
  // the three stages of the 11x11 networks have lean register-resident forms (end of this file)
  if (stage_fwd_is_lean(a)) return launch_stage_fwd_lean<T>(a, G, st);
  if (a.a_compact) { dta_set_error("stage_fwd: halo-free tiles need the lean kernels"); return 1; }
  size_t lds = stage_lds_floats(a, false) * 4;
  if (lds > 160 * 1024) { dta_set_error("stage_fwd: %dx%dx%d patch needs %zu B of LDS", a.Hc, a.Wc, a.C, lds); return 1; }
  // the three stages of the 11x11 network are fully specialised (stencils unroll, no index divisions)
  const bool net = a.apply_bn && a.relu && stage_net_cfg(a);
  if (a.y_fmt == FMT_F16) {     // bf16 mode: conv outputs stored as IEEE half
    if (a.C != 32 && a.C != 64 && a.C != 128) { dta_set_error("stage_fwd: 16-bit conv outputs need 32/64/128 channels"); return 1; }
    if (net && a.C == 32) return launch_stage_fwd_c<T, StageCfg<32, 11, 11, 0, FMT_F16>>(a, G, lds, st);
    if (net && a.C == 64) return launch_stage_fwd_c<T, StageCfg<64, 11, 11, 1, FMT_F16>>(a, G, lds, st);
    if (net && a.C == 128) return launch_stage_fwd_c<T, StageCfg<128, 5, 5, 1, FMT_F16>>(a, G, lds, st);
    switch (a.C) {
      case 32: return launch_stage_fwd_c<T, StageCfg<32, 0, 0, 0, FMT_F16>>(a, G, lds, st);
      case 64: return launch_stage_fwd_c<T, StageCfg<64, 0, 0, 0, FMT_F16>>(a, G, lds, st);
      default: return launch_stage_fwd_c<T, StageCfg<128, 0, 0, 0, FMT_F16>>(a, G, lds, st);
    }
  }
  if (a.y_fmt != FMT_F32) { dta_set_error("stage_fwd: unsupported conv-output format %d", a.y_fmt); return 1; }
  if (net && a.C == 32) return launch_stage_fwd_c<T, StageCfg<32, 11, 11, 0>>(a, G, lds, st);
  if (net && a.C == 64) return launch_stage_fwd_c<T, StageCfg<64, 11, 11, 1>>(a, G, lds, st);
  if (net && a.C == 128) return launch_stage_fwd_c<T, StageCfg<128, 5, 5, 1>>(a, G, lds, st);
  switch (a.C) {
    case 32: return launch_stage_fwd_c<T, StageCfg<32, 0, 0, 0>>(a, G, lds, st);
    case 64: return launch_stage_fwd_c<T, StageCfg<64, 0, 0, 0>>(a, G, lds, st);
    case 128: return launch_stage_fwd_c<T, StageCfg<128, 0, 0, 0>>(a, G, lds, st);
  }
  return launch_stage_fwd_c<T, StageCfg<0, 0, 0, 0>>(a, G, lds, st);
}
template int launch_stage_fwd<float>(const StageArgs&, int, hipStream_t);
template int launch_stage_fwd<bf16_t>(const StageArgs&, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
// Backward of one stage for one patch.
// ------------------------------------------------------------------------------------------------
#ifdef DTA_TICKS
__device__ long long g_ticks[2][32];
#define TICK(i) do { if (threadIdx.x == 0 && blockIdx.x == 700) g_ticks[blockIdx.y][i] = clock64(); } while (0)
#else
#define TICK(i)
#endif
// Backward of one patch (b, g).  tiles_ready: Z, D and the saved attention state (v0..v2) are already in LDS.
template <typename CFG>
__device__ __forceinline__ void stage_bwd_patch(const StageBwdArgs& ba, const StageGeom& s, int b, int g, bool tiles_ready,
                                                float* Z, float* D, float* v0, float* v1, float* v2, float* v3,
                                                float* v4, float* v5, float* v6, float* scratch) {
  const StageArgs& a = ba.f;
  const bool pool = cfg_pool<CFG>(a);
  const int t = threadIdx.x, C = s.C, ld = s.ld;
  const int kind = a.kind[g];
  // padded maps d2 (v3) and d1 (v4).  With tiles_ready the caller has already cleared them BEFORE its barrier:
  // clearing here would race with the first writes into v3 below when no head gradient adds a barrier in between.
  if (kind == KIND_SPATIAL && !tiles_ready)
    for (int i = t; i < 2 * s.vslot; i += 256) v3[i] = 0.f;
  TICK(0);
  if (!tiles_ready) {
  // D = incoming gradient wrt the gated map (issued first: its loads fly together with the activations')
  const float* da_pre = nullptr;
  if (ba.da) {
    const float* da = ba.da + (size_t)g * ba.da_gs + (size_t)b * s.HWz * C;
    if (CFG::fixed) da_pre = da;
    else for (int i = t; i < s.HWz * C; i += 256) { int p = i / C, c = i - p * C; D[p * ld + c] = da[i]; }
  } else if (ba.da_nchw) {
    const float* da = ba.da_nchw + (size_t)g * ba.da_nchw_gs + (size_t)b * C * s.HWz;
    for (int i = t; i < s.HWz * C; i += 256) { int c = i / s.HWz, p = i - c * s.HWz; D[p * ld + c] = da[i]; }
  } else {
    for (int i = t; i < s.HWz * ld; i += 256) D[i] = 0.f;
  }
  TICK(1);
  stage_forward<CFG>(a, s, g, b, kind, Z, v0, v1, v2, scratch, a.attsave != nullptr, da_pre, D);
  }
  TICK(2);
  const float* df = ba.dfeat ? ba.dfeat + (size_t)g * ba.dfeat_gs + (size_t)b * a.F[g] : nullptr;
  float* vec = ba.vec ? ba.vec + (size_t)g * ba.vec_gs + (size_t)b * ba.vec_ld : nullptr;

  if (kind == KIND_SPECTRAL) {
    const float* a1 = a.att[g].p[4]; const float* a2 = a.att[g].p[5];  // dense [o][i] forms
    const float inv = 1.f / (float)s.HWz;
    if (df) {
      for (int i = t; i < s.HWz * C; i += 256) { int p = i / C, c = i - p * C; D[p * ld + c] += df[c] * inv; }
      __syncthreads();
    }
    // dg -> d2 (v3), dh -> d1 (v4), dp (v5)
    colreduce(C, s.HWz, scratch, v3, 1.f, [&](int c, int i) { return D[i * ld + c] * Z[i * ld + c]; });
    if (t < C) v3[t] = v3[t] * v2[t] * (1.f - v2[t]);
    __syncthreads();
    colreduce(C, C, scratch, v4, 1.f, [&](int i, int o) { return a2[o * C + i] * v3[o]; });
    if (t < C) v4[t] = v1[t] > 0.f ? v4[t] : 0.f;
    __syncthreads();
    colreduce(C, C, scratch, v5, inv, [&](int i, int o) { return a1[o * C + i] * v4[o]; });
    for (int i = t; i < s.HWz * C; i += 256) {
      int p = i / C, c = i - p * C;
      D[p * ld + c] = D[p * ld + c] * v2[c] + v5[c];
    }
    if (vec && t < C) { vec[t] = v3[t]; vec[C + t] = v1[t]; vec[2 * C + t] = v4[t]; vec[3 * C + t] = v0[t]; }
    __syncthreads();
  } else if (kind == KIND_SPATIAL) {
    const float* wc = a.att[g].p[0];
    const float* k1 = a.att[g].p[2];
    const float* k2 = a.att[g].p[4];
    const int k = cfg_att_k<CFG>(a, g), r = k / 2, kk = k * k, Wp = s.Wz + 2 * r;
    if (df) {
      const int ps = cfg_att_pool<CFG>(a, g), hp = s.Hz / ps, wp = s.Wz / ps;
      for (int i = t; i < C * hp * wp; i += 256) {
        int c = i / (hp * wp), rem = i - c * hp * wp, ph = rem / wp, pw = rem - ph * wp;
        float m = -3.4e38f; int arg = 0;
        for (int dy = 0; dy < ps; ++dy)
          for (int dx = 0; dx < ps; ++dx) {
            int p = (ph * ps + dy) * s.Wz + pw * ps + dx;
            float v = Z[p * ld + c] * v2[p];
            if (v > m) { m = v; arg = p; }
          }
        D[arg * ld + c] += df[i];
      }
      __syncthreads();
    }
    // ds -> d2 (v3, padded map)
    pixreduce(s.HWz, C, [&](int p, int c) { return D[p * ld + c] * Z[p * ld + c]; },
              [&](int p, float acc) {
                int h = p / s.Wz, w = p - h * s.Wz;
                v3[(h + r) * Wp + w + r] = acc * v2[p] * (1.f - v2[p]);
              });
    __syncthreads();
    TICK(6);
    // dt1 = transposed stencil of d2 with k2, masked by t1 > 0 -> d1 (v4, padded map)
    for (int p = t; p < s.HWz; p += 256) {
      int h = p / s.Wz, w = p - h * s.Wz, pi = (h + r) * Wp + w + r;
      float acc = stencil_at(v3, k2, k, Wp, h, w, true);
      v4[pi] = v1[pi] > 0.f ? acc : 0.f;
    }
    __syncthreads();
    // dm = transposed stencil of d1 with k1, masked by m > 0 -> dm0 (v5, per pixel)
    for (int p = t; p < s.HWz; p += 256) {
      int h = p / s.Wz, w = p - h * s.Wz, pi = (h + r) * Wp + w + r;
      float acc = stencil_at(v4, k1, k, Wp, h, w, true);
      v5[p] = v0[pi] > 0.f ? acc : 0.f;
    }
    __syncthreads();
    TICK(7);
    if (vec) {
      // [dwc (C) | dbc | dK1 (kk) | db1 | dK2 (kk) | db2]
      colreduce(C, s.HWz, scratch, v6, 1.f, [&](int c, int p) { return v5[p] * Z[p * ld + c]; });
      if (t < C) vec[t] = v6[t];
      // bias gradients: dbc = sum dm, db1 = sum d1, db2 = sum d2 (padded maps: borders are zero)
      {
        const int npad = (s.Hz + 2 * r) * Wp;
        float sA = 0.f, sB = 0.f, sC = 0.f;
        for (int p = t; p < s.HWz; p += 256) sA += v5[p];
        for (int q = t; q < npad; q += 256) { sB += v4[q]; sC += v3[q]; }
        sA = wave_sum(sA); sB = wave_sum(sB); sC = wave_sum(sC);
        if ((t & 63) == 0) { scratch[(t >> 6) * 3] = sA; scratch[(t >> 6) * 3 + 1] = sB; scratch[(t >> 6) * 3 + 2] = sC; }
        __syncthreads();
        if (t < 3) {
          const float tot = scratch[t] + scratch[3 + t] + scratch[6 + t] + scratch[9 + t];
          vec[t == 0 ? C : (t == 1 ? C + 1 + kk : C + 2 + 2 * kk)] = tot;
        }
      }
      // stencil-weight gradients: one (kernel, tap) task per lane pair, the pair splits the rows
      for (int task = t >> 1; task < 2 * kk; task += 128) {
        const int half = t & 1;
        const bool first = task < kk;
        const int j = first ? task : task - kk;
        const int ky = j / k, kx = j - ky * k;
        const float* src = first ? v0 : v1;   // conv input map (m for K1, t1 for K2), padded
        const float* dd = first ? v4 : v3;    // grad wrt the conv output, padded
        float acc = 0.f;
        for (int h = half; h < s.Hz; h += 2) {
          const float* sr = src + (h + ky) * Wp + kx;
          const float* dr = dd + (h + r) * Wp + r;
          for (int w = 0; w < s.Wz; ++w) acc += sr[w] * dr[w];
        }
        acc += __shfl_xor(acc, 1);
        if (!half) vec[first ? C + 1 + j : C + 2 + kk + j] = acc;
      }
    }
    TICK(8);
    for (int i = t; i < s.HWz * C; i += 256) {
      int p = i / C, c = i - p * C;
      D[p * ld + c] = D[p * ld + c] * v2[p] + v5[p] * wc[c];
    }
    __syncthreads();
  } else {
    if (df && a.F[g] > 0) {
      for (int i = t; i < C * s.HWz; i += 256) { int c = i / s.HWz, p = i - c * s.HWz; D[p * ld + c] += df[i]; }
      __syncthreads();
    }
  }

  TICK(3);
  // pool + ReLU backward, write dv and the per-patch BatchNorm-backward partial sums
  // the gradient map leaves in fp32 or (bf16 mode) in bf16, like the lean kernels' (half the bytes here and in the
  // BatchNorm-backward apply that reads it); element index dvi, stores through st_dv
  const int dvf = ba.dv_fmt;
  const size_t dvi = (size_t)g * ba.dv_gs + (size_t)b * s.HWc * C;
  // (bf16: lanes t, t ^ 1 own channels c, c + 1 of the same pixel -- the even lane stores the packed pair as one dword;
  //  2-byte stores per lane were measured 19 % slower for the whole kernel)
  auto st_dv = [&](size_t i_, float v_) {
    if (dvf == FMT_F32) { const_cast<float*>(ba.dv)[dvi + i_] = v_; return; }
    const float nb_ = lane_xor1(v_);
    if (!(t & 1)) *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(const_cast<float*>(ba.dv)) + dvi + i_) = pack2_fmt(v_, nb_, FMT_BF16);
  };
  const size_t ybase = (size_t)g * a.y_gs + (size_t)b * s.HWc * a.y_rs;
  const float* coef = a.coef ? a.coef + (size_t)g * a.coef_gs : nullptr;
  const int c = t % C, sl = t / C, nsl = 256 / C;
  float s1 = 0.f, s2 = 0.f;
  if (sl < nsl) {
    // xhat is recovered from the activation instead of re-reading the conv output: a non-zero gradient only
    // survives where the ReLU output r is positive, and there r = gamma * xhat + beta  (gamma = scale / rstd)
    const float mean = a.apply_bn ? coef[c * 4 + 2] : 0.f, rstd = a.apply_bn ? coef[c * 4 + 3] : 0.f;
    const float scale = a.apply_bn ? coef[c * 4 + 0] : 1.f, shift = a.apply_bn ? coef[c * 4 + 1] : 0.f;
    const float gam = scale / (rstd != 0.f ? rstd : 1.f), bet = shift + mean * scale;
    const bool recover = a.apply_bn && a.relu && gam != 0.f;
    const float inv_gam = recover ? 1.f / gam : 0.f;
    if (!pool) {
      for (int p = sl; p < s.HWc; p += nsl) {
        float d = D[p * ld + c], r = Z[p * ld + c];
        if (a.relu && r <= 0.f) d = 0.f;
        st_dv((size_t)p * C + c, d);
        if (a.apply_bn) {
          float xh = recover ? (r - bet) * inv_gam : (ld_fmt(a.y, ybase + (size_t)p * a.y_rs + c, CFG::YF) - mean) * rstd;
          s1 += d; s2 += d * xh;
        }
      }
    } else {
      // one pooled element per iteration: re-read its 2x2 window, route the gradient to the first maximum
      unsigned char* fpos = reinterpret_cast<unsigned char*>(const_cast<float*>(ba.dv)) + (dvi + (size_t)s.HWz * C) * fmt_bytes(dvf);   // compact form: positions
      for (int pz = sl; pz < s.HWz; pz += nsl) {
        int hz = pz / s.Wz, wz = pz - hz * s.Wz;
        const int p00 = (2 * hz) * s.Wc + 2 * wz;
        const int po[4] = {p00, p00 + 1, p00 + s.Wc, p00 + s.Wc + 1};
        float yv[4], m = -3.4e38f;
        int first = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          yv[k] = ld_fmt(a.y, ybase + (size_t)po[k] * a.y_rs + c, CFG::YF);
          float v = yv[k] * scale + shift;
          if (v > m) { m = v; first = k; }
        }
        float d = D[pz * ld + c];
        if (a.relu && m <= 0.f) d = 0.f;
        if (ba.dv_compact) { st_dv((size_t)pz * C + c, d); fpos[(size_t)pz * C + c] = (unsigned char)first; }
        else {
#pragma unroll
          for (int k = 0; k < 4; ++k) st_dv((size_t)po[k] * C + c, (k == first) ? d : 0.f);
        }
        if (a.apply_bn) { s1 += d; s2 += d * (yv[first] - mean) * rstd; }
      }
      // conv-resolution positions the floor pooling dropped get no gradient
      if (!ba.dv_compact)
        for (int p = sl; p < s.HWc; p += nsl) {
          int h = p / s.Wc, w = p - h * s.Wc;
          if ((h >> 1) >= s.Hz || (w >> 1) >= s.Wz) st_dv((size_t)p * C + c, 0.f);
        }
    }
  }
  TICK(4);
  if (ba.bnpart) {
    __syncthreads();
    scratch[t] = s1; scratch[256 + t] = s2;
    __syncthreads();
    if (t < C) {
      float q1 = 0.f, q2 = 0.f;
      for (int k2 = 0; k2 < nsl; ++k2) { q1 += scratch[k2 * C + t]; q2 += scratch[256 + k2 * C + t]; }
      float* o = ba.bnpart + (size_t)g * ba.bnpart_gs + ((size_t)b * C + t) * 2;
      o[0] = q1; o[1] = q2;
    }
  }
  TICK(5);
}

template <typename CFG>
__global__ __launch_bounds__(256, (CFG::fixed && CFG::P == 0) ? 4 : 1) void k_stage_bwd(StageBwdArgs ba) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const StageArgs& a = ba.f;
  const StageGeom s = stage_geom<CFG>(a);
  const int g = blockIdx.y, t = threadIdx.x, C = s.C, ld = s.ld;
  const int kind = a.kind[g];
  float* Z = sm;
  float* D = Z + (size_t)s.HWz * ld;
  float* v0 = D + (size_t)s.HWz * ld;
  float* v1 = v0 + s.vslot; float* v2 = v1 + s.vslot; float* v3 = v2 + s.vslot;
  float* v4 = v3 + s.vslot; float* v5 = v4 + s.vslot; float* v6 = v5 + (s.HWz > C ? s.HWz : C);
  float* scratch = v6 + C;   // (the slot plan keeps the 11x11x32 stage at 4 workgroups per CU)
  // Un-pooled network stage (11x11x32) launched with half a grid: every workgroup owns TWO patches (b, b + gridDim.x).
  // All workgroups are then resident at once, and the second patch's conv output and incoming gradient are fetched
  // into registers while the first patch is processed: its HBM latency (every workgroup bursting at once) is hidden.
  constexpr bool PIPE = CFG::fixed && CFG::P == 0;
  if (PIPE && (int)gridDim.x < a.B) {
    constexpr int NQ = PIPE ? (CFG::H * CFG::W * CFG::C + 255) / 256 : 1;
    constexpr int CQ = PIPE ? CFG::C : 1, NEL = PIPE ? CFG::H * CFG::W * CFG::C : 0;
    float ry[NQ], rq[NQ];
    const float* coef = a.coef + (size_t)g * a.coef_gs;
    const float psc = coef[(t % CQ) * 4 + 0], psh = coef[(t % CQ) * 4 + 1];
#define DTA_STAGE_ISSUE(b_)                                                                          \
    {                                                                                                \
      const size_t y_ = (size_t)g * a.y_gs + (size_t)(b_) * s.HWc * a.y_rs;                          \
      const float* d_ = ba.da + (size_t)g * ba.da_gs + (size_t)(b_) * s.HWz * C;                     \
      _Pragma("unroll") for (int u = 0; u < NQ; ++u) {                                               \
        const int i = t + u * 256;                                                                   \
        if (i < NEL) { ry[u] = ld_fmt<false>(a.y, y_ + (size_t)(i / CQ) * a.y_rs + (i % CQ), CFG::YF); rq[u] = __builtin_nontemporal_load(d_ + i); } \
      }                                                                                              \
    }
#define DTA_STAGE_LAND(b_)                                                                           \
    {                                                                                                \
      if (kind == KIND_SPATIAL)                                                                      \
        for (int i = t; i < 2 * s.vslot; i += 256) v3[i] = 0.f;                                      \
      _Pragma("unroll") for (int u = 0; u < NQ; ++u) {                                               \
        const int i = t + u * 256;                                                                   \
        if (i < NEL) {                                                                               \
          const int p = i / CQ, c = i % CQ;                                                          \
          Z[p * ld + c] = relu_nan(ry[u] * psc + psh);                                             \
          D[p * ld + c] = rq[u];                                                                     \
        }                                                                                            \
      }                                                                                              \
      const float* src = a.attsave + ((size_t)g * a.B + (b_)) * a.attsave_ld;                        \
      if (kind == KIND_SPECTRAL) {                                                                   \
        for (int i = t; i < 3 * C; i += 256) { const int k = i / C; v0[k * s.vslot + (i - k * C)] = __builtin_nontemporal_load(src + i); } \
      } else if (kind == KIND_SPATIAL) {                                                             \
        for (int i = t; i < 3 * s.vslot; i += 256) v0[i] = __builtin_nontemporal_load(src + i);      \
      }                                                                                              \
      __syncthreads();                                                                               \
    }
    const int b0 = blockIdx.x, b1 = blockIdx.x + gridDim.x;
    DTA_STAGE_ISSUE(b0)
    DTA_STAGE_LAND(b0)
    if (b1 < a.B) DTA_STAGE_ISSUE(b1)
    stage_bwd_patch<CFG>(ba, s, b0, g, true, Z, D, v0, v1, v2, v3, v4, v5, v6, scratch);
    if (b1 < a.B) {
      __syncthreads();   // the second patch reuses the LDS tiles
      DTA_STAGE_LAND(b1)
      stage_bwd_patch<CFG>(ba, s, b1, g, true, Z, D, v0, v1, v2, v3, v4, v5, v6, scratch);
    }
#undef DTA_STAGE_LAND
#undef DTA_STAGE_ISSUE
    return;
  }
  stage_bwd_patch<CFG>(ba, s, blockIdx.x, g, false, Z, D, v0, v1, v2, v3, v4, v5, v6, scratch);
}
#ifdef DTA_TICKS
extern "C" int dta_debug_ticks(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ticks), sizeof(long long) * 64); }
#endif

template <typename CFG>
static int launch_stage_bwd_c(const StageBwdArgs& a, int G, size_t lds, hipStream_t st) {
  static DevOnce attr_once;      // (function attributes are per device)
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)k_stage_bwd<CFG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  // (11x11x32 network stage with saved attention state: two patches per workgroup, see the kernel)
  const bool pipe = CFG::fixed && CFG::P == 0 && a.da && a.f.attsave && a.f.apply_bn && a.f.relu && a.f.B >= 512;
  hipLaunchKernelGGL((k_stage_bwd<CFG>), dim3(pipe ? (a.f.B + 1) / 2 : a.f.B, G), dim3(256), lds, st, a);
  DTA_CHECK_LAUNCH("k_stage_bwd");
  return 0;
}

int launch_stage_bwd_lean(const StageBwdArgs& a, int G, hipStream_t st);
// lean form (end of this file): needs the forward's saved attention state, the [B][HWz][C] gradient layout and, for
// spatial groups with a classifier gradient, the un-pooled class pool of the last stage
bool stage_bwd_is_lean(const StageBwdArgs& a, int G) {
  const int lbit = a.f.C == 32 ? 2 : (a.f.C == 64 ? 4 : 8);
  const bool net24 = stage_net24_cfg(a.f) && a.f.y_fmt == FMT_F16;      // 24x24 spectral networks, bf16 mode
  bool ok = (a.f.lean & lbit) && a.f.apply_bn && a.f.relu && (stage_net_cfg(a.f) || net24) && a.f.attsave && !a.da_nchw && a.dv &&
            (a.f.y_fmt == FMT_F32 || a.f.y_fmt == FMT_F16);
  for (int g = 0; g < G; ++g)
    if (a.f.kind[g] == KIND_SPATIAL && a.dfeat && a.f.C != 128) ok = false;
  return ok;
}

int launch_stage_bwd(const StageBwdArgs& a_in, int G, hipStream_t st) {
  StageBwdArgs a = a_in;
  a.f.vslot = stage_vslot_for(a.f, G);
  if (stage_bwd_is_lean(a, G)) return launch_stage_bwd_lean(a, G, st);
  if (a.da_fmt != FMT_F32) { dta_set_error("stage_bwd: a 16-bit incoming gradient map needs the lean kernels"); return 1; }
  if (a.dv_fmt != FMT_F32 && a.dv_fmt != FMT_BF16) { dta_set_error("stage_bwd: unsupported gradient-map format %d", a.dv_fmt); return 1; }
  size_t lds = stage_lds_floats(a.f, true) * 4;
  if (lds > 160 * 1024) { dta_set_error("stage_bwd: %dx%dx%d patch needs %zu B of LDS", a.f.Hc, a.f.Wc, a.f.C, lds); return 1; }
  const bool net = a.f.apply_bn && a.f.relu && stage_net_cfg(a.f);
  if (a.f.y_fmt == FMT_F16) {
    if (a.f.C != 32 && a.f.C != 64 && a.f.C != 128) { dta_set_error("stage_bwd: 16-bit conv outputs need 32/64/128 channels"); return 1; }
    if (net && a.f.C == 32) return launch_stage_bwd_c<StageCfg<32, 11, 11, 0, FMT_F16>>(a, G, lds, st);
    if (net && a.f.C == 64) return launch_stage_bwd_c<StageCfg<64, 11, 11, 1, FMT_F16>>(a, G, lds, st);
    if (net && a.f.C == 128) return launch_stage_bwd_c<StageCfg<128, 5, 5, 1, FMT_F16>>(a, G, lds, st);
    switch (a.f.C) {
      case 32: return launch_stage_bwd_c<StageCfg<32, 0, 0, 0, FMT_F16>>(a, G, lds, st);
      case 64: return launch_stage_bwd_c<StageCfg<64, 0, 0, 0, FMT_F16>>(a, G, lds, st);
      default: return launch_stage_bwd_c<StageCfg<128, 0, 0, 0, FMT_F16>>(a, G, lds, st);
    }
  }
  if (a.f.y_fmt != FMT_F32) { dta_set_error("stage_bwd: unsupported conv-output format %d", a.f.y_fmt); return 1; }
  if (net && a.f.C == 32) return launch_stage_bwd_c<StageCfg<32, 11, 11, 0>>(a, G, lds, st);
  if (net && a.f.C == 64) return launch_stage_bwd_c<StageCfg<64, 11, 11, 1>>(a, G, lds, st);
  if (net && a.f.C == 128) return launch_stage_bwd_c<StageCfg<128, 5, 5, 1>>(a, G, lds, st);
  switch (a.f.C) {
    case 32: return launch_stage_bwd_c<StageCfg<32, 0, 0, 0>>(a, G, lds, st);
    case 64: return launch_stage_bwd_c<StageCfg<64, 0, 0, 0>>(a, G, lds, st);
    case 128: return launch_stage_bwd_c<StageCfg<128, 0, 0, 0>>(a, G, lds, st);
  }
  return launch_stage_bwd_c<StageCfg<0, 0, 0, 0>>(a, G, lds, st);
}

// ------------------------------------------------------------------------------------------------
// BatchNorm backward: reduce per-patch partials -> dgamma, dbeta and the apply coefficients.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bn_bwd_finalize_block(const BnBwdFinalizeArgs& a, int bx, int g, float (*sc)[16]) {
  // block = 8 channels x 128 batch slices (see colsum8)
  const int t = threadIdx.x, C = a.C, c0 = bx * 8;
  const float* part = a.bnpart + (size_t)g * a.bnpart_gs + (size_t)c0 * 2;
  const int c = c0 + (t >> 1);
  const float* coef = a.coef + (size_t)g * a.coef_gs;
  // (the finishing threads' two parameter loads go out with the partials, not behind the reduction's barriers)
  float pgam = 0.f, prstd = 0.f;
  if (t < 16 && !(t & 1) && c < C) { pgam = a.gamma[g][c]; prstd = coef[c * 4 + 3]; }
  const double v = colsum8<2>(part, (size_t)C * 2, a.B, min(8, C - c0), sc);
  // thread 2j holds sum(dv) of channel c0 + j, thread 2j + 1 its sum(dv * xhat): the even thread finishes the channel
  const double d2 = __shfl_down(v, 1);
  if (t < 16 && !(t & 1) && c < C) {
    const double d1 = v;
    float* bc = a.bcoef + (size_t)g * a.bcoef_gs;
    float A = pgam * prstd;
    double n = (double)a.B * a.HW;
    bc[c * 4 + 0] = A;
    bc[c * 4 + 1] = a.training ? (float)(d1 / n) : 0.f;
    bc[c * 4 + 2] = a.training ? (float)(d2 / n) : 0.f;
    bc[c * 4 + 3] = 0.f;
    if (a.dbeta[g]) a.dbeta[g][c] = (float)d1;
    if (a.dgamma[g]) a.dgamma[g][c] = (float)d2;
    // conv bias feeds BN directly: with batch statistics its gradient is exactly zero
    if (a.dconvbias[g]) a.dconvbias[g][c] = a.training ? 0.f : A * (float)d1;
  }
}

__global__ __launch_bounds__(1024) void k_bn_bwd_finalize(BnBwdFinalizeArgs a) {
  __shared__ float sc[16][16];
  bn_bwd_finalize_block(a, blockIdx.x, blockIdx.y, sc);   // grid = (C/8, G)
}

int launch_bn_bwd_finalize(const BnBwdFinalizeArgs& a, int G, hipStream_t st) {
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((a.C + 7) / 8, G), dim3(1024), 0, st, a);
  DTA_CHECK_LAUNCH("k_bn_bwd_finalize");
  return 0;
}

// blocks [0, nbn): BatchNorm finalize (block -> (channel tile, group)); then the column-sum jobs
__global__ __launch_bounds__(1024) void k_bn_bwd_finalize_colsum(BnBwdFinalizeArgs a, int nbx, int nbn, ColsumPair cp) {
  __shared__ float sc[16][16];
  int bx = blockIdx.x;
  if (bx < nbn) { bn_bwd_finalize_block(a, bx % nbx, bx / nbx, sc); return; }
  bx -= nbn;
  if (bx < cp.nblk[0]) { colsum_scatter_block(cp.cs[0], bx, sc); return; }
  colsum_scatter_block(cp.cs[1], bx - cp.nblk[0], sc);
}

int launch_bn_bwd_finalize_colsum(const BnBwdFinalizeArgs& a, int G, const ColsumArgs* cs, int ncs, hipStream_t st) {
  if (ncs <= 0) return launch_bn_bwd_finalize(a, G, st);
  if (ncs > 2) { dta_set_error("bn_bwd_finalize_colsum: at most two column-sum jobs"); return 1; }
  ColsumPair cp = {};
  int extra = 0;
  for (int i = 0; i < ncs; ++i) { cp.cs[i] = cs[i]; cp.nblk[i] = colsum_nblocks(cs[i]); extra += cp.nblk[i]; }
  const int nbx = (a.C + 7) / 8, nbn = nbx * G;
  hipLaunchKernelGGL(k_bn_bwd_finalize_colsum, dim3(nbn + extra), dim3(1024), 0, st, a, nbx, nbn, cp);
  DTA_CHECK_LAUNCH("k_bn_bwd_finalize_colsum");
  return 0;
}

// Apply coefficients of channel cc of group g when no finalize launch ran: the FAN_R rows of batch sums the stage-backward
// launch left are added in a fixed order (every workgroup gets the same bits); `writer` also stores the parameter gradients.
__device__ __forceinline__ void bn_bwd_fan_coef(const BnBwdApplyArgs& a, const BnBwdFanArgs& fa, int g, int cc, bool writer, float& A, float& Bc,
                                                float& Cc) {
  const double* f = fa.fan + ((size_t)g * FAN_R * a.C + cc) * 2;
  double d1 = 0, d2 = 0;
#pragma unroll
  for (int q = 0; q < FAN_R; ++q) { d1 += f[(size_t)q * a.C * 2]; d2 += f[(size_t)q * a.C * 2 + 1]; }
  const double n = (double)a.B * a.H * a.W;
  A = fa.gamma[g][cc] * a.coef[(size_t)g * a.coef_gs + cc * 4 + 3];
  Bc = fa.training ? (float)(d1 / n) : 0.f;
  Cc = fa.training ? (float)(d2 / n) : 0.f;
  if (writer) {
    if (fa.dbeta[g]) fa.dbeta[g][cc] = (float)d1;
    if (fa.dgamma[g]) fa.dgamma[g][cc] = (float)d2;
    // conv bias feeds BN directly: with batch statistics its gradient is exactly zero
    if (fa.dconvbias[g]) fa.dconvbias[g][cc] = fa.training ? 0.f : A * (float)d1;
  }
}

template <typename T, int YF, int DVF>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(BnBwdApplyArgs a) {
  // per channel: dy = k0 * dv + k1 * y + k2  (folded from A*(dv - Bc - ((y-mean)*rstd)*Cc)), coefficients in LDS
  __shared__ float sk[3][128];
  const int b = blockIdx.x, g = blockIdx.y, t = threadIdx.x, C = a.C;
  const int W2 = a.W + 2, Q = (a.H + 2) * W2, HW = a.H * a.W, nch = C / 16;
  const size_t dvbase = (size_t)g * a.dv_gs + (size_t)b * HW * C;      // element index (dv may be bf16)
  const size_t ybase = (size_t)g * a.y_gs + (size_t)b * HW * a.y_rs;   // element index (the conv output may be 16-bit)
  const float* coef = a.coef + (size_t)g * a.coef_gs;
  const float* bc = a.bcoef + (size_t)g * a.bcoef_gs;
  for (int c = t; c < C; c += 256) {
    float A, Bc, Cc;
    const float mean = coef[c * 4 + 2], rstd = coef[c * 4 + 3];
    A = bc[c * 4 + 0]; Bc = bc[c * 4 + 1]; Cc = bc[c * 4 + 2];
    sk[0][c] = A;
    sk[1][c] = -A * Cc * rstd;
    sk[2][c] = A * (Cc * rstd * mean - Bc);
  }
  __syncthreads();
  T* dst = (T*)a.dy_tl + (size_t)g * a.dy_gs + ((size_t)b * a.dy_nc + a.dy_ch0) * Q * 16;
  constexpr int VW = TlVec<T>::VW, PARTS = 16 / VW;
  for (int i = t; i < nch * Q * PARTS; i += 256) {
    int ch = i / (Q * PARTS), rem = i - ch * Q * PARTS, q = rem / PARTS, part = rem % PARTS;
    int hh = q / W2 - 1, ww = q % W2 - 1;
    const bool in = hh >= 0 && hh < a.H && ww >= 0 && ww < a.W;
    float v[VW];
    if (in) {
      const int p = hh * a.W + ww;
      // the VW stored positions of this segment are one aligned group of VW channels (permuted inside for fp32)
      const int cb = ch * 16 + (tl_pos<T>(q, part * VW) & ~(VW - 1));
      float yv[VW], dvv[VW];
#pragma unroll
      for (int k = 0; k < VW; k += 4) {
        float t4[4];
        ld4_fmt(t4, a.y, ybase + (size_t)p * a.y_rs + cb + k, YF);
        yv[k] = t4[0]; yv[k + 1] = t4[1]; yv[k + 2] = t4[2]; yv[k + 3] = t4[3];
        float d4[4];
        ld4_fmt(d4, a.dv, dvbase + (size_t)p * C + cb + k, DVF);
        dvv[k] = d4[0]; dvv[k + 1] = d4[1]; dvv[k + 2] = d4[2]; dvv[k + 3] = d4[3];
      }
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        int c = ch * 16 + tl_pos<T>(q, part * VW + j);
        int k = c - cb;
        v[j] = sk[0][c] * dvv[k] + sk[1][c] * yv[k] + sk[2][c];
      }
    } else {
#pragma unroll
      for (int j = 0; j < VW; ++j) v[j] = 0.f;
    }
    tl_store_vec(dst + ((size_t)ch * Q + q) * 16, part, v);
  }
}

// Same transform, patch image assembled in LDS: pixel-major float4 reads of dv / y (fully coalesced), the tile-layout
// image of the patch (halo included) built in LDS, then one linear 16-byte-per-lane copy to HBM.
template <typename T, int YF, int DVF, bool FAN>
__device__ __forceinline__ void bn_bwd_apply_lds_body(const BnBwdApplyArgs& a, const BnBwdFanArgs* fa, const ColsumPair* cpp) {
  WGSTAMP(3);      // (the last launch of a step is the first stage's)
  extern __shared__ __attribute__((aligned(16))) char smem_apply[];
  if (FAN && (int)blockIdx.x >= a.B) {
    const ColsumPair& cp = *cpp;
    // extra workgroups (group row 0, slice 0 only): the batch column sums of the spatial-attention parameter gradients,
    // which used to ride in the finalize launch
    if (blockIdx.y != 0 || blockIdx.z != 0) return;
    int bx = blockIdx.x - a.B;
    float (*sc)[16] = reinterpret_cast<float (*)[16]>(smem_apply);
    if (bx < cp.nblk[0]) colsum_scatter_block<256>(cp.cs[0], bx, sc);
    else colsum_scatter_block<256>(cp.cs[1], bx - cp.nblk[0], sc);
    return;
  }
  // a workgroup = one patch x one slice of CS channels (blockIdx.z): small LDS images keep 8 workgroups on a CU
  const int b = blockIdx.x, g = blockIdx.y, t = threadIdx.x, C = a.C, CS = a.cslice, c0 = blockIdx.z * CS;
  const int W2 = a.W + 2, Q = (a.H + 2) * W2, HW = a.H * a.W, nch = CS / 16, C4 = CS / 4;
  const int c4sh = 31 - __clz(C4);                      // C is a power of two (32 / 64 / 128)
  float* sk = (float*)smem_apply;                       // [3][CS]
  int* lut = (int*)(smem_apply + 3 * CS * 4);            // [HW] pixel -> haloed row q | pooled element << 10 | window position << 20 | in-window << 22
  T* img = (T*)(smem_apply + ((3 * CS * 4 + HW * 4 + 15) & ~15)); // [nch][Q][16]
  const size_t dvbase = (size_t)g * a.dv_gs + (size_t)b * HW * C;      // element index of the patch (dv may be bf16)
  const size_t ybase = (size_t)g * a.y_gs + (size_t)b * HW * a.y_rs + c0;   // element index (the conv output may be 16-bit)
  const unsigned char* fpos = reinterpret_cast<const unsigned char*>(a.dv) +
                              (dvbase + (size_t)a.Hz * a.Wz * C) * (DVF == FMT_F32 ? 4 : 2) + c0;   // compact form only
  const float* coef = a.coef + (size_t)g * a.coef_gs;
  const float* bc = a.bcoef + (size_t)g * a.bcoef_gs;
  // 16-bit everything (the bf16 networks' lean path): a thread item is 8 channels of a pixel -- one 16-byte load of the
  // half conv output, one of the bf16 gradient, one 16-byte LDS store.  The loads of a thread's FIRST items are issued
  // here, before the coefficient / table prologue and its barrier, so that the two global round trips overlap (an 11x11
  // patch is a single round of items: the launch was one dependent chain coef -> barrier -> loads -> image -> store)
  constexpr bool V8 = sizeof(T) == 2 && YF != FMT_F32 && DVF != FMT_F32;
  constexpr int UB8 = 2;
  const int c8sh = c4sh - 1, total8 = HW << c8sh;
  auto lut_of = [&](int pix) -> int {
    const int hh = pix / a.W, ww = pix - hh * a.W, hz = hh >> 1, wz = ww >> 1;
    const int inw = (a.dv_compact && hz < a.Hz && wz < a.Wz) ? 1 : 0;
    return ((hh + 1) * W2 + ww + 1) | ((inw ? hz * a.Wz + wz : 0) << 10) | ((((hh & 1) << 1) | (ww & 1)) << 20) | (inw << 22);
  };
  u32x4 ry[UB8], rd[UB8];
  u32x2 rfb[UB8];
  int rl[UB8];
  auto issue8 = [&](int i0) {
#pragma unroll
    for (int u = 0; u < UB8; ++u) {
      const int i = i0 + u * 256;
      rl[u] = 0;
      if (i < total8) {
        const int pix = i >> c8sh, c8 = (i - (pix << c8sh)) * 8;
        ry[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(a.y) + ybase + (size_t)pix * a.y_rs + c8));   // last reader of the conv output
        const int l = lut_of(pix);
        rl[u] = l;
        if (!a.dv_compact) rd[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(a.dv) + dvbase + (size_t)pix * C + c0 + c8));
        else if (l >> 22) {
          const int pz = (l >> 10) & 1023;
          rd[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(a.dv) + dvbase + (size_t)pz * C + c0 + c8));
          rfb[u] = *(const u32x2*)(fpos + (size_t)pz * C + c8);
        }
      }
    }
  };
  if constexpr (V8) issue8(t);
  for (int c = t; c < CS; c += 256) {
    const int cc = c0 + c;
    float A, Bc, Cc;
    const float mean = coef[cc * 4 + 2], rstd = coef[cc * 4 + 3];
    if constexpr (FAN) bn_bwd_fan_coef(a, *fa, g, cc, b == 0, A, Bc, Cc);
    else { A = bc[cc * 4 + 0]; Bc = bc[cc * 4 + 1]; Cc = bc[cc * 4 + 2]; }
    sk[c] = A;
    sk[CS + c] = -A * Cc * rstd;
    sk[2 * CS + c] = A * (Cc * rstd * mean - Bc);
  }
  if constexpr (!V8)
    for (int pix = t; pix < HW; pix += 256) lut[pix] = lut_of(pix);
  // dy_compact: the image holds the HW pixels only (no halo rows to zero, 28 % fewer bytes out for an 11x11 patch)
  const int QI = a.dy_compact ? HW : Q;
  const int nvec = nch * QI * 16 * (int)sizeof(T) / 16;
  u32x4* img4 = (u32x4*)img;
  if (!a.dy_compact)
    for (int i = t; i < nvec; i += 256) img4[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  if constexpr (V8) {
    for (int i0 = t; i0 < total8; i0 += 256 * UB8) {
      if (i0 != t) issue8(i0);              // (later rounds of a larger map: loaded here; the first round is in flight)
#pragma unroll
      for (int u = 0; u < UB8; ++u) {
        const int i = i0 + u * 256;
        if (i < total8) {
          const int pix = i >> c8sh, c8 = (i - (pix << c8sh)) * 8;
          const int l = rl[u];
          float yv[8], dvv[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { yv[2 * e] = unpack_lo(ry[u][e], YF); yv[2 * e + 1] = unpack_hi(ry[u][e], YF); }
          if (!a.dv_compact) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { dvv[2 * e] = unpack_lo(rd[u][e], DVF); dvv[2 * e + 1] = unpack_hi(rd[u][e], DVF); }
          } else {
            // expand the pooled stage's compact gradient: the value lands on the window position the forward chose
#pragma unroll
            for (int j = 0; j < 8; ++j) dvv[j] = 0.f;
            if (l >> 22) {
              const int k = (l >> 20) & 3;
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const float dc = (j & 1) ? unpack_hi(rd[u][j >> 1], DVF) : unpack_lo(rd[u][j >> 1], DVF);
                dvv[j] = (int)((rfb[u][j >> 2] >> (8 * (j & 3))) & 0xFFu) == k ? dc : 0.f;
              }
            }
          }
          const int q = a.dy_compact ? pix : (l & 1023);
          u32x4 pk;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v0 = sk[c8 + 2 * j] * dvv[2 * j] + sk[CS + c8 + 2 * j] * yv[2 * j] + sk[2 * CS + c8 + 2 * j];
            const float v1 = sk[c8 + 2 * j + 1] * dvv[2 * j + 1] + sk[CS + c8 + 2 * j + 1] * yv[2 * j + 1] + sk[2 * CS + c8 + 2 * j + 1];
            pk[j] = (unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16);
          }
          *(u32x4*)(img + ((size_t)(c8 >> 4) * QI + q) * 16 + (c8 & 15)) = pk;
        }
      }
    }
  } else {
  const int total = HW * C4;
  constexpr int UB = 4;
  for (int i0 = t; i0 < total; i0 += 256 * UB) {
    float yv[UB][4];
    float dvv[UB][4];
    int lq[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int i = i0 + u * 256;
      if (i < total) {
        const int pix = i >> c4sh, c4 = (i - (pix << c4sh)) * 4;
        ld4_fmt<true>(yv[u], a.y, ybase + (size_t)pix * a.y_rs + c4, YF);   // last reader of the conv output
        const int l = lut[pix];
        lq[u] = l;
        if (!a.dv_compact) ld4_fmt<true>(dvv[u], a.dv, dvbase + (size_t)pix * C + c0 + c4, DVF);
        else {
          // expand the pooled stage's compact gradient: the value lands on the window position the forward chose
#pragma unroll
          for (int j = 0; j < 4; ++j) dvv[u][j] = 0.f;
          if (l >> 22) {
            const int pz = (l >> 10) & 1023, k = (l >> 20) & 3;
            float dc[4];
            ld4_fmt<true>(dc, a.dv, dvbase + (size_t)pz * C + c0 + c4, DVF);
            const unsigned fb = *(const unsigned*)(fpos + (size_t)pz * C + c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) dvv[u][j] = (int)((fb >> (8 * j)) & 0xFFu) == k ? dc[j] : 0.f;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int i = i0 + u * 256;
      if (i < total) {
        const int pix = i >> c4sh, c4 = (i - (pix << c4sh)) * 4;
        const int q = a.dy_compact ? pix : (lq[u] & 1023);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sk[c4 + j] * dvv[u][j] + sk[CS + c4 + j] * yv[u][j] + sk[2 * CS + c4 + j];
        T* row = img + ((size_t)(c4 >> 4) * QI + q) * 16;
        if constexpr (sizeof(T) == 2) {
          u32x2 pk = {(unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16)};
          *(u32x2*)(row + (c4 & 15)) = pk;
        } else {
          // stored position of channel c is (c & 15) ^ (q & 15): an aligned group of 4 stays an aligned group
          const int s = q & 15, grp = ((c4 & 15) ^ s) & ~3, sw = s & 3;
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = v[j ^ sw];
          *(f32x4*)((float*)row + grp) = o;
        }
      }
    }
  }
  }
  __syncthreads();
  u32x4* dst = (u32x4*)((T*)a.dy_tl + (size_t)g * a.dy_gs + ((size_t)b * a.dy_nc + a.dy_ch0 + c0 / 16) * QI * 16);
  for (int i = t; i < nvec; i += 256) dst[i] = img4[i];
}
template <typename T, int YF, int DVF>
__global__ __launch_bounds__(256) void k_bn_bwd_apply_lds(BnBwdApplyArgs a) {
  bn_bwd_apply_lds_body<T, YF, DVF, false>(a, nullptr, nullptr);
}
// DTA_FANIN experiment: coefficients from the stage launch's fan-in rows, column-sum jobs as extra workgroups
template <typename T, int YF, int DVF>
__global__ __launch_bounds__(256) void k_bn_bwd_apply_lds_fan(BnBwdApplyArgs a, BnBwdFanArgs fa, ColsumPair cp) {
  bn_bwd_apply_lds_body<T, YF, DVF, true>(a, &fa, &cp);
}

static size_t bn_bwd_apply_lds_bytes(int CS, int H, int W, size_t esz) {
  return ((3 * CS * 4 + H * W * 4 + 15) & ~15) + (size_t)(CS / 16) * (H + 2) * (W + 2) * 16 * esz;
}
// channels per workgroup of the LDS-image kernel: whole patch if its image is small, else 32-channel slices
static int bn_bwd_apply_cslice(int C, int H, int W, size_t esz) {
  return (C > 32 && bn_bwd_apply_lds_bytes(C, H, W, esz) > 20 * 1024) ? 32 : C;
}
bool bn_bwd_apply_uses_lds(int C, int H, int W, size_t esz) {
  return bn_bwd_apply_lds_bytes(bn_bwd_apply_cslice(C, H, W, esz), H, W, esz) <= 48 * 1024;
}

template <typename T>
int launch_bn_bwd_apply(const BnBwdApplyArgs& a_in, int G, hipStream_t st, const BnBwdFanArgs* fan, const ColsumArgs* cs, int ncs) {
  BnBwdApplyArgs a = a_in;
  ColsumPair cp = {};
  int extra = 0;
  for (int i = 0; i < ncs && i < 2; ++i) { cp.cs[i] = cs[i]; cp.nblk[i] = colsum_nblocks(cs[i]); extra += cp.nblk[i]; }
  a.cslice = bn_bwd_apply_cslice(a.C, a.H, a.W, sizeof(T));
  const size_t lds = bn_bwd_apply_lds_bytes(a.cslice, a.H, a.W, sizeof(T));
  if ((a.dv_compact || a.dy_compact) && lds > 48 * 1024) { dta_set_error("bn_bwd_apply: compact dv / dy need the LDS-image kernel"); return 1; }
  if (a.dy_compact && sizeof(T) != 2) { dta_set_error("bn_bwd_apply: halo-free output tiles are a bf16 layout"); return 1; }
  if (a.y_fmt != FMT_F32 && a.y_fmt != FMT_F16) { dta_set_error("bn_bwd_apply: unsupported conv-output format %d", a.y_fmt); return 1; }
  if (a.dv_fmt != FMT_F32 && a.dv_fmt != FMT_BF16) { dta_set_error("bn_bwd_apply: unsupported gradient-map format %d", a.dv_fmt); return 1; }
  const bool h = a.y_fmt == FMT_F16, d16 = a.dv_fmt == FMT_BF16;
  if (d16 && !h) { dta_set_error("bn_bwd_apply: bf16 gradient maps come with half conv outputs (bf16 mode)"); return 1; }
  if (fan && lds > 48 * 1024) { dta_set_error("bn_bwd_apply: the fan-in form needs the LDS-image kernel"); return 1; }
  if (lds <= 48 * 1024) {
    if (fan) {
      const dim3 grid(a.B + extra, G, a.C / a.cslice);
      if (d16) hipLaunchKernelGGL((k_bn_bwd_apply_lds_fan<T, FMT_F16, FMT_BF16>), grid, dim3(256), lds, st, a, *fan, cp);
      else if (h) hipLaunchKernelGGL((k_bn_bwd_apply_lds_fan<T, FMT_F16, FMT_F32>), grid, dim3(256), lds, st, a, *fan, cp);
      else hipLaunchKernelGGL((k_bn_bwd_apply_lds_fan<T, FMT_F32, FMT_F32>), grid, dim3(256), lds, st, a, *fan, cp);
      DTA_CHECK_LAUNCH("k_bn_bwd_apply_lds_fan");
      return 0;
    }
    const dim3 grid(a.B, G, a.C / a.cslice);
    if (d16) hipLaunchKernelGGL((k_bn_bwd_apply_lds<T, FMT_F16, FMT_BF16>), grid, dim3(256), lds, st, a);
    else if (h) hipLaunchKernelGGL((k_bn_bwd_apply_lds<T, FMT_F16, FMT_F32>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_bn_bwd_apply_lds<T, FMT_F32, FMT_F32>), grid, dim3(256), lds, st, a);
    DTA_CHECK_LAUNCH("k_bn_bwd_apply_lds");
    return 0;
  }
  if (d16) hipLaunchKernelGGL((k_bn_bwd_apply<T, FMT_F16, FMT_BF16>), dim3(a.B, G), dim3(256), 0, st, a);
  else if (h) hipLaunchKernelGGL((k_bn_bwd_apply<T, FMT_F16, FMT_F32>), dim3(a.B, G), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_bn_bwd_apply<T, FMT_F32, FMT_F32>), dim3(a.B, G), dim3(256), 0, st, a);
  DTA_CHECK_LAUNCH("k_bn_bwd_apply");
  return 0;
}
template int launch_bn_bwd_apply<float>(const BnBwdApplyArgs&, int, hipStream_t, const BnBwdFanArgs*, const ColsumArgs*, int);
template int launch_bn_bwd_apply<bf16_t>(const BnBwdApplyArgs&, int, hipStream_t, const BnBwdFanArgs*, const ColsumArgs*, int);

// ================================================================================================
// Lean forms of the three 11x11-network stages.
//
// The kernels above keep a patch as [pixel][channel] floats in LDS and run every step as a loop of scalar LDS / global
// accesses with per-element index arithmetic; on the bench size they are instruction-issue bound (SQ counters: the
// SIMDs issue 50-75 % of the time, >80 % of it address arithmetic).  Here a thread owns ITEMS (pixel, 8-channel octet):
// the conv output arrives as one 16-byte load per item (8 halves, or two for fp32), the gated map leaves as one
// 16-byte tile-row store per item, BatchNorm / ReLU / pooling / gating happen in registers, and LDS only carries what
// crosses threads: the patch for the per-channel sums, the attention vectors / single-channel maps, the reductions.
// Restates the same reference lines as above (Hang2020.py:24-31, :105-124, :149-168).
// ================================================================================================
template <int C_, int HC_, int WC_, int POOL_, int YF_>
struct LeanCfg {
  static constexpr int C = C_, HC = HC_, WC = WC_, POOL = POOL_, YF = YF_;
  static constexpr int HZ = POOL ? HC / 2 : HC, WZ = POOL ? WC / 2 : WC, NP = HZ * WZ, HWC = HC * WC;
  static constexpr int NO = C / 8;                        // octets per pixel
  static constexpr int ITEMS = NP * NO;                   // (pixel, octet) items per patch
  static constexpr int NT = ITEMS > 256 ? 512 : 256;      // threads per workgroup (one item per thread when they fit)
  static constexpr int PPW = ITEMS <= 64 ? 4 : 1;         // patches per workgroup
  static constexpr int TPP = NT / PPW;                    // threads per patch
  static constexpr int IPT = (ITEMS + TPP - 1) / TPP;     // items per thread
  static constexpr int K = C == 32 ? 7 : (C == 64 ? 5 : 3);     // spatial stencil size (Hang2020.py:77-85)
  static constexpr int PS = C == 32 ? 4 : (C == 64 ? 2 : 1);    // spatial class-pool size (:91-99)
  static constexpr int R = K / 2, WP = WZ + 2 * R, HPAD = HZ + 2 * R, NPAD = HPAD * WP;
  static constexpr int W2 = WZ + 2, QZ = (HZ + 2) * W2;   // haloed grid of the gated map's conv tiles
  static constexpr int NPART = NT / C;                    // matvec: input slices per output
  // waves per SIMD the backward is compiled for (register budget 512 / MINW): two 512-thread or three 256-thread
  // workgroups per CU; the 128-wide stage needs its 256 registers
  // (the fp32 step's 64-wide stage holds twice the raw bytes per item: compiled for three waves it spilled 15 registers)
  static constexpr int MINW = HC > 11 ? 2 : (C == 32 ? 4 : (C == 64 ? (YF == FMT_F32 ? 2 : 3) : 2));      // (24x24 crops: 3-5 items per thread)
  static constexpr bool PERSIST = C < 128 && HC <= 11;    // backward: persistent workgroups with next-batch prefetch
  // LDS floats per patch slot: the patch [NP][C], then vectors: spectral pooled|h|gate, spatial m|t1 (padded) | s
  static constexpr int VEC = 3 * C > 2 * NPAD + NP ? 3 * C : 2 * NPAD + NP;
  // the [NP][C] patch image in LDS serves the spatial class pool and the plain network's flatten only: the 24x24
  // configurations (spectral networks only) do without it (73 KiB for the 32-channel stage)
  static constexpr int ZF = HC > 11 ? 0 : NP * C;
  static constexpr int SLOT = ZF + VEC;
  static constexpr int RED = C >= 128 ? 4608 : (1536 > PPW * NT + 2 * C * PPW ? 1536 : PPW * NT + 2 * C * PPW);                        // reduction scratch shared by the workgroup (>= 4 C + 516, >= 256 + 2 C PPW)
  static constexpr int LDS_FWD = 2 * C + PPW * SLOT + RED + C + 2 * K * K;
};

// eight consecutive channels of one conv-output pixel -> floats
template <int YF, bool NT>
__device__ __forceinline__ void lean_ld8(float (&v)[8], const void* base, size_t i) {
  if (YF == FMT_F32) {
    const f32x4* p = (const f32x4*)((const float*)base + i);
    const f32x4 a = NT ? __builtin_nontemporal_load(p) : p[0], b = NT ? __builtin_nontemporal_load(p + 1) : p[1];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
  } else {
    const u32x4* p = (const u32x4*)((const unsigned short*)base + i);
    const u32x4 q = NT ? __builtin_nontemporal_load(p) : *p;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = unpack_lo(q[e], YF); v[2 * e + 1] = unpack_hi(q[e], YF); }
  }
}
// the same eight channels as the raw bytes a prefetch keeps in registers until the previous patch is done with
template <int YF> struct LeanRaw { u32x4 q; };
template <> struct LeanRaw<FMT_F32> { f32x4 a, b; };
template <int YF, bool NT>
__device__ __forceinline__ void lean_ld8_raw(LeanRaw<YF>& r, const void* base, size_t i) {
  if constexpr (YF == FMT_F32) {
    const f32x4* p = (const f32x4*)((const float*)base + i);
    r.a = NT ? __builtin_nontemporal_load(p) : p[0]; r.b = NT ? __builtin_nontemporal_load(p + 1) : p[1];
  } else {
    const u32x4* p = (const u32x4*)((const unsigned short*)base + i);
    r.q = NT ? __builtin_nontemporal_load(p) : *p;
  }
}
template <int YF>
__device__ __forceinline__ void lean_unpack8(float (&v)[8], const LeanRaw<YF>& r) {
  if constexpr (YF == FMT_F32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = r.a[e]; v[4 + e] = r.b[e]; }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = unpack_lo(r.q[e], YF); v[2 * e + 1] = unpack_hi(r.q[e], YF); }
  }
}
// one octet of a tile row: channels 8 * (o & 1) .. + 8 of chunk o / 2 at haloed-grid row q
__device__ __forceinline__ void lean_tl_store8(bf16_t* tile, int NCQ, int q, int o, const float (&v)[8]) {
  (void)NCQ;
  u32x4 u;
#pragma unroll
  for (int e = 0; e < 4; ++e) u[e] = pack2_fmt(v[2 * e], v[2 * e + 1], FMT_BF16);
  *reinterpret_cast<u32x4*>(tile + ((size_t)(o >> 1) * NCQ + q) * 16 + (o & 1) * 8) = u;
}
__device__ __forceinline__ void lean_tl_store8(float* tile, int NCQ, int q, int o, const float (&v)[8]) {
  // fp32 rows are XOR-swizzled (tl_pos<float>): channel c16 sits at c16 ^ (q & 15); an aligned octet stays one
  const int s = q & 15, half = (o & 1) ^ (s >> 3), sw = s & 7;
  float w[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) w[e] = v[e ^ sw];       // out[e ^ sw] = v[e]  <=>  out[e] = v[e ^ sw]
  float* row = tile + ((size_t)(o >> 1) * NCQ + q) * 16 + half * 8;
  *reinterpret_cast<f32x4*>(row) = f32x4{w[0], w[1], w[2], w[3]};
  *reinterpret_cast<f32x4*>(row + 4) = f32x4{w[4], w[5], w[6], w[7]};
}
// zero the halo rows of one patch's tiles (the workspace is borrowed: nothing in it can be assumed)
template <typename T, typename CFG>
__device__ __forceinline__ void lean_tl_halo(T* tile, int lt) {
  constexpr int NH = 2 * CFG::W2 + 2 * CFG::HZ, NCH = CFG::C / 16, VPR = 16 * (int)sizeof(T) / 16;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  for (int i = lt; i < NH * NCH * VPR; i += CFG::TPP) {
    const int v = i % VPR, r = (i / VPR) % NH, ch = i / (VPR * NH);
    int q;
    if (r < CFG::W2) q = r;
    else if (r < 2 * CFG::W2) q = (CFG::HZ + 1) * CFG::W2 + (r - CFG::W2);
    else { const int k = r - 2 * CFG::W2; q = ((k >> 1) + 1) * CFG::W2 + (k & 1) * (CFG::W2 - 1); }
    reinterpret_cast<u32x4*>(tile + ((size_t)ch * CFG::QZ + q) * 16)[v] = zero;
  }
}
// per-channel sum over the pixels of f(p, c) with the patch in LDS ([NP][C]); all 256 threads call it.
// red: RED floats; out[slot][c] for every patch slot.
template <typename CFG, typename F>
__device__ __forceinline__ void lean_colsum(float* red, float* out, int out_stride, float scale, F f) {
  constexpr int C = CFG::C, TPP = CFG::TPP, NP = CFG::NP;
  const int t = threadIdx.x, slot = t / TPP, lt = t % TPP;
  if (TPP >= C) {
    constexpr int RS = TPP / C > 0 ? TPP / C : 1;
    const int c = lt % C, sl = lt / C;
    float acc = 0.f;
#pragma unroll 4
    for (int p = sl; p < NP; p += RS) acc += f(slot, p, c);
    red[t] = acc;
    __syncthreads();
    if (lt < C) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < RS; ++k) s += red[slot * TPP + k * C + lt];
      out[slot * out_stride + lt] = s * scale;
    }
  } else {
    for (int c = lt; c < C; c += TPP) {
      float acc = 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) acc += f(slot, p, c);
      out[slot * out_stride + c] = acc * scale;
    }
  }
  __syncthreads();
}
// per-channel sums over the pixels straight from the owners' registers: part[v][e] is this thread's sum over its items
// of value set v, channel 8 o + e (o = lt % NO is the same for all items of a thread: TPP % NO == 0; threads without an
// item pass zeros).  The lanes of a wave that share o meet through DPP row rotations and two shuffles, the waves of a patch
// through `red` (NV * PPW * NW * C floats).  out[v][slot * out_stride + c] = scale * sum.  All threads call it.
template <int N>
__device__ __forceinline__ float lean_row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N /* row_ror:N */, 0xF, 0xF, false));
}
template <typename CFG, int NV>
__device__ __forceinline__ void lean_colsum_reg(float (&part)[NV][8], float* red, float* const (&out)[NV], int out_stride, float scale) {
  constexpr int C = CFG::C, NO = CFG::NO, TPP = CFG::TPP, PPW = CFG::PPW, NW = TPP / 64;
  static_assert(TPP % 64 == 0 && (NO == 4 || NO == 8 || NO == 16), "lean_colsum_reg: item ownership");
  static_assert(NV * PPW * NW * C <= CFG::RED, "lean_colsum_reg: scratch");
  const int t = threadIdx.x, slot = t / TPP, lt = t % TPP, lane = t & 63, wave = lt >> 6;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = part[v][e];
      if (NO == 4) x += lean_row_ror<4>(x);
      if (NO <= 8) x += lean_row_ror<8>(x);
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);
      part[v][e] = x;
    }
    if (lane < NO) {
      float* r = red + ((v * PPW + slot) * NW + wave) * C + lane * 8;
      *reinterpret_cast<f32x4*>(r) = f32x4{part[v][0], part[v][1], part[v][2], part[v][3]};
      *reinterpret_cast<f32x4*>(r + 4) = f32x4{part[v][4], part[v][5], part[v][6], part[v][7]};
    }
  }
  __syncthreads();
  for (int i = t; i < NV * PPW * C; i += CFG::NT) {
    const int v = i / (PPW * C), sl = (i / C) % PPW, c = i % C;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) acc += red[((v * PPW + sl) * NW + k) * C + c];
    out[v][sl * out_stride + c] = acc * scale;
  }
  __syncthreads();
}
// The scalar mat-vec in two halves, so that callers can fetch a thread's weight slice (C / NPART values) at kernel entry,
// long before the vector it multiplies exists: the weights then cost no latency on the dependent chain.
template <typename CFG>
__device__ __forceinline__ void lean_matvec_load(const float* W, float (&w)[CFG::C / CFG::NPART]) {
  constexpr int C = CFG::C, PER = C / CFG::NPART;
  const int t = threadIdx.x, o = t % C, part = t / C;
#pragma unroll
  for (int k = 0; k < PER; ++k) w[k] = W[(size_t)(part * PER + k) * C + o];
}
template <typename CFG, typename F>
__device__ __forceinline__ void lean_matvec_run(const float (&w)[CFG::C / CFG::NPART], const float* x, int x_stride, float* red, F fin) {
  constexpr int C = CFG::C, NPART = CFG::NPART, PPW = CFG::PPW, PER = C / NPART;
  const int t = threadIdx.x, o = t % C, part = t / C;
  float acc[PPW];
#pragma unroll
  for (int s = 0; s < PPW; ++s) acc[s] = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
#pragma unroll
    for (int s = 0; s < PPW; ++s) acc[s] += w[k] * x[s * x_stride + part * PER + k];
  }
#pragma unroll
  for (int s = 0; s < PPW; ++s) red[(s * NPART + part) * C + o] = acc[s];
  __syncthreads();
  for (int i = t; i < PPW * C; i += CFG::NT) {
    const int s = i / C, oo = i % C;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < NPART; ++k) v += red[(s * NPART + k) * C + oo];
    fin(s, oo, v);
  }
  __syncthreads();
}
// the same with the weight matrix staged in LDS (persistent workgroups: staged once, used for every patch batch)
template <typename CFG, typename F>
__device__ __forceinline__ void lean_matvec_lds(const float* WL, const float* x, int x_stride, float* red, F fin) {
  constexpr int C = CFG::C, PER = C / CFG::NPART;
  const int t = threadIdx.x, o = t % C, part = t / C;
  float w[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) w[k] = WL[(part * PER + k) * C + o];
  lean_matvec_run<CFG>(w, x, x_stride, red, fin);
}
// y[slot][o] = sum_i W[i * C + o] * x[slot][i] for every patch slot of the workgroup (W input-major, so lanes read
// consecutive o): thread (o, part) covers C / NPART inputs for all slots, the parts meet in LDS.  `fin(slot, o, sum)`.
template <typename CFG, typename F>
__device__ __forceinline__ void lean_matvec(const float* W, const float* x, int x_stride, float* red, F fin) {
  constexpr int C = CFG::C, NPART = CFG::NPART, PPW = CFG::PPW, PER = C / NPART;
  if constexpr (C >= 128) {
    // wide layers: a thread owns FOUR consecutive outputs (16-byte weight loads) and 1/NP4 of the inputs, so the whole
    // weight slice of a thread is a handful of loads in flight at once instead of 64 scalar ones in eight round trips
    constexpr int OT = C / 4, NP4 = CFG::NT / OT, PER4 = C / NP4;
    const int t = threadIdx.x, o4 = (t % OT) * 4, part = t / OT;
    float acc[PPW][4];
#pragma unroll
    for (int s = 0; s < PPW; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[s][e] = 0.f;
    f32x4 w[PER4];
#pragma unroll
    for (int k = 0; k < PER4; ++k) w[k] = *reinterpret_cast<const f32x4*>(W + (size_t)(part * PER4 + k) * C + o4);
#pragma unroll
    for (int k = 0; k < PER4; ++k) {
#pragma unroll
      for (int s = 0; s < PPW; ++s) {
        const float xv = x[s * x_stride + part * PER4 + k];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[s][e] += w[k][e] * xv;
      }
    }
    // partials [slot][part][C] -> the same meeting point as the scalar form below (NP4 parts)
    static_assert(PPW * NP4 * C <= CFG::RED, "matvec scratch");
#pragma unroll
    for (int s = 0; s < PPW; ++s)
      *reinterpret_cast<f32x4*>(red + (s * NP4 + part) * C + o4) = f32x4{acc[s][0], acc[s][1], acc[s][2], acc[s][3]};
    __syncthreads();
    for (int i = t; i < PPW * C; i += CFG::NT) {
      const int s = i / C, oo = i % C;
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < NP4; ++k) v += red[(s * NP4 + k) * C + oo];
      fin(s, oo, v);
    }
    __syncthreads();
    return;
  }
  float w[PER];
  lean_matvec_load<CFG>(W, w);
  lean_matvec_run<CFG>(w, x, x_stride, red, fin);
}
// sum over the NO lanes that share a pixel (consecutive lanes: o = item % NO); every lane ends with the total
template <int NO>
__device__ __forceinline__ float lean_octet_sum(float v) {
  v += lane_xor1(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, false));
  if (NO >= 8) v += __shfl_xor(v, 4);
  if (NO >= 16) v += __shfl_xor(v, 8);
  return v;
}

template <typename T, typename CFG>
__global__ __launch_bounds__(CFG::NT) void k_stage_fwd_lean(StageArgs a) {
  WGSTAMP(CFG::C == 32 ? 2 : -1);
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int C = CFG::C, NO = CFG::NO, NP = CFG::NP, TPP = CFG::TPP, IPT = CFG::IPT, PPW = CFG::PPW, WZ = CFG::WZ;
  constexpr int K = CFG::K, R = CFG::R, WP = CFG::WP, NPAD = CFG::NPAD;
  // (groups dispatched last-first, as in the backward kernel below: the heavier spatial branch goes first)
  if (a.lead > 0 && (int)blockIdx.x < a.lead) {        // lead workgroups: the BatchNorm statistics of ALL groups (row 0 only)
    if (blockIdx.y == 0) bn_lead_block<CFG::NT>(a.bnfin, blockIdx.x, a.lead_flag, reinterpret_cast<double*>(sm));
    return;
  }
  const int bx = (int)blockIdx.x - a.lead;
  const int g = gridDim.y - 1 - blockIdx.y, t = threadIdx.x, slot = t / TPP, lt = t % TPP;
  const int b = bx * PPW + slot;
  const bool live = b < a.B;
  const int kind = a.kind[g];
  float* coefL = sm;                                   // [C][2] scale, shift
  float* Zs = sm + 2 * C + slot * CFG::SLOT;           // [NP][C]
  float* vec = Zs + CFG::ZF;                           // spectral: pooled | h | gate ; spatial: m | t1 (padded maps) | s
  float* red = sm + 2 * C + PPW * CFG::SLOT;           // [RED]
  float* vec0 = sm + 2 * C + CFG::ZF;                  // slot 0's vectors (matvec operands are addressed with a slot stride)

  // ---- conv output of this thread's items in flight first ----
  float z[IPT][8];
  const size_t ypatch = (size_t)g * a.y_gs + (size_t)(live ? b : 0) * CFG::HWC * a.y_rs;
  float yraw[IPT][CFG::POOL ? 4 : 1][8];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int it = lt + j * TPP, itc = it < CFG::ITEMS ? it : 0;
    const int p = itc / NO, o = itc % NO;
    if (CFG::POOL) {
      const int hz = p / WZ, wz = p % WZ, p00 = (2 * hz) * CFG::WC + 2 * wz;
      const int po[4] = {p00, p00 + 1, p00 + CFG::WC, p00 + CFG::WC + 1};
#pragma unroll
      for (int k = 0; k < 4; ++k) lean_ld8<CFG::YF, true>(yraw[j][k], a.y, ypatch + (size_t)po[k] * a.y_rs + o * 8);
    } else {
      lean_ld8<CFG::YF, true>(yraw[j][0], a.y, ypatch + (size_t)p * a.y_rs + o * 8);
    }
  }
  // ---- attention weights in flight as well: the spectral mat-vec slices go to registers, the spatial stencils to LDS ----
  constexpr bool PRE = C < 128;                          // (the 128-wide mat-vecs load 16-byte slices at their call)
  float w1[C / CFG::NPART], w2[C / CFG::NPART];
  if (PRE && kind == KIND_SPECTRAL) { lean_matvec_load<CFG>(a.att[g].p[0], w1); lean_matvec_load<CFG>(a.att[g].p[2], w2); }
  float* wL = sm + 2 * C + PPW * CFG::SLOT + CFG::RED;   // [C + 2 K K] wc | k1 | k2
  if (kind == KIND_SPATIAL)
    for (int i = t; i < C + 2 * K * K; i += CFG::NT)
      wL[i] = i < C ? a.att[g].p[0][i] : (i < C + K * K ? a.att[g].p[2][i - C] : a.att[g].p[4][i - C - K * K]);
  // ---- BatchNorm coefficients (from the finalize launch, or derived here in eval mode) ----
  if (a.bn_inkernel) {
    float* lc = red;                                    // [C][4], C <= 128 -> 512 floats, then 514 of scratch
    bn_coef_block<CFG::NT>(a.bnfin, g, lc, red + 4 * C, bx == 0);
    for (int i = t; i < 2 * C; i += CFG::NT) coefL[i] = lc[(i >> 1) * 4 + (i & 1)];
  } else if (a.lead > 0) {
    // the coefficients come from THIS launch's lead workgroups (everything above is in flight while they work)
    const bool ok = bn_lead_wait(a.lead_flag, a.lead_need, reinterpret_cast<int*>(red));
    const float* coef = a.coef + (size_t)g * a.coef_gs;
    for (int i = t; i < C; i += CFG::NT) {
      const float2 q = fan_load2(coef + i * 4);
      coefL[2 * i] = ok ? q.x : __builtin_nanf(""); coefL[2 * i + 1] = ok ? q.y : __builtin_nanf("");
    }
  } else {
    const float* coef = a.coef + (size_t)g * a.coef_gs;
    for (int i = t; i < 2 * C; i += CFG::NT) coefL[i] = coef[(i >> 1) * 4 + (i & 1)];
  }
  if (kind == KIND_SPATIAL)
    for (int i = lt; i < 2 * NPAD; i += TPP) vec[i] = 0.f;      // borders of the padded maps
  __syncthreads();
  // ---- BN -> ReLU -> (2x2 max-pool), patch into LDS ----
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int it = lt + j * TPP;
    const int itc = it < CFG::ITEMS ? it : 0, p = itc / NO, o = itc % NO;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(coefL + (o * 8 + e) * 2);
      sc[e] = q[0]; sh[e] = q[1]; sc[e + 1] = q[2]; sh[e + 1] = q[3];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float m = yraw[j][0][e] * sc[e] + sh[e];
      if (CFG::POOL) {
#pragma unroll
        for (int k = 1; k < 4; ++k) m = max_nan(m, yraw[j][k][e] * sc[e] + sh[e]);
      }
      z[j][e] = (live && it < CFG::ITEMS) ? relu_nan(m) : 0.f;
    }
    // only the plain network's classifier flatten reads the un-gated patch back from LDS
    if (it < CFG::ITEMS && kind != KIND_SPECTRAL && kind != KIND_SPATIAL) {
      *reinterpret_cast<f32x4*>(Zs + p * C + o * 8) = f32x4{z[j][0], z[j][1], z[j][2], z[j][3]};
      *reinterpret_cast<f32x4*>(Zs + p * C + o * 8 + 4) = f32x4{z[j][4], z[j][5], z[j][6], z[j][7]};
    }
  }
  if (kind != KIND_SPECTRAL && kind != KIND_SPATIAL) __syncthreads();
  // tile rows per chunk: the haloed grid, or (a_compact) the pixels only -- the readers keep the halo in LDS
  const int trows = a.a_compact ? NP : CFG::QZ;
  T* tile = a.a_tl ? (T*)a.a_tl + (size_t)g * a.a_gs + ((size_t)(live ? b : 0) * a.a_nc + a.a_ch0) * trows * 16 : nullptr;
  auto trow = [&](int p) { return a.a_compact ? p : (p / WZ + 1) * CFG::W2 + p % WZ + 1; };
  float* feat = (a.feat && live) ? a.feat + (size_t)g * a.feat_gs + (size_t)b * a.F[g] : nullptr;
  float* save = (a.attsave && live) ? a.attsave + ((size_t)g * a.B + b) * a.attsave_ld : nullptr;
  float* sm0 = sm + 2 * C;
  if (kind == KIND_SPECTRAL) {
    float* pooled = vec; float* hL = vec + C; float* gL = vec + 2 * C;
    {
      float part[1][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        part[0][e] = 0.f;
#pragma unroll
        for (int j = 0; j < IPT; ++j) part[0][e] += z[j][e];
      }
      float* const outs[1] = {vec0};
      lean_colsum_reg<CFG, 1>(part, red, outs, CFG::SLOT, 1.f / (float)NP);
    }
    const float* c1 = a.att[g].p[1]; const float* c2 = a.att[g].p[3];
    auto fin1 = [&](int s, int o, float v) { sm0[s * CFG::SLOT + CFG::ZF + C + o] = relu_nan(v + c1[o]); };
    auto fin2 = [&](int s, int o, float v) { sm0[s * CFG::SLOT + CFG::ZF + 2 * C + o] = sigmoidf_(v + c2[o]); };
    if constexpr (PRE) {
      lean_matvec_run<CFG>(w1, vec0, CFG::SLOT, red, fin1);
      lean_matvec_run<CFG>(w2, vec0 + C, CFG::SLOT, red, fin2);
    } else {
      lean_matvec<CFG>(a.att[g].p[0], vec0, CFG::SLOT, red, fin1);
      lean_matvec<CFG>(a.att[g].p[2], vec0 + C, CFG::SLOT, red, fin2);
    }
    if (save) for (int i = lt; i < 3 * C; i += TPP) __builtin_nontemporal_store(vec[i], save + i);
    if (feat) for (int c = lt; c < C; c += TPP) feat[c] = gL[c] * pooled[c];      // mean_p(z * gate) = gate * mean_p(z)
    if (tile && live) {
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const int it = lt + j * TPP;
        if (it >= CFG::ITEMS) continue;
        const int p = it / NO, o = it % NO, h = p / WZ, w = p % WZ;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = z[j][e] * gL[o * 8 + e];
        lean_tl_store8(tile, trows, trow(p), o, v);
      }
    }
  } else if (kind == KIND_SPATIAL) {
    float* mL = vec; float* t1L = vec + NPAD; float* sL = vec + 2 * NPAD;
    const float* wc = wL; const float bc = a.att[g].p[1][0];
    const float* k1 = wL + C; const float b1 = a.att[g].p[3][0];
    const float* k2 = wL + C + K * K; const float b2 = a.att[g].p[5][0];
    // m = relu(channel_pool(z)): 8 channels per lane, the NO lanes of a pixel meet through DPP / shuffles
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = lt + j * TPP, itc = it < CFG::ITEMS ? it : 0, p = itc / NO, o = itc % NO;
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += wc[o * 8 + e] * z[j][e];
      acc = lean_octet_sum<NO>(acc);
      if (it < CFG::ITEMS && o == 0) mL[(p / WZ + R) * WP + p % WZ + R] = relu_nan(acc + bc);
    }
    __syncthreads();
    // the two k x k stencils: the NO lanes of a pixel split the kernel rows
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const float* src = pass ? t1L : mL; const float* kw = pass ? k2 : k1;
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const int it = lt + j * TPP, itc = it < CFG::ITEMS ? it : 0, p = itc / NO, o = itc % NO, h = p / WZ, w = p % WZ;
        float acc = 0.f;
        for (int ky = o; ky < K; ky += NO) {
#pragma unroll
          for (int kx = 0; kx < K; ++kx) acc += kw[ky * K + kx] * src[(h + ky) * WP + w + kx];
        }
        acc = lean_octet_sum<NO>(acc);
        if (it < CFG::ITEMS && o == 0) {
          if (pass == 0) t1L[(h + R) * WP + w + R] = relu_nan(acc + b1);
          else sL[p] = sigmoidf_(acc + b2);
        }
      }
      __syncthreads();
    }
    if (save) {     // m | t1 (padded maps, zero borders included) | s, one slot of a.vslot floats each
      for (int i = lt; i < NPAD; i += TPP) {
        __builtin_nontemporal_store(mL[i], save + i);
        __builtin_nontemporal_store(t1L[i], save + a.vslot + i);
      }
      for (int p = lt; p < NP; p += TPP) __builtin_nontemporal_store(sL[p], save + 2 * a.vslot + p);
    }
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = lt + j * TPP;
      if (it >= CFG::ITEMS) continue;
      const int p = it / NO, o = it % NO, h = p / WZ, w = p % WZ;
      const float sp = sL[p];
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = z[j][e] * sp;
      if (tile && live) lean_tl_store8(tile, trows, trow(p), o, v);
      if (feat) {   // the gated map replaces the patch in LDS for the class pool below
        *reinterpret_cast<f32x4*>(Zs + p * C + o * 8) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(Zs + p * C + o * 8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
      }
    }
    if (a.feat) {
      __syncthreads();
      if (feat) {
        constexpr int PS = CFG::PS, HPc = CFG::HZ / PS, WPc = CFG::WZ / PS;
        for (int i = lt; i < C * HPc * WPc; i += TPP) {       // flatten order of the reference: (c, ph, pw)
          const int c = i / (HPc * WPc), rem = i % (HPc * WPc), ph = rem / WPc, pw = rem % WPc;
          float m = -3.4e38f;
#pragma unroll
          for (int dy = 0; dy < PS; ++dy)
#pragma unroll
            for (int dx = 0; dx < PS; ++dx) m = max_nan(m, Zs[((ph * PS + dy) * WZ + pw * PS + dx) * C + c]);
          feat[i] = m;
        }
      }
    }
  } else if (feat) {   // plain network (vanilla_CNN): the last stage's map is the classifier input, NCHW flatten
    for (int i = lt; i < C * NP; i += TPP) feat[i] = Zs[(i % NP) * C + i / NP];
  }
  if (kind != KIND_SPECTRAL && kind != KIND_SPATIAL && tile && live) {
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = lt + j * TPP;
      if (it >= CFG::ITEMS) continue;
      const int p = it / NO, o = it % NO;
      lean_tl_store8(tile, trows, trow(p), o, z[j]);
    }
  }
  if (tile && live && !a.a_compact) lean_tl_halo<T, CFG>(tile, lt);
}

// ------------------------------------------------------------------------------------------------
// Lean backward of a stage: same item ownership.  Per patch and group it needs
//   spectral: T_c = sum_p D z, two C x C mat-vecs, then dZ = D g + dp ;  spatial: per-pixel sums of D z, two transposed
//   k x k stencils, dZ = D s + dm wc ; then the ReLU / pool mask, the BatchNorm partial sums (sum dv, sum dv xhat) and
//   the attention parameter-gradient vectors.  D = incoming gradient of the gated map (+ the classifier-feature path).
// ------------------------------------------------------------------------------------------------
template <typename CFG>
struct LeanBwd {
  static constexpr int C = CFG::C, NP = CFG::NP, NPAD = CFG::NPAD;
  static constexpr int VEC = 8 * C > 4 * NPAD + 2 * NP + 3 * C ? 8 * C : 4 * NPAD + 2 * NP + 3 * C;
  static constexpr int SLOT = VEC;                        // per-patch vectors only: the maps never leave the registers
  static constexpr int LDS = 4 * C + CFG::PPW * SLOT + CFG::RED + C + 2 * CFG::K * CFG::K + 4 + (C < 128 ? 2 * C * C : 0);
};

template <typename CFG>
__global__ __launch_bounds__(CFG::NT, CFG::MINW) void k_stage_bwd_lean(StageBwdArgs ba) {
  WGSTAMP(CFG::C == 32 ? 0 : (CFG::C == 64 ? 1 : -1));
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const StageArgs& a = ba.f;
  constexpr int C = CFG::C, NO = CFG::NO, NP = CFG::NP, TPP = CFG::TPP, IPT = CFG::IPT, PPW = CFG::PPW, WZ = CFG::WZ;
  constexpr int K = CFG::K, KK = K * K, R = CFG::R, WP = CFG::WP, NPAD = CFG::NPAD;
  constexpr int SLOT = LeanBwd<CFG>::SLOT, NPOS = CFG::POOL ? 4 : 1;
  // storage of the gradient maps follows the conv outputs': bf16 next to half outputs, fp32 next to fp32 (launcher checks)
  constexpr int GF = CFG::YF == FMT_F16 ? FMT_BF16 : FMT_F32;
  // groups are dispatched last-first: Hang2020's spatial branch (group 1) is the heavier program, and of the two
  // workgroups that share a CU the one dispatched first wins the issue arbitration -- with the spectral branch first its
  // workgroups finished at ~21 us and left the spatial ones alone on half-empty CUs until 32 us (same-box alternation:
  // step 0.5211 -> 0.5182 ms; with the forward kernel 0.5128 -> 0.5111)
  const int g = gridDim.y - 1 - blockIdx.y, t = threadIdx.x, slot0 = t / TPP, lt0 = t % TPP;
  const int kind = a.kind[g];
  float* coefL = sm;                                   // [C][4] scale, shift, mean, rstd
  float* sm0 = sm + 4 * C;
  float* vec0 = sm0 + slot0 * SLOT;
  float* red = sm0 + PPW * SLOT;
  constexpr int VOFF = 0;                              // vectors of slot s start at sm0 + s * SLOT + VOFF

  // ---- once per workgroup: BatchNorm coefficients, attention weights (spectral mat-vec slices -> registers,
  //      spatial stencils -> LDS), zero borders of the padded gradient maps ----
  {
    const float* coef = a.coef + (size_t)g * a.coef_gs;
    for (int i = t; i < 4 * C; i += CFG::NT) coefL[i] = coef[i];
  }
  constexpr bool PRE = C < 128;                          // the two C x C mat-vec matrices fit in LDS
  float* wL = sm0 + PPW * SLOT + CFG::RED;               // [C + 2 K K] wc | k1 | k2
  float* W1L = wL + C + 2 * KK + ((4 - (C + 2 * KK) % 4) % 4);      // [C][C] each, 16-byte aligned
  float* W2L = W1L + C * C;
  if (PRE && kind == KIND_SPECTRAL) {
    const f32x4* s1 = reinterpret_cast<const f32x4*>(a.att[g].p[4]); const f32x4* s2 = reinterpret_cast<const f32x4*>(a.att[g].p[5]);
    for (int i = t; i < C * C / 4; i += CFG::NT) {
      reinterpret_cast<f32x4*>(W1L)[i] = s1[i];
      reinterpret_cast<f32x4*>(W2L)[i] = s2[i];
    }
  }
  if (kind == KIND_SPATIAL) {
    for (int i = t; i < C + 2 * KK; i += CFG::NT)
      wL[i] = i < C ? a.att[g].p[0][i] : (i < C + KK ? a.att[g].p[2][i - C] : a.att[g].p[4][i - C - KK]);
    for (int i = lt0; i < NPAD; i += TPP) { vec0[2 * NPAD + NP + i] = 0.f; vec0[3 * NPAD + NP + i] = 0.f; }   // d2, d1 (padded)
  }

  // ---- a workgroup walks patch batches blockIdx.x, + gridDim.x, ...; the next batch's conv output, incoming gradient
  //      and saved attention state are fetched into registers (raw bytes) while the current one is worked on ----
  constexpr int NSAVE = (3 * C > 2 * NPAD + NP ? 3 * C : 2 * NPAD + NP);
  constexpr int NSV = (NSAVE + TPP - 1) / TPP;
  LeanRaw<CFG::YF> yq[IPT][NPOS];
  LeanRaw<GF> dq[IPT];
  float sv[NSV];
  const int nsave = kind == KIND_SPECTRAL ? 3 * C : (kind == KIND_SPATIAL ? 2 * NPAD + NP : 0);
  auto fetch = [&](int bi) {
    const int b = bi * PPW + slot0, bb = b < a.B ? b : 0;
    const size_t ypatch = (size_t)g * a.y_gs + (size_t)bb * CFG::HWC * a.y_rs;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = lt0 + j * TPP, itc = it < CFG::ITEMS ? it : 0;
      const int p = itc / NO, o = itc % NO;
      if (CFG::POOL) {
        const int hz = p / WZ, wz = p % WZ, p00 = (2 * hz) * CFG::WC + 2 * wz;
        const int po[4] = {p00, p00 + 1, p00 + CFG::WC, p00 + CFG::WC + 1};
#pragma unroll
        for (int k = 0; k < NPOS; ++k) lean_ld8_raw<CFG::YF, false>(yq[j][k], a.y, ypatch + (size_t)po[k] * a.y_rs + o * 8);
      } else {
        lean_ld8_raw<CFG::YF, false>(yq[j][0], a.y, ypatch + (size_t)p * a.y_rs + o * 8);
      }
      if (ba.da) lean_ld8_raw<GF, true>(dq[j], ba.da, (size_t)g * ba.da_gs + ((size_t)bb * NP + p) * C + o * 8);
    }
    const float* save = a.attsave + ((size_t)g * a.B + bb) * a.attsave_ld;
#pragma unroll
    for (int k = 0; k < NSV; ++k) {
      const int i = lt0 + k * TPP;
      if (i < nsave) {
        // spectral: pooled | h | gate, contiguous; spatial: m | t1 (padded maps) | s, one slot of a.vslot floats each
        const int src = kind == KIND_SPECTRAL ? i : (i < NPAD ? i : (i < 2 * NPAD ? a.vslot + i - NPAD : 2 * a.vslot + i - 2 * NPAD));
        sv[k] = __builtin_nontemporal_load(save + src);
      }
    }
  };
  const int nb = (a.B + PPW - 1) / PPW;
  // no finalize launch (StageBwdArgs::bn_fan_count): the BatchNorm partial sums of all the patches this workgroup walks,
  // per channel (thread lt of a slot owns channels lt, lt + TPP, ...)
  constexpr int NCH = (C + TPP - 1) / TPP;
  float wgsum[NCH][2];
#pragma unroll
  for (int k = 0; k < NCH; ++k) { wgsum[k][0] = 0.f; wgsum[k][1] = 0.f; }
  // the 128-wide stage has no registers to spare (a loop costs it spills): one batch per workgroup, fetched at the top
  constexpr bool PF = CFG::PERSIST;
  if (PF && (int)blockIdx.x < nb) fetch(blockIdx.x);

  for (int bi = blockIdx.x; bi < nb; bi += gridDim.x) {
  // the thread's place is re-derived from an opaque copy in every round: otherwise the compiler hoists every
  // loop-invariant address out of the loop and pays ~50 registers (= a workgroup per CU) for it
  int lt = lt0;
  asm volatile("" : "+v"(lt));
  const int slot = PPW > 1 ? (int)threadIdx.x / TPP : 0;
  float* vec = sm0 + slot * SLOT;
  const int b = bi * PPW + slot;
  const bool live = b < a.B;
  if (!PF) fetch(bi);
  // ---- the fetched bytes -> floats; saved attention state -> LDS ----
  float yraw[IPT][NPOS][8];
  float D[IPT][8];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
#pragma unroll
    for (int k = 0; k < NPOS; ++k) lean_unpack8<CFG::YF>(yraw[j][k], yq[j][k]);
    if (ba.da) lean_unpack8<GF>(D[j], dq[j]);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) D[j][e] = 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < NSV; ++k) {
    const int i = lt + k * TPP;
    if (i < nsave) vec[i] = sv[k];
  }
  const float* df = (ba.dfeat && live) ? ba.dfeat + (size_t)g * ba.dfeat_gs + (size_t)b * a.F[g] : nullptr;
  __syncthreads();

  // ---- recompute BN -> ReLU -> pool; keep z, the window position of the maximum and xhat there ----
  // RECOMP (several items per thread: the 24x24 crops): only the selected raw conv output of an item stays in registers;
  // z = relu(BN(y)) and xhat are re-derived from it (two FMAs and one LDS read of the coefficients) at each of their three
  // uses instead of living in 16 registers per item through the whole kernel -- the 5-item / 3-item configurations
  // spilled 78 / 44 registers to scratch with them
  constexpr bool RECOMP = CFG::HC > 12;      // (the 12x12 x 128 stage fits its registers; measured 38 -> 49 us with the recomputation)
  float z[RECOMP ? 1 : IPT][8], xh[IPT][8];      // RECOMP: xh holds the selected raw output, z is unused
  unsigned first[IPT];
  unsigned vmask = 0u;                          // RECOMP: bit j = item j exists and its patch is live
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int it = lt + j * TPP, itc = it < CFG::ITEMS ? it : 0, o = itc % NO;
    first[j] = 0u;
    if (live && it < CFG::ITEMS) vmask |= 1u << j;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(coefL + (o * 8 + e) * 4);
      float m = yraw[j][0][e] * q[0] + q[1], ys = yraw[j][0][e];
      if (CFG::POOL) {
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          const float v = yraw[j][k][e] * q[0] + q[1];
          if (v > m) { m = v; ys = yraw[j][k][e]; first[j] = (first[j] & ~(3u << (2 * e))) | ((unsigned)k << (2 * e)); }
        }
      }
      if constexpr (RECOMP) xh[j][e] = ys;
      else {
        z[j][e] = (live && it < CFG::ITEMS) ? relu_nan(m) : 0.f;
        xh[j][e] = (ys - q[2]) * q[3];
      }
    }
  }
  auto ZV = [&](int j, int e) -> float {
    if constexpr (RECOMP) {
      const int it = lt + j * TPP, o = (it < CFG::ITEMS ? it : 0) % NO;
      const f32x2 q = *reinterpret_cast<const f32x2*>(coefL + (o * 8 + e) * 4);
      return ((vmask >> j) & 1u) ? relu_nan(xh[j][e] * q[0] + q[1]) : 0.f;
    } else return z[j][e];
  };
  auto XV = [&](int j, int e) -> float {
    if constexpr (RECOMP) {
      const int it = lt + j * TPP, o = (it < CFG::ITEMS ? it : 0) % NO;
      const f32x2 q = *reinterpret_cast<const f32x2*>(coefL + (o * 8 + e) * 4 + 2);
      return (xh[j][e] - q[0]) * q[1];
    } else return xh[j][e];
  };
  // the raw registers are free again: next batch in flight behind everything below
  if (PF && bi + (int)gridDim.x < nb) fetch(bi + gridDim.x);
  float* bnp = (ba.bnpart && live) ? ba.bnpart + (size_t)g * ba.bnpart_gs + (size_t)b * C * 2 : nullptr;
  float* vout = (ba.vec && live) ? ba.vec + (size_t)g * ba.vec_gs + (size_t)b * ba.vec_ld : nullptr;

  float dv[IPT][8];
  if (kind == KIND_SPECTRAL) {
    float* pooled = vec; float* hL = vec + C; float* gL = vec + 2 * C;
    float* d2L = vec + 3 * C; float* d1L = vec + 4 * C; float* dpL = vec + 5 * C;
    const float inv = 1.f / (float)NP;
    // D += df / NP (the features are the pixel mean of the gated map), T_c = sum_p D z
    {
      float part[1][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) part[0][e] = 0.f;
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const int it = lt + j * TPP, itc = it < CFG::ITEMS ? it : 0, o = itc % NO;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (df) D[j][e] += df[o * 8 + e] * inv;
          part[0][e] += D[j][e] * ZV(j, e);           // z is zero for threads without an item
        }
      }
      float* const outs[1] = {sm0 + VOFF + 3 * C};
      lean_colsum_reg<CFG, 1>(part, red, outs, SLOT, 1.f);
    }
    for (int c = lt; c < C; c += TPP) d2L[c] = d2L[c] * gL[c] * (1.f - gL[c]);
    __syncthreads();
    auto fin2 = [&](int s, int i, float v) {
      float* vs = sm0 + s * SLOT + VOFF;
      vs[4 * C + i] = vs[C + i] > 0.f ? v : 0.f;
    };
    auto fin1 = [&](int s, int i, float v) { sm0[s * SLOT + VOFF + 5 * C + i] = v * inv; };
    if constexpr (PRE) {
      lean_matvec_lds<CFG>(W2L, sm0 + VOFF + 3 * C, SLOT, red, fin2);
      lean_matvec_lds<CFG>(W1L, sm0 + VOFF + 4 * C, SLOT, red, fin1);
    } else {
      lean_matvec<CFG>(a.att[g].p[5], sm0 + VOFF + 3 * C, SLOT, red, fin2);
      lean_matvec<CFG>(a.att[g].p[4], sm0 + VOFF + 4 * C, SLOT, red, fin1);
    }
    if (vout)
      for (int c = lt; c < C; c += TPP) { vout[c] = d2L[c]; vout[C + c] = hL[c]; vout[2 * C + c] = d1L[c]; vout[3 * C + c] = pooled[c]; }
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = lt + j * TPP, itc = it < CFG::ITEMS ? it : 0, o = itc % NO;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dz = D[j][e] * gL[o * 8 + e] + dpL[o * 8 + e];
        dv[j][e] = ZV(j, e) > 0.f ? dz : 0.f;
      }
    }
  } else if (kind == KIND_SPATIAL) {
    float* mL = vec; float* t1L = vec + NPAD; float* sL = vec + 2 * NPAD;
    float* d2L = sL + NP; float* d1L = d2L + NPAD; float* dmL = d1L + NPAD;
    const float* wc = wL; const float* k1 = wL + C; const float* k2 = wL + C + KK;
    // classifier-feature path: only the un-pooled class pool (PS == 1) is handled here; the launcher keeps the older
    // kernel for stages whose features come from a real max-pool
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = lt + j * TPP, itc = it < CFG::ITEMS ? it : 0, p = itc / NO, o = itc % NO;
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (CFG::PS == 1 && df) D[j][e] += df[(o * 8 + e) * NP + p];
        acc += D[j][e] * ZV(j, e);
      }
      acc = lean_octet_sum<NO>(acc);
      if (it < CFG::ITEMS && o == 0) { const float sp = sL[p]; d2L[(p / WZ + R) * WP + p % WZ + R] = acc * sp * (1.f - sp); }
    }
    __syncthreads();
    // d1 = (transposed k2 stencil of d2) masked by t1 > 0 ;  dm = (transposed k1 stencil of d1) masked by m > 0
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const float* src = pass ? d1L : d2L; const float* kw = pass ? k1 : k2; const float* gate = pass ? mL : t1L;
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const int it = lt + j * TPP, itc = it < CFG::ITEMS ? it : 0, p = itc / NO, o = itc % NO, h = p / WZ, w = p % WZ;
        float acc = 0.f;
        for (int ky = o; ky < K; ky += NO) {
#pragma unroll
          for (int kx = 0; kx < K; ++kx) acc += kw[(K - 1 - ky) * K + (K - 1 - kx)] * src[(h + ky) * WP + w + kx];
        }
        acc = lean_octet_sum<NO>(acc);
        if (it < CFG::ITEMS && o == 0) {
          const int pi = (h + R) * WP + w + R;
          const float v = gate[pi] > 0.f ? acc : 0.f;
          if (pass == 0) d1L[pi] = v; else dmL[p] = v;
        }
      }
      __syncthreads();
    }
    float dwp[1][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dwp[0][e] = 0.f;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = lt + j * TPP, itc = it < CFG::ITEMS ? it : 0, p = itc / NO, o = itc % NO;
      const float sp = sL[p], dm = dmL[p];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dz = D[j][e] * sp + dm * wc[o * 8 + e];
        dv[j][e] = ZV(j, e) > 0.f ? dz : 0.f;
        dwp[0][e] += dm * ZV(j, e);                   // z is zero for threads without an item
      }
    }
    if (ba.vec) {
      // [dwc (C) | dbc | dK1 (kk) | db1 | dK2 (kk) | db2]
      float* dwcL = dmL + NP;
      float* const outs[1] = {sm0 + VOFF + (int)(dmL + NP - vec)};
      lean_colsum_reg<CFG, 1>(dwp, red, outs, SLOT, 1.f);
      if (vout) {
        for (int c = lt; c < C; c += TPP) vout[c] = dwcL[c];
        // stencil-weight gradients: task = (kernel, tap), NP products each; LPT lanes per task split the rows
        constexpr int LPT = TPP >= 8 * KK ? 4 : 2;
        for (int task = lt / LPT; task < 2 * KK; task += TPP / LPT) {
          const bool fst = task < KK;
          const int jj = fst ? task : task - KK, ky = jj / K, kx = jj % K;
          const float* src = fst ? mL : t1L; const float* dd = fst ? d1L : d2L;
          float acc = 0.f;
          for (int h = lt % LPT; h < CFG::HZ; h += LPT)
#pragma unroll
            for (int w = 0; w < WZ; ++w) acc += src[(h + ky) * WP + w + kx] * dd[(h + R) * WP + w + R];
          acc += lane_xor1(acc);
          if (LPT == 4) acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, false));
          if (lt % LPT == 0) vout[fst ? C + 1 + jj : C + 2 + KK + jj] = acc;
        }
      }
      // bias gradients: sums of dm, d1, d2 (borders of the padded maps are zero): three waves, one map each
      {
        const int wv = lt >> 6, ln = lt & 63;
        if (wv < 3 && TPP >= 192) {
          const float* mp = wv == 0 ? dmL : (wv == 1 ? d1L : d2L);
          const int n = wv == 0 ? NP : NPAD;
          float acc = 0.f;
          for (int q = ln; q < n; q += 64) acc += mp[q];
          acc = wave_sum(acc);
          if (ln == 0 && vout) vout[wv == 0 ? C : (wv == 1 ? C + 1 + KK : C + 2 + 2 * KK)] = acc;
        } else if (TPP < 192 && lt < 3 && vout) {     // four patches per workgroup (2x2 maps): a handful of terms
          float acc = 0.f;
          if (lt == 0) { for (int p = 0; p < NP; ++p) acc += dmL[p]; }
          else { const float* mp = lt == 1 ? d1L : d2L; for (int q = 0; q < NPAD; ++q) acc += mp[q]; }
          vout[lt == 0 ? C : (lt == 1 ? C + 1 + KK : C + 2 + 2 * KK)] = acc;
        }
      }
    }
  } else {
    // plain stage (vanilla_CNN): the gradient of the map is the incoming gradient (+ the classifier's, NCHW flatten)
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int it = lt + j * TPP, itc = it < CFG::ITEMS ? it : 0, p = itc / NO, o = itc % NO;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float dz = D[j][e];
        if (df) dz += df[(o * 8 + e) * NP + p];
        dv[j][e] = ZV(j, e) > 0.f ? dz : 0.f;
      }
    }
  }
  // ---- outputs: dv (dense, or compact value + window position for pooled stages), BatchNorm partial sums ----
  constexpr bool h16 = GF == FMT_BF16;
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int it = lt + j * TPP;
    if (it >= CFG::ITEMS) continue;
    const int p = it / NO, o = it % NO;
    if (live) {
      const size_t dvi = (size_t)g * ba.dv_gs + (size_t)b * CFG::HWC * C;          // element index of the patch
      auto st8 = [&](size_t i, const float (&q)[8]) {
        if (h16) {
          u32x4 u;
#pragma unroll
          for (int e = 0; e < 4; ++e) u[e] = pack2_fmt(q[2 * e], q[2 * e + 1], FMT_BF16);
          *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(ba.dv) + dvi + i) = u;
        } else {
          *reinterpret_cast<f32x4*>(ba.dv + dvi + i) = f32x4{q[0], q[1], q[2], q[3]};
          *reinterpret_cast<f32x4*>(ba.dv + dvi + i + 4) = f32x4{q[4], q[5], q[6], q[7]};
        }
      };
      if (!CFG::POOL || ba.dv_compact) {
        st8((size_t)p * C + o * 8, dv[j]);
        if (CFG::POOL) {
          unsigned lo = 0u, hi = 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) { lo |= ((first[j] >> (2 * e)) & 3u) << (8 * e); hi |= ((first[j] >> (2 * (e + 4))) & 3u) << (8 * e); }
          unsigned char* fpos = reinterpret_cast<unsigned char*>(ba.dv) + (dvi + (size_t)NP * C) * (h16 ? 2 : 4);
          *reinterpret_cast<u32x2*>(fpos + (size_t)p * C + o * 8) = u32x2{lo, hi};
        }
      } else {
        // dense map of a pooled stage: the gradient lands on the window position of the maximum, zeros elsewhere
        const int hz = p / WZ, wz = p % WZ, p00 = (2 * hz) * CFG::WC + 2 * wz;
        const int po[4] = {p00, p00 + 1, p00 + CFG::WC, p00 + CFG::WC + 1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float q8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) q8[e] = ((first[j] >> (2 * e)) & 3u) == (unsigned)k ? dv[j][e] : 0.f;
          st8((size_t)po[k] * C + o * 8, q8);
        }
      }
    }
  }
  if (CFG::POOL && !ba.dv_compact && live) {
    // conv-resolution positions the floor pooling dropped (last row / column of an odd map) get no gradient
    const size_t dvi = (size_t)g * ba.dv_gs + (size_t)b * CFG::HWC * C;
    for (int i = lt; i < CFG::HWC * (C / 4); i += TPP) {
      const int pix = i / (C / 4), c4 = (i % (C / 4)) * 4, h = pix / CFG::WC, w = pix % CFG::WC;
      if ((h >> 1) >= CFG::HZ || (w >> 1) >= WZ) {
        if (h16) *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(ba.dv) + dvi + (size_t)pix * C + c4) = u32x2{0u, 0u};
        else *reinterpret_cast<f32x4*>(ba.dv + dvi + (size_t)pix * C + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  if (ba.bnpart || ba.bn_fan_count) {
    // sum dv, sum dv xhat per channel (dv is zero for threads without an item and for dead patches)
    float bp[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bp[0][e] = 0.f; bp[1][e] = 0.f;
#pragma unroll
      for (int j = 0; j < IPT; ++j) { bp[0][e] += dv[j][e]; bp[1][e] += dv[j][e] * XV(j, e); }
    }
    constexpr int S1 = 2 * PPW * (TPP / 64) * C;         // the head of red is the column-sum scratch
    float* s1L = red + S1 + slot * 2 * C;
    float* const outs[2] = {red + S1, red + S1 + C};
    lean_colsum_reg<CFG, 2>(bp, red, outs, 2 * C, 1.f);
    if (ba.bn_fan_count) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int c = lt + k * TPP;
        if (c < C) { wgsum[k][0] += s1L[c]; wgsum[k][1] += s1L[C + c]; }
      }
    } else if (bnp)
      for (int c = lt; c < C; c += TPP) *reinterpret_cast<f32x2*>(bnp + c * 2) = f32x2{s1L[c], s1L[C + c]};
  }
  if (!PF) break;       // (compile-time single trip: straight-line code)
  __syncthreads();      // the next batch reuses the vectors and the scratch
  }
  if (ba.bn_fan_count) {
    // ---- this workgroup's row of batch sums -> the logical group's fan-in (kernels.h) ----
    const int lt = t % TPP, slot = t / TPP;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int c = lt + k * TPP;
      if (c < C) { red[(slot * C + c) * 2] = wgsum[k][0]; red[(slot * C + c) * 2 + 1] = wgsum[k][1]; }
    }
    __syncthreads();
    if (t < C) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int sl = 0; sl < PPW; ++sl) { s0 += red[(sl * C + t) * 2]; s1 += red[(sl * C + t) * 2 + 1]; }
      fan_store2(ba.bn_fan_rows + (((size_t)g * gridDim.x + blockIdx.x) * C + t) * 2, s0, s1);
    }
    int* flag = reinterpret_cast<int*>(red + 2 * PPW * C);
    if (!fan_arrive(ba.bn_fan_count, flag)) return;
    // last arriver of its logical group: fold the group's rows in double, fixed order (row slices, then slices in order)
    const int q = fan_group(), x0 = fan_first(q), nrows = fan_members(q);
    constexpr int T = CFG::NT / C;
    const int col = t % C, sl = t / C;
    const float* rows = ba.bn_fan_rows + (size_t)g * gridDim.x * C * 2;
    double d0 = 0, d1 = 0;
#pragma unroll 4
    for (int r = sl; r < nrows; r += T) {
      const float2 v = fan_load2(rows + ((size_t)(x0 + r * FAN_R) * C + col) * 2);
      d0 += (double)v.x; d1 += (double)v.y;
    }
    double* dred = reinterpret_cast<double*>(sm);          // the whole LDS plan is free now (>= 2 NT doubles, checked by the launcher)
    dred[t] = d0; dred[CFG::NT + t] = d1;
    __syncthreads();
    if (t < C) {
      double t0 = 0, t1 = 0;
      for (int j = 0; j < T; ++j) { t0 += dred[j * C + t]; t1 += dred[CFG::NT + j * C + t]; }
      double* o = ba.bn_fan_sums + (((size_t)g * FAN_R + q) * C + t) * 2;
      o[0] = t0; o[1] = t1;
    }
  }
}

// workgroups of `kernel` one CU holds at once (queried once per instantiation)
template <typename KERNEL>
static int lean_wgs_per_cu(KERNEL kernel, int threads, size_t lds) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds) != hipSuccess || n < 1) n = 1;
  return n;
}
// persistent grid: as many workgroups per group as stay resident, evened out so that no workgroup walks a batch more
// than the others need to (nb batches over gx workgroups: ceil(nb / gx) rounds)
static int lean_grid_x(int nb, int G, int wgs_per_cu) {
  const int slots = wgs_per_cu * 256 / (G > 0 ? G : 1);
  if (slots < 1 || nb <= slots) return nb;
  const int rounds = (nb + slots - 1) / slots;
  return (nb + rounds - 1) / rounds;
}
template <typename CFG>
static int launch_stage_bwd_lean_c(const StageBwdArgs& a, int G, hipStream_t st) {
  const size_t lds = (size_t)LeanBwd<CFG>::LDS * 4;
  static_assert(LeanBwd<CFG>::LDS * 4 <= 64 * 1024, "lean stage backward: LDS plan exceeds the default 64 KiB limit");
  static_assert(LeanBwd<CFG>::LDS * 4 >= 2 * CFG::NT * 8, "lean stage backward: the fan-in fold needs 2 NT doubles of LDS");
  constexpr int GF = CFG::YF == FMT_F16 ? FMT_BF16 : FMT_F32;
  if ((a.da && a.da_fmt != GF) || a.dv_fmt != GF) {     // gradient maps are stored like the conv outputs
    dta_set_error("k_stage_bwd_lean: gradient-map storage format does not follow the conv outputs'");
    return 1;
  }
  static const int wpc = lean_wgs_per_cu(k_stage_bwd_lean<CFG>, CFG::NT, lds);
  const int nb = (a.f.B + CFG::PPW - 1) / CFG::PPW;
  hipLaunchKernelGGL((k_stage_bwd_lean<CFG>), dim3(CFG::PERSIST ? lean_grid_x(nb, G, wpc) : nb, G), dim3(CFG::NT), lds, st, a);
  DTA_CHECK_LAUNCH("k_stage_bwd_lean");
  return 0;
}
int launch_stage_bwd_lean(const StageBwdArgs& a, int G, hipStream_t st) {
  const bool h = a.f.y_fmt == FMT_F16;
  if (a.f.Hc != 11 && a.f.Hc != 5) {      // 24x24 crops (spectral networks, bf16 mode)
    if (a.f.C == 32) return launch_stage_bwd_lean_c<LeanCfg<32, 24, 24, 0, FMT_F16>>(a, G, st);
    if (a.f.C == 64) return launch_stage_bwd_lean_c<LeanCfg<64, 24, 24, 1, FMT_F16>>(a, G, st);
    return launch_stage_bwd_lean_c<LeanCfg<128, 12, 12, 1, FMT_F16>>(a, G, st);
  }
  if (a.f.C == 32) return h ? launch_stage_bwd_lean_c<LeanCfg<32, 11, 11, 0, FMT_F16>>(a, G, st) : launch_stage_bwd_lean_c<LeanCfg<32, 11, 11, 0, FMT_F32>>(a, G, st);
  if (a.f.C == 64) return h ? launch_stage_bwd_lean_c<LeanCfg<64, 11, 11, 1, FMT_F16>>(a, G, st) : launch_stage_bwd_lean_c<LeanCfg<64, 11, 11, 1, FMT_F32>>(a, G, st);
  return h ? launch_stage_bwd_lean_c<LeanCfg<128, 5, 5, 1, FMT_F16>>(a, G, st) : launch_stage_bwd_lean_c<LeanCfg<128, 5, 5, 1, FMT_F32>>(a, G, st);
}

template <typename T, typename CFG>
static int launch_stage_fwd_lean_c(const StageArgs& a, int G, hipStream_t st) {
  const size_t lds = (size_t)CFG::LDS_FWD * 4;     // < 24 KiB for the 11x11 networks
  if (lds > 64 * 1024) {
    static DevOnce attr_once;
    if (attr_once.first()) hipFuncSetAttribute((const void*)k_stage_fwd_lean<T, CFG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if (a.lead > 0 && lds < (size_t)CFG::NT / 64 * 24 * 8) { dta_set_error("stage_fwd_lean: no room for the lead workgroups' scratch"); return 1; }
  hipLaunchKernelGGL((k_stage_fwd_lean<T, CFG>), dim3((a.B + CFG::PPW - 1) / CFG::PPW + a.lead, G), dim3(CFG::NT), lds, st, a);
  DTA_CHECK_LAUNCH("k_stage_fwd_lean");
  return 0;
}
template <typename T>
int launch_stage_fwd_lean(const StageArgs& a, int G, hipStream_t st) {
  const bool h = a.y_fmt == FMT_F16;
  if (a.Hc != 11 && a.Hc != 5) {      // 24x24 crops (spectral networks, bf16 mode): several items per thread
    // (bf16 mode only: half conv outputs exist with 16-bit tiles alone, so the float-tile instantiations are never
    //  launched -- and one of them carried 176 B of scratch; stage_fwd_is_lean() admits these shapes for y_fmt F16 only)
    if constexpr (sizeof(T) == 2) {
      if (a.C == 32) return launch_stage_fwd_lean_c<T, LeanCfg<32, 24, 24, 0, FMT_F16>>(a, G, st);
      if (a.C == 64) return launch_stage_fwd_lean_c<T, LeanCfg<64, 24, 24, 1, FMT_F16>>(a, G, st);
      return launch_stage_fwd_lean_c<T, LeanCfg<128, 12, 12, 1, FMT_F16>>(a, G, st);
    } else {
      dta_set_error("lean stage forward: 24x24-class maps are a bf16-mode plan");
      return 1;
    }
  }
  if (a.C == 32) return h ? launch_stage_fwd_lean_c<T, LeanCfg<32, 11, 11, 0, FMT_F16>>(a, G, st) : launch_stage_fwd_lean_c<T, LeanCfg<32, 11, 11, 0, FMT_F32>>(a, G, st);
  if (a.C == 64) return h ? launch_stage_fwd_lean_c<T, LeanCfg<64, 11, 11, 1, FMT_F16>>(a, G, st) : launch_stage_fwd_lean_c<T, LeanCfg<64, 11, 11, 1, FMT_F32>>(a, G, st);
  return h ? launch_stage_fwd_lean_c<T, LeanCfg<128, 5, 5, 1, FMT_F16>>(a, G, st) : launch_stage_fwd_lean_c<T, LeanCfg<128, 5, 5, 1, FMT_F32>>(a, G, st);
}
template int launch_stage_fwd_lean<float>(const StageArgs&, int, hipStream_t);
template int launch_stage_fwd_lean<bf16_t>(const StageArgs&, int, hipStream_t);

// ================================================================================================
// Fused forward tail of Hang2020 on 11x11 patches (kernels.h: TailArgs).
//
// One workgroup = 4 patches x 2 branches = 8 waves; wave (slot, g) owns patch slot `slot` of branch g (0 spectral,
// 1 spatial), lane = (pooled pixel p, channel octet o) as in the lean stage kernels.  Phases (all 512 threads pass every
// barrier):
//   0  conv output (2x2 pool windows) -> BatchNorm -> ReLU -> max-pool in registers.  Spectral waves: pooled spectrum
//      (two shuffles).  Spatial waves finish their whole stage alone: channel pool (DPP / shuffles), the two 3x3 stencils
//      on the 2x2 map in registers, gate, gated map = the 512 class-pool features.
//   1-4 the two 128 x 128 spectral mat-vecs for the four slots with all 512 threads: thread (4 outputs, 8 inputs), weight
//      slices requested at kernel entry, partial sums met in LDS.
//   5  both last heads as ONE [640] x [640][classes] product per slot: thread (4 classes, 64 inputs) streams the transposed
//      weights with 16-byte loads (a wave reads whole rows), the features come as LDS broadcasts ([input][slot] float4).
//   6  blend + weighted cross-entropy + gradient + loss (ce_dev.h, the code of k_blend_ce) on the four score rows in LDS.
// The third conv's output is read once; features, attention state and branch scores are still written for the backward.
// ================================================================================================
constexpr int TAIL_KS = 10, TAIL_KROWS = 64, TAIL_TPS = 512 / TAIL_KS;      // head product: k-slices, rows per slice, threads per slice
static_assert(TAIL_KS * TAIL_KROWS == 640 && 128 % TAIL_KROWS == 0, "tail: slices must not straddle the two heads");
__host__ __device__ inline size_t tail_lds_floats(int ldw) {
  const size_t red = (size_t)16 * 128 * 4 > (size_t)TAIL_KS * 4 * ldw ? (size_t)16 * 128 * 4 : (size_t)TAIL_KS * 4 * ldw;
  return 512 + 2560 + 512 + 512 + red + (size_t)8 * ldw + 16 + 512;      // coef | X | pooled | h | partials | scores | ce scratch (sc, flag, sd)
}

#ifdef DTA_TICKS
__device__ long long g_tail_ticks[2][16];
#define TTICK(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 200)) g_tail_ticks[blockIdx.x ? 1 : 0][i] = wall_clock64(); } while (0)
extern "C" int dta_debug_tail_ticks(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_ticks), sizeof(long long) * 32); }
#else
#define TTICK(i) do { } while (0)
#endif
template <int YF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))      // (up to 256 registers: the head product keeps 16 weight vectors in flight)
void k_tail_fwd(TailArgs a) {
  constexpr int C = 128;
  TTICK(0);
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, slot = wave & 3, g = wave >> 2;
  const int B = a.st.B, N = a.classes, ldw = a.ldw;
  const int b = blockIdx.x * 4 + slot;
  const bool live = b < B;
  float* coefL = sm;                       // [2][C][2] scale, shift
  float* X = coefL + 512;                  // [640][4]: features of the four slots, input-major (0..127 spectral, 128..639 spatial)
  float* pooledT = X + 2560;               // [C][4]
  float* hT = pooledT + 512;               // [C][4]
  float* red = hT + 512;                   // mat-vec partials [16][C][4]; later the head product's [KS][4][ldw]
  const size_t redn = (size_t)16 * 128 * 4 > (size_t)TAIL_KS * 4 * ldw ? (size_t)16 * 128 * 4 : (size_t)TAIL_KS * 4 * ldw;
  float* scS = red + redn;                 // [4][N] spectral scores (row pitch N: what blend_ce_body indexes)
  float* scT = scS + 4 * (size_t)ldw;      // [4][N] spatial scores
  float* cesc = scT + 4 * (size_t)ldw;     // 8 floats + flag, then 256 doubles
  int* is_last = reinterpret_cast<int*>(cesc + 8);
  double* sd = reinterpret_cast<double*>(cesc + 16);

  // ---- everything that depends on nothing goes out first: mat-vec weight slices, the conv output, coefficients ----
  const int o4 = (t & 31) * 4, q = t >> 5;                 // mat-vec thread: outputs o4..o4+3, inputs 8q..8q+7
  f32x4 w1[8], w2[8];
  const float* A1 = a.st.att[0].p[0]; const float* A2 = a.st.att[0].p[2];      // input-major centre-tap matrices
#pragma unroll
  for (int k = 0; k < 8; ++k) w1[k] = *reinterpret_cast<const f32x4*>(A1 + (size_t)(8 * q + k) * C + o4);
  const int p = lane >> 4, o = lane & 15;                  // item: pooled pixel, channel octet
  float yraw[4][8];
  {
    const size_t ypatch = (size_t)g * a.st.y_gs + (size_t)(live ? b : 0) * 25 * a.st.y_rs;
    const int p00 = (2 * (p >> 1)) * 5 + 2 * (p & 1);
    const int po[4] = {p00, p00 + 1, p00 + 5, p00 + 6};
#pragma unroll
    for (int k = 0; k < 4; ++k) lean_ld8<YF, true>(yraw[k], a.st.y, ypatch + (size_t)po[k] * a.st.y_rs + o * 8);
  }
  {
    const int gi = t >> 8, j = t & 255;
    coefL[t] = a.st.coef[(size_t)gi * a.st.coef_gs + (j >> 1) * 4 + (j & 1)];
  }
  float wcv[8];
  if (g == 1) {
    const f32x4 u0 = *reinterpret_cast<const f32x4*>(a.st.att[1].p[0] + o * 8), u1 = *reinterpret_cast<const f32x4*>(a.st.att[1].p[0] + o * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { wcv[e] = u0[e]; wcv[4 + e] = u1[e]; }
  }
  // the head product's first two batches of weight rows (they depend on nothing either): thread (k-slice hks, classes 4 htt ..)
  constexpr int TB = 8, NBLK = TAIL_KROWS / TB;
  const int hks = t / TAIL_TPS, htt = t - hks * TAIL_TPS;
  const bool hact = hks < TAIL_KS && htt * 4 < ldw;
  const float* hwp = a.wt + (size_t)((hact ? hks : 0) * TAIL_KROWS) * ldw + (hact ? htt : 0) * 4;
  f32x4 wb[2][TB];
#pragma unroll
  for (int r = 0; r < TB; ++r) wb[0][r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(hwp + (size_t)r * ldw));
  // ... and what the loss needs besides the scores: labels, class weights, this thread's share of the normaliser
  BlendCeArgs cea = a.ce;
  cea.spec = scS - (size_t)blockIdx.x * 4 * N; cea.spat = scT - (size_t)blockIdx.x * 4 * N; cea.alpha = a.alpha;
  cea.B = B; cea.classes = N;
  CePre cpre = {};
  if (a.ce.labels) ce_prefetch(cea, cpre);
  __syncthreads();
  TTICK(1);
  // ---- BatchNorm -> ReLU -> 2x2 max-pool ----
  float z[8];
  {
    const float* cf = coefL + g * 256 + o * 16;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(cf + e * 2);
      const float sc0 = c4[0], sh0 = c4[1], sc1 = c4[2], sh1 = c4[3];
      float m0 = yraw[0][e] * sc0 + sh0, m1 = yraw[0][e + 1] * sc1 + sh1;
#pragma unroll
      for (int k = 1; k < 4; ++k) { m0 = max_nan(m0, yraw[k][e] * sc0 + sh0); m1 = max_nan(m1, yraw[k][e + 1] * sc1 + sh1); }
      z[e] = live ? relu_nan(m0) : 0.f; z[e + 1] = live ? relu_nan(m1) : 0.f;
    }
  }
  float* save = (a.st.attsave && live) ? a.st.attsave + ((size_t)g * B + b) * a.st.attsave_ld : nullptr;
  if (g == 0) {
    // pooled spectrum: mean over the four pooled pixels (lanes o, o + 16, o + 32, o + 48)
    float ps[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { float x = z[e]; x += __shfl_xor(x, 16); x += __shfl_xor(x, 32); ps[e] = x * 0.25f; }
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) pooledT[(o * 8 + e) * 4 + slot] = ps[e];
      if (save) {
        __builtin_nontemporal_store(f32x4{ps[0], ps[1], ps[2], ps[3]}, reinterpret_cast<f32x4*>(save + o * 8));
        __builtin_nontemporal_store(f32x4{ps[4], ps[5], ps[6], ps[7]}, reinterpret_cast<f32x4*>(save + o * 8 + 4));
      }
    }
  } else {
    // spatial attention on the 2x2 map, all of it inside the wave (reference Hang2020.py:105-124 with k = 3, pool 1)
    const float bc = a.st.att[1].p[1][0], b1 = a.st.att[1].p[3][0], b2 = a.st.att[1].p[5][0];
    const float* k1 = a.st.att[1].p[2]; const float* k2 = a.st.att[1].p[4];
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += wcv[e] * z[e];
    acc = lean_octet_sum<16>(acc);
    const float m = relu_nan(acc + bc);                    // every lane of pixel p holds m[p]
    float mm[4], t1[4], ss[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mm[i] = __shfl(m, 16 * i);
    // a 3x3 "same" stencil on a 2x2 map: out[h][w] = sum_{h', w'} k[(h' - h + 1) * 3 + (w' - w + 1)] * in[h'][w']
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s1 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s1 += k1[((j >> 1) - (i >> 1) + 1) * 3 + ((j & 1) - (i & 1) + 1)] * mm[j];
      t1[i] = relu_nan(s1 + b1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s2 += k2[((j >> 1) - (i >> 1) + 1) * 3 + ((j & 1) - (i & 1) + 1)] * t1[j];
      ss[i] = sigmoidf_(s2 + b2);
    }
    const float sp = p == 0 ? ss[0] : p == 1 ? ss[1] : p == 2 ? ss[2] : ss[3];
    float* feat = (a.st.feat && live) ? a.st.feat + (size_t)a.st.feat_gs + (size_t)b * 512 : nullptr;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = z[e] * sp;
      const int i = (o * 8 + e) * 4 + p;                   // class-pool size 1: feature (c, ph, pw) = gated map (reference flatten order)
      X[(128 + i) * 4 + slot] = v;
      if (feat) feat[i] = v;
    }
    if (save && lane < 16) {                               // m | t1 as zero-bordered 4x4 maps, then s: what the backward reads back
      const int hh = lane >> 2, ww = lane & 3;
      const bool in = hh >= 1 && hh <= 2 && ww >= 1 && ww <= 2;
      const int pi = in ? (hh - 1) * 2 + (ww - 1) : 0;
      const float mv = pi == 0 ? mm[0] : pi == 1 ? mm[1] : pi == 2 ? mm[2] : mm[3];
      const float tv = pi == 0 ? t1[0] : pi == 1 ? t1[1] : pi == 2 ? t1[2] : t1[3];
      __builtin_nontemporal_store(in ? mv : 0.f, save + lane);
      __builtin_nontemporal_store(in ? tv : 0.f, save + a.st.vslot + lane);
      if (lane < 4) __builtin_nontemporal_store(lane == 0 ? ss[0] : lane == 1 ? ss[1] : lane == 2 ? ss[2] : ss[3], save + 2 * a.st.vslot + lane);
    }
  }
  __syncthreads();
  TTICK(2);
  // (the conv output's registers are free now: the second mat-vec's weights go out)
#pragma unroll
  for (int k = 0; k < 8; ++k) w2[k] = *reinterpret_cast<const f32x4*>(A2 + (size_t)(8 * q + k) * C + o4);
  // ---- spectral mat-vecs: h = relu(A1 pooled + c1), gate = sigmoid(A2 h + c2), for the four slots ----
  auto matvec = [&](const f32x4 (&w)[8], const float* xT) {
    float acc[4][4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[s_][e] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xT + (8 * q + k) * 4);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[s_][e] += w[k][e] * xv[s_];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      *reinterpret_cast<f32x4*>(red + ((size_t)q * C + o4 + e) * 4) = f32x4{acc[0][e], acc[1][e], acc[2][e], acc[3][e]};
  };
  const int mo = t >> 2, ms = t & 3, mb = blockIdx.x * 4 + ms;     // meeting point: (output, slot)
  float* msave = (a.st.attsave && mb < B) ? a.st.attsave + (size_t)mb * a.st.attsave_ld : nullptr;      // (group 0)
  matvec(w1, pooledT);
  __syncthreads();
  TTICK(3);
  {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[((size_t)k * C + mo) * 4 + ms];
    v = relu_nan(v + a.st.att[0].p[1][mo]);
    hT[t] = v;
    if (msave) __builtin_nontemporal_store(v, msave + C + mo);
  }
  __syncthreads();
  TTICK(4);
  matvec(w2, hT);
  __syncthreads();
  TTICK(5);
  {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[((size_t)k * C + mo) * 4 + ms];
    const float gate = sigmoidf_(v + a.st.att[0].p[3][mo]);
    const float f = gate * pooledT[t];                     // mean_p(z * gate) = gate * mean_p(z)
    X[t] = f;
    if (msave) __builtin_nontemporal_store(gate, msave + 2 * C + mo);
    if (a.st.feat && mb < B) a.st.feat[(size_t)mb * C + mo] = f;
  }
  __syncthreads();
  TTICK(6);
  // ---- both last heads: out[slot][n] = sum_k X[k][slot] * Wt[k][n], k-slices of 64 inputs ----
  // The slice's 64 weight rows arrive in batches of TB loads, one batch ahead of the one being multiplied (left to itself
  // the compiler waits for every single load before issuing the next: 64 exposed L2 round trips, 19.6 of 32 us); the first
  // batch was requested at kernel entry.
  if (hks < TAIL_KS) {
    for (int n4 = htt; n4 * 4 < ldw; n4 += TAIL_TPS) {
      float acc[4][4];
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[s_][e] = 0.f;
      const float* wp = a.wt + (size_t)(hks * TAIL_KROWS) * ldw + n4 * 4;
      const float* xp = X + (size_t)(hks * TAIL_KROWS) * 4;
      if (n4 != htt) {      // (more than 4 * TAIL_TPS classes: the later class groups fetch their first batch here)
#pragma unroll
        for (int r = 0; r < TB; ++r) wb[0][r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + (size_t)r * ldw));
      }
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk) {
        if (blk + 1 < NBLK) {
#pragma unroll
          for (int r = 0; r < TB; ++r)
            wb[(blk + 1) & 1][r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + (size_t)((blk + 1) * TB + r) * ldw));
        }
        __builtin_amdgcn_sched_barrier(0);      // (the scheduler must not sink the batch's loads to their uses)
#pragma unroll
        for (int r = 0; r < TB; ++r) {
          const f32x4 w = wb[blk & 1][r];
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + (blk * TB + r) * 4);
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[s_][e] += w[e] * xv[s_];
        }
      }
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
        *reinterpret_cast<f32x4*>(red + ((size_t)hks * 4 + s_) * ldw + n4 * 4) = f32x4{acc[s_][0], acc[s_][1], acc[s_][2], acc[s_][3]};
    }
  }
  __syncthreads();
  TTICK(7);
  for (int i = t; i < 4 * ldw; i += 512) {
    const int s_ = i / ldw, n = i - s_ * ldw;
    if (n >= N) continue;
    float vs = red[((size_t)0 * 4 + s_) * ldw + n] + red[((size_t)1 * 4 + s_) * ldw + n];
    float vt = 0.f;
#pragma unroll
    for (int k = 2; k < TAIL_KS; ++k) vt += red[((size_t)k * 4 + s_) * ldw + n];
    vs += a.bias[0][n]; vt += a.bias[1][n];
    scS[(size_t)s_ * N + n] = vs; scT[(size_t)s_ * N + n] = vt;
    const int bb = blockIdx.x * 4 + s_;
    if (bb < B) { a.scores[0][(size_t)bb * N + n] = vs; a.scores[1][(size_t)bb * N + n] = vt; }
  }
  __syncthreads();
  TTICK(8);
  // ---- blend (+ loss): the four score rows are addressed as rows 4 blockIdx.x .. + 3 of a [B][N] matrix ----
  if (a.ce.labels) {
    blend_ce_body(cea, cesc, sd, is_last, &cpre);
  } else if (a.ce.joint) {      // (neither: the caller blends in its own loss launch, dta_net_loss)
    const double wd = 1.0 / (1.0 + exp(-a.alpha[0]));
    const float w = (float)wd, w1_ = (float)(1.0 - wd);
    for (int i = t; i < 4 * N; i += 512) {
      const int bb = blockIdx.x * 4 + i / N;
      if (bb < B) a.ce.joint[(size_t)blockIdx.x * 4 * N + i] = blend2(scS[i], scT[i], w, w1_);
    }
  }
  TTICK(15);
}

// The fused tail serves the third stage of a two-branch Hang2020 on 11x11 patches in training mode (coefficients from the
// BatchNorm finalize launch), with the lean kernels' storage formats.
bool tail_fwd_supported(const StageArgs& st3, int G, int classes) {
  if (G != 2 || st3.kind[0] != KIND_SPECTRAL || st3.kind[1] != KIND_SPATIAL) return false;
  if (st3.C != 128 || st3.Hc != 5 || st3.Wc != 5 || !st3.pool || !st3.apply_bn || !st3.relu || st3.bn_inkernel) return false;
  if (st3.att_k[1] != 3 || st3.att_pool[1] != 1 || st3.F[0] != 128 || st3.F[1] != 512) return false;
  if (!(st3.y_fmt == FMT_F32 || st3.y_fmt == FMT_F16) || !(st3.lean & 1)) return false;
  const int ldw = (classes + 3) / 4 * 4;
  return classes >= 1 && tail_lds_floats(ldw) * 4 <= 150 * 1024;
}

int launch_tail_fwd(const TailArgs& a_in, hipStream_t st) {
  TailArgs a = a_in;
  a.st.vslot = stage_vslot_for(a.st, 2);
  if (!tail_fwd_supported(a.st, 2, a.classes) || a.ldw != (a.classes + 3) / 4 * 4) { dta_set_error("tail_fwd: unsupported configuration"); return 1; }
  const size_t lds = tail_lds_floats(a.ldw) * 4;
  static DevOnce attr_once;
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)k_tail_fwd<FMT_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void*)k_tail_fwd<FMT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  }
  const dim3 grid((a.st.B + 3) / 4);
  if (a.st.y_fmt == FMT_F16) hipLaunchKernelGGL(k_tail_fwd<FMT_F16>, grid, dim3(512), lds, st, a);
  else hipLaunchKernelGGL(k_tail_fwd<FMT_F32>, grid, dim3(512), lds, st, a);
  DTA_CHECK_LAUNCH("k_tail_fwd");
  return 0;
}

}  // namespace dta
