// 3x3 'same' convolution family for the Hang2020 hot path on gfx950 (MI355X):
//   k_pack_input   NCHW fp32 patches  -> tile layout (TL, see common.h)
//   k_pack_conv_w  torch conv weights -> [chunk][tap][n][16] MFMA B-operand image (fwd or dgrad form)
//   k_conv3x3      implicit-GEMM forward / input-gradient conv on the matrix cores (+bias, +BN partials)
//   k_conv_wgrad   weight-gradient conv (K = batch x pixels) + k_wgrad_reduce (split-K reduce, torch layout)
// Replaces nn.Conv2d(3x3, padding="same") forward/backward of /root/reference/src/models/Hang2020.py:18,25
// (conv_module), i.e. the ops torch dispatches to MIOpen/oneDNN in the reference.
#include <stdlib.h>

#include "kernels.h"

namespace dta {

// ------------------------------------------------------------------------------------------------
// pack input
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void pack_input_block(const float* __restrict__ x, T* __restrict__ out, int B, int C, int H,
                                                 int W, int NC, int CG, int b, int ycg, unsigned char* smem,
                                                 bool compact = false) {
  const int HW = H * W, Q = (H + 2) * (W + 2);
  int* lut = (int*)smem;                   // [Q] pixel index of haloed-grid row q, or -1 on the halo
  float* sbase = (float*)(lut + ((Q + 3) & ~3));   // 16-byte aligned
  const int chunk0 = ycg * CG;
  const int nch = min(CG, NC - chunk0);
  const int c0 = chunk0 * 16;
  const int creal = max(0, min(nch * 16, C - c0));
  const float* src = x + ((size_t)b * C + c0) * HW;
  for (int q = threadIdx.x; q < Q; q += 256) {
    int hh = q / (W + 2) - 1, ww = q % (W + 2) - 1;
    lut[q] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? hh * W + ww : -1;
  }
  // The patch's channel planes are one contiguous run of floats whose 16-byte phase depends on (b, c0): shift the LDS
  // image by the same phase so that the body can move as aligned float4 (global AND LDS), with a scalar head/tail.
  const size_t e0 = ((size_t)b * C + c0) * HW;
  const int phase = (int)(e0 & 3);
  float* s = sbase + phase;                // s[i] <-> src[i]; (s + i) is 16-byte aligned exactly when (src + i) is
  const int nreal = creal * HW, ntot = nch * 16 * HW;
  const int head = min(nreal, (4 - phase) & 3);
  const int nvec = (nreal - head) / 4;
  if (threadIdx.x < head) s[threadIdx.x] = src[threadIdx.x];
  for (int k = threadIdx.x; k < nvec; k += 256)
    *reinterpret_cast<float4*>(s + head + 4 * k) = *reinterpret_cast<const float4*>(src + head + 4 * k);
  for (int i = head + 4 * nvec + threadIdx.x; i < ntot; i += 256) s[i] = (i < nreal) ? src[i] : 0.f;
  __syncthreads();
  constexpr int VW = TlVec<T>::VW, PARTS = 16 / VW;
  if (compact) {
    // halo-free tiles [patch][chunk][pixel][16] (bf16 network input only: the conv kernels that read them re-insert
    // the zero halo while staging into LDS, so 28 % fewer bytes are written here and read twice later)
    for (int ch = 0; ch < nch; ++ch) {
      T* dst = out + ((size_t)b * NC + chunk0 + ch) * HW * 16;
      const float* sc = s + ch * 16 * HW;
      for (int r = threadIdx.x; r < HW * PARTS; r += 256) {
        const int p = r / PARTS, part = r % PARTS;
        float v[VW];
#pragma unroll
        for (int j = 0; j < VW; ++j) v[j] = sc[(part * VW + j) * HW + p];
        tl_store_vec(dst + (size_t)p * 16, part, v);
      }
    }
    return;
  }
  for (int ch = 0; ch < nch; ++ch) {
    T* dst = out + ((size_t)b * NC + chunk0 + ch) * Q * 16;
    const float* sc = s + ch * 16 * HW;
    for (int r = threadIdx.x; r < Q * PARTS; r += 256) {
      int q = r / PARTS, part = r % PARTS;
      int p = lut[q];
      float v[VW];
#pragma unroll
      for (int j = 0; j < VW; ++j) v[j] = p >= 0 ? sc[tl_pos<T>(q, part * VW + j) * HW + p] : 0.f;
      tl_store_vec(dst + (size_t)q * 16, part, v);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_pack_input(const float* __restrict__ x, T* __restrict__ out, int B, int C,
                                                    int H, int W, int NC, int CG) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  pack_input_block<T>(x, out, B, C, H, W, NC, CG, blockIdx.x, blockIdx.y, smem);
}

static int pack_input_plan(int C, int H, int W, int* NC, int* CG, size_t* lds) {
  const int HW = H * W;
  *NC = (C + 15) / 16;
  *CG = 2;   // 2 chunks (32 channels) per workgroup measured best on MI355X (more resident workgroups)
  while (*CG > 1 && (size_t)*CG * 16 * HW * 4 > 65536) *CG >>= 1;
  *lds = (size_t)*CG * 16 * HW * 4 + (size_t)(((H + 2) * (W + 2) + 3) & ~3) * 4 + 16;
  if (*lds > 160 * 1024) { dta_set_error("pack_input: %dx%d patch does not fit LDS", H, W); return 1; }
  return 0;
}

template <typename T>
int launch_pack_input(const float* x, void* out, int B, int C, int H, int W, hipStream_t st) {
  int NC, CG; size_t lds;
  if (pack_input_plan(C, H, W, &NC, &CG, &lds)) return 1;
  dim3 grid(B, (NC + CG - 1) / CG);
  hipLaunchKernelGGL(k_pack_input<T>, grid, dim3(256), lds, st, x, (T*)out, B, C, H, W, NC, CG);
  DTA_CHECK_LAUNCH("k_pack_input");
  return 0;
}
template int launch_pack_input<float>(const float*, void*, int, int, int, int, hipStream_t);
template int launch_pack_input<bf16_t>(const float*, void*, int, int, int, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
// pack weights: dst[g][chunk][tap][n][16], row = tap*N+n swizzled like TL rows.
//   mode 0: forward, one source per group                 val = W_g[n][kc][tap]
//   mode 1: forward, G==1, output columns concatenated     val = (n<nsplit ? W_0[n] : W_1[n-nsplit])[kc][tap]
//   mode 2: input-gradient form (transposed + flipped)     val = W_g[kc][n][8-tap]   (W_g is [Kdim][N][9])
// ------------------------------------------------------------------------------------------------
// (an item is one 16-wide row of the image -- (group, chunk, tap, n) decoded ONCE with 32-bit arithmetic, its sixteen values
//  gathered with all loads in flight and written as one contiguous 32- / 64-byte run; the element-per-thread form decoded every
//  element with four 64-bit divisions: 33 us of a 15-network multi-stage step's prep launch)
template <typename T>
__device__ __forceinline__ void pack_conv_w_job(const PackWArgs& a, T* __restrict__ dst, size_t i0, size_t stride) {
  const int N = a.N, NC = a.NC;
  const unsigned rows = (unsigned)a.G * NC * 9 * N;
  for (unsigned r0 = (unsigned)i0; r0 < rows; r0 += (unsigned)stride) {
    unsigned r = r0;
    const int n = r % (unsigned)N; r /= (unsigned)N;
    const int tap = r % 9u; r /= 9u;
    const int chunk = r % (unsigned)NC;
    const int g = r / (unsigned)NC;
    const int row = tap * N + n;
    const float* w;
    size_t base, kstride;      // value of contraction channel kc: w[base + kc * kstride]
    if (a.mode == 0) { w = a.src[g]; base = (size_t)n * a.K * 9 + tap; kstride = 9; }
    else if (a.mode == 1) { w = n < a.nsplit ? a.src[0] : a.src[1]; base = (size_t)(n < a.nsplit ? n : n - a.nsplit) * a.K * 9 + tap; kstride = 9; }
    else { w = a.src[g]; base = (size_t)n * 9 + (8 - tap); kstride = (size_t)N * 9; }
    float v[16];
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
      const int kc = chunk * 16 + tl_pos<T>(row, pos);
      v[pos] = kc < a.K ? w[base + (size_t)kc * kstride] : 0.f;
    }
    T* o = dst + (size_t)r0 * 16;
    if constexpr (sizeof(T) == 2) {
      u32x4 lo, hi;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo[j] = (unsigned)Cvt<T>::to(v[2 * j]) | ((unsigned)Cvt<T>::to(v[2 * j + 1]) << 16);
        hi[j] = (unsigned)Cvt<T>::to(v[8 + 2 * j]) | ((unsigned)Cvt<T>::to(v[8 + 2 * j + 1]) << 16);
      }
      reinterpret_cast<u32x4*>(o)[0] = lo; reinterpret_cast<u32x4*>(o)[1] = hi;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const f32x4 q = {v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]}; reinterpret_cast<f32x4*>(o)[j] = q; }
    }
  }
}

// several packing jobs in one launch: blockIdx.y selects the job
template <typename T>
__global__ void k_pack_conv_w(PackWGroup gr) {
  const PackWArgs& a = gr.job[blockIdx.y];
  pack_conv_w_job<T>(a, (T*)gr.dst[blockIdx.y], blockIdx.x * (size_t)blockDim.x + threadIdx.x,
                     (size_t)gridDim.x * blockDim.x);
}

template <typename T>
int launch_pack_conv_w_group(const PackWGroup& gr, hipStream_t st) {
  if (gr.n == 0) return 0;
  hipLaunchKernelGGL(k_pack_conv_w<T>, dim3(256, gr.n), dim3(256), 0, st, gr);
  DTA_CHECK_LAUNCH("k_pack_conv_w");
  return 0;
}
template int launch_pack_conv_w_group<float>(const PackWGroup&, hipStream_t);
template int launch_pack_conv_w_group<bf16_t>(const PackWGroup&, hipStream_t);

template <typename T>
int launch_pack_conv_w(const PackWArgs& a, void* dst, hipStream_t st) {
  PackWGroup gr;
  gr.n = 1; gr.job[0] = a; gr.dst[0] = dst;
  return launch_pack_conv_w_group<T>(gr, st);
}
template int launch_pack_conv_w<float>(const PackWArgs&, void*, hipStream_t);
template int launch_pack_conv_w<bf16_t>(const PackWArgs&, void*, hipStream_t);

// ------------------------------------------------------------------------------------------------
// Everything the forward needs before its first conv, in ONE launch (each of these is a few microseconds of work
// and none depends on another): input patches -> tiles, conv weights -> MFMA operand images, spectral-attention
// centre taps -> dense matrices, and the clearing of the split-K targets.  blockIdx.y selects the job.
// ------------------------------------------------------------------------------------------------
// The row tables every FULL workgroup of a bf16 conv launch shares (kernels.h: conv_row_tables; rows relative to the
// workgroup's first patch), built once per forward instead of by each of the launch's 200-1000 workgroups.
template <int ROWS>
__device__ __forceinline__ void conv_tab_rows(const ConvTabJob& j, int* tab_s) {
  ConvArgs c;      // (geometry fields only)
  c.HW = j.HW; c.W = j.W; c.Q = j.Q; c.spp = 1; c.ppw = j.ppw; c.pixel_order = j.order;
  int* rowtab = tab_s; int* plq = tab_s + ROWS; int* hist = plq + ROWS; int* flag = hist + 256;
  conv_row_tables<ROWS, 256>(c, rowtab, plq, hist, flag, 0, j.ppw, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * ROWS; i += 256) j.dst[i] = tab_s[i];
}
__device__ __forceinline__ void conv_tab_job(const ConvTabJob& j) {
  __shared__ int tab_s[2 * 576 + 256 + 4];
  if (j.rows == 256) conv_tab_rows<256>(j, tab_s);
  else if (j.rows == 512) conv_tab_rows<512>(j, tab_s);
  else if (j.rows == 576) conv_tab_rows<576>(j, tab_s);
}

template <typename T>
__global__ __launch_bounds__(256) void k_forward_prep(PrepArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int y = blockIdx.y;
  if (a.tabs.n > 0) {      // row tables of the step's conv launches: ONE grid row, block x = job (a row per job would add a
    if (y == 0) {          // thousand empty workgroups per job: +3.5 us measured), and the FIRST row: a one-block job is a
      if ((int)blockIdx.x < a.tabs.n) conv_tab_job(a.tabs.job[blockIdx.x]);   // ~3 us chain that must start when the launch starts,
      return;              // not behind the dispatch of 13 k other workgroups (last row: the launch ran 11 us instead of 8)
    }
    y -= 1;
  }
  if (y < a.ncg * a.nx) {
    const int gi = y / a.ncg;
    if ((int)blockIdx.x >= a.B) return;      // (the grid may be wider than the batch: see launch_forward_prep)
    pack_input_block<T>(a.x[gi], (T*)((char*)a.x_tl + (size_t)gi * a.x_tl_gs), a.B, a.C, a.H, a.W, a.NC, a.CG, blockIdx.x,
                        y - gi * a.ncg, smem, a.x_compact != 0);
    return;
  }
  y -= a.ncg * a.nx;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x, tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (y < a.packs.n) { pack_conv_w_job<T>(a.packs.job[y], (T*)a.packs.dst[y], tid, nthreads); return; }
  y -= a.packs.n;
  if (y < a.spacks.n) { pack_spectral_att_job(a.spacks, y, tid, nthreads); return; }
  y -= a.spacks.n;
  if (y < a.trans.n) { transpose_job(a.trans, y, tid, nthreads); return; }
  y -= a.trans.n;
  if (a.zero) {
    float4* z = (float4*)a.zero;   // workspace regions are 256-byte aligned and padded
    for (size_t i = tid; i < a.zero_n4; i += nthreads) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <typename T>
int launch_forward_prep(PrepArgs a, hipStream_t st) {
  size_t lds;
  if (pack_input_plan(a.C, a.H, a.W, &a.NC, &a.CG, &lds)) return 1;
  a.ncg = (a.NC + a.CG - 1) / a.CG;
  constexpr int PREP_MIN_BLOCKS = 1024;
  // grid x extent: every job is a grid-stride loop, and what the launch costs is mostly the DISPATCH of its workgroups (13 rows x
  // 1024 blocks for a Hang2020 step whose largest job is 0.3 M elements: 9.4 us; x 256: 7.3; x 128: 8.7; x 64: 12.8) -- so
  // about four elements per thread of the largest job, between 256 and 1024 blocks; the input-pack rows need one block per patch
  size_t most = a.zero ? a.zero_n4 : 0;
  for (int j = 0; j < a.packs.n; ++j) { const PackWArgs& w = a.packs.job[j]; const size_t e = (size_t)w.G * w.NC * 9 * w.N * 4; if (e > most) most = e; }      // (row items: 16 elements each, weighed as 4)
  for (int j = 0; j < a.spacks.n; ++j) { const size_t e = (size_t)a.spacks.C[j] * a.spacks.C[j]; if (e > most) most = e; }
  for (int j = 0; j < a.trans.n; ++j) { const size_t e = (size_t)a.trans.cols[j] * a.trans.ld[j]; if (e > most) most = e; }
  int gx = (int)((most + 1023) / 1024);
  gx = gx < 256 ? 256 : (gx > PREP_MIN_BLOCKS ? PREP_MIN_BLOCKS : gx);
  if (a.nx > 0 && gx < a.B) gx = a.B;
  dim3 grid(gx, a.ncg * a.nx + a.packs.n + a.spacks.n + a.trans.n + (a.tabs.n > 0 ? 1 : 0) + (a.zero ? 1 : 0));
  hipLaunchKernelGGL(k_forward_prep<T>, grid, dim3(256), lds, st, a);
  DTA_CHECK_LAUNCH("k_forward_prep");
  return 0;
}
template int launch_forward_prep<float>(PrepArgs, hipStream_t);
template int launch_forward_prep<bf16_t>(PrepArgs, hipStream_t);

// ------------------------------------------------------------------------------------------------
// MFMA fragment helpers.  A 32x32 output tile per MFMA; K step = 2 (fp32, exact) or 16 (bf16).
// ------------------------------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<float> {
  static constexpr int KS = 2;
  typedef float reg;
  // element offset inside a 16-wide TL row for k-step ks (0..7): one float per lane, k = 2*ks + lane/32
  __device__ static __forceinline__ reg load(const float* rowbase, int row, int ks, int lane) {
    return rowbase[tl_pos<float>(row, ks * 2 + (lane >> 5))];
  }
  __device__ static __forceinline__ f32x16 mfma(reg a, reg b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};
template <> struct Frag<bf16_t> {
  static constexpr int KS = 16;
  typedef bf16x8 reg;
  __device__ static __forceinline__ reg load(const bf16_t* rowbase, int row, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(rowbase + tl_pos<bf16_t>(row, (lane >> 5) << 3));
  }
  __device__ static __forceinline__ f32x16 mfma(reg a, reg b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// ------------------------------------------------------------------------------------------------
// k_conv3x3: implicit GEMM.  M = (patch, pixel) rows, N = output channels, K = 9 taps x Cin.
// One workgroup = 4 waves; wave w owns MT consecutive 32-row tiles x all NT 32-column tiles.
// Per 16-channel chunk the haloed input tiles of the workgroup's patches and the [9][N][16] weight
// slab are copied linearly HBM->LDS; the 9 taps are 9 row-shifted reads of the same LDS tile.
// ------------------------------------------------------------------------------------------------
template <typename T, int MT, int NT>
__global__ __launch_bounds__(256, 2) void k_conv3x3(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MWG = 4 * MT * 32;
  constexpr int N = NT * 32;
  int* rowtab = (int*)smem;            // [MWG] global output row or -1
  int* plq = rowtab + MWG;             // [MWG] (pl << 16) | q_topleft
  float* red = (float*)(plq + MWG);    // [4][N]
  float* cmean = red + 4 * N;          // [N]
  T* sx = (T*)(cmean + N);
  const int Q = a.Q, HW = a.HW, W2 = a.W + 2;
  T* sw = sx + (size_t)a.ppw * Q * 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.y;
  const int pg = blockIdx.x / a.spp, split = blockIdx.x - pg * a.spp;
  const int b0 = pg * a.ppw;
  const int npatch = min(a.ppw, a.B - b0);

  for (int lr = tid; lr < MWG; lr += 256) {
    int pl, pix;
    bool valid;
    if (a.spp == 1) { pl = lr / HW; pix = lr - pl * HW; valid = pl < npatch; }
    else { pl = 0; pix = split * MWG + lr; valid = pix < HW; }
    int h = valid ? pix / a.W : 0, w = valid ? pix - h * a.W : 0;
    rowtab[lr] = valid ? (b0 + pl) * HW + pix : -1;
    plq[lr] = valid ? ((pl << 16) | (h * W2 + w)) : 0;
  }
  __syncthreads();

  int base[MT], ql[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int v = plq[(wave * MT + mt) * 32 + (lane & 31)];
    base[mt] = (v >> 16) * Q * 16;
    ql[mt] = v & 0xFFFF;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int vpp = Q * 16 * (int)sizeof(T) / 16;        // 16-byte vectors per patch tile
  const int wvec = 9 * N * 16 * (int)sizeof(T) / 16;   // 16-byte vectors of the weight slab
  const T* xg = (const T*)a.x_tl + (size_t)g * a.x_gs;
  const T* wg = (const T*)a.wp + (size_t)g * a.NC * 9 * N * 16;

  // Staging: 16-byte vectors, thread t owns vectors t, t+256, ...  With PIPE the next chunk's vectors are fetched
  // into registers while the current chunk is being multiplied (global latency hidden behind the MFMAs).
  constexpr bool PIPE = sizeof(T) == 2;
  constexpr int XV = PIPE ? 6 : 1, WV = PIPE ? (9 * N * 2 + 255) / 256 : 1;
  u32x4 rx[XV], rw[WV];
  const int nxv = npatch * vpp;
#define DTA_CONV_FETCH(chunk_)                                                                              \
  {                                                                                                         \
    _Pragma("unroll") for (int u = 0; u < XV; ++u) {                                                        \
      int v = min(tid + u * 256, nxv - 1);                                                                  \
      int pl = v / vpp, o = v - pl * vpp;                                                                   \
      rx[u] = reinterpret_cast<const u32x4*>(xg + (((size_t)(b0 + pl) * a.NC + (chunk_)) * Q) * 16)[o];     \
    }                                                                                                       \
    const u32x4* swp_ = reinterpret_cast<const u32x4*>(wg + (size_t)(chunk_) * 9 * N * 16);                 \
    _Pragma("unroll") for (int u = 0; u < WV; ++u) {                                                        \
      int v = min(tid + u * 256, wvec - 1);                                                                 \
      rw[u] = swp_[v];                                                                                      \
    }                                                                                                       \
  }
  const bool pipe = PIPE && nxv <= XV * 256;
  if (pipe) DTA_CONV_FETCH(0)
  for (int chunk = 0; chunk < a.NC; ++chunk) {
    __syncthreads();
    if (pipe) {
      u32x4* d = reinterpret_cast<u32x4*>(sx);
#pragma unroll
      for (int u = 0; u < XV; ++u) { int v = tid + u * 256; if (v < nxv) d[v] = rx[u]; }
      u32x4* dw = reinterpret_cast<u32x4*>(sw);
#pragma unroll
      for (int u = 0; u < WV; ++u) { int v = tid + u * 256; if (v < wvec) dw[v] = rw[u]; }
    } else {
      u32x4* d = reinterpret_cast<u32x4*>(sx);
      for (int v = tid; v < nxv; v += 256) {
        int pl = v / vpp, o = v - pl * vpp;
        const u32x4* s = reinterpret_cast<const u32x4*>(xg + (((size_t)(b0 + pl) * a.NC + chunk) * Q) * 16);
        d[v] = s[o];
      }
      u32x4* dw = reinterpret_cast<u32x4*>(sw);
      const u32x4* swp = reinterpret_cast<const u32x4*>(wg + (size_t)chunk * 9 * N * 16);
      for (int v = tid; v < wvec; v += 256) dw[v] = swp[v];
    }
    __syncthreads();
    if (pipe && chunk + 1 < a.NC) DTA_CONV_FETCH(chunk + 1)
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = (tap / 3) * W2 + (tap % 3);
      const T* wrow[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wrow[nt] = sw + (size_t)(tap * N + nt * 32 + (lane & 31)) * 16;
      const T* xrow[MT];
      int xq[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { xq[mt] = ql[mt] + toff; xrow[mt] = sx + base[mt] + xq[mt] * 16; }
#pragma unroll
      for (int ks = 0; ks < 16 / Frag<T>::KS; ++ks) {
        typename Frag<T>::reg af[MT], bf[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = Frag<T>::load(xrow[mt], xq[mt], ks, lane);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = Frag<T>::load(wrow[nt], lane & 31, ks, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = Frag<T>::mfma(af[mt], bf[nt], acc[mt][nt]);
      }
    }
  }

  // ---- epilogue: bias, store, per-workgroup (mean, M2) per column for BatchNorm batch statistics ----
  float bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int n = nt * 32 + (lane & 31);
    float bv = 0.f;
    if (a.bias[0]) {
      if (a.bias_mode == 1) bv = n < a.bias_split ? a.bias[0][n] : a.bias[1][n - a.bias_split];
      else bv = a.bias[g][n];
    }
    bias[nt] = bv;
  }
  float csum[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) csum[nt] = 0.f;
  float* yg = a.y + (size_t)g * a.y_gs;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (wave * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      int orow = rowtab[lr];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float v = acc[mt][nt][r] + bias[nt];
        acc[mt][nt][r] = v;
        if (orow >= 0) {
          yg[(size_t)orow * a.y_rs + nt * 32 + (lane & 31)] = v;
          csum[nt] += v;
        }
      }
    }
  }
  if (a.stats == nullptr) return;
  const int cnt = (a.spp == 1) ? npatch * HW : min(MWG, HW - split * MWG);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float v = csum[nt] + __shfl_xor(csum[nt], 32);
    if (lane < 32) red[wave * N + nt * 32 + lane] = v;
  }
  __syncthreads();
  if (tid < N) cmean[tid] = (red[tid] + red[N + tid] + red[2 * N + tid] + red[3 * N + tid]) / (float)cnt;
  __syncthreads();
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float mu = cmean[nt * 32 + (lane & 31)];
    float m2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int lr = (wave * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (rowtab[lr] >= 0) { float d = acc[mt][nt][r] - mu; m2 += d * d; }
      }
    csum[nt] = m2 + __shfl_xor(m2, 32);
  }
  __syncthreads();
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    if (lane < 32) red[wave * N + nt * 32 + lane] = csum[nt];
  __syncthreads();
  if (tid < N) {
    float m2 = red[tid] + red[N + tid] + red[2 * N + tid] + red[3 * N + tid];
    float* o = a.stats + (((size_t)g * gridDim.x + blockIdx.x) * N + tid) * 2;
    if (a.fan_count) fan_store2(o, cmean[tid], m2);
    else { o[0] = cmean[tid]; o[1] = m2; }
  }
  if (a.fan_count) conv_stats_fanin<256>(a, g, N, HW, MWG, reinterpret_cast<double*>(sx), reinterpret_cast<int*>(cmean));
}

void conv_geometry(int HW, int MWG, int B, int* ppw, int* spp, int* nwg) {
  if (HW <= MWG) { *ppw = MWG / HW; *spp = 1; *nwg = (B + *ppw - 1) / *ppw; }
  else { *ppw = 1; *spp = (HW + MWG - 1) / MWG; *nwg = B * *spp; }
}

template <typename T, int MT, int NT>
static int launch_conv_t(ConvArgs a, int G, hipStream_t st) {
  constexpr int MWG = 4 * MT * 32, N = NT * 32;
  int nwg;
  conv_geometry(a.HW, MWG, a.B, &a.ppw, &a.spp, &nwg);
  size_t lds = (size_t)MWG * 8 + (size_t)5 * N * 4 + ((size_t)a.ppw * a.Q * 16 + (size_t)9 * N * 16) * sizeof(T);
  if (lds > 160 * 1024) { dta_set_error("conv3x3: LDS need %zu B exceeds 160 KiB (H=%d W=%d)", lds, a.H, a.W); return 1; }
  static DevOnce attr_once;      // (function attributes are per device)
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)k_conv3x3<T, MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((k_conv3x3<T, MT, NT>), dim3(nwg, G), dim3(256), lds, st, a);
  DTA_CHECK_LAUNCH("k_conv3x3");
  return 0;
}

int conv_mwg(int N) { return N >= 128 ? 256 : 512; }

template <typename T>
int launch_conv3x3(const ConvArgs& a, int G, hipStream_t st) {
  switch (a.N) {
    case 32: return launch_conv_t<T, 4, 1>(a, G, st);
    case 64: return launch_conv_t<T, 4, 2>(a, G, st);
    case 128: return launch_conv_t<T, 2, 4>(a, G, st);
  }
  dta_set_error("conv3x3: unsupported output width %d", a.N);
  return 1;
}
template int launch_conv3x3<float>(const ConvArgs&, int, hipStream_t);   // bf16: conv_bf16.hip

// ------------------------------------------------------------------------------------------------
// k_conv_wgrad: dW[tap][c][n] = sum_{b,q} X[b][c][q + shift(tap)] * dY[b][n][q], K runs over the haloed
// grid rows q in [W+3, Q-W-3) (dY halo rows are zero, so the side-halo rows contribute nothing).
// Workgroup = (channel group of CT*32 inputs, batch split s, group g); wave w -> (c-tile, n-tile), 9 taps.
// fp32: element reads + 32x32x2 MFMA.  bf16: ds_read_b64_tr_b16 transposing reads + 32x32x16 MFMA.
// ------------------------------------------------------------------------------------------------
template <typename T> struct WFrag;
template <> struct WFrag<float> {
  static constexpr int KS = 2;
  typedef float reg;
  // operand element (column col of a 32-wide tile spanning two 16-chunks, haloed-grid row q0 + lane/32)
  __device__ static __forceinline__ reg load(const float* tile0, int rows_per_chunk, int q0, int lane) {
    int q = q0 + (lane >> 5);
    int col = lane & 31;
    return tile0[((size_t)(col >> 4) * rows_per_chunk + q) * 16 + tl_pos<float>(q, col & 15)];
  }
  __device__ static __forceinline__ f32x16 mfma(reg a, reg b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};
template <> struct WFrag<bf16_t> {
  static constexpr int KS = 16;
  typedef bf16x8 reg;
  __device__ static __forceinline__ reg load(const bf16_t* tile0, int rows_per_chunk, int q0, int lane) {
    // lane i of 16-lane group gq supplies the address of 4 consecutive channels (col group i&3) of row
    // q0 + 8*(gq>>1) + 4*half + (i>>2) in chunk (gq&1); it receives channel i of those 4 rows.
    const int gq = lane >> 4, i = lane & 15;
    bf16x8 out;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      int q = q0 + 8 * (gq >> 1) + 4 * half + (i >> 2);
      const bf16_t* p = tile0 + ((size_t)(gq & 1) * rows_per_chunk + q) * 16 + tl_pos<bf16_t>(q, (i & 3) * 4);
      s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
      out[half * 4 + 0] = v[0]; out[half * 4 + 1] = v[1]; out[half * 4 + 2] = v[2]; out[half * 4 + 3] = v[3];
    }
    return out;
  }
  __device__ static __forceinline__ f32x16 mfma(reg a, reg b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <typename T, int NTT>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CT = 4 / NTT;          // c-tiles (32 input channels each) per workgroup
  constexpr int N = NTT * 32;
  constexpr int XCH = CT * 2, YCH = NTT * 2;
  // K is walked in bands of a.bl haloed-grid rows (a multiple of 16, so the row swizzle of the linear LDS copy is
  // band independent); the LDS window of a band holds tile rows [band*bl, band*bl + WR).  See conv_bf16.hip.
  const int Q = a.Q, WR = a.wr, W2 = a.W + 2;
  T* sx = (T*)smem;                    // [XCH][WR][16]
  T* sy = sx + (size_t)XCH * WR * 16;  // [YCH][WR][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = blockIdx.x, s = blockIdx.y, g = blockIdx.z;
  const int ct = wave / NTT, nt = wave % NTT;
  const int chunk0 = cg * XCH;
  const int nxch = max(0, min(XCH, a.NCx - chunk0));

  {  // zero everything once: absent chunks stay zero for the whole kernel
    u32x4 z = {0, 0, 0, 0};
    u32x4* d = reinterpret_cast<u32x4*>(smem);
    int tot = (XCH + YCH) * WR * 16 * (int)sizeof(T) / 16;
    for (int v = tid; v < tot; v += 256) d[v] = z;
  }
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  constexpr int VPR = 16 * (int)sizeof(T) / 16;   // 16-byte vectors per row
  const int vpc = WR * VPR;                       // vectors per chunk window
  const T* xg = (const T*)a.x_tl + (size_t)g * a.x_gs;
  const T* yg = (const T*)a.dy_tl + (size_t)g * a.dy_gs;
  const int q0 = a.W + 3, q1 = Q - a.W - 3;
  const T* xt = sx + (size_t)ct * 2 * WR * 16;
  const T* yt = sy + (size_t)nt * 2 * WR * 16;
  const u32x4 zero4 = {0, 0, 0, 0};

  for (int b = s; b < a.B; b += a.S) {
    for (int band = 0; band < a.nbands; ++band) {
      const int r0 = band * a.bl;
      __syncthreads();
      for (int v = tid; v < nxch * vpc; v += 256) {
        int ch = v / vpc, o = v - ch * vpc;
        const bool in = r0 + o / VPR < Q;
        reinterpret_cast<u32x4*>(sx + (size_t)ch * WR * 16)[o] =
            in ? reinterpret_cast<const u32x4*>(xg + (((size_t)b * a.NCx + chunk0 + ch) * Q + r0) * 16)[o] : zero4;
      }
      for (int v = tid; v < YCH * vpc; v += 256) {
        int ch = v / vpc, o = v - ch * vpc;
        const bool in = r0 + o / VPR < Q;
        reinterpret_cast<u32x4*>(sy + (size_t)ch * WR * 16)[o] =
            in ? reinterpret_cast<const u32x4*>(yg + (((size_t)b * a.NCy + a.ych0 + ch) * Q + r0) * 16)[o] : zero4;
      }
      __syncthreads();
      const int qend = q0 + min(a.bl, (q1 - q0) - r0);
#pragma unroll 1
      for (int q = q0; q < qend; q += WFrag<T>::KS) {
        typename WFrag<T>::reg bf = WFrag<T>::load(yt, WR, q, lane);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int shift = (tap / 3 - 1) * W2 + (tap % 3 - 1);
          typename WFrag<T>::reg af = WFrag<T>::load(xt, WR, q + shift, lane);
          acc[tap] = WFrag<T>::mfma(af, bf, acc[tap]);
        }
      }
    }
  }
  // partial[g][tap][c][s][n]: the S partial sums of one output row are contiguous for the reduction
  float* out = a.partial + (size_t)g * 9 * a.Cpad * a.S * N + (size_t)s * N;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int c = cg * CT * 32 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (c < a.Cpad) out[((size_t)tap * a.Cpad + c) * a.S * N + nt * 32 + (lane & 31)] = acc[tap][r];
    }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(WgradReduceArgs a) { wgrad_reduce_blocks(a, blockIdx.x, gridDim.x); }
__global__ void k_slot_copy(const double* src, float* dst) { dst[0] = (float)src[0]; }
__global__ __launch_bounds__(256) void k_wgrad_reduce_group(WgradReduceGroup gr) {
  if (gr.slot_dst && blockIdx.x == 0 && threadIdx.x == 0) gr.slot_dst[0] = (float)gr.slot_src[0];
  int j = 0;
  while (j + 1 < gr.n && (int)blockIdx.x >= gr.start[j + 1]) ++j;
  wgrad_reduce_blocks(gr.job[j], blockIdx.x - gr.start[j], gr.start[j + 1] - gr.start[j]);
}

int wgrad_cpw(int N) { return (4 / (N / 32)) * 32; }
// 128-column layers (bf16): two 64-column groups per (channel group, slab).  A workgroup's partial tile is written once
// per slab whatever its width, so halving the tile and the slab count halves the split-K traffic (37.7 -> 18.9 MB written
// and read back for the third conv) at the price of each X tile being staged by two workgroups.
int wgrad_ngroups(int N, int bf16) { return (bf16 && N == 128) ? 2 : 1; }

void wgrad_band_plan(int Q, int W, int wr_max, int* bl, int* wr, int* nbands) {
  const int span = Q - 2 * (W + 3);              // haloed-grid rows q0..q1
  const int halo = 2 * (W + 3);
  int b = (span + 15) / 16 * 16;
  auto align = [](int r) { while ((r & 7) != 4) ++r; return r; };
  while (b >= 16 && align(b + halo) > wr_max) b -= 16;
  *bl = b;
  *wr = b >= 16 ? align(b + halo) : 0;
  *nbands = b >= 16 ? (span + b - 1) / b : 0;
}

template <typename T, int NTT>
static int launch_wgrad_t(const WgradArgs& a, int G, int cgroups, hipStream_t st) {
  constexpr int CT = 4 / NTT, NCH = CT * 2 + NTT * 2;
  WgradArgs a2 = a;
  int wr_max = (int)(160 * 1024 / ((size_t)NCH * 16 * sizeof(T)));
  if (wr_max > 1024) wr_max = 1024;
  wgrad_band_plan(a.Q, a.W, wr_max, &a2.bl, &a2.wr, &a2.nbands);
  if (a2.bl < 16) { dta_set_error("conv_wgrad: %dx%d patch is too wide for the band plan", a.H, a.W); return 1; }
  size_t lds = (size_t)NCH * a2.wr * 16 * sizeof(T);
  static DevOnce attr_once;      // (function attributes are per device)
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)k_conv_wgrad<T, NTT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((k_conv_wgrad<T, NTT>), dim3(cgroups, a.S, G), dim3(256), lds, st, a2);
  DTA_CHECK_LAUNCH("k_conv_wgrad");
  return 0;
}

template <typename T>
int launch_conv_wgrad(const WgradArgs& a, int G, hipStream_t st) {
  int cpw = wgrad_cpw(a.N);
  int cgroups = (a.Cpad + cpw - 1) / cpw;
  switch (a.N) {
    case 32: return launch_wgrad_t<T, 1>(a, G, cgroups, st);
    case 64: return launch_wgrad_t<T, 2>(a, G, cgroups, st);
    case 128: return launch_wgrad_t<T, 4>(a, G, cgroups, st);
  }
  dta_set_error("conv_wgrad: unsupported width %d", a.N);
  return 1;
}
template int launch_conv_wgrad<float>(const WgradArgs&, int, hipStream_t);   // bf16: conv_bf16.hip

int wgrad_reduce_nblocks(const WgradReduceArgs& a) {
  size_t total = (size_t)a.G * 9 * a.C * (a.N / 4) * 4;   // four lanes per item
  return (int)min((size_t)8192, (total + 255) / 256);
}
int launch_wgrad_reduce(const WgradReduceArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(wgrad_reduce_nblocks(a)), dim3(256), 0, st, a);
  DTA_CHECK_LAUNCH("k_wgrad_reduce");
  return 0;
}
int launch_wgrad_reduce_group(WgradReduceGroup& gr, hipStream_t st) {
  if (gr.n == 0) {
    if (gr.slot_dst) { hipLaunchKernelGGL(k_slot_copy, dim3(1), dim3(1), 0, st, gr.slot_src, gr.slot_dst); DTA_CHECK_LAUNCH("k_slot_copy"); }
    return 0;
  }
  if (gr.n == 1 && !gr.slot_dst) return launch_wgrad_reduce(gr.job[0], st);
  int total = 0;
  for (int j = 0; j < gr.n; ++j) { gr.start[j] = total; total += wgrad_reduce_nblocks(gr.job[j]); }
  gr.start[gr.n] = total;
  hipLaunchKernelGGL(k_wgrad_reduce_group, dim3(total), dim3(256), 0, st, gr);
  DTA_CHECK_LAUNCH("k_wgrad_reduce_group");
  return 0;
}

}  // namespace dta
