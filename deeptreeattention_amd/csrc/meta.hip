// The site-metadata head of the reference's fusion model (src/models/metadata.py:9-44) around the Hang2020 scores:
//   meta   = ReLU(Linear_16->C(Dropout(BatchNorm1d(Embedding(site)))))                  (metadata.py:9-24)
//   out    = ReLU(Linear_2C->C(cat[meta, hsi_scores]))                                   (metadata.py:40-44)
// forward + backward as a handful of launches (the batched linears are the grouped GEMMs of heads.hip): issued as stock
// torch ops this <0.2 MFLOP-per-sample graph is ~35 launches = 0.17 ms per step, a third of the whole fused HSI branch.
//
// BatchNorm1d over the batch of embedding rows needs no pass over the batch: a row depends only on its site, so with
// n_s = #samples of site s the batch statistics are mean_f = sum_s n_s E[s][f] / B, var_f = sum_s n_s (E[s][f] - mean_f)^2 / B,
// and the backward's batch sums are sums over sites of per-site gradient sums.  Everything is evaluated in a fixed order
// (no float atomics): reruns give the same bits.
#include <atomic>
#include <string.h>

#include "../../include/dta_hip.h"
#include "kernels.h"

using namespace dta;

namespace {

constexpr int MW = 16;              // the site branch's width (metadata.py:12)
constexpr int MAX_SITES = 2048;     // LDS: histogram + 16-wide tables

template <typename T> inline T* at(void* ws, size_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(ws) + off); }
struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
};
struct MetaPlan { size_t x16, xhat_t, rstd, hist, joined, d_meta, d_x16, total; };
MetaPlan meta_plan(int B, int C, int S) {
  MetaPlan p;
  Carver c;
  p.x16 = c.take((size_t)B * MW * 4);            // the site branch's input to its Linear (after BN + dropout)
  p.xhat_t = c.take((size_t)S * MW * 4);         // normalised embedding row per site
  p.rstd = c.take(MW * 4);
  p.hist = c.take((size_t)S * 4);
  p.joined = c.take((size_t)B * 2 * C * 4);      // [ReLU(meta_pre) | hsi scores]
  p.d_meta = c.take((size_t)B * C * 4);
  p.d_x16 = c.take((size_t)B * MW * 4);
  p.total = c.off;
  return p;
}

struct FrontArgs {
  const long long* site; const float* emb; const float* bn_w; const float* bn_b; float* rm; float* rv; long long* nbt;
  const float* drop;      // [B][16] dropout factors (0 or 1 / (1 - p)) or null
  float* x16; float* xhat_t; float* rstd; int* hist;
  int B, S, training; float momentum, eps;
};

// one workgroup: site histogram -> batch statistics -> per-site normalised rows (the [B][16] input of the site Linear is
// formed from them by the join kernel below: a single workgroup streams 200 KB through one CU in 6 us)
// dynamic LDS: tab [S][16] floats (the embedding table, then the normalised rows), hist [S] ints
__global__ __launch_bounds__(1024) void k_meta_front(FrontArgs a) {
  extern __shared__ float tab[];
  __shared__ float mean[MW], rstd[MW];
  int* hist = reinterpret_cast<int*>(tab + a.S * MW);
  const int t = threadIdx.x;
  // (running statistics requested up front: the kernel is one dependent chain)
  float rm0 = 0.f, rv0 = 1.f;
  if (t < MW && a.rm) { rm0 = a.rm[t]; rv0 = a.rv[t]; }
#pragma unroll 4
  for (int i = t; i < a.S * MW; i += 1024) tab[i] = a.emb[i];
  for (int s = t; s < a.S; s += 1024) hist[s] = 0;
  __syncthreads();
  if (a.training) {
#pragma unroll 4
    for (int b = t; b < a.B; b += 1024) {
      const long long sb = a.site[b];
      if (sb >= 0 && sb < a.S) atomicAdd(&hist[(int)sb], 1);                      // integer: order-independent
    }
  }
  __syncthreads();
  if (t < MW) {
    float m, r;
    if (a.training) {
      // rows whose site index is out of range are not in the histogram (k_meta_join poisons them with NaN, so the loss says
      // so): the statistics are those of the rows that exist -- nrow of them, not B -- and the running buffers stay finite
      double s1 = 0;
      int nrow = 0;
#pragma unroll 8
      for (int s = 0; s < a.S; ++s) { s1 += (double)hist[s] * (double)tab[s * MW + t]; nrow += hist[s]; }
      const double cnt = nrow > 0 ? (double)nrow : 1.0;
      const double mu = s1 / cnt;
      double s2 = 0;
#pragma unroll 8
      for (int s = 0; s < a.S; ++s) { const double d = (double)tab[s * MW + t] - mu; s2 += (double)hist[s] * d * d; }
      const double var = s2 / cnt;
      m = (float)mu; r = 1.f / sqrtf((float)var + a.eps);      // (torch's invstd: float)
      if (a.rm) {
        const float unb = nrow > 1 ? (float)s2 / (float)(nrow - 1) : (float)var;
        a.rm[t] = (1.f - a.momentum) * rm0 + a.momentum * m;
        a.rv[t] = (1.f - a.momentum) * rv0 + a.momentum * unb;
        if (t == 0 && a.nbt) a.nbt[0] += 1;
      }
    } else {
      m = rm0; r = rsqrtf(rv0 + a.eps);
    }
    mean[t] = m; rstd[t] = r;
    a.rstd[t] = r;
  }
  __syncthreads();
  for (int i = t; i < a.S * MW; i += 1024) {
    const int f = i & (MW - 1);
    const float xh = (tab[i] - mean[f]) * rstd[f];
    tab[i] = xh;
    a.xhat_t[i] = xh;
  }
  if (a.training)
    for (int s = t; s < a.S; s += 1024) a.hist[s] = hist[s];
}

// joined[b] = [ReLU(x16[b] . mlp_w^T + mlp_b) | scores[b]] with x16[b] = (xhat[site_b] * gamma + beta) * dropout factors formed on
// the fly (and written out once per row for the backward): the 16 -> C linear of the site branch is 16 multiply-adds per
// output, done here instead of in a GEMM launch of its own
__global__ __launch_bounds__(256) void k_meta_join(const float* xhat_t, const float* bn_w, const float* bn_b, const long long* site,
                                                   const float* drop, const float* w, const float* bias, const float* scores,
                                                   float* joined, float* x16, int B, int C, int S) {
  // (a site index outside [0, sites) is a caller bug -- torch's Embedding device-asserts: the row's outputs become NaN so that
  //  it cannot train silently, and nothing is read out of bounds)
  const size_t n = (size_t)B * 2 * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / (2 * C)), c = (int)(i - (size_t)b * 2 * C);
    float v;
    if (c < C) {
      const long long sb = site[b];
      const bool okb = sb >= 0 && sb < S;
      const f32x4* xr = reinterpret_cast<const f32x4*>(xhat_t + (size_t)(okb ? sb : 0) * MW);      // (workspace rows: 64-byte aligned)
      const f32x4* dr = drop ? reinterpret_cast<const f32x4*>(drop + (size_t)b * MW) : nullptr;
      const float* wr = w + (size_t)c * MW;
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < MW / 4; ++q) {
        f32x4 xv = xr[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = xv[j] * bn_w[4 * q + j] + bn_b[4 * q + j];
        if (dr) { const f32x4 dv = dr[q]; xv *= dv; }
        if (c == 0) reinterpret_cast<f32x4*>(x16 + (size_t)b * MW)[q] = xv;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += xv[j] * wr[4 * q + j];
      }
      const float pre = acc + bias[c];
      v = okb ? (pre < 0.f ? 0.f : pre) : __builtin_nanf("");      // (NaN inputs stay NaN, as torch.relu)
    } else {
      v = scores[(size_t)b * C + (c - C)];
    }
    joined[i] = v;
  }
}

struct BackArgs {
  const long long* site; const float* bn_w; const float* drop; const float* d_x16; const float* xhat_t; const float* rstd; const int* hist;
  float* d_emb; float* d_bn_w; float* d_bn_b; int B, S, training;
};
// one workgroup: dropout backward, per-site sums of d(BN output), BatchNorm1d backward in closed form, embedding gradient.
// The per-site sums run over each site's samples IN BATCH ORDER (a fixed order, no atomics): a stable counting sort of the
// batch by site -- thread (site, part) counts, then lists, the samples of its site in the part-th sixteenth of the batch --
// and thread (site, feature) adds its site's listed rows.
// dynamic LDS: D [S][16] floats | off [S + 1] ints | cnt [S][16] ints | list [B] ints | sv [B] ints | dy [B][16] floats | xhat [S][16] floats
__global__ __launch_bounds__(1024) void k_meta_back(BackArgs a) {
  extern __shared__ float D[];
  __shared__ float dbeta[MW], dgamma[MW];
  const int t = threadIdx.x, S = a.S, B = a.B;
  int* off = reinterpret_cast<int*>(D + S * MW);
  int* cnt = off + S + 1;
  int* list = cnt + S * MW;
  int* sv = list + B;
  float* dy = reinterpret_cast<float*>(sv + B);
  // every global value this workgroup needs is requested here, in one go (the kernel is one dependent chain: each later
  // global load would add a round trip to it): sites, dy (x dropout), the normalised table, gamma / rstd of this thread's feature
  float* xh = dy + (size_t)B * MW;                         // [S][16] normalised embedding rows (LDS copy)
  const float gam = a.bn_w[t & (MW - 1)], rs = a.rstd[t & (MW - 1)];
#pragma unroll 4
  for (int i = t; i < S * MW; i += 1024) xh[i] = a.xhat_t[i];
#pragma unroll 4
  for (int b = t; b < B; b += 1024) sv[b] = (int)a.site[b];
  if (a.drop) {
#pragma unroll 16
    for (int i = t; i < B * MW; i += 1024) dy[i] = a.d_x16[i] * a.drop[i];
  } else {
#pragma unroll 16
    for (int i = t; i < B * MW; i += 1024) dy[i] = a.d_x16[i];
  }
  __syncthreads();
  const int per = (B + MW - 1) / MW;                       // samples per part
  // (every loop below reads LDS at addresses known in advance: unrolled so that the reads overlap -- a dependent LDS
  //  round trip per iteration made this one-workgroup kernel 25 us)
  // (lane -> site, so that the lanes of a wave read the SAME sv[b] -- a broadcast; with lane -> part the sixteen parts'
  //  addresses are 64 words apart: one bank, 16-way conflicts, 4.8 + 6.1 us for the two passes)
  for (int i = t; i < S * MW; i += 1024) {
    const int part = i / S, s = i - part * S, lo = part * per, hi = min(B, lo + per);
    int c = 0;
#pragma unroll 8
    for (int b = lo; b < hi; ++b) c += sv[b] == s;
    cnt[s * MW + part] = c;
  }
  __syncthreads();
  // site totals (thread s), then an inclusive scan over the sites (Hillis-Steele, ping-pong between off[] and list[])
  for (int s = t; s < S; s += 1024) {
    int tot = 0;
#pragma unroll
    for (int part = 0; part < MW; ++part) tot += cnt[s * MW + part];
    off[s + 1] = tot;
  }
  if (t == 0) off[0] = 0;
  __syncthreads();
  {
    int* src = off + 1;                                    // S totals
    int* tmp = list;                                       // (list[] is free until the fill pass; B >= 1 ints... S may exceed B:
    int* tmp2 = reinterpret_cast<int*>(D);                 //  the scan's second buffer is D[], S * 16 floats, free until the sums)
    (void)tmp;
    int* bufs[2] = {src, tmp2};
    int cur = 0;
    for (int d = 1; d < S; d <<= 1) {
      for (int s = t; s < S; s += 1024) bufs[cur ^ 1][s] = bufs[cur][s] + (s >= d ? bufs[cur][s - d] : 0);
      __syncthreads();
      cur ^= 1;
    }
    if (cur == 1) {
      for (int s = t; s < S; s += 1024) src[s] = tmp2[s];
      __syncthreads();
    }
  }
  for (int i = t; i < S * MW; i += 1024) {
    const int part = i / S, s = i - part * S, lo = part * per, hi = min(B, lo + per);
    int k = off[s];
#pragma unroll
    for (int q = 0; q < MW; ++q) k += q < part ? cnt[s * MW + q] : 0;
#pragma unroll 8
    for (int b = lo; b < hi; ++b)
      if (sv[b] == s) list[k++] = b;
  }
  __syncthreads();
  for (int i = t; i < S * MW; i += 1024) {
    const int s = i >> 4, f = i & (MW - 1);
    float acc = 0.f;
#pragma unroll 8
    for (int k = off[s]; k < off[s + 1]; ++k) acc += dy[list[k] * MW + f];
    D[i] = acc;
  }
  __syncthreads();
  if (t < MW) {
    float sb = 0.f, sg = 0.f;
#pragma unroll 8
    for (int s = 0; s < S; ++s) { sb += D[s * MW + t]; sg += D[s * MW + t] * xh[s * MW + t]; }
    dbeta[t] = sb; dgamma[t] = sg;
    if (a.d_bn_b) a.d_bn_b[t] = sb;
    if (a.d_bn_w) a.d_bn_w[t] = sg;
  }
  __syncthreads();
  if (!a.d_emb) return;
  for (int i = t; i < S * MW; i += 1024) {
    const int s = i >> 4;
    const float g = gam, r = rs;                            // (i = t + 1024 k: the feature is t mod 16 for every k)
    const int n_s = off[s + 1] - off[s];
    float v;
    if (a.training) {
      // d e_b = rstd (g dy_b - g mean_b(dy) - xhat_b g mean_b(dy xhat)), summed over the samples of site s
      const float m1 = g * dbeta[t & (MW - 1)] / B, m2 = g * dgamma[t & (MW - 1)] / B;
      v = r * (g * D[i] - n_s * m1 - n_s * xh[i] * m2);
    } else {
      v = r * g * D[i];
    }
    a.d_emb[i] = v;
  }
}

// Dynamic LDS a workgroup of the head's two one-workgroup kernels may use on the CURRENT device (cached per device ordinal,
// as DevOnce does): the larger of the device's reported per-workgroup limit and the 160 KB every gfx950 CU has (some ROCm
// stacks report 64 KB for hipDeviceAttributeMaxSharedMemoryPerBlock; hipFuncSetAttribute is what really decides, and the
// callers check its status), minus 4 KB for the kernels' static part.  0 = no device: every shape is then refused, and the
// Python side falls back.
size_t lds_budget() {
  static std::atomic<size_t> cap[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::atomic<size_t>& c = cap[dev & 63];
  size_t v = c.load(std::memory_order_relaxed);
  if (v == 0) {
    int q = 0;
    if (hipDeviceGetAttribute(&q, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || q <= 0) q = 0;
    const size_t lim = (size_t)q > (size_t)160 * 1024 ? (size_t)q : (size_t)160 * 1024;
    v = lim - 4096;
    c.store(v, std::memory_order_relaxed);
  }
  return v;
}
size_t meta_back_lds(int B, int S) { return ((size_t)S * MW + (S + 1) + (size_t)S * MW + B + B + (size_t)B * MW + (size_t)S * MW) * 4; }

int check(int B, int C, int S, const dta_meta_params* p, const long long* site, void* ws, const char* who) {
  if (B < 1 || C < 1 || S < 1 || !p || !site || !ws) { dta_set_error("%s: bad argument", who); return 1; }
  if (S > MAX_SITES) { dta_set_error("%s: at most %d sites", who, MAX_SITES); return 1; }
  if (meta_back_lds(B, S) > lds_budget()) { dta_set_error("%s: batch %d x %d sites needs %zu bytes of LDS, this device offers %zu per workgroup", who, B, S, meta_back_lds(B, S), lds_budget()); return 1; }
  if (!p->emb || !p->bn_w || !p->bn_b || !p->bn_rm || !p->bn_rv || !p->mlp_w || !p->mlp_b || !p->fc_w || !p->fc_b) {
    dta_set_error("%s: null parameter", who); return 1; }
  return 0;
}
int grid1d(size_t n) { return (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024); }

}  // namespace

extern "C" {

size_t dta_meta_head_workspace_bytes(int batch, int classes, int sites) {
  if (batch < 1 || classes < 1 || sites < 1 || sites > MAX_SITES || meta_back_lds(batch, sites) > lds_budget()) {
    dta_set_error("dta_meta_head_workspace_bytes: unsupported shape (sites <= %d; batch x sites needs %zu bytes of LDS, this device offers %zu per workgroup)",
                  MAX_SITES, (batch > 0 && sites > 0) ? meta_back_lds(batch, sites) : (size_t)0, lds_budget());
    return 0; }
  return meta_plan(batch, classes, sites).total;
}

int dta_meta_head_forward(int batch, int classes, int sites, int training, float momentum, float eps, const dta_meta_params* p,
                          const long long* site, const float* scores, const float* drop, void* workspace, float* out,
                          void* stream) {
  const int B = batch, C = classes, S = sites;
  if (check(B, C, S, p, site, workspace, "dta_meta_head_forward")) return 1;
  if (!scores || !out) { dta_set_error("dta_meta_head_forward: null scores / out"); return 1; }
  hipStream_t st = (hipStream_t)stream;
  const MetaPlan pl = meta_plan(B, C, S);
  FrontArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.site = site; fa.emb = p->emb; fa.bn_w = p->bn_w; fa.bn_b = p->bn_b; fa.rm = p->bn_rm; fa.rv = p->bn_rv; fa.nbt = p->bn_nbt;
  fa.drop = training ? drop : nullptr;
  fa.x16 = at<float>(workspace, pl.x16); fa.xhat_t = at<float>(workspace, pl.xhat_t); fa.rstd = at<float>(workspace, pl.rstd);
  fa.hist = at<int>(workspace, pl.hist);
  fa.B = B; fa.S = S; fa.training = training; fa.momentum = momentum; fa.eps = eps;
  static DevOnce front_once;
  if (front_once.first() && hipFuncSetAttribute((const void*)k_meta_front, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget()) != hipSuccess) {
    dta_set_error("dta_meta_head_forward: the device refused %zu bytes of dynamic LDS", lds_budget()); return 1; }
  hipLaunchKernelGGL(k_meta_front, dim3(1), dim3(1024), (size_t)S * (MW + 1) * 4, st, fa);
  DTA_CHECK_LAUNCH("k_meta_front");
  float* joined = at<float>(workspace, pl.joined);
  hipLaunchKernelGGL(k_meta_join, dim3(grid1d((size_t)B * 2 * C)), dim3(256), 0, st, fa.xhat_t, p->bn_w, p->bn_b, site, fa.drop,
                     p->mlp_w, p->mlp_b, scores, joined, fa.x16, B, C, S);
  DTA_CHECK_LAUNCH("k_meta_join");
  GemmArgs g;
  memset(&g, 0, sizeof(g));                      // out[B][C] = ReLU(joined[B][2C] . fc_w[C][2C]^T + fc_b)
  g.A = joined; g.sa_m = 2 * C; g.sa_k = 1; g.Bm = p->fc_w; g.sb_k = 1; g.sb_n = 2 * C;
  g.C = out; g.sc_m = C; g.sc_n = 1; g.bias = p->fc_b; g.M = B; g.N = C; g.K = 2 * C; g.ksplit = 1; g.relu = 1;
  if (launch_gemm(g, st)) return 1;
  return 0;
}

int dta_meta_head_loss(int batch, int classes, const float* out, const long long* labels, float* loss, float* dout, float* scratch,
                       void* stream) {
  if (batch < 1 || classes < 1 || !out || !labels || !loss || !scratch) { dta_set_error("dta_meta_head_loss: bad argument"); return 1; }
  BlendCeArgs a;
  a.spec = out; a.spat = nullptr; a.alpha = nullptr; a.joint = nullptr;
  a.labels = labels; a.weight = nullptr; a.dlogits = dout; a.loss = loss; a.rowtmp = scratch;
  a.B = batch; a.classes = classes; a.gscale = 1.f; a.relu_mask = 1;
  return launch_blend_ce(a, (hipStream_t)stream);
}

int dta_meta_head_backward(int batch, int classes, int sites, int training, const dta_meta_params* p, const long long* site,
                           const float* drop, void* workspace, const float* out, const float* dout, const dta_meta_grads* grads,
                           float* dscores, void* stream) {
  const int B = batch, C = classes, S = sites;
  if (check(B, C, S, p, site, workspace, "dta_meta_head_backward")) return 1;
  if (!dout || !grads || !dscores) { dta_set_error("dta_meta_head_backward: null argument"); return 1; }
  (void)out;
  hipStream_t st = (hipStream_t)stream;
  const MetaPlan pl = meta_plan(B, C, S);
  const float* d_pre = dout;                                // (dta_meta_head_loss already applied the last ReLU's backward)
  const float* joined = at<float>(workspace, pl.joined);
  float* d_meta = at<float>(workspace, pl.d_meta);
  float* d_x16 = at<float>(workspace, pl.d_x16);
  const float* x16 = at<float>(workspace, pl.x16);
  GemmGroup g1;
  GemmArgs g;
  memset(&g, 0, sizeof(g));                      // d(meta)[B][C] = d_pre[B][C] . fc_w[:, :C]
  g.A = d_pre; g.sa_m = C; g.sa_k = 1; g.Bm = p->fc_w; g.sb_k = 2 * C; g.sb_n = 1;
  g.C = d_meta; g.sc_m = C; g.sc_n = 1; g.M = B; g.N = C; g.K = C; g.ksplit = 1;
  g.mask = joined; g.mask_m = 2 * C;             // ... through the site branch's ReLU: joined[:, :C] holds its output
  g1.add(g);
  g.mask = nullptr; g.mask_m = 0;
  g.Bm = p->fc_w + C; g.C = dscores;             // d(hsi scores)[B][C] = d_pre . fc_w[:, C:]
  g1.add(g);
  GemmArgs gw;
  memset(&gw, 0, sizeof(gw));                    // d fc_w[C][2C] = d_pre^T[C][B] . joined[B][2C], d fc_b = column sums of d_pre
  gw.A = d_pre; gw.sa_m = 1; gw.sa_k = C; gw.Bm = joined; gw.sb_k = 2 * C; gw.sb_n = 1;
  gw.C = grads->fc_w; gw.sc_m = 2 * C; gw.sc_n = 1; gw.M = C; gw.N = 2 * C; gw.K = B; gw.ksplit = 1; gw.rowsum_out = grads->fc_b;
  // (launched with the second group: in this one its K = batch chain held back d(hsi scores), which the whole HSI backward
  //  waits for -- 0.606 -> 0.592 ms per step)
  if (launch_gemm_group(g1, st)) return 1;
  GemmGroup g2;
  memset(&g, 0, sizeof(g));                      // d x16[B][16] = d_meta[B][C] . mlp_w[C][16]
  g.A = d_meta; g.sa_m = C; g.sa_k = 1; g.Bm = p->mlp_w; g.sb_k = MW; g.sb_n = 1;
  g.C = d_x16; g.sc_m = MW; g.sc_n = 1; g.M = B; g.N = MW; g.K = C; g.ksplit = 1;
  g2.add(g);
  if (grads->mlp_w) {                            // d mlp_w[C][16] = d_meta^T . x16, d mlp_b = column sums of d_meta
    memset(&g, 0, sizeof(g));
    g.A = d_meta; g.sa_m = 1; g.sa_k = C; g.Bm = x16; g.sb_k = MW; g.sb_n = 1;
    g.C = grads->mlp_w; g.sc_m = MW; g.sc_n = 1; g.M = C; g.N = MW; g.K = B; g.ksplit = 1; g.rowsum_out = grads->mlp_b;
    g2.add(g);
  }
  if (grads->fc_w) g2.add(gw);
  if (launch_gemm_group(g2, st)) return 1;
  BackArgs ba;
  memset(&ba, 0, sizeof(ba));
  ba.site = site; ba.bn_w = p->bn_w; ba.drop = training ? drop : nullptr; ba.d_x16 = d_x16;
  ba.xhat_t = at<float>(workspace, pl.xhat_t); ba.rstd = at<float>(workspace, pl.rstd); ba.hist = at<int>(workspace, pl.hist);
  ba.d_emb = grads->emb; ba.d_bn_w = grads->bn_w; ba.d_bn_b = grads->bn_b; ba.B = B; ba.S = S; ba.training = training;
  const size_t lds = meta_back_lds(B, S);
  static DevOnce attr_once;
  if (attr_once.first() && hipFuncSetAttribute((const void*)k_meta_back, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget()) != hipSuccess) {
    dta_set_error("dta_meta_head_backward: the device refused %zu bytes of dynamic LDS", lds_budget()); return 1; }
  hipLaunchKernelGGL(k_meta_back, dim3(1), dim3(1024), lds, st, ba);
  DTA_CHECK_LAUNCH("k_meta_back");
  return 0;
}

}  // extern "C"
