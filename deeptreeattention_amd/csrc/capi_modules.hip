// Module-level C-ABI entry points: the reference's building blocks used stand-alone (conv_module, spectral_attention,
// spatial_attention, Classifier), composed from the same kernels as the network-level path in capi.hip.
#include <string.h>

#include "../../include/dta_hip.h"
#include "kernels.h"

using namespace dta;

namespace {

struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
};
template <typename T> inline T* at(void* ws, size_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(ws) + off); }

int spectral_k(int C) { return C == 32 ? 3 : C == 64 ? 5 : C == 128 ? 7 : 0; }   // Hang2020.py:136-141
int spatial_k(int C) { return C == 32 ? 7 : C == 64 ? 5 : C == 128 ? 3 : 0; }    // :77-85
int spatial_pool(int C) { return C == 32 ? 4 : C == 64 ? 2 : C == 128 ? 1 : 0; } // :91-99

// ---- conv_module ---------------------------------------------------------------------------------
struct CmPlan {
  int B, Cin, N, H, W, HW, Q, NC, Hz, Wz, esz, MWG, nwg, S, cgroups, Cpad;
  size_t x_tl, wp, y, stats, coef, dv, bnpart, bcoef, dy_tl, wd, wpart, total;
};
int cm_plan(const dta_conv_module_desc* d, CmPlan* p) {
  memset(p, 0, sizeof(*p));
  if (d->batch < 1 || d->in_channels < 1 || d->height < 1 || d->width < 1) { dta_set_error("conv_module: bad descriptor"); return 1; }
  if (d->filters != 32 && d->filters != 64 && d->filters != 128) {
    dta_set_error("conv_module: filters must be 32, 64 or 128 (got %d)", d->filters); return 1; }
  if ((d->height + 2) * (d->width + 2) > 65535) { dta_set_error("conv_module: patch too large"); return 1; }
  p->B = d->batch; p->Cin = d->in_channels; p->N = d->filters; p->H = d->height; p->W = d->width;
  p->HW = p->H * p->W; p->Q = (p->H + 2) * (p->W + 2); p->NC = (p->Cin + 15) / 16;
  p->Hz = d->pool ? p->H / 2 : p->H; p->Wz = d->pool ? p->W / 2 : p->W;
  if (p->Hz < 1 || p->Wz < 1) { dta_set_error("conv_module: map too small to pool"); return 1; }
  p->esz = d->dtype == DTA_BF16 ? 2 : 4;
  p->MWG = conv_mwg(p->N);
  int ppw, spp;
  conv_geometry(p->HW, p->MWG, p->B, &ppw, &spp, &p->nwg);
  int cpw = wgrad_cpw(p->N);
  p->Cpad = p->NC * 16; p->cgroups = (p->Cpad + cpw - 1) / cpw;
  int target = d->dtype == DTA_BF16 ? 256 : 512;
  int S = target / p->cgroups;
  if (S >= 8) S &= ~7;
  p->S = S < 1 ? 1 : (S > p->B ? p->B : S);
  Carver c;
  const size_t e = p->esz;
  p->x_tl = c.take((size_t)p->B * p->NC * p->Q * 16 * e);
  p->wp = c.take((size_t)p->NC * 9 * p->N * 16 * e);
  p->y = c.take((size_t)p->B * p->HW * p->N * 4);
  p->stats = c.take((size_t)p->nwg * p->N * 2 * 4);
  p->coef = c.take((size_t)p->N * 16);
  p->dv = c.take((size_t)p->B * p->HW * p->N * 4);
  p->bnpart = c.take((size_t)p->B * p->N * 8);
  p->bcoef = c.take((size_t)p->N * 16);
  p->dy_tl = c.take((size_t)p->B * (p->N / 16) * p->Q * 16 * e);
  p->wd = c.take((size_t)(p->N / 16) * 9 * 128 * 16 * e);
  p->wpart = c.take((size_t)p->S * 9 * p->Cpad * p->N * 4);
  p->total = c.off;
  return 0;
}

StageArgs cm_stage(const CmPlan& p, const dta_conv_module_desc* d, void* ws) {
  StageArgs s;
  memset(&s, 0, sizeof(s));
  s.kind[0] = KIND_PLAIN;
  s.y = at<float>(ws, p.y); s.y_gs = 0; s.y_rs = p.N;
  s.coef = at<float>(ws, p.coef); s.coef_gs = p.N * 4;
  s.apply_bn = 1; s.relu = 1; s.pool = d->pool ? 1 : 0;
  s.B = p.B; s.C = p.N; s.Hc = p.H; s.Wc = p.W;
  return s;
}

template <typename T>
int cm_forward(const CmPlan& p, const dta_conv_module_desc* d, const float* conv_w, const float* conv_b, const float* bn_w,
               const float* bn_b, float* rm, float* rv, long long* nbt, const float* x, void* ws, float* out, hipStream_t st) {
  if (launch_pack_input<T>(x, at<char>(ws, p.x_tl), p.B, p.Cin, p.H, p.W, st)) return 1;
  PackWArgs pw;
  memset(&pw, 0, sizeof(pw));
  pw.G = 1; pw.NC = p.NC; pw.N = p.N; pw.K = p.Cin; pw.src[0] = conv_w; pw.mode = 0;
  if (launch_pack_conv_w<T>(pw, at<char>(ws, p.wp), st)) return 1;
  ConvArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.x_tl = at<char>(ws, p.x_tl); ca.wp = at<char>(ws, p.wp); ca.bias[0] = conv_b;
  ca.y = at<float>(ws, p.y); ca.y_rs = p.N;
  ca.stats = d->training ? at<float>(ws, p.stats) : nullptr;
  ca.B = p.B; ca.H = p.H; ca.W = p.W; ca.NC = p.NC; ca.N = p.N; ca.Q = p.Q; ca.HW = p.HW;
  if (launch_conv3x3<T>(ca, 1, st)) return 1;
  BnFinalizeArgs bf;
  memset(&bf, 0, sizeof(bf));
  bf.stats = at<float>(ws, p.stats); bf.nwg = p.nwg; bf.N = p.N; bf.HW = p.HW; bf.MWG = p.MWG; bf.B = p.B;
  bf.gamma[0] = bn_w; bf.beta[0] = bn_b; bf.rmean[0] = rm; bf.rvar[0] = rv; bf.nbt[0] = nbt;
  bf.coef = at<float>(ws, p.coef); bf.training = d->training; bf.momentum = d->bn_momentum; bf.eps = d->bn_eps;
  if (launch_bn_finalize(bf, 1, st)) return 1;
  StageArgs sa = cm_stage(p, d, ws);
  sa.a_nchw = out;
  return launch_stage_fwd<T>(sa, 1, st);
}

template <typename T>
int cm_backward(const CmPlan& p, const dta_conv_module_desc* d, const float* conv_w, const float* bn_w, void* ws,
                const float* dout, float* dx_nhwc, float* g_conv_w, float* g_conv_b, float* g_bn_w, float* g_bn_b,
                hipStream_t st) {
  StageBwdArgs sb;
  memset(&sb, 0, sizeof(sb));
  sb.f = cm_stage(p, d, ws);
  sb.da_nchw = dout;
  sb.dv = at<float>(ws, p.dv);
  sb.bnpart = at<float>(ws, p.bnpart);
  if (launch_stage_bwd(sb, 1, st)) return 1;
  BnBwdFinalizeArgs bf;
  memset(&bf, 0, sizeof(bf));
  bf.bnpart = sb.bnpart; bf.B = p.B; bf.C = p.N; bf.HW = p.HW;
  bf.coef = sb.f.coef; bf.coef_gs = sb.f.coef_gs; bf.gamma[0] = bn_w;
  bf.dgamma[0] = g_bn_w; bf.dbeta[0] = g_bn_b; bf.dconvbias[0] = g_conv_b;
  bf.bcoef = at<float>(ws, p.bcoef); bf.bcoef_gs = p.N * 4; bf.training = d->training;
  if (launch_bn_bwd_finalize(bf, 1, st)) return 1;
  BnBwdApplyArgs ap;
  memset(&ap, 0, sizeof(ap));
  ap.dv = sb.dv; ap.y = sb.f.y; ap.y_rs = p.N; ap.coef = sb.f.coef; ap.coef_gs = sb.f.coef_gs;
  ap.bcoef = bf.bcoef; ap.bcoef_gs = bf.bcoef_gs;
  ap.B = p.B; ap.C = p.N; ap.H = p.H; ap.W = p.W;
  ap.dy_tl = at<char>(ws, p.dy_tl); ap.dy_nc = p.N / 16;
  if (launch_bn_bwd_apply<T>(ap, 1, st)) return 1;
  if (g_conv_w) {
    WgradArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.x_tl = at<char>(ws, p.x_tl); wa.NCx = p.NC; wa.dy_tl = ap.dy_tl; wa.NCy = p.N / 16;
    wa.partial = at<float>(ws, p.wpart);
    wa.B = p.B; wa.H = p.H; wa.W = p.W; wa.Q = p.Q; wa.N = p.N; wa.Cpad = p.Cpad; wa.S = p.S;
    if (launch_conv_wgrad<T>(wa, 1, st)) return 1;
    WgradReduceArgs wr;
    memset(&wr, 0, sizeof(wr));
    wr.partial = wa.partial; wr.G = 1; wr.S = p.S; wr.N = p.N; wr.C = p.Cin; wr.Cpad = p.Cpad; wr.dst[0] = g_conv_w;
    if (launch_wgrad_reduce(wr, st)) return 1;
  }
  if (dx_nhwc) {
    if (p.Cin != 32 && p.Cin != 64 && p.Cin != 128) {
      dta_set_error("conv_module backward: input gradient needs in_channels in {32,64,128} (got %d)", p.Cin); return 1; }
    PackWArgs pw;
    memset(&pw, 0, sizeof(pw));
    pw.G = 1; pw.NC = p.N / 16; pw.N = p.Cin; pw.K = p.N; pw.mode = 2; pw.src[0] = conv_w;
    if (launch_pack_conv_w<T>(pw, at<char>(ws, p.wd), st)) return 1;
    ConvArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.x_tl = ap.dy_tl; ca.wp = at<char>(ws, p.wd); ca.y = dx_nhwc; ca.y_rs = p.Cin;
    ca.B = p.B; ca.H = p.H; ca.W = p.W; ca.NC = p.N / 16; ca.N = p.Cin; ca.Q = p.Q; ca.HW = p.HW;
    if (launch_conv3x3<T>(ca, 1, st)) return 1;
  }
  return 0;
}

// ---- attention modules ---------------------------------------------------------------------------
struct AtPlan { int B, C, H, W, HW, kind, K, F, vec_ld; size_t packed, vec, total; };
int at_plan(const dta_attention_desc* d, AtPlan* p) {
  memset(p, 0, sizeof(*p));
  p->B = d->batch; p->C = d->filters; p->H = d->height; p->W = d->width; p->HW = p->H * p->W;
  if (p->B < 1 || p->H < 1 || p->W < 1) { dta_set_error("attention: bad descriptor"); return 1; }
  if (d->kind == 0) {
    p->kind = KIND_SPECTRAL; p->K = spectral_k(p->C); p->F = p->C; p->vec_ld = 4 * p->C;
  } else {
    p->kind = KIND_SPATIAL; p->K = spatial_k(p->C);
    int ps = spatial_pool(p->C);
    if (ps && (p->H / ps < 1 || p->W / ps < 1)) { dta_set_error("spatial_attention: %dx%d map smaller than its %d-pool", p->H, p->W, ps); return 1; }
    p->F = ps ? p->C * (p->H / ps) * (p->W / ps) : 0;
    p->vec_ld = p->C + 2 * p->K * p->K + 3;
  }
  if (!p->K) { dta_set_error("Unknown incoming kernel size %d for attention layers", p->C); return 1; }
  Carver c;
  p->packed = c.take((size_t)4 * p->C * p->C * 4);
  p->vec = c.take((size_t)p->B * p->vec_ld * 4);
  p->total = c.off;
  return 0;
}
StageArgs at_stage(const AtPlan& p, const float* const* prm, const float* x_nhwc, void* ws) {
  StageArgs s;
  memset(&s, 0, sizeof(s));
  s.kind[0] = p.kind; s.F[0] = p.F;
  s.y = x_nhwc; s.y_rs = p.C;
  s.apply_bn = 0; s.relu = 0; s.pool = 0;
  s.B = p.B; s.C = p.C; s.Hc = p.H; s.Wc = p.W;
  if (p.kind == KIND_SPECTRAL) {
    float* pk = at<float>(ws, p.packed);
    s.att[0].p[0] = pk; s.att[0].p[1] = prm[1]; s.att[0].p[2] = pk + p.C * p.C; s.att[0].p[3] = prm[3];
    s.att[0].p[4] = pk + 2 * p.C * p.C; s.att[0].p[5] = pk + 3 * p.C * p.C;
  } else {
    for (int i = 0; i < 6; ++i) s.att[0].p[i] = prm[i];
    s.att_k[0] = p.K; s.att_pool[0] = spatial_pool(p.C);
  }
  return s;
}

}  // namespace

extern "C" {

size_t dta_conv_module_workspace_bytes(const dta_conv_module_desc* d) {
  CmPlan p;
  if (!d || cm_plan(d, &p)) return 0;
  return p.total;
}

int dta_conv_module_forward(const dta_conv_module_desc* d, const float* conv_w, const float* conv_b, const float* bn_w,
                            const float* bn_b, float* bn_rm, float* bn_rv, long long* bn_nbt, const float* x,
                            void* workspace, float* out, void* stream) {
  CmPlan p;
  if (!d || !conv_w || !bn_w || !bn_b || !bn_rm || !bn_rv || !x || !workspace || !out) { dta_set_error("dta_conv_module_forward: null argument"); return 1; }
  if (cm_plan(d, &p)) return 1;
  hipStream_t st = (hipStream_t)stream;
  return d->dtype == DTA_BF16 ? cm_forward<bf16_t>(p, d, conv_w, conv_b, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, x, workspace, out, st)
                              : cm_forward<float>(p, d, conv_w, conv_b, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, x, workspace, out, st);
}

int dta_conv_module_backward(const dta_conv_module_desc* d, const float* conv_w, const float* bn_w, void* workspace,
                             const float* dout, float* dx_nhwc, float* g_conv_w, float* g_conv_b, float* g_bn_w,
                             float* g_bn_b, void* stream) {
  CmPlan p;
  if (!d || !conv_w || !bn_w || !workspace || !dout) { dta_set_error("dta_conv_module_backward: null argument"); return 1; }
  if (cm_plan(d, &p)) return 1;
  hipStream_t st = (hipStream_t)stream;
  return d->dtype == DTA_BF16 ? cm_backward<bf16_t>(p, d, conv_w, bn_w, workspace, dout, dx_nhwc, g_conv_w, g_conv_b, g_bn_w, g_bn_b, st)
                              : cm_backward<float>(p, d, conv_w, bn_w, workspace, dout, dx_nhwc, g_conv_w, g_conv_b, g_bn_w, g_bn_b, st);
}

size_t dta_attention_workspace_bytes(const dta_attention_desc* d) {
  AtPlan p;
  if (!d || at_plan(d, &p)) return 0;
  return p.total;
}

int dta_attention_forward(const dta_attention_desc* d, const float* const params[6], const float* x_nhwc, void* workspace,
                          float* out_nchw, float* feat, void* stream) {
  AtPlan p;
  if (!d || !params || !x_nhwc || !workspace || !out_nchw) { dta_set_error("dta_attention_forward: null argument"); return 1; }
  if (at_plan(d, &p)) return 1;
  hipStream_t st = (hipStream_t)stream;
  if (p.kind == KIND_SPECTRAL)
    if (launch_pack_spectral_att(params[0], params[2], p.C, p.K, at<float>(workspace, p.packed), st)) return 1;
  StageArgs sa = at_stage(p, params, x_nhwc, workspace);
  sa.a_nchw = out_nchw; sa.feat = feat; sa.feat_gs = 0;
  return launch_stage_fwd<float>(sa, 1, st);
}

int dta_attention_backward(const dta_attention_desc* d, const float* const params[6], const float* x_nhwc, void* workspace,
                           const float* dout_nchw, const float* dfeat, float* dx_nhwc, float* const grads[6], void* stream) {
  AtPlan p;
  if (!d || !params || !x_nhwc || !workspace || !dx_nhwc || !grads) { dta_set_error("dta_attention_backward: null argument"); return 1; }
  if (at_plan(d, &p)) return 1;
  hipStream_t st = (hipStream_t)stream;
  StageBwdArgs sb;
  memset(&sb, 0, sizeof(sb));
  sb.f = at_stage(p, params, x_nhwc, workspace);   // packed spectral matrices are still in the workspace
  sb.da_nchw = dout_nchw; sb.dfeat = dfeat;
  sb.dv = dx_nhwc;
  sb.vec = at<float>(workspace, p.vec); sb.vec_ld = p.vec_ld;
  if (launch_stage_bwd(sb, 1, st)) return 1;
  const float* vec = sb.vec;
  const int C = p.C, B = p.B, ld = p.vec_ld;
  if (p.kind == KIND_SPECTRAL) {
    GemmGroup grp;
    for (int which = 0; which < 2; ++which) {
      float* gw = grads[which * 2];
      if (!gw) continue;
      GemmArgs ga;
      memset(&ga, 0, sizeof(ga));
      ga.A = vec + (which ? 0 : 2 * C); ga.sa_m = 1; ga.sa_k = ld;
      ga.Bm = vec + (which ? C : 3 * C); ga.sb_k = ld; ga.sb_n = 1;
      ga.C = gw + p.K / 2; ga.sc_m = (long)C * p.K; ga.sc_n = p.K;
      ga.M = C; ga.N = C; ga.K = B; ga.ksplit = gemm_auto_ksplit(C, C, B);
      ga.rowsum_out = grads[which * 2 + 1];
      grp.add(ga);
    }
    return launch_gemm_group(grp, st);
  }
  const int kk = p.K * p.K;
  ColsumArgs cs;
  memset(&cs, 0, sizeof(cs));
  cs.A = vec; cs.rows = B; cs.cols = C + 2 * kk + 3; cs.lda = ld; cs.nseg = 6;
  int offs[6] = {0, C, C + 1, C + 1 + kk, C + 2 + kk, C + 2 + 2 * kk};
  int lens[6] = {C, 1, kk, 1, kk, 1};
  for (int i = 0; i < 6; ++i) { cs.off[i] = offs[i]; cs.len[i] = lens[i]; cs.dst[i] = grads[i]; cs.dst_stride[i] = 1; }
  return launch_colsum_scatter(cs, st);
}

int dta_linear_forward(const float* x, const float* w, const float* b, int batch, int in_features, int out_features,
                       float* out, void* stream) {
  if (!x || !w || !out || batch < 1 || in_features < 1 || out_features < 1) { dta_set_error("dta_linear_forward: bad argument"); return 1; }
  GemmArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.A = x; ga.sa_m = in_features; ga.sa_k = 1;
  ga.Bm = w; ga.sb_k = 1; ga.sb_n = in_features;
  ga.C = out; ga.sc_m = out_features; ga.sc_n = 1; ga.bias = b;
  ga.M = batch; ga.N = out_features; ga.K = in_features; ga.ksplit = gemm_auto_ksplit(batch, out_features, in_features);
  if (ga.ksplit > 1) hipMemsetAsync(out, 0, (size_t)batch * out_features * 4, (hipStream_t)stream);
  return launch_gemm(ga, (hipStream_t)stream);
}

int dta_linear_backward(const float* x, const float* w, const float* dout, int batch, int in_features, int out_features,
                        float* dx, float* gw, float* gb, void* stream) {
  if (!x || !w || !dout || batch < 1) { dta_set_error("dta_linear_backward: bad argument"); return 1; }
  hipStream_t st = (hipStream_t)stream;
  GemmArgs ga;
  if (dx) {
    memset(&ga, 0, sizeof(ga));
    ga.A = dout; ga.sa_m = out_features; ga.sa_k = 1;
    ga.Bm = w; ga.sb_k = in_features; ga.sb_n = 1;
    ga.C = dx; ga.sc_m = in_features; ga.sc_n = 1;
    ga.M = batch; ga.N = in_features; ga.K = out_features; ga.ksplit = gemm_auto_ksplit(batch, in_features, out_features);
    if (ga.ksplit > 1) hipMemsetAsync(dx, 0, (size_t)batch * in_features * 4, st);
    if (launch_gemm(ga, st)) return 1;
  }
  if (gw) {   // gw / gb arrive zero-filled
    memset(&ga, 0, sizeof(ga));
    ga.A = dout; ga.sa_m = 1; ga.sa_k = out_features;
    ga.Bm = x; ga.sb_k = in_features; ga.sb_n = 1;
    ga.C = gw; ga.sc_m = in_features; ga.sc_n = 1;
    ga.M = out_features; ga.N = in_features; ga.K = batch; ga.ksplit = gemm_auto_ksplit(out_features, in_features, batch);
    ga.rowsum_out = gb;
    if (launch_gemm(ga, st)) return 1;
  }
  return 0;
}

}  // extern "C"
