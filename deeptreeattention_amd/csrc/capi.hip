// C-ABI entry points (include/dta_hip.h) and the launch orchestration of the Hang2020 hot path.
// Host code only decides buffer carving and launch order; all arithmetic is in conv.hip / stage.hip / heads.hip.
#include <math.h>
#include <stdlib.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dta_hip.h"
#include "kernels.h"
#include "xchg_dev.h"

using namespace dta;

// xchg.hip (internal): arguments of the head segment's overlapped reduce-scatter for the exchange's NEXT launch / undo
extern "C" {
int dta_xchg_side_args(dta_xchg* x, const double* alpha_g, long long alpha_slot, dta::XchgArgs* out);
void dta_xchg_side_cancel(dta_xchg* x);
}

static thread_local char g_err[512] = "";
void dta_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

// lean stage kernels in use: bit 0 = forward stages, bits 1..3 = backward of the 32 / 64 / 128-channel stage
constexpr int DTA_LEAN_DEFAULT = 15;

// Optional HIP-event timing of one launch site (bench.py's roofline leg): events are recorded on the same
// stream as the kernel, immediately before and after its launch.  Host-side state only; off by default.
struct Prof {
  static constexpr int SLOTS = 4, CAP = 512;
  int site[SLOTS] = {-1, -1, -1, -1}, n[SLOTS] = {0, 0, 0, 0}, nslots = 0;
  int stride = 1, seen[SLOTS] = {0, 0, 0, 0};      // every stride-th launch of a site is timed (an event pair costs ~5 us of stream time)
  bool armed[SLOTS] = {false, false, false, false};
  bool created = false;
  hipEvent_t ev[SLOTS][CAP][2];
} g_prof;
inline int prof_slot(int site) {
  for (int s = 0; s < g_prof.nslots; ++s)
    if (g_prof.site[s] == site) return s;
  return -1;
}
inline void prof_begin(int site, hipStream_t st) {
  if (g_prof.nslots == 0) return;
  const int s = prof_slot(site);
  if (s < 0) return;
  g_prof.armed[s] = (g_prof.seen[s]++ % g_prof.stride) == 0 && g_prof.n[s] < Prof::CAP;
  if (g_prof.armed[s]) hipEventRecord(g_prof.ev[s][g_prof.n[s]][0], st);
}
inline void prof_end(int site, hipStream_t st) {
  if (g_prof.nslots == 0) return;
  const int s = prof_slot(site);
  if (s >= 0 && g_prof.armed[s]) { hipEventRecord(g_prof.ev[s][g_prof.n[s]][1], st); ++g_prof.n[s]; g_prof.armed[s] = false; }
}

// Developer switches (same-box A/B runs; developer library only, common.h: dev_getenv): the environment is read once
// per process, not per call.  In the product library every switch has its default and nothing reads the environment.
#define DTA_LEAD_DEFAULT 0      // (measured: -1 us per step for two launches fewer, profiles/README.md round 5 -- an experiment, off)
struct Switches { bool no_fused_input, no_tail_merge, bn_inkernel, fp32_act, no_lean; int lean_mask; bool halo_tiles, wgrad_nsplit; int conv_order; bool no_wgrad_pair; int fanin; bool no_tail; int lead; };
inline Switches read_switches() {
  // these change launch plans (and, for DTA_FP32_ACT, roundings): never meant for a training job's environment, so say
  // so once, loudly, when one is set
  static const char* names[] = {"DTA_NO_FUSED_INPUT", "DTA_NO_TAIL_MERGE", "DTA_BN_INKERNEL", "DTA_FP32_ACT", "DTA_NO_LEAN",
                                "DTA_LEAN_MASK", "DTA_HALO_TILES", "DTA_NO_WGRAD_NSPLIT", "DTA_NO_WGRAD_STACK", "DTA_NO_WGRAD_D2", "DTA_PIXEL_ORDER", "DTA_NO_STAGGER", "DTA_NO_WGRAD_PAIR", "DTA_FANIN", "DTA_NO_TAIL", "DTA_LEAD"};
  for (const char* n : names)
    if (dev_getenv(n)) fprintf(stderr, "[libdta_hip] developer switch %s is set: kernel plans (and possibly roundings) differ from the default build\n", n);
  return {dev_getenv("DTA_NO_FUSED_INPUT") != nullptr, dev_getenv("DTA_NO_TAIL_MERGE") != nullptr, dev_getenv("DTA_BN_INKERNEL") != nullptr,
          dev_getenv("DTA_FP32_ACT") != nullptr, dev_getenv("DTA_NO_LEAN") != nullptr,
          dev_getenv("DTA_LEAN_MASK") ? atoi(dev_getenv("DTA_LEAN_MASK")) : DTA_LEAN_DEFAULT, dev_getenv("DTA_HALO_TILES") != nullptr,
          dev_getenv("DTA_NO_WGRAD_NSPLIT") == nullptr, (dev_getenv("DTA_PIXEL_ORDER") ? 1 : 0) | (dev_getenv("DTA_NO_STAGGER") ? 2 : 0), dev_getenv("DTA_NO_WGRAD_PAIR") != nullptr,
          dev_getenv("DTA_FANIN") ? atoi(dev_getenv("DTA_FANIN")) : 0,       // bit 0: forward BatchNorm statistics folded in-launch, bit 1: backward batch sums
          dev_getenv("DTA_NO_TAIL") != nullptr,                              // the fused forward tail (stage.hip: k_tail_fwd) off: stage 3 + head GEMMs + blend / loss launches
          dev_getenv("DTA_LEAD") ? atoi(dev_getenv("DTA_LEAD")) : DTA_LEAD_DEFAULT};      // bit 0: forward BatchNorm finalize as lead workgroups of the stage launch, bit 1: backward
}
Switches g_switches = read_switches();      // re-read only by dta_dev_reload_switches()
inline const Switches& switches() { return g_switches; }
constexpr int CH[3] = {32, 64, 128};
constexpr int SPEC_K[3] = {3, 5, 7};   // reference Hang2020.py:136-141
constexpr int SPAT_K[3] = {7, 5, 3};   // :77-85
constexpr int SPAT_P[3] = {4, 2, 1};   // :91-99

struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
};

// row tables of a conv launch's full workgroups (kernels.h: ConvArgs::tabs): slots 0-2 = the forward convs, 3-4 = the
// input-gradient convs of layers 2 and 3
constexpr size_t CONV_TAB_BYTES = (size_t)2 * 576 * 4;

// All workspace offsets for one network description (identical in forward and backward).
struct Plan {
  int G, B, bands, H, W, classes, esz;
  int cls[MAXG];   // class count of each group's classifier heads (= classes, except for the levels of a multi-stage step)
  int kinds[MAXG];
  int shared_x;   // Hang2020: both branches read the same input, so the first conv is ONE launch over [branch0|branch1] columns
  size_t x_tl_gs;  // bytes between the groups' network-input tiles (0 when shared)
  int NC0;
  int Hc[3], Wc[3], Hz[3], Wz[3], HWc[3], HWz[3], Qin[3];  // conv-res dims, post-pool dims, haloed grid of conv input
  int Cin[3], NCin[3];
  int F[MAXG][3], Fmax[3], vec_ld[3];
  int nwg[3], MWG[3];
  int S[3], cgroups[3], ngroups[3], CpadW[3];
  int wgrad_pair;       // bf16: the second and third conv's weight gradients share one launch (their S are sized for it)
  size_t x_tl, wp[3], wd[3], y[3], stats[3], coef[3], a_tl[3], feat[3], attpk[MAXG][3], scores[MAXG][3];
  size_t dfeat[3], dv[3], bnpart[3], bcoef[3], dy_tl[3], da[3], vec[3], wpart[3];
  size_t attsave[3]; int attsave_ld[3];
  int x_compact;   // bf16, 11x11-class patches: the network-input tiles are stored without their halo rows
  int y_fmt;       // storage of the conv outputs between kernels: fp32, or IEEE half in bf16 mode (common.h)
  int tl_compact;  // bf16 11x11 networks on the lean stage kernels: gated-map and output-gradient tiles are halo-free too
  int Rin[3];      // tile rows per chunk of layer L's conv INPUT tiles and of its output-gradient tiles (Qin or HWc)
  size_t scores_all, scores_bytes, dfeat_all, dfeat_bytes;
  // BatchNorm statistics / batch sums folded inside the producing launches (kernels.h, FanIn): arrival counters (zeroed by
  // the forward's prep launch, self re-arming), per layer FAN_R rows of forward sums and of backward sums
  size_t lead;          // 64 counters of the lead workgroups (stage.hip: bn_lead_block), cleared with the scores by k_forward_prep
  size_t fan_cnt, fan_cnt_bytes, fan_ctr, fan_fwd[3], fan_bwd[3];      // fan_cnt..: the cleared range (rows, then the counters at fan_ctr)
  size_t rowtabs;      // bf16: row tables of the step's five conv launches (conv_tab_slot), built by the forward's prep launch
  size_t fct; int fct_ld;              // Hang2020: the two last heads' weights transposed, [128 + 512][classes padded to 4] (fused forward tail)
  size_t total;
};

int build_plan(const dta_net_desc* d, Plan* p, int years = 0, const int* group_classes = nullptr) {
  memset(p, 0, sizeof(*p));
  if (d->batch < 1 || d->bands < 1 || d->height < 4 || d->width < 4 || d->classes < 1) {
    dta_set_error("bad descriptor: batch=%d bands=%d H=%d W=%d classes=%d", d->batch, d->bands, d->height, d->width, d->classes);
    return 1;
  }
  if (d->height + 2 > 255 || (d->height + 2) * (d->width + 2) > 65535) { dta_set_error("patch too large"); return 1; }
  p->G = d->kind == DTA_NET_HANG2020 ? 2 : 1;
  p->shared_x = d->kind == DTA_NET_HANG2020;
  p->B = d->batch; p->bands = d->bands; p->H = d->height; p->W = d->width; p->classes = d->classes;
  p->esz = d->dtype == DTA_BF16 ? 2 : 4;
  p->y_fmt = (d->dtype == DTA_BF16 && !switches().fp32_act) ? FMT_F16 : FMT_F32;
  switch (d->kind) {
    case DTA_NET_HANG2020: p->kinds[0] = KIND_SPECTRAL; p->kinds[1] = KIND_SPATIAL; break;
    case DTA_NET_SPECTRAL: p->kinds[0] = KIND_SPECTRAL; break;
    case DTA_NET_SPATIAL: p->kinds[0] = KIND_SPATIAL; break;
    case DTA_NET_VANILLA: p->kinds[0] = KIND_PLAIN; break;
    default: dta_set_error("unknown network kind %d", d->kind); return 1;
  }
  if (years > 0) {   // year ensemble: `years` spectral networks, each on its own input
    if (d->kind != DTA_NET_SPECTRAL || years > MAXG) { dta_set_error("an ensemble is 1..%d spectral networks", MAXG); return 1; }
    p->G = years;
    for (int g = 0; g < years; ++g) p->kinds[g] = KIND_SPECTRAL;
  }
  for (int g = 0; g < MAXG; ++g) {
    p->cls[g] = (group_classes && g < p->G) ? group_classes[g] : d->classes;
    if (p->cls[g] < 1) { dta_set_error("bad descriptor: group %d has %d classes", g, p->cls[g]); return 1; }
  }
  const int G = p->G, B = p->B;
  p->NC0 = (d->bands + 15) / 16;
  p->Hc[0] = p->H; p->Wc[0] = p->W; p->Hz[0] = p->H; p->Wz[0] = p->W;
  p->Hc[1] = p->H; p->Wc[1] = p->W; p->Hz[1] = p->H / 2; p->Wz[1] = p->W / 2;
  p->Hc[2] = p->Hz[1]; p->Wc[2] = p->Wz[1]; p->Hz[2] = p->Hz[1] / 2; p->Wz[2] = p->Wz[1] / 2;
  if (p->Hz[2] < 1 || p->Wz[2] < 1) { dta_set_error("patch %dx%d too small for two 2x2 pools", p->H, p->W); return 1; }
  if (d->kind == DTA_NET_VANILLA && CH[2] * p->Hz[2] * p->Wz[2] != 512) {
    // vanilla_CNN's head is Linear(512, classes) (reference Hang2020.py:43): fc_w holds classes x 512 floats, so a patch
    // that flattens to anything else would read / write past the caller's weight and gradient buffers
    dta_set_error("vanilla_CNN: a %dx%d patch flattens to %d features, its head takes 512", p->H, p->W, CH[2] * p->Hz[2] * p->Wz[2]);
    return 1;
  }
  for (int L = 0; L < 3; ++L) {
    p->HWc[L] = p->Hc[L] * p->Wc[L]; p->HWz[L] = p->Hz[L] * p->Wz[L];
    p->Qin[L] = (p->Hc[L] + 2) * (p->Wc[L] + 2);
    p->Cin[L] = L == 0 ? d->bands : CH[L - 1];
    p->NCin[L] = (p->Cin[L] + 15) / 16;
    p->Fmax[L] = 0; p->vec_ld[L] = 1;
    for (int g = 0; g < G; ++g) {
      int f = 0, vl = 1;
      if (p->kinds[g] == KIND_SPECTRAL) { f = CH[L]; vl = 4 * CH[L]; }
      else if (p->kinds[g] == KIND_SPATIAL) {
        int hp = p->Hz[L] / SPAT_P[L], wp = p->Wz[L] / SPAT_P[L];
        if (hp < 1 || wp < 1) { dta_set_error("spatial attention %d: %dx%d map smaller than its %d-pool", L + 1, p->Hz[L], p->Wz[L], SPAT_P[L]); return 1; }
        f = CH[L] * hp * wp; vl = CH[L] + 2 * SPAT_K[L] * SPAT_K[L] + 3;
      } else if (L == 2) f = CH[2] * p->HWz[2];
      p->F[g][L] = f;
      if (f > p->Fmax[L]) p->Fmax[L] = f;
      if (vl > p->vec_ld[L]) p->vec_ld[L] = (vl + 3) & ~3;   // rows 16-byte aligned: the batch GEMMs over them use vector loads
    }
    int Nconv = (L == 0 && p->shared_x) ? 32 * G : CH[L];
    p->MWG[L] = d->dtype == DTA_BF16 ? conv_mwg_bf16(Nconv, p->HWc[L]) : conv_mwg(Nconv);
    if (d->dtype == DTA_BF16 && L > 0 && Nconv == 64) {
      // second conv: two 256-row workgroups per CU; a 24x24 map is exactly one 576-row workgroup of six waves (256-row
      // workgroups cut it into 256 + 256 + 64 rows: a quarter of the tile rows empty and three weight stagings per patch)
      p->MWG[L] = (p->HWc[L] == 576 && !dev_getenv("DTA_NO_CONV2_576")) ? 576 : 256;
    }
    int ppw, spp;
    conv_geometry(p->HWc[L], p->MWG[L], B, &ppw, &spp, &p->nwg[L]);
    // weight-gradient split
    int cpw = wgrad_cpw(Nconv);
    p->CpadW[L] = p->NCin[L] * 16;
    p->cgroups[L] = (p->CpadW[L] + cpw - 1) / cpw;
    int launchG = (L == 0 && p->shared_x) ? 1 : G;
    // bf16 path: register-prefetch pipeline, 1 workgroup per CU; fp32 path: 2 workgroups per CU overlap each other
    int target = d->dtype == DTA_BF16 ? 256 : 512;
    p->ngroups[L] = (switches().wgrad_nsplit ? wgrad_ngroups(Nconv, d->dtype == DTA_BF16) : 1);
    int S = target / (p->cgroups[L] * p->ngroups[L] * launchG);   // floor: never spill into a second round of workgroups
    if (S >= 8) S &= ~7;                           // multiple of 8: whole batch splits per XCD (see k_conv_wgrad_bf16)
    p->S[L] = S < 1 ? 1 : (S > B ? B : S);
  }
  {   // halo-free input tiles need the single-window weight-gradient plan (and the bf16 kernels)
    int bl, wr, nb;
    wgrad_band_plan(p->Qin[0], p->Wc[0], 252, &bl, &wr, &nb);
    p->x_compact = d->dtype == DTA_BF16 && nb == 1 && bl >= 16 && wr <= 256;
    // every layer of the 11x11 networks is a single-band plan; the lean forward kernels write the halo-free gated maps
    const bool net11 = p->H == 11 && p->W == 11;
    p->tl_compact = p->x_compact && net11 && !switches().no_lean && (switches().lean_mask & 1) && !switches().halo_tiles;
    for (int L = 0; L < 3; ++L) p->Rin[L] = p->tl_compact ? p->HWc[L] : p->Qin[L];
  }
  // bf16: the weight gradients of the second and third conv run side by side in one launch (conv_bf16.hip,
  // k_conv_wgrad_bf16_pair): 96 + 160 of the 256 CUs (the third conv's stacked 5x5 windows carry more work per slab),
  // batch splits in whole multiples of 8 workgroups so that a split's workgroups stay on one XCD
  p->wgrad_pair = 0;
  auto pair_ok = [&]() {
    WgradArgs w[2];
    memset(w, 0, sizeof(w));
    for (int k = 0; k < 2; ++k) {
      const int L = k + 1;
      w[k].H = p->Hc[L]; w[k].W = p->Wc[L]; w[k].Q = p->Qin[L]; w[k].N = CH[L]; w[k].Cpad = p->CpadW[L]; w[k].ngroups = p->ngroups[L];
      w[k].x_compact = w[k].y_compact = p->tl_compact; w[k].S = 1; w[k].B = B;
    }
    return wgrad_pair_plan_bf16(w[0], w[1]);
  };
  const int pair_kind = (d->dtype == DTA_BF16 && !switches().no_wgrad_pair) ? pair_ok() : 0;
  if (pair_kind) {
    auto split = [&](int L, int budget) {
      int S = budget / (p->cgroups[L] * p->ngroups[L] * G);
      // (whole XCDs per batch split where the group count allows it; three years do not divide 8: any S then)
      if (pair_kind == 1) while (S > 0 && (S * G) % 8 != 0) --S;
      return S > B ? 0 : S;
    };
    // wide windows (24x24 crops): a launch is 21.7 us fixed + 18.7 us of work at 240 workgroups: equal shares
    const int S2 = split(1, pair_kind == 1 ? 96 : 128), S3 = split(2, pair_kind == 1 ? 160 : 128);      // (measured: 96/160 0.5266 ms, 88/160 0.5298, 128/128 0.5318, 64/192 0.5395)
    if (S2 > 0 && S3 > 0) { p->S[1] = S2; p->S[2] = S3; p->wgrad_pair = 1; }
  }
  Carver c;
  const size_t e = p->esz;
  {
    const size_t one = ((size_t)B * p->NC0 * p->Qin[0] * 16 * e + 255) & ~(size_t)255;
    p->x_tl_gs = p->shared_x ? 0 : one;
    p->x_tl = c.take(one * (p->shared_x ? 1 : G));
  }
  for (int L = 0; L < 3; ++L) {
    int Nconv = (L == 0 && p->shared_x) ? 32 * G : CH[L];
    int gw = (L == 0 && p->shared_x) ? 1 : G;
    p->wp[L] = c.take((size_t)gw * p->NCin[L] * 9 * Nconv * 16 * e);
    p->y[L] = c.take((size_t)G * B * p->HWc[L] * CH[L] * 4);
    p->stats[L] = c.take((size_t)gw * p->nwg[L] * Nconv * 2 * 4);
    p->coef[L] = c.take((size_t)G * CH[L] * 4 * 4);
    if (L < 2) p->a_tl[L] = c.take((size_t)G * B * (CH[L] / 16) * p->Qin[L + 1] * 16 * e);
    p->feat[L] = c.take((size_t)G * B * (p->Fmax[L] > 0 ? p->Fmax[L] : 1) * 4);
    for (int g = 0; g < G; ++g) p->attpk[g][L] = c.take((size_t)4 * CH[L] * CH[L] * 4);
  }
  for (int L = 0; L < 3; ++L) {
    int Nconv = (L == 0 && p->shared_x) ? 32 * G : CH[L];
    int gw = (L == 0 && p->shared_x) ? 1 : G;
    p->fan_fwd[L] = c.take((size_t)gw * FAN_R * Nconv * 3 * 8);
    p->fan_bwd[L] = c.take((size_t)G * FAN_R * CH[L] * 2 * 8);
  }
  c.take((size_t)2 * 3 * MAXG * FAN_R * 4);                    // the counters
  // (rows + counters sit directly in front of the scores: one clearing job for all -- a logical group without
  //  workgroups, e.g. a launch of fewer than FAN_R workgroups, leaves its row untouched, and a zero row adds nothing)
  p->fan_cnt = p->fan_fwd[0];
  p->fan_ctr = c.off - (((size_t)2 * 3 * MAXG * FAN_R * 4 + 255) & ~(size_t)255);
  p->lead = c.take(256);
  p->fan_cnt_bytes = c.off - p->fan_cnt;
  p->scores_all = c.off;
  for (int g = 0; g < G; ++g)
    for (int L = 0; L < 3; ++L) p->scores[g][L] = c.take((size_t)B * p->cls[g] * 4);
  p->scores_bytes = c.off - p->scores_all;
  // backward
  p->dfeat_all = c.off;
  for (int L = 0; L < 3; ++L) p->dfeat[L] = c.take((size_t)G * B * (p->Fmax[L] > 0 ? p->Fmax[L] : 1) * 4);
  p->dfeat_bytes = c.off - p->dfeat_all;
  for (int L = 0; L < 3; ++L) {
    p->dv[L] = c.take((size_t)G * B * p->HWc[L] * CH[L] * 4);
    p->bnpart[L] = c.take((size_t)G * B * CH[L] * 2 * 4);
    p->bcoef[L] = c.take((size_t)G * CH[L] * 4 * 4);
    p->dy_tl[L] = c.take((size_t)G * B * (CH[L] / 16) * p->Qin[L] * 16 * e);
    p->vec[L] = c.take((size_t)G * B * p->vec_ld[L] * 4);
    {   // attention intermediates kept from the forward: 3 vector slots per patch (padded maps for stencil radius <= 3)
      const int pm = (p->Hz[L] + 6) * (p->Wz[L] + 6);
      p->attsave_ld[L] = 3 * (CH[L] > pm ? CH[L] : pm);
      p->attsave[L] = c.take((size_t)G * B * p->attsave_ld[L] * 4);
    }
    if (L > 0) {
      p->wd[L] = c.take((size_t)G * (CH[L] / 16) * 9 * CH[L - 1] * 16 * e);
      p->da[L] = c.take((size_t)G * B * p->HWc[L] * CH[L - 1] * 4);   // grad wrt the conv's input map
    }
  }
  for (int L = 0; L < 3; ++L) {   // split-K slabs of every layer stay live until the one grouped reduction
    int Nconv = (L == 0 && p->shared_x) ? 32 * G : CH[L];
    int launchG = (L == 0 && p->shared_x) ? 1 : G;
    p->wpart[L] = c.take((size_t)launchG * p->S[L] * 9 * p->CpadW[L] * Nconv * 4);
  }
  p->rowtabs = c.take((size_t)5 * CONV_TAB_BYTES);
  p->fct_ld = (p->classes + 3) / 4 * 4;
  p->fct = d->kind == DTA_NET_HANG2020 ? c.take((size_t)(p->F[0][2] + p->F[1][2]) * p->fct_ld * 4) : 0;
  p->total = c.off;
  return 0;
}

// bf16 networks whose input tiles are halo-free: the first conv converts the fp32 input while staging it.  With few
// workgroups (small batches) the conversion's long per-chunk load chain is exposed and the wide, shallow pack job is
// faster (measured crossover: ~100 first-conv workgroups, B ~ 450 for Hang2020)
inline bool fused_input(const Plan& p) {
  const int launchG = p.shared_x ? 1 : p.G;
  // (tiles that keep their halo -- 24x24-class crops -- take the same route as long as a workgroup owns whole patches)
  // (no fused-input kernel exists for 576-row x 64-column workgroups -- two branches on 24x24-class crops: it would spill)
  if (p.shared_x && p.MWG[0] == 576) return false;
  return p.esz == 2 && p.HWc[0] <= p.MWG[0] && p.nwg[0] * launchG >= 100 && !switches().no_fused_input;
}

template <typename T> inline T* at(void* ws, size_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(ws) + off); }

StageArgs stage_args(const Plan& p, const dta_net_desc* d, const dta_subnet_params* nets, void* ws, int L) {
  StageArgs s;
  memset(&s, 0, sizeof(s));
  const int G = p.G, B = p.B, C = CH[L];
  for (int g = 0; g < G; ++g) {
    s.kind[g] = p.kinds[g];
    s.F[g] = p.F[g][L];
    if (p.kinds[g] == KIND_SPECTRAL) {
      float* pk = at<float>(ws, p.attpk[g][L]);
      s.att[g].p[0] = pk; s.att[g].p[1] = nets[g].att[L][1];
      s.att[g].p[2] = pk + C * C; s.att[g].p[3] = nets[g].att[L][3];
      s.att[g].p[4] = pk + 2 * C * C; s.att[g].p[5] = pk + 3 * C * C;
    } else if (p.kinds[g] == KIND_SPATIAL) {
      for (int i = 0; i < 6; ++i) s.att[g].p[i] = nets[g].att[L][i];
      s.att_k[g] = SPAT_K[L]; s.att_pool[g] = SPAT_P[L];
    }
  }
  s.y = at<float>(ws, p.y[L]);
  s.y_fmt = p.y_fmt;
  s.lean = switches().no_lean ? 0 : switches().lean_mask;
  if (L == 0 && p.shared_x) { s.y_gs = 32; s.y_rs = 32 * G; }
  else { s.y_gs = (size_t)B * p.HWc[L] * C; s.y_rs = C; }
  s.coef = at<float>(ws, p.coef[L]); s.coef_gs = C * 4;
  s.apply_bn = 1; s.relu = 1; s.pool = L > 0;
  s.B = B; s.C = C; s.Hc = p.Hc[L]; s.Wc = p.Wc[L];
  if (L < 2) {
    s.a_tl = at<char>(ws, p.a_tl[L]);
    s.a_nc = C / 16; s.a_ch0 = 0;
    s.a_gs = (size_t)B * (C / 16) * p.Rin[L + 1] * 16;
    s.a_compact = p.tl_compact;
  }
  s.attsave = at<float>(ws, p.attsave[L]); s.attsave_ld = p.attsave_ld[L];
  s.feat = at<float>(ws, p.feat[L]);
  s.feat_gs = (size_t)B * (p.Fmax[L] > 0 ? p.Fmax[L] : 1);
  // per-group row length of feat is F[g] (rows packed per group)
  return s;
}

// Geometry (the fields the bf16 launcher's tile choice reads: conv_bf16_rows) of the forward conv of layer L and of the
// input-gradient conv of layer L -- shared by the launches and by the prep launch's row-table jobs, so they cannot disagree.
ConvArgs fwd_conv_geom(const Plan& p, const dta_net_desc* d, int L, bool fan_fwd) {
  ConvArgs ca;
  memset(&ca, 0, sizeof(ca));
  const bool cat = L == 0 && p.shared_x;
  const int Nconv = cat ? 32 * p.G : CH[L];
  ca.mwg = p.MWG[L];
  ca.stats = d->training ? reinterpret_cast<float*>(1) : nullptr;      // (only its null-ness matters here; the launch sets the pointer)
  static const bool no_nsplit = dev_getenv("DTA_NO_NSPLIT") != nullptr;
  // bf16, 128 output channels (third conv) on maps of 12x12 and up: two 64-column groups per row tile -- half-width
  // workgroups stage half the weight slab each, twice as many of them (same-box alternation, 3 x 369 x 24x24: 1.0215 ->
  // 1.0148 ms; the 5x5 maps of the 11x11 networks measured 0.5248 -> 0.5260 ms with it and keep the full-width tile)
  if (p.esz == 2 && Nconv == 128 && p.HWc[L] >= 64 && !fan_fwd && !no_nsplit) ca.ncg = 2;
  ca.B = p.B; ca.H = p.Hc[L]; ca.W = p.Wc[L]; ca.NC = p.NCin[L]; ca.N = Nconv; ca.Q = p.Qin[L]; ca.HW = p.HWc[L];
  return ca;
}
ConvArgs dgrad_conv_geom(const Plan& p, int L) {
  ConvArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.B = p.B; ca.H = p.Hc[L]; ca.W = p.Wc[L]; ca.NC = CH[L] / 16; ca.N = CH[L - 1]; ca.Q = p.Qin[L]; ca.HW = p.HWc[L];
  return ca;
}
inline int conv_tab_slot_fwd(int L) { return L; }
inline int conv_tab_slot_dgrad(int L) { return 2 + L; }
// points ca.tabs at its slot when the launcher's tile is one the prep launch tabulated (whole patches per workgroup)
void use_conv_tabs(const Plan& p, void* ws, ConvArgs& ca, int G, int slot) {
  if (p.esz != 2) return;
  const int rows = conv_bf16_rows(ca, G);
  if (rows <= 0 || ca.HW > rows) return;
  ca.tabs = reinterpret_cast<const int*>(reinterpret_cast<char*>(ws) + p.rowtabs + (size_t)slot * CONV_TAB_BYTES);
  ca.tabs_rows = rows;
}
bool add_conv_tab_job(const Plan& p, void* ws, const ConvArgs& geom, int G, int slot, ConvTabGroup& tg) {
  ConvArgs ca = geom;
  use_conv_tabs(p, ws, ca, G, slot);
  if (!ca.tabs) return false;
  ConvTabJob& j = tg.job[tg.n++];
  j.HW = ca.HW; j.W = ca.W; j.Q = ca.Q; j.rows = ca.tabs_rows; j.ppw = ca.tabs_rows / ca.HW; j.order = switches().conv_order;
  j.dst = const_cast<int*>(ca.tabs);
  return true;
}

// the loss of a single-score network, taken in the same call as its forward (dta_net_forward_loss)
struct LossArgs { const long long* labels; const float* weight; float* loss; float* dlogits; float* scratch; };

// Does this forward end in the fused tail launch (stage.hip: k_tail_fwd)?  A two-branch Hang2020 on 11x11 patches in
// training mode whose caller wants the last heads only and lets the workspace hold the branch scores.
bool tail_plan(const Plan& p, const dta_net_desc* d, float* const (*scores)[3]) {
  if (d->kind != DTA_NET_HANG2020 || !d->training || (d->heads_mask & 7) != 4 || (d->heads_mask & DTA_FORWARD_ONLY)) return false;
  if (switches().no_tail || switches().no_lean || !(switches().lean_mask & 1) || switches().bn_inkernel || (switches().fanin & 1)) return false;
  if (scores && (scores[0][2] || scores[1][2])) return false;
  if (p.H != 11 || p.W != 11 || p.F[0][2] != 128 || p.F[1][2] != 512) return false;
  return true;
}

template <typename T>
int forward_t(const Plan& p, const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha,
              const float* const* xs, void* ws, float* const (*scores)[3], float* joint, hipStream_t st,
              const void* x_tiles = nullptr, const float* gate = nullptr, const LossArgs* loss = nullptr) {
  // x_tiles: the network input already as halo-free bf16 conv tiles (dta_preprocess_crops_tiles): no fp32 input at all
  if (x_tiles && !(p.esz == 2 && p.x_compact && (p.shared_x || p.G == 1))) {
    dta_set_error("input tiles need the bf16 mode, 11x11-class patches and a single input tensor");
    return 1;
  }
  const int G = p.G, B = p.B;
  GemmGroup heads;
  bool tail = tail_plan(p, d, scores);
  if (tail) {
    StageArgs s3 = stage_args(p, d, nets, ws, 2);
    tail = tail_fwd_supported(s3, G, p.classes);
  }
  bool loss_done = false;
  // ---- all weight re-layouts of the step in two launches (forward forms, and when training the transposed
  //      forms the input-gradient convs will need) ----
  PackWGroup packs;
  SpecPackGroup spacks;
  int pack_mode[3];
  for (int L = 0; L < 3; ++L) {
    const int C = CH[L];
    PackWArgs pw;
    memset(&pw, 0, sizeof(pw));
    const bool cat = L == 0 && p.shared_x;   // one launch over the concatenated branch columns
    pw.G = cat ? 1 : G; pw.NC = p.NCin[L]; pw.N = cat ? 32 * G : C; pw.K = p.Cin[L];
    for (int g = 0; g < G; ++g) pw.src[g] = nets[g].conv_w[L];
    pw.mode = (cat && G == 2) ? 1 : 0; pw.nsplit = 32;
    pack_mode[L] = pw.mode;
    packs.job[packs.n] = pw; packs.dst[packs.n++] = at<char>(ws, p.wp[L]);
    if (L > 0 && d->training) {
      memset(&pw, 0, sizeof(pw));
      pw.G = G; pw.NC = C / 16; pw.N = CH[L - 1]; pw.K = C; pw.mode = 2;
      for (int g = 0; g < G; ++g) pw.src[g] = nets[g].conv_w[L];
      packs.job[packs.n] = pw; packs.dst[packs.n++] = at<char>(ws, p.wd[L]);
    }
    for (int g = 0; g < G; ++g)
      if (p.kinds[g] == KIND_SPECTRAL) {
        int j = spacks.n++;
        spacks.w1[j] = nets[g].att[L][0]; spacks.w2[j] = nets[g].att[L][2];
        spacks.packed[j] = at<float>(ws, p.attpk[g][L]); spacks.C[j] = C; spacks.K[j] = SPEC_K[L];
      }
  }
  // frozen-weight inference (DTA_REUSE_PACKED): the re-layouts and row tables of an earlier call are still in the workspace
  const bool reuse = !d->training && (d->heads_mask & DTA_FORWARD_ONLY) && (d->heads_mask & DTA_REUSE_PACKED);
  if (reuse) { packs.n = 0; spacks.n = 0; }
  {
    PrepArgs pa = {};
    // bf16, halo-free input tiles: the first conv reads the caller's fp32 tensor itself (and leaves the bf16 tiles
    // behind for the weight gradient), so there is no input-pack job
    pa.nx = (x_tiles || fused_input(p)) ? 0 : (p.shared_x ? 1 : G); pa.x_tl_gs = p.x_tl_gs;
    for (int g = 0; g < pa.nx; ++g) pa.x[g] = xs[g];
    pa.x_tl = at<char>(ws, p.x_tl); pa.B = B; pa.C = p.bands; pa.H = p.H; pa.W = p.W;
    pa.x_compact = p.x_compact;
    pa.packs = packs; pa.spacks = spacks;
    if (sizeof(T) == 2 && !reuse) {      // row tables of the step's conv launches (full workgroups read them instead of building their own)
      const bool fan_fwd0 = d->training && (switches().fanin & 1);
      for (int L = 0; L < 3; ++L) add_conv_tab_job(p, ws, fwd_conv_geom(p, d, L, fan_fwd0), (L == 0 && p.shared_x) ? 1 : G, conv_tab_slot_fwd(L), pa.tabs);
      if (d->training && !(d->heads_mask & DTA_FORWARD_ONLY))
        for (int L = 1; L < 3; ++L) add_conv_tab_job(p, ws, dgrad_conv_geom(p, L), G, conv_tab_slot_dgrad(L), pa.tabs);
    }
    if (tail) {      // the last heads' weights [classes][F] -> [F][classes padded to 4], spectral rows first
      for (int g = 0; g < 2; ++g) {
        if (!nets[g].fc_w[2] || !nets[g].fc_b[2]) { dta_set_error("Hang2020 forward: the last heads' parameters are missing"); return 1; }
        pa.trans.src[g] = nets[g].fc_w[2];
        pa.trans.dst[g] = at<float>(ws, p.fct) + (size_t)(g ? p.F[0][2] : 0) * p.fct_ld;
        pa.trans.rows[g] = p.classes; pa.trans.cols[g] = p.F[g][2]; pa.trans.ld[g] = p.fct_ld;
      }
      pa.trans.n = 2;
    }
    // the split-K GEMM targets (and, for the DTA_FANIN experiment, the fan-in rows and counters directly in front of them)
    if (switches().fanin) { pa.zero = at<float>(ws, p.fan_cnt); pa.zero_n4 = (p.fan_cnt_bytes + (d->heads_mask ? p.scores_bytes : 0) + 15) / 16; }
    else { pa.zero = at<float>(ws, p.lead); pa.zero_n4 = (256 + (d->heads_mask ? p.scores_bytes : 0) + 15) / 16; }      // (lead counters sit directly in front of the scores)
    if (launch_forward_prep<T>(pa, st)) return 1;
  }
  for (int L = 0; L < 3; ++L) {
    const int C = CH[L];
    const bool cat = L == 0 && p.shared_x;
    const int Nconv = cat ? 32 * G : C;
    const int launchG = cat ? 1 : G;
    // conv
    const bool fan_fwd = d->training && (switches().fanin & 1);
    ConvArgs ca = fwd_conv_geom(p, d, L, fan_fwd);
    use_conv_tabs(p, ws, ca, launchG, conv_tab_slot_fwd(L));
    ca.pixel_order = switches().conv_order;
    if (L == 0) {
      ca.x_tl = x_tiles ? x_tiles : at<char>(ws, p.x_tl); ca.x_gs = p.x_tl_gs / p.esz; ca.x_compact = p.x_compact;
      if (!x_tiles && fused_input(p)) {
        for (int g = 0; g < G; ++g) ca.x_nchw[g] = xs[p.shared_x ? 0 : g];
        ca.Cx = p.bands;
        ca.x_tl_out = (d->heads_mask & DTA_FORWARD_ONLY) ? nullptr : at<char>(ws, p.x_tl);   // only the backward reads the tiles
      }
    } else { ca.x_tl = at<char>(ws, p.a_tl[L - 1]); ca.x_gs = (size_t)B * p.NCin[L] * p.Rin[L] * 16; ca.x_compact = p.tl_compact; }
    ca.wp = at<char>(ws, p.wp[L]);
    for (int g = 0; g < G; ++g) ca.bias[g] = nets[g].conv_b[L];
    ca.bias_mode = pack_mode[L]; ca.bias_split = 32;
    ca.y = at<float>(ws, p.y[L]); ca.y_fmt = p.y_fmt;
    if (cat) { ca.y_gs = 0; ca.y_rs = Nconv; } else { ca.y_gs = (size_t)B * p.HWc[L] * C; ca.y_rs = C; }
    ca.stats = d->training ? at<float>(ws, p.stats[L]) : nullptr;
    // DTA_FANIN=1 (experiment, measured SLOWER than the finalize launch it removes: profiles/README.md, round 4): the conv
    // launch folds its BatchNorm partials itself (FAN_R rows of raw sums), the stage workgroups add those up in their prologue
    if (fan_fwd) { ca.fan_count = at<unsigned>(ws, p.fan_ctr) + (size_t)L * MAXG * FAN_R; ca.fan_sums = at<double>(ws, p.fan_fwd[L]); }
    prof_begin(DTA_SITE_CONV_FWD + L, st);
    if (launch_conv3x3<T>(ca, launchG, st)) return 1;
    prof_end(DTA_SITE_CONV_FWD + L, st);
    // BatchNorm statistics -> per-channel scale/shift
    BnFinalizeArgs bf;
    memset(&bf, 0, sizeof(bf));
    bf.stats = at<float>(ws, p.stats[L]); bf.nwg = p.nwg[L]; bf.N = Nconv; bf.HW = p.HWc[L]; bf.MWG = p.MWG[L]; bf.B = B;
    for (int g = 0; g < G; ++g) {
      bf.gamma[g] = nets[g].bn_w[L]; bf.beta[g] = nets[g].bn_b[L];
      bf.rmean[g] = nets[g].bn_rm[L]; bf.rvar[g] = nets[g].bn_rv[L]; bf.nbt[g] = nets[g].bn_nbt[L];
    }
    bf.cat_mode = (cat && G == 2); bf.nsplit = 32;
    bf.coef = at<float>(ws, p.coef[L]); bf.training = d->training; bf.momentum = d->bn_momentum; bf.eps = d->bn_eps;
    bf.gate = gate;      // (year ensembles: a year the step skips keeps its running statistics)
    // BN + ReLU + pool + attention
    StageArgs sa = stage_args(p, d, nets, ws, L);
    // eval mode: the coefficients are a function of the running statistics only, so every stage workgroup derives
    // them itself and the finalize launch disappears (three dependent launches fewer per inference batch).  Training:
    // combining the conv partials redundantly in every stage workgroup was measured SLOWER than the launch (each of
    // 2048 workgroups pulls 64-128 KB through its XCD's L2: +5 / +14 / +13 us on the three layers against a 4-5 us
    // launch), so it stays a separate wide-and-shallow launch; DTA_BN_INKERNEL=1 re-enables it for experiments
    if (fan_fwd) { bf.fsum = ca.fan_sums; sa.bn_inkernel = 1; sa.bnfin = bn_finalize_kargs(bf); }
    else if ((!d->training || switches().bn_inkernel) && p.nwg[L] <= BN_INKERNEL_MAX_NWG) { sa.bn_inkernel = 1; sa.bnfin = bn_finalize_kargs(bf); }
    else if ((switches().lead & 1) && !(L == 2 && tail) && stage_fwd_will_be_lean(sa, G)) {
      // DTA_LEAD=1 (experiment): no finalize launch -- the first workgroups of the stage launch do its work while the
      // others' loads are in flight
      sa.bnfin = bn_finalize_kargs(bf);
      sa.lead = G * ((sa.bnfin.C + 7) / 8); sa.lead_flag = at<unsigned>(ws, p.lead) + L; sa.lead_need = (unsigned)sa.lead;
    }
    else if (launch_bn_finalize(bf, G, st)) return 1;
    if (d->heads_mask & DTA_FORWARD_ONLY) sa.attsave = nullptr;   // attention state is kept for the backward only
    if (L == 2 && tail) {
      // third stage of both branches + the two last heads + blend (+ loss) in ONE launch
      TailArgs ta;
      memset(&ta, 0, sizeof(ta));
      ta.st = sa;
      ta.wt = at<float>(ws, p.fct); ta.ldw = p.fct_ld;
      for (int g = 0; g < 2; ++g) { ta.bias[g] = nets[g].fc_b[2]; ta.scores[g] = at<float>(ws, p.scores[g][2]); }
      ta.alpha = alpha; ta.classes = p.classes;
      ta.ce.gscale = 1.f;
      if (loss) {
        if (!alpha) { dta_set_error("Hang2020 forward + loss needs alpha"); return 1; }
        ta.ce.labels = loss->labels; ta.ce.weight = loss->weight; ta.ce.loss = loss->loss; ta.ce.dlogits = loss->dlogits;
        ta.ce.rowtmp = loss->scratch; ta.ce.joint = joint;
        loss_done = true;
      } else if (!(d->heads_mask & DTA_SKIP_BLEND)) {
        if (!joint || !alpha) { dta_set_error("Hang2020 forward needs head 3, alpha and a joint output"); return 1; }
        ta.ce.joint = joint;
      }
      prof_begin(DTA_SITE_STAGE_FWD + L, st);
      if (launch_tail_fwd(ta, st)) return 1;
      prof_end(DTA_SITE_STAGE_FWD + L, st);
      continue;
    }
    prof_begin(DTA_SITE_STAGE_FWD + L, st);
    if (launch_stage_fwd<T>(sa, G, st)) return 1;
    prof_end(DTA_SITE_STAGE_FWD + L, st);
    // classifier heads
    if (d->heads_mask & (1 << L)) {
      for (int g = 0; g < G; ++g) {
        if (p.F[g][L] == 0 || nets[g].fc_w[L] == nullptr) continue;
        const int F = p.F[g][L];
        GemmArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.A = at<float>(ws, p.feat[L]) + (size_t)g * sa.feat_gs; ga.sa_m = F; ga.sa_k = 1;
        ga.Bm = nets[g].fc_w[L]; ga.sb_k = 1; ga.sb_n = F;
        float* out = at<float>(ws, p.scores[g][L]);
        // K slices of ~128 (at most 4): measured best for this launch (23 us vs 30 us with 2 slices of 256; 8 slices
        // lose to the longer same-address atomic chains)
        ga.M = B; ga.N = p.cls[g]; ga.K = F; ga.ksplit = min(4, max(1, (F + 63) / 64)); ga.accumulate = 0;
        float* user = (scores && scores[g][L]) ? scores[g][L] : nullptr;
        if (d->kind == DTA_NET_VANILLA && joint) user = joint;
        if (user) {
          out = user;
          if (ga.ksplit > 1) hipMemsetAsync(user, 0, (size_t)B * p.cls[g] * 4, st);
        }
        ga.C = out; ga.sc_m = p.cls[g]; ga.sc_n = 1;
        ga.bias = nets[g].fc_b[L];
        if (!heads.add(ga)) { dta_set_error("too many head GEMMs"); return 1; }
      }
    }
  }
  prof_begin(DTA_SITE_GEMM, st);
  if (launch_gemm_group(heads, st)) return 1;   // all classifier heads of all branches in one launch
  prof_end(DTA_SITE_GEMM, st);
  if (tail) return (loss && !loss_done) ? 1 : 0;      // (blend and loss happened in the tail launch)
  if (loss) {
    // no fused tail for this plan: the loss launch dta_net_loss would issue (blend + CE in one)
    BlendCeArgs a;
    memset(&a, 0, sizeof(a));
    a.gscale = 1.f;
    if (d->kind == DTA_NET_HANG2020) {
      if (!alpha || !(d->heads_mask & 4)) { dta_set_error("Hang2020 forward + loss needs head 3 and alpha"); return 1; }
      a.spec = (scores && scores[0][2]) ? scores[0][2] : at<float>(ws, p.scores[0][2]);
      a.spat = (scores && scores[1][2]) ? scores[1][2] : at<float>(ws, p.scores[1][2]);
      a.alpha = alpha; a.joint = joint;
    } else {
      if (!joint) { dta_set_error("forward + loss: the scores of a single-branch network are passed in `joint`"); return 1; }
      a.spec = joint; a.joint = joint;
    }
    a.labels = loss->labels; a.weight = loss->weight; a.dlogits = loss->dlogits; a.loss = loss->loss; a.rowtmp = loss->scratch;
    a.B = B; a.classes = p.classes;
    return launch_blend_ce(a, st);
  }
  if (d->kind == DTA_NET_HANG2020 && !(d->heads_mask & DTA_SKIP_BLEND)) {
    if (!(d->heads_mask & 4) || !joint || !alpha) { dta_set_error("Hang2020 forward needs head 3, alpha and a joint output"); return 1; }
    BlendArgs ba;
    ba.spec = (scores && scores[0][2]) ? scores[0][2] : at<float>(ws, p.scores[0][2]);
    ba.spat = (scores && scores[1][2]) ? scores[1][2] : at<float>(ws, p.scores[1][2]);
    ba.alpha = alpha; ba.joint = joint; ba.B = B; ba.classes = p.classes;
    if (launch_blend(ba, st)) return 1;
  }
  return 0;
}

template <typename T>
int conv_wgrad_layer(const Plan& p, const dta_net_desc* d, const dta_subnet_grads* grads, void* ws, int L,
                     WgradReduceGroup& reduces, hipStream_t st, const void* x_tiles = nullptr,
                     WgradArgs* defer = nullptr, const WgradArgs* partner = nullptr, dta_xchg* xchg = nullptr,
                     const double* dalpha = nullptr, long long alpha_slot = -1) {
  const int G = p.G, B = p.B, C = CH[L];
  const bool cat = L == 0 && p.shared_x;
  const int Nconv = cat ? 32 * G : C;
  const int launchG = cat ? 1 : G;
  bool want_w = false;
  for (int g = 0; g < G; ++g) want_w |= grads[g].conv_w[L] != nullptr;
  if (!want_w) return 0;
  WgradArgs wa;
  memset(&wa, 0, sizeof(wa));
  if (L == 0) { wa.x_tl = x_tiles ? x_tiles : at<char>(ws, p.x_tl); wa.x_gs = p.x_tl_gs / p.esz; wa.x_compact = p.x_compact; }
  else { wa.x_tl = at<char>(ws, p.a_tl[L - 1]); wa.x_gs = (size_t)B * p.NCin[L] * p.Rin[L] * 16; wa.x_compact = p.tl_compact; }
  wa.NCx = p.NCin[L];
  wa.dy_tl = at<char>(ws, p.dy_tl[L]);
  wa.dy_gs = cat ? 0 : (size_t)B * (C / 16) * p.Rin[L] * 16;
  wa.y_compact = p.tl_compact;
  wa.NCy = Nconv / 16; wa.ych0 = 0; wa.ngroups = p.ngroups[L];
  wa.partial = at<float>(ws, p.wpart[L]);
  wa.B = B; wa.H = p.Hc[L]; wa.W = p.Wc[L]; wa.Q = p.Qin[L]; wa.N = Nconv; wa.Cpad = p.CpadW[L]; wa.S = p.S[L];
  if (defer) *defer = wa;      // (third conv of a paired plan: launched together with the second conv's)
  else {
    prof_begin(DTA_SITE_CONV_WGRAD + L, st);
    bool done = false;
    if (xchg && L == 0 && !partner && sizeof(T) == 2) {
      // data-parallel peer exchange: the head bucket's reduce-scatter rides in this launch as side workgroups
      XchgArgs side;
      if (dta_xchg_side_args(xchg, dalpha, alpha_slot, &side) == 0) {
        const int rc = launch_conv_wgrad_bf16_xchg(wa, launchG, side, st);
        if (rc == 1) return 1;
        if (rc == 0) done = true;
        else dta_xchg_side_cancel(xchg);      // (this plan has no combined kernel: the exchange sums the head itself)
      }
    }
    if (!done && (partner ? launch_conv_wgrad_pair_bf16(wa, *partner, launchG, st) : launch_conv_wgrad<T>(wa, launchG, st))) return 1;
    prof_end(DTA_SITE_CONV_WGRAD + L, st);
  }
  WgradReduceArgs wr;
  memset(&wr, 0, sizeof(wr));
  wr.partial = wa.partial; wr.G = launchG; wr.S = p.S[L]; wr.N = Nconv; wr.C = p.Cin[L]; wr.Cpad = p.CpadW[L];
  wr.mode = (cat && G == 2) ? 1 : 0; wr.nsplit = 32;
  for (int g = 0; g < G; ++g) wr.dst[g] = grads[g].conv_w[L];
  reduces.job[reduces.n++] = wr;   // reduced (and laid out as torch weights) by the caller's grouped launch
  return 0;
}

template <typename T>
int backward_t(const Plan& p, const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, void* ws,
               const float* const (*dscores)[3], const float* djoint, const dta_subnet_grads* grads, double* dalpha,
               int phases, hipStream_t st, const void* x_tiles = nullptr, float* dalpha32 = nullptr,
               const float* gate = nullptr, dta_xchg* xchg = nullptr, long long alpha_slot = -1) {
  // gate (device, per group; year ensembles): gate[g] <= 0 -> group g's score gradient is taken as zero, so every
  // gradient of that group comes out as an exact zero (its launches still run)
  const int G = p.G, B = p.B;
  if (!(phases & 1)) {
    // phase 2 only: the first layer's weight gradient from tensors phase 1 left in the workspace
    WgradReduceGroup reduces;
    if (conv_wgrad_layer<T>(p, d, grads, ws, 0, reduces, st, x_tiles)) return 1;
    return launch_wgrad_reduce_group(reduces, st);
  }
  const float* dsc[MAXG][3] = {};
  BlendBwdArgs blend_fin = {}; bool blend_fin_pending = false;
  WgradArgs pair_conv3; bool pair_pending = false;      // third conv's weight gradient waiting for the second's (paired plan)
  int dsc_mode[MAXG] = {};   // Hang2020: scale mode of the last-head score gradient of each branch
  if (dscores)
    for (int g = 0; g < G; ++g)
      for (int L = 0; L < 3; ++L) dsc[g][L] = dscores[g][L];
  if (d->kind == DTA_NET_HANG2020 && !djoint) {
    // all-heads mode (the Hang et al. multi-head training loss): gradients arrive per classifier head in `dscores`;
    // the sigmoid(alpha) blend is not on the graph, so alpha gets no gradient (dalpha stays as the caller left it)
    if (!dscores) { dta_set_error("Hang2020 backward needs djoint (blended scores) or dscores (per-head mode)"); return 1; }
  } else if (d->kind == DTA_NET_HANG2020) {
    if (!alpha) { dta_set_error("Hang2020 backward needs alpha"); return 1; }
    // the forward kept both branch scores in the workspace unless the caller supplied its own buffers; the
    // Python binding always lets the workspace hold them
    // d(joint)/d(branch score) = sigmoid(alpha) / 1 - sigmoid(alpha): folded into the head GEMMs as an output scale
    // (GemmArgs::sig_mode); d(alpha) is reduced by one extra block of the first GEMM launch below
    BlendBwdArgs bb = {};
    bb.spec = at<float>(ws, p.scores[0][2]); bb.spat = at<float>(ws, p.scores[1][2]);
    bb.alpha = alpha; bb.djoint = djoint; bb.dalpha = dalpha; bb.B = B; bb.classes = p.classes;
    if (dalpha == nullptr) { dta_set_error("Hang2020 backward needs a dalpha destination"); return 1; }
    blend_fin = bb; blend_fin_pending = true;
    dsc[0][2] = djoint; dsc[1][2] = djoint;
    dsc_mode[0] = 1; dsc_mode[1] = 2;
  } else if (d->kind == DTA_NET_VANILLA) {
    dsc[0][2] = djoint ? djoint : dsc[0][2];
  }

  // which levels run the lean stage-backward kernel (they can take / leave 16-bit gradient maps)
  const bool g16 = p.esz == 2 && p.y_fmt == FMT_F16;
  bool lean_lvl[3];
  for (int L = 0; L < 3; ++L) {
    StageBwdArgs probe;
    memset(&probe, 0, sizeof(probe));
    probe.f = stage_args(p, d, nets, ws, L);
    bool any_head = false;
    for (int g = 0; g < G; ++g) any_head |= dsc[g][L] != nullptr && p.F[g][L] > 0;
    probe.dfeat = any_head ? at<float>(ws, p.dfeat[L]) : nullptr;
    probe.dv = at<float>(ws, p.dv[L]);
    lean_lvl[L] = stage_bwd_is_lean(probe, G);
  }
  GemmGroup deferred;   // parameter-gradient GEMMs nothing downstream waits for: one grouped launch at the end
  WgradReduceGroup reduces;   // likewise the split-K reductions of the conv weight gradients
  // data-parallel (torch / RCCL buckets): alpha's finished float64 gradient is rounded ONCE into its fp32 exchange slot
  // by the launch that ends this call (phase 1 or the whole backward)
  if (dalpha32 && blend_fin_pending) { reduces.slot_src = dalpha; reduces.slot_dst = dalpha32; }
  for (int L = 2; L >= 0; --L) {
    const int C = CH[L];
    StageArgs sa = stage_args(p, d, nets, ws, L);
    const size_t fgs = sa.feat_gs;
    // ---- classifier backward -> dfeat, dW, db (gradient buffers arrive zeroed: split-K accumulates) ----
    bool any_head = false;
    GemmGroup dfeat_grp;
    for (int g = 0; g < G; ++g) {
      const int F = p.F[g][L];
      if (!dsc[g][L] || F == 0) continue;
      any_head = true;
      for (int g2 = 0; g2 < G; ++g2)   // a branch without a score gradient at this stage contributes zeros
        if (!dsc[g2][L] && p.F[g2][L] > 0)
          hipMemsetAsync(at<float>(ws, p.dfeat[L]) + (size_t)g2 * fgs, 0, (size_t)B * p.F[g2][L] * 4, st);
      GemmArgs ga;
      memset(&ga, 0, sizeof(ga));
      ga.A = dsc[g][L]; ga.sa_m = p.cls[g]; ga.sa_k = 1;
      ga.Bm = nets[g].fc_w[L]; ga.sb_k = F; ga.sb_n = 1;
      ga.C = at<float>(ws, p.dfeat[L]) + (size_t)g * fgs; ga.sc_m = F; ga.sc_n = 1;
      ga.M = B; ga.N = F; ga.K = p.cls[g]; ga.ksplit = 1;   // plain stores: dfeat needs no clearing
      ga.gate = gate ? gate + g : nullptr;
      if (L == 2 && dsc_mode[g]) { ga.sig_alpha = alpha; ga.sig_mode = dsc_mode[g]; }
      dfeat_grp.add(ga);
      if (grads[g].fc_w[L]) {
        memset(&ga, 0, sizeof(ga));
        ga.A = dsc[g][L]; ga.sa_m = 1; ga.sa_k = p.cls[g];
        ga.Bm = at<float>(ws, p.feat[L]) + (size_t)g * fgs; ga.sb_k = F; ga.sb_n = 1;
        ga.C = grads[g].fc_w[L]; ga.sc_m = F; ga.sc_n = 1;
        ga.M = p.cls[g]; ga.N = F; ga.K = B;
        ga.ksplit = gemm_auto_ksplit(p.cls[g], F, B);
        ga.rowsum_out = grads[g].fc_b[L];      // db[n] = sum_b dscore[b][n]
        ga.gate = gate ? gate + g : nullptr;
        if (L == 2 && dsc_mode[g]) { ga.sig_alpha = alpha; ga.sig_mode = dsc_mode[g]; }
        if (!deferred.add(ga)) { if (launch_gemm_group(deferred, st)) return 1; deferred.n = 0; deferred.add(ga); }
      }
    }
    if (any_head) prof_begin(DTA_SITE_GEMM + 1, st);
    if (blend_fin_pending) {
      if (launch_gemm_group_with_blend_fin(dfeat_grp, blend_fin, st)) return 1;
      blend_fin_pending = false;
    } else if (launch_gemm_group(dfeat_grp, st)) return 1;
    if (any_head) prof_end(DTA_SITE_GEMM + 1, st);
    // ---- attention + pool + ReLU backward ----
    StageBwdArgs sb;
    memset(&sb, 0, sizeof(sb));
    sb.f = sa;
    if (L < 2) { sb.da = at<float>(ws, p.da[L + 1]); sb.da_gs = (size_t)B * p.HWz[L] * C; }
    sb.dfeat = any_head ? at<float>(ws, p.dfeat[L]) : nullptr; sb.dfeat_gs = fgs;
    sb.dv = at<float>(ws, p.dv[L]); sb.dv_gs = (size_t)B * p.HWc[L] * C;
    sb.dv_compact = L > 0 && bn_bwd_apply_uses_lds(C, p.Hc[L], p.Wc[L], p.esz);   // pooled stages: 3/4 of dv is zeros
    // bf16 mode + lean stage kernels: the gradient maps between the backward kernels are stored in bf16 (the conv
    // operand they become is bf16 anyway); `da` was written by the NEXT layer's input-gradient conv in the format that
    // layer chose from lean_lvl[L] (same predicate, evaluated before the loop)
    sb.da_fmt = (L < 2 && lean_lvl[L] && g16) ? FMT_BF16 : FMT_F32;
    sb.dv_fmt = g16 ? FMT_BF16 : FMT_F32;      // (the generic stage kernels store it in bf16 too)
    sb.bnpart = at<float>(ws, p.bnpart[L]); sb.bnpart_gs = (size_t)B * C * 2;
    // DTA_FANIN=2 (experiment, measured no faster than the finalize launch): the batch sums of BatchNorm's backward are
    // folded inside the lean stage launch (FAN_R rows) and the apply launch derives its coefficients from them
    const bool fan_bwd = lean_lvl[L] && (switches().fanin & 2);
    if (fan_bwd) {
      sb.bn_fan_rows = sb.bnpart;      // (the per-patch partials' buffer: a workgroup row needs no more than a patch row)
      sb.bn_fan_sums = at<double>(ws, p.fan_bwd[L]);
      sb.bn_fan_count = at<unsigned>(ws, p.fan_ctr) + (size_t)(3 + L) * MAXG * FAN_R;
    }
    sb.vec = at<float>(ws, p.vec[L]); sb.vec_gs = (size_t)B * p.vec_ld[L]; sb.vec_ld = p.vec_ld[L];
    prof_begin(DTA_SITE_STAGE_BWD + L, st);
    if (launch_stage_bwd(sb, G, st)) return 1;
    prof_end(DTA_SITE_STAGE_BWD + L, st);
    // ---- attention parameter gradients (batch reductions) ----
    ColsumArgs cs_jobs[2]; int ncs_jobs = 0;   // column sums ride in the BatchNorm-finalize launch below
    for (int g = 0; g < G; ++g) {
      const float* vec = at<float>(ws, p.vec[L]) + (size_t)g * sb.vec_gs;
      const int ld = p.vec_ld[L];
      if (p.kinds[g] == KIND_SPECTRAL) {
        const int K = SPEC_K[L];
        for (int which = 0; which < 2; ++which) {   // 0: attention_conv1 (d1 x pooled), 1: attention_conv2 (d2 x h)
          float* gw = grads[g].att[L][which * 2];
          if (!gw) continue;
          GemmArgs ga;
          memset(&ga, 0, sizeof(ga));
          ga.A = vec + (which ? 0 : 2 * C); ga.sa_m = 1; ga.sa_k = ld;
          ga.Bm = vec + (which ? C : 3 * C); ga.sb_k = ld; ga.sb_n = 1;
          ga.C = gw + K / 2; ga.sc_m = (long)C * K; ga.sc_n = K;    // only the centre tap is live
          ga.M = C; ga.N = C; ga.K = B; ga.ksplit = gemm_auto_ksplit(C, C, B);
          ga.rowsum_out = grads[g].att[L][which * 2 + 1];            // bias gradient = sum_b d{1,2}
          if (!deferred.add(ga)) { if (launch_gemm_group(deferred, st)) return 1; deferred.n = 0; deferred.add(ga); }
        }
      } else if (p.kinds[g] == KIND_SPATIAL) {
        const int kk = SPAT_K[L] * SPAT_K[L];
        ColsumArgs cs;
        memset(&cs, 0, sizeof(cs));
        cs.A = vec; cs.rows = B; cs.cols = C + 2 * kk + 3; cs.lda = ld; cs.nseg = 6;
        int offs[6] = {0, C, C + 1, C + 1 + kk, C + 2 + kk, C + 2 + 2 * kk};
        int lens[6] = {C, 1, kk, 1, kk, 1};
        for (int i = 0; i < 6; ++i) { cs.off[i] = offs[i]; cs.len[i] = lens[i]; cs.dst[i] = grads[g].att[L][i]; cs.dst_stride[i] = 1; }
        if (ncs_jobs < 2) cs_jobs[ncs_jobs++] = cs;
        else if (launch_colsum_scatter(cs, st)) return 1;
      }
    }
    // ---- BatchNorm backward ----
    BnBwdFinalizeArgs bf;
    memset(&bf, 0, sizeof(bf));
    bf.bnpart = sb.bnpart; bf.bnpart_gs = sb.bnpart_gs; bf.B = B; bf.C = C; bf.HW = p.HWc[L];
    bf.coef = sa.coef; bf.coef_gs = sa.coef_gs;
    for (int g = 0; g < G; ++g) {
      bf.gamma[g] = nets[g].bn_w[L]; bf.dgamma[g] = grads[g].bn_w[L]; bf.dbeta[g] = grads[g].bn_b[L];
      bf.dconvbias[g] = grads[g].conv_b[L];
    }
    bf.bcoef = at<float>(ws, p.bcoef[L]); bf.bcoef_gs = C * 4; bf.training = d->training;
    if (!fan_bwd && launch_bn_bwd_finalize_colsum(bf, G, cs_jobs, ncs_jobs, st)) return 1;
    BnBwdApplyArgs ap;
    memset(&ap, 0, sizeof(ap));
    ap.dv = sb.dv; ap.dv_gs = sb.dv_gs; ap.y = sa.y; ap.y_gs = sa.y_gs; ap.y_rs = sa.y_rs; ap.y_fmt = sa.y_fmt;
    ap.coef = sa.coef; ap.coef_gs = sa.coef_gs; ap.bcoef = bf.bcoef; ap.bcoef_gs = bf.bcoef_gs;
    ap.B = B; ap.C = C; ap.H = p.Hc[L]; ap.W = p.Wc[L];
    ap.dv_compact = sb.dv_compact; ap.Hz = p.Hz[L]; ap.Wz = p.Wz[L]; ap.dv_fmt = sb.dv_fmt;
    ap.dy_tl = at<char>(ws, p.dy_tl[L]);
    if (L == 0 && p.shared_x) { ap.dy_gs = (size_t)2 * p.Rin[0] * 16; ap.dy_nc = 2 * G; ap.dy_ch0 = 0; }   // group g -> chunks [2g, 2g+2)
    else { ap.dy_gs = (size_t)B * (C / 16) * p.Rin[L] * 16; ap.dy_nc = C / 16; ap.dy_ch0 = 0; }
    ap.dy_compact = p.tl_compact;
    if (fan_bwd) {
      BnBwdFanArgs fa = {};
      fa.fan = sb.bn_fan_sums; fa.training = d->training;
      for (int g = 0; g < G; ++g) { fa.gamma[g] = bf.gamma[g]; fa.dgamma[g] = bf.dgamma[g]; fa.dbeta[g] = bf.dbeta[g]; fa.dconvbias[g] = bf.dconvbias[g]; }
      // (the spatial-attention column sums that rode in the finalize launch ride in this one)
      if (launch_bn_bwd_apply<T>(ap, G, st, &fa, cs_jobs, ncs_jobs)) return 1;
    } else if (launch_bn_bwd_apply<T>(ap, G, st)) return 1;
    // ---- conv weight gradient ----
    // the deferred parameter-gradient GEMMs always ride in the launch of the split-K reductions that ends this call
    // (phase 1 of a data-parallel step: the reductions of layers 3 and 2, so the first gradient bucket is complete
    // when the call returns; the whole backward: all three layers); DTA_NO_TAIL_MERGE=1 flushes them separately
    if (L == 0 && switches().no_tail_merge) {
      prof_begin(DTA_SITE_GEMM + 2, st);
      if (launch_gemm_group(deferred, st)) return 1;
      prof_end(DTA_SITE_GEMM + 2, st);
      deferred.n = 0;
    }
    if (L > 0 || (phases & 2)) {
      bool want1 = false;
      for (int g = 0; g < G; ++g) want1 |= grads[g].conv_w[1] != nullptr;
      if (L == 2 && p.wgrad_pair && want1) {
        if (conv_wgrad_layer<T>(p, d, grads, ws, L, reduces, st, x_tiles, &pair_conv3)) return 1;
        bool want3 = false;
        for (int g = 0; g < G; ++g) want3 |= grads[g].conv_w[2] != nullptr;
        pair_pending = want3;
      } else {
        if (L == 0 && xchg) {
          // the exchange's head bucket (everything but the first conv's weights) must be COMPLETE before the launch whose
          // side workgroups sum it over the ranks: the parameter-gradient GEMMs and the other layers' slab reductions go now
          if (deferred.n > 0 ? launch_gemm_group_with_reduce(deferred, reduces, st, sizeof(T) == 2) : launch_wgrad_reduce_group(reduces, st)) return 1;
          deferred.n = 0; reduces.n = 0;
        }
        if (conv_wgrad_layer<T>(p, d, grads, ws, L, reduces, st, x_tiles, nullptr, (L == 1 && pair_pending) ? &pair_conv3 : nullptr,
                                L == 0 ? xchg : nullptr, dalpha, alpha_slot)) return 1;
      }
    }
    // ---- conv input gradient (feeds the previous stage's gated map) ----
    if (L > 0) {
      PackWArgs pw;
      memset(&pw, 0, sizeof(pw));
      pw.G = G; pw.NC = C / 16; pw.N = CH[L - 1]; pw.K = C; pw.mode = 2;
      for (int g = 0; g < G; ++g) pw.src[g] = nets[g].conv_w[L];
      if (!d->training)   // a training forward already packed the transposed weights into the workspace
        if (launch_pack_conv_w<T>(pw, at<char>(ws, p.wd[L]), st)) return 1;
      ConvArgs ca = dgrad_conv_geom(p, L);
      if (d->training) use_conv_tabs(p, ws, ca, G, conv_tab_slot_dgrad(L));      // (built by the training forward's prep launch)
      ca.pixel_order = switches().conv_order;
      ca.x_tl = ap.dy_tl; ca.x_gs = ap.dy_gs; ca.wp = at<char>(ws, p.wd[L]); ca.x_compact = p.tl_compact;
      ca.y = at<float>(ws, p.da[L]); ca.y_gs = (size_t)B * p.HWc[L] * CH[L - 1]; ca.y_rs = CH[L - 1];
      ca.y_fmt = (lean_lvl[L - 1] && g16) ? FMT_BF16 : FMT_F32;      // read by stage L-1's backward
      prof_begin(DTA_SITE_CONV_DGRAD + L, st);
      if (launch_conv3x3<T>(ca, G, st)) return 1;
      prof_end(DTA_SITE_CONV_DGRAD + L, st);
    }
  }
  if (deferred.n > 0) return launch_gemm_group_with_reduce(deferred, reduces, st, sizeof(T) == 2);
  return launch_wgrad_reduce_group(reduces, st);
}

}  // namespace

extern "C" {

int dta_abi_version(void) { return DTA_ABI_VERSION; }
#ifndef DTA_BUILD_ID
#define DTA_BUILD_ID "unknown"
#endif
const char* dta_build_id(void) { return DTA_BUILD_ID; }

#ifdef DTA_DEV_SWITCHES
int dta_dev_switches_enabled(void) { return 1; }
int dta_dev_reload_switches(void) { g_switches = read_switches(); return 0; }
#else
int dta_dev_switches_enabled(void) { return 0; }
int dta_dev_reload_switches(void) { dta_set_error("dta_dev_reload_switches: this is the product library (no developer switches); load libdta_hip_dev.so"); return 1; }
#endif

int dta_profile_set_stride(int stride) {
  if (stride < 1) { dta_set_error("dta_profile_set_stride: stride must be >= 1"); return 1; }
  g_prof.stride = stride;
  return 0;
}

int dta_profile_enable(int site) {
  if (site < 0) {   // stop timing, forget every site
    g_prof.nslots = 0;
    for (int s = 0; s < Prof::SLOTS; ++s) { g_prof.site[s] = -1; g_prof.n[s] = 0; }
    return 0;
  }
  if (!g_prof.created) {
    for (int s = 0; s < Prof::SLOTS; ++s)
      for (int i = 0; i < Prof::CAP; ++i)
        for (int j = 0; j < 2; ++j)
          if (hipEventCreate(&g_prof.ev[s][i][j]) != hipSuccess) { dta_set_error("hipEventCreate failed"); return 1; }
    g_prof.created = true;
  }
  int s = prof_slot(site);
  if (s < 0) {
    if (g_prof.nslots >= Prof::SLOTS) { dta_set_error("dta_profile_enable: at most %d sites at once", Prof::SLOTS); return 1; }
    s = g_prof.nslots++;
    g_prof.site[s] = site;
  }
  g_prof.n[s] = 0; g_prof.seen[s] = 0; g_prof.armed[s] = false;
  return 0;
}

int dta_profile_collect_site(int site, float* ms, int max) {
  const int s = prof_slot(site);
  if (s < 0) { dta_set_error("dta_profile_collect_site: site %d is not being timed", site); return -1; }
  int n = g_prof.n[s] < max ? g_prof.n[s] : max;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_prof.ev[s][i][1]) != hipSuccess ||
        hipEventElapsedTime(&ms[i], g_prof.ev[s][i][0], g_prof.ev[s][i][1]) != hipSuccess) {
      dta_set_error("profile event readback failed");
      return -1;
    }
  }
  g_prof.n[s] = 0;
  return n;
}

int dta_profile_collect(float* ms, int max) {
  if (g_prof.nslots == 0) return 0;
  return dta_profile_collect_site(g_prof.site[0], ms, max);
}
const char* dta_last_error(void) { return g_err; }

size_t dta_net_workspace_bytes(const dta_net_desc* d) {
  Plan p;
  if (!d || build_plan(d, &p)) return 0;
  return p.total;
}

int dta_net_forward(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, const float* x,
                    void* workspace, float* const scores[2][3], float* joint, void* stream) {
  Plan p;
  if (!d || !nets || !x || !workspace) { dta_set_error("dta_net_forward: null argument"); return 1; }
  if (build_plan(d, &p)) return 1;
  hipStream_t st = (hipStream_t)stream;
  const float* xs[MAXG] = {x, x, x, x};
  if (d->dtype == DTA_BF16) return forward_t<bf16_t>(p, d, nets, alpha, xs, workspace, scores, joint, st);
  if (d->dtype == DTA_F32) return forward_t<float>(p, d, nets, alpha, xs, workspace, scores, joint, st);
  dta_set_error("unknown dtype %d", d->dtype);
  return 1;
}

int dta_net_forward_tiles(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, const void* x_tiles,
                          void* workspace, float* const scores[2][3], float* joint, void* stream) {
  Plan p;
  if (!d || !nets || !x_tiles || !workspace) { dta_set_error("dta_net_forward_tiles: null argument"); return 1; }
  if (build_plan(d, &p)) return 1;
  if (d->dtype != DTA_BF16) { dta_set_error("dta_net_forward_tiles: bf16 mode only"); return 1; }
  const float* xs[MAXG] = {nullptr, nullptr, nullptr, nullptr};
  return forward_t<bf16_t>(p, d, nets, alpha, xs, workspace, scores, joint, (hipStream_t)stream, x_tiles);
}

int dta_net_forward_loss(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, const float* x,
                         const void* x_tiles, void* workspace, const long long* labels, const float* weight, float* joint,
                         float* loss, float* dlogits, float* scratch, void* stream) {
  Plan p;
  if (!d || !nets || (!x && !x_tiles) || !workspace || !labels || !loss || !scratch) { dta_set_error("dta_net_forward_loss: null argument"); return 1; }
  if (d->kind != DTA_NET_HANG2020 && d->kind != DTA_NET_VANILLA) { dta_set_error("dta_net_forward_loss: single-score networks only (Hang2020, vanilla_CNN)"); return 1; }
  if (d->kind == DTA_NET_VANILLA && !joint) { dta_set_error("dta_net_forward_loss: vanilla_CNN's scores are written to `joint`"); return 1; }
  if (build_plan(d, &p)) return 1;
  if (x_tiles && d->dtype != DTA_BF16) { dta_set_error("dta_net_forward_loss: tile input is bf16 mode only"); return 1; }
  dta_net_desc dd = *d;
  if (d->kind == DTA_NET_HANG2020) dd.heads_mask |= DTA_SKIP_BLEND;      // the blend belongs to the loss launch (or the fused tail)
  const LossArgs la = {labels, weight, loss, dlogits, scratch};
  hipStream_t st = (hipStream_t)stream;
  const float* xs[MAXG] = {x, x, x, x};
  if (dd.dtype == DTA_BF16) return forward_t<bf16_t>(p, &dd, nets, alpha, xs, workspace, nullptr, joint, st, x_tiles, nullptr, &la);
  if (dd.dtype == DTA_F32) return forward_t<float>(p, &dd, nets, alpha, xs, workspace, nullptr, joint, st, nullptr, nullptr, &la);
  dta_set_error("unknown dtype %d", d->dtype);
  return 1;
}

int dta_net_backward_tiles(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, const void* x_tiles,
                           void* workspace, const float* const dscores[2][3], const float* djoint,
                           const dta_subnet_grads* grads, double* dalpha, int phases, void* stream) {
  Plan p;
  if (!d || !nets || !x_tiles || !workspace || !grads || !(phases & 3)) { dta_set_error("dta_net_backward_tiles: null argument"); return 1; }
  if (build_plan(d, &p)) return 1;
  if (d->dtype != DTA_BF16) { dta_set_error("dta_net_backward_tiles: bf16 mode only"); return 1; }
  return backward_t<bf16_t>(p, d, nets, alpha, workspace, dscores, djoint, grads, dalpha, phases, (hipStream_t)stream, x_tiles);
}

// ---- year ensemble: `years` spectral networks as the groups of ONE set of launches ----
static int ensemble_desc(const dta_net_desc* d, int years, dta_net_desc* out, Plan* p, const char* who) {
  if (!d || years < 1 || years > MAXG) { dta_set_error("%s: 1..%d years", who, MAXG); return 1; }
  if (d->kind != DTA_NET_SPECTRAL) { dta_set_error("%s: the descriptor's kind must be DTA_NET_SPECTRAL", who); return 1; }
  *out = *d;
  out->heads_mask = 4 | (d->heads_mask & (DTA_FORWARD_ONLY | DTA_REUSE_PACKED));   // the ensemble keeps each year's last head only (reference year.py:30)
  return build_plan(out, p, years);
}

size_t dta_ensemble_workspace_bytes(const dta_net_desc* d, int years) {
  Plan p; dta_net_desc dd;
  if (ensemble_desc(d, years, &dd, &p, "dta_ensemble_workspace_bytes")) return 0;
  return p.total;
}

static int ensemble_forward_impl(const dta_net_desc* d, int years, const dta_subnet_params* nets, const float* const* x,
                                 const float* gate, void* workspace, float* mean_scores, float* kept, void* stream) {
  Plan p; dta_net_desc dd;
  if (!nets || !x || !workspace || !mean_scores) { dta_set_error("dta_ensemble_forward: null argument"); return 1; }
  if (ensemble_desc(d, years, &dd, &p, "dta_ensemble_forward")) return 1;
  for (int g = 0; g < years; ++g)
    if (!x[g]) { dta_set_error("dta_ensemble_forward: null input for year %d", g); return 1; }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dd.dtype == DTA_BF16) rc = forward_t<bf16_t>(p, &dd, nets, nullptr, x, workspace, nullptr, nullptr, st, nullptr, gate);
  else if (dd.dtype == DTA_F32) rc = forward_t<float>(p, &dd, nets, nullptr, x, workspace, nullptr, nullptr, st, nullptr, gate);
  else { dta_set_error("unknown dtype %d", dd.dtype); return 1; }
  if (rc) return rc;
  MeanArgs ma = {};
  for (int g = 0; g < years; ++g) ma.src[g] = at<float>(workspace, p.scores[g][2]);
  ma.n = years; ma.dst = mean_scores; ma.count = (size_t)p.B * p.classes; ma.gate = gate; ma.kept = kept;
  return launch_mean_scores(ma, st);
}

int dta_ensemble_forward_loss(const dta_net_desc* d, int years, const dta_subnet_params* nets, const float* const* x,
                              const float* gate, void* workspace, const long long* labels, const float* weight,
                              float* mean_scores, float* kept, float* loss, float* dscore, float* scratch, void* stream) {
  Plan p; dta_net_desc dd;
  if (!nets || !x || !workspace || !labels || !loss || !scratch) { dta_set_error("dta_ensemble_forward_loss: null argument"); return 1; }
  if (ensemble_desc(d, years, &dd, &p, "dta_ensemble_forward_loss")) return 1;
  for (int g = 0; g < years; ++g)
    if (!x[g]) { dta_set_error("dta_ensemble_forward_loss: null input for year %d", g); return 1; }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dd.dtype == DTA_BF16) rc = forward_t<bf16_t>(p, &dd, nets, nullptr, x, workspace, nullptr, nullptr, st, nullptr, gate);
  else if (dd.dtype == DTA_F32) rc = forward_t<float>(p, &dd, nets, nullptr, x, workspace, nullptr, nullptr, st, nullptr, gate);
  else { dta_set_error("unknown dtype %d", dd.dtype); return 1; }
  if (rc) return rc;
  BlendCeArgs a;
  memset(&a, 0, sizeof(a));
  a.gscale = 1.f;
  for (int g = 0; g < years; ++g) a.src[g] = at<float>(workspace, p.scores[g][2]);
  a.nsrc = years; a.src_gate = gate; a.kept_out = kept; a.joint = mean_scores;
  a.labels = labels; a.weight = weight; a.dlogits = dscore; a.loss = loss; a.rowtmp = scratch; a.B = p.B; a.classes = p.classes;
  return launch_blend_ce(a, st);
}

int dta_ensemble_forward(const dta_net_desc* d, int years, const dta_subnet_params* nets, const float* const* x,
                         void* workspace, float* mean_scores, void* stream) {
  return ensemble_forward_impl(d, years, nets, x, nullptr, workspace, mean_scores, nullptr, stream);
}

int dta_ensemble_forward_gated(const dta_net_desc* d, int years, const dta_subnet_params* nets, const float* const* x,
                               const float* gate, void* workspace, float* mean_scores, float* kept, void* stream) {
  if (!gate) { dta_set_error("dta_ensemble_forward_gated: null gate"); return 1; }
  return ensemble_forward_impl(d, years, nets, x, gate, workspace, mean_scores, kept, stream);
}

int dta_year_flags(const float* const* x, int years, size_t n_per_year, float* flags, float* clear_next, void* stream) {
  if (!x || !flags || flags == clear_next || years < 1 || years > MAXG || n_per_year == 0) { dta_set_error("dta_year_flags: bad argument (1..%d years)", MAXG); return 1; }
  for (int g = 0; g < years; ++g)
    if (!x[g] || ((size_t)x[g] & 15)) { dta_set_error("dta_year_flags: year %d: null or not 16-byte aligned", g); return 1; }
  return launch_year_flags(x, years, n_per_year, flags, clear_next, (hipStream_t)stream);
}

int dta_ensemble_backward(const dta_net_desc* d, int years, const dta_subnet_params* nets, void* workspace,
                          const float* dscore, const dta_subnet_grads* grads, void* stream) {
  return dta_ensemble_backward_phased(d, years, nets, workspace, dscore, grads, 3, stream);
}

int dta_ensemble_backward_phased(const dta_net_desc* d, int years, const dta_subnet_params* nets, void* workspace,
                                 const float* dscore, const dta_subnet_grads* grads, int phases, void* stream) {
  return dta_ensemble_backward_gated(d, years, nets, workspace, dscore, grads, nullptr, phases, stream);
}

int dta_ensemble_backward_gated(const dta_net_desc* d, int years, const dta_subnet_params* nets, void* workspace,
                                const float* dscore, const dta_subnet_grads* grads, const float* gate, int phases,
                                void* stream) {
  Plan p; dta_net_desc dd;
  if (!nets || !workspace || !dscore || !grads || !(phases & 3)) { dta_set_error("dta_ensemble_backward: null argument"); return 1; }
  if (ensemble_desc(d, years, &dd, &p, "dta_ensemble_backward")) return 1;
  const float* dsc[MAXG][3] = {};
  for (int g = 0; g < years; ++g) dsc[g][2] = dscore;   // d(mean)/d(year score) is the same 1/years for every year
  hipStream_t st = (hipStream_t)stream;
  if (dd.dtype == DTA_BF16) return backward_t<bf16_t>(p, &dd, nets, nullptr, workspace, dsc, nullptr, grads, nullptr, phases, st, nullptr, nullptr, gate);
  if (dd.dtype == DTA_F32) return backward_t<float>(p, &dd, nets, nullptr, workspace, dsc, nullptr, grads, nullptr, phases, st, nullptr, nullptr, gate);
  dta_set_error("unknown dtype %d", dd.dtype);
  return 1;
}

int dta_ensemble_backward_xchg(const dta_net_desc* d, int years, const dta_subnet_params* nets, void* workspace,
                               const float* dscore, const dta_subnet_grads* grads, const float* gate, dta_xchg* xchg,
                               void* stream) {
  Plan p; dta_net_desc dd;
  if (!nets || !workspace || !dscore || !grads || !xchg) { dta_set_error("dta_ensemble_backward_xchg: null argument"); return 1; }
  if (ensemble_desc(d, years, &dd, &p, "dta_ensemble_backward_xchg")) return 1;
  const float* dsc[MAXG][3] = {};
  for (int g = 0; g < years; ++g) dsc[g][2] = dscore;
  hipStream_t st = (hipStream_t)stream;
  if (dd.dtype == DTA_BF16) return backward_t<bf16_t>(p, &dd, nets, nullptr, workspace, dsc, nullptr, grads, nullptr, 3, st, nullptr, nullptr, gate, xchg, -1);
  if (dd.dtype == DTA_F32) return backward_t<float>(p, &dd, nets, nullptr, workspace, dsc, nullptr, grads, nullptr, 3, st, nullptr, nullptr, gate, xchg, -1);
  dta_set_error("unknown dtype %d", dd.dtype);
  return 1;
}

// ---- multi-stage step: the levels x years networks of the reference's hierarchical model as the groups of ONE set of
//      launches (reference src/models/multi_stage.py:41-66 one learned_ensemble per level, :277-288 one loss per level) ----
static int multistage_desc(const dta_net_desc* d, int levels, const dta_level* lv, dta_net_desc* out, Plan* p, const char* who) {
  if (!d || !lv || levels < 1 || levels > DTA_MAX_LEVELS) { dta_set_error("%s: 1..%d levels", who, DTA_MAX_LEVELS); return 1; }
  static_assert(DTA_MAX_LEVELS == BLEND_CE_MULTI_MAX, "header and kernel disagree");
  if (d->kind != DTA_NET_SPECTRAL) { dta_set_error("%s: the descriptor's kind must be DTA_NET_SPECTRAL", who); return 1; }
  int cls[MAXG] = {}, G = 0;
  for (int l = 0; l < levels; ++l) {
    if (lv[l].count < 1 || lv[l].first != G) { dta_set_error("%s: level %d: its groups must be [first, first + count) with count >= 1, levels in order and adjacent", who, l); return 1; }
    if (G + lv[l].count > MAXG) { dta_set_error("%s: at most %d networks (levels x kept years) per step", who, MAXG); return 1; }
    if (lv[l].classes < 1) { dta_set_error("%s: level %d has %d classes", who, l, lv[l].classes); return 1; }
    for (int k = 0; k < lv[l].count; ++k) cls[G++] = lv[l].classes;
  }
  *out = *d;
  out->heads_mask = 4 | (d->heads_mask & (DTA_FORWARD_ONLY | DTA_REUSE_PACKED));   // each year's last head only (reference year.py:30)
  out->classes = cls[0];
  return build_plan(out, p, G, cls);
}

size_t dta_multistage_workspace_bytes(const dta_net_desc* d, int levels, const dta_level* lv) {
  Plan p; dta_net_desc dd;
  if (multistage_desc(d, levels, lv, &dd, &p, "dta_multistage_workspace_bytes")) return 0;
  return p.total;
}

int dta_multistage_forward_loss(const dta_net_desc* d, int levels, const dta_level* lv, const dta_subnet_params* nets,
                                const float* const* x, const float* gate, void* workspace, void* stream) {
  Plan p; dta_net_desc dd;
  if (!nets || !x || !workspace) { dta_set_error("dta_multistage_forward_loss: null argument"); return 1; }
  if (multistage_desc(d, levels, lv, &dd, &p, "dta_multistage_forward_loss")) return 1;
  for (int g = 0; g < p.G; ++g)
    if (!x[g]) { dta_set_error("dta_multistage_forward_loss: null input for network %d", g); return 1; }
  for (int l = 0; l < levels; ++l)
    if (!lv[l].labels || !lv[l].loss || !lv[l].scratch) { dta_set_error("dta_multistage_forward_loss: level %d: labels, loss and scratch are required", l); return 1; }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dd.dtype == DTA_BF16) rc = forward_t<bf16_t>(p, &dd, nets, nullptr, x, workspace, nullptr, nullptr, st, nullptr, gate);
  else if (dd.dtype == DTA_F32) rc = forward_t<float>(p, &dd, nets, nullptr, x, workspace, nullptr, nullptr, st, nullptr, gate);
  else { dta_set_error("unknown dtype %d", dd.dtype); return 1; }
  if (rc) return rc;
  BlendCeMulti m;
  memset(&m, 0, sizeof(m));
  m.n = levels;
  for (int l = 0; l < levels; ++l) {
    BlendCeArgs& a = m.a[l];
    a.gscale = 1.f;
    for (int k = 0; k < lv[l].count; ++k) a.src[k] = at<float>(workspace, p.scores[lv[l].first + k][2]);
    a.nsrc = lv[l].count; a.src_gate = gate ? gate + lv[l].first : nullptr; a.kept_out = lv[l].kept; a.joint = lv[l].mean_scores;
    a.labels = lv[l].labels; a.weight = lv[l].weight; a.dlogits = lv[l].dscore; a.loss = lv[l].loss; a.rowtmp = lv[l].scratch;
    a.B = p.B; a.classes = lv[l].classes;
  }
  return launch_blend_ce_multi(m, st);
}

int dta_multistage_forward(const dta_net_desc* d, int levels, const dta_level* lv, const dta_subnet_params* nets,
                           const float* const* x, const float* gate, void* workspace, void* stream) {
  Plan p; dta_net_desc dd;
  if (!nets || !x || !workspace) { dta_set_error("dta_multistage_forward: null argument"); return 1; }
  if (multistage_desc(d, levels, lv, &dd, &p, "dta_multistage_forward")) return 1;
  for (int g = 0; g < p.G; ++g)
    if (!x[g]) { dta_set_error("dta_multistage_forward: null input for network %d", g); return 1; }
  for (int l = 0; l < levels; ++l)
    if (!lv[l].mean_scores) { dta_set_error("dta_multistage_forward: level %d has no score output", l); return 1; }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dd.dtype == DTA_BF16) rc = forward_t<bf16_t>(p, &dd, nets, nullptr, x, workspace, nullptr, nullptr, st, nullptr, gate);
  else if (dd.dtype == DTA_F32) rc = forward_t<float>(p, &dd, nets, nullptr, x, workspace, nullptr, nullptr, st, nullptr, gate);
  else { dta_set_error("unknown dtype %d", dd.dtype); return 1; }
  if (rc) return rc;
  for (int l = 0; l < levels; ++l) {      // each level's mean over its kept years (reference year.py:33)
    MeanArgs ma = {};
    for (int k = 0; k < lv[l].count; ++k) ma.src[k] = at<float>(workspace, p.scores[lv[l].first + k][2]);
    ma.n = lv[l].count; ma.dst = lv[l].mean_scores; ma.count = (size_t)p.B * lv[l].classes;
    ma.gate = gate ? gate + lv[l].first : nullptr; ma.kept = lv[l].kept;
    if (launch_mean_scores(ma, st)) return 1;
  }
  return 0;
}

int dta_multistage_predict(const dta_net_desc* d, int levels, const dta_level* lv, const dta_subnet_params* nets,
                           const float* const* x, const float* gate, void* workspace, float* const* probs,
                           long long* const* top_idx, float* const* top_score, void* stream) {
  Plan p; dta_net_desc dd;
  if (!nets || !x || !workspace || !top_idx || !top_score) { dta_set_error("dta_multistage_predict: null argument"); return 1; }
  if (multistage_desc(d, levels, lv, &dd, &p, "dta_multistage_predict")) return 1;
  for (int g = 0; g < p.G; ++g)
    if (!x[g]) { dta_set_error("dta_multistage_predict: null input for network %d", g); return 1; }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dd.dtype == DTA_BF16) rc = forward_t<bf16_t>(p, &dd, nets, nullptr, x, workspace, nullptr, nullptr, st, nullptr, gate);
  else if (dd.dtype == DTA_F32) rc = forward_t<float>(p, &dd, nets, nullptr, x, workspace, nullptr, nullptr, st, nullptr, gate);
  else { dta_set_error("unknown dtype %d", dd.dtype); return 1; }
  if (rc) return rc;
  SoftmaxMulti m;
  memset(&m, 0, sizeof(m));
  m.n = levels; m.B = p.B;
  for (int l = 0; l < levels; ++l) {
    SoftmaxLevel& a = m.lv[l];
    for (int k = 0; k < lv[l].count; ++k) a.src[k] = at<float>(workspace, p.scores[lv[l].first + k][2]);
    a.nsrc = lv[l].count; a.gate = gate ? gate + lv[l].first : nullptr; a.mean_out = lv[l].mean_scores; a.classes = lv[l].classes;
    a.probs = probs ? probs[l] : nullptr; a.top_idx = top_idx[l]; a.top_score = top_score[l];
  }
  return launch_softmax_top2_multi(m, st);
}

int dta_multistage_backward(const dta_net_desc* d, int levels, const dta_level* lv, const dta_subnet_params* nets,
                            void* workspace, const dta_subnet_grads* grads, const float* gate, void* stream) {
  Plan p; dta_net_desc dd;
  if (!nets || !workspace || !grads) { dta_set_error("dta_multistage_backward: null argument"); return 1; }
  if (multistage_desc(d, levels, lv, &dd, &p, "dta_multistage_backward")) return 1;
  const float* dsc[MAXG][3] = {};
  for (int l = 0; l < levels; ++l) {
    if (!lv[l].dscore) { dta_set_error("dta_multistage_backward: level %d has no score gradient", l); return 1; }
    for (int k = 0; k < lv[l].count; ++k) dsc[lv[l].first + k][2] = lv[l].dscore;   // d(mean)/d(year score) is the same for every kept year
  }
  hipStream_t st = (hipStream_t)stream;
  if (dd.dtype == DTA_BF16) return backward_t<bf16_t>(p, &dd, nets, nullptr, workspace, dsc, nullptr, grads, nullptr, 3, st, nullptr, nullptr, gate);
  if (dd.dtype == DTA_F32) return backward_t<float>(p, &dd, nets, nullptr, workspace, dsc, nullptr, grads, nullptr, 3, st, nullptr, nullptr, gate);
  dta_set_error("unknown dtype %d", dd.dtype);
  return 1;
}

int dta_net_backward(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, void* workspace,
                     const float* const dscores[2][3], const float* djoint, const dta_subnet_grads* grads,
                     double* dalpha, int phases, void* stream) {
  Plan p;
  if (!d || !nets || !workspace || !grads || !(phases & 3)) { dta_set_error("dta_net_backward: null argument"); return 1; }
  if (build_plan(d, &p)) return 1;
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == DTA_BF16) return backward_t<bf16_t>(p, d, nets, alpha, workspace, dscores, djoint, grads, dalpha, phases, st);
  if (d->dtype == DTA_F32) return backward_t<float>(p, d, nets, alpha, workspace, dscores, djoint, grads, dalpha, phases, st);
  dta_set_error("unknown dtype %d", d->dtype);
  return 1;
}

int dta_net_backward_dp(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, const void* x_tiles,
                        void* workspace, const float* const dscores[2][3], const float* djoint,
                        const dta_subnet_grads* grads, double* dalpha, float* dalpha_f32, int phases, void* stream) {
  Plan p;
  if (!d || !nets || !workspace || !grads || !(phases & 3)) { dta_set_error("dta_net_backward_dp: null argument"); return 1; }
  if (build_plan(d, &p)) return 1;
  hipStream_t st = (hipStream_t)stream;
  if (x_tiles && d->dtype != DTA_BF16) { dta_set_error("dta_net_backward_dp: tile input is bf16 mode only"); return 1; }
  if (d->dtype == DTA_BF16) return backward_t<bf16_t>(p, d, nets, alpha, workspace, dscores, djoint, grads, dalpha, phases, st, x_tiles, dalpha_f32);
  if (d->dtype == DTA_F32) return backward_t<float>(p, d, nets, alpha, workspace, dscores, djoint, grads, dalpha, phases, st, nullptr, dalpha_f32);
  dta_set_error("unknown dtype %d", d->dtype);
  return 1;
}

int dta_net_backward_xchg(const dta_net_desc* d, const dta_subnet_params* nets, const double* alpha, const void* x_tiles,
                          void* workspace, const float* const dscores[2][3], const float* djoint,
                          const dta_subnet_grads* grads, double* dalpha, dta_xchg* xchg, long long alpha_slot, void* stream) {
  Plan p;
  if (!d || !nets || !workspace || !grads || !xchg) { dta_set_error("dta_net_backward_xchg: null argument"); return 1; }
  if (build_plan(d, &p)) return 1;
  hipStream_t st = (hipStream_t)stream;
  if (x_tiles && d->dtype != DTA_BF16) { dta_set_error("dta_net_backward_xchg: tile input is bf16 mode only"); return 1; }
  if (d->dtype == DTA_BF16) return backward_t<bf16_t>(p, d, nets, alpha, workspace, dscores, djoint, grads, dalpha, 3, st, x_tiles, nullptr, nullptr, xchg, alpha_slot);
  if (d->dtype == DTA_F32) return backward_t<float>(p, d, nets, alpha, workspace, dscores, djoint, grads, dalpha, 3, st, nullptr, nullptr, nullptr, xchg, alpha_slot);
  dta_set_error("unknown dtype %d", d->dtype);
  return 1;
}

int dta_net_loss(const dta_net_desc* d, const double* alpha, void* workspace, const long long* labels, const float* weight,
                 float* joint, float* loss, float* dlogits, float* scratch, void* stream) {
  Plan p;
  if (!d || !workspace || !labels || !loss || !scratch) { dta_set_error("dta_net_loss: null argument"); return 1; }
  if (build_plan(d, &p)) return 1;
  BlendCeArgs a;
  memset(&a, 0, sizeof(a));
  a.gscale = 1.f;
  if (d->kind == DTA_NET_HANG2020) {
    if (!alpha) { dta_set_error("dta_net_loss: Hang2020 needs alpha"); return 1; }
    a.spec = at<float>(workspace, p.scores[0][2]); a.spat = at<float>(workspace, p.scores[1][2]); a.alpha = alpha; a.joint = joint;
  } else {
    if (!joint) { dta_set_error("dta_net_loss: the scores of a single-branch network are passed in `joint`"); return 1; }
    a.spec = joint; a.joint = joint;
  }
  a.labels = labels; a.weight = weight; a.dlogits = dlogits; a.loss = loss; a.rowtmp = scratch; a.B = p.B; a.classes = p.classes;
  return launch_blend_ce(a, (hipStream_t)stream);
}

int dta_weighted_ce(const float* logits, const long long* labels, const float* weight, int batch, int classes,
                    float* loss, float* dlogits, float* scratch, void* stream) {
  if (!logits || !labels || !loss || !scratch || batch < 1 || classes < 1) { dta_set_error("dta_weighted_ce: bad argument"); return 1; }
  CeArgs a;
  a.logits = logits; a.labels = labels; a.weight = weight; a.dlogits = dlogits; a.loss = loss; a.rowtmp = scratch;
  a.B = batch; a.classes = classes;
  return launch_weighted_ce(a, (hipStream_t)stream);
}

static int weighted_ce_scaled_impl(const float* logits, const long long* labels, const float* weight, int batch, int classes,
                                   float grad_scale, const float* grad_scale_dev, float* loss, float* dlogits, float* scratch,
                                   void* stream) {
  if (!logits || !labels || !loss || !scratch || batch < 1 || classes < 1) { dta_set_error("dta_weighted_ce_scaled: bad argument"); return 1; }
  BlendCeArgs a;
  a.spec = logits; a.spat = nullptr; a.alpha = nullptr; a.joint = nullptr;
  a.labels = labels; a.weight = weight; a.dlogits = dlogits; a.loss = loss; a.rowtmp = scratch;
  a.B = batch; a.classes = classes; a.gscale = grad_scale; a.gscale_dev = grad_scale_dev;
  return launch_blend_ce(a, (hipStream_t)stream);
}
int dta_weighted_ce_scaled(const float* logits, const long long* labels, const float* weight, int batch, int classes,
                           float grad_scale, float* loss, float* dlogits, float* scratch, void* stream) {
  return weighted_ce_scaled_impl(logits, labels, weight, batch, classes, grad_scale, nullptr, loss, dlogits, scratch, stream);
}
int dta_weighted_ce_scaled_dev(const float* logits, const long long* labels, const float* weight, int batch, int classes,
                               const float* grad_scale_dev, float* loss, float* dlogits, float* scratch, void* stream) {
  if (!grad_scale_dev) { dta_set_error("dta_weighted_ce_scaled_dev: null scale"); return 1; }
  return weighted_ce_scaled_impl(logits, labels, weight, batch, classes, 1.f, grad_scale_dev, loss, dlogits, scratch, stream);
}

int dta_softmax_top2(const float* logits, int batch, int classes, float* probs, long long* top_idx, float* top_score,
                     void* stream) {
  if (!logits || !top_idx || !top_score || batch < 1 || classes < 2) { dta_set_error("dta_softmax_top2: bad argument"); return 1; }
  return launch_softmax_top2(logits, batch, classes, probs, top_idx, top_score, (hipStream_t)stream);
}

static int adam_step_impl(float* p, const float* g, float* gz, float* m, float* v, size_t n, double* alpha_p,
                          const double* alpha_g, double* alpha_gz, double* alpha_m, double* alpha_v, int step, float lr,
                          float beta1, float beta2, float eps, float grad_scale, void* stream,
                          const float* active = nullptr, const int* dev_step = nullptr, const float* alpha_g32 = nullptr,
                          int* dev_step_out = nullptr) {
  if ((step < 1 && !active) || (n && (!p || !g || !m || !v))) { dta_set_error("dta_adam_step: bad argument"); return 1; }
  AdamArgs a;
  a.active = active; a.dev_step = dev_step; a.dev_step_out = dev_step_out;
  a.g_inactive = active ? const_cast<float*>(g) : nullptr;
  if (active && !dev_step) { dta_set_error("dta_adam_step_gated: needs a device step counter"); return 1; }
  if (alpha_g32 && (alpha_g32 < g || alpha_g32 >= g + n)) { dta_set_error("dta_adam_step_dp: alpha's exchange slot must lie inside g"); return 1; }
  if (step < 1) step = 1;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.gz = gz; a.alpha_gz = alpha_gz;
  a.alpha_p = alpha_p; a.alpha_g = alpha_g; a.alpha_m = alpha_m; a.alpha_v = alpha_v; a.alpha_g32 = alpha_g32;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale = grad_scale;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));   // O(1): a long run must not become host-bound
  a.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  return launch_adam(a, (hipStream_t)stream);
}

int dta_adam_step(float* p, const float* g, float* m, float* v, size_t n, double* alpha_p, const double* alpha_g,
                  double* alpha_m, double* alpha_v, int step, float lr, float beta1, float beta2, float eps,
                  float grad_scale, void* stream) {
  return adam_step_impl(p, g, nullptr, m, v, n, alpha_p, alpha_g, nullptr, alpha_m, alpha_v, step, lr, beta1, beta2, eps,
                        grad_scale, stream);
}

int dta_adam_step_gated(float* p, float* g, float* m, float* v, size_t n, const float* active, const int* dev_step,
                        int* dev_step_next, float lr, float beta1, float beta2, float eps, float grad_scale, int zero_grad,
                        void* stream) {
  if (!active) { dta_set_error("dta_adam_step_gated: null gate"); return 1; }
  if (dev_step_next == dev_step) { dta_set_error("dta_adam_step_gated: dev_step_next must be a different word than dev_step"); return 1; }
  return adam_step_impl(p, g, zero_grad ? g : nullptr, m, v, n, nullptr, nullptr, nullptr, nullptr, nullptr, 0, lr, beta1, beta2, eps,
                        grad_scale, stream, active, dev_step, nullptr, dev_step_next);
}

int dta_adam_step_multi(int nseg, const dta_adam_segment* segs, float lr, float beta1, float beta2, float eps, float grad_scale,
                        int zero_grad, void* stream) {
  if (nseg < 1 || nseg > DTA_ADAM_MAX_SEGMENTS || !segs) { dta_set_error("dta_adam_step_multi: 1..%d segments", DTA_ADAM_MAX_SEGMENTS); return 1; }
  static_assert(DTA_ADAM_MAX_SEGMENTS == ADAM_MAX_SEG, "header and kernel disagree");
  AdamMulti mm;
  memset(&mm, 0, sizeof(mm));
  mm.n = nseg;
  for (int i = 0; i < nseg; ++i) {
    const dta_adam_segment& sgm = segs[i];
    if (sgm.n && (!sgm.p || !sgm.g || !sgm.m || !sgm.v)) { dta_set_error("dta_adam_step_multi: segment %d: null buffer", i); return 1; }
    if (sgm.active ? !sgm.dev_step : sgm.step < 1) { dta_set_error("dta_adam_step_multi: segment %d: a gated segment needs a device step counter, an ungated one a step count >= 1", i); return 1; }
    if (sgm.dev_step_next && sgm.dev_step_next == sgm.dev_step) { dta_set_error("dta_adam_step_multi: segment %d: dev_step_next must be a different word than dev_step", i); return 1; }
    AdamArgs& a = mm.seg[i];
    a.p = sgm.p; a.g = sgm.g; a.m = sgm.m; a.v = sgm.v; a.n = sgm.n;
    a.gz = zero_grad ? sgm.g : nullptr;
    a.active = sgm.active; a.dev_step = sgm.dev_step; a.dev_step_out = sgm.dev_step_next;
    a.g_inactive = sgm.active ? sgm.g : nullptr;
    a.lr = sgm.lr > 0.f ? sgm.lr : lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale = grad_scale;
    const int step = sgm.active ? 1 : sgm.step;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  }
  return launch_adam_multi(mm, (hipStream_t)stream);
}

int dta_adam_step_zero_grad(float* p, float* g, float* m, float* v, size_t n, double* alpha_p, double* alpha_g,
                            double* alpha_m, double* alpha_v, int step, float lr, float beta1, float beta2, float eps,
                            float grad_scale, void* stream) {
  return adam_step_impl(p, g, g, m, v, n, alpha_p, alpha_g, alpha_g, alpha_m, alpha_v, step, lr, beta1, beta2, eps,
                        grad_scale, stream);
}

int dta_adam_step_dp(float* p, float* g, float* m, float* v, size_t n, double* alpha_p, double* alpha_g,
                     const float* alpha_g_f32, double* alpha_m, double* alpha_v, int step, float lr, float beta1, float beta2,
                     float eps, float grad_scale, int zero_grad, void* stream) {
  return adam_step_impl(p, g, zero_grad ? g : nullptr, m, v, n, alpha_p, alpha_g, zero_grad ? alpha_g : nullptr, alpha_m, alpha_v,
                        step, lr, beta1, beta2, eps, grad_scale, stream, nullptr, nullptr, alpha_g_f32);
}

}  // extern "C"
