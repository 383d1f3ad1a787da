// Device code shared by the loss kernels (heads.hip: k_blend_ce) and the fused forward tail (stage.hip: k_tail_fwd): the
// Hang2020 blend and the class-weighted cross-entropy with its gradient and the last-arriver loss sum.  ONE definition, so
// that the loss of the fused step (tail kernel), of the module path (optim.cross_entropy -> k_blend_ce) and of dta_net_loss
// are the same sequence of float operations: same scores in, same bits out.
#pragma once
#include "kernels.h"

namespace dta {

// The Hang2020 blend (reference Hang2020.py:260-261) exactly as torch evaluates it: sigmoid and 1 - sigmoid in double (alpha
// is a float64 0-dim tensor), each rounded to float when it meets the float32 scores, then two products and a sum with
// no fused multiply-add.  ONE definition for every kernel that blends, so that the stand-alone blend (module path) and
// the blend folded into the loss kernel (fused path) produce the same bits.
__device__ __forceinline__ float blend2(float zs, float zt, float w, float w1) {
#pragma clang fp contract(off)      // (hipcc contracts a * b + c into an FMA by default, site by site; HIP's __fmul_rn is a plain product)
  const float ps = zs * w, pt = zt * w1;
  return ps + pt;
}

// Sum over the first four waves of a workgroup of >= 256 threads (the other waves must pass 0): every thread gets
// ((s0 + s1) + s2) + s3 -- the order block_sum256 uses.  scratch: 8 floats.
__device__ __forceinline__ float ce_block_sum(float v, float* scratch) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && threadIdx.x < 512) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// Blend (optional) + weighted CE + gradient for rows 4 * blockIdx.x .. + 3 (one wave per row: waves 0..3) + the loss by
// the last workgroup to arrive.  Call with EVERY thread of a workgroup of 256 or more threads (waves >= 4 only take part in
// the barriers); a.spec / a.spat may point anywhere a flat load reaches (global memory, or this workgroup's LDS rows
// offset so that row r sits at a.spec + r * classes).  sc: 8 floats, sd: 256 doubles, is_last: one int (all LDS).
// What the loss needs that does NOT depend on the scores: this thread's share of the normaliser sum_i w[y_i], its row's
// label and class weight, alpha.  A caller whose scores arrive late (the fused forward tail) fetches these at kernel entry
// (ce_prefetch) and hands them to blend_ce_body: two dependent global round trips leave the end of its critical path.
struct CePre { float part; long long y; float wy; double alpha0; };
// the same loads and the same additions, in the same order, as blend_ce_body performs them itself
__device__ __forceinline__ void ce_prefetch(const BlendCeArgs& a, CePre& pre) {
  const int t = threadIdx.x;
  const bool worker = t < 256;
  const int row = worker ? (int)blockIdx.x * 4 + (t >> 6) : a.B;
  const int rowc = !worker ? (int)blockIdx.x * 4 : (row < a.B ? row : a.B - 1);
  pre.alpha0 = a.spat ? a.alpha[0] : 0.0;
  pre.y = a.labels[rowc];
  long long yy0[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) yy0[k] = (worker && t + 256 * k < a.B) ? a.labels[t + 256 * k] : -1;
  const bool ok = pre.y >= 0 && pre.y < a.classes;
  pre.wy = ok ? (a.weight ? a.weight[pre.y] : 1.f) : 0.f;
  float part = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (yy0[k] >= 0 && yy0[k] < a.classes) part += a.weight ? a.weight[yy0[k]] : 1.f;
  for (int i0 = t + 1024; worker && i0 < a.B; i0 += 1024) {
    long long yy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) yy[k] = (i0 + 256 * k < a.B) ? a.labels[i0 + 256 * k] : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (yy[k] >= 0 && yy[k] < a.classes) part += a.weight ? a.weight[yy[k]] : 1.f;
  }
  pre.part = part;
}

__device__ __forceinline__ void blend_ce_body(const BlendCeArgs& a, float* sc, double* sd, int* is_last, const CePre* pre = nullptr) {
  const int t = threadIdx.x, lane = t & 63;
  const bool worker = t < 256;                       // (wider workgroups: the other waves only keep the barriers company)
  const int row = worker ? (int)blockIdx.x * 4 + (t >> 6) : a.B;
  // normaliser sum_i w[y_i] (every block needs it for its gradient rows): label loads of four strides in flight at
  // once, then their weight gathers -- two dependent round trips per 1024 labels instead of eight
  // this wave's row first: its score loads, label and blend weight go out together with the normaliser's label loads
  // (behind the normaliser's barrier they would be one more dependent round trip of an all-latency launch)
  const int rowc = !worker ? (int)blockIdx.x * 4 : (row < a.B ? row : a.B - 1);      // (always one of THIS workgroup's rows: LDS-resident callers)
  const double alpha0 = pre ? pre->alpha0 : (a.spat ? a.alpha[0] : 0.0);
  const float* zs = a.spec + (size_t)rowc * a.classes;
  const float* zt = a.spat ? a.spat + (size_t)rowc * a.classes : nullptr;
  // year ensemble: mean over the kept sources, the arithmetic of k_mean_scores (heads.hip)
  float mkept = 0.f;
  unsigned muse = 0u;
#pragma unroll
  for (int k = 0; k < MAXG; ++k)
    if (k < a.nsrc && (!a.src_gate || a.src_gate[k] > 0.f)) { muse |= 1u << k; mkept += 1.f; }
  const float minv = 1.f / mkept;
  auto mean_at = [&](int n) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < MAXG; ++k)
      if ((muse >> k) & 1u) acc += a.src[k][(size_t)rowc * a.classes + n];
    return acc * minv;
  };
  if (a.nsrc > 0 && a.kept_out && blockIdx.x == 0 && t == 0) { a.kept_out[0] = mkept; a.kept_out[1] = minv; }
  float zsr[4], ztr[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int n = lane + 64 * k;
    zsr[k] = n < a.classes ? (a.nsrc > 0 ? mean_at(n) : zs[n]) : 0.f;
    ztr[k] = (zt && n < a.classes) ? zt[n] : 0.f;
  }
  const long long y = pre ? pre->y : a.labels[rowc];
  // the normaliser's first 1024 labels (usually all of them) are requested here, their weight gathers after the row's
  // softmax below: the row arithmetic runs under the sweep's two dependent round trips instead of behind its barrier
  long long yy0[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) yy0[k] = (!pre && worker && t + 256 * k < a.B) ? a.labels[t + 256 * k] : -1;
  const bool ok = y >= 0 && y < a.classes;
  const double wd = 1.0 / (1.0 + exp(-alpha0));
  const float w = (float)wd, w1 = (float)(1.0 - wd);
  auto zval = [&](int n) { return a.nsrc > 0 ? mean_at(n) : (zt ? blend2(zs[n], zt[n], w, w1) : zs[n]); };
  // the first 256 classes of the row live in registers (four per lane); wider rows re-read the rest
  float zc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int n = lane + 64 * k; zc[k] = n < a.classes ? (zt ? blend2(zsr[k], ztr[k], w, w1) : zsr[k]) : -3.4e38f; }
  auto zget = [&](int n, int k) { return k < 4 ? zc[k] : zval(n); };
  float mx = -3.4e38f;
  for (int n = lane, k = 0; n < a.classes; n += 64, ++k) mx = fmaxf(mx, zget(n, k));
  mx = wave_max(mx);
  float se = 0.f;
  for (int n = lane, k = 0; n < a.classes; n += 64, ++k) se += __expf(zget(n, k) - mx);
  se = wave_sum(se);
  const float lse = __logf(se);
  const float wy = pre ? pre->wy : (ok ? (a.weight ? a.weight[y] : 1.f) : 0.f);
  float part = pre ? pre->part : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (yy0[k] >= 0 && yy0[k] < a.classes) part += a.weight ? a.weight[yy0[k]] : 1.f;
  for (int i0 = t + 1024; !pre && worker && i0 < a.B; i0 += 1024) {
    long long yy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) yy[k] = (i0 + 256 * k < a.B) ? a.labels[i0 + 256 * k] : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (yy[k] >= 0 && yy[k] < a.classes) part += a.weight ? a.weight[yy[k]] : 1.f;
  }
  const float poison = (ok || y == -100) ? 0.f : __builtin_nanf("");
  float xold = 0.f;
  if (row < a.B && lane == 0) {
    // device-scope exchange: performed at the coherence point (the per-XCD L2s are not coherent for plain stores); issued
    // before the normaliser's reduction, its return value consumed after it (the wave must not reach the counter below
    // before the exchange HAS been performed)
    xold = __hip_atomic_exchange(a.rowtmp + row, ok ? wy * (lse + mx - zval((int)y)) : poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const float den = ce_block_sum(part, sc);
  if (row < a.B) {
    float* jo = a.joint ? a.joint + (size_t)row * a.classes : nullptr;
    // device-decided factor (1 / kept years): an infinite one says this rank kept NO year -- its scores are NaN (an empty
    // mean, as the reference raises there) and it must contribute nothing to a data-parallel gradient sum: exact zeros
    const float gs = a.nsrc > 0 ? minv : (a.gscale_dev ? a.gscale_dev[0] : a.gscale);
    const bool none_kept = (a.nsrc > 0 || a.gscale_dev) && !(gs < 3.0e38f);
    const float sc2 = (den > 0.f ? wy / den : 0.f) * gs + poison;
    for (int n = lane, k = 0; n < a.classes; n += 64, ++k) {
      const float z = zget(n, k);
      if (jo && (a.nsrc > 0 || jo != zs)) jo[n] = z;
      if (a.dlogits) {
        float dv = none_kept ? 0.f : sc2 * (__expf(z - mx - lse) - ((ok && n == (int)y) ? 1.f : 0.f));
        if (a.relu_mask && !(z > 0.f)) dv = 0.f;      // the scores are a ReLU's output: the gradient w.r.t. its input
        a.dlogits[(size_t)row * a.classes + n] = dv;
      }
    }
  }
  asm volatile("" ::"v"(xold));
  // the last block to arrive sums the row terms (fixed order) into the loss.  No fence: a device-scope release would
  // write back this XCD's whole L2 (measured: 28 us for this kernel); the row terms and the counter are all device-scope
  // atomics, ordered by the waits above and the barrier
  __syncthreads();
  unsigned* counter = reinterpret_cast<unsigned*>(a.rowtmp + a.B + 1);
  // (-DDTA_STRICT_ORDER: the same hand-over by the letter of the memory model -- a release/acquire pair on the counter,
  //  i.e. an L2 write-back per block; for ports to targets whose device-scope atomics are not performed at a common
  //  coherence point.  tests/test_round2_gpu.py compares the fused loss with the three-launch route over many launches.)
#ifdef DTA_STRICT_ORDER
  constexpr int CNT_ORDER = __ATOMIC_ACQ_REL;
#else
  constexpr int CNT_ORDER = __ATOMIC_RELAXED;
#endif
  if (t == 0) *is_last = (__hip_atomic_fetch_add(counter, 1u, CNT_ORDER, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!*is_last) return;
  double acc = 0;
  for (int r = t; worker && r < a.B; r += 256) acc += (double)__hip_atomic_load(a.rowtmp + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (worker) sd[t] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) sd[t] += sd[t + o]; __syncthreads(); }
  if (t == 0) {
    a.loss[0] = (float)(sd[0] / (double)den);
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
  }
}

}  // namespace dta
