// Peer gradient exchange for the data-parallel train step (one process per GPU of ONE node): the sum of the flat
// gradient over ranks + Adam in ONE launch on the compute stream, through IPC-mapped peer memory (xGMI), without a
// collective library, side streams or stream-event hops.
//
// Replaces what Lightning's DDP does for the reference between loss.backward() and optimizer.step()
// (/root/reference/train.py:89-98: every parameter's gradient is all-reduced; src/main.py:136 Adam).
//
// Protocol of one step (epoch e), every rank runs the same kernel with G workgroups:
//   0. announce: rank r writes ready[r] = e into every peer's signal page.  The gradient buffer was completed by the
//      kernels before this one on the same stream, so it is in memory (kernel-boundary write-back).
//   1. every workgroup waits until all ready[*] >= e in its OWN page (peers write, the owner polls locally).
//   A. reduce-scatter by pulling: rank r sums shard r of every rank's gradient buffer in rank order 0..N-1 (one fixed
//      order for everybody: the sums, and so the replicas, are bit-identical) and stores the sums in its staging area
//      (uncached memory of its signal allocation, system-scope stores).  Workgroup j owns the same relative slice of
//      every shard in both phases, so it publishes done[r][j] = e to every page as soon as ITS slice is stored.
//   B. all-gather by pulling, fused with the optimizer: workgroup j waits for done[s][j] >= e of every rank s, pulls
//      slice j of every summed shard (one load per rank in flight, starting at rank r+1 so that the N ranks use the
//      N-1 links of the mesh at the same time) and applies Adam to the local parameters / moments (all ranks hold the
//      full optimizer state, as under DDP); the gradient elements are cleared (or replaced by the sum: keep_grads).
//      done[s][j] also says that rank s has finished reading this rank's gradients of that slice, so clearing is safe.
// Only flags are ever written remotely, and only into uncached memory; bulk data is pulled with system-scope loads
// (remote lines are never served from a stale local L2 line).  Every wait is bounded (wall clock, default 30 min like a
// collective library's watchdog): a missing peer ends the kernel with a status word in pinned host memory instead of
// hanging the GPU, and the abort is STICKY -- every later launch of this exchange returns at once without touching
// parameters, moments or gradients, so a replica never trains on past a failed exchange; the trainers read the status
// word (a pinned host word, no synchronisation) at the start of every step and raise.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/dta_hip.h"
#include "common.h"

namespace dta {

constexpr int XCHG_MAX_WORLD = 8;
constexpr int XCHG_MAX_WGS = 256;
constexpr int XCHG_SIG_BYTES = 16384;    // signal page (XchgSig), then the staging area
constexpr int XCHG_THREADS = 256;

struct XchgSig {
  unsigned ready[16];                               // ready[s]: rank s's gradients of this epoch are complete
  unsigned aborted;                                 // sticky: a wait of an earlier launch of THIS rank timed out -- later launches return at once
  unsigned pad[47];
  unsigned done[XCHG_MAX_WORLD][XCHG_MAX_WGS];      // done[s][j]: workgroup j of rank s has published its sums
};
static_assert(sizeof(XchgSig) <= XCHG_SIG_BYTES, "signal page overflow");

struct XchgArgs {
  const float* grads[XCHG_MAX_WORLD];    // every rank's gradient buffer (own entry: the local one)
  char* sig[XCHG_MAX_WORLD];             // every rank's signal page + staging area
  int rank, world;
  unsigned epoch;
  size_t n, shard;                       // floats; shard is a multiple of 4
  long long timeout_ticks;               // wall_clock64 ticks (100 MHz)
  int* status;                           // pinned host word: 0 ok, else (phase << 8) | peer
  int mode;                              // 0: all-reduce only (g := sum), 1: fused Adam
  int zero_grad;
  float* p; float* g; float* m; float* v;
  double* alpha_p; double* alpha_m; double* alpha_v; double* alpha_g;
  long long alpha_slot;                  // index into g of alpha's fp32 exchange slot, or -1
  float lr, beta1, beta2, eps, bc1, bc2, grad_scale;
};

__device__ __forceinline__ unsigned ld_sys32(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys32(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// 16-byte system-scope accesses (sc0 sc1: never served from / parked in a non-coherent cache line) as raw buffer
// operations: the compiler tracks their completion like any other load (hand-written global_load asm is not tracked, and
// a register copy the compiler places before a hand-placed s_waitcnt reads the destination too early).
typedef __amdgpu_buffer_rsrc_t xrsrc_t;
__device__ __forceinline__ xrsrc_t xchg_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
constexpr int XCHG_SYS = 1 | 16;      // cache policy bits: sc0 | sc1 = system scope
__device__ __forceinline__ f32x4 ld_sys128(xrsrc_t r, size_t float_index) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(float_index * 4), 0, XCHG_SYS));
}
__device__ __forceinline__ void st_sys128(xrsrc_t r, size_t float_index, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (unsigned)(float_index * 4), 0, XCHG_SYS);
}

// spin on a flag of the local page until it reaches `epoch` (wrap-safe) or the budget runs out
__device__ __forceinline__ bool wait_flag(const unsigned* flag, unsigned epoch, long long budget) {
  const long long t0 = wall_clock64();
  while ((int)(ld_sys32(flag) - epoch) < 0) {
    if (wall_clock64() - t0 > budget) return false;
    __builtin_amdgcn_s_sleep(2);
  }
  return true;
}

__global__ void __launch_bounds__(XCHG_THREADS) k_xchg_step(XchgArgs a) {
  __shared__ int s_abort;
  const int t = threadIdx.x, j = blockIdx.x;
  XchgSig* mine = (XchgSig*)a.sig[a.rank];
  if (t == 0) s_abort = (a.world > 1 && ld_sys32(&mine->aborted) != 0) ? 1 : 0;     // (one rank: nothing ever waits)
  __syncthreads();
  if (s_abort) return;                       // an earlier step of this rank timed out: nothing may be applied any more
  const long long tk0 = wall_clock64();
  // alpha's float64 gradient enters the exchange as one float32 slot of the gradient buffer: converted here, once, from
  // the (order-independently accumulated) double, so the slot does not depend on the order of any float atomics
  if (j == 0 && t == 0 && a.alpha_slot >= 0 && a.alpha_g) {
    st_sys32((unsigned*)(a.g + a.alpha_slot), __float_as_uint((float)a.alpha_g[0]));
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
  }
  __syncthreads();
  // 0. announce (one workgroup), 1. wait for every rank's gradients
  if (j == 0 && t < a.world) st_sys32(&((XchgSig*)a.sig[t])->ready[a.rank], a.epoch);
  if (t < a.world && !wait_flag(&mine->ready[t], a.epoch, a.timeout_ticks)) {
    s_abort = 1;
    a.status[0] = (1 << 8) | t;
    st_sys32(&mine->aborted, 1u);
  }
  __syncthreads();
  if (s_abort) return;
  if (t < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");     // drop what this CU / L2 still holds of the peers' buffers
  __syncthreads();
  const long long tk1 = wall_clock64();

  // Workgroup j owns the same relative slice {j*256 + t + k*G*256} (in 16-byte quads) of EVERY shard, in both phases:
  // its phase-B reads of rank s's sums depend only on workgroup j of rank s, so the "sums published" flags are per
  // (rank, workgroup) and no rank-wide counter or last-workgroup election sits on the critical path.
  const size_t nq = a.n / 4, shard_q = a.shard / 4, stride = (size_t)gridDim.x * XCHG_THREADS;
  auto len_q = [&](int s) -> size_t {
    const size_t lo = (size_t)s * shard_q;
    return lo >= nq ? 0 : (nq - lo < shard_q ? nq - lo : shard_q);
  };
  // A. my shard: sum over ranks in rank order -> staging (one 16-byte load per rank in flight per thread).
  //    A single rank has nothing to sum or publish: phase B reads its gradient buffer directly.
  if (a.world > 1) {
    const size_t lo = (size_t)a.rank * a.shard, mylen = len_q(a.rank);
    const xrsrc_t stage = xchg_rsrc(a.sig[a.rank] + XCHG_SIG_BYTES);
    for (size_t q = (size_t)j * XCHG_THREADS + t; q < mylen; q += stride) {
      f32x4 part[XCHG_MAX_WORLD];
#pragma unroll
      for (int s = 0; s < XCHG_MAX_WORLD; ++s)
        if (s < a.world) part[s] = ld_sys128(xchg_rsrc(a.grads[s]), lo + 4 * q);
      f32x4 acc = part[0];
#pragma unroll
      for (int s = 1; s < XCHG_MAX_WORLD; ++s)
        if (s < a.world) acc += part[s];
      st_sys128(stage, 4 * q, acc);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this wave's staging stores have reached memory
    __syncthreads();
    if (t < a.world) st_sys32(&((XchgSig*)a.sig[t])->done[a.rank][j], a.epoch);
  }
  // B. every rank's workgroup j has published its sums (which also says: it has finished reading this rank's gradients
  //    of that slice, so the slice may be cleared)
  if (a.world > 1 && t < a.world && !wait_flag(&mine->done[t][j], a.epoch, a.timeout_ticks)) {
    s_abort = 1;
    a.status[0] = (2 << 8) | t;
    st_sys32(&mine->aborted, 1u);
  }
  __syncthreads();
  if (s_abort) return;
  const float ss = a.lr / a.bc1, rbc2 = rsqrtf(a.bc2);
  for (size_t q = (size_t)j * XCHG_THREADS + t; q < shard_q; q += stride) {
    // one load per shard in flight; rank r starts with shard r+1, so the N ranks pull over N different links at a time
    f32x4 gs[XCHG_MAX_WORLD];
#pragma unroll
    for (int u = 0; u < XCHG_MAX_WORLD; ++u) {
      if (u < a.world) {
        int s = a.rank + 1 + u;
        if (s >= a.world) s -= a.world;
        if (q < len_q(s))
          gs[u] = ld_sys128(xchg_rsrc(a.world > 1 ? (const void*)(a.sig[s] + XCHG_SIG_BYTES) : (const void*)a.grads[0]), 4 * q);
      }
    }
#pragma unroll
    for (int u = 0; u < XCHG_MAX_WORLD; ++u) {
      if (u >= a.world) continue;
      int s = a.rank + 1 + u;
      if (s >= a.world) s -= a.world;
      if (q >= len_q(s)) continue;
      const size_t i = (size_t)s * a.shard + 4 * q;
      if (a.mode == 1) {
        const f32x4 mo = __builtin_nontemporal_load((const f32x4*)(a.m + i));
        const f32x4 vo = __builtin_nontemporal_load((const f32x4*)(a.v + i));
        f32x4 po = *(const f32x4*)(a.p + i), mn, vn;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float g = gs[u][c] * a.grad_scale;
          mn[c] = a.beta1 * mo[c] + (1.f - a.beta1) * g;
          vn[c] = a.beta2 * vo[c] + (1.f - a.beta2) * g * g;
          po[c] -= ss * (mn[c] / (sqrtf(vn[c]) * rbc2 + a.eps));
        }
        __builtin_nontemporal_store(mn, (f32x4*)(a.m + i));
        __builtin_nontemporal_store(vn, (f32x4*)(a.v + i));
        *(f32x4*)(a.p + i) = po;
        if (a.alpha_slot >= 0 && (size_t)(a.alpha_slot & ~3ll) == i) {      // the owner of alpha's slot steps alpha (float64)
          const float graw = gs[u][a.alpha_slot & 3];
          const double g = (double)graw * (double)a.grad_scale;
          const double m2 = (double)a.beta1 * a.alpha_m[0] + (1.0 - (double)a.beta1) * g;
          const double v2 = (double)a.beta2 * a.alpha_v[0] + (1.0 - (double)a.beta2) * g * g;
          a.alpha_m[0] = m2; a.alpha_v[0] = v2;
          a.alpha_p[0] -= ((double)a.lr / (double)a.bc1) * (m2 / (sqrt(v2) / sqrt((double)a.bc2) + (double)a.eps));
          if (a.alpha_g) a.alpha_g[0] = a.zero_grad ? 0.0 : (double)graw;
        }
      }
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *(f32x4*)(a.g + i) = (a.mode == 1 && a.zero_grad) ? z : gs[u];
    }
  }
  // workgroup 0's view of the launch, in 10 ns ticks: [2] waiting for the ranks to arrive, [3] the exchange proper
  if (j == 0 && t == 0) { a.status[2] = (int)(tk1 - tk0); a.status[3] = (int)(wall_clock64() - tk1); }
}

// probe pattern (peer_probe.py computes the same numbers with numpy): exact in float32
__global__ void k_xchg_selftest_fill(float* g, size_t n, int rank, int step) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const long long v = ((long long)i * 2654435761ll + (long long)rank * 40503ll + (long long)step * 9973ll) % 2001ll - 1000ll;
    g[i] = (float)v / 1024.0f;
  }
}

}  // namespace dta

using namespace dta;

struct dta_xchg {
  int rank, world, device;
  size_t n, n_pad, shard;
  float* grads;
  char* sig;
  size_t sig_bytes;
  float* peer_grads[XCHG_MAX_WORLD];
  char* peer_sig[XCHG_MAX_WORLD];
  bool connected;
  unsigned epoch;
  int* status_host;
  int* status_dev;
  double timeout_s;
  int max_wgs;
};

extern "C" {

int dta_xchg_create(int rank, int world, size_t n_floats, dta_xchg** out) {
  if (!out || world < 1 || world > XCHG_MAX_WORLD || rank < 0 || rank >= world || n_floats == 0) {
    dta_set_error("dta_xchg_create: bad argument (1 <= world <= %d ranks of one node)", XCHG_MAX_WORLD);
    return 1;
  }
  dta_xchg* x = (dta_xchg*)calloc(1, sizeof(dta_xchg));
  if (!x) { dta_set_error("dta_xchg_create: out of host memory"); return 1; }
  x->rank = rank; x->world = world; x->n = n_floats;
  x->n_pad = (n_floats + 3) & ~(size_t)3;
  x->shard = (((x->n_pad + world - 1) / world) + 3) & ~(size_t)3;
  x->sig_bytes = XCHG_SIG_BYTES + x->shard * sizeof(float);
  x->timeout_s = 1800.0;    // like a collective library's watchdog: rank skew of seconds (validation or a checkpoint on one rank) is routine
  x->max_wgs = 256;
  const char* why = nullptr;
  hipError_t e = hipGetDevice(&x->device);
  if (e == hipSuccess) { e = hipMalloc((void**)&x->grads, x->n_pad * sizeof(float)); why = "hipMalloc(gradient buffer)"; }
  if (e == hipSuccess) {
    e = hipExtMallocWithFlags((void**)&x->sig, x->sig_bytes, hipDeviceMallocUncached);
    why = "hipExtMallocWithFlags(uncached signal page)";
  }
  if (e == hipSuccess) { e = hipHostMalloc((void**)&x->status_host, 64, hipHostMallocMapped); why = "hipHostMalloc(status)"; }
  if (e == hipSuccess) { e = hipHostGetDevicePointer((void**)&x->status_dev, x->status_host, 0); why = "hipHostGetDevicePointer"; }
  if (e == hipSuccess) { x->status_host[0] = 0; e = hipMemset(x->grads, 0, x->n_pad * sizeof(float)); why = "hipMemset"; }
  if (e == hipSuccess) e = hipMemset(x->sig, 0, x->sig_bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    dta_set_error("dta_xchg_create: %s: %s", why ? why : "hipGetDevice", hipGetErrorString(e));
    if (x->grads) hipFree(x->grads);
    if (x->sig) hipFree(x->sig);
    if (x->status_host) hipHostFree(x->status_host);
    free(x);
    return 1;
  }
  x->peer_grads[rank] = x->grads;
  x->peer_sig[rank] = x->sig;
  x->connected = (world == 1);
  *out = x;
  return 0;
}

float* dta_xchg_grad_buffer(dta_xchg* x) { return x ? x->grads : nullptr; }
size_t dta_xchg_grad_capacity(dta_xchg* x) { return x ? x->n_pad : 0; }

int dta_xchg_export(dta_xchg* x, void* handles) {
  if (!x || !handles) { dta_set_error("dta_xchg_export: bad argument"); return 1; }
  hipIpcMemHandle_t h[2];
  hipError_t e = hipIpcGetMemHandle(&h[0], x->grads);
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h[1], x->sig);
  if (e != hipSuccess) { dta_set_error("dta_xchg_export: hipIpcGetMemHandle: %s", hipGetErrorString(e)); return 1; }
  memcpy(handles, h, sizeof(h));
  return 0;
}

int dta_xchg_connect(dta_xchg* x, const void* all_handles) {
  if (!x || !all_handles) { dta_set_error("dta_xchg_connect: bad argument"); return 1; }
  if (x->connected) return 0;
  const hipIpcMemHandle_t* h = (const hipIpcMemHandle_t*)all_handles;
  for (int s = 0; s < x->world; ++s) {
    if (s == x->rank) continue;
    hipError_t e = hipIpcOpenMemHandle((void**)&x->peer_grads[s], h[2 * s], hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess) e = hipIpcOpenMemHandle((void**)&x->peer_sig[s], h[2 * s + 1], hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      dta_set_error("dta_xchg_connect: hipIpcOpenMemHandle(rank %d): %s", s, hipGetErrorString(e));
      return 1;
    }
  }
  x->connected = true;
  return 0;
}

void dta_xchg_set_timeout(dta_xchg* x, double seconds) { if (x && seconds > 0) x->timeout_s = seconds; }
void dta_xchg_set_max_workgroups(dta_xchg* x, int wgs) { if (x && wgs > 0 && x->epoch == 0) x->max_wgs = wgs < XCHG_MAX_WGS ? wgs : XCHG_MAX_WGS; }

static int xchg_launch(dta_xchg* x, XchgArgs& a, void* stream) {
  if (!x->connected) { dta_set_error("dta_xchg: not connected (dta_xchg_connect)"); return 1; }
  for (int s = 0; s < x->world; ++s) { a.grads[s] = x->peer_grads[s]; a.sig[s] = x->peer_sig[s]; }
  a.rank = x->rank; a.world = x->world; a.n = x->n_pad; a.shard = x->shard;
  a.epoch = ++x->epoch;
  a.timeout_ticks = (long long)(x->timeout_s * 1e8);
  a.status = x->status_dev;
  a.g = x->grads;
  // one workgroup per CU at most, never more than there are element pairs per shard
  size_t wgs = (x->shard / 4 + XCHG_THREADS - 1) / XCHG_THREADS;
  if (wgs > (size_t)x->max_wgs) wgs = x->max_wgs;
  if (wgs > XCHG_MAX_WGS) wgs = XCHG_MAX_WGS;
  if (wgs < 1) wgs = 1;
  hipLaunchKernelGGL(k_xchg_step, dim3((unsigned)wgs), dim3(XCHG_THREADS), 0, (hipStream_t)stream, a);
  DTA_CHECK_LAUNCH("k_xchg_step");
  return 0;
}

int dta_xchg_selftest_fill(dta_xchg* x, int step, void* stream) {
  if (!x) { dta_set_error("dta_xchg_selftest_fill: null exchange"); return 1; }
  hipLaunchKernelGGL(k_xchg_selftest_fill, dim3(64), dim3(256), 0, (hipStream_t)stream, x->grads, x->n_pad, x->rank, step);
  DTA_CHECK_LAUNCH("k_xchg_selftest_fill");
  return 0;
}

int dta_xchg_allreduce(dta_xchg* x, const double* alpha_g, long long alpha_slot, void* stream) {
  if (!x) { dta_set_error("dta_xchg_allreduce: null exchange"); return 1; }
  if (alpha_g && (alpha_slot < 0 || (size_t)alpha_slot >= x->n_pad)) { dta_set_error("dta_xchg_allreduce: alpha's slot lies outside the buffer"); return 1; }
  XchgArgs a;
  memset(&a, 0, sizeof(a));
  a.mode = 0;
  a.alpha_g = (double*)alpha_g;          // read only in this mode
  a.alpha_slot = alpha_g ? alpha_slot : -1;
  return xchg_launch(x, a, stream);
}

int dta_xchg_adam_step(dta_xchg* x, float* p, float* m, float* v, size_t n, double* alpha_p, double* alpha_g,
                       long long alpha_slot, double* alpha_m, double* alpha_v, int step, float lr, float beta1, float beta2,
                       float eps, float grad_scale, int zero_grad, void* stream) {
  if (!x || !p || !m || !v || step < 1) { dta_set_error("dta_xchg_adam_step: bad argument"); return 1; }
  if (n != x->n && n != x->n_pad) { dta_set_error("dta_xchg_adam_step: n = %zu does not match the exchange's %zu elements", n, x->n); return 1; }
  if (n != x->n_pad) { dta_set_error("dta_xchg_adam_step: flat buffers must be padded to a multiple of 4 elements"); return 1; }
  if (alpha_p && (alpha_slot < 0 || (size_t)alpha_slot >= n || !alpha_m || !alpha_v)) {
    dta_set_error("dta_xchg_adam_step: alpha needs its exchange slot inside the gradient buffer and its moments");
    return 1;
  }
  XchgArgs a;
  memset(&a, 0, sizeof(a));
  a.mode = 1; a.zero_grad = zero_grad;
  a.p = p; a.m = m; a.v = v;
  a.alpha_p = alpha_p; a.alpha_m = alpha_m; a.alpha_v = alpha_v; a.alpha_g = alpha_g;
  a.alpha_slot = alpha_p ? alpha_slot : -1;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale = grad_scale;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  return xchg_launch(x, a, stream);
}

int dta_xchg_status(dta_xchg* x) {
  if (!x) return -1;
  const int s = ((volatile int*)x->status_host)[0];
  if (s) dta_set_error("dta_xchg: timed out in phase %d waiting for rank %d (a peer is missing, stuck or not co-scheduled)", s >> 8, s & 255);
  return s;
}

/* Development aid: the last launch as workgroup 0 saw it, in microseconds (host read of pinned words; synchronise first). */
int dta_xchg_last_timing(dta_xchg* x, float* wait_us, float* exchange_us) {
  if (!x) return 1;
  const volatile int* s = (const volatile int*)x->status_host;
  if (wait_us) *wait_us = s[2] * 0.01f;
  if (exchange_us) *exchange_us = s[3] * 0.01f;
  return 0;
}

int dta_xchg_destroy(dta_xchg* x) {
  if (!x) return 0;
  for (int s = 0; s < x->world; ++s) {
    if (s == x->rank || !x->connected) continue;
    if (x->peer_grads[s]) hipIpcCloseMemHandle(x->peer_grads[s]);
    if (x->peer_sig[s]) hipIpcCloseMemHandle(x->peer_sig[s]);
  }
  hipFree(x->grads);
  hipFree(x->sig);
  hipHostFree(x->status_host);
  free(x);
  return 0;
}

}  // extern "C"
