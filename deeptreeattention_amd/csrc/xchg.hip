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
// Overlap with the backward (dta_net_backward_xchg): the buffer is cut into a HEAD segment (everything but the first conv's
// weights, 76 % of the bytes, complete before that layer's weight gradient starts) and a TAIL segment, each with its own
// shards.  The head's phase A runs as 16 side workgroups of the weight-gradient launch itself (xchg_dev.h: own ready /
// done flags), i.e. on the CUs that launch leaves idle and for the ~70 us it takes; this launch then finds the head
// summed, sums only the tail and runs phase B over both.  Without that (one segment, fp32 mode) it does everything.
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
#include "xchg_dev.h"

namespace dta {

// Phase B for one segment: pull every rank's summed shard (one load per rank in flight, starting at rank r+1 so that the N
// ranks use the N-1 links of the mesh at the same time) and apply Adam to the local parameters / moments; the gradient
// elements are cleared (or replaced by the sum: keep_grads).  A single rank reads its own gradient buffer.
__device__ __forceinline__ void xchg_gather_adam(const XchgArgs& a, size_t seg_lo, size_t seg_len, size_t shard, size_t stage_off,
                                                 int j, int nj) {
  const size_t seg_q = seg_len / 4, shard_q = shard / 4, stride = (size_t)nj * XCHG_THREADS;
  const float ss = a.lr / a.bc1, rbc2 = rsqrtf(a.bc2);
  for (size_t q = (size_t)j * XCHG_THREADS + threadIdx.x; q < shard_q; q += stride) {
    f32x4 gs[XCHG_MAX_WORLD];
#pragma unroll
    for (int u = 0; u < XCHG_MAX_WORLD; ++u) {
      if (u < a.world) {
        int s = a.rank + 1 + u;
        if (s >= a.world) s -= a.world;
        if (q < xchg_len_q(seg_q, shard_q, s)) {
          if (a.world > 1) gs[u] = ld_sys128(xchg_rsrc(a.sig[s] + XCHG_SIG_BYTES), stage_off + 4 * q);
          else gs[u] = *(const f32x4*)(a.g + seg_lo + 4 * q);      // (a single rank: its own gradients, written by the launch before: plain loads)
        }
      }
    }
#pragma unroll
    for (int u = 0; u < XCHG_MAX_WORLD; ++u) {
      if (u >= a.world) continue;
      int s = a.rank + 1 + u;
      if (s >= a.world) s -= a.world;
      if (q >= xchg_len_q(seg_q, shard_q, s)) continue;
      const size_t i = seg_lo + (size_t)s * shard + 4 * q;
      if (a.world == 1 && a.alpha_slot >= 0 && a.alpha_g && (size_t)(a.alpha_slot & ~3ll) == i) {
        // a single rank skips the handshake that orders the slot conversion before the sums: the owner of the slot's
        // quad takes alpha's float64 gradient (rounded to float32, as it would travel) directly
        const float ag = (float)a.alpha_g[0];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c == (int)(a.alpha_slot & 3)) gs[u][c] = ag;
      }
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
}

__global__ void __launch_bounds__(XCHG_THREADS) k_xchg_step(XchgArgs a) {
  __shared__ int s_abort;
  const int t = threadIdx.x, j = blockIdx.x, nj = gridDim.x;
  XchgSig* mine = (XchgSig*)a.sig[a.rank];
  if (t == 0) s_abort = (a.world > 1 && ld_sys32(&mine->aborted) != 0) ? 1 : 0;     // (one rank: nothing ever waits)
  __syncthreads();
  if (s_abort) return;                       // an earlier step of this rank timed out: nothing may be applied any more
  const long long tk0 = wall_clock64();
  // alpha's float64 gradient enters the exchange as one float32 slot of the gradient buffer: converted here, once, from
  // the (order-independently accumulated) double, so the slot does not depend on the order of any float atomics
  // (a head segment that was summed ahead of this launch carried it already: xchg_side_job)
  if (j == 0 && t == 0 && a.alpha_slot >= 0 && a.alpha_g && !a.head_presummed && a.world > 1) {
    st_sys32((unsigned*)(a.g + a.alpha_slot), __float_as_uint((float)a.alpha_g[0]));
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
  }
  __syncthreads();
  // 0. announce (one workgroup), 1. wait for every rank's gradients  (a single rank has nobody to wait for and nothing
  //    stale to drop: its launch is the optimizer pass with the slot conversion in front)
  if (a.world > 1) {
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
  }
  const long long tk1 = wall_clock64();

  // Workgroup j owns the same relative slice {j*256 + t + k*G*256} (in 16-byte quads) of EVERY shard, in both phases:
  // its phase-B reads of rank s's sums depend only on workgroup j of rank s, so the "sums published" flags are per
  // (rank, workgroup) and no rank-wide counter or last-workgroup election sits on the critical path.  (A head segment
  // summed ahead by the side workgroups of the weight-gradient launch has its own flags: done_head[s][*].)
  // A. my shards: sum over ranks in rank order -> staging.  A single rank has nothing to sum or publish.
  if (a.world > 1) {
    if (a.split > 0 && !a.head_presummed) xchg_reduce_scatter(a, 0, a.split, a.shard_h, 0, j, nj);
    xchg_reduce_scatter(a, a.split, a.n - a.split, a.shard_t, a.shard_h, j, nj);
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this wave's staging stores have reached memory
    __syncthreads();
    if (t < a.world) st_sys32(&((XchgSig*)a.sig[t])->done[a.rank][j], a.epoch);
  }
  // B. every rank's workgroup j has published its sums (which also says: it has finished reading this rank's gradients
  //    of that slice, so the slice may be cleared); a head summed ahead: all of every rank's side workgroups have
  if (a.world > 1) {
    if (t < a.world && !wait_flag(&mine->done[t][j], a.epoch, a.timeout_ticks)) {
      s_abort = 1;
      a.status[0] = (2 << 8) | t;
      st_sys32(&mine->aborted, 1u);
    }
    if (a.head_presummed && t >= 64 && t < 64 + a.world * XCHG_SIDE_WGS) {
      const int s = (t - 64) / XCHG_SIDE_WGS, k = (t - 64) % XCHG_SIDE_WGS;
      if (!wait_flag(&mine->done_head[s][k], a.epoch, a.timeout_ticks)) {
        s_abort = 1;
        a.status[0] = (4 << 8) | s;
        st_sys32(&mine->aborted, 1u);
      }
    }
  }
  __syncthreads();
  if (s_abort) return;
  if (a.split > 0) xchg_gather_adam(a, 0, a.split, a.shard_h, 0, j, nj);
  xchg_gather_adam(a, a.split, a.n - a.split, a.shard_t, a.shard_h, j, nj);
  // workgroup 0's view of the launch, in 10 ns ticks: [2] waiting for the ranks to arrive, [3] the exchange proper
  if (j == 0 && t == 0) { a.status[2] = (int)(tk1 - tk0); a.status[3] = (int)(wall_clock64() - tk1); }
}

// The head segment's reduce-scatter as a launch of its own (dta_xchg_reduce_head): the same device code the first conv's
// weight-gradient launch runs in its spare workgroups, for plans that have no combined kernel (the caller puts it on a
// side stream beside its last gradient kernel) and for exercising the head / tail protocol without a network around it.
__global__ void __launch_bounds__(XCHG_THREADS) k_xchg_head(XchgArgs side) {
  __shared__ int s_abort;
  xchg_side_job(side, (int)blockIdx.x, &s_abort);
}

// probe pattern (peer_probe.py computes the same numbers with numpy): exact in float32
__global__ void k_xchg_selftest_fill(float* g, size_t n, int rank, int step) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const long long v = ((long long)i * 2654435761ll + (long long)rank * 40503ll + (long long)step * 9973ll) % 2001ll - 1000ll;
    g[i] = (float)v / 1024.0f;
  }
}

// the sum over ranks (in rank order, as the exchange sums) of the pattern above vs the buffer: mismatches are COUNTED on the
// device (status word 4, pinned host memory), so a burst of fill -> exchange -> verify triples needs no host synchronisation
// between its steps -- back-to-back launches are what a training loop issues
__global__ void k_xchg_selftest_verify(const float* g, size_t n, int world, int step, int* mismatches) {
  int bad = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float want = 0.f;
    for (int r = 0; r < world; ++r) {
      const long long v = ((long long)i * 2654435761ll + (long long)r * 40503ll + (long long)step * 9973ll) % 2001ll - 1000ll;
      want += (float)v / 1024.0f;
    }
    bad += g[i] != want;
  }
  if (bad) atomicAdd(mismatches, bad);
}

}  // namespace dta

using namespace dta;

struct dta_xchg {
  int rank, world, device;
  size_t n, n_pad, shard;
  size_t split, shard_h, shard_t;      // segments (dta_xchg_set_split): head [0, split), tail [split, n_pad)
  unsigned side_epoch;                 // epoch whose head segment the side workgroups of a weight-gradient launch summed
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
  x->sig_bytes = XCHG_SIG_BYTES + (x->shard + 16) * sizeof(float);      // (+16: two segments round their shards up separately)
  x->split = 0; x->shard_h = 0; x->shard_t = x->shard;
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
  if (e == hipSuccess) { x->status_host[0] = 0; x->status_host[4] = 0; e = hipMemset(x->grads, 0, x->n_pad * sizeof(float)); why = "hipMemset"; }
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
int dta_xchg_set_split(dta_xchg* x, size_t split_floats) {
  if (!x || x->epoch != 0 || (split_floats & 3) || split_floats >= x->n_pad) { dta_set_error("dta_xchg_set_split: before the first step, a multiple of 4 floats inside the buffer"); return 1; }
  x->split = split_floats;
  if (split_floats == 0) { x->shard_h = 0; x->shard_t = x->shard; return 0; }
  const size_t w = (size_t)x->world;
  x->shard_h = (((split_floats + w - 1) / w) + 3) & ~(size_t)3;
  x->shard_t = ((((x->n_pad - split_floats) + w - 1) / w) + 3) & ~(size_t)3;
  if (x->shard_h + x->shard_t > x->shard + 16) { dta_set_error("dta_xchg_set_split: staging area too small"); return 1; }
  return 0;
}
void dta_xchg_set_max_workgroups(dta_xchg* x, int wgs) { if (x && wgs > 0 && x->epoch == 0) x->max_wgs = wgs < XCHG_MAX_WGS ? wgs : XCHG_MAX_WGS; }

static void xchg_fill_common(dta_xchg* x, XchgArgs& a) {
  for (int s = 0; s < x->world; ++s) { a.grads[s] = x->peer_grads[s]; a.sig[s] = x->peer_sig[s]; }
  a.rank = x->rank; a.world = x->world; a.n = x->n_pad;
  a.split = x->split; a.shard_h = x->shard_h; a.shard_t = x->shard_t;
  a.timeout_ticks = (long long)(x->timeout_s * 1e8);
  a.status = x->status_dev;
  a.g = x->grads;
}

static int xchg_launch(dta_xchg* x, XchgArgs& a, void* stream) {
  if (!x->connected) { dta_set_error("dta_xchg: not connected (dta_xchg_connect)"); return 1; }
  xchg_fill_common(x, a);
  a.epoch = ++x->epoch;
  a.head_presummed = (x->world > 1 && x->split > 0 && x->side_epoch == a.epoch) ? 1 : 0;
  // one workgroup per CU at most, never more than there are element quads per shard
  const size_t big = x->shard_h > x->shard_t ? x->shard_h : x->shard_t;
  size_t wgs = (big / 4 + XCHG_THREADS - 1) / XCHG_THREADS;
  if (x->world > 1) {      // (co-resident workgroups with per-workgroup flags; a single rank is a plain optimizer pass)
    if (wgs > (size_t)x->max_wgs) wgs = x->max_wgs;
    if (wgs > XCHG_MAX_WGS) wgs = XCHG_MAX_WGS;
  } else if (wgs > 2048) wgs = 2048;
  if (wgs < 1) wgs = 1;
  hipLaunchKernelGGL(k_xchg_step, dim3((unsigned)wgs), dim3(XCHG_THREADS), 0, (hipStream_t)stream, a);
  DTA_CHECK_LAUNCH("k_xchg_step");
  return 0;
}

// internal (capi.hip): arguments of the head segment's overlapped reduce-scatter for THE NEXT exchange launch of `x` (its
// epoch), to be run by the side workgroups of the first conv's weight-gradient launch.  Returns 1 when the exchange has no
// head segment to overlap (one rank, or no split set).
int dta_xchg_side_args(dta_xchg* x, const double* alpha_g, long long alpha_slot, dta::XchgArgs* out) {
  if (!x || !x->connected || x->world < 2 || x->split == 0) return 1;
  memset(out, 0, sizeof(*out));
  xchg_fill_common(x, *out);
  out->epoch = x->epoch + 1;
  out->alpha_g = (double*)alpha_g;
  out->alpha_slot = alpha_g ? alpha_slot : -1;
  x->side_epoch = out->epoch;
  return 0;
}

void dta_xchg_side_cancel(dta_xchg* x) { if (x) x->side_epoch = 0; }

int dta_xchg_reduce_head(dta_xchg* x, const double* alpha_g, long long alpha_slot, void* stream) {
  if (!x) { dta_set_error("dta_xchg_reduce_head: null exchange"); return 1; }
  if (alpha_g && (alpha_slot < 0 || (size_t)alpha_slot >= x->split)) { dta_set_error("dta_xchg_reduce_head: alpha's slot must lie inside the head segment"); return 1; }
  if (x->world < 2 || x->split == 0) return 0;      // one rank / one segment: the exchange launch does everything itself
  XchgArgs side;
  if (dta_xchg_side_args(x, alpha_g, alpha_slot, &side)) { dta_set_error("dta_xchg_reduce_head: exchange not connected"); return 1; }
  hipLaunchKernelGGL(k_xchg_head, dim3(XCHG_SIDE_WGS), dim3(XCHG_THREADS), 0, (hipStream_t)stream, side);
  DTA_CHECK_LAUNCH("k_xchg_head");
  return 0;
}

int dta_xchg_selftest_fill(dta_xchg* x, int step, void* stream) {
  if (!x) { dta_set_error("dta_xchg_selftest_fill: null exchange"); return 1; }
  hipLaunchKernelGGL(k_xchg_selftest_fill, dim3(64), dim3(256), 0, (hipStream_t)stream, x->grads, x->n_pad, x->rank, step);
  DTA_CHECK_LAUNCH("k_xchg_selftest_fill");
  return 0;
}

int dta_xchg_selftest_verify(dta_xchg* x, int step, void* stream) {
  if (!x) { dta_set_error("dta_xchg_selftest_verify: null exchange"); return 1; }
  hipLaunchKernelGGL(k_xchg_selftest_verify, dim3(64), dim3(256), 0, (hipStream_t)stream, x->grads, x->n, x->world, step, x->status_dev + 4);
  DTA_CHECK_LAUNCH("k_xchg_selftest_verify");
  return 0;
}
int dta_xchg_selftest_mismatches(dta_xchg* x) { return x ? ((volatile int*)x->status_host)[4] : -1; }

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

int dta_xchg_disconnect(dta_xchg* x) {
  if (!x) return 0;
  for (int s = 0; s < x->world; ++s) {
    if (s == x->rank || !x->connected) continue;
    if (x->peer_grads[s]) { hipIpcCloseMemHandle(x->peer_grads[s]); x->peer_grads[s] = nullptr; }
    if (x->peer_sig[s]) { hipIpcCloseMemHandle(x->peer_sig[s]); x->peer_sig[s] = nullptr; }
  }
  x->connected = false;
  return 0;
}

int dta_xchg_destroy(dta_xchg* x) {
  if (!x) return 0;
  dta_xchg_disconnect(x);
  hipFree(x->grads);
  hipFree(x->sig);
  hipHostFree(x->status_host);
  free(x);
  return 0;
}

}  // extern "C"
