// Device side of the peer gradient exchange (protocol: xchg.hip), shared by the exchange launch (k_xchg_step) and by the
// first conv's weight-gradient launch, whose spare workgroups run the HEAD bucket's reduce-scatter while the matrix
// cores compute the last gradient of the step (reference train.py:89-98: DDP overlaps the gradient all-reduce with
// the backward; here the overlap is workgroups of one launch, not a side stream).
//
// The flat gradient buffer is exchanged as up to two SEGMENTS: head = [0, split) -- everything except the first conv's
// weights (76 % of the bytes), complete before that weight gradient starts -- and tail = [split, n).  Each segment is cut
// into `world` shards; rank r owns shard r of both.  split = 0: one segment (the whole buffer).
#pragma once
#include "common.h"

namespace dta {

constexpr int XCHG_MAX_WORLD = 8;
constexpr int XCHG_MAX_WGS = 256;
constexpr int XCHG_SIDE_WGS = 16;        // workgroups of the overlapped head reduce-scatter (the CUs the weight gradient leaves idle)
constexpr int XCHG_SIG_BYTES = 16384;    // signal page (XchgSig), then the staging area
constexpr int XCHG_THREADS = 256;

struct XchgSig {
  unsigned ready[16];                               // ready[s]: ALL of rank s's gradients of this epoch are complete
  unsigned aborted;                                 // sticky: a wait of an earlier launch of THIS rank timed out -- later launches return at once
  unsigned pad[15];
  unsigned ready_head[16];                          // ready_head[s]: rank s's head segment of this epoch is complete
  unsigned pad2[16];
  unsigned done_head[XCHG_MAX_WORLD][XCHG_SIDE_WGS];   // side workgroup k of rank s has published its part of the head sums
  unsigned done[XCHG_MAX_WORLD][XCHG_MAX_WGS];      // done[s][j]: workgroup j of rank s has published its sums
};
static_assert(sizeof(XchgSig) <= XCHG_SIG_BYTES, "signal page overflow");

struct XchgArgs {
  const float* grads[XCHG_MAX_WORLD];    // every rank's gradient buffer (own entry: the local one)
  char* sig[XCHG_MAX_WORLD];             // every rank's signal page + staging area
  int rank, world;
  unsigned epoch;
  size_t n;                              // floats (multiple of 4)
  size_t split, shard_h, shard_t;        // segments: head [0, split) in shards of shard_h, tail [split, n) in shards of shard_t
  int head_presummed;                    // the head's reduce-scatter of this epoch already ran (xchg_side_job)
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

// quads (16 bytes) of shard s of a segment of `seg_q` quads cut into shards of `shard_q`
__device__ __forceinline__ size_t xchg_len_q(size_t seg_q, size_t shard_q, int s) {
  const size_t lo = (size_t)s * shard_q;
  return lo >= seg_q ? 0 : (seg_q - lo < shard_q ? seg_q - lo : shard_q);
}

// Reduce-scatter by pulling, one segment: this rank sums ITS shard of every rank's gradient buffer in rank order 0..N-1
// (one fixed order for everybody: the sums, and so the replicas, are bit-identical) into its staging area at float offset
// stage_off.  Workgroup j of nj, all threads; one 16-byte load per rank in flight per thread.
__device__ __forceinline__ void xchg_reduce_scatter(const XchgArgs& a, size_t seg_lo, size_t seg_len, size_t shard, size_t stage_off,
                                                    int j, int nj) {
  const size_t lo = seg_lo + (size_t)a.rank * shard, mylen = xchg_len_q(seg_len / 4, shard / 4, a.rank);
  const size_t stride = (size_t)nj * blockDim.x;
  const xrsrc_t stage = xchg_rsrc(a.sig[a.rank] + XCHG_SIG_BYTES);
  for (size_t q = (size_t)j * blockDim.x + threadIdx.x; q < mylen; q += stride) {
    f32x4 part[XCHG_MAX_WORLD];
#pragma unroll
    for (int s = 0; s < XCHG_MAX_WORLD; ++s)
      if (s < a.world) part[s] = ld_sys128(xchg_rsrc(a.grads[s]), lo + 4 * q);
    f32x4 acc = part[0];
#pragma unroll
    for (int s = 1; s < XCHG_MAX_WORLD; ++s)
      if (s < a.world) acc += part[s];
    st_sys128(stage, stage_off + 4 * q, acc);
  }
}

// The head segment's reduce-scatter as a SIDE JOB of another launch (the first conv's weight gradient): workgroup k of
// XCHG_SIDE_WGS.  The head bucket was completed by the kernels before this launch on the stream; the launch's main
// workgroups do not depend on these, so a side workgroup that starts late (or waits for a late rank) delays nobody.
// s_abort: one int of LDS.
__device__ __forceinline__ void xchg_side_job(const XchgArgs& a, int k, int* s_abort) {
  const int t = threadIdx.x;
  XchgSig* mine = (XchgSig*)a.sig[a.rank];
  if (t == 0) *s_abort = ld_sys32(&mine->aborted) != 0 ? 1 : 0;
  __syncthreads();
  if (*s_abort) return;
  // alpha's float64 gradient enters the exchange as one float32 slot of the head segment: converted here, once, from the
  // (order-independently accumulated) double, before the head is announced
  if (k == 0 && t == 0 && a.alpha_slot >= 0 && a.alpha_g) {
    st_sys32((unsigned*)(a.g + a.alpha_slot), __float_as_uint((float)a.alpha_g[0]));
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
  }
  __syncthreads();
  if (k == 0 && t < a.world) st_sys32(&((XchgSig*)a.sig[t])->ready_head[a.rank], a.epoch);
  if (t < a.world && !wait_flag(&mine->ready_head[t], a.epoch, a.timeout_ticks)) {
    *s_abort = 1;
    a.status[0] = (3 << 8) | t;
    st_sys32(&mine->aborted, 1u);
  }
  __syncthreads();
  if (*s_abort) return;
  if (t < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");     // drop what this CU / L2 still holds of the peers' buffers
  __syncthreads();
  xchg_reduce_scatter(a, 0, a.split, a.shard_h, 0, k, XCHG_SIDE_WGS);
  __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): this wave's staging stores have reached memory
  __syncthreads();
  if (t < a.world) st_sys32(&((XchgSig*)a.sig[t])->done_head[a.rank][k], a.epoch);
}

}  // namespace dta
