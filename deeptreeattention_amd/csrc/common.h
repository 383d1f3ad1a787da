// Shared device/host helpers for the gfx950 (MI355X) kernels of the Hang2020 hot path.
// Written for CDNA4 only: 64-lane wavefronts, MFMA 32x32 tiles, 160 KiB LDS per CU.
#pragma once
#include <stdio.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dta {

typedef unsigned short bf16_t;  // storage type of a bfloat16
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ bf16_t f2bf(float f) {  // round-to-nearest-even
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

// Storage formats of the activations / gradients that travel between kernels through HBM.  fp32 everywhere in fp32
// mode; in bf16 mode the conv outputs (pre-BatchNorm) are kept in IEEE half (their next use normalises them, so the
// three extra mantissa bits over bf16 matter and their range is small) and the gradient maps in bf16 (range first).
enum { FMT_F32 = 0, FMT_F16 = 1, FMT_BF16 = 2 };
__host__ __device__ __forceinline__ int fmt_bytes(int fmt) { return fmt == FMT_F32 ? 4 : 2; }
typedef _Float16 hw_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 hw_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pack2_fmt(float lo, float hi, int fmt) {
  if (fmt == FMT_F16) {   // saturating: a value beyond the half range must not become inf (and NaN after BatchNorm)
    // (fminf / fmaxf return the non-NaN operand, v_med3 the minimum: clamp only ordered values, so a NaN stays a NaN
    // and nodata pixels propagate through eval-mode BatchNorm as they do in the reference)
    const float l = lo != lo ? lo : __builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f);
    const float h2 = hi != hi ? hi : __builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f);
    hw_f16x2 h = {(_Float16)l, (_Float16)h2};
    return __builtin_bit_cast(unsigned, h);
  }
  unsigned a = __float_as_uint(lo), b = __float_as_uint(hi);     // bf16, round to nearest even
  a += 0x7FFFu + ((a >> 16) & 1u); b += 0x7FFFu + ((b >> 16) & 1u);
  return (a >> 16) | (b & 0xFFFF0000u);
}
__device__ __forceinline__ float unpack_lo(unsigned u, int fmt) {
  if (fmt == FMT_F16) { hw_f16x2 h = __builtin_bit_cast(hw_f16x2, u); return (float)h[0]; }
  return __uint_as_float(u << 16);
}
__device__ __forceinline__ float unpack_hi(unsigned u, int fmt) {
  if (fmt == FMT_F16) { hw_f16x2 h = __builtin_bit_cast(hw_f16x2, u); return (float)h[1]; }
  return __uint_as_float(u & 0xFFFF0000u);
}
// element i of a tensor stored in `fmt` (pass a compile-time constant inside loops: a run-time format puts every load
// behind a branch and the loads of an unrolled loop no longer overlap)
template <bool NT = false>
__device__ __forceinline__ float ld_fmt(const void* base, size_t i, int fmt) {
  if (fmt == FMT_F32) { const float* p = (const float*)base + i; return NT ? __builtin_nontemporal_load(p) : *p; }
  const unsigned short* p = (const unsigned short*)base + i;
  const unsigned short u = NT ? __builtin_nontemporal_load(p) : *p;
  if (fmt == FMT_F16) return (float)__builtin_bit_cast(_Float16, u);
  return __uint_as_float(((unsigned)u) << 16);
}
// four consecutive elements starting at i (i % 4 == 0, base 16-byte aligned)
template <bool NT = false>
__device__ __forceinline__ void ld4_fmt(float (&v)[4], const void* base, size_t i, int fmt) {
  if (fmt == FMT_F32) {
    const f32x4* p = (const f32x4*)((const float*)base + i);
    const f32x4 q = NT ? __builtin_nontemporal_load(p) : *p;
    v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
  } else {
    const u32x2* p = (const u32x2*)((const unsigned short*)base + i);
    const u32x2 q = NT ? __builtin_nontemporal_load(p) : *p;
    v[0] = unpack_lo(q.x, fmt); v[1] = unpack_hi(q.x, fmt); v[2] = unpack_lo(q.y, fmt); v[3] = unpack_hi(q.y, fmt);
  }
}
__device__ __forceinline__ void st_fmt(void* base, size_t i, float v, int fmt) {
  if (fmt == FMT_F32) ((float*)base)[i] = v;
  else if (fmt == FMT_F16) ((_Float16*)base)[i] = (_Float16)(v != v ? v : __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f));
  else ((unsigned short*)base)[i] = (unsigned short)(pack2_fmt(v, 0.f, FMT_BF16) & 0xFFFFu);
}
// exchange with the neighbouring lane (lane ^ 1) without touching LDS
__device__ __forceinline__ float lane_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false));
}

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  __device__ static __forceinline__ float to(float v) { return v; }
  __device__ static __forceinline__ float from(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
  __device__ static __forceinline__ bf16_t to(float v) { return f2bf(v); }
  __device__ static __forceinline__ float from(bf16_t v) { return bf2f(v); }
};

// ---------------------------------------------------------------------------------------------
// "Tile layout" (TL): the HBM/LDS image every 3x3 convolution operand uses.
//   TL[g][b][chunk][q][16]  with  q = (h+1)*(W+2) + (w+1)  over the zero-haloed (H+2)x(W+2) grid,
//   chunk = c/16, and the 16 channels of a row stored at a swizzled position so that the MFMA
//   fragment reads out of LDS are bank-conflict free (the LDS image is a linear copy of HBM):
//     fp32 : pos = (c%16) ^ (q & 15)   (ds_read_b32, one element per lane; LDS image = linear copy of HBM)
//     bf16 : pos = c%16                (no swizzle: the bf16 kernels pad LDS rows to 48 B while staging instead)
// The same rule applies to packed conv weights with row index (tap*N + n).
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __host__ __forceinline__ int tl_pos(int row, int c16);
template <> __device__ __host__ __forceinline__ int tl_pos<float>(int row, int c16) { return c16 ^ (row & 15); }
template <> __device__ __host__ __forceinline__ int tl_pos<bf16_t>(int row, int c16) { return c16; }

// Vector width (elements per 16-byte store) of a TL row segment.
template <typename T> struct TlVec { static constexpr int VW = 16 / (int)sizeof(T); };

// Store segment `part` (VW consecutive stored positions) of haloed-grid row q: vals[j] must be the value of
// channel tl_pos<T>(q, part*VW + j).  One 16-byte store.
__device__ __forceinline__ void tl_store_vec(float* rowbase, int part, const float* vals) {
  *reinterpret_cast<float4*>(rowbase + part * 4) = make_float4(vals[0], vals[1], vals[2], vals[3]);
}
__device__ __forceinline__ void tl_store_vec(bf16_t* rowbase, int part, const float* vals) {
  uint4 u;
  u.x = (unsigned)f2bf(vals[0]) | ((unsigned)f2bf(vals[1]) << 16);
  u.y = (unsigned)f2bf(vals[2]) | ((unsigned)f2bf(vals[3]) << 16);
  u.z = (unsigned)f2bf(vals[4]) | ((unsigned)f2bf(vals[5]) << 16);
  u.w = (unsigned)f2bf(vals[6]) | ((unsigned)f2bf(vals[7]) << 16);
  *reinterpret_cast<uint4*>(rowbase + part * 8) = u;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// ReLU and the pooling maximum as torch computes them: a NaN operand stays a NaN (v_max_f32 would return the other
// operand), so nodata pixels reach the scores exactly as they do in the reference
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float max_nan(float a, float b) { return (a >= b || a != a) ? a : b; }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// Block-wide sum for 256-thread blocks (4 waves); scratch must hold 4 floats.  All threads get the sum.
__device__ __forceinline__ float block_sum256(float v, float* scratch) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

}  // namespace dta

// hipFuncSetAttribute is a per-device setting: a launch site remembers per device ordinal whether it has made it (one
// process may drive several devices; a process-wide flag would leave the second device with the 64 KiB default).
#include <atomic>
#include <stdlib.h>
// Developer switches (same-box A/B runs of alternative launch plans) exist only in the developer library
// (libdta_hip_dev.so, built with -DDTA_DEV_SWITCHES): the product library reads NOTHING from the environment, so no
// variable of a training job's environment can change a kernel plan or a rounding.
#ifdef DTA_DEV_SWITCHES
inline const char* dev_getenv(const char* name) { return getenv(name); }
#else
inline const char* dev_getenv(const char*) { return nullptr; }
#endif

struct DevOnce {
  std::atomic<unsigned long long> done{0};
  bool first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return true;
    const unsigned long long bit = 1ull << (d & 63);
    return (done.fetch_or(bit) & bit) == 0;
  }
};

// Error plumbing: kernels launch asynchronously on the caller's stream; launch errors are caught here
// and surfaced through dta_last_error() (no exceptions cross the C ABI).
void dta_set_error(const char* fmt, ...);
#ifdef DTA_TRACE_LAUNCHES   /* developer build: name every launch on stderr and wait for it */
#define DTA_TRACE_LAUNCH_(name) do { fprintf(stderr, "[dta] %s\n", name); fflush(stderr); hipDeviceSynchronize(); } while (0)
#else
#define DTA_TRACE_LAUNCH_(name) do { } while (0)
#endif
#define DTA_CHECK_LAUNCH(name)                                                      \
  do {                                                                              \
    DTA_TRACE_LAUNCH_(name);                                                        \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      dta_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));          \
      return 1;                                                                     \
    }                                                                               \
  } while (0)
