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
