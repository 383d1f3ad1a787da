// Crop preprocessing on the device: the step that feeds the Hang2020 hot path (reference src/utils.py:36-79,
// src/augmentation.py:13-14, src/data.py:284-310).  One workgroup per crop:
//   raw crop (int16 / uint8 / float32; band-first as rasterio reads it, or pixel-interleaved as the crops lie on
//   disk) -> drop the first/last `clip` bands -> per-pixel min-max over the bands -> NEAREST resize to SxS ->
//   (training) horizontal + vertical flip -> float32 [bands][S][S] of the NCHW batch.
// Min-max is per pixel, so only the SxS pixels the resize keeps are ever read.  Rounding follows scikit-learn's
// float32 MinMaxScaler step by step (separately rounded multiply and add): results are bit-identical to the
// reference's CPU path (tests/test_preprocess.py).  HBM-bound: sampled pixels in once (the second read of a crop hits
// L2), 4 bytes per output element out, written as whole contiguous rows.
#include "../../include/dta_hip.h"
#include "common.h"

// Bit-exactness with the reference needs the multiply and the add of the scaling rounded SEPARATELY (NumPy does
// `X *= scale; X += min`): no fused multiply-add in this file (hipcc contracts by default, and HIP's __fmul_rn /
// __fadd_rn are plain operators that the contraction sees through; see mul_rounded below).
#pragma clang fp contract(off)

namespace {

using namespace dta;

// x * s rounded to float32 on its own: the empty asm makes the product opaque, so no later add can be fused into it
__device__ __forceinline__ float mul_rounded(float x, float s) {
  float p = x * s;
  asm volatile("" : "+v"(p));
  return p;
}

struct CropArgs {
  const void* raw; const long long* off; const int* hs; const int* ws; float* out;
  int B, Craw, c0, C, S, flip, pitch, cc, np;   // np: output pixels per workgroup (a crop is split into SS/np workgroups)
};

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(short v) { return (float)v; }
__device__ __forceinline__ float to_f(unsigned char v) { return (float)v; }

// ATen's nearest-neighbour source index: min(floor(dst * float(in / out)), in - 1)
__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
  const float scale = (float)in / (float)out;
  const int i = (int)floorf((float)dst * scale);
  return i < in - 1 ? i : in - 1;
}

// LAYOUT 0: raw[c][h][w]   LAYOUT 1: raw[h][w][c]
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void k_preprocess_crops(CropArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x, t = threadIdx.x, S = a.S, SS = S * S, C = a.C;
  const int h = a.hs[b], w = a.ws[b];
  float* out = a.out + (size_t)b * C * SS;
  const int p0 = blockIdx.y * a.np, NP = min(a.np, SS - p0);   // this workgroup's run of output pixels
  if (h <= 0 || w <= 0) {   // missing year: an all-zero tensor (reference data.py:295-296)
    for (int i = t; i < C * NP; i += 256) { const int c = i / NP; out[(size_t)c * SS + p0 + (i - c * NP)] = 0.f; }
    return;
  }
  const T* raw = reinterpret_cast<const T*>(a.raw) + a.off[b];
  int* pix = reinterpret_cast<int*>(sm);          // [np] source pixel (row-major index into the h x w crop)
  float* scl = sm + a.np;                         // [np] 1 / range
  float* mn = scl + a.np;                         // [np] -min * scale
  float* part = mn + a.np;                        // CHW: [2][K][np] partial min / max; HWC: the [cc][pitch] transpose tile
  for (int p = t; p < NP; p += 256) {
    int i = (p0 + p) / S, j = (p0 + p) - i * S;
    if (a.flip) { i = S - 1 - i; j = S - 1 - j; }   // both flips, applied after the resize
    pix[p] = nearest_src(i, h, S) * w + nearest_src(j, w, S);
  }
  __syncthreads();
  const float tiny = 10.f * 1.1920928955078125e-07f;   // scikit-learn: ranges below 10 * eps(float32) are "constant"
  if (LAYOUT == 0) {
    const size_t plane = (size_t)h * w;
    int K = 256 / a.np; if (K < 1) K = 1; if (K > 8) K = 8;
    for (int item = t; item < K * NP; item += 256) {     // (channel slice k, pixel p)
      const int k = item / NP, p = item - k * NP;
      const T* src = raw + pix[p];
      float lo = __builtin_inff(), hi = -__builtin_inff();
#pragma unroll 8
      for (int c = k; c < C; c += K) {
        const float v = to_f(src[(size_t)(a.c0 + c) * plane]);
        lo = fminf(lo, v); hi = fmaxf(hi, v);             // NaNs are passed over, as nanmin / nanmax do
      }
      part[k * a.np + p] = lo; part[(K + k) * a.np + p] = hi;
    }
    __syncthreads();
    for (int p = t; p < NP; p += 256) {
      float lo = part[p], hi = part[K * a.np + p];
      for (int k = 1; k < K; ++k) { lo = fminf(lo, part[k * a.np + p]); hi = fmaxf(hi, part[(K + k) * a.np + p]); }
      float rng = hi - lo;
      if (rng < tiny) rng = 1.f;
      const float s = 1.f / rng;
      scl[p] = s; mn[p] = 0.f - mul_rounded(lo, s);
    }
    __syncthreads();
    for (int i = t; i < C * NP; i += 256) {
      const int c = i / NP, p = i - c * NP;
      const float v = to_f(raw[(size_t)(a.c0 + c) * plane + pix[p]]);
      out[(size_t)c * SS + p0 + p] = mul_rounded(v, scl[p]) + mn[p];
    }
  } else {
    const int lane = t & 63, wv = t >> 6;
    // one wave per pixel (its bands are contiguous), two pixels in flight per wave so that the second pixel's loads
    // overlap the first one's shuffle reduction
    for (int pa = wv; pa < NP; pa += 8) {
      const int pb = pa + 4 < NP ? pa + 4 : pa;
      const T* sa = raw + (size_t)pix[pa] * a.Craw + a.c0;
      const T* sb = raw + (size_t)pix[pb] * a.Craw + a.c0;
      float lo0 = __builtin_inff(), hi0 = -__builtin_inff(), lo1 = lo0, hi1 = hi0;
      for (int c = lane; c < C; c += 64) {
        const float v0 = to_f(sa[c]), v1 = to_f(sb[c]);
        lo0 = fminf(lo0, v0); hi0 = fmaxf(hi0, v0); lo1 = fminf(lo1, v1); hi1 = fmaxf(hi1, v1);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        lo0 = fminf(lo0, __shfl_xor(lo0, o)); hi0 = fmaxf(hi0, __shfl_xor(hi0, o));
        lo1 = fminf(lo1, __shfl_xor(lo1, o)); hi1 = fmaxf(hi1, __shfl_xor(hi1, o));
      }
      if (lane < 2) {
        const int p = lane ? pb : pa;
        const float lo = lane ? lo1 : lo0, hi = lane ? hi1 : hi0;
        float rng = hi - lo;
        if (rng < tiny) rng = 1.f;
        const float s = 1.f / rng;
        scl[p] = s; mn[p] = 0.f - mul_rounded(lo, s);
      }
    }
    __syncthreads();
    const int cc = a.cc, pitch = a.pitch;                 // transpose [pixel][band] -> [band][pixel] through LDS
    for (int cb = 0; cb < C; cb += cc) {
      const int nc = min(cc, C - cb);
      if (lane < nc)      // cc == 64: lane = band of the chunk; four pixels' loads in flight per wave
        for (int pa = wv; pa < NP; pa += 16) {
          float v[4]; int pp[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            pp[u] = pa + 4 * u < NP ? pa + 4 * u : pa;
            v[u] = to_f(raw[(size_t)pix[pp[u]] * a.Craw + a.c0 + cb + lane]);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) part[lane * pitch + pp[u]] = mul_rounded(v[u], scl[pp[u]]) + mn[pp[u]];
        }
      __syncthreads();
      for (int i = t; i < nc * NP; i += 256) {
        const int c = i / NP, p = i - c * NP;
        out[(size_t)(cb + c) * SS + p0 + p] = part[c * pitch + p];
      }
      __syncthreads();
    }
  }
}

// Same preprocessing, output written straight as the bf16 conv tiles the first layer's kernels read
// ([crop][chunk = band / 16][pixel][16], halo-free; bands past C are zero): the float32 NCHW batch (4 bytes per element
// written here, read and converted again by the first conv) never exists.  Values are the float32 results above rounded
// to bf16 (nearest even).  A thread owns (pixel, 16-band chunk) items: with pixel-interleaved crops its 16 bands are one
// contiguous 32-byte run of the raw pixel.
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void k_preprocess_crops_tiles(CropArgs a, unsigned short* tiles) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x, t = threadIdx.x, S = a.S, SS = S * S, C = a.C, NC = (C + 15) / 16;
  const int h = a.hs[b], w = a.ws[b];
  unsigned short* out = tiles + (size_t)b * NC * SS * 16;
  const int p0 = blockIdx.y * a.np, NP = min(a.np, SS - p0);
  if (h <= 0 || w <= 0) {   // missing year: all zeros
    for (int i = t; i < NC * NP * 2; i += 256) {
      const int ch = i / (NP * 2), r = i % (NP * 2);
      reinterpret_cast<u32x4*>(out + ((size_t)ch * SS + p0) * 16)[r] = u32x4{0u, 0u, 0u, 0u};
    }
    return;
  }
  const T* raw = reinterpret_cast<const T*>(a.raw) + a.off[b];
  int* pix = reinterpret_cast<int*>(sm);
  float* scl = sm + a.np;
  float* mn = scl + a.np;
  for (int p = t; p < NP; p += 256) {
    int i = (p0 + p) / S, j = (p0 + p) - i * S;
    if (a.flip) { i = S - 1 - i; j = S - 1 - j; }
    pix[p] = nearest_src(i, h, S) * w + nearest_src(j, w, S);
  }
  __syncthreads();
  const float tiny = 10.f * 1.1920928955078125e-07f;
  const size_t plane = (size_t)h * w;
  auto at = [&](int p, int c) -> float {
    return LAYOUT == 0 ? to_f(raw[(size_t)(a.c0 + c) * plane + pix[p]]) : to_f(raw[(size_t)pix[p] * a.Craw + a.c0 + c]);
  };
  // per-pixel range: 256 / np lanes share a pixel when it pays, else one lane per pixel
  {
    const int lane = t & 63, wv = t >> 6;
    for (int p = wv; p < NP; p += 4) {
      float lo = __builtin_inff(), hi = -__builtin_inff();
      for (int c = lane; c < C; c += 64) { const float v = at(p, c); lo = fminf(lo, v); hi = fmaxf(hi, v); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
      if (lane == 0) {
        float rng = hi - lo;
        if (rng < tiny) rng = 1.f;
        const float s = 1.f / rng;
        scl[p] = s; mn[p] = 0.f - mul_rounded(lo, s);
      }
    }
  }
  __syncthreads();
  for (int i = t; i < NC * NP; i += 256) {
    const int ch = LAYOUT == 0 ? i / NP : i % NC, p = LAYOUT == 0 ? i % NP : i / NC;   // consecutive lanes: consecutive raw bytes
    const float s = scl[p], m = mn[p];
    unsigned pk[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c0 = ch * 16 + 2 * e;
      const float v0 = c0 < C ? mul_rounded(at(p, c0), s) + m : 0.f;
      const float v1 = c0 + 1 < C ? mul_rounded(at(p, c0 + 1), s) + m : 0.f;
      pk[e] = pack2_fmt(v0, v1, FMT_BF16);
    }
    u32x4* dst = reinterpret_cast<u32x4*>(out + ((size_t)ch * SS + p0 + p) * 16);
    dst[0] = u32x4{pk[0], pk[1], pk[2], pk[3]};
    dst[1] = u32x4{pk[4], pk[5], pk[6], pk[7]};
  }
}

template <typename T>
int launch_tiles_t(const CropArgs& a, int layout, unsigned short* tiles, hipStream_t st) {
  const size_t lds = 3 * (size_t)a.np * 4;
  const dim3 grid(a.B, (a.S * a.S + a.np - 1) / a.np);
  if (layout == 0) hipLaunchKernelGGL((k_preprocess_crops_tiles<T, 0>), grid, dim3(256), lds, st, a, tiles);
  else hipLaunchKernelGGL((k_preprocess_crops_tiles<T, 1>), grid, dim3(256), lds, st, a, tiles);
  DTA_CHECK_LAUNCH("k_preprocess_crops_tiles");
  return 0;
}

template <typename T>
int launch_t(const CropArgs& a, int layout, size_t lds, hipStream_t st) {
  if (layout == 0) {
    hipFuncSetAttribute((const void*)k_preprocess_crops<T, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k_preprocess_crops<T, 0>), dim3(a.B, (a.S * a.S + a.np - 1) / a.np), dim3(256), lds, st, a);
  } else {
    hipFuncSetAttribute((const void*)k_preprocess_crops<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k_preprocess_crops<T, 1>), dim3(a.B, (a.S * a.S + a.np - 1) / a.np), dim3(256), lds, st, a);
  }
  DTA_CHECK_LAUNCH("k_preprocess_crops");
  return 0;
}

}  // namespace

extern "C" int dta_preprocess_out_bands(int bands_raw, int clip) { return bands_raw > 3 ? bands_raw - 2 * clip : bands_raw; }

extern "C" int dta_preprocess_crops(const dta_crop_desc* d, const void* raw, const long long* offsets, const int* heights,
                                    const int* widths, float* out, void* stream) {
  if (!d || !raw || !offsets || !heights || !widths || !out) { dta_set_error("dta_preprocess_crops: null argument"); return 1; }
  if (d->batch < 1 || d->bands_raw < 1 || d->size < 1 || d->clip < 0) { dta_set_error("dta_preprocess_crops: bad descriptor"); return 1; }
  CropArgs a;
  a.raw = raw; a.off = offsets; a.hs = heights; a.ws = widths; a.out = out;
  a.B = d->batch; a.Craw = d->bands_raw; a.S = d->size; a.flip = d->flip != 0;
  a.c0 = d->bands_raw > 3 ? d->clip : 0;      // reference utils.py:40-42: bands are dropped only when there are more than 3
  a.C = dta_preprocess_out_bands(d->bands_raw, d->clip);
  if (a.C < 1) { dta_set_error("dta_preprocess_crops: %d bands leave nothing after dropping 2 x %d", d->bands_raw, d->clip); return 1; }
  const int SS = a.S * a.S;
  // a workgroup owns a run of at most 144 output pixels (an 11x11 crop is one workgroup, a 24x24 crop four): the LDS
  // footprint stays near 40 KB, so several workgroups share a CU and hide each other's load latency
  const int parts = (SS + 143) / 144;
  a.np = (SS + parts - 1) / parts;
  size_t lds_floats = 3 * (size_t)a.np;
  a.pitch = (a.np & 1) ? a.np : a.np + 1; a.cc = 0;
  if (d->layout == DTA_CROP_CHW) {
    int K = 256 / a.np; if (K < 1) K = 1; if (K > 8) K = 8;
    lds_floats += 2 * (size_t)K * a.np;
  } else if (d->layout == DTA_CROP_HWC) {
    a.cc = 64;
    lds_floats += (size_t)a.cc * a.pitch;
  } else { dta_set_error("dta_preprocess_crops: unknown layout %d", d->layout); return 1; }
  hipStream_t st = (hipStream_t)stream;
  switch (d->dtype) {
    case DTA_CROP_F32: return launch_t<float>(a, d->layout, lds_floats * 4, st);
    case DTA_CROP_I16: return launch_t<short>(a, d->layout, lds_floats * 4, st);
    case DTA_CROP_U8: return launch_t<unsigned char>(a, d->layout, lds_floats * 4, st);
    default: dta_set_error("dta_preprocess_crops: unknown raw dtype %d", d->dtype); return 1;
  }
}

extern "C" int dta_preprocess_crops_tiles(const dta_crop_desc* d, const void* raw, const long long* offsets, const int* heights,
                                          const int* widths, void* tiles, void* stream) {
  if (!d || !raw || !offsets || !heights || !widths || !tiles) { dta_set_error("dta_preprocess_crops_tiles: null argument"); return 1; }
  if (d->batch < 1 || d->bands_raw < 1 || d->size < 1 || d->clip < 0) { dta_set_error("dta_preprocess_crops_tiles: bad descriptor"); return 1; }
  if (d->layout != DTA_CROP_CHW && d->layout != DTA_CROP_HWC) { dta_set_error("dta_preprocess_crops_tiles: unknown layout %d", d->layout); return 1; }
  CropArgs a;
  a.raw = raw; a.off = offsets; a.hs = heights; a.ws = widths; a.out = nullptr;
  a.B = d->batch; a.Craw = d->bands_raw; a.S = d->size; a.flip = d->flip != 0;
  a.c0 = d->bands_raw > 3 ? d->clip : 0;
  a.C = dta_preprocess_out_bands(d->bands_raw, d->clip);
  if (a.C < 1) { dta_set_error("dta_preprocess_crops_tiles: %d bands leave nothing after dropping 2 x %d", d->bands_raw, d->clip); return 1; }
  const int SS = a.S * a.S, parts = (SS + 143) / 144;
  a.np = (SS + parts - 1) / parts; a.pitch = 0; a.cc = 0;
  hipStream_t st = (hipStream_t)stream;
  unsigned short* o = reinterpret_cast<unsigned short*>(tiles);
  switch (d->dtype) {
    case DTA_CROP_F32: return launch_tiles_t<float>(a, d->layout, o, st);
    case DTA_CROP_I16: return launch_tiles_t<short>(a, d->layout, o, st);
    case DTA_CROP_U8: return launch_tiles_t<unsigned char>(a, d->layout, o, st);
    default: dta_set_error("dta_preprocess_crops_tiles: unknown raw dtype %d", d->dtype); return 1;
  }
}
