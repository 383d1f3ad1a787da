// Development probe (not product): what does this part's HBM deliver to a HAND-WRITTEN streaming kernel?
// (the anchor every "at the copy rate" statement used through round 5 was torch's copy kernel: 4.5-5.1 TB/s)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_stream.hip -o tools/bin/probe_stream
// Kernels: 16-byte (float4) read-only / write-only / copy, temporal and nontemporal, persistent grid-stride form;
// swept: workgroups per CU x independent 16-byte accesses in flight per lane x buffer size.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ src, float* sink, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  for (; i < n; i += stride) acc += src[i];
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) sink[0] = acc.x;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_write(f4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v, dst + i + u * stride); else dst[i + u * stride] = v;
    }
  }
  for (; i < n; i += stride) dst[i] = v;
}

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NTS) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

// block-contiguous form: a workgroup owns a contiguous 64 KB-multiple span (what a tile-per-workgroup kernel does)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read_span(const f4* __restrict__ src, float* sink, size_t n) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  size_t i = b0 + threadIdx.x;
  for (; i + (U - 1) * 256 < b1; i += U * 256) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  for (; i < b1; i += 256) acc += src[i];
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) sink[0] = acc.x;
}

static hipStream_t st;
template <typename F> static float time_us(F f, int reps = 12) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f(); f();
  hipStreamSynchronize(st);
  std::vector<float> t;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(a, st); f(); hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  hipEventDestroy(a); hipEventDestroy(b);
  return t[t.size() / 2];
}

template <int U> static void sweep_u(const f4* src, f4* dst, float* sink, size_t bytes, int ncu) {
  const size_t n = bytes / 16;
  const int wgs[] = {1, 2, 4, 8, 16};
  for (int w : wgs) {
    const int grid = ncu * w;
    float r_t = time_us([&] { hipLaunchKernelGGL((k_read<U, false>), dim3(grid), dim3(256), 0, st, src, sink, n); });
    float r_n = time_us([&] { hipLaunchKernelGGL((k_read<U, true>), dim3(grid), dim3(256), 0, st, src, sink, n); });
    float s_n = time_us([&] { hipLaunchKernelGGL((k_read_span<U, true>), dim3(grid), dim3(256), 0, st, src, sink, n); });
    float w_t = time_us([&] { hipLaunchKernelGGL((k_write<U, false>), dim3(grid), dim3(256), 0, st, dst, n); });
    float w_n = time_us([&] { hipLaunchKernelGGL((k_write<U, true>), dim3(grid), dim3(256), 0, st, dst, n); });
    float c_tt = time_us([&] { hipLaunchKernelGGL((k_copy<U, false, false>), dim3(grid), dim3(256), 0, st, src, dst, n); });
    float c_nn = time_us([&] { hipLaunchKernelGGL((k_copy<U, true, true>), dim3(grid), dim3(256), 0, st, src, dst, n); });
    float c_nt = time_us([&] { hipLaunchKernelGGL((k_copy<U, true, false>), dim3(grid), dim3(256), 0, st, src, dst, n); });
    const double gb = bytes / 1e3;   // bytes / us -> GB/s = bytes / (us * 1e3)
    printf("%6zu MB  wg/CU %2d  U %d | read t %5.0f nt %5.0f span-nt %5.0f | write t %5.0f nt %5.0f | copy(r+w) tt %5.0f nn %5.0f nt-load %5.0f  GB/s\n",
           bytes >> 20, w, U, gb / r_t, gb / r_n, gb / s_n, gb / w_t, gb / w_n, 2 * gb / c_tt, 2 * gb / c_nn, 2 * gb / c_nt);
    fflush(stdout);
  }
}

int main(int argc, char** argv) {
  hipStreamCreate(&st);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  printf("# %s, %d CUs; median of 12 launches each, HIP events on the launch stream; copy = bytes read + bytes written\n", p.name, ncu);
  const size_t maxb = (size_t)2048 << 20;
  f4 *src, *dst; float* sink;
  hipMalloc(&src, maxb); hipMalloc(&dst, maxb); hipMalloc(&sink, 64);
  hipMemset(src, 0, maxb); hipMemset(dst, 0, maxb);
  const size_t sizes[] = {(size_t)256 << 20, (size_t)512 << 20, (size_t)1024 << 20, (size_t)2048 << 20};
  const bool quick = argc > 1;
  for (size_t b : sizes) {
    if (quick && b != ((size_t)1024 << 20)) continue;
    sweep_u<1>(src, dst, sink, b, ncu);
    sweep_u<2>(src, dst, sink, b, ncu);
    sweep_u<4>(src, dst, sink, b, ncu);
    sweep_u<8>(src, dst, sink, b, ncu);
  }
  // torch-like reference point: hipMemcpyAsync device to device
  for (size_t b : sizes) {
    float t = time_us([&] { hipMemcpyAsync(dst, src, b, hipMemcpyDeviceToDevice, st); });
    printf("%6zu MB  hipMemcpyAsync D2D: %5.0f GB/s (r+w)\n", b >> 20, 2.0 * b / 1e3 / t);
  }
  return 0;
}
