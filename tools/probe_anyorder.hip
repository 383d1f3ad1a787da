// Probe: does hipExtAnyOrderLaunch let two independent kernels in one stream overlap on gfx950?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
  float *a, *b;
  hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20);
  hipStream_t st; hipStreamCreate(&st);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipStreamSynchronize(st);
      auto t0 = std::chrono::high_resolution_clock::now();
      for (int i = 0; i < 200; ++i) {
        if (mode == 0) {
          hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, a, 20000);
          hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, b, 20000);
        } else if (mode == 1) {
          hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, nullptr, nullptr, 0, a, 20000);
          hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, b, 20000);
        } else {
          hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, st, a, 20000);
        }
      }
      hipStreamSynchronize(st);
      double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 200;
      printf("mode %d (%s): %.1f us per pair\n", mode, mode == 0 ? "in-order" : mode == 1 ? "second any-order" : "single launch of both", us);
    }
  }
  return 0;
}
