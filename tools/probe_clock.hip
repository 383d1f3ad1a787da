// Developer probe: what do clock64() (s_memtime) and wall_clock64() (s_memrealtime) count on this part?
// One wave spins for a while; both counters are read at the ends, HIP events time the launch.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long* out, int iters) {
  const long long c0 = clock64(), w0 = wall_clock64();
  float x = threadIdx.x;
  for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
  const long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
int main() {
  long long* d; hipMalloc(&d, 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, d, 4000000); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("launch %.3f ms: clock64 %lld ticks (%.1f MHz), wall_clock64 %lld ticks (%.1f MHz), %.2f clock64 ticks per loop iteration\n",
           ms, h[0], h[0] / ms / 1e3, h[1], h[1] / ms / 1e3, (double)h[0] / 4e6);
  }
  return 0;
}
