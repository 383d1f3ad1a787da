// Development probe (not product): what does a cross-stream dependency (event record on one stream, hipStreamWaitEvent
// on another) cost on this part, per event flag set?  Round 1 measured 10-15 us per fork with torch's default events and
// rejected every side-stream overlap on that number; this probe owns its events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_events.hip -o tools/bin/probe_events
// Chain per iteration:   A (s0) -> B -> C (s0), with B either on s0 (baseline) or on s1 behind a fork event and in front
// of a join event.  Kernels spin for a fixed wall-clock time, so the GPU (not the host's enqueue rate) is the bottleneck.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <chrono>

__global__ void k_spin(int* p, int ticks) {   // ticks of the 100 MHz wall clock
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
  if (p && threadIdx.x == 9999) p[0] = 1;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t s0, s1;
  hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  int* flag; hipMalloc(&flag, 64);
  const int reps = 400;
  struct Fl { const char* name; unsigned f; } flags[] = {
    {"default (timing, system fence)", 0},
    {"DisableTiming", hipEventDisableTiming},
    {"DisableTiming|DisableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence},
    {"DisableTiming|ReleaseToDevice", hipEventDisableTiming | hipEventReleaseToDevice},
  };
  for (int spin_us : {3, 10, 30}) {
    const int ticks = spin_us * 100;
    auto run = [&](auto body) {
      for (int i = 0; i < 20; ++i) body();
      hipStreamSynchronize(s0); hipStreamSynchronize(s1);
      std::vector<double> t;
      for (int r = 0; r < 5; ++r) {
        const double a = now_us();
        for (int i = 0; i < reps; ++i) body();
        hipStreamSynchronize(s0); hipStreamSynchronize(s1);
        t.push_back((now_us() - a) / reps);
      }
      std::sort(t.begin(), t.end());
      return t[2];
    };
    const double base = run([&] {
      hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s0, flag, ticks);
      hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s0, flag, ticks);
      hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s0, flag, ticks);
    });
    printf("spin %2d us x3 on one stream: %7.2f us per iteration (%.2f per dependent launch boundary)\n", spin_us, base, (base - 3 * spin_us) / 3);
    for (auto& fl : flags) {
      hipEvent_t e0, e1;
      if (hipEventCreateWithFlags(&e0, fl.f) != hipSuccess || hipEventCreateWithFlags(&e1, fl.f) != hipSuccess) {
        printf("  %-36s: hipEventCreateWithFlags failed\n", fl.name); (void)hipGetLastError(); continue;
      }
      // serial through the side stream: A (s0) -> fork -> B (s1) -> join -> C (s0)
      const double ser = run([&] {
        hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s0, flag, ticks);
        hipEventRecord(e0, s0); hipStreamWaitEvent(s1, e0, 0);
        hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, flag, ticks);
        hipEventRecord(e1, s1); hipStreamWaitEvent(s0, e1, 0);
        hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s0, flag, ticks);
      });
      // overlapped: A -> fork -> [B on s1 || B' on s0] -> join -> C   (ideal = 3 spins + 2 boundaries)
      const double ovl = run([&] {
        hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s0, flag, ticks);
        hipEventRecord(e0, s0); hipStreamWaitEvent(s1, e0, 0);
        hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, flag, ticks);
        hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s0, flag, ticks);
        hipEventRecord(e1, s1); hipStreamWaitEvent(s0, e1, 0);
        hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s0, flag, ticks);
      });
      // fork only (side work nobody joins inside the iteration; joined by the NEXT iteration's A through e1)
      const double fork_only = run([&] {
        hipStreamWaitEvent(s0, e1, 0);
        hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s0, flag, ticks);
        hipEventRecord(e0, s0); hipStreamWaitEvent(s1, e0, 0);
        hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, flag, ticks);
        hipEventRecord(e1, s1);
        hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s0, flag, ticks);
        hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s0, flag, ticks);
      });
      printf("  %-36s: fork+join serial %7.2f (+%5.2f vs one stream) | overlapped 4 kernels %7.2f (ideal %.2f, +%5.2f) | fork now, join next iteration %7.2f (ideal %.2f)\n",
             fl.name, ser, ser - base, ovl, base, ovl - base, fork_only, base);
      hipEventDestroy(e0); hipEventDestroy(e1);
    }
  }
  return 0;
}
