// Development probe (not product): what does a tiny dependent launch cost on this GPU, and how long do the two
// BatchNorm finalize launches of the step really take back-to-back?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_launch.hip -Ideeptreeattention_amd/csrc \
//       -Ldeeptreeattention_amd -ldta_hip -Wl,-rpath,'$ORIGIN/../deeptreeattention_amd' -o tools/bin/probe_launch
#include <stdio.h>
#include <string.h>
#include <vector>
#include "kernels.h"
using namespace dta;

__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ __launch_bounds__(1024) void k_empty1024(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
struct Big { char pad[1800]; int* p; };
__global__ void k_empty_bigarg(Big b) { if (b.p && threadIdx.x == 9999) b.p[0] = 1; }
__global__ void k_touch(float* p, int n) {   // streaming kernel: every block writes 64 KB
  size_t i = (size_t)blockIdx.x * 16384 + threadIdx.x;
  for (int k = 0; k < 64; ++k) p[i + k * 256] = (float)k;
}

template <typename F> float timeit(const char* name, int reps, hipStream_t st, F f) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 10; ++i) f();
  hipStreamSynchronize(st);
  hipEventRecord(a, st);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b, st);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  printf("%-58s %8.2f us per iteration\n", name, ms * 1e3f / reps);
  return ms * 1e3f / reps;
}

int main() {
  hipStream_t st;
  hipStreamCreate(&st);
  const int B = 1024;
  float* buf; hipMalloc(&buf, 512u << 20); hipMemset(buf, 0, 512u << 20);
  float* par; hipMalloc(&par, 1 << 20);
  std::vector<float> ones(1 << 18, 1.f);
  hipMemcpy(par, ones.data(), 1 << 20, hipMemcpyHostToDevice);
  long long* nbt; hipMalloc(&nbt, 64); hipMemset(nbt, 0, 64);
  int* flag; hipMalloc(&flag, 64);
  timeit("empty<<<1,64>>>", 500, st, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, flag); });
  timeit("empty<<<256,256>>>", 500, st, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, flag); });
  timeit("empty<<<2048,256>>>", 500, st, [&] { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, st, flag); });
  timeit("empty1024<<<4,1024>>>", 500, st, [&] { hipLaunchKernelGGL(k_empty1024, dim3(4), dim3(1024), 0, st, flag); });
  Big big; memset(&big, 0, sizeof(big)); big.p = flag;
  timeit("empty_bigarg(1.8 KB kernarg)<<<256,256>>>", 500, st, [&] { hipLaunchKernelGGL(k_empty_bigarg, dim3(256), dim3(256), 0, st, big); });
  float t_touch = timeit("touch 128 MB <<<2048,256>>>", 200, st, [&] { hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, st, buf, 0); });
  float t_pair = timeit("touch + empty<<<4,1024>>>", 200, st, [&] {
    hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, st, buf, 0);
    hipLaunchKernelGGL(k_empty1024, dim3(4), dim3(1024), 0, st, flag); });
  printf("  -> a tiny launch behind a streaming kernel adds %.2f us\n", t_pair - t_touch);

  // the step's BatchNorm finalize launches at the bench geometry
  for (int L = 0; L < 3; ++L) {
    const int C = L == 0 ? 32 : (L == 1 ? 64 : 128), HW = L == 2 ? 25 : 121, G = 2;
    const bool cat = L == 0;
    const int Nconv = cat ? 64 : C, MWG = conv_mwg(Nconv);
    int ppw, spp, nwg;
    conv_geometry(HW, MWG, B, &ppw, &spp, &nwg);
    BnFinalizeArgs bf; memset(&bf, 0, sizeof(bf));
    bf.stats = buf; bf.nwg = nwg; bf.N = Nconv; bf.HW = HW; bf.MWG = MWG; bf.B = B;
    for (int g = 0; g < G; ++g) { bf.gamma[g] = par; bf.beta[g] = par + 4096; bf.rmean[g] = par + 8192 + g * 256; bf.rvar[g] = par + 16384 + g * 256; bf.nbt[g] = nbt + g; }
    bf.cat_mode = cat; bf.nsplit = 32; bf.coef = par + 32768; bf.training = 1; bf.momentum = 0.1f; bf.eps = 1e-5f;
    char name[128];
    snprintf(name, sizeof(name), "bn_finalize layer %d (nwg=%d, C=%d) alone", L, nwg, C);
    float t_f = timeit(name, 300, st, [&] { launch_bn_finalize(bf, G, st); });
    snprintf(name, sizeof(name), "touch + bn_finalize layer %d", L);
    float t_tf = timeit(name, 200, st, [&] { hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, st, buf + (64u << 20), 0); launch_bn_finalize(bf, G, st); });
    printf("  -> behind a streaming kernel it adds %.2f us (alone %.2f)\n", t_tf - t_touch, t_f);
    BnBwdFinalizeArgs bb; memset(&bb, 0, sizeof(bb));
    bb.bnpart = buf; bb.bnpart_gs = (size_t)B * C * 2; bb.B = B; bb.C = C; bb.HW = HW;
    bb.coef = par + 32768; bb.coef_gs = C * 4;
    for (int g = 0; g < G; ++g) { bb.gamma[g] = par; bb.dgamma[g] = par + 65536 + g * 256; bb.dbeta[g] = par + 70000 + g * 256; bb.dconvbias[g] = par + 75000 + g * 256; }
    bb.bcoef = par + 80000; bb.bcoef_gs = C * 4; bb.training = 1;
    snprintf(name, sizeof(name), "bn_bwd_finalize layer %d alone", L);
    float t_b = timeit(name, 300, st, [&] { launch_bn_bwd_finalize(bb, G, st); });
    snprintf(name, sizeof(name), "touch + bn_bwd_finalize layer %d", L);
    float t_tb = timeit(name, 200, st, [&] { hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, st, buf + (64u << 20), 0); launch_bn_bwd_finalize(bb, G, st); });
    printf("  -> behind a streaming kernel it adds %.2f us (alone %.2f)\n", t_tb - t_touch, t_b);
  }
  return 0;
}
