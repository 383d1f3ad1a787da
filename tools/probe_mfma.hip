// Hardware probe (development tool, not product): confirms the gfx950 MFMA fragment maps and the
// ds_read_b64_tr_b16 gather this repo's conv kernels rely on.  Prints PASS/FAIL per item.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ inline unsigned short f2bf(float f) { unsigned u = __float_as_uint(f); u += 0x7FFF + ((u >> 16) & 1); return u >> 16; }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short hf2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return u >> 16; }

__global__ void k_bf16_32(const unsigned short* A, const unsigned short* B, float* D) {  // A[32][16], B[16][32]
  int l = threadIdx.x; bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j]; b[j] = B[(8 * (l >> 5) + j) * 32 + (l & 31)]; }
  f32x16 c = {0}; c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void k_f32_32(const float* A, const float* B, float* D) {  // A[32][2], B[2][32]
  int l = threadIdx.x; f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void k_bf16_16(const unsigned short* A, const unsigned short* B, float* D) {  // A[16][32], B[32][16]
  int l = threadIdx.x; bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j]; b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)]; }
  f32x4 c = {0}; c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void k_f32_16(const float* A, const float* B, float* D) {  // A[16][4], B[4][16]
  int l = threadIdx.x; f32x4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// transpose read: lane i of a 16-lane group supplies the address of 4 consecutive bf16 (row i/4, cols 4*(i%4)..),
// rows `stride` elements apart; expected result: lane i receives column i of that 4x16 block (rows 0..3).
__global__ void k_tr(unsigned short* out, int stride) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int g = l >> 4, i = l & 15;
  unsigned short* p = &lds[g * 4 * stride + (i >> 2) * stride + (i & 3) * 4];
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

template <class F> static bool check(const char* name, int M, int N, const std::vector<float>& ref, F get) {
  double worst = 0; for (int i = 0; i < M * N; ++i) worst = fmax(worst, fabs(ref[i] - get(i)));
  printf("%-28s %s (max abs err %.3g)\n", name, worst < 1e-3 ? "PASS" : "FAIL", worst); return worst < 1e-3;
}
int main() {
  bool ok = true;
  auto rnd = [](int i) { return (float)(((i * 2654435761u) >> 20) % 17) / 8.f - 1.f; };
  {  // bf16 32x32x16
    std::vector<unsigned short> A(32 * 16), B(16 * 32); std::vector<float> R(32 * 32, 0), D(32 * 32);
    for (int i = 0; i < 512; ++i) { A[i] = hf2bf(rnd(i)); B[i] = hf2bf(rnd(i + 7777)); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += bf2f(A[i * 16 + k]) * bf2f(B[k * 32 + j]);
    unsigned short *dA, *dB; float* dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    k_bf16_32<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    ok &= check("mfma_f32_32x32x16_bf16", 32, 32, R, [&](int i) { return D[i]; });
  }
  {  // f32 32x32x2
    std::vector<float> A(64), B(64), R(1024, 0), D(1024);
    for (int i = 0; i < 64; ++i) { A[i] = rnd(i); B[i] = rnd(i + 999); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 2; ++k) R[i * 32 + j] += A[i * 2 + k] * B[k * 32 + j];
    float *dA, *dB, *dD; hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
    k_f32_32<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    ok &= check("mfma_f32_32x32x2f32", 32, 32, R, [&](int i) { return D[i]; });
  }
  {  // bf16 16x16x32
    std::vector<unsigned short> A(512), B(512); std::vector<float> R(256, 0), D(256);
    for (int i = 0; i < 512; ++i) { A[i] = hf2bf(rnd(i + 5)); B[i] = hf2bf(rnd(i + 4242)); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += bf2f(A[i * 32 + k]) * bf2f(B[k * 16 + j]);
    unsigned short *dA, *dB; float* dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    k_bf16_16<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    ok &= check("mfma_f32_16x16x32_bf16", 16, 16, R, [&](int i) { return D[i]; });
  }
  {  // f32 16x16x4
    std::vector<float> A(64), B(64), R(256, 0), D(256);
    for (int i = 0; i < 64; ++i) { A[i] = rnd(i + 3); B[i] = rnd(i + 31); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    float *dA, *dB, *dD; hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
    k_f32_16<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    ok &= check("mfma_f32_16x16x4f32", 16, 16, R, [&](int i) { return D[i]; });
  }
  for (int stride : {16, 40, 24}) {  // tr read, several row strides (elements)
    unsigned short* dO; hipMalloc(&dO, 512); std::vector<unsigned short> O(256);
    k_tr<<<1, 64>>>(dO, stride); hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      int g = l >> 4, i = l & 15; int exp = g * 4 * stride + j * stride + i;
      if (O[l * 4 + j] != exp) { if (bad < 6) printf("  tr stride %d lane %d j %d got %d exp %d\n", stride, l, j, O[l * 4 + j], exp); ++bad; }
    }
    printf("ds_read_b64_tr_b16 stride %-3d   %s\n", stride, bad ? "FAIL" : "PASS"); ok &= !bad;
  }
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("device %s CUs %d clock %d kHz LDS/block %zu\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate, pr.sharedMemPerBlock);
  printf(ok ? "ALL PASS\n" : "SOME FAIL\n");
  return ok ? 0 : 1;
}
