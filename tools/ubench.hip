// Micro-benchmarks that size the 256-bit arithmetic design on gfx950: issue rates of the integer
// multiply-add forms, of f64 FMA, and the achieved rate of our Montgomery multiply and XYZZ mixed add.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench.hip -o tools/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../vimz_amd/csrc/ec.hpp"
using namespace vz;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;

__global__ void k_mad64(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
  for (int i = 0; i < ITERS; i++) {
    x0 = (uint64_t)(uint32_t)x0 * a + x0; x1 = (uint64_t)(uint32_t)x1 * b + x1; x2 = (uint64_t)(uint32_t)x2 * a + x2; x3 = (uint64_t)(uint32_t)x3 * b + x3;
    x4 = (uint64_t)(uint32_t)x4 * a + x4; x5 = (uint64_t)(uint32_t)x5 * b + x5; x6 = (uint64_t)(uint32_t)x6 * a + x6; x7 = (uint64_t)(uint32_t)x7 * b + x7;
  }
  uint64_t r = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}
__global__ void k_mullo(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed;
  uint32_t x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
  for (int i = 0; i < ITERS; i++) {
    x0 = x0 * a + 1; x1 = x1 * a + 1; x2 = x2 * a + 1; x3 = x3 * a + 1; x4 = x4 * a + 1; x5 = x5 * a + 1; x6 = x6 * a + 1; x7 = x7 * a + 1;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_mulhi(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed | 0x80000000u;
  uint32_t x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
  for (int i = 0; i < ITERS; i++) {
    x0 = __umulhi(x0, a) | 0x80000000u; x1 = __umulhi(x1, a) | 0x80000000u; x2 = __umulhi(x2, a) | 0x80000000u; x3 = __umulhi(x3, a) | 0x80000000u;
    x4 = __umulhi(x4, a) | 0x80000000u; x5 = __umulhi(x5, a) | 0x80000000u; x6 = __umulhi(x6, a) | 0x80000000u; x7 = __umulhi(x7, a) | 0x80000000u;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_mul24(uint32_t* out, uint32_t seed) {
  uint32_t a = (threadIdx.x * 2654435761u + seed) & 0xffffff;
  uint32_t x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
  for (int i = 0; i < ITERS; i++) {
    x0 = __umul24(x0 & 0xffffff, a) + 1; x1 = __umul24(x1 & 0xffffff, a) + 1;
    x2 = __umul24(x2 & 0xffffff, a) + 1; x3 = __umul24(x3 & 0xffffff, a) + 1;
    x4 = __umul24(x4 & 0xffffff, a) + 1; x5 = __umul24(x5 & 0xffffff, a) + 1;
    x6 = __umul24(x6 & 0xffffff, a) + 1; x7 = __umul24(x7 & 0xffffff, a) + 1;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_add32(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed;
  uint32_t x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
  for (int i = 0; i < ITERS; i++) {
    x0 = (x0 + a) ^ x1; x1 = (x1 + a) ^ x2; x2 = (x2 + a) ^ x3; x3 = (x3 + a) ^ x4; x4 = (x4 + a) ^ x5; x5 = (x5 + a) ^ x6; x6 = (x6 + a) ^ x7; x7 = (x7 + a) ^ x0;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_fma64(double* out, double seed) {
  double a = 1.0 + threadIdx.x * 1e-9 + seed, b = 1e-9;
  double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
  for (int i = 0; i < ITERS; i++) {
    x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
    x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_fma32(float* out, float seed) {
  float a = 1.0f + threadIdx.x * 1e-6f + seed, b = 1e-6f;
  float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
  for (int i = 0; i < ITERS; i++) {
    x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
    x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
typedef Fp<BnFq> F;
__global__ void k_fpmul(uint32_t* out, uint32_t seed, int iters) {
  F a = F::one(), b = F::r2();
  a.v[0] ^= threadIdx.x + seed; b.v[1] ^= blockIdx.x;
  F c = a, d = b;
  for (int i = 0; i < iters; i++) { a = F::mul(a, b); c = F::mul(c, d); }
  a = F::add(a, c);
  uint32_t r = 0; for (int k = 0; k < 8; k++) r ^= a.v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_madd(uint32_t* out, uint32_t seed, int iters) {
  Affine<F> q; q.x = F::one(); q.y = F::dbl(F::one());  // (1,2) is on BN254 G1
  XYZZ<F> acc = dbl_affine(q);
  acc.X.v[0] ^= threadIdx.x + seed; acc.Y.v[1] ^= blockIdx.x;  // lane-dependent (keeps the work on the VALU)
  for (int i = 0; i < iters; i++) add_mixed(acc, q);
  uint32_t r = 0; for (int k = 0; k < 8; k++) r ^= acc.X.v[k] ^ acc.ZZ.v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

typedef Fp29<BnFq> G;
__global__ void k_fp29mul(uint32_t* out, uint32_t seed, int iters) {
  G a = G::one(), b = G::one();
  a.v[0] ^= (threadIdx.x + seed) & 0xffff; b.v[1] ^= blockIdx.x & 0xffff;
  G c = a, d = b;
  for (int i = 0; i < iters; i++) { a = G::mul(a, b); c = G::mul(c, d); }
  a = G::add(a, c);
  uint32_t r = 0; for (int k = 0; k < 9; k++) r ^= a.v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_madd29(uint32_t* out, uint32_t seed, int iters) {
  Affine<G> q; q.x = G::one(); q.y = G::dbl(G::one());
  XYZZ<G> acc = dbl_affine(q);
  acc.X.v[0] ^= (threadIdx.x + seed) & 0xffff; acc.Y.v[1] ^= blockIdx.x & 0xffff;
  for (int i = 0; i < iters; i++) add_mixed(acc, q);
  uint32_t r = 0; for (int k = 0; k < 9; k++) r ^= acc.X.v[k] ^ acc.ZZ.v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <class K, class... A>
double time_kernel(K k, dim3 g, dim3 b, A... args) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, g, b, 0, 0, args...);  // warm
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, g, b, 0, 0, args...);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 3 * 1e-3;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s %s CUs=%d clock=%d MHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
  void* buf; CK(hipMalloc(&buf, 64 << 20));
  const int blocks = p.multiProcessorCount * 8, tb = 256;
  const double lanes = (double)blocks * tb;
  double t;
  t = time_kernel(k_mad64, dim3(blocks), dim3(tb), (uint32_t*)buf, 1u);  printf("v_mad_u64_u32   : %8.1f Gop/s\n", lanes * ITERS * 8 / t * 1e-9);
  t = time_kernel(k_mullo, dim3(blocks), dim3(tb), (uint32_t*)buf, 1u);  printf("mul_lo+add u32  : %8.1f Gop/s\n", lanes * ITERS * 8 / t * 1e-9);
  t = time_kernel(k_mulhi, dim3(blocks), dim3(tb), (uint32_t*)buf, 1u);  printf("mul_hi(+or) u32 : %8.1f Gop/s\n", lanes * ITERS * 8 / t * 1e-9);
  t = time_kernel(k_mul24, dim3(blocks), dim3(tb), (uint32_t*)buf, 1u);  printf("mul_u24(+and,add): %7.1f Gop/s\n", lanes * ITERS * 8 / t * 1e-9);
  t = time_kernel(k_add32, dim3(blocks), dim3(tb), (uint32_t*)buf, 1u);  printf("add+xor u32 pair: %8.1f Gpair/s\n", lanes * ITERS * 8 / t * 1e-9);
  t = time_kernel(k_fma64, dim3(blocks), dim3(tb), (double*)buf, 0.0);   printf("v_fma_f64       : %8.1f Gop/s\n", lanes * ITERS * 8 / t * 1e-9);
  t = time_kernel(k_fma32, dim3(blocks), dim3(tb), (float*)buf, 0.0f);   printf("v_fma_f32       : %8.1f Gop/s\n", lanes * ITERS * 8 / t * 1e-9);
  int it = 512;
  t = time_kernel(k_fpmul, dim3(blocks), dim3(tb), (uint32_t*)buf, 1u, it); printf("Fp::mul (BN254 Fq, 8x32 CIOS): %8.2f Gmul/s  (%.1f ns per wave-mul-pair)\n", lanes * it * 2 / t * 1e-9, t / it * 1e9);
  t = time_kernel(k_madd, dim3(blocks), dim3(tb), (uint32_t*)buf, 1u, it);  printf("XYZZ mixed add               : %8.2f Gadd/s\n", lanes * it / t * 1e-9);
  t = time_kernel(k_fp29mul, dim3(blocks), dim3(tb), (uint32_t*)buf, 1u, it); printf("Fp29::mul (BN254 Fq, 9x29 columns): %8.2f Gmul/s\n", lanes * it * 2 / t * 1e-9);
  t = time_kernel(k_madd29, dim3(blocks), dim3(tb), (uint32_t*)buf, 1u, it);  printf("XYZZ mixed add (Fp29)             : %8.2f Gadd/s\n", lanes * it / t * 1e-9);
  t = time_kernel(k_madd29, dim3(p.multiProcessorCount), dim3(tb), (uint32_t*)buf, 1u, it);
  printf("  mixed add (Fp29) @ 1 WG/CU       : %8.2f Gadd/s  (%.2f us per dependent add)\n", (double)p.multiProcessorCount * tb * it / t * 1e-9, t / it * 1e6);
  t = time_kernel(k_madd, dim3(p.multiProcessorCount), dim3(tb), (uint32_t*)buf, 1u, it);
  printf("  mixed add (8x32) @ 1 WG/CU       : %8.2f Gadd/s  (%.2f us per dependent add)\n", (double)p.multiProcessorCount * tb * it / t * 1e-9, t / it * 1e6);
  for (int wg = 1; wg <= 8; wg *= 2) {
    t = time_kernel(k_madd, dim3(p.multiProcessorCount * wg), dim3(tb), (uint32_t*)buf, 1u, it);
    printf("  mixed add @ %d WG/CU of 256: %8.2f Gadd/s\n", wg, (double)p.multiProcessorCount * wg * tb * it / t * 1e-9);
  }
  return 0;
}
