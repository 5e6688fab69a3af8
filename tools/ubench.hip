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

// ---- gated experiment: the Montgomery reduction's two multiplications by constants (m = T_lo·p' mod R, then m·p) as
// byte-limb i8 MFMA products with Toeplitz matrices of p' and p, shared by a wave's 64 elements.  An UPPER BOUND on what that variant
// could reach: this kernel does everything the VALU would still have to do — the variable x variable product a·b (81 v_mad_u64_u32,
// no matrix form: both operands differ per lane), its carries, the repacking of T_lo into byte lanes, the carry resolution of the
// 33 and 66 int32 column sums an i8 MFMA returns (8-bit granularity: the matrix cores take 8-bit integers), the repacking into
// 29-bit limbs and the addition of T_hi — and treats the two MFMA products themselves, the operand staging and the cross-lane
// exchange their output layout needs as FREE (the "MFMA results" are opaque register values).  If even this is not 1.3x Fp29::mul,
// the real thing cannot be.
__device__ __forceinline__ uint32_t opaque(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ G mfma_bound_mul(const G& a, const G& b) {
  uint64_t acc[18];
#pragma unroll
  for (int k = 0; k < 18; k++) acc[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)a.v[i] * b.v[j];
  uint32_t t[18];
#pragma unroll
  for (int k = 0; k < 18; k++) { t[k] = (uint32_t)acc[k] & 0x1fffffffu; if (k < 17) acc[k + 1] += acc[k] >> 29; }
  // T_lo (261 bits) as contiguous 32-bit words = packed byte lanes of the MFMA operand
  uint32_t w[9]; { uint64_t buf = 0; int have = 0, o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { buf |= (uint64_t)t[i] << have; have += 29; if (have >= 32) { w[o++] = (uint32_t)buf; buf >>= 32; have -= 32; } }
    w[o] = (uint32_t)buf; }
  // [MFMA 1: 33 int32 column sums of T_lo·p' — free]  ->  m as 33 bytes, packed
  uint32_t mw[9]; { uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 33; k++) { const uint32_t s = opaque(w[k % 9] >> (k & 7)) + c; c = s >> 8; const uint32_t byte = s & 255u; if ((k & 3) == 0) mw[k >> 2] = byte; else mw[k >> 2] |= byte << (8 * (k & 3)); } }
  // [MFMA 2: 66 int32 column sums of m·p — free]  ->  (T + m·p) / 2^261: carries through the low 33 columns, bytes of the high 33
  uint32_t hw[9]; { uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 33; k++) { const uint32_t s = opaque(mw[k % 9] >> (k & 7)) + c; c = s >> 8; }
#pragma unroll
    for (int k = 0; k < 33; k++) { const uint32_t s = opaque(mw[(k + 3) % 9] >> (k & 7)) + c; c = s >> 8; const uint32_t byte = s & 255u; if ((k & 3) == 0) hw[k >> 2] = byte; else hw[k >> 2] |= byte << (8 * (k & 3)); } }
  // back to 29-bit limbs, plus T_hi
  G r; { uint32_t cy = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int bit = 29 * i, ww = bit >> 5, off = bit & 31;
      uint64_t x = (uint64_t)hw[ww] >> off;
      if (off > 3 && ww + 1 < 9) x |= (uint64_t)hw[ww + 1] << (32 - off);
      const uint32_t y = ((uint32_t)x & 0x1fffffffu) + t[9 + i] + cy; cy = y >> 29; r.v[i] = y & 0x1fffffffu;
    } }
  return r;
}
__global__ void k_fp29_mfma_bound(uint32_t* out, uint32_t seed, int iters) {
  G a = G::one(), b = G::one();
  a.v[0] ^= (threadIdx.x + seed) & 0xffff; b.v[1] ^= blockIdx.x & 0xffff;
  G c = a, d = b;
  for (int i = 0; i < iters; i++) { a = mfma_bound_mul(a, b); c = mfma_bound_mul(c, d); }
  a = G::add(a, c);
  uint32_t r = 0; for (int k = 0; k < 9; k++) r ^= a.v[k];
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
  { const double t29 = t;
    t = time_kernel(k_fp29_mfma_bound, dim3(blocks), dim3(tb), (uint32_t*)buf, 1u, it);
    printf("Fp29 product + byte-limb conversions of an i8-MFMA reduction, MFMA itself free (upper bound): %8.2f Gmul/s = %.2fx Fp29::mul\n", lanes * it * 2 / t * 1e-9, t29 / t); }
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
