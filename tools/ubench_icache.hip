// Does the instruction footprint of the MSM tail kernels cost anything next to the bulk accumulation?  
// Two kernels side by side on two streams:
//   bulk      the accumulation's inner loop (a chain of mixed additions per lane, 3 workgroups of 256 per CU: k_accum's occupancy; 18 KB of code)
//   tail      222 workgroups of 256 (k_msm_small's shape), every lane a chain of N dependent FULL additions — once as a loop (one copy of the addition,
//             28 KB of code), once fully unrolled (N copies: N x 28 KB of straight-line code, what the unrolled tree levels of the tail kernels amount to)
// Same arithmetic, same occupancy, same issue-slot demand: whatever the unrolled variant costs on top of the looped one — to itself or to the bulk kernel
// beside it — is the instruction cache's share.  PMC passes cannot show this (they serialise kernels).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_icache.hip -o tools/ubench_icache ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../vimz_amd/csrc/ec.hpp"
using namespace vz;
typedef Fp29<BnFq> G;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256, 3) k_bulk(uint32_t* out, uint32_t seed, int iters) {
  Affine<G> q; q.x = G::one(); q.y = G::dbl(G::one());
  XYZZ<G> acc = dbl_affine(q);
  acc.X.v[0] ^= (threadIdx.x + seed) & 0xffff; acc.Y.v[1] ^= blockIdx.x & 0xffff;
  for (int i = 0; i < iters; i++) add_mixed(acc, q);
  uint32_t r = 0; for (int k = 0; k < 9; k++) r ^= acc.X.v[k] ^ acc.ZZ.v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__device__ __forceinline__ XYZZ<G> start_point(uint32_t seed) {
  Affine<G> q; q.x = G::one(); q.y = G::dbl(G::one());
  XYZZ<G> p = dbl_affine(q);
  p.X.v[0] ^= (threadIdx.x + seed) & 0xffff; p.Y.v[1] ^= blockIdx.x & 0xffff;
  return p;
}
constexpr int TAIL_N = 48;
__global__ void __launch_bounds__(256) k_tail_loop(uint32_t* out, uint32_t seed, int n, int prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);      // (as k_msm_small and k_spmv_cross16 do)
  XYZZ<G> acc = start_point(seed), b = start_point(seed + 7);
#pragma nounroll
  for (int i = 0; i < n; i++) { add_full(acc, b); b.X.v[0] ^= (uint32_t)i & 1u; }
  uint32_t r = 0; for (int k = 0; k < 9; k++) r ^= acc.X.v[k] ^ acc.ZZ.v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_tail_straight(uint32_t* out, uint32_t seed) {
  XYZZ<G> acc = start_point(seed), b = start_point(seed + 7);
  // (written out: the compiler declines to unroll a body of this size)
#define VZ_A1(i) { add_full(acc, b); b.X.v[0] ^= (uint32_t)(i) & 1u; }
#define VZ_A4(i) VZ_A1(i) VZ_A1(i + 1) VZ_A1(i + 2) VZ_A1(i + 3)
#define VZ_A16(i) VZ_A4(i) VZ_A4(i + 4) VZ_A4(i + 8) VZ_A4(i + 12)
  VZ_A16(0) VZ_A16(16) VZ_A16(32)
  static_assert(TAIL_N == 48, "three blocks of sixteen");
  uint32_t r = 0; for (int k = 0; k < 9; k++) r ^= acc.X.v[k] ^ acc.ZZ.v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// in between: eight rounds of a body of six written-out additions (about 310 KB: the size of the real tail kernels after round 5)
__global__ void __launch_bounds__(256) k_tail_mid(uint32_t* out, uint32_t seed, int rounds) {
  XYZZ<G> acc = start_point(seed), b = start_point(seed + 7);
#pragma nounroll
  for (int r = 0; r < rounds; r++) { VZ_A4(r) VZ_A1(r) VZ_A1(r + 1) }
  uint32_t r = 0; for (int k = 0; k < 9; k++) r ^= acc.X.v[k] ^ acc.ZZ.v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s CUs=%d\n", p.gcnArchName, p.multiProcessorCount);
  uint32_t *b1, *b2; CK(hipMalloc((void**)&b1, 64 << 20)); CK(hipMalloc((void**)&b2, 64 << 20));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t a0, a1, t0, t1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  const int bulk_blocks = p.multiProcessorCount * 3, bulk_iters = 512, tail_blocks = 222, reps = 12;
  auto bulk = [&](hipStream_t s) { hipLaunchKernelGGL(k_bulk, dim3(bulk_blocks), dim3(256), 0, s, b1, 1u, bulk_iters); };
  auto tail = [&](hipStream_t s, int straight) {
    if (straight == 1) hipLaunchKernelGGL(k_tail_straight, dim3(tail_blocks), dim3(256), 0, s, b2, 3u);
    else if (straight == 2) hipLaunchKernelGGL(k_tail_mid, dim3(tail_blocks), dim3(256), 0, s, b2, 3u, TAIL_N / 6);
    else hipLaunchKernelGGL(k_tail_loop, dim3(tail_blocks), dim3(256), 0, s, b2, 3u, TAIL_N, straight == 3 ? 1 : 0);
  };
  float ms;
  for (int w = 0; w < 2; w++) { bulk(s1); tail(s2, 0); tail(s2, 1); tail(s2, 2); tail(s2, 3); }
  CK(hipDeviceSynchronize());
  // alone
  CK(hipEventRecord(a0, s1)); bulk(s1); CK(hipEventRecord(a1, s1)); CK(hipEventSynchronize(a1)); CK(hipEventElapsedTime(&ms, a0, a1));
  const double bulk_alone = ms;
  printf("bulk alone: %.3f ms (%d mixed additions per lane, %d workgroups)\n", bulk_alone, bulk_iters, bulk_blocks);
  double tail_alone[4];
  const char* names[4] = {"looped (28 KB)", "written out (2.5 MB)", "8 x 6 written out (310 KB)", "looped, s_setprio(3)"};
  for (int st = 0; st < 4; st++) {
    CK(hipEventRecord(t0, s2)); for (int r = 0; r < reps; r++) tail(s2, st); CK(hipEventRecord(t1, s2)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
    tail_alone[st] = ms / reps;
    printf("tail %-28s alone: %.3f ms per launch (%d dependent full additions per lane, %d workgroups)\n", names[st], tail_alone[st], TAIL_N, tail_blocks);
  }
  // side by side: the bulk kernel on one stream, tail launches back to back on the other for as long as it runs
  for (int round = 0; round < 3; round++)
    for (int st = 0; st < 4; st++) {
      CK(hipDeviceSynchronize());
      const int n_tail = (int)(bulk_alone * 1.6 / tail_alone[st]) + 1;
      CK(hipEventRecord(a0, s1)); CK(hipEventRecord(t0, s2));
      bulk(s1);
      for (int r = 0; r < n_tail; r++) tail(s2, st);
      CK(hipEventRecord(a1, s1)); CK(hipEventRecord(t1, s2));
      CK(hipEventSynchronize(a1)); CK(hipEventSynchronize(t1));
      float mb, mt; CK(hipEventElapsedTime(&mb, a0, a1)); CK(hipEventElapsedTime(&mt, t0, t1));
      printf("side by side, tail %-28s: bulk %.3f ms (x%.2f), tail %.3f ms per launch (x%.2f) over %d launches\n", names[st], mb, mb / bulk_alone,
             mt / n_tail, mt / n_tail / tail_alone[st], n_tail);
    }
  return 0;
}
