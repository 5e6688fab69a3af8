// Pippenger multi-scalar multiplication for gfx950 — replaces `cpu_best_multiexp` / `pasta_msm`
// behind `CommitmentEngine::commit` in nova-snark 0.23.0 (SURVEY.md §8a rows M1/M2, §8b "MSM" seam).
//
// Pipeline (all on one HIP stream, no host round trip until the K window sums come back):
//   1. k_hist      signed-digit recode of every scalar (window c bits, digits in [-2^(c-1), 2^(c-1)]),
//                  histogram of (window, |digit|) buckets with global atomics.
//   2. k_scan      one workgroup: exclusive scan of bucket sizes -> entry offsets, and of
//                  ceil(size/SUB) -> sub-bucket offsets.  Large buckets (witness scalars are ~95 % bits
//                  and bytes, so bucket "1" of window 0 can hold a third of all points) are split into
//                  sub-buckets of at most SUB entries so no thread owns an unbounded chain.
//   3. k_scatter   counting-sort scatter of (point index | sign) into bucket order.
//   4. k_accum     one thread per sub-bucket: gathers its affine bases (64 B each, served from L2 /
//                  Infinity Cache — the base table is reused by every window) and accumulates in XYZZ.
//   5. k_combine   log2 passes folding the sub-bucket partials of each bucket pairwise.
//   6. k_reduce    per window: chunked running sums + LDS tree -> Σ b·B_b.
//   7. host        Horner over the K window sums (K·c doublings) and one inversion to affine.
// Addition order inside a bucket depends on atomics, but the result is an exact group element, so
// the affine output is bit-identical run to run and to the CPU oracle.
#pragma once
#include "msm_api.hpp"

namespace vz {

// ---- device helpers --------------------------------------------------------------------------

// bits [lo, lo+c) of a 256-bit little-endian integer held in 8 registers (c <= 16)
__device__ __forceinline__ uint32_t window_bits(const uint32_t* s, int lo, int c) {
  int limb = lo >> 5, off = lo & 31;
  if (limb >= 8) return 0;
  uint64_t v = s[limb];
  if (limb + 1 < 8) v |= (uint64_t)s[limb + 1] << 32;
  return (uint32_t)(v >> off) & ((1u << c) - 1);
}

template <class S>
__device__ __forceinline__ void load_scalar(const uint32_t* __restrict__ scalars, size_t i, int mont, uint32_t* s) {
  const uint4* p = reinterpret_cast<const uint4*>(scalars + 8 * i);
  uint4 a = p[0], b = p[1];
  S x;
  x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w; x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w;
  if (mont) x = S::from_mont(x);
#pragma unroll
  for (int k = 0; k < 8; k++) s[k] = x.v[k];
}

// Calls f(window, bucket_index_in_window, negative) for every non-zero signed digit.
template <class Fn>
__device__ __forceinline__ void for_each_digit(const uint32_t* s, int c, int K, Fn f) {
  uint32_t carry = 0;
  const uint32_t half = 1u << (c - 1);
  for (int w = 0; w < K; w++) {
    uint32_t d = window_bits(s, w * c, c) + carry;
    uint32_t neg = d > half;
    uint32_t mag = neg ? (1u << c) - d : d;
    carry = neg;
    if (mag) f(w, mag - 1, neg);
  }
}

// Wave-aggregated "fetch-and-increment": lanes of a wave that target the same counter are served by one
// atomic.  Witness scalars are ~80 % bits, so most lanes of a wave hit the same (window 0, digit 1) bucket and
// plain per-lane atomics serialise on one address (measured 1.4 ms vs 0.2 ms for the histogram at n = 315 k).
// Called by the active lanes only; falls back to per-lane atomics once the groups get small.
__device__ __forceinline__ uint32_t agg_atomic_inc(uint32_t* __restrict__ ctr, uint32_t g) {
  const uint32_t lane = __lane_id();
  bool done = false;
  uint32_t pos = 0;
  for (int round = 0; round < 4; round++) {
    const uint64_t pending = __ballot(!done);
    if (pending == 0) break;
    const int leader = __ffsll((unsigned long long)pending) - 1;
    const uint32_t lg = __shfl(g, leader);
    const bool mine = !done && g == lg;
    const uint64_t same = __ballot(mine);
    uint32_t base = 0;
    if (mine && lane == (uint32_t)leader) base = atomicAdd(&ctr[g], (uint32_t)__popcll(same));
    base = __shfl(base, leader);
    if (mine) { pos = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull)); done = true; }
    if (__popcll(same) < 4) break;
  }
  if (!done) pos = atomicAdd(&ctr[g], 1u);
  return pos;
}

template <class S>
__global__ void k_hist(const uint32_t* __restrict__ scalars, size_t n, int mont, int c, int K, uint32_t nbw,
                       uint32_t* __restrict__ counts) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t s[8];
    load_scalar<S>(scalars, i, mont, s);
    for_each_digit(s, c, K, [&](int w, uint32_t b, uint32_t) { agg_atomic_inc(counts, (uint32_t)w * nbw + b); });
  }
}

// One workgroup of 1024 threads; nb is a few 10^4..10^6.
template <int SUB>
__global__ void __launch_bounds__(1024) k_scan(const uint32_t* __restrict__ counts, uint32_t nb,
                                               uint32_t* __restrict__ bucket_off, uint32_t* __restrict__ sub_off,
                                               uint32_t* __restrict__ totals) {
  __shared__ uint32_t sh_e[1024], sh_s[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (nb + 1023) / 1024;
  const uint32_t lo = min(nb, t * per), hi = min(nb, lo + per);
  uint32_t e = 0, s = 0;
  for (uint32_t b = lo; b < hi; b++) { uint32_t cnt = counts[b]; e += cnt; s += (cnt + SUB - 1) / SUB; }
  sh_e[t] = e; sh_s[t] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
    uint32_t ve = 0, vs = 0;
    if (t >= d) { ve = sh_e[t - d]; vs = sh_s[t - d]; }
    __syncthreads();
    sh_e[t] += ve; sh_s[t] += vs;
    __syncthreads();
  }
  uint32_t be = sh_e[t] - e, bs = sh_s[t] - s;
  for (uint32_t b = lo; b < hi; b++) {
    uint32_t cnt = counts[b];
    bucket_off[b] = be; sub_off[b] = bs;
    be += cnt; bs += (cnt + SUB - 1) / SUB;
  }
  if (t == 1023) { bucket_off[nb] = sh_e[1023]; sub_off[nb] = sh_s[1023]; totals[0] = sh_s[1023]; totals[1] = sh_e[1023]; }
}

template <class S>
__global__ void k_scatter(const uint32_t* __restrict__ scalars, size_t n, int mont, int c, int K, uint32_t nbw,
                          const uint32_t* __restrict__ bucket_off, uint32_t* __restrict__ cursor,
                          uint32_t* __restrict__ sorted) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t s[8];
    load_scalar<S>(scalars, i, mont, s);
    for_each_digit(s, c, K, [&](int w, uint32_t b, uint32_t neg) {
      uint32_t g = (uint32_t)w * nbw + b;
      uint32_t pos = bucket_off[g] + agg_atomic_inc(cursor, g);
      sorted[pos] = (uint32_t)i | (neg << 31);
    });
  }
}

template <class F>
__device__ __forceinline__ Affine<F> load_affine(const uint32_t* __restrict__ bases, uint32_t idx) {
  const uint4* p = reinterpret_cast<const uint4*>(bases + 16 * (size_t)idx);
  uint4 a = p[0], b = p[1], c = p[2], d = p[3];
  Affine<F> q;
  q.x.v[0] = a.x; q.x.v[1] = a.y; q.x.v[2] = a.z; q.x.v[3] = a.w; q.x.v[4] = b.x; q.x.v[5] = b.y; q.x.v[6] = b.z; q.x.v[7] = b.w;
  q.y.v[0] = c.x; q.y.v[1] = c.y; q.y.v[2] = c.z; q.y.v[3] = c.w; q.y.v[4] = d.x; q.y.v[5] = d.y; q.y.v[6] = d.z; q.y.v[7] = d.w;
  return q;
}

template <class F>
__device__ __forceinline__ void store_xyzz(XYZZ<F>* dst, const XYZZ<F>& p) {
  uint4* o = reinterpret_cast<uint4*>(dst);
  const F* f[4] = {&p.X, &p.Y, &p.ZZ, &p.ZZZ};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    o[2 * k] = make_uint4(f[k]->v[0], f[k]->v[1], f[k]->v[2], f[k]->v[3]);
    o[2 * k + 1] = make_uint4(f[k]->v[4], f[k]->v[5], f[k]->v[6], f[k]->v[7]);
  }
}
template <class F>
__device__ __forceinline__ XYZZ<F> load_xyzz(const XYZZ<F>* src) {
  const uint4* o = reinterpret_cast<const uint4*>(src);
  XYZZ<F> p;
  F* f[4] = {&p.X, &p.Y, &p.ZZ, &p.ZZZ};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint4 a = o[2 * k], b = o[2 * k + 1];
    f[k]->v[0] = a.x; f[k]->v[1] = a.y; f[k]->v[2] = a.z; f[k]->v[3] = a.w;
    f[k]->v[4] = b.x; f[k]->v[5] = b.y; f[k]->v[6] = b.z; f[k]->v[7] = b.w;
  }
  return p;
}

template <class F>
__global__ void __launch_bounds__(256) k_accum(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                               const uint32_t* __restrict__ bucket_off, const uint32_t* __restrict__ sub_off,
                                               uint32_t nb, const uint32_t* __restrict__ totals,
                                               XYZZ<F>* __restrict__ partial, uint32_t* __restrict__ sub_bucket,
                                               uint32_t* __restrict__ sub_k) {
  const uint32_t total = totals[0];
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  // largest b with sub_off[b] <= s  (empty buckets share an offset with their successor; pick the last)
  uint32_t lo = 0, hi = nb;  // invariant: sub_off[lo] <= s < sub_off[hi]
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (sub_off[mid] <= s) lo = mid; else hi = mid;
  }
  const uint32_t b = lo, k = s - sub_off[b];
  const uint32_t beg = bucket_off[b] + k * MSM_SUB;
  const uint32_t end = min(bucket_off[b + 1], beg + MSM_SUB);
  XYZZ<F> acc = XYZZ<F>::identity();
  for (uint32_t e = beg; e < end; e++) {
    uint32_t ent = sorted[e];
    Affine<F> q = load_affine<F>(bases, ent & 0x7fffffffu);
    if ((ent >> 31) && !aff_is_identity(q)) q.y = F::neg(q.y);
    add_mixed(acc, q);
  }
  store_xyzz(&partial[s], acc);
  sub_bucket[s] = b; sub_k[s] = k;
}

template <class F>
__global__ void __launch_bounds__(256) k_combine(XYZZ<F>* __restrict__ partial, const uint32_t* __restrict__ sub_bucket,
                                                 const uint32_t* __restrict__ sub_k, const uint32_t* __restrict__ totals,
                                                 uint32_t stride) {
  const uint32_t total = totals[0];
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total || s + stride >= total) return;
  if (sub_k[s] % (2 * stride) != 0) return;
  if (sub_bucket[s + stride] != sub_bucket[s]) return;
  XYZZ<F> a = load_xyzz(&partial[s]);
  XYZZ<F> b = load_xyzz(&partial[s + stride]);
  add_full(a, b);
  store_xyzz(&partial[s], a);
}

// grid = K workgroups of T threads (T = min(256, nbw)); window sum = Σ_{idx} (idx+1)·B_idx
template <class F>
__global__ void __launch_bounds__(256) k_reduce(const XYZZ<F>* __restrict__ partial, const uint32_t* __restrict__ counts,
                                                const uint32_t* __restrict__ sub_off, uint32_t nbw,
                                                XYZZ<F>* __restrict__ window_sums) {
  __shared__ XYZZ<F> sh[256];
  const uint32_t w = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  const uint32_t ch = nbw / T;
  const uint32_t lo = t * ch;
  XYZZ<F> run = XYZZ<F>::identity(), sum = XYZZ<F>::identity();
  for (uint32_t j = ch; j-- > 0;) {
    uint32_t g = w * nbw + lo + j;
    if (counts[g]) { XYZZ<F> B = load_xyzz(&partial[sub_off[g]]); add_full(run, B); }
    add_full(sum, run);
  }
  // sum = Σ (j+1)·B_{lo+j}; add lo·run
  if (lo) {
    XYZZ<F> acc = XYZZ<F>::identity();
    for (int bit = 31 - __clz(lo); bit >= 0; bit--) {
      acc = dbl(acc);
      if ((lo >> bit) & 1) add_full(acc, run);
    }
    add_full(sum, acc);
  }
  sh[t] = sum;
  __syncthreads();
  for (uint32_t d = T >> 1; d > 0; d >>= 1) {
    if (t < d) { XYZZ<F> a = sh[t]; add_full(a, sh[t + d]); sh[t] = a; }
    __syncthreads();
  }
  if (t == 0) store_xyzz(&window_sums[w], sh[0]);
}

// ---- host driver ---------------------------------------------------------------------------------

template <class C>
hipError_t msm_run(hipStream_t stream, MsmWorkspace& ws, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n,
                   int scalars_mont, int c_override, Affine<typename C::Base>* out_affine_mont, MsmStats* stats,
                   hipEvent_t* ev /* 7 events or nullptr */) {
  typedef typename C::Base F;
  typedef typename C::Scalar S;
  static_assert(sizeof(XYZZ<F>) == 128, "XYZZ layout");
  if (n == 0) { out_affine_mont->x = F::zero(); out_affine_mont->y = F::zero(); return hipSuccess; }
  if (n >= (1u << 31)) return hipErrorInvalidValue;
  const MsmPlan pl = msm_plan(n, S::Params::BITS, c_override);
  if (pl.K > MSM_MAX_WINDOWS || pl.c > 16 || pl.c < 2) return hipErrorInvalidValue;
  const size_t entries = (size_t)pl.K * n;
  const size_t max_subs = entries / MSM_SUB + pl.nb + 1;
  VZ_HIP_CHECK(ws.reserve(pl.nb, entries, max_subs));

  VZ_HIP_CHECK(hipMemsetAsync(ws.counts, 0, 4 * (size_t)pl.nb, stream));
  VZ_HIP_CHECK(hipMemsetAsync(ws.cursor, 0, 4 * (size_t)pl.nb, stream));
  const int TB = 256;
#define VZ_EV(i) do { if (ev) VZ_HIP_CHECK(hipEventRecord(ev[i], stream)); } while (0)
  VZ_EV(0);
  const unsigned gs = (unsigned)std::min<size_t>((n + TB - 1) / TB, 256 * 16);
  hipLaunchKernelGGL(k_hist<S>, dim3(gs), dim3(TB), 0, stream, d_scalars, n, scalars_mont, pl.c, pl.K, pl.nbw, ws.counts);
  VZ_EV(1);
  hipLaunchKernelGGL(k_scan<MSM_SUB>, dim3(1), dim3(1024), 0, stream, ws.counts, pl.nb, ws.bucket_off, ws.sub_off, ws.totals);
  VZ_EV(2);
  hipLaunchKernelGGL(k_scatter<S>, dim3(gs), dim3(TB), 0, stream, d_scalars, n, scalars_mont, pl.c, pl.K, pl.nbw,
                     ws.bucket_off, ws.cursor, ws.sorted);
  VZ_EV(3);
  XYZZ<F>* partial = reinterpret_cast<XYZZ<F>*>(ws.partial);
  const unsigned ga = (unsigned)((max_subs + TB - 1) / TB);
  hipLaunchKernelGGL(k_accum<F>, dim3(ga), dim3(TB), 0, stream, d_bases, ws.sorted, ws.bucket_off, ws.sub_off, pl.nb,
                     ws.totals, partial, ws.sub_bucket, ws.sub_k);
  VZ_EV(4);
  // a bucket holds at most n entries -> at most ceil(n/SUB) sub-buckets -> that many halving passes
  const size_t max_m = (n + MSM_SUB - 1) / MSM_SUB;
  for (size_t stride = 1; stride < max_m; stride <<= 1)
    hipLaunchKernelGGL(k_combine<F>, dim3(ga), dim3(TB), 0, stream, partial, ws.sub_bucket, ws.sub_k, ws.totals, (uint32_t)stride);
  VZ_EV(5);
  const unsigned T = pl.nbw < 256 ? pl.nbw : 256;
  XYZZ<F>* wsum = reinterpret_cast<XYZZ<F>*>(ws.window_sums);
  hipLaunchKernelGGL(k_reduce<F>, dim3(pl.K), dim3(T), 0, stream, partial, ws.counts, ws.sub_off, pl.nbw, wsum);
  VZ_EV(6);
#undef VZ_EV
  VZ_HIP_CHECK(hipGetLastError());
  VZ_HIP_CHECK(hipMemcpyAsync(ws.host_pinned, wsum, sizeof(XYZZ<F>) * pl.K, hipMemcpyDeviceToHost, stream));
  uint32_t h_tot[2] = {0, 0};
  if (stats) VZ_HIP_CHECK(hipMemcpyAsync(h_tot, ws.totals, 8, hipMemcpyDeviceToHost, stream));
  VZ_HIP_CHECK(hipStreamSynchronize(stream));

  const XYZZ<F>* hw = reinterpret_cast<const XYZZ<F>*>(ws.host_pinned);
  XYZZ<F> acc = XYZZ<F>::identity();
  for (int w = pl.K - 1; w >= 0; w--) {
    for (int k = 0; k < pl.c; k++) acc = dbl(acc);
    add_full(acc, hw[w]);
  }
  *out_affine_mont = to_affine(acc);
  if (stats && ev) for (int i = 0; i < 6; i++) VZ_HIP_CHECK(hipEventElapsedTime(&stats->ms[i], ev[i], ev[i + 1]));
  if (stats) { stats->c = pl.c; stats->K = pl.K; stats->subs = h_tot[0]; stats->entries = h_tot[1]; }
  return hipSuccess;
}

}  // namespace vz
