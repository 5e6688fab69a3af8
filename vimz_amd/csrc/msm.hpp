// Pippenger multi-scalar multiplication for gfx950 — replaces `cpu_best_multiexp` / `pasta_msm`
// behind `CommitmentEngine::commit` in nova-snark 0.23.0 (SURVEY.md §8a rows M1/M2, §8b "MSM" seam).
//
// Pipeline (all on one HIP stream, no host round trip until the window sums come back):
//   1. k_hist_lds / k_prefix_scan    signed-digit recode of every scalar (window c bits, digits in [-2^(c-1), 2^(c-1)]);
//                  per-workgroup LDS histograms of the (window, |digit|) buckets, then bucket totals and per-workgroup
//                  offsets — no global atomics (k_hist / k_scatter with wave-aggregated global atomics remain as the
//                  fallback for windows whose counters do not fit LDS).
//   2. (scan)      exclusive scan of bucket sizes -> entry offsets, and of ceil(size/SUB) -> sub-bucket offsets, by the last workgroup
//                  of k_prefix_scan to finish (k_scan, a launch of its own, on the fallback path).  Buckets are split into sub-buckets of
//                  at most SUB entries so no thread owns a long chain.
//   3. k_scatter_lds  counting-sort scatter of (point index | sign) into bucket order, ranks from LDS counters.
//   4. k_accum     one thread per sub-bucket: gathers its affine bases (64 B each — one sector —, from L2 / Infinity Cache — the key is
//                  re-read by every window) and accumulates in XYZZ over the 9x29-bit coordinate field.
//   5. k_combine   folds the sub-bucket partials of each bucket; hot buckets (listed by the scan) by workgroup trees in the same launch;
//                  the second stage of the very heavy ones rides in k_reduce's prologue.
//   6. k_reduce    per window: chunked running sums + LDS tree -> sum_b b*B_b.
//      (unit scalars, when split: k_ones_partial + k_tree256 -> one extra "window sum")
//   7. host        Horner over the K window sums (K*c doublings) and one inversion to affine (msm_finish).
// Optional window tables (k_build_tables, vimz_bases_precompute): one bucket set for all windows, reduced as virtual windows of
// MSM_VWIN buckets by k_reduce (R_v, S_v), no Horner.
// Addition order inside a bucket is arbitrary, but the result is an exact group element, so the affine output is
// bit-identical run to run and to the CPU oracle.
#pragma once
#include <cstdlib>
#include <cstring>
#include <mutex>
#include "msm_api.hpp"
#include "ec_mem.hpp"

namespace vz {

// The two shape parameters of the large-MSM pipeline that are chosen from the input size can be pinned from the environment, ONE
// variable read once:  VIMZ_TUNE="sort_blocks=256,combine_lane_bits=4"  (workgroups of the LDS counting sort; log2 of the lanes per
// bucket in k_combine).  Results never depend on them — tests/test_gpu_ivc.py folds the same rows under other values and requires
// the identical proof.  (The switches of rounds 1-2 whose A/B is settled — launcher thread, copied window sums, CU masks, stream
// priorities, sub-bucket length, late folds, one producer stream — are gone; DESIGN.md §9 keeps what they measured.)
// small_lean (0: off, the default): the fused small MSM's wide tail levels by one lane per addition instead of four — 15-20 % fewer
// instructions per small MSM, 7-27 µs more latency: measured SLOWER in every regime (three segments 1095-1101 -> 1054-1078 -> 1029-1051
// steps/s for lean = 0 / 1 / 2, one chain 819 -> 799 -> 770): with the GPU 98 % busy the step is still bound by its latency chains.
// reduce_planes (0: off, the default): the shared bucket set of a table MSM reduced by bit planes (k_reduce_planes) instead of chunked running sums
// (k_reduce) — alone on the GPU the reduce falls from 0.171 to 0.083 ms (305 k dense points, c = 15) and the whole MSM from 0.873 to 0.786 ms, but
// every bucket is then added into half of the 14 planes: 131 k full additions instead of 32 k, a tenth of the accumulation's work on top — and inside
// a fold, where the GPU is busy throughout, that costs more than the shorter tail gains: 1 176 against 1 198 steps/s over 256 rows, 936 against 938 in
// the 20-row window, one chain 829 against 822 (round 5, same box, profiles/r05_reduce_planes.txt).
struct MsmTuning { int sort_blocks = 0, combine_lane_bits = -1, small_lean = 0, witness_sub = 0, ones_dense = 1, reduce_planes = 0, accum_lds_kb = 0, dense_sub = 0; };
inline const MsmTuning& msm_tuning() {
  static const MsmTuning t = [] {
    MsmTuning r;
    if (const char* e = getenv("VIMZ_TUNE")) {
      if (const char* q = strstr(e, "sort_blocks=")) r.sort_blocks = atoi(q + 12);
      if (const char* q = strstr(e, "combine_lane_bits=")) r.combine_lane_bits = atoi(q + 18);
      if (const char* q = strstr(e, "small_lean=")) r.small_lean = atoi(q + 11);
      if (const char* q = strstr(e, "witness_sub=")) { const int v = atoi(q + 12); if (v >= 2 && v <= MSM_SUB) r.witness_sub = v; }
      if (const char* q = strstr(e, "dense_sub=")) { const int v = atoi(q + 10); if (v >= 2 && v <= MSM_SUB) r.dense_sub = v; }
      if (const char* q = strstr(e, "ones_dense=")) r.ones_dense = atoi(q + 11);
      if (const char* q = strstr(e, "reduce_planes=")) r.reduce_planes = atoi(q + 14);
      if (const char* q = strstr(e, "accum_lds_kb=")) { const int v = atoi(q + 13); if (v >= 0 && v <= 160) r.accum_lds_kb = v; }
    }
    return r;
  }();
  return t;
}

// ---- device helpers --------------------------------------------------------------------------

// bits [lo, lo+c) of a 256-bit little-endian integer held in 8 registers (c <= 16)
__device__ __forceinline__ uint32_t window_bits(const uint32_t* s, int lo, int c) {
  int limb = lo >> 5, off = lo & 31;
  if (limb >= 8) return 0;
  uint64_t v = s[limb];
  if (limb + 1 < 8) v |= (uint64_t)s[limb + 1] << 32;
  return (uint32_t)(v >> off) & ((1u << c) - 1);
}

// returns false when the scalar is to be skipped (zero, or the unit when `skip_ones`: units are summed by k_ones_partial)
template <class S>
__device__ __forceinline__ bool load_scalar(const uint32_t* __restrict__ scalars, size_t i, int mont, int skip_ones, uint32_t* s) {
  const uint4* p = reinterpret_cast<const uint4*>(scalars + 8 * i);
  uint4 a = p[0], b = p[1];
  S x;
  x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w; x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w;
  if (x.is_zero()) return false;
  if (skip_ones) {
    bool one;
    if (mont) one = x.eq(S::one());
    else { uint32_t o = x.v[0] ^ 1u; for (int k = 1; k < 8; k++) o |= x.v[k]; one = o == 0; }
    if (one) return false;
  }
  if (mont) x = S::from_mont(x);
#pragma unroll
  for (int k = 0; k < 8; k++) s[k] = x.v[k];
  return true;
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
__global__ void k_hist(const uint32_t* __restrict__ scalars, size_t n, int mont, int skip_ones, int c, int K, uint32_t nbw /* bucket stride per window: 2^(c-1), or 0 with window tables */,
                       uint32_t* __restrict__ counts) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t s[8];
    if (!load_scalar<S>(scalars, i, mont, skip_ones, s)) continue;
    for_each_digit(s, c, K, [&](int w, uint32_t b, uint32_t) { agg_atomic_inc(counts, (uint32_t)w * nbw + b); });
  }
}

// One workgroup of 1024 threads; nb is a few 10^4..10^6.
__device__ __forceinline__ void scan_body(const uint32_t* __restrict__ counts, uint32_t nb,
                                               uint32_t* __restrict__ bucket_off, uint32_t* __restrict__ sub_off,
                                               uint32_t* __restrict__ totals, uint32_t sub,
                                               uint32_t* __restrict__ heavy /* [0] = count (zeroed by the caller), then ids of buckets with > heavy_min sub-buckets */,
                                               uint32_t heavy_min, uint32_t heavy_cap) {
  // One pass, coalesced: the buckets are taken 1024 at a time (thread t: bucket base + t); each block is scanned inside its waves
  // by shuffles and across the 16 wave totals through LDS (one barrier per block, the totals double-buffered), carrying the running
  // totals along.  (This kernel sits between the two sort passes of every MSM, on a step's dependent chain.  The first version gave
  // every thread 24 consecutive buckets — two strided passes over the counters and a twenty-barrier Hillis-Steele scan: 50 µs.)
  __shared__ uint32_t wv_e[2][16], wv_s[2][16];
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  uint32_t carry_e = 0, carry_s = 0;
  for (uint32_t base = 0, it = 0; base < nb; base += 1024, it++) {
    const uint32_t b = base + t;
    const uint32_t cnt = b < nb ? counts[b] : 0u, m = (cnt + sub - 1) / sub;
    uint32_t ie = cnt, is = m;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint32_t ve = __shfl_up(ie, d), vs = __shfl_up(is, d);
      if (lane >= d) { ie += ve; is += vs; }
    }
    if (lane == 63) { wv_e[it & 1][wv] = ie; wv_s[it & 1][wv] = is; }
    __syncthreads();
    uint32_t oe = 0, os = 0, te = 0, ts = 0;
    for (uint32_t q = 0; q < 16; q++) { const uint32_t xe = wv_e[it & 1][q], xs = wv_s[it & 1][q]; if (q < wv) { oe += xe; os += xs; } te += xe; ts += xs; }
    if (b < nb) {
      bucket_off[b] = carry_e + oe + ie - cnt; sub_off[b] = carry_s + os + is - m;
      if (m > heavy_min) { const uint32_t slot = atomicAdd(&heavy[0], 1u); if (slot < heavy_cap) heavy[1 + slot] = b; }
    }
    carry_e += te; carry_s += ts;
  }
  if (t == 0) { bucket_off[nb] = carry_e; sub_off[nb] = carry_s; totals[0] = carry_s; totals[1] = carry_e; }
}
template <int SUB>
__global__ void __launch_bounds__(1024) k_scan(const uint32_t* __restrict__ counts, uint32_t nb, uint32_t* __restrict__ bucket_off, uint32_t* __restrict__ sub_off,
                                               uint32_t* __restrict__ totals, uint32_t sub, uint32_t* __restrict__ heavy, uint32_t heavy_min, uint32_t heavy_cap) {
  scan_body(counts, nb, bucket_off, sub_off, totals, sub, heavy, heavy_min, heavy_cap);
}

template <class S>
__global__ void k_scatter(const uint32_t* __restrict__ scalars, size_t n, int mont, int skip_ones, int c, int K, uint32_t nbw,
                          uint32_t pt_stride /* 0, or the table row length: entry = window * pt_stride + point */,
                          const uint32_t* __restrict__ bucket_off, uint32_t* __restrict__ cursor,
                          uint32_t* __restrict__ sorted) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t s[8];
    if (!load_scalar<S>(scalars, i, mont, skip_ones, s)) continue;
    for_each_digit(s, c, K, [&](int w, uint32_t b, uint32_t neg) {
      uint32_t g = (uint32_t)w * nbw + b;
      uint32_t pos = bucket_off[g] + agg_atomic_inc(cursor, g);
      sorted[pos] = ((uint32_t)i + (uint32_t)w * pt_stride) | (neg << 31);
    });
  }
}

// Curve data in HBM: coordinates of COORD_WORDS (10) words, 9 used — every point starts 16-byte aligned.
// ---- contention-free counting sort ---------------------------------------------------------------------------------
// The cross-term vector T of a real fold is far from uniform: wires that were 1 in every row so far share one running
// value, so tens of thousands of scalars are EQUAL and hit the same 24 buckets.  Device-scope atomics on one address
// serialise across XCDs (~0.5 us each), which made k_hist / k_scatter 5x slower on real data than on random data.
// Here each workgroup histograms its own contiguous chunk of scalars in LDS (all buckets fit: 24 x 1024 counters = 96 KiB),
// a scan kernel turns the per-workgroup histograms into global bucket sizes and per-workgroup offsets, and the scatter
// ranks its entries with LDS atomics again.  No global atomic is issued at all.
constexpr uint32_t SORT_BLOCKS = 256;      // at most one workgroup per CU (msm_launch picks fewer for small inputs)
constexpr uint32_t SORT_THREADS = 1024;

template <class S>
__global__ void __launch_bounds__(SORT_THREADS) k_hist_lds(const uint32_t* __restrict__ scalars, size_t n, int mont, int skip_ones, int c, int K,
                                                           uint32_t nbw, uint32_t nb, uint32_t* __restrict__ block_hist /* [SORT_BLOCKS][nb] */) {
  extern __shared__ uint32_t lds_cnt[];
  for (uint32_t g = threadIdx.x; g < nb; g += SORT_THREADS) lds_cnt[g] = 0;
  __syncthreads();
  const size_t chunk = (n + gridDim.x - 1) / gridDim.x;
  const size_t lo = blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += SORT_THREADS) {
    uint32_t s[8];
    if (!load_scalar<S>(scalars, i, mont, skip_ones, s)) continue;
    for_each_digit(s, c, K, [&](int w, uint32_t b, uint32_t) { atomicAdd(&lds_cnt[(uint32_t)w * nbw + b], 1u); });
  }
  __syncthreads();
  uint32_t* out = block_hist + (size_t)blockIdx.x * nb;
  for (uint32_t g = threadIdx.x; g < nb; g += SORT_THREADS) out[g] = lds_cnt[g];
}

// The per-bucket prefix over the sort workgroups' histograms and the scan in ONE launch (round 4; k_block_prefix + k_scan before): every workgroup
// turns its 1024 buckets' per-workgroup histograms into exclusive prefixes (block_hist[blk][g]) and totals (counts[g]); the LAST workgroup to finish (a ticket in totals[2]) runs the scan over all totals.  In a fold a launch — however small — waits ~0.1 ms for
// its turn next to three segments' kernels (profiles/r04_msm_chain_gaps_HD.txt: k_scan 93 µs, k_block_prefix 120 µs, an EMPTY k_combine_heavy2 114 µs):
// the large MSM's chain is two launches shorter.  Release / acquire as in k_msm_small: stores, fence, agent-scope ticket; fence, plain loads.
template <int DUMMY>
__global__ void __launch_bounds__(1024) k_prefix_scan(uint32_t* __restrict__ block_hist, uint32_t nb, uint32_t* __restrict__ counts, uint32_t nblocks,
                                                      uint32_t* __restrict__ heavy, uint32_t* __restrict__ bucket_off, uint32_t* __restrict__ sub_off,
                                                      uint32_t* __restrict__ totals, uint32_t sub, uint32_t heavy_min, uint32_t heavy_cap) {
  __shared__ uint32_t s_last;
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < nb) {
    uint32_t run = 0;
    for (uint32_t blk = 0; blk < nblocks; blk++) {
      const size_t idx = (size_t)blk * nb + g;
      const uint32_t v = block_hist[idx];
      block_hist[idx] = run;
      run += v;
    }
    counts[g] = run;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&totals[2], 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (threadIdx.x == 0) { __hip_atomic_store(&totals[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); heavy[0] = 0; }      // (the ticket for the next MSM; the heavy list starts empty)
  __syncthreads();
  scan_body(counts, nb, bucket_off, sub_off, totals, sub, heavy, heavy_min, heavy_cap);
}

template <class S>
__global__ void __launch_bounds__(SORT_THREADS) k_scatter_lds(const uint32_t* __restrict__ scalars, size_t n, int mont, int skip_ones, int c, int K,
                                                              uint32_t nbw, uint32_t nb, uint32_t pt_stride, const uint32_t* __restrict__ bucket_off,
                                                              const uint32_t* __restrict__ block_hist, uint32_t* __restrict__ sorted) {
  extern __shared__ uint32_t lds_pos[];
  const uint32_t* mine = block_hist + (size_t)blockIdx.x * nb;
  for (uint32_t g = threadIdx.x; g < nb; g += SORT_THREADS) lds_pos[g] = bucket_off[g] + mine[g];
  __syncthreads();
  const size_t chunk = (n + gridDim.x - 1) / gridDim.x;
  const size_t lo = blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += SORT_THREADS) {
    uint32_t s[8];
    if (!load_scalar<S>(scalars, i, mont, skip_ones, s)) continue;
    for_each_digit(s, c, K, [&](int w, uint32_t b, uint32_t neg) {
      const uint32_t pos = atomicAdd(&lds_pos[(uint32_t)w * nbw + b], 1u);
      sorted[pos] = ((uint32_t)i + (uint32_t)w * pt_stride) | (neg << 31);
    });
  }
}

template <class F>
__global__ void __launch_bounds__(256, 3) k_accum(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                               const uint32_t* __restrict__ bucket_off, const uint32_t* __restrict__ sub_off,
                                               uint32_t nb, const uint32_t* __restrict__ totals,
                                               uint32_t* __restrict__ partial, uint32_t sub) {
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
  const uint32_t beg = bucket_off[b] + k * sub;
  const uint32_t end = min(bucket_off[b + 1], beg + sub);
  XYZZ<F> acc = XYZZ<F>::identity();
  // software pipeline: the gather of entry e+1 (index, then the 64-byte base: one sector) is in flight while entry e is added
  uint32_t ent = beg < end ? sorted[beg] : 0u;
  RawAffine raw = load_affine_raw(bases, ent & 0x7fffffffu);
  for (uint32_t e = beg; e < end; e++) {
    const uint32_t ent_n = e + 1 < end ? sorted[e + 1] : ent;
    const RawAffine raw_n = load_affine_raw(bases, ent_n & 0x7fffffffu);
    Affine<F> q = affine_of_raw<F>(raw);
    if ((ent >> 31) && !aff_is_identity(q)) q.y = F::neg(q.y);
    add_mixed(acc, q);
    ent = ent_n; raw = raw_n;
  }
  store_xyzz(partial, s, acc);
}

// ---- unit scalars -------------------------------------------------------------------------------------------------
// A fresh witness is ~80 % bits, so its commitment is mostly "the sum of the bases whose wire is 1".  Pushing those
// through the sort makes one bucket hold a third of all points and serialises ~10^3 device-scope atomics on one
// address.  Instead the units are summed directly: thread t adds the bases of the unit scalars among t, t+G, t+2G, ...
// (coalesced scalar reads; which subset a thread takes is irrelevant, everything is summed), then a two-level LDS tree.
constexpr uint32_t ONES_THREADS = 16384;   // partial sums of level 0
template <class S>
__device__ __forceinline__ bool scalar_is_one(const uint32_t* __restrict__ scalars, size_t i, int mont) {
  const uint4* p = reinterpret_cast<const uint4*>(scalars + 8 * i);
  const uint4 a = p[0], b = p[1];
  if (mont) return a.x == S::Params::R1.w[0] && a.y == S::Params::R1.w[1] && a.z == S::Params::R1.w[2] && a.w == S::Params::R1.w[3] &&
                   b.x == S::Params::R1.w[4] && b.y == S::Params::R1.w[5] && b.z == S::Params::R1.w[6] && b.w == S::Params::R1.w[7];
  return a.x == 1u && (a.y | a.z | a.w | b.x | b.y | b.z | b.w) == 0u;
}
template <class S, class F>
__global__ void __launch_bounds__(256) k_ones_partial(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ bases, size_t n, int mont,
                                                      uint32_t* __restrict__ out /* ONES_THREADS XYZZ */) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  XYZZ<F> acc = XYZZ<F>::identity();
  for (size_t i = t; i < n; i += ONES_THREADS)
    if (scalar_is_one<S>(scalars, i, mont)) { Affine<F> q = load_affine<F>(bases, (uint32_t)i); add_mixed(acc, q); }
  store_xyzz(out, t, acc);
}
// The same sums with the unit scalars COMPACTED first (round 4): in the loop above a wave runs an addition whenever one of its 64 lanes
// holds a unit — at 40 % units that is every iteration, 19 additions per lane for 7.5 useful ones.  Here a wave walks 64 consecutive scalars
// at a time, queues the indices of the units in LDS (ballot + prefix count) and adds 64 queued bases at a time, one per lane; which lane
// adds which base is irrelevant, everything is summed.  The queue is the wave's own: LDS operations of one wave execute in order.
template <class S, class F>
__global__ void __launch_bounds__(256) k_ones_dense(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ bases, size_t n, int mont,
                                                    uint32_t* __restrict__ out /* ONES_THREADS XYZZ */) {
  __shared__ uint32_t queue[4][128];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const size_t wave = t >> 6, nwaves = ONES_THREADS / 64;
  volatile uint32_t* q = queue[wv];
  XYZZ<F> acc = XYZZ<F>::identity();
  uint32_t cnt = 0;                                     // (wave-uniform)
  for (size_t base = wave * 64; base < n; base += nwaves * 64) {
    const size_t i = base + lane;
    const bool one = i < n && scalar_is_one<S>(scalars, i, mont);
    const uint64_t m = __ballot(one);
    if (one) q[cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
    cnt += (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
    if (cnt >= 64) {
      cnt -= 64;
      const uint32_t idx = q[cnt + lane];
      __builtin_amdgcn_wave_barrier();
      Affine<F> pt = load_affine<F>(bases, idx);
      add_mixed(acc, pt);
    }
  }
  if (lane < cnt) { Affine<F> pt = load_affine<F>(bases, q[lane]); add_mixed(acc, pt); }
  store_xyzz(out, t, acc);
}
// in: m points, out: ceil(m/256) points (one LDS tree per workgroup)
template <class F>
__global__ void __launch_bounds__(256) k_tree256(const uint32_t* __restrict__ in, uint32_t m, uint32_t* __restrict__ out) {
  __shared__ XYZZ<F> sh[256];
  const uint32_t t = threadIdx.x, i = blockIdx.x * 256 + t;
  sh[t] = i < m ? load_xyzz<F>(in, i) : XYZZ<F>::identity();
  __syncthreads();
  quad_tree256<F>(sh);
  if (t == 0) store_xyzz(out, blockIdx.x, sh[0]);
}


constexpr uint32_t MSM_HEAVY_PARTS = 32, MSM_HEAVY_SPLIT = 1024;
constexpr uint32_t MSM_HEAVY_SPLIT_MIN = 2048;   // buckets with fewer partials than this are left to one workgroup (one tree, no second stage)
constexpr uint32_t COMBINE_HEAVY_BLOCKS = 512, COMBINE_SPLIT_BLOCKS = 1024;

// ONE launch folds every bucket's sub-bucket partials into partial[sub_off[b]], three kinds of workgroups side by side
// (they touch disjoint buckets, and run as separate kernels they cost three serial tails of dependent additions):
//   blocks [0, nbn)                      ordinary buckets (<= heavy_min partials): 2..16 lanes per bucket, strided sums + a short LDS tree;
//   next COMBINE_HEAVY_BLOCKS            heavy buckets (list written by k_scan), one workgroup each: strided sums + 8-level tree;
//   next COMBINE_SPLIT_BLOCKS            very heavy buckets (>= MSM_HEAVY_SPLIT_MIN partials; among the first MSM_HEAVY_SPLIT of the list):
//                                        stage 1 of a two-stage fold, MSM_HEAVY_PARTS workgroups per bucket into a scratch row
//                                        the last of a bucket's workgroups to finish folds the row (ticket in heavy_done).
// Why buckets get that heavy: with 254-bit scalars and c = 11 the top window holds one bit plus a carry, so a third of ALL points
// of a dense MSM meet in one or two buckets; repeated cross-term values and the small values of a witness add hot buckets.
template <class F>
__global__ void __launch_bounds__(256) k_combine(uint32_t* __restrict__ partial, const uint32_t* __restrict__ sub_off, uint32_t nb, uint32_t nbn,
                                                 const uint32_t* __restrict__ heavy, uint32_t heavy_cap, uint32_t* scratch /* (written and, by the last part, read: no restrict) */,
                                                 uint32_t lane_bits /* log2 of the lanes per ordinary bucket: 0..4 */, uint32_t heavy_min,
                                                 uint32_t* __restrict__ heavy_done /* one ticket per split bucket */) {
  __shared__ XYZZ<F> sh[256];
  __shared__ uint32_t s_last;
  const uint32_t t = threadIdx.x;
  if (blockIdx.x < nbn) {
    // The lanes of a bucket are idle for most of an LDS tree (8, 4, 2, 1 of 16 active), and an addition costs the wave the same
    // whether one lane or all of them take part: the host picks few lanes per bucket (about a third of the mean number of
    // partials), so the strided sums — where every lane works — carry most of the additions and the tree is short.
    const uint32_t L = 1u << lane_bits, lane = t & (L - 1u);
    const uint32_t b = blockIdx.x * (256u >> lane_bits) + (t >> lane_bits);
    uint32_t s0 = 0, m = 0;
    if (b < nb) { s0 = sub_off[b]; m = sub_off[b + 1] - s0; }
    if (m > heavy_min && heavy[0] <= heavy_cap) m = 0;        // on k_scan's list: another workgroup of this launch folds it
                                                              // (a list that overflowed is ignored: every bucket is folded here)
    if (!__syncthreads_or(m >= 2 ? 1 : 0)) return;            // nothing to fold among this workgroup's buckets
    XYZZ<F> acc = XYZZ<F>::identity();
    if (lane < m) acc = load_xyzz<F>(partial, s0 + lane);
    for (uint32_t k = lane + L; k < m; k += L) { XYZZ<F> q = load_xyzz<F>(partial, s0 + k); add_full(acc, q); }
    sh[t] = acc;
    __syncthreads();
    for (uint32_t d = L >> 1; d > 0; d >>= 1) {
      if (lane < d && lane + d < m) { XYZZ<F> a = sh[t]; add_full(a, sh[t + d]); sh[t] = a; }
      __syncthreads();
    }
    if (lane == 0 && m >= 2) store_xyzz(partial, s0, sh[t]);
    return;
  }
  const uint32_t count = heavy[0] <= heavy_cap ? heavy[0] : 0u;
  if (blockIdx.x < nbn + COMBINE_HEAVY_BLOCKS) {
    for (uint32_t h = blockIdx.x - nbn; h < count; h += COMBINE_HEAVY_BLOCKS) {
      const uint32_t b = heavy[1 + h];
      const uint32_t s0 = sub_off[b], m = sub_off[b + 1] - s0;
      if (h < MSM_HEAVY_SPLIT && m >= MSM_HEAVY_SPLIT_MIN) continue;      // folded in two stages
      XYZZ<F> acc = XYZZ<F>::identity();
      for (uint32_t k = t; k < m; k += 256) { XYZZ<F> q = load_xyzz<F>(partial, s0 + k); add_full(acc, q); }
      __syncthreads();
      sh[t] = acc;
      __syncthreads();
      quad_tree256<F>(sh);
      if (t == 0) store_xyzz(partial, s0, sh[0]);
    }
    return;
  }
  const uint32_t nsplit = min(count, MSM_HEAVY_SPLIT);
  for (uint32_t it = blockIdx.x - nbn - COMBINE_HEAVY_BLOCKS; it < nsplit * MSM_HEAVY_PARTS; it += COMBINE_SPLIT_BLOCKS) {
    const uint32_t b = heavy[1 + it / MSM_HEAVY_PARTS], part = it % MSM_HEAVY_PARTS;
    const uint32_t s0 = sub_off[b], m = sub_off[b + 1] - s0;
    if (m < MSM_HEAVY_SPLIT_MIN) continue;
    const uint32_t per = (m + MSM_HEAVY_PARTS - 1) / MSM_HEAVY_PARTS, lo = part * per, hi = min(m, lo + per);
    XYZZ<F> acc = XYZZ<F>::identity();
    for (uint32_t k = lo + t; k < hi; k += 256) { XYZZ<F> q = load_xyzz<F>(partial, s0 + k); add_full(acc, q); }
    __syncthreads();
    sh[t] = acc;
    __syncthreads();
    quad_tree256<F>(sh);
    if (t == 0) store_xyzz(scratch, it, sh[0]);
    // Stage 2 by the LAST of the bucket's MSM_HEAVY_PARTS workgroups to get here (a ticket per bucket): a 32-leaf tree of four-lane additions over
    // the scratch row, into the bucket's slot.  (Until round 5 in k_reduce's prologue — which every workgroup that reads the bucket would now repeat.)
    const uint32_t hsl = it / MSM_HEAVY_PARTS;
    if (t == 0) { __threadfence(); s_last = atomicAdd(&heavy_done[hsl], 1u) == MSM_HEAVY_PARTS - 1 ? 1u : 0u; }
    __syncthreads();
    if (s_last) {
      __threadfence();
      if (t == 0) __hip_atomic_store(&heavy_done[hsl], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the ticket for the next MSM)
      sh[t] = t < MSM_HEAVY_PARTS ? load_xyzz<F>(scratch, (size_t)hsl * MSM_HEAVY_PARTS + t) : XYZZ<F>::identity();
      __syncthreads();
      for (uint32_t d = MSM_HEAVY_PARTS / 2; d > 0; d >>= 1) quad_level<F>(sh, d, [](uint32_t e) { return e; }, [d](uint32_t e) { return e + d; });
      if (t == 0) store_xyzz(partial, s0, sh[0]);
    }
    __syncthreads();
  }
}

// grid = K workgroups of T threads (T = min(256, nbw)); window sum = Σ_{idx} (idx+1)·B_idx
// plain_sums (optional): also Σ_idx B_idx of every window — with window tables and ONE bucket set shared by all windows (c = 13..16)
// the 2^(c-1) buckets are reduced as V = 2^(c-1)/1024 "virtual windows" of 1024 buckets each by this same kernel (V workgroups side by
// side, the depth of the c = 11 reduce), and the host finishes  Σ_v R_v + 1024·Σ_v v·S_v  (msm_finish).
template <class F>
__global__ void __launch_bounds__(256) k_reduce(const uint32_t* __restrict__ partial, const uint32_t* __restrict__ counts,
                                                const uint32_t* __restrict__ sub_off, uint32_t nbw,
                                                uint32_t* __restrict__ window_sums, uint32_t* __restrict__ plain_sums = nullptr) {
  __shared__ XYZZ<F> sh[256];
  const uint32_t w = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  const uint32_t ch = nbw / T;
  const uint32_t lo = t * ch;
  XYZZ<F> run = XYZZ<F>::identity(), sum = XYZZ<F>::identity();
  for (uint32_t j = ch; j-- > 0;) {
    uint32_t g = w * nbw + lo + j;
    if (counts[g]) { XYZZ<F> B = load_xyzz<F>(partial, sub_off[g]); add_full(run, B); }
    add_full(sum, run);
  }
  // sum = Σ (j+1)·B_{lo+j}; the chunk's offset contributes lo·run = ch·t·run_t.  Σ_t t·run_t is the sum over s >= 1 of the suffix
  // sums sfx_s = Σ_{t>=s} run_t, so one suffix scan over the threads' chunk totals replaces a per-thread double-and-add
  // (≈15 dependent operations for a 10-bit offset):  window sum = Σ_t (sum_t + ch·sfx_t·[t >= 1]).
  sh[t] = run;
  __syncthreads();
  for (uint32_t d = 1; d < T; d <<= 1) {
    const bool act = t + d < T;
    XYZZ<F> o = XYZZ<F>::identity();
    if (act) o = sh[t + d];
    __syncthreads();
    if (act) { XYZZ<F> a = sh[t]; add_full(a, o); sh[t] = a; }
    __syncthreads();
  }
  if (t >= 1) {
    XYZZ<F> sfx = sh[t];
    for (uint32_t k = 1; k < ch; k <<= 1) sfx = dbl(sfx);     // ch is a power of two
    add_full(sum, sfx);
  } else if (plain_sums) store_xyzz(plain_sums, w, sh[0]);    // (the suffix sum at 0 is the sum of all the window's buckets)
  __syncthreads();
  sh[t] = sum;
  __syncthreads();
  if (T == 256) quad_tree256<F>(sh);
  else for (uint32_t d = T >> 1; d > 0; d >>= 1) {
    if (t < d) { XYZZ<F> a = sh[t]; add_full(a, sh[t + d]); sh[t] = a; }
    __syncthreads();
  }
  if (t == 0) store_xyzz(window_sums, w, sh[0]);
}


// The shared bucket set of a table MSM reduced by BIT PLANES:  Σ_g (g + 1)·B_g = Σ_g B_g + Σ_p 2^p · Σ_{g: bit p of g set} B_g.  Every plane sum (and
// the plain sum, as two halves) is a TREE over its buckets — workgroup (plane, q) sums 256·lpt of them: lpt per thread, then an eight-level tree of
// four-lane additions — all (P + 2)·G workgroups side by side, and the last one to finish (a ticket in totals[3]) folds each plane's G partial sums.
// Depth: lpt + 8 + log2 G dependent additions, against the chunked running sums of k_reduce (2·nbw/256/V + 8 + 3 + 8); more additions in total
// (every bucket is read by half the planes), but this tail is latency, not throughput.  Output: P + 2 sums (planes 0..P-1, plain low half, plain
// high half); msm_finish does the Horner over the planes.
template <class F>
__global__ void __launch_bounds__(256) k_reduce_planes(const uint32_t* __restrict__ partial, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ sub_off,
                                                       uint32_t nbw, uint32_t P /* log2 nbw */, uint32_t G /* workgroups per plane */, uint32_t lpt,
                                                       uint32_t* stage /* (P + 2)·G partial sums: written by all, read by the last */, uint32_t* __restrict__ ticket,
                                                       uint32_t* __restrict__ out /* P + 2 sums */) {
  __shared__ XYZZ<F> sh[256];
  __shared__ uint32_t s_last;
  const uint32_t t = threadIdx.x, plane = blockIdx.x / G, q = blockIdx.x % G;
  XYZZ<F> acc = XYZZ<F>::identity();
  for (uint32_t i = 0; i < lpt; i++) {
    const uint32_t j = q * (256u * lpt) + i * 256u + t;      // leaf of this plane: 0 .. nbw/2
    uint32_t g;
    if (plane < P) g = ((j >> plane) << (plane + 1)) | (1u << plane) | (j & ((1u << plane) - 1u));      // the j-th bucket index with bit `plane` set
    else g = (plane - P) * (nbw >> 1) + j;                                                            // plain sum, low / high half
    if (counts[g]) { XYZZ<F> B = load_xyzz<F>(partial, sub_off[g]); add_full(acc, B); }
  }
  sh[t] = acc;
  __syncthreads();
  quad_tree256<F>(sh);
  if (t == 0) { store_xyzz(stage, blockIdx.x, sh[0]); __threadfence(); s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u; }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (t == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (for the next MSM)
  const uint32_t total = (P + 2) * G;      // <= 256
  sh[t] = t < total ? load_xyzz<F>(stage, t) : XYZZ<F>::identity();
  __syncthreads();
  for (uint32_t d = G >> 1; d > 0; d >>= 1) {
    const uint32_t pairs = (P + 2) * d;
    for (uint32_t base = 0; base < pairs; base += 64)
      quad_level<F>(sh, min(64u, pairs - base), [=](uint32_t e) { return ((base + e) / d) * G + (base + e) % d; }, [=](uint32_t e) { return ((base + e) / d) * G + (base + e) % d + d; });
  }
  if (t < P + 2) store_xyzz(out, t, sh[t * G]);
}

// ---- fused small MSM -------------------------------------------------------------------------------------------------
// Below ~16 k points the general pipeline is pure latency: ten launches, a sort through global memory, a per-window reduce
// that is 23-31 dependent additions deep.  Nova's augmented circuits need four such MSMs per step (7.6 k points each), on the
// critical path.  Here one launch does everything: workgroup (w, q) owns window w (c = 7: 37 windows of 64 signed buckets)
// and point chunk q (<= 1536 points): digits -> counting sort in LDS -> one thread per sub-bucket of <= 8 entries ->
// segmented tree over a bucket's sub-buckets -> sum_b (b+1)·B_b as the sum of the 64 suffix sums (scan + tree, 12 deep) ->
// the last workgroup of a window to finish adds the chunk results.  Depth: 8 + log2(max sub-buckets) + 12 + log2(chunks).
// (SMALL_C, SMALL_CHUNK, SMALL_MAXQ, MSM_SMALL_MAX: msm_api.hpp)
// 256 threads, one wave per SIMD.  (512 threads on half-length sub-buckets — two waves per SIMD, 210 VGPRs, no spills — measured
// SLOWER: 0.251 vs 0.187 ms at 7.7 k points, 0.167 vs 0.145 ms for one chunk: the accumulation phase does not get shorter with a second
// wave per SIMD, and the segment tree grows.)
constexpr uint32_t SMALL_THREADS = 256;
constexpr uint32_t SMALL_NBW = 1u << (SMALL_C - 1), SMALL_SUB = 8 /* longest sub-bucket */, SMALL_PER_THREAD = SMALL_CHUNK / SMALL_THREADS;
static_assert(SMALL_CHUNK / SMALL_SUB + SMALL_NBW <= SMALL_THREADS, "one thread per sub-bucket");
static_assert(SMALL_NBW == 64, "k_msm_small scans the buckets of a window with one wave");
// (1536 points per workgroup: at most 1536/8 + 64 = 256 sub-buckets, one per thread; 7.6 k points -> 37 x 5 = 185 workgroups,
//  fewer than the 256 CUs, so no two workgroups' single-wave scan phases share a SIMD)

// Signed digit w of a scalar WITHOUT walking the carries of the windows below it (a workgroup of the fused kernels owns one window):
// the recoding's digits lie in [-(half-1), half], so the carry into window w is 1 exactly when the low c·w bits exceed what w digits
// can hold without one,  T_w = Σ_{j<w} half·2^(cj)  (bit c-1 of every lower window) — one masked 256-bit comparison.
// (The walk cost ~1 µs per window below w: the top windows of k_msm_small started their accumulation 40 µs after window 0.)
template <int C>
struct DigitThreshold {
  uint32_t w[8];
  constexpr DigitThreshold() : w{} { for (int p = C - 1; p < 256; p += C) w[p >> 5] |= 1u << (p & 31); }
};
template <int C>
__device__ __forceinline__ int signed_digit(const uint32_t* s, int w) {
  constexpr DigitThreshold<C> T{};
  const int lo = C * w;
  uint32_t br = 0;                                       // borrow of T_w − (s mod 2^lo)
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int nb = lo - 32 * i;
    const uint32_t m = nb >= 32 ? 0xffffffffu : nb <= 0 ? 0u : (1u << nb) - 1u;
    const uint64_t d = (uint64_t)(T.w[i] & m) - (uint64_t)(s[i] & m) - br;
    br = (uint32_t)(d >> 63);
  }
  const uint32_t d = window_bits(s, lo, C) + br;
  return d > (1u << (C - 1)) ? (int)d - (1 << C) : (int)d;
}

template <class S, class F, int LEAN /* the option small_lean: its code is only in the instantiations that run it */>
__global__ void __launch_bounds__(SMALL_THREADS) k_msm_small(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ scalars, uint32_t n, int mont,
                                                   uint32_t Q, uint32_t chunk, uint32_t* __restrict__ chunk_out /* K*Q points */,
                                                   uint32_t* __restrict__ done /* K counters, zero between launches; nullptr: k_msm_small_sum follows */,
                                                   uint32_t* __restrict__ window_sums,
                                                   const uint32_t* __restrict__ tables /* or nullptr: row w holds 2^(7w)·P_i, row length tstride */, uint32_t tstride) {
  constexpr int lean = LEAN;      // (levels with a wave's worth of additions by one lane each instead of four: half the instructions, 3.4 µs more per level)
  __shared__ XYZZ<F> sh[SMALL_THREADS];
  __shared__ uint32_t cnt[SMALL_NBW], off[SMALL_NBW + 1], soff[SMALL_NBW + 1], cur[SMALL_NBW];
  __shared__ uint16_t list[SMALL_CHUNK];
  __shared__ uint8_t subb[SMALL_THREADS];
  __shared__ uint32_t s_maxm, s_ticket, wcnt[SMALL_THREADS / 64];
  // These waves sit on the critical path of a folding step while bulk kernels (the large MSM's accumulation, the batch
  // producer) fill the same SIMDs: raise their issue priority over the resident bulk waves.
  VZ_SET_CRIT_PRIO();
  const uint32_t t = threadIdx.x, w = blockIdx.x, q = blockIdx.y;
  const uint32_t lo = q * chunk, hi = min(n, lo + chunk);
  const uint32_t* __restrict__ wbases = tables ? tables + (size_t)AFFINE_WORDS * ((size_t)w * tstride) : bases;
  if (t < SMALL_NBW) { cnt[t] = 0; cur[t] = 0; }
  __syncthreads();
  int dig[SMALL_PER_THREAD];
#pragma unroll
  for (int k = 0; k < (int)SMALL_PER_THREAD; k++) {
    dig[k] = 0;
    const uint32_t i = lo + t + SMALL_THREADS * k;
    if (i < hi) {
      uint32_t sc[8];
      if (load_scalar<S>(scalars, i, mont, 0, sc)) {
        dig[k] = signed_digit<SMALL_C>(sc, (int)w);
        if (dig[k]) atomicAdd(&cnt[(dig[k] < 0 ? -dig[k] : dig[k]) - 1], 1u);
      }
    }
  }
  __syncthreads();
  if (t < 64) {
    // Wave 0, one lane per bucket: the sub-bucket length is the SHORTEST (4..8) whose sub-buckets still get a thread each — the
    // accumulation below is that many dependent mixed additions on every wave of the workgroup (≈4.7 µs each) — and a bucket's
    // entries are spread evenly over its sub-buckets.  (8 always fits: SMALL_CHUNK/8 + 64 = 256.)
    const uint32_t c0 = cnt[t];
    uint32_t m4 = (c0 + 3) / 4, m5 = (c0 + 4) / 5, m6 = (c0 + 5) / 6, m7 = (c0 + 6) / 7;
    for (int o = 32; o > 0; o >>= 1) { m4 += __shfl_xor(m4, o); m5 += __shfl_xor(m5, o); m6 += __shfl_xor(m6, o); m7 += __shfl_xor(m7, o); }
    const uint32_t sub = m4 <= SMALL_THREADS ? 4u : m5 <= SMALL_THREADS ? 5u : m6 <= SMALL_THREADS ? 6u : m7 <= SMALL_THREADS ? 7u : 8u;
    const uint32_t m = (c0 + sub - 1) / sub;
    uint32_t ic = c0, im = m, mx = m;                       // inclusive scans over the 64 lanes
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t uc = __shfl_up(ic, o), um = __shfl_up(im, o);
      if ((int)t >= o) { ic += uc; im += um; }
    }
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, o));
    off[t] = ic - c0; soff[t] = im - m;
    if (t == 63) { off[SMALL_NBW] = ic; soff[SMALL_NBW] = im; s_maxm = mx; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < (int)SMALL_PER_THREAD; k++) {
    if (dig[k]) {
      const uint32_t b = (uint32_t)(dig[k] < 0 ? -dig[k] : dig[k]) - 1;
      const uint32_t pos = atomicAdd(&cur[b], 1u);
      list[off[b] + pos] = (uint16_t)((t + SMALL_THREADS * k) | (dig[k] < 0 ? 0x8000u : 0u));
    }
  }
  if (t < SMALL_NBW) for (uint32_t sb = soff[t]; sb < soff[t + 1]; sb++) subb[sb] = (uint8_t)t;
  __syncthreads();
  const uint32_t nsubs = soff[SMALL_NBW];
  XYZZ<F> acc = XYZZ<F>::identity();
  uint32_t kk = 0, mb = 0;
  if (t < nsubs) {
    const uint32_t b = subb[t];
    kk = t - soff[b]; mb = soff[b + 1] - soff[b];
    const uint32_t cb = off[b + 1] - off[b];
    const uint32_t beg = off[b] + kk * cb / mb, end = off[b] + (kk + 1) * cb / mb;
    // (one wave per SIMD: nothing else hides the gather of an 80-byte table entry, so entry e+1 is in flight while e is added)
    uint32_t ent = beg < end ? list[beg] : 0u;
    Affine<F> pt = load_affine<F>(wbases, lo + (ent & 0x7fffu));
    for (uint32_t e = beg; e < end; e++) {
      const uint32_t ent_n = e + 1 < end ? list[e + 1] : ent;
      const Affine<F> pt_n = load_affine<F>(wbases, lo + (ent_n & 0x7fffu));
      if ((ent & 0x8000u) && !aff_is_identity(pt)) pt.y = F::neg(pt.y);
      add_mixed(acc, pt);
      ent = ent_n; pt = pt_n;
    }
  }
  sh[t] = acc;
  __syncthreads();
  // A bucket's sub-buckets are contiguous: stride-doubling tree inside each segment.  A level's scattered pairs are compacted
  // (ballot + popcount) into `list`, free by now, and added four lanes per addition (ec_mem.hpp: quad_level), 64 pairs a round.
  for (uint32_t d = 1; d < s_maxm; d <<= 1) {
    const bool act = t < nsubs && (kk & (2 * d - 1)) == 0 && kk + d < mb;
    const uint64_t bal = __ballot(act);
    if ((t & 63) == 0) wcnt[t >> 6] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t wv = 0; wv < SMALL_THREADS / 64; wv++) { before += wv < (t >> 6) ? wcnt[wv] : 0u; total += wcnt[wv]; }
    if (act) list[before + (uint32_t)__popcll(bal & ((1ull << (t & 63)) - 1ull))] = (uint16_t)t;
    __syncthreads();
    if (lean && total >= 48) {
      // one lane per addition, the pairs packed into the first waves: a wave's 64 additions cost what 16 cost four lanes each
      const uint32_t wave0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t & ~63u));
      XYZZ<F> a; uint32_t ia = 0; bool mine = false;
      if (wave0 < total) {
        mine = t < total;
        if (mine) { ia = list[t]; a = sh[ia]; add_full(a, sh[ia + d]); }
      }
      __syncthreads();
      if (mine) sh[ia] = a;
      __syncthreads();
    } else
    for (uint32_t base = 0; base < total; base += 64)
      quad_level<F>(sh, min(64u, total - base), [&](uint32_t e) { return (uint32_t)list[base + e]; }, [&](uint32_t e) { return (uint32_t)list[base + e] + d; });
  }
  XYZZ<F> bk = XYZZ<F>::identity();
  if (t < SMALL_NBW && cnt[t]) bk = sh[soff[t]];
  __syncthreads();
  if (Q > 1 && done) {
    // The chunks of a window share its buckets: publish this chunk's 64 bucket sums; the last workgroup of the window to arrive
    // adds them bucket by bucket and alone runs the weighted bucket reduction below (it used to run in every chunk's workgroup,
    // twelve dependent additions on one wave each, followed by a tree over the chunk results: same depth, a fifth of the
    // instructions at five chunks).  Release: stores, fence, agent-scope atomic ticket — all by the publishing wave; acquire: fence
    // (invalidates this CU's vector cache and the L2's non-local lines), then plain 16-byte loads.  (Per-word agent-scope atomic
    // loads here were issued one at a time: 72 serial round trips to another XCD's data, 40 µs.)
    // (the publishing wave alone runs the release fence: it writes this XCD's L2 back, and with every wave of the launch's 222
    //  workgroups doing so at the same moment the write-backs queued for ~25 µs)
    if (t < SMALL_NBW) {
      store_xyzz(chunk_out, ((size_t)w * Q + q) * SMALL_NBW + t, bk);
      __threadfence();
      if (t == 0) s_ticket = atomicAdd(&done[w], 1u);
    }
    __syncthreads();
    if (s_ticket != Q - 1) return;
    __threadfence();

    // sh = four slots of 64 buckets; slot 0 accumulates, the chunks arrive four (then three) at a time and every addition
    // is a four-lane one: (0 += 1, 2 += 3), 0 += 2
    const uint32_t b = t & (SMALL_NBW - 1), sl = t / SMALL_NBW;
    for (uint32_t loaded = 0; loaded < Q;) {
      const uint32_t slot0 = loaded ? 1u : 0u, k = min(Q - loaded, 4u - slot0), m = slot0 + k;
      if (sl >= slot0 && sl < m) {
        sh[t] = load_xyzz<F>(chunk_out, ((size_t)w * Q + loaded + sl - slot0) * SMALL_NBW + b);
      }
      __syncthreads();
      if (lean) {
        // slots (0 += 1) and (2 += 3) side by side on waves 0 and 2, one lane per bucket; then 0 += 2 on wave 0
        if ((sl == 0 && m >= 2) || (sl == 2 && m == 4)) { XYZZ<F> a = sh[t]; add_full(a, sh[t + SMALL_NBW]); sh[t] = a; }      // (a wave reads and writes its own slot pair only)
        __syncthreads();
        if (sl == 0 && m >= 3) { XYZZ<F> a = sh[t]; add_full(a, sh[t + 2 * SMALL_NBW]); sh[t] = a; }
        __syncthreads();
      } else {
      if (m >= 2) quad_level<F>(sh, SMALL_NBW, [](uint32_t e) { return e; }, [](uint32_t e) { return e + SMALL_NBW; });
      if (m == 4) quad_level<F>(sh, SMALL_NBW, [](uint32_t e) { return e + 2 * SMALL_NBW; }, [](uint32_t e) { return e + 3 * SMALL_NBW; });
      if (m >= 3) quad_level<F>(sh, SMALL_NBW, [](uint32_t e) { return e; }, [](uint32_t e) { return e + 2 * SMALL_NBW; });
      }
      loaded += k;
    }
    if (t == 0) __hip_atomic_store(&done[w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (t < SMALL_NBW) sh[t] = bk;
    __syncthreads();
  }
  // The rest is twelve levels of at most 64 dependent additions: four lanes per addition (ec_mem.hpp: quad_level), all four waves busy.
  if (lean >= 2) {      // (experiment: the six scan levels by one lane per bucket on wave 0: 19 k fewer instructions per window, 20 µs more latency)
    for (uint32_t d = 1; d < SMALL_NBW; d <<= 1) {
      XYZZ<F> a; const bool mine = t < SMALL_NBW - d;
      if (t < 64) { if (mine) { a = sh[t]; add_full(a, sh[t + d]); } }
      __syncthreads();
      if (mine) sh[t] = a;
      __syncthreads();
    }
  } else
  for (uint32_t d = 1; d < SMALL_NBW; d <<= 1)      // inclusive suffix sums of the buckets
    quad_level<F>(sh, SMALL_NBW - d, [](uint32_t e) { return e; }, [d](uint32_t e) { return e + d; });
  for (uint32_t d = SMALL_NBW / 2; d > 0; d >>= 1)  // sum_b (b+1)·B_b = sum of the suffix sums
    quad_level<F>(sh, d, [](uint32_t e) { return e; }, [d](uint32_t e) { return e + d; });
  if (Q == 1 || done) { if (t == 0) store_xyzz(window_sums, w, sh[0]); return; }
  if (t == 0) store_xyzz(chunk_out, (size_t)w * Q + q, sh[0]);          // k_msm_small_sum follows (VIMZ_DEBUG_SMALL_SUM_KERNEL)
}

// window sum = sum of the window's chunk results as a kernel of its own (VIMZ_DEBUG_SMALL_SUM_KERNEL=1): the fallback for the
// in-kernel "last workgroup sums" tail of k_msm_small, whose cross-workgroup hand-off relies on fences and agent-scope atomics
// across the part's eight L2s.
template <class F>
__global__ void __launch_bounds__(64) k_msm_small_sum(const uint32_t* __restrict__ chunk_out, uint32_t Q, uint32_t* __restrict__ window_sums) {
  __shared__ XYZZ<F> sh[SMALL_MAXQ];
  const uint32_t t = threadIdx.x, w = blockIdx.x;
  if (t < SMALL_MAXQ) sh[t] = t < Q ? load_xyzz<F>(chunk_out, (size_t)w * Q + t) : XYZZ<F>::identity();
  __syncthreads();
  for (uint32_t d = SMALL_MAXQ / 2; d > 0; d >>= 1) {
    if (t < d) { XYZZ<F> a = sh[t]; add_full(a, sh[t + d]); sh[t] = a; }
    __syncthreads();
  }
  if (t == 0) store_xyzz(window_sums, w, sh[0]);
}

// ---- fixed-base small MSM: precomputed multiples, no buckets ---------------------------------------------------------------------------
// The four small MSMs of an IVC step run over FIXED slices of the keys, and an MI355X has 288 GB: with every multiple
// m·2^(7w)·P_i (m = 1..64) of a slice resident (1.46 GB per 7.7 k points) a signed digit selects its point and the MSM is the plain sum
// of n·37 points — no digit sort, no bucket accumulation chains, no weighted bucket reduction (twelve dependent additions).  Depth:
// three mixed additions per thread (four points each), the 256-leaf tree of a workgroup (nine four-lane rounds), the tree over a
// window's ≤ 32 workgroup sums by the last workgroup of the window to arrive; the host adds the 37 window sums as before.
constexpr uint32_t FIXED_PER_THREAD = 4, FIXED_CHUNK = 256 * FIXED_PER_THREAD, FIXED_MAXQ = 64;
static_assert(MSM_SMALL_MAX <= (size_t)FIXED_CHUNK * FIXED_MAXQ, "a window's workgroup sums fit one tree");

template <class F>
__global__ void __launch_bounds__(256) k_build_multiples(const uint32_t* __restrict__ tables, size_t n, int K, uint32_t nm, uint32_t* __restrict__ mult) {
  const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;       // (w, i)
  if (g >= n * (size_t)K) return;
  const Affine<F> B = load_affine<F>(tables, (uint32_t)g);
  XYZZ<F> acc = from_affine(B);
  uint32_t* dst = mult + (size_t)AFFINE_WORDS * (g * nm);
  for (uint32_t m = 0; m < nm; m++) {
    const Affine<F> a = to_affine(acc);
    store_affine_at<F>(dst + (size_t)AFFINE_WORDS * m, a);
    add_mixed(acc, B);
  }
}
template <class C>
hipError_t build_multiples(hipStream_t stream, const uint32_t* d_tables, size_t n, int c, int K, uint32_t* d_mult) {
  typedef typename C::Coord F;
  if (!n) return hipSuccess;
  if (n * (size_t)K >= (1u << 31)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_build_multiples<F>, dim3((unsigned)((n * (size_t)K + 255) / 256)), dim3(256), 0, stream, d_tables, n, K, 1u << (c - 1), d_mult);
  return hipGetLastError();
}

// grid (Q, K): workgroup (q, w) sums the points selected by window w's digits of scalars [q·FIXED_CHUNK, (q+1)·FIXED_CHUNK).
template <class S, class F>
__global__ void __launch_bounds__(256) k_msm_fixed(const uint32_t* __restrict__ mult, uint32_t tstride, const uint32_t* __restrict__ scalars, uint32_t n, int mont,
                                                   uint32_t Q, uint32_t* __restrict__ partial /* K x Q */, uint32_t* __restrict__ done /* K counters */,
                                                   uint32_t* __restrict__ window_sums) {
  __shared__ XYZZ<F> sh[256];
  __shared__ uint32_t s_ticket;
  VZ_SET_CRIT_PRIO();
  const uint32_t t = threadIdx.x, q = blockIdx.x, w = blockIdx.y;
  constexpr uint32_t NM = SMALL_NBW;
  // the four table entries of this thread are requested before the first of them is needed
  Affine<F> pt[FIXED_PER_THREAD];
  bool neg[FIXED_PER_THREAD], have[FIXED_PER_THREAD];
#pragma unroll
  for (int k = 0; k < (int)FIXED_PER_THREAD; k++) {
    const uint32_t i = q * FIXED_CHUNK + t + 256u * k;
    have[k] = false; neg[k] = false;
    if (i < n) {
      uint32_t sc[8];
      if (load_scalar<S>(scalars, i, mont, 0, sc)) {
        const int d = signed_digit<SMALL_C>(sc, (int)w);
        if (d) {
          have[k] = true; neg[k] = d < 0;
          const uint32_t m = (uint32_t)(d < 0 ? -d : d) - 1u;
          const uint32_t* src = mult + (size_t)AFFINE_WORDS * (((size_t)w * tstride + i) * NM + m);
          pt[k] = load_affine_at<F>(src);
        }
      }
    }
  }
  XYZZ<F> acc = XYZZ<F>::identity();
#pragma unroll
  for (int k = 0; k < (int)FIXED_PER_THREAD; k++) {
    if (have[k]) {
      if (neg[k] && !aff_is_identity(pt[k])) pt[k].y = F::neg(pt[k].y);
      add_mixed(acc, pt[k]);
    }
  }
  sh[t] = acc;
  __syncthreads();
  quad_level<F>(sh, 64, [](uint32_t e) { return e; }, [](uint32_t e) { return e + 128; });
  quad_level<F>(sh, 64, [](uint32_t e) { return e + 64; }, [](uint32_t e) { return e + 192; });
  for (uint32_t d = 64; d > 0; d >>= 1) quad_level<F>(sh, d, [](uint32_t e) { return e; }, [d](uint32_t e) { return e + d; });
  if (Q == 1) { if (t == 0) store_xyzz(window_sums, w, sh[0]); return; }
  // publish; the last workgroup of the window to arrive sums the Q workgroup sums (release: store, fence, agent-scope ticket — by
  // the one publishing wave; acquire: fence, then plain loads — as k_msm_small's chunk merge)
  if (t < 64) {
    if (t == 0) store_xyzz(partial, (size_t)w * Q + q, sh[0]);
    __threadfence();
    if (t == 0) s_ticket = atomicAdd(&done[w], 1u);
  }
  __syncthreads();
  if (s_ticket != Q - 1) return;
  __threadfence();
  if (t < FIXED_MAXQ) {
    XYZZ<F> v = XYZZ<F>::identity();
    if (t < Q) {
      v = load_xyzz<F>(partial, (size_t)w * Q + t);
    }
    sh[t] = v;
  }
  __syncthreads();
  uint32_t top = 1; while (top < Q) top <<= 1;
  for (uint32_t d = top >> 1; d > 0; d >>= 1) quad_level<F>(sh, d, [](uint32_t e) { return e; }, [d](uint32_t e) { return e + d; });
  if (t == 0) { store_xyzz(window_sums, w, sh[0]); __hip_atomic_store(&done[w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

// ---- host driver ---------------------------------------------------------------------------------


// ---- window tables: one bucket set for all windows ---------------------------------------------------------------------
// With T_j[i] = 2^(c·j)·P_i precomputed (the commitment key is fixed for the whole proof and HBM is plentiful), digit j of
// scalar i is an entry (T_j[i], bucket |d|): every window shares ONE set of 2^(c-1) buckets, so c can be large (fewer
// digits per scalar => fewer sort entries and fewer additions), the bucket reduction runs once, and no Horner is left.
//   level 1: workgroup w owns buckets [256w, 256w+256): R_w = Σ_t (t+1)·B, S_w = Σ_t B      (double-and-add + LDS trees)
//   level 2: total = Σ_w R_w + 256·Σ_w w·S_w
template <class F>
__device__ __forceinline__ XYZZ<F> small_mul(const XYZZ<F>& p, uint32_t k) {
  XYZZ<F> acc = XYZZ<F>::identity();
  if (!k) return acc;
  for (int bit = 31 - __clz(k); bit >= 0; bit--) { acc = dbl(acc); if ((k >> bit) & 1) add_full(acc, p); }
  return acc;
}
// Table construction: tables[j][i] = 2^(c j) * P_i in affine internal form (row 0 = the key itself).
template <class F>
__global__ void __launch_bounds__(256) k_build_tables(const uint32_t* __restrict__ bases, size_t n, int c, int K, uint32_t* __restrict__ tables) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> p = load_affine<F>(bases, (uint32_t)i);
  store_affine<F>(tables, i, p);
  for (int j = 1; j < K; j++) {
    XYZZ<F> acc = from_affine(p);
    for (int k = 0; k < c; k++) acc = dbl(acc);
    p = to_affine(acc);
    store_affine<F>(tables, (size_t)j * n + i, p);
  }
}
template <class C>
hipError_t build_tables(hipStream_t stream, const uint32_t* d_bases, size_t n, int c, int K, uint32_t* d_tables) {
  typedef typename C::Coord F;
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_build_tables<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_bases, n, c, K, d_tables);
  return hipGetLastError();
}

// Enqueue the whole pipeline on `stream` and the copy of the K window sums into `pinned_dst` (host-pinned,
// K * XYZZ_WORDS words).  Does not synchronise: the caller waits on the stream (or an event recorded after this call)
// and then calls msm_finish.  `ws` must not be used by another stream concurrently.
template <class C>
hipError_t msm_launch(hipStream_t stream, MsmWorkspace& ws, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n,
                      int scalars_mont, int c_override, void* pinned_dst, MsmPlan* plan_out, hipEvent_t* ev, int split_ones,
                      const BaseTables* tb) {
  typedef typename C::Coord F;
  typedef typename C::Scalar S;
  if (n == 0 || n >= (1u << 31)) return hipErrorInvalidValue;
  // The window sums go straight into the caller's pinned buffer (host memory the device can write): the copy that used to follow —
  // a launch and a dependent hop between the last kernel and the host's wake-up — is gone.
  constexpr bool direct = true;
  // tables made for the fused small path (window SMALL_C): its window sums then only need adding — no Horner on the host
  static const bool no_small = getenv("VIMZ_DEBUG_NO_SMALL_MSM") != nullptr;
  const bool small_fmt = tb && tb->d && tb->c == SMALL_C;             // (ignored, not an error, when the fused path is switched off)
  const bool small_tb = small_fmt && !no_small && c_override <= 0;
  bool tabled = tb && tb->d && !small_fmt && c_override <= 0;
  // tables with per-window bucket sets only fit the window this size would get anyway; otherwise they are not used
  if (tabled && tb->own && tb->c != msm_plan(n, S::Params::BITS, 0).c) tabled = false;
  const bool own = tabled && tb->own;
  if (small_tb && (n > MSM_SMALL_MAX || tb->K != (S::Params::BITS + SMALL_C) / SMALL_C)) return hipErrorInvalidValue;
  if (!no_small && !tabled && c_override <= 0 && n <= MSM_SMALL_MAX) {      // fused single-launch path
    MsmPlan ps; ps.c = SMALL_C; ps.K = (S::Params::BITS + SMALL_C) / SMALL_C; ps.nbw = SMALL_NBW; ps.nb = SMALL_NBW * (uint32_t)ps.K; ps.split_ones = 0; ps.tabled = small_tb ? 2 : 0;
    *plan_out = ps;
    VZ_HIP_CHECK(ws.reserve_small());
    const uint32_t Q = (uint32_t)((n + SMALL_CHUNK - 1) / SMALL_CHUNK), chunk = (uint32_t)((n + Q - 1) / Q);
    uint32_t* done = reinterpret_cast<uint32_t*>(ws.small_buf);
    uint32_t* chunk_out = done + 128;
    if (small_tb && tb->mult) {      // every multiple resident: the digits select their points (k_msm_fixed)
      const uint32_t Qf = (uint32_t)((n + FIXED_CHUNK - 1) / FIXED_CHUNK);
      if (ev) for (int i = 0; i < 4; i++) VZ_HIP_CHECK(hipEventRecord(ev[i], stream));
      hipLaunchKernelGGL((k_msm_fixed<S, F>), dim3(Qf, ps.K), dim3(256), 0, stream, tb->mult + (size_t)AFFINE_WORDS * SMALL_NBW * tb->offset, (uint32_t)tb->n_total, d_scalars, (uint32_t)n,
                         scalars_mont, Qf, chunk_out, done, direct ? reinterpret_cast<uint32_t*>(pinned_dst) : reinterpret_cast<uint32_t*>(ws.window_sums));
      if (ev) for (int i = 4; i < 7; i++) VZ_HIP_CHECK(hipEventRecord(ev[i], stream));
      VZ_HIP_CHECK(hipGetLastError());
      if (!direct) VZ_HIP_CHECK(hipMemcpyAsync(pinned_dst, ws.window_sums, 4 * (size_t)XYZZ_WORDS * ps.K, hipMemcpyDeviceToHost, stream));
      return hipSuccess;
    }
    if (ev) for (int i = 0; i < 4; i++) VZ_HIP_CHECK(hipEventRecord(ev[i], stream));
    static const bool sum_kernel = getenv("VIMZ_DEBUG_SMALL_SUM_KERNEL") != nullptr;
#define VZ_SMALL(LEAN) hipLaunchKernelGGL((k_msm_small<S, F, LEAN>), dim3(ps.K, Q), dim3(SMALL_THREADS), 0, stream, d_bases, d_scalars, (uint32_t)n, scalars_mont, Q, chunk, chunk_out, \
                       sum_kernel ? (uint32_t*)nullptr : done, direct ? reinterpret_cast<uint32_t*>(pinned_dst) : reinterpret_cast<uint32_t*>(ws.window_sums), \
                       small_tb ? tb->d + (size_t)AFFINE_WORDS * tb->offset : (const uint32_t*)nullptr, small_tb ? (uint32_t)tb->n_total : 0u)
    switch (msm_tuning().small_lean) { case 0: VZ_SMALL(0); break; case 1: VZ_SMALL(1); break; default: VZ_SMALL(2); break; }
#undef VZ_SMALL
    if (Q > 1 && sum_kernel) hipLaunchKernelGGL(k_msm_small_sum<F>, dim3(ps.K), dim3(64), 0, stream, chunk_out, Q, direct ? reinterpret_cast<uint32_t*>(pinned_dst) : reinterpret_cast<uint32_t*>(ws.window_sums));
    if (ev) for (int i = 4; i < 7; i++) VZ_HIP_CHECK(hipEventRecord(ev[i], stream));
    VZ_HIP_CHECK(hipGetLastError());
    if (!direct) VZ_HIP_CHECK(hipMemcpyAsync(pinned_dst, ws.window_sums, 4 * (size_t)XYZZ_WORDS * ps.K, hipMemcpyDeviceToHost, stream));
    return hipSuccess;
  }
  MsmPlan pl = msm_plan(n, S::Params::BITS, tabled ? tb->c : c_override);
  if (tabled) {            // one bucket set shared by all windows — or (own) the usual ones, whose sums then need no Horner
    if (tb->K != pl.K || pl.nbw < 256 || (size_t)tb->K * tb->n_total >= (1u << 31)) return hipErrorInvalidValue;
    if (!own && 2 * (pl.nbw / std::min<uint32_t>(pl.nbw, MSM_VWIN)) + 1 > (uint32_t)MSM_MAX_WINDOWS) return hipErrorInvalidValue;
    if (own) pl.tabled = 3; else { pl.nb = pl.nbw; pl.tabled = 1; }
    d_bases = tb->d + (size_t)AFFINE_WORDS * tb->offset;
  }
  if (pl.K + 1 > MSM_MAX_WINDOWS || pl.c > 16 || pl.c < 2) return hipErrorInvalidValue;
  pl.split_ones = split_ones;
  // shared bucket set: as virtual windows (k_reduce), or — VIMZ_TUNE=reduce_planes=1 — by bit planes (k_reduce_planes) where the plane workgroups' partial sums fit one tree
  const bool no_planes = !msm_tuning().reduce_planes;
  uint32_t Pl = 0; while ((1u << Pl) < pl.nbw) Pl++;
  uint32_t Gp = std::min<uint32_t>(16u, pl.nbw / 1024u);
  while (Gp > 1 && (Pl + 2) * Gp > 256) Gp >>= 1;
  const bool planes = tabled && !own && !no_planes && pl.nbw >= 1024 && (1u << Pl) == pl.nbw && (Pl + 2) * Gp <= 256 && (int)Pl + 2 + 1 <= MSM_MAX_WINDOWS;
  if (planes) pl.tabled = 4;
  // (ONE assignment, every field final: the plan object is often shared — the producer's issuer thread launches row r + k while the folding thread
  //  finishes row r with the same plan, msm_finish — and a launch must never be seen half-described)
  *plan_out = pl;
  const size_t entries = (size_t)pl.K * n;
  // small MSMs are latency-bound (one dependent addition ~ 6-10 us): shorter chains per thread, more threads
  // (a witness commitment — split_ones — of an HD-sized circuit is a few 10^5 entries spread thinly over the buckets: one thread per bucket and a
  //  chain of a dozen additions each on a quarter of the GPU; pieces of 8 give twice the threads half the chain: 1 045-1 061 -> 1 088-1 097 steps/s at
  //  contrast HD, one chain 808 -> 823; at 4K the buckets are three times as full and the long pieces stay (558 against 542).  VIMZ_TUNE=witness_sub=N pins it.)
  const int wsub = msm_tuning().witness_sub;
  // (longer pieces for the dense MSM(T) — 24 / 32 entries, half the partials for k_combine — measured within the noise at 256 rows and
  //  worse in the 20-row window and on one chain: 842 against 876, 786 against 812)
  const int dsub = tb && tb->sub_hint > 0 ? tb->sub_hint : msm_tuning().dense_sub;      // (a caller that knows its vector is sparse — ivc.hip's boolean-row form — asks for shorter pieces)
  const uint32_t sub = n < (1u << 15) ? 8u : split_ones ? (uint32_t)(wsub > 0 ? wsub : n < (1u << 19) ? 8 : MSM_SUB) : (uint32_t)(dsub >= 2 && dsub <= (int)MSM_SUB ? dsub : MSM_SUB);   // MSM_SUB for everything large
  const size_t max_subs = entries / sub + pl.nb + 1;
  VZ_HIP_CHECK(ws.reserve(pl.nb, entries, max_subs));
  const int TB = 256;
#define VZ_EV(i) do { if (ev) VZ_HIP_CHECK(hipEventRecord(ev[i], stream)); } while (0)
  VZ_EV(0);
  const unsigned gs = (unsigned)std::min<size_t>((n + TB - 1) / TB, 256 * 16);
  // all buckets' counters fit in one workgroup's LDS at the default window (24 x 1024 x 4 B = 96 KiB): contention-free sort
  const bool lds_sort = (size_t)pl.nb * 4 <= 144 * 1024;
  if (!lds_sort) {      // (the LDS sort writes every counter itself and needs no cursors; each fill is a launch of its own)
    VZ_HIP_CHECK(hipMemsetAsync(ws.counts, 0, 4 * (size_t)pl.nb, stream));
    VZ_HIP_CHECK(hipMemsetAsync(ws.cursor, 0, 4 * (size_t)pl.nb, stream));
    VZ_HIP_CHECK(hipMemsetAsync(ws.heavy, 0, 4, stream));
  }
  const int sort_blocks_env = msm_tuning().sort_blocks;
  // every sort workgroup zeroes, writes out and later re-reads all nb counters (96 KiB at the default window): with one workgroup
  // per CU at 305 k points each handled 1.2 k scalars for 24.5 k counters, and k_block_prefix walked 256 rows — fixed costs.
  // About 4 k scalars per workgroup (75 workgroups here) measured best: one proof 384 -> 394 steps/s, three 599 -> 614.
  const uint32_t sort_blocks = sort_blocks_env > 0 && sort_blocks_env <= (int)SORT_BLOCKS ? (uint32_t)sort_blocks_env
                                                                                           : (uint32_t)std::min<size_t>(SORT_BLOCKS, std::max<size_t>(32, n / 4096));
  const uint32_t bstride = tabled && !own ? 0u : pl.nbw, pstride = tabled ? (uint32_t)tb->n_total : 0u;
  if (lds_sort) {
    VZ_HIP_CHECK(ws.reserve_block_hist((size_t)SORT_BLOCKS * pl.nb));
    {   // the >64 KiB dynamic-LDS opt-in is per device and per kernel instantiation; contexts fold from several host threads
      static std::mutex attr_mu;
      static uint64_t attr_devices = 0;
      int dev = 0;
      VZ_HIP_CHECK(hipGetDevice(&dev));
      std::lock_guard<std::mutex> g(attr_mu);
      if (!((attr_devices >> (dev & 63)) & 1ull)) {
        VZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist_lds<S>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_lds<S>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_devices |= 1ull << (dev & 63);
      }
    }
    hipLaunchKernelGGL(k_hist_lds<S>, dim3(sort_blocks), dim3(SORT_THREADS), pl.nb * 4, stream, d_scalars, n, scalars_mont, split_ones, pl.c, pl.K, bstride, pl.nb, ws.block_hist);
  } else {
    hipLaunchKernelGGL(k_hist<S>, dim3(gs), dim3(TB), 0, stream, d_scalars, n, scalars_mont, split_ones, pl.c, pl.K, bstride, ws.counts);
  }
  VZ_EV(1);
  // lanes per ordinary bucket in k_combine, from the mean number of partials per bucket (upper bound: every digit non-zero): two
  // lanes up to ~24 partials (measured at 312 k dense points, 19 per bucket: 16 lanes 0.28 ms, 8: 0.155, 4: 0.137, 2: 0.117;
  // one lane and a heavy list of every bucket: 3.5 ms); buckets above 16 partials per lane go to the heavy list
  // (a witness — split_ones — has far fewer entries than its upper bound: an eighth is assumed, which gives its buckets two lanes where
  //  the bound gave four: k_combine 31.4 -> 29.1 M instructions per step)
  const size_t mean_parts = (split_ones ? entries / 8 : entries) / sub / pl.nb + 1;
  const int lane_bits_env = msm_tuning().combine_lane_bits;
  const uint32_t lane_bits = lane_bits_env >= 0 ? (uint32_t)lane_bits_env : mean_parts > 96 ? 4u : mean_parts > 48 ? 3u : mean_parts > 24 ? 2u : 1u;
  const uint32_t heavy_min = 16u << lane_bits;
  if (lds_sort)      // per-workgroup histograms -> prefixes and totals, and (last workgroup) the scan: one launch
    hipLaunchKernelGGL(k_prefix_scan<0>, dim3((pl.nb + 1023) / 1024), dim3(1024), 0, stream, ws.block_hist, pl.nb, ws.counts, sort_blocks, ws.heavy, ws.bucket_off, ws.sub_off,
                       ws.totals, sub, heavy_min, MsmWorkspace::HEAVY_CAP);
  else
    hipLaunchKernelGGL(k_scan<MSM_SUB>, dim3(1), dim3(1024), 0, stream, ws.counts, pl.nb, ws.bucket_off, ws.sub_off, ws.totals, sub, ws.heavy, heavy_min, MsmWorkspace::HEAVY_CAP);
  VZ_EV(2);
  if (lds_sort)
    hipLaunchKernelGGL(k_scatter_lds<S>, dim3(sort_blocks), dim3(SORT_THREADS), pl.nb * 4, stream, d_scalars, n, scalars_mont, split_ones, pl.c, pl.K, bstride, pl.nb,
                       pstride, ws.bucket_off, ws.block_hist, ws.sorted);
  else
    hipLaunchKernelGGL(k_scatter<S>, dim3(gs), dim3(TB), 0, stream, d_scalars, n, scalars_mont, split_ones, pl.c, pl.K, bstride, pstride,
                       ws.bucket_off, ws.cursor, ws.sorted);
  VZ_EV(3);
  uint32_t* partial = reinterpret_cast<uint32_t*>(ws.partial);
  const unsigned ga = (unsigned)((max_subs + TB - 1) / TB);
  // (accum_lds_kb, an experiment kept as an option: LDS the kernel never touches, so that only 160 / kb of its workgroups fit a CU whatever the number of
  //  launches in flight and the critical chain's kernels — 180-230 registers a lane — always find room.  Measured zero-sum at two workgroups per CU and 2 %
  //  worse at one, in the 20-row window, over 256 rows and on one chain: profiles/r05_segments_queues_sweep.txt)
  const size_t accum_lds = (size_t)msm_tuning().accum_lds_kb * 1024;
  if (accum_lds > 65536) { static const hipError_t once = hipFuncSetAttribute((const void*)k_accum<F>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)once; }
  hipLaunchKernelGGL(k_accum<F>, dim3(ga), dim3(TB), accum_lds, stream, d_bases, ws.sorted, ws.bucket_off, ws.sub_off, pl.nb,
                     ws.totals, partial, sub);
  VZ_EV(4);
  const unsigned per_wg = 256u >> lane_bits, nbn = (pl.nb + per_wg - 1) / per_wg;
  hipLaunchKernelGGL(k_combine<F>, dim3(nbn + COMBINE_HEAVY_BLOCKS + COMBINE_SPLIT_BLOCKS), dim3(256), 0, stream, partial, ws.sub_off, pl.nb, nbn,
                     (const uint32_t*)ws.heavy, MsmWorkspace::HEAVY_CAP, ws.heavy_scratch, lane_bits, heavy_min, ws.heavy_done);
  VZ_EV(5);      // (stage 2 of the very heavy buckets: the last of a bucket's workgroups, inside k_combine)
  const unsigned T = pl.nbw < 256 ? pl.nbw : 256;
  uint32_t* wsum = direct ? reinterpret_cast<uint32_t*>(pinned_dst) : reinterpret_cast<uint32_t*>(ws.window_sums);
  const uint32_t vw = std::min<uint32_t>(pl.nbw, MSM_VWIN);      // shared bucket set: virtual windows of vw buckets
  const int V = (int)(pl.nbw / vw);
  const int kout = planes ? (int)Pl + 2 : tabled && !own ? 2 * V : pl.K;     // sums produced: planes + two plain halves; (R_v, S_v) per virtual window; or one per window
  if (planes) {
    hipLaunchKernelGGL(k_reduce_planes<F>, dim3((Pl + 2) * Gp), dim3(256), 0, stream, (const uint32_t*)partial, (const uint32_t*)ws.counts, (const uint32_t*)ws.sub_off, pl.nbw, Pl, Gp,
                       pl.nbw / (512u * Gp), ws.plane_scratch, ws.totals + 3, wsum);
  } else if (tabled && !own) {
    hipLaunchKernelGGL(k_reduce<F>, dim3(V), dim3(vw < 256 ? vw : 256), 0, stream, (const uint32_t*)partial, (const uint32_t*)ws.counts, (const uint32_t*)ws.sub_off, vw, wsum, wsum + (size_t)XYZZ_WORDS * V);
  } else
    hipLaunchKernelGGL(k_reduce<F>, dim3(pl.K), dim3(T), 0, stream, (const uint32_t*)partial, (const uint32_t*)ws.counts, (const uint32_t*)ws.sub_off, pl.nbw, wsum);
  VZ_EV(6);
#undef VZ_EV
  if (split_ones) {   // sum of the bases with unit scalar -> window_sums[K]
    uint32_t* lvl0 = reinterpret_cast<uint32_t*>(ws.ones_partial);
    if (msm_tuning().ones_dense) hipLaunchKernelGGL((k_ones_dense<S, F>), dim3(ONES_THREADS / 256), dim3(256), 0, stream, d_scalars, d_bases, n, scalars_mont, lvl0);
    else hipLaunchKernelGGL((k_ones_partial<S, F>), dim3(ONES_THREADS / 256), dim3(256), 0, stream, d_scalars, d_bases, n, scalars_mont, lvl0);
    uint32_t* lvl1 = lvl0 + (size_t)XYZZ_WORDS * ONES_THREADS;
    hipLaunchKernelGGL(k_tree256<F>, dim3(ONES_THREADS / 256), dim3(256), 0, stream, lvl0, ONES_THREADS, lvl1);
    hipLaunchKernelGGL(k_tree256<F>, dim3(1), dim3(256), 0, stream, lvl1, ONES_THREADS / 256, wsum + (size_t)XYZZ_WORDS * kout);
  }
  VZ_HIP_CHECK(hipGetLastError());
  if (!direct) VZ_HIP_CHECK(hipMemcpyAsync(pinned_dst, wsum, 4 * (size_t)XYZZ_WORDS * (kout + (split_ones ? 1 : 0)), hipMemcpyDeviceToHost, stream));
  return hipSuccess;
}

// Host tail: Horner over the K window sums (converted to the standard form, whose host multiply is the fast 4x64 path)
// and one inversion.  Result affine, standard Montgomery form.
template <class C>
Affine<typename C::Base> msm_finish(const MsmPlan& pl, const void* pinned) {
  typedef typename C::Coord F;
  typedef typename C::Base FS;
  const uint32_t* hw = reinterpret_cast<const uint32_t*>(pinned);
  auto host_point = [&](int w) {
    XYZZ<FS> p; const uint32_t* d = hw + (size_t)XYZZ_WORDS * w;
    FS* f[4] = {&p.X, &p.Y, &p.ZZ, &p.ZZZ};
    for (int k = 0; k < 4; k++) { F t; for (int i = 0; i < 9; i++) t.v[i] = d[COORD_WORDS * k + i]; *f[k] = t.to_std(); }
    return p;
  };
  XYZZ<FS> acc = XYZZ<FS>::identity();
  if (pl.tabled == 4) {
    // bit planes of the shared bucket set (k_reduce_planes): Σ_p 2^p·S_p + plain low + plain high
    int P = 0; while ((1u << P) < pl.nbw) P++;
    for (int q = P - 1; q >= 0; q--) { acc = dbl(acc); add_full(acc, host_point(q)); }
    add_full(acc, host_point(P)); add_full(acc, host_point(P + 1));
    if (pl.split_ones) add_full(acc, host_point(P + 2));
    return to_affine(acc);
  }
  if (pl.tabled == 1) {
    // one bucket set shared by all windows, reduced as V virtual windows of MSM_VWIN buckets: bucket b = MSM_VWIN·v + idx weighs
    // (idx + 1) + MSM_VWIN·v, so the sum is  Σ_v R_v + MSM_VWIN·Σ_v v·S_v  with  Σ_v v·S_v = Σ_{s>=1} Σ_{v>=s} S_v
    const uint32_t vw = pl.nbw < MSM_VWIN ? pl.nbw : MSM_VWIN;
    const int V = (int)(pl.nbw / vw);
    XYZZ<FS> run = XYZZ<FS>::identity();
    for (int v = V - 1; v >= 1; v--) { add_full(run, host_point(V + v)); add_full(acc, run); }
    for (uint32_t k = 1; k < vw; k <<= 1) acc = dbl(acc);
    for (int v = 0; v < V; v++) add_full(acc, host_point(v));
    if (pl.split_ones) add_full(acc, host_point(2 * V));
    return to_affine(acc);
  }
  const int kout = pl.K;      // tabled == 2, 3: K sums of one bucket set each, already weighted
  for (int w = kout - 1; w >= 0; w--) {
    if (!pl.tabled) for (int k = 0; k < pl.c; k++) acc = dbl(acc);
    add_full(acc, host_point(w));
  }
  if (pl.split_ones) add_full(acc, host_point(kout));
  return to_affine(acc);
}

// Σ_{scalar_i = 1} P_i as a call of its own (the unit-scalar tail of msm_launch without a Pippenger around it): the sum S_1 of the key's points on the
// boolean rows whose fresh bit is one (ivc.hip: the boolean-row form of the cross term's commitment).  Result: one XYZZ point at pinned_dst.
template <class C>
hipError_t ones_launch(hipStream_t stream, MsmWorkspace& ws, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n, int scalars_mont, void* pinned_dst) {
  typedef typename C::Coord F;
  typedef typename C::Scalar S;
  if (!ws.ones_partial) VZ_HIP_CHECK(hipMalloc(&ws.ones_partial, 4 * (size_t)XYZZ_WORDS * (16384 + 64 + 512)));      // (only this: a general reserve() here would be freed and re-made — hipFree synchronises the device — by the first real MSM on this workspace)
  uint32_t* lvl0 = reinterpret_cast<uint32_t*>(ws.ones_partial);
  uint32_t* lvl1 = lvl0 + (size_t)XYZZ_WORDS * ONES_THREADS;
  uint32_t* top = lvl1 + (size_t)XYZZ_WORDS * (ONES_THREADS / 256);      // (inside the spare slots of ones_partial)
  hipLaunchKernelGGL((k_ones_dense<S, F>), dim3(ONES_THREADS / 256), dim3(256), 0, stream, d_scalars, d_bases, n, scalars_mont, lvl0);
  hipLaunchKernelGGL(k_tree256<F>, dim3(ONES_THREADS / 256), dim3(256), 0, stream, lvl0, ONES_THREADS, lvl1);
  hipLaunchKernelGGL(k_tree256<F>, dim3(1), dim3(256), 0, stream, lvl1, ONES_THREADS / 256, top);
  VZ_HIP_CHECK(hipGetLastError());
  return hipMemcpyAsync(pinned_dst, top, 4 * (size_t)XYZZ_WORDS, hipMemcpyDeviceToHost, stream);
}
template <class C>
XYZZ<typename C::Base> ones_finish(const void* pinned) {
  typedef typename C::Coord F;
  typedef typename C::Base FS;
  const uint32_t* d = reinterpret_cast<const uint32_t*>(pinned);
  XYZZ<FS> p; FS* f[4] = {&p.X, &p.Y, &p.ZZ, &p.ZZZ};
  for (int k = 0; k < 4; k++) { F t; for (int i = 0; i < 9; i++) t.v[i] = d[COORD_WORDS * k + i]; *f[k] = t.to_std(); }
  return p;
}

template <class C>
hipError_t msm_run(hipStream_t stream, MsmWorkspace& ws, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n,
                   int scalars_mont, int c_override, Affine<typename C::Base>* out_affine_mont, MsmStats* stats,
                   hipEvent_t* ev /* 7 events or nullptr */, int split_ones, const BaseTables* tb) {
  typedef typename C::Base FS;
  if (n == 0) { out_affine_mont->x = FS::zero(); out_affine_mont->y = FS::zero(); return hipSuccess; }
  if (!ws.host_pinned) VZ_HIP_CHECK(hipHostMalloc(&ws.host_pinned, 4 * XYZZ_WORDS * MSM_MAX_WINDOWS));
  MsmPlan pl;
  VZ_HIP_CHECK(msm_launch<C>(stream, ws, d_bases, d_scalars, n, scalars_mont, c_override, ws.host_pinned, &pl, ev, split_ones, tb));
  uint32_t h_tot[2] = {0, 0};
  if (stats) VZ_HIP_CHECK(hipMemcpyAsync(h_tot, ws.totals, 8, hipMemcpyDeviceToHost, stream));
  VZ_HIP_CHECK(hipStreamSynchronize(stream));
  *out_affine_mont = msm_finish<C>(pl, ws.host_pinned);
  if (stats && ev) for (int i = 0; i < 6; i++) VZ_HIP_CHECK(hipEventElapsedTime(&stats->ms[i], ev[i], ev[i + 1]));
  if (stats) { stats->c = pl.c; stats->K = pl.K; stats->subs = h_tot[0]; stats->entries = h_tot[1]; }
  return hipSuccess;
}

}  // namespace vz
