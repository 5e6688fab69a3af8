// Pippenger multi-scalar multiplication for gfx950 — replaces `cpu_best_multiexp` / `pasta_msm`
// behind `CommitmentEngine::commit` in nova-snark 0.23.0 (SURVEY.md §8a rows M1/M2, §8b "MSM" seam).
//
// Pipeline (all on one HIP stream, no host round trip until the K window sums come back):
//   1. k_hist      signed-digit recode of every scalar (window c bits, digits in [-2^(c-1), 2^(c-1)]),
//                  histogram of (window, |digit|) buckets with global atomics.
//   2. (scan)      exclusive scan of bucket sizes -> entry offsets, and of ceil(size/SUB) -> sub-bucket offsets (k_scan; on the
//                  LDS-sort path the last workgroup of k_prefix_scan).  Large buckets (witness scalars are ~95 % bits
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
#include <hip/hip_runtime.h>
#include <vector>
#include <algorithm>
#include "ec.hpp"

namespace vz {

#define VZ_HIP_CHECK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return _e; } while (0)

constexpr int MSM_SUB = 16;          // max entries one thread accumulates in k_accum: the chain of dependent additions per thread;
                                     // the partials of a bucket are then folded by k_combine.  8 / 12 / 16 / 24 give the same steps/s within the run-to-run noise
                                     // (shorter pieces: k_accum wastes fewer lanes at its end, k_combine has more partials to fold —
                                     // in the bench 0.27 + 0.31, 0.29 + 0.25, 0.32 + 0.25, 0.53 + 0.19 ms)
constexpr int MSM_MAX_WINDOWS = 96;
constexpr uint32_t MSM_VWIN = 1024;  // window tables with one shared bucket set: its 2^(c-1) buckets are reduced as virtual windows of this many

struct MsmPlan {
  int c;            // window bits
  int K;            // windows
  uint32_t nbw;     // buckets per window = 2^(c-1)
  uint32_t nb;      // total buckets
  int split_ones;   // unit scalars summed separately (window_sums[K])
  int tabled;       // 1: window tables, one bucket set, (R_v, S_v) of nbw / MSM_VWIN virtual windows, no Horner; 2: tables of the fused small path (K sums, no Horner);
                    // 3: tables with per-window bucket sets; 4: one bucket set reduced by BIT PLANES (k_reduce_planes: log2(nbw) + 2 sums)
};

// The fused single-launch path for small MSMs (k_msm_small): window, points per workgroup chunk, chunks, size limit.
constexpr int SMALL_C = 7;
constexpr uint32_t SMALL_CHUNK = 1536, SMALL_MAXQ = 32;
// (measured on MI355X, dense scalars, tools/small_msm_crossover.py: fused 0.22 / 0.29 / 0.31 / 0.35 / 0.45 ms at 16 k / 24.6 k / 27.7 k / 32.8 k / 49 k
//  points against 0.28 / 0.34 / 0.35 / 0.34 / 0.42 ms through the general pipeline: the hand-over is at 20 chunks)
constexpr size_t MSM_SMALL_MAX = (size_t)SMALL_CHUNK * 20;
static_assert(MSM_SMALL_MAX <= (size_t)SMALL_CHUNK * SMALL_MAXQ, "chunk results of a window fit the workspace");

// Precomputed window tables of a commitment key: d[j][i] = 2^(c j) * P_i, affine internal form, row length n_total.
// own != 0: every window keeps its own bucket set, as without tables — the tables only spare the host the Horner over the window sums
// (the large-MSM default window, c = 11); own == 0: one bucket set shared by all windows (c = 13..16: fewer entries, deeper reduce).
struct BaseTables {
  const uint32_t* d; size_t n_total; size_t offset; int c, K; int own = 0;
  // fused small path only: every multiple of the table rows — mult[(w·n_total + i)·2^(c−1) + (m−1)] = m·2^(c·w)·P_i, m = 1..2^(c−1) —
  // so that a digit SELECTS its point: no buckets, no sort, the MSM is one sum (k_msm_fixed).  189 KB per base point at c = 7.
  const uint32_t* mult = nullptr;
  int sub_hint = 0;      // large path: entries per accumulation thread (0: the default, 16); a caller whose vector is mostly zeros asks for shorter pieces
};

static inline MsmPlan msm_plan(size_t n, int scalar_bits, int c_override) {
  MsmPlan p;
  p.split_ones = 0; p.tabled = 0;
  int c = c_override;
  if (c <= 0) {
    // Measured on MI355X (profiles/r01_msm_phases.txt): k_accum is throughput-bound (~n*K mixed adds) while
    // k_reduce is a latency-bound serial chain whose depth grows with 2^c / 256, so the optimum sits at a
    // much smaller window than the classic ln(n) rule: c = 11 from 2^15 points up, shrinking below.
    c = n >= (1u << 15) ? 11 : n >= (1u << 12) ? 9 : n >= 256 ? 7 : 5;
  }
  p.c = c;
  p.K = (scalar_bits + 1 + c - 1) / c;
  p.nbw = 1u << (c - 1);
  p.nb = p.nbw * (uint32_t)p.K;
  return p;
}

struct MsmWorkspace {  // device buffers, grown on demand and reused across calls
  uint32_t* counts = nullptr;      // nb
  uint32_t* cursor = nullptr;      // nb
  uint32_t* bucket_off = nullptr;  // nb + 1
  uint32_t* sub_off = nullptr;     // nb + 1
  uint32_t* sorted = nullptr;      // K * n
  void* partial = nullptr;         // max_subs * sizeof(XYZZ)
  void* window_sums = nullptr;     // K * sizeof(XYZZ)
  uint32_t* totals = nullptr;      // [0] = total subs
  uint32_t* block_hist = nullptr;  // [256][nb] per-workgroup histograms of the LDS counting sort
  size_t cap_block_hist = 0;
  void* ones_partial = nullptr;    // (16384 + 64) XYZZ partial sums of the unit-scalar path
  uint32_t* heavy = nullptr;       // [0] = count, then ids of buckets with many sub-buckets
  uint32_t* heavy_scratch = nullptr;   // 1024 x 32 partial sums of the split heavy buckets
  uint32_t* heavy_done = nullptr;      // 1024 tickets: the last of a split bucket's 32 workgroups folds its scratch row (k_combine); zero between launches
  uint32_t* plane_scratch = nullptr;   // 256 partial sums of k_reduce_planes
  static constexpr uint32_t HEAVY_CAP = 65536;
  size_t cap_nb = 0, cap_entries = 0, cap_subs = 0;
  void* host_pinned = nullptr;     // MSM_MAX_WINDOWS * 128 B

  hipError_t reserve(uint32_t nb, size_t entries, size_t subs) {
    if (nb > cap_nb) {
      hipFree(counts); hipFree(cursor); hipFree(bucket_off); hipFree(sub_off);
      VZ_HIP_CHECK(hipMalloc(&counts, 4 * (size_t)nb)); VZ_HIP_CHECK(hipMalloc(&cursor, 4 * (size_t)nb));
      VZ_HIP_CHECK(hipMalloc(&bucket_off, 4 * ((size_t)nb + 1))); VZ_HIP_CHECK(hipMalloc(&sub_off, 4 * ((size_t)nb + 1)));
      cap_nb = nb;
    }
    if (entries > cap_entries) {
      hipFree(sorted); VZ_HIP_CHECK(hipMalloc(&sorted, 4 * entries)); cap_entries = entries;
    }
    if (subs > cap_subs) {
      hipFree(partial);
      VZ_HIP_CHECK(hipMalloc(&partial, 4 * XYZZ_WORDS * subs)); cap_subs = subs;
    }
    if (!window_sums) VZ_HIP_CHECK(hipMalloc(&window_sums, 4 * XYZZ_WORDS * MSM_MAX_WINDOWS));
    // (totals[2] is the ticket of k_prefix_scan: zero between launches.  hipMemset on device memory is a null-stream operation that may
    //  return before it has run and the MSM streams are non-blocking: synchronise)
    if (!totals) { VZ_HIP_CHECK(hipMalloc(&totals, 64)); VZ_HIP_CHECK(hipMemset(totals, 0, 64)); VZ_HIP_CHECK(hipStreamSynchronize(nullptr)); }
    if (!heavy) VZ_HIP_CHECK(hipMalloc(&heavy, 4 * (HEAVY_CAP + 1)));
    if (!heavy_scratch) VZ_HIP_CHECK(hipMalloc(&heavy_scratch, 4 * (size_t)XYZZ_WORDS * 1024 * 32));
    if (!heavy_done) { VZ_HIP_CHECK(hipMalloc(&heavy_done, 4 * 1024)); VZ_HIP_CHECK(hipMemset(heavy_done, 0, 4 * 1024)); VZ_HIP_CHECK(hipStreamSynchronize(nullptr)); }
    if (!plane_scratch) VZ_HIP_CHECK(hipMalloc(&plane_scratch, 4 * (size_t)XYZZ_WORDS * 256));
    if (!ones_partial) VZ_HIP_CHECK(hipMalloc(&ones_partial, 4 * (size_t)XYZZ_WORDS * (16384 + 64 + 512)));
    if (!host_pinned) VZ_HIP_CHECK(hipHostMalloc(&host_pinned, 4 * XYZZ_WORDS * MSM_MAX_WINDOWS));
    return hipSuccess;
  }
  void* small_buf = nullptr;       // fused small MSM: 128 completion counters, then MSM_MAX_WINDOWS x 16 chunks x 64 bucket sums
  hipError_t reserve_small() {
    if (!window_sums) VZ_HIP_CHECK(hipMalloc(&window_sums, 4 * XYZZ_WORDS * MSM_MAX_WINDOWS));
    if (!host_pinned) VZ_HIP_CHECK(hipHostMalloc(&host_pinned, 4 * XYZZ_WORDS * MSM_MAX_WINDOWS));
    // NOTE: hipMemset on device memory runs on the null stream and may return before it has executed; the MSM streams are
    // non-blocking, so every such fill is followed by a null-stream synchronise (a fill landing after the first kernel's
    // writes cost a day: it zeroed chunk results of the very first small MSM of a prover, only under heavy multi-stream load).
    if (!totals) { VZ_HIP_CHECK(hipMalloc(&totals, 64)); VZ_HIP_CHECK(hipMemset(totals, 0, 64)); VZ_HIP_CHECK(hipStreamSynchronize(nullptr)); }
    if (small_buf) return hipSuccess;
    const size_t bytes = 512 + 4 * (size_t)XYZZ_WORDS * MSM_MAX_WINDOWS * SMALL_MAXQ * (1u << (SMALL_C - 1));
    VZ_HIP_CHECK(hipMalloc(&small_buf, bytes));
    VZ_HIP_CHECK(hipMemset(small_buf, 0, 512));          // the completion counters (the chunk results are written before they are read)
    return hipStreamSynchronize(nullptr);
  }
  hipError_t reserve_block_hist(size_t words) {
    if (words <= cap_block_hist) return hipSuccess;
    hipFree(block_hist); block_hist = nullptr; cap_block_hist = 0;
    VZ_HIP_CHECK(hipMalloc(&block_hist, 4 * words));
    cap_block_hist = words;
    return hipSuccess;
  }
  void release() {
    hipFree(counts); hipFree(cursor); hipFree(bucket_off); hipFree(sub_off); hipFree(sorted);
    hipFree(partial); hipFree(window_sums); hipFree(totals); hipFree(heavy); hipFree(heavy_scratch); hipFree(heavy_done); hipFree(plane_scratch); hipFree(small_buf); hipFree(ones_partial); hipFree(block_hist);
    if (host_pinned) hipHostFree(host_pinned);
    *this = MsmWorkspace();
  }
};

struct MsmStats { int c, K; uint32_t subs, entries; float ms[6]; };  // ms: hist, scan, scatter, accum, combine, reduce

// Defined in msm.hpp; explicitly instantiated per curve in msm_inst_*.hip.
template <class C>
hipError_t msm_launch(hipStream_t stream, MsmWorkspace& ws, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n,
                      int scalars_mont, int c_override, void* pinned_dst, MsmPlan* plan_out, hipEvent_t* ev, int split_ones,
                      const BaseTables* tb = nullptr);
template <class C>
hipError_t build_tables(hipStream_t stream, const uint32_t* d_bases, size_t n, int c, int K, uint32_t* d_tables);
// d_mult[(w·n + i)·2^(c−1) + (m−1)] = m·d_tables[w][i] for m = 1..2^(c−1)  (n·K·2^(c−1) affine points)
template <class C>
hipError_t build_multiples(hipStream_t stream, const uint32_t* d_tables, size_t n, int c, int K, uint32_t* d_mult);
template <class C>
Affine<typename C::Base> msm_finish(const MsmPlan& pl, const void* pinned);
template <class C>
hipError_t msm_run(hipStream_t stream, MsmWorkspace& ws, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n,
                   int scalars_mont, int c_override, Affine<typename C::Base>* out_affine_mont, MsmStats* stats,
                   hipEvent_t* ev /* 7 events or nullptr */, int split_ones, const BaseTables* tb = nullptr);
// Σ_{scalar_i = 1} P_i alone: one XYZZ point (device form) at pinned_dst; ones_finish converts it
template <class C>
hipError_t ones_launch(hipStream_t stream, MsmWorkspace& ws, const uint32_t* d_bases, const uint32_t* d_scalars, size_t n, int scalars_mont, void* pinned_dst);
template <class C>
XYZZ<typename C::Base> ones_finish(const void* pinned);

}  // namespace vz
