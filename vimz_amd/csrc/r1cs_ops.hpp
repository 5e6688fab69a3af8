// Relaxed-R1CS kernels of one Nova fold (SURVEY.md §8a rows V1, V2, F1', X1) over vectors resident in HBM.
//   k_spmv3        (A,B,C)·z for a fresh witness — replaces R1CSShape::multiply_vec (nova-snark 0.23.0).
//                  CSR with a coefficient dictionary: 8 B per non-zero (column, dictionary index); the few-10^3-entry
//                  dictionary and the hot part of z live in L2.  One thread per constraint row.
//   k_cross_term   T = Az1∘Bz2 + Az2∘Bz1 − u1·Cz2 − u2·Cz1          (commit_T's cross term)
//   k_fold5        W, E, Az, Bz, Cz  <-  x1 + r·x2  in one pass  (RelaxedR1CSWitness::fold + the running products,
//                  kept by linearity so the running instance never needs an SpMV again)
//   k_check_relaxed  counts rows with Az∘Bz != u·Cz + E             (is_sat_relaxed)
// All element-wise kernels are HBM streaming kernels: 32 B per element per operand, two 16-byte accesses per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "vecops.hpp"
#include "fp29.hpp"

namespace vz {

struct CsrDev { const uint32_t* row_ptr; const uint32_t* col; const uint32_t* coef; };

// The coefficient dictionary as the device holds it: entry i = [c·2^256 mod p | c·2^261 mod p], 8 words each (index 2i and 2i + 1 of an array of field
// elements).  The second form is the coefficient for the reduced-radix product below: Fp29::mul(c·2^261, z·2^256) = c·z·2^256, the vectors' own form.
template <class F>
inline std::vector<F> dict_for_device(const std::vector<F>& dict) {
  std::vector<F> out(2 * dict.size());
  for (size_t i = 0; i < dict.size(); i++) { F t = dict[i]; out[2 * i] = t; for (int k = 0; k < 5; k++) t = F::dbl(t); out[2 * i + 1] = t; }
  return out;
}
template <class F> using R29 = Fp29<typename F::Params>;
// x·2^256 (canonical, 8 words) -> x·2^261 in reduced radix: five modular doublings, then shifts and masks
template <class F> __device__ __forceinline__ R29<F> r29_of(const F& y) { F t = y; for (int k = 0; k < 5; k++) t = F::dbl(t); return R29<F>::pack(t.v); }
// a sum of lazily reduced terms back to 8 canonical words
template <class F> __device__ __forceinline__ F fe_of29(const R29<F>& a) { F r; a.weak_reduce().canon().unpack(r.v); return r; }

constexpr uint32_t SPMV_LONG = 6;    // (matrix,row) items with more terms than this go to the wave-per-item kernel: the short
                                     // kernel's duration is its longest serial row (≈1 µs per dependent gather + multiply)

// acc += c * v, skipping the multiply for the values a fresh witness is made of (0 and 1).  The product runs in the curve code's reduced-radix arithmetic
// (fp29.hpp: 162 multiply-adds and ~65 other instructions against the 8 x 32-bit CIOS's 585) and the row's sum stays lazily reduced — every term is below
// 1.5 p, a lane adds at most a few dozen — until fe_of29 (round 5; the vectors keep their 8-word Montgomery form in HBM).
template <class F>
__device__ __forceinline__ void spmv_term(R29<F>& acc, const uint32_t* __restrict__ dict, uint32_t coef, const uint32_t* __restrict__ z, uint32_t col) {
  const F v = load_fe<F>(z, col);
  const bool zr = v.is_zero(), on = v.eq(F::one());
  if (__ballot(!zr) == 0ull) return;                 // (wave-uniform tests: per-lane branches around a multiplication are flattened
  F t = load_fe<F>(dict, 2 * (size_t)coef);          //  into predicated code that every wave executes in full)
  // most coefficients are ±1 as well (differences, selections): c·v is then ±v
  const bool cp = t.eq(F::one()), cm = t.eq(F::neg(F::one()));
  if (!on && cp) t = v;
  if (!on && cm) t = F::neg(v);
  R29<F> term = R29<F>::pack(t.v);
  if (__ballot(!(zr || on || cp || cm)) != 0ull) {
    const R29<F> m = R29<F>::mul(R29<F>::pack(load_fe<F>(dict, 2 * (size_t)coef + 1).v), R29<F>::pack(v.v));
    if (!(on || cp || cm)) term = m;
  }
  if (!zr) acc = R29<F>::add(acc, term);
}
// butterfly step of a row's partial sums: lanes `off` apart exchange and add
template <class F>
__device__ __forceinline__ void spmv_exchange_add(R29<F>& acc, int off) {
  R29<F> o;
#pragma unroll
  for (int w = 0; w < 9; w++) o.v[w] = __shfl_xor(acc.v[w], off);
  acc = R29<F>::add(acc, o);
}

// x·y where y comes from a FRESH instance — a witness wire or a row of (A,B,C)·z of one step: nineteen in twenty of the step
// circuits' constraints are boolean ones (b·(b − 1) = 0), whose products are 0, 1 and −1, and the rows of a wave are the same
// constraint for neighbouring pixels — so most waves never reach the multiplication (≈ 40 instructions of tests against ≈ 380).
template <class F>
__device__ __forceinline__ F mul_fresh(const F& x, const F& y) {
  const bool z = y.is_zero(), o = y.eq(F::one()), m = y.eq(F::neg(F::one()));
  F res = F::zero();
  // a WAVE-uniform branch (the ballot is a scalar): a per-lane `if` around the multiplication is flattened into predicated code
  // that every wave executes in full — measured: 23.5 -> 20.3 M instructions per step with the per-lane form
  if (__ballot(!(z || o || m)) != 0ull) res = F::mul(x, y);
  if (o) res = x;
  if (m) res = F::neg(x);
  if (z) res = F::zero();
  return res;
}

// One thread per (matrix, row), matrix-major so that consecutive lanes read consecutive rows; rows longer than SPMV_LONG are
// left to k_spmv_long.  (One thread per row looping over A, B, C serialised three dependent gather chains.)
template <class F>
__global__ void __launch_bounds__(256) k_spmv3(CsrDev A, CsrDev B, CsrDev C, const uint32_t* __restrict__ dict, size_t nrows,
                                               const uint32_t* __restrict__ z, uint32_t* __restrict__ az, uint32_t* __restrict__ bz, uint32_t* __restrict__ cz) {
  VZ_GRID_STRIDE(i, 3 * nrows) {
    const uint32_t m = (uint32_t)(i / nrows);
    const size_t r = i - (size_t)m * nrows;
    const CsrDev M = m == 0 ? A : (m == 1 ? B : C);
    uint32_t* out = m == 0 ? az : (m == 1 ? bz : cz);
    const uint32_t lo = M.row_ptr[r], hi = M.row_ptr[r + 1];
    if (hi - lo > SPMV_LONG) continue;
    R29<F> acc = R29<F>::zero();
    for (uint32_t k = lo; k < hi; k++) spmv_term<F>(acc, dict, M.coef[k], z, M.col[k]);
    store_fe(out, r, fe_of29<F>(acc));
  }
}

// Long (matrix,row) items: packed (matrix << 30 | row).  They are the substituted-bit rows of Num2Bits(240) (240 terms), the
// Poseidon partial rounds (up to ~70 terms) and, nine in ten, rows of 9..32 terms (byte recompositions, full rounds); with one
// thread per row they stalled whole waves.  The first n_med items (<= SPMV_MED terms, list sorted by the host) get 16 lanes
// each, four to a wave; the rest one wave each.  Lanes stride over the terms, a shuffle tree adds the partial sums.
constexpr uint32_t SPMV_MED = 32;
template <class F>
__global__ void __launch_bounds__(256) k_spmv_long(CsrDev A, CsrDev B, CsrDev C, const uint32_t* __restrict__ dict, const uint32_t* __restrict__ items,
                                                   uint32_t n_items, uint32_t n_med, const uint32_t* __restrict__ z, uint32_t* __restrict__ az,
                                                   uint32_t* __restrict__ bz, uint32_t* __restrict__ cz) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t quads = (n_med + 3) >> 2, units = quads + (n_items - n_med);
  for (uint32_t u = wave; u < units; u += nwaves) {
    const bool quad = u < quads;
    const uint32_t it = quad ? 4 * u + (lane >> 4) : n_med + (u - quads);
    const uint32_t l = quad ? (lane & 15u) : lane, step = quad ? 16u : 64u;
    const bool live = !quad || it < n_med;
    const uint32_t packed = live ? items[it] : 0u, m = packed >> 30, r = packed & 0x3fffffffu;
    const CsrDev M = m == 0 ? A : (m == 1 ? B : C);
    uint32_t lo = 0, hi = 0;
    if (live) { lo = M.row_ptr[r]; hi = M.row_ptr[r + 1]; }
    R29<F> acc = R29<F>::zero();
    uint32_t added = 0;
    for (uint32_t k = lo + l; k < hi; k += step) {
      spmv_term<F>(acc, dict, M.coef[k], z, M.col[k]);
      if ((++added & 31u) == 0) acc = acc.weak_reduce();      // (a lane of a very long row: keep the lazy sum far below 2^261)
    }
    acc = acc.weak_reduce();                             // below 3 p; six doublings of the bound stay below 2^261 / p = 128
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      if (quad && off >= 16) continue;                 // (uniform per wave)
      spmv_exchange_add<F>(acc, off);
      if (off == 8) acc = acc.weak_reduce();
    }
    if (live && l == 0) store_fe(m == 0 ? az : (m == 1 ? bz : cz), r, fe_of29<F>(acc));
  }
}
// Host side of the list: items of at most SPMV_MED terms first; returns their number.
template <class Len>
inline uint32_t spmv_sort_items(std::vector<uint32_t>& items, Len term_count) {
  std::vector<uint32_t> med, lng;
  for (uint32_t it : items) (term_count(it) <= SPMV_MED ? med : lng).push_back(it);
  items = med; items.insert(items.end(), lng.begin(), lng.end());
  return (uint32_t)med.size();
}
inline unsigned spmv_long_blocks(uint32_t n_items, uint32_t n_med) {
  const uint32_t units = (n_med + 3) / 4 + (n_items - n_med);
  return (unsigned)std::min<uint32_t>((units + 3) / 4, 4096);
}

// (A,B,C)·z over `nrows` rows starting at `row0` and, in the same pass, the cross term of those rows: for the verifier
// circuits' 7.6 k rows (both curves) the three launches spmv3 -> spmv_long -> cross_term were three latency-bound hops on the
// critical path of every step.  16 lanes per row: they stride over the terms of A, B and C in turn, a 4-level shuffle butterfly
// adds the partial sums, lane 0 stores the products and T.  (T is skipped when az1 is null: step 0 folds into the zero instance.)
// (One wave per row with 16 lanes per matrix side by side measured slower: 0.81 vs 0.72 ms for the secondary half of a step.)
template <class F>
__global__ void __launch_bounds__(256) k_spmv_cross16(CsrDev A, CsrDev B, CsrDev C, const uint32_t* __restrict__ dict, uint32_t row0, uint32_t nrows,
                                                      const uint32_t* __restrict__ z, uint32_t* __restrict__ az, uint32_t* __restrict__ bz, uint32_t* __restrict__ cz,
                                                      const uint32_t* __restrict__ az1, const uint32_t* __restrict__ bz1, const uint32_t* __restrict__ cz1, F u1, F u2,
                                                      uint32_t* __restrict__ T) {
  VZ_SET_CRIT_PRIO();       // critical path of a step, next to bulk kernels on the same SIMDs
  const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l = threadIdx.x & 15u;
  const bool live = g < nrows;
  const uint32_t r = row0 + (live ? g : 0u);
  R29<F> acc[3];
#pragma unroll
  for (int m = 0; m < 3; m++) {
    const CsrDev M = m == 0 ? A : (m == 1 ? B : C);
    acc[m] = R29<F>::zero();
    if (live) {
      const uint32_t lo = M.row_ptr[r], hi = M.row_ptr[r + 1];
      // four terms of this lane at a time: their index loads, then their value / coefficient gathers, are in flight together (one
      // term per iteration made every term two dependent memory latencies: the 254-term rows of a bit decomposition cost 16 of them)
      for (uint32_t k0 = lo + l; k0 < hi; k0 += 64) {
        uint32_t col[4], cf[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint32_t kk = k0 + 16u * j; const bool on = kk < hi; col[j] = on ? M.col[kk] : 0xffffffffu; cf[j] = on ? M.coef[kk] : 0u; }
        F v[4], c[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { v[j] = col[j] != 0xffffffffu ? load_fe<F>(z, col[j]) : F::zero(); c[j] = load_fe<F>(dict, 2 * (size_t)cf[j]); }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const bool zr = v[j].is_zero(), on = v[j].eq(F::one());
          if (__ballot(!zr) == 0ull) continue;
          F t = c[j];
          const bool cp = t.eq(F::one()), cm = t.eq(F::neg(F::one()));      // (the verifier circuits' rows are wires with coefficient ±1 almost throughout)
          if (!on && cp) t = v[j];
          if (!on && cm) t = F::neg(v[j]);
          R29<F> term = R29<F>::pack(t.v);
          if (__ballot(!(zr || on || cp || cm)) != 0ull) {      // (spmv_term's product: the coefficient's second form, reduced radix)
            const R29<F> mm = R29<F>::mul(R29<F>::pack(load_fe<F>(dict, 2 * (size_t)cf[j] + 1).v), R29<F>::pack(v[j].v));
            if (!(on || cp || cm)) term = mm;
          }
          if (!zr) acc[m] = R29<F>::add(acc[m], term);
        }
        acc[m] = acc[m].weak_reduce();      // (four terms a round: a 254-term row's lane never holds more than 3 p + 4 · 1.5 p)
      }
    }
  }
#pragma unroll
  for (int m = 0; m < 3; m++) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) spmv_exchange_add<F>(acc[m], off);      // below 16 · 3 p
    acc[m] = acc[m].weak_reduce().canon();
  }
  if (!live || l != 0) return;
  F a2, b2, c2;
  acc[0].unpack(a2.v); acc[1].unpack(b2.v); acc[2].unpack(c2.v);
  store_fe(az, r, a2); store_fe(bz, r, b2); store_fe(cz, r, c2);
  if (!az1) return;
  // T = az1·bz2 + az2·bz1 − u1·cz2 − u2·cz1 with two reductions: the running instance's entries and the scalars moved to the 2^261 form (r29_of), the fresh
  // products as they stand (canonical, 2^256 form), so that every product comes out in the vectors' 2^256 form
  typedef R29<F> G;
  const G t1 = G::mul_add2(r29_of<F>(load_fe<F>(az1, r)), acc[1], acc[0], r29_of<F>(load_fe<F>(bz1, r)));       // < 1.5 p
  const F c1 = load_fe<F>(cz1, r);
  G t2;
  if (u2.eq(F::one())) t2 = G::add(G::mul(r29_of<F>(u1), acc[2]), G::pack(c1.v));                                // the fresh instance's u is one: < 2.5 p
  else t2 = G::mul_add2(r29_of<F>(u1), acc[2], r29_of<F>(u2), G::pack(c1.v));
  store_fe(T, r, fe_of29<F>(G::template sub<3>(t1, t2)));
}

// The BOOLEAN-ROW form of a cross term's commitment (ivc.hip).  On a row b·(b − 1) = 0 — rows [0, nb) of a step circuit, circuit/builder.hpp — the cross term
// of the running instance (a = AZ_1[i], u_1) and a fresh one (b = az_2[i] in {0, 1}) is  T_i = b ? a − u_1 : −a,  so with C_A = Σ_{i<nb} a_i·ck_i (kept by
// linearity) and S_1 = Σ_{b_i = 1} ck_i (a unit-scalar sum)
//     Σ_{i<nb} T_i·ck_i = 2·Σ_{b_i = 1} T_i·ck_i + u_1·S_1 − C_A:
// the dense MSM runs over the vector T' = (2·T_i where b_i = 1, 0 where b_i = 0; T_i on the other rows) — half the points — and the host adds u_1·S_1 − C_A.
// Exact for ANY fresh value: a row whose az is neither 0 nor 1 gets T_i + a_i (its −a_i is inside −C_A).
template <class F>
__device__ __forceinline__ F bool_row_masked(const F& t, const F& a, const F& az_fresh) {
  const bool z = az_fresh.is_zero(), o = az_fresh.eq(F::one());
  F r = F::zero();
  if (o) r = F::dbl(t);
  if (!z && !o) r = F::add(t, a);
  return r;
}
template <class F>
__global__ void __launch_bounds__(256) k_cross_term(size_t n, const uint32_t* __restrict__ az1, const uint32_t* __restrict__ bz1, const uint32_t* __restrict__ cz1, F u1,
                                                    const uint32_t* __restrict__ az2, const uint32_t* __restrict__ bz2, const uint32_t* __restrict__ cz2, F u2,
                                                    uint32_t* __restrict__ T, uint32_t* __restrict__ Tm = nullptr, uint32_t nb = 0) {
  VZ_GRID_STRIDE(i, n) {
    const F a1 = load_fe<F>(az1, i), a2 = load_fe<F>(az2, i);
    F t = mul_fresh(a1, load_fe<F>(bz2, i));            // (instance 2 is the fresh one wherever a fold calls this)
    t = F::add(t, mul_fresh(load_fe<F>(bz1, i), a2));
    t = F::sub(t, mul_fresh(u1, load_fe<F>(cz2, i)));
    t = F::sub(t, mul_fresh(load_fe<F>(cz1, i), u2));
    store_fe(T, i, t);
    if (Tm) store_fe(Tm, i, i < nb ? bool_row_masked(t, a1, a2) : t);
  }
}

// The step rows of a primary fold and of a later cross term in ONE pass, queued the moment a step's challenge r is known:
//   (AZ, BZ, CZ) += r·(az, bz, cz) of the row just folded;
//   E += r·T,  T = Tin − hasB·r_prev·negB the cross term of that row (see below; no error vector is folded at step 0);
//   out <- cross term of the folded products with a coming row's (az', bz', cz')      (out may be Tin: element-wise, read first).
// Lookahead (ivc.hip): the cross term of step i+2 is computed against the running instance of step i+1, one whole step early —
// T(U + r·u, u') = T(U, u') + r·T(u, u') — so the vector actually folded into E is Tin + r_prev·T(u_prev, u) = Tin − r_prev·negB.
template <class F>
__global__ void __launch_bounds__(256) k_fold_cross(size_t n, uint32_t* __restrict__ AZ, uint32_t* __restrict__ BZ, uint32_t* __restrict__ CZ, uint32_t* __restrict__ E,
                                                    const uint32_t* Tin, int fold_E, F r, int hasB, F r_prev, const uint32_t* __restrict__ negB, F u1_new,
                                                    const uint32_t* __restrict__ az, const uint32_t* __restrict__ bz, const uint32_t* __restrict__ cz,
                                                    uint32_t* out, const uint32_t* __restrict__ azn, const uint32_t* __restrict__ bzn, const uint32_t* __restrict__ czn, F u2,
                                                    uint32_t* __restrict__ outm = nullptr, uint32_t nb = 0) {
  VZ_GRID_STRIDE(i, n) {
    // (az, bz, cz and the coming row's are fresh products: mul_fresh; the cross term Tin is zero on every linear row)
    const F a = F::add(load_fe<F>(AZ, i), mul_fresh(r, load_fe<F>(az, i)));
    const F b = F::add(load_fe<F>(BZ, i), mul_fresh(r, load_fe<F>(bz, i)));
    const F c = F::add(load_fe<F>(CZ, i), mul_fresh(r, load_fe<F>(cz, i)));
    store_fe(AZ, i, a); store_fe(BZ, i, b); store_fe(CZ, i, c);
    if (fold_E) {
      F t = load_fe<F>(Tin, i);
      if (hasB) t = F::sub(t, mul_fresh(r_prev, load_fe<F>(negB, i)));
      if (__ballot(!t.is_zero()) != 0ull) { const F e = F::add(load_fe<F>(E, i), F::mul(r, t)); if (!t.is_zero()) store_fe(E, i, e); }
    }
    if (out) {
      const F an = load_fe<F>(azn, i);
      F t = mul_fresh(a, load_fe<F>(bzn, i));
      t = F::add(t, mul_fresh(b, an));
      t = F::sub(t, mul_fresh(u1_new, load_fe<F>(czn, i)));
      t = F::sub(t, mul_fresh(c, u2));
      store_fe(out, i, t);
      if (outm) store_fe(outm, i, i < nb ? bool_row_masked(t, a, an) : t);      // (the vector the MSM takes: boolean-row form, see k_cross_term)
    }
  }
}

// MINUS the cross term of two fresh instances (u = 1 both):  negB = c0 + c1 − a0·b1 − a1·b0.  For satisfied rows (c = a·b) this is
// (a1 − a0)·(b1 − b0): between two consecutive image rows 80 % zeros, 15 % ones and the 13 k Poseidon rows (measured on the sample
// image) — a vector whose commitment costs a twentieth of a dense one (the unit scalars are summed directly, MSM `split_ones`).
template <class F>
__global__ void __launch_bounds__(256) k_fresh_cross_neg(size_t n, const uint32_t* __restrict__ az0, const uint32_t* __restrict__ bz0, const uint32_t* __restrict__ cz0,
                                                         const uint32_t* __restrict__ az1, const uint32_t* __restrict__ bz1, const uint32_t* __restrict__ cz1,
                                                         uint32_t* __restrict__ out) {
  VZ_GRID_STRIDE(i, n) {
    F t = F::add(load_fe<F>(cz0, i), load_fe<F>(cz1, i));
    t = F::sub(t, F::mul(load_fe<F>(az0, i), load_fe<F>(bz1, i)));
    t = F::sub(t, F::mul(load_fe<F>(az1, i), load_fe<F>(bz0, i)));
    store_fe(out, i, t);
  }
}

// x1[i] += r * x2[i]
template <class F>
__global__ void __launch_bounds__(256) k_axpy_inplace(size_t n, uint32_t* __restrict__ x1, F r, const uint32_t* __restrict__ x2) {
  VZ_GRID_STRIDE(i, n) store_fe(x1, i, F::add(load_fe<F>(x1, i), F::mul(r, load_fe<F>(x2, i))));
}

// Upload from PINNED host memory by a kernel that reads the host buffer over the link itself.  hipMemcpyAsync(pinned -> device) is
// asynchronous on paper; with several contexts of one process folding at once the call itself was seen to block for 6-7 ms now and
// then (the copy path's own queue), which in a 20-row window decides between 560 and 750 steps/s.  A kernel launch never takes that
// path.  bytes must be a multiple of 16; the host buffer must stay unchanged until the stream has passed this point (as for a copy).
static __global__ void __launch_bounds__(256) k_copy16(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
  VZ_GRID_STRIDE(i, n16) dst[i] = src[i];
}
static __global__ void __launch_bounds__(256) k_copy4(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t n4) {
  VZ_GRID_STRIDE(i, n4) dst[i] = src[i];
}
// either direction: one side is device memory, the other pinned host memory (bytes a multiple of 4)
static inline hipError_t copy_pinned(hipStream_t s, void* dst, const void* src, size_t bytes) {
  if (!bytes) return hipSuccess;
  if ((bytes & 15) || (((uintptr_t)dst | (uintptr_t)src) & 15)) {
    if (bytes & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_copy4, dim3((unsigned)std::min<size_t>(1024, (bytes / 4 + 255) / 256)), dim3(256), 0, s, (uint32_t*)dst, (const uint32_t*)src, bytes / 4);
    return hipGetLastError();
  }
  const size_t n16 = bytes / 16;
  hipLaunchKernelGGL(k_copy16, dim3((unsigned)std::min<size_t>(1024, (n16 + 255) / 256)), dim3(256), 0, s, (uint4*)dst, (const uint4*)src, n16);
  return hipGetLastError();
}
static inline hipError_t upload_pinned(hipStream_t s, void* dst, const void* src_pinned, size_t bytes) { return copy_pinned(s, dst, src_pinned, bytes); }

struct Fold5 { uint32_t* x1[5]; const uint32_t* x2[5]; size_t n[5]; };
template <class F>
__global__ void __launch_bounds__(256) k_fold5(Fold5 a, F r) {
#pragma unroll 1
  for (int v = 0; v < 5; v++) {
    uint32_t* x1 = a.x1[v]; const uint32_t* x2 = a.x2[v];
    if (!x1) continue;
    VZ_GRID_STRIDE(i, a.n[v]) {       // (x2 is a fresh witness or fresh products wherever a step calls this; a merge's operands are dense)
      const F y = load_fe<F>(x2, i);
      if (!y.is_zero()) store_fe(x1, i, F::add(load_fe<F>(x1, i), mul_fresh(r, y)));
    }
  }
}

template <class F>
__global__ void __launch_bounds__(256) k_check_relaxed(size_t n, const uint32_t* __restrict__ az, const uint32_t* __restrict__ bz, const uint32_t* __restrict__ cz,
                                                       F u, const uint32_t* __restrict__ E, uint32_t* __restrict__ bad /* [0]=count, [1]=first row+1 */) {
  VZ_GRID_STRIDE(i, n) {
    F lhs = F::mul(load_fe<F>(az, i), load_fe<F>(bz, i));
    F rhs = F::mul(u, load_fe<F>(cz, i));
    if (E) rhs = F::add(rhs, load_fe<F>(E, i));
    if (!lhs.eq(rhs)) { atomicAdd(&bad[0], 1u); atomicMin(&bad[1], (uint32_t)i); }
  }
}

// counts elements where a != b (raw limbs)
static __global__ void __launch_bounds__(256) k_count_diff(size_t n, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ bad) {
  VZ_GRID_STRIDE(i, n) {
    uint32_t d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) d |= a[8 * i + k] ^ b[8 * i + k];
    if (d) atomicAdd(&bad[0], 1u);
  }
}

// counts elements whose limbs are not below the modulus (data imported from outside)
template <class F>
__global__ void __launch_bounds__(256) k_count_unreduced(size_t n, const uint32_t* __restrict__ a, uint32_t* __restrict__ bad) {
  VZ_GRID_STRIDE(i, n) {
    F x = load_fe<F>(a, i);
    if (!x.is_reduced()) atomicAdd(&bad[0], 1u);
  }
}

}  // namespace vz
