// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).
//
// Relaxed-R1CS algebra of one Nova fold, restating what nova-snark 0.23.0 performs inside
// `RecursiveSNARK::prove_step` (reached from vimz/src/nova_snark_backend/folding.rs:35-41
// through nova_scotia::create_recursive_circuit; crate not vendored, vimz/Cargo.lock:3577-3606;
// algebra as in SURVEY.md Appendix F / the Nova paper §4):
//   R1CSShape::multiply_vec      (A z, B z, C z)          -> spmv()
//   commit_T cross term          T = Az1∘Bz2 + Az2∘Bz1 − u1·Cz2 − u2·Cz1   -> cross_term()
//   RelaxedR1CSWitness::fold     W = W1 + r·W2, E = E1 + r·T               -> axpy()
//   is_sat_relaxed               Az∘Bz == u·Cz + E                         -> is_sat_relaxed()
// Vector layout used across this repo: z = [u | X (public IO) | W (aux)] in Circom wire
// order (wire 0 is the constant one / u).
#pragma once
#include <vector>
#include <thread>
#include "field.hpp"

namespace orc {

template <class F>
static void spmv(size_t nrows, const uint32_t* row_ptr, const uint32_t* col, const u64* val_canon,
                 const F* z, F* out, int threads = 1) {
  auto work = [&](size_t lo, size_t hi) {
    for (size_t r = lo; r < hi; r++) {
      F acc = F::zero();
      for (uint32_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) acc = acc + F::from_canonical(val_canon + 4 * (size_t)k) * z[col[k]];
      out[r] = acc;
    }
  };
  if (threads <= 1) { work(0, nrows); return; }
  std::vector<std::thread> th; size_t chunk = (nrows + threads - 1) / threads;
  for (int t = 0; t < threads; t++) th.emplace_back(work, std::min(nrows, t * chunk), std::min(nrows, (t + 1) * chunk));
  for (auto& x : th) x.join();
}

template <class F>
static void cross_term(size_t n, const F* az1, const F* bz1, const F* cz1, const F& u1,
                       const F* az2, const F* bz2, const F* cz2, const F& u2, F* T) {
  for (size_t i = 0; i < n; i++) T[i] = az1[i] * bz2[i] + az2[i] * bz1[i] - u1 * cz2[i] - u2 * cz1[i];
}

template <class F>
static void axpy(size_t n, const F* a, const F& r, const F* b, F* out) {
  for (size_t i = 0; i < n; i++) out[i] = a[i] + r * b[i];
}

template <class F>
static long first_unsat_relaxed(size_t n, const F* az, const F* bz, const F* cz, const F& u, const F* E) {
  for (size_t i = 0; i < n; i++) {
    F rhs = u * cz[i];
    if (E) rhs = rhs + E[i];
    if (az[i] * bz[i] != rhs) return (long)i;
  }
  return -1;
}

}  // namespace orc
