// circomlib Poseidon parameters over BN254 Fr, regenerated from the Poseidon reference Grain LFSR
// (circomlib is a package.json dependency of the reference, circuits/package.json:7,
// not vendored).  Host code; the tables are uploaded to the GPU for the witness kernels.
#pragma once
#include <mutex>
#include <stdexcept>
#include <vector>
#include "builder.hpp"

namespace vz {
namespace cb {

template <class F>
struct PoseidonTableT {
  int t = 0, rf = 8, rp = 0;
  std::vector<F> C;  // (rf+rp)*t
  std::vector<F> M;  // t*t row-major
};
typedef PoseidonTableT<Fe> PoseidonTable;

inline int poseidon_rp(int t) {
  static const int RP[16] = {56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68};
  return RP[t - 2];
}

class GrainLfsr {
  bool s_[80];
 public:
  GrainLfsr(unsigned field_bits, unsigned t, unsigned rf, unsigned rp) {
    int k = 0;
    auto put = [&](unsigned v, int w) { for (int i = w - 1; i >= 0; i--) s_[k++] = (v >> i) & 1u; };
    put(1, 2); put(0, 4); put(field_bits, 12); put(t, 12); put(rf, 10); put(rp, 10);
    while (k < 80) s_[k++] = true;
    for (int i = 0; i < 160; i++) step();
  }
  bool step() {
    bool b = s_[62] ^ s_[51] ^ s_[38] ^ s_[23] ^ s_[13] ^ s_[0];
    for (int i = 0; i < 79; i++) s_[i] = s_[i + 1];
    s_[79] = b;
    return b;
  }
  bool filtered() { for (;;) { bool keep = step(); bool v = step(); if (keep) return v; } }
  // 254-bit big-endian sample into canonical little-endian limbs
  void sample(uint32_t out[8]) {
    for (int i = 0; i < 8; i++) out[i] = 0;
    for (int i = 253; i >= 0; i--) if (filtered()) out[i >> 5] |= 1u << (i & 31);
  }
};

// Parameters for the prime of field parameters P.  For BnFr these are circomlib's; for the other fields the same
// procedure (Grain LFSR seeded with field=1, sbox=0, n=254|255, t, R_F, R_P; rejection-sampled round constants; Cauchy MDS)
// over that prime — used by the augmented circuits' hashes on the second curve of the cycle (aug/).
template <class P>
inline const PoseidonTableT<Fp<P>>& poseidon_table_t(int t) {
  typedef Fp<P> F;
  static PoseidonTableT<F> cache[18];
  static std::mutex mu;                       // several provers (one per row segment) may fold from different threads
  std::lock_guard<std::mutex> guard(mu);
  PoseidonTableT<F>& T = cache[t];
  if (T.t == t) return T;
  T.rf = 8; T.rp = poseidon_rp(t);
  GrainLfsr g((unsigned)P::BITS, (unsigned)t, (unsigned)T.rf, (unsigned)T.rp);
  const int nbits = P::BITS;
  auto sample = [&](uint32_t* out) { for (int i = 0; i < 8; i++) out[i] = 0; for (int i = nbits - 1; i >= 0; i--) if (g.filtered()) out[i >> 5] |= 1u << (i & 31); };
  auto below_modulus = [](const uint32_t* v) {
    for (int i = 7; i >= 0; i--) if (v[i] != P::MOD.w[i]) return v[i] < P::MOD.w[i];
    return false;
  };
  auto to_fe_reduced = [&](uint32_t* v) {  // v < 2^BITS < 2p: one conditional subtraction
    if (!below_modulus(v)) { uint64_t br = 0; for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)v[i] - P::MOD.w[i] - br; v[i] = (uint32_t)d; br = (d >> 32) & 1; } }
    F x; for (int i = 0; i < 8; i++) x.v[i] = v[i];
    return F::to_mont(x);
  };
  while ((int)T.C.size() < (T.rf + T.rp) * t) {
    uint32_t v[8]; sample(v);
    if (below_modulus(v)) T.C.push_back(to_fe_reduced(v));  // rejection sampling for round constants
  }
  std::vector<F> xs(t), ys(t);
  for (int i = 0; i < t; i++) { uint32_t v[8]; sample(v); xs[i] = to_fe_reduced(v); }  // MDS seeds: mod p, no rejection
  for (int i = 0; i < t; i++) { uint32_t v[8]; sample(v); ys[i] = to_fe_reduced(v); }
  T.M.resize((size_t)t * t);
  for (int i = 0; i < t; i++) for (int j = 0; j < t; j++) T.M[(size_t)i * t + j] = F::pow_pm2(F::add(xs[i], ys[j]));
  T.t = t;
  return T;
}
inline const PoseidonTable& poseidon_table(int t) { return poseidon_table_t<BnFr>(t); }

// ---- partial rounds in sparse form ------------------------------------------------------------------------------------
// A partial round x <- M·S0(x + c) applies the S-box to lane 0 only, so the change of basis diag(1, N) on the other lanes
// commutes with it.  Pushing that basis through the rounds (Poseidon paper, App. B) turns every partial round's dense
// matrix into  [[d00, v^T], [w', I]]  (2t-1 products instead of t^2) with transformed constants; one (t-1)x(t-1) matrix is
// applied after the last partial round.  Lane 0 — the only value a circuit needs from these rounds — is unchanged.
template <class F>
struct PoseidonSparseT {
  int t = 0;
  std::vector<F> ctil;   // rp * t      transformed round constants
  std::vector<F> row;    // rp * t      first row of each round's sparse matrix
  std::vector<F> col;    // rp * (t-1)  first column below the corner
  std::vector<F> Pfin;   // (t-1)^2     basis change applied to lanes 1.. after the last partial round
};

template <class F>
inline std::vector<F> mat_inverse(std::vector<F> a, int n) {   // Gauss-Jordan, row-major n x n
  std::vector<F> inv((size_t)n * n, F::zero());
  for (int i = 0; i < n; i++) inv[(size_t)i * n + i] = F::one();
  for (int c = 0; c < n; c++) {
    int piv = c; while (piv < n && a[(size_t)piv * n + c].is_zero()) piv++;
    if (piv == n) throw std::runtime_error("poseidon: singular matrix");
    if (piv != c) for (int j = 0; j < n; j++) { std::swap(a[(size_t)piv * n + j], a[(size_t)c * n + j]); std::swap(inv[(size_t)piv * n + j], inv[(size_t)c * n + j]); }
    const F pi = F::pow_pm2(a[(size_t)c * n + c]);
    for (int j = 0; j < n; j++) { a[(size_t)c * n + j] = F::mul(a[(size_t)c * n + j], pi); inv[(size_t)c * n + j] = F::mul(inv[(size_t)c * n + j], pi); }
    for (int r = 0; r < n; r++) {
      if (r == c) continue;
      const F f = a[(size_t)r * n + c];
      if (f.is_zero()) continue;
      for (int j = 0; j < n; j++) { a[(size_t)r * n + j] = F::sub(a[(size_t)r * n + j], F::mul(f, a[(size_t)c * n + j])); inv[(size_t)r * n + j] = F::sub(inv[(size_t)r * n + j], F::mul(f, inv[(size_t)c * n + j])); }
    }
  }
  return inv;
}

template <class P>
inline const PoseidonSparseT<Fp<P>>& poseidon_sparse_t(int t) {
  typedef Fp<P> F;
  const PoseidonTableT<F>& T = poseidon_table_t<P>(t);
  static PoseidonSparseT<F> cache[18];
  static std::mutex mu;
  std::lock_guard<std::mutex> guard(mu);
  PoseidonSparseT<F>& S = cache[t];
  if (S.t == t) return S;
  const int m = t - 1, rp = T.rp;
  S.ctil.resize((size_t)rp * t); S.row.resize((size_t)rp * t); S.col.resize((size_t)rp * m);
  std::vector<F> N((size_t)m * m, F::zero());
  for (int i = 0; i < m; i++) N[(size_t)i * m + i] = F::one();
  for (int r = 0; r < rp; r++) {
    const F* c = &T.C[(size_t)(T.rf / 2 + r) * t];
    const std::vector<F> Ninv = mat_inverse(N, m);
    S.ctil[(size_t)r * t] = c[0];
    for (int i = 0; i < m; i++) { F acc = F::zero(); for (int k = 0; k < m; k++) acc = F::add(acc, F::mul(Ninv[(size_t)i * m + k], c[1 + k])); S.ctil[(size_t)r * t + 1 + i] = acc; }
    std::vector<F> D((size_t)t * t);                 // D = M · diag(1, N)
    for (int i = 0; i < t; i++) {
      D[(size_t)i * t] = T.M[(size_t)i * t];
      for (int j = 0; j < m; j++) { F acc = F::zero(); for (int k = 0; k < m; k++) acc = F::add(acc, F::mul(T.M[(size_t)i * t + 1 + k], N[(size_t)k * m + j])); D[(size_t)i * t + 1 + j] = acc; }
    }
    std::vector<F> Dhat((size_t)m * m), w(m);
    for (int i = 0; i < m; i++) { w[i] = D[(size_t)(1 + i) * t]; for (int j = 0; j < m; j++) Dhat[(size_t)i * m + j] = D[(size_t)(1 + i) * t + 1 + j]; }
    const std::vector<F> Dinv = mat_inverse(Dhat, m);
    for (int j = 0; j < t; j++) S.row[(size_t)r * t + j] = D[j];
    for (int i = 0; i < m; i++) { F acc = F::zero(); for (int k = 0; k < m; k++) acc = F::add(acc, F::mul(Dinv[(size_t)i * m + k], w[k])); S.col[(size_t)r * m + i] = acc; }
    N.swap(Dhat);
  }
  S.Pfin = N;
  S.t = t;
  return S;
}

// The permutation on values: state[0..t) in place.  wires (optional): receives x^2, x^4, x^5 of every S-box in (round, lane)
// order, skipping S-boxes whose input is the constant (0 + C) — lane 0 of round 0 when lane0_const — i.e. exactly the wires
// the constraint systems allocate (aug/cs.hpp).
template <class FP>
inline void poseidon_permute(Fp<FP>* s, int t, bool lane0_const, std::vector<Fp<FP>>* wires) {
  typedef Fp<FP> F;
  const PoseidonTableT<F>& P = poseidon_table_t<FP>(t);
  const PoseidonSparseT<F>& S = poseidon_sparse_t<FP>(t);
  F u[POSEIDON_MAX_T];
  auto sbox = [&](F& x, bool emit) { const F x2 = F::sqr(x), x4 = F::sqr(x2), x5 = F::mul(x4, x); if (wires && emit) { wires->push_back(x2); wires->push_back(x4); wires->push_back(x5); } x = x5; };
  auto dot = [](const F* a, const F* b, int n) {      // F::dot takes at most 12 terms
    F acc = F::dot(a, b, n < 12 ? n : 12);
    for (int k = 12; k < n; k += 12) acc = F::add(acc, F::dot(a + k, b + k, n - k < 12 ? n - k : 12));
    return acc;
  };
  auto mix = [&]() {
    for (int i = 0; i < t; i++) u[i] = dot(&P.M[(size_t)i * t], s, t);
    for (int i = 0; i < t; i++) s[i] = u[i];
  };
  const int half = P.rf / 2;
  for (int r = 0; r < half; r++) {
    for (int i = 0; i < t; i++) s[i] = F::add(s[i], P.C[(size_t)r * t + i]);
    for (int i = 0; i < t; i++) sbox(s[i], !(r == 0 && i == 0 && lane0_const));
    mix();
  }
  const int m = t - 1;
  for (int r = 0; r < P.rp; r++) {
    const F* ct = &S.ctil[(size_t)r * t]; const F* row = &S.row[(size_t)r * t]; const F* col = &S.col[(size_t)r * m];
    for (int i = 0; i < t; i++) s[i] = F::add(s[i], ct[i]);
    sbox(s[0], true);
    const F n0 = dot(row, s, t);
    for (int i = 1; i < t; i++) s[i] = F::add(s[i], F::mul(col[i - 1], s[0]));
    s[0] = n0;
  }
  for (int i = 0; i < m; i++) u[i] = dot(&S.Pfin[(size_t)i * m], s + 1, m);
  for (int i = 0; i < m; i++) s[1 + i] = u[i];
  for (int r = half + P.rp; r < P.rf + P.rp; r++) {
    for (int i = 0; i < t; i++) s[i] = F::add(s[i], P.C[(size_t)r * t + i]);
    for (int i = 0; i < t; i++) sbox(s[i], true);
    mix();
  }
}

// One HashJob of a step circuit's witness program on the host, with the wires the GPU kernel (witness.hpp: poseidon_group) writes:
// x^2, x^4, x^5 of every S-box in (round, lane) order; round-0 S-boxes of lane 0 (state 0 + C) and of lanes whose input is the
// constant zero (bit i of fold_mask0 set for lane i) are folded into constants and emit nothing; when the output is bound to a
// public-output wire (`bound`) the x^5 of (last round, lane 0) is the substituted signal and is not a wire either.
// state: s[0..t) in, permuted in place (s[0] = the hash); returns the number of wires written.
template <class FP>
inline uint32_t poseidon_job_wires(Fp<FP>* s, int t, uint32_t fold_mask0, bool bound, Fp<FP>* wires) {
  typedef Fp<FP> F;
  const PoseidonTableT<F>& P = poseidon_table_t<FP>(t);
  const PoseidonSparseT<F>& S = poseidon_sparse_t<FP>(t);
  F u[POSEIDON_MAX_T];
  uint32_t n = 0;
  auto sbox = [&](F& x, bool emit, bool skip_x5) {
    const F x2 = F::sqr(x), x4 = F::sqr(x2), x5 = F::mul(x4, x);
    if (emit) { wires[n++] = x2; wires[n++] = x4; if (!skip_x5) wires[n++] = x5; }
    x = x5;
  };
  auto dot = [](const F* a, const F* b, int k) { return F::dot(a, b, k); };      // t <= 9 terms
  auto mix = [&]() {
    for (int i = 0; i < t; i++) u[i] = dot(&P.M[(size_t)i * t], s, t);
    for (int i = 0; i < t; i++) s[i] = u[i];
  };
  const int half = P.rf / 2, R = P.rf + P.rp;
  for (int r = 0; r < half; r++) {
    for (int i = 0; i < t; i++) s[i] = F::add(s[i], P.C[(size_t)r * t + i]);
    for (int i = 0; i < t; i++) sbox(s[i], !(r == 0 && (i == 0 || ((fold_mask0 >> i) & 1u))), false);
    mix();
  }
  const int m = t - 1;
  for (int r = 0; r < P.rp; r++) {
    const F* ct = &S.ctil[(size_t)r * t]; const F* row = &S.row[(size_t)r * t]; const F* col = &S.col[(size_t)r * m];
    for (int i = 0; i < t; i++) s[i] = F::add(s[i], ct[i]);
    sbox(s[0], true, false);
    const F n0 = dot(row, s, t);
    for (int i = 1; i < t; i++) s[i] = F::add(s[i], F::mul(col[i - 1], s[0]));
    s[0] = n0;
  }
  for (int i = 0; i < m; i++) u[i] = dot(&S.Pfin[(size_t)i * m], s + 1, m);
  for (int i = 0; i < m; i++) s[1 + i] = u[i];
  for (int r = half + P.rp; r < R; r++) {
    for (int i = 0; i < t; i++) s[i] = F::add(s[i], P.C[(size_t)r * t + i]);
    for (int i = 0; i < t; i++) sbox(s[i], true, bound && r == R - 1 && i == 0);
    mix();
  }
  return n;
}
// wires a job of width t writes, given the number of its non-constant inputs
inline uint32_t poseidon_job_wire_count(int t, int rp, uint32_t nonconst_inputs, bool bound) { return 3u * (nonconst_inputs + 3u * t + rp + 4u * t) - (bound ? 1u : 0u); }

// Numeric hash on the host (IVC state chain, transcript, instance hashes).
template <class FP>
inline Fp<FP> poseidon_hash_t(const Fp<FP>* in, int n) {
  typedef Fp<FP> F;
  F s[POSEIDON_MAX_T];
  s[0] = F::zero();
  for (int i = 0; i < n; i++) s[i + 1] = in[i];
  poseidon_permute<FP>(s, n + 1, true, nullptr);
  return s[0];
}
// The dense textbook form (kept for the tests of the sparse form).
template <class FP>
inline Fp<FP> poseidon_hash_dense_t(const Fp<FP>* in, int n) {
  typedef Fp<FP> Fe;
  const int t = n + 1;
  const PoseidonTableT<Fe>& P = poseidon_table_t<FP>(t);
  Fe s[POSEIDON_MAX_T], u[POSEIDON_MAX_T];
  s[0] = Fe::zero();
  for (int i = 0; i < n; i++) s[i + 1] = in[i];
  for (int r = 0; r < P.rf + P.rp; r++) {
    for (int i = 0; i < t; i++) s[i] = Fe::add(s[i], P.C[(size_t)r * t + i]);
    const bool full = r < P.rf / 2 || r >= P.rf / 2 + P.rp;
    for (int i = 0; i < (full ? t : 1); i++) { Fe x2 = Fe::sqr(s[i]); Fe x4 = Fe::sqr(x2); s[i] = Fe::mul(x4, s[i]); }
    for (int i = 0; i < t; i++) {
      Fe acc = Fe::zero();
      for (int j = 0; j < t; j++) acc = Fe::add(acc, Fe::mul(P.M[(size_t)i * t + j], s[j]));
      u[i] = acc;
    }
    for (int i = 0; i < t; i++) s[i] = u[i];
  }
  return s[0];
}
inline Fe poseidon_hash(const Fe* in, int n) { return poseidon_hash_t<BnFr>(in, n); }

}  // namespace cb
}  // namespace vz
