// circomlib Poseidon parameters over BN254 Fr, regenerated from the Poseidon reference Grain LFSR
// (circomlib is a package.json dependency of the reference, circuits/package.json:7,
// not vendored).  Host code; the tables are uploaded to the GPU for the witness kernels.
#pragma once
#include <mutex>
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

// Numeric permutation on the host (used for the IVC state chain between witness batches).
template <class FP>
inline Fp<FP> poseidon_hash_t(const Fp<FP>* in, int n) {
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
