// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).
//
// circomlib Poseidon over BN254 Fr, as included by the reference's hashers
// (circuits/src/utils/hashers.circom:4 `include ".../circomlib/circuits/poseidon.circom"`;
// circomlib ^2.0.5 is a package.json dependency, circuits/package.json:7, NOT vendored).
// Published algorithm restated: Hades permutation, x^5 S-box,
// R_F = 8, R_P = table[t-2], round constants and Cauchy MDS drawn from the Poseidon
// reference Grain-LFSR; state = [0, inputs...], output = state[0].
//
// The hashers on top follow circuits/src/utils/hashers.circom:
//   PairHasher        :7-16     Poseidon(2)(a,b)
//   _WindowFoldHasher :40-73    first window absorbs 8, then 7 per round with the running
//                               hash in lane 0 of the inputs, **ceil(L/8) rounds only**
//                               (so the tail of the row is never absorbed: SURVEY.md F5)
//   ArrayHasher       :19-23    = _WindowFoldHasher(L, 8)
//   HeadTailHasher    :115-120  PairHasher(head, ArrayHasher(tail))
#pragma once
#include <vector>
#include <mutex>
#include "field.hpp"

namespace orc {

struct PoseidonParams {
  int t, rf, rp;
  std::vector<BnFr> C;  // (rf+rp)*t round constants
  std::vector<BnFr> M;  // t*t, row-major: new[i] = sum_j M[i*t+j]*s[j]
};

static const int POSEIDON_RP[16] = {56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68};

struct Grain {
  unsigned char s[80];
  int n;
  Grain(int field_bits, int t, int rf, int rp) : n(field_bits) {
    int k = 0;
    auto app = [&](unsigned v, int w) { for (int i = w - 1; i >= 0; i--) s[k++] = (v >> i) & 1; };
    app(1, 2); app(0, 4); app((unsigned)field_bits, 12); app((unsigned)t, 12);
    app((unsigned)rf, 10); app((unsigned)rp, 10);
    for (int i = 0; i < 30; i++) s[k++] = 1;
    for (int i = 0; i < 160; i++) next();
  }
  int next() {
    int b = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0];
    memmove(s, s + 1, 79); s[79] = (unsigned char)b;
    return b;
  }
  int bit() { for (;;) { int a = next(); int b = next(); if (a) return b; } }
  void sample(u64* out) {  // n-bit big-endian sample -> little-endian limbs
    memset(out, 0, 32);
    for (int i = n - 1; i >= 0; i--) if (bit()) out[i / 64] |= 1ull << (i % 64);
  }
};

static inline const PoseidonParams& poseidon_params(int t) {
  static PoseidonParams cache[18];
  static std::mutex mu;
  std::lock_guard<std::mutex> g(mu);
  PoseidonParams& P = cache[t];
  if (P.t == t) return P;
  P.t = t; P.rf = 8; P.rp = POSEIDON_RP[t - 2];
  Grain gr(254, t, P.rf, P.rp);
  const u64* mod = BnFr::P().p;
  while ((int)P.C.size() < (P.rf + P.rp) * t) {
    u64 v[4]; gr.sample(v);
    if (cmp4(v, mod) < 0) P.C.push_back(BnFr::from_canonical(v));  // rejection sampling
  }
  std::vector<BnFr> xs(t), ys(t);
  for (int i = 0; i < t; i++) { u64 v[4]; gr.sample(v); xs[i] = BnFr::from_canonical(v); }  // mod r, no rejection
  for (int i = 0; i < t; i++) { u64 v[4]; gr.sample(v); ys[i] = BnFr::from_canonical(v); }
  P.M.resize(t * t);
  for (int i = 0; i < t; i++) for (int j = 0; j < t; j++) P.M[i * t + j] = (xs[i] + ys[j]).inv();
  return P;
}

static inline BnFr poseidon(const BnFr* in, int n) {
  int t = n + 1;
  const PoseidonParams& P = poseidon_params(t);
  std::vector<BnFr> s(t), u(t);
  s[0] = BnFr::zero();
  for (int i = 0; i < n; i++) s[i + 1] = in[i];
  int half = P.rf / 2;
  for (int r = 0; r < P.rf + P.rp; r++) {
    for (int i = 0; i < t; i++) s[i] = s[i] + P.C[r * t + i];
    if (r < half || r >= half + P.rp) { for (int i = 0; i < t; i++) s[i] = s[i].pow5(); }
    else s[0] = s[0].pow5();
    for (int i = 0; i < t; i++) {
      BnFr acc = BnFr::zero();
      for (int j = 0; j < t; j++) acc = acc + P.M[i * t + j] * s[j];
      u[i] = acc;
    }
    s.swap(u);
  }
  return s[0];
}

static inline BnFr pair_hash(const BnFr& a, const BnFr& b) { BnFr in[2] = {a, b}; return poseidon(in, 2); }

static inline BnFr array_hash(const BnFr* arr, int L) {
  const int W = 8;
  int rounds = (L + W - 1) / W;
  int first = L < W ? L : W;
  BnFr h = poseidon(arr, first);
  int processed = first;
  for (int r = 0; r < rounds - 1; r++) {
    int remaining = L - processed;
    int cur = remaining < W - 1 ? remaining : W - 1;
    BnFr in[8];
    in[0] = h;
    for (int i = 0; i < cur; i++) in[i + 1] = arr[processed + i];
    h = poseidon(in, cur + 1);
    processed += cur;
  }
  return h;
}

static inline BnFr head_tail_hash(const BnFr& head, const BnFr* tail, int L) {
  return pair_hash(head, array_hash(tail, L));
}

}  // namespace orc
