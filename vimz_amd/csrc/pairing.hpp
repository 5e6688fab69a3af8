// The optimal ate pairing on BN254 (alt_bn128), host only — what the EVM's pairing precompile (address 8) behind contracts/*Verifier.sol
// computes, for the PRODUCT-side decider verification (`Decider::verify`, reached from vimz/src/sonobe_backend/decider.rs:31-50
// `verify_final_proof`, mod.rs:80): vimz_decider_verify / vimz_decider_verify_key in groth16.hip.
//
// Tower: Fq2 = Fq[u]/(u² + 1);  Fq6 = Fq2[v]/(v³ − ξ), ξ = 9 + u;  Fq12 = Fq6[w]/(w² − v).  G2 is the D-type twist y² = x³ + 3/ξ over Fq2,
// untwisted by (x, y) -> (x w², y w³).  Miller loop over 6t + 2 = 29793968203157093288 with affine line functions (one Fq2 inversion a
// step: this is a verifier that runs a handful of times a proof, not a hot loop) and the two Frobenius corrections; the final
// exponentiation is a plain square-and-multiply by (q¹² − 1)/r — no cyclotomic tricks, no magic constants beyond the two moduli (the Frobenius
// coefficients ξ^((q−1)/3), ξ^((q−1)/2) and the exponent are computed from them at first use).
// Checked in the CPU suite against the reference's own vectors: the six committed marketplace/proofs/*.proof verify through
// vimz_decider_verify_key with the constants of contracts/*Verifier.sol (tests/test_decider_verify_host.py).
#pragma once
#include <array>
#include <vector>
#include "ec.hpp"

namespace vz {
namespace pairing {

typedef Fp<BnFq> Fq;

struct Fq2 {
  Fq c0, c1;
  static constexpr bool LAZY = false;
  static VZ_HD Fq2 zero() { Fq2 r; r.c0 = Fq::zero(); r.c1 = Fq::zero(); return r; }
  static VZ_HD Fq2 one() { Fq2 r; r.c0 = Fq::one(); r.c1 = Fq::zero(); return r; }
  VZ_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  VZ_HD bool is_zero_mod() const { return is_zero(); }
  VZ_HD bool eq(const Fq2& b) const { return c0.eq(b.c0) && c1.eq(b.c1); }
  VZ_HD Fq2 canon() const { return *this; }
  static VZ_HD Fq2 add(const Fq2& a, const Fq2& b) { Fq2 r; r.c0 = Fq::add(a.c0, b.c0); r.c1 = Fq::add(a.c1, b.c1); return r; }
  static VZ_HD Fq2 sub(const Fq2& a, const Fq2& b) { Fq2 r; r.c0 = Fq::sub(a.c0, b.c0); r.c1 = Fq::sub(a.c1, b.c1); return r; }
  template <int K> static VZ_HD Fq2 sub(const Fq2& a, const Fq2& b) { return sub(a, b); }
  static VZ_HD Fq2 neg(const Fq2& a) { return sub(zero(), a); }
  static VZ_HD Fq2 dbl(const Fq2& a) { return add(a, a); }
  static VZ_HD Fq2 mul(const Fq2& a, const Fq2& b) {      // Karatsuba: three base-field products
    const Fq t0 = Fq::mul(a.c0, b.c0), t1 = Fq::mul(a.c1, b.c1);
    const Fq t2 = Fq::mul(Fq::add(a.c0, a.c1), Fq::add(b.c0, b.c1));
    Fq2 r; r.c0 = Fq::sub(t0, t1); r.c1 = Fq::sub(Fq::sub(t2, t0), t1); return r;
  }
  static VZ_HD Fq2 sqr(const Fq2& a) {
    const Fq t = Fq::mul(a.c0, a.c1);
    Fq2 r; r.c0 = Fq::mul(Fq::add(a.c0, a.c1), Fq::sub(a.c0, a.c1)); r.c1 = Fq::dbl(t); return r;
  }
  static VZ_HD Fq2 pow_pm2(const Fq2& a) {      // the inverse (0 -> 0): conj(a) / (c0² + c1²)
    const Fq n = Fq::pow_pm2(Fq::add(Fq::sqr(a.c0), Fq::sqr(a.c1)));
    Fq2 r; r.c0 = Fq::mul(a.c0, n); r.c1 = Fq::neg(Fq::mul(a.c1, n)); return r;
  }
  static VZ_HD Fq2 conj(const Fq2& a) { Fq2 r; r.c0 = a.c0; r.c1 = Fq::neg(a.c1); return r; }
  static VZ_HD Fq2 scale(const Fq2& a, const Fq& k) { Fq2 r; r.c0 = Fq::mul(a.c0, k); r.c1 = Fq::mul(a.c1, k); return r; }
  static VZ_HD Fq2 mul_xi(const Fq2& a) {       // (9 + u)(c0 + c1 u) = (9 c0 − c1) + (9 c1 + c0) u
    const Fq n0 = Fq::dbl(Fq::dbl(Fq::dbl(a.c0))), n1 = Fq::dbl(Fq::dbl(Fq::dbl(a.c1)));
    Fq2 r; r.c0 = Fq::sub(Fq::add(n0, a.c0), a.c1); r.c1 = Fq::add(Fq::add(n1, a.c1), a.c0); return r;
  }
};
typedef Affine<Fq2> G2PAff;      // a point of BN254 G2 (twist coordinates); identity = (0, 0)
typedef Affine<Fq> G1PAff;

inline Fq fq_u64(uint64_t v) { Fq c = Fq::zero(); c.v[0] = (uint32_t)v; c.v[1] = (uint32_t)(v >> 32); return Fq::to_mont(c); }

struct Fq6 {
  Fq2 c0, c1, c2;
  static Fq6 zero() { Fq6 r; r.c0 = r.c1 = r.c2 = Fq2::zero(); return r; }
  static Fq6 one() { Fq6 r = zero(); r.c0 = Fq2::one(); return r; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
  bool eq(const Fq6& b) const { return c0.eq(b.c0) && c1.eq(b.c1) && c2.eq(b.c2); }
  static Fq6 add(const Fq6& a, const Fq6& b) { Fq6 r; r.c0 = Fq2::add(a.c0, b.c0); r.c1 = Fq2::add(a.c1, b.c1); r.c2 = Fq2::add(a.c2, b.c2); return r; }
  static Fq6 sub(const Fq6& a, const Fq6& b) { Fq6 r; r.c0 = Fq2::sub(a.c0, b.c0); r.c1 = Fq2::sub(a.c1, b.c1); r.c2 = Fq2::sub(a.c2, b.c2); return r; }
  static Fq6 neg(const Fq6& a) { return sub(zero(), a); }
  static Fq6 mul(const Fq6& a, const Fq6& b) {      // schoolbook with v³ = ξ
    const Fq2 a0b0 = Fq2::mul(a.c0, b.c0), a1b1 = Fq2::mul(a.c1, b.c1), a2b2 = Fq2::mul(a.c2, b.c2);
    const Fq2 t12 = Fq2::sub(Fq2::sub(Fq2::mul(Fq2::add(a.c1, a.c2), Fq2::add(b.c1, b.c2)), a1b1), a2b2);      // a1 b2 + a2 b1
    const Fq2 t01 = Fq2::sub(Fq2::sub(Fq2::mul(Fq2::add(a.c0, a.c1), Fq2::add(b.c0, b.c1)), a0b0), a1b1);      // a0 b1 + a1 b0
    const Fq2 t02 = Fq2::sub(Fq2::sub(Fq2::mul(Fq2::add(a.c0, a.c2), Fq2::add(b.c0, b.c2)), a0b0), a2b2);      // a0 b2 + a2 b0
    Fq6 r; r.c0 = Fq2::add(a0b0, Fq2::mul_xi(t12)); r.c1 = Fq2::add(t01, Fq2::mul_xi(a2b2)); r.c2 = Fq2::add(t02, a1b1); return r;
  }
  static Fq6 mul_v(const Fq6& a) { Fq6 r; r.c0 = Fq2::mul_xi(a.c2); r.c1 = a.c0; r.c2 = a.c1; return r; }      // · v
  static Fq6 inv(const Fq6& a) {
    const Fq2 A = Fq2::sub(Fq2::sqr(a.c0), Fq2::mul_xi(Fq2::mul(a.c1, a.c2)));
    const Fq2 B = Fq2::sub(Fq2::mul_xi(Fq2::sqr(a.c2)), Fq2::mul(a.c0, a.c1));
    const Fq2 Cc = Fq2::sub(Fq2::sqr(a.c1), Fq2::mul(a.c0, a.c2));
    const Fq2 F = Fq2::add(Fq2::mul(a.c0, A), Fq2::mul_xi(Fq2::add(Fq2::mul(a.c2, B), Fq2::mul(a.c1, Cc))));
    const Fq2 fi = Fq2::pow_pm2(F);
    Fq6 r; r.c0 = Fq2::mul(A, fi); r.c1 = Fq2::mul(B, fi); r.c2 = Fq2::mul(Cc, fi); return r;
  }
};

struct Fq12 {
  Fq6 c0, c1;      // c0 + c1 w
  static Fq12 one() { Fq12 r; r.c0 = Fq6::one(); r.c1 = Fq6::zero(); return r; }
  bool is_one() const { return c0.eq(Fq6::one()) && c1.is_zero(); }
  static Fq12 mul(const Fq12& a, const Fq12& b) {
    const Fq6 t0 = Fq6::mul(a.c0, b.c0), t1 = Fq6::mul(a.c1, b.c1);
    const Fq6 t2 = Fq6::mul(Fq6::add(a.c0, a.c1), Fq6::add(b.c0, b.c1));
    Fq12 r; r.c0 = Fq6::add(t0, Fq6::mul_v(t1)); r.c1 = Fq6::sub(Fq6::sub(t2, t0), t1); return r;
  }
  static Fq12 sqr(const Fq12& a) { return mul(a, a); }
};

// big exponents as little-endian 32-bit words
typedef std::vector<uint32_t> Big;
inline Big big_from(const uint32_t* w, int n) { return Big(w, w + n); }
inline void big_trim(Big& a) { while (a.size() > 1 && a.back() == 0) a.pop_back(); }
inline Big big_mul(const Big& a, const Big& b) {
  Big r(a.size() + b.size(), 0);
  for (size_t i = 0; i < a.size(); i++) {
    uint64_t c = 0;
    for (size_t j = 0; j < b.size(); j++) { const uint64_t t = (uint64_t)a[i] * b[j] + r[i + j] + c; r[i + j] = (uint32_t)t; c = t >> 32; }
    r[i + b.size()] += (uint32_t)c;
  }
  big_trim(r); return r;
}
inline Big big_sub_small(Big a, uint32_t s) { uint64_t br = s; for (size_t i = 0; i < a.size() && br; i++) { const uint64_t d = (uint64_t)a[i] - br; a[i] = (uint32_t)d; br = (d >> 32) & 1; } big_trim(a); return a; }
inline int big_cmp(const Big& a, const Big& b) {
  if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
  for (size_t i = a.size(); i-- > 0;) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
inline int big_bits(const Big& a) { int n = (int)a.size() * 32; while (n > 0 && !((a[(n - 1) >> 5] >> ((n - 1) & 31)) & 1u)) n--; return n; }
// quotient of a / d by schoolbook shift-subtract (exact or not; a few thousand bits, once per process); *rem_zero: the division was exact
inline Big big_div(const Big& a, const Big& d, bool* rem_zero) {
  Big q(a.size(), 0), rem(1, 0);
  for (int i = big_bits(a) - 1; i >= 0; i--) {
    // rem = rem * 2 + bit
    uint32_t carry = (a[i >> 5] >> (i & 31)) & 1u;
    for (size_t k = 0; k < rem.size(); k++) { const uint32_t nc = rem[k] >> 31; rem[k] = (rem[k] << 1) | carry; carry = nc; }
    if (carry) rem.push_back(carry);
    big_trim(rem);
    if (big_cmp(rem, d) >= 0) {
      uint64_t br = 0;
      for (size_t k = 0; k < rem.size(); k++) { const uint64_t x = (uint64_t)rem[k] - (k < d.size() ? d[k] : 0) - br; rem[k] = (uint32_t)x; br = (x >> 32) & 1; }
      big_trim(rem);
      q[i >> 5] |= 1u << (i & 31);
    }
  }
  if (rem_zero) *rem_zero = rem.size() == 1 && rem[0] == 0;
  big_trim(q); return q;
}
inline Big big_div_small(const Big& a, uint32_t d, uint32_t* rem) {
  Big q(a.size(), 0); uint64_t r = 0;
  for (size_t i = a.size(); i-- > 0;) { const uint64_t cur = (r << 32) | a[i]; q[i] = (uint32_t)(cur / d); r = cur % d; }
  if (rem) *rem = (uint32_t)r;
  big_trim(q); return q;
}

inline Fq2 fq2_pow(const Fq2& a, const Big& e) {
  Fq2 acc = Fq2::one();
  for (int i = big_bits(e) - 1; i >= 0; i--) { acc = Fq2::sqr(acc); if ((e[i >> 5] >> (i & 31)) & 1u) acc = Fq2::mul(acc, a); }
  return acc;
}
inline Fq12 fq12_pow(const Fq12& a, const Big& e) {
  Fq12 acc = Fq12::one();
  for (int i = big_bits(e) - 1; i >= 0; i--) { acc = Fq12::sqr(acc); if ((e[i >> 5] >> (i & 31)) & 1u) acc = Fq12::mul(acc, a); }
  return acc;
}

struct Consts {
  Fq2 xi, twist_b, g12, g13;      // ξ, 3/ξ, ξ^((q−1)/3), ξ^((q−1)/2)
  Big final_exp;                   // (q¹² − 1) / r
  bool ok = false;
};
inline const Consts& consts() {
  static const Consts K = [] {
    Consts k;
    k.xi.c0 = fq_u64(9); k.xi.c1 = Fq::one();
    Fq2 three = Fq2::zero(); three.c0 = fq_u64(3);
    k.twist_b = Fq2::mul(three, Fq2::pow_pm2(k.xi));
    const Big q = big_from(BnFq::MOD.w, 8), r = big_from(BnFr::MOD.w, 8);
    const Big qm1 = big_sub_small(q, 1);
    uint32_t rem3 = 1, rem2 = 1;
    k.g12 = fq2_pow(k.xi, big_div_small(qm1, 3, &rem3));
    k.g13 = fq2_pow(k.xi, big_div_small(qm1, 2, &rem2));
    Big q2 = big_mul(q, q), q4 = big_mul(q2, q2), q8 = big_mul(q4, q4), q12 = big_mul(q8, q4);
    bool exact = false;
    k.final_exp = big_div(big_sub_small(q12, 1), r, &exact);
    k.ok = exact && rem3 == 0 && rem2 == 0;
    return k;
  }();
  return K;
}

inline bool g1_on_curve(const G1PAff& p) {      // (0, 0) = identity
  if (p.x.is_zero() && p.y.is_zero()) return true;
  return Fq::sqr(p.y).eq(Fq::add(Fq::mul(Fq::sqr(p.x), p.x), fq_u64(3)));
}
inline bool g2_is_identity(const G2PAff& p) { return p.x.is_zero() && p.y.is_zero(); }
inline bool g2_on_curve(const G2PAff& p) {
  if (g2_is_identity(p)) return true;
  return Fq2::sqr(p.y).eq(Fq2::add(Fq2::mul(Fq2::sqr(p.x), p.x), consts().twist_b));
}
// r·P == identity (EIP-197 requires G2 operands in the order-r subgroup)
inline bool g2_in_subgroup(const G2PAff& p) {
  if (g2_is_identity(p)) return true;
  XYZZ<Fq2> acc = XYZZ<Fq2>::identity();
  for (int i = 255; i >= 0; i--) { acc = dbl(acc); if ((BnFr::MOD.w[i >> 5] >> (i & 31)) & 1u) add_mixed(acc, p); }
  return acc.is_identity();
}

// the line through T (slope lam, twist coordinates) evaluated at P ∈ G1, up to a factor in Fq:  yP − lam·xP·w + (lam·xT − yT)·w³
inline Fq12 line(const Fq2& lam, const G2PAff& T, const G1PAff& P) {
  Fq12 l; l.c0 = Fq6::zero(); l.c1 = Fq6::zero();
  l.c0.c0.c0 = P.y;
  l.c1.c0 = Fq2::neg(Fq2::scale(lam, P.x));
  l.c1.c1 = Fq2::sub(Fq2::mul(lam, T.x), T.y);
  return l;
}
// the vertical line x = xT at P:  xP − xT·w²  (w² = v)
inline Fq12 line_vertical(const G2PAff& T, const G1PAff& P) {
  Fq12 l; l.c0 = Fq6::zero(); l.c1 = Fq6::zero();
  l.c0.c0.c0 = P.x; l.c0.c1 = Fq2::neg(T.x);
  return l;
}
struct MillerPoint { G2PAff p; bool inf = false; };
// f *= l_{T,T}(P); T = 2T
inline void step_double(Fq12& f, MillerPoint& T, const G1PAff& P) {
  if (T.inf) return;
  if (T.p.y.is_zero()) { f = Fq12::mul(f, line_vertical(T.p, P)); T.inf = true; return; }
  const Fq2 xx = Fq2::sqr(T.p.x);
  const Fq2 lam = Fq2::mul(Fq2::add(Fq2::dbl(xx), xx), Fq2::pow_pm2(Fq2::dbl(T.p.y)));
  f = Fq12::mul(f, line(lam, T.p, P));
  G2PAff n; n.x = Fq2::sub(Fq2::sqr(lam), Fq2::dbl(T.p.x)); n.y = Fq2::sub(Fq2::mul(lam, Fq2::sub(T.p.x, n.x)), T.p.y);
  T.p = n;
}
// f *= l_{T,Q}(P); T = T + Q
inline void step_add(Fq12& f, MillerPoint& T, const G2PAff& Q, const G1PAff& P) {
  if (T.inf) { T.p = Q; T.inf = false; return; }
  if (T.p.x.eq(Q.x)) {
    if (T.p.y.eq(Q.y)) { step_double(f, T, P); return; }
    f = Fq12::mul(f, line_vertical(T.p, P)); T.inf = true; return;
  }
  const Fq2 lam = Fq2::mul(Fq2::sub(Q.y, T.p.y), Fq2::pow_pm2(Fq2::sub(Q.x, T.p.x)));
  f = Fq12::mul(f, line(lam, T.p, P));
  G2PAff n; n.x = Fq2::sub(Fq2::sub(Fq2::sqr(lam), T.p.x), Q.x); n.y = Fq2::sub(Fq2::mul(lam, Fq2::sub(T.p.x, n.x)), T.p.y);
  T.p = n;
}
inline Fq12 miller_loop(const G2PAff& Q, const G1PAff& P) {
  if (g2_is_identity(Q) || (P.x.is_zero() && P.y.is_zero())) return Fq12::one();
  const Consts& K = consts();
  static const uint64_t ATE = 11347224129447541672ull;      // 6t + 2 = 29793968203157093288 = 2^64 + this: 65 bits, the loop starts below the leading one
  Fq12 f = Fq12::one();
  MillerPoint T; T.p = Q;
  for (int i = 63; i >= 0; i--) {
    f = Fq12::sqr(f);
    step_double(f, T, P);
    if ((ATE >> i) & 1ull) step_add(f, T, Q, P);
  }
  // pi(Q) = (conj(x)·ξ^((q−1)/3), conj(y)·ξ^((q−1)/2));  −pi²(Q)
  G2PAff Q1; Q1.x = Fq2::mul(Fq2::conj(Q.x), K.g12); Q1.y = Fq2::mul(Fq2::conj(Q.y), K.g13);
  G2PAff Q2; Q2.x = Fq2::mul(Fq2::conj(Q1.x), K.g12); Q2.y = Fq2::neg(Fq2::mul(Fq2::conj(Q1.y), K.g13));
  step_add(f, T, Q1, P);
  step_add(f, T, Q2, P);
  return f;
}

// prod e(P_k, Q_k) == 1: one Miller loop per pair, ONE final exponentiation (what the precompile answers)
inline bool product_is_one(const std::vector<std::pair<G1PAff, G2PAff>>& pairs) {
  Fq12 f = Fq12::one();
  for (auto& pq : pairs) f = Fq12::mul(f, miller_loop(pq.second, pq.first));
  return fq12_pow(f, consts().final_exp).is_one();
}

}  // namespace pairing
}  // namespace vz
