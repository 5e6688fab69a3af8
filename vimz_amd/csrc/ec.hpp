// Short-Weierstrass (a = 0) group arithmetic in extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): the cheapest accumulator for bucket sums — a mixed
// addition of an affine base costs 8M + 2S and needs no field inversion.  Identity: ZZ = 0.
// Affine identity is (0,0), the encoding halo2curves/pasta_curves hand over.
#pragma once
#include "fp.hpp"
#include "fp29.hpp"

namespace vz {

template <class F>
struct Affine { F x, y; };

template <class F>
struct XYZZ {
  F X, Y, ZZ, ZZZ;
  static VZ_HD XYZZ identity() { XYZZ r; r.X = F::zero(); r.Y = F::zero(); r.ZZ = F::zero(); r.ZZZ = F::zero(); return r; }
  VZ_HD bool is_identity() const { return ZZ.is_zero(); }
};

template <class F>
VZ_HD bool aff_is_identity(const Affine<F>& a) { return a.x.is_zero() && a.y.is_zero(); }

template <class F>
VZ_HD XYZZ<F> from_affine(const Affine<F>& a) {
  XYZZ<F> r;
  if (aff_is_identity(a)) return XYZZ<F>::identity();
  r.X = a.x; r.Y = a.y; r.ZZ = F::one(); r.ZZZ = F::one();
  return r;
}

// ---- bounds (multiples of p) the lazily reduced coordinate field Fp29 relies on ------------------------------------------------
// Every XYZZ point these formulas produce or consume satisfies   X < 5.3 p,  Y < 3.4 p,  ZZ, ZZZ < 1.5 p;   affine operands are
// canonical (< p).  A product needs Bx·By <= 64 and is < (Bx·By/128 + 1) p; sub<K>(a, b) = a − b + K p needs b < K p.  The bound of
// every intermediate is noted on its line.  For the canonical Fp<P> (host) the same code is ordinary modular arithmetic.

// 2 * affine point (mdbl-2008-s-1); caller guarantees the point is not the identity.
template <class F>
VZ_HD XYZZ<F> dbl_affine(const Affine<F>& a) {
  XYZZ<F> r;
  F U = F::dbl(a.y);                                   // 2
  F V = F::sqr(U);                                     // 1.04
  F W = F::mul(U, V);                                  // 1.02
  F S = F::mul(a.x, V);                                // 1.01
  F X2 = F::sqr(a.x);                                  // 1.01
  F M = F::add(F::dbl(X2), X2);                        // 3.1
  r.X = F::template sub<4>(F::sqr(M), F::dbl(S));      // 1.08 − 2.02 + 4 = 5.08
  r.Y = F::template sub<2>(F::mul(M, F::template sub<6>(S, r.X)), F::mul(W, a.y));     // 3.1·7.01 -> 1.17;  1.17 − 1.01 + 2 = 3.17
  r.ZZ = V; r.ZZZ = W;
  return r;
}

// dbl-2008-s-1
template <class F>
VZ_HD XYZZ<F> dbl(const XYZZ<F>& p) {
  if (p.is_identity()) return p;
  XYZZ<F> r;
  F U = F::dbl(p.Y);                                   // 6.8
  F V = F::sqr(U);                                     // 46.3 -> 1.37
  F W = F::mul(U, V);                                  // 9.3 -> 1.08
  F S = F::mul(p.X, V);                                // 7.3 -> 1.06
  F X2 = F::sqr(p.X);                                  // 28.1 -> 1.22
  F M = F::add(F::dbl(X2), X2);                        // 3.66
  r.X = F::template sub<4>(F::sqr(M), F::dbl(S));      // 1.11 − 2.12 + 4 = 5.11
  r.Y = F::template sub<2>(F::mul(M, F::template sub<6>(S, r.X)), F::mul(W, p.Y));     // 3.66·7.06 -> 1.21;  W·Y: 3.7 -> 1.03;  3.21
  r.ZZ = F::mul(V, p.ZZ);
  r.ZZZ = F::mul(W, p.ZZZ);
  return r;
}

// a·b − c·d for c < K·p.  The lazily reduced device form sums both products' columns and reduces ONCE (fp29.hpp: mul_add2 — 81 of a
// mixed addition's 1 390 multiply instructions and one of its nine reductions); the canonical form multiplies twice.
template <int K, class F>
VZ_HD F diff_of_products(const F& a, const F& b, const F& c, const F& d) {
  if constexpr (F::LAZY) return F::mul_add2(a, b, F::template sub<K>(F::zero(), c), d);
  else return F::template sub<2>(F::mul(a, b), F::mul(c, d));
}

// acc += q (affine), madd-2008-s.  Handles identity on either side, doubling and cancellation.
template <class F>
VZ_HD void add_mixed(XYZZ<F>& acc, const Affine<F>& q) {
  if (aff_is_identity(q)) return;
  if (acc.is_identity()) { acc.X = q.x; acc.Y = q.y; acc.ZZ = F::one(); acc.ZZZ = F::one(); return; }
  F U2 = F::mul(q.x, acc.ZZ);                          // 1.02
  F S2 = F::mul(q.y, acc.ZZZ);                         // 1.02
  F Pv = F::template sub<6>(U2, acc.X);                // 1.02 − X + 6 < 7.1
  F R = F::template sub<4>(S2, acc.Y);                 // 1.02 − Y + 4 < 5.1
  if (Pv.is_zero_mod()) {
    if (R.is_zero_mod()) acc = dbl_affine(q); else acc = XYZZ<F>::identity();
    return;
  }
  F PP = F::sqr(Pv);                                   // 50.4 -> 1.40
  F PPP = F::mul(Pv, PP);                              // 9.9 -> 1.08
  F Q = F::mul(acc.X, PP);                             // 7.4 -> 1.06
  F X3 = F::template sub<4>(F::sqr(R), F::add(PPP, F::dbl(Q)));     // R²: 26 -> 1.21;  PPP + 2Q < 3.2;  1.21 + 4 = 5.21
  F Y3 = diff_of_products<4>(R, F::template sub<6>(Q, X3), acc.Y, PPP);   // 5.1·7.06 + 4·1.08 = 40.4 -> 1.32   (two reductions: 3.29)
  acc.X = X3; acc.Y = Y3;
  acc.ZZ = F::mul(acc.ZZ, PP);                         // 2.1 -> 1.02
  acc.ZZZ = F::mul(acc.ZZZ, PPP);
}

// acc += q (XYZZ), add-2008-s.
template <class F>
VZ_HD void add_full(XYZZ<F>& acc, const XYZZ<F>& q) {
  if (q.is_identity()) return;
  if (acc.is_identity()) { acc = q; return; }
  F U1 = F::mul(acc.X, q.ZZ);                          // 5.3·1.5 -> 1.07
  F U2 = F::mul(q.X, acc.ZZ);                          // 1.07
  F S1 = F::mul(acc.Y, q.ZZZ);                         // 3.4·1.5 -> 1.04
  F S2 = F::mul(q.Y, acc.ZZZ);                         // 1.04
  F Pv = F::template sub<2>(U2, U1);                   // 3.07
  F R = F::template sub<2>(S2, S1);                    // 3.04
  if (Pv.is_zero_mod()) {
    if (R.is_zero_mod()) acc = dbl(acc); else acc = XYZZ<F>::identity();
    return;
  }
  F PP = F::sqr(Pv);                                   // 9.5 -> 1.08
  F PPP = F::mul(Pv, PP);                              // 1.03
  F Q = F::mul(U1, PP);                                // 1.01
  F X3 = F::template sub<4>(F::sqr(R), F::add(PPP, F::dbl(Q)));     // 1.08 + 4 = 5.08   (PPP + 2Q < 3.1)
  F Y3 = diff_of_products<2>(R, F::template sub<6>(Q, X3), S1, PPP);     // 3.04·7.01 + 2·1.03 = 23.4 -> 1.19   (two reductions: 3.17)
  acc.X = X3; acc.Y = Y3;
  acc.ZZ = F::mul(F::mul(acc.ZZ, q.ZZ), PP);
  acc.ZZZ = F::mul(F::mul(acc.ZZZ, q.ZZZ), PPP);
}

template <class F>
VZ_HD Affine<F> to_affine(const XYZZ<F>& p) {  // one inversion; identity -> (0,0); coordinates canonical
  Affine<F> a;
  if (p.is_identity()) { a.x = F::zero(); a.y = F::zero(); return a; }
  F zi3 = F::pow_pm2(p.ZZZ);                 // 1/ZZZ
  F zi2 = F::sqr(F::mul(zi3, p.ZZ));       // (ZZ/ZZZ)^2 = 1/ZZ   (ZZ^3 = ZZZ^2)
  a.x = F::mul(p.X, zi2).canon();
  a.y = F::mul(p.Y, zi3).canon();
  return a;
}

// The four curves.  `Base` = coordinate field, `Scalar` = scalar field (group order).
// `Coord` = the device-internal form of the coordinate field (fp29.hpp).
struct BnG1 { typedef Fp<BnFq> Base; typedef Fp29<BnFq> Coord; typedef Fp<BnFr> Scalar; };
struct Grumpkin { typedef Fp<BnFr> Base; typedef Fp29<BnFr> Coord; typedef Fp<BnFq> Scalar; };
struct Pallas { typedef Fp<PallasFp> Base; typedef Fp29<PallasFp> Coord; typedef Fp<VestaFq> Scalar; };
struct Vesta { typedef Fp<VestaFq> Base; typedef Fp29<VestaFq> Coord; typedef Fp<PallasFp> Scalar; };

// y² = x³ + b of the curve whose coordinates live in Fp<P> — for validating points that come from outside (proof blobs): the XYZZ
// formulas never use b, so an off-curve point would silently compute on another curve.
template <class P> struct CurveB;
template <> struct CurveB<BnFq> { static constexpr int value = 3; };        // BN254 G1
template <> struct CurveB<BnFr> { static constexpr int value = -17; };      // Grumpkin
template <> struct CurveB<PallasFp> { static constexpr int value = 5; };
template <> struct CurveB<VestaFq> { static constexpr int value = 5; };
template <class P>
inline bool aff_on_curve(const Affine<Fp<P>>& q) {      // the identity's encoding (0,0) counts as on the curve
  typedef Fp<P> F;
  if (aff_is_identity(q)) return true;
  F b = F::zero();
  const int bv = CurveB<P>::value;
  for (int i = 0; i < (bv < 0 ? -bv : bv); i++) b = F::add(b, F::one());
  if (bv < 0) b = F::neg(b);
  return F::sqr(q.y).eq(F::add(F::mul(F::sqr(q.x), q.x), b));
}

// Device storage of curve data: coordinates at a stride of 10 words (40 B): affine = 20 words, XYZZ = 40 words.
constexpr int COORD_WORDS = 10;
// A resident AFFINE point (commitment keys, window tables — the data the large MSM's accumulation GATHERS, 0.6 GB of tables at HD) is 64 bytes:
// x then y as 256-bit integers (the canonical Montgomery residues x·2^261 mod p), ONE 64-byte sector per point, 64-byte aligned; the nine 29-bit
// limbs are cut out after the load (Fp29::pack: shifts and masks).  Until round 5 it was 80 bytes (nine limbs in ten words per coordinate): two
// sectors per gathered point.  XYZZ accumulators (lazily reduced, up to 261 bits) keep ten words per coordinate.
constexpr int AFFINE_WORDS = 16;
constexpr int XYZZ_WORDS = 4 * COORD_WORDS;

}  // namespace vz
