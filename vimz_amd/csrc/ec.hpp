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

// 2 * affine point (mdbl-2008-s-1); caller guarantees the point is not the identity.
template <class F>
VZ_HD XYZZ<F> dbl_affine(const Affine<F>& a) {
  XYZZ<F> r;
  F U = F::dbl(a.y);
  F V = F::sqr(U);
  F W = F::mul(U, V);
  F S = F::mul(a.x, V);
  F X2 = F::sqr(a.x);
  F M = F::add(F::dbl(X2), X2);
  r.X = F::sub(F::sqr(M), F::dbl(S));
  r.Y = F::sub(F::mul(M, F::sub(S, r.X)), F::mul(W, a.y));
  r.ZZ = V; r.ZZZ = W;
  return r;
}

// dbl-2008-s-1
template <class F>
VZ_HD XYZZ<F> dbl(const XYZZ<F>& p) {
  if (p.is_identity()) return p;
  XYZZ<F> r;
  F U = F::dbl(p.Y);
  F V = F::sqr(U);
  F W = F::mul(U, V);
  F S = F::mul(p.X, V);
  F X2 = F::sqr(p.X);
  F M = F::add(F::dbl(X2), X2);
  r.X = F::sub(F::sqr(M), F::dbl(S));
  r.Y = F::sub(F::mul(M, F::sub(S, r.X)), F::mul(W, p.Y));
  r.ZZ = F::mul(V, p.ZZ);
  r.ZZZ = F::mul(W, p.ZZZ);
  return r;
}

// acc += q (affine), madd-2008-s.  Handles identity on either side, doubling and cancellation.
template <class F>
VZ_HD void add_mixed(XYZZ<F>& acc, const Affine<F>& q) {
  if (aff_is_identity(q)) return;
  if (acc.is_identity()) { acc.X = q.x; acc.Y = q.y; acc.ZZ = F::one(); acc.ZZZ = F::one(); return; }
  F U2 = F::mul(q.x, acc.ZZ);
  F S2 = F::mul(q.y, acc.ZZZ);
  F Pv = F::sub(U2, acc.X);
  F R = F::sub(S2, acc.Y);
  if (Pv.is_zero()) {
    if (R.is_zero()) acc = dbl_affine(q); else acc = XYZZ<F>::identity();
    return;
  }
  F PP = F::sqr(Pv);
  F PPP = F::mul(Pv, PP);
  F Q = F::mul(acc.X, PP);
  F X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
  F Y3 = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(acc.Y, PPP));
  acc.X = X3; acc.Y = Y3;
  acc.ZZ = F::mul(acc.ZZ, PP);
  acc.ZZZ = F::mul(acc.ZZZ, PPP);
}

// acc += q (XYZZ), add-2008-s.
template <class F>
VZ_HD void add_full(XYZZ<F>& acc, const XYZZ<F>& q) {
  if (q.is_identity()) return;
  if (acc.is_identity()) { acc = q; return; }
  F U1 = F::mul(acc.X, q.ZZ);
  F U2 = F::mul(q.X, acc.ZZ);
  F S1 = F::mul(acc.Y, q.ZZZ);
  F S2 = F::mul(q.Y, acc.ZZZ);
  F Pv = F::sub(U2, U1);
  F R = F::sub(S2, S1);
  if (Pv.is_zero()) {
    if (R.is_zero()) acc = dbl(acc); else acc = XYZZ<F>::identity();
    return;
  }
  F PP = F::sqr(Pv);
  F PPP = F::mul(Pv, PP);
  F Q = F::mul(U1, PP);
  F X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
  F Y3 = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(S1, PPP));
  acc.X = X3; acc.Y = Y3;
  acc.ZZ = F::mul(F::mul(acc.ZZ, q.ZZ), PP);
  acc.ZZZ = F::mul(F::mul(acc.ZZZ, q.ZZZ), PPP);
}

template <class F>
VZ_HD Affine<F> to_affine(const XYZZ<F>& p) {  // one inversion; identity -> (0,0)
  Affine<F> a;
  if (p.is_identity()) { a.x = F::zero(); a.y = F::zero(); return a; }
  F zi3 = F::pow_pm2(p.ZZZ);                 // 1/ZZZ
  F zi2 = F::sqr(F::mul(zi3, p.ZZ));       // (ZZ/ZZZ)^2 = 1/ZZ   (ZZ^3 = ZZZ^2)
  a.x = F::mul(p.X, zi2);
  a.y = F::mul(p.Y, zi3);
  return a;
}

// The four curves.  `Base` = coordinate field, `Scalar` = scalar field (group order).
// `Coord` = the device-internal form of the coordinate field (fp29.hpp).
struct BnG1 { typedef Fp<BnFq> Base; typedef Fp29<BnFq> Coord; typedef Fp<BnFr> Scalar; };
struct Grumpkin { typedef Fp<BnFr> Base; typedef Fp29<BnFr> Coord; typedef Fp<BnFq> Scalar; };
struct Pallas { typedef Fp<PallasFp> Base; typedef Fp29<PallasFp> Coord; typedef Fp<VestaFq> Scalar; };
struct Vesta { typedef Fp<VestaFq> Base; typedef Fp29<VestaFq> Coord; typedef Fp<PallasFp> Scalar; };

// Device storage of curve data: coordinates at a stride of 10 words (40 B): affine = 20 words, XYZZ = 40 words.
constexpr int COORD_WORDS = 10;
constexpr int AFFINE_WORDS = 2 * COORD_WORDS;
constexpr int XYZZ_WORDS = 4 * COORD_WORDS;

}  // namespace vz
