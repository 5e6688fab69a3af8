// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).
//
// Short-Weierstrass curves y^2 = x^3 + b (a = 0) in Jacobian coordinates:
// BN254 G1 (b=3) and Grumpkin (b=-17) are the reference's cycle
// (vimz/src/nova_snark_backend/mod.rs:19-20); Pallas/Vesta (b=5) are the cycle
// BASELINE.json's north_star names (SURVEY.md finding F1, Appendix E).
// Restates the standard Jacobian formulas (EFD "dbl-2009-l", "add-2007-bl",
// "madd-2007-bl") that halo2curves / pasta_curves implement (not vendored).
// Affine identity is encoded (0,0) as those crates do.
//
// Pippenger MSM follows the structure of nova-snark 0.23.0 `cpu_best_multiexp`
// (SURVEY.md §2 native table, †): window c ≈ ln(n), per-thread chunks, buckets,
// running-sum reduction, Horner combine.
#pragma once
#include <thread>
#include <vector>
#include <cmath>
#include "field.hpp"

namespace orc {

template <class F>
struct Affine {
  F x, y;
  bool is_identity() const { return x.is_zero() && y.is_zero(); }
};

template <class F, int BI>  // BI: curve constant b as signed small integer
struct Curve {
  typedef F Base;
  typedef Affine<F> Aff;
  struct Jac {
    F X, Y, Z;
    bool is_identity() const { return Z.is_zero(); }
  };
  static F b() { return F::from_i64(BI); }
  static Jac identity() { Jac j; j.X = F::zero(); j.Y = F::one(); j.Z = F::zero(); return j; }
  static Jac from_affine(const Aff& a) {
    if (a.is_identity()) return identity();
    Jac j; j.X = a.x; j.Y = a.y; j.Z = F::one(); return j;
  }
  static bool on_curve(const Aff& a) {
    if (a.is_identity()) return true;
    return a.y.sqr() == a.x.sqr() * a.x + b();
  }
  static Aff to_affine(const Jac& p) {
    Aff a;
    if (p.is_identity()) { a.x = F::zero(); a.y = F::zero(); return a; }
    F zi = p.Z.inv(); F zi2 = zi.sqr();
    a.x = p.X * zi2; a.y = p.Y * zi2 * zi;
    return a;
  }
  static Jac dbl(const Jac& p) {
    if (p.is_identity()) return p;
    F A = p.X.sqr(), B = p.Y.sqr(), C = B.sqr();
    F D = ((p.X + B).sqr() - A - C).dbl();
    F E = A.dbl() + A;
    F Fv = E.sqr();
    Jac r;
    r.X = Fv - D.dbl();
    r.Y = E * (D - r.X) - C.dbl().dbl().dbl();
    r.Z = (p.Y * p.Z).dbl();
    return r;
  }
  static Jac add(const Jac& p, const Jac& q) {
    if (p.is_identity()) return q;
    if (q.is_identity()) return p;
    F Z1Z1 = p.Z.sqr(), Z2Z2 = q.Z.sqr();
    F U1 = p.X * Z2Z2, U2 = q.X * Z1Z1;
    F S1 = p.Y * q.Z * Z2Z2, S2 = q.Y * p.Z * Z1Z1;
    if (U1 == U2) {
      if (S1 == S2) return dbl(p);
      return identity();
    }
    F H = U2 - U1;
    F I = H.dbl().sqr();
    F J = H * I;
    F rr = (S2 - S1).dbl();
    F V = U1 * I;
    Jac r;
    r.X = rr.sqr() - J - V.dbl();
    r.Y = rr * (V - r.X) - (S1 * J).dbl();
    r.Z = ((p.Z + q.Z).sqr() - Z1Z1 - Z2Z2) * H;
    return r;
  }
  static Jac add_mixed(const Jac& p, const Aff& q) {
    if (q.is_identity()) return p;
    if (p.is_identity()) return from_affine(q);
    F Z1Z1 = p.Z.sqr();
    F U2 = q.x * Z1Z1;
    F S2 = q.y * p.Z * Z1Z1;
    if (p.X == U2) {
      if (p.Y == S2) return dbl(p);
      return identity();
    }
    F H = U2 - p.X;
    F HH = H.sqr();
    F I = HH.dbl().dbl();
    F J = H * I;
    F rr = (S2 - p.Y).dbl();
    F V = p.X * I;
    Jac r;
    r.X = rr.sqr() - J - V.dbl();
    r.Y = rr * (V - r.X) - (p.Y * J).dbl();
    r.Z = (p.Z + H).sqr() - Z1Z1 - HH;
    return r;
  }
  static Aff neg(const Aff& a) { Aff r; r.x = a.x; r.y = a.is_identity() ? a.y : a.y.neg(); return r; }

  // scalar: canonical 4-limb little-endian integer
  static Jac mul(const Aff& p, const u64* k) {
    Jac acc = identity();
    for (int i = 255; i >= 0; i--) {
      acc = dbl(acc);
      if ((k[i / 64] >> (i % 64)) & 1) acc = add_mixed(acc, p);
    }
    return acc;
  }

  static inline unsigned window(const u64* k, int lo, int c) {
    // bits [lo, lo+c) of a 256-bit integer
    if (lo >= 256) return 0;
    int limb = lo / 64, off = lo % 64;
    u64 v = k[limb] >> off;
    if (off + c > 64 && limb + 1 < 4) v |= k[limb + 1] << (64 - off);
    return (unsigned)(v & ((1ull << c) - 1));
  }

  // Pippenger over one contiguous chunk.
  static Jac msm_serial(const Aff* bases, const u64* scalars, size_t n, int c) {
    if (n == 0) return identity();
    int nwin = (256 + c - 1) / c;
    std::vector<Jac> win(nwin);
    std::vector<Jac> buckets((size_t)1 << c);
    for (int w = 0; w < nwin; w++) {
      for (auto& bk : buckets) bk = identity();
      bool any = false;
      for (size_t i = 0; i < n; i++) {
        unsigned d = window(scalars + 4 * i, w * c, c);
        if (d) { buckets[d] = add_mixed(buckets[d], bases[i]); any = true; }
      }
      Jac run = identity(), sum = identity();
      if (any) for (size_t bkt = buckets.size() - 1; bkt >= 1; bkt--) {
        run = add(run, buckets[bkt]);
        sum = add(sum, run);
      }
      win[w] = sum;
    }
    Jac acc = identity();
    for (int w = nwin - 1; w >= 0; w--) {
      for (int k = 0; k < c; k++) acc = dbl(acc);
      acc = add(acc, win[w]);
    }
    return acc;
  }

  static Jac msm(const Aff* bases, const u64* scalars, size_t n, int threads) {
    if (n == 0) return identity();
    int c = n < 32 ? 3 : (int)std::ceil(std::log((double)n));
    if (threads <= 1 || n < 1024) return msm_serial(bases, scalars, n, c);
    std::vector<Jac> part(threads);
    std::vector<std::thread> th;
    size_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
      th.emplace_back([&, t]() {
        size_t lo = std::min(n, (size_t)t * chunk), hi = std::min(n, lo + chunk);
        int cc = (hi - lo) < 32 ? 3 : (int)std::ceil(std::log((double)(hi - lo)));
        part[t] = msm_serial(bases + lo, scalars + 4 * lo, hi - lo, cc);
      });
    }
    for (auto& x : th) x.join();
    Jac acc = identity();
    for (int t = 0; t < threads; t++) acc = add(acc, part[t]);
    return acc;
  }
};

typedef Curve<BnFq, 3> BnG1;          // scalar field BnFr
typedef Curve<BnFr, -17> Grumpkin;    // scalar field BnFq
typedef Curve<PallasFp, 5> Pallas;    // scalar field VestaFq
typedef Curve<VestaFq, 5> Vesta;      // scalar field PallasFp

}  // namespace orc
