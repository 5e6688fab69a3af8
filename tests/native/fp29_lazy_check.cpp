// Host check of the lazily reduced coordinate field (vimz_amd/csrc/fp29.hpp) and of the bound discipline of the curve formulas
// written against it (vimz_amd/csrc/ec.hpp): the very same headers the device kernels compile, run here on the CPU.
//   1. field: mul / sqr / add / sub<K> / canon / is_zero_mod against the canonical 8x32 arithmetic of fp.hpp, on canonical operands
//      and on the largest representatives the formulas admit;
//   2. curve: long random chains of add_mixed / add_full / dbl on XYZZ<Fp29> against XYZZ<Fp> (every value canonical), compared
//      after to_affine, with the documented bounds (X < 5.3 p, Y < 3.4 p, ZZ, ZZZ < 1.5 p) and limb normalisation asserted after
//      every operation — including the doubling, cancellation and identity branches.
#include <cstdio>
#include <cstdlib>
#include <random>
#include "ec.hpp"
using namespace vz;

static std::mt19937_64 rng(12345);
static int g_trial, g_step, g_op;

template <class P> Fp<P> rand_fp() {
  Fp<P> c; for (int i = 0; i < 8; i++) c.v[i] = (uint32_t)rng();
  c.v[7] &= 0x0fffffffu;                       // < 2^252 < p: canonical integer
  return Fp<P>::to_mont(c);
}
// value of a 9x29 element as a long double multiple of p (bounds check only)
template <class P> long double ratio(const Fp29<P>& a) {
  long double v = 0, p = 0;
  for (int i = 8; i >= 0; i--) { v = v * 536870912.0L + a.v[i]; p = p * 536870912.0L + Fp29<P>::MOD29.l[i]; }
  return v / p;
}
template <class P> bool limbs_ok(const Fp29<P>& a) { for (int i = 0; i < 9; i++) if (a.v[i] >> 29) return false; return true; }
// a + k·p as a (normalised) 29-bit-limb integer: a larger representative of the same residue
template <class P> Fp29<P> lift(const Fp29<P>& a, uint32_t k) {
  Fp29<P> r; uint64_t c = 0;
  for (int i = 0; i < 9; i++) { c += a.v[i] + (uint64_t)Fp29<P>::MOD29.l[i] * k; r.v[i] = (uint32_t)c & 0x1fffffffu; c >>= 29; }
  return r;
}

template <class P> int field_checks(const char* name) {
  typedef Fp<P> S; typedef Fp29<P> G;
  int bad = 0;
  for (int it = 0; it < 20000; it++) {
    S a = rand_fp<P>(), b = rand_fp<P>();
    G x = G::from_std(a), y = G::from_std(b);
    const uint32_t ka = rng() % 7, kb = rng() % 7;
    G xl = lift<P>(x, ka), yl = lift<P>(y, kb);             // representatives up to 7 p + (p-1) < 8 p
    if (!(xl.canon().eq(x))) bad++;
    if ((ka + 1) * (kb + 1) <= 64) {
      G m = G::mul(xl, yl);
      if (!limbs_ok<P>(m) || ratio<P>(m) >= (ka + 1.0L) * (kb + 1.0L) / 128 + 1 || !m.to_std().eq(S::mul(a, b))) bad++;
      G q = G::sqr(xl);
      if (!limbs_ok<P>(q) || !q.to_std().eq(S::sqr(a))) bad++;
      if (!G::mul(xl, xl).canon().eq(q.canon())) bad++;
    }
    G s = G::add(xl, yl);
    if (!limbs_ok<P>(s) || !s.canon().to_std().eq(S::add(a, b))) bad++;
    G d = G::template sub<8>(xl, yl);                        // yl < 8 p
    if (!limbs_ok<P>(d) || ratio<P>(d) >= ka + 1 + 8 || !lift<P>(G::zero(), 0).is_zero()) bad++;
    {  // d may exceed 8 p: reduce by hand for the comparison
      G t = d; for (int r = 0; r < 3; r++) t = G::cond_sub(t.v, ct_times29(G::MOD29, 8));
      if (!t.to_std().eq(S::sub(a, b))) bad++;
    }
    // is_zero_mod: exactly the multiples of p below 8 p
    for (uint32_t k = 0; k < 8; k++) if (!lift<P>(G::zero(), k).is_zero_mod()) bad++;
    if (!a.is_zero() && xl.is_zero_mod()) bad++;
    G nz = G::neg(x);
    if (!G::add(nz, x).is_zero_mod() || !limbs_ok<P>(nz)) bad++;
    // weak_reduce: any representative below 2^261 comes back below 3 p as the same residue (sums of lazily reduced products)
    {
      const uint32_t kw = (uint32_t)(rng() % 120);             // x + kw·p < 121 p < 2^261
      const G big = lift<P>(x, kw), w = big.weak_reduce();
      if (!limbs_ok<P>(w) || ratio<P>(w) >= 3.0L || !w.canon().eq(x)) bad++;
      G top; for (int i = 0; i < 9; i++) top.v[i] = 0x1fffffffu;            // 2^261 − 1 itself
      const G wt = top.weak_reduce();
      if (!limbs_ok<P>(wt) || ratio<P>(wt) >= 3.0L) bad++;
    }
  }
  printf("%s: field %d mismatches\n", name, bad);
  return bad;
}

template <class C> int curve_checks(const char* name, const Affine<typename C::Base>& gen) {
  typedef typename C::Base S; typedef typename C::Coord G;
  int bad = 0;
  // a table of affine points k·G in both representations
  const int NP = 64;
  std::vector<Affine<S>> ps(NP); std::vector<Affine<G>> pg(NP);
  XYZZ<S> run = from_affine(gen);
  for (int i = 0; i < NP; i++) {
    ps[i] = to_affine(run);
    pg[i].x = G::from_std(ps[i].x); pg[i].y = G::from_std(ps[i].y);
    for (int k = 0; k < 3 + i % 5; k++) add_mixed(run, gen);
  }
  auto check = [&](const XYZZ<G>& a, const XYZZ<S>& b, const char* what) {
    if (!limbs_ok<typename S::Params>(a.X) || !limbs_ok<typename S::Params>(a.Y) || !limbs_ok<typename S::Params>(a.ZZ) || !limbs_ok<typename S::Params>(a.ZZZ)) { if (bad++ < 5) printf("%s: limbs %s\n", name, what); }
    if (ratio<typename S::Params>(a.X) >= 5.3L || ratio<typename S::Params>(a.Y) >= 3.4L || ratio<typename S::Params>(a.ZZ) >= 1.5L || ratio<typename S::Params>(a.ZZZ) >= 1.5L) { if (bad++ < 5) printf("%s: bound %s %Lf %Lf %Lf %Lf\n", name, what, ratio<typename S::Params>(a.X), ratio<typename S::Params>(a.Y), ratio<typename S::Params>(a.ZZ), ratio<typename S::Params>(a.ZZZ)); }
    Affine<G> ag = to_affine(a); Affine<S> as = to_affine(b);
    if (!ag.x.to_std().eq(as.x) || !ag.y.to_std().eq(as.y)) { if (bad++ < 5) printf("%s: value %s (trial %d step %d op %d)\n", name, what, g_trial, g_step, g_op); }
  };
  for (int trial = 0; trial < 40; trial++) {
    XYZZ<G> a = XYZZ<G>::identity(), c = XYZZ<G>::identity();
    XYZZ<S> b = XYZZ<S>::identity(), d = XYZZ<S>::identity();
    for (int step = 0; step < 300; step++) {
      const int op = rng() % 10, i = rng() % NP;
      g_trial = trial; g_step = step; g_op = op;
      Affine<G> qg = pg[i]; Affine<S> qs = ps[i];
      if (rng() & 1) { qg.y = G::neg(qg.y); qs.y = S::neg(qs.y); }
      switch (op) {
        case 0: case 1: case 2: case 3: add_mixed(a, qg); add_mixed(b, qs); break;
        case 4: add_mixed(c, qg); add_mixed(d, qs); break;
        case 5: add_full(a, c); add_full(b, d); break;
        case 6: a = dbl(a); b = dbl(b); break;
        case 7: add_full(a, a); add_full(b, b); break;                                   // doubling branch of the full addition
        case 8: { Affine<G> t = to_affine(a); Affine<S> u = to_affine(b); add_mixed(a, t); add_mixed(b, u); break; }     // doubling branch of the mixed addition
        default: { Affine<G> t = to_affine(a); Affine<S> u = to_affine(b);                                                 // cancellation
                   if (!aff_is_identity(t)) { t.y = G::neg(t.y); u.y = S::neg(u.y); add_mixed(a, t); add_mixed(b, u); if (!a.is_identity() || !b.is_identity()) bad++; } break; }
      }
      check(a, b, "a"); check(c, d, "c");
    }
  }
  printf("%s: curve %d mismatches\n", name, bad);
  return bad;
}

template <class F> Affine<F> pt(long x, long y) {
  auto f = [](long v) { F r = F::zero(); F o = F::one(); for (long k = 0; k < (v < 0 ? -v : v); k++) r = F::add(r, o); return v < 0 ? F::neg(r) : r; };
  Affine<F> a; a.x = f(x); a.y = f(y); return a;
}

int main() {
  int bad = 0;
  bad += field_checks<BnFr>("BnFr"); bad += field_checks<BnFq>("BnFq"); bad += field_checks<PallasFp>("PallasFp"); bad += field_checks<VestaFq>("VestaFq");
  bad += curve_checks<BnG1>("BnG1", pt<Fp<BnFq>>(1, 2));
  bad += curve_checks<Pallas>("Pallas", pt<Fp<PallasFp>>(-1, 2));
  bad += curve_checks<Vesta>("Vesta", pt<Fp<VestaFq>>(-1, 2));
  {   // Grumpkin: y^2 = x^3 - 17, generator (1, sqrt(-16))
    Affine<Fp<BnFr>> g; g.x = Fp<BnFr>::one();
    Fp<BnFr> y = Fp<BnFr>::zero();
    const uint32_t w[8] = {0x823f272cu, 0x833fc48du, 0xf1181294u, 0x2d270d45u, 0x06a45d63u, 0xcf135e75u, 0x00000002u, 0x00000000u};
    for (int i = 0; i < 8; i++) y.v[i] = w[i];
    g.y = Fp<BnFr>::to_mont(y);
    bad += curve_checks<Grumpkin>("Grumpkin", g);
  }
  printf("total %d\n", bad);
  return bad ? 1 : 0;
}
