// Host check of Fp::dot (lazy-reduction dot product, vimz_amd/csrc/fp.hpp) against the sum of single products: random, zero, one
// and maximal representations, 1..12 terms, all four fields.  Built and run by tests/test_host_field.py.
#include <cstdio>
#include <random>
#include "fp.hpp"
using namespace vz;
template <class P> int check(const char* name) {
  typedef Fp<P> F;
  std::mt19937_64 rng(7);
  int bad = 0;
  F pm1 = F::neg(F::one());                 // p-1 in Montgomery form is some value; also use raw maximal representations
  F maxrep; for (int i = 0; i < 8; i++) maxrep.v[i] = P::MOD.w[i]; maxrep.v[0] -= 1;     // the integer p-1 as a representation
  for (int it = 0; it < 40000; it++) {
    int n = 1 + (int)(rng() % 12);
    F a[12], b[12];
    for (int k = 0; k < n; k++) {
      auto rnd = [&]() { F x; for (int i = 0; i < 8; i++) x.v[i] = (uint32_t)rng(); x.v[7] &= 0x3fffffffu; return F::mul(x, F::r2()); };
      int mode = (int)(rng() % 8);
      a[k] = mode == 0 ? maxrep : mode == 1 ? pm1 : mode == 2 ? F::zero() : rnd();
      mode = (int)(rng() % 8);
      b[k] = mode == 0 ? maxrep : mode == 1 ? pm1 : mode == 2 ? F::one() : rnd();
    }
    F want = F::zero();
    for (int k = 0; k < n; k++) want = F::add(want, F::mul(a[k], b[k]));
    F got = F::dot(a, b, n);
    if (!got.eq(want)) { if (bad < 3) printf("%s mismatch n=%d it=%d\n", name, n, it); bad++; }
  }
  // all-maximal
  { F a[12], b[12]; for (int k = 0; k < 12; k++) a[k] = b[k] = maxrep; F want = F::zero(); for (int k = 0; k < 12; k++) want = F::add(want, F::mul(a[k], b[k])); if (!F::dot(a, b, 12).eq(want)) { printf("%s all-max mismatch\n", name); bad++; } }
  printf("%s: %d mismatches\n", name, bad);
  return bad;
}
int main() { return check<BnFr>("BnFr") + check<BnFq>("BnFq") + check<PallasFp>("PallasFp") + check<VestaFq>("VestaFq"); }
