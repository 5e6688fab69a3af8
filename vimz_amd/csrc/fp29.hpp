// Reduced-radix field arithmetic for the curve coordinates: 9 limbs of 29 bits (Montgomery, R' = 2^261).
//
// Why a second representation.  On gfx950 the multiply primitive is v_mad_u64_u32 (32x32 + 64 -> 64) and there is no
// add-with-carry-in form of it, so the saturated 8 x 32-bit CIOS of fp.hpp spends more instructions shuffling
// carries than multiplying (measured: 128 mad + 142 64-bit adds + 315 moves per product, and every mad depends on
// the previous one's carry: ~1280 cycles of latency per product at one wave per SIMD).  With 29-bit limbs a
// 58-bit partial product leaves 6 bits of headroom, so a whole column of 9 products (plus 9 reduction products)
// accumulates in one 64-bit register with plain mad chains — no carry handling, columns independent of each other
// (instruction-level parallelism for the latency-bound bucket reductions), carries resolved once per column by a
// shift.  162 mad + ~110 simple ops per product.
//
// Only device-resident curve data uses this form (commitment key, bucket accumulators); the C ABI keeps the
// standard [u64;4] Montgomery form (R = 2^256) and converts at upload / result time.  Storage stride is 10 words
// (40 B) per element so points stay 16-byte aligned: affine 80 B, XYZZ 160 B.
#pragma once
#include "fp.hpp"

namespace vz {

struct L29x9 { uint32_t l[9]; };

constexpr L29x9 ct_split29(const U256& x) {
  L29x9 r{};
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, w = bit >> 5, off = bit & 31;
    uint64_t v = (uint64_t)x.w[w] >> off;
    if (off > 3 && w + 1 < 8) v |= (uint64_t)x.w[w + 1] << (32 - off);
    r.l[i] = (uint32_t)(v & 0x1fffffffu);
  }
  return r;
}
constexpr L29x9 ct_times29(const L29x9& x, uint32_t k) {      // k·x in 29-bit limbs (k·x < 2^261)
  L29x9 r{}; uint64_t c = 0;
  for (int i = 0; i < 9; i++) { c += (uint64_t)x.l[i] * k; r.l[i] = (uint32_t)(c & 0x1fffffffu); c >>= 29; }
  return r;
}

// LAZY REDUCTION.  Nine 29-bit limbs hold integers below 2^261 = 128·2^254 > 128 p, seven bits more than a residue needs, and
// the Montgomery product of x < Bx·p and y < By·p is below (Bx·By/128 + 1)·p (p/R' < 2^-7).  So values are kept as normalised
// limbs (each < 2^29) of ANY representative below 128 p: `mul`/`sqr` never subtract p at the end (operand bounds must satisfy
// Bx·By <= 64: result < 1.5 p), `add` only propagates carries, `sub<K>` computes a − b + K·p for a caller-stated K·p >= b.
// Every formula in ec.hpp carries its bounds in comments.  `canon()` brings a value below 8 p to [0, p) — needed only where
// limbs are compared or leave this representation (to_std, table entries, equality) — and `is_zero_mod()` tests ≡ 0 (mod p)
// of a value below 8 p in three instructions on the common path (k = v0·p^-1 mod 2^29 must be < 8 for v = k·p).
// Measured: the conditional subtractions were 14 % of a mixed addition's issue slots (profiles/r02_ubench_fp29.txt).
template <class P>
struct Fp29 {
  typedef P Params;
  static constexpr int NW = 9;      // limbs used
  static constexpr int NWS = 10;    // storage stride in words
  static constexpr uint32_t MASK = 0x1fffffffu;
  static constexpr L29x9 MOD29 = ct_split29(P::MOD);
  static constexpr L29x9 ONE29 = ct_split29(ct_pow2_mod(261, P::MOD));
  static constexpr L29x9 R2_29 = ct_split29(ct_pow2_mod(522, P::MOD));
  static constexpr uint32_t N0_29 = P::N0 & MASK;            // -p^-1 mod 2^29
  static constexpr uint32_t PINV_29 = (0u - P::N0) & MASK;   //  p^-1 mod 2^29
  static constexpr bool LAZY = true;

  uint32_t v[9];

  static VZ_HD Fp29 zero() { Fp29 r; for (int i = 0; i < 9; i++) r.v[i] = 0; return r; }
  static VZ_HD Fp29 one() { Fp29 r; for (int i = 0; i < 9; i++) r.v[i] = ONE29.l[i]; return r; }
  VZ_HD bool is_zero() const { uint32_t o = 0; for (int i = 0; i < 9; i++) o |= v[i]; return o == 0; }    // the integer 0 (identity markers)
  VZ_HD bool eq(const Fp29& b) const { uint32_t o = 0; for (int i = 0; i < 9; i++) o |= v[i] ^ b.v[i]; return o == 0; }   // same limbs: canon() first

  // ≡ 0 (mod p) for a value below 8 p
  VZ_HD bool is_zero_mod() const {
    const uint32_t k = (v[0] * PINV_29) & MASK;
    if (k >= 8u) return false;
    uint64_t c = 0; uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { c += (uint64_t)MOD29.l[i] * k; o |= v[i] ^ ((uint32_t)c & MASK); c >>= 29; }
    return o == 0;
  }
  // t (limbs < 2^29) -> t - p if t >= p
  static VZ_HD Fp29 cond_sub(const uint32_t* t, const L29x9& m) {
    uint32_t s[9]; uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { uint32_t d = t[i] - m.l[i] - br; br = d >> 31; s[i] = d & MASK; }
    Fp29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = br ? t[i] : s[i];
    return r;
  }
  static VZ_HD Fp29 reduce_once(const uint32_t* t) { return cond_sub(t, MOD29); }
  // value < 8 p -> [0, p)
  VZ_HD Fp29 canon() const {
    constexpr L29x9 P4 = ct_times29(MOD29, 4), P2 = ct_times29(MOD29, 2);
    Fp29 r = cond_sub(v, P4);
    r = cond_sub(r.v, P2);
    return cond_sub(r.v, MOD29);
  }
  // ANY value below 2^261 (normalised limbs) -> the same residue below 3 p (below 1.9 p for BN254's two fields), without a comparison:
  // q = floor(v / 2^253), k = floor(q·KNUM / 256) with KNUM = floor(2^29 / (top limb of p + 1)) <= 2^261 / p, so k·p <= q·2^253 <= v, and what is
  // left is below 2^253 + p + q·(2^253 − KNUM·p/256).  One multiply-subtract chain — for sums of a few dozen lazily reduced products
  // (witness.hpp: the Poseidon rounds), where canon()'s comparisons would sit on a chain of dependent operations.
  static constexpr uint32_t KNUM = (1u << 29) / (MOD29.l[8] + 1u);
  VZ_HD Fp29 weak_reduce() const {
    const uint32_t k = ((v[8] >> 21) * KNUM) >> 8;
    Fp29 r; int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { c += (int64_t)v[i] - (int64_t)((uint64_t)k * MOD29.l[i]); r.v[i] = (uint32_t)c & MASK; c >>= 29; }
    return r;
  }
  // a + b (no reduction): bound Ba + Bb
  static VZ_HD Fp29 add(const Fp29& a, const Fp29& b) {
    Fp29 r; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { uint32_t x = a.v[i] + b.v[i] + c; c = x >> 29; r.v[i] = x & MASK; }
    return r;
  }
  // a − b + K·p, for b < K·p: bound Ba + K
  template <int K>
  static VZ_HD Fp29 sub(const Fp29& a, const Fp29& b) {
    constexpr L29x9 KP = ct_times29(MOD29, K);
    Fp29 r; int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { int32_t x = (int32_t)(a.v[i] + KP.l[i]) - (int32_t)b.v[i] + c; c = x >> 29; r.v[i] = (uint32_t)x & MASK; }
    return r;
  }
  static VZ_HD Fp29 neg(const Fp29& a) { return sub<1>(zero(), a); }      // a <= p (canonical operands: the affine bases)
  static VZ_HD Fp29 dbl(const Fp29& a) { return add(a, a); }

  // Montgomery reduction of the 18 product columns: result < (column value)/2^261 + p, limbs normalised, NO final subtraction
  static VZ_HD Fp29 redc(uint64_t* acc) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const uint32_t m = ((uint32_t)acc[i] * N0_29) & MASK;
#pragma unroll
      for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)m * MOD29.l[j];
      acc[i + 1] += acc[i] >> 29;   // low 29 bits of acc[i] are now zero
    }
    Fp29 r;
#pragma unroll
    for (int k = 9; k < 18; k++) {
      r.v[k - 9] = (uint32_t)acc[k] & MASK;
      if (k < 17) acc[k + 1] += acc[k] >> 29;
    }
    return r;
  }
  // Montgomery product a*b / 2^261: column-wise product (a column of 9 + 9 products of 29-bit limbs fits 64 bits), then the
  // word-by-word reduction.  Operand bounds Ba·Bb <= 64 -> result < 1.5 p.
  static VZ_HD Fp29 mul(const Fp29& a, const Fp29& b) {
    uint64_t acc[18];
#pragma unroll
    for (int k = 0; k < 18; k++) acc[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)a.v[i] * b.v[j];
    return redc(acc);
  }
  // (a*b + c*d) / 2^261 with ONE reduction: a column then holds 18 + 9 products of 29-bit limbs, < 2^62.8.  Operand bounds
  // Ba·Bb + Bc·Bd <= 64 -> result < 1.5 p.
  static VZ_HD Fp29 mul_add2(const Fp29& a, const Fp29& b, const Fp29& c, const Fp29& d) {
    uint64_t acc[18];
#pragma unroll
    for (int k = 0; k < 18; k++) acc[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)a.v[i] * b.v[j];
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)c.v[i] * d.v[j];
    return redc(acc);
  }
  // a^2: 45 products instead of 81 (cross terms against the doubled limbs, 2a_j < 2^30: a column holds at most four such
  // products, one square and nine reduction products: < 2^62.4)
  static VZ_HD Fp29 sqr(const Fp29& a) {
    uint64_t acc[18];
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
#pragma unroll
    for (int k = 0; k < 18; k++) acc[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      acc[2 * i] += (uint64_t)a.v[i] * a.v[i];
#pragma unroll
      for (int j = i + 1; j < 9; j++) acc[i + j] += (uint64_t)a.v[i] * d[j];
    }
    return redc(acc);
  }

  static VZ_HD Fp29 pow_pm2(const Fp29& a) {  // a^(p-2)   (a < 8 p; result < 1.5 p)
    uint32_t e[8]; uint64_t br = 2;
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)P::MOD.w[i] - br; e[i] = (uint32_t)d; br = (d >> 32) & 1; }
    Fp29 acc = one();
    for (int i = 255; i >= 0; i--) {
      acc = sqr(acc);
      if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, a);
    }
    return acc;
  }

  // ---- conversions with the standard representation (fp.hpp, 8 x 32 limbs, R = 2^256) ---------------------
  static VZ_HD Fp29 pack(const uint32_t* w8) {   // 256-bit integer -> 29-bit limbs (no arithmetic)
    Fp29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int bit = 29 * i, w = bit >> 5, off = bit & 31;
      uint64_t x = (uint64_t)w8[w] >> off;
      if (off > 3 && w + 1 < 8) x |= (uint64_t)w8[w + 1] << (32 - off);
      r.v[i] = (uint32_t)x & MASK;
    }
    return r;
  }
  VZ_HD void unpack(uint32_t* w8) const {        // 29-bit limbs (value < 2^256) -> 256-bit integer
    uint64_t buf = 0; int have = 0, o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      buf |= (uint64_t)v[i] << have; have += 29;
      if (have >= 32) { w8[o++] = (uint32_t)buf; buf >>= 32; have -= 32; }
    }
    if (o < 8) w8[o] = (uint32_t)buf;
  }
  // y = x*2^256 (standard Montgomery)  ->  x*2^261, canonical
  static VZ_HD Fp29 from_std(const Fp<P>& y) {
    Fp<P> t = y;
#pragma unroll
    for (int k = 0; k < 5; k++) t = Fp<P>::dbl(t);
    return pack(t.v);
  }
  // x*2^261 (any representative below 8 p) -> x*2^256: canonicalise, then five modular halvings
  VZ_HD Fp<P> to_std() const {
    const Fp29 cn = canon();
    uint32_t w[8]; cn.unpack(w);
    for (int k = 0; k < 5; k++) {
      uint64_t c = 0;
      if (w[0] & 1) { for (int i = 0; i < 8; i++) { c += (uint64_t)w[i] + P::MOD.w[i]; w[i] = (uint32_t)c; c >>= 32; } }
      for (int i = 0; i < 8; i++) w[i] = (w[i] >> 1) | (i < 7 ? w[i + 1] << 31 : (uint32_t)c << 31);
    }
    Fp<P> r; for (int i = 0; i < 8; i++) r.v[i] = w[i];
    return r;
  }
};

}  // namespace vz
