// 256-bit prime-field arithmetic for gfx950 (and the host side of the same library).
//
// Representation: 8 x 32-bit little-endian limbs, Montgomery form with R = 2^256 — bit-compatible
// with the `[u64; 4]` Montgomery limbs halo2curves / pasta_curves hand across the FFI,
// so buffers are reinterpreted, never converted.  CDNA4 has no 64x64 multiplier in the VALU; the
// multiply-accumulate primitive is v_mad_u64_u32 (32x32+64 -> 64), which `(u64)a * b + c` lowers to.
//
// All moduli here are < 2^255, so 2p < 2^256 and the CIOS accumulator never needs a tenth limb.
// Constants (R mod p, R^2 mod p, -p^-1 mod 2^32) are derived at compile time from the modulus alone.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VZ_HD __host__ __device__ __forceinline__
#else
#define VZ_HD inline
#endif

namespace vz {

struct U256 { uint32_t w[8]; };

// ---- compile-time big-integer helpers (used only to derive constants) -------------------------
constexpr bool ct_geq(const U256& a, const U256& b) {
  for (int i = 7; i >= 0; i--) { if (a.w[i] != b.w[i]) return a.w[i] > b.w[i]; }
  return true;
}
constexpr U256 ct_sub(const U256& a, const U256& b) {
  U256 r{}; uint64_t br = 0;
  for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.w[i] - b.w[i] - br; r.w[i] = (uint32_t)d; br = (d >> 32) & 1; }
  return r;
}
constexpr U256 ct_dbl_mod(const U256& a, const U256& p) {  // 2a mod p, a < p < 2^255
  U256 r{}; uint32_t c = 0;
  for (int i = 0; i < 8; i++) { r.w[i] = (a.w[i] << 1) | c; c = a.w[i] >> 31; }
  return ct_geq(r, p) ? ct_sub(r, p) : r;
}
constexpr U256 ct_pow2_mod(int k, const U256& p) {  // 2^k mod p
  U256 x{}; x.w[0] = 1;
  for (int i = 0; i < k; i++) x = ct_dbl_mod(x, p);
  return x;
}
constexpr uint32_t ct_n0(uint32_t p0) {  // -p^-1 mod 2^32
  uint32_t inv = 1;
  for (int i = 0; i < 5; i++) inv *= 2 - p0 * inv;
  return 0u - inv;
}
constexpr uint64_t ct_n0_64(uint64_t p0) {  // -p^-1 mod 2^64
  uint64_t inv = 1;
  for (int i = 0; i < 6; i++) inv *= 2 - p0 * inv;
  return 0ull - inv;
}
constexpr int ct_bits(const U256& p) {
  for (int i = 255; i >= 0; i--) if ((p.w[i / 32] >> (i % 32)) & 1) return i + 1;
  return 0;
}

#define VZ_FIELD(NAME, w7, w6, w5, w4, w3, w2, w1, w0)                         \
  struct NAME {                                                                \
    static constexpr U256 MOD = {{w0, w1, w2, w3, w4, w5, w6, w7}};            \
    static constexpr U256 R1 = ct_pow2_mod(256, MOD);                          \
    static constexpr U256 R2 = ct_pow2_mod(512, MOD);                          \
    static constexpr uint32_t N0 = ct_n0(w0);                                  \
    static constexpr uint64_t N0_64 = ct_n0_64((uint64_t)w0 | ((uint64_t)w1 << 32)); \
    static constexpr int BITS = ct_bits(MOD);                                  \
  };

// SURVEY.md Appendix E (moduli from contracts/ContrastVerifier.sol:35-38 and the Pasta definitions)
VZ_FIELD(BnFr, 0x30644e72u, 0xe131a029u, 0xb85045b6u, 0x8181585du, 0x2833e848u, 0x79b97091u, 0x43e1f593u, 0xf0000001u)
VZ_FIELD(BnFq, 0x30644e72u, 0xe131a029u, 0xb85045b6u, 0x8181585du, 0x97816a91u, 0x6871ca8du, 0x3c208c16u, 0xd87cfd47u)
VZ_FIELD(PallasFp, 0x40000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x224698fcu, 0x094cf91bu, 0x992d30edu, 0x00000001u)
VZ_FIELD(VestaFq, 0x40000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x224698fcu, 0x0994a8ddu, 0x8c46eb21u, 0x00000001u)

// Wave issue priority of the kernels on a step's critical chain (k_msm_small, k_spmv_cross16, k_msm_fixed) over the bulk waves on the same SIMDs.
// tools/ubench_icache.hip: at priority 3 a chain of dependent additions runs at its stand-alone speed beside the accumulation's loop — and the loop at 45 % of its own.
#ifndef VZ_CRIT_PRIO
#define VZ_CRIT_PRIO 3
#endif
#define VZ_SET_CRIT_PRIO() do { if (VZ_CRIT_PRIO) __builtin_amdgcn_s_setprio(VZ_CRIT_PRIO); } while (0)

template <class P>
struct Fp {
  typedef P Params;
  uint32_t v[8];

  static VZ_HD Fp zero() { Fp r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
  static VZ_HD Fp one() { Fp r; for (int i = 0; i < 8; i++) r.v[i] = P::R1.w[i]; return r; }
  static VZ_HD Fp r2() { Fp r; for (int i = 0; i < 8; i++) r.v[i] = P::R2.w[i]; return r; }
  VZ_HD bool is_zero() const { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= v[i]; return o == 0; }
  VZ_HD bool eq(const Fp& b) const { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= v[i] ^ b.v[i]; return o == 0; }
  // limbs < p: the invariant every operation here assumes of its operands (data from outside is checked with this)
  VZ_HD bool is_reduced() const {
    uint64_t br = 0;
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)v[i] - P::MOD.w[i] - br; br = (d >> 32) & 1; }
    return br != 0;
  }

  // r = a - p if a >= p  (a < 2p)
  static VZ_HD Fp reduce_once(const uint32_t* t) {
    uint32_t s[8]; uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)t[i] - P::MOD.w[i] - br; s[i] = (uint32_t)d; br = (d >> 32) & 1; }
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = br ? t[i] : s[i];
    return r;
  }

  static VZ_HD Fp add(const Fp& a, const Fp& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    {  // host pass: 4 x 64-bit limbs
      typedef unsigned __int128 u128;
      uint64_t A[4], B[4], M[4], t[4], s[4];
      __builtin_memcpy(A, a.v, 32); __builtin_memcpy(B, b.v, 32); __builtin_memcpy(M, P::MOD.w, 32);
      u128 c = 0;
      for (int i = 0; i < 4; i++) { c += (u128)A[i] + B[i]; t[i] = (uint64_t)c; c >>= 64; }
      uint64_t br = 0;
      for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - M[i] - br; s[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
      const uint64_t keep = (uint64_t)0 - br;      // branch-free select: the borrow is a coin flip on random data
      for (int i = 0; i < 4; i++) s[i] = (t[i] & keep) | (s[i] & ~keep);
      Fp r; __builtin_memcpy(r.v, s, 32);
      return r;
    }
#endif
    uint32_t t[8]; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; t[i] = (uint32_t)c; c >>= 32; }
    return reduce_once(t);  // a + b < 2p < 2^256: no carry out
  }
  static VZ_HD Fp sub(const Fp& a, const Fp& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    {
      typedef unsigned __int128 u128;
      uint64_t A[4], B[4], M[4], t[4];
      __builtin_memcpy(A, a.v, 32); __builtin_memcpy(B, b.v, 32); __builtin_memcpy(M, P::MOD.w, 32);
      uint64_t br = 0;
      for (int i = 0; i < 4; i++) { u128 d = (u128)A[i] - B[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
      const uint64_t mask = (uint64_t)0 - br;
      u128 c = 0;
      for (int i = 0; i < 4; i++) { c += (u128)t[i] + (M[i] & mask); t[i] = (uint64_t)c; c >>= 64; }
      Fp r; __builtin_memcpy(r.v, t, 32);
      return r;
    }
#endif
    uint32_t t[8]; uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.v[i] - b.v[i] - br; t[i] = (uint32_t)d; br = (d >> 32) & 1; }
    uint32_t mask = br ? 0xffffffffu : 0u;
    Fp r; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (uint64_t)t[i] + (P::MOD.w[i] & mask); r.v[i] = (uint32_t)c; c >>= 32; }
    return r;
  }
  static VZ_HD Fp neg(const Fp& a) { return sub(zero(), a); }
  static VZ_HD Fp dbl(const Fp& a) { return add(a, a); }
  // The interface the curve formulas (ec.hpp) are written against, shared with the lazily reduced Fp29: there sub<K> is
  // a − b + K·p without a reduction; here every value is canonical and K is irrelevant.
  static constexpr bool LAZY = false;
  template <int K> static VZ_HD Fp sub(const Fp& a, const Fp& b) { return sub(a, b); }
  VZ_HD bool is_zero_mod() const { return is_zero(); }
  VZ_HD Fp canon() const { return *this; }

  // CIOS Montgomery product a*b/R mod p.
  static VZ_HD Fp mul(const Fp& a, const Fp& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    // Host pass: CIOS over 4 x 64-bit limbs (x86-64 has the 64x64->128 multiplier the GPU lacks), in the "no-carry" form
    // valid for moduli whose top limb is below 2^63 - 1 (all four fields here): the two carry chains never overflow a limb.
    typedef unsigned __int128 u128;
    uint64_t A[4], B[4], M[4], t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    __builtin_memcpy(A, a.v, 32); __builtin_memcpy(B, b.v, 32); __builtin_memcpy(M, P::MOD.w, 32);  // little-endian host
    const uint64_t n0 = P::N0_64;
#define VZ_MAC(hi, lo, x, y, z) do { u128 _p = (u128)(x) * (y) + (z); lo = (uint64_t)_p; hi = (uint64_t)(_p >> 64); } while (0)
#define VZ_MAC2(hi, lo, x, y, z, w) do { u128 _p = (u128)(x) * (y) + (z) + (w); lo = (uint64_t)_p; hi = (uint64_t)(_p >> 64); } while (0)
    for (int i = 0; i < 4; i++) {
      const uint64_t bi = B[i];
      uint64_t Ah, Ch, lo, m;
      VZ_MAC(Ah, t0, A[0], bi, t0);
      m = t0 * n0;
      VZ_MAC(Ch, lo, m, M[0], t0);
      VZ_MAC2(Ah, t1, A[1], bi, t1, Ah);
      VZ_MAC2(Ch, t0, m, M[1], t1, Ch);
      VZ_MAC2(Ah, t2, A[2], bi, t2, Ah);
      VZ_MAC2(Ch, t1, m, M[2], t2, Ch);
      VZ_MAC2(Ah, t3, A[3], bi, t3, Ah);
      VZ_MAC2(Ch, t2, m, M[3], t3, Ch);
      t3 = Ch + Ah;
      (void)lo;
    }
#undef VZ_MAC
#undef VZ_MAC2
    {
      uint64_t t[4] = {t0, t1, t2, t3}, s[4];
      uint64_t br = 0;
      for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - M[i] - br; s[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
      const uint64_t keep = (uint64_t)0 - br;
      for (int i = 0; i < 4; i++) s[i] = (t[i] & keep) | (s[i] & ~keep);
      Fp r; __builtin_memcpy(r.v, s, 32);
      return r;
    }
#else
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t c = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) { c += (uint64_t)a.v[j] * b.v[i] + t[j]; t[j] = (uint32_t)c; c >>= 32; }
      uint32_t t8 = t[8] + (uint32_t)c;
      uint32_t m = t[0] * P::N0;
      c = (uint64_t)m * P::MOD.w[0] + t[0]; c >>= 32;
#pragma unroll
      for (int j = 1; j < 8; j++) { c += (uint64_t)m * P::MOD.w[j] + t[j]; t[j - 1] = (uint32_t)c; c >>= 32; }
      c += t8; t[7] = (uint32_t)c; t[8] = (uint32_t)(c >> 32);
    }
    return reduce_once(t);
#endif
  }
  static VZ_HD Fp sqr(const Fp& a) { return mul(a, a); }

  // sum_k a[k]*b[k] (Montgomery), n <= 12.  Host pass: the 512-bit products are accumulated unreduced and reduced once — the
  // Poseidon matrix rows of the verifier circuits' witnesses are dot products of 9, and a reduction costs as much as a product.
  // (12 p^2 < 2^512 for p < 2^254.2, which covers the four fields here; the reduced value is below 12 p (p / 2^256) + p < 5 p,
  // brought under p by conditional subtractions.)
  static VZ_HD Fp dot(const Fp* a, const Fp* b, int n) {
#if !defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned __int128 u128;
    uint64_t acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, M[4];
    __builtin_memcpy(M, P::MOD.w, 32);
    for (int k = 0; k < n; k++) {
      uint64_t A[4], B[4];
      __builtin_memcpy(A, a[k].v, 32); __builtin_memcpy(B, b[k].v, 32);
      for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)A[j] * B[i] + acc[i + j]; acc[i + j] = (uint64_t)c; c >>= 64; }
        for (int j = i + 4; j < 9 && c; j++) { c += acc[j]; acc[j] = (uint64_t)c; c >>= 64; }
      }
    }
    for (int i = 0; i < 4; i++) {
      const uint64_t m = acc[i] * P::N0_64;
      u128 c = 0;
      for (int j = 0; j < 4; j++) { c += (u128)m * M[j] + acc[i + j]; acc[i + j] = (uint64_t)c; c >>= 64; }
      for (int j = i + 4; j < 9; j++) { c += acc[j]; acc[j] = (uint64_t)c; c >>= 64; }
    }
    uint64_t r[4] = {acc[4], acc[5], acc[6], acc[7]};      // < 5 p < 2^257: acc[8] carries at most one bit
    uint64_t top = acc[8];
    for (int round = 0; round < 4; round++) {              // subtract p while the value is >= p (at most four times)
      uint64_t d[4], br = 0;
      for (int i = 0; i < 4; i++) { u128 x = (u128)r[i] - M[i] - br; d[i] = (uint64_t)x; br = (uint64_t)(x >> 64) & 1; }
      const uint64_t ge = top | (br ^ 1);                  // value >= p
      const uint64_t take = (uint64_t)0 - (ge ? 1 : 0);
      top = top - ((top && br) ? 1 : 0);                   // the borrow comes out of the top bit
      for (int i = 0; i < 4; i++) r[i] = (d[i] & take) | (r[i] & ~take);
    }
    Fp o; __builtin_memcpy(o.v, r, 32);
    return o;
#else
    Fp acc = zero();
    for (int k = 0; k < n; k++) acc = add(acc, mul(a[k], b[k]));
    return acc;
#endif
  }

  // canonical <-> Montgomery
  static VZ_HD Fp to_mont(const Fp& canon) { return mul(canon, r2()); }
  static VZ_HD Fp from_mont(const Fp& m) { Fp o = zero(); o.v[0] = 1; return mul(m, o); }

  static VZ_HD Fp pow_pm2(const Fp& a) {  // a^(p-2): inverse by Fermat (0 -> 0)
    uint32_t e[8]; uint64_t br = 2;
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)P::MOD.w[i] - br; e[i] = (uint32_t)d; br = (d >> 32) & 1; }
    Fp acc = one();
    for (int i = 255; i >= 0; i--) {
      acc = sqr(acc);
      if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, a);
    }
    return acc;
  }
};

}  // namespace vz
