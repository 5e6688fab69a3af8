// ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported/linked by the product path
// (vimz_amd/…); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use anything in oracle/.
//
// 256-bit prime-field arithmetic, Montgomery form with R = 2^256, 4 x 64-bit
// little-endian limbs — the representation halo2curves 0.1.0 / pasta_curves 0.5.1
// use for the fields nova-snark 0.23.0 folds over (reference call sites:
// vimz/src/nova_snark_backend/mod.rs:19-20 `G1 = bn256::Point`, `G2 = grumpkin::Point`;
// SURVEY.md Appendix E lists the moduli, taken from contracts/ContrastVerifier.sol:35-38).
// Those crates are NOT vendored under /root/reference (vimz/Cargo.lock:2711,3958); this
// restates the textbook CIOS Montgomery algorithm they implement.
//
// Deliberately different from the product's device code (8 x 32-bit limbs,
// vimz_amd/csrc/fp.hpp): this one uses unsigned __int128 on 64-bit limbs.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

namespace orc {

typedef uint64_t u64;
typedef unsigned __int128 u128;

struct FieldParams {
  u64 p[4];    // modulus
  u64 r1[4];   // R mod p   (Montgomery one)
  u64 r2[4];   // R^2 mod p
  u64 n0;      // -p^{-1} mod 2^64
  int bits;    // bit length of p
};

static inline int cmp4(const u64* a, const u64* b) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] < b[i]) return -1;
    if (a[i] > b[i]) return 1;
  }
  return 0;
}
static inline u64 add4(u64* o, const u64* a, const u64* b) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; o[i] = (u64)c; c >>= 64; }
  return (u64)c;
}
static inline u64 sub4(u64* o, const u64* a, const u64* b) {
  u64 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - b[i] - br;
    o[i] = (u64)d; br = (u64)(d >> 64) & 1;
  }
  return br;
}

// Build params from a big-endian hex string of the modulus.
static inline void parse_hex(const char* hex, u64* out) {
  memset(out, 0, 32);
  if (hex[0] == '0' && (hex[1] == 'x' || hex[1] == 'X')) hex += 2;
  size_t n = strlen(hex);
  for (size_t i = 0; i < n; i++) {
    char c = hex[n - 1 - i];
    u64 v = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : c - 'A' + 10;
    out[i / 16] |= v << (4 * (i % 16));
  }
}

static inline FieldParams make_params(const char* hex) {
  FieldParams P;
  parse_hex(hex, P.p);
  // n0 = -p^{-1} mod 2^64 by Newton iteration
  u64 inv = 1;
  for (int i = 0; i < 6; i++) inv *= 2 - P.p[0] * inv;
  P.n0 = (u64)0 - inv;
  // R mod p: start from 1 and double 256 times mod p; R^2: 512 times.
  u64 x[4] = {1, 0, 0, 0};
  for (int i = 0; i < 512; i++) {
    u64 c = add4(x, x, x);
    u64 t[4];
    if (c || cmp4(x, P.p) >= 0) { sub4(t, x, P.p); memcpy(x, t, 32); }
    if (i == 255) memcpy(P.r1, x, 32);
  }
  memcpy(P.r2, x, 32);
  P.bits = 0;
  for (int i = 255; i >= 0; i--) if ((P.p[i / 64] >> (i % 64)) & 1) { P.bits = i + 1; break; }
  return P;
}

// Field element bound to a params tag.  Tag::P() returns const FieldParams&.
template <class Tag>
struct Fe {
  u64 l[4];  // Montgomery form

  static const FieldParams& P() { return Tag::P(); }

  static Fe zero() { Fe r; memset(r.l, 0, 32); return r; }
  static Fe one() { Fe r; memcpy(r.l, P().r1, 32); return r; }
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const Fe& o) const { return memcmp(l, o.l, 32) == 0; }
  bool operator!=(const Fe& o) const { return !(*this == o); }

  // CIOS Montgomery multiplication.
  static Fe mont_mul(const Fe& a, const Fe& b) {
    const FieldParams& F = P();
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < 4; j++) {
        c += (u128)a.l[j] * b.l[i] + t[j];
        t[j] = (u64)c; c >>= 64;
      }
      c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
      u64 m = t[0] * F.n0;
      c = (u128)m * F.p[0] + t[0]; c >>= 64;
      for (int j = 1; j < 4; j++) {
        c += (u128)m * F.p[j] + t[j];
        t[j - 1] = (u64)c; c >>= 64;
      }
      c += t[4]; t[3] = (u64)c; t[4] = t[5] + (u64)(c >> 64);
    }
    Fe r;
    if (t[4] || cmp4(t, F.p) >= 0) sub4(r.l, t, F.p); else memcpy(r.l, t, 32);
    return r;
  }
  Fe operator*(const Fe& o) const { return mont_mul(*this, o); }
  Fe sqr() const { return mont_mul(*this, *this); }
  Fe operator+(const Fe& o) const {
    Fe r; u64 c = add4(r.l, l, o.l);
    if (c || cmp4(r.l, P().p) >= 0) { u64 t[4]; sub4(t, r.l, P().p); memcpy(r.l, t, 32); }
    return r;
  }
  Fe operator-(const Fe& o) const {
    Fe r; u64 b = sub4(r.l, l, o.l);
    if (b) { u64 t[4]; add4(t, r.l, P().p); memcpy(r.l, t, 32); }
    return r;
  }
  Fe neg() const { return zero() - *this; }
  Fe dbl() const { return *this + *this; }

  // canonical (non-Montgomery) little-endian limbs in/out
  static Fe from_canonical(const u64* c) {
    Fe a; memcpy(a.l, c, 32);
    // reduce if >= p (inputs are expected < p, but be safe)
    while (cmp4(a.l, P().p) >= 0) { u64 t[4]; sub4(t, a.l, P().p); memcpy(a.l, t, 32); }
    Fe r2; memcpy(r2.l, P().r2, 32);
    return mont_mul(a, r2);
  }
  static Fe from_u64(u64 v) { u64 c[4] = {v, 0, 0, 0}; return from_canonical(c); }
  static Fe from_i64(int64_t v) { return v >= 0 ? from_u64((u64)v) : from_u64((u64)(-v)).neg(); }
  void to_canonical(u64* out) const {
    Fe o; o.l[0] = 1; o.l[1] = o.l[2] = o.l[3] = 0;
    Fe r = mont_mul(*this, o);
    memcpy(out, r.l, 32);
  }
  static Fe from_mont_limbs(const u64* m) { Fe a; memcpy(a.l, m, 32); return a; }

  Fe pow(const u64* e) const {  // e: 4 limbs canonical exponent
    Fe acc = one();
    for (int i = 255; i >= 0; i--) {
      acc = acc.sqr();
      if ((e[i / 64] >> (i % 64)) & 1) acc = acc * *this;
    }
    return acc;
  }
  Fe inv() const {  // Fermat; inv(0) = 0
    u64 e[4]; u64 two[4] = {2, 0, 0, 0};
    sub4(e, P().p, two);
    return pow(e);
  }
  Fe pow5() const { Fe x2 = sqr(); Fe x4 = x2.sqr(); return x4 * *this; }
};

// ---- the four fields -------------------------------------------
struct TagBnFr { static const FieldParams& P() { static FieldParams p = make_params("30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001"); return p; } };
struct TagBnFq { static const FieldParams& P() { static FieldParams p = make_params("30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47"); return p; } };
struct TagPallasFp { static const FieldParams& P() { static FieldParams p = make_params("40000000000000000000000000000000224698fc094cf91b992d30ed00000001"); return p; } };
struct TagVestaFq { static const FieldParams& P() { static FieldParams p = make_params("40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001"); return p; } };

typedef Fe<TagBnFr> BnFr;
typedef Fe<TagBnFq> BnFq;
typedef Fe<TagPallasFp> PallasFp;
typedef Fe<TagVestaFq> VestaFq;

}  // namespace orc
