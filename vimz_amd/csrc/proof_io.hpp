// Host-side plumbing shared by the proof objects that leave the library (spartan.hip: the compressed proof; merge.hip: the merged
// proof of several row segments): the SHA3-256 transcript chain, word-stream writer / reader with range and curve checks on
// everything read from an untrusted blob, host scalar multiplication.
#pragma once
#include "ivc_internal.hpp"

namespace {

// ---- transcript: SHA3-256 chain ---------------------------------------------------------------------------------------------------
struct Transcript {
  uint8_t st[32];
  explicit Transcript(const char* label) { memset(st, 0, 32); absorb_bytes("init", label, strlen(label)); }
  void absorb_bytes(const char* tag, const void* data, size_t n) {
    Sha3 h; h.update(st, 32);
    uint8_t t[8] = {0}; for (int i = 0; i < 8 && tag[i]; i++) t[i] = (uint8_t)tag[i];
    h.update(t, 8);
    const uint64_t len = n; h.update(&len, 8);
    if (n) h.update(data, n);
    h.finish(st);
  }
  void absorb_words(const char* tag, const uint64_t* w, size_t nwords) { absorb_bytes(tag, w, 8 * nwords); }
  template <class F> void absorb_fe(const char* tag, const F& mont) { F c = F::from_mont(mont); absorb_bytes(tag, c.v, 32); }
  void challenge(uint32_t out[4]) {               // 128 bits
    Sha3 h; h.update(st, 32); const uint8_t c = 'c'; h.update(&c, 1); h.finish(st);
    memcpy(out, st, 16);
  }
  template <class F> F challenge_fe() { uint32_t w[4]; challenge(w); F c = F::zero(); for (int i = 0; i < 4; i++) c.v[i] = w[i]; return F::to_mont(c); }
};

// ---- serialisation helpers -------------------------------------------------------------------------------------------------------------
struct Writer {
  std::vector<uint64_t> w;
  template <class F> void fe(const F& mont) { F c = F::from_mont(mont); const size_t o = w.size(); w.resize(o + 4); memcpy(&w[o], c.v, 32); }
  void u256(const U256w& x) { w.insert(w.end(), x.w, x.w + 4); }
  void word(uint64_t x) { w.push_back(x); }
  template <class F> void point(const Affine<F>& p) { fe(p.x); fe(p.y); }
};
struct Reader {
  const uint64_t* w; size_t n, pos = 0; bool ok = true;
  bool need(size_t k) { if (pos + k > n) ok = false; return ok; }
  template <class F> F fe() { F c = F::zero(); if (need(4)) { memcpy(c.v, w + pos, 32); pos += 4; if (!c.is_reduced()) { ok = false; return F::zero(); } } return F::to_mont(c); }
  U256w u256() { U256w x{}; if (need(4)) { memcpy(x.w, w + pos, 32); pos += 4; } return x; }
  uint64_t word() { uint64_t x = 0; if (need(1)) x = w[pos++]; return x; }
  // (an untrusted point must be on its curve: the addition formulas do not depend on b)
  template <class F> Affine<F> point() { Affine<F> p; p.x = fe<F>(); p.y = fe<F>(); if (ok && !aff_on_curve(p)) { ok = false; p.x = p.y = F::zero(); } return p; }
};

// host scalar·point on curve C (scalar: canonical little-endian words)
template <class FS>
XYZZ<FS> host_mul(const Affine<FS>& p, const uint32_t* k, int bits) {
  XYZZ<FS> acc = XYZZ<FS>::identity();
  for (int i = bits - 1; i >= 0; i--) { acc = dbl(acc); if ((k[i >> 5] >> (i & 31)) & 1) add_mixed(acc, p); }
  return acc;
}
template <class FS, class F>
XYZZ<FS> host_mul_fe(const Affine<FS>& p, const F& k_mont) { F c = F::from_mont(k_mont); return host_mul<FS>(p, c.v, 256); }

}  // namespace
