// The two checks that make the decider FULL — what Sonobe's `DeciderEth` circuit does with the CycleFold instance and the reference's opt-in
// `light-test` feature leaves out (vimz/src/sonobe_backend/decider.rs:13-21 instantiates the full one; vimz/Cargo.toml:56-59, vimz/Makefile:1-2 and
// contracts/light-test/*.sol are the shrunken variant).  Sonobe (folding-schemes @ d312916) is not vendored: the CONSTRAINTS below are our statement of
// the two checks its documentation lists for `cf_U_i` — parity unpinned, like aug/decider.hpp; the public inputs and the 25 words are untouched.
//
//   5. cfU_i.cmW = Σ_k cfW_k · G_k   and   cfU_i.cmE = Σ_k cfE_k · G_k             the running CycleFold instance's two Pedersen commitments, opened
//      Grumpkin's coordinates live in Fr: NATIVE arithmetic.  Every scalar (an element of Fq, 254 bits) is given by its bits; a fixed-base sum in
//      2-bit windows: entry (d + 1)·4^j·G_k chosen by the two bits (one product, the coordinates are then linear in b0, b1, b0·b1), added to the
//      scalar's accumulator by an affine addition (3 rows): 3 rows per bit with the bit's own.  Accumulators start at an independent generator H
//      (per scalar) / H2 (the sum over scalars), so that an addition's two operands can only share their x for a prover who knows a discrete-log
//      relation among (H, H2, G_k) — the assumption the commitment's binding rests on anyway; the constant H2 + n·H + Σ_j 4^j·ΣG_k is taken off
//      at the end.  H, H2: hash-derived like the key itself (ckgen_impl.hpp), labels "vimz-decider-H" 0 and 1.
//   6. (A·z) ∘ (B·z) = u·(C·z) + E  over Fq for z = (u, cfW, x)                     the running CycleFold instance's relaxed R1CS, row by row
//      NON-native.  Every element is an integer given by bits (those of check 5) in limbs of 90 bits.  One row: signed small coefficients give the
//      three columns of A, B, C as linear combinations; X = A·B − u·C − E − k·q = 0 over the integers is shown by the Chinese remainder theorem —
//      modulo the circuit's own prime r (one product of the native images) and modulo 2^270 (the three low columns of the limb products with
//      range-checked carries); |X| < r·2^270 by the bounds noted at each step (computed from the shape's coefficients, not assumed).  A linear
//      combination with large coefficients (the 128-term bit sums of the CycleFold circuit: four rows of 1 313) is first reduced modulo q the same way.
//      ≈ 570 rows per row of the CycleFold shape.
// Together ≈ 2.75 M constraints on top of the light circuit, whatever the step circuit.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <mutex>
#include <string>
#include <thread>
#include "cyclefold.hpp"
#include "../ckgen_impl.hpp"

namespace vz {
namespace aug {

constexpr int CFO_BITS = 254, CFO_WINDOWS = 127;      // scalars of Grumpkin: 254 bits, 2-bit windows
constexpr int NN_L = 90;                               // limb width of the non-native relation

// ---- what the full decider knows of the CycleFold commitment key -------------------------------------------------------------------------------
struct CfOpeningKey {
  uint32_t n = 0;                                      // generators in use (max of the witness and error vector lengths)
  std::vector<Affine<CfFr>> table;                     // [k][j][d] = (d + 1) · 4^j · G_k
  Affine<CfFr> H, H2;
  const Affine<CfFr>& entry(uint32_t k, int j, int d) const { return table[((size_t)k * CFO_WINDOWS + j) * 4 + d]; }
  static Affine<CfFr> derived_generator(uint64_t idx) {
    static const SqrtParams sp = sqrt_params<CfFr>();
    CkLabel L; memset(&L, 0, sizeof(L));
    static const char tag[] = "vimz-decider-H";
    memcpy(L.bytes, tag, sizeof(tag) - 1); L.len = (int)sizeof(tag) - 1;
    Affine<CfFr> p; ckgen_point<CfFr>(L, sp, CurveB<BnFr>::value, idx, &p.x, &p.y);
    return p;
  }
  static void batch_affine(const std::vector<XYZZ<CfFr>>& in, Affine<CfFr>* out) {
    std::vector<CfFr> zi(in.size());
    for (size_t i = 0; i < in.size(); i++) zi[i] = in[i].ZZZ;
    batch_inv(zi);
    for (size_t i = 0; i < in.size(); i++) {
      if (in[i].is_identity()) { out[i].x = out[i].y = CfFr::zero(); continue; }
      const CfFr zi2 = CfFr::sqr(CfFr::mul(zi[i], in[i].ZZ));
      out[i].x = CfFr::mul(in[i].X, zi2); out[i].y = CfFr::mul(in[i].Y, zi[i]);
    }
  }
  void build(const Affine<CfFr>* gens, uint32_t count) {
    n = count;
    H = derived_generator(0); H2 = derived_generator(1);
    table.assign((size_t)n * CFO_WINDOWS * 4, Affine<CfFr>());
    const unsigned T = std::max(1u, std::min(16u, affinity_cpus()));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++) th.emplace_back([&, t] {
      std::vector<XYZZ<CfFr>> pts((size_t)CFO_WINDOWS * 4);
      for (uint32_t k = t; k < n; k += T) {
        XYZZ<CfFr> B = from_affine(gens[k]);
        for (int j = 0; j < CFO_WINDOWS; j++) {
          XYZZ<CfFr> p2 = dbl(B), p3 = p2; add_full(p3, B);
          XYZZ<CfFr> p4 = dbl(p2);
          pts[4 * j] = B; pts[4 * j + 1] = p2; pts[4 * j + 2] = p3; pts[4 * j + 3] = p4;
          B = p4;
        }
        batch_affine(pts, &table[(size_t)k * CFO_WINDOWS * 4]);
      }
    });
    for (auto& x : th) x.join();
  }
  // H2 + cnt·H + Σ_{k<cnt} Σ_j 4^j·G_k: what the sum of cnt scalars' chains holds beyond Σ s_k·G_k
  Affine<CfFr> offset(uint32_t cnt) const {
    std::lock_guard<std::mutex> g(off_mu);
    for (auto& o : offs) if (o.first == cnt) return o.second;
    XYZZ<CfFr> acc = from_affine(H2);
    for (uint32_t k = 0; k < cnt; k++) { add_mixed(acc, H); for (int j = 0; j < CFO_WINDOWS; j++) add_mixed(acc, entry(k, j, 0)); }
    offs.emplace_back(cnt, to_affine(acc));
    return offs.back().second;
  }
  mutable std::mutex off_mu; mutable std::vector<std::pair<uint32_t, Affine<CfFr>>> offs;      // (computed once per length: 127 additions per scalar)
};

// ---- 320-bit wrap-around integers: the low words of an exact quotient -------------------------------------------------------------------------
struct W5 {
  uint64_t w[5];
  static W5 zero() { W5 r; for (auto& x : r.w) x = 0; return r; }
  static W5 from256(const U256w& v) { W5 r; for (int i = 0; i < 4; i++) r.w[i] = v.w[i]; r.w[4] = 0; return r; }
  static W5 add(const W5& a, const W5& b) { W5 r; unsigned __int128 c = 0; for (int i = 0; i < 5; i++) { c += (unsigned __int128)a.w[i] + b.w[i]; r.w[i] = (uint64_t)c; c >>= 64; } return r; }
  static W5 neg(const W5& a) { W5 r; unsigned __int128 c = 1; for (int i = 0; i < 5; i++) { c += (unsigned __int128)(~a.w[i]); r.w[i] = (uint64_t)c; c >>= 64; } return r; }
  static W5 sub(const W5& a, const W5& b) { return add(a, neg(b)); }
  static W5 mul(const W5& a, const W5& b) {
    W5 r = zero();
    for (int i = 0; i < 5; i++) { unsigned __int128 c = 0; for (int j = 0; i + j < 5; j++) { c += (unsigned __int128)a.w[i] * b.w[j] + r.w[i + j]; r.w[i + j] = (uint64_t)c; c >>= 64; } }
    return r;
  }
  static W5 pow2(int k) { W5 r = zero(); if (k < 320) r.w[k >> 6] = 1ull << (k & 63); return r; }
  bool bit(int k) const { return k < 320 && ((w[k >> 6] >> (k & 63)) & 1u); }
  // the inverse of an odd number modulo 2^320 (Newton)
  static W5 inv_odd(const W5& m) { W5 x = zero(); x.w[0] = 1; W5 two = zero(); two.w[0] = 2; for (int it = 0; it < 9; it++) x = mul(x, sub(two, mul(m, x))); return x; }
};

// ---- the gadgets ----------------------------------------------------------------------------------------------------------------------------------
struct CfFullIn {
  const CfOpeningKey* key = nullptr;
  const cb::BuilderT<CfFq>* shape = nullptr;           // the CycleFold circuit
  const CfFq* W = nullptr;                              // the running CycleFold witness: wires 1 .. n_w-1-CF_IO of Z (Montgomery); nullptr: shape mode / zeros
  const CfFq* E = nullptr;                              // its error vector
};

struct DeciderCfGadget {
  typedef CfFr F;
  typedef Num<F> N;
  typedef cb::LCT<F> LC;
  CS<BnFr>& cs;
  explicit DeciderCfGadget(CS<BnFr>& c) : cs(c) {}
  bool shape() const { return cs.shape(); }

  N bool_wire(bool v) {
    N b = cs.alloc(v ? F::one() : F::zero());
    if (cs.b) cs.b->enforce(b.lc, b.lc - LC::constant(F::one()), LC());
    return b;
  }
  static bool bit_of(const U256w& v, int k) { return k < 256 && ((v.w[k >> 6] >> (k & 63)) & 1u); }

  // ---- check 5: Σ_k s_k·G_k from the scalars' bits (bits[k*254 + i]); returns the sum's chain value  H2 + cnt·H + Σ(s_k + off)·G_k  as (x, y) ----
  struct XY { N x, y; };
  struct ChainTemplate { bool have = false; uint32_t row0 = 0, bits0 = 0, chain0 = 0, cnt = 0; } tmpl;      // shape mode: the first opening's rows, to copy from
  XY open_chains(const CfOpeningKey& key, const std::vector<N>& bits, const U256w* vals /* witness mode: the scalars */, uint32_t cnt) {
    const bool sh = shape();
    // witness mode: all the chains' values at once, one batched inversion per window
    std::vector<F> wv;                                   // [k][j][4]: product, slope, x, y
    std::vector<Affine<F>> S(cnt);
    if (!sh) {
      wv.resize((size_t)cnt * CFO_WINDOWS * 4);
      std::vector<Affine<F>> acc(cnt, key.H);
      std::vector<F> den(cnt);
      for (int j = 0; j < CFO_WINDOWS; j++) {
        for (uint32_t k = 0; k < cnt; k++) {
          const int d = (int)bit_of(vals[k], 2 * j) + 2 * (int)bit_of(vals[k], 2 * j + 1);
          den[k] = F::sub(key.entry(k, j, d).x, acc[k].x);
          if (den[k].is_zero()) cs.bad = true;
        }
        batch_inv(den);
        for (uint32_t k = 0; k < cnt; k++) {
          const bool b0 = bit_of(vals[k], 2 * j), b1 = bit_of(vals[k], 2 * j + 1);
          const Affine<F>& q = key.entry(k, j, (int)b0 + 2 * (int)b1);
          const F lam = F::mul(F::sub(q.y, acc[k].y), den[k]);
          Affine<F> r;
          r.x = F::sub(F::sub(F::sqr(lam), acc[k].x), q.x);
          r.y = F::sub(F::mul(lam, F::sub(acc[k].x, r.x)), acc[k].y);
          F* o = &wv[((size_t)k * CFO_WINDOWS + j) * 4];
          o[0] = (b0 && b1) ? F::one() : F::zero(); o[1] = lam; o[2] = r.x; o[3] = r.y;
          acc[k] = r;
        }
      }
      S = acc;
    }
    std::vector<XY> Sn(cnt);
    for (uint32_t k = 0; k < cnt; k++) {
      if (!sh) {      // the chain's wires, in the order the shape allocates them
        cs.w.insert(cs.w.end(), wv.begin() + (size_t)k * CFO_WINDOWS * 4, wv.begin() + (size_t)(k + 1) * CFO_WINDOWS * 4);
        Sn[k].x.v = S[k].x; Sn[k].y.v = S[k].y;
        continue;
      }
      constexpr uint32_t PER = 4 * CFO_WINDOWS;      // rows and wires of one scalar's chain: (product, slope, x, y) per window
      const uint32_t bits_k = bits[(size_t)k * CFO_BITS].lc.t[0].w;
      static const bool no_copy = getenv("VIMZ_DEBUG_DECIDER_NO_ROW_COPY") != nullptr;      // (the long way, for comparing the two)
      if (tmpl.have && k < tmpl.cnt && !no_copy) {
        // The same scalar index was opened before (the witness opening precedes the error vector's over the same generators): its chain's rows are the
        // same rows over other wires — copied with the wire numbers moved instead of synthesised again (0.8 s of a 3.3 s set-up)
        cb::BuilderT<F>& B = *cs.b;
        const uint32_t nb0 = B.alloc(PER);
        if (nb0 != cs.base + (uint32_t)cs.w.size()) throw std::runtime_error("decider: wire allocation out of step");
        cs.w.insert(cs.w.end(), PER, F::zero());
        const uint32_t ob = tmpl.bits0 + (uint32_t)CFO_BITS * k, oc = tmpl.chain0 + PER * k;
        auto move = [&](uint32_t w) -> uint32_t {
          if (w == 0) return 0;
          if (w >= ob && w < ob + (uint32_t)CFO_BITS) return w - ob + bits_k;
          if (w >= oc && w < oc + PER) return w - oc + nb0;
          throw std::runtime_error("decider: a chain row reaches outside its scalar");
        };
        for (uint32_t r = tmpl.row0 + PER * k; r < tmpl.row0 + PER * (k + 1); r++)
          for (cb::Csr* M : {&B.A, &B.B, &B.C}) {
            const uint32_t lo = M->row_ptr[r], hi = M->row_ptr[r + 1];
            for (uint32_t q = lo; q < hi; q++) { M->col.push_back(move(M->col[q])); M->coef.push_back(M->coef[q]); }
            M->row_ptr.push_back((uint32_t)M->col.size());
          }
        Sn[k].x = cs.wire(nb0 + PER - 2, F::zero()); Sn[k].y = cs.wire(nb0 + PER - 1, F::zero());
        continue;
      }
      if (k == 0 && !tmpl.have) { tmpl.row0 = cs.b->n_constraints(); tmpl.bits0 = bits_k; tmpl.chain0 = cs.base + (uint32_t)cs.w.size(); }
      if (!tmpl.have && !no_copy && parallel_first_opening(key, bits, cnt, Sn)) break;      // (every scalar's chain synthesised on the host's threads and merged in order)
      Sn[k] = scalar_chain(cs, key, k, &bits[(size_t)k * CFO_BITS]);
    }
    if (sh && !tmpl.have) { tmpl.have = true; tmpl.cnt = cnt; }
    // the sum over the scalars
    XY tot; tot.x = cs.constant(key.H2.x); tot.y = cs.constant(key.H2.y);
    for (uint32_t k = 0; k < cnt; k++) tot = add_incomplete(tot, Sn[k]);
    return tot;
  }
  // P + Q for two finite points with different x (3 rows; the slope multiplies from the right: it is the only full-size wire of the B matrix)
  XY add_incomplete(const XY& p, const XY& q) { return add_incomplete(cs, p, q); }
  static XY add_incomplete(CS<BnFr>& c, const XY& p, const XY& q) {
    F lamv = F::zero();
    if (!c.b) {
      const F d = F::sub(q.x.v, p.x.v);
      if (d.is_zero()) c.bad = true;
      lamv = F::mul(F::sub(q.y.v, p.y.v), F::pow_pm2(d));
    }
    N l = c.alloc(lamv);
    c.enforce(c.sub(q.x, p.x), l, c.sub(q.y, p.y));
    XY r;
    r.x = c.alloc(F::sub(F::sub(F::sqr(lamv), p.x.v), q.x.v));
    c.enforce(l, l, c.add(r.x, c.add(p.x, q.x)));
    r.y = c.alloc(F::sub(F::mul(lamv, F::sub(p.x.v, r.x.v)), p.y.v));
    c.enforce(c.sub(p.x, r.x), l, c.add(r.y, p.y));
    return r;
  }
  // shape mode: the chain of scalar k — per window the product of its two bits, the table entry as a linear combination, the addition — appended to c
  static XY scalar_chain(CS<BnFr>& c, const CfOpeningKey& key, uint32_t k, const N* kbits) {
    XY acc; acc.x = c.constant(key.H.x); acc.y = c.constant(key.H.y);
    for (int j = 0; j < CFO_WINDOWS; j++) {
      const N& b0 = kbits[2 * j]; const N& b1 = kbits[2 * j + 1];
      N p = c.alloc(F::zero());
      c.enforce(b0, b1, p);
      const Affine<F>* e = &key.entry(k, j, 0);      // e[d], d = b0 + 2 b1
      auto lookup = [&](const F& c00, const F& c10, const F& c01, const F& c11) {
        N r = c.constant(c00);
        r = c.add(r, c.scale(b0, F::sub(c10, c00)));
        r = c.add(r, c.scale(b1, F::sub(c01, c00)));
        r = c.add(r, c.scale(p, F::sub(F::add(c11, c00), F::add(c10, c01))));
        r.konst = false;
        return r;
      };
      XY q; q.x = lookup(e[0].x, e[1].x, e[2].x, e[3].x); q.y = lookup(e[0].y, e[1].y, e[2].y, e[3].y);
      acc = add_incomplete(c, acc, q);
    }
    return acc;
  }
  // The first opening's chains on the host's threads (1.0 s of a 3 s set-up on one): every thread synthesises a run of scalars into a builder of its own — wire 0,
  // then the run's bits, then its chain wires — and the runs are merged in order: rows copied with the wire numbers moved, coefficients appended to the dictionary
  // (no search: a table constant occurs once).  Same rows, same wires, same coefficient VALUES as the sequential synthesis whatever the number of threads.
  bool parallel_first_opening(const CfOpeningKey& key, const std::vector<N>& bits, uint32_t cnt, std::vector<XY>& Sn) {
    const unsigned T = std::max(1u, std::min(16u, affinity_cpus()));
    if (T < 2 || cnt < 4 * T) return false;
    constexpr uint32_t PER = 4 * CFO_WINDOWS;
    struct Part { cb::BuilderT<F> b; uint32_t k0 = 0, k1 = 0; std::string err; };
    std::vector<Part> parts(T);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++) {
      parts[t].k0 = (uint32_t)((uint64_t)cnt * t / T); parts[t].k1 = (uint32_t)((uint64_t)cnt * (t + 1) / T);
      th.emplace_back([&, t] {
        Part& P = parts[t];
        try {
          const uint32_t n = P.k1 - P.k0;
          P.b.n_wires = 1 + n * (uint32_t)CFO_BITS;
          CS<BnFr> c; c.b = &P.b; c.base = P.b.n_wires;
          std::vector<N> kb((size_t)CFO_BITS);
          for (uint32_t k = P.k0; k < P.k1; k++) {
            for (int i = 0; i < CFO_BITS; i++) kb[i] = c.wire(1 + (k - P.k0) * (uint32_t)CFO_BITS + (uint32_t)i, F::zero());
            scalar_chain(c, key, k, kb.data());
          }
        } catch (const std::exception& e) { P.err = e.what(); }
      });
    }
    const double t_th0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    for (auto& x : th) x.join();
    if (getenv("VIMZ_DEBUG_TIMING")) fprintf(stderr, "[timing] decider: %u threads' chains joined after %.0f ms\n", T, 1e3 * (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_th0));
    for (auto& P : parts) if (!P.err.empty()) throw std::runtime_error(P.err);
    cb::BuilderT<F>& B = *cs.b;
    for (auto& P : parts) {
      const uint32_t n = P.k1 - P.k0, lbits = 1, lchain = 1 + n * (uint32_t)CFO_BITS;
      if (P.b.n_constraints() != n * PER || P.b.n_wires != lchain + n * PER) throw std::runtime_error("decider: a run of chains has an unexpected size");
      const uint32_t d0 = (uint32_t)B.dict.size();
      B.dict.insert(B.dict.end(), P.b.dict.begin(), P.b.dict.end());
      for (uint32_t k = P.k0; k < P.k1; k++) {
        const uint32_t j = k - P.k0, bits_k = bits[(size_t)k * CFO_BITS].lc.t[0].w;
        const uint32_t nb0 = B.alloc(PER);
        if (nb0 != cs.base + (uint32_t)cs.w.size()) throw std::runtime_error("decider: wire allocation out of step");
        cs.w.insert(cs.w.end(), PER, F::zero());
        const uint32_t ob = lbits + j * (uint32_t)CFO_BITS, oc = lchain + j * PER;
        auto move = [&](uint32_t w) -> uint32_t {
          if (w == 0) return 0;
          if (w >= ob && w < ob + (uint32_t)CFO_BITS) return w - ob + bits_k;
          if (w >= oc && w < oc + PER) return w - oc + nb0;
          throw std::runtime_error("decider: a chain row reaches outside its scalar");
        };
        for (uint32_t r = j * PER; r < (j + 1) * PER; r++) {
          const cb::Csr* Ms[3] = {&P.b.A, &P.b.B, &P.b.C}; cb::Csr* Md[3] = {&B.A, &B.B, &B.C};
          for (int m = 0; m < 3; m++) {
            for (uint32_t q = Ms[m]->row_ptr[r]; q < Ms[m]->row_ptr[r + 1]; q++) { Md[m]->col.push_back(move(Ms[m]->col[q])); Md[m]->coef.push_back(d0 + Ms[m]->coef[q]); }
            Md[m]->row_ptr.push_back((uint32_t)Md[m]->col.size());
          }
        }
        Sn[k].x = cs.wire(nb0 + PER - 2, F::zero()); Sn[k].y = cs.wire(nb0 + PER - 1, F::zero());
      }
    }
    return true;
  }

  // ---- check 6: non-native integers -------------------------------------------------------------------------------------------------------------
  struct Big { N l[3]; N nat; double lb = 0; U256w v; };      // limbs of NN_L bits (the last one shorter), the image modulo r, log2 of a bound, the integer (witness mode)
  N pack_bits(const std::vector<N>& bits, size_t at, int lo, int hi) {      // Σ_{i in [lo, hi)} 2^(i - lo) bits[at + i]
    N r = cs.zero(); r.konst = false;
    F val = F::zero();
    for (int i = hi - 1; i >= lo; i--) { val = F::dbl(val); val = F::add(val, bits[at + i].v); }
    r.v = val;
    if (cs.b) { LC acc; acc.t.reserve(hi - lo); for (int i = lo; i < hi; i++) { const LC& bl = bits[at + i].lc; if (bl.t.size() != 1) throw std::runtime_error("decider: a bit that is not a wire"); acc.t.push_back({bl.t[0].w, F::mul(bl.t[0].c, cb::f_pow2<F>(i - lo))}); } r.lc = acc; }
    return r;
  }
  // bits (all wires, ascending) -> limbs; wires = true: the limbs become wires of their own (for values many rows use)
  Big big_from_bits(const std::vector<N>& bits, size_t at, int nbits, const U256w& v, bool wires) {
    Big g; g.lb = nbits; g.v = v;
    for (int j = 0; j < 3; j++) {
      const int lo = NN_L * j, hi = std::min(NN_L * (j + 1), nbits);
      if (lo >= hi) { g.l[j] = cs.zero(); continue; }
      N pk = pack_bits(bits, at, lo, hi);
      if (wires) { N w = cs.alloc(pk.v); cs.enforce_equal(w, pk); g.l[j] = w; } else g.l[j] = pk;
    }
    g.nat = cs.add(g.l[0], cs.add(cs.scale(g.l[1], cb::f_pow2<F>(NN_L)), cs.scale(g.l[2], cb::f_pow2<F>(2 * NN_L))));
    return g;
  }
  static F f_from_w5_low(const W5& x, int lo, int hi) {      // bits [lo, hi) of x as a field element (hi - lo <= 128)
    F c = F::zero();
    for (int i = lo; i < hi; i++) if (x.bit(i)) c.v[(i - lo) >> 5] |= 1u << ((i - lo) & 31);
    return F::to_mont(c);
  }
  struct Plan { int nk, ncol; };
  // |X'| < 2^lb for X' = X + k·q: the quotient's size and how many low columns the 2-adic half of the argument needs
  static Plan plan(double lb) {
    Plan p;
    p.nk = (int)std::ceil(lb - 253.5) + 1;                     // q > 2^253.5: |k| < 2^(lb - 253.5) < 2^nk
    if (p.nk < 1) p.nk = 1;
    const double lbx = std::max(lb, (double)p.nk + 254.0) + 1.0;      // |X' − k·q| < 2^lbx
    p.ncol = (int)std::ceil((lbx - 253.5) / NN_L);            // r·2^(90·ncol) > 2^lbx   (r > 2^253.5)
    if (p.ncol < 1) p.ncol = 1;
    if (p.ncol > 3) throw std::runtime_error("decider: a row of the CycleFold shape is too large for the non-native check");
    return p;
  }
  const U256w qv = NonNative<BnFr, BnFq>::modulus();
  F q_limb(int j) const {      // limb j of q
    F c = F::zero();
    for (int i = NN_L * j; i < std::min(NN_L * (j + 1), 256); i++) if (bit_of(qv, i)) c.v[(i - NN_L * j) >> 5] |= 1u << ((i - NN_L * j) & 31);
    return F::to_mont(c);
  }
  // cols[j] (j < ncol), nat: the columns and the native image of X' (the part without the quotient), |X'| < 2^lb, |cols[j]| < 2^col_lb;
  // xlow = X' modulo 2^320 (witness mode).  Proves X' ≡ 0 (mod q): allocates the quotient, the carries.
  void finish_mod_q(N* cols, const N& nat, double lb, double col_lb, const Plan& pl, const W5& xlow) {
    const int nk = pl.nk, ncol = pl.ncol, nkb = nk + 1;
    if (nkb > 3 * NN_L) throw std::runtime_error("decider: quotient too wide");
    // k' = k + 2^nk in [0, 2^(nk+1))
    W5 kp = W5::zero();
    if (!cs.b) {
      static const W5 qinv = W5::inv_odd(W5::from256(NonNative<BnFr, BnFq>::modulus()));
      kp = W5::add(W5::mul(xlow, qinv), W5::pow2(nk));
      for (int i = nkb; i < 320; i++) if (kp.bit(i)) { cs.bad = true; kp = W5::zero(); break; }
    }
    std::vector<N> kb((size_t)nkb);
    for (int i = 0; i < nkb; i++) kb[i] = bool_wire(kp.bit(i));
    N kl[3], knat;
    for (int j = 0; j < 3; j++) { const int lo = NN_L * j, hi = std::min(NN_L * (j + 1), nkb); kl[j] = lo < hi ? pack_bits(kb, 0, lo, hi) : cs.zero(); }
    knat = cs.add(kl[0], cs.add(cs.scale(kl[1], cb::f_pow2<F>(NN_L)), cs.scale(kl[2], cb::f_pow2<F>(2 * NN_L))));
    // the constant 2^nk·q: its low columns and its native image
    const W5 cq = W5::mul(W5::pow2(nk), W5::from256(qv));      // modulo 2^320 >= 2^270: enough for three columns
    const F qnat = F::add(q_limb(0), F::add(F::mul(q_limb(1), cb::f_pow2<F>(NN_L)), F::mul(q_limb(2), cb::f_pow2<F>(2 * NN_L))));
    F p2nk = F::one(); for (int i = 0; i < nk; i++) p2nk = F::dbl(p2nk);
    // native: X' − k'·q + 2^nk·q = 0 (mod r)
    N nz = cs.add(cs.sub(nat, cs.scale(knat, qnat)), cs.constant(F::mul(p2nk, qnat)));
    cs.enforce_zero(nz);
    if (!cs.b && !nz.v.is_zero()) cs.bad = true;
    // 2-adic: column by column, carries by range check.  |column| < 2^col_lb + 3·2^180 + 2^90 + |carry in|
    static const F inv_limb = F::pow_pm2(cb::f_pow2<F>(NN_L));
    N carry = cs.zero();
    double carry_lb = -1e9;
    for (int j = 0; j < ncol; j++) {
      N t = cs.add(cols[j], carry);
      for (int m = 0; m <= j && m < 3; m++) { const int nq = j - m; if (nq < 3) t = cs.sub(t, cs.scale(kl[m], q_limb(nq))); }
      t = cs.add(t, cs.constant(f_from_w5_low(cq, NN_L * j, NN_L * (j + 1))));
      const double mag = std::log2(std::exp2(col_lb) + 3 * std::exp2(2.0 * NN_L) + std::exp2((double)NN_L) + (carry_lb > 0 ? std::exp2(carry_lb) : 0.0));
      if (mag > 250.0) throw std::runtime_error("decider: a column of the non-native check does not fit the field");
      const int cbits = (int)std::ceil(mag - NN_L) + 1;      // |carry| < 2^cbits
      N c = cs.scale(t, inv_limb);                         // the carry, as a linear combination: t = c·2^90 exactly when the low limb vanishes
      cs.bits(cs.addc(c, cb::f_pow2<F>(cbits)), cbits + 1);
      carry = c; carry_lb = cbits;
    }
  }

  // ---- the relation ----------------------------------------------------------------------------------------------------------------------------
  struct Coef { bool small; int64_t s; bool neg; U256w mag; int bits; F nat; };      // a dictionary coefficient as a signed integer
  static Coef coef_of(const CfFq& m) {
    Coef c;
    const CfFq pos = CfFq::from_mont(m), ng = CfFq::from_mont(CfFq::neg(m));
    auto nbits = [](const CfFq& y) { for (int k = 255; k >= 0; k--) if ((y.v[k >> 5] >> (k & 31)) & 1u) return k + 1; return 0; };
    const int bp = nbits(pos), bn = nbits(ng);
    c.neg = bn < bp;
    const CfFq& a = c.neg ? ng : pos;
    c.bits = c.neg ? bn : bp;
    for (int i = 0; i < 4; i++) c.mag.w[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
    c.small = c.bits <= 10;
    c.s = c.small ? (c.neg ? -(int64_t)c.mag.w[0] : (int64_t)c.mag.w[0]) : 0;
    // the image modulo r (|a| may exceed r: by halves)
    F lo = F::zero(), hi = F::zero();
    for (int i = 0; i < 4; i++) { lo.v[i] = a.v[i]; hi.v[i] = a.v[4 + i]; }
    c.nat = F::add(F::to_mont(lo), F::mul(F::to_mont(hi), cb::f_pow2<F>(128)));
    if (c.neg) c.nat = F::neg(c.nat);
    return c;
  }
  struct RowLc { N col[3]; N nat; double sum_abs = 0; double lb = 0; W5 val; };      // columns Σ a_i z_i[j], image, Σ|a_i|, log2 bound, value modulo 2^320
  static W5 w5_signed(const W5& x, bool neg) { return neg ? W5::neg(x) : x; }

  void relation(const cb::BuilderT<CfFq>& sh, const std::vector<Big>& z, const std::vector<N>& ebits, const CfFq* Ev) {
    const uint32_t nc = sh.n_constraints();
    std::vector<Coef> dict(sh.dict.size());
    for (size_t i = 0; i < dict.size(); i++) dict[i] = coef_of(sh.dict[i]);
    const Big& u = z[0];
    auto row_lc = [&](const cb::Csr& M, uint32_t r) {
      RowLc o; o.val = W5::zero();
      const uint32_t lo = M.row_ptr[r], hi = M.row_ptr[r + 1];
      bool all_small = true;
      for (uint32_t k = lo; k < hi; k++) all_small = all_small && dict[M.coef[k]].small;
      if (all_small) {
        for (int j = 0; j < 3; j++) { o.col[j] = cs.zero(); o.col[j].konst = false; }
        o.nat = cs.zero(); o.nat.konst = false;
        double bound = 0;
        for (uint32_t k = lo; k < hi; k++) {
          const Coef& c = dict[M.coef[k]]; const Big& zz = z[M.col[k]];
          const F cf = cb::f_from_i64<F>(c.s);
          for (int j = 0; j < 3; j++) o.col[j] = cs.add(o.col[j], cs.scale(zz.l[j], cf));
          o.nat = cs.add(o.nat, cs.scale(zz.nat, cf));
          o.sum_abs += std::fabs((double)c.s);
          bound += std::fabs((double)c.s) * std::exp2(zz.lb);
          if (!cs.b) { W5 t = W5::zero(); t.w[0] = (uint64_t)std::llabs(c.s); o.val = W5::add(o.val, w5_signed(W5::mul(t, W5::from256(zz.v)), c.s < 0)); }
        }
        o.lb = hi > lo ? std::log2(bound) : -1e9;
        return o;
      }
      // large coefficients: v = Σ a_i z_i mod q as a fresh 256-bit integer, shown by the same argument (no products: the coefficients are constants)
      double bound = std::exp2(256.0);
      for (uint32_t k = lo; k < hi; k++) bound += std::exp2((double)dict[M.coef[k]].bits + z[M.col[k]].lb);
      const double lb = std::log2(bound);
      const Plan pl = plan(lb);
      CfFq vq = CfFq::zero();
      W5 xlow = W5::zero();
      N cols[3], nat = cs.zero(); nat.konst = false;
      for (int j = 0; j < 3; j++) { cols[j] = cs.zero(); cols[j].konst = false; }
      for (uint32_t k = lo; k < hi; k++) {
        const Coef& c = dict[M.coef[k]]; const Big& zz = z[M.col[k]];
        // limbs of |a|
        F al[3];
        for (int m = 0; m < 3; m++) { F t = F::zero(); for (int i = NN_L * m; i < std::min(NN_L * (m + 1), 256); i++) if (bit_of(c.mag, i)) t.v[(i - NN_L * m) >> 5] |= 1u << ((i - NN_L * m) & 31); al[m] = F::to_mont(t); if (c.neg) al[m] = F::neg(al[m]); }
        for (int j = 0; j < pl.ncol; j++) for (int m = 0; m <= j; m++) if (!al[m].is_zero()) cols[j] = cs.add(cols[j], cs.scale(zz.l[j - m], al[m]));
        nat = cs.add(nat, cs.scale(zz.nat, c.nat));
        if (!cs.b) {
          vq = CfFq::add(vq, CfFq::mul(sh.dict[M.coef[k]], from_u256<CfFq>(reduce_q(zz.v))));
          xlow = W5::add(xlow, w5_signed(W5::mul(W5::from256(c.mag), W5::from256(zz.v)), c.neg));
        }
      }
      const U256w vv = cs.b ? U256w{{0, 0, 0, 0}} : to_u256(vq);
      std::vector<N> vb(256);
      for (int i = 0; i < 256; i++) vb[i] = bool_wire(bit_of(vv, i));
      Big v = big_from_bits(vb, 0, 256, vv, false);
      for (int j = 0; j < pl.ncol; j++) cols[j] = cs.sub(cols[j], v.l[j]);
      nat = cs.sub(nat, v.nat);
      xlow = W5::sub(xlow, W5::from256(vv));
      finish_mod_q(cols, nat, lb, std::log2(3.0 * (hi - lo) + 1.0) + 2.0 * NN_L, pl, xlow);
      for (int j = 0; j < 3; j++) o.col[j] = v.l[j];
      o.nat = v.nat; o.sum_abs = 1; o.lb = 256; o.val = W5::from256(vv);
      return o;
    };
    for (uint32_t r = 0; r < nc; r++) {
      RowLc A = row_lc(sh.A, r), B = row_lc(sh.B, r);
      const bool hasC = sh.C.row_ptr[r + 1] > sh.C.row_ptr[r];
      RowLc Cc; if (hasC) Cc = row_lc(sh.C, r);
      const U256w ev = (cs.b || !Ev) ? U256w{{0, 0, 0, 0}} : to_u256(Ev[r]);
      Big e = big_from_bits(ebits, (size_t)r * CFO_BITS, CFO_BITS, ev, false);
      const double lb = std::log2(std::exp2(A.lb + B.lb) + (hasC ? std::exp2(u.lb + Cc.lb) : 0.0) + std::exp2((double)CFO_BITS));
      const Plan pl = plan(lb);
      // the low columns of A·B − u·C − E
      N cols[3];
      for (int j = 0; j < pl.ncol; j++) {
        N t = cs.neg(e.l[j]); t.konst = false;
        for (int m = 0; m <= j; m++) {
          t = cs.add(t, cs.mul(A.col[m], B.col[j - m]));
          if (hasC) t = cs.sub(t, cs.mul(u.l[m], Cc.col[j - m]));
        }
        cols[j] = t;
      }
      N nat = cs.sub(cs.mul(A.nat, B.nat), e.nat);
      if (hasC) nat = cs.sub(nat, cs.mul(u.nat, Cc.nat));
      W5 xlow = W5::zero();
      if (!cs.b) { xlow = W5::sub(W5::mul(A.val, B.val), W5::from256(ev)); if (hasC) xlow = W5::sub(xlow, W5::mul(W5::from256(u.v), Cc.val)); }
      const double col_lb = std::log2(3.0 * (A.sum_abs * B.sum_abs + (hasC ? Cc.sum_abs : 0.0)) + 1.0) + 2.0 * NN_L;
      finish_mod_q(cols, nat, lb, col_lb, pl, xlow);
    }
  }
  U256w reduce_q(const U256w& v) const {      // any 256-bit integer modulo q
    U256w a = v;
    auto geq = [&](const U256w& x) { for (int i = 3; i >= 0; i--) if (x.w[i] != qv.w[i]) return x.w[i] > qv.w[i]; return true; };
    while (geq(a)) { unsigned __int128 br = 0; for (int i = 0; i < 4; i++) { const unsigned __int128 d = (unsigned __int128)a.w[i] - qv.w[i] - (uint64_t)br; a.w[i] = (uint64_t)d; br = (d >> 64) & 1; } }
    return a;
  }

  // ---- both checks, appended to the light circuit.  cu, cx, cW*, cE*: the wires of cfU_i the light circuit hashes (aug/decider.hpp) -----------------
  void synthesize(const CfFullIn& in, const N& cu, const N cx[CF_IO][4], const N& cWx, const N& cWy, const N& cEx, const N& cEy, const CfRelaxed& cfU) {
    typedef EcGadgets<BnFr> Ec;
    const cb::BuilderT<CfFq>& sh = *in.shape;
    const uint32_t nW = sh.n_wires - 1 - CF_IO, nE = sh.n_constraints();
    if (in.key->n < std::max(nW, nE)) throw std::runtime_error("decider: the opening key holds fewer generators than the CycleFold vectors need");
    const bool have = !cs.b && in.W && in.E;
    static const bool dbg_t = getenv("VIMZ_DEBUG_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_l = now();
    auto lap = [&](const char* what) { if (dbg_t) { const double t = now(); fprintf(stderr, "[timing] decider checks 5-6 (%s): %s %.0f ms\n", cs.b ? "shape" : "witness", what, 1e3 * (t - t_l)); t_l = t; } };
    // the scalars and their bits
    std::vector<U256w> wv(nW, U256w{{0, 0, 0, 0}}), ev(nE, U256w{{0, 0, 0, 0}});
    if (have) { for (uint32_t k = 0; k < nW; k++) wv[k] = to_u256(in.W[k]); for (uint32_t k = 0; k < nE; k++) ev[k] = to_u256(in.E[k]); }
    std::vector<N> wbits((size_t)nW * CFO_BITS), ebits((size_t)nE * CFO_BITS);
    for (uint32_t k = 0; k < nW; k++) for (int i = 0; i < CFO_BITS; i++) wbits[(size_t)k * CFO_BITS + i] = bool_wire(bit_of(wv[k], i));
    for (uint32_t k = 0; k < nE; k++) for (int i = 0; i < CFO_BITS; i++) ebits[(size_t)k * CFO_BITS + i] = bool_wire(bit_of(ev[k], i));
    lap("bits");
    // check 5
    Ec ec(cs, CycleSide<BnFr>::b(), CycleSide<BnFr>::G());
    auto opened = [&](const std::vector<N>& bits, const U256w* vals, uint32_t cnt, const N& cx_, const N& cy_) {
      XY tot = open_chains(*in.key, bits, vals, cnt);
      const Affine<F> K = in.key->offset(cnt);
      Ec::Pt cm; cm.x = cx_; cm.y = cy_; cm.inf = cs.is_zero(cm.y);
      Ec::Pt kc; kc.x = cs.constant(K.x); kc.y = cs.constant(K.y); kc.inf = cs.zero();
      Ec::Pt s = ec.add(cm, kc);
      cs.enforce_equal(s.x, tot.x); cs.enforce_equal(s.y, tot.y);
      if (!cs.b && (!s.x.v.eq(tot.x.v) || !s.y.v.eq(tot.y.v))) cs.bad = true;
    };
    opened(wbits, wv.data(), nW, cWx, cWy);
    lap("opening of cmW");
    opened(ebits, ev.data(), nE, cEx, cEy);
    lap("opening of cmE");
    // check 6: z = (u, W, x) as integers
    std::vector<Big> z(sh.n_wires);
    { const F uc = F::from_mont(cu.v);
      U256w uv; for (int i = 0; i < 4; i++) uv.w[i] = (uint64_t)uc.v[2 * i] | ((uint64_t)uc.v[2 * i + 1] << 32);
      std::vector<N> ub = cs.bits(cu, 192);      // u_i = Σ of 2 i challenges of 129 bits: far below 2^192
      // (bit 0 of cs.bits is a linear combination, not a wire: give the limbs wires of their own from packs built by hand)
      Big g; g.lb = 192; g.v = uv;
      for (int j = 0; j < 3; j++) { const int lo = NN_L * j, hi = std::min(NN_L * (j + 1), 192); N pk = cs.pack(ub, lo, hi); N w = cs.alloc(pk.v); cs.enforce_equal(w, pk); g.l[j] = w; }
      g.nat = cu;
      z[0] = g; }
    for (uint32_t k = 0; k < nW; k++) z[1 + k] = big_from_bits(wbits, (size_t)k * CFO_BITS, CFO_BITS, wv[k], true);
    for (int k = 0; k < CF_IO; k++) {
      std::vector<N> xb;
      for (int j = 0; j < 4; j++) { std::vector<N> bj = cs.bits(cx[k][j], 64); xb.insert(xb.end(), bj.begin(), bj.end()); }
      Big g; g.lb = 256; g.v = cfU.x[k];
      for (int j = 0; j < 3; j++) { const int lo = NN_L * j, hi = std::min(NN_L * (j + 1), 256); N pk = cs.pack(xb, lo, hi); N w = cs.alloc(pk.v); cs.enforce_equal(w, pk); g.l[j] = w; }
      g.nat = cs.add(cs.add(cx[k][0], cs.scale(cx[k][1], cb::f_pow2<F>(64))), cs.add(cs.scale(cx[k][2], cb::f_pow2<F>(128)), cs.scale(cx[k][3], cb::f_pow2<F>(192))));
      z[1 + nW + k] = g;
    }
    lap("integers of z");
    relation(sh, z, ebits, have ? in.E : nullptr);
    lap("relation");
  }
};

}  // namespace aug
}  // namespace vz
