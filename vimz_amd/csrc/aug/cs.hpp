// Constraint system + gadgets for Nova's augmented circuits.
//
// nova-snark builds its augmented circuit with bellperson: `synthesize` is run once to extract the R1CS shape and
// once per step with values to produce the witness (nova-snark 0.23.0, src/circuit.rs / src/gadgets/*, not vendored).
// This file follows that pattern with our own gadgets: the same code path emits the constraints (shape mode: `b`
// set) and computes every wire value (both modes), so the witness of a step is the result of running the circuit.
// Field-generic: the primary circuit lives over BN254 Fr, the secondary over BN254 Fq (= Grumpkin's scalar field).
//
// Host code (runs per folding step on the CPU, like the reference's): values only in witness mode — no linear
// combinations are built, inversions are batched (Montgomery's trick) by the gadgets that need many.
#pragma once
#include <stdint.h>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>
#include <algorithm>
#if defined(__linux__)
#include <sched.h>
#endif
#include "../circuit/builder.hpp"
#include "../circuit/poseidon_params.hpp"
#include "../ec.hpp"

namespace vz {
namespace aug {

template <class F>
struct Num {
  cb::LCT<F> lc;   // only maintained in shape mode
  F v;
  bool konst = false;   // value is a circuit constant (same in every run): products with it cost nothing
};

// One helper thread per prover for the two independent scalar-multiplication chains of a step (host cores are plentiful
// next to a GPU; the chains are the largest sequential piece of the verifier circuit's witness).
struct Worker {
  // The next job arrives a few milliseconds after the last one while a proof is being folded, and waking a sleeping thread
  // costs more than the job saves: the helper spins for a while after each job and only then goes to sleep.
  std::thread th; std::mutex m; std::condition_variable cv;
  std::function<void()> job;
  std::atomic<int> state{0};          // 0 idle, 1 job posted, 2 job done
  std::atomic<bool> sleeping{false}, stop{false};
  // how long an idle helper spins before it sleeps (VIMZ_WORKER_SPIN_US, default 100 µs).  It used to be 20 ms — "waking a sleeping thread costs more
  // than the job saves" was measured with the lost wake-up above still in place; without it 0 / 200 / 2 000 / 20 000 µs give the same steps/s (one chain
  // 753-755, CycleFold 452-457 / 820-833), and idle helpers that do not burn a core each are what a process under a CPU quota needs
  static long spin_us() { static const long v = [] { const char* e = getenv("VIMZ_WORKER_SPIN_US"); const long x = e ? atol(e) : 100; return x < 0 ? 0 : x; }(); return v; }
  Worker() { th = std::thread([this] { loop(); }); }
  ~Worker() { stop = true; { std::lock_guard<std::mutex> g(m); } cv.notify_all(); th.join(); }
  void loop() {
    for (;;) {
      const auto t0 = std::chrono::steady_clock::now();
      int spins = 0;
      while (state.load(std::memory_order_acquire) != 1) {
        if (stop) return;
        std::this_thread::yield();      // free on an idle core; hands the core over when there is only one
        if ((++spins & 63) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us())) {
          std::unique_lock<std::mutex> lk(m);
          sleeping.store(true, std::memory_order_seq_cst);
          cv.wait(lk, [this] { return state.load(std::memory_order_seq_cst) == 1 || stop; });
          sleeping = false;
          if (stop) return;
          break;
        }
      }
      job();
      state.store(2, std::memory_order_release);
    }
  }
  void start(std::function<void()> f) {
    job = std::move(f);
    // (sequentially consistent, like the helper's `sleeping = true` and its re-check of `state` under the lock: with a release store the read of
    //  `sleeping` below could pass it — the helper goes to sleep, nobody wakes it, wait() spins for ever.  Seen as a hang once the spin window was short.)
    state.store(1, std::memory_order_seq_cst);
    if (sleeping.load(std::memory_order_seq_cst)) { { std::lock_guard<std::mutex> g(m); } cv.notify_all(); }
  }
  void wait() { while (state.load(std::memory_order_acquire) != 2) std::this_thread::yield(); state.store(0, std::memory_order_relaxed); }
};

// Wires of one hash whose inputs recur: H(pz, i+1, z_{i+1}, U_new) computed at the end of a step is exactly the hash the next
// step's circuit recomputes to check its incoming instance, so its S-box wires are kept and replayed.
template <class F>
struct HashCache { std::vector<F> in, wires; F out; bool valid = false; };
// The instance hash is H(H(digest, i, z_0, z_i), U):  `pre` / `rest` hold the two hashes a step's circuit ends with, replayed as the
// incoming-hash check of the next step;  `next_pre` is the statement part of THIS step's output hash, which the prover may compute
// ahead (AugCircuit::precompute_statement) while it waits for the commitments the rest of the circuit needs.
template <class F>
struct AugCache { HashCache<F> pre, rest, next_pre; };

// Montgomery batch inversion; zeros are left as zero.
template <class F>
inline void batch_inv(std::vector<F>& a) {
  const size_t n = a.size();
  if (!n) return;
  std::vector<F> pre(n);
  F acc = F::one();
  for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!a[i].is_zero()) acc = F::mul(acc, a[i]); }
  F inv = F::pow_pm2(acc);
  for (size_t i = n; i-- > 0;) {
    if (a[i].is_zero()) continue;
    const F ai = a[i];
    a[i] = F::mul(inv, pre[i]);
    inv = F::mul(inv, ai);
  }
}

// Host CPUs this process may run on: the scheduler's affinity mask (what `taskset`, a launcher's binding or a container's cpuset
// leave), not the machine's count — helper threads beyond it only take turns with the threads they are meant to help.
inline unsigned affinity_cpus() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
#if defined(__linux__)
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int k = CPU_COUNT(&set); if (k >= 1) n = std::min<unsigned>(n, (unsigned)k); }
#endif
  return n;
}

template <class FP>
struct CS {
  typedef Fp<FP> F;
  typedef cb::LCT<F> LC;
  typedef Num<F> N;

  cb::BuilderT<F>* b = nullptr;   // shape mode: constraints and wires are appended here
  uint32_t base = 0;              // index of the first wire this synthesis allocates
  std::vector<F> w;               // values of the wires allocated by this synthesis, in order
  bool bad = false;               // witness mode: some value did not fit its range (the witness will not satisfy)
  Worker* worker = nullptr;       // witness mode: optional helper threads for the two scalar-multiplication chains
  Worker* worker2 = nullptr;
  std::function<void(const uint32_t*)>* on_challenge = nullptr;   // witness mode: called once with the challenge's low 128 bits (see circuit.hpp)

  bool shape() const { return b != nullptr; }

  N constant(const F& c) const { N r; r.v = c; r.konst = true; if (b) r.lc = LC::constant(c); return r; }
  N constant_u(uint64_t c) const { return constant(cb::f_from_u64<F>(c)); }
  N zero() const { return constant(F::zero()); }
  N one() const { return constant(F::one()); }
  N alloc(const F& v) {
    const uint32_t idx = base + (uint32_t)w.size();
    w.push_back(v);
    N r; r.v = v;
    if (b) { const uint32_t got = b->alloc(1); if (got != idx) throw std::runtime_error("aug: wire allocation out of step"); r.lc = LC::wire(idx); }
    return r;
  }
  N wire(uint32_t idx, const F& v) const { N r; r.v = v; if (b) r.lc = LC::wire(idx); return r; }   // an existing wire of the host circuit

  N add(const N& x, const N& y) const { N r; r.v = F::add(x.v, y.v); r.konst = x.konst && y.konst; if (b) r.lc = x.lc + y.lc; return r; }
  N sub(const N& x, const N& y) const { N r; r.v = F::sub(x.v, y.v); r.konst = x.konst && y.konst; if (b) r.lc = x.lc - y.lc; return r; }
  N scale(const N& x, const F& k) const { N r; r.v = F::mul(x.v, k); r.konst = x.konst; if (b) r.lc = x.lc.scaled(k); return r; }
  N scale_u(const N& x, uint64_t k) const { return scale(x, cb::f_from_u64<F>(k)); }
  N addc(const N& x, const F& c) const { N r; r.v = F::add(x.v, c); r.konst = x.konst; if (b) r.lc = x.lc + LC::constant(c); return r; }
  N neg(const N& x) const { N r; r.v = F::neg(x.v); r.konst = x.konst; if (b) r.lc = LC() - x.lc; return r; }
  N one_minus(const N& x) const { return sub(one(), x); }

  void enforce(const N& x, const N& y, const N& z) { if (b) b->enforce(x.lc, y.lc, z.lc); }   // x * y = z
  void enforce_zero(const N& x) { if (b) b->enforce(x.lc, LC::constant(F::one()), LC()); }
  void enforce_equal(const N& x, const N& y) { enforce_zero(sub(x, y)); }

  N mul(const N& x, const N& y) {
    if (x.konst) return scale(y, x.v);
    if (y.konst) return scale(x, y.v);
    N r = alloc(F::mul(x.v, y.v));
    enforce(x, y, r);
    return r;
  }
  N sqr(const N& x) { return mul(x, x); }
  // s ? x : y   (s boolean)
  N select(const N& s, const N& x, const N& y) { return add(y, mul(s, sub(x, y))); }
  // 1 if x == 0 else 0; `inv_hint` (optional) = 1/x computed elsewhere
  N is_zero(const N& x, const F* inv_hint = nullptr) {
    const F inv = inv_hint ? *inv_hint : F::pow_pm2(x.v);
    N i = alloc(inv);
    N out = alloc(x.v.is_zero() ? F::one() : F::zero());
    enforce(x, i, one_minus(out));
    enforce(x, out, zero());
    return out;
  }

  // Little-endian bits of x (n of them).  Bit 0 is the substituted signal x - sum_{k>=1} 2^k b_k (circom's convention),
  // so the gadget costs n constraints and n-1 wires and proves x < 2^n.
  std::vector<N> bits(const N& x, int n) { return bits_of(x, F::from_mont(x.v), n); }
  // The decomposition forced CANONICAL (circom's Num2Bits_strict, arkworks' to_bits_le): n = the field's bit length, and the bits are
  // also compared with p - 1.  Of the two bit vectors with sum 2^k b_k = x mod p — those of x and, when it is below 2^n, of x + p — the
  // plain gadget accepts both; a hash-to-challenge decomposition that did would let the prover choose between two challenges (ADVICE
  // r3).  Walking down from the top bit, `run` = the AND of the bits at the positions where p - 1 has a one; at every position where
  // p - 1 has a zero, run · b_k = 0 (a one there, with everything above equal to p - 1, would make the number exceed it).  One
  // constraint per bit below the leading ones of p - 1, one wire per further one.
  std::vector<N> bits_strict(const N& x) {
    const int n = FP::BITS;
    F c = F::from_mont(x.v);
    if (!b && alias_attack) {      // test hook of the negative test (witness mode only): the bits of x + p where they fit n bits
      F a; uint64_t cy = 0;
      for (int i = 0; i < 8; i++) { cy += (uint64_t)c.v[i] + FP::MOD.w[i]; a.v[i] = (uint32_t)cy; cy >>= 32; }
      bool fits = cy == 0;
      for (int k = n; k < 256; k++) if ((a.v[k >> 5] >> (k & 31)) & 1u) fits = false;
      if (fits) { c = a; alias_used++; }
    }
    std::vector<N> r = bits_of(x, c, n);
    uint32_t pm1[8];
    { uint64_t br = 1; for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)FP::MOD.w[i] - br; pm1[i] = (uint32_t)d; br = (d >> 32) & 1; } }
    N run; bool have_run = false;      // (no run yet: the leading ones of p - 1 — the AND of nothing is the constant one)
    for (int k = n - 1; k >= 0; k--) {
      if ((pm1[k >> 5] >> (k & 31)) & 1u) {
        if (!have_run) { run = r[k]; have_run = true; }
        else run = mul(run, r[k]);
      } else if (have_run) {
        enforce(run, r[k], zero());
        if (!b && !alias_attack && !F::mul(run.v, r[k].v).is_zero()) bad = true;
      } else {      // (cannot happen for a modulus of n bits: its top bit is a one)
        enforce_zero(r[k]);
      }
    }
    return r;
  }
  bool alias_attack = false; int alias_used = 0;
  std::vector<N> bits_of(const N& x, const F& c, int n) {
    for (int k = n; k < 256; k++) if ((c.v[k >> 5] >> (k & 31)) & 1u) bad = true;
    std::vector<N> r((size_t)n);
    LC rest;
    for (int k = 1; k < n; k++) {
      const bool bit = (c.v[k >> 5] >> (k & 31)) & 1u;
      r[k] = alloc(bit ? F::one() : F::zero());
      if (b) { b->enforce(r[k].lc, r[k].lc - LC::constant(F::one()), LC()); rest.t.push_back({r[k].lc.t[0].w, F::neg(cb::f_pow2<F>(k))}); }
    }
    r[0].v = (c.v[0] & 1u) ? F::one() : F::zero();
    if (b) { r[0].lc = x.lc + rest; b->enforce(r[0].lc, r[0].lc - LC::constant(F::one()), LC()); }
    return r;
  }
  // sum_{k in [lo,hi)} 2^(k-lo) bits[k]
  N pack(const std::vector<N>& bits, int lo, int hi) const {
    N r = zero();
    F val = F::zero();
    for (int k = hi - 1; k >= lo; k--) { val = F::dbl(val); val = F::add(val, bits[k].v); }
    if (b) { LC acc; for (int k = lo; k < hi; k++) acc = LC::axpy(acc, cb::f_pow2<F>(k - lo), bits[k].lc); r.lc = acc; }
    r.v = val; r.konst = false;
    return r;
  }

  // ---- Poseidon (circomlib construction: state = [0, inputs], x^5, output state[0]) over this field ------------------
  N poseidon(const std::vector<N>& in) {
    const int t = (int)in.size() + 1;
    if (t > POSEIDON_MAX_T) throw std::runtime_error("aug poseidon: too many inputs");
    if (!b) {   // witness mode: values only, partial rounds in sparse form; emits exactly the wires the shape mode allocates
      bool plain = true;
      for (auto& x : in) plain = plain && !x.konst;
      if (plain) {
        F s[POSEIDON_MAX_T];
        s[0] = F::zero();
        for (int i = 1; i < t; i++) s[i] = in[i - 1].v;
        cb::poseidon_permute<FP>(s, t, true, &w);
        N r; r.v = s[0];
        return r;
      }
    }
    const cb::PoseidonTableT<F>& P = cb::poseidon_table_t<FP>(t);
    std::vector<N> st((size_t)t), nx((size_t)t);
    st[0] = zero();
    for (int i = 1; i < t; i++) st[i] = in[i - 1];
    const int R = P.rf + P.rp;
    for (int r = 0; r < R; r++) {
      const bool full = r < P.rf / 2 || r >= P.rf / 2 + P.rp;
      for (int i = 0; i < t; i++) st[i] = addc(st[i], P.C[(size_t)r * t + i]);
      for (int i = 0; i < (full ? t : 1); i++) {
        if (st[i].konst) { F x2 = F::sqr(st[i].v), x4 = F::sqr(x2); st[i] = constant(F::mul(x4, st[i].v)); continue; }
        N x2 = mul(st[i], st[i]); N x4 = mul(x2, x2); st[i] = mul(x4, st[i]);
      }
      for (int i = 0; i < t; i++) {
        F acc = F::zero(); bool k = true;
        for (int j = 0; j < t; j++) { acc = F::add(acc, F::mul(P.M[(size_t)i * t + j], st[j].v)); k = k && st[j].konst; }
        nx[i].v = acc; nx[i].konst = k;
        if (b) { LC l; for (int j = 0; j < t; j++) l = LC::axpy(l, P.M[(size_t)i * t + j], st[j].lc); nx[i].lc = l; }
      }
      st.swap(nx);
    }
    return st[0];
  }
  // hash() with replay: `read` (if its inputs match) supplies the wires; `write` receives them for a later replay
  N hash_cached(const std::vector<N>& in, const HashCache<F>* read, HashCache<F>* write) {
    if (b) return hash(in);
    if (read && read->valid && read->in.size() == in.size()) {
      bool same = true;
      for (size_t i = 0; i < in.size() && same; i++) same = read->in[i].eq(in[i].v);
      if (same) { w.insert(w.end(), read->wires.begin(), read->wires.end()); N r; r.v = read->out; if (write && write != read) *write = *read; return r; }
    }
    const size_t w0 = w.size();
    N h = hash(in);
    if (write) {
      write->in.resize(in.size());
      for (size_t i = 0; i < in.size(); i++) write->in[i] = in[i].v;
      write->wires.assign(w.begin() + w0, w.end());
      write->out = h.v; write->valid = true;
    }
    return h;
  }
  // Hash of any number of elements: the first permutation absorbs 8, every further one the running hash + 7.
  N hash(const std::vector<N>& in) {
    const size_t n = in.size();
    const size_t first = n < 8 ? n : 8;
    N h = poseidon(std::vector<N>(in.begin(), in.begin() + first));
    size_t pos = first;
    while (pos < n) {
      const size_t cur = n - pos < 7 ? n - pos : 7;
      std::vector<N> nx; nx.push_back(h);
      nx.insert(nx.end(), in.begin() + pos, in.begin() + pos + cur);
      h = poseidon(nx);
      pos += cur;
    }
    return h;
  }
};

// Native (no circuit) evaluation of the same hash, for the verifier and for values the prover needs ahead of synthesis.
template <class FP>
inline Fp<FP> hash_native(const std::vector<Fp<FP>>& in) {
  typedef Fp<FP> F;
  const size_t n = in.size();
  const size_t first = n < 8 ? n : 8;
  F h = cb::poseidon_hash_t<FP>(in.data(), (int)first);
  size_t pos = first;
  while (pos < n) {
    const size_t cur = n - pos < 7 ? n - pos : 7;
    F nx[8]; nx[0] = h;
    for (size_t i = 0; i < cur; i++) nx[1 + i] = in[pos + i];
    h = cb::poseidon_hash_t<FP>(nx, (int)cur + 1);
    pos += cur;
  }
  return h;
}

// ====================================================================================================================
// Elliptic-curve gadgets: y^2 = x^3 + b over the circuit's field (the OTHER curve of the cycle, whose points are the
// commitments this circuit folds).  Affine coordinates, identity encoded as (0,0) with an explicit flag.
// Both curves of the cycle have prime order, so a finite point never has y = 0: inf = IsZero(y).
// ====================================================================================================================
template <class FP>
struct EcGadgets {
  typedef Fp<FP> F;
  typedef CS<FP> Cs;
  typedef Num<F> N;
  struct Pt { N x, y, inf; };

  Cs& cs;
  F curve_b;
  Affine<F> G;      // a fixed finite point: stands in for an identity operand of a scalar multiplication
  EcGadgets(Cs& c, const F& bb, const Affine<F>& g) : cs(c), curve_b(bb), G(g) {}

  Pt constant_identity() { Pt p; p.x = cs.zero(); p.y = cs.zero(); p.inf = cs.one(); return p; }

  // Allocate a point from its value.  check: constrain it to be the identity or on the curve (for prover-supplied points;
  // points that arrive through a checked hash were produced by an earlier instance of this circuit and need no check).
  Pt alloc(const Affine<F>& p, bool check) {
    Pt r; r.x = cs.alloc(p.x); r.y = cs.alloc(p.y);
    r.inf = cs.is_zero(r.y);
    if (check) {
      cs.enforce(r.x, r.inf, cs.zero());
      N xx = cs.mul(r.x, r.x); N xxx = cs.mul(xx, r.x);
      cs.enforce(r.y, r.y, cs.add(xxx, cs.scale(cs.one_minus(r.inf), curve_b)));
    }
    return r;
  }
  Pt select(const N& s, const Pt& a, const Pt& c) { Pt r; r.x = cs.select(s, a.x, c.x); r.y = cs.select(s, a.y, c.y); r.inf = cs.select(s, a.inf, c.inf); return r; }

  struct XY { N x, y; };
  // 2P for a finite P; lam = 3x^2 / 2y supplied by the caller (batched inversion)
  XY dbl(const XY& p, const F& lam) {
    N xx = cs.mul(p.x, p.x);
    N l = cs.alloc(lam);
    cs.enforce(l, cs.scale_u(p.y, 2), cs.scale_u(xx, 3));
    XY r;
    r.x = cs.alloc(F::sub(F::sqr(lam), F::dbl(p.x.v)));
    cs.enforce(l, l, cs.add(r.x, cs.scale_u(p.x, 2)));
    r.y = cs.alloc(F::sub(F::mul(lam, F::sub(p.x.v, r.x.v)), p.y.v));
    cs.enforce(l, cs.sub(p.x, r.x), cs.add(r.y, p.y));
    return r;
  }
  // P + Q for finite P, Q with different x; lam = (yQ - yP)/(xQ - xP) supplied by the caller
  XY add_distinct(const XY& p, const XY& q, const F& lam) {
    N l = cs.alloc(lam);
    cs.enforce(l, cs.sub(q.x, p.x), cs.sub(q.y, p.y));
    XY r;
    r.x = cs.alloc(F::sub(F::sub(F::sqr(lam), p.x.v), q.x.v));
    cs.enforce(l, l, cs.add(r.x, cs.add(p.x, q.x)));
    r.y = cs.alloc(F::sub(F::mul(lam, F::sub(p.x.v, r.x.v)), p.y.v));
    cs.enforce(l, cs.sub(p.x, r.x), cs.add(r.y, p.y));
    return r;
  }

  // Native double-and-add chains of (2^nbits + k)·P for several (P, same k) at once, then every slope by two batched
  // inversions.  bits: the low nbits of the scalar (canonical words), the leading one is implicit.
  // wires: the 9·nbits values scalar_mul allocates inside its loop, in its order (x², λ_d, 2P, λ_a, 2P + P, the two selects'
  // products) — the chain has them all at hand, so in witness mode the gadget only appends them;  last: the final accumulator
  struct ChainHints { std::vector<F> lam_d, lam_a, wires; Affine<F> last; };
  static void chain_hints(const std::vector<Affine<F>>& Ps, const uint32_t* k, int nbits, std::vector<ChainHints>& out) {
    const size_t m = Ps.size();
    out.assign(m, ChainHints());
    std::vector<XYZZ<F>> D(m * (size_t)nbits), A(m * (size_t)nbits);
    for (size_t j = 0; j < m; j++) {
      XYZZ<F> acc = from_affine(Ps[j]);
      for (int i = nbits - 1; i >= 0; i--) {
        XYZZ<F> d = vz::dbl(acc); XYZZ<F> a = d; add_mixed(a, Ps[j]);
        D[j * nbits + i] = d; A[j * nbits + i] = a;
        acc = ((k[i >> 5] >> (i & 31)) & 1u) ? a : d;
      }
    }
    // affine coordinates of every intermediate point: one batched inversion of the ZZZ's
    std::vector<F> zi(2 * m * (size_t)nbits);
    for (size_t i = 0; i < m * (size_t)nbits; i++) { zi[2 * i] = D[i].ZZZ; zi[2 * i + 1] = A[i].ZZZ; }
    batch_inv(zi);
    std::vector<Affine<F>> Da(m * (size_t)nbits), Aa(m * (size_t)nbits);
    auto norm = [](const XYZZ<F>& p, const F& zi3) { Affine<F> a; F zi2 = F::sqr(F::mul(zi3, p.ZZ)); a.x = F::mul(p.X, zi2); a.y = F::mul(p.Y, zi3); return a; };
    for (size_t i = 0; i < m * (size_t)nbits; i++) { Da[i] = norm(D[i], zi[2 * i]); Aa[i] = norm(A[i], zi[2 * i + 1]); }
    // slopes: doubling of the accumulator before step i, addition D_i + P
    std::vector<F> den(2 * m * (size_t)nbits);
    std::vector<Affine<F>> accs(m * (size_t)nbits);
    for (size_t j = 0; j < m; j++) {
      Affine<F> acc = Ps[j];
      for (int i = nbits - 1; i >= 0; i--) {
        accs[j * nbits + i] = acc;
        den[2 * (j * nbits + i)] = F::dbl(acc.y);
        den[2 * (j * nbits + i) + 1] = F::sub(Ps[j].x, Da[j * nbits + i].x);
        acc = ((k[i >> 5] >> (i & 31)) & 1u) ? Aa[j * nbits + i] : Da[j * nbits + i];
      }
    }
    batch_inv(den);
    for (size_t j = 0; j < m; j++) {
      out[j].lam_d.resize(nbits); out[j].lam_a.resize(nbits);
      std::vector<F>& w = out[j].wires;
      w.resize(9 * (size_t)nbits);
      size_t o = 0;
      for (int i = nbits - 1; i >= 0; i--) {
        const size_t ix = j * nbits + i;
        const F xx = F::sqr(accs[ix].x);
        out[j].lam_d[i] = F::mul(F::add(F::dbl(xx), xx), den[2 * ix]);
        out[j].lam_a[i] = F::mul(F::sub(Ps[j].y, Da[ix].y), den[2 * ix + 1]);
        const bool bit = (k[i >> 5] >> (i & 31)) & 1u;
        w[o++] = xx; w[o++] = out[j].lam_d[i]; w[o++] = Da[ix].x; w[o++] = Da[ix].y;
        w[o++] = out[j].lam_a[i]; w[o++] = Aa[ix].x; w[o++] = Aa[ix].y;
        w[o++] = bit ? F::sub(Aa[ix].x, Da[ix].x) : F::zero();
        w[o++] = bit ? F::sub(Aa[ix].y, Da[ix].y) : F::zero();
      }
      out[j].last = (k[0] & 1u) ? Aa[j * nbits] : Da[j * nbits];
    }
  }

  // Both chains on helper threads (one each), started as soon as the challenge is known: the calling thread goes on with the part of
  // the circuit that needs the challenge only (non-native folds, the first permutation of the output hash) and collects them after.
  // Without helpers (or with one) the missing chains are computed at wait().
  struct ChainJob {
    std::vector<Affine<F>> ps[2]; std::vector<ChainHints> o[2]; const uint32_t* k = nullptr; int nbits = 0; Worker* w[2] = {nullptr, nullptr};
    void start(const Affine<F>& p0, const Affine<F>& p1, const uint32_t* k_, int nbits_, Worker* w0, Worker* w1) {
      ps[0] = {p0}; ps[1] = {p1}; k = k_; nbits = nbits_; w[0] = w0; w[1] = w1;
      for (int j = 0; j < 2; j++) if (w[j]) w[j]->start([this, j] { chain_hints(ps[j], k, nbits, o[j]); });
    }
    void wait(std::vector<ChainHints>& out) {
      for (int j = 0; j < 2; j++) { if (w[j]) w[j]->wait(); else chain_hints(ps[j], k, nbits, o[j]); }
      out.clear(); out.push_back(std::move(o[0][0])); out.push_back(std::move(o[1][0]));
    }
  };

  // (2^nbits + sum bits[i] 2^i) · P.  An identity P gives the identity.  9 constraints per bit.
  // The accumulator is an even multiple >= 2 of P when P is added, so the two x-coordinates always differ.
  Pt scalar_mul(const Pt& P, const std::vector<N>& bits, int nbits, const ChainHints& h) {
    XY Pe; Pe.x = cs.select(P.inf, cs.constant(G.x), P.x); Pe.y = cs.select(P.inf, cs.constant(G.y), P.y);
    XY acc = Pe;
    if (!cs.b && h.wires.size() == 9 * (size_t)nbits) {       // witness mode: the chain's values, as computed with the slopes
      cs.w.insert(cs.w.end(), h.wires.begin(), h.wires.end());
      acc.x.v = h.last.x; acc.y.v = h.last.y; acc.x.konst = acc.y.konst = false;
    } else
    for (int i = nbits - 1; i >= 0; i--) {
      XY d = dbl(acc, h.lam_d[i]);
      XY a = add_distinct(d, Pe, h.lam_a[i]);
      acc.x = cs.select(bits[i], a.x, d.x);
      acc.y = cs.select(bits[i], a.y, d.y);
    }
    Pt r;
    N fin = cs.one_minus(P.inf);
    r.x = cs.mul(fin, acc.x); r.y = cs.mul(fin, acc.y); r.inf = P.inf;
    return r;
  }
  static Affine<F> scalar_operand(const Affine<F>& P, const Affine<F>& G) { return aff_is_identity(P) ? G : P; }

  // P + Q where either may be the identity.  P = ±Q (both finite) makes the constraints unsatisfiable: it cannot be
  // provoked (Q is a hash-derived multiple) and never yields a wrong sum.
  Pt add(const Pt& P, const Pt& Q) {
    N both = cs.mul(P.inf, Q.inf);
    N skip = cs.sub(cs.add(P.inf, Q.inf), both);
    N dx = cs.sub(Q.x, P.x), dy = cs.sub(Q.y, P.y);
    N e = cs.mul(skip, cs.one_minus(dx));
    N dxe = cs.add(dx, e);                       // skip ? 1 : dx
    const F inv = F::pow_pm2(dxe.v);
    if (dxe.v.is_zero()) cs.bad = true;
    N iv = cs.alloc(inv);
    cs.enforce(dxe, iv, cs.one());
    const F lam = F::mul(dy.v, inv);
    N l = cs.alloc(lam);
    cs.enforce(l, dxe, dy);
    XY r;
    r.x = cs.alloc(F::sub(F::sub(F::sqr(lam), P.x.v), Q.x.v));
    cs.enforce(l, l, cs.add(r.x, cs.add(P.x, Q.x)));
    r.y = cs.alloc(F::sub(F::mul(lam, F::sub(P.x.v, r.x.v)), P.y.v));
    cs.enforce(l, cs.sub(P.x, r.x), cs.add(r.y, P.y));
    Pt out;
    N tx = cs.select(Q.inf, P.x, r.x), ty = cs.select(Q.inf, P.y, r.y);
    out.x = cs.select(P.inf, Q.x, tx); out.y = cs.select(P.inf, Q.y, ty);
    out.inf = both;
    return out;
  }
};

// ====================================================================================================================
// Non-native fold  X' = X + rho·x  (mod m), m = the other field's modulus.  X, X' as four 64-bit limbs, x < 2^250 given by
// its bits, rho = 2^128 + rho1·2^64 + rho0.  Proved as the integer identity X + rho·x = k·m + X' limb by limb with signed
// carries (offset by 2^68, 69-bit range checks).  X' is any 256-bit representative (not forced below m), as in nova-snark's
// BigNat folding.
// ====================================================================================================================
struct U256w { uint64_t w[4]; };

template <class FP, class OtherP>
struct NonNative {
  typedef Fp<FP> F;
  typedef CS<FP> Cs;
  typedef Num<F> N;

  static U256w modulus() { U256w m; for (int i = 0; i < 4; i++) m.w[i] = (uint64_t)OtherP::MOD.w[2 * i] | ((uint64_t)OtherP::MOD.w[2 * i + 1] << 32); return m; }
  static F from_u64(uint64_t v) { return cb::f_from_u64<F>(v); }
  static F from_u128(unsigned __int128 v) { F x = F::zero(); for (int i = 0; i < 4; i++) x.v[i] = (uint32_t)(v >> (32 * i)); return F::to_mont(x); }

  // native value of the fold (used by the host prover and the verifier as well)
  static U256w fold_value(const U256w& X, const uint32_t rho_low128[4], const U256w& x) {
    typedef Fp<OtherP> G;
    G Xg, xg, rg = G::zero();
    for (int i = 0; i < 4; i++) { Xg.v[2 * i] = (uint32_t)X.w[i]; Xg.v[2 * i + 1] = (uint32_t)(X.w[i] >> 32); xg.v[2 * i] = (uint32_t)x.w[i]; xg.v[2 * i + 1] = (uint32_t)(x.w[i] >> 32); }
    for (int i = 0; i < 4; i++) rg.v[i] = rho_low128[i];
    rg.v[4] = 1;
    auto geq_mod = [](const G& a) { for (int i = 7; i >= 0; i--) if (a.v[i] != OtherP::MOD.w[i]) return a.v[i] > OtherP::MOD.w[i]; return true; };
    auto sub_mod = [](G& a) { uint64_t br = 0; for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.v[i] - OtherP::MOD.w[i] - br; a.v[i] = (uint32_t)d; br = (d >> 32) & 1; } };
    while (geq_mod(Xg)) sub_mod(Xg);      // X may be any 256-bit representative
    while (geq_mod(xg)) sub_mod(xg);
    G r = G::add(G::to_mont(Xg), G::mul(G::to_mont(rg), G::to_mont(xg)));
    r = G::from_mont(r);
    U256w o; for (int i = 0; i < 4; i++) o.w[i] = (uint64_t)r.v[2 * i] | ((uint64_t)r.v[2 * i + 1] << 32);
    return o;
  }

  // X: limb values + Nums; rho0, rho1: Nums (64-bit each); xbits: 250 bits of x.  Returns the limbs of X'.
  static void fold(Cs& cs, const N X[4], const U256w& Xv, const N& rho0, const N& rho1, const uint32_t rho_low128[4],
                   const std::vector<N>& xbits, const U256w& xv, N out[4], U256w& outv) {
    typedef unsigned __int128 u128;
    const U256w m = modulus();
    outv = fold_value(Xv, rho_low128, xv);
    // k = (X + rho·x - X') / m exactly, and k < 2^126: k = (S - X') · m^{-1} mod 2^128 from the low 128 bits alone
    const u128 rlow = (u128)rho_low128[0] | ((u128)rho_low128[1] << 32) | ((u128)rho_low128[2] << 64) | ((u128)rho_low128[3] << 96);
    const u128 xlow = (u128)xv.w[0] | ((u128)xv.w[1] << 64), Xlow = (u128)Xv.w[0] | ((u128)Xv.w[1] << 64), olow = (u128)outv.w[0] | ((u128)outv.w[1] << 64);
    const u128 mlow = (u128)m.w[0] | ((u128)m.w[1] << 64);
    u128 minv = 1;
    for (int i = 0; i < 7; i++) minv *= 2 - mlow * minv;     // Newton: inverse of the odd m modulo 2^128
    const u128 k = (Xlow + rlow * xlow - olow) * minv;
    N kN = cs.alloc(from_u128(k));
    std::vector<N> kb = cs.bits(kN, 128);
    N k0 = cs.pack(kb, 0, 64), k1 = cs.pack(kb, 64, 128);
    for (int j = 0; j < 4; j++) { out[j] = cs.alloc(from_u64(outv.w[j])); cs.bits(out[j], 64); }
    N xl[4]; for (int j = 0; j < 4; j++) xl[j] = cs.pack(xbits, 64 * j, j == 3 ? (int)xbits.size() : 64 * (j + 1));
    N p0[4], p1[4];                                          // rho_a * x_b (a = 0,1): one constraint each; rho_2 = 1
    for (int c = 0; c < 4; c++) { p0[c] = cs.mul(rho0, xl[c]); p1[c] = cs.mul(rho1, xl[c]); }
    // limb equations  t_j = X_j + sum rho_a x_b - sum k_a m_b - X'_j + c_{j-1} = c_j 2^64,  |c_j| < 2^67, c_5 = 0.
    // Every t_j is far below the field size, so the carries can be read off with field arithmetic.
    const F two64 = cb::f_pow2<F>(64), off = cb::f_pow2<F>(68);
    static const F inv_two64 = F::pow_pm2(cb::f_pow2<F>(64));
    N carry_in = cs.zero();
    for (int j = 0; j < 6; j++) {
      N t = carry_in;
      if (j < 4) { t = cs.add(t, cs.sub(X[j], out[j])); t = cs.add(t, p0[j]); t = cs.sub(t, cs.scale(k0, from_u64(m.w[j]))); }
      if (j >= 1 && j - 1 < 4) { t = cs.add(t, p1[j - 1]); t = cs.sub(t, cs.scale(k1, from_u64(m.w[j - 1]))); }
      if (j >= 2 && j - 2 < 4) t = cs.add(t, xl[j - 2]);
      if (j == 5) { cs.enforce_zero(t); if (!t.v.is_zero()) cs.bad = true; break; }
      N cN = cs.alloc(F::add(F::mul(t.v, inv_two64), off));   // c_j + 2^68 in [0, 2^69)
      cs.bits(cN, 69);
      N c_signed = cs.addc(cN, F::neg(off));
      cs.enforce_zero(cs.sub(t, cs.scale(c_signed, two64)));
      carry_in = c_signed;
    }
  }

  // The same integer identity for an x that is ANY element of the other field, given as four 64-bit limbs (CycleFold: the coordinates
  // of the points a CycleFold instance speaks about; the caller range-checks the limbs or takes them from a checked hash).
  // rho·x < 2^129·2^256, so the quotient k may need 131 bits: three limbs k0, k1, k2 (k2 < 8).
  static void fold_limbs(Cs& cs, const N X[4], const U256w& Xv, const N& rho0, const N& rho1, const uint32_t rho_low128[4],
                         const N xl[4], const U256w& xv, N out[4], U256w& outv) {
    typedef unsigned __int128 u128;
    const U256w m = modulus();
    outv = fold_value(Xv, rho_low128, xv);
    // k = (X + rho·x − X') / m exactly and k < 2^192: from the low 192 bits alone, with m^{-1} mod 2^192
    auto mul_lo3 = [](const uint64_t* a, const uint64_t* b, uint64_t* o) {
      uint64_t r[3] = {0, 0, 0};
      for (int i = 0; i < 3; i++) { u128 carry = 0; for (int j = 0; i + j < 3; j++) { const u128 cur = (u128)a[i] * b[j] + r[i + j] + carry; r[i + j] = (uint64_t)cur; carry = cur >> 64; } }
      o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
    };
    auto add3 = [](const uint64_t* a, const uint64_t* b, uint64_t* o) { u128 c = 0; for (int i = 0; i < 3; i++) { c += (u128)a[i] + b[i]; o[i] = (uint64_t)c; c >>= 64; } };
    auto sub3 = [](const uint64_t* a, const uint64_t* b, uint64_t* o) { uint64_t br = 0; for (int i = 0; i < 3; i++) { const u128 d = (u128)a[i] - b[i] - br; o[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } };
    const uint64_t rl[3] = {(uint64_t)rho_low128[0] | ((uint64_t)rho_low128[1] << 32), (uint64_t)rho_low128[2] | ((uint64_t)rho_low128[3] << 32), 1};
    uint64_t minv[3] = {1, 0, 0};
    for (int it = 0; it < 8; it++) {      // Newton: inverse of the odd m modulo 2^192
      uint64_t t[3], two[3] = {2, 0, 0};
      mul_lo3(m.w, minv, t); sub3(two, t, t); mul_lo3(minv, t, minv);
    }
    uint64_t sl[3], kq[3];
    mul_lo3(rl, xv.w, sl); add3(sl, Xv.w, sl); sub3(sl, outv.w, sl); mul_lo3(sl, minv, kq);
    if (kq[2] >= 8) cs.bad = true;
    F kf = F::zero();
    for (int i = 0; i < 3; i++) { kf.v[2 * i] = (uint32_t)kq[i]; kf.v[2 * i + 1] = (uint32_t)(kq[i] >> 32); }
    N kN = cs.alloc(F::to_mont(kf));
    std::vector<N> kb = cs.bits(kN, 131);
    N k0 = cs.pack(kb, 0, 64), k1 = cs.pack(kb, 64, 128), k2 = cs.pack(kb, 128, 131);
    for (int j = 0; j < 4; j++) { out[j] = cs.alloc(from_u64(outv.w[j])); cs.bits(out[j], 64); }
    N p0[4], p1[4];
    for (int c = 0; c < 4; c++) { p0[c] = cs.mul(rho0, xl[c]); p1[c] = cs.mul(rho1, xl[c]); }
    const F two64 = cb::f_pow2<F>(64), off = cb::f_pow2<F>(68);
    static const F inv_two64 = F::pow_pm2(cb::f_pow2<F>(64));
    N carry_in = cs.zero();
    for (int j = 0; j < 6; j++) {
      N t = carry_in;
      if (j < 4) { t = cs.add(t, cs.sub(X[j], out[j])); t = cs.add(t, p0[j]); t = cs.sub(t, cs.scale(k0, from_u64(m.w[j]))); }
      if (j >= 1 && j - 1 < 4) { t = cs.add(t, p1[j - 1]); t = cs.sub(t, cs.scale(k1, from_u64(m.w[j - 1]))); }
      if (j >= 2 && j - 2 < 4) { t = cs.add(t, xl[j - 2]); t = cs.sub(t, cs.scale(k2, from_u64(m.w[j - 2]))); }
      if (j == 5) { cs.enforce_zero(t); if (!t.v.is_zero()) cs.bad = true; break; }
      N cN = cs.alloc(F::add(F::mul(t.v, inv_two64), off));   // c_j + 2^68 in [0, 2^69)
      cs.bits(cN, 69);
      N c_signed = cs.addc(cN, F::neg(off));
      cs.enforce_zero(cs.sub(t, cs.scale(c_signed, two64)));
      carry_in = c_signed;
    }
  }
};

}  // namespace aug
}  // namespace vz
