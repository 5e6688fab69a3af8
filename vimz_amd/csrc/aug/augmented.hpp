// An augmented circuit as the IVC prover and verifier use it: the step circuit's R1CS with Nova's verifier circuit
// (circuit.hpp) appended, its per-step witness generator, and the digest that binds the shape into the transcript.
//
// Wire layout of the augmented circuit:
//     [ 1 | step_out (len_z) | step_in (len_z) | step circuit's other wires | augmented wires ... | X0 | X1 ]
// The two public IOs are the LAST two wires; everything between wire 1 and them is the committed witness W.
// Constraint rows: the step circuit's rows first, then the verifier circuit's.
#pragma once
#include "circuit.hpp"
#include <cstdlib>
#include <cstring>
#include <memory>
#include "../keccak.hpp"

namespace vz {
namespace aug {

struct Sha3 {   // SHA3-256, streaming (keccak.hpp holds the permutation)
  uint64_t st[25]; int pos = 0;
  Sha3() { for (auto& x : st) x = 0; }
  void update(const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data;
    for (size_t i = 0; i < n; i++) { st[pos >> 3] ^= (uint64_t)p[i] << (8 * (pos & 7)); if (++pos == 136) { keccak_f1600(st); pos = 0; } }
  }
  template <class T> void vec(const std::vector<T>& v) { uint64_t n = v.size(); update(&n, 8); if (n) update(v.data(), n * sizeof(T)); }
  void finish(uint8_t out[32]) {
    st[pos >> 3] ^= (uint64_t)0x06 << (8 * (pos & 7));
    st[16] ^= 0x8000000000000000ULL;
    keccak_f1600(st);
    memcpy(out, st, 32);
  }
};

template <class FP> struct CycleSide;   // curve constants of the OTHER curve as seen from field FP
template <> struct CycleSide<BnFr> {    // primary: folds commitments on Grumpkin, y^2 = x^3 - 17 over Fr
  typedef BnFq Other;
  static Fp<BnFr> b() { return cb::f_from_i64<Fp<BnFr>>(-17); }
  static Affine<Fp<BnFr>> G() {
    Affine<Fp<BnFr>> g; g.x = Fp<BnFr>::one();
    Fp<BnFr> y = Fp<BnFr>::zero();   // sqrt(-16): the generator barretenberg uses, (1, 0x2cf135e7506a45d632d270d45f1181294833fc48d823f272c)
    const uint32_t w[8] = {0x823f272cu, 0x833fc48du, 0xf1181294u, 0x2d270d45u, 0x06a45d63u, 0xcf135e75u, 0x00000002u, 0x00000000u};
    for (int i = 0; i < 8; i++) y.v[i] = w[i];
    g.y = Fp<BnFr>::to_mont(y);
    return g;
  }
};
template <> struct CycleSide<BnFq> {    // secondary: folds commitments on BN254 G1, y^2 = x^3 + 3 over Fq
  typedef BnFr Other;
  static Fp<BnFq> b() { return cb::f_from_u64<Fp<BnFq>>(3); }
  static Affine<Fp<BnFq>> G() { Affine<Fp<BnFq>> g; g.x = Fp<BnFq>::one(); g.y = cb::f_from_u64<Fp<BnFq>>(2); return g; }
};

template <class FP>
struct AugCircuit {
  typedef Fp<FP> F;
  typedef typename CycleSide<FP>::Other OP;
  std::unique_ptr<cb::BuilderT<F>> owned;
  cb::BuilderT<F>& b;       // the circuit: owned here (secondary) or living in a vimz_circuit (primary: the step circuit's copy)
  bool primary = true;
  AugCircuit() : owned(new cb::BuilderT<F>()), b(*owned) {}
  explicit AugCircuit(cb::BuilderT<F>& ext) : b(ext) {}
  uint32_t len_z = 0, step_wires = 0, step_constraints = 0;
  F digest;                 // SHA3-256 of the shape, truncated to 250 bits
  mutable AugCache<F> cache;                  // the output hashes of the last witness() call, replayed by the next (cs.hpp)
  mutable std::unique_ptr<Worker> worker, worker2;     // helper threads for the two scalar-multiplication chains (created on first use)
  mutable std::function<void(const uint32_t*)> on_challenge;      // witness(): called with the challenge once the parts of the circuit that only need it are done
  Worker *shared_w = nullptr, *shared_w2 = nullptr;    // ... or the owner's (an IVC hands both of its circuits the same two)
  bool use_worker = affinity_cpus() > 2 && !getenv("VIMZ_AUG_NO_THREADS");      // (two helper threads per circuit: pointless on one or two cores)

  uint32_t n_wires() const { return b.n_wires; }
  uint32_t n_constraints() const { return b.n_constraints(); }
  uint32_t aug_wires() const { return b.n_wires - step_wires; }     // includes the two public IOs
  uint32_t x0_wire() const { return b.n_wires - 2; }

  // The trivial step circuit of the secondary curve: one state element, z_out = z_in.
  void init_trivial_step() {
    b = cb::BuilderT<F>();
    b.len_z = 1; b.n_priv = 0; b.n_wires = 3;
    b.enforce(cb::LCT<F>::constant(F::one()), cb::LCT<F>::wire(2), cb::LCT<F>::wire(1));
    b.n_linear = 1;
  }
  // `b` holds a step circuit in the [1 | out | in | ...] layout: append the verifier circuit.
  void finish(bool is_primary) {
    primary = is_primary;
    len_z = b.len_z; step_wires = b.n_wires; step_constraints = b.n_constraints();
    CS<FP> cs; cs.b = &b; cs.base = b.n_wires;
    AugIn<FP> in; in.digest = F::zero(); in.i = 0; in.U = RelaxedInst<F>::zero(); in.u = FreshInst<F>::zero(); in.T.x = in.T.y = F::zero();
    std::vector<Num<F>> zi, zn;
    for (uint32_t k = 0; k < len_z; k++) { zi.push_back(cs.wire(1 + len_z + k, F::zero())); zn.push_back(cs.wire(1 + k, F::zero())); }
    synthesize_augmented<FP, OP>(cs, in, zi, zn, primary, CycleSide<FP>::b(), CycleSide<FP>::G());
    Sha3 h;
    const uint64_t hdr[6] = {0x3130677561ull /* "aug01" */, b.n_wires, b.n_constraints(), len_z, step_wires, (uint64_t)primary};
    h.update(hdr, sizeof(hdr));
    h.vec(b.A.row_ptr); h.vec(b.A.col); h.vec(b.A.coef); h.vec(b.B.row_ptr); h.vec(b.B.col); h.vec(b.B.coef);
    h.vec(b.C.row_ptr); h.vec(b.C.col); h.vec(b.C.coef); h.vec(b.dict);
    uint8_t d[32]; h.finish(d);
    F c; memcpy(c.v, d, 32); c.v[7] &= 0x03ffffffu;
    digest = F::to_mont(c);
  }

  // `b` already holds a finished augmented circuit (a copy of one that finish() made): take over its description instead of synthesising it again
  void adopt(bool is_primary, uint32_t lz, uint32_t sw, uint32_t sc, const F& dg) { primary = is_primary; len_z = lz; step_wires = sw; step_constraints = sc; digest = dg; }

  // Witness of the verifier part for one step.  z_i / z_next: the step circuit's state (values).  aug: receives the
  // aug_wires() values (Montgomery) in wire order, the public IOs last.
  AugOut<FP> witness(const AugIn<FP>& in, const F* z_i, const F* z_next, std::vector<F>& aug, bool* bad) const {
    CS<FP> cs; cs.base = step_wires;
    cs.w.reserve(aug_wires());
    if (use_worker) {
      if (shared_w) { cs.worker = shared_w; cs.worker2 = shared_w2; }
      else { if (!worker) worker.reset(new Worker()); if (!worker2) worker2.reset(new Worker()); cs.worker = worker.get(); cs.worker2 = worker2.get(); }
    }
    cs.on_challenge = &on_challenge;
    std::vector<Num<F>> zi(len_z), zn(len_z);
    for (uint32_t k = 0; k < len_z; k++) { zi[k].v = z_i[k]; zn[k].v = z_next[k]; }
    AugOut<FP> o = synthesize_augmented<FP, OP>(cs, in, zi, zn, primary, CycleSide<FP>::b(), CycleSide<FP>::G(), &cache);
    if (cs.w.size() != aug_wires()) throw std::runtime_error("aug: witness length differs from the shape");
    if (bad) *bad = cs.bad;
    aug.swap(cs.w);
    return o;
  }
  // The statement part H(digest, i + 1, z_0, z_next) of the output hash of step i, with its wires: nothing in it depends on the
  // commitments step i's witness() waits for, so the prover calls this while they are computed.  (A stale or missing entry only
  // costs witness() the permutation: the cache is keyed on the inputs.)
  void precompute_statement(uint64_t i_next, const std::vector<F>& z0, const F* z_next) const {
    CS<FP> cs;
    std::vector<Num<F>> in(2 + 2 * (size_t)len_z);
    in[0].v = digest; in[1].v = cb::f_from_u64<F>(i_next);
    for (uint32_t k = 0; k < len_z; k++) { in[2 + k].v = k < z0.size() ? z0[k] : F::zero(); in[2 + len_z + k].v = z_next[k]; }
    cs.hash_cached(in, nullptr, &cache.next_pre);
  }

};

}  // namespace aug
}  // namespace vz
