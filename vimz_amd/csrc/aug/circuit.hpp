// Nova's augmented circuit F' (Kothapalli–Setty–Tzialla, "Nova", CRYPTO 2022, Fig. 4), the part of
// RecursiveSNARK::prove_step that SURVEY.md §8a lists as S1 (primary) and S2 (secondary half).  nova-snark 0.23.0's
// version (src/circuit.rs: NovaAugmentedCircuit::synthesize) is not vendored with the reference; this is our own
// statement of the same relation with the same public interface (two public IOs per instance, 250-bit hashes that fit both
// fields, 128-bit challenges, BigNat-style folding of the non-native IO), not a byte-compatible copy of its wiring.
//
// One instance of the circuit, over field F (native), does for the OTHER circuit's instances (commitments = points with
// coordinates in F, public IO = elements of the other field):
//     is_base = (i == 0)
//     check   z_i == z_0                                                if is_base (the chain starts from the claimed state)
//     check   u.x0 == trunc250(H(digest, i, z_0, z_i, U))              unless is_base
//     rho     = 2^128 + low128(H(H(digest, i, z_0, z_i, U), u.W, u.x0, u.x1, T))      Fiat–Shamir challenge, leading one explicit
//     U'      = NIFS.V(U or the zero instance if is_base, u, T, rho):
//               W' = W + rho·u.W,  E' = E + rho·T,  u' = u + rho,  X' = X + rho·x (mod the other field's prime)
//     U_new   = primary and is_base ? zero instance : U'
//     z_{i+1} = F(z_i)                                                  (the step circuit; identity on the secondary)
//     public IO  X0 = u.x1 (passed through),  X1 = trunc250(H(digest, i+1, z_0, z_{i+1}, U_new))
// digest (of the shape) and z_0 are witness wires that every hash absorbs, as in nova-snark's circuit: the verifier recomputes the
// final hash with the true digest and the claimed z_0, and the chain of hash checks carries both back to step 0, where z_i is
// tied to z_0.
#pragma once
#include "cs.hpp"

namespace vz {
namespace aug {

template <class F>
struct RelaxedInst {      // a relaxed instance of the OTHER circuit, as this circuit's field sees it
  Affine<F> W, E;
  F u;                    // 1 + sum of challenges: a small integer, the same in both fields
  U256w X0, X1;           // elements of the other field (any 256-bit representative)
  static RelaxedInst zero() { RelaxedInst r; r.W.x = r.W.y = r.E.x = r.E.y = r.u = F::zero(); for (int i = 0; i < 4; i++) r.X0.w[i] = r.X1.w[i] = 0; return r; }
};
template <class F>
struct FreshInst {        // a fresh (strict) instance of the OTHER circuit
  Affine<F> W;
  F x0, x1;               // 250-bit hash outputs: the same integers in both fields
  static FreshInst zero() { FreshInst r; r.W.x = r.W.y = r.x0 = r.x1 = F::zero(); return r; }
};

template <class F> inline U256w to_u256(const F& mont) { F c = F::from_mont(mont); U256w o; for (int i = 0; i < 4; i++) o.w[i] = (uint64_t)c.v[2 * i] | ((uint64_t)c.v[2 * i + 1] << 32); return o; }
template <class F> inline F from_u256(const U256w& x) { F c; for (int i = 0; i < 4; i++) { c.v[2 * i] = (uint32_t)x.w[i]; c.v[2 * i + 1] = (uint32_t)(x.w[i] >> 32); } return F::to_mont(c); }
template <class F> inline F trunc250(const F& mont) { F c = F::from_mont(mont); c.v[7] &= 0x03ffffffu; return F::to_mont(c); }

// The elements a relaxed instance contributes to the hashes, in order.
template <class F>
// (u and the IO limbs first: they only need the folding challenge, so the permutation that absorbs them — the first of the two — runs
//  while the commitments are still being folded; the commitments go into the second)
inline void absorb_relaxed(const RelaxedInst<F>& U, std::vector<F>& out) {
  out.push_back(U.u);
  for (int j = 0; j < 4; j++) out.push_back(cb::f_from_u64<F>(U.X0.w[j]));
  for (int j = 0; j < 4; j++) out.push_back(cb::f_from_u64<F>(U.X1.w[j]));
  out.push_back(U.W.x); out.push_back(U.W.y); out.push_back(U.E.x); out.push_back(U.E.y);
}
// trunc250(H(digest, i, z_0, z, U)) outside any circuit (verifier; also the prover's bookkeeping)
template <class FP>
inline Fp<FP> instance_hash_native(const Fp<FP>& digest, uint64_t i, const std::vector<Fp<FP>>& z0, const std::vector<Fp<FP>>& z, const RelaxedInst<Fp<FP>>& U, Fp<FP>* full = nullptr) {
  typedef Fp<FP> F;
  // two levels: H(H(digest, i, z_0, z_i), U) — the statement part does not depend on the folding challenge, so the prover has it
  // (and its S-box wires) ready before the commitments a step waits for arrive (AugCache::next_pre)
  std::vector<F> st; st.push_back(digest); st.push_back(cb::f_from_u64<F>(i));
  st.insert(st.end(), z0.begin(), z0.end());
  st.insert(st.end(), z.begin(), z.end());
  std::vector<F> in; in.push_back(hash_native<FP>(st));
  absorb_relaxed(U, in);
  F h = hash_native<FP>(in);
  if (full) *full = h;
  return trunc250(h);
}

template <class FP>
struct AugIn {
  typedef Fp<FP> F;
  F digest; uint64_t i = 0;
  std::vector<F> z0;       // the initial state the chain claims to start from (len_z elements; empty = zeros)
  RelaxedInst<F> U; FreshInst<F> u; Affine<F> T;
};
template <class FP>
struct AugOut {
  typedef Fp<FP> F;
  RelaxedInst<F> U_new;
  uint32_t rho_low[4];     // low 128 bits of the challenge (rho = 2^128 + this)
  F x0, x1;                // this instance's public IO
  uint32_t x0_wire = 0, x1_wire = 0;
};

// FP: the circuit's field; OP: the other field of the cycle.  z_i / z_next: the step circuit's state wires (already in the
// constraint system).  curve_b, G: the other curve (y^2 = x^3 + b over F) and a fixed finite point on it.
#ifdef VZ_AUG_TIMING
extern double g_t[16]; extern const char* g_n[16];
#define VZ_T(k, name) do { const double _n = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); g_n[k] = name; g_t[k] += _n - _tl; _tl = _n; } while (0)
#define VZ_T0() double _tl = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count()
#else
#define VZ_T(k, name) do {} while (0)
#define VZ_T0() do {} while (0)
#endif
template <class FP, class OP>
AugOut<FP> synthesize_augmented(CS<FP>& cs, const AugIn<FP>& in, const std::vector<Num<Fp<FP>>>& z_i, const std::vector<Num<Fp<FP>>>& z_next,
                                bool is_primary, const Fp<FP>& curve_b, const Affine<Fp<FP>>& G, AugCache<Fp<FP>>* cache = nullptr) {
  typedef Fp<FP> F;
  typedef Num<F> N;
  typedef EcGadgets<FP> Ec;
  typedef typename Ec::Pt Pt;
  typedef NonNative<FP, OP> NN;
  Ec ec(cs, curve_b, G);
  AugOut<FP> out;
  VZ_T0();

  // ---- inputs --------------------------------------------------------------------------------------------------------
  N dg = cs.alloc(in.digest);
  std::vector<N> z0(z_i.size());
  for (size_t k = 0; k < z_i.size(); k++) z0[k] = cs.alloc(k < in.z0.size() ? in.z0[k] : F::zero());
  N iN = cs.alloc(cb::f_from_u64<F>(in.i));
  std::vector<F> inv = {iN.v, in.U.W.y, in.U.E.y, in.u.W.y, in.T.y};
  batch_inv(inv);
  auto alloc_pt = [&](const Affine<F>& p, const F& yinv, bool check) {
    Pt r; r.x = cs.alloc(p.x); r.y = cs.alloc(p.y); r.inf = cs.is_zero(r.y, &yinv);
    if (check) {
      cs.enforce(r.x, r.inf, cs.zero());
      N xx = cs.mul(r.x, r.x); N xxx = cs.mul(xx, r.x);
      cs.enforce(r.y, r.y, cs.add(xxx, cs.scale(cs.one_minus(r.inf), curve_b)));
    }
    return r;
  };
  Pt UW = alloc_pt(in.U.W, inv[1], false), UE = alloc_pt(in.U.E, inv[2], false);
  N Uu = cs.alloc(in.U.u);
  N UX0[4], UX1[4];
  for (int j = 0; j < 4; j++) UX0[j] = cs.alloc(cb::f_from_u64<F>(in.U.X0.w[j]));
  for (int j = 0; j < 4; j++) UX1[j] = cs.alloc(cb::f_from_u64<F>(in.U.X1.w[j]));
  Pt uW = alloc_pt(in.u.W, inv[3], true);
  N ux0 = cs.alloc(in.u.x0), ux1 = cs.alloc(in.u.x1);
  Pt T = alloc_pt(in.T, inv[4], true);
  N is_base = cs.is_zero(iN, &inv[0]);
  N nb = cs.one_minus(is_base);
  const bool base = in.i == 0;
  // the base case starts from the claimed initial state: is_base·(z_i − z_0) = 0
  for (size_t k = 0; k < z_i.size(); k++) {
    cs.enforce(is_base, cs.sub(z_i[k], z0[k]), cs.zero());
    if (base && !z_i[k].v.eq(z0[k].v)) cs.bad = true;
  }

  VZ_T(0, "inputs");
  // ---- consistency of the incoming instance with the previous step's output hash ---------------------------------------
  std::vector<N> hst; hst.push_back(dg); hst.push_back(iN);
  hst.insert(hst.end(), z0.begin(), z0.end());
  hst.insert(hst.end(), z_i.begin(), z_i.end());
  std::vector<N> hin; hin.push_back(cs.hash_cached(hst, cache ? &cache->pre : nullptr, nullptr));
  hin.push_back(Uu);
  for (int j = 0; j < 4; j++) hin.push_back(UX0[j]);
  for (int j = 0; j < 4; j++) hin.push_back(UX1[j]);
  hin.push_back(UW.x); hin.push_back(UW.y); hin.push_back(UE.x); hin.push_back(UE.y);
  N h_chk = cs.hash_cached(hin, cache ? &cache->rest : nullptr, nullptr);
  VZ_T(1, "hash_in");
  std::vector<N> hb = cs.bits(h_chk, FP::BITS);
  N h250 = cs.pack(hb, 0, 250);
  cs.enforce(nb, cs.sub(h250, ux0), cs.zero());
  if (!base && !h250.v.eq(ux0.v)) cs.bad = true;

  VZ_T(2, "bits_in");
  // ---- challenge ----------------------------------------------------------------------------------------------------------
  N hr = cs.hash({h_chk, uW.x, uW.y, ux0, ux1, T.x, T.y});
  VZ_T(3, "hash_rho");
  std::vector<N> rb = cs.bits_strict(hr);      // (canonical: of the two bit vectors that sum to hr mod p only one is accepted — ADVICE r3; the truncated instance hashes above and below stay plain: both of their candidate values are functions of the same preimage)
  N rho0 = cs.pack(rb, 0, 64), rho1 = cs.pack(rb, 64, 128);
  { F c = F::from_mont(hr.v); for (int k = 0; k < 4; k++) out.rho_low[k] = c.v[k]; }
  N rho = cs.add(cs.add(rho0, cs.scale(rho1, cb::f_pow2<F>(64))), cs.constant(cb::f_pow2<F>(128)));

  // ---- NIFS.V --------------------------------------------------------------------------------------------------------------
  // The two 128-step scalar multiplications are the long pole of the witness: their chains (cs.hpp: chain_hints) start now, on
  // helper threads, and everything below that only needs the challenge is evaluated under them.
  typename Ec::ChainJob chains;
  chains.start(Ec::scalar_operand(in.u.W, G), Ec::scalar_operand(in.T, G), out.rho_low, 128, cs.worker, cs.worker2);
  Pt We, Ee;   // the running instance, or the zero instance in the base case
  We.x = cs.mul(nb, UW.x); We.y = cs.mul(nb, UW.y); We.inf = cs.add(UW.inf, cs.mul(is_base, cs.one_minus(UW.inf)));
  Ee.x = cs.mul(nb, UE.x); Ee.y = cs.mul(nb, UE.y); Ee.inf = cs.add(UE.inf, cs.mul(is_base, cs.one_minus(UE.inf)));
  N ue = cs.mul(nb, Uu);
  N X0e[4], X1e[4];
  for (int j = 0; j < 4; j++) { X0e[j] = cs.mul(nb, UX0[j]); X1e[j] = cs.mul(nb, UX1[j]); }
  U256w X0v = in.U.X0, X1v = in.U.X1;
  if (base) for (int j = 0; j < 4; j++) X0v.w[j] = X1v.w[j] = 0;

  std::vector<N> xb0 = cs.bits(ux0, 250), xb1 = cs.bits(ux1, 250);
  VZ_T(4, "select_bits");
  N un = cs.add(ue, rho);
  N X0n[4], X1n[4]; U256w X0nv, X1nv;
  NN::fold(cs, X0e, X0v, rho0, rho1, out.rho_low, xb0, to_u256(in.u.x0), X0n, X0nv);
  NN::fold(cs, X1e, X1v, rho0, rho1, out.rho_low, xb1, to_u256(in.u.x1), X1n, X1nv);
  VZ_T(8, "nonnative");
  // the primary's base case outputs the zero instance (its incoming fresh instance is a dummy)
  N uo = un;
  if (is_primary) {
    uo = cs.mul(nb, un);
    for (int j = 0; j < 4; j++) { X0n[j] = cs.mul(nb, X0n[j]); X1n[j] = cs.mul(nb, X1n[j]); }
    if (base) for (int j = 0; j < 4; j++) X0nv.w[j] = X1nv.w[j] = 0;
  }
  // ---- output hash, first part: the statement (computed ahead, AugCache::next_pre) and the permutation that absorbs u and the IO limbs
  std::vector<N> host; host.push_back(dg); host.push_back(cs.addc(iN, F::one()));
  host.insert(host.end(), z0.begin(), z0.end());
  host.insert(host.end(), z_next.begin(), z_next.end());
  std::vector<N> hout; hout.push_back(cs.hash_cached(host, cache ? &cache->next_pre : nullptr, cache ? &cache->pre : nullptr));
  hout.push_back(uo);
  for (int j = 0; j < 4; j++) hout.push_back(X0n[j]);
  for (int j = 0; j < 4; j++) hout.push_back(X1n[j]);
  VZ_T(9, "sel_out");
  const size_t w_h1 = cs.w.size();
  N h1 = cs.poseidon(std::vector<N>(hout.begin(), hout.begin() + 8));      // (hash() of the 14 elements = this, then h1 with the other six)
  const size_t w_h1_end = cs.w.size();
  VZ_T(10, "hash_out_1");

  // ---- the folded commitments ------------------------------------------------------------------------------------------------------
  // (the prover's hook: what only waits for the challenge on the device — the folds, the next cross term and its commitment — is
  //  queued here, in the time this thread would otherwise spend waiting for the chains.  Right after the chains' start instead: the
  //  large MSM begins 60 µs earlier still, its accumulation lands on the secondary half — 659 against 677 steps/s)
  if (cs.on_challenge && *cs.on_challenge) (*cs.on_challenge)(out.rho_low);
  std::vector<typename Ec::ChainHints> hints;
  chains.wait(hints);
  VZ_T(5, "chain_hints");
  Pt rW = ec.scalar_mul(uW, rb, 128, hints[0]);
  Pt rT = ec.scalar_mul(T, rb, 128, hints[1]);
  VZ_T(6, "scalar_mul_gadget");
  Pt Wn = ec.add(We, rW), En = ec.add(Ee, rT);
  VZ_T(7, "ec_add");
  Pt Wo = Wn, Eo = En;
  if (is_primary) {
    Wo.x = cs.mul(nb, Wn.x); Wo.y = cs.mul(nb, Wn.y);
    Eo.x = cs.mul(nb, En.x); Eo.y = cs.mul(nb, En.y);
  }
  out.U_new.W.x = Wo.x.v; out.U_new.W.y = Wo.y.v; out.U_new.E.x = Eo.x.v; out.U_new.E.y = Eo.y.v; out.U_new.u = uo.v;
  out.U_new.X0 = X0nv; out.U_new.X1 = X1nv;

  // ---- output hash, second part ---------------------------------------------------------------------------------------------------------
  hout.push_back(Wo.x); hout.push_back(Wo.y); hout.push_back(Eo.x); hout.push_back(Eo.y);
  const size_t w_h2 = cs.w.size();
  std::vector<N> h2in; h2in.push_back(h1);
  h2in.insert(h2in.end(), hout.begin() + 8, hout.end());
  N h_new = cs.poseidon(h2in);
  if (cache && !cs.b) {      // kept for the next step's incoming-hash check: inputs, the two permutations' wires in order, the value
    HashCache<F>& hc = cache->rest;
    hc.in.resize(hout.size());
    for (size_t k = 0; k < hout.size(); k++) hc.in[k] = hout[k].v;
    hc.wires.assign(cs.w.begin() + w_h1, cs.w.begin() + w_h1_end);
    hc.wires.insert(hc.wires.end(), cs.w.begin() + w_h2, cs.w.end());
    hc.out = h_new.v; hc.valid = true;
  }
  VZ_T(11, "hash_out_2");
  std::vector<N> hnb = cs.bits(h_new, FP::BITS);
  N hn250 = cs.pack(hnb, 0, 250);

  // ---- public IO (the last two wires of the circuit) -------------------------------------------------------------------------
  N p0 = cs.alloc(ux1.v), p1 = cs.alloc(hn250.v);
  cs.enforce_equal(p0, ux1);
  cs.enforce_equal(p1, hn250);
  out.x0 = p0.v; out.x1 = p1.v;
  out.x0_wire = cs.base + (uint32_t)cs.w.size() - 2; out.x1_wire = out.x0_wire + 1;
  VZ_T(12, "tail");
  return out;
}

}  // namespace aug
}  // namespace vz
