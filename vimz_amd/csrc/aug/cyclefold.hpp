// Nova with CycleFold (Kothapalli–Setty, "CycleFold", 2023) on the BN254 / Grumpkin cycle: the folding scheme of the reference's second
// backend, `Nova<G1, G2, C, KZG, Pedersen, false>` (vimz/src/sonobe_backend/folding.rs:22), whose `prove_step` loop is
// vimz/src/sonobe_backend/folding.rs:52-66.  SURVEY.md §8 row N1.  Sonobe (folding-schemes @ d312916, vimz/Cargo.toml:46-48) is not
// vendored with the reference: the two circuits below state the construction of the paper with the interface Sonobe documents (two
// public IOs per main instance, one CycleFold instance per folded commitment, 128-bit challenges) — our wiring, not a byte-compatible
// copy.  Gadgets: cs.hpp (Poseidon, curve arithmetic, non-native folds).
//
// MAIN circuit F' over BN254 Fr, appended to the step circuit F like Nova's (augmented.hpp: same wire layout, the two public IOs last).
// Instances of F' are committed on BN254 G1 (the KZG SRS or any other generators: vimz_bases_upload), so their commitments have
// coordinates in Fq: NON-native here.  F' therefore folds only the SCALARS of the main instance natively,
//     u' = u + r,   x' = x + r·x_in        (u_in = 1, no error term on the incoming instance)
// takes the folded commitments W' = W + r·W_in and E' = E + r·T as hints, and has their correctness shown by two instances of the
// CYCLEFOLD circuit over Fq (where G1 arithmetic is native):  public (r, P1, P2, P3),  P3 = P1 + r·P2.  Those instances are committed
// on Grumpkin (native coordinates here) and F' folds them — two per step — into a running CycleFold instance: commitments by
// in-circuit curve arithmetic, the seven public elements each by the non-native fold  x' = x + r_cf·x_in (mod q).
//     is_base = (i == 0);   z_i == z_0 if is_base
//     u_in.x0 == H(H(dg, i, z_0, z_i), U)                      unless is_base        U = (u, x0, x1, limbs of W, limbs of E)
//     u_in.x1 == H(dg, cfU)                                     unless is_base        cfU = (u, 28 limbs of x, W, E)
//     r    = 2^128 + low128(hr),   hr = H(h_U, limbs(W_in), x_in, limbs(T))
//     r_1  = 2^128 + low128(h1),   h1 = H(h_cf, hr, cf1.W, limbs(W'), cf1.T)        cf1.x = (r, U.W, W_in, W')
//     r_2  = 2^128 + low128(h2),   h2 = H(h1, cf2.W, limbs(E'), cf2.T)              cf2.x = (r, U.E, T,    E')
//     cfU' = NIFS.V(NIFS.V(cfU, cf1, r_1), cf2, r_2);     base case: U' and cfU' are the zero instances
//     public IO:  x0 = H(H(dg, i+1, z_0, z_{i+1}), U'),  x1 = H(dg, cfU')
// The IVC proof after n steps is (U_n, W_n), (u_n, w_n), (cfU_n, cfW_n): the verifier recomputes the two hashes from the claimed z_0,
// z_n and n, and checks the three instances against their witnesses (Sonobe's Nova::verify).
#pragma once
#include "augmented.hpp"

namespace vz {
namespace aug {

typedef Fp<BnFr> CfFr;
typedef Fp<BnFq> CfFq;

struct NnPoint {                 // a BN254 G1 point as the main circuit sees it: canonical coordinates, identity = (0, 0)
  U256w x, y;
  static NnPoint zero() { NnPoint p; for (int i = 0; i < 4; i++) p.x.w[i] = p.y.w[i] = 0; return p; }
};
inline NnPoint nn_point(const Affine<CfFq>& p) { NnPoint r; r.x = to_u256(p.x); r.y = to_u256(p.y); return r; }

struct CfMainRelaxed {           // running main instance
  NnPoint W, E; CfFr u, x0, x1;
  static CfMainRelaxed zero() { CfMainRelaxed r; r.W = r.E = NnPoint::zero(); r.u = r.x0 = r.x1 = CfFr::zero(); return r; }
};
struct CfMainFresh { NnPoint W; CfFr x0, x1; static CfMainFresh zero() { CfMainFresh r; r.W = NnPoint::zero(); r.x0 = r.x1 = CfFr::zero(); return r; } };
constexpr int CF_IO = 7;         // public elements of a CycleFold instance: r, P1, P2, P3
struct CfRelaxed {               // running CycleFold instance: commitments on Grumpkin, u a small integer, x elements of Fq
  Affine<CfFr> W, E; CfFr u; U256w x[CF_IO];
  static CfRelaxed zero() { CfRelaxed r; r.W.x = r.W.y = r.E.x = r.E.y = r.u = CfFr::zero(); for (auto& e : r.x) for (int i = 0; i < 4; i++) e.w[i] = 0; return r; }
};

inline void cf_push_limbs(const U256w& v, std::vector<CfFr>& out) { for (int j = 0; j < 4; j++) out.push_back(cb::f_from_u64<CfFr>(v.w[j])); }
// H(H(dg, i, z_0, z), U) and H(dg, cfU) outside any circuit (verifier; the prover's bookkeeping)
inline CfFr cf_hash_main(const CfFr& dg, uint64_t i, const std::vector<CfFr>& z0, const CfFr* z, const CfMainRelaxed& U) {
  std::vector<CfFr> st = {dg, cb::f_from_u64<CfFr>(i)};
  st.insert(st.end(), z0.begin(), z0.end());
  st.insert(st.end(), z, z + z0.size());
  std::vector<CfFr> in = {hash_native<BnFr>(st), U.u, U.x0, U.x1};
  cf_push_limbs(U.W.x, in); cf_push_limbs(U.W.y, in); cf_push_limbs(U.E.x, in); cf_push_limbs(U.E.y, in);
  return hash_native<BnFr>(in);
}
inline CfFr cf_hash_cf(const CfFr& dg, const CfRelaxed& U) {
  std::vector<CfFr> in = {dg, U.u};
  for (auto& e : U.x) cf_push_limbs(e, in);
  in.push_back(U.W.x); in.push_back(U.W.y); in.push_back(U.E.x); in.push_back(U.E.y);
  return hash_native<BnFr>(in);
}
inline void cf_low128(const CfFr& h, uint32_t out[4]) { const CfFr c = CfFr::from_mont(h); for (int k = 0; k < 4; k++) out[k] = c.v[k]; }
inline U256w cf_challenge_u256(const uint32_t low[4]) { U256w r; r.w[0] = (uint64_t)low[0] | ((uint64_t)low[1] << 32); r.w[1] = (uint64_t)low[2] | ((uint64_t)low[3] << 32); r.w[2] = 1; r.w[3] = 0; return r; }
// the three challenges of a step, as the circuit derives them
struct CfChallenges { CfFr h_U, h_cf, hr, h1, h2; uint32_t r[4], r1[4], r2[4]; };
inline void cf_challenge_main(CfChallenges& c, const CfMainFresh& u, const NnPoint& T) {
  std::vector<CfFr> in = {c.h_U};
  cf_push_limbs(u.W.x, in); cf_push_limbs(u.W.y, in); in.push_back(u.x0); in.push_back(u.x1); cf_push_limbs(T.x, in); cf_push_limbs(T.y, in);
  c.hr = hash_native<BnFr>(in); cf_low128(c.hr, c.r);
}
inline void cf_challenge_cf1(CfChallenges& c, const Affine<CfFr>& W, const NnPoint& Wn, const Affine<CfFr>& T) {
  std::vector<CfFr> in = {c.h_cf, c.hr, W.x, W.y};
  cf_push_limbs(Wn.x, in); cf_push_limbs(Wn.y, in); in.push_back(T.x); in.push_back(T.y);
  c.h1 = hash_native<BnFr>(in); cf_low128(c.h1, c.r1);
}
inline void cf_challenge_cf2(CfChallenges& c, const Affine<CfFr>& W, const NnPoint& En, const Affine<CfFr>& T) {
  std::vector<CfFr> in = {c.h1, W.x, W.y};
  cf_push_limbs(En.x, in); cf_push_limbs(En.y, in); in.push_back(T.x); in.push_back(T.y);
  c.h2 = hash_native<BnFr>(in); cf_low128(c.h2, c.r2);
}

struct CfMainIn {
  CfFr digest; uint64_t i = 0; std::vector<CfFr> z0;
  CfMainRelaxed U; CfMainFresh u; NnPoint T;      // the fold this step's circuit verifies
  NnPoint Wn, En;                                  // hints: the folded commitments (the identity in the base case)
  CfRelaxed cfU;
  Affine<CfFr> cf1W, cf1T, cf2W, cf2T;             // the two CycleFold instances' witness commitments, and their folds' cross-term commitments
  static CfMainIn zero() {
    CfMainIn in; in.digest = CfFr::zero(); in.U = CfMainRelaxed::zero(); in.u = CfMainFresh::zero(); in.T = in.Wn = in.En = NnPoint::zero(); in.cfU = CfRelaxed::zero();
    for (auto* p : {&in.cf1W, &in.cf1T, &in.cf2W, &in.cf2T}) p->x = p->y = CfFr::zero();
    return in;
  }
};
struct CfMainOut {
  CfMainRelaxed U_new; CfRelaxed cfU_new;
  uint32_t r[4], r1[4], r2[4];
  CfFr x0, x1;
};

// The four hashes a step ends with are the ones the next step's circuit recomputes to check its incoming instance: their wires are kept
// and replayed (cs.hpp: HashCache) — st = H(dg, i, z_0, z), U = the main instance's, cf_pre / cf_fin = the CycleFold instance's in two parts
// (the first 29 elements — u and the limbs — end on a block boundary of the sponge; the last block absorbs the commitments).
struct CfHashCache { HashCache<CfFr> st, U, cf_pre, cf_fin; };
inline CfMainOut synthesize_cf_main(CS<BnFr>& cs, const CfMainIn& in, const std::vector<Num<CfFr>>& z_i, const std::vector<Num<CfFr>>& z_next, CfHashCache* cache = nullptr) {
  typedef CfFr F;
  typedef Num<F> N;
  typedef EcGadgets<BnFr> Ec;
  typedef Ec::Pt Pt;
  typedef NonNative<BnFr, BnFq> NN;
  Ec ec(cs, CycleSide<BnFr>::b(), CycleSide<BnFr>::G());
  CfMainOut out;
  struct NnVar { N x[4], y[4]; NnPoint v; };
  auto nn_alloc = [&](const NnPoint& p, bool range) {
    NnVar r; r.v = p;
    for (int j = 0; j < 4; j++) { r.x[j] = cs.alloc(cb::f_from_u64<F>(p.x.w[j])); if (range) cs.bits(r.x[j], 64); }
    for (int j = 0; j < 4; j++) { r.y[j] = cs.alloc(cb::f_from_u64<F>(p.y.w[j])); if (range) cs.bits(r.y[j], 64); }
    return r;
  };
  auto push_nn = [](std::vector<N>& h, const NnVar& p) { for (int j = 0; j < 4; j++) h.push_back(p.x[j]); for (int j = 0; j < 4; j++) h.push_back(p.y[j]); };

  // ---- inputs ------------------------------------------------------------------------------------------------------------------------
  N dg = cs.alloc(in.digest);
  std::vector<N> z0(z_i.size());
  for (size_t k = 0; k < z_i.size(); k++) z0[k] = cs.alloc(k < in.z0.size() ? in.z0[k] : F::zero());
  N iN = cs.alloc(cb::f_from_u64<F>(in.i));
  // (limbs that arrive through a checked hash were produced — and range-checked — by the previous instance of this circuit)
  NnVar UW = nn_alloc(in.U.W, false), UE = nn_alloc(in.U.E, false);
  N Uu = cs.alloc(in.U.u), Ux0 = cs.alloc(in.U.x0), Ux1 = cs.alloc(in.U.x1);
  NnVar uW = nn_alloc(in.u.W, true);
  N ux0 = cs.alloc(in.u.x0), ux1 = cs.alloc(in.u.x1);
  NnVar T = nn_alloc(in.T, true), Wn = nn_alloc(in.Wn, true), En = nn_alloc(in.En, true);
  std::vector<F> inv = {iN.v, in.cfU.W.y, in.cfU.E.y, in.cf1W.y, in.cf1T.y, in.cf2W.y, in.cf2T.y};
  batch_inv(inv);
  auto alloc_pt = [&](const Affine<F>& p, const F& yinv, bool check) {
    Pt r; r.x = cs.alloc(p.x); r.y = cs.alloc(p.y); r.inf = cs.is_zero(r.y, &yinv);
    if (check) {
      cs.enforce(r.x, r.inf, cs.zero());
      N xx = cs.mul(r.x, r.x); N xxx = cs.mul(xx, r.x);
      cs.enforce(r.y, r.y, cs.add(xxx, cs.scale(cs.one_minus(r.inf), ec.curve_b)));
    }
    return r;
  };
  Pt cW = alloc_pt(in.cfU.W, inv[1], false), cE = alloc_pt(in.cfU.E, inv[2], false);
  N cu = cs.alloc(in.cfU.u);
  N cx[CF_IO][4];
  for (int k = 0; k < CF_IO; k++) for (int j = 0; j < 4; j++) cx[k][j] = cs.alloc(cb::f_from_u64<F>(in.cfU.x[k].w[j]));
  Pt c1W = alloc_pt(in.cf1W, inv[3], true), c1T = alloc_pt(in.cf1T, inv[4], true), c2W = alloc_pt(in.cf2W, inv[5], true), c2T = alloc_pt(in.cf2T, inv[6], true);
  N is_base = cs.is_zero(iN, &inv[0]);
  N nb = cs.one_minus(is_base);
  const bool base = in.i == 0;
  for (size_t k = 0; k < z_i.size(); k++) {
    cs.enforce(is_base, cs.sub(z_i[k], z0[k]), cs.zero());
    if (base && !z_i[k].v.eq(z0[k].v)) cs.bad = true;
  }

  // ---- the incoming instance carries the hashes of the two running instances ----------------------------------------------------------------
  std::vector<N> hst = {dg, iN};
  hst.insert(hst.end(), z0.begin(), z0.end());
  hst.insert(hst.end(), z_i.begin(), z_i.end());
  std::vector<N> hin = {cs.hash_cached(hst, cache ? &cache->st : nullptr, nullptr), Uu, Ux0, Ux1};
  push_nn(hin, UW); push_nn(hin, UE);
  N h_U = cs.hash_cached(hin, cache ? &cache->U : nullptr, nullptr);
  cs.enforce(nb, cs.sub(h_U, ux0), cs.zero());
  if (!base && !h_U.v.eq(ux0.v)) cs.bad = true;
  std::vector<N> hcin = {dg, cu};
  for (int k = 0; k < CF_IO; k++) for (int j = 0; j < 4; j++) hcin.push_back(cx[k][j]);
  static_assert(2 + 4 * CF_IO == 30, "the sponge absorbs 8 + 7 + 7 + 7 = 29 elements in four blocks; the rest and the commitments go into the fifth");
  N h_cf_pre = cs.hash_cached(std::vector<N>(hcin.begin(), hcin.begin() + 29), cache ? &cache->cf_pre : nullptr, nullptr);
  N h_cf = cs.hash_cached({h_cf_pre, hcin[29], cW.x, cW.y, cE.x, cE.y}, cache ? &cache->cf_fin : nullptr, nullptr);
  cs.enforce(nb, cs.sub(h_cf, ux1), cs.zero());
  if (!base && !h_cf.v.eq(ux1.v)) cs.bad = true;

  // ---- challenge and the native half of NIFS.V on the main instance --------------------------------------------------------------------------
  std::vector<N> hrin = {h_U};
  push_nn(hrin, uW); hrin.push_back(ux0); hrin.push_back(ux1); push_nn(hrin, T);
  N hr = cs.hash(hrin);
  std::vector<N> rb = cs.bits_strict(hr);
  N rho0 = cs.pack(rb, 0, 64), rho1 = cs.pack(rb, 64, 128);
  cf_low128(hr.v, out.r);
  N rho = cs.add(cs.add(rho0, cs.scale(rho1, cb::f_pow2<F>(64))), cs.constant(cb::f_pow2<F>(128)));
  N un = cs.mul(nb, cs.add(Uu, rho));
  N x0n = cs.mul(nb, cs.add(Ux0, cs.mul(rho, ux0))), x1n = cs.mul(nb, cs.add(Ux1, cs.mul(rho, ux1)));
  // the base case outputs the zero instance: its hinted commitments must be the identity
  for (int j = 0; j < 4; j++) for (const N* l : {&Wn.x[j], &Wn.y[j], &En.x[j], &En.y[j]}) { cs.enforce(is_base, *l, cs.zero()); if (base && !l->v.is_zero()) cs.bad = true; }
  out.U_new.W = base ? NnPoint::zero() : in.Wn; out.U_new.E = base ? NnPoint::zero() : in.En;
  out.U_new.u = un.v; out.U_new.x0 = x0n.v; out.U_new.x1 = x1n.v;

  // ---- the two CycleFold instances: public elements as limbs ------------------------------------------------------------------------------------
  struct Elem { N l[4]; U256w v; };
  auto elem = [](const N* l, const U256w& v) { Elem e; for (int j = 0; j < 4; j++) e.l[j] = l[j]; e.v = v; return e; };
  Elem er; er.l[0] = rho0; er.l[1] = rho1; er.l[2] = cs.one(); er.l[3] = cs.zero(); er.v = cf_challenge_u256(out.r);
  const Elem cf1x[CF_IO] = {er, elem(UW.x, in.U.W.x), elem(UW.y, in.U.W.y), elem(uW.x, in.u.W.x), elem(uW.y, in.u.W.y), elem(Wn.x, in.Wn.x), elem(Wn.y, in.Wn.y)};
  const Elem cf2x[CF_IO] = {er, elem(UE.x, in.U.E.x), elem(UE.y, in.U.E.y), elem(T.x, in.T.x), elem(T.y, in.T.y), elem(En.x, in.En.x), elem(En.y, in.En.y)};

  // ---- their challenges ----------------------------------------------------------------------------------------------------------------------------
  std::vector<N> h1in = {h_cf, hr, c1W.x, c1W.y};
  push_nn(h1in, Wn); h1in.push_back(c1T.x); h1in.push_back(c1T.y);
  N h1 = cs.hash(h1in);
  std::vector<N> r1b = cs.bits_strict(h1);
  N r1_0 = cs.pack(r1b, 0, 64), r1_1 = cs.pack(r1b, 64, 128);
  cf_low128(h1.v, out.r1);
  std::vector<N> h2in = {h1, c2W.x, c2W.y};
  push_nn(h2in, En); h2in.push_back(c2T.x); h2in.push_back(c2T.y);
  N h2 = cs.hash(h2in);
  std::vector<N> r2b = cs.bits_strict(h2);
  N r2_0 = cs.pack(r2b, 0, 64), r2_1 = cs.pack(r2b, 64, 128);
  cf_low128(h2.v, out.r2);
  const F p128 = cb::f_pow2<F>(128), p64 = cb::f_pow2<F>(64);
  N r1 = cs.add(cs.add(r1_0, cs.scale(r1_1, p64)), cs.constant(p128)), r2 = cs.add(cs.add(r2_0, cs.scale(r2_1, p64)), cs.constant(p128));

  // ---- NIFS.V on the running CycleFold instance, twice ---------------------------------------------------------------------------------------------
  Pt We, Ee;        // the running instance, or the zero instance in the base case
  We.x = cs.mul(nb, cW.x); We.y = cs.mul(nb, cW.y); We.inf = cs.add(cW.inf, cs.mul(is_base, cs.one_minus(cW.inf)));
  Ee.x = cs.mul(nb, cE.x); Ee.y = cs.mul(nb, cE.y); Ee.inf = cs.add(cE.inf, cs.mul(is_base, cs.one_minus(cE.inf)));
  N ue = cs.mul(nb, cu);
  N xe[CF_IO][4]; U256w xev[CF_IO];
  for (int k = 0; k < CF_IO; k++) { for (int j = 0; j < 4; j++) xe[k][j] = cs.mul(nb, cx[k][j]); xev[k] = in.cfU.x[k]; if (base) for (int j = 0; j < 4; j++) xev[k].w[j] = 0; }
  // The four 128-step scalar multiplications are the long pole of the witness: both pairs' chains (cs.hpp: chain_hints, one batch per
  // pair) start now, on the two helper threads; the non-native folds and the hashes that do not need the folded commitments run under them.
  struct PairJob {
    std::vector<Affine<F>> ps; std::vector<Ec::ChainHints> o; const uint32_t* k = nullptr; Worker* w = nullptr;
    void start(const Affine<F>& a, const Affine<F>& b2, const uint32_t* k_, Worker* w_) { ps = {a, b2}; k = k_; w = w_; if (w) w->start([this] { Ec::chain_hints(ps, k, 128, o); }); }
    void wait() { if (w) w->wait(); else Ec::chain_hints(ps, k, 128, o); }
  } job1, job2;
  job1.start(Ec::scalar_operand(in.cf1W, ec.G), Ec::scalar_operand(in.cf1T, ec.G), out.r1, cs.worker);
  job2.start(Ec::scalar_operand(in.cf2W, ec.G), Ec::scalar_operand(in.cf2T, ec.G), out.r2, cs.worker2);
  N u1 = cs.add(ue, r1), u2 = cs.add(u1, r2);
  N x1v[CF_IO][4]; U256w x1vv[CF_IO];
  for (int k = 0; k < CF_IO; k++) NN::fold_limbs(cs, xe[k], xev[k], r1_0, r1_1, out.r1, cf1x[k].l, cf1x[k].v, x1v[k], x1vv[k]);
  N x2v[CF_IO][4]; U256w x2vv[CF_IO];
  for (int k = 0; k < CF_IO; k++) NN::fold_limbs(cs, x1v[k], x1vv[k], r2_0, r2_1, out.r2, cf2x[k].l, cf2x[k].v, x2v[k], x2vv[k]);
  N ou = cs.mul(nb, u2);
  N ox[CF_IO][4];
  for (int k = 0; k < CF_IO; k++) for (int j = 0; j < 4; j++) { ox[k][j] = cs.mul(nb, x2v[k][j]); if (base) x2vv[k].w[j] = 0; }
  // public IO, the parts that do not wait for the chains: the hash of the new main instance, four of the five blocks of the CycleFold one
  std::vector<N> host = {dg, cs.addc(iN, F::one())};
  host.insert(host.end(), z0.begin(), z0.end());
  host.insert(host.end(), z_next.begin(), z_next.end());
  std::vector<N> hout = {cs.hash_cached(host, nullptr, cache ? &cache->st : nullptr), un, x0n, x1n};
  push_nn(hout, Wn); push_nn(hout, En);
  N hU_new = cs.hash_cached(hout, nullptr, cache ? &cache->U : nullptr);
  std::vector<N> hcout = {dg, ou};
  for (int k = 0; k < CF_IO; k++) for (int j = 0; j < 4; j++) hcout.push_back(ox[k][j]);
  N hcf_pre = cs.hash_cached(std::vector<N>(hcout.begin(), hcout.begin() + 29), nullptr, cache ? &cache->cf_pre : nullptr);
  // the folded commitments
  job1.wait(); job2.wait();
  Pt W1 = ec.add(We, ec.scalar_mul(c1W, r1b, 128, job1.o[0])), E1 = ec.add(Ee, ec.scalar_mul(c1T, r1b, 128, job1.o[1]));
  Pt W2 = ec.add(W1, ec.scalar_mul(c2W, r2b, 128, job2.o[0])), E2 = ec.add(E1, ec.scalar_mul(c2T, r2b, 128, job2.o[1]));
  // masked by the base case
  N oWx = cs.mul(nb, W2.x), oWy = cs.mul(nb, W2.y), oEx = cs.mul(nb, E2.x), oEy = cs.mul(nb, E2.y);
  out.cfU_new.W.x = oWx.v; out.cfU_new.W.y = oWy.v; out.cfU_new.E.x = oEx.v; out.cfU_new.E.y = oEy.v; out.cfU_new.u = ou.v;
  for (int k = 0; k < CF_IO; k++) out.cfU_new.x[k] = x2vv[k];
  N hcf_new = cs.hash_cached({hcf_pre, hcout[29], oWx, oWy, oEx, oEy}, nullptr, cache ? &cache->cf_fin : nullptr);
  N p0 = cs.alloc(hU_new.v), p1 = cs.alloc(hcf_new.v);
  cs.enforce_equal(p0, hU_new);
  cs.enforce_equal(p1, hcf_new);
  out.x0 = p0.v; out.x1 = p1.v;
  return out;
}

// ---- the CycleFold circuit over Fq: public (r, P1, P2, P3) with P3 = P1 + r·P2 on BN254 G1, r = 2^128 + 128 bits ----------------------------------
// Wire layout [1 | witness | r | P1.x | P1.y | P2.x | P2.y | P3.x | P3.y]: the seven public elements are the LAST seven wires.
struct CfCircuitOut { Affine<CfFq> P3; };
inline CfCircuitOut synthesize_cyclefold(CS<BnFq>& cs, const uint32_t r_low[4], const Affine<CfFq>& P1v, const Affine<CfFq>& P2v) {
  typedef CfFq F;
  typedef Num<F> N;
  typedef EcGadgets<BnFq> Ec;
  Ec ec(cs, CycleSide<BnFq>::b(), CycleSide<BnFq>::G());
  F lowv = F::zero(); for (int k = 0; k < 4; k++) lowv.v[k] = r_low[k];
  N low = cs.alloc(F::to_mont(lowv));
  std::vector<N> rb = cs.bits(low, 128);
  Ec::Pt P1 = ec.alloc(P1v, true), P2 = ec.alloc(P2v, true);
  std::vector<Ec::ChainHints> hints;
  Ec::chain_hints({Ec::scalar_operand(P2v, ec.G)}, r_low, 128, hints);
  Ec::Pt Q = ec.scalar_mul(P2, rb, 128, hints[0]);
  Ec::Pt P3 = ec.add(P1, Q);
  const N pub[CF_IO] = {cs.addc(low, cb::f_pow2<F>(128)), P1.x, P1.y, P2.x, P2.y, P3.x, P3.y};
  for (int k = 0; k < CF_IO; k++) { N p = cs.alloc(pub[k].v); cs.enforce_equal(p, pub[k]); }
  CfCircuitOut o; o.P3.x = P3.x.v; o.P3.y = P3.y.v;
  return o;
}

struct CfCircuit {
  cb::BuilderT<CfFq> b;
  void finish() {
    b = cb::BuilderT<CfFq>();
    CS<BnFq> cs; cs.b = &b; cs.base = b.n_wires;
    const uint32_t z[4] = {0, 0, 0, 0};
    Affine<CfFq> id; id.x = id.y = CfFq::zero();
    synthesize_cyclefold(cs, z, id, id);
  }
  uint32_t n_wires() const { return b.n_wires; }
  uint32_t n_constraints() const { return b.n_constraints(); }
  // wires[0] = 1, then the circuit's wires in order (Montgomery), the seven public elements last
  Affine<CfFq> witness(const uint32_t r_low[4], const Affine<CfFq>& P1, const Affine<CfFq>& P2, std::vector<CfFq>& wires, bool* bad) const {
    CS<BnFq> cs; cs.base = 1;
    cs.w.reserve(n_wires());
    CfCircuitOut o = synthesize_cyclefold(cs, r_low, P1, P2);
    if (cs.w.size() + 1 != n_wires()) throw std::runtime_error("cyclefold: witness length differs from the shape");
    if (bad) *bad = cs.bad;
    wires.clear(); wires.reserve(n_wires());
    wires.push_back(CfFq::one());
    wires.insert(wires.end(), cs.w.begin(), cs.w.end());
    return o.P3;
  }
};

// ---- the main circuit appended to a step circuit (the counterpart of AugCircuit<BnFr>) -----------------------------------------------------------------
struct CfMainCircuit {
  cb::BuilderT<CfFr>& b;
  explicit CfMainCircuit(cb::BuilderT<CfFr>& ext) : b(ext) {}
  uint32_t len_z = 0, step_wires = 0, step_constraints = 0;
  CfFr digest;                 // SHA3-256 of both shapes, truncated to 250 bits
  mutable std::unique_ptr<Worker> worker, worker2;
  mutable CfHashCache cache;            // the output hashes of the last witness() call, replayed by the next
  bool use_worker = affinity_cpus() > 2 && !getenv("VIMZ_AUG_NO_THREADS");      // (two helper threads per circuit: pointless on one or two cores)
  uint32_t n_wires() const { return b.n_wires; }
  uint32_t n_constraints() const { return b.n_constraints(); }
  uint32_t aug_wires() const { return b.n_wires - step_wires; }
  void finish(const CfCircuit& cf) {
    len_z = b.len_z; step_wires = b.n_wires; step_constraints = b.n_constraints();
    CS<BnFr> cs; cs.b = &b; cs.base = b.n_wires;
    std::vector<Num<CfFr>> zi, zn;
    for (uint32_t k = 0; k < len_z; k++) { zi.push_back(cs.wire(1 + len_z + k, CfFr::zero())); zn.push_back(cs.wire(1 + k, CfFr::zero())); }
    synthesize_cf_main(cs, CfMainIn::zero(), zi, zn);
    Sha3 h;
    const uint64_t hdr[6] = {0x31306d6663ull /* "cfm01" */, b.n_wires, b.n_constraints(), len_z, step_wires, cf.b.n_wires};
    h.update(hdr, sizeof(hdr));
    for (const cb::Csr* M : {&b.A, &b.B, &b.C}) { h.vec(M->row_ptr); h.vec(M->col); h.vec(M->coef); }
    h.vec(b.dict);
    for (const cb::Csr* M : {&cf.b.A, &cf.b.B, &cf.b.C}) { h.vec(M->row_ptr); h.vec(M->col); h.vec(M->coef); }
    h.vec(cf.b.dict);
    uint8_t d[32]; h.finish(d);
    CfFr c; memcpy(c.v, d, 32); c.v[7] &= 0x03ffffffu;
    digest = CfFr::to_mont(c);
  }
  // alias_attack (test hook of vimz_strict_bits_selfcheck, never set by the prover): decompose hash outputs as h + p where that fits;
  // returns how many decompositions were aliased
  CfMainOut witness(const CfMainIn& in, const CfFr* z_i, const CfFr* z_next, std::vector<CfFr>& aug, bool* bad, int* alias_attack = nullptr) const {
    CS<BnFr> cs; cs.base = step_wires; cs.alias_attack = alias_attack != nullptr;
    cs.w.reserve(aug_wires());
    if (use_worker) { if (!worker) worker.reset(new Worker()); if (!worker2) worker2.reset(new Worker()); cs.worker = worker.get(); cs.worker2 = worker2.get(); }
    std::vector<Num<CfFr>> zi(len_z), zn(len_z);
    for (uint32_t k = 0; k < len_z; k++) { zi[k].v = z_i[k]; zn[k].v = z_next[k]; }
    CfMainOut o = synthesize_cf_main(cs, in, zi, zn, &cache);
    if (cs.w.size() != aug_wires()) throw std::runtime_error("cyclefold main circuit: witness length differs from the shape");
    if (bad) *bad = cs.bad;
    if (alias_attack) *alias_attack = cs.alias_used;
    aug.swap(cs.w);
    return o;
  }
};

}  // namespace aug
}  // namespace vz
