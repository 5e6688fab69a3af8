// The decider circuit of the Nova + CycleFold path — the statement the reference's Sonobe backend proves with Groth16 before it goes on
// chain (`DeciderEth<.., Groth16<Bn254>, ..>`, vimz/src/sonobe_backend/decider.rs:13-21; `Decider::prove`, mod.rs:72-78; checked by
// contracts/*Verifier.sol:785-810).  Sonobe (folding-schemes @ d312916) is not vendored, so — like F' and the CycleFold circuit (aug/cyclefold.hpp) —
// this is OUR statement of the construction its documentation describes, parity unpinned: what the on-chain decider's SNARK attests about the
// final fold U_{i+1} = NIFS.V(U_i, u_i) of an IVC proof (U_i, u_i, cfU_i) for the statement (i, z_0, z_i):
//
//   public inputs   i, z_0, z_i, h_inst
//                   h_inst = H(dg, rho, U_i.cmW, U_i.cmE, u_i.cmW, cmT, U_{i+1}.cmW, U_{i+1}.cmE, c_W, c_E, e_W, e_E) binds the words the contract sees
//                   (commitments as 64-bit limbs): the verifier recomputes it from the calldata and checks U_{i+1}.cm* = U_i.cm* + rho·(u_i.cmW | cmT)
//                   with the curve precompiles, the two KZG openings with the pairing precompile, and this proof with the pairing precompile
//   1. u_i.x0 = H(H(dg, i, z_0, z_i), U_i)  and  u_i.x1 = H(dg, cfU_i)         the hashes the last instance of F' carries (aug/cyclefold.hpp)
//   2. rho < 2^128;  u' = U.u + rho,  x' = U.x + rho·u_i.x                      NIFS.V on the scalars (the challenge itself is derived by the
//                                                                              verifier from the transcript of the records, outside)
//   3. (A·Z)∘(B·Z) = u'·(C·Z) + E   for Z = (u', W', x0', x1')                  the folded main instance satisfies F' ∪ step circuit, row by row
//   4. e_W = Σ_j W'_j c_W^j,  e_E = Σ_k E_k c_E^k                               the evaluations the two KZG openings are about
// The CycleFold instance cfU_i is bound through its hash (1) and checked outside the circuit (vimz_cf_verify: its relaxed relation over Fq);
// Sonobe's circuit checks it in non-native arithmetic.
#pragma once
#include "cyclefold.hpp"

namespace vz {
namespace aug {

struct DeciderIn {
  CfFr digest; uint64_t i = 0; std::vector<CfFr> z0, zi;
  CfMainRelaxed U; CfMainFresh u; CfRelaxed cfU;
  uint32_t r_low[4] = {0, 0, 0, 0};            // rho: the 128-bit challenge of the final fold (cyclefold_merge.hip: chal 'b')
  NnPoint cmT, Wn, En;                         // cross-term commitment, folded commitments
  CfFr cW, cE, eW, eE;                         // KZG challenges and evaluations
  const CfFr* Wf = nullptr; const CfFr* Ef = nullptr;      // folded witness (wires 1 .. n_wires-3 of Z) and error vector; nullptr in shape mode
};

inline CfFr decider_rho(const uint32_t low[4]) { CfFr c = CfFr::zero(); for (int k = 0; k < 4; k++) c.v[k] = low[k]; return CfFr::to_mont(c); }      // the final fold's 128-bit challenge
inline CfFr decider_instance_hash(const DeciderIn& in) {
  std::vector<CfFr> h = {in.digest, decider_rho(in.r_low)};
  for (const NnPoint* p : {&in.U.W, &in.U.E, &in.u.W, &in.cmT, &in.Wn, &in.En}) { cf_push_limbs(p->x, h); cf_push_limbs(p->y, h); }
  h.push_back(in.cW); h.push_back(in.cE); h.push_back(in.eW); h.push_back(in.eE);
  return hash_native<BnFr>(h);
}

// shape mode (cs.b set): appends the circuit to cs.b, public inputs first;  witness mode: cs.w = the assignment after wire 0
inline void synthesize_decider(CS<BnFr>& cs, const cb::BuilderT<CfFr>& main, uint32_t len_z, const DeciderIn& in) {
  typedef CfFr F;
  typedef Num<F> N;
  typedef cb::LCT<F> LC;
  const bool shape = cs.shape();
  const uint32_t nw = main.n_wires, nc = main.n_constraints();
  // ---- public inputs: i, z_0, z_i, h_inst -------------------------------------------------------------------------------------------
  N iN = cs.alloc(cb::f_from_u64<F>(in.i));
  std::vector<N> z0(len_z), zi(len_z);
  for (uint32_t k = 0; k < len_z; k++) z0[k] = cs.alloc(k < in.z0.size() ? in.z0[k] : F::zero());
  for (uint32_t k = 0; k < len_z; k++) zi[k] = cs.alloc(k < in.zi.size() ? in.zi[k] : F::zero());
  N h_pub = cs.alloc(shape ? F::zero() : decider_instance_hash(in));
  // ---- private inputs ----------------------------------------------------------------------------------------------------------------
  struct NnVar { N x[4], y[4]; };
  auto nn_alloc = [&](const NnPoint& p) { NnVar r; for (int j = 0; j < 4; j++) r.x[j] = cs.alloc(cb::f_from_u64<F>(p.x.w[j])); for (int j = 0; j < 4; j++) r.y[j] = cs.alloc(cb::f_from_u64<F>(p.y.w[j])); return r; };
  auto push_nn = [](std::vector<N>& h, const NnVar& p) { for (int j = 0; j < 4; j++) h.push_back(p.x[j]); for (int j = 0; j < 4; j++) h.push_back(p.y[j]); };
  N dg = cs.alloc(in.digest);
  NnVar UW = nn_alloc(in.U.W), UE = nn_alloc(in.U.E), uW = nn_alloc(in.u.W), T = nn_alloc(in.cmT), Wn = nn_alloc(in.Wn), En = nn_alloc(in.En);
  N Uu = cs.alloc(in.U.u), Ux0 = cs.alloc(in.U.x0), Ux1 = cs.alloc(in.U.x1), ux0 = cs.alloc(in.u.x0), ux1 = cs.alloc(in.u.x1);
  N cu = cs.alloc(in.cfU.u);
  N cx[CF_IO][4];
  for (int k = 0; k < CF_IO; k++) for (int j = 0; j < 4; j++) cx[k][j] = cs.alloc(cb::f_from_u64<F>(in.cfU.x[k].w[j]));
  N cWx = cs.alloc(in.cfU.W.x), cWy = cs.alloc(in.cfU.W.y), cEx = cs.alloc(in.cfU.E.x), cEy = cs.alloc(in.cfU.E.y);
  N cW = cs.alloc(in.cW), cE = cs.alloc(in.cE), eW = cs.alloc(in.eW), eE = cs.alloc(in.eE);
  F rl = F::zero(); for (int k = 0; k < 4; k++) rl.v[k] = in.r_low[k];
  N r_lo = cs.alloc(F::to_mont(rl));
  // ---- 1. the hashes the last instance of F' carries ----------------------------------------------------------------------------------
  {
    std::vector<N> hst = {dg, iN};
    hst.insert(hst.end(), z0.begin(), z0.end());
    hst.insert(hst.end(), zi.begin(), zi.end());
    std::vector<N> hin = {cs.hash(hst), Uu, Ux0, Ux1};
    push_nn(hin, UW); push_nn(hin, UE);
    N hU = cs.hash(hin);
    cs.enforce_equal(hU, ux0);
    if (!shape && !hU.v.eq(ux0.v)) cs.bad = true;
    std::vector<N> hc = {dg, cu};
    for (int k = 0; k < CF_IO; k++) for (int j = 0; j < 4; j++) hc.push_back(cx[k][j]);
    hc.push_back(cWx); hc.push_back(cWy); hc.push_back(cEx); hc.push_back(cEy);
    N hcf = cs.hash(hc);
    cs.enforce_equal(hcf, ux1);
    if (!shape && !hcf.v.eq(ux1.v)) cs.bad = true;
  }
  // ---- 2. NIFS.V on the scalars ---------------------------------------------------------------------------------------------------------
  cs.bits(r_lo, 128);
  N rho = r_lo;
  N un = cs.add(Uu, rho);                                   // (u_i.u = 1)
  N x0n = cs.add(Ux0, cs.mul(rho, ux0)), x1n = cs.add(Ux1, cs.mul(rho, ux1));
  // ---- the instance hash the public input carries --------------------------------------------------------------------------------------
  {
    std::vector<N> h = {dg, rho};
    for (const NnVar* p : {&UW, &UE, &uW, &T, &Wn, &En}) push_nn(h, *p);
    h.push_back(cW); h.push_back(cE); h.push_back(eW); h.push_back(eE);
    N hi = cs.hash(h);
    cs.enforce_equal(hi, h_pub);
    if (!shape && !hi.v.eq(h_pub.v)) cs.bad = true;
  }
  // ---- 3. the folded main instance satisfies its relaxed R1CS ---------------------------------------------------------------------------
  // Z = (u', W'_1 .. W'_{nw-3}, x0', x1'): the witness entries are this circuit's variables, the three instance scalars the values above
  std::vector<N> Wv(nw >= 3 ? nw - 3 : 0), Ev(nc);
  for (uint32_t j = 0; j < Wv.size(); j++) Wv[j] = cs.alloc(in.Wf ? in.Wf[j] : F::zero());
  for (uint32_t k = 0; k < nc; k++) Ev[k] = cs.alloc(in.Ef ? in.Ef[k] : F::zero());
  auto zvar = [&](uint32_t w) -> const N& { return w == 0 ? un : w == nw - 2 ? x0n : w == nw - 1 ? x1n : Wv[w - 1]; };
  auto row_lc = [&](const cb::Csr& M, uint32_t r) {
    N acc; acc.v = F::zero(); acc.konst = false;
    if (shape) {
      // (terms arrive sorted by the main shape's wire; this circuit's wire numbers are monotone in it except for the three scalars: merge generally)
      LC lc;
      for (uint32_t k = M.row_ptr[r]; k < M.row_ptr[r + 1]; k++) lc = LC::axpy(lc, main.dict[M.coef[k]], zvar(M.col[k]).lc);
      acc.lc = lc;
    } else {
      F s = F::zero();
      for (uint32_t k = M.row_ptr[r]; k < M.row_ptr[r + 1]; k++) s = F::add(s, F::mul(main.dict[M.coef[k]], zvar(M.col[k]).v));
      acc.v = s;
    }
    return acc;
  };
  for (uint32_t r = 0; r < nc; r++) {
    N az = row_lc(main.A, r), bz = row_lc(main.B, r);
    N rhs = Ev[r];
    if (main.C.row_ptr[r + 1] > main.C.row_ptr[r]) {
      N cz = row_lc(main.C, r);
      N t = cs.alloc(F::mul(un.v, cz.v));
      cs.enforce(un, cz, t);
      rhs = cs.add(t, Ev[r]);
    }
    cs.enforce(az, bz, rhs);
    if (!shape && !F::mul(az.v, bz.v).eq(rhs.v)) cs.bad = true;
  }
  // ---- 4. the evaluations the KZG openings are about (Horner from the top coefficient) ---------------------------------------------------
  auto horner = [&](const std::vector<N>& v, const N& c, const N& e) {
    if (v.empty()) { cs.enforce_equal(e, cs.zero()); return; }
    N acc = v.back();
    for (size_t j = v.size() - 1; j-- > 0;) {
      N nx = cs.alloc(F::add(F::mul(acc.v, c.v), v[j].v));
      cs.enforce(acc, c, cs.sub(nx, v[j]));
      acc = nx;
    }
    cs.enforce_equal(acc, e);
    if (!shape && !acc.v.eq(e.v)) cs.bad = true;
  };
  horner(Wv, cW, eW);
  horner(Ev, cE, eE);
}

struct DeciderCircuit {
  cb::BuilderT<CfFr> b;
  uint32_t n_public = 0, len_z = 0;
  void finish(const cb::BuilderT<CfFr>& main, uint32_t lz) {
    b = cb::BuilderT<CfFr>(); len_z = lz; n_public = 2 * lz + 2;
    CS<BnFr> cs; cs.b = &b; cs.base = b.n_wires;
    DeciderIn in; in.digest = CfFr::zero(); in.U = CfMainRelaxed::zero(); in.u = CfMainFresh::zero(); in.cfU = CfRelaxed::zero();
    in.cmT = in.Wn = in.En = NnPoint::zero(); in.cW = in.cE = in.eW = in.eE = CfFr::zero();
    synthesize_decider(cs, main, lz, in);
  }
  // the full assignment (wire 0 = 1, then the public inputs); *bad: some check of the statement fails on these inputs
  std::vector<CfFr> witness(const cb::BuilderT<CfFr>& main, const DeciderIn& in, bool* bad) const {
    CS<BnFr> cs; cs.base = 1;
    cs.w.reserve(b.n_wires);
    synthesize_decider(cs, main, len_z, in);
    if (cs.w.size() + 1 != b.n_wires) throw std::runtime_error("decider circuit: witness length differs from the shape");
    if (bad) *bad = cs.bad;
    std::vector<CfFr> z; z.reserve(b.n_wires);
    z.push_back(CfFr::one());
    z.insert(z.end(), cs.w.begin(), cs.w.end());
    return z;
  }
};

}  // namespace aug
}  // namespace vz
