// Export of an augmented circuit's R1CS through the C ABI (same table codes as vimz_circuit_export).
#pragma once
#include "../../../include/vimz_hip.h"
#include "augmented.hpp"

namespace vz {
namespace aug {

template <class FP>
inline int64_t export_r1cs(const AugCircuit<FP>& c, int what, void* buf, size_t cap) {
  typedef Fp<FP> F;
  const cb::BuilderT<F>& b = c.b;
  const void* src = nullptr; size_t bytes = 0;
  auto vec = [&](const auto& v) { src = v.data(); bytes = v.size() * sizeof(v[0]); };
  switch (what) {
    case VIMZ_CX_A_ROWPTR: vec(b.A.row_ptr); break;
    case VIMZ_CX_A_COL: vec(b.A.col); break;
    case VIMZ_CX_A_COEF: vec(b.A.coef); break;
    case VIMZ_CX_B_ROWPTR: vec(b.B.row_ptr); break;
    case VIMZ_CX_B_COL: vec(b.B.col); break;
    case VIMZ_CX_B_COEF: vec(b.B.coef); break;
    case VIMZ_CX_C_ROWPTR: vec(b.C.row_ptr); break;
    case VIMZ_CX_C_COL: vec(b.C.col); break;
    case VIMZ_CX_C_COEF: vec(b.C.coef); break;
    case VIMZ_CX_DICT_MONT: vec(b.dict); break;
    case VIMZ_CX_DICT_CANON: {
      bytes = b.dict.size() * 32;
      if (buf && cap >= bytes) { F* o = (F*)buf; for (size_t i = 0; i < b.dict.size(); i++) o[i] = F::from_mont(b.dict[i]); }
      return (int64_t)bytes;
    }
    case VIMZ_IX_INFO: {
      bytes = 32;
      if (buf && cap >= bytes) { uint64_t o[4] = {c.n_wires(), c.n_constraints(), c.step_wires, c.step_constraints}; memcpy(buf, o, 32); }
      return 32;
    }
    default: return VIMZ_ERR_INVALID;
  }
  if (buf && cap >= bytes && bytes) memcpy(buf, src, bytes);
  return (int64_t)bytes;
}

// flat canonical I/O of one run of the verifier circuit (parity hook): see vimz_augcircuit_witness
template <class FP>
inline int witness_flat(const AugCircuit<FP>& c, const uint64_t* in, uint64_t* wires_out, uint64_t* outputs) {
  typedef Fp<FP> F;
  auto fe = [&](int k) { F x; memcpy(x.v, in + 4 * k, 32); return F::to_mont(x); };
  auto u256 = [&](int k) { U256w x; memcpy(x.w, in + 4 * k, 32); return x; };
  AugIn<FP> a;
  a.digest = fe(0); a.i = in[4];
  a.z0.push_back(fe(2));
  F zi = fe(3);
  a.U.W.x = fe(4); a.U.W.y = fe(5); a.U.E.x = fe(6); a.U.E.y = fe(7); a.U.u = fe(8); a.U.X0 = u256(9); a.U.X1 = u256(10);
  a.u.W.x = fe(11); a.u.W.y = fe(12); a.u.x0 = fe(13); a.u.x1 = fe(14);
  a.T.x = fe(15); a.T.y = fe(16);
  std::vector<F> aug; bool bad = false;
  AugOut<FP> o = c.witness(a, &zi, &zi, aug, &bad);
  auto put = [&](uint64_t* dst, const F& m) { F x = F::from_mont(m); memcpy(dst, x.v, 32); };
  if (wires_out) {
    put(wires_out, F::one()); put(wires_out + 4, zi); put(wires_out + 8, zi);
    for (size_t k = 0; k < aug.size(); k++) put(wires_out + 4 * (3 + k), aug[k]);
  }
  if (outputs) {
    put(outputs, o.U_new.W.x); put(outputs + 4, o.U_new.W.y); put(outputs + 8, o.U_new.E.x); put(outputs + 12, o.U_new.E.y); put(outputs + 16, o.U_new.u);
    memcpy(outputs + 20, o.U_new.X0.w, 32); memcpy(outputs + 24, o.U_new.X1.w, 32);
    uint64_t rho[4] = {(uint64_t)o.rho_low[0] | ((uint64_t)o.rho_low[1] << 32), (uint64_t)o.rho_low[2] | ((uint64_t)o.rho_low[3] << 32), 1, 0};
    memcpy(outputs + 28, rho, 32);
    put(outputs + 32, o.x0); put(outputs + 36, o.x1);
    uint64_t flag[4] = {bad ? 1u : 0u, 0, 0, 0}; memcpy(outputs + 40, flag, 32);
  }
  return VIMZ_OK;
}

}  // namespace aug
}  // namespace vz
