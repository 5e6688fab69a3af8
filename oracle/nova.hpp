// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).
//
// Nova's non-interactive folding verifier and the IVC hash chain, stated natively (no circuit) for the
// BN254 / Grumpkin cycle the reference instantiates (vimz/src/nova_snark_backend/mod.rs:19-20).  This is the relation
// the augmented circuits of the product (vimz_amd/csrc/aug/) must enforce; the tests compare the circuits' outputs with
// it and use it as the independent acceptance check of an IVC proof (RecursiveSNARK::verify, reached from
// vimz/src/nova_snark_backend/folding.rs:45-56).  Source of the algorithm: Kothapalli–Setty–Tzialla, "Nova" (CRYPTO
// 2022), Fig. 4 and §4 (NIFS); nova-snark 0.23.0 is not vendored (vimz/Cargo.lock:3577-3606), so parameters that the
// paper leaves open are the product's documented choices restated here from that description:
//   * hash: circomlib-construction Poseidon over the circuit's own field, parameters from the Poseidon reference
//     Grain LFSR for that prime; arbitrary-length input absorbed 8 elements first, then (running hash + 7) per call;
//   * instance hash trunc250(H(digest, i, z_0, z_i, U)) with U = (W.x, W.y, E.x, E.y, u, X0 limbs[4], X1 limbs[4]), 64-bit limbs
//     (the shape digest and the initial state are absorbed by every hash, as in Nova Fig. 4: "hash(vk, i, z_0, z_i, U_i)");
//   * base case: z_i must equal z_0;
//   * challenge rho = 2^128 + low128(H(H_full(digest,i,z_0,z_i,U), u.W.x, u.W.y, u.x0, u.x1, T.x, T.y));
//   * NIFS.V: W' = W + rho·w, E' = E + rho·T, u' = u + rho, X' = X + rho·x mod (other field's prime).
#pragma once
#include <vector>
#include <mutex>
#include "field.hpp"
#include "curve.hpp"
#include "poseidon.hpp"

namespace orc {

template <class F>
struct PoseidonParamsG { int t = 0, rf = 8, rp = 0; std::vector<F> C, M; };

template <class F>
static const PoseidonParamsG<F>& poseidon_params_g(int t) {
  static PoseidonParamsG<F> cache[18];
  static std::mutex mu;
  std::lock_guard<std::mutex> g(mu);
  PoseidonParamsG<F>& P = cache[t];
  if (P.t == t) return P;
  P.rf = 8; P.rp = POSEIDON_RP[t - 2];
  const u64* mod = F::P().p;
  int bits = 256; while (bits > 0 && !((mod[(bits - 1) / 64] >> ((bits - 1) % 64)) & 1)) bits--;
  Grain gr(bits, t, P.rf, P.rp);
  while ((int)P.C.size() < (P.rf + P.rp) * t) { u64 v[4]; gr.sample(v); if (cmp4(v, mod) < 0) P.C.push_back(F::from_canonical(v)); }
  std::vector<F> xs(t), ys(t);
  for (int i = 0; i < t; i++) { u64 v[4]; gr.sample(v); xs[i] = F::from_canonical(v); }
  for (int i = 0; i < t; i++) { u64 v[4]; gr.sample(v); ys[i] = F::from_canonical(v); }
  P.M.resize(t * t);
  for (int i = 0; i < t; i++) for (int j = 0; j < t; j++) P.M[i * t + j] = (xs[i] + ys[j]).inv();
  P.t = t;
  return P;
}

template <class F>
static F poseidon_g(const F* in, int n) {
  const int t = n + 1;
  const PoseidonParamsG<F>& P = poseidon_params_g<F>(t);
  std::vector<F> s(t), u(t);
  s[0] = F::zero();
  for (int i = 0; i < n; i++) s[i + 1] = in[i];
  for (int r = 0; r < P.rf + P.rp; r++) {
    for (int i = 0; i < t; i++) s[i] = s[i] + P.C[r * t + i];
    const bool full = r < P.rf / 2 || r >= P.rf / 2 + P.rp;
    if (full) for (int i = 0; i < t; i++) s[i] = s[i].pow5(); else s[0] = s[0].pow5();
    for (int i = 0; i < t; i++) { F acc = F::zero(); for (int j = 0; j < t; j++) acc = acc + P.M[i * t + j] * s[j]; u[i] = acc; }
    s.swap(u);
  }
  return s[0];
}

template <class F>
static F nova_hash(const std::vector<F>& in) {
  size_t n = in.size(), take = n < 8 ? n : 8;
  F h = poseidon_g<F>(in.data(), (int)take);
  for (size_t pos = take; pos < n;) {
    size_t cur = n - pos < 7 ? n - pos : 7;
    std::vector<F> blk; blk.push_back(h);
    for (size_t k = 0; k < cur; k++) blk.push_back(in[pos + k]);
    h = poseidon_g<F>(blk.data(), (int)blk.size());
    pos += cur;
  }
  return h;
}

template <class F> static F low_bits(const F& x, int nbits) {   // the integer x mod 2^nbits, as a field element
  u64 c[4]; x.to_canonical(c);
  for (int k = nbits; k < 256; k++) c[k / 64] &= ~(1ull << (k % 64));
  return F::from_canonical(c);
}

// A relaxed instance as seen from the field F that its commitments' coordinates live in.
template <class F> struct NovaRelaxed { Affine<F> W, E; F u; u64 X0[4], X1[4]; };
template <class F> struct NovaFresh { Affine<F> W; F x0, x1; };

template <class F>
static F nova_instance_hash_full(const F& digest, u64 i, const std::vector<F>& z0, const std::vector<F>& z, const NovaRelaxed<F>& U) {
  // two levels: the statement part (digest, i, z_0, z_i) is hashed on its own, then absorbed with the instance
  std::vector<F> st = {digest, F::from_u64(i)};
  st.insert(st.end(), z0.begin(), z0.end());
  st.insert(st.end(), z.begin(), z.end());
  std::vector<F> in = {nova_hash<F>(st)};
  in.push_back(U.u);
  for (int k = 0; k < 4; k++) in.push_back(F::from_u64(U.X0[k]));
  for (int k = 0; k < 4; k++) in.push_back(F::from_u64(U.X1[k]));
  in.push_back(U.W.x); in.push_back(U.W.y); in.push_back(U.E.x); in.push_back(U.E.y);
  return nova_hash<F>(in);
}

// One run of the augmented circuit's relation.  C: the curve whose points are being folded (coordinates in F = C::Base);
// G: the other field (the folded instances' scalar field).  Returns false when the incoming hash does not match, or when the base
// case does not start from z_0.
template <class C, class G>
static bool nova_step(bool is_primary, const typename C::Base& digest, u64 i, const std::vector<typename C::Base>& z_0,
                      const std::vector<typename C::Base>& z_i, const std::vector<typename C::Base>& z_next,
                      const NovaRelaxed<typename C::Base>& U_in, const NovaFresh<typename C::Base>& u, const Affine<typename C::Base>& T,
                      NovaRelaxed<typename C::Base>& U_new, u64 rho_out[4], typename C::Base& x1_out) {
  typedef typename C::Base F;
  const bool base = i == 0;
  if (base) for (size_t k = 0; k < z_i.size(); k++) if (z_i[k] != z_0[k]) return false;      // the chain starts from the claimed state
  const F h_full = nova_instance_hash_full<F>(digest, i, z_0, z_i, U_in);
  if (!base && low_bits(h_full, 250) != u.x0) return false;
  std::vector<F> rin = {h_full, u.W.x, u.W.y, u.x0, u.x1, T.x, T.y};
  const F hr = nova_hash<F>(rin);
  u64 rho[4]; low_bits(hr, 128).to_canonical(rho); rho[2] = 1; rho[3] = 0;      // rho = 2^128 + low 128 bits
  memcpy(rho_out, rho, 32);
  NovaRelaxed<F> U = U_in;
  if (base) { U.W.x = U.W.y = U.E.x = U.E.y = U.u = F::zero(); memset(U.X0, 0, 32); memset(U.X1, 0, 32); }
  NovaRelaxed<F> R;
  R.W = C::to_affine(C::add(C::from_affine(U.W), C::mul(u.W, rho)));
  R.E = C::to_affine(C::add(C::from_affine(U.E), C::mul(T, rho)));
  R.u = U.u + F::from_canonical(rho);
  const G rg = G::from_canonical(rho);
  u64 x0c[4], x1c[4]; u.x0.to_canonical(x0c); u.x1.to_canonical(x1c);
  (G::from_canonical(U.X0) + rg * G::from_canonical(x0c)).to_canonical(R.X0);
  (G::from_canonical(U.X1) + rg * G::from_canonical(x1c)).to_canonical(R.X1);
  if (is_primary && base) { R.W.x = R.W.y = R.E.x = R.E.y = R.u = F::zero(); memset(R.X0, 0, 32); memset(R.X1, 0, 32); }
  U_new = R;
  x1_out = low_bits(nova_instance_hash_full<F>(digest, i + 1, z_0, z_next, R), 250);
  return true;
}

}  // namespace orc
