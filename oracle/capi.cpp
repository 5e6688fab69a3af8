// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).
// C API over the oracle so tests can drive it with ctypes.  All field elements cross this
// API as canonical (non-Montgomery) little-endian 4 x u64 unless the name says `mont`.
#include <cstdio>
#include <chrono>
#include "field.hpp"
#include "curve.hpp"
#include "poseidon.hpp"
#include "steps.hpp"
#include "r1cs.hpp"
#include "witness.hpp"
#include "nova.hpp"

using namespace orc;

#define FIELD_SWITCH(fid, ...)                          \
  switch (fid) {                                         \
    case 0: { typedef BnFr F; __VA_ARGS__; } break;             \
    case 1: { typedef BnFq F; __VA_ARGS__; } break;             \
    case 2: { typedef PallasFp F; __VA_ARGS__; } break;         \
    default: { typedef VestaFq F; __VA_ARGS__; } break;         \
  }
#define CURVE_SWITCH(cid, BODY)                          \
  switch (cid) {                                         \
    case 0: { typedef BnG1 C; BODY; } break;             \
    case 1: { typedef Grumpkin C; BODY; } break;         \
    case 2: { typedef Pallas C; BODY; } break;           \
    default: { typedef Vesta C; BODY; } break;           \
  }

template <class C>
static typename C::Aff load_aff(const u64* xy) {
  typename C::Aff a;
  a.x = C::Base::from_canonical(xy); a.y = C::Base::from_canonical(xy + 4);
  return a;
}
template <class C>
static void store_aff(const typename C::Aff& a, u64* xy) { a.x.to_canonical(xy); a.y.to_canonical(xy + 4); }

template <class F> static NovaRelaxed<F> load_relaxed(const u64* U) {
  NovaRelaxed<F> r;
  r.W.x = F::from_canonical(U); r.W.y = F::from_canonical(U + 4); r.E.x = F::from_canonical(U + 8); r.E.y = F::from_canonical(U + 12);
  r.u = F::from_canonical(U + 16); memcpy(r.X0, U + 20, 32); memcpy(r.X1, U + 24, 32);
  return r;
}
template <class F> static void store_relaxed(const NovaRelaxed<F>& r, u64* U) {
  r.W.x.to_canonical(U); r.W.y.to_canonical(U + 4); r.E.x.to_canonical(U + 8); r.E.y.to_canonical(U + 12); r.u.to_canonical(U + 16);
  memcpy(U + 20, r.X0, 32); memcpy(U + 24, r.X1, 32);
}
template <class Cv, class G>
static int nova_step_c(int is_primary, const u64* pz, u64 i, const u64* z_0, const u64* z_i, const u64* z_next, int len_z, const u64* U, const u64* u, const u64* T,
                       u64* U_new, u64* rho, u64* x1) {
  typedef typename Cv::Base F;
  std::vector<F> z0(len_z), zi(len_z), zn(len_z);
  for (int k = 0; k < len_z; k++) { z0[k] = F::from_canonical(z_0 + 4 * k); zi[k] = F::from_canonical(z_i + 4 * k); zn[k] = F::from_canonical(z_next + 4 * k); }
  NovaFresh<F> uf; uf.W.x = F::from_canonical(u); uf.W.y = F::from_canonical(u + 4); uf.x0 = F::from_canonical(u + 8); uf.x1 = F::from_canonical(u + 12);
  Affine<F> Tp; Tp.x = F::from_canonical(T); Tp.y = F::from_canonical(T + 4);
  NovaRelaxed<F> R; F x1f;
  if (!nova_step<Cv, G>(is_primary != 0, F::from_canonical(pz), i, z0, zi, zn, load_relaxed<F>(U), uf, Tp, R, rho, x1f)) return 0;
  store_relaxed(R, U_new); x1f.to_canonical(x1);
  return 1;
}
template <class F>
static long r1cs_check_relaxed_t(size_t nrows, size_t ncols, const uint32_t* const* row_ptr, const uint32_t* const* col, const uint32_t* const* coef,
                                 const u64* dict, size_t ndict, const u64* z, const u64* u, const u64* E, int threads) {
  std::vector<F> D(ndict), Z(ncols);
  for (size_t k = 0; k < ndict; k++) D[k] = F::from_canonical(dict + 4 * k);
  for (size_t k = 0; k < ncols; k++) Z[k] = F::from_canonical(z + 4 * k);
  std::vector<F> out[3];
  for (int m = 0; m < 3; m++) {
    out[m].resize(nrows);
    auto work = [&, m](size_t lo, size_t hi) {
      for (size_t r = lo; r < hi; r++) { F acc = F::zero(); for (uint32_t k = row_ptr[m][r]; k < row_ptr[m][r + 1]; k++) acc = acc + D[coef[m][k]] * Z[col[m][k]]; out[m][r] = acc; }
    };
    std::vector<std::thread> th; const int nt = threads > 0 ? threads : 1; size_t chunk = (nrows + nt - 1) / nt;
    for (int t = 0; t < nt; t++) th.emplace_back(work, std::min(nrows, t * chunk), std::min(nrows, (t + 1) * chunk));
    for (auto& x : th) x.join();
  }
  const F uu = F::from_canonical(u);
  for (size_t r = 0; r < nrows; r++) {
    F rhs = uu * out[2][r];
    if (E) rhs = rhs + F::from_canonical(E + 4 * r);
    if (out[0][r] * out[1][r] != rhs) return (long)r;
  }
  return -1;
}

template <class Fn>
static void par_chunks(size_t n, int threads, Fn fn) {
  const size_t T = threads > 1 ? (size_t)threads : 1, chunk = (n + T - 1) / T;
  if (T == 1) { fn((size_t)0, n); return; }
  std::vector<std::thread> th;
  for (size_t t = 0; t < T; t++) { const size_t lo = std::min(n, t * chunk), hi = std::min(n, lo + chunk); if (lo < hi) th.emplace_back(fn, lo, hi); }
  for (auto& x : th) x.join();
}

extern "C" {

// ---------------- fields ----------------
void orc_field_modulus(int fid, u64* out) { FIELD_SWITCH(fid, memcpy(out, F::P().p, 32)); }
void orc_field_consts(int fid, u64* r1, u64* r2, u64* n0) {
  FIELD_SWITCH(fid, { memcpy(r1, F::P().r1, 32); memcpy(r2, F::P().r2, 32); *n0 = F::P().n0; });
}
void orc_f_add(int fid, const u64* a, const u64* b, u64* o) { FIELD_SWITCH(fid, (F::from_canonical(a) + F::from_canonical(b)).to_canonical(o)); }
void orc_f_sub(int fid, const u64* a, const u64* b, u64* o) { FIELD_SWITCH(fid, (F::from_canonical(a) - F::from_canonical(b)).to_canonical(o)); }
void orc_f_mul(int fid, const u64* a, const u64* b, u64* o) { FIELD_SWITCH(fid, (F::from_canonical(a) * F::from_canonical(b)).to_canonical(o)); }
void orc_f_inv(int fid, const u64* a, u64* o) { FIELD_SWITCH(fid, F::from_canonical(a).inv().to_canonical(o)); }
void orc_f_to_mont(int fid, const u64* a, u64* o, size_t n) { FIELD_SWITCH(fid, for (size_t i = 0; i < n; i++) { F x = F::from_canonical(a + 4 * i); memcpy(o + 4 * i, x.l, 32); }); }
void orc_f_from_mont(int fid, const u64* a, u64* o, size_t n) { FIELD_SWITCH(fid, for (size_t i = 0; i < n; i++) F::from_mont_limbs(a + 4 * i).to_canonical(o + 4 * i)); }

// ---------------- curves ----------------
int orc_curve_on(int cid, const u64* xy) { CURVE_SWITCH(cid, return C::on_curve(load_aff<C>(xy)) ? 1 : 0); return 0; }
void orc_curve_add(int cid, const u64* p, const u64* q, u64* o) {
  CURVE_SWITCH(cid, store_aff<C>(C::to_affine(C::add(C::from_affine(load_aff<C>(p)), C::from_affine(load_aff<C>(q)))), o));
}
void orc_curve_mul(int cid, const u64* p, const u64* k, u64* o) {
  CURVE_SWITCH(cid, store_aff<C>(C::to_affine(C::mul(load_aff<C>(p), k)), o));
}
// bases: n x 8 limbs canonical affine; scalars: n x 4 limbs canonical; out: affine canonical
void orc_msm(int cid, const u64* bases, const u64* scalars, size_t n, int threads, u64* out) {
  CURVE_SWITCH(cid, {
    std::vector<C::Aff> b(n);
    for (size_t i = 0; i < n; i++) b[i] = load_aff<C>(bases + 8 * i);
    store_aff<C>(C::to_affine(C::msm(b.data(), scalars, n, threads)), out);
  });
}
// Same but bases already in Montgomery form (what the product keeps resident); returns seconds spent in the
// MSM proper (conversion excluded) for the cpu_baseline leg.
double orc_msm_mont_timed(int cid, const u64* bases_mont, const u64* scalars, size_t n, int threads, u64* out) {
  double secs = 0;
  CURVE_SWITCH(cid, {
    auto t0 = std::chrono::steady_clock::now();
    auto r = C::msm(reinterpret_cast<const C::Aff*>(bases_mont), scalars, n, threads);
    secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    store_aff<C>(C::to_affine(r), out);
  });
  return secs;
}
// Deterministic test bases: P_i = (seed + i + 1) * G computed incrementally (G canonical affine in).
void orc_curve_seq_bases(int cid, const u64* g, const u64* start, size_t n, u64* out) {
  CURVE_SWITCH(cid, {
    C::Aff G = load_aff<C>(g);
    C::Jac acc = C::mul(G, start);
    for (size_t i = 0; i < n; i++) {
      acc = C::add_mixed(acc, G);
      store_aff<C>(C::to_affine(acc), out + 8 * i);
    }
  });
}

// ---------------- Poseidon / hashers (BN254 Fr) ----------------
void orc_poseidon(const u64* in, int n, u64* out) {
  std::vector<BnFr> v(n);
  for (int i = 0; i < n; i++) v[i] = BnFr::from_canonical(in + 4 * i);
  poseidon(v.data(), n).to_canonical(out);
}
void orc_poseidon_constants(int t, u64* C_out, u64* M_out, int* rf, int* rp) {
  const PoseidonParams& P = poseidon_params(t);
  *rf = P.rf; *rp = P.rp;
  if (C_out) for (size_t i = 0; i < P.C.size(); i++) P.C[i].to_canonical(C_out + 4 * i);
  if (M_out) for (size_t i = 0; i < P.M.size(); i++) P.M[i].to_canonical(M_out + 4 * i);
}
void orc_array_hash(const u64* in, int L, u64* out) {
  std::vector<BnFr> v(L);
  for (int i = 0; i < L; i++) v[i] = BnFr::from_canonical(in + 4 * i);
  array_hash(v.data(), L).to_canonical(out);
}
void orc_head_tail_hash(const u64* head, const u64* tail, int L, u64* out) {
  std::vector<BnFr> v(L);
  for (int i = 0; i < L; i++) v[i] = BnFr::from_canonical(tail + 4 * i);
  head_tail_hash(BnFr::from_canonical(head), v.data(), L).to_canonical(out);
}
// running image hash: acc_{k+1} = HeadTailHasher(width)(acc_k, row_k)  (circuits/image_running_hash.circom:8-19)
void orc_image_hash(const u64* rows, int nrows, int width, u64* out) {
  BnFr acc = BnFr::zero();
  std::vector<BnFr> v(width);
  for (int r = 0; r < nrows; r++) {
    for (int i = 0; i < width; i++) v[i] = BnFr::from_canonical(rows + 4 * ((size_t)r * width + i));
    acc = head_tail_hash(acc, v.data(), width);
  }
  acc.to_canonical(out);
}

// ---------------- step semantics ----------------
int orc_ivc_state_len(int t) { return ivc_state_len(t); }
int orc_step_input_width(int t, int width, int width2, int rows_in, int rows_out, int crop_h) {
  StepShape s = {width, width2, rows_in, rows_out, crop_h};
  return step_input_width(t, s);
}
// returns 1 if the step relation holds, 0 otherwise; z_out always written.
int orc_step_eval(int t, int width, int width2, int rows_in, int rows_out, int crop_h,
                  const u64* z_in, const u64* inputs, u64* z_out) {
  StepShape s = {width, width2, rows_in, rows_out, crop_h};
  int L = ivc_state_len(t);
  std::vector<BnFr> z(L);
  for (int i = 0; i < L; i++) z[i] = BnFr::from_canonical(z_in + 4 * i);
  StepOut r = step_eval(t, s, z, inputs);
  for (int i = 0; i < L; i++) r.z[i].to_canonical(z_out + 4 * i);
  return r.ok ? 1 : 0;
}

// ---------------- relaxed R1CS algebra ----------------
#define VEC(F, name, ptr, n) std::vector<F> name(n); for (size_t _i = 0; _i < (size_t)(n); _i++) name[_i] = F::from_canonical((ptr) + 4 * _i)
#define OUT(vec, ptr) for (size_t _i = 0; _i < vec.size(); _i++) vec[_i].to_canonical((ptr) + 4 * _i)

void orc_spmv(int fid, size_t nrows, size_t ncols, const uint32_t* row_ptr, const uint32_t* col, const u64* val,
              const u64* z, u64* out, int threads) {
  FIELD_SWITCH(fid, { VEC(F, zz, z, ncols); std::vector<F> o(nrows); spmv<F>(nrows, row_ptr, col, val, zz.data(), o.data(), threads); OUT(o, out); });
}
void orc_cross_term(int fid, size_t n, const u64* az1, const u64* bz1, const u64* cz1, const u64* u1,
                    const u64* az2, const u64* bz2, const u64* cz2, const u64* u2, u64* T) {
  FIELD_SWITCH(fid, {
    VEC(F, a1, az1, n); VEC(F, b1, bz1, n); VEC(F, c1, cz1, n); VEC(F, a2, az2, n); VEC(F, b2, bz2, n); VEC(F, c2, cz2, n);
    std::vector<F> t(n);
    cross_term<F>(n, a1.data(), b1.data(), c1.data(), F::from_canonical(u1), a2.data(), b2.data(), c2.data(), F::from_canonical(u2), t.data());
    OUT(t, T);
  });
}
void orc_axpy(int fid, size_t n, const u64* a, const u64* r, const u64* b, u64* out) {
  FIELD_SWITCH(fid, { VEC(F, aa, a, n); VEC(F, bb, b, n); std::vector<F> o(n); axpy<F>(n, aa.data(), F::from_canonical(r), bb.data(), o.data()); OUT(o, out); });
}
// The same two vector operations over `threads` host threads, conversions included (the CPU baseline of bench.py times them on all
// cores; the single-threaded forms above are what the parity tests call).
void orc_cross_term_mt(int fid, size_t n, const u64* az1, const u64* bz1, const u64* cz1, const u64* u1,
                       const u64* az2, const u64* bz2, const u64* cz2, const u64* u2, u64* T, int threads) {
  FIELD_SWITCH(fid, {
    const F U1 = F::from_canonical(u1); const F U2 = F::from_canonical(u2);
    par_chunks(n, threads, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; i++) {
        const F a1 = F::from_canonical(az1 + 4 * i); const F b1 = F::from_canonical(bz1 + 4 * i); const F c1 = F::from_canonical(cz1 + 4 * i);
        const F a2 = F::from_canonical(az2 + 4 * i); const F b2 = F::from_canonical(bz2 + 4 * i); const F c2 = F::from_canonical(cz2 + 4 * i);
        F t; cross_term<F>(1, &a1, &b1, &c1, U1, &a2, &b2, &c2, U2, &t);
        t.to_canonical(T + 4 * i);
      }
    });
  });
}
void orc_axpy_mt(int fid, size_t n, const u64* a, const u64* r, const u64* b, u64* out, int threads) {
  FIELD_SWITCH(fid, {
    const F R = F::from_canonical(r);
    par_chunks(n, threads, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; i++) { const F x = F::from_canonical(a + 4 * i); const F y = F::from_canonical(b + 4 * i); F o; axpy<F>(1, &x, R, &y, &o); o.to_canonical(out + 4 * i); }
    });
  });
}
// returns -1 if Az∘Bz == u·Cz + E for every row, else the first failing row.  E may be NULL (= 0).
long orc_first_unsat(int fid, size_t n, const u64* az, const u64* bz, const u64* cz, const u64* u, const u64* E) {
  long res = -1;
  FIELD_SWITCH(fid, {
    VEC(F, a, az, n); VEC(F, b, bz, n); VEC(F, c, cz, n);
    std::vector<F> e; if (E) { e.resize(n); for (size_t i = 0; i < n; i++) e[i] = F::from_canonical(E + 4 * i); }
    res = first_unsat_relaxed<F>(n, a.data(), b.data(), c.data(), F::from_canonical(u), E ? e.data() : nullptr);
  });
  return res;
}

// ---------------- witness program executor ----------------
// sizes[] = {n_wires, len_z, n_priv, n_decomp, n_groups, n_jobs, n_chains, n_fops}; tables are the raw POD arrays
// exported by the product (format: witness.hpp).  z_wires_out: n_wires x 4 canonical; z_state_out: len_z x 4.
int orc_witness_execute(const uint32_t* sizes, const void* decomp, const void* groups, const void* instr, const void* rows,
                        const void* jobs, const void* chains, const void* fops, const void* zout, const void* lc_terms, const u64* dict_canon,
                        const u64* z_in, const u64* priv, u64* z_wires_out, u64* z_state_out) {
  wp::Program P;
  P.n_wires = sizes[0]; P.len_z = sizes[1]; P.n_priv = sizes[2];
  P.decomp = (const wp::DecompGroup*)decomp; P.n_decomp = sizes[3];
  P.groups = (const wp::LaneGroup*)groups; P.n_groups = sizes[4];
  P.instr = (const wp::LaneInstr*)instr; P.rows = (const wp::LaneRow*)rows;
  P.jobs = (const wp::HashJob*)jobs; P.n_jobs = sizes[5];
  P.chains = (const wp::Chain*)chains; P.n_chains = sizes[6];
  P.fops = (const wp::FieldOp*)fops; P.n_fops = sizes[7];
  P.zout = (const wp::ZOut*)zout;
  P.lc_terms = (const wp::LcTerm*)lc_terms; P.dict_canon = dict_canon;
  std::vector<BnFr> z, zo;
  int st = wp::execute(P, z_in, priv, z, zo);
  if (st == 2) return 2;
  if (z_wires_out) for (size_t i = 0; i < z.size(); i++) z[i].to_canonical(z_wires_out + 4 * i);
  if (z_state_out) for (size_t i = 0; i < zo.size(); i++) zo[i].to_canonical(z_state_out + 4 * i);
  return st;
}
int orc_witness_struct_sizes(int which) {
  switch (which) {
    case 0: return sizeof(wp::DecompGroup); case 1: return sizeof(wp::LaneGroup); case 2: return sizeof(wp::LaneInstr);
    case 3: return sizeof(wp::LaneRow); case 4: return sizeof(wp::HashJob); case 5: return sizeof(wp::Chain);
    case 6: return sizeof(wp::FieldOp); default: return sizeof(wp::ZOut);
  }
}
// A z, B z, C z with dictionary-compressed CSR (coef = index into dict, canonical); then the first row where
// Az*Bz != Cz (or -1).  Also returns the three products when the out pointers are non-NULL.
long orc_r1cs_check(size_t nrows, size_t ncols, const uint32_t* const* row_ptr, const uint32_t* const* col, const uint32_t* const* coef,
                    const u64* dict, size_t ndict, const u64* z, u64* az_out, u64* bz_out, u64* cz_out, int threads) {
  std::vector<BnFr> D(ndict), Z(ncols);
  for (size_t i = 0; i < ndict; i++) D[i] = BnFr::from_canonical(dict + 4 * i);
  for (size_t i = 0; i < ncols; i++) Z[i] = BnFr::from_canonical(z + 4 * i);
  std::vector<BnFr> out[3];
  for (int m = 0; m < 3; m++) {
    out[m].resize(nrows);
    auto work = [&, m](size_t lo, size_t hi) {
      for (size_t r = lo; r < hi; r++) {
        BnFr acc = BnFr::zero();
        for (uint32_t k = row_ptr[m][r]; k < row_ptr[m][r + 1]; k++) acc = acc + D[coef[m][k]] * Z[col[m][k]];
        out[m][r] = acc;
      }
    };
    std::vector<std::thread> th; size_t chunk = (nrows + threads - 1) / (threads > 0 ? threads : 1);
    for (int t = 0; t < threads; t++) th.emplace_back(work, std::min(nrows, t * chunk), std::min(nrows, (t + 1) * chunk));
    for (auto& x : th) x.join();
  }
  u64* outs[3] = {az_out, bz_out, cz_out};
  for (int m = 0; m < 3; m++) if (outs[m]) for (size_t r = 0; r < nrows; r++) out[m][r].to_canonical(outs[m] + 4 * r);
  for (size_t r = 0; r < nrows; r++) if (out[0][r] * out[1][r] != out[2][r]) return (long)r;
  return -1;
}

// ---------------- Nova IVC relation (nova.hpp) ----------------
// chain hash over field fid (0 = BN254 Fr, 1 = BN254 Fq)
void orc_nova_hash(int fid, const u64* in, int n, u64* out) {
  FIELD_SWITCH(fid, { std::vector<F> v(n); for (int i = 0; i < n; i++) v[i] = F::from_canonical(in + 4 * i); nova_hash<F>(v).to_canonical(out); });
}
// trunc250(H(digest, i, z_0, z, U)); U = 7 canonical elements (W.x, W.y, E.x, E.y, u, X0, X1)
void orc_nova_instance_hash(int fid, const u64* pz, u64 i, const u64* z_0, const u64* z, int len_z, const u64* U, u64* out) {
  FIELD_SWITCH(fid, {
    std::vector<F> z0(len_z), zz(len_z); for (int k = 0; k < len_z; k++) { z0[k] = F::from_canonical(z_0 + 4 * k); zz[k] = F::from_canonical(z + 4 * k); }
    low_bits(nova_instance_hash_full<F>(F::from_canonical(pz), i, z0, zz, load_relaxed<F>(U)), 250).to_canonical(out);
  });
}
// side 0: circuit over Fr folding Grumpkin commitments; side 1: circuit over Fq folding BN254-G1 commitments.
// Returns 1 when the relation holds (incoming hash matches or base case), 0 otherwise.
int orc_nova_step(int side, int is_primary, const u64* pz, u64 i, const u64* z_0, const u64* z_i, const u64* z_next, int len_z, const u64* U, const u64* u, const u64* T,
                  u64* U_new, u64* rho, u64* x1) {
  return side == 0 ? nova_step_c<Grumpkin, BnFq>(is_primary, pz, i, z_0, z_i, z_next, len_z, U, u, T, U_new, rho, x1)
                   : nova_step_c<BnG1, BnFr>(is_primary, pz, i, z_0, z_i, z_next, len_z, U, u, T, U_new, rho, x1);
}
// Relaxed R1CS over field fid with dictionary-compressed CSR: first row where Az∘Bz != u·Cz + E (E may be NULL), or -1.
long orc_r1cs_check_relaxed(int fid, size_t nrows, size_t ncols, const uint32_t* const* row_ptr, const uint32_t* const* col, const uint32_t* const* coef,
                            const u64* dict, size_t ndict, const u64* z, const u64* u, const u64* E, int threads) {
  FIELD_SWITCH(fid, return r1cs_check_relaxed_t<F>(nrows, ncols, row_ptr, col, coef, dict, ndict, z, u, E, threads));
  return -2;
}

}  // extern "C"
