// R1CS + witness-program builder for the VIMz step circuits (host code, runs once at set-up).
//
// The reference obtains its R1CS by running `circom --O2 --r1cs` over circuits/nova_snark/*.circom
// (circuits/build_circuits.sh:47) and loading the file with nova_scotia::circom::reader::load_r1cs
// (vimz/src/nova_snark_backend/folding.rs:22).  Neither circom nor its outputs exist in this tree
// (SURVEY.md F4), so this builder re-states the templates directly as constraints, following circom's
// --O2 convention that only multiplications cost a constraint (linear relations are substituted away):
// the non-linear constraint counts it produces equal circuits/nova_snark/circuit_parameters.csv.
//
// Matrices are accumulated straight into CSR with a coefficient dictionary: the distinct coefficients
// (powers of two, small integers, Poseidon MDS products) number a few 10^4, so the GPU SpMV reads
// 8 bytes per non-zero (column + dictionary index) instead of 36.
#pragma once
#include <stdint.h>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>
#include <stdexcept>

#include "../fp.hpp"
#include "program.hpp"

namespace vz {
namespace cb {

template <class F> inline F f_from_u64(uint64_t v) { F x = F::zero(); x.v[0] = (uint32_t)v; x.v[1] = (uint32_t)(v >> 32); return F::to_mont(x); }
template <class F> inline F f_from_i64(int64_t v) { return v >= 0 ? f_from_u64<F>((uint64_t)v) : F::neg(f_from_u64<F>((uint64_t)(-v))); }
template <class F> inline F f_pow2(int k) {      // 2^k, k < 256 (a table: the builders ask for the same few powers millions of times)
  struct Table { F v[256]; Table() { for (int i = 0; i < 256; i++) { F x = F::zero(); x.v[i >> 5] = 1u << (i & 31); v[i] = F::to_mont(x); } } };
  static const Table T;
  return T.v[k];
}

template <class Fe> struct TermT { uint32_t w; Fe c; };

// Linear combination over wires; wire 0 is the constant one.  Terms sorted by wire, no zero coefficients.
template <class Fe>
struct LCT {
  typedef TermT<Fe> Term;
  std::vector<Term> t;
  LCT() {}
  static LCT constant(const Fe& c) { LCT r; if (!c.is_zero()) r.t.push_back({0, c}); return r; }
  static LCT constant_i(int64_t v) { return constant(f_from_i64<Fe>(v)); }
  static LCT wire(uint32_t w) { LCT r; r.t.push_back({w, Fe::one()}); return r; }
  static LCT wire(uint32_t w, const Fe& c) { LCT r; if (!c.is_zero()) r.t.push_back({w, c}); return r; }
  bool is_const() const { return t.empty() || (t.size() == 1 && t[0].w == 0); }
  Fe const_value() const { return t.empty() ? Fe::zero() : t[0].c; }
  LCT scaled(const Fe& k) const {
    LCT r; if (k.is_zero()) return r;
    r.t.reserve(t.size());
    for (auto& x : t) r.t.push_back({x.w, Fe::mul(x.c, k)});
    return r;
  }
  static LCT axpy(const LCT& a, const Fe& k, const LCT& b) {  // a + k*b
    LCT r; r.t.reserve(a.t.size() + b.t.size());
    // (sums and differences are nearly all of a builder's calls: k = ±1 needs no multiplication — 56 M of them, 2 of the 3.5 s of building crop_step(HD))
    const bool k1 = k.eq(Fe::one()), km1 = !k1 && k.eq(Fe::neg(Fe::one()));
    auto kb = [&](const Fe& c) { return k1 ? c : km1 ? Fe::neg(c) : Fe::mul(c, k); };
    size_t i = 0, j = 0;
    while (i < a.t.size() || j < b.t.size()) {
      if (j >= b.t.size() || (i < a.t.size() && a.t[i].w < b.t[j].w)) r.t.push_back(a.t[i++]);
      else if (i >= a.t.size() || b.t[j].w < a.t[i].w) { Fe c = kb(b.t[j].c); if (!c.is_zero()) r.t.push_back({b.t[j].w, c}); j++; }
      else { Fe c = Fe::add(a.t[i].c, kb(b.t[j].c)); if (!c.is_zero()) r.t.push_back({a.t[i].w, c}); i++; j++; }
    }
    return r;
  }
  LCT operator+(const LCT& b) const { return axpy(*this, Fe::one(), b); }
  LCT operator-(const LCT& b) const { return axpy(*this, Fe::neg(Fe::one()), b); }
  LCT add_const(int64_t v) const { return *this + constant_i(v); }
  // *this += b.  Sums that grow one fresh wire at a time (a multiplexer's products, a decoder's outputs) append — `a = a + b` merged, and copied, the
  // whole sum for every term: quadratic in the sum's length.  Same terms in the same order either way.
  void add_in_place(const LCT& b) {
    if (b.t.empty()) return;
    if (t.empty() || t.back().w < b.t.front().w) t.insert(t.end(), b.t.begin(), b.t.end());
    else *this = *this + b;
  }
};

typedef Fp<BnFr> Fe;                 // the step circuits' field (the reference compiles them for bn128)
typedef TermT<Fe> Term;
typedef LCT<Fe> LC;
inline Fe fe_from_u64(uint64_t v) { return f_from_u64<Fe>(v); }
inline Fe fe_from_i64(int64_t v) { return f_from_i64<Fe>(v); }
inline Fe fe_pow2(int k) { return f_pow2<Fe>(k); }

struct FeKey {
  uint32_t v[8];
  bool operator==(const FeKey& o) const { return memcmp(v, o.v, 32) == 0; }
};
struct FeKeyHash { size_t operator()(const FeKey& k) const { uint64_t h = 1469598103934665603ull; for (int i = 0; i < 8; i++) { h ^= k.v[i]; h *= 1099511628211ull; } return (size_t)h; } };

struct Csr {
  std::vector<uint32_t> row_ptr{0};
  std::vector<uint32_t> col;
  std::vector<uint32_t> coef;  // index into Builder::dict
};

template <class Fe>
struct BuilderT {
  typedef LCT<Fe> LC;
  uint32_t n_wires = 1;   // wire 0 = one
  uint32_t len_z = 0, n_priv = 0;
  Csr A, B, C;
  std::vector<Fe> dict;
  std::unordered_map<FeKey, uint32_t, FeKeyHash> dict_ix;
  uint32_t n_linear = 0;  // constraints kept although linear (circom reports them separately)
  uint32_t n_bool = 0;    // rows [0, n_bool) are b·(b − 1) = 0 with A = the wire b, B = b − 1, C empty (group_boolean_rows_first)

  // witness program
  std::vector<DecompGroup> decomp;
  std::vector<LaneGroup> lane_groups;
  std::vector<LaneInstr> lane_instr;
  std::vector<LaneRow> lane_rows;
  std::vector<HashJob> jobs;
  std::vector<Chain> chains;
  std::vector<FieldOp> fops;
  std::vector<LcTerm> lc_terms;
  bool gpu_witness = true;   // false: a circuit whose witness program the GPU kernels cannot run (none at present)
  std::vector<ZOut> zout;

  uint32_t alloc(uint32_t n) { uint32_t b = n_wires; n_wires += n; return b; }
  uint32_t n_constraints() const { return (uint32_t)A.row_ptr.size() - 1; }

  uint32_t coef_id(const Fe& c) {
    FeKey k; memcpy(k.v, c.v, 32);
    auto it = dict_ix.find(k);
    if (it != dict_ix.end()) return it->second;
    uint32_t id = (uint32_t)dict.size(); dict.push_back(c); dict_ix.emplace(k, id);
    return id;
  }
  void push_row(Csr& M, const LC& lc) {
    for (auto& x : lc.t) { M.col.push_back(x.w); M.coef.push_back(coef_id(x.c)); }
    M.row_ptr.push_back((uint32_t)M.col.size());
  }
  void enforce(const LC& a, const LC& b, const LC& c) { push_row(A, a); push_row(B, b); push_row(C, c); }
  // a * b = new wire
  uint32_t mul_wire(const LC& a, const LC& b) { uint32_t w = alloc(1); enforce(a, b, LC::wire(w)); return w; }
  void mul_into(const LC& a, const LC& b, uint32_t w) { enforce(a, b, LC::wire(w)); }
};

typedef BuilderT<Fe> Builder;

// Constraint order is ours to choose (circom's is not pinned by anything the reference commits).  The boolean constraints b·(b − 1) = 0 —
// nineteen in twenty of a step circuit's rows, interleaved with a comparator's or a multiplexer's row every fifteen or so — are moved to
// the front, stably: the element-wise kernels of a fold (k_fold_cross, k_cross_term, k_fold5) skip their field multiplications for a
// WAVE whose fresh products are all 0 / ±1 (r1cs_ops.hpp: mul_fresh), and only rows of one kind side by side make whole waves of them.
template <class Fe>
inline uint32_t group_boolean_rows_first(BuilderT<Fe>& b) {
  const uint32_t n = b.n_constraints();
  const uint32_t one = b.coef_id(Fe::one()), mone = b.coef_id(Fe::neg(Fe::one()));
  auto len = [](const Csr& M, uint32_t r) { return M.row_ptr[r + 1] - M.row_ptr[r]; };
  std::vector<uint32_t> order; order.reserve(n);
  std::vector<uint8_t> boolean(n, 0);
  uint32_t nb = 0;
  for (uint32_t r = 0; r < n; r++) {
    if (len(b.A, r) != 1 || len(b.B, r) != 2 || len(b.C, r) != 0) continue;
    const uint32_t a = b.A.row_ptr[r], k = b.B.row_ptr[r];
    const uint32_t w = b.A.col[a];
    if (w == 0 || b.A.coef[a] != one) continue;
    if (b.B.col[k] != 0 || b.B.coef[k] != mone || b.B.col[k + 1] != w || b.B.coef[k + 1] != one) continue;
    boolean[r] = 1; nb++;
  }
  for (uint32_t r = 0; r < n; r++) if (boolean[r]) order.push_back(r);
  for (uint32_t r = 0; r < n; r++) if (!boolean[r]) order.push_back(r);
  auto permute = [&](Csr& M) {
    Csr o; o.row_ptr.reserve(n + 1); o.col.reserve(M.col.size()); o.coef.reserve(M.coef.size());
    for (uint32_t r : order) {
      for (uint32_t k = M.row_ptr[r]; k < M.row_ptr[r + 1]; k++) { o.col.push_back(M.col[k]); o.coef.push_back(M.coef[k]); }
      o.row_ptr.push_back((uint32_t)o.col.size());
    }
    M = std::move(o);
  };
  permute(b.A); permute(b.B); permute(b.C);
  b.n_bool = nb;
  return nb;
}

}  // namespace cb
}  // namespace vz
