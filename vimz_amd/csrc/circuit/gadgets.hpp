// Gadgets: the circomlib / VIMz templates re-stated once, each emitting BOTH the R1CS rows and the
// witness-program entries (program.hpp) that compute the wires it introduces.
//   Num2Bits / LessEqThan / Mux1 / IsZero   circomlib
//   Decompressor / DecompressorGray          circuits/src/utils/pixels.circom:6-29, 67-89
//   Poseidon / PairHasher / ArrayHasher / HeadTailHasher   circuits/src/utils/hashers.circom:7-73,115-120
#pragma once
#include <algorithm>
#include <memory>
#include "builder.hpp"
#include "poseidon_params.hpp"

namespace vz {
namespace cb {

// ====================================================================================================
// Field-level values (hash inputs/outputs, IVC state)
// ====================================================================================================
struct FV { LC lc; ValRef ref; };
inline FV fv_wire(uint32_t w) { return FV{LC::wire(w), ValRef{REF_WIRE, w}}; }
inline FV fv_zero() { return FV{LC(), ValRef{REF_CONST_ZERO, 0}}; }

// ---- Poseidon ----------------------------------------------------------------------------------------
// Template of one permutation over local variables (0 = one, v>=1 = S-box wire slot v-1), shared by every
// instance with the same width and the same set of constant-zero inputs.
struct PoseidonTemplate {
  struct LTerm { uint32_t var; uint32_t coef; };
  struct Sbox {
    int round, lane;
    int input_lane;              // >= 0: round-0 S-box whose input is (instance input `lane`) + C ; -1: `in` below
    std::vector<LTerm> in;       // local LC of the S-box input (rounds >= 1); for round 0 only the constant term
    uint32_t slot;               // x2 = slot, x4 = slot+1, x5 = slot+2
  };
  int t; uint32_t const_mask;
  std::vector<Sbox> sboxes;
  std::vector<LTerm> out;        // state[0] after the last round
  uint32_t n_slots;
  uint32_t last_lane0_x5_slot;   // the wire eliminated when the output is bound to a public-output wire
  Fe m00_inv;                    // 1 / (coefficient of that x5 in `out`)
};

struct LocalLC {  // LC over local variables with explicit Fe coefficients (template construction only)
  std::vector<std::pair<uint32_t, Fe>> t;
  static LocalLC constant(const Fe& c) { LocalLC r; if (!c.is_zero()) r.t.push_back({0, c}); return r; }
  static LocalLC var(uint32_t v) { LocalLC r; r.t.push_back({v, Fe::one()}); return r; }
  void add_scaled(const LocalLC& b, const Fe& k) {
    std::vector<std::pair<uint32_t, Fe>> r; r.reserve(t.size() + b.t.size());
    size_t i = 0, j = 0;
    while (i < t.size() || j < b.t.size()) {
      if (j >= b.t.size() || (i < t.size() && t[i].first < b.t[j].first)) r.push_back(t[i++]);
      else if (i >= t.size() || b.t[j].first < t[i].first) { Fe c = Fe::mul(b.t[j].second, k); if (!c.is_zero()) r.push_back({b.t[j].first, c}); j++; }
      else { Fe c = Fe::add(t[i].second, Fe::mul(b.t[j].second, k)); if (!c.is_zero()) r.push_back({t[i].first, c}); i++; j++; }
    }
    t.swap(r);
  }
  bool is_const() const { return t.empty() || (t.size() == 1 && t[0].first == 0); }
  Fe const_value() const { return t.empty() ? Fe::zero() : t[0].second; }
};

struct Gadgets {
  Builder& b;
  std::map<std::pair<int, uint32_t>, std::unique_ptr<PoseidonTemplate>> templates;
  int cur_chain = -1;

  explicit Gadgets(Builder& bb) : b(bb) {}

  const PoseidonTemplate& poseidon_template(int t, uint32_t const_mask) {
    auto key = std::make_pair(t, const_mask);
    auto it = templates.find(key);
    if (it != templates.end()) return *it->second;
    const PoseidonTable& P = poseidon_table(t);
    auto T = std::make_unique<PoseidonTemplate>();
    T->t = t; T->const_mask = const_mask;
    std::vector<LocalLC> st(t);
    // round-0 inputs: lane 0 and masked lanes are the constant 0; the others are instance inputs, represented
    // here by a placeholder variable that never reaches an LC (round 0 always applies the S-box to every lane).
    uint32_t next_slot = 0;
    const int R = P.rf + P.rp;
    for (int r = 0; r < R; r++) {
      const bool full = r < P.rf / 2 || r >= P.rf / 2 + P.rp;
      for (int i = 0; i < t; i++) {
        const Fe& c = P.C[(size_t)r * t + i];
        const bool sbox_here = full || i == 0;
        if (r == 0) {
          const bool is_const_in = (i == 0) || ((const_mask >> i) & 1);
          if (is_const_in) {  // constant-folded S-box: (0 + C)^5
            Fe x2 = Fe::sqr(c), x4 = Fe::sqr(x2);
            st[i] = LocalLC::constant(Fe::mul(x4, c));
          } else {
            PoseidonTemplate::Sbox s; s.round = 0; s.lane = i; s.input_lane = i; s.slot = next_slot; next_slot += 3;
            s.in.push_back({0, b.coef_id(c)});
            T->sboxes.push_back(std::move(s));
            st[i] = LocalLC::var(T->sboxes.back().slot + 2 + 1);
          }
          continue;
        }
        LocalLC in = st[i];
        in.add_scaled(LocalLC::constant(c), Fe::one());
        if (!sbox_here) { st[i] = in; continue; }
        if (in.is_const()) {  // cannot happen after round 0 for these parameters, kept for completeness
          Fe v = in.const_value(); Fe x2 = Fe::sqr(v), x4 = Fe::sqr(x2);
          st[i] = LocalLC::constant(Fe::mul(x4, v));
          continue;
        }
        PoseidonTemplate::Sbox s; s.round = r; s.lane = i; s.input_lane = -1; s.slot = next_slot; next_slot += 3;
        for (auto& term : in.t) s.in.push_back({term.first, b.coef_id(term.second)});
        T->sboxes.push_back(std::move(s));
        st[i] = LocalLC::var(T->sboxes.back().slot + 2 + 1);
      }
      std::vector<LocalLC> nx(t);
      for (int i = 0; i < t; i++) for (int j = 0; j < t; j++) nx[i].add_scaled(st[j], P.M[(size_t)i * t + j]);
      st.swap(nx);
    }
    for (auto& term : st[0].t) T->out.push_back({term.first, b.coef_id(term.second)});
    T->n_slots = next_slot;
    // last round is full: its lane-0 S-box is the first S-box of that round
    for (auto& s : T->sboxes) if (s.round == R - 1 && s.lane == 0) T->last_lane0_x5_slot = s.slot + 2;
    T->m00_inv = Fe::pow_pm2(P.M[0]);
    auto* raw = T.get();
    templates.emplace(key, std::move(T));
    return *raw;
  }

  void begin_chain(int phase) {
    Chain c; c.job_off = (uint32_t)b.jobs.size(); c.job_cnt = 0; c.phase = (uint32_t)phase; c.pad = 0;
    b.chains.push_back(c);
    cur_chain = (int)b.chains.size() - 1;
  }

  // Poseidon(n)(inputs); if out_wire != 0 the result is bound to that (public output) wire.
  FV poseidon(const std::vector<FV>& in, uint32_t out_wire = 0) {
    const int t = (int)in.size() + 1;
    if (t > POSEIDON_MAX_T) throw std::runtime_error("poseidon: too many inputs");
    if (cur_chain < 0) throw std::runtime_error("poseidon outside a chain");
    uint32_t mask = 0;
    for (int i = 1; i < t; i++) if (in[i - 1].ref.kind == REF_CONST_ZERO) mask |= 1u << i;
    const PoseidonTemplate& T = poseidon_template(t, mask);
    const uint32_t n_w = T.n_slots - (out_wire ? 1 : 0);
    const uint32_t base = b.alloc(n_w);
    auto slot_wire = [&](uint32_t slot) -> uint32_t {  // slots after the eliminated x5 shift down by one
      if (out_wire && slot > T.last_lane0_x5_slot) return base + slot - 1;
      return base + slot;
    };
    // LC (in CSR form) of a local LC, with the eliminated variable substituted when bound
    auto push_local = [&](Csr& M, const std::vector<PoseidonTemplate::LTerm>& lc, const LC* extra) {
      // extra: an instance LC to add (round-0 inputs).  Local terms never contain the eliminated var
      // (it only occurs in `out`), so a plain remap is enough; merge with `extra` keeping wire order.
      std::vector<std::pair<uint32_t, uint32_t>> terms;  // (wire, coef id)
      terms.reserve(lc.size() + (extra ? extra->t.size() : 0));
      for (auto& x : lc) terms.push_back({x.var == 0 ? 0u : slot_wire(x.var - 1), x.coef});
      if (extra) {
        // merge: both sorted by wire; coefficients on equal wires must be added (only wire 0 can collide)
        std::vector<std::pair<uint32_t, uint32_t>> merged; merged.reserve(terms.size() + extra->t.size());
        size_t i = 0, j = 0;
        while (i < terms.size() || j < extra->t.size()) {
          if (j >= extra->t.size() || (i < terms.size() && terms[i].first < extra->t[j].w)) merged.push_back(terms[i++]);
          else if (i >= terms.size() || extra->t[j].w < terms[i].first) { merged.push_back({extra->t[j].w, b.coef_id(extra->t[j].c)}); j++; }
          else { Fe c = Fe::add(b.dict[terms[i].second], extra->t[j].c); if (!c.is_zero()) merged.push_back({terms[i].first, b.coef_id(c)}); i++; j++; }
        }
        terms.swap(merged);
      }
      for (auto& x : terms) { M.col.push_back(x.first); M.coef.push_back(x.second); }
      M.row_ptr.push_back((uint32_t)M.col.size());
    };
    const uint32_t one_id = b.coef_id(Fe::one());
    auto push_wire = [&](Csr& M, uint32_t w) { M.col.push_back(w); M.coef.push_back(one_id); M.row_ptr.push_back((uint32_t)M.col.size()); };
    for (auto& s : T.sboxes) {
      const LC* extra = s.input_lane >= 0 ? &in[s.input_lane - 1].lc : nullptr;
      const uint32_t x2 = slot_wire(s.slot), x4 = slot_wire(s.slot + 1);
      push_local(b.A, s.in, extra); push_local(b.B, s.in, extra); push_wire(b.C, x2);     // in * in = x2
      push_wire(b.A, x2); push_wire(b.B, x2); push_wire(b.C, x4);                          // x2 * x2 = x4
      push_wire(b.A, x4); push_local(b.B, s.in, extra);                                    // x4 * in = x5
      if (out_wire && s.slot + 2 == T.last_lane0_x5_slot) {
        // x5 = (out - sum_{other} coef * var) / m00
        LC rhs = LC::wire(out_wire);
        for (auto& x : T.out) {
          if (x.var - 1 == T.last_lane0_x5_slot) continue;
          rhs = LC::axpy(rhs, Fe::neg(b.dict[x.coef]), x.var == 0 ? LC::constant(Fe::one()) : LC::wire(slot_wire(x.var - 1)));
        }
        b.push_row(b.C, rhs.scaled(T.m00_inv));
      } else {
        push_wire(b.C, slot_wire(s.slot + 2));
      }
    }
    HashJob J; memset(&J, 0, sizeof(J));
    J.t = (uint32_t)t; J.wire_base = base; J.out_wire = out_wire; J.chain = (uint32_t)cur_chain;
    for (int i = 0; i < t - 1; i++) J.in[i] = in[i].ref;
    b.jobs.push_back(J);
    b.chains[cur_chain].job_cnt++;
    FV r;
    r.ref = ValRef{REF_JOB, (uint32_t)b.jobs.size() - 1};
    if (out_wire) r.lc = LC::wire(out_wire);
    else {
      for (auto& x : T.out) r.lc.t.push_back({x.var == 0 ? 0u : slot_wire(x.var - 1), b.dict[x.coef]});
      std::sort(r.lc.t.begin(), r.lc.t.end(), [](const Term& a, const Term& c) { return a.w < c.w; });
    }
    return r;
  }

  FV pair_hash(const FV& a, const FV& c, uint32_t out_wire = 0) { return poseidon({a, c}, out_wire); }

  // _WindowFoldHasher(L, 8): ceil(L/8) permutations, the tail of the array is never absorbed.
  FV array_hash(const std::vector<FV>& arr, uint32_t out_wire = 0) {
    const int L = (int)arr.size(), W = 8;
    const int rounds = (L + W - 1) / W;
    const int first = L < W ? L : W;
    std::vector<FV> in(arr.begin(), arr.begin() + first);
    FV h = poseidon(in, rounds == 1 ? out_wire : 0);
    int processed = first;
    for (int r = 0; r < rounds - 1; r++) {
      const int remaining = L - processed;
      const int cur = remaining < W - 1 ? remaining : W - 1;
      std::vector<FV> nx; nx.push_back(h);
      for (int i = 0; i < cur; i++) nx.push_back(arr[processed + i]);
      h = poseidon(nx, r == rounds - 2 ? out_wire : 0);
      processed += cur;
    }
    return h;
  }

  // ==================================================================================================
  // Bit decomposition of packed input rows
  // ==================================================================================================
  struct Decomp { uint32_t src, count, bit_base; };

  // Num2Bits(240) of each element of a packed row (Decompressor / DecompressorGray share it)
  Decomp decompress_row(uint32_t src_wire, uint32_t count) {
    Decomp d; d.src = src_wire; d.count = count; d.bit_base = b.alloc(239 * count);
    DecompGroup g; g.src_wire = src_wire; g.count = count; g.bit_base = d.bit_base; g.nbits = 240;
    b.decomp.push_back(g);
    const LC one = LC::constant(Fe::one());
    for (uint32_t j = 0; j < count; j++) {
      // bit 0 is the substituted signal: b0 = in - sum_{k>=1} 2^k b_k
      b.enforce(bit0_lc(d, j), bit0_lc(d, j) - one, LC());
      for (int k = 1; k < 240; k++) { LC w = LC::wire(bit_wire(d, j, k)); b.enforce(w, w - one, LC()); }
    }
    return d;
  }
  static uint32_t bit_wire(const Decomp& d, uint32_t j, int k) { return d.bit_base + (uint32_t)(k - 1) * d.count + j; }
  LC bit0_lc(const Decomp& d, uint32_t j) {
    LC r; r.t.reserve(240);
    r.t.push_back({d.src + j, Fe::one()});
    std::vector<Term> bits;
    for (int k = 1; k < 240; k++) bits.push_back({bit_wire(d, j, k), Fe::neg(fe_pow2(k))});
    LC bl; bl.t = bits;  // wires increasing in k (bit-major layout) -> already sorted
    return r + bl;
  }
  // value of byte `byte_idx` (0..29) of element j as an LC over its bits
  LC byte_lc(const Decomp& d, uint32_t j, int byte_idx) {
    LC r;
    for (int m = 0; m < 8; m++) {
      const int k = 8 * byte_idx + m;
      if (k == 0) r.add_in_place(bit0_lc(d, j));
      else r.add_in_place(LC::wire(bit_wire(d, j, k), fe_pow2(m)));
    }
    return r;
  }

  // ==================================================================================================
  // Integer lane programs
  // ==================================================================================================
  struct LV { LC lc; int reg; };

  struct LaneCtx {
    Gadgets& g; Builder& b;
    LaneGroup grp;
    std::vector<Decomp> rows;        // rows[a] readable by LDB
    bool recording = false;          // true while running lane 0 (records the tape)
    bool counting = false;           // dry run: count slots only
    uint32_t lane = 0, slot = 0; int next_reg = 0;
    std::vector<LaneInstr> tape;

    LaneCtx(Gadgets& gg) : g(gg), b(gg.b) { memset(&grp, 0, sizeof(grp)); }
    int reg() { if (next_reg >= 60) throw std::runtime_error("lane program uses too many registers"); return next_reg++; }
    void ins(uint8_t op, int d, int a, int bb, int32_t imm, int32_t imm2) { if (recording) tape.push_back(LaneInstr{op, (uint8_t)d, (uint8_t)a, (uint8_t)bb, imm, imm2}); }
    uint32_t x() const { return lane % grp.pixels; }
    uint32_t colour() const { return (lane / grp.pixels) % grp.colours; }
    uint32_t wire_of_slot(uint32_t s) const { return grp.wire_base + s * grp.lanes + lane; }

    LV imm(int64_t v) { LV r; r.lc = LC::constant_i(v); r.reg = reg(); ins(LOP_LI, r.reg, 0, 0, (int32_t)v, 0); return r; }
    LV zin(int idx) { LV r; r.lc = LC::wire(1 + b.len_z + idx); r.reg = reg(); ins(LOP_LDZ, r.reg, 0, 0, idx, 0); return r; }
    // byte of row `row` at pixel x*xmul + dx, colour col (3 = lane colour); zero outside the row
    LV byte(int row, int dx, int col, int xmul = 1) {
      LV r; r.reg = reg(); ins(LOP_LDB, r.reg, row, col, dx, xmul);
      if (counting) return r;
      const long px = (long)x() * xmul + dx;
      const int c = col == 3 ? (int)colour() : col;
      const Decomp& d = rows[row];
      if (px < 0 || px >= (long)d.count * 10) return r;  // zero LC
      r.lc = g.byte_lc(d, (uint32_t)(px / 10), (int)(px % 10) * 3 + c);
      return r;
    }
    LV add(const LV& a, const LV& c) { LV r; if (!counting) r.lc = a.lc + c.lc; r.reg = reg(); ins(LOP_ADD, r.reg, a.reg, c.reg, 0, 0); return r; }
    LV sub(const LV& a, const LV& c) { LV r; if (!counting) r.lc = a.lc - c.lc; r.reg = reg(); ins(LOP_SUB, r.reg, a.reg, c.reg, 0, 0); return r; }
    LV muli(const LV& a, int32_t k) { LV r; if (!counting) r.lc = a.lc.scaled(fe_from_i64(k)); r.reg = reg(); ins(LOP_MULI, r.reg, a.reg, 0, k, 0); return r; }
    LV addi(const LV& a, int32_t k) { LV r; if (!counting) r.lc = a.lc.add_const(k); r.reg = reg(); ins(LOP_ADDI, r.reg, a.reg, 0, k, 0); return r; }
    // new signal <== a * c  (one constraint, one wire)
    LV mul(const LV& a, const LV& c) {
      LV r; r.reg = reg(); ins(LOP_MUL, r.reg, a.reg, c.reg, 0, 0);
      const uint32_t s = slot++; ins(LOP_EMIT, 0, r.reg, 0, (int32_t)s, 0);
      if (counting) return r;
      const uint32_t w = wire_of_slot(s);
      b.enforce(a.lc, c.lc, LC::wire(w));
      r.lc = LC::wire(w);
      return r;
    }
    // Num2Bits(nbits)(v): returns the LCs of the bits (bit 0 substituted)
    std::vector<LC> num2bits(const LV& v, int nbits) {
      const uint32_t s0 = slot; slot += (uint32_t)(nbits - 1);
      ins(LOP_BITS, 0, v.reg, 0, nbits, (int32_t)s0);
      std::vector<LC> bits;
      if (counting) return bits;
      bits.resize(nbits);
      LC rest;
      for (int k = 1; k < nbits; k++) { bits[k] = LC::wire(wire_of_slot(s0 + k - 1)); rest = rest + bits[k].scaled(fe_pow2(k)); }
      bits[0] = v.lc - rest;
      const LC one = LC::constant(Fe::one());
      for (int k = 0; k < nbits; k++) b.enforce(bits[k], bits[k] - one, LC());
      return bits;
    }
    // LessEqThan(n)(a, c) with a free output (n+1 constraints)
    LV less_eq(int n, const LV& a, const LV& c) {
      LV v = sub(addi(a, (int32_t)((1 << n) - 1)), c);   // a + 2^n - (c + 1)
      std::vector<LC> bits = num2bits(v, n + 1);
      LV r; r.reg = reg(); ins(LOP_LEQ, r.reg, a.reg, c.reg, n, 0);
      if (!counting) r.lc = LC::constant(Fe::one()) - bits[n];
      return r;
    }
    // LessEqThan(n)(a, c) === 1 : the top bit is the constant 0, n constraints remain
    void assert_less_eq(int n, const LV& a, const LV& c) {
      LV v = sub(addi(a, (int32_t)((1 << n) - 1)), c);
      num2bits(v, n);
    }
    // ---- crop support: lane index, masks/shifts, equality hint, register-relative byte load ----
    LV lane_index(int32_t add) { LV r; r.reg = reg(); ins(LOP_LANE, r.reg, 0, 0, add, 0); if (!counting) r.lc = LC::constant_i((int64_t)x() + add); return r; }
    LV andi(const LV& a, int32_t m, const LC& lc) { LV r; r.reg = reg(); ins(LOP_ANDI, r.reg, a.reg, 0, m, 0); r.lc = lc; return r; }   // caller supplies the LC
    LV shri(const LV& a, int32_t k) { LV r; r.reg = reg(); ins(LOP_SHRI, r.reg, a.reg, 0, k, 0); return r; }                            // value only
    // signal <-- (a == c): a hinted wire (no constraint of its own; the caller constrains it)
    LV eq_hint(const LV& a, const LV& c) {
      LV r; r.reg = reg(); ins(LOP_EQ, r.reg, a.reg, c.reg, 0, 0);
      const uint32_t s = slot++; ins(LOP_EMIT, 0, r.reg, 0, (int32_t)s, 0);
      if (!counting) r.lc = LC::wire(wire_of_slot(s));
      return r;
    }
    LV eq_value(const LV& a, const LV& c, const LC& lc) { LV r; r.reg = reg(); ins(LOP_EQ, r.reg, a.reg, c.reg, 0, 0); r.lc = lc; return r; }
    // byte (colour 0) of row `row` at pixel x + off(reg): value only, the caller supplies the LC
    LV byte_rel(int row, const LV& off, const LC& lc) { LV r; r.reg = reg(); ins(LOP_LDBR, r.reg, row, off.reg, 0, 1); r.lc = lc; return r; }
    void enforce(const LC& a, const LC& c, const LC& d) { if (!counting) b.enforce(a, c, d); }

    // Mux1: out = (c1 - c0) * s + c0
    LV mux(const LV& s, const LV& c0, const LV& c1) {
      LV d = sub(c1, c0);
      LV p = mul(d, s);
      LV r = add(p, c0);
      return r;
    }
  };

  // Runs `body(ctx)` for every lane of a group: a dry run sizes the group, lane 0 records the tape.
  template <class Body>
  void lane_group(uint32_t lanes, uint32_t pixels, uint32_t colours, const std::vector<Decomp>& rows, Body body) {
    LaneCtx probe(*this);
    probe.grp.lanes = lanes; probe.grp.pixels = pixels; probe.grp.colours = colours; probe.rows = rows;
    probe.counting = true; probe.recording = true;
    body(probe);
    LaneCtx cx(*this);
    cx.rows = rows;
    cx.grp.lanes = lanes; cx.grp.pixels = pixels; cx.grp.colours = colours;
    cx.grp.slots = probe.slot;
    cx.grp.wire_base = b.alloc(probe.slot * lanes);
    cx.grp.prog_off = (uint32_t)b.lane_instr.size(); cx.grp.prog_len = (uint32_t)probe.tape.size();
    cx.grp.row_off = (uint32_t)b.lane_rows.size(); cx.grp.row_cnt = (uint32_t)rows.size();
    for (auto& r : rows) b.lane_rows.push_back(LaneRow{r.src, r.count});
    b.lane_instr.insert(b.lane_instr.end(), probe.tape.begin(), probe.tape.end());
    b.lane_groups.push_back(cx.grp);
    for (uint32_t l = 0; l < lanes; l++) { cx.lane = l; cx.slot = 0; cx.next_reg = 0; body(cx); }
  }
};

}  // namespace cb
}  // namespace vz
