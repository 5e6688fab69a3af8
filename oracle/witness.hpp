// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).
//
// Independent CPU executor of the product's witness program (the data a step circuit is compiled to,
// format documented in vimz_amd/csrc/circuit/program.hpp and re-declared here so that no product code
// is linked): the counterpart of running circom's generated witness calculator for one step, which is
// what nova-scotia does per fold (reference call site
// vimz/src/nova_snark_backend/folding.rs:35-41 via nova_scotia::create_recursive_circuit).
// Semantics of the operations follow circomlib and poseidon.hpp.
#pragma once
#include <cstdint>
#include <vector>
#include "poseidon.hpp"

namespace orc {
namespace wp {

struct ValRef { uint32_t kind, idx; };  // 0 const zero, 1 wire, 2 job output, 3 field-op output, 4 step_in
struct DecompGroup { uint32_t src_wire, count, bit_base, nbits; };
struct LaneInstr { uint8_t op, d, a, b; int32_t imm, imm2; };
struct LaneRow { uint32_t src_wire, count; };
struct LaneGroup { uint32_t lanes, pixels, colours, wire_base, slots, prog_off, prog_len, row_off, row_cnt, rows_out, row_stride_a, pad; };
struct HashJob { uint32_t t, wire_base, out_wire, chain; ValRef in[8]; };
struct Chain { uint32_t job_off, job_cnt, phase, pad; };
struct FieldOp { uint32_t op, wire, bound, early; ValRef a, b, c; };
struct LcTerm { uint32_t wire, coef; };
struct ZOut { ValRef ref; int64_t add; };

enum { LOP_LDB = 1, LOP_LDZ, LOP_LI, LOP_ADD, LOP_SUB, LOP_MUL, LOP_MULI, LOP_ADDI, LOP_LEQ, LOP_SEL, LOP_BITS, LOP_EMIT, LOP_ROWSEL,
       LOP_LANE, LOP_ANDI, LOP_SHRI, LOP_EQ, LOP_LDBR };

struct Program {
  uint32_t n_wires, len_z, n_priv;
  const DecompGroup* decomp; size_t n_decomp;
  const LaneGroup* groups; size_t n_groups;
  const LaneInstr* instr;
  const LaneRow* rows;
  const HashJob* jobs; size_t n_jobs;
  const Chain* chains; size_t n_chains;
  const FieldOp* fops; size_t n_fops;
  const ZOut* zout;
  const LcTerm* lc_terms; const u64* dict_canon;   // for FOP_LC (may be null when the program has none)
};

static inline BnFr fe_i64(long long v) { return BnFr::from_i64(v); }

// Numeric Poseidon that also records the S-box wires in the layout the builder assigns:
// (round, lane) order, x2,x4,x5 per S-box; round-0 S-boxes on constant inputs are folded (no wires);
// when the job is bound to an output wire the last round's lane-0 x5 is not a wire.
static inline BnFr poseidon_job(const HashJob& J, const BnFr* in, const bool* in_const, std::vector<BnFr>& z) {
  const int t = (int)J.t;
  const PoseidonParams& P = poseidon_params(t);
  std::vector<BnFr> s(t), u(t);
  s[0] = BnFr::zero();
  for (int i = 1; i < t; i++) s[i] = in[i - 1];
  uint32_t w = J.wire_base;
  const int R = P.rf + P.rp, half = P.rf / 2;
  for (int r = 0; r < R; r++) {
    const bool full = r < half || r >= half + P.rp;
    for (int i = 0; i < t; i++) {
      s[i] = s[i] + P.C[r * t + i];
      if (!(full || i == 0)) continue;
      BnFr x2 = s[i].sqr(), x4 = x2.sqr(), x5 = x4 * s[i];
      const bool folded = r == 0 && (i == 0 || in_const[i - 1]);
      if (!folded) {
        z[w++] = x2; z[w++] = x4;
        const bool eliminated = J.out_wire != 0 && r == R - 1 && i == 0;
        if (!eliminated) z[w++] = x5;
      }
      s[i] = x5;
    }
    for (int i = 0; i < t; i++) {
      BnFr acc = BnFr::zero();
      for (int j = 0; j < t; j++) acc = acc + P.M[i * t + j] * s[j];
      u[i] = acc;
    }
    s.swap(u);
  }
  if (J.out_wire) z[J.out_wire] = s[0];
  return s[0];
}

// Executes the whole program for one step.  Returns 0 on success, 1 if the step relation cannot be satisfied
// (a range decomposition does not exist), 2 on malformed input.  z (n_wires) and z_out are written.
static inline int execute(const Program& Pg, const u64* z_in_canon, const u64* priv_canon, std::vector<BnFr>& z, std::vector<BnFr>& z_out) {
  int status = 0;
  z.assign(Pg.n_wires, BnFr::zero());
  z[0] = BnFr::one();
  const uint32_t in0 = 1 + Pg.len_z, priv0 = 1 + 2 * Pg.len_z;
  for (uint32_t i = 0; i < Pg.len_z; i++) z[in0 + i] = BnFr::from_canonical(z_in_canon + 4 * i);
  for (uint32_t i = 0; i < Pg.n_priv; i++) z[priv0 + i] = BnFr::from_canonical(priv_canon + 4 * i);
  auto priv_limbs = [&](uint32_t wire) { return priv_canon + 4 * (size_t)(wire - priv0); };

  // 1. bit decompositions
  for (size_t g = 0; g < Pg.n_decomp; g++) {
    const DecompGroup& D = Pg.decomp[g];
    for (uint32_t j = 0; j < D.count; j++) {
      const u64* v = priv_limbs(D.src_wire + j);
      if (D.nbits < 256 && ((D.nbits >= 192 ? v[3] >> (D.nbits - 192) : 1) != 0)) status = 1;  // does not fit nbits
      for (uint32_t k = 1; k < D.nbits; k++) if ((v[k / 64] >> (k % 64)) & 1) z[D.bit_base + (k - 1) * D.count + j] = BnFr::one();
    }
  }
  // 2. lane programs
  for (size_t g = 0; g < Pg.n_groups; g++) {
    const LaneGroup& G = Pg.groups[g];
    for (uint32_t lane = 0; lane < G.lanes; lane++) {
      long long r[64] = {0};
      const uint32_t x = lane % G.pixels, col = (lane / G.pixels) % G.colours;
      for (uint32_t pc = 0; pc < G.prog_len; pc++) {
        const LaneInstr& I = Pg.instr[G.prog_off + pc];
        switch (I.op) {
          case LOP_LDB: {
            const LaneRow& R = Pg.rows[G.row_off + I.a];
            const long px = (long)x * I.imm2 + I.imm;
            const int c = I.b == 3 ? (int)col : I.b;
            if (px < 0 || px >= (long)R.count * 10) { r[I.d] = 0; break; }
            const u64* v = priv_limbs(R.src_wire + (uint32_t)(px / 10));
            const int byte = (int)(px % 10) * 3 + c;
            r[I.d] = (long long)((v[byte / 8] >> (8 * (byte % 8))) & 0xff);
            break;
          }
          case LOP_LDZ: {
            const u64* v = z_in_canon + 4 * I.imm;
            if (v[1] | v[2] | v[3] || v[0] >> 40) status = 1;
            r[I.d] = (long long)(v[0] & ((1ull << 40) - 1));
            break;
          }
          case LOP_LI: r[I.d] = I.imm; break;
          case LOP_ADD: r[I.d] = r[I.a] + r[I.b]; break;
          case LOP_SUB: r[I.d] = r[I.a] - r[I.b]; break;
          case LOP_MUL: r[I.d] = r[I.a] * r[I.b]; break;
          case LOP_MULI: r[I.d] = r[I.a] * I.imm; break;
          case LOP_ADDI: r[I.d] = r[I.a] + I.imm; break;
          case LOP_LEQ: {  // LessEqThan(n): bit n of (a + 2^n - (b+1)) clear
            const long long v = r[I.a] + (1ll << I.imm) - (r[I.b] + 1);
            r[I.d] = ((v >> I.imm) & 1) ? 0 : 1;
            break;
          }
          case LOP_SEL: r[I.d] = r[I.a] ? r[I.b] : r[I.imm]; break;
          case LOP_BITS: {
            const long long v = r[I.a];
            if (v < 0 || v >= (1ll << I.imm)) { status = 1; break; }
            for (int k = 1; k < I.imm; k++) if ((v >> k) & 1) z[G.wire_base + (uint32_t)(I.imm2 + k - 1) * G.lanes + lane] = BnFr::one();
            break;
          }
          case LOP_EMIT: z[G.wire_base + (uint32_t)I.imm * G.lanes + lane] = fe_i64(r[I.a]); break;
          case LOP_LANE: r[I.d] = (long long)x + I.imm; break;
          case LOP_ANDI: r[I.d] = r[I.a] & (long long)I.imm; break;
          case LOP_SHRI: r[I.d] = r[I.a] >> I.imm; break;
          case LOP_EQ: r[I.d] = r[I.a] == r[I.b] ? 1 : 0; break;
          case LOP_LDBR: {
            const LaneRow& R = Pg.rows[G.row_off + I.a];
            const long long px = (long long)x * I.imm2 + I.imm + r[I.b];
            if (px < 0 || px >= (long long)R.count * 10) { r[I.d] = 0; break; }
            const u64* v = priv_limbs(R.src_wire + (uint32_t)(px / 10));
            const int byte = (int)(px % 10) * 3;
            r[I.d] = (long long)((v[byte / 8] >> (8 * (byte % 8))) & 0xff);
            break;
          }
          default: return 2;
        }
      }
    }
  }
  // 3. hash jobs (phase A chains, then phase B) and field ops
  std::vector<BnFr> job_out(Pg.n_jobs, BnFr::zero()), fop_out(Pg.n_fops, BnFr::zero());
  auto value = [&](const ValRef& v) -> BnFr {
    switch (v.kind) {
      case 1: return z[v.idx];
      case 2: return job_out[v.idx];
      case 3: return fop_out[v.idx];
      case 4: return z[in0 + v.idx];
      default: return BnFr::zero();
    }
  };
  auto run_fops = [&](uint32_t early) -> int {
    for (size_t f = 0; f < Pg.n_fops; f++) {
      const FieldOp& F = Pg.fops[f];
      if (F.early != early) continue;
      if (F.op == 1) {  // IsZero: inv, out
        BnFr in = value(F.a);
        BnFr inv = in.is_zero() ? BnFr::zero() : in.inv();
        BnFr out = BnFr::one() - in * inv;
        z[F.wire] = inv; z[F.wire + 1] = out;
        fop_out[f] = out;
      } else if (F.op == 2) {  // Mux1 on full-width values
        BnFr s = value(F.a), c0 = value(F.b), c1 = value(F.c);
        BnFr prod = (c1 - c0) * s, out = prod + c0;
        z[F.wire] = F.bound ? out : prod;
        fop_out[f] = out;
      } else if (F.op == 3) {  // value of a stored linear combination of wires
        if (!Pg.lc_terms || !Pg.dict_canon) return 2;
        BnFr acc = BnFr::zero();
        for (uint32_t k = 0; k < F.b.idx; k++) {
          const LcTerm& T = Pg.lc_terms[F.a.idx + k];
          acc = acc + BnFr::from_canonical(Pg.dict_canon + 4 * (size_t)T.coef) * z[T.wire];
        }
        fop_out[f] = acc;
      } else return 2;
    }
    return 0;
  };
  const int phase_order[3] = {0, 2, 1};   // row hashes; early field ops, then the chains that use them; the state hashes
  for (int pi = 0; pi < 3; pi++) {
    const int phase = phase_order[pi];
    if (phase == 2 && run_fops(1)) return 2;
    for (size_t c = 0; c < Pg.n_chains; c++) {
      const Chain& C = Pg.chains[c];
      if ((int)C.phase != phase) continue;
      for (uint32_t k = 0; k < C.job_cnt; k++) {
        const HashJob& J = Pg.jobs[C.job_off + k];
        BnFr in[8]; bool cst[8];
        for (uint32_t i = 0; i + 1 < J.t; i++) { in[i] = value(J.in[i]); cst[i] = J.in[i].kind == 0; }
        job_out[C.job_off + k] = poseidon_job(J, in, cst, z);
      }
    }
  }
  if (run_fops(0)) return 2;
  z_out.resize(Pg.len_z);
  for (uint32_t i = 0; i < Pg.len_z; i++) {
    z_out[i] = value(Pg.zout[i].ref) + fe_i64(Pg.zout[i].add);
    z[1 + i] = z_out[i];
  }
  return status;
}

}  // namespace wp
}  // namespace orc
