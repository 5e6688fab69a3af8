// GPU witness generation for a BATCH of folding steps — replaces the circom witness generator that
// nova-scotia spawns as a child process once per step (reference call site
// vimz/src/nova_snark_backend/folding.rs:35-41).  Executes the witness program (circuit/program.hpp).
//
// All rows of a batch are independent given the IVC states, so the batch dimension supplies the
// parallelism the sequential Poseidon chains lack:
//   k_wit_inputs   z[0] = 1, step_out, step_in, private inputs (canonical -> Montgomery)
//   k_wit_decomp   Num2Bits(240) of every packed element; thread = (row, bit, element), bit-major wires
//                  so consecutive lanes write consecutive 32-byte wires
//   k_wit_lanes    the integer lane programs; thread = (row, lane), registers in LDS, slot-major wires
//   k_wit_chains   Poseidon chains; 16 lanes co-operate on one permutation (lane i owns state[i], the MDS
//                  row is gathered with wave shuffles), 4 chains per wave
//   k_wit_fops     IsZero / Mux1 on full-width values
// Layout in HBM: Z[row][wire], 32-byte Montgomery elements, row stride = n_wires.
#pragma once
#include <hip/hip_runtime.h>
#include "fp.hpp"
#include "fp29.hpp"
#include "circuit/program.hpp"
#include "vecops.hpp"

namespace vz {

typedef Fp<BnFr> Fr;

struct WitnessDev {  // device copies of the program tables
  const DecompGroup* decomp; uint32_t n_decomp;
  const LaneGroup* groups; uint32_t n_groups;
  const LaneInstr* instr;
  const LaneRow* rows;
  const HashJob* jobs; uint32_t n_jobs;
  const Chain* chains; uint32_t n_chains;
  const FieldOp* fops; uint32_t n_fops;
  const LcTerm* lc_terms; const uint32_t* dict;   // FOP_LC: value = sum dict[coef] * Z[wire]   (dictionary in Montgomery form)
  const uint32_t* pc3; const uint32_t* pm3;   // Poseidon constants t=3 (Montgomery): C then M
  const uint32_t* pc9; const uint32_t* pm9;   // t=9
  // partial rounds in sparse form (circuit/poseidon_params.hpp: poseidon_sparse_t): per (round, lane) the triple
  // (transformed round constant, first-row entry, first-column entry [0 for lane 0]); then the (t-1)^2 basis change
  const uint32_t* ps3; const uint32_t* pf3;
  const uint32_t* ps9; const uint32_t* pf9;
  uint32_t rp3, rp9;
  // the same tables in the reduced-radix form of fp29.hpp (9 limbs of 29 bits, R' = 2^261; 12 words per element): the chains' permutations
  // run in it (poseidon_group29) unless VIMZ_DEBUG_POSEIDON_STD is set
  const uint32_t *pc3_29, *pm3_29, *ps3_29, *pf3_29, *pc9_29, *pm9_29, *ps9_29, *pf9_29;
  uint32_t poseidon29;
  uint32_t n_wires, len_z, n_priv;
};

__device__ __forceinline__ Fr fr_from_small(long long v) {  // small signed integer -> Montgomery
  Fr x = Fr::zero();
  unsigned long long m = v < 0 ? (unsigned long long)(-v) : (unsigned long long)v;
  x.v[0] = (uint32_t)m; x.v[1] = (uint32_t)(m >> 32);
  x = Fr::to_mont(x);
  return v < 0 ? Fr::neg(x) : x;
}

// status bits per row
enum : uint32_t { WIT_UNSAT = 1u, WIT_BAD_INPUT = 2u };

static __global__ void __launch_bounds__(256) k_wit_inputs(WitnessDev P, const uint32_t* __restrict__ priv, const uint32_t* __restrict__ zs,
                                                    uint32_t* __restrict__ Z, uint32_t first_row) {
  const uint32_t row = blockIdx.y;
  const size_t zrow = (size_t)row * P.n_wires;
  const uint32_t n = 1 + 2 * P.len_z + P.n_priv;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    Fr v;
    if (i == 0) v = Fr::one();
    else if (i <= P.len_z) v = Fr::to_mont(load_fe<Fr>(zs, (size_t)(first_row + row + 1) * P.len_z + (i - 1)));       // step_out = z_{k+1}
    else if (i <= 2 * P.len_z) v = Fr::to_mont(load_fe<Fr>(zs, (size_t)(first_row + row) * P.len_z + (i - 1 - P.len_z)));  // step_in = z_k
    else v = Fr::to_mont(load_fe<Fr>(priv, (size_t)row * P.n_priv + (i - 1 - 2 * P.len_z)));
    store_fe(Z, zrow + i, v);
  }
}

static __global__ void __launch_bounds__(256) k_wit_decomp(WitnessDev P, uint32_t g, const uint32_t* __restrict__ priv, uint32_t* __restrict__ Z,
                                                    uint32_t* __restrict__ status) {
  const DecompGroup D = P.decomp[g];
  const uint32_t row = blockIdx.y;
  const uint32_t total = (D.nbits - 1) * D.count;
  const uint32_t priv0 = 1 + 2 * P.len_z;
  const Fr one = Fr::one(), zero = Fr::zero();
  for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const uint32_t k = idx / D.count + 1, j = idx % D.count;
    const uint32_t* v = priv + 8 * ((size_t)row * P.n_priv + (D.src_wire - priv0 + j));
    const uint32_t bit = (v[k >> 5] >> (k & 31)) & 1u;
    store_fe(Z, (size_t)row * P.n_wires + D.bit_base + idx, bit ? one : zero);
    if (k == 1) {  // one lane per element checks the range
      uint32_t hi = 0;
      for (uint32_t b = D.nbits; b < 256; b++) hi |= (v[b >> 5] >> (b & 31)) & 1u;
      if (hi) atomicOr(&status[row], WIT_UNSAT);
    }
  }
}

constexpr int LANE_TB = 64;
constexpr int LANE_REGS = 60;

static __global__ void __launch_bounds__(LANE_TB) k_wit_lanes(WitnessDev P, uint32_t g, const uint32_t* __restrict__ priv, const uint32_t* __restrict__ zs,
                                                        uint32_t first_row, uint32_t* __restrict__ Z, uint32_t* __restrict__ status) {
  __shared__ long long regs[LANE_REGS][LANE_TB];
  const LaneGroup G = P.groups[g];
  const uint32_t row = blockIdx.y;
  const uint32_t lane = blockIdx.x * LANE_TB + threadIdx.x;
  if (lane >= G.lanes) return;
  const uint32_t tid = threadIdx.x;
  const uint32_t x = lane % G.pixels, col = (lane / G.pixels) % G.colours;
  const uint32_t priv0 = 1 + 2 * P.len_z;
  const size_t zrow = (size_t)row * P.n_wires;
  const uint32_t* prow = priv + 8 * (size_t)row * P.n_priv;
  uint32_t st = 0;
  const Fr one = Fr::one(), zero = Fr::zero();
  for (uint32_t pc = 0; pc < G.prog_len; pc++) {
    const LaneInstr I = P.instr[G.prog_off + pc];
    switch (I.op) {
      case LOP_LDB: {
        const LaneRow R = P.rows[G.row_off + I.a];
        const long px = (long)x * I.imm2 + I.imm;
        const int c = I.b == 3 ? (int)col : (int)I.b;
        long long v = 0;
        if (px >= 0 && px < (long)R.count * 10) {
          const uint32_t* e = prow + 8 * (size_t)(R.src_wire - priv0 + (uint32_t)(px / 10));
          const int byte = (int)(px % 10) * 3 + c;
          v = (e[byte >> 2] >> (8 * (byte & 3))) & 0xff;
        }
        regs[I.d][tid] = v;
        break;
      }
      case LOP_LDZ: {
        const uint32_t* e = zs + 8 * ((size_t)(first_row + row) * P.len_z + (uint32_t)I.imm);
        uint32_t hi = (e[1] >> 8) | e[2] | e[3] | e[4] | e[5] | e[6] | e[7];
        if (hi) st |= WIT_UNSAT;
        regs[I.d][tid] = (long long)(((unsigned long long)(e[1] & 0xff) << 32) | e[0]);
        break;
      }
      case LOP_LI: regs[I.d][tid] = I.imm; break;
      case LOP_ADD: regs[I.d][tid] = regs[I.a][tid] + regs[I.b][tid]; break;
      case LOP_SUB: regs[I.d][tid] = regs[I.a][tid] - regs[I.b][tid]; break;
      case LOP_MUL: regs[I.d][tid] = regs[I.a][tid] * regs[I.b][tid]; break;
      case LOP_MULI: regs[I.d][tid] = regs[I.a][tid] * I.imm; break;
      case LOP_ADDI: regs[I.d][tid] = regs[I.a][tid] + I.imm; break;
      case LOP_LEQ: {
        const long long v = regs[I.a][tid] + (1ll << I.imm) - (regs[I.b][tid] + 1);
        regs[I.d][tid] = ((v >> I.imm) & 1) ? 0 : 1;
        break;
      }
      case LOP_SEL: regs[I.d][tid] = regs[I.a][tid] ? regs[I.b][tid] : regs[I.imm][tid]; break;
      case LOP_BITS: {
        const long long v = regs[I.a][tid];
        const bool ok = v >= 0 && v < (1ll << I.imm);
        if (!ok) st |= WIT_UNSAT;
        for (int k = 1; k < I.imm; k++) {
          const bool bit = ok && ((v >> k) & 1);
          store_fe(Z, zrow + G.wire_base + (size_t)(I.imm2 + k - 1) * G.lanes + lane, bit ? one : zero);
        }
        break;
      }
      case LOP_EMIT:
        store_fe(Z, zrow + G.wire_base + (size_t)I.imm * G.lanes + lane, fr_from_small(regs[I.a][tid]));
        break;
      case LOP_LANE: regs[I.d][tid] = (long long)x + I.imm; break;
      case LOP_ANDI: regs[I.d][tid] = regs[I.a][tid] & (long long)I.imm; break;
      case LOP_SHRI: regs[I.d][tid] = regs[I.a][tid] >> I.imm; break;
      case LOP_EQ: regs[I.d][tid] = regs[I.a][tid] == regs[I.b][tid] ? 1 : 0; break;
      case LOP_LDBR: {   // byte (colour 0) of row a at pixel x*imm2 + imm + r[b]; zero outside the row
        const LaneRow R = P.rows[G.row_off + I.a];
        const long long px = (long long)x * I.imm2 + I.imm + regs[I.b][tid];
        long long v = 0;
        if (px >= 0 && px < (long long)R.count * 10) {
          const uint32_t* e = prow + 8 * (size_t)(R.src_wire - priv0 + (uint32_t)(px / 10));
          const int byte = (int)(px % 10) * 3;
          v = (e[byte >> 2] >> (8 * (byte & 3))) & 0xff;
        }
        regs[I.d][tid] = v;
        break;
      }
      default: st |= WIT_BAD_INPUT; break;
    }
  }
  if (st) atomicOr(&status[row], st);
}

// ---- Poseidon chains: 16 lanes per chain ---------------------------------------------------------------
__device__ __forceinline__ Fr shfl_fe(const Fr& v, int src_lane) {
  Fr r;
#pragma unroll
  for (int k = 0; k < 8; k++) r.v[k] = __shfl(v.v[k], src_lane);
  return r;
}

__device__ __forceinline__ Fr wit_value(const WitnessDev& P, const ValRef& r, const uint32_t* __restrict__ Zrow, const uint32_t* __restrict__ job_out_row,
                                         const uint32_t* __restrict__ priv_row = nullptr) {
  switch (r.kind) {
    case REF_WIRE:
      if (priv_row) return Fr::to_mont(load_fe<Fr>(priv_row, r.idx - (1 + 2 * P.len_z)));   // hash-only pass: private inputs only
      return load_fe<Fr>(Zrow, r.idx);
    case REF_JOB: return load_fe<Fr>(job_out_row, r.idx);
    case REF_FOP: return load_fe<Fr>(job_out_row, P.n_jobs + r.idx);
    case REF_ZIN: return load_fe<Fr>(Zrow, 1 + P.len_z + r.idx);
    default: return Fr::zero();
  }
}

// One Poseidon permutation by a 16-lane group (lane li < T owns state[li]).  T is a template parameter so the MDS
// row is held in registers and its T products are independent instructions (the chain is latency-bound: one wave per
// SIMD, so instruction-level parallelism inside the round is what shortens it).
template <int T>
__device__ __forceinline__ Fr poseidon_group(const WitnessDev& P, const HashJob& J, bool live, Fr s, bool in_const, uint32_t li, int lane_base,
                                             uint32_t* __restrict__ Zrow) {   // Zrow == nullptr: compute the hash only, write no wires
  const uint32_t* PC = T == 3 ? P.pc3 : P.pc9;
  const uint32_t* PM = T == 3 ? P.pm3 : P.pm9;
  const uint32_t* PS = T == 3 ? P.ps3 : P.ps9;
  const uint32_t* PF = T == 3 ? P.pf3 : P.pf9;
  const uint32_t rp = T == 3 ? P.rp3 : P.rp9;
  const uint32_t R = 8 + rp;
  const bool mine = li < (uint32_t)T;
  Fr Mrow[T];
#pragma unroll
  for (int j = 0; j < T; j++) Mrow[j] = mine ? load_fe<Fr>(PM, (size_t)li * T + j) : Fr::zero();
  // number of non-folded round-0 S-boxes before this lane / in total (lanes 1..T-1 with non-constant input)
  const unsigned long long nf_mask = __ballot(live && li >= 1 && mine && !in_const);
  const uint32_t grp_mask16 = (uint32_t)((nf_mask >> lane_base) & 0xffffull);
  const uint32_t nf_before = __popc(grp_mask16 & ((1u << li) - 1u));
  const uint32_t nf0 = __popc(grp_mask16);
  const uint32_t elim_slot = 3 * (nf0 + 3 * T + rp + 3 * T) + 2;  // x5 of (last round, lane 0)
  const bool bound = J.out_wire != 0;
  auto emit = [&](uint32_t index, const Fr& x2, const Fr& x4, const Fr& x5) {
    const uint32_t slot = 3 * index;
    auto wire_of = [&](uint32_t sl) { return J.wire_base + sl - ((bound && sl > elim_slot) ? 1u : 0u); };
    store_fe(Zrow, wire_of(slot), x2);
    store_fe(Zrow, wire_of(slot + 1), x4);
    if (!(bound && slot + 2 == elim_slot)) store_fe(Zrow, wire_of(slot + 2), x5);
  };
  auto full_round = [&](uint32_t r) {
    if (mine) s = Fr::add(s, load_fe<Fr>(PC, (size_t)r * T + li));
    if (mine) {
      const Fr x2 = Fr::sqr(s), x4 = Fr::sqr(x2), x5 = Fr::mul(x4, s);
      const bool folded = r == 0 && in_const;
      if (live && !folded && Zrow) {
        uint32_t index;
        if (r == 0) index = nf_before;
        else if (r < 4) index = nf0 + (r - 1) * T + li;
        else index = nf0 + 3 * T + rp + (r - 4 - rp) * T + li;
        emit(index, x2, x4, x5);
      }
      s = x5;
    }
    // MDS: new[i] = sum_j M[i][j] * s[j] — T independent products, then a short add tree
    Fr prod[T];
#pragma unroll
    for (int j = 0; j < T; j++) prod[j] = Fr::mul(Mrow[j], shfl_fe(s, lane_base + j));
#pragma unroll
    for (int stride = 1; stride < T; stride <<= 1)
#pragma unroll
      for (int j = 0; j + stride < T; j += 2 * stride) prod[j] = Fr::add(prod[j], prod[j + stride]);
    s = prod[0];
  };
  for (uint32_t r = 0; r < 4; r++) full_round(r);
  // Partial rounds in sparse form: only lane 0 goes through the S-box, and its value is the one the dense form has; the other
  // lanes live in a changed basis (undone by PF below).  Three dependent multiplication slots per round instead of twelve:
  //   [x2 on lane 0 | row_i·s_i on the others, and beside them row_0·s | col_i·s_0]  [x4]  [x4·(row_0·s) on lane 0 | x4·(col_i·s_0) on the others],
  // then the lanes' row products are added by a shuffle butterfly.
  const bool l0 = li == 0;
  Fr ct = Fr::zero(), rw = Fr::zero(), cl = Fr::zero();
  if (mine) { const size_t q = 3 * (size_t)li; ct = load_fe<Fr>(PS, q); rw = load_fe<Fr>(PS, q + 1); cl = load_fe<Fr>(PS, q + 2); }
  for (uint32_t r = 0; r < rp; r++) {
    Fr ctn = Fr::zero(), rwn = Fr::zero(), cln = Fr::zero();       // next round's constants: in flight during this round
    if (mine && r + 1 < rp) { const size_t q = 3 * ((size_t)(r + 1) * T + li); ctn = load_fe<Fr>(PS, q); rwn = load_fe<Fr>(PS, q + 1); cln = load_fe<Fr>(PS, q + 2); }
    s = Fr::add(s, ct);
    // Three dependent multiplications per round instead of four: x5 = x4·s is needed only multiplied by a constant, and s·(that constant)
    // does not wait for the S-box — lane 0: x2 | row_0·s, then x4, then x4·(row_0·s); lane i: row_i·s_i | col_i·s_0, then x4·(col_i·s_0).
    const Fr sl0 = shfl_fe(s, lane_base);                          // lane 0's state, before its S-box
    const Fr m1 = Fr::mul(l0 ? s : rw, s);                         // lane 0: x2; lane i: row_i·s_i
    const Fr u = Fr::mul(l0 ? rw : cl, sl0);                       // lane 0: row_0·s; lane i: col_i·s_0
    Fr x4 = Fr::zero();
    if (l0) { x4 = Fr::sqr(m1); if (live && Zrow) emit(nf0 + 3 * T + r, m1, x4, Fr::mul(x4, s)); }
    const Fr x4b = shfl_fe(x4, lane_base);
    const Fr m2 = Fr::mul(x4b, u);                                 // lane 0: row_0·x5; lane i: col_i·x5
    Fr p = l0 ? m2 : (mine ? m1 : Fr::zero());
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      Fr o;
#pragma unroll
      for (int k = 0; k < 8; k++) o.v[k] = __shfl_xor(p.v[k], off);
      p = Fr::add(p, o);
    }
    s = l0 ? p : (mine ? Fr::add(s, m2) : s);
    ct = ctn; rw = rwn; cl = cln;
  }
  {  // back to the standard basis on lanes 1..T-1
    Fr acc = Fr::zero();
#pragma unroll
    for (int j = 0; j < T - 1; j++) {
      const Fr sj = shfl_fe(s, lane_base + 1 + j);
      if (mine && !l0) acc = Fr::add(acc, Fr::mul(load_fe<Fr>(PF, (size_t)(li - 1) * (T - 1) + j), sj));
    }
    if (mine && !l0) s = acc;
  }
  for (uint32_t r = 4 + rp; r < R; r++) full_round(r);
  return s;
}

// ---- the same permutation in the reduced-radix arithmetic of the curve code (fp29.hpp) ----------------------------------------------
// A chain is ONE dependent sequence of multiplications — 17 permutations of 65 rounds for a row hash — on one wave per SIMD, so what
// counts is the latency of a multiplication, not its instruction count.  The saturated CIOS of fp.hpp is one carry chain of 128
// dependent multiply-adds (measured here: 7.2 µs per partial round, 8.0 ms for a row-hash chain); the 29-bit-limb product accumulates
// its 17 columns independently and reduces lazily.  Values stay below 2 p between rounds (weak_reduce29), products take operands whose
// bounds multiply to at most 64 (fp29.hpp), sums of at most nine products stay below 14 p.
typedef Fp29<BnFr> F29;
constexpr int F29_STRIDE = 12;      // words per table element (three 16-byte loads)
__device__ __forceinline__ F29 load29(const uint32_t* __restrict__ base, size_t idx) {
  const uint4* q = reinterpret_cast<const uint4*>(base + (size_t)F29_STRIDE * idx);
  const uint4 a = q[0], b = q[1], c = q[2];
  F29 r; r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w; r.v[8] = c.x;
  return r;
}
__device__ __forceinline__ F29 shfl29(const F29& v, int src_lane) {
  F29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.v[k] = __shfl(v.v[k], src_lane);
  return r;
}
// (a sum of up to nine products, below 14 p, back below 2 p: Fp29::weak_reduce — 1.65 p for BN254 Fr, KNUM = 169)
static_assert(F29::KNUM == 169u, "the bounds in poseidon_group29 are written for BN254 Fr's quotient estimate");
__device__ __forceinline__ F29 weak_reduce29(const F29& a) { return a.weak_reduce(); }
// a wire of the S-boxes as it leaves the chain kernel: the 256-bit integer x·2^261 mod p (+ p at most once); k_wit_chain_wires_std turns
// the job's wires into the standard form afterwards, all of them side by side (a conversion per wire inside the chain would sit in its way)
__device__ __forceinline__ void store29_raw(uint32_t* __restrict__ Zrow, uint32_t wire, const F29& x) {
  uint32_t w[8]; x.unpack(w);
  uint4* q = reinterpret_cast<uint4*>(Zrow + 8 * (size_t)wire);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]); q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

template <int T>
__device__ __forceinline__ F29 poseidon_group29(const WitnessDev& P, const HashJob& J, bool live, F29 s, bool in_const, uint32_t li, int lane_base,
                                                uint32_t* __restrict__ Zrow) {   // as poseidon_group; s below 2 p in, below 2 p out
  const uint32_t* PC = T == 3 ? P.pc3_29 : P.pc9_29;
  const uint32_t* PM = T == 3 ? P.pm3_29 : P.pm9_29;
  const uint32_t* PS = T == 3 ? P.ps3_29 : P.ps9_29;
  const uint32_t* PF = T == 3 ? P.pf3_29 : P.pf9_29;
  const uint32_t rp = T == 3 ? P.rp3 : P.rp9;
  const uint32_t R = 8 + rp;
  const bool mine = li < (uint32_t)T;
  F29 Mrow[T];
#pragma unroll
  for (int j = 0; j < T; j++) Mrow[j] = mine ? load29(PM, (size_t)li * T + j) : F29::zero();
  const unsigned long long nf_mask = __ballot(live && li >= 1 && mine && !in_const);
  const uint32_t grp_mask16 = (uint32_t)((nf_mask >> lane_base) & 0xffffull);
  const uint32_t nf_before = __popc(grp_mask16 & ((1u << li) - 1u));
  const uint32_t nf0 = __popc(grp_mask16);
  const uint32_t elim_slot = 3 * (nf0 + 3 * T + rp + 3 * T) + 2;
  const bool bound = J.out_wire != 0;
  auto emit = [&](uint32_t index, const F29& x2, const F29& x4, const F29& x5) {      // (each below 1.5 p < 2^256)
    const uint32_t slot = 3 * index;
    auto wire_of = [&](uint32_t sl) { return J.wire_base + sl - ((bound && sl > elim_slot) ? 1u : 0u); };
    store29_raw(Zrow, wire_of(slot), x2);
    store29_raw(Zrow, wire_of(slot + 1), x4);
    if (!(bound && slot + 2 == elim_slot)) store29_raw(Zrow, wire_of(slot + 2), x5);
  };
  auto full_round = [&](uint32_t r) {
    if (mine) {
      s = F29::add(s, load29(PC, (size_t)r * T + li));                         // < 3
      const F29 x2 = F29::sqr(s), x4 = F29::sqr(x2), x5 = F29::mul(x4, s);     // 9, 2.25, 4.5 -> each < 1.5
      const bool folded = r == 0 && in_const;
      if (live && !folded && Zrow) {
        uint32_t index;
        if (r == 0) index = nf_before;
        else if (r < 4) index = nf0 + (r - 1) * T + li;
        else index = nf0 + 3 * T + rp + (r - 4 - rp) * T + li;
        emit(index, x2, x4, x5);
      }
      s = x5;
    }
    F29 prod[T];
#pragma unroll
    for (int j = 0; j < T; j++) prod[j] = F29::mul(Mrow[j], shfl29(s, lane_base + j));      // 1 · 1.5
#pragma unroll
    for (int stride = 1; stride < T; stride <<= 1)
#pragma unroll
      for (int j = 0; j + stride < T; j += 2 * stride) prod[j] = F29::add(prod[j], prod[j + stride]);
    s = weak_reduce29(prod[0]);                                                // T · 1.5 <= 13.5 -> < 1.65
  };
  for (uint32_t r = 0; r < 4; r++) full_round(r);
  const bool l0 = li == 0;
  F29 ct = F29::zero(), rw = F29::zero(), cl = F29::zero();
  if (mine) { const size_t q = 3 * (size_t)li; ct = load29(PS, q); rw = load29(PS, q + 1); cl = load29(PS, q + 2); }
  for (uint32_t r = 0; r < rp; r++) {
    F29 ctn = F29::zero(), rwn = F29::zero(), cln = F29::zero();
    if (mine && r + 1 < rp) { const size_t q = 3 * ((size_t)(r + 1) * T + li); ctn = load29(PS, q); rwn = load29(PS, q + 1); cln = load29(PS, q + 2); }
    s = F29::add(s, ct);                                           // < 3
    const F29 sl0 = shfl29(s, lane_base);
    const F29 m1 = F29::mul(l0 ? s : rw, s);                       // lane 0: x2 (3·3); lane i: row_i·s_i
    const F29 u = F29::mul(l0 ? rw : cl, sl0);                     // lane 0: row_0·s; lane i: col_i·s_0
    F29 x4 = F29::zero();
    if (l0) { x4 = F29::sqr(m1); if (live && Zrow) emit(nf0 + 3 * T + r, m1, x4, F29::mul(x4, s)); }
    const F29 x4b = shfl29(x4, lane_base);
    const F29 m2 = F29::mul(x4b, u);                               // lane 0: row_0·x5; lane i: col_i·x5
    F29 p = l0 ? m2 : (mine ? m1 : F29::zero());
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      F29 o;
#pragma unroll
      for (int k = 0; k < 9; k++) o.v[k] = __shfl_xor(p.v[k], off);
      p = F29::add(p, o);                                          // <= T terms below 1.5
    }
    s = weak_reduce29(l0 ? p : (mine ? F29::add(s, m2) : s));      // 13.5 | 3 + 1.5 -> < 1.65
    ct = ctn; rw = rwn; cl = cln;
  }
  {
    F29 acc = F29::zero();
#pragma unroll
    for (int j = 0; j < T - 1; j++) {
      const F29 sj = shfl29(s, lane_base + 1 + j);
      if (mine && !l0) acc = F29::add(acc, F29::mul(load29(PF, (size_t)(li - 1) * (T - 1) + j), sj));      // <= 8 terms below 1.5
    }
    if (mine && !l0) s = weak_reduce29(acc);
  }
  for (uint32_t r = 4 + rp; r < R; r++) full_round(r);
  return s;
}

// grid.x covers chains of `phase` in groups of 16 lanes (4 chains per wave); grid.y = row.
// job_out: [row][n_jobs + n_fops] Montgomery.
// priv_rows != nullptr selects the hash-only pass: inputs come from the canonical private-input rows, no wire is written
// (used once per fold call over ALL rows to get the row hashes the IVC state chain needs).
static __global__ void __launch_bounds__(64) k_wit_chains(WitnessDev P, uint32_t phase, uint32_t* __restrict__ Z, uint32_t* __restrict__ job_out,
                                                   const uint32_t* __restrict__ priv_rows) {
  const uint32_t row = blockIdx.y;
  const uint32_t sub = threadIdx.x >> 4, li = threadIdx.x & 15;
  const int lane_base = (int)(threadIdx.x & ~15u);
  uint32_t want = blockIdx.x * 4 + sub, seen = 0, cid = 0xffffffffu;
  for (uint32_t c = 0; c < P.n_chains; c++) if (P.chains[c].phase == phase) { if (seen == want) { cid = c; break; } seen++; }
  const bool active_chain = cid != 0xffffffffu;
  uint32_t* Zrow = priv_rows ? nullptr : Z + 8 * (size_t)row * P.n_wires;
  const uint32_t* prow = priv_rows ? priv_rows + 8 * (size_t)row * P.n_priv : nullptr;
  uint32_t* jrow = job_out + 8 * (size_t)row * (P.n_jobs + P.n_fops);
  const uint32_t njobs = active_chain ? P.chains[cid].job_cnt : 0;
  // every lane of the wave runs as many iterations as the longest chain in it, so the shuffles stay convergent
  uint32_t max_jobs = njobs;
  for (int off = 32; off >= 16; off >>= 1) max_jobs = max(max_jobs, (uint32_t)__shfl_xor((int)max_jobs, off));
  Fr prev_out = Fr::zero();  // output of the previous job of this chain, kept in registers (no memory round trip)
  if (P.poseidon29) {        // (wave-uniform)
    F29 prev29 = F29::zero();
    for (uint32_t k = 0; k < max_jobs; k++) {
      const bool live = k < njobs;
      HashJob J;
      if (live) J = P.jobs[P.chains[cid].job_off + k]; else { J.t = 3; J.wire_base = 0; J.out_wire = 0; }
      const uint32_t t = J.t;
      F29 s = F29::zero();
      bool in_const = true;
      if (live && li >= 1 && li < t) {
        const ValRef ref = J.in[li - 1];
        in_const = ref.kind == REF_CONST_ZERO;
        if (ref.kind == REF_JOB && k > 0 && ref.idx == P.chains[cid].job_off + k - 1) s = prev29;
        else s = F29::from_std(wit_value(P, ref, Zrow, jrow, prow));
      }
      const bool any9 = __any(live && t == 9), any3 = __any(live && t != 9);
      F29 out = F29::zero();
      if (any9) { F29 o = poseidon_group29<9>(P, J, live && t == 9, s, in_const, li, lane_base, Zrow); if (t == 9) out = o; }
      if (any3) { F29 o = poseidon_group29<3>(P, J, live && t == 3, s, in_const, li, lane_base, Zrow); if (t != 9) out = o; }
      prev29 = shfl29(out, lane_base);
      if (live && li == 0) {
        const Fr o = out.to_std();
        store_fe(jrow, P.chains[cid].job_off + k, o);
        if (J.out_wire && Zrow) store_fe(Zrow, J.out_wire, o);
      }
    }
    return;
  }
  for (uint32_t k = 0; k < max_jobs; k++) {
    const bool live = k < njobs;
    HashJob J;
    if (live) J = P.jobs[P.chains[cid].job_off + k]; else { J.t = 3; J.wire_base = 0; J.out_wire = 0; }
    const uint32_t t = J.t;
    // initial state: lane 0 -> 0, lane i -> input i-1
    Fr s = Fr::zero();
    bool in_const = true;
    if (live && li >= 1 && li < t) {
      const ValRef ref = J.in[li - 1];
      in_const = ref.kind == REF_CONST_ZERO;
      if (ref.kind == REF_JOB && k > 0 && ref.idx == P.chains[cid].job_off + k - 1) s = prev_out;
      else s = wit_value(P, ref, Zrow, jrow, prow);
    }
    // the four chains of a wave may mix widths: run both variants under wave-uniform votes so shuffles stay convergent
    const bool any9 = __any(live && t == 9), any3 = __any(live && t != 9);      // (idle 16-lane groups and finished chains run neither: a wave of row-hash chains used to run the narrow variant beside the wide one)
    Fr out = Fr::zero();
    if (any9) { Fr o = poseidon_group<9>(P, J, live && t == 9, s, in_const, li, lane_base, Zrow); if (t == 9) out = o; }
    if (any3) { Fr o = poseidon_group<3>(P, J, live && t == 3, s, in_const, li, lane_base, Zrow); if (t != 9) out = o; }
    prev_out = shfl_fe(out, lane_base);
    if (live && li == 0) {
      store_fe(jrow, P.chains[cid].job_off + k, out);
      if (J.out_wire && Zrow) store_fe(Zrow, J.out_wire, out);
    }
  }
}

// After k_wit_chains in the reduced-radix form: the S-box wires of the jobs of `phase` hold x·2^261 mod p (store29_raw); x·2^256 is that
// times 2^-5 — one Montgomery product with the plain integer 2^251 mod p.  grid (jobs, rows); job_off as for k_wit_scatter.
static __global__ void __launch_bounds__(128) k_wit_chain_wires_std(WitnessDev P, const uint32_t* __restrict__ job_off, uint32_t phase, uint32_t* __restrict__ Z) {
  const uint32_t j = blockIdx.x, row = blockIdx.y;
  uint32_t ph = 0xffffffffu;
  for (uint32_t c = 0; c < P.n_chains; c++) if (j >= P.chains[c].job_off && j < P.chains[c].job_off + P.chains[c].job_cnt) { ph = P.chains[c].phase; break; }
  if (ph != phase) return;
  constexpr U256 C251 = ct_pow2_mod(251, BnFr::MOD);
  Fr c; for (int i = 0; i < 8; i++) c.v[i] = C251.w[i];
  const uint32_t cnt = job_off[j + 1] - job_off[j];
  uint32_t* dst = Z + 8 * ((size_t)row * P.n_wires + P.jobs[j].wire_base);
  for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) store_fe(dst, i, Fr::mul(load_fe<Fr>(dst, i), c));
}

// Head batch of a fold call (prover_internal.hpp: fold_head_batch): its Poseidon jobs were evaluated on the host — a sequential
// chain of 17 permutations is 1.1 ms on a CPU core and 10 ms on one wave — and arrive as one staging row per witness row
// (job j's wires at [job_off[j], job_off[j+1])) plus the job outputs.  grid (jobs, rows): copy them to their wires.
static __global__ void __launch_bounds__(128) k_wit_scatter(WitnessDev P, const uint32_t* __restrict__ job_off, const uint32_t* __restrict__ stage, uint32_t stage_row,
                                                             const uint32_t* __restrict__ jobvals, uint32_t* __restrict__ Z, uint32_t* __restrict__ job_out) {
  const uint32_t j = blockIdx.x, row = blockIdx.y;
  const HashJob& J = P.jobs[j];
  const uint32_t lo = job_off[j], cnt = job_off[j + 1] - lo;
  const uint32_t* src = stage + 8 * ((size_t)row * stage_row + lo);
  uint32_t* dst = Z + 8 * ((size_t)row * P.n_wires + J.wire_base);
  for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) store_fe(dst, i, load_fe<Fr>(src, i));
  if (threadIdx.x == 0) {
    const size_t jr = (size_t)row * (P.n_jobs + P.n_fops) + j;
    const Fr out = load_fe<Fr>(jobvals, jr);
    store_fe(job_out, jr, out);
    if (J.out_wire) store_fe(Z, (size_t)row * P.n_wires + J.out_wire, out);
  }
}

// Field ops of one stage (early = before the phase-2 chains, 0 = after every chain), FOP_LC excepted (k_wit_fops_lc).
static __global__ void __launch_bounds__(64) k_wit_fops(WitnessDev P, uint32_t* __restrict__ Z, uint32_t* __restrict__ job_out, uint32_t rows, uint32_t early) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  uint32_t* Zrow = Z + 8 * (size_t)row * P.n_wires;
  uint32_t* jrow = job_out + 8 * (size_t)row * (P.n_jobs + P.n_fops);
  for (uint32_t f = 0; f < P.n_fops; f++) {
    const FieldOp F = P.fops[f];
    if (F.early != early) continue;
    if (F.op == FOP_ISZERO) {
      const Fr in = wit_value(P, F.a, Zrow, jrow);
      const Fr inv = Fr::pow_pm2(in);                 // 0 -> 0
      const Fr out = Fr::sub(Fr::one(), Fr::mul(in, inv));
      store_fe(Zrow, F.wire, inv); store_fe(Zrow, F.wire + 1, out);
      store_fe(jrow, P.n_jobs + f, out);
    } else if (F.op == FOP_MUX) {
      const Fr s = wit_value(P, F.a, Zrow, jrow), c0 = wit_value(P, F.b, Zrow, jrow), c1 = wit_value(P, F.c, Zrow, jrow);
      const Fr prod = Fr::mul(Fr::sub(c1, c0), s), out = Fr::add(prod, c0);
      store_fe(Zrow, F.wire, F.bound ? out : prod);
      store_fe(jrow, P.n_jobs + f, out);
    }
  }
}

// FOP_LC: the value of a stored linear combination of wires (crop: a packed element of the cropped row = 12.8 k terms, all
// but ten of them zero).  One wave per (field op, row): lanes stride over the terms, zero wires skip the multiply, shuffle tree.
static __global__ void __launch_bounds__(64) k_wit_fops_lc(WitnessDev P, const uint32_t* __restrict__ Z, uint32_t* __restrict__ job_out, uint32_t early) {
  const uint32_t f = blockIdx.x, row = blockIdx.y, lane = threadIdx.x;
  const FieldOp F = P.fops[f];
  if (F.op != FOP_LC || F.early != early) return;
  const size_t zrow = (size_t)row * P.n_wires;
  Fr acc = Fr::zero();
  for (uint32_t k = lane; k < F.b.idx; k += 64) {
    const LcTerm T = P.lc_terms[F.a.idx + k];
    const Fr v = load_fe<Fr>(Z, zrow + T.wire);
    if (v.is_zero()) continue;
    acc = Fr::add(acc, Fr::mul(load_fe<Fr>(P.dict, 2 * (size_t)T.coef), v));      // (the dictionary holds two forms per coefficient: r1cs_ops.hpp)
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    Fr o;
#pragma unroll
    for (int w = 0; w < 8; w++) o.v[w] = __shfl_xor(acc.v[w], off);
    acc = Fr::add(acc, o);
  }
  if (lane == 0) store_fe(job_out, (size_t)row * (P.n_jobs + P.n_fops) + P.n_jobs + f, acc);
}

}  // namespace vz
