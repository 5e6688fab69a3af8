// Witness-program format: what the step-circuit builder emits next to the R1CS, and what the GPU
// witness kernels (witness.hpp) execute.  It plays the role of the `.wasm` / C++ witness generator
// that circom emits and nova-scotia spawns once per step: data describing how
// to compute every wire of the step circuit from (step_in, private inputs).
//
// Wire order (iden3 / nova-scotia convention):
//   z = [ 1 | step_out (len_z) | step_in (len_z) | private inputs (n_priv) | intermediates ]
//
// Four kinds of work, each a flat POD table so the same bytes drive the HIP kernels and can be handed
// to the CPU oracle's independent executor in the tests:
//   DecompGroup  bit decomposition (Num2Bits(240)) of a run of packed input elements; bit-major layout.
//   LaneGroup    a short integer program run once per lane (lane = pixel x colour): pixel arithmetic,
//                comparator bits, multiplexers.  Wires of the group are laid out slot-major
//                (wire = base + slot * lanes + lane) so a wave writes contiguous memory.
//   HashJob      one circomlib Poseidon permutation; jobs are grouped in chains (sequential), chains are
//                independent.  Phase A chains do not depend on the IVC state (row hashes), phase B do.
//   FieldOp      the few full-width operations outside Poseidon (IsZero, Mux1 on hashes).
#pragma once
#include <stdint.h>

namespace vz {

enum : uint32_t { REF_CONST_ZERO = 0, REF_WIRE = 1, REF_JOB = 2, REF_FOP = 3, REF_ZIN = 4 };
struct ValRef { uint32_t kind; uint32_t idx; };  // how a full-width value is obtained at witness time

struct DecompGroup {
  uint32_t src_wire;   // first packed input wire
  uint32_t count;      // N elements
  uint32_t bit_base;   // wire of (bit k>=1, element j) = bit_base + (k-1)*N + j ; bit 0 is not a wire
  uint32_t nbits;      // 240
};

// ---- integer lane machine -------------------------------------------------------------------------
enum : uint8_t {
  LOP_LDB = 1,   // r[d] = byte: row a, colour b (3 = the lane's colour), pixel = x*imm2 + imm (0 if out of range)
  LOP_LDZ,       // r[d] = step_in[imm] as a small integer (error if >= 2^40)
  LOP_LI,        // r[d] = imm
  LOP_ADD, LOP_SUB, LOP_MUL,   // r[d] = r[a] op r[b]
  LOP_MULI, LOP_ADDI,          // r[d] = r[a] op imm
  LOP_LEQ,       // r[d] = LessEqThan(imm)(r[a], r[b])  (value only; bits come from LOP_BITS)
  LOP_SEL,       // r[d] = r[a] ? r[b] : r[imm]
  LOP_BITS,      // bits 1..imm-1 of r[a] -> slots imm2.. ; r[a] must be in [0, 2^imm) else the row is UNSAT
  LOP_EMIT,      // wire slot imm = r[a] as a field element
  LOP_ROWSEL,    // (reserved)
  LOP_LANE,      // r[d] = x + imm                      (x = lane % pixels)
  LOP_ANDI, LOP_SHRI,   // r[d] = r[a] & imm ; r[d] = r[a] >> imm
  LOP_EQ,        // r[d] = (r[a] == r[b])
  LOP_LDBR       // r[d] = byte: row a, colour 0, pixel = x*imm2 + imm + r[b]   (register-relative; 0 if out of range)
};
struct LaneInstr { uint8_t op, d, a, b; int32_t imm, imm2; };

struct LaneRow { uint32_t src_wire; uint32_t count; };  // a packed input row the lanes can read bytes from
struct LaneGroup {
  uint32_t lanes;        // L
  uint32_t pixels;       // pixels per output row (lane % pixels = x); colour = (lane / pixels) % colours
  uint32_t colours;      // 3 or 1
  uint32_t wire_base;    // wire(slot, lane) = wire_base + slot * lanes + lane
  uint32_t slots;        // wires per lane
  uint32_t prog_off, prog_len;   // into the LaneInstr table
  uint32_t row_off, row_cnt;     // into the LaneRow table
  uint32_t rows_out;     // number of output rows sharing the program (resize); lane / (pixels*colours) = output row
  uint32_t row_stride_a; // LDB row index = a + out_row * row_stride_a  (resize: input row i -> i + out_row)
  uint32_t pad;
};

// ---- Poseidon jobs ----------------------------------------------------------------------------------
constexpr int POSEIDON_MAX_T = 9;
struct HashJob {
  uint32_t t;                        // state width (inputs = t-1)
  uint32_t wire_base;                // S-box wires x2,x4,x5 in (round, lane) order, folded / bound ones skipped
  uint32_t out_wire;                 // 0 = output is not a wire; else the public-output wire the hash is bound to
  uint32_t chain;                    // chain id
  ValRef in[POSEIDON_MAX_T - 1];
};
struct Chain { uint32_t job_off, job_cnt, phase, pad; };  // phase 0 = A (row data only), 1 = B (needs the hashed state in step_in),
                                                          // 2 = after the early field ops, before B (may use only the predictable part of step_in)

enum : uint32_t { FOP_ISZERO = 1, FOP_MUX = 2, FOP_LC = 3 };
struct LcTerm { uint32_t wire; uint32_t coef; };   // coef = index into the R1CS coefficient dictionary
struct FieldOp {
  uint32_t op;
  uint32_t wire;        // ISZERO: inv wire (out wire = wire+1) ; MUX: product/out wire
  uint32_t bound;       // MUX: 1 if `wire` is the bound public output (holds the mux result), 0 if it holds (c1-c0)*s
  uint32_t early;       // 1: evaluated before the phase-B hash chains (its value feeds them), 0: after all chains
  ValRef a, b, c;       // ISZERO: a = in ; MUX: a = s, b = c0, c = c1 ; LC: value = sum of lc_terms[a.idx .. a.idx + b.idx)
};

struct ZOut { ValRef ref; int64_t add; };  // step_out[i] = value(ref) + add

struct ProgramHeader {
  uint32_t n_wires, len_z, n_priv, n_constraints;
  uint32_t n_decomp, n_lane_groups, n_lane_instr, n_lane_rows;
  uint32_t n_jobs, n_chains, n_fops, n_chains_a;
};

}  // namespace vz
