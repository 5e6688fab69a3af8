// Internals of the folding prover shared by the two chains that drive it:
//   prover.hip  the NIFS accumulator over the step circuit's own instances (row segments fold independently and merge)
//   ivc.hip     Nova IVC: the augmented circuits on the BN254/Grumpkin cycle (RecursiveSNARK::prove_step in full)
// Both use the same batch producer (witness kernels, per-row SpMV, witness-commitment MSM on a low-priority stream).
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstring>
#include <vector>
#include <algorithm>

#include "internal.hpp"
#include "circuit_handle.hpp"
#include "r1cs_ops.hpp"
#include "witness.hpp"

using namespace vz;
typedef cb::Fe Fe;                 // host Montgomery Fr
typedef Fp<BnFq> Fq;
typedef Affine<Fq> G1Aff;
typedef XYZZ<Fq> G1;

#define P_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) return vz_fail(ctx, VIMZ_ERR_HIP, #x, _e); } while (0)

namespace {

template <class T>
hipError_t upload(const std::vector<T>& v, const T** out) {
  *out = nullptr;
  if (v.empty()) return hipSuccess;
  void* d; hipError_t e = hipMalloc(&d, v.size() * sizeof(T));
  if (e != hipSuccess) return e;
  e = hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  *out = (const T*)d;
  return e;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// scalar (canonical 128-bit, little-endian words) * affine point
G1 scalar_mul(const G1Aff& p, const uint32_t* k, int bits) {
  G1 acc = G1::identity();
  for (int i = bits - 1; i >= 0; i--) {
    acc = dbl(acc);
    if ((k[i >> 5] >> (i & 31)) & 1) add_mixed(acc, p);
  }
  return acc;
}

}  // namespace

enum { PH_WITNESS = 0, PH_ZCHAIN, PH_SPMV, PH_MSM_W, PH_CROSS, PH_MSM_T, PH_RO, PH_FOLD, PH_HOST_EC, PH_COUNT };

struct vimz_prover {
  vimz_ctx* ctx = nullptr;
  const vimz_circuit* circuit = nullptr;
  const vimz_bases* ck = nullptr;
  uint32_t n_wires = 0, n_c = 0, len_z = 0, n_priv = 0, n_aux = 0, n_jobs = 0, n_fops = 0;
  // Layout.  Accumulator mode: public IO X = wires [1, 1+2 len_z), committed witness = wires [c0 = 1+2 len_z, n_wires).
  // IVC mode (augmented circuit, aug/augmented.hpp): the step circuit's wires/rows come first ([0, step_wires) / [0, step_c)),
  // the verifier circuit's after them, public IO = the last two wires, committed witness = wires [c0 = 1, n_wires - 2).
  // The batch producer only touches the step part: wires [c0, step_wires) and rows [0, step_c).
  bool ivc = false;
  uint32_t c0 = 0, step_wires = 0, step_c = 0;
  const uint32_t* long_items_aug = nullptr; uint32_t n_long_aug = 0, n_med_aug = 0;   // long (matrix,row) items of the verifier rows
  size_t max_batch = 0;
  // device: shape
  CsrDev A{}, B{}, C{};
  const uint32_t* dict = nullptr;
  const uint32_t* long_items = nullptr; uint32_t n_long = 0, n_med = 0;   // the first n_med items have <= SPMV_MED terms
  WitnessDev wd{};
  std::vector<void*> owned;   // every device allocation, for cleanup
  // device: batch buffers
  uint32_t *priv_d = nullptr, *zs_d = nullptr, *Z_d = nullptr, *job_out_d = nullptr, *status_d = nullptr;
  // device: running instance and per-step scratch
  uint32_t *Zrun = nullptr, *E = nullptr, *AZ = nullptr, *BZ = nullptr, *CZ = nullptr, *T = nullptr, *az2 = nullptr, *bz2 = nullptr, *cz2 = nullptr;
  uint32_t* bad_d = nullptr;
  // second stream: everything of a step that does not depend on the running instance (the fresh instance's
  // (A,B,C)·z and its witness commitment) is issued for the whole batch up front and overlaps the sequential chain
  hipStream_t sB = nullptr;
  struct BatchBuf {                       // double-buffered: batch k+1 is produced while batch k is folded
    uint32_t *Z = nullptr, *job_out = nullptr, *status = nullptr, *az = nullptr, *bz = nullptr, *cz = nullptr;
    void* pin = nullptr;                  // [batch][MSM_MAX_WINDOWS] window sums of the witness commitments (pinned)
    uint32_t* status_host = nullptr;      // pinned
    std::vector<hipEvent_t> ev;           // per row: fresh-instance work done
    hipEvent_t wit_done = nullptr;
  } buf[2];
  MsmWorkspace wsB;
  MsmPlan planB{};
  // per fold call: all private inputs, all IVC states and all row hashes resident
  uint32_t *priv_all_d = nullptr, *zs_all_d = nullptr, *job_all_d = nullptr;
  size_t cap_priv_all = 0, cap_zs_all = 0, cap_job_all = 0;
  std::vector<void*> retired;      // outgrown per-call buffers (see grow())
  // host: running instance
  G1Aff comm_W{}, comm_E{};
  Fe u = Fe::zero();
  std::vector<Fe> z_cur, z0;      // IVC state (Montgomery)
  Fe ro = Fe::zero(), zdigest = Fe::zero();
  uint64_t steps = 0;
  double phase_s[PH_COUNT] = {};
  uint64_t phase_n[PH_COUNT] = {};
  std::vector<uint32_t> last_status;
};

namespace {

Fe fe_from_canon(const uint64_t* c) { Fe x; memcpy(x.v, c, 32); return Fe::to_mont(x); }
void fe_to_canon(const Fe& m, uint64_t* out) { Fe c = Fe::from_mont(m); memcpy(out, c.v, 32); }

// numeric value of a reference on the host, for the IVC state chain (phase-B jobs / field ops only)
struct HostEval {
  const vimz_prover* P; const cb::Builder* b;
  const uint64_t* priv;            // canonical private inputs of this row
  const Fe* job_a;                 // phase-A job outputs of this row (Montgomery), indexed by job
  std::vector<Fe> job_b, fop;      // computed here
  const Fe* zin;
  Fe value(const ValRef& r) const {
    switch (r.kind) {
      case REF_WIRE:
        if (r.idx > b->len_z && r.idx <= 2 * b->len_z) return zin[r.idx - 1 - b->len_z];        // a step_in wire
        if (r.idx >= 1 + 2 * b->len_z && r.idx < 1 + 2 * b->len_z + b->n_priv) return fe_from_canon(priv + 4 * (size_t)(r.idx - (1 + 2 * b->len_z)));
        return Fe::zero();  // the builder never references other wires from phase-B inputs
      case REF_JOB: return b->chains[b->jobs[r.idx].chain].phase != 1 ? job_a[r.idx] : job_b[r.idx];   // phases 0 and 2: computed on the GPU ahead of the chain
      case REF_FOP: return b->fops[r.idx].op == FOP_LC ? job_a[b->jobs.size() + r.idx] : fop[r.idx];
      case REF_ZIN: return zin[r.idx];
      default: return Fe::zero();
    }
  }
};

}  // namespace


// (A,B,C)·z.  part 0: every row; 1: the step circuit's rows [0, step_c); 2: the verifier circuit's rows [step_c, n_c).
static void launch_spmv(vimz_prover* p, hipStream_t s, const uint32_t* z, uint32_t* az, uint32_t* bz, uint32_t* cz, int part = 0) {
  if (part == 0 || part == 1) {
    const size_t rows = p->step_c;
    hipLaunchKernelGGL(k_spmv3<Fr>, dim3(stream_grid(3 * rows)), dim3(256), 0, s, p->A, p->B, p->C, p->dict, rows, z, az, bz, cz);
    if (p->n_long) {
      hipLaunchKernelGGL(k_spmv_long<Fr>, dim3(spmv_long_blocks(p->n_long, p->n_med)), dim3(256), 0, s, p->A, p->B, p->C, p->dict, p->long_items, p->n_long, p->n_med, z, az, bz, cz);
    }
  }
  if ((part == 0 || part == 2) && p->n_c > p->step_c) {
    const size_t rows = p->n_c - p->step_c, o = p->step_c;
    CsrDev A2{p->A.row_ptr + o, p->A.col, p->A.coef}, B2{p->B.row_ptr + o, p->B.col, p->B.coef}, C2{p->C.row_ptr + o, p->C.col, p->C.coef};
    hipLaunchKernelGGL(k_spmv3<Fr>, dim3(stream_grid(3 * rows)), dim3(256), 0, s, A2, B2, C2, p->dict, rows, z, az + 8 * o, bz + 8 * o, cz + 8 * o);
    if (p->n_long_aug) {
      hipLaunchKernelGGL(k_spmv_long<Fr>, dim3(spmv_long_blocks(p->n_long_aug, p->n_med_aug)), dim3(256), 0, s, p->A, p->B, p->C, p->dict, p->long_items_aug, p->n_long_aug, p->n_med_aug, z, az, bz, cz);
    }
  }
}

// Host IVC-state chain for `rows` rows: zs[(r+1)] from zs[r] and the row hashes (phase-A job outputs) of row r.
static void host_state_chain(const vimz_prover* p, const uint64_t* inputs, size_t rows, const Fe* jobA, size_t jstride, std::vector<Fe>& zs) {
  const cb::Builder& b = p->circuit->build->b;
  HostEval ev; ev.P = p; ev.b = &b;
  for (size_t r = 0; r < rows; r++) {
    ev.priv = inputs + 4 * r * p->n_priv; ev.job_a = jobA + r * jstride; ev.zin = zs.data() + r * p->len_z;
    ev.job_b.assign(p->n_jobs, Fe::zero()); ev.fop.assign(p->n_fops, Fe::zero());
    for (auto& c : b.chains) {
      if (c.phase != 1) continue;
      for (uint32_t k = 0; k < c.job_cnt; k++) {
        const HashJob& J = b.jobs[c.job_off + k];
        Fe in[POSEIDON_MAX_T];
        for (uint32_t i = 0; i + 1 < J.t; i++) in[i] = ev.value(J.in[i]);
        ev.job_b[c.job_off + k] = cb::poseidon_hash(in, (int)J.t - 1);
      }
    }
    for (uint32_t f = 0; f < p->n_fops; f++) {
      const FieldOp& F = b.fops[f];
      if (F.op == FOP_ISZERO) { Fe in = ev.value(F.a); ev.fop[f] = in.is_zero() ? Fe::one() : Fe::zero(); }
      else if (F.op == FOP_MUX) { Fe sv = ev.value(F.a), c0 = ev.value(F.b), c1 = ev.value(F.c); ev.fop[f] = Fe::add(Fe::mul(Fe::sub(c1, c0), sv), c0); }
    }
    Fe* zn = zs.data() + (r + 1) * p->len_z;
    for (uint32_t i = 0; i < p->len_z; i++) zn[i] = Fe::add(ev.value(b.zout[i].ref), cb::fe_from_i64(b.zout[i].add));
  }
}

// Per-call device buffers that scale with the number of rows.  hipFree synchronises the whole device: with other provers folding
// on the same GPU that is a wait for everything they have queued (measured: one fold_prepare in fifteen took 320 ms instead of
// 30 ms, a fifth of the bench runs lost 30 %).  So an outgrown buffer is retired (freed with the prover), capacity at least
// doubles, and the first allocation is sized for 1024 rows.
static hipError_t grow(std::vector<void*>& retired, uint32_t** d, size_t* cap, size_t bytes, size_t floor_bytes) {
  if (bytes <= *cap) return hipSuccess;
  if (*d) retired.push_back(*d);
  *d = nullptr;
  const size_t want = std::max(std::max(bytes, floor_bytes), 2 * *cap);
  *cap = 0;
  hipError_t e = hipMalloc((void**)d, want);
  if (e == hipSuccess) *cap = want;
  return e;
}

int vz_prover_create_layout(vimz_ctx* ctx, const vimz_circuit* circuit, const vimz_bases* ck, size_t max_batch, int ivc, uint32_t step_wires, uint32_t step_c, vimz_prover** out);

// One call of vimz_prover_fold / vimz_ivc_fold: the inputs, the IVC state chain and the batch schedule.
struct FoldJob {
  bool batch0_started = false;     // fold_prepare already ran the state-independent part of batch 0's witness (see there)
  const uint64_t* step_inputs = nullptr;   // nsteps x n_priv canonical, or
  const uint64_t* witnesses = nullptr;     // nsteps x step_wires canonical (circom .wtns order)
  size_t nsteps = 0;
  std::vector<Fe> zs;                      // (nsteps + 1) x len_z IVC states, Montgomery
  size_t nbatches = 0, Bk = 0;
  uint32_t nA = 0, nB = 0, nE = 0;         // Poseidon chains of phase A (row data only) / B (need the hashed state) / 2 (after the early field ops)
  bool early_fops = false;
  BaseTables tbl{};
  static constexpr size_t pin_stride = 4 * (size_t)XYZZ_WORDS * MSM_MAX_WINDOWS;
};

// Witness kernels of `rows` rows starting at global row `first` (program order: decompositions, inputs, lane programs, row
// hashes, early field ops, the chains that use them, state hashes, late field ops).  ahead = true stops before the state
// hashes: the ahead-of-time pass of circuits whose IVC state needs values computed from the witness (crop: the hash of the
// cropped row), run with only the predictable part of step_in filled in.
// part: 0 = everything; 1 = only what does not depend on the IVC state (bit decompositions, input wires, the row-hash chains
// with their wires); 2 = the rest, after part 1 (the input wires are rewritten: the state ones were not known in part 1).
static int launch_witness(vimz_prover* p, hipStream_t st, uint32_t* Z, uint32_t* job_out, uint32_t* status, const uint32_t* priv, size_t first, size_t rows,
                          const FoldJob& J, bool ahead, int part = 0) {
  vimz_ctx* ctx = p->ctx;
  const cb::Builder& b = p->circuit->build->b;
  const WitnessDev& W = p->wd;
  const unsigned R = (unsigned)rows;
  if (part != 2)
    for (uint32_t gI = 0; gI < W.n_decomp; gI++) {
      const uint32_t total = (b.decomp[gI].nbits - 1) * b.decomp[gI].count;
      hipLaunchKernelGGL(k_wit_decomp, dim3((total + 255) / 256, R), dim3(256), 0, st, W, gI, priv, Z, status);
    }
  hipLaunchKernelGGL(k_wit_inputs, dim3(((1 + 2 * p->len_z + p->n_priv) + 255) / 256, R), dim3(256), 0, st, W, priv, (const uint32_t*)p->zs_all_d, Z, (uint32_t)first);
  if (part == 1) {
    hipLaunchKernelGGL(k_wit_chains, dim3((J.nA + 3) / 4, R), dim3(64), 0, st, W, 0u, Z, job_out, (const uint32_t*)nullptr);
    P_TRY(hipGetLastError());
    return VIMZ_OK;
  }
  for (uint32_t gI = 0; gI < W.n_groups; gI++)
    hipLaunchKernelGGL(k_wit_lanes, dim3((b.lane_groups[gI].lanes + LANE_TB - 1) / LANE_TB, R), dim3(LANE_TB), 0, st, W, gI, priv, (const uint32_t*)p->zs_all_d, (uint32_t)first, Z, status);
  if (J.nA && !ahead && part != 2) hipLaunchKernelGGL(k_wit_chains, dim3((J.nA + 3) / 4, R), dim3(64), 0, st, W, 0u, Z, job_out, (const uint32_t*)nullptr);
  if (J.early_fops) {
    hipLaunchKernelGGL(k_wit_fops_lc, dim3(p->n_fops, R), dim3(64), 0, st, W, (const uint32_t*)Z, job_out, 1u);
    hipLaunchKernelGGL(k_wit_fops, dim3((R + 63) / 64), dim3(64), 0, st, W, Z, job_out, (uint32_t)rows, 1u);
  }
  if (J.nE) hipLaunchKernelGGL(k_wit_chains, dim3((J.nE + 3) / 4, R), dim3(64), 0, st, W, 2u, Z, job_out, (const uint32_t*)nullptr);
  if (!ahead) {
    if (J.nB) hipLaunchKernelGGL(k_wit_chains, dim3((J.nB + 3) / 4, R), dim3(64), 0, st, W, 1u, Z, job_out, (const uint32_t*)nullptr);
    if (p->n_fops) hipLaunchKernelGGL(k_wit_fops, dim3((R + 63) / 64), dim3(64), 0, st, W, Z, job_out, (uint32_t)rows, 0u);
  }
  P_TRY(hipGetLastError());
  return VIMZ_OK;
}

// Stage 0 (caller holds the lock, device set): every private input to HBM; ONE hash-only pass of the phase-A Poseidon chains
// over ALL rows (they depend on the row data only); the host then runs the whole IVC state chain z_0..z_n and uploads it.
static int fold_prepare(vimz_prover* p, FoldJob& J, bool start_batch0 = false) {
  vimz_ctx* ctx = p->ctx;
  hipStream_t s = ctx->stream;
  const cb::Builder& b = p->circuit->build->b;
  const WitnessDev& W = p->wd;
  const size_t nsteps = J.nsteps, jstride = p->n_jobs + p->n_fops, B = p->max_batch, sw = p->step_wires;
  for (auto& c : b.chains) (c.phase == 0 ? J.nA : c.phase == 1 ? J.nB : J.nE)++;
  for (auto& f : b.fops) if (f.early) J.early_fops = true;
  J.zs.assign((nsteps + 1) * p->len_z, Fe::zero());
  std::vector<Fe>& zs = J.zs;
  for (uint32_t i = 0; i < p->len_z; i++) zs[i] = p->z_cur[i];
  double t0 = now_s();
  if (J.witnesses) {
    // external witnesses: the state chain is read off their public wires, and checked for continuity
    for (size_t r = 0; r < nsteps; r++) {
      const uint64_t* w = J.witnesses + 4 * r * sw;
      for (uint32_t i = 0; i < p->len_z; i++) {
        if (!fe_from_canon(w + 4 * (1 + p->len_z + i)).eq(zs[r * p->len_z + i])) {
          char msg[128]; snprintf(msg, sizeof(msg), "witness %llu: step_in does not continue the IVC state", (unsigned long long)r);
          return vz_fail(ctx, VIMZ_ERR_UNSAT, msg);
        }
        zs[(r + 1) * p->len_z + i] = fe_from_canon(w + 4 * (1 + i));
      }
    }
  } else {
    P_TRY(grow(p->retired, &p->priv_all_d, &p->cap_priv_all, 32 * nsteps * (size_t)p->n_priv, 32 * 1024 * (size_t)p->n_priv));
    P_TRY(grow(p->retired, &p->zs_all_d, &p->cap_zs_all, 32 * (nsteps + 1) * (size_t)p->len_z, 32 * 1025 * (size_t)p->len_z));
    P_TRY(grow(p->retired, &p->job_all_d, &p->cap_job_all, 32 * nsteps * jstride, 32 * 1024 * jstride));
    P_TRY(hipMemcpyAsync(p->priv_all_d, J.step_inputs, 32 * nsteps * (size_t)p->n_priv, hipMemcpyHostToDevice, s));
    P_TRY(hipMemsetAsync(p->job_all_d, 0, 32 * nsteps * jstride, s));
    // Both this pass and the first witness batch are one Poseidon-chain latency long (≈10 ms, whatever the row count), and the
    // first fold waits for both.  For the first batch's rows the chains therefore run once, with their wires, into the batch
    // buffer on the producer's stream, side by side with the hash-only pass of the remaining rows; fold_issue(0) adds the rest.
    size_t rows0 = 0;
    if (start_batch0 && J.nA && !J.nE && !J.early_fops) {
      const size_t nb0 = (nsteps + B - 1) / B;
      rows0 = std::min((nsteps + nb0 - 1) / nb0, nsteps);
      auto& b0 = p->buf[0];
      P_TRY(hipEventRecord(b0.wit_done, s));
      P_TRY(hipStreamWaitEvent(p->sB, b0.wit_done, 0));
      P_TRY(hipMemsetAsync(b0.status, 0, 4 * rows0, p->sB));
      int rc = launch_witness(p, p->sB, b0.Z, b0.job_out, b0.status, p->priv_all_d, 0, rows0, J, false, 1);
      if (rc) return rc;
      P_TRY(hipMemcpyAsync(p->job_all_d, b0.job_out, 32 * rows0 * jstride, hipMemcpyDeviceToDevice, p->sB));
      P_TRY(hipEventRecord(b0.wit_done, p->sB));
      J.batch0_started = true;
    }
    for (size_t off = rows0; off < nsteps && J.nA; off += 32768) {
      const unsigned rows = (unsigned)std::min<size_t>(32768, nsteps - off);
      hipLaunchKernelGGL(k_wit_chains, dim3((J.nA + 3) / 4, rows), dim3(64), 0, s, W, 0u, (uint32_t*)nullptr, p->job_all_d + 8 * off * jstride,
                         (const uint32_t*)(p->priv_all_d + 8 * off * p->n_priv));
    }
    P_TRY(hipGetLastError());
    if (J.nE || J.early_fops) {
      // Ahead-of-time pass: the IVC state of these circuits absorbs values computed from the witness (crop: the hash of the cropped
      // row), which depend on step_in only through its predictable part (state elements that are step_in[j] + constant, e.g. the
      // row counter).  Fill that part in, run the witness kernels up to the phase-2 chains batch by batch, keep the job outputs.
      std::vector<Fe> zp((nsteps + 1) * p->len_z, Fe::zero());
      for (uint32_t i = 0; i < p->len_z; i++) zp[i] = zs[i];
      for (size_t r = 0; r < nsteps; r++)
        for (uint32_t i = 0; i < p->len_z; i++)
          if (b.zout[i].ref.kind == REF_ZIN) zp[(r + 1) * p->len_z + i] = Fe::add(zp[r * p->len_z + b.zout[i].ref.idx], cb::fe_from_i64(b.zout[i].add));
      std::vector<Fe> zc(zp.size());
      for (size_t i = 0; i < zp.size(); i++) zc[i] = Fe::from_mont(zp[i]);
      P_TRY(hipMemcpyAsync(p->zs_all_d, zc.data(), 32 * zc.size(), hipMemcpyHostToDevice, s));
      P_TRY(hipStreamSynchronize(s));
      int rc;
      for (size_t first = 0; first < nsteps; first += B) {
        const size_t rows = std::min(B, nsteps - first);
        P_TRY(hipMemsetAsync(p->buf[0].status, 0, 4 * rows, s));
        if ((rc = launch_witness(p, s, p->buf[0].Z, p->job_all_d + 8 * first * jstride, p->buf[0].status, p->priv_all_d + 8 * first * p->n_priv, first, rows, J, true))) return rc;
      }
    }
    std::vector<Fe> jobA(nsteps * jstride);
    if (J.batch0_started) P_TRY(hipStreamWaitEvent(s, p->buf[0].wit_done, 0));
    P_TRY(hipMemcpyAsync(jobA.data(), p->job_all_d, 32 * nsteps * jstride, hipMemcpyDeviceToHost, s));
    P_TRY(hipStreamSynchronize(s));
    p->phase_s[PH_WITNESS] += now_s() - t0; t0 = now_s();
    host_state_chain(p, J.step_inputs, nsteps, jobA.data(), jstride, zs);
    {
      std::vector<Fe> zc(zs.size());
      for (size_t i = 0; i < zs.size(); i++) zc[i] = Fe::from_mont(zs[i]);
      P_TRY(hipMemcpyAsync(p->zs_all_d, zc.data(), 32 * zc.size(), hipMemcpyHostToDevice, s));
      P_TRY(hipStreamSynchronize(s));
    }
    p->phase_s[PH_ZCHAIN] += now_s() - t0; p->phase_n[PH_ZCHAIN] += nsteps; p->phase_n[PH_WITNESS] += nsteps;
  }
  J.tbl = p->ck->tb(0);
  J.nbatches = (nsteps + B - 1) / B;
  J.Bk = (nsteps + J.nbatches - 1) / J.nbatches;     // even batches (85 rows -> 43 + 42, not 64 + 21): no short tail batch
  return VIMZ_OK;
}

// Producer: batch k on the low-priority stream — full witness generation of the step circuit, then per row the step rows of
// (A,B,C)·z and the commitment to the step circuit's wires.  Records bb.wit_done and one event per row.
static int fold_issue(vimz_prover* p, const FoldJob& J, size_t k) {
  vimz_ctx* ctx = p->ctx;
  const cb::Builder& b = p->circuit->build->b;
  const WitnessDev& W = p->wd;
  const size_t nw = p->n_wires, nc = p->n_c, sw = p->step_wires;
  auto& bb = p->buf[k & 1];
  const size_t first = k * J.Bk, rows = std::min(J.Bk, J.nsteps - first);
  hipStream_t sb = p->sB;
  const bool started = k == 0 && J.batch0_started;      // (its status words already hold the decompositions' range checks)
  if (!started) P_TRY(hipMemsetAsync(bb.status, 0, 4 * rows, sb));
  if (J.witnesses) {
    P_TRY(hipMemcpy2DAsync(bb.Z, 32 * nw, J.witnesses + 4 * first * sw, 32 * sw, 32 * sw, rows, hipMemcpyHostToDevice, sb));
    launch_to_mont<Fr>(sb, bb.Z, rows * nw);
  } else {
    int rc = launch_witness(p, sb, bb.Z, bb.job_out, bb.status, p->priv_all_d + 8 * first * p->n_priv, first, rows, J, false, started ? 2 : 0);
    if (rc) return rc;
  }
  P_TRY(hipGetLastError());
  P_TRY(hipMemcpyAsync(bb.status_host, bb.status, 4 * rows, hipMemcpyDeviceToHost, sb));
  P_TRY(hipEventRecord(bb.wit_done, sb));
  for (size_t r = 0; r < rows; r++) {
    const uint32_t* Zi = bb.Z + 8 * r * nw;
    launch_spmv(p, sb, Zi, bb.az + 8 * r * nc, bb.bz + 8 * r * nc, bb.cz + 8 * r * nc, 1);
    P_TRY(msm_launch<BnG1>(sb, p->wsB, p->ck->d, Zi + 8 * (size_t)p->c0, sw - p->c0, 1, 0, (char*)bb.pin + r * FoldJob::pin_stride, &p->planB, nullptr, 1, p->ck->tables ? &J.tbl : nullptr));
    P_TRY(hipEventRecord(bb.ev[r], sb));
  }
  return VIMZ_OK;
}
