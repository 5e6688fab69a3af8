// Internals of the folding prover shared by the two chains that drive it:
//   prover.hip  the NIFS accumulator over the step circuit's own instances (row segments fold independently and merge)
//   ivc.hip     Nova IVC: the augmented circuits on the BN254/Grumpkin cycle (RecursiveSNARK::prove_step in full)
// Both use the same batch producer (witness kernels, per-row SpMV, witness-commitment MSM on a low-priority stream).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>

#include "internal.hpp"
#include "circuit_handle.hpp"
#include "r1cs_ops.hpp"
#include "witness.hpp"
#include "aug/cs.hpp"      // (affinity_cpus)

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

// A few host threads per prover for the head batch's Poseidon chains (fold_head_batch).  run(n, f) executes f(0..n-1) on the
// workers and the calling thread and returns when all are done.
struct HostPool {
  std::vector<std::thread> th;
  std::mutex m; std::condition_variable cv, cv_done;
  std::function<void(size_t)> fn; size_t n = 0; std::atomic<size_t> next{0}; size_t done = 0, active = 0; uint64_t gen = 0; bool stop = false;
  explicit HostPool(unsigned workers) { for (unsigned i = 0; i < workers; i++) th.emplace_back([this] { loop(); }); }
  ~HostPool() { { std::lock_guard<std::mutex> g(m); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
  void drain() {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= n) break;
      fn(i);
      std::lock_guard<std::mutex> g(m);
      if (++done == n) cv_done.notify_all();
    }
  }
  void loop() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv.wait(lk, [&] { return stop || gen != seen; });
      if (stop) return;
      seen = gen; active++;
      lk.unlock(); drain(); lk.lock();
      if (--active == 0) cv_done.notify_all();
    }
  }
  std::mutex run_mu;       // one run at a time: callers of different provers queue up instead of oversubscribing the host's cores
  void run(size_t count, std::function<void(size_t)> f) {
    if (!count) return;
    std::lock_guard<std::mutex> one(run_mu);
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return active == 0; });      // stragglers of the previous call have left drain(): fn / n may change
    fn = std::move(f); n = count; next = 0; done = 0; gen++;
    lk.unlock();
    cv.notify_all();
    drain();
    lk.lock();
    while (!cv_done.wait_for(lk, std::chrono::seconds(10), [&] { return done == n; }))
      fprintf(stderr, "[vimz] host pool: %zu of %zu tasks done after 10 s (next %zu, active workers %zu)\n", done, n, next.load(), active);
  }
};

// ONE pool per process (defined in prover.hip), shared by every prover: the GPU boxes grant a process 16 CPUs, every IVC keeps about
// three threads busy while it folds, and three concurrent segments with a pool of fourteen workers each — all evaluating head
// batches at once in a short fold call — got the whole process throttled (20-row windows: 550 to 780 steps/s from run to run).
HostPool& vz_shared_pool();

enum { PH_WITNESS = 0, PH_ZCHAIN, PH_SPMV, PH_MSM_W, PH_CROSS, PH_MSM_T, PH_RO, PH_FOLD, PH_HOST_EC, PH_COUNT };

// one segment's end state handed to the next (vimz_prover::start_from / end_to)
struct StartLink {
  std::mutex mu; std::condition_variable cv; bool ready = false, failed = false; std::vector<vz::cb::Fe> z;
  void publish(const vz::cb::Fe* zz, size_t n) { { std::lock_guard<std::mutex> g(mu); z.assign(zz, zz + n); ready = true; } cv.notify_all(); }
  void fail() { { std::lock_guard<std::mutex> g(mu); if (!ready) { failed = true; ready = true; } } cv.notify_all(); }
  bool wait(std::vector<vz::cb::Fe>& out) { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return ready; }); if (failed) return false; out = z; return true; }
};

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
  // rows [0, n_bool) of the step circuit are b·(b − 1) = 0 (builder.hpp); want_s1: the producer also sums, per row, the key's points on those rows whose
  // fresh bit is one (S_1 of the boolean-row form of the cross term's commitment, ivc.hip) into the row's pinned slot S1_SLOT
  uint32_t n_bool = 0; bool want_s1 = false;
  static constexpr size_t S1_SLOT = 4 * (size_t)XYZZ_WORDS * (MSM_MAX_WINDOWS - 1);
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
    // IVC lookahead (want_d): per row minus the fresh x fresh cross term with the row before it, and its commitment (fold_issue_d)
    uint32_t* d = nullptr; void* pin_d = nullptr;
    std::vector<hipEvent_t> ev_p, ev_d;   // per row: (A,B,C)·z done on the producer's stream; the commitment to d done on stream sD
    std::vector<uint8_t> has_d;
    // Host-side hand-over of a row: behind the row's last kernel the producer's stream stores this filling's generation into a pinned
    // word (k_row_flag) and the fold thread reads that word (wait_row_flag) — no runtime call on the producer's event from another
    // thread.  [0, B): the fresh instance's work (beside ev[r]); [B, 2B): the lookahead's commitment (beside ev_d[r]).
    uint32_t* row_flag = nullptr; uint32_t gen = 0; size_t flag_rows = 0;
  } buf[2];
  bool want_d = false;
  hipStream_t sD = nullptr;      // = sH (a stream of its own for the lookahead's commitments — a sixth — shifted the streams' hardware
                                 // queues and cost one proof 10 %, with nothing ever queued on it)
  MsmPlan planD{};
  MsmWorkspace wsB;
  MsmPlan planB{};
  // head batch of a fold call: Poseidon jobs evaluated on the host, everything else on a stream of its own (fold_head_batch)
  hipStream_t sH = nullptr; hipEvent_t ev_head = nullptr, ev_hash = nullptr;
  MsmWorkspace wsH;
  std::vector<uint32_t> job_stage_off;     // [n_jobs + 1] wire offsets of the jobs inside a staging row
  std::vector<uint32_t> job_fold_mask;     // per job: lanes whose round-0 S-box is folded (constant-zero inputs)
  const uint32_t* job_stage_off_d = nullptr;
  Fe* stage_host = nullptr; Fe* jobvals_host = nullptr; Fe* zs_host = nullptr;   // pinned
  uint32_t *stage_d = nullptr, *jobvals_d = nullptr;
  size_t head_rows_cap = 0;
  size_t last_head_rows = 0;             // rows of the last fold call whose Poseidon chains were evaluated on the host (the head batch)
  // row-hash jobs of a coming call's head rows, evaluated ahead (head_precompute): the call over exactly these inputs finds them done
  const uint64_t* pre_inputs = nullptr; size_t pre_rows = 0;
  bool head_eligible = false;
  // DEFERRED START STATE of a fold call (vimz_ivc_fold_segments): the segments of one proof are folded concurrently, and segment k starts at the
  // state segment k-1 ends in — which follows from the row hashes of k-1's rows, one Poseidon-chain latency (4-5 ms) after ITS call began.  So
  // every segment's call begins at once (inputs uploaded, the row-hash chains of its first batch running, with their wires), and only where the
  // host state chain needs the start state does segment k wait for its predecessor's end state (start_from), set it (on_start: the IVC
  // layer's z_0) and hand its own end state on (end_to).  Every row is hashed once; no segment waits for a hash-only pass of another's rows.
  StartLink* start_from = nullptr; StartLink* end_to = nullptr; std::function<void()> on_start;
  // ... and (a rank of a sharded proof, vimz_ivc_fold_segments_begin) the rows' digests — the job values the state chain reads — are handed to the caller
  // the moment the call's own chain pass has produced them: the rank exchanges them with the other ranks instead of hashing its rows a second time
  std::function<void(const Fe* job_values, size_t nsteps, size_t jstride)> on_digests;
  bool suppress_head = false;      // this call is one of several concurrent segments of a proof too long for a head batch (HEAD_JOB_MAX): its rows are hashed on the GPU
  // per fold call: all private inputs, all IVC states and all row hashes resident
  uint32_t *priv_all_d = nullptr, *zs_all_d = nullptr, *job_all_d = nullptr;
  size_t cap_priv_all = 0, cap_zs_all = 0, cap_job_all = 0;
  const uint64_t* preloaded_inputs = nullptr; size_t preloaded_rows = 0;      // rows already in priv_all_d (vz_prover_preload_inputs): the next fold call over exactly them skips its upload
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
// The IVC state chain is the one strictly serial part of a fold (and what a sharded proof's later segments wait for): two Poseidon
// permutations per row at contrast, ≈ 50 µs.  Most circuits' state elements are INDEPENDENT chains (contrast: the hash chain of the
// original rows, the hash chain of the transformed rows, the constant factor): the value-only chain (no wires wanted) runs every
// independent component on a thread of its own.  plan: components of the graph  state element i — state element k  when z_out[i]
// reads z_in[k]; per component the phase-B jobs and field ops its outputs need, in evaluation order.
struct ChainPlan { std::vector<std::vector<uint32_t>> outs, jobs, fops; };
static ChainPlan chain_plan(const cb::Builder& b) {
  const uint32_t L = b.len_z, nj = (uint32_t)b.jobs.size(), nf = (uint32_t)b.fops.size();
  std::vector<uint32_t> dep_job(nj, 0), dep_fop(nf, 0);
  std::vector<int> done_job(nj, 0), done_fop(nf, 0);
  std::function<uint32_t(const ValRef&)> deps = [&](const ValRef& r) -> uint32_t {
    switch (r.kind) {
      case REF_ZIN: return r.idx < 32 ? 1u << r.idx : 0xffffffffu;
      case REF_WIRE: return (r.idx > L && r.idx <= 2 * L) ? 1u << (r.idx - 1 - L) : 0u;
      case REF_JOB: {
        if (r.idx >= nj || b.chains[b.jobs[r.idx].chain].phase != 1) return 0u;
        if (!done_job[r.idx]) { done_job[r.idx] = 1; uint32_t d = 0; for (uint32_t i = 0; i + 1 < b.jobs[r.idx].t; i++) d |= deps(b.jobs[r.idx].in[i]); dep_job[r.idx] = d; }
        return dep_job[r.idx];
      }
      case REF_FOP: {
        if (r.idx >= nf || b.fops[r.idx].op == FOP_LC) return 0u;
        if (!done_fop[r.idx]) { done_fop[r.idx] = 1; const FieldOp& F = b.fops[r.idx]; uint32_t d = deps(F.a); if (F.op == FOP_MUX) d |= deps(F.b) | deps(F.c); dep_fop[r.idx] = d; }
        return dep_fop[r.idx];
      }
      default: return 0u;
    }
  };
  std::vector<uint32_t> parent(L);
  for (uint32_t i = 0; i < L; i++) parent[i] = i;
  std::function<uint32_t(uint32_t)> find = [&](uint32_t x) { return parent[x] == x ? x : parent[x] = find(parent[x]); };
  for (uint32_t i = 0; i < L; i++) { const uint32_t d = deps(b.zout[i].ref); for (uint32_t k = 0; k < L && k < 32; k++) if ((d >> k) & 1u) parent[find(i)] = find(k); }
  ChainPlan P;
  std::vector<int> comp_of(L, -1);
  for (uint32_t i = 0; i < L; i++) {
    const uint32_t root = find(i);
    if (comp_of[root] < 0) { comp_of[root] = (int)P.outs.size(); P.outs.emplace_back(); P.jobs.emplace_back(); P.fops.emplace_back(); }
    P.outs[comp_of[root]].push_back(i);
  }
  for (size_t c = 0; c < P.outs.size(); c++) {
    std::vector<uint8_t> need_job(nj, 0), need_fop(nf, 0);
    std::function<void(const ValRef&)> mark = [&](const ValRef& r) {
      if (r.kind == REF_JOB && r.idx < nj && b.chains[b.jobs[r.idx].chain].phase == 1 && !need_job[r.idx]) {
        need_job[r.idx] = 1; for (uint32_t i = 0; i + 1 < b.jobs[r.idx].t; i++) mark(b.jobs[r.idx].in[i]);
      } else if (r.kind == REF_FOP && r.idx < nf && b.fops[r.idx].op != FOP_LC && !need_fop[r.idx]) {
        need_fop[r.idx] = 1; const FieldOp& F = b.fops[r.idx]; mark(F.a); if (F.op == FOP_MUX) { mark(F.b); mark(F.c); }
      }
    };
    for (uint32_t i : P.outs[c]) mark(b.zout[i].ref);
    for (auto& ch : b.chains) if (ch.phase == 1) for (uint32_t k = 0; k < ch.job_cnt; k++) if (need_job[ch.job_off + k]) P.jobs[c].push_back(ch.job_off + k);
    for (uint32_t f = 0; f < nf; f++) if (need_fop[f]) P.fops[c].push_back(f);
  }
  return P;
}

// stage / jobvals (optional, head batch): also emit the wires of the state-dependent (phase-B) jobs into the row's staging area and
// their outputs into the row's job values, so that no Poseidon work of these rows is left for the GPU.
static void host_state_chain(const vimz_prover* p, const uint64_t* inputs, size_t rows, const Fe* jobA, size_t jstride, std::vector<Fe>& zs, size_t zs_row0 = 0,
                             Fe* stage = nullptr, size_t stage_row = 0, Fe* jobvals = nullptr) {
  const cb::Builder& b = p->circuit->build->b;
  if (!stage && rows >= 16) {       // value-only chain: one thread per independent component that has Poseidon work
    const ChainPlan P = chain_plan(b);
    size_t heavy = 0;
    for (auto& j : P.jobs) if (!j.empty()) heavy++;
    if (heavy >= 2) {
      auto run = [&](size_t c) {
        HostEval ev; ev.P = p; ev.b = &b;
        ev.job_b.assign(p->n_jobs, Fe::zero()); ev.fop.assign(p->n_fops, Fe::zero());
        for (size_t r = 0; r < rows; r++) {
          ev.priv = inputs + 4 * r * p->n_priv; ev.job_a = jobA + r * jstride; ev.zin = zs.data() + (zs_row0 + r) * p->len_z;
          for (uint32_t j : P.jobs[c]) {
            const HashJob& J = b.jobs[j];
            Fe in[POSEIDON_MAX_T];
            for (uint32_t i = 0; i + 1 < J.t; i++) in[i] = ev.value(J.in[i]);
            ev.job_b[j] = cb::poseidon_hash(in, (int)J.t - 1);
          }
          for (uint32_t f : P.fops[c]) {
            const FieldOp& F = b.fops[f];
            if (F.op == FOP_ISZERO) { Fe in = ev.value(F.a); ev.fop[f] = in.is_zero() ? Fe::one() : Fe::zero(); }
            else if (F.op == FOP_MUX) { Fe sv = ev.value(F.a), c0 = ev.value(F.b), c1 = ev.value(F.c); ev.fop[f] = Fe::add(Fe::mul(Fe::sub(c1, c0), sv), c0); }
          }
          Fe* zn = zs.data() + (zs_row0 + r + 1) * p->len_z;
          for (uint32_t i : P.outs[c]) zn[i] = Fe::add(ev.value(b.zout[i].ref), cb::fe_from_i64(b.zout[i].add));
        }
      };
      std::vector<std::thread> th;
      size_t mine = P.outs.size();                   // the calling thread takes the first heavy component and all light ones
      for (size_t c = 0; c < P.outs.size(); c++) {
        if (P.jobs[c].empty()) continue;
        if (mine == P.outs.size()) { mine = c; continue; }
        th.emplace_back(run, c);
      }
      for (size_t c = 0; c < P.outs.size(); c++) if (P.jobs[c].empty() || c == mine) run(c);
      for (auto& t : th) t.join();
      return;
    }
  }
  HostEval ev; ev.P = p; ev.b = &b;
  for (size_t r = 0; r < rows; r++) {
    ev.priv = inputs + 4 * r * p->n_priv; ev.job_a = jobA + r * jstride; ev.zin = zs.data() + (zs_row0 + r) * p->len_z;
    ev.job_b.assign(p->n_jobs, Fe::zero()); ev.fop.assign(p->n_fops, Fe::zero());
    for (auto& c : b.chains) {
      if (c.phase != 1) continue;
      for (uint32_t k = 0; k < c.job_cnt; k++) {
        const uint32_t j = c.job_off + k;
        const HashJob& J = b.jobs[j];
        if (stage) {
          Fe st[POSEIDON_MAX_T]; st[0] = Fe::zero();
          for (uint32_t i = 0; i + 1 < J.t; i++) st[1 + i] = ev.value(J.in[i]);
          cb::poseidon_job_wires<BnFr>(st, (int)J.t, p->job_fold_mask[j], J.out_wire != 0, stage + r * stage_row + p->job_stage_off[j]);
          ev.job_b[j] = st[0];
          jobvals[r * jstride + j] = st[0];
        } else {
          Fe in[POSEIDON_MAX_T];
          for (uint32_t i = 0; i + 1 < J.t; i++) in[i] = ev.value(J.in[i]);
          ev.job_b[j] = cb::poseidon_hash(in, (int)J.t - 1);
        }
      }
    }
    for (uint32_t f = 0; f < p->n_fops; f++) {
      const FieldOp& F = b.fops[f];
      if (F.op == FOP_ISZERO) { Fe in = ev.value(F.a); ev.fop[f] = in.is_zero() ? Fe::one() : Fe::zero(); }
      else if (F.op == FOP_MUX) { Fe sv = ev.value(F.a), c0 = ev.value(F.b), c1 = ev.value(F.c); ev.fop[f] = Fe::add(Fe::mul(Fe::sub(c1, c0), sv), c0); }
    }
    Fe* zn = zs.data() + (zs_row0 + r + 1) * p->len_z;
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
// The rows of the prover's NEXT fold call, uploaded now (blocking, on the calling thread): a caller that is about to start several segments' calls side by side does this
// for each of them first, while nothing runs — an upload issued from a segment's own thread while its siblings' first kernels start was seen to sit in the runtime for 7 ms
// once in eight proofs (profiles/r06_slow_pass_trace.txt).
static int vz_prover_preload_inputs(vimz_prover* p, const uint64_t* step_inputs, size_t nsteps);

// One call of vimz_prover_fold / vimz_ivc_fold: the inputs, the IVC state chain and the batch schedule.
struct FoldJob {
  int started_batch = -1;          // fold_prepare already ran the state-independent part of this batch's witness (see there)
  const uint64_t* step_inputs = nullptr;   // nsteps x n_priv canonical, or
  const uint64_t* witnesses = nullptr;     // nsteps x step_wires canonical (circom .wtns order)
  size_t nsteps = 0;
  std::vector<Fe> zs;                      // (nsteps + 1) x len_z IVC states, Montgomery
  size_t nbatches = 0;
  std::vector<size_t> bfirst, brows;       // batch k = rows [bfirst[k], bfirst[k] + brows[k]); buffer k & 1
  size_t first(size_t k) const { return bfirst[k]; }
  size_t rows(size_t k) const { return brows[k]; }
  uint32_t nA = 0, nB = 0, nE = 0;         // Poseidon chains of phase A (row data only) / B (need the hashed state) / 2 (after the early field ops)
  bool early_fops = false;
  size_t pre_rows = 0;            // head rows whose row-hash jobs were evaluated ahead of this call (head_precompute), taken over in fold_prepare
  BaseTables tbl{};
  // head batch on the host (fold_head_batch): batch 0 is already issued when fold_prepare returns, and the IVC states of the rows
  // after it arrive later, from a helper thread (the hash-only pass over those rows takes one Poseidon-chain latency on the GPU)
  bool head = false;
  size_t next_issue = 0;                   // first batch not yet handed to fold_issue
  std::thread helper; std::atomic<size_t> states_upto{0};   // rows whose IVC states (zs, on host and device) exist
  std::atomic<int> helper_done{1}; int helper_rc = VIMZ_OK; std::string helper_err;
  std::vector<Fe> jobA_rest;
  ~FoldJob() { if (helper.joinable()) helper.join(); }
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
  // (the chains compute in the reduced-radix form and leave their S-box wires in it: converted right behind each chain launch)
  auto chain_wires_std = [&](uint32_t phase) {
    if (W.poseidon29 && p->n_jobs) hipLaunchKernelGGL(k_wit_chain_wires_std, dim3(p->n_jobs, R), dim3(128), 0, st, W, p->job_stage_off_d, phase, Z);
  };
  if (part != 2)
    for (uint32_t gI = 0; gI < W.n_decomp; gI++) {
      const uint32_t total = (b.decomp[gI].nbits - 1) * b.decomp[gI].count;
      hipLaunchKernelGGL(k_wit_decomp, dim3((total + 255) / 256, R), dim3(256), 0, st, W, gI, priv, Z, status);
    }
  hipLaunchKernelGGL(k_wit_inputs, dim3(((1 + 2 * p->len_z + p->n_priv) + 255) / 256, R), dim3(256), 0, st, W, priv, (const uint32_t*)p->zs_all_d, Z, (uint32_t)first);
  if (part == 1) {
    hipLaunchKernelGGL(k_wit_chains, dim3((J.nA + 3) / 4, R), dim3(64), 0, st, W, 0u, Z, job_out, (const uint32_t*)nullptr);
    chain_wires_std(0u);
    P_TRY(hipGetLastError());
    return VIMZ_OK;
  }
  for (uint32_t gI = 0; gI < W.n_groups; gI++)
    hipLaunchKernelGGL(k_wit_lanes, dim3((b.lane_groups[gI].lanes + LANE_TB - 1) / LANE_TB, R), dim3(LANE_TB), 0, st, W, gI, priv, (const uint32_t*)p->zs_all_d, (uint32_t)first, Z, status);
  if (J.nA && !ahead && part != 2) { hipLaunchKernelGGL(k_wit_chains, dim3((J.nA + 3) / 4, R), dim3(64), 0, st, W, 0u, Z, job_out, (const uint32_t*)nullptr); chain_wires_std(0u); }
  if (J.early_fops) {
    hipLaunchKernelGGL(k_wit_fops_lc, dim3(p->n_fops, R), dim3(64), 0, st, W, (const uint32_t*)Z, job_out, 1u);
    hipLaunchKernelGGL(k_wit_fops, dim3((R + 63) / 64), dim3(64), 0, st, W, Z, job_out, (uint32_t)rows, 1u);
  }
  if (J.nE) { hipLaunchKernelGGL(k_wit_chains, dim3((J.nE + 3) / 4, R), dim3(64), 0, st, W, 2u, Z, job_out, (const uint32_t*)nullptr); chain_wires_std(2u); }
  if (!ahead) {
    if (J.nB) { hipLaunchKernelGGL(k_wit_chains, dim3((J.nB + 3) / 4, R), dim3(64), 0, st, W, 1u, Z, job_out, (const uint32_t*)nullptr); chain_wires_std(1u); }
    if (p->n_fops) hipLaunchKernelGGL(k_wit_fops, dim3((R + 63) / 64), dim3(64), 0, st, W, Z, job_out, (uint32_t)rows, 0u);
  }
  P_TRY(hipGetLastError());
  return VIMZ_OK;
}

// Even split of rows [first, first + n) into batches of at most B rows, appended to the job's batch table
// (85 rows -> 43 + 42, not 64 + 21: no short tail batch).
static void plan_batches(FoldJob& J, size_t first, size_t n, size_t B) {
  if (!n) return;
  const size_t nb = (n + B - 1) / B, per = (n + nb - 1) / nb;
  for (size_t off = 0; off < n; off += per) { J.bfirst.push_back(first + off); J.brows.push_back(std::min(per, n - off)); }
}

template <int DUMMY>
__global__ void k_row_flag(uint32_t* flag, uint32_t v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
static inline hipError_t mark_row(hipStream_t st, vimz_prover::BatchBuf& bb, size_t idx) {
  hipLaunchKernelGGL(k_row_flag<0>, dim3(1), dim3(1), 0, st, bb.row_flag + idx, bb.gen);
  return hipGetLastError();
}
// The fold thread's wait for row idx of a batch that has been issued (the word is written in stream order behind the row's results, which
// went to pinned memory too), and the ONLY way the fold's streams depend on the producer's: what they launch for a row is launched after
// this returned, never behind a hipStreamWaitEvent on the producer's event.  The producer's streams have the lowest priority, the fold's
// the highest; a high-priority queue stalled in a barrier behind a producer's event still counts as having work, so the producer's queues
// were served in leftovers exactly when the fold was waiting for them: a producer that once fell behind stayed behind, and provers
// created after others had folded on the same context ran 10× slower (20–250 steps/s instead of 800–930: DESIGN.md §5c).  Nor is the
// event synchronised on from this thread (that goes through the runtime's lock of the producer's stream, which the producer's
// thread needs for every launch); it is looked at now and then, for a device error, and ends the wait if it completed.
static inline hipError_t wait_row_flag(const vimz_prover::BatchBuf& bb, size_t idx, hipEvent_t ev) {
  const uint32_t want = bb.gen;
  const uint32_t* f = bb.row_flag + idx;
  double t_start = 0;
  for (uint32_t spins = 1; __atomic_load_n(f, __ATOMIC_ACQUIRE) != want; spins++) {
    if (spins & 31) continue;
    std::this_thread::yield();
    if ((spins & 0x3ffff) == 0) {
      const hipError_t q = hipEventQuery(ev);
      if (q == hipSuccess) break;                     // (recorded behind the word's kernel)
      if (q != hipErrorNotReady) return q;
      if (t_start == 0) t_start = now_s(); else if (now_s() - t_start > 120.0) return hipErrorNotReady;      // fails loudly instead of hanging
    }
  }
  return hipSuccess;
}
static int fold_issue(vimz_prover* p, const FoldJob& J, size_t k);

// Host CPUs this process can really use: what the OS shows, capped by the cgroup's CPU quota (cpu.max / cfs_quota_us).  A GPU box shows
// 256 logical CPUs and grants 16; threads beyond the quota get the whole process throttled for the rest of a 100 ms period.
static bool getenv_once(const char* name) {      // (debugging switches are read once; each call site has its own name)
  static std::mutex m; static std::vector<std::pair<std::string, bool>> seen;
  std::lock_guard<std::mutex> g(m);
  for (auto& e : seen) if (e.first == name) return e.second;
  seen.emplace_back(name, getenv(name) != nullptr);
  return seen.back().second;
}
static unsigned usable_cpus() {
  static const unsigned v = [] {
    unsigned n = aug::affinity_cpus();      // (the affinity mask, not the machine: aug/cs.hpp)
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0}; long per = 0;
      if (fscanf(f, "%31s %ld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) { const long k = atol(q) / per; if (k >= 1) n = std::min<unsigned>(n, (unsigned)k); }
      fclose(f);
    } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      long quota = -1, per = 100000;
      if (fscanf(g, "%ld", &quota) != 1) quota = -1;
      fclose(g);
      if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%ld", &per) != 1) per = 100000; fclose(h); }
      if (quota > 0 && per > 0 && quota / per >= 1) n = std::min<unsigned>(n, (unsigned)(quota / per));
    }
    // the ranks of a node share the quota (and, unless the launcher binds them, the affinity mask): a rank's share is what its helper pools may count on
    // (a mask narrower than the machine means the launcher has bound this rank to cores of its own: that IS its share)
    const bool bound = aug::affinity_cpus() < std::max(1u, std::thread::hardware_concurrency());
    if (const char* lw = getenv("LOCAL_WORLD_SIZE")) { const long k = atol(lw); if (k > 1 && !bound) n = std::max(1u, n / (unsigned)k); }
    return n;
  }();
  return v;
}

// Waits of the folding thread: the runtime's own.  (Round 4 tried polling the stream / event with a yield between polls on ranks with few
// cores: +3 % on two cores — and, with four ranks sharing one GPU, ONE of the four then waited 14-28 ms per step for its small MSM,
// every run: 65 instead of 540 steps/s for the proof set.  Removed.)
static inline hipError_t vz_wait_stream(hipStream_t s) { return hipStreamSynchronize(s); }
static inline hipError_t vz_wait_event(hipEvent_t e) { return hipEventSynchronize(e); }

// Rows of a call whose Poseidon jobs are evaluated on the host (VIMZ_HEAD_ROWS overrides; 0 switches the head batch off).
// Round 2 chose 24: the first GPU-produced batch then needed one Poseidon-chain latency on the low-priority producer stream — 10 ms alone,
// 16-18 ms next to the first folds.  Since round 4 the chains run in the reduced-radix arithmetic (witness.hpp: poseidon_group29) and a
// GPU-produced first batch is ready after 6 ms; the head's Poseidon work — 1.8 ms per row and core — now only pays where the whole call
// fits in it on a machine with cores to spare:
//   a rank's share of the cores below six: no head (two cores, 20 rows: 667-707 steps/s against 441-445 with all rows in the head; 256 rows: 950-1 007 / 925);
//   otherwise lone calls of at most HEAD_CALL_MAX = 24 rows, and the segments of one proof (vimz_ivc_fold_segments) while the proof has at most HEAD_JOB_MAX = 28
//   rows in all: every row in the head; longer calls / proofs: none.  (Rounds 2-4: calls of at most 48 rows.  Since round 5 the
//   segments of one proof start together and hash every row once on the GPU — deferred start states, merge.hip — and the crossover moved: three segments of
//   7 / 8 / 10 / 12 / 14 / 16 rows each run at 950 / 901 / 923 / 952 / 967 / 966 steps/s with a head batch against 852 / 878 / 930 / 972 / 997 / 1 018 without,
//   profiles/r05_head_crossover.txt)
//   (256 rows: 1 174-1 176 against 1 087-1 100).  A head shorter than the call AND short (2-8 rows of 10) is the one thing to avoid: the rest
//   then waits for two chain latencies, the hash-only pass and the batch's own (210-450 steps/s).
std::atomic<long>& vz_head_rows_override();      // vimz_set_head_rows: -1 = the policy below
constexpr size_t HEAD_CALL_MAX = 24;      // a lone call: its rows' 2 x 24 chains are four rounds of the 14-worker pool, about one chain latency of the GPU
constexpr size_t HEAD_JOB_MAX = 28;       // the concurrent segments of ONE proof share that pool: a head batch only while ALL their rows fit four rounds
static size_t head_rows_wanted(size_t nsteps = 0) {
  static const long env = getenv("VIMZ_HEAD_ROWS") ? atol(getenv("VIMZ_HEAD_ROWS")) : -1;
  const long ov = vz_head_rows_override().load(std::memory_order_relaxed);
  if (ov >= 0) return (size_t)ov;
  if (env >= 0) return (size_t)env;
  static const bool few_cores = usable_cpus() < 6;
  if (few_cores) return 0;
  return nsteps == 0 || nsteps <= HEAD_CALL_MAX ? 24 : 0;
}

// The IVC's lookahead schedule (ivc.hip) is an option: VIMZ_IVC_LOOKAHEAD=1 (read once).  It takes the large MSM off a
// step's dependent chains at the price of one more (cheap) commitment per row from the producer; measured 572 against 651 steps/s for
// one proof at contrast HD — the producer's two low-priority streams then run at the pace of the folds and the proof waits for rows.
static bool ivc_lookahead_enabled() { static const bool v = getenv("VIMZ_IVC_LOOKAHEAD") && atoi(getenv("VIMZ_IVC_LOOKAHEAD")) != 0; return v; }

// IVC lookahead: row r of batch k gets negB = −T(previous row, this row) over the step rows and its commitment, on the row's own
// producer stream `sp` behind its commitment to the witness (same workspace, used one after the other).  The previous row is the one
// before it in the batch, or the last row of the batch before (the other buffer — still intact: it is rewritten by batch k+1, whose
// producer waits for this); its products may come from the other producer stream: ev_p.
static int fold_issue_d(vimz_prover* p, const FoldJob& J, size_t k, size_t r, hipStream_t sp, MsmWorkspace& ws) {
  vimz_ctx* ctx = p->ctx;
  auto& bb = p->buf[k & 1];
  bb.has_d[r] = 0;
  if (!p->want_d) return VIMZ_OK;
  const size_t nc = p->n_c, sc = p->step_c;
  const uint32_t *az0, *bz0, *cz0; hipEvent_t prev;
  if (r > 0) { az0 = bb.az + 8 * (r - 1) * nc; bz0 = bb.bz + 8 * (r - 1) * nc; cz0 = bb.cz + 8 * (r - 1) * nc; prev = bb.ev_p[r - 1]; }
  else if (k > 0) { auto& ob = p->buf[(k - 1) & 1]; const size_t lr = J.rows(k - 1) - 1; az0 = ob.az + 8 * lr * nc; bz0 = ob.bz + 8 * lr * nc; cz0 = ob.cz + 8 * lr * nc; prev = ob.ev_p[lr]; }
  else return VIMZ_OK;        // first row of a call: its cross term is computed against the running instance directly
  P_TRY(hipStreamWaitEvent(sp, prev, 0));      // (the previous row's (A,B,C)·z may come from the other producer stream)
  uint32_t* d = bb.d + 8 * r * sc;
  hipLaunchKernelGGL(k_fresh_cross_neg<Fr>, dim3(stream_grid(sc)), dim3(256), 0, sp, sc, az0, bz0, cz0, bb.az + 8 * r * nc, bb.bz + 8 * r * nc, bb.cz + 8 * r * nc, d);
  P_TRY(hipGetLastError());
  P_TRY(msm_launch<BnG1>(sp, ws, p->ck->d, d, sc, 1, 0, (char*)bb.pin_d + r * FoldJob::pin_stride, &p->planD, nullptr, 1, nullptr));
  P_TRY(mark_row(sp, bb, bb.flag_rows + r));
  P_TRY(hipEventRecord(bb.ev_d[r], sp));
  bb.has_d[r] = 1;
  return VIMZ_OK;
}


// HEAD BATCH.  A fold call cannot start folding before the first rows' witnesses exist, and those contain the row-hash Poseidon
// chains: 17 dependent permutations, 10 ms on the GPU whatever the row count (one wave per chain, ~0.6 ms per permutation) but
// 1.1 ms on a CPU core (66 µs per permutation).  So the Poseidon jobs of the first few rows of a call are evaluated on host
// threads — outputs for the IVC state chain and every S-box wire — while the GPU does the rest of those rows' witnesses (bit
// decompositions, lane programs, field ops) on a stream of its own and, on the producer's stream, already runs the chains of the
// NEXT batch.  The first fold of a call starts after ~3 ms instead of ~20 ms (VIMZ_DEBUG_TIMING=1 prints it).
// Leaves batch 0 fully issued (events bb.wit_done, bb.ev[r] recorded on p->sH) and the IVC states zs[0..rows] filled in.
// pinned + device staging for the largest head this prover can see (first use; kept: pinned allocations take milliseconds and must not
// recur in later calls).  Caller holds the context's lock and has set the device.
static int head_reserve(vimz_prover* p, size_t rows) {
  vimz_ctx* ctx = p->ctx;
  if (rows <= p->head_rows_cap) return VIMZ_OK;
  const size_t jstride = p->n_jobs + p->n_fops, stage_row = p->job_stage_off.back();
  const size_t cap_rows = std::max(rows, std::min(head_rows_wanted(), p->max_batch));
  if (p->stage_host) { hipHostFree(p->stage_host); hipHostFree(p->jobvals_host); hipHostFree(p->zs_host); p->retired.push_back(p->stage_d); p->retired.push_back(p->jobvals_d); }
  p->stage_host = nullptr; p->jobvals_host = nullptr; p->zs_host = nullptr; p->stage_d = nullptr; p->jobvals_d = nullptr; p->head_rows_cap = 0;
  p->pre_rows = 0; p->pre_inputs = nullptr;
  P_TRY(hipHostMalloc((void**)&p->stage_host, 32 * cap_rows * stage_row));
  P_TRY(hipHostMalloc((void**)&p->jobvals_host, 32 * cap_rows * jstride));
  P_TRY(hipHostMalloc((void**)&p->zs_host, 32 * (cap_rows + 1) * (size_t)p->len_z));
  P_TRY(hipMalloc((void**)&p->stage_d, 32 * cap_rows * stage_row));
  P_TRY(hipMalloc((void**)&p->jobvals_d, 32 * cap_rows * jstride));
  p->head_rows_cap = cap_rows;
  return VIMZ_OK;
}
// The row-hash chains (phase A) of `rows` head rows on the host pool, one task per (row, chain): outputs into jobvals_host, every S-box wire
// into stage_host in the witness program's own wire order.  Depends on the rows' private inputs only — not on the IVC state.
static void head_rowhash_jobs(vimz_prover* p, const uint64_t* inputs, size_t rows) {
  const cb::Builder& b = p->circuit->build->b;
  const size_t jstride = p->n_jobs + p->n_fops, stage_row = p->job_stage_off.back();
  std::vector<uint32_t> chainsA;
  for (uint32_t c = 0; c < b.chains.size(); c++) if (b.chains[c].phase == 0) chainsA.push_back(c);
  Fe* stage = p->stage_host; Fe* jobvals = p->jobvals_host;
  memset(jobvals, 0, 32 * rows * jstride);
  const uint32_t priv0 = 1 + 2 * b.len_z;
  vz_shared_pool().run(rows * chainsA.size(), [&](size_t task) {
    const size_t r = task / chainsA.size();
    const Chain& C = b.chains[chainsA[task % chainsA.size()]];
    Fe prev = Fe::zero();
    for (uint32_t k = 0; k < C.job_cnt; k++) {
      const uint32_t j = C.job_off + k;
      const HashJob& Jb = b.jobs[j];
      Fe st[POSEIDON_MAX_T]; st[0] = Fe::zero();
      for (uint32_t i = 0; i + 1 < Jb.t; i++) {
        const ValRef& ref = Jb.in[i];
        if (ref.kind == REF_WIRE) st[1 + i] = fe_from_canon(inputs + 4 * (r * p->n_priv + (ref.idx - priv0)));
        else if (ref.kind == REF_JOB) st[1 + i] = ref.idx + 1 == j ? prev : jobvals[r * jstride + ref.idx];
        else st[1 + i] = Fe::zero();
      }
      cb::poseidon_job_wires<BnFr>(st, (int)Jb.t, p->job_fold_mask[j], Jb.out_wire != 0, stage + r * stage_row + p->job_stage_off[j]);
      prev = st[0];
      jobvals[r * jstride + j] = st[0];
    }
  });
}
// Would a fold call of `nsteps` rows on this prover put ALL its rows into the host-evaluated head batch (fold_prepare's rule)?
static bool head_takes_whole_call(const vimz_prover* p, size_t nsteps) {
  const cb::Builder& b = p->circuit->build->b;
  bool nA = false, nE = false, early = false;
  for (auto& c : b.chains) { if (c.phase == 0) nA = true; else if (c.phase != 1) nE = true; }
  for (auto& f : b.fops) if (f.early) early = true;
  return nsteps && p->head_eligible && nA && !nE && !early && std::min(std::min(head_rows_wanted(nsteps), p->max_batch), nsteps) == nsteps;
}
// The head rows' row-hash jobs evaluated AHEAD of the fold call over exactly these inputs (same pointer): the call's head batch finds
// them done, and jobvals_host holds these rows' digests meanwhile (what vimz_prover_row_digests would return) — the segments of a short
// call hash every row once instead of twice (merge.hip: vimz_ivc_fold_segments).  Caller holds the lock, device set.
static int head_precompute(vimz_prover* p, const uint64_t* inputs, size_t rows) {
  int rc = head_reserve(p, rows);
  if (rc) return rc;
  head_rowhash_jobs(p, inputs, rows);
  p->pre_inputs = inputs; p->pre_rows = rows;
  return VIMZ_OK;
}

static int fold_head_batch(vimz_prover* p, FoldJob& J, size_t rows) {
  vimz_ctx* ctx = p->ctx;
  const cb::Builder& b = p->circuit->build->b;
  const WitnessDev& W = p->wd;
  const size_t jstride = p->n_jobs + p->n_fops, stage_row = p->job_stage_off.back(), nw = p->n_wires, nc = p->n_c, sw = p->step_wires;
  const bool pre = J.pre_rows >= rows && rows <= p->head_rows_cap;      // (evaluated ahead: head_precompute)
  { int rc = head_reserve(p, rows); if (rc) return rc; }
  auto& bb = p->buf[0];
  bb.gen++;      // (row flags of this filling: wait_row_flag)
  hipStream_t sh = p->sH;
  static const bool dbg_t = getenv("VIMZ_DEBUG_TIMING") != nullptr;
  const double th0 = now_s();
  // the state-independent GPU part of these rows starts now, under the host's Poseidon work
  P_TRY(hipStreamWaitEvent(sh, bb.wit_done, 0));          // (recorded by the caller behind the upload of the private inputs)
  P_TRY(hipMemsetAsync(bb.status, 0, 4 * rows, sh));
  for (uint32_t gI = 0; gI < W.n_decomp; gI++) {
    const uint32_t total = (b.decomp[gI].nbits - 1) * b.decomp[gI].count;
    hipLaunchKernelGGL(k_wit_decomp, dim3((total + 255) / 256, (unsigned)rows), dim3(256), 0, sh, W, gI, (const uint32_t*)p->priv_all_d, bb.Z, bb.status);
  }
  // 1. row-hash chains (phase A) of every head row, one task per (row, chain) — unless they were evaluated ahead
  Fe* stage = p->stage_host; Fe* jobvals = p->jobvals_host;
  const uint64_t* inputs = J.step_inputs;
  if (!pre) head_rowhash_jobs(p, inputs, rows);
  const double th1 = now_s();
  // 2. the IVC state chain of these rows, with the wires of the state hashes
  host_state_chain(p, inputs, rows, jobvals, jstride, J.zs, 0, stage, stage_row, jobvals);
  const double th2 = now_s();
  for (size_t i = 0; i < (rows + 1) * (size_t)p->len_z; i++) p->zs_host[i] = Fe::from_mont(J.zs[i]);
  // 3. upload, scatter, the rest of the witness, then per row (A,B,C)·z and the witness commitment
  P_TRY(upload_pinned(sh, p->zs_all_d, p->zs_host, 32 * (rows + 1) * (size_t)p->len_z));
  P_TRY(upload_pinned(sh, p->stage_d, stage, 32 * rows * stage_row));
  P_TRY(upload_pinned(sh, p->jobvals_d, jobvals, 32 * rows * jstride));
  const double th3 = now_s();
  const unsigned R = (unsigned)rows;
  hipLaunchKernelGGL(k_wit_inputs, dim3(((1 + 2 * p->len_z + p->n_priv) + 255) / 256, R), dim3(256), 0, sh, W, (const uint32_t*)p->priv_all_d, (const uint32_t*)p->zs_all_d, bb.Z, 0u);
  for (uint32_t gI = 0; gI < W.n_groups; gI++)
    hipLaunchKernelGGL(k_wit_lanes, dim3((b.lane_groups[gI].lanes + LANE_TB - 1) / LANE_TB, R), dim3(LANE_TB), 0, sh, W, gI, (const uint32_t*)p->priv_all_d, (const uint32_t*)p->zs_all_d, 0u, bb.Z, bb.status);
  if (p->n_jobs) hipLaunchKernelGGL(k_wit_scatter, dim3(p->n_jobs, R), dim3(128), 0, sh, W, p->job_stage_off_d, (const uint32_t*)p->stage_d, (uint32_t)stage_row, (const uint32_t*)p->jobvals_d, bb.Z, bb.job_out);
  if (p->n_fops) hipLaunchKernelGGL(k_wit_fops, dim3((R + 63) / 64), dim3(64), 0, sh, W, bb.Z, bb.job_out, (uint32_t)rows, 0u);
  P_TRY(hipGetLastError());
  const double th4 = now_s();
  P_TRY(copy_pinned(sh, bb.status_host, bb.status, 4 * rows));
  P_TRY(hipEventRecord(bb.wit_done, sh));
  const double th5 = now_s();
  for (size_t r = 0; r < rows; r++) {
    const uint32_t* Zi = bb.Z + 8 * r * nw;
    launch_spmv(p, sh, Zi, bb.az + 8 * r * nc, bb.bz + 8 * r * nc, bb.cz + 8 * r * nc, 1);
    if (p->ivc) P_TRY(hipEventRecord(bb.ev_p[r], sh));
    if (p->want_s1) P_TRY(ones_launch<BnG1>(sh, p->wsH, p->ck->d, bb.az + 8 * r * nc, p->n_bool, 1, (char*)bb.pin + r * FoldJob::pin_stride + vimz_prover::S1_SLOT));
    P_TRY(msm_launch<BnG1>(sh, p->wsH, p->ck->d, Zi + 8 * (size_t)p->c0, sw - p->c0, 1, 0, (char*)bb.pin + r * FoldJob::pin_stride, &p->planB, nullptr, 1, p->ck->tables ? &J.tbl : nullptr));
    P_TRY(mark_row(sh, bb, r));
    P_TRY(hipEventRecord(bb.ev[r], sh));
    { int rc = fold_issue_d(p, J, 0, r, sh, p->wsH); if (rc) return rc; }
  }
  P_TRY(hipEventRecord(p->ev_head, sh));
  if (dbg_t) fprintf(stderr, "[timing] head batch of %zu rows: launches + row-hash chains on the pool %.2f ms, state chain %.2f ms, uploads %.2f, witness launches %.2f, status copy + event %.2f, per-row spmv + msm launches %.2f ms\n", rows, 1e3 * (th1 - th0), 1e3 * (th2 - th1), 1e3 * (th3 - th2), 1e3 * (th4 - th3), 1e3 * (th5 - th4), 1e3 * (now_s() - th5));
  return VIMZ_OK;
}

// Stage 0 (caller holds the lock, device set): every private input to HBM; the row hashes (phase-A Poseidon chains) of ALL rows
// — they depend on the row data only —; the host then runs the IVC state chain z_0..z_n and uploads it.
// start_batch0 (the fold calls): the first batch is issued from here — as a host-evaluated head batch where the circuit allows it
// (fold_head_batch), in which case the states of the remaining rows are finished by a helper thread (fold_states_wait) —, else
// with its state-independent part started ahead.
// z_cur was replaced as the state the NEXT rows start from before any row was folded (reset; a deferred start): what hangs on it
static void prover_start_state_changed(vimz_prover* p) {
  p->z0 = p->z_cur;
  p->zdigest = Fe::zero();
  for (uint32_t i = 0; i < p->len_z; i++) { Fe in[2] = {p->zdigest, p->z_cur[i]}; p->zdigest = cb::poseidon_hash(in, 2); }
}
// VIMZ_DEBUG_TIMING: where a fold call's prologue spends its time, on one clock for all the provers of a process
static inline void dbg_stamp(const vimz_prover* p, const char* what) {
  static const bool on = getenv("VIMZ_DEBUG_TIMING") != nullptr;
  static const double t_origin = now_s();
  if (on) fprintf(stderr, "[timing] %p %10.3f ms  %s\n", (const void*)p, 1e3 * (now_s() - t_origin), what);
}
static int vz_prover_preload_inputs(vimz_prover* p, const uint64_t* step_inputs, size_t nsteps) {
  vimz_ctx* ctx = p->ctx;
  if (!nsteps || !step_inputs) return VIMZ_OK;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  P_TRY(grow(p->retired, &p->priv_all_d, &p->cap_priv_all, 32 * nsteps * (size_t)p->n_priv, 32 * 1024 * (size_t)p->n_priv));
  P_TRY(hipMemcpyAsync(p->priv_all_d, step_inputs, 32 * nsteps * (size_t)p->n_priv, hipMemcpyHostToDevice, ctx->stream));
  P_TRY(hipStreamSynchronize(ctx->stream));
  p->preloaded_inputs = step_inputs; p->preloaded_rows = nsteps;
  return VIMZ_OK;
}

static int fold_prepare(vimz_prover* p, FoldJob& J, bool start_batch0 = false) {
  vimz_ctx* ctx = p->ctx;
  dbg_stamp(p, "fold_prepare: enter");
  hipStream_t s = ctx->stream;
  const cb::Builder& b = p->circuit->build->b;
  const WitnessDev& W = p->wd;
  const size_t nsteps = J.nsteps, jstride = p->n_jobs + p->n_fops, B = p->max_batch, sw = p->step_wires;
  // (a head evaluated ahead belongs to the one call that follows it, over the very same input buffer: taken over or dropped here)
  J.pre_rows = p->pre_inputs && p->pre_inputs == J.step_inputs ? p->pre_rows : 0;
  p->pre_rows = 0; p->pre_inputs = nullptr;
  for (auto& c : b.chains) (c.phase == 0 ? J.nA : c.phase == 1 ? J.nB : J.nE)++;
  for (auto& f : b.fops) if (f.early) J.early_fops = true;
  struct EndGuard { StartLink* l; ~EndGuard() { if (l) l->fail(); } } end_guard{p->end_to};      // (a call that stops early must not leave its successor waiting)
  if ((p->start_from || p->end_to || p->on_digests) && (J.witnesses || !start_batch0 || J.nE || J.early_fops)) return vz_fail(ctx, VIMZ_ERR_INVALID, "fold: a deferred start state needs the plain witness schedule");
  J.zs.assign((nsteps + 1) * p->len_z, Fe::zero());
  std::vector<Fe>& zs = J.zs;
  for (uint32_t i = 0; i < p->len_z; i++) zs[i] = p->z_cur[i];
  J.tbl = p->ck->tb(0);
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
    plan_batches(J, 0, nsteps, B);
    J.nbatches = J.bfirst.size();
    J.states_upto = nsteps;
    return VIMZ_OK;
  }
  P_TRY(grow(p->retired, &p->priv_all_d, &p->cap_priv_all, 32 * nsteps * (size_t)p->n_priv, 32 * 1024 * (size_t)p->n_priv));
  P_TRY(grow(p->retired, &p->zs_all_d, &p->cap_zs_all, 32 * (nsteps + 1) * (size_t)p->len_z, 32 * 1025 * (size_t)p->len_z));
  P_TRY(grow(p->retired, &p->job_all_d, &p->cap_job_all, 32 * nsteps * jstride, 32 * 1024 * jstride));
  dbg_stamp(p, "fold_prepare: buffers grown");
  if (p->preloaded_inputs == J.step_inputs && p->preloaded_rows == nsteps) { p->preloaded_inputs = nullptr; p->preloaded_rows = 0; }      // (uploaded ahead by the caller that starts several segments together)
  else { p->preloaded_inputs = nullptr; P_TRY(hipMemcpyAsync(p->priv_all_d, J.step_inputs, 32 * nsteps * (size_t)p->n_priv, hipMemcpyHostToDevice, s)); }
  dbg_stamp(p, "fold_prepare: inputs upload queued");
  P_TRY(hipMemsetAsync(p->job_all_d, 0, 32 * nsteps * jstride, s));
  { static const bool dbg_t = getenv("VIMZ_DEBUG_TIMING") != nullptr; if (dbg_t) fprintf(stderr, "[timing] prepare: buffers + upload of the inputs %.2f ms\n", 1e3 * (now_s() - t0)); }
  const bool plain = J.nA && !J.nE && !J.early_fops;           // no ahead-of-time witness pass needed (everything but crop)
  const bool deferred = p->start_from || p->end_to || p->on_digests;      // (vimz_prover: deferred start state — needs the plain schedule: the head batch wants the state at once)
  const size_t head = start_batch0 && plain && p->head_eligible && !deferred && !p->suppress_head ? std::min(std::min(head_rows_wanted(nsteps), B), nsteps) : 0;
  p->last_head_rows = head;
  if (head) {
    J.head = true;
    plan_batches(J, 0, head, B);
    plan_batches(J, head, nsteps - head, B);
  } else plan_batches(J, 0, nsteps, B);
  J.nbatches = J.bfirst.size();
  // Both the hash-only pass and a batch's witness are one Poseidon-chain latency long (≈10 ms, whatever the row count).  For one
  // batch — the first GPU-produced one — the chains therefore run once, WITH their wires, into its batch buffer on the producer's
  // stream, side by side with the hash-only pass of the rows after it; fold_issue adds the rest of that batch's witness.
  size_t hashed_from = 0;             // rows before this one get their row hashes elsewhere (host / started batch)
  if (start_batch0 && plain) {
    const size_t kb = head ? 1 : 0;
    if (kb < J.nbatches) {
      auto& bk = p->buf[kb & 1];
      const size_t f = J.first(kb), n = J.rows(kb);
      P_TRY(hipEventRecord(bk.wit_done, s));
      P_TRY(hipStreamWaitEvent(p->sB, bk.wit_done, 0));
      P_TRY(hipMemsetAsync(bk.status, 0, 4 * n, p->sB));
      int rc = launch_witness(p, p->sB, bk.Z, bk.job_out, bk.status, p->priv_all_d + 8 * f * p->n_priv, f, n, J, false, 1);
      if (rc) return rc;
      P_TRY(hipMemcpyAsync(p->job_all_d + 8 * f * jstride, bk.job_out, 32 * n * jstride, hipMemcpyDeviceToDevice, p->sB));
      P_TRY(hipEventRecord(bk.wit_done, p->sB));
      J.started_batch = (int)kb;
      hashed_from = f + n;
    } else hashed_from = nsteps;
    if (head) P_TRY(hipEventRecord(p->buf[0].wit_done, s));      // the head batch's GPU work waits for the upload of the inputs
  }
  // hash-only pass over the remaining rows: behind the started batch's chains on the producer's stream when there is a head batch
  // (the main stream must stay free for the first folds), on the main stream otherwise
  hipStream_t sh = head ? p->sB : s;
  for (size_t off = hashed_from; off < nsteps && J.nA; off += 32768) {
    const unsigned rows = (unsigned)std::min<size_t>(32768, nsteps - off);
    hipLaunchKernelGGL(k_wit_chains, dim3((J.nA + 3) / 4, rows), dim3(64), 0, sh, W, 0u, (uint32_t*)nullptr, p->job_all_d + 8 * off * jstride,
                       (const uint32_t*)(p->priv_all_d + 8 * off * p->n_priv));
  }
  P_TRY(hipGetLastError());
  if (head) {
    int rc = fold_head_batch(p, J, head);
    if (rc) return rc;
    J.next_issue = 1;
    p->phase_s[PH_WITNESS] += now_s() - t0; p->phase_n[PH_WITNESS] += head;
    J.states_upto = head;
    if (head == nsteps) return VIMZ_OK;
    // the states of the remaining rows: a helper thread waits for their row hashes (first those of the started batch, ready one
    // chain latency earlier than the rest), runs the host chain and uploads the states; fold_issue_when_ready() looks at
    // states_upto before a GPU-produced batch is issued.  (Every blocking call — the pageable download above all — lives in the
    // helper: issued from here it held up the first fold by a whole chain latency.)
    J.states_upto = head;
    const size_t mid = hashed_from;           // rows [head, mid): hashed by the started batch (ev_head2); [mid, nsteps): by the hash-only pass
    P_TRY(hipEventRecord(p->ev_hash, p->sB));
    J.jobA_rest.resize((nsteps - head) * jstride);
    J.helper_done = 0;
    hipEvent_t ev_mid = J.started_batch >= 0 ? p->buf[J.started_batch & 1].wit_done : p->ev_hash;
    J.helper = std::thread([p, &J, head, mid, nsteps, jstride, ev_mid] {
      vimz_ctx* ctx = p->ctx;
      auto failed = [&](const char* what, hipError_t e) { J.helper_rc = VIMZ_ERR_HIP; J.helper_err = std::string(what) + ": " + hipGetErrorString(e); J.helper_done = 1; };
      hipError_t e = hipSetDevice(ctx->device);
      if (e != hipSuccess) return failed("helper: hipSetDevice", e);
      hipStream_t sx = nullptr;            // own stream: nothing of the fold is queued behind these copies
      int plo = 0, phi = 0; hipDeviceGetStreamPriorityRange(&plo, &phi); const int pmid = (plo + phi) / 2;
      if ((e = vz_stream_acquire(ctx, pmid, &sx)) != hipSuccess) return failed("helper: stream", e);      // (recycled: no stream churn per fold call)
      const size_t cuts[3] = {head, mid, nsteps};
      hipEvent_t evs[2] = {ev_mid, p->ev_hash};
      for (int part = 0; part < 2; part++) {
        const size_t lo = cuts[part], hi = cuts[part + 1];
        if (hi <= lo) continue;
        if ((e = hipEventSynchronize(evs[part])) != hipSuccess) { vz_stream_release(ctx, pmid, sx); return failed("helper: row hashes", e); }
        const double t1 = now_s();
        Fe* ja = J.jobA_rest.data() + (lo - head) * jstride;
        if ((e = hipMemcpyAsync(ja, p->job_all_d + 8 * lo * jstride, 32 * (hi - lo) * jstride, hipMemcpyDeviceToHost, sx)) != hipSuccess ||
            (e = hipStreamSynchronize(sx)) != hipSuccess) { vz_stream_release(ctx, pmid, sx); return failed("helper: download", e); }
        host_state_chain(p, J.step_inputs + 4 * lo * (size_t)p->n_priv, hi - lo, ja, jstride, J.zs, lo);
        std::vector<Fe> zc((hi - lo) * (size_t)p->len_z);
        for (size_t i = 0; i < zc.size(); i++) zc[i] = Fe::from_mont(J.zs[(lo + 1) * p->len_z + i]);
        if ((e = hipMemcpyAsync(p->zs_all_d + 8 * (lo + 1) * (size_t)p->len_z, zc.data(), 32 * zc.size(), hipMemcpyHostToDevice, sx)) != hipSuccess ||
            (e = hipStreamSynchronize(sx)) != hipSuccess) { vz_stream_release(ctx, pmid, sx); return failed("helper: upload", e); }
        p->phase_s[PH_ZCHAIN] += now_s() - t1; p->phase_n[PH_ZCHAIN] += hi - lo;
        J.states_upto.store(hi, std::memory_order_release);
      }
      vz_stream_release(ctx, pmid, sx);
      J.helper_done = 1;
    });
    return VIMZ_OK;
  }
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
  // (waited for on the HOST: a barrier on this high-priority stream behind the producer's low-priority event is the inversion of DESIGN.md §5c —
  //  the stalled barrier keeps the low-priority queue from being served, and with four processes on one GPU one of them then ran at 60 ms per step)
  dbg_stamp(p, "fold_prepare: first batch's chains launched");
  if (J.started_batch >= 0) P_TRY(hipEventSynchronize(p->buf[J.started_batch & 1].wit_done));
  dbg_stamp(p, "fold_prepare: chains done");
  P_TRY(hipMemcpyAsync(jobA.data(), p->job_all_d, 32 * nsteps * jstride, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  dbg_stamp(p, "fold_prepare: row hashes on the host");
  if (p->on_digests) p->on_digests(jobA.data(), nsteps, jstride);
  p->phase_s[PH_WITNESS] += now_s() - t0; t0 = now_s();
  if (p->start_from) {      // the state this call starts from: the predecessor segment's end state (it follows from ITS row hashes, ready about now)
    std::vector<Fe> zst;
    if (!p->start_from->wait(zst) || zst.size() != p->len_z) return vz_fail(ctx, VIMZ_ERR_INVALID, "fold: the segment before this one failed before its end state was known");
    for (uint32_t i = 0; i < p->len_z; i++) { p->z_cur[i] = zst[i]; zs[i] = zst[i]; }
    prover_start_state_changed(p);
    if (p->on_start) p->on_start();
    p->phase_s[PH_ZCHAIN] += now_s() - t0;
    dbg_stamp(p, "fold_prepare: start state arrived");
  }
  host_state_chain(p, J.step_inputs, nsteps, jobA.data(), jstride, zs);
  if (p->end_to) { p->end_to->publish(zs.data() + nsteps * (size_t)p->len_z, p->len_z); end_guard.l = nullptr; }
  dbg_stamp(p, "fold_prepare: state chain done");
  {
    std::vector<Fe> zc(zs.size());
    for (size_t i = 0; i < zs.size(); i++) zc[i] = Fe::from_mont(zs[i]);
    P_TRY(hipMemcpyAsync(p->zs_all_d, zc.data(), 32 * zc.size(), hipMemcpyHostToDevice, s));
    P_TRY(hipStreamSynchronize(s));
  }
  p->phase_s[PH_ZCHAIN] += now_s() - t0; p->phase_n[PH_ZCHAIN] += nsteps; p->phase_n[PH_WITNESS] += nsteps;
  J.states_upto = nsteps;
  dbg_stamp(p, "fold_prepare: states uploaded, return");
  return VIMZ_OK;
}

// Hand batch k to the producer as soon as the IVC states of its rows exist (called at the top of batch k-1 and then once per row
// until it has happened; wait = true before batch k is consumed).
static int fold_issue_when_ready(vimz_prover* p, FoldJob& J, size_t k, bool wait) {
  if (k >= J.nbatches || J.next_issue > k) return VIMZ_OK;
  const size_t need = J.first(k) + J.rows(k);
  const double t_wait = now_s(); uint64_t spins = 0;
  while (J.states_upto.load(std::memory_order_acquire) < need) {
    if (J.helper_done.load() && J.states_upto.load() < need) {       // the helper stopped early
      if (J.helper.joinable()) J.helper.join();
      p->ctx->err = J.helper_err.empty() ? std::string("fold: the state chain helper stopped early") : J.helper_err;
      return J.helper_rc ? J.helper_rc : VIMZ_ERR_HIP;
    }
    if (!wait) return VIMZ_OK;
    std::this_thread::yield();
    if ((++spins & 0xfffff) == 0 && now_s() - t_wait > 10.0) {      // a stuck helper must not hang the caller silently
      fprintf(stderr, "[vimz] fold: waiting for the IVC states of batch %zu for %.0f s (states for %zu rows, need %zu, helper done %d)\n", k, now_s() - t_wait,
              J.states_upto.load(), need, J.helper_done.load());
      if (now_s() - t_wait > 120.0) return vz_fail(p->ctx, VIMZ_ERR_HIP, "fold: the state chain helper did not finish within 120 s");
    }
  }
  int rc = fold_issue(p, J, k);
  if (rc) return rc;
  J.next_issue = k + 1;
  return VIMZ_OK;
}

// Producer: batch k on the low-priority stream — full witness generation of the step circuit, then per row the step rows of
// (A,B,C)·z and the commitment to the step circuit's wires.  Records bb.wit_done and one event per row.
static int fold_issue(vimz_prover* p, const FoldJob& J, size_t k) {
  vimz_ctx* ctx = p->ctx;
  const size_t nw = p->n_wires, nc = p->n_c, sw = p->step_wires;
  auto& bb = p->buf[k & 1];
  bb.gen++;      // (row flags of this filling: wait_row_flag)
  const size_t first = J.first(k), rows = J.rows(k);
  hipStream_t sb = p->sB;
  if (J.head && k == 1) P_TRY(hipStreamWaitEvent(sb, p->ev_head, 0));      // (orders the two MSM workspaces' users; the head batch is long done)
  // this batch overwrites the buffer of batch k-2, whose last row the first row of batch k-1 was differenced against (fold_issue_d)
  if (p->want_d && k >= 1 && p->buf[(k - 1) & 1].has_d[0]) { P_TRY(hipStreamWaitEvent(sb, p->buf[(k - 1) & 1].ev_d[0], 0)); P_TRY(hipStreamWaitEvent(p->sH, p->buf[(k - 1) & 1].ev_d[0], 0)); }
  const bool started = (int)k == J.started_batch;      // (its status words already hold the decompositions' range checks)
  if (!started) P_TRY(hipMemsetAsync(bb.status, 0, 4 * rows, sb));
  if (J.witnesses) {
    P_TRY(hipMemcpy2DAsync(bb.Z, 32 * nw, J.witnesses + 4 * first * sw, 32 * sw, 32 * sw, rows, hipMemcpyHostToDevice, sb));
    launch_to_mont<Fr>(sb, bb.Z, rows * nw);
  } else {
    int rc = launch_witness(p, sb, bb.Z, bb.job_out, bb.status, p->priv_all_d + 8 * first * p->n_priv, first, rows, J, false, started ? 2 : 0);
    if (rc) return rc;
  }
  P_TRY(hipGetLastError());
  P_TRY(copy_pinned(sb, bb.status_host, bb.status, 4 * rows));
  P_TRY(hipEventRecord(bb.wit_done, sb));
  // Per row: the step rows of (A,B,C)·z, the commitment to the witness and (lookahead) the fresh x fresh commitment — a chain of some
  // twenty-five dependent launches, ≈1 ms (1.6 ms with the lookahead's) whatever the GPU has free.  In IVC mode the rows alternate
  // between the producer's stream and the head batch's (idle after the first rows of a call), each with its own MSM workspace.
  const bool two = p->ivc && p->sH && p->sH != sb;
  if (two) P_TRY(hipStreamWaitEvent(p->sH, bb.wit_done, 0));
  for (size_t r = 0; r < rows; r++) {
    const uint32_t* Zi = bb.Z + 8 * r * nw;
    hipStream_t st = two && (r & 1) ? p->sH : sb;
    MsmWorkspace& ws = two && (r & 1) ? p->wsH : p->wsB;
    launch_spmv(p, st, Zi, bb.az + 8 * r * nc, bb.bz + 8 * r * nc, bb.cz + 8 * r * nc, 1);
    if (p->ivc) P_TRY(hipEventRecord(bb.ev_p[r], st));
    if (p->want_s1) P_TRY(ones_launch<BnG1>(st, ws, p->ck->d, bb.az + 8 * r * nc, p->n_bool, 1, (char*)bb.pin + r * FoldJob::pin_stride + vimz_prover::S1_SLOT));
    P_TRY(msm_launch<BnG1>(st, ws, p->ck->d, Zi + 8 * (size_t)p->c0, sw - p->c0, 1, 0, (char*)bb.pin + r * FoldJob::pin_stride, &p->planB, nullptr, 1, p->ck->tables ? &J.tbl : nullptr));
    P_TRY(mark_row(st, bb, r));
    P_TRY(hipEventRecord(bb.ev[r], st));
    { int rc = fold_issue_d(p, J, k, r, st, ws); if (rc) return rc; }
  }
  return VIMZ_OK;
}
