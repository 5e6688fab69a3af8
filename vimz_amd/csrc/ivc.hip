// Nova IVC on the BN254 / Grumpkin cycle: RecursiveSNARK::{new, prove_step, verify} in full (SURVEY.md §8a rows S1, S2, X1;
// reference entry points vimz/src/nova_snark_backend/folding.rs:27-56 through nova_scotia::create_recursive_circuit).
//
// Per step i (z_i -> z_{i+1}), following nova-snark's prove_step order:
//   1. NIFS on the secondary curve: fold the previous fresh secondary instance u2 into the running U2
//      (cross term and its commitment MSM(T2) were queued on the GPU as soon as u2's witness existed);
//   2. primary augmented circuit: the step circuit's part of the witness, its commitment and (A,B,C)·z come from the batch
//      producer (prover_internal.hpp); the verifier circuit's wires (aug/circuit.hpp: checks u2's hash, derives the challenge,
//      folds U2 <- U2 + rho·u2 in-circuit, hashes the result) are computed on the host, uploaded, committed (MSM over the
//      verifier slice of ck) and multiplied (verifier rows of A,B,C) on the GPU  ->  fresh primary instance u1;
//   3. NIFS on the primary curve: cross term, MSM(T1), fused fold of W, E, Az, Bz, Cz;
//   4. secondary augmented circuit (7.6 k constraints over BN254 Fq): host witness, then SpMV / MSM(W2) / cross term /
//      MSM(T2) on the GPU with the same kernels instantiated for Fq / Grumpkin.
// The challenges are the ones the circuits derive (rho = 2^128 + low 128 bits of a Poseidon hash), so the folded instances
// the prover holds are exactly the ones the circuits compute: vimz_ivc_verify checks that.
//
// Schedule: streams 1 and 2 (high priority) carry the two halves of a step, stream 3 the one large MSM — the step
// rows' cross term, written by k_fold_cross together with the fold of the step rows and queued from INSIDE the secondary circuit's
// evaluation (AugCircuit::on_challenge) —, two low-priority streams the batch producer, whose launches come from an issuer thread.
// Option VIMZ_IVC_LOOKAHEAD=1: the cross term of step i+2 against the running instance of step i+1 (T1Slot, fold_issue_d).
#include "ivc_internal.hpp"

namespace {

// read the results of the queued secondary MSMs (comm_W of the fresh instance, comm_T of its fold)
int finish_secondary(vimz_ivc* v) {
  if (!v->pending_sec) return VIMZ_OK;
  vimz_ctx* ctx = v->ctx;
  double t0 = now_s();
  P_TRY(vz_wait_stream(v->s2));
  v->u2.W = msm_finish<Grumpkin>(v->plan_W2, v->pin + 2 * v->pin_res);      // overlaps the rest of MSM(T2)
  P_TRY(vz_wait_stream(ctx->stream));
  v->ph_s[IP_WAIT_SEC] += now_s() - t0; v->ph_n[IP_WAIT_SEC]++;
  if (ctx->profiling) { float ms = 0; if (hipEventElapsedTime(&ms, v->ev_b0, v->ev_b1) == hipSuccess) { v->ph_s[IP_RESERVED] += ms * 1e-3; v->ph_n[IP_RESERVED]++; } }
  if (v->sec_T_valid) v->T2 = msm_finish<Grumpkin>(v->plan_T2, v->pin + 3 * v->pin_res);
  else { v->T2.x = Fe::zero(); v->T2.y = Fe::zero(); }
  static const bool dbg = getenv("VIMZ_DEBUG_CHECK_MSM") != nullptr;
  if (dbg) {   // recompute both on the main stream, alone
    G2Aff r; MsmStats st;
    P_TRY(msm_run<Grumpkin>(ctx->stream, ctx->msm_ws, v->ck2->d, v->sec.z2 + 8, v->sec.n_w - 3, 1, 0, &r, &st, nullptr, 0, nullptr));
    if (!r.x.eq(v->u2.W.x) || !r.y.eq(v->u2.W.y)) fprintf(stderr, "[dbg] step %llu: MSM(W2) on stream 2 differs from the recomputation\n", (unsigned long long)v->i);
    if (v->sec_T_valid) {
      P_TRY(msm_run<Grumpkin>(ctx->stream, ctx->msm_ws, v->ck2->d, v->sec.T, v->sec.n_c, 1, 0, &r, &st, nullptr, 0, nullptr));
      if (!r.x.eq(v->T2.x) || !r.y.eq(v->T2.y)) fprintf(stderr, "[dbg] step %llu: MSM(T2) differs from the recomputation\n", (unsigned long long)v->i);
    }
  }
  v->pending_sec = false;
  return VIMZ_OK;
}

int ivc_fold_core(vimz_ivc* v, const uint64_t* step_inputs, const uint64_t* witnesses, size_t nsteps) {
  if (!nsteps) return VIMZ_OK;
  vimz_ctx* ctx = v->ctx;
  if (v->broken) return vz_fail(ctx, VIMZ_ERR_INVALID, "this IVC failed in the middle of a step and cannot be folded any further");
  vimz_prover* p = v->pri;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t nw = p->n_wires, nc = p->n_c, sw = p->step_wires, sc = p->step_c;
  const size_t aw1 = v->c1->aug_wires();
  SecDev& S = v->sec;
  int rc;
  const double t_all = now_s();
  FoldJob job; job.step_inputs = step_inputs; job.witnesses = witnesses; job.nsteps = nsteps;
  dbg_stamp(p, "ivc fold: lock taken");
  if ((rc = fold_prepare(p, job, true))) return rc;
  // Any return between here and the end that does not first restore a consistent state (rows folded so far, no cross term queued
  // ahead, state = that of the last folded row) leaves work of a half-done step behind: such an IVC refuses further folds.
  struct BrokenGuard { vimz_ivc* v; bool armed = true; ~BrokenGuard() { if (armed) v->broken = true; } } guard{v};
  auto settle = [&](size_t rows_done) {      // the rows folded so far stay folded: drop what was queued ahead, finish the pending secondary commitments
    hipStreamSynchronize(p->sB);
    if (p->sD) hipStreamSynchronize(p->sD);
    hipStreamSynchronize(v->s3); v->t1[0].step = v->t1[1].step = -1;
    const int rc2 = finish_secondary(v);
    for (uint32_t q = 0; q < p->len_z; q++) p->z_cur[q] = job.zs[rows_done * p->len_z + q];
    if (!rc2) guard.armed = false;
  };
  static const bool dbg_timing = getenv("VIMZ_DEBUG_TIMING") != nullptr;
  const double t_prep = now_s() - t_all;
  double t_first = 0, t_wait0 = 0, t_hook = 0, t_hk[5] = {0, 0, 0, 0, 0}, t_wp[5] = {0, 0, 0, 0, 0};      // (t_wp: the parts of wait_primary_msm)      // (t_hk: fused fold launch, event records, large-MSM launches, fold5 launch, row-flag waits)
  const std::vector<Fe>& zs = job.zs;
  const size_t pin_stride = FoldJob::pin_stride;
  char* pin_aug1 = v->pin + 5 * v->pin_res;
  char* pin_w2 = pin_aug1 + 32 * aw1;
  const Fq zero_q = Fq::zero();

  // ---- the step rows' cross term and its commitment (the one large MSM of a step), on stream 3 ------------------------------------
  // (a helper thread for these launches measured no gain once the producer had its issuer thread, and costs a spinning core)
  typedef vimz_prover::BatchBuf BufRef;
  // Boolean-row form of the step rows' commitment (r1cs_ops.hpp: bool_row_masked): needs the producer's S_1 per row and C_A of the running instance.
  // Not with MSM helpers (base-range split) and not over caller-supplied witnesses (the producer's own rows have bits on the boolean rows by construction).
  const bool trick = v->bool_rows && v->helpers.empty() && !witnesses;
  if (trick && v->i > 0 && !v->ca_valid) {      // (after an import: C_A = Σ_{i<n_bool} AZ[i]·ck_i by one MSM)
    G1Aff r; MsmStats st;
    P_TRY(hipStreamSynchronize(v->s3));
    P_TRY(msm_run<BnG1>(s, ctx->msm_ws, p->ck->d, p->AZ, p->n_bool, 1, 0, &r, &st, nullptr, 0, p->ck->tables ? &job.tbl : nullptr));
    v->CA = from_affine(r);
  }
  v->ca_valid = false;      // (valid again when this call ends with C_A kept up to its last fold)
  G1Aff S1_row; S1_row.x = S1_row.y = Fq::zero();      // S_1 of the row being folded: Σ ck_i over the boolean rows whose fresh bit is one (producer, pinned)
  bool pend_ca = false; uint32_t pend_rho[4] = {0, 0, 0, 0}; G1Aff pend_S1 = S1_row; int pend_slot[2] = {-1, -1};
  auto apply_pending_ca = [&]() {
    if (!pend_ca) return;
    if (!aff_is_identity(pend_S1)) { const uint32_t k[5] = {pend_rho[0], pend_rho[1], pend_rho[2], pend_rho[3], 1u}; G1 t = scalar_mul(pend_S1, k, 129); add_full(v->CA, t); }
    for (int q : pend_slot) if (q >= 0) v->t1[q].ca_at = v->CA;
    pend_ca = false;
  };
  // commitment to the vector of slot `par` (base-range split: the first share stays here, helper h commits to rows [off_h, off_h + n_h)
  // with its replica of the key)
  auto queue_msm = [&](int par) -> int {
    auto& S = v->t1[par];
    const size_t parts = v->helpers.size() + 1, share = (sc + parts - 1) / parts;
    v->t1_main_n = std::min(share, sc);
    if (!v->helpers.empty()) {
      P_TRY(hipEventRecord(v->ev_T, v->s3));
      size_t off = v->t1_main_n;
      for (auto& h : v->helpers) {
        h.off = off; h.n = off < sc ? std::min(share, sc - off) : 0; off += h.n;
        if (!h.n) continue;
        P_TRY(hipSetDevice(h.ctx->device));
        P_TRY(hipStreamWaitEvent(h.s, v->ev_T, 0));
        P_TRY(hipMemcpyAsync(h.T, S.buf + 8 * h.off, 32 * h.n, hipMemcpyDefault, h.s));
        P_TRY(msm_launch<BnG1>(h.s, h.ws, h.ck->d + (size_t)AFFINE_WORDS * h.off, h.T, h.n, 1, 0, h.pin, &h.plan, nullptr, 0, nullptr));
      }
      P_TRY(hipSetDevice(ctx->device));
    }
    double tq = now_s();
    P_TRY(msm_launch<BnG1>(v->s3, v->ws3, p->ck->d, S.tricked ? S.bufm : S.buf, v->t1_main_n, 1, 0, S.pin, &S.plan, ctx->profiling ? S.ev : nullptr, 0,
                           p->ck->tables && v->helpers.empty() ? &job.tbl : nullptr));
    t_hk[2] += now_s() - tq; tq = now_s();
    if (ctx->profiling) P_TRY(hipMemcpyAsync(S.pin + v->pin_res, v->ws3.totals, 8, hipMemcpyDeviceToHost, v->s3));   // (pinned: stays asynchronous)
    P_TRY(hipEventRecord(S.done, v->s3));
    t_hk[1] += now_s() - tq;
    return VIMZ_OK;
  };
  // the cross term of step `step` = row `row` of `b2` against the running instance as it is now (first row of a call)
  auto launch_direct = [&](BufRef& b2, size_t row, uint64_t step) -> int {
    auto& S = v->t1[step & 1];
    P_TRY(hipEventRecord(v->ev_fold, s));
    P_TRY(hipStreamWaitEvent(v->s3, v->ev_fold, 0));
    P_TRY(wait_row_flag(b2, row, b2.ev[row]));      // (on the host: prover_internal.hpp, wait_row_flag)
    hipLaunchKernelGGL(k_cross_term<Fr>, dim3(stream_grid(sc)), dim3(256), 0, v->s3, sc, p->AZ, p->BZ, p->CZ, v->u1_run,
                       b2.az + 8 * row * nc, b2.bz + 8 * row * nc, b2.cz + 8 * row * nc, Fe::one(), S.buf, trick ? S.bufm : (uint32_t*)nullptr, p->n_bool);
    P_TRY(hipGetLastError());
    S.step = (int64_t)step; S.hasB = false;
    S.tricked = trick; S.u_at = v->u1_run; S.ca_at = v->CA;
    return queue_msm((int)(step & 1));
  };
  // When step i's challenge is known: k_fold_cross folds the step rows (running products, error vector) and writes the cross term(s)
  // that come next —  `next1` (row i+1, against the running instance it will meet: needed unless a lookahead produced it already)
  // and/or `next2` (row i+2 against the running instance of step i+1: the lookahead, completed by −rho_{i+1}·negB_{i+2} when used).
  struct RowAt { BufRef* b; size_t row; };
  struct FoldArgs { uint64_t i; BufRef* cur; size_t r; Fe rho; bool need1; RowAt next1; bool look; RowAt next2; };
  auto launch_fold_and_cross = [&](const FoldArgs& a) -> int {
    const int par = (int)(a.i & 1);
    auto& Scur = v->t1[par]; auto& Soth = v->t1[par ^ 1];
    const bool fold_E = a.i > 0, hasB = fold_E && Scur.hasB;
    const uint32_t *az = a.cur->az + 8 * a.r * nc, *bz = a.cur->bz + 8 * a.r * nc, *cz = a.cur->cz + 8 * a.r * nc;
    if (hasB) P_TRY(wait_row_flag(*a.cur, a.cur->flag_rows + a.r, a.cur->ev_d[a.r]));
    // first output of the pass: next1 into the other slot if it is needed, else the lookahead into this slot (read, then overwritten)
    const RowAt* t = a.need1 ? &a.next1 : a.look ? &a.next2 : nullptr;
    uint32_t* out = a.need1 ? Soth.buf : a.look ? Scur.buf : nullptr;
    uint32_t* outm = !trick ? nullptr : a.need1 ? Soth.bufm : a.look ? Scur.bufm : nullptr;
    double tq = now_s();
    if (t) P_TRY(wait_row_flag(*t->b, t->row, t->b->ev[t->row]));      // (the producer is a batch ahead: no wait in the steady state)
    t_hk[4] += now_s() - tq; tq = now_s();
    hipLaunchKernelGGL(k_fold_cross<Fr>, dim3(stream_grid(sc)), dim3(256), 0, v->s3, sc, p->AZ, p->BZ, p->CZ, p->E, (const uint32_t*)Scur.buf, fold_E ? 1 : 0, a.rho,
                       hasB ? 1 : 0, v->rho_prev, hasB ? (const uint32_t*)(a.cur->d + 8 * a.r * sc) : (const uint32_t*)nullptr, v->u1_run, az, bz, cz,
                       out, t ? t->b->az + 8 * t->row * nc : nullptr, t ? t->b->bz + 8 * t->row * nc : nullptr, t ? t->b->cz + 8 * t->row * nc : nullptr, Fe::one(),
                       outm, p->n_bool);
    P_TRY(hipGetLastError());
    if (a.need1 && a.look) {      // both (the first step of a call): the lookahead as a pass of its own, after this slot's vector was read
      P_TRY(wait_row_flag(*a.next2.b, a.next2.row, a.next2.b->ev[a.next2.row]));
      hipLaunchKernelGGL(k_cross_term<Fr>, dim3(stream_grid(sc)), dim3(256), 0, v->s3, sc, p->AZ, p->BZ, p->CZ, v->u1_run,
                         a.next2.b->az + 8 * a.next2.row * nc, a.next2.b->bz + 8 * a.next2.row * nc, a.next2.b->cz + 8 * a.next2.row * nc, Fe::one(), Scur.buf,
                         trick ? Scur.bufm : (uint32_t*)nullptr, p->n_bool);
      P_TRY(hipGetLastError());
    }
    t_hk[0] += now_s() - tq; tq = now_s();
    P_TRY(hipEventRecord(v->ev_fused, v->s3)); v->fused_recorded = true;
    t_hk[1] += now_s() - tq;
    // (boolean-row form: u_at is the running instance's u after this fold; ca_at follows once C_A has this fold in it — queue_folds, after the launches)
    if (a.need1) { Soth.step = (int64_t)a.i + 1; Soth.hasB = false; Soth.tricked = trick; Soth.u_at = v->u1_run; int rc2 = queue_msm(par ^ 1); if (rc2) return rc2; }
    if (a.look) { Scur.step = (int64_t)a.i + 2; Scur.hasB = true; Scur.tricked = trick; Scur.u_at = v->u1_run; int rc2 = queue_msm(par); if (rc2) return rc2; }
    return VIMZ_OK;
  };
  // ---- the batch producer's launches (some forty per row, sixty with the lookahead's commitment) come from a thread of their own: issued
  // from this one, a whole batch at a time, they held every step up by 0.15-0.23 ms on average.  Batch b may be issued once its IVC
  // states exist and this thread allows it (`allowed` > b: the buffer it overwrites has been folded and the streams told to wait).
  std::atomic<size_t> allowed{0}, issued{job.next_issue};
  std::atomic<int> issuer_rc{VIMZ_OK};
  std::atomic<bool> issuer_stop{false};
  std::thread issuer([&, b0 = job.next_issue] {
    if (hipSetDevice(ctx->device) != hipSuccess) { issuer_rc = VIMZ_ERR_HIP; return; }
    for (size_t b = b0; b < job.nbatches; b++) {
      const size_t need = job.first(b) + job.rows(b);
      while (allowed.load(std::memory_order_acquire) <= b || job.states_upto.load(std::memory_order_acquire) < need) {
        if (issuer_stop.load()) return;
        if (job.helper_done.load() && job.states_upto.load() < need) { issuer_rc = job.helper_rc ? job.helper_rc : VIMZ_ERR_HIP; return; }
        // (a batch is wanted a whole batch of folds after it may be issued: sleeping polls, not a spinning core — the GPU boxes
        // grant a process 16 CPUs and throttle the whole group beyond that)
        std::this_thread::sleep_for(std::chrono::microseconds(100));
      }
      const double _ti = now_s();
      const int rc2 = fold_issue(p, job, b);
      if (dbg_timing) fprintf(stderr, "[timing] producer: batch %zu (%zu rows): ready at %.1f ms, issue took %.1f ms\n", b, job.rows(b), 1e3 * (_ti - t_all), 1e3 * (now_s() - _ti));
      if (rc2) { issuer_rc = rc2; return; }
      issued.store(b + 1, std::memory_order_release);
    }
  });
  struct IssuerJoin { std::thread& t; std::atomic<bool>& stop; ~IssuerJoin() { stop = true; if (t.joinable()) t.join(); } } issuer_join{issuer, issuer_stop};
  auto wait_issued = [&](size_t b) -> int {
    const double t_wait = now_s(); uint64_t spins = 0;
    while (issued.load(std::memory_order_acquire) <= b) {
      const int irc = issuer_rc.load();
      if (irc) { if (ctx->err.empty()) ctx->err = job.helper_err.empty() ? std::string("fold: the batch producer stopped early") : job.helper_err; return irc; }
      std::this_thread::yield();
      if ((++spins & 0xfffff) == 0 && now_s() - t_wait > 120.0) return vz_fail(ctx, VIMZ_ERR_HIP, "fold: batch not issued within 120 s");
    }
    return VIMZ_OK;
  };
  for (size_t k = 0; k < job.nbatches; k++) {
    auto& bb = p->buf[k & 1];
    const size_t first = job.first(k), rows = job.rows(k);
    double t0 = now_s();
    allowed.store(k + 2, std::memory_order_release);       // the next batch is produced while this one is folded
    if ((rc = wait_issued(k))) return rc;                  // (batch 0 of a head-batch call is already out)
    if (k == 0) dbg_stamp(p, "ivc fold: first batch issued");
    P_TRY(vz_wait_event(bb.wit_done));
    v->ph_s[IP_PRODUCER] += now_s() - t0;
    if (k == 0) { t_wait0 = now_s() - t0; t_first = now_s() - t_all; dbg_stamp(p, "ivc fold: first batch's witnesses ready"); }
    for (size_t r = 0; r < rows; r++) if (bb.status_host[r]) {
      char msg[128]; snprintf(msg, sizeof(msg), "step %llu: the step relation is not satisfiable for these rows", (unsigned long long)(v->i + r));
      // the batches folded so far stay folded: leave the IVC consistent at that point (cross terms of this batch's rows were queued ahead)
      settle(first);
      return vz_fail(ctx, VIMZ_ERR_UNSAT, msg);
    }
    for (size_t r = 0; r < rows; r++) {
      const uint64_t i = v->i;
      uint32_t* Zi = bb.Z + 8 * r * nw;
      uint32_t *az = bb.az + 8 * r * nc, *bz = bb.bz + 8 * r * nc, *cz = bb.cz + 8 * r * nc;
      // ---- 1. the previous fresh secondary instance is complete once its two MSMs are back -------------------------------------
      // (while they run: the statement part of this step's output hash, which depends on nothing they produce)
      v->c1->precompute_statement(i + 1, v->z0, zs.data() + (first + r + 1) * p->len_z);
      // Boolean-row form, the host's share — three scalar multiplications of S_1 points per step, done HERE, where the thread would otherwise wait for the
      // previous step's secondary MSMs (0.5 ms): C_A takes in the previous fold (rho_{i-1}·S_1(row i-1)), then this step's completion u·S_1(row i) − C_A
      G1 boolCorr = G1::identity();
      if (trick) {
        apply_pending_ca();
        P_TRY(wait_row_flag(bb, r, bb.ev[r]));      // (the producer is a batch ahead: no wait in the steady state)
        S1_row = to_affine(ones_finish<BnG1>((const char*)bb.pin + r * pin_stride + vimz_prover::S1_SLOT));
        auto& sl = v->t1[i & 1];
        if (i > 0 && sl.step == (int64_t)i && sl.tricked) {
          if (!aff_is_identity(S1_row)) { const Fe uc = Fe::from_mont(sl.u_at); boolCorr = scalar_mul(S1_row, uc.v, 254); }
          G1 nca = sl.ca_at; if (!nca.is_identity()) nca.Y = Fq::neg(nca.Y);
          add_full(boolCorr, nca);
        }
      }
      if ((rc = finish_secondary(v))) return rc;
      // ---- 2. primary verifier circuit on the host: folds (U2, u2) and hashes the result ----------------------------------------
      t0 = now_s();
      AugIn<BnFr> in1; in1.digest = v->c1->digest; in1.z0 = v->z0; in1.i = i; in1.U = v->U2; in1.u = v->u2; in1.T = v->T2;
      std::vector<Fe> aug1; bool bad = false;
      AugOut<BnFr> o1 = v->c1->witness(in1, zs.data() + (first + r) * p->len_z, zs.data() + (first + r + 1) * p->len_z, aug1, &bad);
      if (bad) { settle(first + r); return vz_fail(ctx, VIMZ_ERR_UNSAT, "primary verifier circuit: inconsistent incoming instance"); }   // (nothing of this row is queued yet)
      v->ph_s[IP_SYNTH1] += now_s() - t0; v->ph_n[IP_SYNTH1]++;
      t0 = now_s();
      v->U2 = o1.U_new;
      // ---- fresh primary instance: upload the verifier wires, finish (A,B,C)·z and the commitment ---------------------------------
      memcpy(pin_aug1, aug1.data(), 32 * aw1);
      // (waited for on the HOST, not by a barrier on this stream: a high-priority queue stalled behind the producer's event keeps the
      //  producer's low-priority queues from being served — once a producer fell behind it stayed behind, 10× slower: DESIGN.md §5c)
      P_TRY(wait_row_flag(bb, r, bb.ev[r]));
      P_TRY(upload_pinned(s, Zi + 8 * sw, pin_aug1, 32 * aw1));
      // the commitment to the verifier wires needs the upload only: it starts first, on stream 2
      P_TRY(hipEventRecord(v->ev_fork, s));
      P_TRY(hipStreamWaitEvent(v->s2, v->ev_fork, 0));
      P_TRY(msm_launch<BnG1>(v->s2, v->ws2, p->ck->d + (size_t)AFFINE_WORDS * (sw - 1), Zi + 8 * sw, aw1 - 2, 1, 0, v->pin, &v->plan_aug, nullptr, 0, v->tb_aug.d ? &v->tb_aug : nullptr));
      // ---- 3. NIFS on the primary curve ------------------------------------------------------------------------------------------------
      // verifier rows: (A,B,C)·z and their part of the cross term in one launch, then its commitment over the matching slice of ck
      P_TRY(hipStreamWaitEvent(s, v->ev_fold, 0));          // (the running products come from the previous step's fold on stream 3)
      hipLaunchKernelGGL(k_spmv_cross16<Fr>, dim3((unsigned)((16 * (nc - sc) + 255) / 256)), dim3(256), 0, s, p->A, p->B, p->C, p->dict, (uint32_t)sc, (uint32_t)(nc - sc),
                         Zi, az, bz, cz, i > 0 ? p->AZ : nullptr, p->BZ, p->CZ, v->u1_run, Fe::one(), p->T);
      P_TRY(hipGetLastError());
      if (i > 0) {
        if (v->t1[i & 1].step != (int64_t)i) {       // first row of a call: nothing was queued ahead
          if ((rc = launch_direct(bb, r, i))) return rc;
          auto& sl = v->t1[i & 1];
          if (sl.tricked) {      // (its completion: u·S_1(row i) − C_A as they are now)
            boolCorr = G1::identity();
            if (!aff_is_identity(S1_row)) { const Fe uc = Fe::from_mont(sl.u_at); boolCorr = scalar_mul(S1_row, uc.v, 254); }
            G1 nca = sl.ca_at; if (!nca.is_identity()) nca.Y = Fq::neg(nca.Y);
            add_full(boolCorr, nca);
          }
        }
        P_TRY(msm_launch<BnG1>(s, ctx->msm_ws, p->ck->d + (size_t)AFFINE_WORDS * sc, p->T + 8 * sc, nc - sc, 1, 0, v->pin + 4 * v->pin_res, &v->plan_T1v, nullptr, 0, v->tb_T1v.d ? &v->tb_T1v : nullptr));
      }
      P_TRY(hipEventRecord(v->ev_a, s));
      if (i > 0) {   // the same fold on the secondary's witness vectors: nothing in this half of the step reads them, so it is queued
                     // behind the verifier rows' work (it used to open the half: 50-160 µs under load before the upload could start)
        const Fq rho2 = rho_element<Fq>(o1.rho_low);
        Fold5 f;
        f.x1[0] = S.Zrun; f.x2[0] = S.z2; f.n[0] = S.n_w;
        f.x1[1] = v->sec_T_valid ? S.E : nullptr; f.x2[1] = S.T; f.n[1] = S.n_c;
        f.x1[2] = S.AZ; f.x2[2] = S.az2; f.n[2] = S.n_c;
        f.x1[3] = S.BZ; f.x2[3] = S.bz2; f.n[3] = S.n_c;
        f.x1[4] = S.CZ; f.x2[4] = S.cz2; f.n[4] = S.n_c;
        hipLaunchKernelGGL(k_fold5<Fq>, dim3(256), dim3(256), 0, s, f, rho2);
        v->u2_run = Fq::add(v->u2_run, rho2);
      }
      v->ph_s[IP_LAUNCH] += now_s() - t0;
      t0 = now_s();
      v->c2.precompute_statement(i + 1, v->z0_sec, &zero_q);      // likewise for the secondary circuit, under the primary half's MSMs
      G1Aff cW_step = msm_finish<BnG1>(p->planB, (char*)bb.pin + r * pin_stride);
      // the large MSM's host tail (Horner over 24 window sums, ≈0.1 ms) is taken whenever its stream turns out to be done:
      // before the small ones if it already is, so that it overlaps what is still running
      G1Aff T1_step; bool t1_step_done = false;
      auto& slot = v->t1[i & 1];
      auto take_T1_step = [&](bool wait) {
        if (t1_step_done || i == 0) return hipSuccess;
        // (polled: hipEventSynchronize on this event — recorded by the launcher thread, no timing — returned 0.3 ms late)
        hipError_t q = hipEventQuery(slot.done);
        if (wait) while (q == hipErrorNotReady) { std::this_thread::yield(); q = hipEventQuery(slot.done); }
        if (q == hipErrorNotReady) return hipSuccess;
        if (q != hipSuccess) return q;
        for (auto& h : v->helpers) {
          if (!h.n) continue;
          q = wait ? hipStreamSynchronize(h.s) : hipStreamQuery(h.s);
          if (q == hipErrorNotReady) return hipSuccess;
          if (q != hipSuccess) return q;
        }
        T1_step = msm_finish<BnG1>(slot.plan, slot.pin);
        if (!v->helpers.empty()) {                      // host-side sum of the partial commitments (<= 8 points)
          G1 acc = from_affine(T1_step);
          for (auto& h : v->helpers) if (h.n) { const G1Aff part = msm_finish<BnG1>(h.plan, h.pin); add_mixed(acc, part); }
          T1_step = to_affine(acc);
        }
        t1_step_done = true;
        return hipSuccess;
      };
      // A lookahead cross term was taken against the running instance one step back: the commitment is completed by
      // rho_{i-1}·comm(T(u_{i-1}, u_i)) = −rho_{i-1}·comm(negB_i) — one 129-bit scalar multiplication on the host, done here, while
      // the device works on the verifier rows (the producer committed to negB_i long ago)
      G1 lookB = G1::identity();
      double tw = now_s();
      t_wp[0] += tw - t0;
      if (i > 0 && slot.hasB) {
        P_TRY(wait_row_flag(bb, bb.flag_rows + r, bb.ev_d[r]));
        const G1Aff cD = msm_finish<BnG1>(p->planD, (char*)bb.pin_d + r * pin_stride);
        if (!aff_is_identity(cD)) {
          G1 acc = from_affine(cD);                       // the leading one of rho = 2^128 + low
          for (int bit = 127; bit >= 0; bit--) { acc = dbl(acc); if ((v->rho_prev_low[bit >> 5] >> (bit & 31)) & 1u) add_mixed(acc, cD); }
          lookB = acc;
        }
      }
      t_wp[1] += now_s() - tw; tw = now_s();
      P_TRY(take_T1_step(false));
      P_TRY(vz_wait_stream(v->s2));
      t_wp[2] += now_s() - tw; tw = now_s();
      G1Aff cW_aug = msm_finish<BnG1>(v->plan_aug, v->pin);      // the small MSM is back first: its tail overlaps the other one
      P_TRY(take_T1_step(false));
      P_TRY(vz_wait_event(v->ev_a));
      t_wp[3] += now_s() - tw;
      v->ph_s[IP_WAIT_PRI] += now_s() - t0; v->ph_n[IP_WAIT_PRI]++;
      t0 = now_s();
      {
        static const bool dbg = getenv("VIMZ_DEBUG_CHECK_MSM") != nullptr;
        if (dbg) {
          G1Aff r; MsmStats st;
          P_TRY(msm_run<BnG1>(s, ctx->msm_ws, p->ck->d + (size_t)AFFINE_WORDS * (sw - 1), Zi + 8 * sw, aw1 - 2, 1, 0, &r, &st, nullptr, 0, nullptr));
          if (!r.x.eq(cW_aug.x) || !r.y.eq(cW_aug.y)) fprintf(stderr, "[dbg] step %llu: MSM(aug) on stream 2 differs from the recomputation\n", (unsigned long long)i);
          P_TRY(msm_run<BnG1>(s, ctx->msm_ws, p->ck->d, Zi + 8, sw - 1, 1, 0, &r, &st, nullptr, 1, nullptr));
          if (!r.x.eq(cW_step.x) || !r.y.eq(cW_step.y)) fprintf(stderr, "[dbg] step %llu: producer MSM(W) differs from the recomputation\n", (unsigned long long)i);
        }
      }
      G1 sum = from_affine(cW_step); add_mixed(sum, cW_aug);
      FreshInst<Fq> u1; u1.W = to_affine(sum); u1.x0 = cross_field<Fq>(o1.x0); u1.x1 = cross_field<Fq>(o1.x1);
      G1Aff T1; T1.x = Fq::zero(); T1.y = Fq::zero();
      if (i > 0) {
        G1Aff Tv = msm_finish<BnG1>(v->plan_T1v, v->pin + 4 * v->pin_res);      // overlaps the rest of the large MSM
        v->ph_s[IP_SYNTH2] += now_s() - t0;
        t0 = now_s();
        P_TRY(take_T1_step(true));
        v->ph_s[IP_WAIT_PRI] += now_s() - t0;
        t_wp[4] += now_s() - t0;
        G1 ts = from_affine(T1_step); add_mixed(ts, Tv);
        if (slot.tricked) add_full(ts, boolCorr);
        if (slot.hasB) { G1 nb = lookB; if (!nb.is_identity()) nb.Y = Fq::neg(nb.Y); add_full(ts, nb); }
        T1 = to_affine(ts);
        t0 = now_s();
      }
      {
        static const bool dbgp = getenv("VIMZ_DEBUG_CHECK_POINTS") != nullptr;
        if (dbgp) {
          auto on = [](const G1Aff& q) { if (aff_is_identity(q)) return true; return Fq::sqr(q.y).eq(Fq::add(Fq::mul(Fq::sqr(q.x), q.x), cb::f_from_u64<Fq>(3))); };
          if (!on(cW_step)) fprintf(stderr, "[dbg] step %llu: producer comm_W is not on the curve\n", (unsigned long long)i);
          if (!on(cW_aug)) {
            fprintf(stderr, "[dbg] step %llu: comm_W(aug) is not on the curve; window sums off curve:", (unsigned long long)i);
            const uint32_t* hw = reinterpret_cast<const uint32_t*>(v->pin);
            for (int w = 0; w < v->plan_aug.K; w++) {
              typedef BnG1::Coord C29; XYZZ<Fq> pt; Fq* f[4] = {&pt.X, &pt.Y, &pt.ZZ, &pt.ZZZ};
              for (int k = 0; k < 4; k++) { C29 t; for (int q = 0; q < 9; q++) t.v[q] = hw[(size_t)XYZZ_WORDS * w + COORD_WORDS * k + q]; *f[k] = t.to_std(); }
              if (!on(to_affine(pt))) fprintf(stderr, " %d", w);
            }
            fprintf(stderr, "\n");
          }
          if (!on(T1)) fprintf(stderr, "[dbg] step %llu: comm_T1 is not on the curve\n", (unsigned long long)i);
        }
      }
      if (i > 0 && ctx->profiling) {       // HIP-event durations of the phases of this MSM(T), accumulated for the roofline figure
        float ms[6];
        for (int q = 0; q < 6; q++) { P_TRY(hipEventElapsedTime(&ms[q], slot.ev[q], slot.ev[q + 1])); ctx->last_msm.ms[q] = ms[q]; ctx->msm_tot_ms[q] += ms[q]; }
        memcpy(&ctx->last_msm.subs, slot.pin + v->pin_res, 8);
        ctx->last_msm.c = slot.plan.c; ctx->last_msm.K = slot.plan.K;
        ctx->msm_tot_calls++; ctx->msm_tot_points += v->t1_main_n; ctx->msm_tot_entries += ctx->last_msm.entries;
      }
      // ---- 4. secondary verifier circuit on the host: folds (U1, u1) ---------------------------------------------------------------------
      AugIn<BnFq> in2; in2.digest = v->c2.digest; in2.z0 = v->z0_sec; in2.i = i; in2.U = v->U1; in2.u = u1; in2.T = T1;
      // What only waits for this step's challenge on the device — the fold of the running instance, the next step's cross term over the
      // step rows and its commitment: the longest dependent chain of a step — is queued from INSIDE the circuit's evaluation, when the
      // challenge and everything that depends on nothing else are done and the thread would wait for the two scalar-multiplication
      // chains (AugCircuit::on_challenge): the large MSM starts ≈0.1 ms earlier than after witness() returns.
      int hook_rc = VIMZ_OK; bool hook_ran = false; FoldArgs fa{};
      auto queue_folds = [&](const uint32_t* rho_low) -> int {
        const Fe rho1 = rho_element<Fe>(rho_low);
        // the rows whose step rows' cross terms come next: row i+1 (unless a lookahead produced its cross term already) and, one whole
        // step ahead, row i+2 — if its batch is out and the producer differenced it against row i+1 (fold_issue_d)
        auto row_at = [&](size_t ahead, RowAt* out) -> int {
          out->b = nullptr; out->row = 0;
          if (r + ahead < rows) { out->b = &bb; out->row = r + ahead; return VIMZ_OK; }
          if (k + 1 >= job.nbatches || r + ahead - rows >= job.rows(k + 1)) return VIMZ_OK;
          int rc2 = wait_issued(k + 1);      // its per-row events must have been recorded
          if (rc2) return rc2;
          out->b = &p->buf[(k + 1) & 1]; out->row = r + ahead - rows;
          return VIMZ_OK;
        };
        int rc2;
        fa.i = i; fa.cur = &bb; fa.r = r; fa.rho = rho1;
        if ((rc2 = row_at(1, &fa.next1))) return rc2;
        fa.need1 = fa.next1.b && v->t1[(i + 1) & 1].step != (int64_t)i + 1;
        fa.look = false;
        if (fa.next1.b && v->lookahead && v->helpers.empty() && p->want_d) {
          if ((rc2 = row_at(2, &fa.next2))) return rc2;
          fa.look = fa.next2.b && fa.next2.b->has_d[fa.next2.row];
        }
        // the witness and the verifier rows (everything k_fold_cross does not touch)
        Fold5 f;
        f.x1[0] = p->Zrun; f.x2[0] = Zi; f.n[0] = nw;
        f.x1[1] = i > 0 ? p->E + 8 * sc : nullptr; f.x2[1] = p->T + 8 * sc; f.n[1] = nc - sc;
        f.x1[2] = p->AZ + 8 * sc; f.x2[2] = az + 8 * sc; f.n[2] = nc - sc;
        f.x1[3] = p->BZ + 8 * sc; f.x2[3] = bz + 8 * sc; f.n[3] = nc - sc;
        f.x1[4] = p->CZ + 8 * sc; f.x2[4] = cz + 8 * sc; f.n[4] = nc - sc;
        v->u1_run = Fe::add(v->u1_run, rho1);
        if ((rc2 = launch_fold_and_cross(fa))) return rc2;
        // on stream 2, idle until the secondary witness is uploaded: this pass overlaps that upload instead of preceding it
        // (everything it reads is complete — the host has waited for all three streams)
        double tq = now_s();
        hipLaunchKernelGGL(k_fold5<Fr>, dim3(512), dim3(256), 0, v->s2, f, rho1);
        t_hk[3] += now_s() - tq; tq = now_s();
        P_TRY(hipEventRecord(v->ev_fold, v->s2));  // the verifier rows of the next step may start here
        t_hk[1] += now_s() - tq;
        if (trick) {      // C_A of the running instance takes this fold in — C_A += rho·S_1(row i) — at the top of the next step (apply_pending_ca); the vectors just queued were computed against it
          pend_ca = true; memcpy(pend_rho, rho_low, 16); pend_S1 = S1_row;
          pend_slot[0] = fa.need1 ? (int)((i + 1) & 1) : -1; pend_slot[1] = fa.look ? (int)(i & 1) : -1;
        }
        return VIMZ_OK;
      };
      v->c2.on_challenge = [&](const uint32_t* rho_low) { const double th = now_s(); hook_ran = true; hook_rc = queue_folds(rho_low); t_hook += now_s() - th; };
      std::vector<Fq> aug2;
      AugOut<BnFq> o2 = v->c2.witness(in2, &zero_q, &zero_q, aug2, &bad);
      v->c2.on_challenge = nullptr;
      // (the secondary running instance has already been folded with this step's challenge, and with the hook the primary one too:
      //  the guard marks the IVC broken)
      if (hook_ran && hook_rc) return hook_rc;
      if (bad) return vz_fail(ctx, VIMZ_ERR_UNSAT, "secondary verifier circuit: inconsistent incoming instance");
      v->ph_s[IP_SYNTH2] += now_s() - t0; v->ph_n[IP_SYNTH2]++;
      t0 = now_s();
      if (!hook_ran && (rc = queue_folds(o2.rho_low))) return rc;
      v->U1 = o2.U_new;
      // fresh secondary instance on the device: [1 | z_out | z_in | verifier wires]
      {
        Fq* w2 = (Fq*)pin_w2;
        w2[0] = Fq::one(); w2[1] = zero_q; w2[2] = zero_q;
        memcpy(w2 + 3, aug2.data(), 32 * aug2.size());
        if (ctx->profiling) P_TRY(hipEventRecord(v->ev_b0, s));
        P_TRY(upload_pinned(s, S.z2, pin_w2, 32 * (size_t)S.n_w));
        P_TRY(hipEventRecord(v->ev_fork, s));
        P_TRY(hipStreamWaitEvent(v->s2, v->ev_fork, 0));
        P_TRY(msm_launch<Grumpkin>(v->s2, v->ws2, v->ck2->d, S.z2 + 8, S.n_w - 3, 1, 0, v->pin + 2 * v->pin_res, &v->plan_W2, nullptr, 0, v->tb_ck2.d ? &v->tb_ck2 : nullptr));
        v->sec_T_valid = i > 0;    // U2 is still the zero instance after step 0: its cross term with anything is zero
        hipLaunchKernelGGL(k_spmv_cross16<Fq>, dim3((unsigned)((16 * (size_t)S.n_c + 255) / 256)), dim3(256), 0, s, S.A, S.B, S.C, S.dict, 0u, S.n_c, S.z2, S.az2, S.bz2, S.cz2,
                           v->sec_T_valid ? S.AZ : nullptr, S.BZ, S.CZ, v->u2_run, Fq::one(), S.T);
        if (v->sec_T_valid)
          P_TRY(msm_launch<Grumpkin>(s, ctx->msm_ws, v->ck2->d, S.T, S.n_c, 1, 0, v->pin + 3 * v->pin_res, &v->plan_T2, nullptr, 0, v->tb_ck2.d ? &v->tb_ck2 : nullptr));
        if (ctx->profiling) P_TRY(hipEventRecord(v->ev_b1, s));
        P_TRY(hipGetLastError());
        v->u2.x0 = cross_field<Fe>(o2.x0); v->u2.x1 = cross_field<Fe>(o2.x1);
        v->pending_sec = true;
      }
      v->rho_prev = rho_element<Fe>(o2.rho_low); memcpy(v->rho_prev_low, o2.rho_low, sizeof(v->rho_prev_low));
      v->ph_s[IP_LAUNCH] += now_s() - t0;
      v->i++; p->steps++;
    }
    // this buffer is rewritten by batch k+2: the folds that read it must have finished
    P_TRY(vz_wait_stream(s));
    P_TRY(vz_wait_stream(v->s2));
    P_TRY(vz_wait_event(v->ev_fold));
    // (the fused fold of the last row sits on stream 3 behind up to two large MSMs: the producers wait for it, not the host)
    if (v->fused_recorded) { P_TRY(hipStreamWaitEvent(p->sB, v->ev_fused, 0)); P_TRY(hipStreamWaitEvent(p->sH, v->ev_fused, 0)); }
  }
  if ((rc = finish_secondary(v))) return rc;
  P_TRY(hipStreamSynchronize(p->sB));
  if (p->sD) P_TRY(hipStreamSynchronize(p->sD));
  P_TRY(hipStreamSynchronize(v->s3));
  v->t1[0].step = v->t1[1].step = -1;
  for (uint32_t k = 0; k < p->len_z; k++) p->z_cur[k] = zs[nsteps * p->len_z + k];
  guard.armed = false;
  if (trick) apply_pending_ca();
  v->ca_valid = trick;
  v->ph_s[IP_TOTAL] += now_s() - t_all; v->ph_n[IP_TOTAL] += nsteps;
  if (dbg_timing) fprintf(stderr, "[timing] wait_primary_msm per step: finish(W) + statement %.3f, lookahead's host share %.3f, small MSM(W aug) %.3f, verifier rows + small MSM(T) %.3f, large MSM(T) %.3f ms\n",
                          1e3 * t_wp[0] / (double)nsteps, 1e3 * t_wp[1] / (double)nsteps, 1e3 * t_wp[2] / (double)nsteps, 1e3 * t_wp[3] / (double)nsteps, 1e3 * t_wp[4] / (double)nsteps);
  if (dbg_timing) fprintf(stderr, "[timing] fold of %zu steps: %.1f ms (prepare %.1f, first batch ready at %.1f after waiting %.1f; launches queued from inside the secondary circuit: %.3f ms per step = fused fold %.3f + 3 event records %.3f + large MSM %.3f + fold5 %.3f + row flags %.3f)\n", nsteps, 1e3 * (now_s() - t_all), 1e3 * t_prep, 1e3 * t_first, 1e3 * t_wait0, 1e3 * t_hook / (double)nsteps,
                          1e3 * t_hk[0] / (double)nsteps, 1e3 * t_hk[1] / (double)nsteps, 1e3 * t_hk[2] / (double)nsteps, 1e3 * t_hk[3] / (double)nsteps, 1e3 * t_hk[4] / (double)nsteps);
  return VIMZ_OK;
}

// The two augmented circuits of an IVC, synthesised once: see circuit_handle.hpp.  (Three IVCs of one proof used to spend 0.12 s each on the same synthesis and
// the same SHA3 over the same 20 MB of CSR, side by side, before their device buffers could be made: VERDICT r5 #5.)
struct IvcPrimaryShape { cb::Builder b; uint32_t len_z = 0, step_wires = 0, step_constraints = 0; Fe digest; };
struct IvcSecondaryShape { cb::BuilderT<Fq> b; uint32_t len_z = 0, step_wires = 0, step_constraints = 0; Fq digest; };
std::shared_ptr<IvcPrimaryShape> ivc_primary_shape(const vimz_circuit* sc) {
  std::lock_guard<std::mutex> g(sc->ivc_mu);
  if (!sc->ivc_shape) {
    auto ps = std::make_shared<IvcPrimaryShape>();
    ps->b = sc->build->b;
    AugCircuit<BnFr> a(ps->b);
    a.finish(true);
    ps->len_z = a.len_z; ps->step_wires = a.step_wires; ps->step_constraints = a.step_constraints; ps->digest = a.digest;
    sc->ivc_shape = ps;
  }
  return std::static_pointer_cast<IvcPrimaryShape>(sc->ivc_shape);
}
std::shared_ptr<IvcSecondaryShape> ivc_secondary_shape() {
  static std::mutex mu;
  static std::shared_ptr<IvcSecondaryShape> cached;
  std::lock_guard<std::mutex> g(mu);
  if (!cached) {
    auto ss = std::make_shared<IvcSecondaryShape>();
    AugCircuit<BnFq> a;
    a.init_trivial_step();
    a.finish(false);
    ss->b = a.b; ss->len_z = a.len_z; ss->step_wires = a.step_wires; ss->step_constraints = a.step_constraints; ss->digest = a.digest;
    cached = ss;
  }
  return cached;
}

template <class F>
bool fetch_elements(hipStream_t s, const uint32_t* d, size_t idx, size_t n, F* out) {
  return hipMemcpyAsync(out, d + 8 * idx, 32 * n, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
}

}  // namespace

extern "C" {

// Synthesises the augmented circuits a Nova IVC over `c` needs (the step circuit with the verifier circuit appended + its digest; the secondary circuit) and
// keeps them with the circuit object: vimz_ivc_create then only copies.  Optional — vimz_ivc_create does it on first use — and host-only: a set-up calls it
// on the thread that built the circuit, while the GPU contexts come up.
int vimz_circuit_prepare_ivc(const vimz_circuit* c) {
  if (!c || !c->build) return VIMZ_ERR_INVALID;
  try { ivc_primary_shape(c); ivc_secondary_shape(); } catch (const std::exception&) { return VIMZ_ERR_INVALID; }
  return VIMZ_OK;
}

void vimz_ivc_free(vimz_ivc* v) {
  if (!v) return;
  if (v->orphan_merged) v->orphan_merged(v);
  if (v->spartan_free) v->spartan_free(v);
  if (v->pri) vimz_prover_free(v->pri);
  if (v->ctx) {
    std::lock_guard<std::mutex> g(v->ctx->mu);
    hipSetDevice(v->ctx->device);
    hipStreamSynchronize(v->ctx->stream);
    { int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
      if (v->s2 && v->s2 != v->ctx->stream) vz_stream_release(v->ctx, hi, v->s2);
      if (v->s3 && v->s3 != v->ctx->stream) vz_stream_release(v->ctx, (lo + hi) / 2, v->s3); }
    if (v->ev_fork) hipEventDestroy(v->ev_fork);
    if (v->ev_fold) hipEventDestroy(v->ev_fold);
    if (v->ev_fused) hipEventDestroy(v->ev_fused);
    for (int q = 0; q < 2; q++) if (v->t1[q].done) hipEventDestroy(v->t1[q].done);
    for (int q = 0; q < 7; q++) if (v->ev_alt[q]) hipEventDestroy(v->ev_alt[q]);
    if (v->pin_t1b) hipHostFree(v->pin_t1b);
    if (v->ev_b0) hipEventDestroy(v->ev_b0);
    if (v->ev_b1) hipEventDestroy(v->ev_b1);
    if (v->ev_a) hipEventDestroy(v->ev_a);
    if (v->ev_T) hipEventDestroy(v->ev_T);
    for (auto& h : v->helpers) {
      hipSetDevice(h.ctx->device);
      if (h.s) { hipStreamSynchronize(h.s); hipStreamDestroy(h.s); }
      h.ws.release();
      hipFree(h.T);
      if (h.pin) hipHostFree(h.pin);
    }
    hipSetDevice(v->ctx->device);
    v->ws2.release(); v->ws3.release();
    for (auto& mp : v->ipc_mappings) if (mp.ptr) hipIpcCloseMemHandle(mp.ptr);
    for (int k = 0; k < vimz_ivc::MERGED_SPARES; k++) { if (v->merged_spare_dev[k]) hipFree(v->merged_spare_dev[k]); if (v->merged_spare_pin[k]) hipHostFree(v->merged_spare_pin[k]); }
    for (void* d : v->owned) hipFree(d);
    if (v->pin) hipHostFree(v->pin);
  }
  delete v;
}

int vimz_ivc_create(vimz_ctx* ctx, const vimz_circuit* step_circuit, const vimz_bases* ck1, const vimz_bases* ck2, size_t max_batch, vimz_ivc** out) {
  if (!ctx || !step_circuit || !ck1 || !ck2 || !out || max_batch == 0) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_create: bad argument");
  if (ck1->curve != VIMZ_CURVE_BN254_G1 || ck2->curve != VIMZ_CURVE_GRUMPKIN) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_create: keys must be on BN254 G1 (primary) and Grumpkin (secondary)");
  std::unique_ptr<vimz_ivc> v(new vimz_ivc());
  v->ctx = ctx; v->ck1 = ck1; v->ck2 = ck2;
  static const bool dbg_create = getenv("VIMZ_DEBUG_TIMING") != nullptr;
  double t_c = now_s();
  auto lap = [&](const char* what) { if (dbg_create) { const double t = now_s(); fprintf(stderr, "[timing] ivc_create(%p): %s %.1f ms\n", (void*)ctx, what, 1e3 * (t - t_c)); t_c = t; } };
  try {
    // both augmented circuits come ready-made: the primary one from the step circuit object's cache, the secondary one (the same for every IVC) from the process's
    const std::shared_ptr<IvcPrimaryShape> ps = ivc_primary_shape(step_circuit);
    const std::shared_ptr<IvcSecondaryShape> ss = ivc_secondary_shape();
    v->circ1.reset(new vimz_circuit());
    v->circ1->transformation = step_circuit->transformation; v->circ1->shape = step_circuit->shape;
    v->circ1->build.reset(new cb::CircuitBuild());
    v->circ1->build->b = ps->b;
    v->c1.reset(new AugCircuit<BnFr>(v->circ1->build->b));
    v->c1->adopt(true, ps->len_z, ps->step_wires, ps->step_constraints, ps->digest);
    v->c2.b = ss->b;
    v->c2.adopt(false, ss->len_z, ss->step_wires, ss->step_constraints, ss->digest);
  } catch (const std::exception& e) { return vz_fail(ctx, VIMZ_ERR_INVALID, e.what()); }
  const uint32_t nw2 = v->c2.n_wires(), nc2 = v->c2.n_constraints();
  if (ck2->n < nw2 || ck2->n < nc2) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_create: secondary commitment key shorter than the secondary circuit");
  lap("circuit copy + both verifier circuits");
  int rc = vz_prover_create_layout(ctx, v->circ1.get(), ck1, max_batch, 1, v->c1->step_wires, v->c1->step_constraints, &v->pri);
  if (rc) return rc;
  lap("prover layout (uploads, device buffers, pinned, streams)");
  std::unique_lock<std::mutex> lk(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  SecDev& S = v->sec;
  S.n_w = nw2; S.n_c = nc2;
  const cb::BuilderT<Fq>& b2 = v->c2.b;
  hipError_t e = hipSuccess;
  auto fail = [&](const char* what) { const hipError_t ee = e; lk.unlock(); vimz_ivc_free(v.release()); return vz_fail(ctx, VIMZ_ERR_HIP, what, ee); };
#define UP2(vec, dst) do { e = upload(vec, &dst); if (dst) v->owned.push_back((void*)dst); if (e != hipSuccess) return fail("upload " #vec); } while (0)
  UP2(b2.A.row_ptr, S.A.row_ptr); UP2(b2.A.col, S.A.col); UP2(b2.A.coef, S.A.coef);
  UP2(b2.B.row_ptr, S.B.row_ptr); UP2(b2.B.col, S.B.col); UP2(b2.B.coef, S.B.coef);
  UP2(b2.C.row_ptr, S.C.row_ptr); UP2(b2.C.col, S.C.col); UP2(b2.C.coef, S.C.coef);
  { const Fq* d = nullptr; const std::vector<Fq> dd = dict_for_device(b2.dict); UP2(dd, d); S.dict = (const uint32_t*)d; }
  {
    std::vector<uint32_t> items;
    const cb::Csr* Ms[3] = {&b2.A, &b2.B, &b2.C};
    for (uint32_t m = 0; m < 3; m++)
      for (uint32_t r = 0; r + 1 < Ms[m]->row_ptr.size(); r++)
        if (Ms[m]->row_ptr[r + 1] - Ms[m]->row_ptr[r] > SPMV_LONG) items.push_back((m << 30) | r);
    S.n_med = spmv_sort_items(items, [&](uint32_t it) { const cb::Csr* M = Ms[it >> 30]; const uint32_t r = it & 0x3fffffffu; return M->row_ptr[r + 1] - M->row_ptr[r]; });
    S.n_long = (uint32_t)items.size();
    UP2(items, S.long_items);
  }
#undef UP2
  auto dalloc = [&](uint32_t** dst, size_t bytes) { e = hipMalloc((void**)dst, bytes); if (e == hipSuccess) { v->owned.push_back(*dst); e = hipMemset(*dst, 0, bytes); } return e; };
  uint32_t** vw[] = {&S.Zrun, &S.z2};
  uint32_t** vc[] = {&S.E, &S.AZ, &S.BZ, &S.CZ, &S.az2, &S.bz2, &S.cz2, &S.T};
  for (auto d : vw) if (dalloc(d, 32 * (size_t)nw2) != hipSuccess) return fail("device allocation");
  for (auto d : vc) if (dalloc(d, 32 * (size_t)nc2) != hipSuccess) return fail("device allocation");
  if (dalloc(&S.bad, 64) != hipSuccess) return fail("device allocation");
  { int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (getenv("VIMZ_DEBUG_NO_S2")) v->s2 = ctx->stream;
    else if ((e = vz_stream_acquire(ctx, hi, &v->s2)) != hipSuccess) return fail("stream");
    if (getenv("VIMZ_DEBUG_NO_S2")) v->s3 = ctx->stream;
    else if ((e = vz_stream_acquire(ctx, (lo + hi) / 2, &v->s3)) != hipSuccess) return fail("stream");
    if ((e = hipEventCreateWithFlags(&v->ev_fork, hipEventDisableTiming)) != hipSuccess) return fail("event");
    if ((e = hipEventCreateWithFlags(&v->ev_fold, hipEventDisableTiming)) != hipSuccess) return fail("event");
    if ((e = hipEventCreateWithFlags(&v->ev_fused, hipEventDisableTiming)) != hipSuccess) return fail("event");
    if ((e = hipEventCreate(&v->ev_b0)) != hipSuccess || (e = hipEventCreate(&v->ev_b1)) != hipSuccess) return fail("event");
    if ((e = hipEventCreateWithFlags(&v->ev_a, hipEventDisableTiming)) != hipSuccess) return fail("event"); }
  lap("secondary circuit on the device, streams, events");
  if (!getenv("VIMZ_DEBUG_NO_SMALL_TABLES")) {
    // tables of the three fixed slices the per-step small MSMs run over (verifier wires and verifier rows of ck1, the head of ck2):
    // rows 2^(7w)·P_i and every multiple of them, shared by all IVCs over these keys (vz_small_tables)
    const cb::Builder& b1 = v->circ1->build->b;
    const size_t sw = v->c1->step_wires, sc = v->c1->step_constraints, aw = v->c1->aug_wires();
    // (every multiple resident — k_msm_fixed, 1.46 GB per slice — is 18-40 % faster alone and SLOWER in a fold: its gathers sweep
    //  the caches the large MSM's accumulation lives in; option VIMZ_IVC_MULT_TABLES=1)
    const bool mult = getenv("VIMZ_IVC_MULT_TABLES") != nullptr && atoi(getenv("VIMZ_IVC_MULT_TABLES")) != 0;
    if ((rc = vz_small_tables(ctx, const_cast<vimz_bases*>(ck1), sw - 1, aw - 2, mult, &v->tb_aug))) { lk.unlock(); vimz_ivc_free(v.release()); return rc; }
    if ((rc = vz_small_tables(ctx, const_cast<vimz_bases*>(ck1), sc, b1.n_constraints() - sc, mult, &v->tb_T1v))) { lk.unlock(); vimz_ivc_free(v.release()); return rc; }
    if ((rc = vz_small_tables(ctx, const_cast<vimz_bases*>(ck2), 0, std::max<size_t>(nw2 - 3, nc2), mult, &v->tb_ck2))) { lk.unlock(); vimz_ivc_free(v.release()); return rc; }
  }
  lap("small-MSM tables");
  v->pin_res = 4 * (size_t)XYZZ_WORDS * MSM_MAX_WINDOWS;
  v->pin_totals = 5 * v->pin_res + 32 * (size_t)v->c1->aug_wires() + 32 * (size_t)nw2;
  if ((e = hipHostMalloc((void**)&v->pin, v->pin_totals + 64)) != hipSuccess) return fail("pinned");
  // the two slots of the step rows' cross term (T1Slot): vector, pinned window sums + totals, events
  if ((e = hipHostMalloc((void**)&v->pin_t1b, 2 * (v->pin_res + 64))) != hipSuccess) return fail("pinned");
  for (int q = 0; q < 2; q++) {
    if (dalloc(&v->t1[q].buf, 32 * (size_t)v->c1->step_constraints) != hipSuccess) return fail("device allocation");
    if ((e = hipEventCreateWithFlags(&v->t1[q].done, hipEventDisableTiming)) != hipSuccess) return fail("event");
    v->t1[q].pin = v->pin_t1b + q * (v->pin_res + 64);
  }
  {   // boolean-row form of the step rows' cross-term commitment: a second vector per slot (the one the MSM takes)
    const char* e2 = getenv("VIMZ_IVC_BOOL_ROWS");
    v->bool_rows = v->pri->n_bool > 0 && !(e2 && atoi(e2) == 0);
    if (v->bool_rows) for (int q = 0; q < 2; q++) if (dalloc(&v->t1[q].bufm, 32 * (size_t)v->c1->step_constraints) != hipSuccess) return fail("device allocation");
    v->pri->want_s1 = v->bool_rows;
  }
  for (int q = 0; q < 7; q++) if ((e = hipEventCreate(&v->ev_alt[q])) != hipSuccess) return fail("event");
  v->t1[0].ev = ctx->ev; v->t1[1].ev = v->ev_alt;
  if (v->c2.use_worker) {       // one pair of helper threads for both circuits
    for (auto& w : v->chain_w) w.reset(new aug::Worker());
    v->c1->shared_w = v->c2.shared_w = v->chain_w[0].get(); v->c1->shared_w2 = v->c2.shared_w2 = v->chain_w[1].get();
  }
  v->lookahead = ivc_lookahead_enabled();
  v->pri->want_d = v->lookahead;
  if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return fail("sync");   // the hipMemset fills above ran on the null stream
  lap("pinned, slots, workers, final sync");
  v->z0.assign(v->c1->len_z, Fe::zero());
  v->U1 = RelaxedInst<Fq>::zero(); v->U2 = RelaxedInst<Fe>::zero(); v->u2 = FreshInst<Fe>::zero(); v->T2.x = v->T2.y = Fe::zero();
  *out = v.release();
  return VIMZ_OK;
}

// A helper GPU for the large cross-term commitment of every step.  helper_ctx: a context on another device (or,
// for tests on a one-GPU box, a second context on the same device); ck_on_helper: a replica of ck_primary resident there.
int vimz_ivc_add_msm_helper(vimz_ivc* v, vimz_ctx* helper_ctx, const vimz_bases* ck_on_helper) {
  if (!v || !helper_ctx || !ck_on_helper) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = v->ctx;
  if (ck_on_helper->curve != VIMZ_CURVE_BN254_G1 || ck_on_helper->n < v->ck1->n) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_add_msm_helper: the helper's key must be a replica of ck_primary");
  if (v->helpers.size() >= 7) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_add_msm_helper: at most seven helpers");
  std::lock_guard<std::mutex> g(ctx->mu);
  vimz_ivc::MsmHelper h; h.ctx = helper_ctx; h.ck = ck_on_helper;
  P_TRY(hipSetDevice(helper_ctx->device));
  int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
  P_TRY(hipStreamCreateWithPriority(&h.s, hipStreamNonBlocking, (lo + hi) / 2));
  P_TRY(hipMalloc((void**)&h.T, 32 * (size_t)v->pri->step_c));
  P_TRY(hipHostMalloc(&h.pin, 4 * (size_t)XYZZ_WORDS * MSM_MAX_WINDOWS));
  if (helper_ctx->device != ctx->device) { int can = 0; hipDeviceCanAccessPeer(&can, helper_ctx->device, ctx->device); if (can) hipDeviceEnablePeerAccess(ctx->device, 0); }
  P_TRY(hipSetDevice(ctx->device));
  if (!v->ev_T) P_TRY(hipEventCreateWithFlags(&v->ev_T, hipEventDisableTiming));
  v->helpers.push_back(std::move(h));
  return VIMZ_OK;
}

int vimz_ivc_reset(vimz_ivc* v, const uint64_t* z0) {
  if (!v || !z0) return VIMZ_ERR_INVALID;
  int rc = vimz_prover_reset(v->pri, z0);
  if (rc) return rc;
  vimz_ctx* ctx = v->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  SecDev& S = v->sec;
  uint32_t* zw[] = {S.Zrun, S.z2};
  uint32_t* zc[] = {S.E, S.AZ, S.BZ, S.CZ, S.az2, S.bz2, S.cz2, S.T};
  for (auto d : zw) P_TRY(hipMemsetAsync(d, 0, 32 * (size_t)S.n_w, s));
  for (auto d : zc) P_TRY(hipMemsetAsync(d, 0, 32 * (size_t)S.n_c, s));
  P_TRY(hipStreamSynchronize(s));
  v->i = 0;
  for (uint32_t k = 0; k < v->c1->len_z; k++) v->z0[k] = v->pri->z_cur[k];
  v->U1 = RelaxedInst<Fq>::zero(); v->U2 = RelaxedInst<Fe>::zero(); v->u2 = FreshInst<Fe>::zero(); v->T2.x = v->T2.y = Fe::zero();
  v->u1_run = Fe::zero(); v->u2_run = Fq::zero();
  v->CA = G1::identity(); v->ca_valid = true;
  v->pending_sec = false; v->sec_T_valid = false; v->t1[0].step = v->t1[1].step = -1; v->broken = false;
  memset(v->ph_s, 0, sizeof(v->ph_s)); memset(v->ph_n, 0, sizeof(v->ph_n));
  return VIMZ_OK;
}

int vimz_ivc_fold(vimz_ivc* v, const uint64_t* step_inputs, size_t nsteps) {
  if (!v || (!step_inputs && nsteps)) return VIMZ_ERR_INVALID;
  const cb::Builder& b = v->circ1->build->b;
  if (b.zout.empty() && nsteps) return vz_fail(v->ctx, VIMZ_ERR_INVALID, "this circuit was loaded from an .r1cs and has no witness program: use vimz_ivc_fold_witness");
  if (!b.gpu_witness && nsteps) return vz_fail(v->ctx, VIMZ_ERR_INVALID, "no GPU witness kernels for this step circuit: supply witnesses with vimz_ivc_fold_witness");
  try { return ivc_fold_core(v, step_inputs, nullptr, nsteps); } catch (const std::exception& e) { return vz_fail(v->ctx, VIMZ_ERR_INVALID, e.what()); }
}
int vimz_ivc_fold_witness(vimz_ivc* v, const uint64_t* witnesses, size_t nsteps) {
  if (!v || (!witnesses && nsteps)) return VIMZ_ERR_INVALID;
  try { return ivc_fold_core(v, nullptr, witnesses, nsteps); } catch (const std::exception& e) { return vz_fail(v->ctx, VIMZ_ERR_INVALID, e.what()); }
}

// A fingerprint of the HOST a benchmark line was measured on, so that a box-to-box gap is attributable from the line alone (bench.py):
// out[0] = µs per Poseidon permutation (t = 9, BN254 Fr) on one host core — the unit of the verifier circuits' witness generation;
// out[1] = µs per empty kernel launch + stream synchronise on the context's stream (median of 200) — the unit of every host/GPU hand-over;
// out[2] = host cores this process may use (affinity mask and cgroup quota); out[3] = µs per hipEventRecord + hipEventSynchronize (median of 200).
namespace vz { __global__ void k_fingerprint_empty() {} }
int vimz_host_fingerprint(vimz_ctx* ctx, double out[4]) {
  if (!ctx || !out) return VIMZ_ERR_INVALID;
  {
    std::vector<Fe> in(8);
    for (int k = 0; k < 8; k++) in[k] = cb::f_from_u64<Fe>(0x9e3779b97f4a7c15ull * (k + 1));
    const int N = 2000;
    const double t0 = now_s();
    for (int i = 0; i < N; i++) in[i & 7] = hash_native<BnFr>(in);
    out[0] = 1e6 * (now_s() - t0) / N;
    if (in[0].is_zero()) out[0] = -1;      // (keeps the loop)
  }
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  P_TRY(hipStreamSynchronize(ctx->stream));
  std::vector<double> a(200), b(200);
  hipEvent_t ev; P_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int i = 0; i < 200; i++) {
    double t0 = now_s();
    hipLaunchKernelGGL(vz::k_fingerprint_empty, dim3(1), dim3(64), 0, ctx->stream);
    hipStreamSynchronize(ctx->stream);
    a[i] = 1e6 * (now_s() - t0);
    t0 = now_s();
    hipEventRecord(ev, ctx->stream); hipEventSynchronize(ev);
    b[i] = 1e6 * (now_s() - t0);
  }
  hipEventDestroy(ev);
  P_TRY(hipGetLastError());
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  out[1] = a[100]; out[2] = (double)usable_cpus(); out[3] = b[100];
  return VIMZ_OK;
}
int vimz_ivc_info(const vimz_ivc* v, uint64_t info[12]) {
  if (!v || !info) return VIMZ_ERR_INVALID;
  const cb::Builder& b1 = v->circ1->build->b; const cb::BuilderT<Fq>& b2 = v->c2.b;
  info[0] = v->i; info[1] = b1.n_wires; info[2] = b1.n_constraints(); info[3] = v->c1->step_wires; info[4] = v->c1->step_constraints;
  info[5] = b2.n_wires; info[6] = b2.n_constraints(); info[7] = v->c1->len_z; info[8] = v->c1->aug_wires();
  info[9] = b1.A.col.size() + b1.B.col.size() + b1.C.col.size(); info[10] = b2.A.col.size() + b2.B.col.size() + b2.C.col.size(); info[11] = v->pri->last_head_rows;
  return VIMZ_OK;
}
int vimz_ivc_state(const vimz_ivc* v, uint64_t* z_current, uint64_t* steps) {
  if (!v) return VIMZ_ERR_INVALID;
  if (z_current) for (uint32_t k = 0; k < v->pri->len_z; k++) fe_to_canon(v->pri->z_cur[k], z_current + 4 * k);
  if (steps) *steps = v->i;
  return VIMZ_OK;
}
int vimz_ivc_state_chain(vimz_ivc* v, const uint64_t* z_start, const uint64_t* step_inputs, size_t nsteps, uint64_t* zs_out) {
  if (!v) return VIMZ_ERR_INVALID;
  return vimz_prover_state_chain(v->pri, z_start, step_inputs, nsteps, zs_out);
}
size_t vimz_ivc_digest_stride(const vimz_ivc* v) { return v ? vimz_prover_digest_stride(v->pri) : 0; }
int vimz_ivc_row_digests(vimz_ivc* v, const uint64_t* step_inputs, size_t nsteps, uint64_t* digests_out) {
  if (!v) return VIMZ_ERR_INVALID;
  return vimz_prover_row_digests(v->pri, step_inputs, nsteps, digests_out);
}
int vimz_ivc_chain_from_digests(vimz_ivc* v, const uint64_t* z_start, const uint64_t* step_inputs, const uint64_t* digests, size_t nsteps, uint64_t* zs_out) {
  if (!v) return VIMZ_ERR_INVALID;
  return vimz_prover_chain_from_digests(v->pri, z_start, step_inputs, digests, nsteps, zs_out);
}
int vimz_ivc_profile(const vimz_ivc* v, double seconds[8], uint64_t counts[8]) {
  if (!v) return VIMZ_ERR_INVALID;
  if (seconds) memcpy(seconds, v->ph_s, sizeof(double) * IP_COUNT);
  if (counts) memcpy(counts, v->ph_n, sizeof(uint64_t) * IP_COUNT);
  return VIMZ_OK;
}

int vimz_ivc_verify(vimz_ivc* v, uint64_t num_steps, const uint64_t* z0, uint32_t* result) {
  if (!v || !result || !z0) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = v->ctx;
  vimz_prover* p = v->pri;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  SecDev& S = v->sec;
  uint32_t res = 0;
  // 0. the statement: the proof must be about the claimed number of steps from the claimed initial state
  //    (RecursiveSNARK::verify(pp, num_steps, z0_primary, z0_secondary), reached from folding.rs:53-55)
  std::vector<Fe> z0c(p->len_z);
  for (uint32_t k = 0; k < p->len_z; k++) {
    Fe c; memcpy(c.v, z0 + 4 * k, 32);
    if (!c.is_reduced()) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_verify: z0 element not below the modulus");
    z0c[k] = Fe::to_mont(c);
  }
  if (v->i != num_steps) res |= 4096;
  for (uint32_t k = 0; k < p->len_z; k++) if (!z0c[k].eq(v->z0[k])) res |= 4096;
  if (v->i == 0) {       // zero steps: the "proof" is the initial state itself
    for (uint32_t k = 0; k < p->len_z; k++) if (!p->z_cur[k].eq(z0c[k])) res |= 4096;
    *result = res; return VIMZ_OK;
  }
  // 1. the two hashes carried by the last fresh secondary instance, recomputed from the CLAIMED z0
  {
    Fe h1 = instance_hash_native<BnFr>(v->c1->digest, v->i, z0c, p->z_cur, v->U2);
    if (!h1.eq(v->u2.x0)) res |= 1;
    std::vector<Fq> zq = {Fq::zero()};
    Fq h2 = instance_hash_native<BnFq>(v->c2.digest, v->i, v->z0_sec, zq, v->U1);
    if (!cross_field<Fe>(h2).eq(v->u2.x1)) res |= 2;
  }
  const uint32_t init[2] = {0, 0xffffffffu};
  uint32_t bad[2];
  uint64_t pt[8];
  int rc;
  // 2. running primary instance
  launch_spmv(p, s, p->Zrun, p->az2, p->bz2, p->cz2, 0);
  P_TRY(hipMemcpyAsync(p->bad_d, init, 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_check_relaxed<Fr>, dim3(stream_grid(p->n_c)), dim3(256), 0, s, (size_t)p->n_c, p->az2, p->bz2, p->cz2, v->u1_run, (const uint32_t*)p->E, p->bad_d);
  P_TRY(hipMemcpyAsync(bad, p->bad_d, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 4;
  // the running products kept by linearity must equal the recomputed ones (they feed the next cross term)
  P_TRY(hipMemcpyAsync(p->bad_d, init, 8, hipMemcpyHostToDevice, s));
  { const uint32_t* kept[3] = {p->AZ, p->BZ, p->CZ}; const uint32_t* fresh[3] = {p->az2, p->bz2, p->cz2};
    for (int m = 0; m < 3; m++) hipLaunchKernelGGL(k_count_diff, dim3(stream_grid(p->n_c)), dim3(256), 0, s, (size_t)p->n_c, kept[m], fresh[m], p->bad_d); }
  P_TRY(hipMemcpyAsync(bad, p->bad_d, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 2048;
  if ((rc = vz_msm_device(ctx, p->ck, 0, p->Zrun + 8, p->n_wires - 3, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (memcmp(pt, v->U1.W.x.v, 32) || memcmp(pt + 4, v->U1.W.y.v, 32)) res |= 8;
  if ((rc = vz_msm_device(ctx, p->ck, 0, p->E, p->n_c, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (memcmp(pt, v->U1.E.x.v, 32) || memcmp(pt + 4, v->U1.E.y.v, 32)) res |= 16;
  {
    Fe e[3];
    if (!fetch_elements(s, p->Zrun, 0, 1, &e[0]) || !fetch_elements(s, p->Zrun, p->n_wires - 2, 2, &e[1])) return vz_fail(ctx, VIMZ_ERR_HIP, "verify: download");
    if (!e[0].eq(v->u1_run) || !cross_field<Fq>(e[0]).eq(v->U1.u)) res |= 1024;
    if (memcmp(to_u256(e[1]).w, v->U1.X0.w, 32) || memcmp(to_u256(e[2]).w, v->U1.X1.w, 32)) res |= 1024;
  }
  // 3. running secondary instance
  sec_spmv<Fq>(S, s, S.Zrun, S.az2, S.bz2, S.cz2);     // (the fresh products are recomputed below)
  P_TRY(hipMemcpyAsync(S.bad, init, 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_check_relaxed<Fq>, dim3(stream_grid(S.n_c)), dim3(256), 0, s, (size_t)S.n_c, S.az2, S.bz2, S.cz2, v->u2_run, (const uint32_t*)S.E, S.bad);
  P_TRY(hipMemcpyAsync(bad, S.bad, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 32;
  P_TRY(hipMemcpyAsync(S.bad, init, 8, hipMemcpyHostToDevice, s));
  { const uint32_t* kept[3] = {S.AZ, S.BZ, S.CZ}; const uint32_t* fresh[3] = {S.az2, S.bz2, S.cz2};
    for (int m = 0; m < 3; m++) hipLaunchKernelGGL(k_count_diff, dim3(stream_grid(S.n_c)), dim3(256), 0, s, (size_t)S.n_c, kept[m], fresh[m], S.bad); }
  P_TRY(hipMemcpyAsync(bad, S.bad, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 2048;
  if ((rc = vz_msm_device(ctx, v->ck2, 0, S.Zrun + 8, S.n_w - 3, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (memcmp(pt, v->U2.W.x.v, 32) || memcmp(pt + 4, v->U2.W.y.v, 32)) res |= 64;
  if ((rc = vz_msm_device(ctx, v->ck2, 0, S.E, S.n_c, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (memcmp(pt, v->U2.E.x.v, 32) || memcmp(pt + 4, v->U2.E.y.v, 32)) res |= 128;
  {
    Fq e[3];
    if (!fetch_elements(s, S.Zrun, 0, 1, &e[0]) || !fetch_elements(s, S.Zrun, S.n_w - 2, 2, &e[1])) return vz_fail(ctx, VIMZ_ERR_HIP, "verify: download");
    if (!e[0].eq(v->u2_run) || !cross_field<Fe>(e[0]).eq(v->U2.u)) res |= 1024;
    if (memcmp(to_u256(e[1]).w, v->U2.X0.w, 32) || memcmp(to_u256(e[2]).w, v->U2.X1.w, 32)) res |= 1024;
  }
  // 4. the last fresh secondary instance (strict R1CS)
  sec_spmv<Fq>(S, s, S.z2, S.az2, S.bz2, S.cz2);
  P_TRY(hipMemcpyAsync(S.bad, init, 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_check_relaxed<Fq>, dim3(stream_grid(S.n_c)), dim3(256), 0, s, (size_t)S.n_c, S.az2, S.bz2, S.cz2, Fq::one(), (const uint32_t*)nullptr, S.bad);
  P_TRY(hipMemcpyAsync(bad, S.bad, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 256;
  if ((rc = vz_msm_device(ctx, v->ck2, 0, S.z2 + 8, S.n_w - 3, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (memcmp(pt, v->u2.W.x.v, 32) || memcmp(pt + 4, v->u2.W.y.v, 32)) res |= 512;
  {
    Fq e[3];
    if (!fetch_elements(s, S.z2, 0, 1, &e[0]) || !fetch_elements(s, S.z2, S.n_w - 2, 2, &e[1])) return vz_fail(ctx, VIMZ_ERR_HIP, "verify: download");
    if (!e[0].eq(Fq::one()) || !cross_field<Fe>(e[1]).eq(v->u2.x0) || !cross_field<Fe>(e[2]).eq(v->u2.x1)) res |= 1024;
  }
  *result = res;
  return VIMZ_OK;
}

int64_t vimz_ivc_export(vimz_ivc* v, int side, int what, void* buf, size_t cap) {
  if (!v || (side != 0 && side != 1)) return VIMZ_ERR_INVALID;
  if (what < 100 || what == VIMZ_IX_INFO) return side == 0 ? export_r1cs(*v->c1, what, buf, cap) : export_r1cs(v->c2, what, buf, cap);
  vimz_ctx* ctx = v->ctx;
  vimz_prover* p = v->pri;
  SecDev& S = v->sec;
  auto put = [](uint64_t* dst, const auto& m) { auto x = std::decay_t<decltype(m)>::from_mont(m); memcpy(dst, x.v, 32); };
  if (what == VIMZ_IX_INSTANCE || what == VIMZ_IX_FRESH_INSTANCE || what == VIMZ_IX_PARAMS) {
    std::vector<uint64_t> o;
    auto push = [&](const auto& m) { o.resize(o.size() + 4); put(o.data() + o.size() - 4, m); };
    auto push_u = [&](const U256w& x) { o.insert(o.end(), x.w, x.w + 4); };
    if (what == VIMZ_IX_INSTANCE) {
      if (side == 0) { push(v->U1.W.x); push(v->U1.W.y); push(v->U1.E.x); push(v->U1.E.y); push(v->U1.u); push_u(v->U1.X0); push_u(v->U1.X1); }
      else { push(v->U2.W.x); push(v->U2.W.y); push(v->U2.E.x); push(v->U2.E.y); push(v->U2.u); push_u(v->U2.X0); push_u(v->U2.X1); }
    } else if (what == VIMZ_IX_FRESH_INSTANCE) {
      if (side != 1) return VIMZ_ERR_INVALID;
      push(v->u2.W.x); push(v->u2.W.y); push(v->u2.x0); push(v->u2.x1);
    } else {
      if (side == 0) { push(v->c1->digest); for (auto& z : v->z0) push(z); for (auto& z : p->z_cur) push(z); }
      else { push(v->c2.digest); push(Fq::zero()); push(Fq::zero()); }
    }
    const size_t bytes = o.size() * 8;
    if (buf && cap >= bytes) memcpy(buf, o.data(), bytes);
    return (int64_t)bytes;
  }
  const uint32_t* src = nullptr; size_t n = 0;
  switch (what) {
    case VIMZ_IX_RUNNING_Z: src = side == 0 ? p->Zrun : S.Zrun; n = side == 0 ? p->n_wires : S.n_w; break;
    case VIMZ_IX_RUNNING_E: src = side == 0 ? p->E : S.E; n = side == 0 ? p->n_c : S.n_c; break;
    case VIMZ_IX_FRESH_Z: if (side != 1) return VIMZ_ERR_INVALID; src = S.z2; n = S.n_w; break;
    default: return VIMZ_ERR_INVALID;
  }
  const size_t bytes = 32 * n;
  if (!buf || cap < bytes) return (int64_t)bytes;
  std::lock_guard<std::mutex> g(ctx->mu);
  if (hipSetDevice(ctx->device) != hipSuccess) return VIMZ_ERR_HIP;
  hipStream_t s = ctx->stream;
  int rc = vz_ensure_scratch(ctx, bytes); if (rc) return rc;
  if (side == 0) launch_from_mont<Fr>(s, src, (uint32_t*)ctx->scratch, n); else launch_from_mont<Fq>(s, src, (uint32_t*)ctx->scratch, n);
  if (hipMemcpyAsync(buf, ctx->scratch, bytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return VIMZ_ERR_HIP;
  return (int64_t)bytes;
}

// ---- the proof as an object of its own: RecursiveSNARK serialisation / checkpoint-resume ------------------------------------------
// Blob = header | host state | device vectors (Montgomery limbs as they sit in HBM).  It holds everything vimz_ivc_verify reads and
// everything the next vimz_ivc_fold needs, so a proof can be verified by another process (import into a fresh vimz_ivc built for the
// same step circuit and keys) or folding can resume after a restart.
namespace {
struct ProofHeader { uint64_t magic, steps, n_w1, n_c1, n_w2, n_c2, len_z, flags; };
const uint64_t PROOF_MAGIC = 0x3243564956ull;   // "VIVC2"
struct ProofHost {
  RelaxedInst<Fe> U2; RelaxedInst<Fq> U1; FreshInst<Fe> u2; G2Aff T2; Fe u1_run; Fq u2_run; Fe digest1; Fq digest2;
};
size_t proof_vec_bytes(const vimz_ivc* v) {
  const size_t nw1 = v->pri->n_wires, nc1 = v->pri->n_c, nw2 = v->sec.n_w, nc2 = v->sec.n_c;
  return 32 * (nw1 + 4 * nc1 + 2 * nw2 + 8 * nc2);
}
}  // namespace

size_t vimz_ivc_proof_size(const vimz_ivc* v) {
  if (!v) return 0;
  return sizeof(ProofHeader) + sizeof(ProofHost) + 64 * (size_t)v->pri->len_z + proof_vec_bytes(v);
}

int vimz_ivc_proof_export(vimz_ivc* v, uint8_t* blob, size_t cap) {
  if (!v || !blob || cap < vimz_ivc_proof_size(v)) return vz_fail(v ? v->ctx : nullptr, VIMZ_ERR_INVALID, "vimz_ivc_proof_export: buffer too small");
  vimz_ctx* ctx = v->ctx; vimz_prover* p = v->pri; SecDev& S = v->sec;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  ProofHeader h{PROOF_MAGIC, v->i, p->n_wires, p->n_c, S.n_w, S.n_c, p->len_z, (uint64_t)(v->sec_T_valid ? 1 : 0)};
  ProofHost hs; memset(&hs, 0, sizeof(hs));
  hs.U2 = v->U2; hs.U1 = v->U1; hs.u2 = v->u2; hs.T2 = v->T2; hs.u1_run = v->u1_run; hs.u2_run = v->u2_run;
  hs.digest1 = v->c1->digest; hs.digest2 = v->c2.digest;
  uint8_t* o = blob;
  memcpy(o, &h, sizeof(h)); o += sizeof(h);
  memcpy(o, &hs, sizeof(hs)); o += sizeof(hs);
  memcpy(o, v->z0.data(), 32 * p->len_z); o += 32 * p->len_z;
  memcpy(o, p->z_cur.data(), 32 * p->len_z); o += 32 * p->len_z;
  const uint32_t* src[] = {p->Zrun, p->E, p->AZ, p->BZ, p->CZ, S.Zrun, S.z2, S.E, S.AZ, S.BZ, S.CZ, S.az2, S.bz2, S.cz2, S.T};
  const size_t len[] = {p->n_wires, p->n_c, p->n_c, p->n_c, p->n_c, S.n_w, S.n_w, S.n_c, S.n_c, S.n_c, S.n_c, S.n_c, S.n_c, S.n_c, S.n_c};
  for (int k = 0; k < 15; k++) { P_TRY(hipMemcpyAsync(o, src[k], 32 * len[k], hipMemcpyDeviceToHost, s)); o += 32 * len[k]; }
  P_TRY(hipStreamSynchronize(s));
  return VIMZ_OK;
}

int vimz_ivc_proof_import(vimz_ivc* v, const uint8_t* blob, size_t len) {
  if (!v || !blob || len < sizeof(ProofHeader)) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = v->ctx; vimz_prover* p = v->pri; SecDev& S = v->sec;
  ProofHeader h; memcpy(&h, blob, sizeof(h));
  if (h.magic != PROOF_MAGIC || h.n_w1 != p->n_wires || h.n_c1 != p->n_c || h.n_w2 != S.n_w || h.n_c2 != S.n_c || h.len_z != p->len_z || len < vimz_ivc_proof_size(v))
    return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_proof_import: the blob does not match this IVC's circuits");
  ProofHost hs; memcpy(&hs, blob + sizeof(h), sizeof(hs));
  if (!hs.digest1.eq(v->c1->digest) || !hs.digest2.eq(v->c2.digest)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_proof_import: shape digest differs");
  // The blob is untrusted: every field element in it must be below its modulus BEFORE anything of this IVC changes (the Montgomery
  // arithmetic and the raw-limb comparisons of vimz_ivc_verify assume reduced operands).
  {
    bool ok = hs.U2.W.x.is_reduced() && hs.U2.W.y.is_reduced() && hs.U2.E.x.is_reduced() && hs.U2.E.y.is_reduced() && hs.U2.u.is_reduced() &&
              hs.U1.W.x.is_reduced() && hs.U1.W.y.is_reduced() && hs.U1.E.x.is_reduced() && hs.U1.E.y.is_reduced() && hs.U1.u.is_reduced() &&
              hs.u2.W.x.is_reduced() && hs.u2.W.y.is_reduced() && hs.u2.x0.is_reduced() && hs.u2.x1.is_reduced() &&
              hs.T2.x.is_reduced() && hs.T2.y.is_reduced() && hs.u1_run.is_reduced() && hs.u2_run.is_reduced();
    const uint8_t* zp = blob + sizeof(h) + sizeof(hs);
    for (uint32_t k = 0; k < 2 * p->len_z && ok; k++) { Fe z; memcpy(z.v, zp + 32 * k, 32); ok = z.is_reduced(); }
    if (!ok) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_proof_import: a field element of the blob is not below its modulus");
  }
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const uint8_t* o = blob + sizeof(h) + sizeof(hs) + 64 * (size_t)p->len_z;
  // device vectors: staged, range-checked on the GPU, and only then copied over the IVC's own
  const size_t ln[] = {p->n_wires, p->n_c, p->n_c, p->n_c, p->n_c, S.n_w, S.n_w, S.n_c, S.n_c, S.n_c, S.n_c, S.n_c, S.n_c, S.n_c, S.n_c};
  const size_t vec_bytes = proof_vec_bytes(v);
  int rc = vz_ensure_scratch(ctx, vec_bytes + 64); if (rc) return rc;
  uint32_t* stage = (uint32_t*)ctx->scratch;
  uint32_t* badc = stage + vec_bytes / 4;
  P_TRY(hipMemcpyAsync(stage, o, vec_bytes, hipMemcpyHostToDevice, s));
  P_TRY(hipMemsetAsync(badc, 0, 8, s));
  {
    size_t off = 0;
    for (int k = 0; k < 15; k++) {
      if (k < 5) hipLaunchKernelGGL(k_count_unreduced<Fr>, dim3(stream_grid(ln[k])), dim3(256), 0, s, ln[k], (const uint32_t*)(stage + off), badc);
      else hipLaunchKernelGGL(k_count_unreduced<Fq>, dim3(stream_grid(ln[k])), dim3(256), 0, s, ln[k], (const uint32_t*)(stage + off), badc);
      off += 8 * ln[k];
    }
  }
  uint32_t nbad = 0;
  P_TRY(hipMemcpyAsync(&nbad, badc, 4, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (nbad) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_proof_import: a vector element of the blob is not below its modulus");
  // commit
  uint32_t* dst[] = {p->Zrun, p->E, p->AZ, p->BZ, p->CZ, S.Zrun, S.z2, S.E, S.AZ, S.BZ, S.CZ, S.az2, S.bz2, S.cz2, S.T};
  { size_t off = 0; for (int k = 0; k < 15; k++) { P_TRY(hipMemcpyAsync(dst[k], stage + off, 32 * ln[k], hipMemcpyDeviceToDevice, s)); off += 8 * ln[k]; } }
  P_TRY(hipStreamSynchronize(s));
  const uint8_t* zp = blob + sizeof(h) + sizeof(hs);
  memcpy(v->z0.data(), zp, 32 * p->len_z);
  memcpy(p->z_cur.data(), zp + 32 * p->len_z, 32 * p->len_z);
  p->z0 = v->z0;
  v->i = h.steps; p->steps = h.steps;
  v->ca_valid = false;
  v->U2 = hs.U2; v->U1 = hs.U1; v->u2 = hs.u2; v->T2 = hs.T2; v->u1_run = hs.u1_run; v->u2_run = hs.u2_run;
  v->sec_T_valid = (h.flags & 1) != 0; v->pending_sec = false; v->t1[0].step = v->t1[1].step = -1; v->broken = false;
  v->c1->cache = aug::AugCache<Fe>(); v->c2.cache = aug::AugCache<Fq>();
  return VIMZ_OK;
}

}  // extern "C"
