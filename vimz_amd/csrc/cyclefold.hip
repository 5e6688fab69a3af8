// Nova + CycleFold IVC on the GPU: the `prove_step` loop of the reference's Sonobe backend (vimz/src/sonobe_backend/folding.rs:52-66,
// `Nova<G1, G2, C, KZG, Pedersen, false>` of folding.rs:22) — SURVEY.md §8 row N1.  Circuits: aug/cyclefold.hpp (ours; Sonobe is not
// vendored).  Same kernels as the Nova IVC of ivc.hip: the batch producer fills the step circuit's part of every row's witness,
// commits to it and multiplies the step rows; MSM / SpMV / cross term / folds of prover_internal.hpp and r1cs_ops.hpp; the CycleFold
// circuit's instances (1.4 k constraints over Fq, committed on Grumpkin) run through the secondary-side instantiations.
//
// Step i (z_i -> z_{i+1}), given the running pair (U_i, W_i), the incoming pair (u_i, w_i) = the instance of F' made by step i-1, and the
// running CycleFold pair (cfU_i, cfW_i):
//   1. T = cross term of (U_i, W_i) and (u_i, w_i), cmT = MSM(T)                                      [the one large MSM of a step]
//   2. r = the challenge F' derives;  (U_{i+1}, W_{i+1}) = fold;  the commitments W' = W + r·W_in, E' = E + r·cmT on the host
//   3. two CycleFold instances (r, U.W, u.W, W') and (r, U.E, cmT, E'): witness on the host, commitment, (A,B,C)·z and cross term
//      against the running CycleFold instance on the GPU, challenge, fold — one after the other
//   4. F' of step i on the host (verifies 1-3 in-circuit, hashes the new running instances), uploaded behind the step circuit's wires
//      of the row; verifier rows of (A,B,C)·z and the commitment to the verifier wires on the GPU  ->  (u_{i+1}, w_{i+1}).
// This is a straight, synchronous schedule (no lookahead, no work queued from inside the circuit evaluation): a first version of the
// row, measured in DESIGN.md §9d, not a tuned one.
#include "cyclefold_internal.hpp"
#ifdef VIMZ_TESTING
#include "../../include/vimz_hip_testing.h"
#endif

namespace {

struct CfWitness { std::vector<Fq> wires; G1Aff P3; bool bad = false; };
// The two CycleFold instances of a step: commitments, (A,B,C)·z, cross terms against the running CycleFold instance, challenges, folds.
// Both witness commitments run side by side on stream 2; on the main stream the chain is cross term 1 -> commitment -> challenge -> fold ->
// cross term 2 (against the folded instance) -> commitment -> challenge -> fold.  second(): blocks until the second witness exists.
template <class WaitSecond>
int run_cyclefold_pair(vimz_cf* v, const CfWitness& w1, const CfWitness& w2, WaitSecond second, CfChallenges& ch, CfMainIn& in) {
  vimz_ctx* ctx = v->ctx;
  hipStream_t s = ctx->stream;
  SecDev& S = v->sec;
  if (w1.bad) return vz_fail(ctx, VIMZ_ERR_UNSAT, "cyclefold circuit: a commitment to fold is not on the curve");
  char* pin_cf = v->pin + 4 * v->pin_res;
  char* pin_cf2 = pin_cf + 32 * (size_t)S.n_w;
  const size_t nwit = S.n_w - 1 - CF_IO;
  const BaseTables* tb = v->tb_ck2.d ? &v->tb_ck2 : nullptr;
  auto cross = [&](const uint32_t* z, uint32_t* az, uint32_t* bz, uint32_t* cz, bool have_run) {
    hipLaunchKernelGGL(k_spmv_cross16<Fq>, dim3((unsigned)((16 * (size_t)S.n_c + 255) / 256)), dim3(256), 0, s, S.A, S.B, S.C, S.dict, 0u, S.n_c, z, az, bz, cz,
                       have_run ? S.AZ : nullptr, S.BZ, S.CZ, v->cf_u_run, Fq::one(), S.T);
  };
  auto fold = [&](const uint32_t* z, const uint32_t* az, const uint32_t* bz, const uint32_t* cz, bool have_run, const Fq& rq) {
    Fold5 f;
    f.x1[0] = S.Zrun; f.x2[0] = z; f.n[0] = S.n_w;
    f.x1[1] = have_run ? S.E : nullptr; f.x2[1] = S.T; f.n[1] = S.n_c;
    f.x1[2] = S.AZ; f.x2[2] = az; f.n[2] = S.n_c;
    f.x1[3] = S.BZ; f.x2[3] = bz; f.n[3] = S.n_c;
    f.x1[4] = S.CZ; f.x2[4] = cz; f.n[4] = S.n_c;
    hipLaunchKernelGGL(k_fold5<Fq>, dim3(64), dim3(256), 0, s, f, rq);
    v->cf_u_run = Fq::add(v->cf_u_run, rq);
  };
  // first instance
  memcpy(pin_cf, w1.wires.data(), 32 * (size_t)S.n_w);
  P_TRY(upload_pinned(s, S.z2, pin_cf, 32 * (size_t)S.n_w));
  P_TRY(hipEventRecord(v->ev_fork, s));
  P_TRY(hipStreamWaitEvent(v->s2, v->ev_fork, 0));
  P_TRY(msm_launch<Grumpkin>(v->s2, v->ws2, v->ck2->d, S.z2 + 8, nwit, 1, 0, v->pin + 2 * v->pin_res, &v->plan_cfW, nullptr, 0, tb));
  const bool run1 = !v->cf_u_run.is_zero();        // the running instance is the zero instance until the first fold: no cross term
  cross(S.z2, S.az2, S.bz2, S.cz2, run1);
  P_TRY(hipGetLastError());
  if (run1) P_TRY(msm_launch<Grumpkin>(s, ctx->msm_ws, v->ck2->d, S.T, S.n_c, 1, 0, v->pin + 3 * v->pin_res, &v->plan_cfT, nullptr, 0, tb));
  // second instance: its witness commitment needs the upload only — on its own stream, next to the first one's
  second();
  if (w2.bad) return vz_fail(ctx, VIMZ_ERR_UNSAT, "cyclefold circuit: a commitment to fold is not on the curve");
  memcpy(pin_cf2, w2.wires.data(), 32 * (size_t)S.n_w);
  P_TRY(upload_pinned(v->s4, v->z3, pin_cf2, 32 * (size_t)S.n_w));
  P_TRY(hipEventRecord(v->ev_z3, v->s4));
  P_TRY(msm_launch<Grumpkin>(v->s4, v->ws4, v->ck2->d, v->z3 + 8, nwit, 1, 0, v->pin_w2, &v->plan_cfW2, nullptr, 0, tb));
  P_TRY(hipStreamSynchronize(v->s2));
  in.cf1W = msm_finish<Grumpkin>(v->plan_cfW, v->pin + 2 * v->pin_res);
  P_TRY(hipStreamSynchronize(s));
  in.cf1T = run1 ? msm_finish<Grumpkin>(v->plan_cfT, v->pin + 3 * v->pin_res) : g2_identity();
  cf_challenge_cf1(ch, in.cf1W, in.Wn, in.cf1T);
  fold(S.z2, S.az2, S.bz2, S.cz2, run1, rho_element<Fq>(ch.r1));
  P_TRY(hipStreamWaitEvent(s, v->ev_z3, 0));
  cross(v->z3, v->az3, v->bz3, v->cz3, true);
  P_TRY(hipGetLastError());
  P_TRY(msm_launch<Grumpkin>(s, ctx->msm_ws, v->ck2->d, S.T, S.n_c, 1, 0, v->pin + 3 * v->pin_res, &v->plan_cfT, nullptr, 0, tb));
  P_TRY(hipStreamSynchronize(v->s4));
  in.cf2W = msm_finish<Grumpkin>(v->plan_cfW2, v->pin_w2);
  P_TRY(hipStreamSynchronize(s));
  in.cf2T = msm_finish<Grumpkin>(v->plan_cfT, v->pin + 3 * v->pin_res);
  cf_challenge_cf2(ch, in.cf2W, in.En, in.cf2T);
  fold(v->z3, v->az3, v->bz3, v->cz3, true, rho_element<Fq>(ch.r2));
  P_TRY(hipGetLastError());
  return VIMZ_OK;
}

int cf_fold_core(vimz_cf* v, const uint64_t* step_inputs, size_t nsteps) {
  if (!nsteps) return VIMZ_OK;
  vimz_ctx* ctx = v->ctx;
  if (v->broken) return vz_fail(ctx, VIMZ_ERR_INVALID, "this CycleFold IVC failed in the middle of a step and cannot be folded any further");
  vimz_prover* p = v->pri;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t nw = p->n_wires, nc = p->n_c, sw = p->step_wires, sc = p->step_c;
  const size_t aw = v->c1->aug_wires();
  int rc;
  const double t_all = now_s();
  FoldJob job; job.step_inputs = step_inputs; job.nsteps = nsteps;
  if ((rc = fold_prepare(p, job, true))) return rc;
  struct BrokenGuard { vimz_cf* v; bool armed = true; ~BrokenGuard() { if (armed) v->broken = true; } } guard{v};
  struct Ahead { vimz_cf* v; ~Ahead() { hipStreamSynchronize(v->s3); v->t_step_for = v->t_ver_for = -1; } } ahead{v};      // nothing queued ahead survives the call
  const std::vector<Fe>& zs = job.zs;
  const size_t pin_stride = FoldJob::pin_stride;
  char* pin_aug = v->pin + 4 * v->pin_res + 64 * (size_t)v->sec.n_w;
  if (!ctx->msm_ws.host_pinned) P_TRY(hipHostMalloc(&ctx->msm_ws.host_pinned, 4 * XYZZ_WORDS * MSM_MAX_WINDOWS));
  for (size_t k = 0; k < job.nbatches; k++) {
    auto& bb = p->buf[k & 1];
    const size_t first = job.first(k), rows = job.rows(k);
    double t0 = now_s();
    if ((rc = fold_issue_when_ready(p, job, k, true))) return rc;
    if ((rc = fold_issue_when_ready(p, job, k + 1, false))) return rc;
    P_TRY(hipEventSynchronize(bb.wit_done));
    v->ph_s[CP_PRODUCER] += now_s() - t0;
    for (size_t r = 0; r < rows; r++) if (bb.status_host[r]) {
      char msg[128]; snprintf(msg, sizeof(msg), "step %llu: the step relation is not satisfiable for these rows", (unsigned long long)(v->i + r));
      hipStreamSynchronize(p->sB);
      for (uint32_t q = 0; q < p->len_z; q++) p->z_cur[q] = zs[first * p->len_z + q];      // the batches folded so far stay folded
      guard.armed = false;
      return vz_fail(ctx, VIMZ_ERR_UNSAT, msg);
    }
    for (size_t r = 0; r < rows; r++) {
      const uint64_t i = v->i;
      uint32_t* Zi = bb.Z + 8 * r * nw;
      uint32_t *az = bb.az + 8 * r * nc, *bz = bb.bz + 8 * r * nc, *cz = bb.cz + 8 * r * nc;
      const Fe* z_i = zs.data() + (first + r) * p->len_z;
      const Fe* z_n = zs.data() + (first + r + 1) * p->len_z;
      CfMainIn in = CfMainIn::zero();
      in.digest = v->c1->digest; in.i = i; in.z0 = v->z0; in.U = v->U; in.u = v->u; in.cfU = v->cfU;
      CfChallenges ch;
      // (the hashes of the two running instances are the public IO of the incoming instance — F' of the previous step computed them)
      ch.h_U = i > 0 ? v->u.x0 : cf_hash_main(in.digest, i, v->z0, z_i, v->U);
      ch.h_cf = i > 0 ? v->u.x1 : cf_hash_cf(in.digest, v->cfU);
      G1Aff Wn = g1_identity(), En = g1_identity();
      if (i > 0) {
        // ---- 1. cross term of the running and the incoming pair, and its commitment ------------------------------------------------
        t0 = now_s();
        const uint32_t *pZ = r ? bb.Z + 8 * (r - 1) * nw : v->Zl, *paz = r ? bb.az + 8 * (r - 1) * nc : v->azl, *pbz = r ? bb.bz + 8 * (r - 1) * nc : v->bzl,
                       *pcz = r ? bb.cz + 8 * (r - 1) * nc : v->czl;
        const bool have_run = !v->u_run.is_zero();       // step 1 folds into the zero instance: no cross term
        G1Aff cT = g1_identity();
        if (have_run && v->t_step_for == (int64_t)i && v->t_ver_for == (int64_t)i) {      // both shares were queued by the previous step
          P_TRY(hipEventSynchronize(v->ev_ts));
          const G1Aff c_step = msm_finish<BnG1>(v->plan_Ts, v->pin_ts);
          P_TRY(hipStreamSynchronize(s));
          const G1Aff c_ver = msm_finish<BnG1>(v->plan_Tv, v->pin + v->pin_res);
          G1 sum = aff_is_identity(c_step) ? G1::identity() : from_affine(c_step);
          add_mixed(sum, c_ver);
          cT = to_affine(sum);
        } else if (have_run) {                                                                 // first row of a call: all rows at once
          P_TRY(hipStreamSynchronize(v->s3));
          hipLaunchKernelGGL(k_cross_term<Fr>, dim3(stream_grid(nc)), dim3(256), 0, s, nc, p->AZ, p->BZ, p->CZ, v->u_run, paz, pbz, pcz, Fe::one(), p->T);
          P_TRY(hipGetLastError());
          P_TRY(msm_launch<BnG1>(s, ctx->msm_ws, p->ck->d, p->T, nc, 1, 0, ctx->msm_ws.host_pinned, &v->plan_T, nullptr, 0, p->ck->tables ? &job.tbl : nullptr));
          P_TRY(hipStreamSynchronize(s));
          cT = msm_finish<BnG1>(v->plan_T, ctx->msm_ws.host_pinned);
        }
        v->t_step_for = v->t_ver_for = -1;
        v->ph_s[CP_CROSS_MSM] += now_s() - t0; v->ph_n[CP_CROSS_MSM]++;
        // ---- 2. challenge, fold of the main pair -------------------------------------------------------------------------------------
        t0 = now_s();
        in.T = nn_point(cT);
        cf_challenge_main(ch, v->u, in.T);
        const Fe rho = cf_r_element_fr(ch.r);
        Fold5 f;
        f.x1[0] = p->Zrun; f.x2[0] = pZ; f.n[0] = nw;
        f.x1[1] = have_run ? p->E : nullptr; f.x2[1] = p->T; f.n[1] = nc;
        f.x1[2] = p->AZ; f.x2[2] = paz; f.n[2] = nc;
        f.x1[3] = p->BZ; f.x2[3] = pbz; f.n[3] = nc;
        f.x1[4] = p->CZ; f.x2[4] = pcz; f.n[4] = nc;
        hipLaunchKernelGGL(k_fold5<Fr>, dim3(2048), dim3(256), 0, s, f, rho);
        P_TRY(hipGetLastError());
        v->u_run = Fe::add(v->u_run, rho);
        // the next step folds THIS row's instance: the step rows of that cross term and their commitment start now, on stream 3
        P_TRY(hipEventRecord(v->ev_fold, s));
        P_TRY(hipStreamWaitEvent(v->s3, v->ev_fold, 0));
        { const double tw = now_s(); P_TRY(wait_row_flag(bb, r, bb.ev[r])); v->ph_s[6] += now_s() - tw; }
        hipLaunchKernelGGL(k_cross_term<Fr>, dim3(stream_grid(sc)), dim3(256), 0, v->s3, sc, p->AZ, p->BZ, p->CZ, v->u_run, az, bz, cz, Fe::one(), p->T);
        P_TRY(hipGetLastError());
        P_TRY(msm_launch<BnG1>(v->s3, v->ws3, p->ck->d, p->T, sc, 1, 0, v->pin_ts, &v->plan_Ts, nullptr, 0, p->ck->tables ? &job.tbl : nullptr));
        P_TRY(hipEventRecord(v->ev_ts, v->s3));
        v->t_step_for = (int64_t)i + 1;
        // ---- 3. the two CycleFold instances -------------------------------------------------------------------------------------------
        // (both witnesses need the challenge only: the second is computed on a helper thread while the first instance is on the GPU)
        CfWitness w1, w2;
        aug::Worker* helper = v->c1->use_worker ? v->c1->worker.get() : nullptr;
        const G1Aff UE_now = v->UE;
        auto make2 = [&] { w2.P3 = v->cf.witness(ch.r, UE_now, cT, w2.wires, &w2.bad); };
        if (helper) helper->start(make2);
        w1.P3 = v->cf.witness(ch.r, v->UW, v->uW, w1.wires, &w1.bad);
        // (the folded commitments W' = W + r·W_in and E' = E + r·cmT are what the two witnesses end in: no second scalar multiplication)
        Wn = w1.P3; in.Wn = nn_point(Wn);
        bool waited = false;
        auto second = [&] { if (waited) return; waited = true; if (helper) helper->wait(); else make2(); En = w2.P3; in.En = nn_point(En); };
        rc = run_cyclefold_pair(v, w1, w2, second, ch, in);
        second();                                        // (an early error return must not leave the helper running on this frame)
        if (rc) return rc;
        v->ph_s[CP_CF] += now_s() - t0; v->ph_n[CP_CF]++;
      }
      // ---- 4. F' of this step on the host ---------------------------------------------------------------------------------------------------
      t0 = now_s();
      std::vector<Fe> aug; bool bad = false;
      CfMainOut o = v->c1->witness(in, z_i, z_n, aug, &bad);
      if (bad) return vz_fail(ctx, VIMZ_ERR_UNSAT, "cyclefold main circuit: inconsistent incoming instance");
      if (i > 0 && (memcmp(o.r, ch.r, 16) || memcmp(o.r1, ch.r1, 16) || memcmp(o.r2, ch.r2, 16)))
        return vz_fail(ctx, VIMZ_ERR_INVALID, "cyclefold main circuit: its challenges differ from the prover's");
      v->ph_s[CP_SYNTH] += now_s() - t0; v->ph_n[CP_SYNTH]++;
      t0 = now_s();
      v->last_in = in; v->last_out = o; v->last_zi.assign(z_i, z_i + p->len_z); v->last_zn.assign(z_n, z_n + p->len_z); v->have_last = true;
      v->U = o.U_new; v->UW = i > 0 ? Wn : g1_identity(); v->UE = i > 0 ? En : g1_identity();
      v->cfU = o.cfU_new;
      // ---- fresh main instance: verifier wires behind the step circuit's, verifier rows of (A,B,C)·z, commitment ---------------------------
      memcpy(pin_aug, aug.data(), 32 * aw);
      // (waited for on the HOST, not by a barrier on this stream: ivc.hip)
      { const double tw = now_s(); P_TRY(wait_row_flag(bb, r, bb.ev[r])); v->ph_s[6] += now_s() - tw; }
      P_TRY(upload_pinned(s, Zi + 8 * sw, pin_aug, 32 * aw));
      P_TRY(hipEventRecord(v->ev_fork, s));
      P_TRY(hipStreamWaitEvent(v->s2, v->ev_fork, 0));
      P_TRY(msm_launch<BnG1>(v->s2, v->ws2, p->ck->d + (size_t)AFFINE_WORDS * (sw - 1), Zi + 8 * sw, aw - 2, 1, 0, v->pin, &v->plan_aug, nullptr, 0, nullptr));
      // (the verifier rows' products, and with them those rows' share of the next step's cross term and its commitment)
      const bool cross_ahead = v->t_step_for == (int64_t)i + 1;
      hipLaunchKernelGGL(k_spmv_cross16<Fr>, dim3((unsigned)((16 * (nc - sc) + 255) / 256)), dim3(256), 0, s, p->A, p->B, p->C, p->dict, (uint32_t)sc, (uint32_t)(nc - sc),
                         Zi, az, bz, cz, cross_ahead ? (const uint32_t*)p->AZ : (const uint32_t*)nullptr, p->BZ, p->CZ, v->u_run, Fe::one(), p->T);
      P_TRY(hipGetLastError());
      if (cross_ahead) {      // (same stream: it follows the kernel that writes its scalars; read back at the start of the next step)
        P_TRY(msm_launch<BnG1>(s, ctx->msm_ws, p->ck->d + (size_t)AFFINE_WORDS * sc, p->T + 8 * sc, nc - sc, 1, 0, v->pin + v->pin_res, &v->plan_Tv, nullptr, 0, nullptr));
        v->t_ver_for = (int64_t)i + 1;
      }
      { const double tw = now_s();
        P_TRY(wait_row_flag(bb, r, bb.ev[r]));      // (a pinned word, not the producer's event: prover_internal.hpp)
        v->ph_s[6] += now_s() - tw; }
      const G1Aff cW_step = msm_finish<BnG1>(p->planB, (char*)bb.pin + r * pin_stride);
      { const double tw = now_s(); P_TRY(hipStreamSynchronize(v->s2)); v->ph_s[7] += now_s() - tw; }
      const G1Aff cW_aug = msm_finish<BnG1>(v->plan_aug, v->pin);
      G1 sum = from_affine(cW_step); if (aff_is_identity(cW_step)) sum = G1::identity();
      add_mixed(sum, cW_aug);
      v->uW = to_affine(sum);
      v->u.W = nn_point(v->uW); v->u.x0 = o.x0; v->u.x1 = o.x1;
      v->ph_s[CP_FRESH] += now_s() - t0; v->ph_n[CP_FRESH]++;
      v->i++; p->steps++;
    }
    // the incoming pair of the next step outlives this batch's buffer
    const size_t lr = rows - 1;
    P_TRY(hipMemcpyAsync(v->Zl, bb.Z + 8 * lr * nw, 32 * nw, hipMemcpyDeviceToDevice, s));
    P_TRY(hipMemcpyAsync(v->azl, bb.az + 8 * lr * nc, 32 * nc, hipMemcpyDeviceToDevice, s));
    P_TRY(hipMemcpyAsync(v->bzl, bb.bz + 8 * lr * nc, 32 * nc, hipMemcpyDeviceToDevice, s));
    P_TRY(hipMemcpyAsync(v->czl, bb.cz + 8 * lr * nc, 32 * nc, hipMemcpyDeviceToDevice, s));
    P_TRY(hipStreamSynchronize(s));
    P_TRY(hipStreamSynchronize(v->s2));
  }
  P_TRY(hipStreamSynchronize(p->sB));
  for (uint32_t k = 0; k < p->len_z; k++) p->z_cur[k] = zs[nsteps * p->len_z + k];
  guard.armed = false;
  v->ph_s[CP_TOTAL] += now_s() - t_all; v->ph_n[CP_TOTAL] += nsteps;
  return VIMZ_OK;
}

// the words of VIMZ_IX_LAST_STEP: what a step's F' was given and what it returned
std::vector<uint64_t> last_step_words(const CfMainIn& L, const CfMainOut& O, const std::vector<Fe>& zi, const std::vector<Fe>& zn) {
  std::vector<uint64_t> o;
  auto push = [&](const auto& m) { auto x = std::decay_t<decltype(m)>::from_mont(m); o.resize(o.size() + 4); memcpy(o.data() + o.size() - 4, x.v, 32); };
  auto push_u = [&](const U256w& x) { o.insert(o.end(), x.w, x.w + 4); };
  auto push_i = [&](uint64_t x) { o.push_back(x); o.push_back(0); o.push_back(0); o.push_back(0); };
  auto push_low = [&](const uint32_t* r) { o.push_back((uint64_t)r[0] | ((uint64_t)r[1] << 32)); o.push_back((uint64_t)r[2] | ((uint64_t)r[3] << 32)); o.push_back(0); o.push_back(0); };
  auto push_main = [&](const CfMainRelaxed& U) { push_u(U.W.x); push_u(U.W.y); push_u(U.E.x); push_u(U.E.y); push(U.u); push(U.x0); push(U.x1); };
  auto push_cf = [&](const CfRelaxed& U) { push(U.W.x); push(U.W.y); push(U.E.x); push(U.E.y); push(U.u); for (auto& e : U.x) push_u(e); };
  push_i(L.i);
  for (auto& z : zi) push(z);
  for (auto& z : zn) push(z);
  push_main(L.U); push_u(L.u.W.x); push_u(L.u.W.y); push(L.u.x0); push(L.u.x1);
  push_u(L.T.x); push_u(L.T.y); push_u(L.Wn.x); push_u(L.Wn.y); push_u(L.En.x); push_u(L.En.y);
  push_cf(L.cfU);
  for (const G2Aff* P : {&L.cf1W, &L.cf1T, &L.cf2W, &L.cf2T}) { push(P->x); push(P->y); }
  push_main(O.U_new); push_cf(O.cfU_new); push(O.x0); push(O.x1);
  push_low(O.r); push_low(O.r1); push_low(O.r2);
  return o;
}

template <class F>
int64_t export_builder(const cb::BuilderT<F>& b, uint32_t step_wires, uint32_t step_c, int what, void* buf, size_t cap) {
  const void* src = nullptr; size_t bytes = 0;
  auto vec = [&](const auto& v) { src = v.data(); bytes = v.size() * sizeof(v[0]); };
  switch (what) {
    case VIMZ_CX_A_ROWPTR: vec(b.A.row_ptr); break;
    case VIMZ_CX_A_COL: vec(b.A.col); break;
    case VIMZ_CX_A_COEF: vec(b.A.coef); break;
    case VIMZ_CX_B_ROWPTR: vec(b.B.row_ptr); break;
    case VIMZ_CX_B_COL: vec(b.B.col); break;
    case VIMZ_CX_B_COEF: vec(b.B.coef); break;
    case VIMZ_CX_C_ROWPTR: vec(b.C.row_ptr); break;
    case VIMZ_CX_C_COL: vec(b.C.col); break;
    case VIMZ_CX_C_COEF: vec(b.C.coef); break;
    case VIMZ_CX_DICT_MONT: vec(b.dict); break;
    case VIMZ_CX_DICT_CANON: {
      bytes = b.dict.size() * 32;
      if (buf && cap >= bytes) { F* o = (F*)buf; for (size_t i = 0; i < b.dict.size(); i++) o[i] = F::from_mont(b.dict[i]); }
      return (int64_t)bytes;
    }
    case VIMZ_IX_INFO: {
      if (buf && cap >= 32) { uint64_t o[4] = {b.n_wires, b.n_constraints(), step_wires, step_c}; memcpy(buf, o, 32); }
      return 32;
    }
    default: return VIMZ_ERR_INVALID;
  }
  if (buf && cap >= bytes && bytes) memcpy(buf, src, bytes);
  return (int64_t)bytes;
}

}  // namespace

extern "C" {

void vimz_cf_free(vimz_cf* v) {
  if (!v) return;
  if (v->orphan_merged) v->orphan_merged(v);
  if (v->pri) vimz_prover_free(v->pri);
  if (v->ctx) {
    std::lock_guard<std::mutex> g(v->ctx->mu);
    hipSetDevice(v->ctx->device);
    hipStreamSynchronize(v->ctx->stream);
    { int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
      vz_stream_release(v->ctx, hi, v->s2); vz_stream_release(v->ctx, (lo + hi) / 2, v->s3); vz_stream_release(v->ctx, hi, v->s4); }
    if (v->ev_z3) hipEventDestroy(v->ev_z3);
    if (v->pin_w2) hipHostFree(v->pin_w2);
    v->ws4.release();
    if (v->ev_fork) hipEventDestroy(v->ev_fork);
    if (v->ev_fold) hipEventDestroy(v->ev_fold);
    if (v->ev_ts) hipEventDestroy(v->ev_ts);
    if (v->pin_ts) hipHostFree(v->pin_ts);
    v->ws2.release(); v->ws3.release();
    for (void* d : v->owned) hipFree(d);
    if (v->pin) hipHostFree(v->pin);
  }
  delete v;
}

int vimz_cf_create(vimz_ctx* ctx, const vimz_circuit* step_circuit, const vimz_bases* ck1, const vimz_bases* ck2, size_t max_batch, vimz_cf** out) {
  if (!ctx || !step_circuit || !ck1 || !ck2 || !out || max_batch == 0) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_create: bad argument");
  if (ck1->curve != VIMZ_CURVE_BN254_G1 || ck2->curve != VIMZ_CURVE_GRUMPKIN) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_create: keys must be on BN254 G1 (main) and Grumpkin (CycleFold)");
  std::unique_ptr<vimz_cf> v(new vimz_cf());
  v->ctx = ctx; v->ck1 = ck1; v->ck2 = ck2;
  try {
    v->cf.finish();
    v->circ.reset(new vimz_circuit());
    v->circ->transformation = step_circuit->transformation; v->circ->shape = step_circuit->shape;
    v->circ->build.reset(new cb::CircuitBuild());
    v->circ->build->b = step_circuit->build->b;
    v->c1.reset(new CfMainCircuit(v->circ->build->b));
    v->c1->finish(v->cf);
    if (v->c1->use_worker) { v->c1->worker.reset(new aug::Worker()); v->c1->worker2.reset(new aug::Worker()); }
  } catch (const std::exception& e) { return vz_fail(ctx, VIMZ_ERR_INVALID, e.what()); }
  const uint32_t nw2 = v->cf.n_wires(), nc2 = v->cf.n_constraints();
  if (ck2->n < nw2 || ck2->n < nc2) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_create: CycleFold commitment key shorter than the CycleFold circuit");
  int rc = vz_prover_create_layout(ctx, v->circ.get(), ck1, max_batch, 1, v->c1->step_wires, v->c1->step_constraints, &v->pri);
  if (rc) return rc;
  std::unique_lock<std::mutex> lk(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  SecDev& S = v->sec;
  S.n_w = nw2; S.n_c = nc2;
  const cb::BuilderT<Fq>& b2 = v->cf.b;
  hipError_t e = hipSuccess;
  auto fail = [&](const char* what) { const hipError_t ee = e; lk.unlock(); vimz_cf_free(v.release()); return vz_fail(ctx, VIMZ_ERR_HIP, what, ee); };
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
  const size_t nw = v->circ->build->b.n_wires, nc = v->circ->build->b.n_constraints();
  if (dalloc(&v->Zl, 32 * nw) != hipSuccess) return fail("device allocation");
  for (auto d : {&v->azl, &v->bzl, &v->czl}) if (dalloc(d, 32 * nc) != hipSuccess) return fail("device allocation");
  if (dalloc(&v->z3, 32 * (size_t)nw2) != hipSuccess) return fail("device allocation");
  for (auto d : {&v->az3, &v->bz3, &v->cz3}) if (dalloc(d, 32 * (size_t)nc2) != hipSuccess) return fail("device allocation");
  { int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
    if ((e = vz_stream_acquire(ctx, hi, &v->s2)) != hipSuccess) return fail("stream");
    if ((e = vz_stream_acquire(ctx, (lo + hi) / 2, &v->s3)) != hipSuccess) return fail("stream");
    if ((e = vz_stream_acquire(ctx, hi, &v->s4)) != hipSuccess) return fail("stream");
    if ((e = hipEventCreateWithFlags(&v->ev_z3, hipEventDisableTiming)) != hipSuccess) return fail("event");
    if ((e = hipEventCreateWithFlags(&v->ev_fork, hipEventDisableTiming)) != hipSuccess) return fail("event");
    if ((e = hipEventCreateWithFlags(&v->ev_fold, hipEventDisableTiming)) != hipSuccess) return fail("event");
    if ((e = hipEventCreateWithFlags(&v->ev_ts, hipEventDisableTiming)) != hipSuccess) return fail("event"); }
  if ((e = hipHostMalloc((void**)&v->pin_ts, 4 * (size_t)XYZZ_WORDS * MSM_MAX_WINDOWS)) != hipSuccess) return fail("pinned");
  if ((e = hipHostMalloc((void**)&v->pin_w2, 4 * (size_t)XYZZ_WORDS * MSM_MAX_WINDOWS)) != hipSuccess) return fail("pinned");
  if ((rc = vz_small_tables(ctx, const_cast<vimz_bases*>(ck2), 0, std::max<size_t>(nw2 - 1 - CF_IO, nc2), false, &v->tb_ck2))) { lk.unlock(); vimz_cf_free(v.release()); return rc; }
  v->pin_res = 4 * (size_t)XYZZ_WORDS * MSM_MAX_WINDOWS;
  if ((e = hipHostMalloc((void**)&v->pin, 4 * v->pin_res + 64 * (size_t)nw2 + 32 * (size_t)v->c1->aug_wires() + 64)) != hipSuccess) return fail("pinned");
  if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return fail("sync");
  v->z0.assign(v->c1->len_z, Fe::zero());
  v->U = CfMainRelaxed::zero(); v->u = CfMainFresh::zero(); v->cfU = CfRelaxed::zero();
  v->UW = v->UE = v->uW = g1_identity();
  *out = v.release();
  return VIMZ_OK;
}

int vimz_cf_reset(vimz_cf* v, const uint64_t* z0) {
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
  P_TRY(hipMemsetAsync(v->Zl, 0, 32 * (size_t)v->pri->n_wires, s));
  for (auto d : {v->azl, v->bzl, v->czl}) P_TRY(hipMemsetAsync(d, 0, 32 * (size_t)v->pri->n_c, s));
  P_TRY(hipStreamSynchronize(s));
  v->i = 0;
  for (uint32_t k = 0; k < v->c1->len_z; k++) v->z0[k] = v->pri->z_cur[k];
  v->U = CfMainRelaxed::zero(); v->u = CfMainFresh::zero(); v->cfU = CfRelaxed::zero();
  v->UW = v->UE = v->uW = g1_identity();
  v->u_run = Fe::zero(); v->cf_u_run = Fq::zero(); v->broken = false; v->t_step_for = v->t_ver_for = -1;
  memset(v->ph_s, 0, sizeof(v->ph_s)); memset(v->ph_n, 0, sizeof(v->ph_n));
  return VIMZ_OK;
}

int vimz_cf_fold(vimz_cf* v, const uint64_t* step_inputs, size_t nsteps) {
  if (!v || (!step_inputs && nsteps)) return VIMZ_ERR_INVALID;
  const cb::Builder& b = v->circ->build->b;
  if ((b.zout.empty() || !b.gpu_witness) && nsteps) return vz_fail(v->ctx, VIMZ_ERR_INVALID, "vimz_cf_fold: this step circuit has no GPU witness program");
  try { return cf_fold_core(v, step_inputs, nsteps); } catch (const std::exception& e) { return vz_fail(v->ctx, VIMZ_ERR_INVALID, e.what()); }
}

int vimz_cf_info(const vimz_cf* v, uint64_t info[12]) {
  if (!v || !info) return VIMZ_ERR_INVALID;
  const cb::Builder& b1 = v->circ->build->b; const cb::BuilderT<Fq>& b2 = v->cf.b;
  info[0] = v->i; info[1] = b1.n_wires; info[2] = b1.n_constraints(); info[3] = v->c1->step_wires; info[4] = v->c1->step_constraints;
  info[5] = b2.n_wires; info[6] = b2.n_constraints(); info[7] = v->c1->len_z; info[8] = v->c1->aug_wires();
  info[9] = b1.A.col.size() + b1.B.col.size() + b1.C.col.size(); info[10] = b2.A.col.size() + b2.B.col.size() + b2.C.col.size(); info[11] = CF_IO;
  return VIMZ_OK;
}
int vimz_cf_state(const vimz_cf* v, uint64_t* z_current, uint64_t* steps) {
  if (!v) return VIMZ_ERR_INVALID;
  if (z_current) for (uint32_t k = 0; k < v->c1->len_z; k++) fe_to_canon(v->pri->z_cur[k], z_current + 4 * k);
  if (steps) *steps = v->i;
  return VIMZ_OK;
}
int vimz_cf_profile(const vimz_cf* v, double seconds[8], uint64_t counts[8]) {
  if (!v) return VIMZ_ERR_INVALID;
  for (int k = 0; k < 8; k++) { if (seconds) seconds[k] = v->ph_s[k]; if (counts) counts[k] = v->ph_n[k]; }
  return VIMZ_OK;
}

// Nova::verify of Sonobe's IVC proof (reached from vimz/src/sonobe_backend/folding.rs:69-75): the two hashes carried by the last
// instance of F', recomputed from the claimed statement; the running main pair (relaxed), the last pair of F' (strict), the running
// CycleFold pair (relaxed); every commitment re-opened.
int vimz_cf_verify(vimz_cf* v, uint64_t num_steps, const uint64_t* z0, uint32_t* result) {
  if (!v || !result || !z0) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = v->ctx;
  vimz_prover* p = v->pri;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  SecDev& S = v->sec;
  uint32_t res = 0;
  std::vector<Fe> z0c(p->len_z);
  for (uint32_t k = 0; k < p->len_z; k++) {
    Fe c; memcpy(c.v, z0 + 4 * k, 32);
    if (!c.is_reduced()) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_verify: z0 element not below the modulus");
    z0c[k] = Fe::to_mont(c);
  }
  if (v->i != num_steps) res |= 4096;
  for (uint32_t k = 0; k < p->len_z; k++) if (!z0c[k].eq(v->z0[k])) res |= 4096;
  if (v->i == 0) {
    for (uint32_t k = 0; k < p->len_z; k++) if (!p->z_cur[k].eq(z0c[k])) res |= 4096;
    *result = res; return VIMZ_OK;
  }
  if (!cf_hash_main(v->c1->digest, v->i, z0c, p->z_cur.data(), v->U).eq(v->u.x0)) res |= 1;
  if (!cf_hash_cf(v->c1->digest, v->cfU).eq(v->u.x1)) res |= 2;
  const uint32_t init[2] = {0, 0xffffffffu};
  uint32_t bad[2];
  uint64_t pt[8];
  int rc;
  auto same_pt = [&](const uint64_t* got, const auto& P) { return !memcmp(got, P.x.v, 32) && !memcmp(got + 4, P.y.v, 32); };
  // running main pair
  launch_spmv(p, s, p->Zrun, p->az2, p->bz2, p->cz2, 0);
  P_TRY(hipMemcpyAsync(p->bad_d, init, 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_check_relaxed<Fr>, dim3(stream_grid(p->n_c)), dim3(256), 0, s, (size_t)p->n_c, p->az2, p->bz2, p->cz2, v->u_run, (const uint32_t*)p->E, p->bad_d);
  P_TRY(hipMemcpyAsync(bad, p->bad_d, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 4;
  P_TRY(hipMemcpyAsync(p->bad_d, init, 8, hipMemcpyHostToDevice, s));
  { const uint32_t* kept[3] = {p->AZ, p->BZ, p->CZ}; const uint32_t* fresh[3] = {p->az2, p->bz2, p->cz2};
    for (int m = 0; m < 3; m++) hipLaunchKernelGGL(k_count_diff, dim3(stream_grid(p->n_c)), dim3(256), 0, s, (size_t)p->n_c, kept[m], fresh[m], p->bad_d); }
  P_TRY(hipMemcpyAsync(bad, p->bad_d, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 2048;
  if ((rc = vz_msm_device(ctx, p->ck, 0, p->Zrun + 8, p->n_wires - 3, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (!same_pt(pt, v->UW)) res |= 8;
  if ((rc = vz_msm_device(ctx, p->ck, 0, p->E, p->n_c, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (!same_pt(pt, v->UE)) res |= 16;
  {
    Fe e[3];
    if (!fetch(s, p->Zrun, 0, 1, &e[0]) || !fetch(s, p->Zrun, p->n_wires - 2, 2, &e[1])) return vz_fail(ctx, VIMZ_ERR_HIP, "verify: download");
    if (!e[0].eq(v->u_run) || !e[0].eq(v->U.u) || !e[1].eq(v->U.x0) || !e[2].eq(v->U.x1)) res |= 1024;
    const NnPoint w = nn_point(v->UW), ee = nn_point(v->UE);
    if (memcmp(&w, &v->U.W, sizeof(w)) || memcmp(&ee, &v->U.E, sizeof(ee))) res |= 1024;
  }
  // the last pair of F' (strict)
  launch_spmv(p, s, v->Zl, p->az2, p->bz2, p->cz2, 0);
  P_TRY(hipMemcpyAsync(p->bad_d, init, 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_check_relaxed<Fr>, dim3(stream_grid(p->n_c)), dim3(256), 0, s, (size_t)p->n_c, p->az2, p->bz2, p->cz2, Fe::one(), (const uint32_t*)nullptr, p->bad_d);
  P_TRY(hipMemcpyAsync(bad, p->bad_d, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 256;
  if ((rc = vz_msm_device(ctx, p->ck, 0, v->Zl + 8, p->n_wires - 3, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (!same_pt(pt, v->uW)) res |= 512;
  {
    Fe e[3];
    if (!fetch(s, v->Zl, 0, 1, &e[0]) || !fetch(s, v->Zl, p->n_wires - 2, 2, &e[1])) return vz_fail(ctx, VIMZ_ERR_HIP, "verify: download");
    if (!e[0].eq(Fe::one()) || !e[1].eq(v->u.x0) || !e[2].eq(v->u.x1)) res |= 1024;
    const NnPoint w = nn_point(v->uW);
    if (memcmp(&w, &v->u.W, sizeof(w))) res |= 1024;
  }
  // running CycleFold pair
  sec_spmv<Fq>(S, s, S.Zrun, S.az2, S.bz2, S.cz2);
  P_TRY(hipMemcpyAsync(S.bad, init, 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_check_relaxed<Fq>, dim3(stream_grid(S.n_c)), dim3(256), 0, s, (size_t)S.n_c, S.az2, S.bz2, S.cz2, v->cf_u_run, (const uint32_t*)S.E, S.bad);
  P_TRY(hipMemcpyAsync(bad, S.bad, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 32;
  P_TRY(hipMemcpyAsync(S.bad, init, 8, hipMemcpyHostToDevice, s));
  { const uint32_t* kept[3] = {S.AZ, S.BZ, S.CZ}; const uint32_t* fresh[3] = {S.az2, S.bz2, S.cz2};
    for (int m = 0; m < 3; m++) hipLaunchKernelGGL(k_count_diff, dim3(stream_grid(S.n_c)), dim3(256), 0, s, (size_t)S.n_c, kept[m], fresh[m], S.bad); }
  P_TRY(hipMemcpyAsync(bad, S.bad, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 2048;
  if ((rc = vz_msm_device(ctx, v->ck2, 0, S.Zrun + 8, S.n_w - 1 - CF_IO, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (!same_pt(pt, v->cfU.W)) res |= 64;
  if ((rc = vz_msm_device(ctx, v->ck2, 0, S.E, S.n_c, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (!same_pt(pt, v->cfU.E)) res |= 128;
  {
    Fq e[1 + CF_IO];
    if (!fetch(s, S.Zrun, 0, 1, &e[0]) || !fetch(s, S.Zrun, S.n_w - CF_IO, CF_IO, &e[1])) return vz_fail(ctx, VIMZ_ERR_HIP, "verify: download");
    if (!e[0].eq(v->cf_u_run) || !cross_field<Fe>(e[0]).eq(v->cfU.u)) res |= 1024;
    for (int k = 0; k < CF_IO; k++) if (memcmp(to_u256(e[1 + k]).w, v->cfU.x[k].w, 32)) res |= 1024;
  }
  *result = res;
  return VIMZ_OK;
}

// side 0 = main circuit (F + F'), side 1 = CycleFold circuit.  R1CS tables (VIMZ_CX_*), VIMZ_IX_INFO, and
//   VIMZ_IX_INSTANCE        side 0: comm_W.x, comm_W.y, comm_E.x, comm_E.y (canonical Fq), u, x0, x1 (canonical Fr)      [7 elements]
//                           side 1: comm_W.x, comm_W.y, comm_E.x, comm_E.y (canonical Fr), u, x[0..7) (canonical Fq)       [12 elements]
//   VIMZ_IX_FRESH_INSTANCE  side 0: comm_W.x, comm_W.y (Fq), x0, x1 (Fr)
//   VIMZ_IX_PARAMS          side 0: digest, z0..., z_i...
//   VIMZ_IX_RUNNING_Z / VIMZ_IX_RUNNING_E (both sides), VIMZ_IX_FRESH_Z (side 0: the last instance of F')
int64_t vimz_cf_export(vimz_cf* v, int side, int what, void* buf, size_t cap) {
  if (!v || (side != 0 && side != 1)) return VIMZ_ERR_INVALID;
  if (what < 100 || what == VIMZ_IX_INFO)
    return side == 0 ? export_builder(v->circ->build->b, v->c1->step_wires, v->c1->step_constraints, what, buf, cap) : export_builder(v->cf.b, 0, 0, what, buf, cap);
  vimz_ctx* ctx = v->ctx;
  vimz_prover* p = v->pri;
  SecDev& S = v->sec;
  if (what == VIMZ_IX_INSTANCE || what == VIMZ_IX_FRESH_INSTANCE || what == VIMZ_IX_PARAMS || what == VIMZ_IX_LAST_STEP) {
    std::vector<uint64_t> o;
    auto push = [&](const auto& m) { auto x = std::decay_t<decltype(m)>::from_mont(m); o.resize(o.size() + 4); memcpy(o.data() + o.size() - 4, x.v, 32); };
    auto push_u = [&](const U256w& x) { o.insert(o.end(), x.w, x.w + 4); };
    if (what == VIMZ_IX_LAST_STEP) {
      // the last step's F': i (one element), z_i, z_{i+1}, U (7), u (4), cmT (2), the hinted W', E' (4), cfU (12), cf1.W, cf1.T, cf2.W, cf2.T (8),
      // then what it returned: U' (7), cfU' (12), x0, x1, and the three challenges' low 128 bits (3)
      if (side != 0 || !v->have_last) return VIMZ_ERR_INVALID;
      o = last_step_words(v->last_in, v->last_out, v->last_zi, v->last_zn);
    } else if (what == VIMZ_IX_INSTANCE) {
      if (side == 0) { push_u(v->U.W.x); push_u(v->U.W.y); push_u(v->U.E.x); push_u(v->U.E.y); push(v->U.u); push(v->U.x0); push(v->U.x1); }
      else { push(v->cfU.W.x); push(v->cfU.W.y); push(v->cfU.E.x); push(v->cfU.E.y); push(v->cfU.u); for (auto& e : v->cfU.x) push_u(e); }
    } else if (what == VIMZ_IX_FRESH_INSTANCE) {
      if (side != 0) return VIMZ_ERR_INVALID;
      push_u(v->u.W.x); push_u(v->u.W.y); push(v->u.x0); push(v->u.x1);
    } else {
      if (side != 0) return VIMZ_ERR_INVALID;
      push(v->c1->digest); for (auto& z : v->z0) push(z); for (auto& z : p->z_cur) push(z);
    }
    const size_t bytes = o.size() * 8;
    if (buf && cap >= bytes) memcpy(buf, o.data(), bytes);
    return (int64_t)bytes;
  }
  const uint32_t* src = nullptr; size_t n = 0;
  switch (what) {
    case VIMZ_IX_RUNNING_Z: src = side == 0 ? p->Zrun : S.Zrun; n = side == 0 ? p->n_wires : S.n_w; break;
    case VIMZ_IX_RUNNING_E: src = side == 0 ? p->E : S.E; n = side == 0 ? p->n_c : S.n_c; break;
    case VIMZ_IX_FRESH_Z: if (side != 0) return VIMZ_ERR_INVALID; src = v->Zl; n = p->n_wires; break;
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

// ---- the proof as an object of its own (Sonobe's `ivc_proof()` / `from_ivc_proof`; checkpoint / resume): everything vimz_cf_verify reads and
// the next vimz_cf_fold needs.  Blob = header | host state | device vectors (Montgomery limbs as they sit in HBM).
namespace {
struct CfProofHeader { uint64_t magic, steps, n_w1, n_c1, n_w2, n_c2, len_z, flags; };
const uint64_t CF_PROOF_MAGIC = 0x3146435a56ull;   // "VZCF1"
struct CfProofHost { CfMainRelaxed U; G1Aff UW, UE; CfMainFresh u; G1Aff uW; CfRelaxed cfU; Fe u_run; Fq cf_u_run; Fe digest; };
size_t cf_vec_bytes(const vimz_cf* v) {
  const size_t nw1 = v->pri->n_wires, nc1 = v->pri->n_c, nw2 = v->sec.n_w, nc2 = v->sec.n_c;
  return 32 * (2 * nw1 + 7 * nc1 + nw2 + 4 * nc2);
}
}  // namespace

size_t vimz_cf_proof_size(const vimz_cf* v) {
  if (!v) return 0;
  return sizeof(CfProofHeader) + sizeof(CfProofHost) + 64 * (size_t)v->pri->len_z + cf_vec_bytes(v);
}

int vimz_cf_proof_export(vimz_cf* v, uint8_t* blob, size_t cap) {
  if (!v || !blob || cap < vimz_cf_proof_size(v)) return vz_fail(v ? v->ctx : nullptr, VIMZ_ERR_INVALID, "vimz_cf_proof_export: buffer too small");
  if (v->broken) return vz_fail(v->ctx, VIMZ_ERR_INVALID, "vimz_cf_proof_export: this IVC failed in the middle of a step");
  vimz_ctx* ctx = v->ctx; vimz_prover* p = v->pri; SecDev& S = v->sec;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  CfProofHeader h{CF_PROOF_MAGIC, v->i, p->n_wires, p->n_c, S.n_w, S.n_c, p->len_z, 0};
  CfProofHost hs; memset((void*)&hs, 0, sizeof(hs));
  hs.U = v->U; hs.UW = v->UW; hs.UE = v->UE; hs.u = v->u; hs.uW = v->uW; hs.cfU = v->cfU; hs.u_run = v->u_run; hs.cf_u_run = v->cf_u_run; hs.digest = v->c1->digest;
  uint8_t* o = blob;
  memcpy(o, &h, sizeof(h)); o += sizeof(h);
  memcpy(o, (const void*)&hs, sizeof(hs)); o += sizeof(hs);
  memcpy(o, v->z0.data(), 32 * p->len_z); o += 32 * p->len_z;
  memcpy(o, p->z_cur.data(), 32 * p->len_z); o += 32 * p->len_z;
  const uint32_t* src[] = {p->Zrun, p->E, p->AZ, p->BZ, p->CZ, v->Zl, v->azl, v->bzl, v->czl, S.Zrun, S.E, S.AZ, S.BZ, S.CZ};
  const size_t len[] = {p->n_wires, p->n_c, p->n_c, p->n_c, p->n_c, p->n_wires, p->n_c, p->n_c, p->n_c, S.n_w, S.n_c, S.n_c, S.n_c, S.n_c};
  for (int k = 0; k < 14; k++) { P_TRY(hipMemcpyAsync(o, src[k], 32 * len[k], hipMemcpyDeviceToHost, s)); o += 32 * len[k]; }
  P_TRY(hipStreamSynchronize(s));
  return VIMZ_OK;
}

int vimz_cf_proof_import(vimz_cf* v, const uint8_t* blob, size_t len) {
  if (!v || !blob || len < sizeof(CfProofHeader)) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = v->ctx; vimz_prover* p = v->pri; SecDev& S = v->sec;
  CfProofHeader h; memcpy(&h, blob, sizeof(h));
  if (h.magic != CF_PROOF_MAGIC || h.n_w1 != p->n_wires || h.n_c1 != p->n_c || h.n_w2 != S.n_w || h.n_c2 != S.n_c || h.len_z != p->len_z || len < vimz_cf_proof_size(v))
    return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_proof_import: the blob does not match this IVC's circuits");
  CfProofHost hs; memcpy((void*)&hs, blob + sizeof(h), sizeof(hs));
  if (!hs.digest.eq(v->c1->digest)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_proof_import: shape digest differs");
  // The blob is untrusted: every field element must be below its modulus and every point on its curve BEFORE anything of this IVC changes.
  {
    auto q_ok = [](const U256w& x) { Fq c; for (int i = 0; i < 4; i++) { c.v[2 * i] = (uint32_t)x.w[i]; c.v[2 * i + 1] = (uint32_t)(x.w[i] >> 32); } return c.is_reduced(); };
    auto g1_ok = [](const G1Aff& P) { return P.x.is_reduced() && P.y.is_reduced() && aff_on_curve(P); };
    auto g2_ok = [](const G2Aff& P) { return P.x.is_reduced() && P.y.is_reduced() && aff_on_curve(P); };
    auto same = [](const G1Aff& P, const NnPoint& n) { const NnPoint m = nn_point(P); return !memcmp(&m, &n, sizeof(m)); };
    bool ok = g1_ok(hs.UW) && g1_ok(hs.UE) && g1_ok(hs.uW) && same(hs.UW, hs.U.W) && same(hs.UE, hs.U.E) && same(hs.uW, hs.u.W) &&
              hs.U.u.is_reduced() && hs.U.x0.is_reduced() && hs.U.x1.is_reduced() && hs.u.x0.is_reduced() && hs.u.x1.is_reduced() &&
              g2_ok(hs.cfU.W) && g2_ok(hs.cfU.E) && hs.cfU.u.is_reduced() && hs.u_run.is_reduced() && hs.cf_u_run.is_reduced();
    for (int k = 0; k < CF_IO && ok; k++) ok = q_ok(hs.cfU.x[k]);
    const uint8_t* zp = blob + sizeof(h) + sizeof(hs);
    for (uint32_t k = 0; k < 2 * p->len_z && ok; k++) { Fe z; memcpy(z.v, zp + 32 * k, 32); ok = z.is_reduced(); }
    if (!ok) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_proof_import: a field element of the blob is not below its modulus, or a point is not on its curve");
  }
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const uint8_t* o = blob + sizeof(h) + sizeof(hs) + 64 * (size_t)p->len_z;
  const size_t ln[] = {p->n_wires, p->n_c, p->n_c, p->n_c, p->n_c, p->n_wires, p->n_c, p->n_c, p->n_c, S.n_w, S.n_c, S.n_c, S.n_c, S.n_c};
  const size_t vec_bytes = cf_vec_bytes(v);
  int rc = vz_ensure_scratch(ctx, vec_bytes + 64); if (rc) return rc;
  uint32_t* stage = (uint32_t*)ctx->scratch;
  uint32_t* badc = stage + vec_bytes / 4;
  P_TRY(hipMemcpyAsync(stage, o, vec_bytes, hipMemcpyHostToDevice, s));
  P_TRY(hipMemsetAsync(badc, 0, 8, s));
  {
    size_t off = 0;
    for (int k = 0; k < 14; k++) {
      if (k < 9) hipLaunchKernelGGL(k_count_unreduced<Fr>, dim3(stream_grid(ln[k])), dim3(256), 0, s, ln[k], (const uint32_t*)(stage + off), badc);
      else hipLaunchKernelGGL(k_count_unreduced<Fq>, dim3(stream_grid(ln[k])), dim3(256), 0, s, ln[k], (const uint32_t*)(stage + off), badc);
      off += 8 * ln[k];
    }
  }
  uint32_t nbad = 0;
  P_TRY(hipMemcpyAsync(&nbad, badc, 4, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (nbad) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_proof_import: a vector element of the blob is not below its modulus");
  uint32_t* dst[] = {p->Zrun, p->E, p->AZ, p->BZ, p->CZ, v->Zl, v->azl, v->bzl, v->czl, S.Zrun, S.E, S.AZ, S.BZ, S.CZ};
  { size_t off = 0; for (int k = 0; k < 14; k++) { P_TRY(hipMemcpyAsync(dst[k], stage + off, 32 * ln[k], hipMemcpyDeviceToDevice, s)); off += 8 * ln[k]; } }
  P_TRY(hipStreamSynchronize(s));
  const uint8_t* zp = blob + sizeof(h) + sizeof(hs);
  memcpy(v->z0.data(), zp, 32 * p->len_z);
  memcpy(p->z_cur.data(), zp + 32 * p->len_z, 32 * p->len_z);
  p->z0 = v->z0;
  v->i = h.steps; p->steps = h.steps;
  v->U = hs.U; v->UW = hs.UW; v->UE = hs.UE; v->u = hs.u; v->uW = hs.uW; v->cfU = hs.cfU; v->u_run = hs.u_run; v->cf_u_run = hs.cf_u_run;
  v->t_step_for = v->t_ver_for = -1; v->broken = false;
  v->c1->cache = CfHashCache();
  return VIMZ_OK;
}

/* IVC state chain only (as vimz_ivc_state_chain): where a row segment proven by another vimz_cf starts */
int vimz_cf_state_chain(vimz_cf* v, const uint64_t* z_start, const uint64_t* step_inputs, size_t nsteps, uint64_t* zs_out) {
  if (!v) return VIMZ_ERR_INVALID;
  return vimz_prover_state_chain(v->pri, z_start, step_inputs, nsteps, zs_out);
}
/* the same in its two parts (see vimz_prover_row_digests): the row digests of any run of rows on any prover's GPU, then the serial host chain */
size_t vimz_cf_digest_stride(const vimz_cf* v) { return v ? vimz_prover_digest_stride(v->pri) : 0; }
int vimz_cf_row_digests(vimz_cf* v, const uint64_t* step_inputs, size_t nsteps, uint64_t* digests_out) {
  if (!v) return VIMZ_ERR_INVALID;
  return vimz_prover_row_digests(v->pri, step_inputs, nsteps, digests_out);
}
int vimz_cf_chain_from_digests(vimz_cf* v, const uint64_t* z_start, const uint64_t* step_inputs, const uint64_t* digests, size_t nsteps, uint64_t* zs_out) {
  if (!v) return VIMZ_ERR_INVALID;
  return vimz_prover_chain_from_digests(v->pri, z_start, step_inputs, digests, nsteps, zs_out);
}

// KZG openings of the running main instance's two commitments (what Sonobe's decider adds to the proof for the on-chain verifier, decider.rs:13-21;
// calldata words kzg_*): with ck_main = the SRS's G1 powers, comm_W commits to the polynomial whose coefficients are the witness wires
// [1, wires - 2) of the running vector, comm_E to the error vector's.  which = 0: W, 1: E.  z, eval_out canonical; proof_xy canonical affine.
int vimz_cf_kzg_open(vimz_cf* v, int which, const uint64_t z[4], uint64_t eval_out[4], uint64_t proof_xy[8]) {
  if (!v || !z || !eval_out || !proof_xy || (which != 0 && which != 1)) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = v->ctx; vimz_prover* p = v->pri;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  return which == 0 ? vz_kzg_open_device(ctx, p->ck, 0, VIMZ_FIELD_BN254_FR, p->Zrun + 8, p->n_wires - 3, z, VIMZ_FORM_CANONICAL, eval_out, proof_xy)
                    : vz_kzg_open_device(ctx, p->ck, 0, VIMZ_FIELD_BN254_FR, p->E, p->n_c, z, VIMZ_FORM_CANONICAL, eval_out, proof_xy);
}

#ifdef VIMZ_TESTING      // ---- test hooks: only in libvimz_hip_testing.so (include/vimz_hip_testing.h) ----
// test hook: overwrite one element of a witness vector on the device (soundness tests flip wires and expect vimz_cf_verify / the
// oracle verifier to reject).  which: 0 running main Z, 1 last fresh main Z, 2 running CycleFold Z, 3 running main E, 4 running CycleFold E.
int vimz_cf_poke(vimz_cf* v, int which, size_t index, const uint64_t value[4]) {
  if (!v || !value) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = v->ctx;
  vimz_prover* p = v->pri;
  SecDev& S = v->sec;
  uint32_t* dst = nullptr; size_t n = 0; bool fq = false;
  switch (which) {
    case 0: dst = p->Zrun; n = p->n_wires; break;
    case 1: dst = v->Zl; n = p->n_wires; break;
    case 2: dst = S.Zrun; n = S.n_w; fq = true; break;
    case 3: dst = p->E; n = p->n_c; break;
    case 4: dst = S.E; n = S.n_c; fq = true; break;
    default: return VIMZ_ERR_INVALID;
  }
  if (index >= n) return VIMZ_ERR_INVALID;
  uint32_t m[8];
  if (fq) { Fq c; memcpy(c.v, value, 32); if (!c.is_reduced()) return VIMZ_ERR_INVALID; c = Fq::to_mont(c); memcpy(m, c.v, 32); }
  else { Fe c; memcpy(c.v, value, 32); if (!c.is_reduced()) return VIMZ_ERR_INVALID; c = Fe::to_mont(c); memcpy(m, c.v, 32); }
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  P_TRY(hipMemcpy(dst + 8 * index, m, 32, hipMemcpyHostToDevice));
  return VIMZ_OK;
}


// Host-only self-check of the two circuits (no GPU): `steps` steps of the recursion over the trivial step circuit z_out = z_in with
// made-up commitments (multiples of the generators: the circuits never open a commitment), every witness checked row by row against its
// R1CS, every in-circuit fold against the same fold in field / curve arithmetic.  result: 0 = all good; bit 0 a CycleFold witness
// violates its shape, bit 1 its result is wrong, bit 2 an F' witness violates its shape, bit 3 F' flagged its inputs, bit 4 a folded
// CycleFold public element differs from x + r·x_in mod q, bit 5 the folded scalars of the main instance differ, bit 6 the output
// hashes differ from the verifier's, bit 7 more flipped wires than the allowed handful went unnoticed.  counts (optional, 8 words): F' wires /
// constraints, CycleFold wires / constraints, then the flip test of the last step: wires of F + F' flipped, unnoticed, wires of the CycleFold
// circuit flipped, unnoticed.
static int cf_selfcheck_core(int steps, uint32_t* result, uint64_t counts[8], std::vector<uint64_t>* record) {
  if (!result || steps < 1 || steps > 64) return VIMZ_ERR_INVALID;
  try {
    CfCircuit cf; cf.finish();
    cb::BuilderT<Fe> b;
    b.len_z = 1; b.n_priv = 0; b.n_wires = 3;
    b.enforce(cb::LCT<Fe>::constant(Fe::one()), cb::LCT<Fe>::wire(2), cb::LCT<Fe>::wire(1));
    b.n_linear = 1;
    CfMainCircuit c1(b); c1.use_worker = getenv("VIMZ_CF_SELFCHECK_WORKERS") != nullptr; c1.finish(cf);      // (helper threads on: the sanitizer runs)
    if (counts) { counts[0] = b.n_wires; counts[1] = b.n_constraints(); counts[2] = cf.n_wires(); counts[3] = cf.n_constraints(); }
    auto sat = [](const auto& bld, const auto& z) -> bool {
      typedef std::decay_t<decltype(z[0])> F;
      const cb::Csr* Ms[3] = {&bld.A, &bld.B, &bld.C};
      for (uint32_t r = 0; r < bld.n_constraints(); r++) {
        F acc[3];
        for (int m = 0; m < 3; m++) { acc[m] = F::zero(); for (uint32_t k = Ms[m]->row_ptr[r]; k < Ms[m]->row_ptr[r + 1]; k++) acc[m] = F::add(acc[m], F::mul(bld.dict[Ms[m]->coef[k]], z[Ms[m]->col[k]])); }
        if (!F::mul(acc[0], acc[1]).eq(acc[2])) return false;
      }
      return true;
    };
    // number of wires from `from` on whose increment by one leaves every row that mentions them satisfied
    auto flips_unnoticed = [](const auto& bld, const auto& z0v, size_t from) -> uint64_t {
      typedef std::decay_t<decltype(z0v[0])> F;
      const cb::Csr* Ms[3] = {&bld.A, &bld.B, &bld.C};
      const uint32_t nr = bld.n_constraints();
      std::vector<std::vector<uint32_t>> rows_of(z0v.size());
      for (int m = 0; m < 3; m++) for (uint32_t r = 0; r < nr; r++) for (uint32_t k = Ms[m]->row_ptr[r]; k < Ms[m]->row_ptr[r + 1]; k++) {
        auto& v = rows_of[Ms[m]->col[k]]; if (v.empty() || v.back() != r) v.push_back(r);
      }
      auto z = z0v;
      uint64_t un = 0;
      for (size_t w = from; w < z.size(); w++) {
        const F keep = z[w]; z[w] = F::add(keep, F::one());
        bool noticed = false;
        for (uint32_t r : rows_of[w]) {
          F acc[3];
          for (int m = 0; m < 3; m++) { acc[m] = F::zero(); for (uint32_t k = Ms[m]->row_ptr[r]; k < Ms[m]->row_ptr[r + 1]; k++) acc[m] = F::add(acc[m], F::mul(bld.dict[Ms[m]->coef[k]], z[Ms[m]->col[k]])); }
          if (!F::mul(acc[0], acc[1]).eq(acc[2])) { noticed = true; break; }
        }
        if (!noticed) { un++; if (getenv("VIMZ_CF_SELFCHECK_VERBOSE")) fprintf(stderr, "[selfcheck] wire %zu of %zu: +1 unnoticed (%zu rows mention it)\n", w, z.size(), rows_of[w].size()); }
        z[w] = keep;
      }
      return un;
    };
    uint64_t flips_buf[4] = {0, 0, 0, 0}; uint64_t* flips_out = flips_buf;
    uint32_t res = 0;
    const G1Aff g1 = CycleSide<BnFq>::G(); const G2Aff g2 = CycleSide<BnFr>::G();
    auto fake1 = [&](uint64_t k) { const uint32_t w[2] = {(uint32_t)k, (uint32_t)(k >> 32)}; return to_affine(scalar_mul(g1, w, 64)); };
    auto fake2 = [&](uint64_t k) { const uint32_t w[2] = {(uint32_t)k, (uint32_t)(k >> 32)}; return to_affine(host_mul<Fe>(g2, w, 64)); };
    std::vector<Fe> z0 = {cb::f_from_u64<Fe>(7)};
    CfMainRelaxed U = CfMainRelaxed::zero(); G1Aff UW = g1_identity(), UE = g1_identity();
    CfMainFresh u = CfMainFresh::zero(); G1Aff uW = g1_identity();
    CfRelaxed cfU = CfRelaxed::zero();
    typedef Fp<BnFq> Q;
    auto q_of = [](const U256w& x) { return from_u256<Q>(x); };
    for (int i = 0; i < steps; i++) {
      CfMainIn in = CfMainIn::zero();
      in.digest = c1.digest; in.i = (uint64_t)i; in.z0 = z0; in.U = U; in.u = u; in.cfU = cfU;
      CfChallenges ch; ch.h_U = cf_hash_main(c1.digest, i, z0, z0.data(), U); ch.h_cf = cf_hash_cf(c1.digest, cfU);
      G1Aff Wn = g1_identity(), En = g1_identity();
      U256w want_x[CF_IO];
      if (i > 0) {
        const G1Aff cT = i > 1 ? fake1(0x1000 + i) : g1_identity();
        in.T = nn_point(cT);
        cf_challenge_main(ch, u, in.T);
        Wn = g1_fold(UW, ch.r, uW); En = g1_fold(UE, ch.r, cT);
        in.Wn = nn_point(Wn); in.En = nn_point(En);
        const G1Aff P1s[2] = {UW, UE}, P2s[2] = {uW, cT}, P3s[2] = {Wn, En};
        for (int c = 0; c < 2; c++) {
          std::vector<Q> wires; bool bad = false;
          const G1Aff P3 = cf.witness(ch.r, P1s[c], P2s[c], wires, &bad);
          if (bad || !sat(cf.b, wires)) res |= 1;
          else if (i == steps - 1 && c == 0) { const uint64_t un = flips_unnoticed(cf.b, wires, 1); if (flips_out) { flips_out[2] = wires.size() - 1; flips_out[3] = un; } if (un > 4) res |= 128; }
          if (!P3.x.eq(P3s[c].x) || !P3.y.eq(P3s[c].y)) res |= 2;
        }
        in.cf1W = fake2(0x2000 + i); in.cf1T = i > 1 ? fake2(0x3000 + i) : g2_identity();
        in.cf2W = fake2(0x4000 + i); in.cf2T = fake2(0x5000 + i);
        cf_challenge_cf1(ch, in.cf1W, in.Wn, in.cf1T);
        cf_challenge_cf2(ch, in.cf2W, in.En, in.cf2T);
        // the folded public elements in field arithmetic
        const U256w x1[CF_IO] = {cf_challenge_u256(ch.r), in.U.W.x, in.U.W.y, in.u.W.x, in.u.W.y, in.Wn.x, in.Wn.y};
        const U256w x2[CF_IO] = {cf_challenge_u256(ch.r), in.U.E.x, in.U.E.y, in.T.x, in.T.y, in.En.x, in.En.y};
        const Q r1 = rho_element<Q>(ch.r1), r2 = rho_element<Q>(ch.r2);
        for (int k = 0; k < CF_IO; k++) want_x[k] = to_u256(Q::add(Q::add(q_of(cfU.x[k]), Q::mul(r1, q_of(x1[k]))), Q::mul(r2, q_of(x2[k]))));
      }
      std::vector<Fe> aug; bool bad = false;
      CfMainOut o = c1.witness(in, z0.data(), z0.data(), aug, &bad);
      if (bad) res |= 8;
      std::vector<Fe> z = {Fe::one(), z0[0], z0[0]};
      z.insert(z.end(), aug.begin(), aug.end());
      if (!sat(b, z)) res |= 4;
      else if (i == steps - 1) {    // EVERY wire of F' matters: adding one to it must violate a row that mentions it (a wire no row notices is
                                    // unconstrained — the only ones allowed are the inverse hints of is-zero tests whose argument IS zero)
        uint64_t unnoticed = flips_unnoticed(b, z, 3);
        if (flips_out) { flips_out[0] = z.size() - 3; flips_out[1] = unnoticed; }
        if (unnoticed > 8) res |= 128;
      }
      if (i > 0) {
        if (memcmp(o.r, ch.r, 16) || memcmp(o.r1, ch.r1, 16) || memcmp(o.r2, ch.r2, 16)) res |= 64;
        for (int k = 0; k < CF_IO; k++) if (memcmp(o.cfU_new.x[k].w, want_x[k].w, 32)) res |= 16;
        const Fe rho = rho_element<Fe>(ch.r);
        if (!o.U_new.u.eq(Fe::add(U.u, rho)) || !o.U_new.x0.eq(Fe::add(U.x0, Fe::mul(rho, u.x0))) || !o.U_new.x1.eq(Fe::add(U.x1, Fe::mul(rho, u.x1)))) res |= 32;
        // the folded CycleFold commitments in curve arithmetic
        const uint32_t k1[5] = {ch.r1[0], ch.r1[1], ch.r1[2], ch.r1[3], 1u}, k2[5] = {ch.r2[0], ch.r2[1], ch.r2[2], ch.r2[3], 1u};
        auto fold2 = [&](const G2Aff& P, const G2Aff& A, const G2Aff& B) {
          G2 acc = aff_is_identity(P) ? G2::identity() : from_affine(P);
          if (!aff_is_identity(A)) { G2 t = host_mul<Fe>(A, k1, 129); add_full(acc, t); }
          if (!aff_is_identity(B)) { G2 t = host_mul<Fe>(B, k2, 129); add_full(acc, t); }
          return to_affine(acc);
        };
        const G2Aff Wc = fold2(cfU.W, in.cf1W, in.cf2W), Ec = fold2(cfU.E, in.cf1T, in.cf2T);
        if (!Wc.x.eq(o.cfU_new.W.x) || !Wc.y.eq(o.cfU_new.W.y) || !Ec.x.eq(o.cfU_new.E.x) || !Ec.y.eq(o.cfU_new.E.y)) res |= 16;
      }
      if (record && i == steps - 1) {      // digest, z_0, then the VIMZ_IX_LAST_STEP words of this step
        record->clear();
        auto put = [&](const Fe& m) { const Fe x = Fe::from_mont(m); record->resize(record->size() + 4); memcpy(record->data() + record->size() - 4, x.v, 32); };
        put(c1.digest); put(z0[0]);
        const std::vector<uint64_t> w = last_step_words(in, o, z0, z0);
        record->insert(record->end(), w.begin(), w.end());
      }
      U = o.U_new; UW = i > 0 ? Wn : g1_identity(); UE = i > 0 ? En : g1_identity(); cfU = o.cfU_new;
      if (!cf_hash_main(c1.digest, i + 1, z0, z0.data(), U).eq(o.x0) || !cf_hash_cf(c1.digest, cfU).eq(o.x1)) res |= 64;
      uW = fake1(0x6000 + i); u.W = nn_point(uW); u.x0 = o.x0; u.x1 = o.x1;
    }
    if (counts) { counts[4] = flips_buf[0]; counts[5] = flips_buf[1]; counts[6] = flips_buf[2]; counts[7] = flips_buf[3]; }
    *result = res;
    return VIMZ_OK;
  } catch (const std::exception& e) { return vz_fail(nullptr, VIMZ_ERR_INVALID, e.what()); }
}
int vimz_cf_selfcheck(int steps, uint32_t* result, uint64_t counts[8]) { return cf_selfcheck_core(steps, result, counts, nullptr); }
// test hook, host only: `jobs` trivial jobs through one helper thread (aug::Worker), each posted and waited for; returns how many ran.  With
// VIMZ_WORKER_SPIN_US=0 the helper sleeps between jobs: every post is a wake-up (a lost one would hang this call).
int64_t vimz_worker_selftest(int jobs) {
  if (jobs < 0) return VIMZ_ERR_INVALID;
  aug::Worker w;
  std::atomic<int64_t> ran{0};
  for (int i = 0; i < jobs; i++) { w.start([&] { ran.fetch_add(1, std::memory_order_relaxed); }); w.wait(); }
  return ran.load();
}
// the same run's LAST step for an outside restatement of the relation (tests/_cyclefold.py::step_relation, on the CPU): digest, z_0 (one element:
// the trivial step circuit's state), then the words of VIMZ_IX_LAST_STEP.  Returns the byte size (copies when buf is large enough).
int64_t vimz_cf_selfcheck_last_step(int steps, void* buf, size_t cap) {
  uint32_t res = 0; std::vector<uint64_t> rec;
  const int rc = cf_selfcheck_core(steps, &res, nullptr, &rec);
  if (rc) return rc;
  if (res) return VIMZ_ERR_UNSAT;
  const size_t bytes = 8 * rec.size();
  if (buf && cap >= bytes) memcpy(buf, rec.data(), bytes);
  return (int64_t)bytes;
}

}  // extern "C"
// Host only: the canonical decomposition (aug/cs.hpp: bits_strict) against the aliased witness; layout of `out`: include/vimz_hip_testing.h.
template <class FP>
static void strict_bits_gadget_check(int values, uint64_t out[8]) {
  typedef Fp<FP> F;
  typedef aug::CS<FP> CSx;
  // shape: wire 0 = 1, wire 1 = x, then the gadget's wires; the first FP::BITS rows are the plain Num2Bits rows, the rest the comparison
  cb::BuilderT<F> b;
  b.len_z = 0; b.n_priv = 0; b.n_wires = 2;
  size_t plain_rows = 0;
  {
    CSx cs; cs.b = &b; cs.base = 2;
    typename CSx::N x = cs.wire(1, F::zero());
    cs.bits(x, FP::BITS);                       // only to count the plain gadget's rows
    plain_rows = b.n_constraints();
  }
  cb::BuilderT<F> bs;
  bs.len_z = 0; bs.n_priv = 0; bs.n_wires = 2;
  { CSx cs; cs.b = &bs; cs.base = 2; typename CSx::N x = cs.wire(1, F::zero()); cs.bits_strict(x); }
  const cb::Csr* Ms[3] = {&bs.A, &bs.B, &bs.C};
  auto violated = [&](const std::vector<F>& z, bool* plain_bad, bool* strict_bad) {
    *plain_bad = *strict_bad = false;
    for (uint32_t r = 0; r < bs.n_constraints(); r++) {
      F acc[3];
      for (int m = 0; m < 3; m++) { acc[m] = F::zero(); for (uint32_t k = Ms[m]->row_ptr[r]; k < Ms[m]->row_ptr[r + 1]; k++) acc[m] = F::add(acc[m], F::mul(bs.dict[Ms[m]->coef[k]], z[Ms[m]->col[k]])); }
      if (!F::mul(acc[0], acc[1]).eq(acc[2])) { if (r < plain_rows) *plain_bad = true; else *strict_bad = true; }
    }
  };
  uint64_t ctr = 0x243f6a8885a308d3ull;
  auto next = [&] { ctr = ctr * 6364136223846793005ull + 1442695040888963407ull; return ctr >> 7; };
  for (int t = 0; t < values; t++) {
    F x;
    if (t == 0) x = F::zero();                                             // h = 0: aliased by p itself
    else if (t == 1) x = F::neg(F::one());                                 // h = p - 1: the largest canonical value, no alias
    else if (t == 2) { F c = F::zero(); uint64_t br = 0; uint32_t top[8] = {0, 0, 0, 0, 0, 0, 0, 1u << ((FP::BITS - 1) & 31)};      // h = 2^n - p - 1: the largest aliasable value
      top[7] = 0; uint32_t pw[8] = {0}; pw[(FP::BITS) >> 5] = 1u << (FP::BITS & 31);
      for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)pw[i] - FP::MOD.w[i] - br; c.v[i] = (uint32_t)d; br = (d >> 32) & 1; }
      { uint64_t b1 = 1; for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)c.v[i] - b1; c.v[i] = (uint32_t)d; b1 = (d >> 32) & 1; } }
      x = F::to_mont(c); }
    else { x = F::mul(F::mul(cb::f_from_u64<F>(next()), cb::f_from_u64<F>(next())), F::mul(cb::f_from_u64<F>(next()), cb::f_from_u64<F>(next()))); x = F::mul(x, F::mul(cb::f_from_u64<F>(next()), cb::f_from_u64<F>(next()))); }      // (six 57-bit factors: reduced, about uniform)
    out[0]++;
    for (int attack = 0; attack < 2; attack++) {
      CSx cs; cs.base = 2; cs.alias_attack = attack != 0;
      typename CSx::N xn = cs.wire(1, x);
      cs.bits_strict(xn);
      if (attack && !cs.alias_used) continue;
      std::vector<F> z = {F::one(), x};
      z.insert(z.end(), cs.w.begin(), cs.w.end());
      bool pb, sb; violated(z, &pb, &sb);
      if (!attack) { if (pb || sb || cs.bad) out[4]++; }
      else { out[1]++; if (!pb) out[2]++; if (!pb && sb) out[3]++; }
    }
  }
}
extern "C" {
int vimz_strict_bits_selfcheck(int field, int values, uint64_t out[8]) {
  if (!out || values < 3 || values > 100000 || (field != 0 && field != 1)) return VIMZ_ERR_INVALID;
  try {
    for (int k = 0; k < 8; k++) out[k] = 0;
    if (field == 0) strict_bits_gadget_check<BnFr>(values, out); else strict_bits_gadget_check<BnFq>(values, out);
    if (field != 0) return VIMZ_OK;
    // inside F': the recursion over the trivial step circuit with every challenge decomposition aliased where an alias exists
    CfCircuit cf; cf.finish();
    cb::BuilderT<Fe> b;
    b.len_z = 1; b.n_priv = 0; b.n_wires = 3;
    b.enforce(cb::LCT<Fe>::constant(Fe::one()), cb::LCT<Fe>::wire(2), cb::LCT<Fe>::wire(1));
    b.n_linear = 1;
    CfMainCircuit c1(b); c1.use_worker = false; c1.finish(cf);
    auto sat = [&](const std::vector<Fe>& z) {
      const cb::Csr* Ms[3] = {&b.A, &b.B, &b.C};
      for (uint32_t r = 0; r < b.n_constraints(); r++) {
        Fe acc[3];
        for (int m = 0; m < 3; m++) { acc[m] = Fe::zero(); for (uint32_t k = Ms[m]->row_ptr[r]; k < Ms[m]->row_ptr[r + 1]; k++) acc[m] = Fe::add(acc[m], Fe::mul(b.dict[Ms[m]->coef[k]], z[Ms[m]->col[k]])); }
        if (!Fe::mul(acc[0], acc[1]).eq(acc[2])) return false;
      }
      return true;
    };
    const G1Aff g1 = CycleSide<BnFq>::G(); const G2Aff g2 = CycleSide<BnFr>::G();
    auto fake1 = [&](uint64_t k) { const uint32_t w[2] = {(uint32_t)k, (uint32_t)(k >> 32)}; return to_affine(scalar_mul(g1, w, 64)); };
    auto fake2 = [&](uint64_t k) { const uint32_t w[2] = {(uint32_t)k, (uint32_t)(k >> 32)}; return to_affine(host_mul<Fe>(g2, w, 64)); };
    std::vector<Fe> z0 = {cb::f_from_u64<Fe>(11)};
    CfMainRelaxed U = CfMainRelaxed::zero(); G1Aff UW = g1_identity(), UE = g1_identity();
    CfMainFresh u = CfMainFresh::zero(); G1Aff uW = g1_identity();
    CfRelaxed cfU = CfRelaxed::zero();
    const int steps = std::min(12, std::max(4, values / 8));
    for (int i = 0; i < steps; i++) {
      CfMainIn in = CfMainIn::zero();
      in.digest = c1.digest; in.i = (uint64_t)i; in.z0 = z0; in.U = U; in.u = u; in.cfU = cfU;
      CfChallenges ch; ch.h_U = cf_hash_main(c1.digest, i, z0, z0.data(), U); ch.h_cf = cf_hash_cf(c1.digest, cfU);
      G1Aff Wn = g1_identity(), En = g1_identity();
      if (i > 0) {
        const G1Aff cT = i > 1 ? fake1(0x1100 + i) : g1_identity();
        in.T = nn_point(cT);
        cf_challenge_main(ch, u, in.T);
        Wn = g1_fold(UW, ch.r, uW); En = g1_fold(UE, ch.r, cT);
        in.Wn = nn_point(Wn); in.En = nn_point(En);
        in.cf1W = fake2(0x2100 + i); in.cf1T = i > 1 ? fake2(0x3100 + i) : g2_identity();
        in.cf2W = fake2(0x4100 + i); in.cf2T = fake2(0x5100 + i);
      }
      // the honest run carries the recursion forward; the aliased run of the same step must not be satisfiable
      std::vector<Fe> aug; bool bad = false;
      CfMainOut o = c1.witness(in, z0.data(), z0.data(), aug, &bad);
      std::vector<Fe> z = {Fe::one(), z0[0], z0[0]}; z.insert(z.end(), aug.begin(), aug.end());
      if (bad || !sat(z)) out[4]++;
      if (i > 0) {
        std::vector<Fe> aug2; bool bad2 = false; int used = 0;
        c1.witness(in, z0.data(), z0.data(), aug2, &bad2, &used);
        out[5]++;
        if (used) {
          out[7]++;
          std::vector<Fe> z2 = {Fe::one(), z0[0], z0[0]}; z2.insert(z2.end(), aug2.begin(), aug2.end());
          if (sat(z2)) out[6]++;
        }
      }
      U = o.U_new; UW = i > 0 ? Wn : g1_identity(); UE = i > 0 ? En : g1_identity(); cfU = o.cfU_new;
      uW = fake1(0x6100 + i); u.W = nn_point(uW); u.x0 = o.x0; u.x1 = o.x1;
    }
    return VIMZ_OK;
  } catch (const std::exception& e) { return vz_fail(nullptr, VIMZ_ERR_INVALID, e.what()); }
}
#endif      // VIMZ_TESTING

}  // extern "C"

