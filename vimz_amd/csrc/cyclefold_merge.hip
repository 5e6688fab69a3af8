// ONE proof object out of several row segments' Nova + CycleFold proofs (vimz_cf_merge*): see the protocol comment below.  The prover itself:
// cyclefold.hip.
#include "cyclefold_internal.hpp"
#include "decider_view.hpp"
#ifdef VIMZ_TESTING
#include "../../include/vimz_hip_testing.h"
#endif

// ---- ONE proof object out of several row segments' CycleFold proofs: the "host-side sequential final fold" of BASELINE.json's north_star for
// this scheme (what merge.hip is for the Nova IVC).  Row segments of an image folded concurrently — each a vimz_cf of its own, on one GPU or
// one per GPU — are folded into one verifiable object by out-of-circuit NIFS, in row order:
//     acc = U_1;  acc = acc (+) u_1;          then for j = 2..S:   acc = acc (+) U_j   (two relaxed instances: E += r·T + r²·E_j),
//     cfacc = cfU_1;                                                acc = acc (+) u_j   (the segment's last instance of F', strict),
//                                                                   cfacc = cfacc (+) cfU_j          (on Grumpkin, over Fq).
// Record j = (n_j, z_start_j, z_end_j, U_j, u_j, cfU_j, T1_j, T2_j, Tc_j): the segment's statement and instances and the commitments to
// the three cross terms (T1_1 = Tc_1 = identity).  Transcript: h_0 = SHA3("vimz-cf-merge-v1" ‖ digest ‖ len_z); h = SHA3(h_{j-1} ‖ record j
// without the T's); r1 = chal(h ‖ 'a' ‖ T1), r2 = chal(h ‖ 'b' ‖ T1 ‖ T2), rc = chal(h ‖ 'c' ‖ Tc) — first 16 bytes of a SHA3-256 as a
// little-endian integer —; h_j = SHA3(h ‖ 'n' ‖ T1 ‖ T2 ‖ Tc).  The verifier replays the records (for every segment: the two hashes its
// last instance carries, recomputed from (n_j, z_start_j, z_end_j); z_end_{j-1} = z_start_j; the folds of the instances), obtains
// (n = Σ n_j, z_start_1, z_end_S, acc, cfacc) and checks ONE main and ONE CycleFold relaxed instance against their witnesses.
// Several such objects — one RUN of segments each, e.g. one per GPU — are folded left to right (vimz_cf_merge_merged): Node(A, B) requires
// A.z_end = B.z_start;  h = SHA3("vimz-cf-merge-node-v1" ‖ A.h ‖ B.h ‖ T_p ‖ T_q), r_p = chal(h ‖ 'p'), r_q = chal(h ‖ 'q');  both pairs of
// relaxed accumulators fold with E += r·T + r²·E_B.  The verifier recomputes every run's accumulator from its records on its own.
// Ours, like the circuits.
struct CfSegRec { uint64_t n = 0; std::vector<Fe> zs, ze; CfMainRelaxed U; G1Aff UW, UE; CfMainFresh u; G1Aff uW; CfRelaxed cfU; G1Aff T1, T2; G2Aff Tc; };
struct CfAcc { uint8_t h[32] = {}; uint64_t n = 0; std::vector<Fe> zs, ze; G1Aff cW, cE; Fe u, x0, x1; G2Aff qW, qE; Fq qu; Fq qx[CF_IO]; };
struct CfJunction { G1Aff Tp; G2Aff Tq; };
struct vimz_cf_merged {
  vimz_cf* vk = nullptr;                  // shapes, keys, context: must outlive this object
  std::vector<CfSegRec> segs;
  // RUNS: segs[run_start[k] .. run_start[k+1]) were folded in row order by vimz_cf_merge on one GPU (one run per GPU of a sharded proof);
  // run k > 0 was folded into the runs before it as a whole (vimz_cf_merge_merged: two relaxed accumulators), junction k-1 holds that fold's
  // two cross-term commitments
  std::vector<uint32_t> run_start{0};
  std::vector<CfJunction> junctions;
  CfAcc acc;
  uint32_t* dev = nullptr;                // one allocation
  uint32_t *Zp = nullptr, *Ep = nullptr, *AZp = nullptr, *BZp = nullptr, *CZp = nullptr, *Tp = nullptr;
  uint32_t *Zq = nullptr, *Eq = nullptr, *AZq = nullptr, *BZq = nullptr, *CZq = nullptr, *Tq = nullptr;
  bool broken = false;
  double seconds[4] = {};                 // GPU cross terms + commitments, folds, host, total
};

namespace {
const uint64_t CF_MERGED_MAGIC = 0x32474d46435a56ull;      // "VZCFMG2"
template <class F> void cfm_fe(Sha3& h, const F& m) { const F c = F::from_mont(m); h.update(c.v, 32); }
template <class F> void cfm_pt(Sha3& h, const Affine<F>& p) { cfm_fe(h, p.x); cfm_fe(h, p.y); }
void cfm_chal(const uint8_t h[32], char tag, const uint8_t* extra, size_t n, uint32_t out[4]) {
  Sha3 s; s.update(h, 32); s.update(&tag, 1); if (n) s.update(extra, n);
  uint8_t d[32]; s.finish(d); memcpy(out, d, 16);
}
template <class F> F cfm_fe128(const uint32_t r[4]) { F c = F::zero(); for (int k = 0; k < 4; k++) c.v[k] = r[k]; return F::to_mont(c); }
template <class FS> Affine<FS> cfm_axpy(const Affine<FS>& a, const uint32_t r[4], const Affine<FS>& b) {      // a + r·b, r of 128 bits
  XYZZ<FS> acc = aff_is_identity(a) ? XYZZ<FS>::identity() : from_affine(a);
  if (!aff_is_identity(b)) { XYZZ<FS> t = host_mul<FS>(b, r, 128); add_full(acc, t); }
  return to_affine(acc);
}
void cfm_points_bytes(const G1Aff* pts, int n, std::vector<uint8_t>& out) {
  out.clear();
  for (int k = 0; k < n; k++) for (const Fq* c : {&pts[k].x, &pts[k].y}) { const Fq x = Fq::from_mont(*c); const uint8_t* b = (const uint8_t*)x.v; out.insert(out.end(), b, b + 32); }
}
// h = SHA3(h_prev ‖ record without its cross-term commitments)
void cfm_segment_hash(const uint8_t prev[32], const CfSegRec& s, uint8_t out[32]) {
  Sha3 h; h.update(prev, 32);
  h.update(&s.n, 8);
  for (auto& z : s.zs) cfm_fe(h, z);
  for (auto& z : s.ze) cfm_fe(h, z);
  cfm_pt(h, s.UW); cfm_pt(h, s.UE); cfm_fe(h, s.U.u); cfm_fe(h, s.U.x0); cfm_fe(h, s.U.x1);
  cfm_pt(h, s.uW); cfm_fe(h, s.u.x0); cfm_fe(h, s.u.x1);
  cfm_pt(h, s.cfU.W); cfm_pt(h, s.cfU.E); cfm_fe(h, s.cfU.u);
  for (auto& e : s.cfU.x) h.update(e.w, 32);
  h.finish(out);
}
struct CfmChallenges { uint32_t r1[4], r2[4], rc[4]; };
void cfm_challenges(const uint8_t h[32], const CfSegRec& s, CfmChallenges& c, uint8_t h_next[32]) {
  std::vector<uint8_t> b;
  const G1Aff t12[2] = {s.T1, s.T2};
  cfm_points_bytes(t12, 1, b); cfm_chal(h, 'a', b.data(), b.size(), c.r1);
  cfm_points_bytes(t12, 2, b); cfm_chal(h, 'b', b.data(), b.size(), c.r2);
  std::vector<uint8_t> q;
  for (const Fe* x : {&s.Tc.x, &s.Tc.y}) { const Fe v = Fe::from_mont(*x); const uint8_t* p8 = (const uint8_t*)v.v; q.insert(q.end(), p8, p8 + 32); }
  cfm_chal(h, 'c', q.data(), q.size(), c.rc);
  Sha3 n; n.update(h, 32); const char tag = 'n'; n.update(&tag, 1); n.update(b.data(), b.size()); n.update(q.data(), q.size()); n.finish(h_next);
}
Fq cfm_q(const U256w& x) { return from_u256<Fq>(x); }
// The statement-side of absorbing one segment: checks (flags: bit 0 hash of its main chain, bit 1 of its CycleFold chain, bit 2 not adjacent /
// empty) and the folds of the instances.  first: acc is uninitialised.
void cfm_absorb(const vimz_cf* vk, CfAcc& acc, const CfSegRec& s, bool first, uint32_t* flags, CfmChallenges* ch_out) {
  const Fe dg = vk->c1->digest;
  if (s.n == 0) *flags |= 4;
  if (!cf_hash_main(dg, s.n, s.zs, s.ze.data(), s.U).eq(s.u.x0)) *flags |= 1;
  if (!cf_hash_cf(dg, s.cfU).eq(s.u.x1)) *flags |= 2;
  { const NnPoint a = nn_point(s.UW), b = nn_point(s.UE), c = nn_point(s.uW);
    if (memcmp(&a, &s.U.W, sizeof(a)) || memcmp(&b, &s.U.E, sizeof(b)) || memcmp(&c, &s.u.W, sizeof(c))) *flags |= 1; }
  uint8_t prev[32];
  if (first) {
    Sha3 h0; const char* tag = "vimz-cf-merge-v1"; h0.update(tag, strlen(tag)); cfm_fe(h0, dg); const uint64_t lz = vk->c1->len_z; h0.update(&lz, 8); h0.finish(prev);
  } else {
    memcpy(prev, acc.h, 32);
    if (acc.ze.size() != s.zs.size()) *flags |= 4;
    else for (size_t k = 0; k < s.zs.size(); k++) if (!acc.ze[k].eq(s.zs[k])) *flags |= 4;
  }
  uint8_t h[32]; cfm_segment_hash(prev, s, h);
  CfmChallenges ch; cfm_challenges(h, s, ch, acc.h);
  if (ch_out) *ch_out = ch;
  if (first) {
    acc.n = 0; acc.zs = s.zs;
    acc.cW = s.UW; acc.cE = s.UE; acc.u = s.U.u; acc.x0 = s.U.x0; acc.x1 = s.U.x1;
    acc.qW = s.cfU.W; acc.qE = s.cfU.E; acc.qu = cross_field<Fq>(s.cfU.u);
    for (int k = 0; k < CF_IO; k++) acc.qx[k] = cfm_q(s.cfU.x[k]);
  } else {
    const Fe r1 = cfm_fe128<Fe>(ch.r1);
    acc.cW = cfm_axpy(acc.cW, ch.r1, s.UW);
    acc.cE = cfm_axpy(acc.cE, ch.r1, cfm_axpy(s.T1, ch.r1, s.UE));          // E + r·(T + r·E_j)
    acc.u = Fe::add(acc.u, Fe::mul(r1, s.U.u)); acc.x0 = Fe::add(acc.x0, Fe::mul(r1, s.U.x0)); acc.x1 = Fe::add(acc.x1, Fe::mul(r1, s.U.x1));
    const Fq rc = cfm_fe128<Fq>(ch.rc);
    acc.qW = cfm_axpy(acc.qW, ch.rc, s.cfU.W);
    acc.qE = cfm_axpy(acc.qE, ch.rc, cfm_axpy(s.Tc, ch.rc, s.cfU.E));
    acc.qu = Fq::add(acc.qu, Fq::mul(rc, cross_field<Fq>(s.cfU.u)));
    for (int k = 0; k < CF_IO; k++) acc.qx[k] = Fq::add(acc.qx[k], Fq::mul(rc, cfm_q(s.cfU.x[k])));
  }
  const Fe r2 = cfm_fe128<Fe>(ch.r2);
  acc.cW = cfm_axpy(acc.cW, ch.r2, s.uW);
  acc.cE = cfm_axpy(acc.cE, ch.r2, s.T2);
  acc.u = Fe::add(acc.u, r2); acc.x0 = Fe::add(acc.x0, Fe::mul(r2, s.u.x0)); acc.x1 = Fe::add(acc.x1, Fe::mul(r2, s.u.x1));
  acc.n += s.n; acc.ze = s.ze;
}
// Node(A, B): the statement-side fold of two runs' accumulators (flags bit 2: not adjacent); A becomes the result
void cfm_node(CfAcc& A, const CfAcc& B, const CfJunction& J, uint32_t* flags, uint32_t rp_out[4], uint32_t rq_out[4]) {
  if (A.ze.size() != B.zs.size()) *flags |= 4;
  else for (size_t k = 0; k < B.zs.size(); k++) if (!A.ze[k].eq(B.zs[k])) *flags |= 4;
  Sha3 hh; const char* tag = "vimz-cf-merge-node-v1"; hh.update(tag, strlen(tag)); hh.update(A.h, 32); hh.update(B.h, 32);
  cfm_pt(hh, J.Tp); cfm_pt(hh, J.Tq);
  uint8_t h[32]; hh.finish(h);
  uint32_t rp[4], rq[4];
  cfm_chal(h, 'p', nullptr, 0, rp); cfm_chal(h, 'q', nullptr, 0, rq);
  if (rp_out) memcpy(rp_out, rp, 16);
  if (rq_out) memcpy(rq_out, rq, 16);
  const Fe rpf = cfm_fe128<Fe>(rp); const Fq rqf = cfm_fe128<Fq>(rq);
  A.cW = cfm_axpy(A.cW, rp, B.cW);
  A.cE = cfm_axpy(A.cE, rp, cfm_axpy(J.Tp, rp, B.cE));
  A.u = Fe::add(A.u, Fe::mul(rpf, B.u)); A.x0 = Fe::add(A.x0, Fe::mul(rpf, B.x0)); A.x1 = Fe::add(A.x1, Fe::mul(rpf, B.x1));
  A.qW = cfm_axpy(A.qW, rq, B.qW);
  A.qE = cfm_axpy(A.qE, rq, cfm_axpy(J.Tq, rq, B.qE));
  A.qu = Fq::add(A.qu, Fq::mul(rqf, B.qu));
  for (int k = 0; k < CF_IO; k++) A.qx[k] = Fq::add(A.qx[k], Fq::mul(rqf, B.qx[k]));
  A.n += B.n; A.ze = B.ze;
  memcpy(A.h, h, 32);
}
// the whole statement side: every run from its records, the runs folded left to right
bool cfm_replay(const vimz_cf* vk, const std::vector<CfSegRec>& segs, const std::vector<uint32_t>& run_start, const std::vector<CfJunction>& junctions, CfAcc& out, uint32_t* flags) {
  if (segs.empty() || run_start.empty() || run_start[0] != 0 || junctions.size() + 1 != run_start.size()) return false;
  for (size_t k = 0; k < run_start.size(); k++) {
    const size_t lo = run_start[k], hi = k + 1 < run_start.size() ? run_start[k + 1] : segs.size();
    if (hi <= lo || hi > segs.size()) return false;
    CfAcc a;
    for (size_t j = lo; j < hi; j++) cfm_absorb(vk, a, segs[j], j == lo, flags, nullptr);
    if (k == 0) out = a; else cfm_node(out, a, junctions[k - 1], flags, nullptr, nullptr);
  }
  return true;
}
CfSegRec cfm_record_of(const vimz_cf* v) {
  CfSegRec s; s.n = v->i; s.zs = v->z0; s.ze = v->pri->z_cur;
  s.U = v->U; s.UW = v->UW; s.UE = v->UE; s.u = v->u; s.uW = v->uW; s.cfU = v->cfU;
  s.T1 = s.T2 = g1_identity(); s.Tc = g2_identity();
  return s;
}
template <class F>
void cfm_fold(hipStream_t s, uint32_t* Z, uint32_t* E, uint32_t* AZ, uint32_t* BZ, uint32_t* CZ, size_t nw, size_t nc,
              const uint32_t* z2, const uint32_t* T, const uint32_t* az2, const uint32_t* bz2, const uint32_t* cz2, const uint32_t* E2, const F& r) {
  Fold5 f;
  f.x1[0] = Z; f.x2[0] = z2; f.n[0] = nw;
  f.x1[1] = T ? E : nullptr; f.x2[1] = T; f.n[1] = nc;
  f.x1[2] = AZ; f.x2[2] = az2; f.n[2] = nc;
  f.x1[3] = BZ; f.x2[3] = bz2; f.n[3] = nc;
  f.x1[4] = CZ; f.x2[4] = cz2; f.n[4] = nc;
  hipLaunchKernelGGL(k_fold5<F>, dim3(1024), dim3(256), 0, s, f, r);
  if (E2) {      // the other operand's own error vector: E += r²·E_2
    Fold5 g; for (int k = 0; k < 5; k++) { g.x1[k] = nullptr; g.x2[k] = nullptr; g.n[k] = 0; }
    g.x1[0] = E; g.x2[0] = E2; g.n[0] = nc;
    hipLaunchKernelGGL(k_fold5<F>, dim3(1024), dim3(256), 0, s, g, F::mul(r, r));
  }
}
// absorb `next` into m on the device and in the records (caller holds both contexts' locks and has set the device)
int cfm_merge_locked(vimz_cf_merged* m, vimz_cf* next) {
  vimz_cf* vk = m->vk; vimz_ctx* ctx = vk->ctx; vimz_prover* p = vk->pri; vimz_prover* pn = next->pri;
  hipStream_t s = ctx->stream;
  const size_t nw = p->n_wires, nc = p->n_c, nw2 = vk->sec.n_w, nc2 = vk->sec.n_c;
  const SecDev& Sn = next->sec;
  const bool first = m->segs.empty();
  const double t_all = now_s();
  if (m->run_start.size() > 1) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merge: this object already holds several runs (vimz_cf_merge_merged): segments are folded into a single run");
  if (next->i == 0) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merge: the segment has no steps");
  if (next->broken) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merge: the segment's IVC failed in the middle of a step");
  if (!first) { for (uint32_t k = 0; k < p->len_z; k++) if (!m->acc.ze[k].eq(next->z0[k])) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merge: the segment does not start at the state the merged proof ends in"); }
  P_TRY(hipStreamSynchronize(next->ctx->stream));
  CfSegRec rec = cfm_record_of(next);
  struct Poison { vimz_cf_merged* m; bool armed = false; ~Poison() { if (armed) m->broken = true; } } poison{m};
  uint64_t pt[8];
  auto to_g1 = [&](G1Aff* out) { memcpy(out->x.v, pt, 32); memcpy(out->y.v, pt + 4, 32); };
  int rc;
  double t0 = now_s();
  Fe u_acc = first ? Fe::zero() : m->acc.u; Fq qu_acc = first ? Fq::zero() : m->acc.qu;
  uint8_t prev[32], h[32];
  if (first) { Sha3 h0; const char* tag = "vimz-cf-merge-v1"; h0.update(tag, strlen(tag)); cfm_fe(h0, vk->c1->digest); const uint64_t lz = vk->c1->len_z; h0.update(&lz, 8); h0.finish(prev); }
  else memcpy(prev, m->acc.h, 32);
  cfm_segment_hash(prev, rec, h);
  if (first) {
    const uint32_t* src[] = {pn->Zrun, pn->E, pn->AZ, pn->BZ, pn->CZ, Sn.Zrun, Sn.E, Sn.AZ, Sn.BZ, Sn.CZ};
    uint32_t* dst[] = {m->Zp, m->Ep, m->AZp, m->BZp, m->CZp, m->Zq, m->Eq, m->AZq, m->BZq, m->CZq};
    const size_t len[] = {nw, nc, nc, nc, nc, nw2, nc2, nc2, nc2, nc2};
    for (int k = 0; k < 10; k++) P_TRY(hipMemcpyAsync(dst[k], src[k], 32 * len[k], hipMemcpyDeviceToDevice, s));
    u_acc = next->u_run; qu_acc = next->cf_u_run;
    poison.armed = true;
  } else {
    // acc (+) U_j: cross term of two relaxed instances, its commitment, challenge, fold
    hipLaunchKernelGGL(k_cross_term<Fr>, dim3(stream_grid(nc)), dim3(256), 0, s, nc, m->AZp, m->BZp, m->CZp, u_acc, pn->AZ, pn->BZ, pn->CZ, next->u_run, m->Tp);
    P_TRY(hipGetLastError());
    if ((rc = vz_msm_device(ctx, p->ck, 0, m->Tp, nc, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
    to_g1(&rec.T1);
    // cfacc (+) cfU_j on Grumpkin
    hipLaunchKernelGGL(k_cross_term<Fq>, dim3(stream_grid(nc2)), dim3(256), 0, s, nc2, m->AZq, m->BZq, m->CZq, qu_acc, Sn.AZ, Sn.BZ, Sn.CZ, next->cf_u_run, m->Tq);
    P_TRY(hipGetLastError());
    if ((rc = vz_msm_device(ctx, vk->ck2, 0, m->Tq, nc2, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
    memcpy(rec.Tc.x.v, pt, 32); memcpy(rec.Tc.y.v, pt + 4, 32);
    m->seconds[0] += now_s() - t0; t0 = now_s();
    uint32_t r1[4], rcq[4]; std::vector<uint8_t> b;
    cfm_points_bytes(&rec.T1, 1, b); cfm_chal(h, 'a', b.data(), b.size(), r1);
    { std::vector<uint8_t> q; for (const Fe* x : {&rec.Tc.x, &rec.Tc.y}) { const Fe v = Fe::from_mont(*x); const uint8_t* p8 = (const uint8_t*)v.v; q.insert(q.end(), p8, p8 + 32); } cfm_chal(h, 'c', q.data(), q.size(), rcq); }
    poison.armed = true;
    const Fe r1f = cfm_fe128<Fe>(r1); const Fq rcf = cfm_fe128<Fq>(rcq);
    cfm_fold<Fr>(s, m->Zp, m->Ep, m->AZp, m->BZp, m->CZp, nw, nc, pn->Zrun, m->Tp, pn->AZ, pn->BZ, pn->CZ, pn->E, r1f);
    cfm_fold<Fq>(s, m->Zq, m->Eq, m->AZq, m->BZq, m->CZq, nw2, nc2, Sn.Zrun, m->Tq, Sn.AZ, Sn.BZ, Sn.CZ, Sn.E, rcf);
    P_TRY(hipGetLastError());
    u_acc = Fe::add(u_acc, Fe::mul(r1f, next->u_run)); qu_acc = Fq::add(qu_acc, Fq::mul(rcf, next->cf_u_run));
    m->seconds[1] += now_s() - t0; t0 = now_s();
  }
  // acc (+) u_j: the segment's last instance of F' (strict)
  hipLaunchKernelGGL(k_cross_term<Fr>, dim3(stream_grid(nc)), dim3(256), 0, s, nc, m->AZp, m->BZp, m->CZp, u_acc, next->azl, next->bzl, next->czl, Fe::one(), m->Tp);
  P_TRY(hipGetLastError());
  if ((rc = vz_msm_device(ctx, p->ck, 0, m->Tp, nc, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  to_g1(&rec.T2);
  m->seconds[0] += now_s() - t0; t0 = now_s();
  uint32_t r2[4]; { std::vector<uint8_t> b; const G1Aff t12[2] = {rec.T1, rec.T2}; cfm_points_bytes(t12, 2, b); cfm_chal(h, 'b', b.data(), b.size(), r2); }
  cfm_fold<Fr>(s, m->Zp, m->Ep, m->AZp, m->BZp, m->CZp, nw, nc, next->Zl, m->Tp, next->azl, next->bzl, next->czl, nullptr, cfm_fe128<Fe>(r2));
  P_TRY(hipGetLastError());
  P_TRY(hipStreamSynchronize(s));
  m->seconds[1] += now_s() - t0; t0 = now_s();
  // the records and the folded instances (host curve arithmetic); the device vectors above used the same challenges
  uint32_t flags = 0; CfmChallenges ch;
  cfm_absorb(vk, m->acc, rec, first, &flags, &ch);
  if (flags || memcmp(ch.r2, r2, 16)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merge: the segment's proof is inconsistent (hashes / statement)");
  m->segs.push_back(std::move(rec));
  poison.armed = false;
  m->seconds[2] += now_s() - t0; m->seconds[3] += now_s() - t_all;
  return VIMZ_OK;
}
void cfm_orphan_dependents(vimz_cf* v) {
  std::lock_guard<std::mutex> g(v->ctx->mu);
  hipSetDevice(v->ctx->device);
  hipStreamSynchronize(v->ctx->stream);
  for (vimz_cf_merged* m : v->merged_dependents) { if (m->dev) hipFree(m->dev); m->dev = nullptr; m->vk = nullptr; m->broken = true; }
  v->merged_dependents.clear();
}
bool cfm_same_shapes(const vimz_cf* a, const vimz_cf* b) {
  return a->ctx->device == b->ctx->device && a->ck1 == b->ck1 && a->ck2 == b->ck2 && a->pri->n_wires == b->pri->n_wires && a->pri->n_c == b->pri->n_c &&
         a->c1->digest.eq(b->c1->digest);
}
}  // namespace

extern "C" {

void vimz_cf_merged_free(vimz_cf_merged* m) {
  if (!m) return;
  if (m->vk) {
    std::lock_guard<std::mutex> g(m->vk->ctx->mu);
    auto& d = m->vk->merged_dependents;
    d.erase(std::remove(d.begin(), d.end(), m), d.end());
    if (m->dev) { hipSetDevice(m->vk->ctx->device); hipStreamSynchronize(m->vk->ctx->stream); hipFree(m->dev); }
  }
  delete m;
}
// the merged proof of one segment; `first` is left unchanged, supplies shapes / keys / context and must outlive the object
int vimz_cf_merged_create(vimz_cf* first, vimz_cf_merged** out) {
  if (!first || !out) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = first->ctx;
  std::unique_ptr<vimz_cf_merged> m(new vimz_cf_merged());
  m->vk = first;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  const size_t nw = first->pri->n_wires, nc = first->pri->n_c, nw2 = first->sec.n_w, nc2 = first->sec.n_c;
  P_TRY(hipMalloc((void**)&m->dev, 32 * (nw + 5 * nc + nw2 + 5 * nc2)));
  uint32_t* q = m->dev;
  m->Zp = q; q += 8 * nw; m->Ep = q; q += 8 * nc; m->AZp = q; q += 8 * nc; m->BZp = q; q += 8 * nc; m->CZp = q; q += 8 * nc; m->Tp = q; q += 8 * nc;
  m->Zq = q; q += 8 * nw2; m->Eq = q; q += 8 * nc2; m->AZq = q; q += 8 * nc2; m->BZq = q; q += 8 * nc2; m->CZq = q; q += 8 * nc2; m->Tq = q;
  int rc = cfm_merge_locked(m.get(), first);
  if (rc) { hipFree(m->dev); m->dev = nullptr; return rc; }
  first->merged_dependents.push_back(m.get()); first->orphan_merged = cfm_orphan_dependents;
  *out = m.release();
  return VIMZ_OK;
}
// fold the proof of the NEXT row segment in (same device; read in place, left unchanged): it must start at the state m ends in
int vimz_cf_merge(vimz_cf_merged* m, vimz_cf* next) {
  if (!m || !next || !m->vk) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = m->vk->ctx;
  if (m->broken) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merge: this merged proof failed in the middle of a merge");
  if (!cfm_same_shapes(m->vk, next)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merge: the segment was proven for another circuit, other keys or on another device");
  std::mutex* a = &ctx->mu; std::mutex* b = &next->ctx->mu;
  if (a == b) { std::lock_guard<std::mutex> g(*a); P_TRY(hipSetDevice(ctx->device)); return cfm_merge_locked(m, next); }
  if (b < a) std::swap(a, b);
  std::lock_guard<std::mutex> g1(*a); std::lock_guard<std::mutex> g2(*b);
  P_TRY(hipSetDevice(ctx->device));
  return cfm_merge_locked(m, next);
}
// fold another merged object — ONE run, e.g. what another GPU made of its rows — in: it must start at the state m ends in.  `other` is read in
// place and left unchanged (same device).
int vimz_cf_merge_merged(vimz_cf_merged* m, vimz_cf_merged* other) {
  if (!m || !other || m == other || !m->vk || !other->vk) return VIMZ_ERR_INVALID;
  vimz_cf* vk = m->vk; vimz_ctx* ctx = vk->ctx;
  if (m->broken || other->broken) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merge_merged: one of the objects failed in the middle of a merge");
  if (!cfm_same_shapes(vk, other->vk)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merge_merged: the objects are about different circuits, keys or devices");
  if (other->run_start.size() != 1 || m->segs.empty() || other->segs.empty()) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merge_merged: the object to fold in must hold exactly one run");
  std::mutex* a = &ctx->mu; std::mutex* b = &other->vk->ctx->mu;
  std::unique_lock<std::mutex> l1, l2;
  if (a == b) l1 = std::unique_lock<std::mutex>(*a);
  else { if (b < a) std::swap(a, b); l1 = std::unique_lock<std::mutex>(*a); l2 = std::unique_lock<std::mutex>(*b); }
  P_TRY(hipSetDevice(ctx->device));
  for (size_t k = 0; k < m->acc.ze.size(); k++) if (!m->acc.ze[k].eq(other->acc.zs[k])) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merge_merged: the object does not start at the state this one ends in");
  hipStream_t s = ctx->stream;
  vimz_prover* p = vk->pri; const size_t nw = p->n_wires, nc = p->n_c, nw2 = vk->sec.n_w, nc2 = vk->sec.n_c;
  P_TRY(hipStreamSynchronize(other->vk->ctx->stream));
  const double t0 = now_s();
  CfJunction J; uint64_t pt[8]; int rc;
  hipLaunchKernelGGL(k_cross_term<Fr>, dim3(stream_grid(nc)), dim3(256), 0, s, nc, m->AZp, m->BZp, m->CZp, m->acc.u, other->AZp, other->BZp, other->CZp, other->acc.u, m->Tp);
  P_TRY(hipGetLastError());
  if ((rc = vz_msm_device(ctx, p->ck, 0, m->Tp, nc, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  memcpy(J.Tp.x.v, pt, 32); memcpy(J.Tp.y.v, pt + 4, 32);
  hipLaunchKernelGGL(k_cross_term<Fq>, dim3(stream_grid(nc2)), dim3(256), 0, s, nc2, m->AZq, m->BZq, m->CZq, m->acc.qu, other->AZq, other->BZq, other->CZq, other->acc.qu, m->Tq);
  P_TRY(hipGetLastError());
  if ((rc = vz_msm_device(ctx, vk->ck2, 0, m->Tq, nc2, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  memcpy(J.Tq.x.v, pt, 32); memcpy(J.Tq.y.v, pt + 4, 32);
  m->seconds[0] += now_s() - t0;
  const double t1 = now_s();
  CfAcc A = m->acc; uint32_t fl = 0, rp[4], rq[4];
  cfm_node(A, other->acc, J, &fl, rp, rq);
  if (fl) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merge_merged: the objects are not adjacent");
  struct Poison { vimz_cf_merged* m; bool armed = true; ~Poison() { if (armed) m->broken = true; } } poison{m};
  cfm_fold<Fr>(s, m->Zp, m->Ep, m->AZp, m->BZp, m->CZp, nw, nc, other->Zp, m->Tp, other->AZp, other->BZp, other->CZp, other->Ep, cfm_fe128<Fe>(rp));
  cfm_fold<Fq>(s, m->Zq, m->Eq, m->AZq, m->BZq, m->CZq, nw2, nc2, other->Zq, m->Tq, other->AZq, other->BZq, other->CZq, other->Eq, cfm_fe128<Fq>(rq));
  P_TRY(hipGetLastError());
  P_TRY(hipStreamSynchronize(s));
  m->run_start.push_back((uint32_t)m->segs.size());
  m->segs.insert(m->segs.end(), other->segs.begin(), other->segs.end());
  m->junctions.push_back(J);
  m->acc = A;
  poison.armed = false;
  m->seconds[1] += now_s() - t1; m->seconds[3] += now_s() - t0;
  return VIMZ_OK;
}

// The object as bytes (a GPU's run on its way to the rank that folds the runs; a proof on disk): the records, then the folded witnesses and
// running products of both sides as they sit in HBM.  load: into the context of `vk` (a vimz_cf for the same step circuit and keys); every
// element range-checked, every point checked to be on its curve, the accumulator recomputed from the records.
size_t vimz_cf_merged_size(const vimz_cf_merged* m) {
  if (!m || !m->vk) return 0;
  const size_t nw = m->vk->pri->n_wires, nc = m->vk->pri->n_c, nw2 = m->vk->sec.n_w, nc2 = m->vk->sec.n_c;
  return (size_t)vimz_cf_merged_records(m, nullptr, 0) + 32 * (nw + 4 * nc + nw2 + 4 * nc2);
}
int vimz_cf_merged_save(vimz_cf_merged* m, uint8_t* blob, size_t cap) {
  if (!m || !m->vk || !blob) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = m->vk->ctx;
  if (m->broken) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merged_save: this object failed in the middle of a merge");
  if (cap < vimz_cf_merged_size(m)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merged_save: buffer too small");
  const int64_t rb = vimz_cf_merged_records(m, blob, cap);
  if (rb < 0) return (int)rb;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t nw = m->vk->pri->n_wires, nc = m->vk->pri->n_c, nw2 = m->vk->sec.n_w, nc2 = m->vk->sec.n_c;
  const uint32_t* src[] = {m->Zp, m->Ep, m->AZp, m->BZp, m->CZp, m->Zq, m->Eq, m->AZq, m->BZq, m->CZq};
  const size_t len[] = {nw, nc, nc, nc, nc, nw2, nc2, nc2, nc2, nc2};
  uint8_t* o = blob + rb;
  for (int k = 0; k < 10; k++) { P_TRY(hipMemcpyAsync(o, src[k], 32 * len[k], hipMemcpyDeviceToHost, s)); o += 32 * len[k]; }
  P_TRY(hipStreamSynchronize(s));
  return VIMZ_OK;
}
// parse and replay the records of an untrusted blob / ticket (w: 8-byte words) into m (vk set); *rec_words_out = their length
static int cfm_parse_records(vimz_cf* vk, const uint64_t* w, size_t nwords, vimz_cf_merged* m, size_t* rec_words_out, const char* who) {
  vimz_ctx* ctx = vk->ctx; vimz_prover* p = vk->pri;
  const size_t lz = p->len_z, nw = p->n_wires, nc = p->n_c, nw2 = vk->sec.n_w, nc2 = vk->sec.n_c;
  if (nwords < 8) return vz_fail(ctx, VIMZ_ERR_INVALID, who);
  const uint64_t S = w[1], R = w[7];
  if (w[0] != CF_MERGED_MAGIC || w[2] != lz || w[3] != nw || w[4] != nc || w[5] != nw2 || w[6] != nc2 || S == 0 || S > 4096 || R == 0 || R > S)
    return vz_fail(ctx, VIMZ_ERR_INVALID, "merged CycleFold proof: the records do not match this prover's circuits");
  const size_t seg_words = 1 + 4 * (2 * lz + 7 + 4 + 5 + CF_IO + 6);
  const size_t rec_words = 8 + R + S * seg_words + (R - 1) * 16;
  if (nwords < rec_words) return vz_fail(ctx, VIMZ_ERR_INVALID, "merged CycleFold proof: records too short");
  m->run_start.clear();
  size_t pos = 8;
  for (uint64_t k = 0; k < R; k++) { if (w[pos] >= S || (k && w[pos] <= m->run_start.back())) return vz_fail(ctx, VIMZ_ERR_INVALID, "merged CycleFold proof: malformed runs"); m->run_start.push_back((uint32_t)w[pos++]); }
  if (m->run_start[0] != 0) return vz_fail(ctx, VIMZ_ERR_INVALID, "merged CycleFold proof: malformed runs");
  bool ok = true;
  auto fe = [&](auto* dst) { typedef std::decay_t<decltype(*dst)> F; F c; memcpy(c.v, w + pos, 32); pos += 4; if (!c.is_reduced()) ok = false; *dst = F::to_mont(c); };
  auto u256 = [&](U256w* dst) { memcpy(dst->w, w + pos, 32); pos += 4; Fq c; memcpy(c.v, dst->w, 32); if (!c.is_reduced()) ok = false; };
  auto g1 = [&](G1Aff* P) { fe(&P->x); fe(&P->y); if (ok && !aff_on_curve(*P)) ok = false; };
  auto g2 = [&](G2Aff* P) { fe(&P->x); fe(&P->y); if (ok && !aff_on_curve(*P)) ok = false; };
  m->segs.assign(S, CfSegRec());
  for (auto& sg : m->segs) {
    sg.n = w[pos++];
    sg.zs.resize(lz); sg.ze.resize(lz);
    for (auto& z : sg.zs) fe(&z);
    for (auto& z : sg.ze) fe(&z);
    g1(&sg.UW); g1(&sg.UE); fe(&sg.U.u); fe(&sg.U.x0); fe(&sg.U.x1); sg.U.W = nn_point(sg.UW); sg.U.E = nn_point(sg.UE);
    g1(&sg.uW); fe(&sg.u.x0); fe(&sg.u.x1); sg.u.W = nn_point(sg.uW);
    g2(&sg.cfU.W); g2(&sg.cfU.E); fe(&sg.cfU.u); for (auto& e : sg.cfU.x) u256(&e);
    g1(&sg.T1); g1(&sg.T2); g2(&sg.Tc);
  }
  m->junctions.assign(R - 1, CfJunction());
  for (auto& j : m->junctions) { g1(&j.Tp); g2(&j.Tq); }
  if (!ok || pos != rec_words) return vz_fail(ctx, VIMZ_ERR_INVALID, "merged CycleFold proof: an element of the records is not below its modulus, or a point is not on its curve");
  uint32_t fl = 0;
  if (!cfm_replay(vk, m->segs, m->run_start, m->junctions, m->acc, &fl)) return vz_fail(ctx, VIMZ_ERR_INVALID, "merged CycleFold proof: malformed records");
  *rec_words_out = rec_words;
  return VIMZ_OK;
}

int vimz_cf_merged_load(vimz_cf* vk, const uint8_t* blob, size_t len, vimz_cf_merged** out) {
  if (!vk || !blob || !out || len < 64) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = vk->ctx; vimz_prover* p = vk->pri;
  const size_t nw = p->n_wires, nc = p->n_c, nw2 = vk->sec.n_w, nc2 = vk->sec.n_c;
  const uint64_t* w = reinterpret_cast<const uint64_t*>(blob);      // (blobs come from numpy / malloc: 8-byte aligned)
  if ((uintptr_t)blob % 8) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merged_load: blob not 8-byte aligned");
  std::unique_ptr<vimz_cf_merged> m(new vimz_cf_merged());
  m->vk = vk;
  size_t rec_words = 0;
  int rcp = cfm_parse_records(vk, w, len / 8, m.get(), &rec_words, "vimz_cf_merged_load: blob too short");
  if (rcp) return rcp;
  if (len < 8 * rec_words + 32 * (nw + 4 * nc + nw2 + 4 * nc2)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merged_load: blob too short");
  // (hash / adjacency failures of the replay are the verifier's to report: the object loads and vimz_cf_merged_verify rejects it)
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  P_TRY(hipMalloc((void**)&m->dev, 32 * (nw + 5 * nc + nw2 + 5 * nc2)));
  uint32_t* q = m->dev;
  m->Zp = q; q += 8 * nw; m->Ep = q; q += 8 * nc; m->AZp = q; q += 8 * nc; m->BZp = q; q += 8 * nc; m->CZp = q; q += 8 * nc; m->Tp = q; q += 8 * nc;
  m->Zq = q; q += 8 * nw2; m->Eq = q; q += 8 * nc2; m->AZq = q; q += 8 * nc2; m->BZq = q; q += 8 * nc2; m->CZq = q; q += 8 * nc2; m->Tq = q;
  uint32_t* dst[] = {m->Zp, m->Ep, m->AZp, m->BZp, m->CZp, m->Zq, m->Eq, m->AZq, m->BZq, m->CZq};
  const size_t ln[] = {nw, nc, nc, nc, nc, nw2, nc2, nc2, nc2, nc2};
  const uint8_t* o = blob + 8 * rec_words;
  uint32_t* badc = m->Tp;      // (scratch until the first merge)
  hipError_t e = hipMemsetAsync(badc, 0, 8, s);
  for (int k = 0; k < 10 && e == hipSuccess; k++) {
    e = hipMemcpyAsync(dst[k], o, 32 * ln[k], hipMemcpyHostToDevice, s); o += 32 * ln[k];
    if (e != hipSuccess) break;
    if (k < 5) hipLaunchKernelGGL(k_count_unreduced<Fr>, dim3(stream_grid(ln[k])), dim3(256), 0, s, ln[k], (const uint32_t*)dst[k], badc);
    else hipLaunchKernelGGL(k_count_unreduced<Fq>, dim3(stream_grid(ln[k])), dim3(256), 0, s, ln[k], (const uint32_t*)dst[k], badc);
  }
  uint32_t nbad = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&nbad, badc, 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess || nbad) { hipFree(m->dev); m->dev = nullptr; return e != hipSuccess ? vz_fail(ctx, VIMZ_ERR_HIP, "vimz_cf_merged_load: upload", e) : vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merged_load: a vector element is not below its modulus"); }
  vk->merged_dependents.push_back(m.get()); vk->orphan_merged = cfm_orphan_dependents;
  *out = m.release();
  return VIMZ_OK;
}

static std::vector<uint64_t> cfm_records_words(const uint64_t shape[5], const std::vector<CfSegRec>& segs, const std::vector<uint32_t>& run_start, const std::vector<CfJunction>& junctions);
// The hand-over between the processes of one node without the host round trip (as vimz_ivc_merged_share / _open_shared: the ticket = the
// records and a HIP IPC handle of the object's one device allocation; the receiver copies the ten vectors device-to-device).
static const uint64_t CF_SHARE_MAGIC = 0x3148534643565aull;      // "ZVCFSH1"
int64_t vimz_cf_merged_share(vimz_cf_merged* m, void* buf, size_t cap) {
  if (!m || !m->vk || !m->dev) return VIMZ_ERR_INVALID;
  vimz_cf* vk = m->vk; vimz_ctx* ctx = vk->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  if (m->broken) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merged_share: this object failed in the middle of a merge");
  const uint64_t shape[5] = {vk->pri->len_z, vk->pri->n_wires, vk->pri->n_c, vk->sec.n_w, vk->sec.n_c};
  const std::vector<uint64_t> rec = cfm_records_words(shape, m->segs, m->run_start, m->junctions);
  const size_t bytes = 8 * (4 + 8 + rec.size());
  if (!buf || cap < bytes) return (int64_t)bytes;
  if (hipSetDevice(ctx->device) != hipSuccess) return VIMZ_ERR_HIP;
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, m->dev);
  if (e != hipSuccess) return vz_fail(ctx, VIMZ_ERR_HIP, "vimz_cf_merged_share: hipIpcGetMemHandle", e);
  std::vector<uint64_t> o = {CF_SHARE_MAGIC, (uint64_t)(vk->pri->n_wires + 5 * (size_t)vk->pri->n_c + vk->sec.n_w + 5 * (size_t)vk->sec.n_c), 0, 0};
  uint64_t hw[8]; memcpy(hw, &h, 64); o.insert(o.end(), hw, hw + 8);
  o.insert(o.end(), rec.begin(), rec.end());
  memcpy(buf, o.data(), bytes);
  return (int64_t)bytes;
}
int vimz_cf_merged_open_shared(vimz_cf* vk, const uint8_t* ticket, size_t len, vimz_cf_merged** out) {
  if (!vk || !ticket || !out || (len & 7) || len < 8 * 20) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = vk->ctx; vimz_prover* p = vk->pri;
  const size_t nw = p->n_wires, nc = p->n_c, nw2 = vk->sec.n_w, nc2 = vk->sec.n_c;
  std::vector<uint64_t> words(len / 8); memcpy(words.data(), ticket, len);
  const size_t elements = nw + 5 * nc + nw2 + 5 * nc2;
  if (words[0] != CF_SHARE_MAGIC || words[1] != elements) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merged_open_shared: not a ticket for this prover's circuits");
  hipIpcMemHandle_t h; memcpy(&h, words.data() + 4, 64);
  std::unique_ptr<vimz_cf_merged> m(new vimz_cf_merged());
  m->vk = vk;
  size_t rec_words = 0;
  int rcp = cfm_parse_records(vk, words.data() + 12, words.size() - 12, m.get(), &rec_words, "vimz_cf_merged_open_shared: ticket too short");
  if (rcp) return rcp;
  if (rec_words != words.size() - 12) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merged_open_shared: malformed ticket");
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  void* remote = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&remote, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess || !remote) return vz_fail(ctx, VIMZ_ERR_HIP, "vimz_cf_merged_open_shared: hipIpcOpenMemHandle (not the same node, or no peer access between the two GPUs)", e);
  e = hipMalloc((void**)&m->dev, 32 * elements);
  if (e != hipSuccess) { hipIpcCloseMemHandle(remote); return vz_fail(ctx, VIMZ_ERR_HIP, "vimz_cf_merged_open_shared: hipMalloc", e); }
  uint32_t* q = m->dev;
  m->Zp = q; q += 8 * nw; m->Ep = q; q += 8 * nc; m->AZp = q; q += 8 * nc; m->BZp = q; q += 8 * nc; m->CZp = q; q += 8 * nc; m->Tp = q; q += 8 * nc;
  m->Zq = q; q += 8 * nw2; m->Eq = q; q += 8 * nc2; m->AZq = q; q += 8 * nc2; m->BZq = q; q += 8 * nc2; m->CZq = q; q += 8 * nc2; m->Tq = q;
  e = hipMemcpyAsync(m->dev, remote, 32 * elements, hipMemcpyDefault, s);
  uint32_t* badc = m->Tq;      // (scratch until the first merge; the copied value there is scratch too)
  uint32_t nbad = 1;
  if (e == hipSuccess) e = hipMemsetAsync(badc, 0, 8, s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_count_unreduced<Fr>, dim3(stream_grid(nw + 4 * nc)), dim3(256), 0, s, (size_t)(nw + 4 * nc), (const uint32_t*)m->Zp, badc);
    hipLaunchKernelGGL(k_count_unreduced<Fq>, dim3(stream_grid(nw2 + 4 * nc2)), dim3(256), 0, s, (size_t)(nw2 + 4 * nc2), (const uint32_t*)m->Zq, badc);
    e = hipMemcpyAsync(&nbad, badc, 4, hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipIpcCloseMemHandle(remote);
  if (e != hipSuccess || nbad) { hipFree(m->dev); m->dev = nullptr; return e != hipSuccess ? vz_fail(ctx, VIMZ_ERR_HIP, "vimz_cf_merged_open_shared: copy", e) : vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merged_open_shared: a vector element is not below its modulus"); }
  vk->merged_dependents.push_back(m.get()); vk->orphan_merged = cfm_orphan_dependents;
  *out = m.release();
  return VIMZ_OK;
}

int vimz_cf_merged_info(const vimz_cf_merged* m, uint64_t info[8]) {
  if (!m || !info || !m->vk) return VIMZ_ERR_INVALID;
  info[0] = m->acc.n; info[1] = m->segs.size(); info[2] = m->vk->pri->len_z; info[3] = m->vk->pri->n_wires; info[4] = m->vk->pri->n_c;
  info[5] = m->vk->sec.n_w; info[6] = m->vk->sec.n_c; info[7] = m->broken ? 1 : 0;
  return VIMZ_OK;
}
int vimz_cf_merged_state(const vimz_cf_merged* m, uint64_t* z_start, uint64_t* z_end, uint64_t* steps) {
  if (!m) return VIMZ_ERR_INVALID;
  for (size_t k = 0; k < m->acc.zs.size(); k++) { if (z_start) fe_to_canon(m->acc.zs[k], z_start + 4 * k); if (z_end) fe_to_canon(m->acc.ze[k], z_end + 4 * k); }
  if (steps) *steps = m->acc.n;
  return VIMZ_OK;
}
int vimz_cf_merged_profile(const vimz_cf_merged* m, double seconds[4]) {
  if (!m || !seconds) return VIMZ_ERR_INVALID;
  for (int k = 0; k < 4; k++) seconds[k] = m->seconds[k];
  return VIMZ_OK;
}
// The statement part as canonical little-endian words: header (magic, segments, len_z, main wires, main constraints, CycleFold wires,
// CycleFold constraints, runs), the first segment of every run, then per segment: n; z_start; z_end; U = comm_W.x, .y, comm_E.x, .y, u, x0, x1; u = comm_W.x, .y, x0, x1;
// cfU = comm_W.x, .y, comm_E.x, .y, u, x[0..7); T1.x, .y; T2.x, .y; Tc.x, .y  (every element four words); then per junction T_p.x, .y, T_q.x, .y.
static std::vector<uint64_t> cfm_records_words(const uint64_t shape[5] /* len_z, main wires, main constraints, CycleFold wires, CycleFold constraints */,
                                               const std::vector<CfSegRec>& segs, const std::vector<uint32_t>& run_start, const std::vector<CfJunction>& junctions) {
  std::vector<uint64_t> o = {CF_MERGED_MAGIC, segs.size(), shape[0], shape[1], shape[2], shape[3], shape[4], run_start.size()};
  for (uint32_t r0 : run_start) o.push_back(r0);
  auto push = [&](const auto& v) { auto x = std::decay_t<decltype(v)>::from_mont(v); o.resize(o.size() + 4); memcpy(o.data() + o.size() - 4, x.v, 32); };
  auto push_u = [&](const U256w& x) { o.insert(o.end(), x.w, x.w + 4); };
  for (auto& s : segs) {
    o.push_back(s.n);
    for (auto& z : s.zs) push(z);
    for (auto& z : s.ze) push(z);
    push(s.UW.x); push(s.UW.y); push(s.UE.x); push(s.UE.y); push(s.U.u); push(s.U.x0); push(s.U.x1);
    push(s.uW.x); push(s.uW.y); push(s.u.x0); push(s.u.x1);
    push(s.cfU.W.x); push(s.cfU.W.y); push(s.cfU.E.x); push(s.cfU.E.y); push(s.cfU.u); for (auto& e : s.cfU.x) push_u(e);
    push(s.T1.x); push(s.T1.y); push(s.T2.x); push(s.T2.y); push(s.Tc.x); push(s.Tc.y);
  }
  for (auto& j : junctions) { push(j.Tp.x); push(j.Tp.y); push(j.Tq.x); push(j.Tq.y); }
  return o;
}
int64_t vimz_cf_merged_records(const vimz_cf_merged* m, void* buf, size_t cap) {
  if (!m || !m->vk) return VIMZ_ERR_INVALID;
  const uint64_t shape[5] = {m->vk->pri->len_z, m->vk->pri->n_wires, m->vk->pri->n_c, m->vk->sec.n_w, m->vk->sec.n_c};
  const std::vector<uint64_t> o = cfm_records_words(shape, m->segs, m->run_start, m->junctions);
  const size_t bytes = o.size() * 8;
  if (buf && cap >= bytes) memcpy(buf, o.data(), bytes);
  return (int64_t)bytes;
}
#ifdef VIMZ_TESTING      // (include/vimz_hip_testing.h: only in libvimz_hip_testing.so)
// Host only, no GPU (test hook for the CPU suite): two runs of made-up segment records — commitments are multiples of the generators, the hashes
// each segment's last instance carries are the true ones of its made-up statement — replayed by cfm_replay.  Output: digest (4 words), the records
// (vimz_cf_merged_records' layout, len_z = 1), then the accumulator the replay arrives at: n, z_start, z_end (one element each), comm_W.x, .y, comm_E.x,
// .y, u, x0, x1 of the main side, comm_W.x, .y, comm_E.x, .y, u, x[0..7) of the CycleFold side.  An outside replay (tests/_cyclefold.py) must agree.
int64_t vimz_cf_selfcheck_merge(int segs_run0, int segs_run1, void* buf, size_t cap) {
  if (segs_run0 < 1 || segs_run0 > 8 || segs_run1 < 0 || segs_run1 > 8) return VIMZ_ERR_INVALID;
  try {
    CfCircuit cfc; cfc.finish();
    cb::BuilderT<Fe> b;
    b.len_z = 1; b.n_priv = 0; b.n_wires = 3;
    b.enforce(cb::LCT<Fe>::constant(Fe::one()), cb::LCT<Fe>::wire(2), cb::LCT<Fe>::wire(1));
    b.n_linear = 1;
    vimz_cf vk;
    vk.c1.reset(new CfMainCircuit(b)); vk.c1->use_worker = false; vk.c1->finish(cfc);
    const Fe dg = vk.c1->digest;
    const G1Aff g1 = CycleSide<BnFq>::G(); const G2Aff g2 = CycleSide<BnFr>::G();
    uint64_t ctr = 0x9e3779b97f4a7c15ull;
    auto next = [&] { ctr = ctr * 6364136223846793005ull + 1442695040888963407ull; return ctr >> 8; };
    auto p1 = [&] { const uint64_t k = next(); const uint32_t w[2] = {(uint32_t)k, (uint32_t)(k >> 32)}; return to_affine(scalar_mul(g1, w, 56)); };
    auto p2 = [&] { const uint64_t k = next(); const uint32_t w[2] = {(uint32_t)k, (uint32_t)(k >> 32)}; return to_affine(host_mul<Fe>(g2, w, 56)); };
    auto fe = [&] { return Fe::mul(cb::f_from_u64<Fe>(next()), cb::f_from_u64<Fe>(next())); };
    const int S = segs_run0 + segs_run1;
    std::vector<CfSegRec> segs((size_t)S);
    Fe z = cb::f_from_u64<Fe>(5);
    for (int j = 0; j < S; j++) {
      CfSegRec& s = segs[(size_t)j];
      s.n = 1 + (uint64_t)(next() % 7);
      s.zs = {z}; z = fe(); s.ze = {z};
      s.UW = p1(); s.UE = p1(); s.U.W = nn_point(s.UW); s.U.E = nn_point(s.UE); s.U.u = fe(); s.U.x0 = fe(); s.U.x1 = fe();
      s.cfU.W = p2(); s.cfU.E = p2(); s.cfU.u = cb::f_from_u64<Fe>(next());
      for (auto& e : s.cfU.x) e = to_u256(Fq::mul(cb::f_from_u64<Fq>(next()), cb::f_from_u64<Fq>(next())));
      s.uW = p1(); s.u.W = nn_point(s.uW);
      s.u.x0 = cf_hash_main(dg, s.n, s.zs, s.ze.data(), s.U); s.u.x1 = cf_hash_cf(dg, s.cfU);
      const bool first_of_run = j == 0 || j == segs_run0;
      s.T1 = first_of_run ? g1_identity() : p1(); s.T2 = p1(); s.Tc = first_of_run ? g2_identity() : p2();
    }
    std::vector<uint32_t> run_start = {0};
    std::vector<CfJunction> junctions;
    if (segs_run1) { run_start.push_back((uint32_t)segs_run0); CfJunction J; J.Tp = p1(); J.Tq = p2(); junctions.push_back(J); }
    CfAcc acc; uint32_t fl = 0;
    if (!cfm_replay(&vk, segs, run_start, junctions, acc, &fl) || fl) return VIMZ_ERR_UNSAT;
    std::vector<uint64_t> o;
    auto push = [&](const auto& v) { auto x = std::decay_t<decltype(v)>::from_mont(v); o.resize(o.size() + 4); memcpy(o.data() + o.size() - 4, x.v, 32); };
    push(dg);
    const uint64_t shape[5] = {1, b.n_wires, b.n_constraints(), cfc.n_wires(), cfc.n_constraints()};
    const std::vector<uint64_t> rec = cfm_records_words(shape, segs, run_start, junctions);
    o.push_back(rec.size()); o.insert(o.end(), rec.begin(), rec.end());
    o.push_back(acc.n); push(acc.zs[0]); push(acc.ze[0]);
    push(acc.cW.x); push(acc.cW.y); push(acc.cE.x); push(acc.cE.y); push(acc.u); push(acc.x0); push(acc.x1);
    push(acc.qW.x); push(acc.qW.y); push(acc.qE.x); push(acc.qE.y); push(acc.qu); for (auto& e : acc.qx) push(e);
    const size_t bytes = 8 * o.size();
    if (buf && cap >= bytes) memcpy(buf, o.data(), bytes);
    return (int64_t)bytes;
  } catch (const std::exception& e) { return vz_fail(nullptr, VIMZ_ERR_INVALID, e.what()); }
}
#endif      // VIMZ_TESTING
// side 0 / 1 = main / CycleFold; what = VIMZ_IX_RUNNING_Z, VIMZ_IX_RUNNING_E (canonical), VIMZ_IX_INSTANCE (side 0: comm_W.x, .y, comm_E.x, .y,
// u, x0, x1; side 1: comm_W.x, .y, comm_E.x, .y, u, x[0..7)) of the folded instances
int64_t vimz_cf_merged_export(vimz_cf_merged* m, int side, int what, void* buf, size_t cap) {
  if (!m || !m->vk || (side != 0 && side != 1)) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = m->vk->ctx;
  if (what == VIMZ_IX_INSTANCE) {
    std::vector<uint64_t> o;
    auto push = [&](const auto& v) { auto x = std::decay_t<decltype(v)>::from_mont(v); o.resize(o.size() + 4); memcpy(o.data() + o.size() - 4, x.v, 32); };
    if (side == 0) { push(m->acc.cW.x); push(m->acc.cW.y); push(m->acc.cE.x); push(m->acc.cE.y); push(m->acc.u); push(m->acc.x0); push(m->acc.x1); }
    else { push(m->acc.qW.x); push(m->acc.qW.y); push(m->acc.qE.x); push(m->acc.qE.y); push(m->acc.qu); for (auto& e : m->acc.qx) push(e); }
    const size_t bytes = o.size() * 8;
    if (buf && cap >= bytes) memcpy(buf, o.data(), bytes);
    return (int64_t)bytes;
  }
  const uint32_t* src = nullptr; size_t n = 0;
  switch (what) {
    case VIMZ_IX_RUNNING_Z: src = side == 0 ? m->Zp : m->Zq; n = side == 0 ? m->vk->pri->n_wires : m->vk->sec.n_w; break;
    case VIMZ_IX_RUNNING_E: src = side == 0 ? m->Ep : m->Eq; n = side == 0 ? m->vk->pri->n_c : m->vk->sec.n_c; break;
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
// verify(vk, num_steps, z0) of a merged proof: replay of the records (bit 0 / 1: a segment's main / CycleFold hash; bit 12: statement — step
// count, initial state, adjacency; bit 13: the folded instances differ from the replay), then ONE main (bit 2 relation, 3 comm_W, 4 comm_E) and
// ONE CycleFold (bit 5, 6, 7) relaxed instance against the folded witnesses; bit 10 instance scalars differ from the vectors.
int vimz_cf_merged_verify(vimz_cf_merged* m, uint64_t num_steps, const uint64_t* z0, uint32_t* result) {
  if (!m || !m->vk || !z0 || !result) return VIMZ_ERR_INVALID;
  vimz_cf* vk = m->vk; vimz_ctx* ctx = vk->ctx; vimz_prover* p = vk->pri; SecDev& S = vk->sec;
  uint32_t res = m->broken ? 8192 : 0;
  CfAcc R; uint32_t fl = 0;
  if (!m->segs.empty() && !cfm_replay(vk, m->segs, m->run_start, m->junctions, R, &fl)) fl |= 4;
  if (fl & 1) res |= 1;
  if (fl & 2) res |= 2;
  if (fl & 4) res |= 4096;
  if (m->segs.empty() || R.n != num_steps) res |= 4096;
  for (uint32_t k = 0; k < p->len_z && !m->segs.empty(); k++) { Fe c; memcpy(c.v, z0 + 4 * k, 32); if (!c.is_reduced() || !Fe::to_mont(c).eq(R.zs[k])) res |= 4096; }
  if (m->segs.empty()) { *result = res; return VIMZ_OK; }
  auto same = [](const auto& a, const auto& b) { return a.x.eq(b.x) && a.y.eq(b.y); };
  if (!same(R.cW, m->acc.cW) || !same(R.cE, m->acc.cE) || !R.u.eq(m->acc.u) || !R.x0.eq(m->acc.x0) || !R.x1.eq(m->acc.x1) || !same(R.qW, m->acc.qW) || !same(R.qE, m->acc.qE) ||
      !R.qu.eq(m->acc.qu) || memcmp(R.h, m->acc.h, 32)) res |= 8192;
  for (int k = 0; k < CF_IO; k++) if (!R.qx[k].eq(m->acc.qx[k])) res |= 8192;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const uint32_t init[2] = {0, 0xffffffffu};
  uint32_t bad[2]; uint64_t pt[8]; int rc;
  auto same_pt = [&](const uint64_t* got, const auto& P) { return !memcmp(got, P.x.v, 32) && !memcmp(got + 4, P.y.v, 32); };
  launch_spmv(p, s, m->Zp, p->az2, p->bz2, p->cz2, 0);
  P_TRY(hipMemcpyAsync(p->bad_d, init, 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_check_relaxed<Fr>, dim3(stream_grid(p->n_c)), dim3(256), 0, s, (size_t)p->n_c, p->az2, p->bz2, p->cz2, R.u, (const uint32_t*)m->Ep, p->bad_d);
  P_TRY(hipMemcpyAsync(bad, p->bad_d, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 4;
  if ((rc = vz_msm_device(ctx, p->ck, 0, m->Zp + 8, p->n_wires - 3, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (!same_pt(pt, R.cW)) res |= 8;
  if ((rc = vz_msm_device(ctx, p->ck, 0, m->Ep, p->n_c, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (!same_pt(pt, R.cE)) res |= 16;
  { Fe e[3];
    if (!fetch(s, m->Zp, 0, 1, &e[0]) || !fetch(s, m->Zp, p->n_wires - 2, 2, &e[1])) return vz_fail(ctx, VIMZ_ERR_HIP, "verify: download");
    if (!e[0].eq(R.u) || !e[1].eq(R.x0) || !e[2].eq(R.x1)) res |= 1024; }
  sec_spmv<Fq>(S, s, m->Zq, S.az2, S.bz2, S.cz2);
  P_TRY(hipMemcpyAsync(S.bad, init, 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_check_relaxed<Fq>, dim3(stream_grid(S.n_c)), dim3(256), 0, s, (size_t)S.n_c, S.az2, S.bz2, S.cz2, R.qu, (const uint32_t*)m->Eq, S.bad);
  P_TRY(hipMemcpyAsync(bad, S.bad, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 32;
  if ((rc = vz_msm_device(ctx, vk->ck2, 0, m->Zq + 8, S.n_w - 1 - CF_IO, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (!same_pt(pt, R.qW)) res |= 64;
  if ((rc = vz_msm_device(ctx, vk->ck2, 0, m->Eq, S.n_c, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (!same_pt(pt, R.qE)) res |= 128;
  { Fq e[1 + CF_IO];
    if (!fetch(s, m->Zq, 0, 1, &e[0]) || !fetch(s, m->Zq, S.n_w - CF_IO, CF_IO, &e[1])) return vz_fail(ctx, VIMZ_ERR_HIP, "verify: download");
    if (!e[0].eq(R.qu)) res |= 1024;
    for (int k = 0; k < CF_IO; k++) if (!e[1 + k].eq(R.qx[k])) res |= 1024; }
  *result = res;
  return VIMZ_OK;
}
// KZG openings of the FOLDED main instance of a merged object (vimz_cf_kzg_open for a merged proof).  For a merged proof of ONE segment that is
// U_{i+1} = NIFS(U_i, u_i): the instance Sonobe's decider opens (decider.rs:13-21).  which = 0: comm_W, 1: comm_E; canonical in and out.
}  // extern "C"
int vz_cf_merged_decider_view(vimz_cf_merged* m, CfDeciderView* out) {
  if (!m || !m->vk || !out) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = m->vk->ctx;
  if (m->broken || m->segs.size() != 1 || m->run_start.size() != 1) return vz_fail(ctx, VIMZ_ERR_INVALID, "decider: the merged proof must hold exactly one segment (U_{i+1} = NIFS.V(U_i, u_i))");
  const CfSegRec& s = m->segs[0];
  out->vk = m->vk; out->n = s.n; out->zs = s.zs; out->ze = s.ze;
  out->U = s.U; out->u = s.u; out->cfU = s.cfU; out->UW = s.UW; out->UE = s.UE; out->uW = s.uW; out->cmT = s.T2;
  // the challenge as cfm_absorb derives it for the first segment of a run
  uint8_t prev[32], h[32], hn[32];
  { Sha3 h0; const char* tag = "vimz-cf-merge-v1"; h0.update(tag, strlen(tag)); cfm_fe(h0, m->vk->c1->digest); const uint64_t lz = m->vk->c1->len_z; h0.update(&lz, 8); h0.finish(prev); }
  cfm_segment_hash(prev, s, h);
  CfmChallenges ch; cfm_challenges(h, s, ch, hn);
  memcpy(out->r, ch.r2, 16);
  out->cW = m->acc.cW; out->cE = m->acc.cE; out->un = m->acc.u; out->x0n = m->acc.x0; out->x1n = m->acc.x1;
  out->Zp = m->Zp; out->Ep = m->Ep;
  return VIMZ_OK;
}
extern "C" {
int vimz_cf_merged_kzg_open(vimz_cf_merged* m, int which, const uint64_t z[4], uint64_t eval_out[4], uint64_t proof_xy[8]) {
  if (!m || !m->vk || !z || !eval_out || !proof_xy || (which != 0 && which != 1)) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = m->vk->ctx; vimz_prover* p = m->vk->pri;
  if (m->broken) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_cf_merged_kzg_open: this object failed in the middle of a merge");
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  return which == 0 ? vz_kzg_open_device(ctx, p->ck, 0, VIMZ_FIELD_BN254_FR, m->Zp + 8, p->n_wires - 3, z, VIMZ_FORM_CANONICAL, eval_out, proof_xy)
                    : vz_kzg_open_device(ctx, p->ck, 0, VIMZ_FIELD_BN254_FR, m->Ep, p->n_c, z, VIMZ_FORM_CANONICAL, eval_out, proof_xy);
}
}  // extern "C"
