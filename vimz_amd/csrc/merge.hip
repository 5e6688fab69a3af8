// vimz_ivc_merge*: ONE verifiable proof object out of several Nova IVC proofs of contiguous row segments (protocol: merge_internal.hpp;
// DESIGN.md §6b).  The reference's fold_input returns one RecursiveSNARK (vimz/src/nova_snark_backend/folding.rs:27-43) and
// BASELINE.json's north_star shards a proof by row segments with a "host-side sequential final fold": this is that fold for IVC
// proofs — relaxed + relaxed NIFS on both curves of the cycle, the cross terms and the folds of the witness vectors on the GPU
// (k_cross_term, the Pippenger MSM, k_fold5), the instance arithmetic and the SHA3 hash tree on the host.
#include <thread>
#include "ivc_internal.hpp"
#include "proof_io.hpp"
#include <unistd.h>
#include "merge_internal.hpp"

namespace {

template <class F> void sha_fe(Sha3& h, const F& m) { const F c = F::from_mont(m); h.update(c.v, 32); }
template <class F> void sha_pt(Sha3& h, const Affine<F>& p) { sha_fe(h, p.x); sha_fe(h, p.y); }
void sha_u256(Sha3& h, const U256w& x) { h.update(x.w, 32); }
void sha_word(Sha3& h, uint64_t x) { h.update(&x, 8); }
void chal(const uint8_t h[32], char tag, uint32_t out[4]) {
  Sha3 s; s.update(h, 32); s.update(&tag, 1);
  uint8_t o[32]; s.finish(o); memcpy(out, o, 16);
}
template <class F> F fe128(const uint32_t r[4]) { F c = F::zero(); for (int k = 0; k < 4; k++) c.v[k] = r[k]; return F::to_mont(c); }
// a + r·b on the host (r: 128 bits)
template <class FS> Affine<FS> pt_axpy(const Affine<FS>& a, const uint32_t r[4], const Affine<FS>& b) {
  XYZZ<FS> t = host_mul<FS>(b, r, 128);
  XYZZ<FS> s = from_affine(a);
  add_full(s, t);
  return to_affine(s);
}

// Leaf(j): the accumulator an IVC proof's record stands for.  r_out: the challenge its pending fresh secondary instance is folded with.
MAcc leaf_acc(const vimz_ivc* vk, const MSeg& s, uint32_t r_out[4]) {
  MAcc a;
  Sha3 h; h.update("vimz-merge-leaf-v1", 18);
  sha_fe(h, vk->c1->digest); sha_fe(h, vk->c2.digest); sha_word(h, vk->c1->len_z);
  sha_word(h, s.n);
  for (auto& z : s.zs) sha_fe(h, z);
  for (auto& z : s.ze) sha_fe(h, z);
  sha_pt(h, s.U1.W); sha_pt(h, s.U1.E); sha_fe(h, s.U1.u); sha_u256(h, s.U1.X0); sha_u256(h, s.U1.X1);
  sha_pt(h, s.U2.W); sha_pt(h, s.U2.E); sha_fe(h, s.U2.u); sha_u256(h, s.U2.X0); sha_u256(h, s.U2.X1);
  sha_pt(h, s.u2.W); sha_fe(h, s.u2.x0); sha_fe(h, s.u2.x1);
  sha_pt(h, s.T);
  h.finish(a.h);
  chal(a.h, 'q', r_out);
  a.n = s.n; a.zs = s.zs; a.ze = s.ze;
  a.P.cW = s.U1.W; a.P.cE = s.U1.E; a.P.u = cross_field<Fe>(s.U1.u); a.P.X0 = from_u256<Fe>(s.U1.X0); a.P.X1 = from_u256<Fe>(s.U1.X1);
  const Fq rq = fe128<Fq>(r_out);
  a.Q.cW = pt_axpy<Fe>(s.U2.W, r_out, s.u2.W);
  a.Q.cE = pt_axpy<Fe>(s.U2.E, r_out, s.T);
  a.Q.u = Fq::add(cross_field<Fq>(s.U2.u), rq);
  a.Q.X0 = Fq::add(from_u256<Fq>(s.U2.X0), Fq::mul(rq, cross_field<Fq>(s.u2.x0)));
  a.Q.X1 = Fq::add(from_u256<Fq>(s.U2.X1), Fq::mul(rq, cross_field<Fq>(s.u2.x1)));
  return a;
}

// Node(A, B; Tp, Tq).  The caller has checked A.ze == B.zs.
MAcc node_acc(const MAcc& A, const MAcc& B, const G1Aff& Tp, const G2Aff& Tq, uint32_t rp[4], uint32_t rq[4]) {
  MAcc a;
  Sha3 h; h.update("vimz-merge-node-v1", 18);
  h.update(A.h, 32); h.update(B.h, 32); sha_pt(h, Tp); sha_pt(h, Tq);
  h.finish(a.h);
  chal(a.h, 'p', rp); chal(a.h, 'q', rq);
  a.n = A.n + B.n; a.zs = A.zs; a.ze = B.ze;
  const Fe fp = fe128<Fe>(rp); const Fq fq = fe128<Fq>(rq);
  // E = E_A + r·(T + r·E_B): two 128-bit multiplications instead of one by r and one by the 256-bit r²
  a.P.cW = pt_axpy<Fq>(A.P.cW, rp, B.P.cW);
  a.P.cE = pt_axpy<Fq>(A.P.cE, rp, pt_axpy<Fq>(Tp, rp, B.P.cE));
  a.P.u = Fe::add(A.P.u, Fe::mul(fp, B.P.u)); a.P.X0 = Fe::add(A.P.X0, Fe::mul(fp, B.P.X0)); a.P.X1 = Fe::add(A.P.X1, Fe::mul(fp, B.P.X1));
  a.Q.cW = pt_axpy<Fe>(A.Q.cW, rq, B.Q.cW);
  a.Q.cE = pt_axpy<Fe>(A.Q.cE, rq, pt_axpy<Fe>(Tq, rq, B.Q.cE));
  a.Q.u = Fq::add(A.Q.u, Fq::mul(fq, B.Q.u)); a.Q.X0 = Fq::add(A.Q.X0, Fq::mul(fq, B.Q.X0)); a.Q.X1 = Fq::add(A.Q.X1, Fq::mul(fq, B.Q.X1));
  return a;
}

bool same_state(const std::vector<Fe>& a, const std::vector<Fe>& b) {
  if (a.size() != b.size()) return false;
  for (size_t k = 0; k < a.size(); k++) if (!a[k].eq(b[k])) return false;
  return true;
}

// ---- device side ------------------------------------------------------------------------------------------------------------------------
struct DevAcc {      // where an accumulator's vectors live
  const uint32_t *Zp, *Ep, *AZp, *BZp, *CZp, *Zq, *Eq, *AZq, *BZq, *CZq;
};

size_t merged_words(const vimz_ivc* v) {      // 32-byte elements of the one device allocation
  const size_t nw1 = v->pri->n_wires, nc1 = v->pri->n_c, nw2 = v->sec.n_w, nc2 = v->sec.n_c;
  return nw1 + 4 * nc1 + 2 * (nw2 + 4 * nc2) + nc2;
}

// (an object that is dropped on an error path gives its device and pinned memory back; the caller holds the context's lock)
// (its buffers go back to the verifier-key IVC as the spare set when that has none)
void release_buffers(vimz_ivc_merged* m) {
  vimz_ivc* vk = m->vk;
  int slot = -1;
  if (vk && m->dev && m->pin) for (int k = 0; k < vimz_ivc::MERGED_SPARES && slot < 0; k++) if (!vk->merged_spare_dev[k]) slot = k;
  if (slot >= 0) { vk->merged_spare_dev[slot] = m->dev; vk->merged_spare_pin[slot] = m->pin; }
  else {
    if (m->dev) {
      if (vk) { auto& ex = vk->ipc_exports; ex.erase(std::remove_if(ex.begin(), ex.end(), [&](const vimz_ivc::IpcExport& e) { return e.dev == m->dev; }), ex.end()); }   // (the address may come back as another allocation)
      hipFree(m->dev);
    }
    if (m->pin) hipHostFree(m->pin);
  }
  m->dev = nullptr; m->pin = nullptr;
}
void unregister_merged(vimz_ivc_merged* m) {
  if (!m->vk) return;
  auto& d = m->vk->merged_dependents;
  d.erase(std::remove(d.begin(), d.end(), m), d.end());
}
// vimz_ivc_free of a verifier key that still has merged proofs: they lose their buffers and their key (every later call fails cleanly)
void orphan_dependents(vimz_ivc* v) {
  std::lock_guard<std::mutex> g(v->ctx->mu);
  hipSetDevice(v->ctx->device);
  hipStreamSynchronize(v->ctx->stream);
  for (vimz_ivc_merged* m : v->merged_dependents) {
    if (m->dev) hipFree(m->dev);
    if (m->pin) hipHostFree(m->pin);
    m->dev = nullptr; m->pin = nullptr; m->vk = nullptr; m->broken = true;
  }
  v->merged_dependents.clear();
}
struct MergedDrop { void operator()(vimz_ivc_merged* m) const { if (!m) return; unregister_merged(m); release_buffers(m); delete m; } };
typedef std::unique_ptr<vimz_ivc_merged, MergedDrop> MergedPtr;

int merged_alloc(vimz_ivc* vk, MergedPtr& m) {
  vimz_ctx* ctx = vk->ctx;
  m.reset(new vimz_ivc_merged());
  m->vk = vk;
  vk->merged_dependents.push_back(m.get()); vk->orphan_merged = orphan_dependents;
  const size_t nw1 = vk->pri->n_wires, nc1 = vk->pri->n_c, nw2 = vk->sec.n_w, nc2 = vk->sec.n_c;
  int slot = -1;
  for (int k = 0; k < vimz_ivc::MERGED_SPARES && slot < 0; k++) if (vk->merged_spare_dev[k]) slot = k;
  if (slot >= 0) { m->dev = vk->merged_spare_dev[slot]; m->pin = vk->merged_spare_pin[slot]; vk->merged_spare_dev[slot] = nullptr; vk->merged_spare_pin[slot] = nullptr; }
  else P_TRY(hipMalloc((void**)&m->dev, 32 * merged_words(vk)));
  uint32_t* d = m->dev;
  auto take = [&](size_t n) { uint32_t* r = d; d += 8 * n; return r; };
  m->Zp = take(nw1); m->Ep = take(nc1); m->AZp = take(nc1); m->BZp = take(nc1); m->CZp = take(nc1);
  m->Zq = take(nw2); m->Eq = take(nc2); m->AZq = take(nc2); m->BZq = take(nc2); m->CZq = take(nc2);
  m->leaf_q[0] = take(nw2); for (int k = 1; k < 5; k++) m->leaf_q[k] = take(nc2);
  m->Tq = take(nc2);
  if (!m->pin) P_TRY(hipHostMalloc(&m->pin, 4 * (size_t)XYZZ_WORDS * MSM_MAX_WINDOWS));
  return VIMZ_OK;
}

// The record of an IVC proof at rest (after a fold call): statement, both running instances, the pending fresh secondary instance and
// the commitment to the cross term of (U2, u2), which the IVC computed for its own next step.
MSeg record_of(const vimz_ivc* v) {
  MSeg s;
  s.n = v->i; s.zs = v->z0; s.ze = v->pri->z_cur;
  s.U1 = v->U1; s.U2 = v->U2; s.u2 = v->u2;
  if (v->sec_T_valid) s.T = v->T2; else { s.T.x = Fe::zero(); s.T.y = Fe::zero(); }
  return s;
}

// Z, E, AZ, BZ, CZ of  U2 + r·u2  of the IVC `v` into dst[0..5)  (the secondary half of Leaf(v)), on stream s
hipError_t leaf_fold_secondary(const vimz_ivc* v, const uint32_t r[4], uint32_t* const dst[5], hipStream_t s) {
  const SecDev& S = v->sec;
  const uint32_t* run[5] = {S.Zrun, S.E, S.AZ, S.BZ, S.CZ};
  const size_t len[5] = {S.n_w, S.n_c, S.n_c, S.n_c, S.n_c};
  for (int k = 0; k < 5; k++) { const hipError_t e = hipMemcpyAsync(dst[k], run[k], 32 * len[k], hipMemcpyDeviceToDevice, s); if (e != hipSuccess) return e; }
  Fold5 f;
  f.x1[0] = dst[0]; f.x2[0] = S.z2; f.n[0] = S.n_w;
  f.x1[1] = v->sec_T_valid ? dst[1] : nullptr; f.x2[1] = S.T; f.n[1] = S.n_c;      // (after one step U2 is still the zero instance: T = 0)
  f.x1[2] = dst[2]; f.x2[2] = S.az2; f.n[2] = S.n_c;
  f.x1[3] = dst[3]; f.x2[3] = S.bz2; f.n[3] = S.n_c;
  f.x1[4] = dst[4]; f.x2[4] = S.cz2; f.n[4] = S.n_c;
  hipLaunchKernelGGL(k_fold5<Fq>, dim3(64), dim3(256), 0, s, f, fe128<Fq>(r));
  return hipGetLastError();
}

// Node(m, B) on the device and the host: m absorbs the accumulator B whose vectors sit at `bv`.  Caller holds the lock of m's context.
int node_merge(vimz_ivc_merged* m, const MAcc& B, const DevAcc& bv) {
  vimz_ivc* vk = m->vk; vimz_ctx* ctx = vk->ctx; vimz_prover* p = vk->pri;
  hipStream_t s = ctx->stream, s2 = vk->s2;
  const size_t nw1 = p->n_wires, nc1 = p->n_c, nw2 = vk->sec.n_w, nc2 = vk->sec.n_c;
  if (m->broken) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merge: this merged proof failed in the middle of an earlier merge and cannot be used");
  if (!same_state(m->acc.ze, B.zs)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merge: the incoming segment does not start at the state the merged proof ends in");
  if (m->acc.n + B.n < m->acc.n) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merge: step count overflow");
  const double t0 = now_s();
  // (from here on a failure leaves the vectors half folded: the guard marks the object unusable)
  struct Poison { vimz_ivc_merged* m; bool armed = true; ~Poison() { if (armed) m->broken = true; } } poison{m};
  // cross terms and their commitments: the primary one (the only large MSM of a merge) on the context's stream, the secondary one beside it
  hipLaunchKernelGGL(k_cross_term<Fr>, dim3(stream_grid(nc1)), dim3(256), 0, s, nc1, (const uint32_t*)m->AZp, (const uint32_t*)m->BZp, (const uint32_t*)m->CZp, m->acc.P.u,
                     bv.AZp, bv.BZp, bv.CZp, B.P.u, p->T);
  MsmPlan plan_p, plan_q;
  if (!ctx->msm_ws.host_pinned) P_TRY(hipHostMalloc(&ctx->msm_ws.host_pinned, 4 * XYZZ_WORDS * MSM_MAX_WINDOWS));
  const BaseTables tb1 = vk->ck1->tb(0);
  P_TRY(msm_launch<BnG1>(s, ctx->msm_ws, vk->ck1->d, p->T, nc1, 1, 0, ctx->msm_ws.host_pinned, &plan_p, nullptr, 0, vk->ck1->tables ? &tb1 : nullptr));
  P_TRY(hipEventRecord(vk->ev_fork, s));        // (everything B's vectors depend on is ordered before this point of s)
  P_TRY(hipStreamWaitEvent(s2, vk->ev_fork, 0));
  hipLaunchKernelGGL(k_cross_term<Fq>, dim3(stream_grid(nc2)), dim3(256), 0, s2, nc2, (const uint32_t*)m->AZq, (const uint32_t*)m->BZq, (const uint32_t*)m->CZq, m->acc.Q.u,
                     bv.AZq, bv.BZq, bv.CZq, B.Q.u, m->Tq);
  P_TRY(msm_launch<Grumpkin>(s2, vk->ws2, vk->ck2->d, m->Tq, nc2, 1, 0, m->pin, &plan_q, nullptr, 0, vk->tb_ck2.d ? &vk->tb_ck2 : nullptr));
  P_TRY(hipStreamSynchronize(s2));
  const G2Aff Tq = msm_finish<Grumpkin>(plan_q, m->pin);
  P_TRY(hipStreamSynchronize(s));
  const G1Aff Tp = msm_finish<BnG1>(plan_p, ctx->msm_ws.host_pinned);
  const double t1 = now_s();
  uint32_t rp[4], rq[4];
  // the challenges need the hash only: the folds of the vectors are queued before the host's share of the instance arithmetic
  {
    Sha3 h; h.update("vimz-merge-node-v1", 18); h.update(m->acc.h, 32); h.update(B.h, 32); sha_pt(h, Tp); sha_pt(h, Tq);
    uint8_t hh[32]; h.finish(hh); chal(hh, 'p', rp); chal(hh, 'q', rq);
  }
  const Fe fp = fe128<Fe>(rp); const Fq fq = fe128<Fq>(rq);
  {
    Fold5 f;
    f.x1[0] = m->Zp; f.x2[0] = bv.Zp; f.n[0] = nw1;
    f.x1[1] = m->Ep; f.x2[1] = p->T; f.n[1] = nc1;
    f.x1[2] = m->AZp; f.x2[2] = bv.AZp; f.n[2] = nc1;
    f.x1[3] = m->BZp; f.x2[3] = bv.BZp; f.n[3] = nc1;
    f.x1[4] = m->CZp; f.x2[4] = bv.CZp; f.n[4] = nc1;
    hipLaunchKernelGGL(k_fold5<Fr>, dim3(2048), dim3(256), 0, s, f, fp);
    hipLaunchKernelGGL(k_axpy_inplace<Fr>, dim3(stream_grid(nc1)), dim3(256), 0, s, nc1, m->Ep, Fe::sqr(fp), bv.Ep);
    Fold5 g;
    g.x1[0] = m->Zq; g.x2[0] = bv.Zq; g.n[0] = nw2;
    g.x1[1] = m->Eq; g.x2[1] = m->Tq; g.n[1] = nc2;
    g.x1[2] = m->AZq; g.x2[2] = bv.AZq; g.n[2] = nc2;
    g.x1[3] = m->BZq; g.x2[3] = bv.BZq; g.n[3] = nc2;
    g.x1[4] = m->CZq; g.x2[4] = bv.CZq; g.n[4] = nc2;
    hipLaunchKernelGGL(k_fold5<Fq>, dim3(64), dim3(256), 0, s2, g, fq);
    hipLaunchKernelGGL(k_axpy_inplace<Fq>, dim3(stream_grid(nc2)), dim3(256), 0, s2, nc2, m->Eq, Fq::sqr(fq), bv.Eq);
    P_TRY(hipGetLastError());
  }
  MOp op; op.kind = 1; op.leaf = 0; op.Tp = Tp; op.Tq = Tq;
  uint32_t rp2[4], rq2[4];
  m->acc = node_acc(m->acc, B, Tp, Tq, rp2, rq2);
  m->ops.push_back(op);
  P_TRY(hipStreamSynchronize(s2));
  P_TRY(hipStreamSynchronize(s));
  const double t2 = now_s();
  m->seconds[1] += t1 - t0; m->seconds[2] += t2 - t1; m->seconds[3] += t2 - t0;
  poison.armed = false;
  return VIMZ_OK;
}

// locks of two contexts in a fixed order (or one, when they are the same)
struct TwoLocks {
  std::unique_lock<std::mutex> a, b;
  TwoLocks(vimz_ctx* x, vimz_ctx* y) : a(x->mu, std::defer_lock), b(y->mu, std::defer_lock) { if (x == y) a.lock(); else std::lock(a, b); }
};

bool same_shapes(const vimz_ivc* a, const vimz_ivc* b) {
  return a->pri->n_wires == b->pri->n_wires && a->pri->n_c == b->pri->n_c && a->sec.n_w == b->sec.n_w && a->sec.n_c == b->sec.n_c && a->pri->len_z == b->pri->len_z &&
         a->c1->digest.eq(b->c1->digest) && a->c2.digest.eq(b->c2.digest) && a->ctx->device == b->ctx->device;
}

}  // namespace

namespace vz {

bool merged_replay(const vimz_ivc* vk, const std::vector<MSeg>& segs, const std::vector<MOp>& ops, MAcc* out, uint32_t* flags) {
  uint32_t fl = 0;
  std::vector<MAcc> st;
  std::vector<uint8_t> used(segs.size(), 0);
  size_t next_leaf = 0;
  for (auto& o : ops) {
    if (o.kind == 0) {
      // leaves appear in row order, each exactly once (the chaining check below then makes the whole a contiguous run of rows)
      if (o.leaf != next_leaf || o.leaf >= segs.size()) { *flags = fl | 8192; return false; }
      next_leaf++;
      const MSeg& s = segs[o.leaf];
      if (s.n == 0 || s.zs.size() != vk->c1->len_z || s.ze.size() != vk->c1->len_z) { *flags = fl | 8192; return false; }
      const Fe h1 = instance_hash_native<BnFr>(vk->c1->digest, s.n, s.zs, s.ze, s.U2);
      if (!h1.eq(s.u2.x0)) fl |= 1;
      const std::vector<Fq> zq = {Fq::zero()};
      const Fq h2 = instance_hash_native<BnFq>(vk->c2.digest, s.n, zq, zq, s.U1);
      if (!cross_field<Fe>(h2).eq(s.u2.x1)) fl |= 2;
      uint32_t r[4];
      st.push_back(leaf_acc(vk, s, r));
    } else if (o.kind == 1) {
      if (st.size() < 2) { *flags = fl | 8192; return false; }
      MAcc B = std::move(st.back()); st.pop_back();
      MAcc A = std::move(st.back()); st.pop_back();
      if (!same_state(A.ze, B.zs) || A.n + B.n < A.n) fl |= 4096;
      uint32_t rp[4], rq[4];
      st.push_back(node_acc(A, B, o.Tp, o.Tq, rp, rq));
    } else { *flags = fl | 8192; return false; }
  }
  if (st.size() != 1 || next_leaf != segs.size()) { *flags = fl | 8192; return false; }
  *out = std::move(st[0]);
  *flags = fl;
  return true;
}

}  // namespace vz

extern "C" {

void vimz_ivc_merged_free(vimz_ivc_merged* m) {
  if (!m) return;
  if (m->vk && m->vk->ctx) {
    std::lock_guard<std::mutex> g(m->vk->ctx->mu);
    unregister_merged(m);
    hipSetDevice(m->vk->ctx->device);
    hipStreamSynchronize(m->vk->ctx->stream);
    if (m->vk->s2) hipStreamSynchronize(m->vk->s2);
    release_buffers(m);
  }
  delete m;
}

// Leaf: the merged proof of ONE segment.  `segment` (unchanged) also supplies the shapes, keys and context: it must outlive the object.
int vimz_ivc_merged_create(vimz_ivc* v, vimz_ivc_merged** out) {
  if (!v || !out) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = v->ctx;
  if (v->i == 0) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_create: nothing has been folded");
  if (v->broken || v->pending_sec) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_create: the IVC is not at rest");
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  const double t0 = now_s();
  MergedPtr m;
  int rc = merged_alloc(v, m);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  vimz_prover* p = v->pri;
  const uint32_t* src[5] = {p->Zrun, p->E, p->AZ, p->BZ, p->CZ}; uint32_t* dst[5] = {m->Zp, m->Ep, m->AZp, m->BZp, m->CZp};
  const size_t len[5] = {p->n_wires, p->n_c, p->n_c, p->n_c, p->n_c};
  for (int k = 0; k < 5; k++) P_TRY(hipMemcpyAsync(dst[k], src[k], 32 * len[k], hipMemcpyDeviceToDevice, s));
  m->segs.push_back(record_of(v));
  uint32_t r[4];
  m->acc = leaf_acc(v, m->segs[0], r);
  uint32_t* q[5] = {m->Zq, m->Eq, m->AZq, m->BZq, m->CZq};
  P_TRY(leaf_fold_secondary(v, r, q, s));
  MOp op; op.kind = 0; op.leaf = 0; op.Tp.x = op.Tp.y = Fq::zero(); op.Tq.x = op.Tq.y = Fe::zero();
  m->ops.push_back(op);
  P_TRY(hipGetLastError());
  P_TRY(hipStreamSynchronize(s));
  m->seconds[0] += now_s() - t0; m->seconds[3] += now_s() - t0;
  *out = m.release();
  return VIMZ_OK;
}

// Node(m, Leaf(next)): the proof of the next row segment is folded into the merged proof.  `next` is read in place and left unchanged
// (it may go on folding); it must start at the state the merged proof ends in, be of the same circuits and sit on the same device.
int vimz_ivc_merge(vimz_ivc_merged* m, vimz_ivc* next) {
  if (!m || !next || !m->vk) return VIMZ_ERR_INVALID;
  vimz_ivc* vk = m->vk; vimz_ctx* ctx = vk->ctx;
  if (!same_shapes(vk, next)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merge: the segment is of other circuits or on another device (export it and use vimz_ivc_merged_load there)");
  if (next->i == 0 || next->broken || next->pending_sec) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merge: the incoming IVC has folded nothing or is not at rest");
  TwoLocks lk(ctx, next->ctx);
  P_TRY(hipSetDevice(ctx->device));
  const double t0 = now_s();
  MSeg rec = record_of(next);
  if (!same_state(m->acc.ze, rec.zs)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merge: the incoming segment does not start at the state the merged proof ends in");
  uint32_t r[4];
  if (m->broken) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merge: this merged proof failed in the middle of an earlier merge and cannot be used");
  const MAcc B = leaf_acc(vk, rec, r);
  P_TRY(leaf_fold_secondary(next, r, m->leaf_q, ctx->stream));
  vimz_prover* np = next->pri;
  const DevAcc bv{np->Zrun, np->E, np->AZ, np->BZ, np->CZ, m->leaf_q[0], m->leaf_q[1], m->leaf_q[2], m->leaf_q[3], m->leaf_q[4]};
  m->seconds[0] += now_s() - t0;
  const size_t ops_before = m->ops.size();
  MOp op; op.kind = 0; op.leaf = (uint32_t)m->segs.size(); op.Tp.x = op.Tp.y = Fq::zero(); op.Tq.x = op.Tq.y = Fe::zero();
  m->segs.push_back(std::move(rec)); m->ops.push_back(op);
  const int rc = node_merge(m, B, bv);
  if (rc) { m->segs.pop_back(); m->ops.resize(ops_before); }
  return rc;
}

// Node(m, other): two merged proofs (of adjacent runs of segments) become one.  `other` is left unchanged.
int vimz_ivc_merge_merged(vimz_ivc_merged* m, vimz_ivc_merged* other) {
  if (!m || !other || m == other || !m->vk || !other->vk) return VIMZ_ERR_INVALID;
  vimz_ivc* vk = m->vk; vimz_ctx* ctx = vk->ctx;
  if (!same_shapes(vk, other->vk)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merge_merged: the proofs are of other circuits or on another device");
  TwoLocks lk(ctx, other->vk->ctx);
  P_TRY(hipSetDevice(ctx->device));
  const DevAcc bv{other->Zp, other->Ep, other->AZp, other->BZp, other->CZp, other->Zq, other->Eq, other->AZq, other->BZq, other->CZq};
  const size_t segs_before = m->segs.size(), ops_before = m->ops.size();
  for (auto& s : other->segs) m->segs.push_back(s);
  for (auto o : other->ops) { if (o.kind == 0) o.leaf += (uint32_t)segs_before; m->ops.push_back(o); }
  const int rc = node_merge(m, other->acc, bv);
  if (rc) { m->segs.resize(segs_before); m->ops.resize(ops_before); }
  return rc;
}

// info: steps, segments, ops, len_z, primary wires, primary constraints, secondary wires, secondary constraints
int vimz_ivc_merged_info(const vimz_ivc_merged* m, uint64_t info[8]) {
  if (!m || !info || !m->vk) return VIMZ_ERR_INVALID;
  const vimz_ivc* vk = m->vk;
  std::lock_guard<std::mutex> g(vk->ctx->mu);
  info[0] = m->acc.n; info[1] = m->segs.size(); info[2] = m->ops.size(); info[3] = vk->pri->len_z;
  info[4] = vk->pri->n_wires; info[5] = vk->pri->n_c; info[6] = vk->sec.n_w; info[7] = vk->sec.n_c;
  return VIMZ_OK;
}
int vimz_ivc_merged_state(const vimz_ivc_merged* m, uint64_t* z_start, uint64_t* z_end, uint64_t* steps) {
  if (!m || !m->vk) return VIMZ_ERR_INVALID;
  std::lock_guard<std::mutex> g(m->vk->ctx->mu);
  const uint32_t lz = m->vk->pri->len_z;
  if (z_start) for (uint32_t k = 0; k < lz; k++) fe_to_canon(m->acc.zs[k], z_start + 4 * k);
  if (z_end) for (uint32_t k = 0; k < lz; k++) fe_to_canon(m->acc.ze[k], z_end + 4 * k);
  if (steps) *steps = m->acc.n;
  return VIMZ_OK;
}
int vimz_ivc_merged_profile(const vimz_ivc_merged* m, double seconds[4]) {
  if (!m || !seconds) return VIMZ_ERR_INVALID;
  memcpy(seconds, m->seconds, sizeof(m->seconds));
  return VIMZ_OK;
}

// The statement part of the proof — header, segment records, ops (with the cross-term commitments) — as canonical little-endian words:
// what a verifier replays.  Returns the size in bytes (copies when buf is large enough).
int64_t vimz_ivc_merged_records(const vimz_ivc_merged* m, void* buf, size_t cap) {
  if (!m || !m->vk) return VIMZ_ERR_INVALID;
  std::lock_guard<std::mutex> g(m->vk->ctx->mu);        // (a merge on another thread grows the records)
  const size_t bytes = 8 * records_words(m);
  if (buf && cap >= bytes) { Writer w; write_records(m, w); if (8 * w.w.size() != bytes) return VIMZ_ERR_INVALID; memcpy(buf, w.w.data(), bytes); }
  return (int64_t)bytes;
}

// side 0 / 1; what = VIMZ_IX_RUNNING_Z, VIMZ_IX_RUNNING_E (canonical vectors), VIMZ_IX_INSTANCE (comm_W, comm_E, u, X0, X1: 7 canonical elements)
int64_t vimz_ivc_merged_export(vimz_ivc_merged* m, int side, int what, void* buf, size_t cap) {
  if (!m || !m->vk || (side != 0 && side != 1)) return VIMZ_ERR_INVALID;
  vimz_ivc* vk = m->vk; vimz_ctx* ctx = vk->ctx;
  if (what == VIMZ_IX_INSTANCE) {
    Writer w;
    if (side == 0) { w.point(m->acc.P.cW); w.point(m->acc.P.cE); w.fe(m->acc.P.u); w.fe(m->acc.P.X0); w.fe(m->acc.P.X1); }
    else { w.point(m->acc.Q.cW); w.point(m->acc.Q.cE); w.fe(m->acc.Q.u); w.fe(m->acc.Q.X0); w.fe(m->acc.Q.X1); }
    if (buf && cap >= 8 * w.w.size()) memcpy(buf, w.w.data(), 8 * w.w.size());
    return (int64_t)(8 * w.w.size());
  }
  const uint32_t* src = nullptr; size_t n = 0;
  if (what == VIMZ_IX_RUNNING_Z) { src = side == 0 ? m->Zp : m->Zq; n = side == 0 ? vk->pri->n_wires : vk->sec.n_w; }
  else if (what == VIMZ_IX_RUNNING_E) { src = side == 0 ? m->Ep : m->Eq; n = side == 0 ? vk->pri->n_c : vk->sec.n_c; }
  else return VIMZ_ERR_INVALID;
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

// The proof as bytes (for a verifier or a merging rank in another process): records ‖ Zp ‖ Ep ‖ Zq ‖ Eq (vectors as Montgomery limbs).
size_t vimz_ivc_merged_size(const vimz_ivc_merged* m) {
  if (!m || !m->vk) return 0;
  const vimz_ivc* vk = m->vk;
  return 8 * records_words(m) + 32 * ((size_t)vk->pri->n_wires + vk->pri->n_c + vk->sec.n_w + vk->sec.n_c);
}
int vimz_ivc_merged_save(vimz_ivc_merged* m, uint8_t* blob, size_t cap) {
  if (!m || !m->vk) return VIMZ_ERR_INVALID;
  if (!blob || cap < vimz_ivc_merged_size(m)) return vz_fail(m->vk->ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_save: buffer too small");
  vimz_ivc* vk = m->vk; vimz_ctx* ctx = vk->ctx;
  Writer w; write_records(m, w);
  memcpy(blob, w.w.data(), 8 * w.w.size());
  uint8_t* o = blob + 8 * w.w.size();
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const uint32_t* src[4] = {m->Zp, m->Ep, m->Zq, m->Eq};
  const size_t len[4] = {vk->pri->n_wires, vk->pri->n_c, vk->sec.n_w, vk->sec.n_c};
  for (int k = 0; k < 4; k++) { P_TRY(hipMemcpyAsync(o, src[k], 32 * len[k], hipMemcpyDeviceToHost, s)); o += 32 * len[k]; }
  P_TRY(hipStreamSynchronize(s));
  return VIMZ_OK;
}

// vk: an IVC created for the same step circuit and keys (its folding state is neither read nor changed; it must outlive the object).
// The blob is untrusted: every element is range-checked, every point must be on its curve, the instances are RECOMPUTED from the
// records; whether the proof is valid is vimz_ivc_merged_verify's business.
int vimz_ivc_merged_load(vimz_ivc* vk, const uint8_t* blob, size_t len, vimz_ivc_merged** out) {
  if (!vk || !blob || !out || (len & 7) || len < 64) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = vk->ctx; vimz_prover* p = vk->pri;
  std::vector<uint64_t> words(len / 8); memcpy(words.data(), blob, len);
  Reader in{words.data(), words.size()};
  std::vector<MSeg> segs; std::vector<MOp> ops;
  if (!read_records(in, vk, segs, ops)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_load: the blob does not match this verifier key's circuits, or is malformed (an element not below its modulus, a point off its curve)");
  const size_t nw1 = p->n_wires, nc1 = p->n_c, nw2 = vk->sec.n_w, nc2 = vk->sec.n_c;
  const size_t vec_bytes = 32 * ((size_t)nw1 + nc1 + nw2 + nc2);
  if (!in.ok || 8 * in.pos + vec_bytes != len) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_load: malformed blob (length, an element not below its modulus, or a point off its curve)");
  MAcc acc; uint32_t fl = 0;
  if (!merged_replay(vk, segs, ops, &acc, &fl)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_load: malformed op sequence");
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  MergedPtr m;
  int rc = merged_alloc(vk, m);
  auto drop = [] {};        // (MergedDrop releases the buffers on every early return)
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const uint8_t* o = blob + 8 * in.pos;
  uint32_t* dst[4] = {m->Zp, m->Ep, m->Zq, m->Eq}; const size_t ln[4] = {nw1, nc1, nw2, nc2};
  hipError_t e = hipSuccess;
  for (int k = 0; k < 4 && e == hipSuccess; k++) { e = hipMemcpyAsync(dst[k], o, 32 * ln[k], hipMemcpyHostToDevice, s); o += 32 * ln[k]; }
  uint32_t nbad = 1;
  if (e == hipSuccess) {
    const uint32_t zero2[2] = {0, 0};
    e = hipMemcpyAsync(p->bad_d, zero2, 8, hipMemcpyHostToDevice, s);
    hipLaunchKernelGGL(k_count_unreduced<Fr>, dim3(stream_grid(nw1)), dim3(256), 0, s, (size_t)nw1, (const uint32_t*)m->Zp, p->bad_d);
    hipLaunchKernelGGL(k_count_unreduced<Fr>, dim3(stream_grid(nc1)), dim3(256), 0, s, (size_t)nc1, (const uint32_t*)m->Ep, p->bad_d);
    hipLaunchKernelGGL(k_count_unreduced<Fq>, dim3(stream_grid(nw2)), dim3(256), 0, s, (size_t)nw2, (const uint32_t*)m->Zq, p->bad_d);
    hipLaunchKernelGGL(k_count_unreduced<Fq>, dim3(stream_grid(nc2)), dim3(256), 0, s, (size_t)nc2, (const uint32_t*)m->Eq, p->bad_d);
    if (e == hipSuccess) e = hipMemcpyAsync(&nbad, p->bad_d, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  if (e != hipSuccess) { drop(); return vz_fail(ctx, VIMZ_ERR_HIP, "vimz_ivc_merged_load: staging", e); }
  if (nbad) { drop(); return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_load: a vector element of the blob is not below its modulus"); }
  // the running products (needed only to merge further) are recomputed
  launch_spmv(p, s, m->Zp, m->AZp, m->BZp, m->CZp, 0);
  sec_spmv<Fq>(vk->sec, s, m->Zq, m->AZq, m->BZq, m->CZq);
  e = hipGetLastError(); if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) { drop(); return vz_fail(ctx, VIMZ_ERR_HIP, "vimz_ivc_merged_load: products", e); }
  m->segs = std::move(segs); m->ops = std::move(ops); m->acc = std::move(acc);
  *out = m.release();
  return VIMZ_OK;
}

// ---- a merged proof handed to another process of the node WITHOUT a host round trip (the ranks' final fold as a tree) ----
// The ticket a rank publishes: {magic, elements, 0, 0} ‖ the HIP IPC handle of the object's one device allocation (64 bytes) ‖ records.
// The receiving rank maps the allocation (dmabuf IPC; between GPUs the copy below runs over xGMI), copies the folded witnesses AND the
// running products device-to-device into an object of its own and unmaps it: no save -> /dev/shm -> load, no SpMV to recompute the
// products.  The sender keeps its object alive until the receiver says it is done.
static const uint64_t SHARE_MAGIC = 0x314853475a56ull;      // "VZGSH1"
static size_t shared_elements(const vimz_ivc* vk) { return (size_t)vk->pri->n_wires + 4 * (size_t)vk->pri->n_c + vk->sec.n_w + 4 * (size_t)vk->sec.n_c; }

int64_t vimz_ivc_merged_share(vimz_ivc_merged* m, void* buf, size_t cap) {
  if (!m || !m->vk || !m->dev) return VIMZ_ERR_INVALID;
  vimz_ivc* vk = m->vk; vimz_ctx* ctx = vk->ctx;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "ticket layout");
  std::lock_guard<std::mutex> g(ctx->mu);
  if (m->broken) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_share: this merged proof failed in the middle of a merge");
  const size_t bytes = 8 * (4 + 8 + records_words(m));
  if (!buf || cap < bytes) return (int64_t)bytes;
  if (hipSetDevice(ctx->device) != hipSuccess) return VIMZ_ERR_HIP;
  // everything that wrote the vectors has been waited for by the call that queued it (merged_create / node_merge / load end in a
  // synchronise); the other process reads them through its own queue
  hipIpcMemHandle_t h;
  const vimz_ivc::IpcExport* known = nullptr;
  for (auto& ex : vk->ipc_exports) if (ex.dev == m->dev) known = &ex;
  uint64_t gen = 0;
  if (known) { memcpy(&h, known->handle, 64); gen = known->gen; }
  else {
    const hipError_t e = hipIpcGetMemHandle(&h, m->dev);
    if (e != hipSuccess) return vz_fail(ctx, VIMZ_ERR_HIP, "vimz_ivc_merged_share: hipIpcGetMemHandle", e);
    static std::atomic<uint64_t> next_gen{1};
    gen = ((uint64_t)getpid() << 32) | (next_gen.fetch_add(1) & 0xffffffffull);
    vimz_ivc::IpcExport ex; ex.dev = m->dev; memcpy(ex.handle, &h, 64); ex.gen = gen; vk->ipc_exports.push_back(ex);
  }
  Writer w; w.word(SHARE_MAGIC); w.word(shared_elements(vk)); w.word(gen); w.word(0);
  uint64_t hw[8]; memcpy(hw, &h, 64); for (int k = 0; k < 8; k++) w.word(hw[k]);
  write_records(m, w);
  if (8 * w.w.size() != bytes) return VIMZ_ERR_INVALID;
  memcpy(buf, w.w.data(), bytes);
  return (int64_t)bytes;
}

// vk as for vimz_ivc_merged_load.  The records are parsed and replayed like an untrusted blob's (range and curve checks, instances
// recomputed); the vectors are range-checked on the device; the running products are TAKEN, not recomputed — they only serve further
// merges, and vimz_ivc_merged_verify recomputes them (result bit 11) like everything else.
int vimz_ivc_merged_open_shared(vimz_ivc* vk, const uint8_t* ticket, size_t len, vimz_ivc_merged** out) {
  if (!vk || !ticket || !out || (len & 7) || len < 8 * (4 + 8 + 8)) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = vk->ctx; vimz_prover* p = vk->pri;
  std::vector<uint64_t> words(len / 8); memcpy(words.data(), ticket, len);
  if (words[0] != SHARE_MAGIC || words[1] != shared_elements(vk)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_open_shared: not a ticket for this verifier key's circuits");
  hipIpcMemHandle_t h; memcpy(&h, words.data() + 4, 64);
  Reader in{words.data() + 12, words.size() - 12};
  std::vector<MSeg> segs; std::vector<MOp> ops;
  if (!read_records(in, vk, segs, ops) || !in.ok || in.pos != words.size() - 12)
    return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_open_shared: malformed records (length, an element not below its modulus, or a point off its curve)");
  MAcc acc; uint32_t fl = 0;
  if (!merged_replay(vk, segs, ops, &acc, &fl)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_open_shared: malformed op sequence");
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  MergedPtr m;
  int rc = merged_alloc(vk, m);
  if (rc) return rc;
  void* remote = nullptr;
  hipError_t e = hipSuccess;
  const uint64_t gen = words[2];
  { auto& mps = vk->ipc_mappings;
    for (size_t k = 0; k < mps.size(); k++) if (!memcmp(mps[k].handle, &h, 64)) {
      if (mps[k].gen == gen) remote = mps[k].ptr;
      else { hipIpcCloseMemHandle(mps[k].ptr); mps.erase(mps.begin() + k); }      // same handle bytes, another allocation: the cached mapping is stale
      break;
    } }
  if (!remote) {
    e = hipIpcOpenMemHandle(&remote, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess || !remote) return vz_fail(ctx, VIMZ_ERR_HIP, "vimz_ivc_merged_open_shared: hipIpcOpenMemHandle (not the same node, or no peer access between the two GPUs)", e);
    if (vk->ipc_mappings.size() >= vimz_ivc::IPC_MAPPINGS) { hipIpcCloseMemHandle(vk->ipc_mappings.front().ptr); vk->ipc_mappings.erase(vk->ipc_mappings.begin()); }
    vimz_ivc::IpcMapping mp; memcpy(mp.handle, &h, 64); mp.ptr = remote; mp.gen = gen; vk->ipc_mappings.push_back(mp);
  }
  hipStream_t s = ctx->stream;
  const size_t nw1 = p->n_wires, nc1 = p->n_c, nw2 = vk->sec.n_w, nc2 = vk->sec.n_c;
  e = hipMemcpyAsync(m->dev, remote, 32 * shared_elements(vk), hipMemcpyDefault, s);
  uint32_t nbad = 1;
  if (e == hipSuccess) {
    const uint32_t zero2[2] = {0, 0};
    e = hipMemcpyAsync(p->bad_d, zero2, 8, hipMemcpyHostToDevice, s);
    // (the layout of merged_alloc: Zp | Ep AZp BZp CZp | Zq | Eq AZq BZq CZq)
    hipLaunchKernelGGL(k_count_unreduced<Fr>, dim3(stream_grid(nw1 + 4 * nc1)), dim3(256), 0, s, (size_t)(nw1 + 4 * nc1), (const uint32_t*)m->Zp, p->bad_d);
    hipLaunchKernelGGL(k_count_unreduced<Fq>, dim3(stream_grid(nw2 + 4 * nc2)), dim3(256), 0, s, (size_t)(nw2 + 4 * nc2), (const uint32_t*)m->Zq, p->bad_d);
    if (e == hipSuccess) e = hipMemcpyAsync(&nbad, p->bad_d, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  if (e != hipSuccess) {      // (a mapping that failed once is not kept)
    auto& mps = vk->ipc_mappings;
    for (size_t k = 0; k < mps.size(); k++) if (mps[k].ptr == remote) { hipIpcCloseMemHandle(remote); mps.erase(mps.begin() + k); break; }
    return vz_fail(ctx, VIMZ_ERR_HIP, "vimz_ivc_merged_open_shared: copy from the other process's allocation", e);
  }
  if (nbad) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_open_shared: a vector element is not below its modulus");
  m->segs = std::move(segs); m->ops = std::move(ops); m->acc = std::move(acc);
  *out = m.release();
  return VIMZ_OK;
}

// The verifier of a merged proof: RecursiveSNARK::verify(pp, num_steps, z0, ...) (reached from folding.rs:53-55) for the object
// S segments were merged into.  The instances are RECOMPUTED from the records (hash checks of every segment, adjacency, the fold
// tree) — nothing the prover says about them is used — and then ONE primary and ONE secondary relaxed instance are checked against
// the witnesses.  result: 0 = accepted; bit 0 / 1 a segment's primary / secondary chain hash; bit 2 primary relation; bit 3 / 4 primary
// comm_W / comm_E; bit 5 secondary relation; bit 6 / 7 secondary comm_W / comm_E; bit 10 the public entries of a witness vector differ
// from the instance; bit 11 the kept running products differ from the recomputed ones (bookkeeping for further merges, not part of
// the proof); bit 12 the statement: total steps, initial state, or segments not adjacent; bit 13 malformed.
int vimz_ivc_merged_verify(vimz_ivc_merged* m, uint64_t num_steps, const uint64_t* z0, uint32_t* result) {
  if (!m || !z0 || !result) return VIMZ_ERR_INVALID;
  if (!m->vk) { *result = 8192; return VIMZ_OK; }      // its verifier key was freed: nothing left to check against
  vimz_ivc* vk = m->vk; vimz_ctx* ctx = vk->ctx; vimz_prover* p = vk->pri;
  const SecDev& S = vk->sec;
  uint32_t res = 0;
  if (m->broken) { *result = 8192; return VIMZ_OK; }
  MAcc R;
  if (!merged_replay(vk, m->segs, m->ops, &R, &res)) { *result = res | 8192; return VIMZ_OK; }
  if (R.n != num_steps) res |= 4096;
  for (uint32_t k = 0; k < p->len_z; k++) {
    Fe c; memcpy(c.v, z0 + 4 * k, 32);
    if (!c.is_reduced()) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_verify: z0 element not below the modulus");
    if (!Fe::to_mont(c).eq(R.zs[k])) res |= 4096;
  }
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const uint32_t init[2] = {0, 0xffffffffu};
  uint32_t bad[2];
  uint64_t pt[8];
  int rc;
  // primary
  launch_spmv(p, s, m->Zp, p->az2, p->bz2, p->cz2, 0);
  P_TRY(hipMemcpyAsync(p->bad_d, init, 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_check_relaxed<Fr>, dim3(stream_grid(p->n_c)), dim3(256), 0, s, (size_t)p->n_c, (const uint32_t*)p->az2, (const uint32_t*)p->bz2, (const uint32_t*)p->cz2, R.P.u, (const uint32_t*)m->Ep, p->bad_d);
  P_TRY(hipMemcpyAsync(bad, p->bad_d, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 4;
  P_TRY(hipMemcpyAsync(p->bad_d, init, 8, hipMemcpyHostToDevice, s));
  { const uint32_t* kept[3] = {m->AZp, m->BZp, m->CZp}; const uint32_t* fresh[3] = {p->az2, p->bz2, p->cz2};
    for (int q = 0; q < 3; q++) hipLaunchKernelGGL(k_count_diff, dim3(stream_grid(p->n_c)), dim3(256), 0, s, (size_t)p->n_c, kept[q], fresh[q], p->bad_d); }
  P_TRY(hipMemcpyAsync(bad, p->bad_d, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 2048;
  if ((rc = vz_msm_device(ctx, vk->ck1, 0, m->Zp + 8, p->n_wires - 3, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (memcmp(pt, R.P.cW.x.v, 32) || memcmp(pt + 4, R.P.cW.y.v, 32)) res |= 8;
  if ((rc = vz_msm_device(ctx, vk->ck1, 0, m->Ep, p->n_c, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (memcmp(pt, R.P.cE.x.v, 32) || memcmp(pt + 4, R.P.cE.y.v, 32)) res |= 16;
  {
    Fe e[3];
    P_TRY(hipMemcpyAsync(&e[0], m->Zp, 32, hipMemcpyDeviceToHost, s));
    P_TRY(hipMemcpyAsync(&e[1], m->Zp + 8 * (size_t)(p->n_wires - 2), 64, hipMemcpyDeviceToHost, s));
    P_TRY(hipStreamSynchronize(s));
    if (!e[0].eq(R.P.u) || !e[1].eq(R.P.X0) || !e[2].eq(R.P.X1)) res |= 1024;
  }
  // secondary (scratch: the leaf buffers)
  uint32_t *az = m->leaf_q[2], *bz = m->leaf_q[3], *cz = m->leaf_q[4];
  sec_spmv<Fq>(S, s, m->Zq, az, bz, cz);
  P_TRY(hipMemcpyAsync(S.bad, init, 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_check_relaxed<Fq>, dim3(stream_grid(S.n_c)), dim3(256), 0, s, (size_t)S.n_c, (const uint32_t*)az, (const uint32_t*)bz, (const uint32_t*)cz, R.Q.u, (const uint32_t*)m->Eq, S.bad);
  P_TRY(hipMemcpyAsync(bad, S.bad, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 32;
  P_TRY(hipMemcpyAsync(S.bad, init, 8, hipMemcpyHostToDevice, s));
  { const uint32_t* kept[3] = {m->AZq, m->BZq, m->CZq}; const uint32_t* fresh[3] = {az, bz, cz};
    for (int q = 0; q < 3; q++) hipLaunchKernelGGL(k_count_diff, dim3(stream_grid(S.n_c)), dim3(256), 0, s, (size_t)S.n_c, kept[q], fresh[q], S.bad); }
  P_TRY(hipMemcpyAsync(bad, S.bad, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 2048;
  if ((rc = vz_msm_device(ctx, vk->ck2, 0, m->Zq + 8, S.n_w - 3, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (memcmp(pt, R.Q.cW.x.v, 32) || memcmp(pt + 4, R.Q.cW.y.v, 32)) res |= 64;
  if ((rc = vz_msm_device(ctx, vk->ck2, 0, m->Eq, S.n_c, 1, 0, pt, VIMZ_FORM_MONTGOMERY))) return rc;
  if (memcmp(pt, R.Q.cE.x.v, 32) || memcmp(pt + 4, R.Q.cE.y.v, 32)) res |= 128;
  {
    Fq e[3];
    P_TRY(hipMemcpyAsync(&e[0], m->Zq, 32, hipMemcpyDeviceToHost, s));
    P_TRY(hipMemcpyAsync(&e[1], m->Zq + 8 * (size_t)(S.n_w - 2), 64, hipMemcpyDeviceToHost, s));
    P_TRY(hipStreamSynchronize(s));
    if (!e[0].eq(R.Q.u) || !e[1].eq(R.Q.X0) || !e[2].eq(R.Q.X1)) res |= 1024;
  }
  *result = res;
  return VIMZ_OK;
}

// fold_input in ONE call (vimz/src/nova_snark_backend/folding.rs:27-43): the rows are proven as n_seg contiguous segments — segment k
// by segs[k], IVCs of the same circuits on their own contexts of one device, reset by this call —, folded concurrently from a thread
// each and merged into ONE object.  Segment k starts at the state segment k−1 ends in: the row digests of all but the last segment
// are computed at once (each on its successor's context) while segment 0 already folds; only the short host chains are serial.
// seconds (optional) = {waiting for start states, merge, total}.  Segments that get no rows (nsteps < n_seg) are left out.
int vimz_ivc_fold_segments(vimz_ivc* const* segs, size_t n_seg, const uint64_t* z0, const uint64_t* step_inputs, size_t nsteps, vimz_ivc_merged** out, double seconds[3]) {
  return vimz_ivc_fold_segments_dg(segs, n_seg, z0, step_inputs, nsteps, nullptr, out, seconds);
}
// the same when the caller already holds the rows' digests (vimz_ivc_row_digests over exactly these rows: a rank of a sharded proof
// has just exchanged them with the other ranks) — they are not computed a second time
int vimz_ivc_fold_segments_dg(vimz_ivc* const* segs, size_t n_seg, const uint64_t* z0, const uint64_t* step_inputs, size_t nsteps, const uint64_t* digests,
                              vimz_ivc_merged** out, double seconds[3]) {
  if (!segs || !n_seg || !z0 || !out || !step_inputs || !nsteps) return VIMZ_ERR_INVALID;
  for (size_t k = 0; k < n_seg; k++) if (!segs[k]) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = segs[0]->ctx;
  for (size_t k = 1; k < n_seg; k++) {
    if (!same_shapes(segs[0], segs[k])) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_fold_segments: the segments' IVCs must be of the same circuits and on the same device");
    for (size_t j = 0; j < k; j++) if (segs[j]->ctx == segs[k]->ctx) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_fold_segments: every segment needs a context of its own");
  }
  const double t_all = now_s();
  const size_t S = std::min(n_seg, nsteps), base = nsteps / S, rem = nsteps % S;
  std::vector<size_t> lo(S), hi(S);
  for (size_t k = 0, at = 0; k < S; k++) { lo[k] = at; at += base + (k < rem ? 1 : 0); hi[k] = at; }
  const size_t n_priv = segs[0]->pri->n_priv, lz = segs[0]->pri->len_z;
  const size_t stride = vimz_ivc_digest_stride(segs[0]);
  std::vector<int> rc_fold(S, VIMZ_OK), rc_dig(S, VIMZ_OK);
  std::vector<std::vector<uint64_t>> dig(S);
  std::vector<std::thread> th_dig, th_fold;
  bool pre_mode = stride && S > 1 && getenv("VIMZ_DEBUG_NO_HEAD_PRECOMPUTE") == nullptr;      // (also with the caller's digests: the folds then start a pool round apart instead of together after three)
  const bool too_long_for_head = nsteps > HEAD_JOB_MAX;
  for (size_t k = 0; k < S; k++) segs[k]->pri->suppress_head = too_long_for_head;      // (cleared after the folds)
  if (too_long_for_head) pre_mode = false;      // (the segments share ONE host pool: past four rounds of it the GPU's chain pass is the shorter wait — profiles/r05_head_crossover.txt)
  for (size_t k = 0; k < S && pre_mode; k++) pre_mode = head_takes_whole_call(segs[k]->pri, hi[k] - lo[k]) && !segs[k]->broken;
  if (stride && S > 1 && !digests && !pre_mode && getenv("VIMZ_DEBUG_NO_DEFERRED_START") != nullptr)
    for (size_t k = 0; k + 1 < S; k++) {
      dig[k].resize(4 * stride * (hi[k] - lo[k]));
      th_dig.emplace_back([&, k] { rc_dig[k] = vimz_ivc_row_digests(segs[k + 1], step_inputs + 4 * n_priv * lo[k], hi[k] - lo[k], dig[k].data()); });
    }
  std::vector<uint64_t> z(z0, z0 + 4 * lz), zs;
  double t_chain = 0;
  int rc = VIMZ_OK;
  size_t started = 0;
  // Short calls whose segments fit whole into the host-evaluated head batch (the driver's 20-row window: 7 rows per segment): each
  // segment's row-hash chains are evaluated ONCE, with their wires, into its head staging right before its fold is started — their
  // outputs are the digests the NEXT segment's start state needs, and the fold finds its head rows hashed.  In segment order, so the GPU
  // gets its first fold after one round of the host pool (14 tasks) and the later segments follow a round apart; before, the rows of
  // all but the last segment were hashed twice (70 pool tasks in front of the last segment's first fold instead of 42).
  const bool pre = pre_mode;
  auto drop_pre = [&] { for (size_t k = 0; k < S; k++) { std::lock_guard<std::mutex> g(segs[k]->ctx->mu); segs[k]->pri->pre_rows = 0; segs[k]->pri->pre_inputs = nullptr; } };   // (on an early return: no later call may take these over)
  std::vector<uint64_t> dig_prev;      // the previous segment's digests, copied out of its staging before its fold may touch it
  for (size_t k = 0; k < S && !rc && pre; k++) {
    if (k > 0) {
      const double t0 = now_s();
      const size_t n = hi[k - 1] - lo[k - 1];
      zs.assign(4 * lz * (n + 1), 0);
      rc = vimz_ivc_chain_from_digests(segs[k], z.data(), step_inputs + 4 * n_priv * lo[k - 1], digests ? digests + 4 * stride * lo[k - 1] : dig_prev.data(), n, zs.data());
      if (rc) { if (segs[k]->ctx != ctx) ctx->err = segs[k]->ctx->err; break; }
      z.assign(zs.end() - 4 * lz, zs.end());
      t_chain += now_s() - t0;
    }
    {
      vimz_ctx* c = segs[k]->ctx;
      std::lock_guard<std::mutex> g(c->mu);
      if (hipSetDevice(c->device) != hipSuccess) { rc = vz_fail(c, VIMZ_ERR_HIP, "vimz_ivc_fold_segments: hipSetDevice"); if (c != ctx) ctx->err = c->err; break; }
      rc = head_precompute(segs[k]->pri, step_inputs + 4 * n_priv * lo[k], hi[k] - lo[k]);
      if (rc) { if (c != ctx) ctx->err = c->err; break; }
      if (k + 1 < S && !digests) { const uint64_t* jv = reinterpret_cast<const uint64_t*>(segs[k]->pri->jobvals_host); dig_prev.assign(jv, jv + 4 * stride * (hi[k] - lo[k])); }
    }
    if ((rc = vimz_ivc_reset(segs[k], z.data()))) break;
    th_fold.emplace_back([&, k] { rc_fold[k] = vimz_ivc_fold(segs[k], step_inputs + 4 * n_priv * lo[k], hi[k] - lo[k]); });
    started = k + 1;
  }
  // GPU-evaluated row hashes, every row hashed ONCE: all segments' fold calls begin now — inputs uploaded, the row-hash chains of their first
  // batches running side by side, with their wires — and segment k takes its start state from segment k-1's host state chain the moment that
  // exists (vimz_prover::start_from / end_to), one chain latency after the start.  (Until round 5 segment k first waited for a hash-only pass over
  // segment k-1's rows on its own context and then hashed its own rows: two chain latencies in front of the later segments' first folds.)
  const bool deferred = !pre && stride && S > 1 && !digests && getenv("VIMZ_DEBUG_NO_DEFERRED_START") == nullptr;
  std::vector<std::unique_ptr<StartLink>> links;
  if (deferred) {
    for (size_t k = 0; k + 1 < S; k++) links.emplace_back(new StartLink());
    for (size_t k = 0; k < S && !rc; k++) {
      if ((rc = vimz_ivc_reset(segs[k], z.data()))) break;      // (segments after the first: a placeholder, replaced when their start state arrives)
      vimz_prover* pk = segs[k]->pri; vimz_ivc* vk = segs[k];
      pk->start_from = k > 0 ? links[k - 1].get() : nullptr; pk->end_to = k + 1 < S ? links[k].get() : nullptr;
      pk->on_start = [vk] { for (uint32_t q = 0; q < vk->c1->len_z; q++) vk->z0[q] = vk->pri->z_cur[q]; };
    }
    // (every segment's rows on its device before any segment's call starts: prover_internal.hpp, vz_prover_preload_inputs)
    static const bool no_preload = getenv("VIMZ_DEBUG_NO_PRELOAD") != nullptr;
    for (size_t k = 0; k < S && !rc && !no_preload; k++) rc = vz_prover_preload_inputs(segs[k]->pri, step_inputs + 4 * n_priv * lo[k], hi[k] - lo[k]);
    for (size_t k = 0; k < S && !rc; k++) {
      th_fold.emplace_back([&, k] { rc_fold[k] = vimz_ivc_fold(segs[k], step_inputs + 4 * n_priv * lo[k], hi[k] - lo[k]); if (rc_fold[k] && k + 1 < S) links[k]->fail(); });
      started = k + 1;
    }
  }
  for (size_t k = 0; k < S && !rc && !pre && !deferred; k++) {
    if (k > 0) {
      const double t0 = now_s();
      const size_t n = hi[k - 1] - lo[k - 1];
      zs.assign(4 * lz * (n + 1), 0);
      if (stride && digests) rc = vimz_ivc_chain_from_digests(segs[k], z.data(), step_inputs + 4 * n_priv * lo[k - 1], digests + 4 * stride * lo[k - 1], n, zs.data());
      else if (stride) {
        th_dig[k - 1].join();
        rc = rc_dig[k - 1];
        if (!rc) rc = vimz_ivc_chain_from_digests(segs[k], z.data(), step_inputs + 4 * n_priv * lo[k - 1], dig[k - 1].data(), n, zs.data());
      } else rc = vimz_ivc_state_chain(segs[k], z.data(), step_inputs + 4 * n_priv * lo[k - 1], n, zs.data());
      if (rc) { if (segs[k]->ctx != ctx) ctx->err = segs[k]->ctx->err; break; }
      z.assign(zs.end() - 4 * lz, zs.end());
      t_chain += now_s() - t0;
    }
    if ((rc = vimz_ivc_reset(segs[k], z.data()))) break;
    th_fold.emplace_back([&, k] { rc_fold[k] = vimz_ivc_fold(segs[k], step_inputs + 4 * n_priv * lo[k], hi[k] - lo[k]); });
    started = k + 1;
  }
  for (auto& t : th_fold) t.join();
  for (size_t k = 0; k < S; k++) { segs[k]->pri->start_from = nullptr; segs[k]->pri->end_to = nullptr; segs[k]->pri->on_start = nullptr; segs[k]->pri->suppress_head = false; segs[k]->pri->preloaded_inputs = nullptr; segs[k]->pri->preloaded_rows = 0; }
  for (size_t k = 0; k < th_dig.size(); k++) if (th_dig[k].joinable()) th_dig[k].join();
  for (size_t k = 0; k < started && !rc; k++) if (rc_fold[k]) { rc = rc_fold[k]; if (segs[k]->ctx != ctx) ctx->err = segs[k]->ctx->err; }
  if (rc) { if (pre) drop_pre(); return rc; }
  const double t_m = now_s();
  vimz_ivc_merged* m = nullptr;
  if ((rc = vimz_ivc_merged_create(segs[0], &m))) return rc;
  for (size_t k = 1; k < S; k++) if ((rc = vimz_ivc_merge(m, segs[k]))) { vimz_ivc_merged_free(m); return rc; }
  if (seconds) { seconds[0] = t_chain; seconds[1] = now_s() - t_m; seconds[2] = now_s() - t_all; }
  *out = m;
  return VIMZ_OK;
}


// ---- fold_input for a run of rows whose START STATE is not known yet: a rank of a sharded proof (vimz_amd/distributed.py::prove_sharded) ----------------
// The rank's state follows from the digests of the rows of the ranks before it, which those ranks are hashing right now.  Instead of hashing its own rows
// for the exchange and then again inside its fold, the rank BEGINS its fold: the segments' calls start at once (inputs uploaded, the row-hash chains of their
// first batches running, with their wires), the rows' digests are handed out the moment those chain passes have produced them (vimz_ivc_pending_digests),
// the caller exchanges them, chains over the rows before its own and provides the start state (vimz_ivc_pending_start); the segments continue — segment k
// from segment k−1's end state, as in vimz_ivc_fold_segments — and vimz_ivc_pending_finish joins and merges.  Every row is hashed once on every rank.
struct vimz_ivc_pending {
  std::vector<vimz_ivc*> segs; size_t S = 0; std::vector<size_t> lo, hi;
  size_t nsteps = 0, stride = 0, lz = 0;
  std::vector<std::unique_ptr<StartLink>> links;      // links[k]: the state segment k starts from (links[0]: the caller's)
  std::vector<std::thread> th; std::vector<int> rc_fold;
  std::vector<Fe> digests; std::mutex mu; std::condition_variable cv; size_t delivered = 0; bool failed = false, started = false;
  double t_all = 0;
};
int vimz_ivc_fold_segments_begin(vimz_ivc* const* segs, size_t n_seg, const uint64_t* step_inputs, size_t nsteps, vimz_ivc_pending** out) {
  if (!segs || !n_seg || !out || !step_inputs || !nsteps) return VIMZ_ERR_INVALID;
  for (size_t k = 0; k < n_seg; k++) if (!segs[k]) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = segs[0]->ctx;
  for (size_t k = 1; k < n_seg; k++) {
    if (!same_shapes(segs[0], segs[k])) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_fold_segments_begin: the segments' IVCs must be of the same circuits and on the same device");
    for (size_t j = 0; j < k; j++) if (segs[j]->ctx == segs[k]->ctx) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_fold_segments_begin: every segment needs a context of its own");
  }
  const size_t stride = vimz_ivc_digest_stride(segs[0]);
  if (!stride) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_fold_segments_begin: this circuit's row digests depend on the IVC state (use vimz_ivc_state_chain and vimz_ivc_fold_segments)");
  std::unique_ptr<vimz_ivc_pending> P(new vimz_ivc_pending());
  P->t_all = now_s();
  const size_t S = std::min(n_seg, nsteps), base = nsteps / S, rem = nsteps % S;
  P->S = S; P->nsteps = nsteps; P->stride = stride; P->lz = segs[0]->pri->len_z;
  P->segs.assign(segs, segs + S); P->lo.resize(S); P->hi.resize(S); P->rc_fold.assign(S, VIMZ_OK);
  for (size_t k = 0, at = 0; k < S; k++) { P->lo[k] = at; at += base + (k < rem ? 1 : 0); P->hi[k] = at; }
  P->digests.assign(nsteps * stride, Fe::zero());
  for (size_t k = 0; k < S; k++) P->links.emplace_back(new StartLink());
  const size_t n_priv = segs[0]->pri->n_priv;
  std::vector<uint64_t> zero(4 * P->lz, 0);
  vimz_ivc_pending* q = P.get();
  for (size_t k = 0; k < S; k++) {
    const int rc = vimz_ivc_reset(segs[k], zero.data());      // (a placeholder: replaced when the segment's start state arrives)
    if (rc) { for (size_t j = 0; j < k; j++) { vimz_prover* pj = segs[j]->pri; pj->start_from = pj->end_to = nullptr; pj->on_start = nullptr; pj->on_digests = nullptr; } return rc; }
    vimz_prover* pk = segs[k]->pri; vimz_ivc* vk = segs[k];
    pk->start_from = q->links[k].get(); pk->end_to = k + 1 < S ? q->links[k + 1].get() : nullptr;
    pk->on_start = [vk] { for (uint32_t i = 0; i < vk->c1->len_z; i++) vk->z0[i] = vk->pri->z_cur[i]; };
    // only the chains' own outputs travel (the other slots of a job row are scratch of earlier batches)
    const cb::Builder& b = pk->circuit->build->b;
    pk->on_digests = [q, k, &b](const Fe* jv, size_t n, size_t js) {
      { std::lock_guard<std::mutex> g(q->mu);
        for (size_t r = 0; r < n; r++) for (auto& c : b.chains) if (c.phase == 0) for (uint32_t j = c.job_off; j < c.job_off + c.job_cnt; j++) q->digests[(q->lo[k] + r) * js + j] = jv[r * js + j];
        q->delivered++; }
      q->cv.notify_all();
    };
  }
  for (size_t k = 0; k < S; k++)
    P->th.emplace_back([q, k, step_inputs, n_priv] {
      q->rc_fold[k] = vimz_ivc_fold(q->segs[k], step_inputs + 4 * n_priv * q->lo[k], q->hi[k] - q->lo[k]);
      if (q->rc_fold[k]) { { std::lock_guard<std::mutex> g(q->mu); q->failed = true; } q->cv.notify_all(); if (k + 1 < q->S) q->links[k + 1]->fail(); }
    });
  *out = P.release();
  return VIMZ_OK;
}
// digests_out: nsteps x vimz_ivc_digest_stride elements (what vimz_ivc_row_digests returns for these rows); blocks until every segment's chain pass is done
int vimz_ivc_pending_digests(vimz_ivc_pending* p, uint64_t* digests_out) {
  if (!p || !digests_out) return VIMZ_ERR_INVALID;
  std::unique_lock<std::mutex> g(p->mu);
  p->cv.wait(g, [&] { return p->delivered == p->S || p->failed; });
  if (p->delivered != p->S) return vz_fail(p->segs[0]->ctx, VIMZ_ERR_INVALID, "vimz_ivc_pending_digests: a segment's fold failed before its rows were hashed");
  memcpy(digests_out, p->digests.data(), 32 * p->digests.size());
  return VIMZ_OK;
}
int vimz_ivc_pending_start(vimz_ivc_pending* p, const uint64_t* z_start) {
  if (!p || !z_start || p->started) return VIMZ_ERR_INVALID;
  std::vector<Fe> z(p->lz);
  for (size_t i = 0; i < p->lz; i++) { Fe c; memcpy(c.v, z_start + 4 * i, 32); if (!c.is_reduced()) return vz_fail(p->segs[0]->ctx, VIMZ_ERR_INVALID, "vimz_ivc_pending_start: a state element is not below the modulus"); z[i] = Fe::to_mont(c); }
  p->started = true;
  p->links[0]->publish(z.data(), z.size());
  return VIMZ_OK;
}
// joins the segments' folds and merges them into ONE object (out may be nullptr: cancel — also what happens when no start state was ever given); frees p.
// seconds (optional) = {0, merge, total since begin}.
int vimz_ivc_pending_finish(vimz_ivc_pending* p, vimz_ivc_merged** out, double seconds[3]) {
  if (!p) return VIMZ_ERR_INVALID;
  std::unique_ptr<vimz_ivc_pending> P(p);
  if (!p->started) p->links[0]->fail();
  for (auto& t : p->th) t.join();
  for (size_t k = 0; k < p->S; k++) { vimz_prover* pk = p->segs[k]->pri; pk->start_from = pk->end_to = nullptr; pk->on_start = nullptr; pk->on_digests = nullptr; }
  vimz_ctx* ctx = p->segs[0]->ctx;
  int rc = VIMZ_OK;
  for (size_t k = 0; k < p->S && !rc; k++) if (p->rc_fold[k]) { rc = p->rc_fold[k]; if (p->segs[k]->ctx != ctx) ctx->err = p->segs[k]->ctx->err; }
  if (rc || !out) return rc ? rc : (p->started ? VIMZ_OK : VIMZ_ERR_INVALID);
  const double t_m = now_s();
  vimz_ivc_merged* m = nullptr;
  if ((rc = vimz_ivc_merged_create(p->segs[0], &m))) return rc;
  for (size_t k = 1; k < p->S; k++) if ((rc = vimz_ivc_merge(m, p->segs[k]))) { vimz_ivc_merged_free(m); return rc; }
  if (seconds) { seconds[0] = 0; seconds[1] = now_s() - t_m; seconds[2] = now_s() - p->t_all; }
  *out = m;
  return VIMZ_OK;
}
// Rows of a fold call of `nsteps` rows whose Poseidon chains the library would evaluate on the host (its policy, or what vimz_set_head_rows pinned)
size_t vimz_head_rows_policy(size_t nsteps) { return std::min(head_rows_wanted(nsteps), nsteps); }
// ... and for a proof of `nsteps` rows made as concurrent segments (vimz_ivc_fold_segments): non-zero while the segments take head batches
size_t vimz_head_rows_policy_segments(size_t nsteps) { return nsteps <= HEAD_JOB_MAX ? std::min(head_rows_wanted(1), nsteps) : 0; }

}  // extern "C"
