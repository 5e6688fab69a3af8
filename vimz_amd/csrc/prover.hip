// Folding prover, ACCUMULATOR mode (Nova IVC in full — augmented circuits, secondary curve — is ivc.hip; both share the batch
// producer in prover_internal.hpp).  The per-row work of `fold_input` (vimz/src/nova_snark_backend/folding.rs:27-43 ->
// nova_scotia::create_recursive_circuit -> RecursiveSNARK::prove_step) on the step circuit's own
// instances, as a host loop over HIP kernels with every vector resident in HBM:
//
//   per batch of rows   GPU witness generation (witness.hpp)  — replaces one circom child process per step
//   per step            (A,B,C)·z2  ->  comm_W2 = MSM(ck, W2)  ->  T  ->  comm_T = MSM(ck, T)
//                       r = RO(...) on the host  ->  one fused fold of W, E and the running Az,Bz,Cz
//
// The instance folded is the step circuit's R1CS itself with public IO X = (z_{i+1}, z_i); the fold algebra, the commitments
// and the acceptance check (is_sat_relaxed + commitment openings) are Nova's NIFS, the transcript is ours.
// Such accumulators of different row segments merge (vimz_prover_merge*): BASELINE.json north_star's sharding picture.
#include "prover_internal.hpp"

// (leaked on purpose: worker threads must not be joined from a static destructor while the library is being unloaded)
HostPool& vz_shared_pool() {
  static HostPool* pool = [] { const unsigned hw = usable_cpus(); return new HostPool(hw > 2 ? std::min(15u, hw - 2) : 0u); }();
  return *pool;
}

std::atomic<long>& vz_head_rows_override() { static std::atomic<long> v{-1}; return v; }

extern "C" {

void vimz_prover_free(vimz_prover* p) {
  if (!p) return;
  if (p->ctx) {
    std::lock_guard<std::mutex> g(p->ctx->mu);
    hipSetDevice(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
    { int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi); vz_stream_release(p->ctx, lo, p->sB); vz_stream_release(p->ctx, lo, p->sH); }
    if (p->ev_head) hipEventDestroy(p->ev_head);
    if (p->ev_hash) hipEventDestroy(p->ev_hash);
    p->wsH.release();
    if (p->stage_host) hipHostFree(p->stage_host);
    if (p->jobvals_host) hipHostFree(p->jobvals_host);
    if (p->zs_host) hipHostFree(p->zs_host);
    hipFree(p->stage_d); hipFree(p->jobvals_d);
    for (auto& bb : p->buf) {
      for (auto e : bb.ev) hipEventDestroy(e);
      if (bb.wit_done) hipEventDestroy(bb.wit_done);
      for (auto e : bb.ev_p) hipEventDestroy(e);
      for (auto e : bb.ev_d) hipEventDestroy(e);
      if (bb.pin_d) hipHostFree(bb.pin_d);
      if (bb.pin) hipHostFree(bb.pin);
      if (bb.status_host) hipHostFree(bb.status_host);
      if (bb.row_flag) hipHostFree(bb.row_flag);
    }
    p->wsB.release();
    hipFree(p->priv_all_d); hipFree(p->zs_all_d); hipFree(p->job_all_d);
    for (void* d : p->retired) hipFree(d);
    for (void* d : p->owned) hipFree(d);
  }
  delete p;
}

int vimz_prover_create(vimz_ctx* ctx, const vimz_circuit* circuit, const vimz_bases* ck, size_t max_batch, vimz_prover** out) {
  return vz_prover_create_layout(ctx, circuit, ck, max_batch, 0, 0, 0, out);
}
}  // extern "C"

// step_wires != 0: `circuit` is an augmented circuit (aug/augmented.hpp) whose first step_wires wires / step_c rows are the
// step circuit; the prover is laid out for IVC (public IO = the last two wires, everything else committed).
int vz_prover_create_layout(vimz_ctx* ctx, const vimz_circuit* circuit, const vimz_bases* ck, size_t max_batch, int ivc, uint32_t step_wires, uint32_t step_c, vimz_prover** out) {
  if (!ctx || !circuit || !ck || !out || max_batch == 0) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_prover_create: bad argument");
  if (ck->curve != VIMZ_CURVE_BN254_G1) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_prover_create: the step circuits are over BN254 Fr; the key must be on BN254 G1");
  const cb::Builder& b = circuit->build->b;
  const uint32_t c0 = ivc ? 1u : 1 + 2 * b.len_z;
  const uint32_t n_aux = ivc ? b.n_wires - 3 : b.n_wires - 1 - 2 * b.len_z;
  if (ck->n < n_aux || ck->n < b.n_constraints()) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_prover_create: commitment key shorter than max(witness, constraints)");
  for (auto& J : b.jobs) if (J.t != 3 && J.t != 9) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_prover_create: the GPU Poseidon kernel supports widths 3 and 9 only (row widths must be multiples of 8... of the window fold)");
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  auto* p = new vimz_prover();
  p->ctx = ctx; p->circuit = circuit; p->ck = ck; p->max_batch = max_batch;
  static const bool dbg_create = getenv("VIMZ_DEBUG_TIMING") != nullptr;
  double t_c = now_s();
  auto lap = [&](const char* what) { if (dbg_create) { const double t = now_s(); fprintf(stderr, "[timing] prover_create(%p): %s %.1f ms\n", (void*)ctx, what, 1e3 * (t - t_c)); t_c = t; } };
  p->n_wires = b.n_wires; p->n_c = b.n_constraints(); p->len_z = b.len_z; p->n_priv = b.n_priv; p->n_aux = n_aux;
  p->n_jobs = (uint32_t)b.jobs.size(); p->n_fops = (uint32_t)b.fops.size();
  p->ivc = ivc != 0; p->c0 = c0; p->step_wires = ivc ? step_wires : b.n_wires; p->step_c = ivc ? step_c : b.n_constraints();
  p->n_bool = std::min<uint32_t>(b.n_bool, p->step_c);
  auto fail_free = [&](const char* what, hipError_t e) { for (void* d : p->owned) hipFree(d); delete p; return vz_fail(ctx, VIMZ_ERR_HIP, what, e); };
  hipError_t e;
#define UP(vec, dst) do { e = upload(vec, &dst); if (dst) p->owned.push_back((void*)dst); if (e != hipSuccess) return fail_free("upload " #vec, e); } while (0)
  UP(b.A.row_ptr, p->A.row_ptr); UP(b.A.col, p->A.col); UP(b.A.coef, p->A.coef);
  UP(b.B.row_ptr, p->B.row_ptr); UP(b.B.col, p->B.col); UP(b.B.coef, p->B.coef);
  UP(b.C.row_ptr, p->C.row_ptr); UP(b.C.col, p->C.col); UP(b.C.coef, p->C.coef);
  { const Fe* d = nullptr; const std::vector<Fe> dd = dict_for_device(b.dict); UP(dd, d); p->dict = (const uint32_t*)d; }      // (both forms of every coefficient: r1cs_ops.hpp)
  {
    std::vector<uint32_t> items, items_aug;
    const cb::Csr* Ms[3] = {&b.A, &b.B, &b.C};
    for (uint32_t m = 0; m < 3; m++)
      for (uint32_t r = 0; r + 1 < Ms[m]->row_ptr.size(); r++)
        if (Ms[m]->row_ptr[r + 1] - Ms[m]->row_ptr[r] > SPMV_LONG) (r < p->step_c ? items : items_aug).push_back((m << 30) | r);
    auto terms = [&](uint32_t it) { const cb::Csr* M = Ms[it >> 30]; const uint32_t r = it & 0x3fffffffu; return M->row_ptr[r + 1] - M->row_ptr[r]; };
    p->n_med = spmv_sort_items(items, terms); p->n_med_aug = spmv_sort_items(items_aug, terms);
    p->n_long = (uint32_t)items.size(); p->n_long_aug = (uint32_t)items_aug.size();
    UP(items, p->long_items); UP(items_aug, p->long_items_aug);
  }
  lap("CSR + dictionary uploads");
  WitnessDev& W = p->wd;
  UP(b.decomp, W.decomp); UP(b.lane_groups, W.groups); UP(b.lane_instr, W.instr); UP(b.lane_rows, W.rows);
  UP(b.jobs, W.jobs); UP(b.chains, W.chains); UP(b.fops, W.fops); UP(b.lc_terms, W.lc_terms);
  W.dict = p->dict;
  W.n_decomp = (uint32_t)b.decomp.size(); W.n_groups = (uint32_t)b.lane_groups.size(); W.n_jobs = p->n_jobs;
  W.n_chains = (uint32_t)b.chains.size(); W.n_fops = p->n_fops; W.n_wires = b.n_wires; W.len_z = b.len_z; W.n_priv = b.n_priv;
  {
    const cb::PoseidonTable& t3 = cb::poseidon_table(3); const cb::PoseidonTable& t9 = cb::poseidon_table(9);
    const Fe* d;
    UP(t3.C, d); W.pc3 = (const uint32_t*)d; UP(t3.M, d); W.pm3 = (const uint32_t*)d;
    UP(t9.C, d); W.pc9 = (const uint32_t*)d; UP(t9.M, d); W.pm9 = (const uint32_t*)d;
    W.rp3 = (uint32_t)t3.rp; W.rp9 = (uint32_t)t9.rp;
    for (int t : {3, 9}) {
      const auto& S = cb::poseidon_sparse_t<BnFr>(t);
      const int rp = t == 3 ? t3.rp : t9.rp;
      std::vector<Fe> packed((size_t)rp * t * 3);
      for (int r = 0; r < rp; r++)
        for (int i = 0; i < t; i++) {
          Fe* q = &packed[3 * ((size_t)r * t + i)];
          q[0] = S.ctil[(size_t)r * t + i]; q[1] = S.row[(size_t)r * t + i]; q[2] = i ? S.col[(size_t)r * (t - 1) + i - 1] : Fe::zero();
        }
      UP(packed, d); (t == 3 ? W.ps3 : W.ps9) = (const uint32_t*)d;
      UP(S.Pfin, d); (t == 3 ? W.pf3 : W.pf9) = (const uint32_t*)d;
      // the same four tables in the reduced-radix form the chain kernel computes in (witness.hpp: poseidon_group29)
      auto to29 = [](const std::vector<Fe>& v) {
        std::vector<uint32_t> o(v.size() * (size_t)F29_STRIDE, 0u);
        for (size_t i = 0; i < v.size(); i++) { const F29 x = F29::from_std(v[i]); for (int k = 0; k < 9; k++) o[i * F29_STRIDE + k] = x.v[k]; }
        return o;
      };
      const uint32_t* w;
      const std::vector<uint32_t> c29 = to29(t == 3 ? t3.C : t9.C), m29 = to29(t == 3 ? t3.M : t9.M), s29 = to29(packed), f29 = to29(S.Pfin);
      UP(c29, w); (t == 3 ? W.pc3_29 : W.pc9_29) = w;
      UP(m29, w); (t == 3 ? W.pm3_29 : W.pm9_29) = w;
      UP(s29, w); (t == 3 ? W.ps3_29 : W.ps9_29) = w;
      UP(f29, w); (t == 3 ? W.pf3_29 : W.pf9_29) = w;
    }
    W.poseidon29 = getenv("VIMZ_DEBUG_POSEIDON_STD") ? 0u : 1u;
  }
#undef UP
  lap("witness program + Poseidon tables");
  auto dalloc = [&](uint32_t** dst, size_t bytes) { e = hipMalloc((void**)dst, bytes ? bytes : 32); if (e == hipSuccess) { p->owned.push_back(*dst); e = hipMemset(*dst, 0, bytes ? bytes : 32); } return e; };   // (null-stream fills: synchronised below)
  const size_t B = max_batch;
  if (dalloc(&p->priv_d, 32 * B * p->n_priv) != hipSuccess || dalloc(&p->zs_d, 32 * (B + 1) * p->len_z) != hipSuccess ||
      dalloc(&p->Z_d, 32 * B * (size_t)p->n_wires) != hipSuccess || dalloc(&p->job_out_d, 32 * B * (size_t)(p->n_jobs + p->n_fops)) != hipSuccess ||
      dalloc(&p->status_d, 4 * B) != hipSuccess || dalloc(&p->Zrun, 32 * (size_t)p->n_wires) != hipSuccess ||
      dalloc(&p->E, 32 * (size_t)p->n_c) != hipSuccess || dalloc(&p->AZ, 32 * (size_t)p->n_c) != hipSuccess ||
      dalloc(&p->BZ, 32 * (size_t)p->n_c) != hipSuccess || dalloc(&p->CZ, 32 * (size_t)p->n_c) != hipSuccess ||
      dalloc(&p->T, 32 * (size_t)p->n_c) != hipSuccess || dalloc(&p->az2, 32 * (size_t)p->n_c) != hipSuccess ||
      dalloc(&p->bz2, 32 * (size_t)p->n_c) != hipSuccess || dalloc(&p->cz2, 32 * (size_t)p->n_c) != hipSuccess ||
      dalloc(&p->bad_d, 64) != hipSuccess)
    return fail_free("device allocation", e);
  lap("device buffers (running instance, batch 0)");
  { int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
    // (the producer sits on the lowest of HIP's three priority levels; confining it to a CU mask instead measured no better: DESIGN.md §9)
    if ((e = vz_stream_acquire(ctx, lo, &p->sB)) != hipSuccess) return fail_free("stream", e); }
  for (int k = 0; k < 2; k++) {
    auto& bb = p->buf[k];
    if (k == 0) { bb.Z = p->Z_d; bb.job_out = p->job_out_d; bb.status = p->status_d; }   // buffer 0 is shared with the witness hook
    else if (dalloc(&bb.Z, 32 * B * (size_t)p->n_wires) != hipSuccess || dalloc(&bb.job_out, 32 * B * (size_t)(p->n_jobs + p->n_fops)) != hipSuccess ||
             dalloc(&bb.status, 4 * B) != hipSuccess) return fail_free("device allocation", e);
    if (dalloc(&bb.az, 32 * B * (size_t)p->n_c) != hipSuccess || dalloc(&bb.bz, 32 * B * (size_t)p->n_c) != hipSuccess ||
        dalloc(&bb.cz, 32 * B * (size_t)p->n_c) != hipSuccess) return fail_free("device allocation", e);
    bb.ev.resize(B);
    for (size_t i = 0; i < B; i++) if ((e = hipEventCreateWithFlags(&bb.ev[i], hipEventDisableTiming)) != hipSuccess) return fail_free("event", e);
    if ((e = hipEventCreateWithFlags(&bb.wit_done, hipEventDisableTiming)) != hipSuccess) return fail_free("event", e);
    if ((e = hipHostMalloc(&bb.pin, 4 * (size_t)XYZZ_WORDS * MSM_MAX_WINDOWS * B)) != hipSuccess) return fail_free("pinned", e);
    if ((e = hipHostMalloc((void**)&bb.status_host, 4 * B)) != hipSuccess) return fail_free("pinned", e);
    if ((e = hipHostMalloc((void**)&bb.row_flag, 8 * B)) != hipSuccess) return fail_free("pinned", e);
    memset(bb.row_flag, 0, 8 * B); bb.flag_rows = B;
    bb.has_d.assign(B, 0);
    if (p->ivc) {        // per row: (A,B,C)·z done (the rows of a batch alternate between two producer streams)
      bb.ev_p.resize(B);
      for (size_t i = 0; i < B; i++) if ((e = hipEventCreateWithFlags(&bb.ev_p[i], hipEventDisableTiming)) != hipSuccess) return fail_free("event", e);
    }
    if (p->ivc && ivc_lookahead_enabled()) {     // lookahead of the IVC's large MSM (fold_issue_d)
      if (dalloc(&bb.d, 32 * B * (size_t)p->step_c) != hipSuccess) return fail_free("device allocation", e);
      if ((e = hipHostMalloc(&bb.pin_d, 4 * (size_t)XYZZ_WORDS * MSM_MAX_WINDOWS * B)) != hipSuccess) return fail_free("pinned", e);
      bb.ev_d.resize(B);
      for (size_t i = 0; i < B; i++) if ((e = hipEventCreateWithFlags(&bb.ev_d[i], hipEventDisableTiming)) != hipSuccess) return fail_free("event", e);
    }
  }
  lap("batch buffers, events, pinned");
  {   // head batch of a fold call (prover_internal.hpp: fold_head_batch): staging layout of the Poseidon jobs' wires, its stream
    int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
    if ((e = vz_stream_acquire(ctx, lo, &p->sH)) != hipSuccess) return fail_free("stream", e);
    p->sD = p->sH;
    if ((e = hipEventCreateWithFlags(&p->ev_head, hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&p->ev_hash, hipEventDisableTiming)) != hipSuccess)
      return fail_free("event", e);
    const cb::PoseidonTable& t3 = cb::poseidon_table(3); const cb::PoseidonTable& t9 = cb::poseidon_table(9);
    p->job_stage_off.assign(1, 0); p->job_fold_mask.clear();
    for (auto& J : b.jobs) {
      uint32_t mask = 0, nonconst = 0;
      for (uint32_t i = 0; i + 1 < J.t; i++) { if (J.in[i].kind == REF_CONST_ZERO) mask |= 1u << (i + 1); else nonconst++; }
      p->job_fold_mask.push_back(mask);
      p->job_stage_off.push_back(p->job_stage_off.back() + cb::poseidon_job_wire_count((int)J.t, J.t == 3 ? t3.rp : t9.rp, nonconst, J.out_wire != 0));
    }
    e = upload(p->job_stage_off, &p->job_stage_off_d); if (p->job_stage_off_d) p->owned.push_back((void*)p->job_stage_off_d);
    if (e != hipSuccess) return fail_free("upload job layout", e);
    bool ok = !b.jobs.empty() && b.gpu_witness && !b.zout.empty();
    for (auto& c : b.chains) ok = ok && c.phase <= 1;
    for (auto& f : b.fops) ok = ok && !f.early && f.op != FOP_LC;
    p->head_eligible = ok;
  }
  if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return fail_free("sync", e);   // the hipMemset fills above ran on the null stream
  lap("head-batch layout + sync of the fills");
  p->z_cur.assign(p->len_z, Fe::zero()); p->z0 = p->z_cur;
  *out = p;
  return VIMZ_OK;
}

extern "C" {
// Start a new IVC: z0 (len_z canonical elements).
// Rows of a short fold call whose Poseidon chains are evaluated on the HOST (the head batch, prover_internal.hpp): rows >= 0 pins the number for
// every later call of this process (0 = every row's witness entirely on the GPU), -1 restores the library's policy.  Returns the previous setting.
long vimz_set_head_rows(long rows) { return vz_head_rows_override().exchange(rows < 0 ? -1 : rows); }

int vimz_prover_reset(vimz_prover* p, const uint64_t* z0) {
  if (!p || !z0) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = p->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  for (uint32_t i = 0; i < p->len_z; i++) p->z_cur[i] = fe_from_canon(z0 + 4 * i);
  p->steps = 0; p->u = Fe::zero();
  p->comm_W.x = Fq::zero(); p->comm_W.y = Fq::zero(); p->comm_E = p->comm_W;
  // RO state seeded with a digest of the shape (stand-in for nova-snark's pp digest)
  Fe seed[6] = {cb::fe_from_u64(0x56494d7a), cb::fe_from_u64(p->n_c), cb::fe_from_u64(p->n_wires), cb::fe_from_u64(p->len_z),
                cb::fe_from_u64((uint64_t)p->circuit->transformation), cb::fe_from_u64((uint64_t)p->circuit->shape.width)};
  p->ro = cb::poseidon_hash(seed, 6);
  prover_start_state_changed(p);
  memset(p->phase_s, 0, sizeof(p->phase_s)); memset(p->phase_n, 0, sizeof(p->phase_n));
  P_TRY(hipMemsetAsync(p->Zrun, 0, 32 * (size_t)p->n_wires, ctx->stream));
  P_TRY(hipMemsetAsync(p->E, 0, 32 * (size_t)p->n_c, ctx->stream));
  P_TRY(hipMemsetAsync(p->AZ, 0, 32 * (size_t)p->n_c, ctx->stream));
  P_TRY(hipMemsetAsync(p->BZ, 0, 32 * (size_t)p->n_c, ctx->stream));
  P_TRY(hipMemsetAsync(p->CZ, 0, 32 * (size_t)p->n_c, ctx->stream));
  P_TRY(hipStreamSynchronize(ctx->stream));
  return VIMZ_OK;
}

// Witness generation for `rows` steps starting from the prover's current IVC state: the stage 0 of a fold (row hashes, the
// ahead-of-time pass where the circuit needs one, the host's state chain) and the witness kernels of one batch, on the main
// stream.  Leaves Z_d filled and advances nothing.  Caller holds the lock.
static int witness_batch_locked(vimz_prover* p, const uint64_t* inputs, size_t rows, std::vector<Fe>& zs) {
  vimz_ctx* ctx = p->ctx;
  hipStream_t s = ctx->stream;
  FoldJob job; job.step_inputs = inputs; job.nsteps = rows;
  int rc = fold_prepare(p, job);
  if (rc) return rc;
  zs = job.zs;
  double t0 = now_s();
  P_TRY(hipMemsetAsync(p->status_d, 0, 4 * rows, s));
  if ((rc = launch_witness(p, s, p->Z_d, p->job_out_d, p->status_d, p->priv_all_d, 0, rows, job, false))) return rc;
  p->last_status.assign(rows, 0);
  P_TRY(hipMemcpyAsync(p->last_status.data(), p->status_d, 4 * rows, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  p->phase_s[PH_WITNESS] += now_s() - t0;
  return VIMZ_OK;
}

// Witness generation only (parity hook for row W): fills the batch buffer and returns the full witness vectors.
// inputs: rows x n_priv canonical; z_wires_out (optional): rows x n_wires canonical; zs_out (optional): (rows+1) x len_z;
// status_out (optional): rows x uint32 (bit 0 = step relation unsatisfiable).
int vimz_prover_witness(vimz_prover* p, const uint64_t* inputs, size_t rows, uint64_t* z_wires_out, uint64_t* zs_out, uint32_t* status_out) {
  if (!p || !inputs || rows == 0 || rows > p->max_batch) return vz_fail(p ? p->ctx : nullptr, VIMZ_ERR_INVALID, "vimz_prover_witness: bad argument");
  if (!p->circuit->build->b.gpu_witness || p->circuit->build->b.zout.empty()) return vz_fail(p->ctx, VIMZ_ERR_INVALID, "vimz_prover_witness: this circuit has no GPU witness program");
  vimz_ctx* ctx = p->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  std::vector<Fe> zs;
  int rc = witness_batch_locked(p, inputs, rows, zs);
  if (rc) return rc;
  if (zs_out) for (size_t i = 0; i < zs.size(); i++) fe_to_canon(zs[i], zs_out + 4 * i);
  if (status_out) memcpy(status_out, p->last_status.data(), 4 * rows);
  if (z_wires_out) {
    const size_t n = rows * (size_t)p->n_wires;
    rc = vz_ensure_scratch(ctx, 32 * n); if (rc) return rc;
    launch_from_mont<Fr>(ctx->stream, p->Z_d, (uint32_t*)ctx->scratch, n);
    P_TRY(hipMemcpyAsync(z_wires_out, ctx->scratch, 32 * n, hipMemcpyDeviceToHost, ctx->stream));
    P_TRY(hipStreamSynchronize(ctx->stream));
  }
  return VIMZ_OK;
}

// (A,B,C)·z on the GPU for a host vector z (n_wires canonical) -> three n_constraints canonical vectors (parity hook, row V1).
int vimz_prover_spmv(vimz_prover* p, const uint64_t* z, uint64_t* az, uint64_t* bz, uint64_t* cz) {
  if (!p || !z || !az || !bz || !cz) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = p->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  int rc = vz_ensure_scratch(ctx, 32 * (size_t)p->n_wires); if (rc) return rc;
  uint32_t* zd = (uint32_t*)ctx->scratch;
  P_TRY(hipMemcpyAsync(zd, z, 32 * (size_t)p->n_wires, hipMemcpyHostToDevice, s));
  launch_to_mont<Fr>(s, zd, p->n_wires);
  launch_spmv(p, s, zd, p->az2, p->bz2, p->cz2);
  uint32_t* src[3] = {p->az2, p->bz2, p->cz2}; uint64_t* dst[3] = {az, bz, cz};
  for (int m = 0; m < 3; m++) {
    launch_from_mont<Fr>(s, src[m], p->T, p->n_c);
    P_TRY(hipMemcpyAsync(dst[m], p->T, 32 * (size_t)p->n_c, hipMemcpyDeviceToHost, s));
    P_TRY(hipStreamSynchronize(s));
  }
  return VIMZ_OK;
}

static void ro_absorb_point(const G1Aff& pt, Fe* out2) {  // (x_lo128, x_hi | parity(y) << 126) of the canonical coordinates
  Fq xc = Fq::from_mont(pt.x), yc = Fq::from_mont(pt.y);
  Fe lo = Fe::zero(), hi = Fe::zero();
  for (int i = 0; i < 4; i++) { lo.v[i] = xc.v[i]; hi.v[i] = xc.v[4 + i]; }
  hi.v[3] |= (yc.v[0] & 1u) << 30;
  out2[0] = Fe::to_mont(lo); out2[1] = Fe::to_mont(hi);
}


// Fold `nsteps` more rows.  step_inputs: nsteps x n_priv canonical elements, in the flattened order of
// vimz/src/nova_snark_backend/input.rs:57-96 (row_orig rows, then row_tran rows; redact: block then indicator).
//
// Schedule (all resident):
//   0. every private input to HBM; ONE hash-only pass of the phase-A Poseidon chains over ALL rows (they depend on the
//      row data only) -> the host runs the whole IVC state chain z_0..z_n (two pair hashes per row) and uploads it;
//   1. per batch, on stream B: full witness generation, then per row (A,B,C)·z and the witness commitment;
//      batch k+1 is produced while
//   2. stream A folds batch k sequentially: cross term, MSM(T), challenge, fused fold.
static int fold_core(vimz_prover* p, const uint64_t* step_inputs, const uint64_t* witnesses, size_t nsteps);
int vimz_prover_fold(vimz_prover* p, const uint64_t* step_inputs, size_t nsteps) {
  if (!p || (!step_inputs && nsteps)) return VIMZ_ERR_INVALID;
  if (p->circuit->build->b.zout.empty() && nsteps) return vz_fail(p->ctx, VIMZ_ERR_INVALID, "this circuit was loaded from an .r1cs and has no witness program: use vimz_prover_fold_witness");
  if (!p->circuit->build->b.gpu_witness && nsteps) return vz_fail(p->ctx, VIMZ_ERR_INVALID, "no GPU witness kernels for this step circuit: supply witnesses with vimz_prover_fold_witness");
  return fold_core(p, step_inputs, nullptr, nsteps);
}
int vimz_prover_fold_witness(vimz_prover* p, const uint64_t* witnesses, size_t nsteps) {
  if (!p || (!witnesses && nsteps)) return VIMZ_ERR_INVALID;
  return fold_core(p, nullptr, witnesses, nsteps);
}
}  // extern "C"
static int fold_core(vimz_prover* p, const uint64_t* step_inputs, const uint64_t* witnesses, size_t nsteps) {
  if (!nsteps) return VIMZ_OK;
  if (p->ivc) return vz_fail(p->ctx, VIMZ_ERR_INVALID, "this prover belongs to an IVC (vimz_ivc_*): fold through vimz_ivc_fold");
  vimz_ctx* ctx = p->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t nw = p->n_wires, nc = p->n_c;
  int rc;
  double t0;
  FoldJob job; job.step_inputs = step_inputs; job.witnesses = witnesses; job.nsteps = nsteps;
  if ((rc = fold_prepare(p, job, true))) return rc;
  const std::vector<Fe>& zs = job.zs;
  const BaseTables& tbl = job.tbl;
  const size_t pin_stride = FoldJob::pin_stride, nbatches = job.nbatches;

  // ---- 2. consumer: the sequential chain on stream A ------------------------------------------------------------
  for (size_t k = 0; k < nbatches; k++) {
    auto& bb = p->buf[k & 1];
    const size_t first = job.first(k), rows = job.rows(k);
    if ((rc = fold_issue_when_ready(p, job, k, true))) return rc;            // (batch 0 of a head-batch call is already out)
    if ((rc = fold_issue_when_ready(p, job, k + 1, false))) return rc;       // the next batch is produced while this one is folded
    P_TRY(hipEventSynchronize(bb.wit_done));
    for (size_t r = 0; r < rows; r++) if (bb.status_host[r]) {
      char msg[128]; snprintf(msg, sizeof(msg), "step %llu: the step relation is not satisfiable for these rows", (unsigned long long)(p->steps + r));
      hipStreamSynchronize(p->sB);
      for (uint32_t q = 0; q < p->len_z; q++) p->z_cur[q] = zs[first * p->len_z + q];   // the batches folded so far stay folded
      return vz_fail(ctx, VIMZ_ERR_UNSAT, msg);
    }
    // Software-pipelined sequential chain.  While the GPU runs MSM(T) of row r the host finishes the previous row's
    // commitment folds and prepares row r+1 (witness-commitment Horner, state digest); the cross term and MSM(T) of
    // row r+1 are enqueued right behind the fold of row r, so stream A never waits for host arithmetic other than the
    // MSM tail (Horner + inversion) and the challenge hash.
    struct Prep { G1Aff cW2; Fe zdig; bool ready = false; };
    std::vector<Prep> prep(rows);
    auto do_prep = [&](size_t r) -> int {          // host side of the fresh instance of row r
      double tp = now_s();
      P_TRY(wait_row_flag(bb, r, bb.ev[r]));
      prep[r].cW2 = msm_finish<BnG1>(p->planB, (char*)bb.pin + r * pin_stride);
      Fe zd = r == 0 ? p->zdigest : prep[r - 1].zdig;
      const Fe* znext = zs.data() + (first + r + 1) * p->len_z;
      for (uint32_t i = 0; i < p->len_z; i++) { Fe in[2] = {zd, znext[i]}; zd = cb::poseidon_hash(in, 2); }
      prep[r].zdig = zd; prep[r].ready = true;
      p->phase_s[PH_MSM_W] += now_s() - tp; p->phase_n[PH_MSM_W]++;
      return VIMZ_OK;
    };
    MsmPlan planT{};
    auto launch_T = [&](size_t r) -> int {         // cross term + MSM(T) of row r, asynchronous on stream A
      P_TRY(wait_row_flag(bb, r, bb.ev[r]));      // (on the host, not by a barrier on this high-priority stream: prover_internal.hpp)
      hipLaunchKernelGGL(k_cross_term<Fr>, dim3(stream_grid(nc)), dim3(256), 0, s, nc, p->AZ, p->BZ, p->CZ, p->u,
                         bb.az + 8 * r * nc, bb.bz + 8 * r * nc, bb.cz + 8 * r * nc, Fe::one(), p->T);
      P_TRY(hipGetLastError());
      if (!ctx->msm_ws.host_pinned) P_TRY(hipHostMalloc(&ctx->msm_ws.host_pinned, 4 * XYZZ_WORDS * MSM_MAX_WINDOWS));
      P_TRY(msm_launch<BnG1>(s, ctx->msm_ws, p->ck->d, p->T, nc, 1, 0, ctx->msm_ws.host_pinned, &planT, ctx->profiling ? ctx->ev : nullptr, 0, p->ck->tables ? &tbl : nullptr));
      if (ctx->profiling)   // into the last (unused) point slot of the pinned result buffer: a pageable destination would make the copy synchronous
        P_TRY(hipMemcpyAsync((char*)ctx->msm_ws.host_pinned + 4 * (size_t)XYZZ_WORDS * (MSM_MAX_WINDOWS - 1), ctx->msm_ws.totals, 8, hipMemcpyDeviceToHost, s));
      return VIMZ_OK;
    };
    struct Deferred { bool pending = false; G1Aff cW2, cT; Fe r128; } dfr;
    auto flush_deferred = [&]() {                  // comm_W, comm_E <- folded (host EC, off the GPU's critical path)
      if (!dfr.pending) return;
      double te = now_s();
      G1 a = from_affine(p->comm_W); G1 rb = scalar_mul(dfr.cW2, dfr.r128.v, 128); add_full(a, rb); p->comm_W = to_affine(a);
      G1 e1 = from_affine(p->comm_E); G1 rt = scalar_mul(dfr.cT, dfr.r128.v, 128); add_full(e1, rt); p->comm_E = to_affine(e1);
      dfr.pending = false;
      p->phase_s[PH_HOST_EC] += now_s() - te; p->phase_n[PH_HOST_EC]++;
    };
    size_t r0 = 0;
    if ((rc = do_prep(0))) return rc;
    if (p->steps == 0) {
      // base case: the running instance IS the first fresh instance (u = 1, E = 0), as RecursiveSNARK::new does
      P_TRY(wait_row_flag(bb, 0, bb.ev[0]));
      P_TRY(hipMemcpyAsync(p->Zrun, bb.Z, 32 * nw, hipMemcpyDeviceToDevice, s));
      P_TRY(hipMemcpyAsync(p->AZ, bb.az, 32 * nc, hipMemcpyDeviceToDevice, s));
      P_TRY(hipMemcpyAsync(p->BZ, bb.bz, 32 * nc, hipMemcpyDeviceToDevice, s));
      P_TRY(hipMemcpyAsync(p->CZ, bb.cz, 32 * nc, hipMemcpyDeviceToDevice, s));
      p->comm_W = prep[0].cW2; p->u = Fe::one(); p->zdigest = prep[0].zdig;
      Fe ab[4]; Fe cw[2]; ro_absorb_point(prep[0].cW2, cw);
      ab[0] = p->ro; ab[1] = cw[0]; ab[2] = cw[1]; ab[3] = p->zdigest;
      p->ro = cb::poseidon_hash(ab, 4);
      p->steps++;
      r0 = 1;
      if (rows > 1 && (rc = do_prep(1))) return rc;
    }
    if (r0 < rows && (rc = launch_T(r0))) return rc;
    for (size_t r = r0; r < rows; r++) {
      // host work hidden behind MSM(T) of row r
      flush_deferred();
      if ((rc = fold_issue_when_ready(p, job, k + 1, false))) return rc;
      if (r + 1 < rows && (rc = do_prep(r + 1))) return rc;
      t0 = now_s();
      P_TRY(hipStreamSynchronize(s));
      const G1Aff cT = msm_finish<BnG1>(planT, ctx->msm_ws.host_pinned);
      if (ctx->profiling) {
        float ms[6];
        for (int i = 0; i < 6; i++) { P_TRY(hipEventElapsedTime(&ms[i], ctx->ev[i], ctx->ev[i + 1])); ctx->last_msm.ms[i] = ms[i]; ctx->msm_tot_ms[i] += ms[i]; }
        memcpy(&ctx->last_msm.subs, (char*)ctx->msm_ws.host_pinned + 4 * (size_t)XYZZ_WORDS * (MSM_MAX_WINDOWS - 1), 8);
        ctx->last_msm.c = planT.c; ctx->last_msm.K = planT.K;
        ctx->msm_tot_calls++; ctx->msm_tot_points += nc; ctx->msm_tot_entries += ctx->last_msm.entries;
      }
      p->phase_s[PH_MSM_T] += now_s() - t0; p->phase_n[PH_MSM_T]++;
      t0 = now_s();
      // r = low 128 bits of Poseidon(ro, comm_W2, comm_T, digest of the IVC states so far)
      p->zdigest = prep[r].zdig;
      Fe ab[6]; Fe a2[2];
      ab[0] = p->ro; ro_absorb_point(prep[r].cW2, a2); ab[1] = a2[0]; ab[2] = a2[1]; ro_absorb_point(cT, a2); ab[3] = a2[0]; ab[4] = a2[1]; ab[5] = p->zdigest;
      p->ro = cb::poseidon_hash(ab, 6);
      Fe rc_canon = Fe::from_mont(p->ro);
      Fe r128 = Fe::zero(); for (int i = 0; i < 4; i++) r128.v[i] = rc_canon.v[i];
      const Fe rm = Fe::to_mont(r128);
      p->phase_s[PH_RO] += now_s() - t0; p->phase_n[PH_RO]++;
      Fold5 f;
      f.x1[0] = p->Zrun; f.x2[0] = bb.Z + 8 * r * nw; f.n[0] = nw;
      f.x1[1] = p->E; f.x2[1] = p->T; f.n[1] = nc;
      f.x1[2] = p->AZ; f.x2[2] = bb.az + 8 * r * nc; f.n[2] = nc;
      f.x1[3] = p->BZ; f.x2[3] = bb.bz + 8 * r * nc; f.n[3] = nc;
      f.x1[4] = p->CZ; f.x2[4] = bb.cz + 8 * r * nc; f.n[4] = nc;
      hipLaunchKernelGGL(k_fold5<Fr>, dim3(2048), dim3(256), 0, s, f, rm);
      P_TRY(hipGetLastError());
      p->u = Fe::add(p->u, rm);
      if (r + 1 < rows && (rc = launch_T(r + 1))) return rc;     // next row's GPU work is queued before any host EC
      dfr.pending = true; dfr.cW2 = prep[r].cW2; dfr.cT = cT; dfr.r128 = r128;
      p->steps++;
    }
    flush_deferred();
    // this buffer is rewritten by batch k+2: the folds that read it must have finished
    P_TRY(hipStreamSynchronize(s));
  }
  P_TRY(hipStreamSynchronize(p->sB));
  for (uint32_t i = 0; i < p->len_z; i++) p->z_cur[i] = zs[nsteps * p->len_z + i];
  return VIMZ_OK;
}

// ---- the IVC state chain in its two parts -------------------------------------------------------------------------------------------
// (1) the ROW DIGESTS — the outputs of the state-independent Poseidon chains of every row (the row hashes) —, which depend on the
//     row data only and are the expensive part (34 permutations per row at contrast HD, ≈ 150 at 8K): any GPU can compute them for
//     any rows, so the ranks of a sharded proof each hash their own rows side by side;
// (2) the serial chain z_i -> z_{i+1} over those digests: two or three small permutations per row on the host (≈ 12-25 µs).
// Plain circuits only (no Poseidon work that depends on step_in other than through the state hashes: everything but crop).
static bool chain_is_plain(const vimz_prover* p) {
  const cb::Builder& b = p->circuit->build->b;
  size_t nA = 0, nE = 0;
  for (auto& c : b.chains) { if (c.phase == 0) nA++; else if (c.phase != 1) nE++; }
  for (auto& f : b.fops) if (f.early) return false;
  return nA > 0 && nE == 0;
}
// caller holds the lock and has set the device; out: nsteps x (n_jobs + n_fops) elements (Montgomery)
static int compute_row_digests(vimz_prover* p, const uint64_t* step_inputs, size_t nsteps, Fe* out) {
  vimz_ctx* ctx = p->ctx;
  const cb::Builder& b = p->circuit->build->b;
  const size_t jstride = p->n_jobs + p->n_fops;
  std::vector<uint32_t> chainsA;
  for (uint32_t c = 0; c < b.chains.size(); c++) if (b.chains[c].phase == 0) chainsA.push_back(c);
  if (nsteps <= 24 && p->head_eligible && head_rows_wanted()) {      // (48 chains: four rounds of the pool, about one chain latency of the GPU; 96 rows until round 4)
    // Short inputs: the row-hash chains on the host's thread pool (0.4 ms per chain of 17 permutations and core) instead of one
    // Poseidon-chain latency of the GPU (≈10 ms whatever the row count) — the start states of a few short row segments that are
    // about to be folded concurrently must not cost as much as folding them.
    for (size_t i = 0; i < nsteps * jstride; i++) out[i] = Fe::zero();
    const uint32_t priv0 = 1 + 2 * b.len_z;
    vz_shared_pool().run(nsteps * chainsA.size(), [&](size_t task) {
      const size_t r = task / chainsA.size();
      const Chain& C = b.chains[chainsA[task % chainsA.size()]];
      Fe prev = Fe::zero();
      for (uint32_t k = 0; k < C.job_cnt; k++) {
        const uint32_t j = C.job_off + k;
        const HashJob& Jb = b.jobs[j];
        Fe in[POSEIDON_MAX_T];
        for (uint32_t i = 0; i + 1 < Jb.t; i++) {
          const ValRef& ref = Jb.in[i];
          if (ref.kind == REF_WIRE) in[i] = fe_from_canon(step_inputs + 4 * (r * p->n_priv + (ref.idx - priv0)));
          else if (ref.kind == REF_JOB) in[i] = ref.idx + 1 == j ? prev : out[r * jstride + ref.idx];
          else in[i] = Fe::zero();
        }
        prev = cb::poseidon_hash(in, (int)Jb.t - 1);
        out[r * jstride + j] = prev;
      }
    });
    return VIMZ_OK;
  }
  hipStream_t s = ctx->stream;
  P_TRY(grow(p->retired, &p->priv_all_d, &p->cap_priv_all, 32 * nsteps * (size_t)p->n_priv, 32 * 1024 * (size_t)p->n_priv));
  P_TRY(grow(p->retired, &p->job_all_d, &p->cap_job_all, 32 * nsteps * jstride, 32 * 1024 * jstride));
  P_TRY(hipMemcpyAsync(p->priv_all_d, step_inputs, 32 * nsteps * (size_t)p->n_priv, hipMemcpyHostToDevice, s));
  P_TRY(hipMemsetAsync(p->job_all_d, 0, 32 * nsteps * jstride, s));
  for (size_t off = 0; off < nsteps; off += 32768) {
    const unsigned rows = (unsigned)std::min<size_t>(32768, nsteps - off);
    hipLaunchKernelGGL(k_wit_chains, dim3((chainsA.size() + 3) / 4, rows), dim3(64), 0, s, p->wd, 0u, (uint32_t*)nullptr, p->job_all_d + 8 * off * jstride,
                       (const uint32_t*)(p->priv_all_d + 8 * off * p->n_priv));
  }
  P_TRY(hipGetLastError());
  P_TRY(hipMemcpyAsync(out, p->job_all_d, 32 * nsteps * jstride, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  return VIMZ_OK;
}

extern "C" {
// elements per row of vimz_prover_row_digests' output (0: this circuit's digests depend on the state — crop —, use vimz_prover_state_chain)
size_t vimz_prover_digest_stride(const vimz_prover* p) { return p && chain_is_plain(p) ? (size_t)p->n_jobs + p->n_fops : 0; }
// part (1): digests_out = nsteps x stride elements (4 x u64 each, Montgomery limbs: opaque to the caller)
int vimz_prover_row_digests(vimz_prover* p, const uint64_t* step_inputs, size_t nsteps, uint64_t* digests_out) {
  if (!p || !digests_out || (!step_inputs && nsteps)) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = p->ctx;
  if (!chain_is_plain(p)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_prover_row_digests: this circuit's row digests depend on the IVC state: use vimz_prover_state_chain");
  if (!nsteps) return VIMZ_OK;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  return compute_row_digests(p, step_inputs, nsteps, reinterpret_cast<Fe*>(digests_out));
}
// part (2): zs_out[(nsteps+1) x len_z] canonical from z_start, the rows' private inputs and their digests (host only)
int vimz_prover_chain_from_digests(vimz_prover* p, const uint64_t* z_start, const uint64_t* step_inputs, const uint64_t* digests, size_t nsteps, uint64_t* zs_out) {
  if (!p || !z_start || !zs_out || ((!step_inputs || !digests) && nsteps)) return VIMZ_ERR_INVALID;
  if (!chain_is_plain(p)) return vz_fail(p->ctx, VIMZ_ERR_INVALID, "vimz_prover_chain_from_digests: use vimz_prover_state_chain for this circuit");
  const size_t jstride = p->n_jobs + p->n_fops;
  const Fe* dg = reinterpret_cast<const Fe*>(digests);
  for (size_t i = 0; i < nsteps * jstride; i++) if (!dg[i].is_reduced()) return vz_fail(p->ctx, VIMZ_ERR_INVALID, "vimz_prover_chain_from_digests: a digest element is not below the modulus");
  std::vector<Fe> zs((nsteps + 1) * p->len_z, Fe::zero());
  for (uint32_t i = 0; i < p->len_z; i++) { Fe c; memcpy(c.v, z_start + 4 * i, 32); if (!c.is_reduced()) return vz_fail(p->ctx, VIMZ_ERR_INVALID, "z_start element not below the modulus"); zs[i] = Fe::to_mont(c); }
  if (nsteps) host_state_chain(p, step_inputs, nsteps, dg, jstride, zs);
  for (size_t i = 0; i < zs.size(); i++) fe_to_canon(zs[i], zs_out + 4 * i);
  return VIMZ_OK;
}

// IVC state chain only (no folding): zs_out[(nsteps+1) x len_z] canonical, starting from z_start.  Both parts on this GPU: the row
// digests (one hash-only GPU pass, or the host pool for short inputs), then the host's chain.  A multi-GPU driver uses it — or the
// two parts separately — to find the state at which a row segment starts.
int vimz_prover_state_chain(vimz_prover* p, const uint64_t* z_start, const uint64_t* step_inputs, size_t nsteps, uint64_t* zs_out) {
  if (!p || !z_start || !zs_out || (!step_inputs && nsteps)) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = p->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  const std::vector<Fe> keep = p->z_cur;
  for (uint32_t i = 0; i < p->len_z; i++) p->z_cur[i] = fe_from_canon(z_start + 4 * i);
  FoldJob job; job.step_inputs = step_inputs; job.nsteps = nsteps;
  int rc = VIMZ_OK;
  if (nsteps && chain_is_plain(p)) {
    const size_t jstride = p->n_jobs + p->n_fops;
    std::vector<Fe> dg(nsteps * jstride);
    rc = compute_row_digests(p, step_inputs, nsteps, dg.data());
    if (!rc) {
      job.zs.assign((nsteps + 1) * p->len_z, Fe::zero());
      for (uint32_t i = 0; i < p->len_z; i++) job.zs[i] = p->z_cur[i];
      host_state_chain(p, step_inputs, nsteps, dg.data(), jstride, job.zs);
    }
  } else if (nsteps) rc = fold_prepare(p, job);      // (the stage 0 of a fold, with the ahead-of-time witness pass crop needs)
  else job.zs = p->z_cur;
  p->z_cur = keep;
  if (rc) return rc;
  for (size_t i = 0; i < job.zs.size(); i++) fe_to_canon(job.zs[i], zs_out + 4 * i);
  return VIMZ_OK;
}

// ---- export / merge of running instances: the host-side final fold of row segments folded on different GPUs ----------
// Blob: header (8 x u64) | u | comm_W | comm_E | ro | zdigest | z_cur | z0 | Zrun | E | AZ | BZ | CZ   (Montgomery limbs)
struct BlobHeader { uint64_t magic, n_wires, n_c, len_z, steps, transformation, width, reserved; };
static const uint64_t BLOB_MAGIC = 0x315a4d4956ull;  // "VIMZ1"

size_t vimz_prover_export_size(const vimz_prover* p) {
  if (!p) return 0;
  return sizeof(BlobHeader) + 32 + 64 + 64 + 32 + 32 + 64 * (size_t)p->len_z + 32 * ((size_t)p->n_wires + 4 * (size_t)p->n_c);
}

int vimz_prover_export(vimz_prover* p, uint8_t* blob, size_t cap) {
  if (!p || !blob || cap < vimz_prover_export_size(p)) return vz_fail(p ? p->ctx : nullptr, VIMZ_ERR_INVALID, "vimz_prover_export: buffer too small");
  vimz_ctx* ctx = p->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  BlobHeader h{BLOB_MAGIC, p->n_wires, p->n_c, p->len_z, p->steps, (uint64_t)p->circuit->transformation, (uint64_t)p->circuit->shape.width, 0};
  uint8_t* o = blob;
  memcpy(o, &h, sizeof(h)); o += sizeof(h);
  memcpy(o, p->u.v, 32); o += 32;
  memcpy(o, &p->comm_W, 64); o += 64; memcpy(o, &p->comm_E, 64); o += 64;
  memcpy(o, p->ro.v, 32); o += 32; memcpy(o, p->zdigest.v, 32); o += 32;
  memcpy(o, p->z_cur.data(), 32 * p->len_z); o += 32 * p->len_z;
  memcpy(o, p->z0.data(), 32 * p->len_z); o += 32 * p->len_z;
  const uint32_t* src[5] = {p->Zrun, p->E, p->AZ, p->BZ, p->CZ};
  const size_t len[5] = {p->n_wires, p->n_c, p->n_c, p->n_c, p->n_c};
  for (int k = 0; k < 5; k++) { P_TRY(hipMemcpyAsync(o, src[k], 32 * len[k], hipMemcpyDeviceToHost, s)); o += 32 * len[k]; }
  P_TRY(hipStreamSynchronize(s));
  return VIMZ_OK;
}

// Fold the exported relaxed instance (U2, W2) into this prover's running instance (U1, W1) — Nova's NIFS for two relaxed
// instances:  T = Az1∘Bz2 + Az2∘Bz1 − u1·Cz2 − u2·Cz1,  E = E1 + r·T + r²·E2,  W = W1 + r·W2,  u = u1 + r·u2,
// comm_W = comm_W1 + r·comm_W2,  comm_E = comm_E1 + r·comm_T + r²·comm_E2,  r = Poseidon(ro1, ro2, comm_T) mod 2^128.
struct OtherInstance {       // a relaxed instance to fold in: host scalars + device vectors (Montgomery)
  Fe u, ro, zdigest; G1Aff cW, cE; std::vector<Fe> z_cur, z0; uint64_t steps;
  const uint32_t *Z, *E, *AZ, *BZ, *CZ;
};

// caller holds p->ctx->mu, has set the device, and guarantees the other instance's vectors are complete and stay valid
static int merge_core(vimz_prover* p, const OtherInstance& o) {
  vimz_ctx* ctx = p->ctx;
  hipStream_t s = ctx->stream;
  const size_t nw = p->n_wires, nc = p->n_c;
  if (o.steps == 0) return VIMZ_OK;
  // Segments are merged in row order and must be ADJACENT: the incoming segment starts at the state this one ends in.  Without this
  // check segments that are out of order, duplicated or from another image merge into an accumulator that still "verifies".
  for (uint32_t k = 0; k < p->len_z; k++)
    if (!o.z0[k].eq(p->z_cur[k])) return vz_fail(ctx, VIMZ_ERR_INVALID, "merge: the incoming segment does not start at the state this accumulator ends in");
  if (p->steps == 0) {   // this prover is empty: adopt the other instance
    P_TRY(hipMemcpyAsync(p->Zrun, o.Z, 32 * nw, hipMemcpyDeviceToDevice, s)); P_TRY(hipMemcpyAsync(p->E, o.E, 32 * nc, hipMemcpyDeviceToDevice, s));
    P_TRY(hipMemcpyAsync(p->AZ, o.AZ, 32 * nc, hipMemcpyDeviceToDevice, s)); P_TRY(hipMemcpyAsync(p->BZ, o.BZ, 32 * nc, hipMemcpyDeviceToDevice, s));
    P_TRY(hipMemcpyAsync(p->CZ, o.CZ, 32 * nc, hipMemcpyDeviceToDevice, s)); P_TRY(hipStreamSynchronize(s));
    p->u = o.u; p->comm_W = o.cW; p->comm_E = o.cE; p->ro = o.ro; p->zdigest = o.zdigest; p->z_cur = o.z_cur; p->z0 = o.z0; p->steps = o.steps;
    return VIMZ_OK;
  }
  hipLaunchKernelGGL(k_cross_term<Fr>, dim3(stream_grid(nc)), dim3(256), 0, s, nc, p->AZ, p->BZ, p->CZ, p->u, o.AZ, o.BZ, o.CZ, o.u, p->T);
  P_TRY(hipGetLastError());
  uint64_t pt[8];
  int rc = vz_msm_device(ctx, p->ck, 0, p->T, nc, 1, 0, pt, VIMZ_FORM_MONTGOMERY);
  if (rc) return rc;
  G1Aff cT; memcpy(cT.x.v, pt, 32); memcpy(cT.y.v, pt + 4, 32);
  Fe ab[4]; Fe a2[2]; ro_absorb_point(cT, a2);
  ab[0] = p->ro; ab[1] = o.ro; ab[2] = a2[0]; ab[3] = a2[1];
  p->ro = cb::poseidon_hash(ab, 4);
  { Fe zz[2] = {p->zdigest, o.zdigest}; p->zdigest = cb::poseidon_hash(zz, 2); }
  Fe rc_canon = Fe::from_mont(p->ro);
  Fe r128 = Fe::zero(); for (int i = 0; i < 4; i++) r128.v[i] = rc_canon.v[i];
  const Fe rm = Fe::to_mont(r128), rm2 = Fe::sqr(rm);
  Fold5 f;
  f.x1[0] = p->Zrun; f.x2[0] = o.Z; f.n[0] = nw;
  f.x1[1] = p->E; f.x2[1] = p->T; f.n[1] = nc;
  f.x1[2] = p->AZ; f.x2[2] = o.AZ; f.n[2] = nc;
  f.x1[3] = p->BZ; f.x2[3] = o.BZ; f.n[3] = nc;
  f.x1[4] = p->CZ; f.x2[4] = o.CZ; f.n[4] = nc;
  hipLaunchKernelGGL(k_fold5<Fr>, dim3(2048), dim3(256), 0, s, f, rm);
  hipLaunchKernelGGL(k_axpy_inplace<Fr>, dim3(stream_grid(nc)), dim3(256), 0, s, nc, p->E, rm2, o.E);
  P_TRY(hipGetLastError());
  Fe r2c = Fe::from_mont(rm2);   // r^2 as a canonical 256-bit scalar
  G1 a = from_affine(p->comm_W); G1 t1 = scalar_mul(o.cW, r128.v, 128); add_full(a, t1); p->comm_W = to_affine(a);
  G1 e = from_affine(p->comm_E); G1 t2 = scalar_mul(cT, r128.v, 128); add_full(e, t2);
  G1 t3 = scalar_mul(o.cE, r2c.v, 256); add_full(e, t3); p->comm_E = to_affine(e);
  p->u = Fe::add(p->u, Fe::mul(rm, o.u));
  p->steps += o.steps;
  p->z_cur = o.z_cur;     // segments are merged in row order: the merged chain ends where the later segment ends
  P_TRY(hipStreamSynchronize(s));
  return VIMZ_OK;
}

int vimz_prover_merge(vimz_prover* p, const uint8_t* blob, size_t len) {
  if (!p || !blob || len < sizeof(BlobHeader)) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = p->ctx;
  BlobHeader h; memcpy(&h, blob, sizeof(h));
  if (h.magic != BLOB_MAGIC || h.n_wires != p->n_wires || h.n_c != p->n_c || h.len_z != p->len_z || len < vimz_prover_export_size(p))
    return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_prover_merge: blob does not match this prover's circuit");
  if (h.transformation != (uint64_t)p->circuit->transformation || h.width != (uint64_t)p->circuit->shape.width)
    return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_prover_merge: blob is of another transformation or row width");
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t nw = p->n_wires, nc = p->n_c;
  const uint8_t* o = blob + sizeof(h);
  OtherInstance oi; oi.steps = h.steps;
  memcpy(oi.u.v, o, 32); o += 32;
  memcpy(&oi.cW, o, 64); o += 64; memcpy(&oi.cE, o, 64); o += 64;
  memcpy(oi.ro.v, o, 32); o += 32; memcpy(oi.zdigest.v, o, 32); o += 32;
  oi.z_cur.resize(p->len_z); oi.z0.resize(p->len_z);
  memcpy(oi.z_cur.data(), o, 32 * p->len_z); o += 32 * p->len_z; memcpy(oi.z0.data(), o, 32 * p->len_z); o += 32 * p->len_z;
  if (h.steps == 0) return VIMZ_OK;
  {   // untrusted host state: limbs must be below the modulus (the vectors are range-checked on the device below)
    bool ok = oi.u.is_reduced() && oi.ro.is_reduced() && oi.zdigest.is_reduced() && oi.cW.x.is_reduced() && oi.cW.y.is_reduced() && oi.cE.x.is_reduced() && oi.cE.y.is_reduced();
    for (auto& z : oi.z_cur) ok = ok && z.is_reduced();
    for (auto& z : oi.z0) ok = ok && z.is_reduced();
    if (!ok) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_prover_merge: a field element of the blob is not below its modulus");
  }
  // stage the other instance's vectors in the (idle) batch buffer 0: Z | E | AZ | BZ | CZ
  auto& bb = p->buf[0];
  uint32_t *Z2 = bb.Z, *E2 = p->az2, *AZ2 = bb.az, *BZ2 = bb.bz, *CZ2 = bb.cz;
  P_TRY(hipMemcpyAsync(Z2, o, 32 * nw, hipMemcpyHostToDevice, s)); o += 32 * nw;
  P_TRY(hipMemcpyAsync(E2, o, 32 * nc, hipMemcpyHostToDevice, s)); o += 32 * nc;
  P_TRY(hipMemcpyAsync(AZ2, o, 32 * nc, hipMemcpyHostToDevice, s)); o += 32 * nc;
  P_TRY(hipMemcpyAsync(BZ2, o, 32 * nc, hipMemcpyHostToDevice, s)); o += 32 * nc;
  P_TRY(hipMemcpyAsync(CZ2, o, 32 * nc, hipMemcpyHostToDevice, s)); o += 32 * nc;
  {
    const uint32_t zero2[2] = {0, 0};
    P_TRY(hipMemcpyAsync(p->bad_d, zero2, 8, hipMemcpyHostToDevice, s));
    const uint32_t* vs[5] = {Z2, E2, AZ2, BZ2, CZ2}; const size_t ns[5] = {nw, nc, nc, nc, nc};
    for (int k = 0; k < 5; k++) hipLaunchKernelGGL(k_count_unreduced<Fr>, dim3(stream_grid(ns[k])), dim3(256), 0, s, ns[k], vs[k], p->bad_d);
    uint32_t nbad = 0;
    P_TRY(hipMemcpyAsync(&nbad, p->bad_d, 4, hipMemcpyDeviceToHost, s));
    P_TRY(hipStreamSynchronize(s));
    if (nbad) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_prover_merge: a vector element of the blob is not below its modulus");
  }
  oi.Z = Z2; oi.E = E2; oi.AZ = AZ2; oi.BZ = BZ2; oi.CZ = CZ2;
  return merge_core(p, oi);
}

// Same final fold when both provers live on the same GPU (row segments folded concurrently on one device): the vectors
// are read in place, no host round trip.  `src` is left unchanged.
int vimz_prover_merge_prover(vimz_prover* p, vimz_prover* src) {
  if (!p || !src || p == src) return VIMZ_ERR_INVALID;
  if (p->n_wires != src->n_wires || p->n_c != src->n_c || p->len_z != src->len_z || p->ctx->device != src->ctx->device)
    return vz_fail(p->ctx, VIMZ_ERR_INVALID, "vimz_prover_merge_prover: provers must share circuit shape and device");
  vimz_ctx* ctx = p->ctx;
  std::unique_lock<std::mutex> g1(p->ctx->mu, std::defer_lock), g2(src->ctx->mu, std::defer_lock);
  if (p->ctx == src->ctx) g1.lock(); else std::lock(g1, g2);
  P_TRY(hipSetDevice(ctx->device));
  P_TRY(hipStreamSynchronize(src->ctx->stream));
  OtherInstance oi;
  oi.u = src->u; oi.ro = src->ro; oi.zdigest = src->zdigest; oi.cW = src->comm_W; oi.cE = src->comm_E; oi.z_cur = src->z_cur; oi.z0 = src->z0;
  oi.steps = src->steps; oi.Z = src->Zrun; oi.E = src->E; oi.AZ = src->AZ; oi.BZ = src->BZ; oi.CZ = src->CZ;
  return merge_core(p, oi);
}

// Running instance: comm_W, comm_E (affine canonical), u, X = (z_i, z_0 ...) — here X is read back from Zrun.
int vimz_prover_instance(vimz_prover* p, uint64_t comm_W[8], uint64_t comm_E[8], uint64_t u[4], uint64_t* z_current, uint64_t* steps) {
  if (!p) return VIMZ_ERR_INVALID;
  if (comm_W) { Fq x = Fq::from_mont(p->comm_W.x), y = Fq::from_mont(p->comm_W.y); memcpy(comm_W, x.v, 32); memcpy(comm_W + 4, y.v, 32); }
  if (comm_E) { Fq x = Fq::from_mont(p->comm_E.x), y = Fq::from_mont(p->comm_E.y); memcpy(comm_E, x.v, 32); memcpy(comm_E + 4, y.v, 32); }
  if (u) fe_to_canon(p->u, u);
  if (z_current) for (uint32_t i = 0; i < p->len_z; i++) fe_to_canon(p->z_cur[i], z_current + 4 * i);
  if (steps) *steps = p->steps;
  return VIMZ_OK;
}

// Download the running witness vectors (canonical): z_run n_wires ([u | X | W]), E n_constraints.
int vimz_prover_running(vimz_prover* p, uint64_t* z_run, uint64_t* E) {
  if (!p) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = p->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t mx = std::max((size_t)p->n_wires, (size_t)p->n_c);
  int rc = vz_ensure_scratch(ctx, 32 * mx); if (rc) return rc;
  if (z_run) { launch_from_mont<Fr>(s, p->Zrun, (uint32_t*)ctx->scratch, p->n_wires); P_TRY(hipMemcpyAsync(z_run, ctx->scratch, 32 * (size_t)p->n_wires, hipMemcpyDeviceToHost, s)); P_TRY(hipStreamSynchronize(s)); }
  if (E) { launch_from_mont<Fr>(s, p->E, (uint32_t*)ctx->scratch, p->n_c); P_TRY(hipMemcpyAsync(E, ctx->scratch, 32 * (size_t)p->n_c, hipMemcpyDeviceToHost, s)); P_TRY(hipStreamSynchronize(s)); }
  return VIMZ_OK;
}

// verify_folded_proof (vimz/src/nova_snark_backend/folding.rs:45-56 -> RecursiveSNARK::verify's is_sat_relaxed):
// recompute (A,B,C)·Z from the folded witness, check Az∘Bz = u·Cz + E row by row, and re-open both commitments.
// result: 0 = accepted; bit 0 = relation violated, bit 1 = comm_W mismatch, bit 2 = comm_E mismatch, bit 3 = running products drifted.
int vimz_prover_verify(vimz_prover* p, uint32_t* result) {
  if (!p || !result) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = p->ctx;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t nc = p->n_c;
  uint32_t res = 0;
  launch_spmv(p, s, p->Zrun, p->az2, p->bz2, p->cz2);
  uint32_t init[2] = {0, 0xffffffffu};
  P_TRY(hipMemcpyAsync(p->bad_d, init, 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_check_relaxed<Fr>, dim3(stream_grid(nc)), dim3(256), 0, s, nc, p->az2, p->bz2, p->cz2, p->u, (const uint32_t*)p->E, p->bad_d);
  uint32_t bad[2];
  P_TRY(hipMemcpyAsync(bad, p->bad_d, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 1;
  // running products must equal the recomputed ones (linearity bookkeeping)
  P_TRY(hipMemcpyAsync(p->bad_d, init, 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_check_relaxed<Fr>, dim3(stream_grid(nc)), dim3(256), 0, s, nc, p->AZ, p->BZ, p->CZ, p->u, (const uint32_t*)p->E, p->bad_d);
  P_TRY(hipMemcpyAsync(bad, p->bad_d, 8, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  if (bad[0]) res |= 8;
  uint64_t pt[8];
  int rc = vz_msm_device(ctx, p->ck, 0, p->Zrun + 8 * (size_t)p->c0, p->n_aux, 1, 0, pt, VIMZ_FORM_MONTGOMERY);
  if (rc) return rc;
  if (memcmp(pt, p->comm_W.x.v, 32) || memcmp(pt + 4, p->comm_W.y.v, 32)) res |= 2;
  rc = vz_msm_device(ctx, p->ck, 0, p->E, nc, 1, 0, pt, VIMZ_FORM_MONTGOMERY);
  if (rc) return rc;
  if (memcmp(pt, p->comm_E.x.v, 32) || memcmp(pt + 4, p->comm_E.y.v, 32)) res |= 4;
  *result = res;
  return VIMZ_OK;
}

// seconds[9] / counts[9]: witness, state chain (host), spmv, msm(W), cross term, msm(T), RO (host), fold, host EC
int vimz_prover_profile(const vimz_prover* p, double seconds[9], uint64_t counts[9]) {
  if (!p) return VIMZ_ERR_INVALID;
  if (seconds) memcpy(seconds, p->phase_s, sizeof(double) * PH_COUNT);
  if (counts) memcpy(counts, p->phase_n, sizeof(uint64_t) * PH_COUNT);
  return VIMZ_OK;
}

}  // extern "C"
