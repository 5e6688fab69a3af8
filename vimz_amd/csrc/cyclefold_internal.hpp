// Internals of the Nova + CycleFold prover shared by cyclefold.hip (prove_step loop, verifier, proof I/O, self-check) and cyclefold_merge.hip
// (ONE object out of several segments' proofs).
#pragma once
#include "ivc_internal.hpp"
#include "aug/cyclefold.hpp"
#include "proof_io.hpp"

enum { CP_CROSS_MSM = 0, CP_CF = 1, CP_SYNTH = 2, CP_FRESH = 3, CP_PRODUCER = 4, CP_TOTAL = 5, CP_COUNT = 8 };

struct vimz_cf {
  vimz_ctx* ctx = nullptr;
  std::unique_ptr<vimz_circuit> circ;             // the step circuit's copy, with F' appended
  std::unique_ptr<CfMainCircuit> c1;
  CfCircuit cf;
  vimz_prover* pri = nullptr;
  const vimz_bases *ck1 = nullptr, *ck2 = nullptr;
  SecDev sec;                                      // the CycleFold circuit on the device (field Fq, commitments on Grumpkin)
  std::vector<void*> owned;
  char* pin = nullptr; size_t pin_res = 0;         // pinned: 4 MSM results, then staging for the two host-made witnesses
  hipStream_t s2 = nullptr; hipEvent_t ev_fork = nullptr; MsmWorkspace ws2;
  MsmPlan plan_T{}, plan_aug{}, plan_cfW{}, plan_cfT{};
  // The step rows of the NEXT step's cross term need the running pair as folded by this step and the producer's products of this row
  // only — not F' of this step: they are queued on a third stream right behind the fold and run, with their commitment (the one large
  // MSM of a step), under the CycleFold instances and F' on the host.  The verifier rows' share follows F' (k_spmv_cross16 writes it
  // with the products).  *_for = the step a kept result belongs to (-1: none; never kept across calls).
  hipStream_t s3 = nullptr; hipEvent_t ev_fold = nullptr, ev_ts = nullptr; MsmWorkspace ws3; char* pin_ts = nullptr;
  MsmPlan plan_Ts{}, plan_Tv{};
  int64_t t_step_for = -1, t_ver_for = -1;
  BaseTables tb_ck2{};            // window tables of the head of ck_cyclefold: the CycleFold instances' small MSMs only add window sums
  uint32_t *Zl = nullptr, *azl = nullptr, *bzl = nullptr, *czl = nullptr;   // the last fresh main instance's vectors (the incoming pair of the next step)
  uint32_t *z3 = nullptr, *az3 = nullptr, *bz3 = nullptr, *cz3 = nullptr;   // the second CycleFold instance of a step (the first uses sec.z2, ...)
  MsmPlan plan_cfW2{}; hipStream_t s4 = nullptr; MsmWorkspace ws4; hipEvent_t ev_z3 = nullptr; char* pin_w2 = nullptr;   // ... and commits to its witness on a stream of its own
  // host state of the recursion
  uint64_t i = 0;
  std::vector<Fe> z0;
  CfMainRelaxed U; G1Aff UW{}, UE{};               // running main instance (commitments also as curve points)
  CfMainFresh u; G1Aff uW{};                       // incoming main instance
  CfRelaxed cfU;
  Fe u_run = Fe::zero(); Fq cf_u_run = Fq::zero();
  bool broken = false;
  double ph_s[CP_COUNT] = {}; uint64_t ph_n[CP_COUNT] = {};
  // what the last step's F' was given and what it returned (VIMZ_IX_LAST_STEP: the oracle-side restatement of the step relation replays it)
  CfMainIn last_in = CfMainIn::zero(); CfMainOut last_out{}; std::vector<Fe> last_zi, last_zn; bool have_last = false;
  // merged proofs that use this prover as their verifier key: freeing it first orphans them (buffers released, later calls fail cleanly)
  std::vector<struct vimz_cf_merged*> merged_dependents;
  void (*orphan_merged)(vimz_cf*) = nullptr;
};

namespace {

G1Aff g1_identity() { G1Aff p; p.x = Fq::zero(); p.y = Fq::zero(); return p; }
G2Aff g2_identity() { G2Aff p; p.x = Fe::zero(); p.y = Fe::zero(); return p; }
// P + (2^128 + low)·Q on BN254 G1 (host)
G1Aff g1_fold(const G1Aff& P, const uint32_t low[4], const G1Aff& Q) {
  const uint32_t k[5] = {low[0], low[1], low[2], low[3], 1u};
  G1 a = from_affine(P);
  if (aff_is_identity(P)) a = G1::identity();
  if (!aff_is_identity(Q)) { G1 t = scalar_mul(Q, k, 129); add_full(a, t); }
  return to_affine(a);
}
Fe cf_r_element_fr(const uint32_t low[4]) { return rho_element<Fe>(low); }

template <class F>
bool fetch(hipStream_t s, const uint32_t* d, size_t idx, size_t n, F* out) {
  return hipMemcpyAsync(out, d + 8 * idx, 32 * n, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
}

// One CycleFold instance: witness, commitments, challenge, fold into the running CycleFold instance.  which = 1 / 2.
}  // namespace
