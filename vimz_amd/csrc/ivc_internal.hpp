// Internals of the Nova IVC object shared by ivc.hip (RecursiveSNARK::{new, prove_step, verify}) and spartan.hip
// (CompressedSNARK::{prove, verify}).
#pragma once
#include <memory>
#include "prover_internal.hpp"
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include "aug/export.hpp"

using namespace vz::aug;
typedef Affine<Fe> G2Aff;          // Grumpkin point: coordinates in BN254 Fr
typedef XYZZ<Fe> G2;

struct SecDev {                    // secondary circuit on the device (field BN254 Fq, commitments on Grumpkin)
  CsrDev A{}, B{}, C{};
  const uint32_t* dict = nullptr;
  const uint32_t* long_items = nullptr; uint32_t n_long = 0, n_med = 0;
  uint32_t n_w = 0, n_c = 0;
  uint32_t *Zrun = nullptr, *E = nullptr, *AZ = nullptr, *BZ = nullptr, *CZ = nullptr;   // running instance
  uint32_t *z2 = nullptr, *az2 = nullptr, *bz2 = nullptr, *cz2 = nullptr, *T = nullptr;  // fresh instance, cross term
  uint32_t* bad = nullptr;
};

template <class To, class From>
To cross_field(const From& m) {     // the same integer (< both primes) as an element of the other field
  From c = From::from_mont(m);
  To t; for (int k = 0; k < 8; k++) t.v[k] = c.v[k];
  return To::to_mont(t);
}
template <class F>
F rho_element(const uint32_t low[4]) { F c = F::zero(); for (int k = 0; k < 4; k++) c.v[k] = low[k]; c.v[4] = 1; return F::to_mont(c); }

template <class F>
static void sec_spmv(const SecDev& S, hipStream_t s, const uint32_t* z, uint32_t* az, uint32_t* bz, uint32_t* cz) {
  hipLaunchKernelGGL(k_spmv3<F>, dim3(stream_grid(3 * (size_t)S.n_c)), dim3(256), 0, s, S.A, S.B, S.C, S.dict, (size_t)S.n_c, z, az, bz, cz);
  if (S.n_long) {
    hipLaunchKernelGGL(k_spmv_long<F>, dim3(spmv_long_blocks(S.n_long, S.n_med)), dim3(256), 0, s, S.A, S.B, S.C, S.dict, S.long_items, S.n_long, S.n_med, z, az, bz, cz);
  }
}

enum { IP_SYNTH1 = 0, IP_SYNTH2, IP_WAIT_SEC, IP_WAIT_PRI, IP_LAUNCH, IP_PRODUCER, IP_RESERVED, IP_TOTAL, IP_COUNT };

struct vimz_ivc {
  vimz_ctx* ctx = nullptr;
  std::unique_ptr<vimz_circuit> circ1;            // the step circuit's copy, with the verifier circuit appended
  std::unique_ptr<AugCircuit<BnFr>> c1;
  AugCircuit<BnFq> c2;
  vimz_prover* pri = nullptr;
  const vimz_bases *ck1 = nullptr, *ck2 = nullptr;
  SecDev sec;
  std::vector<void*> owned;
  char* pin = nullptr;                            // pinned: 5 MSM results, then staging for the two host-made witnesses
  size_t pin_res = 0, pin_totals = 0;
  MsmPlan plan_aug{}, plan_T1v{}, plan_W2{}, plan_T2{};
  // the witness commitment and the cross-term commitment of one instance are independent: they run side by side
  hipStream_t s2 = nullptr; hipEvent_t ev_fork = nullptr; MsmWorkspace ws2;
  // the step rows of the primary cross term need the folded running instance and the producer's products only — not the
  // verifier circuit of their step: they are queued on a third stream right behind the previous fold and run under the
  // secondary half of that step and the host's verifier circuit
  hipStream_t s3 = nullptr; hipEvent_t ev_fold = nullptr; MsmWorkspace ws3;
  // The step rows' cross term of step i (against the running instance of step i, or — lookahead — of step i-1) lives in slot i & 1:
  // its vector, the pinned window sums and plan of its commitment, the events of that MSM.  step = the step it belongs to (-1: none).
  // Boolean-row form (r1cs_ops.hpp: bool_row_masked; VERDICT r5 #2a): the MSM takes `bufm` — half the points of `buf` — and the commitment is completed on the
  // host by u_at·S_1(row) − ca_at, with u_at / ca_at = the running instance's u and C_A = Σ_{i<n_bool} AZ[i]·ck_i at the time the vector was computed.
  struct T1Slot { int64_t step = -1; bool hasB = false; uint32_t* buf = nullptr; char* pin = nullptr; MsmPlan plan{}; hipEvent_t done = nullptr; hipEvent_t* ev = nullptr;
                  uint32_t* bufm = nullptr; bool tricked = false; Fe u_at = Fe::zero(); G1 ca_at = G1::identity(); };
  bool bool_rows = false;          // the boolean-row form is in use (step circuits built here: their boolean rows come first; VIMZ_IVC_BOOL_ROWS=0 turns it off)
  G1 CA = G1::identity(); bool ca_valid = false;      // C_A of the running instance, kept by linearity (C_A += rho·S_1 per fold); recomputed by one MSM when not valid
  T1Slot t1[2];
  hipEvent_t ev_alt[7] = {};       // profiling events of slot 1 (slot 0 uses the context's)
  char* pin_t1b = nullptr;         // pinned window sums (+ totals) of slot 1
  bool lookahead = false;          // step i+2's cross term against the running instance of step i+1 (VIMZ_IVC_LOOKAHEAD=1); off with MSM helpers
  Fe rho_prev = Fe::zero(); uint32_t rho_prev_low[4] = {};      // the previous step's folding challenge
  bool broken = false;             // a step failed after its folds were queued (cannot happen with an honest witness): no further folds
  hipEvent_t ev_fused = nullptr; bool fused_recorded = false;      // the fused fold + cross term of the step rows on stream 3 (k_fold_cross)
  std::unique_ptr<aug::Worker> chain_w[2];                         // the two circuits' scalar-multiplication chains (they never run at the same time)
  hipEvent_t ev_b0 = nullptr, ev_b1 = nullptr;   // profiling: GPU time of the secondary half on the main stream
  hipEvent_t ev_a = nullptr;                      // the primary half's results on the main stream are back
  // window tables (2^(7w)·P_i) of the three base slices the per-step small MSMs run over: verifier wires and verifier rows of
  // ck1, the head of ck2.  Their window sums only need adding: no 254 doublings on the host per commitment (23 MB each)
  BaseTables tb_aug{}, tb_T1v{}, tb_ck2{};
  // host state of the recursion
  uint64_t i = 0;
  std::vector<Fe> z0;                             // the initial state the chain starts from (absorbed by every instance hash)
  std::vector<Fq> z0_sec{Fq::zero()};             // the secondary's trivial step circuit starts from [0]
  RelaxedInst<Fe> U2;       // running secondary instance (commitments on Grumpkin), as the primary circuit sees it
  RelaxedInst<Fq> U1;       // running primary instance (commitments on BN254 G1), as the secondary circuit sees it
  FreshInst<Fe> u2;         // last fresh secondary instance
  G2Aff T2;                 // commitment to the cross term of (U2, u2)
  Fe u1_run = Fe::zero(); Fq u2_run = Fq::zero();   // the running scalars in the fields their vectors live in
  bool pending_sec = false, sec_T_valid = false;
  double ph_s[IP_COUNT] = {}; uint64_t ph_n[IP_COUNT] = {};
  // Helper GPUs for the one large MSM of a step ("the per-step dense MSM(T) is split by base range across the G GPUs ...
  // each returns one partial point, host adds"): helper h holds a replica of ck_primary, receives its slice of the cross term
  // (device-to-device copy on its own stream) and commits to it; the host adds the partial commitments (vimz_ivc_add_msm_helper).
  struct MsmHelper { vimz_ctx* ctx = nullptr; const vimz_bases* ck = nullptr; hipStream_t s = nullptr; MsmWorkspace ws; uint32_t* T = nullptr; void* pin = nullptr; MsmPlan plan{}; size_t off = 0, n = 0; };
  std::vector<MsmHelper> helpers; hipEvent_t ev_T = nullptr; size_t t1_main_n = 0;
  // one set of a merged proof's buffers (merge.hip) kept from the last one that was freed: the next vimz_ivc_merged_create over this
  // IVC needs no allocation (device + pinned: 0.5-2 ms next to running kernels, inside a timed fold_input)
  // (two sets: a rank of a sharded proof holds its own merged proof and, while it folds another rank's in, that one's copy —
  //  a hipMalloc of 50 MB next to four processes' kernels was measured at 30-60 ms inside a 20-row timed window)
  enum { MERGED_SPARES = 2 };
  uint32_t* merged_spare_dev[MERGED_SPARES] = {nullptr, nullptr}; void* merged_spare_pin[MERGED_SPARES] = {nullptr, nullptr};
  // hand-over of merged proofs between the processes of a node (vimz_ivc_merged_share / _open_shared): a merged proof's device
  // allocation is recycled from proof to proof (the spare set above), so its IPC handle is made once (export side) and the other
  // process's mapping of it is kept open (import side: at most IPC_MAPPINGS, oldest closed first) — the second and later proofs of a
  // pair of ranks pay no hipIpcGetMemHandle / hipIpcOpenMemHandle (1-3 ms each next to running kernels)
  enum { IPC_MAPPINGS = 8 };
  // gen: the GENERATION of an exported allocation — (exporter's pid << 32) | a counter, new for every allocation that is exported; it travels in the
  // ticket, and an importer whose cached mapping carries the same handle bytes under another generation (the exporter freed that allocation and the
  // runtime reissued the handle for a new one) closes the stale mapping and opens the new allocation instead of reading freed memory through the old one
  struct IpcExport { const void* dev = nullptr; unsigned char handle[64]; uint64_t gen = 0; };
  struct IpcMapping { unsigned char handle[64]; void* ptr = nullptr; uint64_t gen = 0; };
  std::vector<IpcExport> ipc_exports; std::vector<IpcMapping> ipc_mappings;
  // the merged proofs that use this IVC as their verifier key: freeing the IVC first orphans them (their buffers are released, every
  // later call on them fails cleanly) instead of leaving them with a dangling pointer
  std::vector<struct vimz_ivc_merged*> merged_dependents;
  void (*orphan_merged)(vimz_ivc*) = nullptr;
  // CompressedSNARK (spartan.hip): transposed shapes, scratch — built on first use, released with the IVC
  void* spartan_cache = nullptr; void (*spartan_free)(vimz_ivc*) = nullptr;
};

