/* vimz_hip.h — C ABI of the MI355X (gfx950) Nova-folding accelerator for VIMz.
 *
 * This is the drop-in boundary: the entry points a patched nova-snark 0.23.0 / nova-scotia 0.5.0 would
 * bind over FFI in place of its CPU arithmetic (INTEGRATION.md shows the Rust stubs).
 * The reference has no FFI seam today — the seam is cut inside the third-party crates reached from
 *   vimz/src/nova_snark_backend/folding.rs:22-23  (load_r1cs, create_public_params)
 *   vimz/src/nova_snark_backend/folding.rs:35-41  (create_recursive_circuit -> RecursiveSNARK::prove_step)
 *   vimz/src/nova_snark_backend/folding.rs:53-55  (RecursiveSNARK::verify)
 * Each function below names the crate interface it replaces.
 *
 * Conventions
 *  - Plain C: opaque handles, pointers and sizes.  No C++/torch types cross this boundary.
 *  - A field element is 4 x uint64_t little-endian limbs (32 bytes).  `form` says whether a buffer holds
 *    canonical integers (VIMZ_FORM_CANONICAL, what .r1cs/.wtns/JSON carry) or Montgomery residues with
 *    R = 2^256 (VIMZ_FORM_MONTGOMERY, the in-memory layout of halo2curves / pasta_curves field types, so
 *    Rust slices can be passed without conversion).
 *  - An affine point is {x, y} = 8 limbs (64 bytes); the identity is (0, 0).
 *  - The caller owns every host buffer; the library owns device memory behind handles (explicit *_free).
 *  - Every function returns VIMZ_OK (0) or a negative error code and never throws or aborts;
 *    vimz_last_error() returns a description for the calling context.
 *  - One vimz_ctx per GPU.  Entry points are thread-safe (serialised per context); work is issued on the
 *    context's own HIP stream.
 *  - Results are exact field / group elements: outputs are bit-identical to the CPU reference.
 */
#ifndef VIMZ_HIP_H
#define VIMZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIMZ_OK 0
#define VIMZ_ERR_INVALID (-1)  /* bad argument */
#define VIMZ_ERR_HIP (-2)      /* HIP runtime failure (see vimz_last_error) */
#define VIMZ_ERR_NO_DEVICE (-3)/* no gfx950 device / extension cannot run */
#define VIMZ_ERR_UNSAT (-4)    /* witness does not satisfy the step relation */

#define VIMZ_FORM_CANONICAL 0
#define VIMZ_FORM_MONTGOMERY 1

/* curves: coordinate field / scalar field */
#define VIMZ_CURVE_BN254_G1 0 /* Fq / Fr : primary curve of the reference (nova_snark_backend/mod.rs:19) */
#define VIMZ_CURVE_GRUMPKIN 1 /* Fr / Fq : secondary curve (nova_snark_backend/mod.rs:20) */
#define VIMZ_CURVE_PALLAS 2
#define VIMZ_CURVE_VESTA 3

#define VIMZ_FIELD_BN254_FR 0
#define VIMZ_FIELD_BN254_FQ 1
#define VIMZ_FIELD_PALLAS_FP 2
#define VIMZ_FIELD_VESTA_FQ 3

typedef struct vimz_ctx vimz_ctx;
typedef struct vimz_bases vimz_bases; /* commitment key resident in HBM */
typedef struct vimz_vec vimz_vec;     /* vector of field elements resident in HBM (Montgomery form) */

/* ---- context ------------------------------------------------------------------------------------- */
int vimz_ctx_create(int device, vimz_ctx** out);
void vimz_ctx_destroy(vimz_ctx* ctx);
const char* vimz_last_error(const vimz_ctx* ctx);
const char* vimz_version(void);
/* name: caller buffer; cus / hbm_bytes may be NULL */
int vimz_device_info(vimz_ctx* ctx, char* name, size_t name_len, int* cus, uint64_t* hbm_bytes);
int vimz_sync(vimz_ctx* ctx);

/* A marker for a profiler's kernel trace: an empty kernel `k_trace_marker` of `id` (1..1024) workgroups on the context's stream, waited for
 * (bench.py brackets its timed region with ids 1 and 2; tools/trace_busy.py cuts the trace there). */
int vimz_trace_marker(vimz_ctx* ctx, int id);
/* Rows of a short fold call whose Poseidon chains are evaluated on the host (the "head batch"; the proof is bit-identical either way): rows >= 0 pins
 * the number for every later call of this process (0: every row's witness entirely on the GPU), -1 restores the library's policy (24 for calls of at
 * most 24 rows — for the segments of one proof: proofs of at most 28 rows in all — on six or more host cores, else 0).  Returns the previous setting. */
long vimz_set_head_rows(long rows);
/* A fingerprint of the host a benchmark line was measured on: out[0] = µs per Poseidon permutation (t = 9) on one host core, out[1] = µs per
 * empty kernel launch + stream synchronise (median of 200), out[2] = host cores this process may use, out[3] = µs per event record + synchronise. */
int vimz_host_fingerprint(vimz_ctx* ctx, double out[4]);
/* Stream-ordered timing with HIP events on the context's stream (used by bench.py for the roofline). */
int vimz_timer_start(vimz_ctx* ctx);
int vimz_timer_stop(vimz_ctx* ctx, float* ms_out);
/* When enabled, vimz_msm* record per-kernel HIP-event durations, readable with vimz_msm_last_profile. */
int vimz_set_profiling(vimz_ctx* ctx, int enabled);
/* ms[6] = {hist, scan, scatter, accumulate, combine, reduce}; info[4] = {window bits, windows, sub-buckets, entries} */
int vimz_msm_last_profile(vimz_ctx* ctx, float ms[6], uint32_t info[4]);
/* sums over every profiled MSM of the context since the last reset (the large MSM(T) launches of a fold, on the stream they run on):
 * ms[6] as above; counts[3] = {MSM calls, points, bucket entries (= mixed additions)} */
int vimz_msm_profile_totals(vimz_ctx* ctx, double ms[6], uint64_t counts[3], int reset);

/* ---- commitment key (replaces the `ck: Vec<G::PreprocessedGroupElement>` of nova-snark's
 *      CommitmentKey, built by PublicParams::setup reached from folding.rs:23) ------------------------- */
int vimz_bases_upload(vimz_ctx* ctx, int curve, const uint64_t* xy, size_t n, int form, vimz_bases** out);
/* Derive n generators on the GPU: ck_i = try-and-increment(SHAKE256(label || LE64(i) || LE32(ctr))), see
 * vimz_amd/csrc/ckgen.hpp (the role of nova-snark's `CommitmentKey::setup(b"ck", n)`; not byte-compatible with it). */
int vimz_bases_generate(vimz_ctx* ctx, int curve, const char* label, size_t label_len, size_t n, vimz_bases** out);
int vimz_bases_download(vimz_ctx* ctx, const vimz_bases* b, size_t offset, uint64_t* xy, size_t n, int form);
/* Precompute window tables T_j[i] = 2^(window_bits*j) * P_i (window_bits 0 -> 16; K x the key's footprint in HBM).  The key
 * is fixed for a whole proof, so every later MSM over it (with window_bits = 0) uses them:
 *   window_bits 12..16: ONE bucket set shared by all windows — fewer digits per scalar, one bucket reduction, no Horner;
 *   window_bits 11 (the window a large MSM uses anyway): the usual per-window bucket sets, whose sums the host then only adds
 *   (no Horner: ~0.1 ms less on the host per MSM).  MSMs too small for that window ignore such tables.
 *   window_bits 7 (keys of at most 30720 points): EVERY multiple m·2^(7w)·P_i, m = 1..64 (189 KB per point): a digit selects its point,
 *   the MSM is one sum without buckets — what the per-step commitments over the verifier circuits' fixed key slices use.
 * Results are unchanged. */
int vimz_bases_precompute(vimz_ctx* ctx, vimz_bases* b, int window_bits);
size_t vimz_bases_len(const vimz_bases* b);
void vimz_bases_free(vimz_ctx* ctx, vimz_bases* b);

/* ---- input pipeline: pixel packing on the device (replaces pyvimz's compress_by_rows / compress_by_blocks,
 *      pyvimz/pyvimz/img/ops.py:4-70, whose hex strings `VIMzInput` carries, vimz/src/input.rs:9-62).  pixels: height x width x
 *      channels bytes (channels 3 = RGB, 1 = grey).  block = 0: one output row of ceil(width/10) elements per image row;
 *      block = 40: 40 x 40 blocks of 160 elements, blocks row-major (the redact input).  out: vimz_pack_count(...) canonical
 *      elements (4 x u64 each). ---------------------------------------------------------------------------------------------- */
size_t vimz_pack_count(size_t height, size_t width, int block);
int vimz_pack_pixels(vimz_ctx* ctx, const uint8_t* pixels, size_t height, size_t width, int channels, int block, uint64_t* out);

/* ---- device vectors -------------------------------------------------------------------------------- */
int vimz_vec_alloc(vimz_ctx* ctx, int field, size_t n, vimz_vec** out); /* zero-filled */
int vimz_vec_upload(vimz_ctx* ctx, vimz_vec* v, size_t offset, const uint64_t* host, size_t n, int form);
int vimz_vec_download(vimz_ctx* ctx, const vimz_vec* v, size_t offset, uint64_t* host, size_t n, int form);
size_t vimz_vec_len(const vimz_vec* v);
void vimz_vec_free(vimz_ctx* ctx, vimz_vec* v);

/* ---- MSM: replaces `G::vartime_multiscalar_mul(scalars, bases)` behind `CE::commit(ck, v)`
 *      (nova-snark 0.23.0 provider).  Computes sum_i scalars[i] * bases[i]
 *      over the first n bases.  window_bits = 0 lets the library choose.  out_xy: affine, `out_form`. ---- */
int vimz_msm(vimz_ctx* ctx, const vimz_bases* bases, const uint64_t* scalars, size_t n, int form,
             int window_bits, uint64_t out_xy[8], int out_form);
/* same, scalars already resident: elements [offset, offset+n) of v against bases [base_offset, base_offset+n) */
int vimz_msm_vec(vimz_ctx* ctx, const vimz_bases* bases, size_t base_offset, const vimz_vec* v, size_t offset,
                 size_t n, int window_bits, uint64_t out_xy[8], int out_form);

/* flags: VIMZ_MSM_SPLIT_ONES sums the bases of unit scalars with a dedicated tree kernel instead of the bucket sort
 * (witness vectors are ~80 % bits); the result is identical. */
#define VIMZ_MSM_SPLIT_ONES 1
int vimz_msm_vec_ex(vimz_ctx* ctx, const vimz_bases* bases, size_t base_offset, const vimz_vec* v, size_t offset,
                    size_t n, int window_bits, int flags, uint64_t out_xy[8], int out_form);

/* ---- sparse R1CS mat-vec and the cross-term commitment for a CALLER-SUPPLIED shape (SURVEY.md §8b seam 2): replaces
 *      `R1CSShape::multiply_vec(&self, z) -> (Az, Bz, Cz)` and `R1CSShape::commit_T(ck, U1, W1, U2, W2) -> (T, comm_T)` of
 *      nova-snark 0.23.0, reached from NIFS::prove / is_sat* inside RecursiveSNARK::{prove_step, verify}
 *      (vimz/src/nova_snark_backend/folding.rs:35-41, 53-55).  The matrices are handed over as nova-snark holds them: three lists
 *      of (row, col, value) triplets, in any order; they stay resident behind the handle (CSR + coefficient dictionary). ------- */
typedef struct vimz_r1cs vimz_r1cs;
typedef struct { const uint32_t* row; const uint32_t* col; const uint64_t* val /* nnz x 4 limbs */; size_t nnz; } vimz_coo;
/* field: the shape's scalar field (VIMZ_FIELD_*); ncols = length of z = [W, u, X] as the caller orders it */
int vimz_r1cs_upload(vimz_ctx* ctx, int field, size_t nrows, size_t ncols, const vimz_coo* A, const vimz_coo* B, const vimz_coo* C, int form,
                     vimz_r1cs** out);
void vimz_r1cs_free(vimz_ctx* ctx, vimz_r1cs* shape);
/* info = {rows, cols, nnz A, nnz B, nnz C, distinct coefficients, long rows, field} */
int vimz_r1cs_info(const vimz_r1cs* shape, uint64_t info[8]);
/* (Az, Bz, Cz) = (A, B, C)·z; every vector a vimz_vec of the shape's field (z: >= ncols elements, outputs: >= nrows) */
int vimz_spmv3(vimz_ctx* ctx, const vimz_r1cs* shape, const vimz_vec* z, vimz_vec* az, vimz_vec* bz, vimz_vec* cz);
/* T = AZ1∘BZ2 + AZ2∘BZ1 − u1·CZ2 − u2·CZ1 (written to T_out, nrows elements) and comm_T = Σ T_i·ck_i (affine, out_form).
 * z1 / z2: the full assignment vectors of the two instances (running and fresh); u1 / u2: their relaxation scalars in `form`
 * (u2 = 1 for a fresh instance).  ck: at least nrows generators on the curve whose scalar field is the shape's field. */
int vimz_commit_T(vimz_ctx* ctx, const vimz_r1cs* shape, const vimz_bases* ck, const vimz_vec* z1, const uint64_t u1[4], const vimz_vec* z2,
                  const uint64_t u2[4], int form, vimz_vec* T_out, uint64_t comm_T[8], int out_form);
/* R1CSShape::is_sat_relaxed for a resident assignment z = [W, u-slot, X] with relaxation scalar u (in `form`) and error vector E (NULL = 0):
 * bad_rows = number of rows with (A·z)∘(B·z) != u·(C·z) + E, first_bad = the first of them (all ones if none).  nova-snark 0.23.0,
 * reached from RecursiveSNARK::verify (folding.rs:53-55). */
int vimz_r1cs_check_relaxed(vimz_ctx* ctx, const vimz_r1cs* shape, const vimz_vec* z, const uint64_t u[4], int form, const vimz_vec* E,
                            uint64_t* bad_rows, uint64_t* first_bad);
/* x1 <- x1 + r * x2 over the first n elements (r in `form`): the fold of a resident vector — RelaxedR1CSWitness::fold for W and E,
 * nova-snark 0.23.0 (reached from NIFS::prove, folding.rs:35-41).  Both vectors of the same field, x1 != x2. */
int vimz_vec_axpy(vimz_ctx* ctx, vimz_vec* x1, const uint64_t r[4], int form, const vimz_vec* x2, size_t n);

/* ---- step circuits: R1CS shape + witness program (replaces the `.r1cs` that nova_scotia::circom::reader::load_r1cs
 *      reads at vimz/src/nova_snark_backend/folding.rs:22 and the circom witness generator named by
 *      Config::witness_generator_file(), folding.rs:36).  Host-only: usable without a GPU. -------------------- */
typedef struct vimz_circuit vimz_circuit;
/* (all nine step circuits, crop included, have GPU witness programs; circuits loaded from an .r1cs have none and are folded from
 * supplied witnesses with vimz_prover_fold_witness / vimz_ivc_fold_witness.)
 * transformation ids follow the reference's enum order (vimz/src/transformation.rs:7-18) */
#define VIMZ_T_BLUR 0
#define VIMZ_T_BRIGHTNESS 1
#define VIMZ_T_CONTRAST 2
#define VIMZ_T_CROP 3
#define VIMZ_T_GRAYSCALE 4
#define VIMZ_T_HASH 5
#define VIMZ_T_REDACT 6
#define VIMZ_T_RESIZE 7
#define VIMZ_T_SHARPNESS 8
/* width = packed elements per original row (128 HD / 384 4K / 768 8K; 160 for redact blocks);
 * width2 / rows_in / rows_out: resize geometry (64,3,2 at HD; w/2,2,1 at 4K/8K); crop_height: crop only. */
int vimz_circuit_build(int transformation, int width, int width2, int rows_in, int rows_out, int crop_height,
                       vimz_circuit** out);
void vimz_circuit_free(vimz_circuit* c);
const char* vimz_circuit_last_error(void);
/* iden3 binary formats: load a circom-built `.r1cs` (BN254 Fr, as many public outputs as inputs — a Nova step circuit), and
 * read a `.wtns`.  A loaded circuit has no witness program; fold it with vimz_prover_fold_witness. */
int vimz_circuit_load_r1cs(const uint8_t* data, size_t len, vimz_circuit** out);
int vimz_wtns_load(const uint8_t* data, size_t len, uint64_t* out, size_t cap_elems, size_t* n_out);
#define VIMZ_CIRCUIT_INFO_LEN 16
/* info = {wires, constraints (incl. linear), linear constraints, len_z, private inputs, nnz A, nnz B, nnz C,
 *         dictionary size, decomposition groups, lane groups, lane instructions, lane rows, hash jobs, chains, field ops} */
/* Optional, host only: synthesises ahead of time what vimz_ivc_create needs of this circuit — the augmented primary circuit (the step circuit with Nova's
 * verifier circuit appended, nova-snark's `circuit_primary` of `PublicParams::setup`, reached from vimz/src/nova_snark_backend/folding.rs:20-25) and its digest —
 * and keeps it with the circuit object; the IVCs of a proof made as several segments then share one synthesis.  Call it on the thread that built the circuit. */
int vimz_circuit_prepare_ivc(const vimz_circuit* c);
int vimz_circuit_info(const vimz_circuit* c, uint64_t info[VIMZ_CIRCUIT_INFO_LEN]);
/* raw tables (CSR of A,B,C with a coefficient dictionary; witness-program tables of vimz_amd/csrc/circuit/program.hpp) */
#define VIMZ_CX_A_ROWPTR 0
#define VIMZ_CX_A_COL 1
#define VIMZ_CX_A_COEF 2
#define VIMZ_CX_B_ROWPTR 3
#define VIMZ_CX_B_COL 4
#define VIMZ_CX_B_COEF 5
#define VIMZ_CX_C_ROWPTR 6
#define VIMZ_CX_C_COL 7
#define VIMZ_CX_C_COEF 8
#define VIMZ_CX_DICT_MONT 9
#define VIMZ_CX_DICT_CANON 10
#define VIMZ_CX_DECOMP 11
#define VIMZ_CX_LANE_GROUPS 12
#define VIMZ_CX_LANE_INSTR 13
#define VIMZ_CX_LANE_ROWS 14
#define VIMZ_CX_JOBS 15
#define VIMZ_CX_CHAINS 16
#define VIMZ_CX_FOPS 17
#define VIMZ_CX_ZOUT 18
#define VIMZ_CX_LC_TERMS 19
/* returns the table's size in bytes (copies it when buf != NULL and cap is large enough), negative on error */
int64_t vimz_circuit_export(const vimz_circuit* c, int what, void* buf, size_t cap);

/* ---- folding prover: the body of `fold_input` (vimz/src/nova_snark_backend/folding.rs:27-43), i.e. the per-row loop of
 *      nova_scotia::create_recursive_circuit around nova-snark's RecursiveSNARK::prove_step, with W, E and the running
 *      (A,B,C)·Z resident in HBM.  This is the NIFS *accumulator* over the step circuit's own instances, X = (z_{i+1}, z_i):
 *      accumulators of different row segments merge (vimz_prover_merge), which is the multi-GPU sharding of BASELINE.json's
 *      north_star.  RecursiveSNARK::prove_step in full — augmented verifier circuits on both curves — is vimz_ivc_* below. ------------------------ */
typedef struct vimz_prover vimz_prover;
/* ck must be on BN254 G1 with at least max(wires, constraints) generators; max_batch = rows whose witnesses are
 * generated together (memory: max_batch * wires * 32 bytes). */
int vimz_prover_create(vimz_ctx* ctx, const vimz_circuit* circuit, const vimz_bases* ck, size_t max_batch, vimz_prover** out);
void vimz_prover_free(vimz_prover* p);
/* z0: len_z canonical elements (Transformation::ivc_initial_state, vimz/src/transformation.rs:25-40) */
int vimz_prover_reset(vimz_prover* p, const uint64_t* z0);
/* step_inputs: nsteps x private-input-count canonical elements in the per-step order of
 * vimz/src/nova_snark_backend/input.rs:57-96.  Returns VIMZ_ERR_UNSAT if a row violates the step relation. */
int vimz_prover_fold(vimz_prover* p, const uint64_t* step_inputs, size_t nsteps);
/* Same fold, but the witnesses come from outside (circom's generator): witnesses = nsteps x wires canonical elements in
 * iden3 wire order; step i's public inputs must equal step i-1's public outputs (checked). */
int vimz_prover_fold_witness(vimz_prover* p, const uint64_t* witnesses, size_t nsteps);
/* verify_folded_proof (folding.rs:45-56): 0 = accepted; bit0 relation, bit1 comm_W, bit2 comm_E, bit3 running products */
int vimz_prover_verify(vimz_prover* p, uint32_t* result);
int vimz_prover_instance(vimz_prover* p, uint64_t comm_W[8], uint64_t comm_E[8], uint64_t u[4], uint64_t* z_current, uint64_t* steps);
int vimz_prover_running(vimz_prover* p, uint64_t* z_run, uint64_t* E);
/* IVC state chain only: zs_out = (nsteps+1) x len_z canonical elements starting at z_start (one hash-only GPU pass over the
 * rows + the host pair-hash chain).  Lets a multi-GPU driver find the state at which each row segment starts. */
int vimz_prover_state_chain(vimz_prover* p, const uint64_t* z_start, const uint64_t* step_inputs, size_t nsteps, uint64_t* zs_out);
/* The chain in its two parts, for proofs sharded over GPUs: (1) the ROW DIGESTS (the state-independent row hashes: the expensive part,
 * one GPU pass) of any run of rows — every rank hashes its own rows side by side —, (2) the serial chain over rows whose digests are
 * known (host only, ≈ 12-25 µs per row).  stride = elements per row of the digests (opaque Montgomery limbs); 0 for a circuit whose
 * digests depend on the state (crop): use vimz_prover_state_chain there. */
size_t vimz_prover_digest_stride(const vimz_prover* p);
int vimz_prover_row_digests(vimz_prover* p, const uint64_t* step_inputs, size_t nsteps, uint64_t* digests_out /* nsteps x stride x 4 */);
int vimz_prover_chain_from_digests(vimz_prover* p, const uint64_t* z_start, const uint64_t* step_inputs, const uint64_t* digests, size_t nsteps, uint64_t* zs_out);
/* Host-side final fold of row segments folded on different GPUs (north_star: "host-side sequential final fold"):
 * export one prover's running relaxed instance as a byte blob, merge it into another's (NIFS for two relaxed instances). */
size_t vimz_prover_export_size(const vimz_prover* p);
int vimz_prover_export(vimz_prover* p, uint8_t* blob, size_t cap);
int vimz_prover_merge(vimz_prover* p, const uint8_t* blob, size_t len);
/* the same final fold between two provers on the same GPU (segments folded concurrently on one device), no host copy */
int vimz_prover_merge_prover(vimz_prover* p, vimz_prover* src);
/* seconds[9]/counts[9]: witness, state chain (host), spmv, msm(W), cross term, msm(T), RO (host), fold, host EC */
int vimz_prover_profile(const vimz_prover* p, double seconds[9], uint64_t counts[9]);
/* parity hooks: GPU witness generation alone (replaces the circom witness generator process) and
 * the GPU sparse mat-vec alone (R1CSShape::multiply_vec; row V1).  All buffers canonical. */
int vimz_prover_witness(vimz_prover* p, const uint64_t* inputs, size_t rows, uint64_t* z_wires_out, uint64_t* zs_out, uint32_t* status_out);
int vimz_prover_spmv(vimz_prover* p, const uint64_t* z, uint64_t* az, uint64_t* bz, uint64_t* cz);

/* ---- Nova IVC: RecursiveSNARK::{new, prove_step, verify} in full, with the augmented verifier circuits on the
 *      BN254 / Grumpkin cycle (reference entry vimz/src/nova_snark_backend/folding.rs:27-56,
 *      curves nova_snark_backend/mod.rs:19-20).  nova-snark 0.23.0's own circuit is not vendored: the circuits here
 *      (vimz_amd/csrc/aug/) state the same relation — see DESIGN.md §5 — and are NOT byte-compatible with its proof object.
 *
 *      Per step: the step circuit's witness / commitment / (A,B,C)·z come from the batch producer exactly as in
 *      vimz_prover_fold; the verifier circuit's ~7.6 k wires are computed on the host (as in the reference), committed and
 *      multiplied on the GPU; the secondary circuit (7.6 k constraints over BN254 Fq, commitments on Grumpkin) is folded on
 *      the GPU with the same kernels instantiated for that field/curve. ---------------------------------------------------- */
typedef struct vimz_ivc vimz_ivc;
/* ck_primary on BN254 G1 (>= max(wires, constraints) of the augmented step circuit), ck_secondary on Grumpkin (>= 8192). */
int vimz_ivc_create(vimz_ctx* ctx, const vimz_circuit* step_circuit, const vimz_bases* ck_primary, const vimz_bases* ck_secondary,
                    size_t max_batch, vimz_ivc** out);
void vimz_ivc_free(vimz_ivc* v);
int vimz_ivc_reset(vimz_ivc* v, const uint64_t* z0);
/* One proof on several GPUs: every step's large cross-term commitment MSM(T) is split by base range between this
 * IVC's GPU and the registered helpers (at most 7); each returns one partial point, the host adds them.  helper_ctx: a context on
 * another device; ck_on_helper: a replica of ck_primary resident there (same generators).  The proof is unchanged. */
int vimz_ivc_add_msm_helper(vimz_ivc* v, vimz_ctx* helper_ctx, const vimz_bases* ck_on_helper);
/* same inputs as vimz_prover_fold / vimz_prover_fold_witness */
int vimz_ivc_fold(vimz_ivc* v, const uint64_t* step_inputs, size_t nsteps);
int vimz_ivc_fold_witness(vimz_ivc* v, const uint64_t* witnesses, size_t nsteps);
/* RecursiveSNARK::verify(pp, num_steps, z0_primary, z0_secondary) (reached from folding.rs:53-55): the proof must be about exactly
 * `num_steps` steps from the initial state `z0` (len_z canonical elements; the secondary's z0 is the constant [0]); both output
 * hashes (recomputed from the CLAIMED z0), is_sat_relaxed of both running instances, is_sat of the last secondary instance,
 * every commitment re-opened.  result: 0 = accepted; bit 12 the proof's step count or initial state differ from the claimed ones; bit 0/1 hash of the primary/secondary chain; bit 2 primary relaxed relation;
 * bit 3 primary comm_W; bit 4 primary comm_E; bit 5 secondary relaxed relation; bit 6/7 secondary comm_W/comm_E; bit 8 last
 * secondary instance's relation; bit 9 its comm_W; bit 10 instance scalars differ from the witness vectors; bit 11 the running products (A,B,C)·Z kept for the next
 * fold violate the relation (prover-side bookkeeping, not part of the proof). */
int vimz_ivc_verify(vimz_ivc* v, uint64_t num_steps, const uint64_t* z0, uint32_t* result);
/* info[0..11]: steps, primary wires, primary constraints, step wires, step constraints, secondary wires, secondary constraints,
 * len_z, verifier-circuit wires (primary), nnz(A+B+C) primary, nnz secondary, rows of the last fold call whose Poseidon chains ran on the host */
int vimz_ivc_info(const vimz_ivc* v, uint64_t info[12]);
int vimz_ivc_state(const vimz_ivc* v, uint64_t* z_current /* len_z x 4 */, uint64_t* steps);
/* IVC state chain only (as vimz_prover_state_chain): where a row segment proven by another IVC starts */
int vimz_ivc_state_chain(vimz_ivc* v, const uint64_t* z_start, const uint64_t* step_inputs, size_t nsteps, uint64_t* zs_out);
/* the same in its two parts (see vimz_prover_row_digests): what vimz_amd/distributed.py::prove_sharded uses across ranks */
size_t vimz_ivc_digest_stride(const vimz_ivc* v);
int vimz_ivc_row_digests(vimz_ivc* v, const uint64_t* step_inputs, size_t nsteps, uint64_t* digests_out);
int vimz_ivc_chain_from_digests(vimz_ivc* v, const uint64_t* z_start, const uint64_t* step_inputs, const uint64_t* digests, size_t nsteps, uint64_t* zs_out);
/* seconds[8]/counts[8]: verifier-circuit witness primary (host), secondary (host), wait for secondary MSMs, wait for primary MSMs,
 * uploads+launches, producer wait, GPU time of the secondary half on the main stream (only while profiling is on), total */
int vimz_ivc_profile(const vimz_ivc* v, double seconds[8], uint64_t counts[8]);
/* The proof as an object of its own (RecursiveSNARK serialisation; checkpoint / resume): everything vimz_ivc_verify reads and the
 * next vimz_ivc_fold needs.  Import into a vimz_ivc created for the same step circuit and keys, then verify or keep folding. */
size_t vimz_ivc_proof_size(const vimz_ivc* v);
int vimz_ivc_proof_export(vimz_ivc* v, uint8_t* blob, size_t cap);
int vimz_ivc_proof_import(vimz_ivc* v, const uint8_t* blob, size_t len);
/* ---- CompressedSNARK::{setup, prove, verify} (vimz/src/nova_snark_backend/mod.rs:52-67: Spartan relaxed-R1CS SNARK with
 *      inner-product-argument openings over both curves; README.md:196 counts it into the total proof time).  Our own statement of
 *      the construction (vimz_amd/csrc/spartan.hip), NOT byte-compatible with nova-snark's proof object.  The proof covers the two
 *      running instances and the last fresh secondary instance of the IVC and carries (steps, z_0, z_n). ---------------------------- */
size_t vimz_ivc_compressed_size(vimz_ivc* v);                    /* bytes of a compressed proof for this IVC's shapes */
/* prove: blob receives vimz_ivc_compressed_size(v) bytes; seconds (optional) = {setup (first call only), prove} */
int vimz_ivc_compress(vimz_ivc* v, uint8_t* blob, size_t cap, double seconds[2]);
/* verify(vk, num_steps, z0): `v` supplies the verifier key — any vimz_ivc created for the same step circuit and commitment keys;
 * its folding state is neither read nor changed.  result: 0 = accepted; bit 0 / 1 primary / secondary chain hash; bit 2 / 3 / 4 the
 * argument for the primary running / secondary running / last fresh secondary instance; bit 12 statement mismatch; bit 13 malformed. */
int vimz_ivc_verify_compressed(vimz_ivc* v, const uint8_t* blob, size_t len, uint64_t num_steps, const uint64_t* z0, uint32_t* result);

/* KZG opening of a committed, device-resident vector (the `KZG::prove` of Sonobe's decider for the final commitments, reached from
 * vimz/src/sonobe_backend/decider.rs:13-21; calldata words kzg_*: vimz_amd/calldata.py): `srs` = the SRS's G1 powers [τ^i]G as bases
 * (vimz_bases_upload), so that vimz_msm_vec over them commits to p(X) = Σ v_i X^i.  eval_out = p(z); proof_xy = the commitment to
 * (p(X) − p(z)) / (X − z) (affine, `form`; the identity as zeros).  z and eval_out in `form`.  The pairing check belongs to the verifier. */
int vimz_kzg_open(vimz_ctx* ctx, const vimz_bases* srs, size_t base_offset, const vimz_vec* v, size_t offset, size_t n, const uint64_t z[4], int form,
                  uint64_t eval_out[4], uint64_t proof_xy[8]);

/* ---- Nova + CycleFold IVC: the `prove_step` loop of the reference's Sonobe backend (vimz/src/sonobe_backend/folding.rs:52-66; scheme
 *      `Nova<G1, G2, C, KZG<Bn254>, Pedersen<G2>, false>`, folding.rs:22) — SURVEY.md §8 row N1.  The main circuit F' (step circuit + our
 *      statement of CycleFold's augmented circuit, vimz_amd/csrc/aug/cyclefold.hpp) is committed on BN254 G1 — ck_main may be a KZG SRS's
 *      G1 powers (vimz_bases_upload) or any other generators —, the CycleFold circuit (1.4 k constraints over Fq) on Grumpkin.  Sonobe is
 *      not vendored with the reference: NOT byte-compatible with its proof object, and the decider (Groth16 + KZG, decider.rs:13-21) is
 *      not built; the calldata layout of its output is vimz_amd/calldata.py.  One proof = (U_n, W_n), (u_n, w_n), (cfU_n, cfW_n). ---------- */
typedef struct vimz_cf vimz_cf;
int vimz_cf_create(vimz_ctx* ctx, const vimz_circuit* step_circuit, const vimz_bases* ck_main, const vimz_bases* ck_cyclefold, size_t max_batch, vimz_cf** out);
void vimz_cf_free(vimz_cf* v);
int vimz_cf_reset(vimz_cf* v, const uint64_t* z0);
/* Folding::prove_step for every row (same inputs as vimz_ivc_fold) */
int vimz_cf_fold(vimz_cf* v, const uint64_t* step_inputs, size_t nsteps);
/* Folding::verify(vp, ivc_proof) (folding.rs:69-75) for the claimed statement: 0 = accepted; bit 0 / 1 the hash of the main / CycleFold running
 * instance carried by the last instance of F'; bit 2 main relaxed relation; bit 3 / 4 main comm_W / comm_E; bit 5 CycleFold relaxed relation;
 * bit 6 / 7 its comm_W / comm_E; bit 8 the last instance of F' (strict); bit 9 its comm_W; bit 10 instance scalars differ from the witness
 * vectors; bit 11 prover-side running products; bit 12 the proof's step count or initial state differ from the claimed ones. */
int vimz_cf_verify(vimz_cf* v, uint64_t num_steps, const uint64_t* z0, uint32_t* result);
/* info[0..12): steps, main wires, main constraints, step wires, step constraints, CycleFold wires, CycleFold constraints, len_z,
 * F' wires, nnz main, nnz CycleFold, public elements of a CycleFold instance */
int vimz_cf_info(const vimz_cf* v, uint64_t info[12]);
int vimz_cf_state(const vimz_cf* v, uint64_t* z_current, uint64_t* steps);
/* seconds[8]/counts[8]: cross term + MSM(T), the two CycleFold instances, F' on the host, fresh instance (upload, verifier rows,
 * commitment), producer wait, total, and — parts of the fresh-instance phase — the wait for the producer's row, for the verifier wires' commitment */
int vimz_cf_profile(const vimz_cf* v, double seconds[8], uint64_t counts[8]);
/* side 0 = main circuit, 1 = CycleFold circuit; what = VIMZ_CX_* (R1CS tables), VIMZ_IX_INFO, VIMZ_IX_INSTANCE (side 0: comm_W.x, comm_W.y,
 * comm_E.x, comm_E.y, u, x0, x1; side 1: comm_W.x, comm_W.y, comm_E.x, comm_E.y, u, x[0..7)), VIMZ_IX_FRESH_INSTANCE (side 0: comm_W.x,
 * comm_W.y, x0, x1), VIMZ_IX_PARAMS (side 0: digest, z0..., z_i...), VIMZ_IX_RUNNING_Z, VIMZ_IX_RUNNING_E, VIMZ_IX_FRESH_Z (side 0),
 * VIMZ_IX_LAST_STEP (side 0: inputs and outputs of the last step's F', for an independent restatement of the step relation) */
int64_t vimz_cf_export(vimz_cf* v, int side, int what, void* buf, size_t cap);
/* KZG openings of the running main instance's commitments at z (vimz_kzg_open over ck_main as the SRS): which = 0 comm_W (coefficients = the witness
 * wires [1, wires - 2) of the running vector), 1 comm_E.  Canonical in and out. */
int vimz_cf_kzg_open(vimz_cf* v, int which, const uint64_t z[4], uint64_t eval_out[4], uint64_t proof_xy[8]);
/* The proof as an object of its own (Sonobe's `ivc_proof()` / `from_ivc_proof`; checkpoint / resume): everything vimz_cf_verify reads and the
 * next vimz_cf_fold needs.  Import into a vimz_cf created for the same step circuit and keys, then verify or keep folding. */
size_t vimz_cf_proof_size(const vimz_cf* v);
int vimz_cf_proof_export(vimz_cf* v, uint8_t* blob, size_t cap);
int vimz_cf_proof_import(vimz_cf* v, const uint8_t* blob, size_t len);
/* ONE proof object out of several row segments' CycleFold proofs (the north_star's "host-side sequential final fold" for this scheme; what
 * vimz_ivc_merge* is for the Nova IVC): segments folded concurrently — each a vimz_cf of its own on the same device — are folded in row order
 * by out-of-circuit NIFS: the running main instances pairwise (two relaxed instances), each segment's last instance of F' (strict), the
 * running CycleFold instances pairwise on Grumpkin.  The verifier replays the segment records (hashes from each segment's statement,
 * adjacency, folds of the instances) and checks ONE main and ONE CycleFold relaxed instance.  Protocol: vimz_amd/csrc/cyclefold_merge.hip, ours. */
typedef struct vimz_cf_merged vimz_cf_merged;
/* the merged proof of one segment; `first` is left unchanged, supplies shapes / keys / context and must outlive the object */
int vimz_cf_merged_create(vimz_cf* first, vimz_cf_merged** out);
void vimz_cf_merged_free(vimz_cf_merged* m);
/* fold the proof of the NEXT row segment in (same device; read in place, left unchanged): it must start at the state m ends in */
int vimz_cf_merge(vimz_cf_merged* m, vimz_cf* next_segment);
/* fold another merged object — ONE run of segments, e.g. what another GPU made of its rows (north_star's sharding) — in, as a whole: two relaxed
 * accumulators on each side.  It must start at the state m ends in; read in place, left unchanged; afterwards m takes no more single segments. */
int vimz_cf_merge_merged(vimz_cf_merged* m, vimz_cf_merged* other);
/* the object as bytes (records + the folded witnesses and running products), and back — into the context of `vk`, any vimz_cf for the same step
 * circuit and keys; untrusted input: ranges, curve membership, the accumulator recomputed from the records */
size_t vimz_cf_merged_size(const vimz_cf_merged* m);
int vimz_cf_merged_save(vimz_cf_merged* m, uint8_t* blob, size_t cap);
int vimz_cf_merged_load(vimz_cf* vk, const uint8_t* blob, size_t len, vimz_cf_merged** out);
/* the hand-over between two processes of one node without the host round trip, as vimz_ivc_merged_share / _open_shared: a ticket (records + HIP IPC
 * handle), a device-to-device copy of the ten vectors on the receiving side */
int64_t vimz_cf_merged_share(vimz_cf_merged* m, void* ticket, size_t cap);
int vimz_cf_merged_open_shared(vimz_cf* vk, const uint8_t* ticket, size_t len, vimz_cf_merged** out);
/* result: 0 = accepted; bit 0 / 1 a segment's main / CycleFold hash; bit 2 / 3 / 4 main relaxed relation / comm_W / comm_E; bit 5 / 6 / 7 the
 * same of the CycleFold instance; bit 10 instance scalars differ from the vectors; bit 12 statement (step count, initial state, adjacency);
 * bit 13 the stored folded instances differ from the replay of the records */
int vimz_cf_merged_verify(vimz_cf_merged* m, uint64_t num_steps, const uint64_t* z0, uint32_t* result);
/* info[0..8): steps, segments, len_z, main wires, main constraints, CycleFold wires, CycleFold constraints, broken */
int vimz_cf_merged_info(const vimz_cf_merged* m, uint64_t info[8]);
int vimz_cf_merged_state(const vimz_cf_merged* m, uint64_t* z_start, uint64_t* z_end, uint64_t* steps);
int vimz_cf_merged_profile(const vimz_cf_merged* m, double seconds[4]);   /* cross terms + commitments, folds, host, total */
/* the statement part as canonical little-endian 64-bit words (layout: cyclefold_merge.hip); returns the byte size */
int64_t vimz_cf_merged_records(const vimz_cf_merged* m, void* buf, size_t cap);
/* side 0 / 1; what = VIMZ_IX_RUNNING_Z, VIMZ_IX_RUNNING_E, VIMZ_IX_INSTANCE (side 0: 7 elements, side 1: 12 — as vimz_cf_export) of the folded instances */
int64_t vimz_cf_merged_export(vimz_cf_merged* m, int side, int what, void* buf, size_t cap);
/* KZG openings (vimz_kzg_open over ck_main as the SRS) of the FOLDED main instance: which = 0 comm_W, 1 comm_E.  For a merged proof of one segment the
 * folded instance is U_{i+1} = NIFS(U_i, u_i), the one Sonobe's decider opens (decider.rs:13-21).  Canonical in and out. */
int vimz_cf_merged_kzg_open(vimz_cf_merged* m, int which, const uint64_t z[4], uint64_t eval_out[4], uint64_t proof_xy[8]);
/* IVC state chain only (as vimz_ivc_state_chain): the state at which a row segment proven by another vimz_cf starts */
int vimz_cf_state_chain(vimz_cf* v, const uint64_t* z_start, const uint64_t* step_inputs, size_t nsteps, uint64_t* zs_out);
/* ... in its two parts, as vimz_ivc_row_digests / vimz_ivc_chain_from_digests */
size_t vimz_cf_digest_stride(const vimz_cf* v);
int vimz_cf_row_digests(vimz_cf* v, const uint64_t* step_inputs, size_t nsteps, uint64_t* digests_out);
int vimz_cf_chain_from_digests(vimz_cf* v, const uint64_t* z_start, const uint64_t* step_inputs, const uint64_t* digests, size_t nsteps, uint64_t* zs_out);
/* (test hooks — vimz_cf_poke, the host-only self-checks — are declared in vimz_hip_testing.h and exist only in libvimz_hip_testing.so) */

/* ---- the decider of the Nova + CycleFold path: `Decider::preprocess` / `Decider::prove` / `Decider::verify` of the Sonobe backend
 *      (vimz/src/sonobe_backend/mod.rs:72-80; `DeciderEth<.., Groth16<Bn254>, ..>`, decider.rs:13-21; verify_final_proof, decider.rs:31-50) — the 25 words
 *      `contracts/<T>Verifier.sol::verifyOpaqueNovaProofWithInputs` takes (ContrastVerifier.sol:785-810; solidity.rs:13-27).  Groth16 over BN254 as
 *      published, with the PUBLIC-INPUT LAYOUT of the reference's contracts (pp_hash, i, z_0, z_i, the folded commitments and cmT as 5 x 55-bit
 *      limbs per coordinate, the KZG challenges and evaluations: ContrastVerifier.sol:700-772) — pinned: tests/_novadecider.py restates the
 *      contract, accepts the reference's six committed proofs with the reference's keys, and accepts this library's words with this library's
 *      key.  The circuit's CONSTRAINTS (vimz_amd/csrc/aug/decider.hpp, decider_cf.hpp) are ours, parity unpinned: Sonobe's circuit and keys come out
 *      of crates that are not vendored, so the committed `.proof` bytes cannot be reproduced.  Two variants, as in the reference: the FULL decider
 *      (decider.rs:13-21: also opens the running CycleFold instance's two commitments and checks its relaxed relation inside the circuit — natively over
 *      Grumpkin, non-natively over Fq; ≈ 2.75 M constraints on top of the step circuit's) and the LIGHT one (the opt-in `light-test` feature,
 *      vimz/Cargo.toml:56-59, vimz/Makefile:1-2, the contracts under contracts/light-test/: the CycleFold instance bound by its hash only).  Same public inputs, same
 *      25 words.  NTTs, the G1 / G2 multi-scalar multiplications, the final fold, the KZG openings and the keys' fixed-base multiplications run on
 *      the GPU; verification is host code (pairing.hpp).  The 25-word layout is the reference contracts' INTERFACE, not by itself a sound on-chain
 *      decider: U_i's and u_i's commitments and the fold's r reach the contract as unconstrained calldata (aug/decider.hpp). ---------- */
typedef struct vimz_decider vimz_decider;
/* KZG::setup (inside `prepare_folding`, vimz/src/sonobe_backend/folding.rs:36-48): srs = [tau^i]G1 for i < n as a commitment key (use it as
 * ck_main of vimz_cf_create), vk_g2_out = [tau]G2 (x.c0, x.c1, y.c0, y.c1 canonical).  tau comes from the OS's randomness and is forgotten. */
int vimz_kzg_setup(vimz_ctx* ctx, size_t n, vimz_bases** srs_out, uint64_t vk_g2_out[16]);
/* Decider::preprocess.  prover: supplies shapes, keys and context (must outlive the object).  kzg_vk_g2 (optional): [tau]G2 of the SRS the prover's
 * ck_main is made of (needed by vimz_decider_verify; part of vimz_decider_vk; checked against that SRS: e(srs[1], G2) = e(G1, [tau]G2)).
 * light: 0 = the full decider (the reference's default), non-zero = the `light-test` variant.  The full decider bakes the first generators of the
 * prover's CycleFold commitment key into the circuit.  The Groth16 trapdoor comes from the OS's randomness and is wiped after use (a locally trusted
 * setup; seeded test setups exist only in libvimz_hip_testing.so).  seconds (optional) = {circuit synthesis, QAP evaluation at the trapdoor, key
 * points on the GPU, total} */
int vimz_decider_setup(vimz_cf* prover, const uint64_t kzg_vk_g2[16], int light, vimz_decider** out, double seconds[4]);
void vimz_decider_free(vimz_decider* d);
/* info = {constraints, wires, public inputs (36 + 2 len_z), domain size, non-zeros of A, B, C, rows of the CycleFold checks (0: light decider)} */
int vimz_decider_info(const vimz_decider* d, uint64_t info[8]);
/* the verifying key — the constants of a contract generated for this circuit — as canonical words: pp_hash (4), len_z (1), alpha (G1: x, y), beta,
 * gamma, delta (G2: x.c0, x.c1, y.c0, y.c1), the number of IC points, the IC points, KZG G_1 (G1), G_2, VK (G2); returns the byte size (copies
 * when cap suffices) */
int64_t vimz_decider_vk(const vimz_decider* d, void* buf, size_t cap);
/* The key pair at rest (bytes; layout: vimz_amd/csrc/groth16.hip): a set-up made once per circuit, or keys made elsewhere — a ceremony's, converted to this
 * layout: the library then never sees a trapdoor.  _save returns the byte size (copies when cap suffices); _load checks sizes, the public-parameter hash and the
 * verifying part's points, and takes the queries as they are (a loaded key is trusted like any common reference string). */
int64_t vimz_decider_key_save(vimz_decider* d, void* buf, size_t cap);
int vimz_decider_key_load(vimz_cf* prover, const void* buf, size_t len, vimz_decider** out);
/* Decider::prove for the IVC proof `ivc` holds (same shapes and keys as the decider's prover; left unchanged; at least one step): final fold, KZG
 * openings, Groth16 proof.  words_out: the 25 calldata words (vimz_amd/calldata.py names them), public_out: the info[2] public inputs; canonical,
 * 4 little-endian limbs each.  VIMZ_ERR_UNSAT when the proof does not satisfy the decider's statement (full decider: including a running CycleFold
 * witness that violates its relation or does not open its commitments).
 * seconds (optional) = {final fold + KZG openings, witness + sparse products on the host, NTTs, multi-scalar multiplications, total, 0} */
int vimz_decider_prove(vimz_decider* d, vimz_cf* ivc, uint64_t* public_out, uint64_t words_out[100], double seconds[6]);
/* Decider::verify (verify_final_proof, decider.rs:31-50): the checks of contracts/ContrastVerifier.sol:685-783 on (steps, z_0, z_i, 25 words).
 * *result = 0: accepted; else bits: 1 fewer than two steps, 2 KZG opening of cmW, 4 of cmE, 8 Groth16, 16 a word pair is not a curve point. */
int vimz_decider_verify(const vimz_decider* d, uint64_t steps, const uint64_t* z0, const uint64_t* zi, const uint64_t words[100], uint32_t* result);
/* the same against a key given as words (vimz_decider_vk's layout): no context, no GPU */
int vimz_decider_verify_key(const uint64_t* key_words, size_t n_key_words, uint64_t steps, const uint64_t* z0, const uint64_t* zi, uint32_t len_z,
                            const uint64_t words[100], uint32_t* result);

/* ---- ONE proof object out of several row segments: the "host-side sequential final fold" of BASELINE.json's north_star for IVC proofs.
 *      fold_input returns ONE RecursiveSNARK (vimz/src/nova_snark_backend/folding.rs:27-43); row segments of an image folded
 *      concurrently (S proofs on one GPU, or one per GPU) are merged into one verifiable object by out-of-circuit NIFS on both curves —
 *      the final fold CompressedSNARK::prove performs on the last fresh secondary instance (mod.rs:56-59), extended to S segments
 *      (protocol: vimz_amd/csrc/merge_internal.hpp; ours, like the augmented circuits).  The verifier replays the
 *      segments' hash checks, their adjacency and the fold tree from the records, then checks ONE primary and ONE secondary relaxed
 *      instance (or one compressed argument each).  The cross terms, their commitments and the vector folds run on the GPU. -------- */
typedef struct vimz_ivc_merged vimz_ivc_merged;
/* the merged proof of one segment; `segment` is left unchanged and supplies shapes / keys / context: keep it alive while the object is
 * in use (freeing it first orphans the object: its buffers are released, every later call on it fails, freeing it stays safe) */
int vimz_ivc_merged_create(vimz_ivc* segment, vimz_ivc_merged** out);
void vimz_ivc_merged_free(vimz_ivc_merged* m);
/* fold the proof of the NEXT row segment in (same device; read in place, left unchanged): it must start at the state m ends in */
int vimz_ivc_merge(vimz_ivc_merged* m, vimz_ivc* next_segment);
/* the same for two merged proofs of adjacent runs of segments (e.g. one per GPU, brought over with vimz_ivc_merged_save / _load) */
int vimz_ivc_merge_merged(vimz_ivc_merged* m, vimz_ivc_merged* next);
/* fold_input in ONE call (folding.rs:27-43): `nsteps` rows from state z0 proven as n_segments contiguous segments — segment k by
 * segments[k]: IVCs of the same circuits, each on a context of its own on one device; they are reset by this call —, folded concurrently
 * (a host thread each) and merged into one object.  The segments' start states come from the state chain in its two parts: the row
 * digests of all but the last segment at once, each on its successor's context, then short serial host chains.
 * seconds (optional) = {waiting for start states, merge, total}. */
int vimz_ivc_fold_segments(vimz_ivc* const* segments, size_t n_segments, const uint64_t* z0, const uint64_t* step_inputs, size_t nsteps,
                           vimz_ivc_merged** out, double seconds[3]);
/* the same when the caller already holds the rows' digests (vimz_ivc_row_digests over exactly these rows, e.g. a rank of a sharded
 * proof that has just exchanged them with the other ranks); digests == NULL: as above */
int vimz_ivc_fold_segments_dg(vimz_ivc* const* segments, size_t n_segments, const uint64_t* z0, const uint64_t* step_inputs, size_t nsteps,
                              const uint64_t* digests, vimz_ivc_merged** out, double seconds[3]);
/* fold_input for a run of rows whose START STATE is not known yet — a rank of a sharded proof: the segments' fold calls begin at once, the rows' digests (what
 * vimz_ivc_row_digests returns for them) are handed out as soon as the calls' own chain passes have produced them, the caller exchanges digests with the other
 * ranks, chains over the rows before its own (vimz_ivc_chain_from_digests) and provides the state; every row is hashed once.  _digests blocks; _finish joins the
 * folds, merges them into ONE object and frees the handle (out == NULL, or no start state given: cancels).  Circuits whose digests depend on the state: refused.
 * Between _begin and _start the segments hold their contexts (their folds wait for the state inside their calls): call nothing else on those IVCs meanwhile
 * (vimz_ivc_chain_from_digests is host-only and takes no context: fine). */
typedef struct vimz_ivc_pending vimz_ivc_pending;
int vimz_ivc_fold_segments_begin(vimz_ivc* const* segs, size_t n_seg, const uint64_t* step_inputs, size_t nsteps, vimz_ivc_pending** out);
int vimz_ivc_pending_digests(vimz_ivc_pending* p, uint64_t* digests_out);
int vimz_ivc_pending_start(vimz_ivc_pending* p, const uint64_t* z_start);
int vimz_ivc_pending_finish(vimz_ivc_pending* p, vimz_ivc_merged** out, double seconds[3]);
/* rows of a fold call of `nsteps` rows whose Poseidon chains the library would evaluate on the host (its policy, or what vimz_set_head_rows pinned) */
size_t vimz_head_rows_policy(size_t nsteps);
/* the same for a proof of `nsteps` rows made as concurrent segments (vimz_ivc_fold_segments): non-zero while its segments take head batches */
size_t vimz_head_rows_policy_segments(size_t nsteps);
/* RecursiveSNARK::verify(pp, num_steps, z0) for the merged object.  result: 0 = accepted; bit 0 / 1 a segment's primary / secondary chain
 * hash; bit 2 primary relaxed relation; bit 3 / 4 primary comm_W / comm_E; bit 5 secondary relation; bit 6 / 7 secondary comm_W / comm_E;
 * bit 10 public entries of a witness vector differ from the instance; bit 11 kept running products (bookkeeping for further merges);
 * bit 12 statement (total steps, initial state, or segments not adjacent); bit 13 malformed. */
int vimz_ivc_merged_verify(vimz_ivc_merged* m, uint64_t num_steps, const uint64_t* z0, uint32_t* result);
/* info = {steps, segments, ops, len_z, primary wires, primary constraints, secondary wires, secondary constraints} */
int vimz_ivc_merged_info(const vimz_ivc_merged* m, uint64_t info[8]);
int vimz_ivc_merged_state(const vimz_ivc_merged* m, uint64_t* z_start, uint64_t* z_end, uint64_t* steps);
/* seconds = {leaf work, wait for the cross-term commitments, folds + host instance arithmetic, total} */
int vimz_ivc_merged_profile(const vimz_ivc_merged* m, double seconds[4]);
/* The proof as bytes: records ‖ folded witnesses.  vimz_ivc_merged_load takes `vk` = any vimz_ivc created for the same step circuit
 * and keys (neither read nor changed; must outlive the object); the blob is untrusted (range and curve checks; instances recomputed). */
size_t vimz_ivc_merged_size(const vimz_ivc_merged* m);
int vimz_ivc_merged_save(vimz_ivc_merged* m, uint8_t* blob, size_t cap);
int vimz_ivc_merged_load(vimz_ivc* vk, const uint8_t* blob, size_t len, vimz_ivc_merged** out);
/* The same hand-over between two processes of ONE node without the host round trip (one rank per GPU; the ranks' final fold as a tree,
 * the counterpart of benchmark.sh:25-58's processes ending in one proof): _share writes a ticket — the records and a HIP IPC handle of
 * the object's device allocation (returns its size; copies when cap suffices) — that travels as a small message; _open_shared maps the
 * allocation, copies the folded witnesses AND the running products device-to-device (over xGMI between GPUs) into an object of its own
 * and unmaps it.  Records are checked as in _load; the products are taken as they are (vimz_ivc_merged_verify recomputes them).  The
 * sharing process keeps its object until the receiver is done.  Fails with VIMZ_ERR_HIP where IPC / peer access is unavailable: use
 * _save / _load then. */
int64_t vimz_ivc_merged_share(vimz_ivc_merged* m, void* ticket, size_t cap);
int vimz_ivc_merged_open_shared(vimz_ivc* vk, const uint8_t* ticket, size_t len, vimz_ivc_merged** out);
/* what an independent verifier replays: header {magic, segments, ops, len_z, n_w1, n_c1, n_w2, n_c2}, per segment {n, z_start, z_end,
 * U1 (W, E, u, X0, X1), U2, u2 (W, x0, x1), T}, per op {kind, leaf [, T_p, T_q]} — canonical little-endian words; returns the byte size */
int64_t vimz_ivc_merged_records(const vimz_ivc_merged* m, void* buf, size_t cap);
/* side 0 / 1; what = VIMZ_IX_RUNNING_Z, VIMZ_IX_RUNNING_E (canonical), VIMZ_IX_INSTANCE (comm_W, comm_E, u, X0, X1 of the folded instance) */
int64_t vimz_ivc_merged_export(vimz_ivc_merged* m, int side, int what, void* buf, size_t cap);
/* CompressedSNARK::{prove, verify} for the merged object: one argument for the folded primary and one for the folded secondary
 * instance (vimz_amd/csrc/spartan.hip); the blob carries the records.  result bits as vimz_ivc_verify_compressed, bit 2 / 3 = the
 * primary / secondary argument. */
size_t vimz_ivc_merged_compressed_size(const vimz_ivc_merged* m);
int vimz_ivc_merged_compress(vimz_ivc_merged* m, uint8_t* blob, size_t cap, double seconds[2]);
int vimz_ivc_verify_merged_compressed(vimz_ivc* vk, const uint8_t* blob, size_t len, uint64_t num_steps, const uint64_t* z0, uint32_t* result);

/* Everything an independent verifier needs, canonical little-endian 4 x u64 per element (the parity tests hand these to the
 * CPU oracle's verifier).  side 0 = primary (BN254 Fr / G1), 1 = secondary (BN254 Fq / Grumpkin).
 *   what = VIMZ_CX_{A,B,C}_{ROWPTR,COL,COEF}, VIMZ_CX_DICT_CANON : the augmented circuit's R1CS
 *   what = VIMZ_IX_*  below.  Returns the byte size (copies when buf is large enough). */
#define VIMZ_IX_RUNNING_Z 100   /* running witness vector [u | W | X0 X1] (wire order) */
#define VIMZ_IX_RUNNING_E 101
#define VIMZ_IX_FRESH_Z 102     /* side 1 only: the last fresh secondary witness vector */
#define VIMZ_IX_INSTANCE 103    /* running instance: comm_W.x, comm_W.y, comm_E.x, comm_E.y, u, X0, X1  (7 elements; coordinates in the
                                   commitment curve's base field, u/X in this side's scalar field) */
#define VIMZ_IX_FRESH_INSTANCE 104  /* side 1 only: comm_W.x, comm_W.y, x0, x1 */
#define VIMZ_IX_LAST_STEP 107   /* vimz_cf_export, side 0: what the last step's F' was given and returned (layout: cyclefold.hip) */
#define VIMZ_IX_INFO 106        /* u64[4]: wires, constraints, step wires, step constraints */
#define VIMZ_IX_PARAMS 105      /* digest, z0..., z_i...  (1 + 2 len_z elements of this side's field; secondary len_z = 1) */
int64_t vimz_ivc_export(vimz_ivc* v, int side, int what, void* buf, size_t cap);
/* Host-only hooks for the parity tests of the verifier circuit itself (no GPU): build the augmented circuit of one side over a
 * trivial step circuit (z_out = z_in, arity 1), export it, and run its witness generator on given inputs.
 *   inputs (canonical elements): digest, i, z_0, z_i, U = (W.x, W.y, E.x, E.y, u, X0, X1), u = (W.x, W.y, x0, x1), T = (x, y)   -> 17 elements
 *   wires_out: n_wires elements (the full witness incl. the constant and the trivial step)
 *   outputs: U_new (7), rho (1), x0, x1 (2), flag (1: 1 = some range check failed)                                      -> 11 elements */
typedef struct vimz_augcircuit vimz_augcircuit;
int vimz_augcircuit_build(int side, vimz_augcircuit** out);
void vimz_augcircuit_free(vimz_augcircuit* c);
int64_t vimz_augcircuit_export(const vimz_augcircuit* c, int what, void* buf, size_t cap);   /* VIMZ_CX_* R1CS codes, VIMZ_IX_INFO */
int vimz_augcircuit_witness(const vimz_augcircuit* c, const uint64_t* inputs, uint64_t* wires_out, uint64_t* outputs);

/* ---- field-arithmetic probes (element-wise on the GPU; used by the parity tests to pin the device
 *      Montgomery arithmetic against the oracle).  op: 0 add, 1 sub, 2 mul, 3 inverse (b ignored). -------- */
int vimz_field_op(vimz_ctx* ctx, int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* group probes: out[i] = p[i] + q[i] (affine canonical in/out), through the device XYZZ formulas */
int vimz_curve_add(vimz_ctx* ctx, int curve, const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* out_xy, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* VIMZ_HIP_H */
