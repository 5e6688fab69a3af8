/* vimz_hip_testing.h — test hooks of the MI355X Nova-folding library.
 *
 * NOT part of the product ABI: libvimz_hip.so is built without them (tests/test_abi.py checks that it exports none of these symbols).
 * They are compiled under -DVIMZ_TESTING into vimz_amd/libvimz_hip_testing.so (the same sources otherwise) and, with the host passes
 * under ThreadSanitizer / AddressSanitizer, into vimz_amd/csrc/build/libvimz_hip_{tsan,asan}.so for the GPU-free ones
 * (tests/native/host_sanitize.cpp, run by the CPU suite).
 */
#ifndef VIMZ_HIP_TESTING_H
#define VIMZ_HIP_TESTING_H
#include "vimz_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* overwrite one element (canonical) of a witness vector on the device — which = 0 running main Z, 1 last fresh main Z, 2 running CycleFold Z,
 * 3 running main E, 4 running CycleFold E (soundness tests flip wires and expect vimz_cf_verify and the oracle-side verifier to reject) */
int vimz_cf_poke(vimz_cf* v, int which, size_t index, const uint64_t value[4]);
/* host only, no GPU — `steps` steps of the Nova + CycleFold recursion over the trivial step circuit with made-up commitments, every witness
 * checked against its R1CS and every in-circuit fold against field / curve arithmetic (result 0 = good; counts: F' wires, constraints,
 * CycleFold wires, constraints, then of the last step's flip test — every wire incremented by one must violate a row — wires of F'
 * flipped, unnoticed, wires of the CycleFold circuit flipped, unnoticed). */
int vimz_cf_selfcheck(int steps, uint32_t* result, uint64_t counts[8]);
/* the LAST step of the same host-only run, for an outside restatement of the relation F' enforces: digest, z_0 (one element), then the words of
 * VIMZ_IX_LAST_STEP.  Returns the byte size (copies when buf is large enough); negative on error or when the self-check itself fails. */
int64_t vimz_cf_selfcheck_last_step(int steps, void* buf, size_t cap);
/* host only: two runs of made-up, self-consistent segment records replayed by the library; output = digest, word count + records, the
 * accumulator arrived at (layout: cyclefold_merge.hip) — for an outside replay of the merge transcript in the CPU suite.  Returns the byte size. */
int64_t vimz_cf_selfcheck_merge(int segs_run0, int segs_run1, void* buf, size_t cap);
/* host only: `jobs` trivial jobs posted to and awaited from one helper thread of the verifier circuits' witness generators; returns how many
 * ran.  With VIMZ_WORKER_SPIN_US=0 the helper sleeps between jobs, so that every post is a wake-up (a lost one hangs the call). */
int64_t vimz_worker_selftest(int jobs);
/* host only: the canonical bit decomposition of hash outputs (aug/cs.hpp: bits_strict) against the ALIASED witness — the bits of h + p for
 * hash values h with h + p below 2^254 — on the gadget alone and inside F' (field 0 = BN254 Fr, 1 = Fq).  out[0] = values tried,
 * out[1] = of them aliasable, out[2] = aliased witnesses that satisfy the plain gadget's rows (must equal out[1]: the attack is real),
 * out[3] = aliased witnesses whose only violated rows are rows of the comparison with p - 1 (must equal out[1]: the fix is what stops it),
 * out[4] = honest witnesses that violate a row (must be 0), out[5] = steps of F' run with aliased challenge decompositions,
 * out[6] = of them left satisfiable (must be 0), out[7] = steps in which an alias existed. */
int vimz_strict_bits_selfcheck(int field, int values, uint64_t out[8]);
/* the negative test of the compressed proof's public-slot binding (tests/test_gpu_compress.py): while on, vimz_ivc_compress plays a cheating
 * prover that claims x0 + 1 for the last fresh instance and hides the difference under the generator of that wire's slot */
void vimz_test_forge_public_slot(int on);
/* deterministic TEST setups of the decider path (vimz_kzg_setup / vimz_decider_setup with the toxic waste derived from `seed`: anyone who knows the
 * seed can forge) — for reproducible keys in tests and benchmarks only */
/* host only, no GPU: the decider circuit over `steps` steps of the Nova + CycleFold recursion of the trivial step circuit (made-up commitments, the running
 * witness folded on the host), its final fold, witness and R1CS; full != 0: the FULL decider, over real CycleFold instances kept on the host; result 0 = good
 * (bits: vimz_amd/csrc/groth16.hip); counts = {decider constraints, wires, public inputs, main constraints} */
int vimz_decider_selfcheck(int steps, int full, uint32_t* result, uint64_t counts[4]);
int vimz_testing_kzg_setup_seeded(vimz_ctx* ctx, const uint8_t* seed, size_t seed_len, size_t n, vimz_bases** srs_out, uint64_t vk_g2_out[16]);
int vimz_testing_decider_setup_seeded(vimz_cf* prover, const uint64_t kzg_vk_g2[16], int light, const uint8_t* seed, size_t seed_len, vimz_decider** out, double seconds[4]);

#ifdef __cplusplus
}
#endif
#endif
