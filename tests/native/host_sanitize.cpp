// GPU-free code paths of the library under ThreadSanitizer / AddressSanitizer + UBSan (vimz_amd/csrc/Makefile: `make sanitize` builds
// build/host_tsan and build/host_asan from this file over a library whose HOST passes are instrumented).  Test infrastructure: the CPU
// suite runs both (tests/test_sanitizers.py); nothing of the product uses it.
//
// What runs: the helper threads of the verifier circuits' witness generators (posted / awaited 20 000 times with the helper asleep
// between jobs, then with the default spin window), Nova's verifier circuit on two host threads at once — each with its own helper
// threads for the scalar-multiplication chains, the way concurrent row segments run them —, the Nova + CycleFold recursion over the
// trivial step circuit with helper threads (every witness against its R1CS, the flip test), the merge transcript replay, the canonical
// bit decomposition's negative test, the step-circuit builder, and the iden3 loaders on malformed bytes.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "vimz_hip_testing.h"

static int fails = 0;
#define EXPECT(c, ...) do { if (!(c)) { fails++; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

static void aug_thread(int side, int runs, std::atomic<int>* bad_rc) {
  vimz_augcircuit* c = nullptr;
  if (vimz_augcircuit_build(side, &c) != VIMZ_OK) { bad_rc->fetch_add(1); return; }
  // a non-base step whose incoming points are real curve points: both 128-step chains run on the helper threads (the hash check of
  // the made-up instance fails — the witness is flagged, which is not what this run is about)
  std::vector<uint64_t> in(4 * 17, 0), out(4 * 11, 0);
  auto set = [&](int k, uint64_t v0, uint64_t v1 = 0, uint64_t v2 = 0, uint64_t v3 = 0) { in[4 * k] = v0; in[4 * k + 1] = v1; in[4 * k + 2] = v2; in[4 * k + 3] = v3; };
  set(0, 12345); set(1, 1); set(2, 5); set(3, 5); set(8, 1);
  if (side == 1) { set(11, 1); set(12, 2); set(15, 1); set(16, 2); }                         // BN254 G1 generator (1, 2): coordinates in Fq
  else { set(11, 1); set(12, 0x833fc48d823f272cull, 0x2d270d45f1181294ull, 0xcf135e7506a45d63ull, 0x2ull);   // Grumpkin generator (1, sqrt(-16)): coordinates in Fr
         set(15, 1); set(16, 0x833fc48d823f272cull, 0x2d270d45f1181294ull, 0xcf135e7506a45d63ull, 0x2ull); }
  for (int r = 0; r < runs; r++) {
    in[4 * 13] = (uint64_t)r + 7;      // another statement every time
    if (vimz_augcircuit_witness(c, in.data(), nullptr, out.data()) != VIMZ_OK) bad_rc->fetch_add(1);
  }
  vimz_augcircuit_free(c);
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  // 1. helper threads: every post a wake-up, then the default spin window
  setenv("VIMZ_WORKER_SPIN_US", "0", 1);
  const int jobs = quick ? 2000 : 20000;
  EXPECT(vimz_worker_selftest(jobs) == jobs, "worker selftest (sleeping helper)");
  // 2. Nova's verifier circuit on two threads at once, each with helper threads of its own
  {
    std::atomic<int> bad{0};
    std::thread a(aug_thread, 0, quick ? 2 : 6, &bad), b(aug_thread, 1, quick ? 2 : 6, &bad);
    a.join(); b.join();
    EXPECT(bad.load() == 0, "augmented circuit witness runs: %d failed", bad.load());
  }
  // 3. Nova + CycleFold recursion on the host, helper threads on
  {
    setenv("VIMZ_CF_SELFCHECK_WORKERS", "1", 1);
    uint32_t res = 1; uint64_t counts[8] = {0};
    const int rc = vimz_cf_selfcheck(quick ? 2 : 3, &res, counts);
    EXPECT(rc == VIMZ_OK && res == 0, "vimz_cf_selfcheck rc %d result %#x", rc, res);
    EXPECT(counts[4] > 20000 && counts[5] <= 8, "flip test: %llu wires, %llu unnoticed", (unsigned long long)counts[4], (unsigned long long)counts[5]);
    const int64_t n = vimz_cf_selfcheck_last_step(2, nullptr, 0);
    EXPECT(n > 0, "vimz_cf_selfcheck_last_step size %lld", (long long)n);
    if (n > 0) { std::vector<uint8_t> buf((size_t)n); EXPECT(vimz_cf_selfcheck_last_step(2, buf.data(), buf.size()) == n, "last step copy"); }
  }
  // 4. merge transcript replay
  {
    const int64_t n = vimz_cf_selfcheck_merge(3, 2, nullptr, 0);
    EXPECT(n > 0, "vimz_cf_selfcheck_merge size %lld", (long long)n);
    if (n > 0) { std::vector<uint8_t> buf((size_t)n); EXPECT(vimz_cf_selfcheck_merge(3, 2, buf.data(), buf.size()) == n, "merge replay copy"); }
  }
  // 5. canonical decomposition of hash outputs
  for (int field = 0; field < 2; field++) {
    uint64_t o[8];
    const int rc = vimz_strict_bits_selfcheck(field, quick ? 16 : 48, o);
    EXPECT(rc == VIMZ_OK && o[1] > 0 && o[2] == o[1] && o[3] == o[1] && o[4] == 0 && o[6] == 0, "strict bits field %d: rc %d tried %llu aliasable %llu plain-ok %llu only-strict-bad %llu honest-bad %llu F' steps %llu sat %llu",
           field, rc, (unsigned long long)o[0], (unsigned long long)o[1], (unsigned long long)o[2], (unsigned long long)o[3], (unsigned long long)o[4], (unsigned long long)o[5], (unsigned long long)o[6]);
  }
  // 6. the step-circuit builder and the loaders on bytes that are not what they claim to be
  {
    vimz_circuit* c = nullptr;
    EXPECT(vimz_circuit_build(VIMZ_T_HASH, 128, 0, 0, 0, 0, &c) == VIMZ_OK && c, "vimz_circuit_build(hash, HD)");
    if (c) { uint64_t info[VIMZ_CIRCUIT_INFO_LEN]; EXPECT(vimz_circuit_info(c, info) == VIMZ_OK && info[1] == 6672, "hash_step(HD): %llu constraints", (unsigned long long)info[1]); vimz_circuit_free(c); }
    std::vector<uint8_t> junk(4096);
    for (size_t k = 0; k < junk.size(); k++) junk[k] = (uint8_t)(k * 131 + 7);
    memcpy(junk.data(), "r1cs\x01\0\0\0\x03\0\0\0", 12);
    for (size_t len : {size_t(0), size_t(11), size_t(12), size_t(40), size_t(200), junk.size()}) {
      vimz_circuit* r = nullptr;
      EXPECT(vimz_circuit_load_r1cs(junk.data(), len, &r) != VIMZ_OK, "malformed .r1cs of %zu bytes accepted", len);
      if (r) vimz_circuit_free(r);
    }
    memcpy(junk.data(), "wtns\x02\0\0\0\x02\0\0\0", 12);
    for (size_t len : {size_t(0), size_t(12), size_t(60), junk.size()}) {
      std::vector<uint64_t> w(64); size_t n = 0;
      EXPECT(vimz_wtns_load(junk.data(), len, w.data(), 16, &n) != VIMZ_OK, "malformed .wtns of %zu bytes accepted", len);
    }
  }
  if (fails) { fprintf(stderr, "host_sanitize: %d check(s) failed\n", fails); return 1; }
  printf("host_sanitize ok\n");
  return 0;
}
