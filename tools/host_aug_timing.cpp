// Section timing of the secondary verifier circuit's witness generator on the host (aug/circuit.hpp, VZ_T markers).
// build: cd vimz_amd/csrc && /opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -DVZ_AUG_TIMING -I. -o /tmp/host_aug_timing ../../tools/host_aug_timing.cpp -lpthread
#include <chrono>
#include <cstdio>
#include <vector>
#include "circuit/circuits.hpp"
#include "aug/augmented.hpp"
using namespace vz; using namespace vz::cb; using namespace vz::aug;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
namespace vz { namespace aug { double g_t[16]; const char* g_n[16]; } }
int main() {
  typedef BnFq FP; typedef Fp<FP> F;
  AugCircuit<FP> c; c.init_trivial_step(); c.finish(false);
  AugIn<FP> in; in.digest = f_from_u64<F>(5); in.z0.push_back(F::zero()); in.i = 0;
  in.U = RelaxedInst<F>::zero(); in.u = FreshInst<F>::zero(); in.T.x = in.T.y = F::zero();
  // fresh instance: some point on G1 (commitments the secondary folds live on BN254 G1: y^2 = x^3 + 3): use G = (1,2)
  in.u.W.x = f_from_u64<F>(1); in.u.W.y = f_from_u64<F>(2);
  F z = F::zero(); std::vector<F> aug; bool bad = false;
  AugOut<FP> o = c.witness(in, &z, &z, aug, &bad);
  // chain a few steps so that the timed step is a generic one
  for (int i = 1; i < 4; i++) { in.i = i; in.U = o.U_new; in.u.x0 = F::zero(); /* hash check will fail (bad flag) but the work is the same */ in.T = in.u.W; o = c.witness(in, &z, &z, aug, &bad); }
  for (int rep = 0; rep < 3; rep++) {
    for (int k = 0; k < 16; k++) g_t[k] = 0;
    double t0 = now();
    for (int k = 0; k < 300; k++) o = c.witness(in, &z, &z, aug, &bad);
    double tot = (now() - t0) / 300 * 1e6;
    printf("witness: %.1f us;", tot);
    for (int k = 0; k < 16; k++) if (g_n[k]) printf(" %s=%.1f", g_n[k], g_t[k] / 300 * 1e6);
    printf("\n");
  }
  return 0;
}
