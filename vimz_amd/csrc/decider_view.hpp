// What the decider (groth16.hip) reads of a merged Nova + CycleFold proof of ONE segment — U_{i+1} = NIFS.V(U_i, u_i) with its folded
// witness — without knowing the object's layout (cyclefold_merge.hip owns it).
#pragma once
#include "cyclefold_internal.hpp"

struct vimz_cf_merged;
struct CfDeciderView {
  vimz_cf* vk = nullptr;
  uint64_t n = 0; std::vector<Fe> zs, ze;                       // the statement: steps, z_0, z_i
  aug::CfMainRelaxed U; aug::CfMainFresh u; aug::CfRelaxed cfU;  // the IVC proof's instances
  G1Aff UW, UE, uW, cmT;                                         // ... their commitments as points, and the final fold's cross-term commitment
  uint32_t r[4] = {0, 0, 0, 0};                                  // its challenge (128 bits)
  G1Aff cW, cE; Fe un, x0n, x1n;                                 // U_{i+1}
  const uint32_t* Zp = nullptr; const uint32_t* Ep = nullptr;   // its witness (Z = [u' | W' | x0' x1']) and error vector on the device
};
// fails unless m holds exactly one segment
int vz_cf_merged_decider_view(vimz_cf_merged* m, CfDeciderView* out);
