#pragma once
#include <memory>
#include <mutex>
#include "circuit/circuits.hpp"
struct vimz_circuit {
  int transformation;
  vz::cb::StepShape shape;
  std::unique_ptr<vz::cb::CircuitBuild> build;
  // The augmented PRIMARY circuit of a Nova IVC over this step circuit (ivc.hip: IvcPrimaryShape — the step circuit with the verifier circuit appended, and the
  // digest of that shape: 0.12 s of host work, most of it SHA3 over 20 MB of CSR) — synthesised once per circuit object and shared by the IVCs made from it
  // (a proof of S concurrent segments makes S of them); vimz_circuit_prepare_ivc builds it ahead of time, under the contexts' creation.
  mutable std::mutex ivc_mu;
  mutable std::shared_ptr<void> ivc_shape;
};
