#pragma once
#include <memory>
#include "circuit/circuits.hpp"
struct vimz_circuit {
  int transformation;
  vz::cb::StepShape shape;
  std::unique_ptr<vz::cb::CircuitBuild> build;
};
