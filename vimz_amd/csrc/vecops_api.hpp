// Host-side launchers of the element-wise field kernels (definitions: vecops.hpp, instantiated per field in
// vecops_inst_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "ec.hpp"

namespace vz {
template <class F> void launch_to_mont(hipStream_t s, uint32_t* v, size_t n);
template <class F> void launch_from_mont(hipStream_t s, const uint32_t* v, uint32_t* o, size_t n);
template <class F> void launch_field_probe(hipStream_t s, int op, const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n);
template <class F> void launch_curve_add_probe(hipStream_t s, const uint32_t* p, const uint32_t* q, uint32_t* o, size_t n);
}  // namespace vz
