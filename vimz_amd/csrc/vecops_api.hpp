// Host-side launchers of the element-wise field kernels (definitions: vecops.hpp, instantiated per field in
// vecops_inst_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "ec.hpp"

namespace vz {
template <class F> void launch_to_mont(hipStream_t s, uint32_t* v, size_t n);
template <class F> void launch_from_mont(hipStream_t s, const uint32_t* v, uint32_t* o, size_t n);
template <class F> void launch_field_probe(hipStream_t s, int op, const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n);
// coordinate-form conversion of n affine points: standard (16 words/pt, Montgomery or canonical) <-> internal (20 words/pt)
template <class F> void launch_points_to_internal(hipStream_t s, const uint32_t* std_pts, int canonical, uint32_t* out29, size_t n);
template <class F> void launch_points_from_internal(hipStream_t s, const uint32_t* in29, int canonical, uint32_t* std_pts, size_t n);
template <class F> void launch_curve_add_probe(hipStream_t s, const uint32_t* p, const uint32_t* q, uint32_t* o, size_t n);
}  // namespace vz
