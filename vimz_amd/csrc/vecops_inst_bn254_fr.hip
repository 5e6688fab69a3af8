// Explicit instantiation of the element-wise field kernels for one field.
#include "vecops.hpp"
namespace vz {
typedef Fp<BnFr> F_;
template void launch_to_mont<F_>(hipStream_t, uint32_t*, size_t);
template void launch_from_mont<F_>(hipStream_t, const uint32_t*, uint32_t*, size_t);
template void launch_field_probe<F_>(hipStream_t, int, const uint32_t*, const uint32_t*, uint32_t*, size_t);
template void launch_points_to_internal<F_>(hipStream_t, const uint32_t*, int, uint32_t*, size_t);
template void launch_points_from_internal<F_>(hipStream_t, const uint32_t*, int, uint32_t*, size_t);
template void launch_curve_add_probe<F_>(hipStream_t, const uint32_t*, const uint32_t*, uint32_t*, size_t);
}
