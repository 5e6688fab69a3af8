// Explicit instantiation of the MSM pipeline for one curve (one TU per curve so the build parallelises).
#include "msm.hpp"
namespace vz {
template hipError_t msm_run<Vesta>(hipStream_t, MsmWorkspace&, const uint32_t*, const uint32_t*, size_t, int, int,
                                  Affine<Vesta::Base>*, MsmStats*, hipEvent_t*, int, const BaseTables*);
template hipError_t msm_launch<Vesta>(hipStream_t, MsmWorkspace&, const uint32_t*, const uint32_t*, size_t, int, int, void*, MsmPlan*, hipEvent_t*, int, const BaseTables*);
template hipError_t build_tables<Vesta>(hipStream_t, const uint32_t*, size_t, int, int, uint32_t*);
template hipError_t build_multiples<Vesta>(hipStream_t, const uint32_t*, size_t, int, int, uint32_t*);
template Affine<Vesta::Base> msm_finish<Vesta>(const MsmPlan&, const void*);
}
