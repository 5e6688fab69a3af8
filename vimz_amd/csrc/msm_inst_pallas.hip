// Explicit instantiation of the MSM pipeline for one curve (one TU per curve so the build parallelises).
#include "msm.hpp"
namespace vz {
template hipError_t msm_run<Pallas>(hipStream_t, MsmWorkspace&, const uint32_t*, const uint32_t*, size_t, int, int,
                                  Affine<Pallas::Base>*, MsmStats*, hipEvent_t*, int, const BaseTables*);
template hipError_t msm_launch<Pallas>(hipStream_t, MsmWorkspace&, const uint32_t*, const uint32_t*, size_t, int, int, void*, MsmPlan*, hipEvent_t*, int, const BaseTables*);
template hipError_t build_tables<Pallas>(hipStream_t, const uint32_t*, size_t, int, int, uint32_t*);
template hipError_t build_multiples<Pallas>(hipStream_t, const uint32_t*, size_t, int, int, uint32_t*);
template Affine<Pallas::Base> msm_finish<Pallas>(const MsmPlan&, const void*);
}
