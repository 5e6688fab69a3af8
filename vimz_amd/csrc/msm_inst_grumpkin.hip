// Explicit instantiation of the MSM pipeline for one curve (one TU per curve so the build parallelises).
#include "msm.hpp"
namespace vz {
template hipError_t msm_run<Grumpkin>(hipStream_t, MsmWorkspace&, const uint32_t*, const uint32_t*, size_t, int, int,
                                  Affine<Grumpkin::Base>*, MsmStats*, hipEvent_t*, int, const BaseTables*);
template hipError_t msm_launch<Grumpkin>(hipStream_t, MsmWorkspace&, const uint32_t*, const uint32_t*, size_t, int, int, void*, MsmPlan*, hipEvent_t*, int, const BaseTables*);
template hipError_t build_tables<Grumpkin>(hipStream_t, const uint32_t*, size_t, int, int, uint32_t*);
template hipError_t build_multiples<Grumpkin>(hipStream_t, const uint32_t*, size_t, int, int, uint32_t*);
template Affine<Grumpkin::Base> msm_finish<Grumpkin>(const MsmPlan&, const void*);
}
