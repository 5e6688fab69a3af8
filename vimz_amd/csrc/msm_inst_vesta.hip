// Explicit instantiation of the MSM pipeline for one curve (one TU per curve so the build parallelises).
#include "msm.hpp"
namespace vz {
template hipError_t msm_run<Vesta>(hipStream_t, MsmWorkspace&, const uint32_t*, const uint32_t*, size_t, int, int,
                                  Affine<Vesta::Base>*, MsmStats*, hipEvent_t*, int);
template hipError_t msm_launch<Vesta>(hipStream_t, MsmWorkspace&, const uint32_t*, const uint32_t*, size_t, int, int, void*, MsmPlan*, hipEvent_t*, int);
template Affine<Vesta::Base> msm_finish<Vesta>(const MsmPlan&, const void*);
}
