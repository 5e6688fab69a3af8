// Explicit instantiation of the MSM pipeline for one curve (one TU per curve so the build parallelises).
#include "msm.hpp"
namespace vz {
template hipError_t msm_run<BnG1>(hipStream_t, MsmWorkspace&, const uint32_t*, const uint32_t*, size_t, int, int,
                                  Affine<BnG1::Base>*, MsmStats*, hipEvent_t*, int, const BaseTables*);
template hipError_t msm_launch<BnG1>(hipStream_t, MsmWorkspace&, const uint32_t*, const uint32_t*, size_t, int, int, void*, MsmPlan*, hipEvent_t*, int, const BaseTables*);
template hipError_t build_tables<BnG1>(hipStream_t, const uint32_t*, size_t, int, int, uint32_t*);
template hipError_t build_multiples<BnG1>(hipStream_t, const uint32_t*, size_t, int, int, uint32_t*);
template Affine<BnG1::Base> msm_finish<BnG1>(const MsmPlan&, const void*);
template hipError_t ones_launch<BnG1>(hipStream_t, MsmWorkspace&, const uint32_t*, const uint32_t*, size_t, int, void*);
template XYZZ<BnG1::Base> ones_finish<BnG1>(const void*);
}
