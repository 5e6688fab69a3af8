// Explicit instantiation of commitment-key derivation over one coordinate field.
#include "ckgen_impl.hpp"
namespace vz {
template hipError_t ckgen_run<Fp<VestaFq>>(hipStream_t, const CkLabel&, int, size_t, size_t, uint32_t*);
}
