// Commitment-key derivation on the GPU: n independent generators
//   ck_i = try-and-increment( SHAKE256(label ‖ LE64(i) ‖ LE32(ctr)) )
// x = the 32 output bytes as a little-endian integer with the bits above the field size cleared,
// rejected if x >= p or x^3 + b is a non-residue; y = sqrt(x^3 + b) with parity equal to output bit 255.
// nova-snark derives its key from SHAKE256("ck") too but through halo2curves' hash-to-curve, which is
// not vendored; byte-compatibility with that is NOT claimed (DESIGN.md "parity unpinned" list) — a
// caller that needs nova-snark's exact key uploads it with vimz_bases_upload instead.
#pragma once
#include <hip/hip_runtime.h>
#include "ec.hpp"

namespace vz {

struct SqrtParams {      // Tonelli–Shanks constants for one field, computed on the host at first use
  uint32_t q[8];         // odd part of p-1
  uint32_t q1h[8];       // (q+1)/2
  uint32_t z[8];         // g^q for a non-residue g (Montgomery form)
  int s;                 // 2-adicity
};
struct CkLabel { uint8_t bytes[64]; int len; };

template <class F> hipError_t ckgen_run(hipStream_t stream, const CkLabel& label, int b_small, size_t first, size_t n, uint32_t* d_out);
template <class F> SqrtParams sqrt_params();

}  // namespace vz
