// Loads / stores of curve data in its resident form: coordinates of COORD_WORDS (10) words, 9 used (fp29.hpp) — every point
// starts 16-byte aligned: affine 80 B (five 16-byte accesses), XYZZ 160 B.
#pragma once
#include <hip/hip_runtime.h>
#include "ec.hpp"

namespace vz {

template <class F>
__device__ __forceinline__ void load_words20(const uint32_t* __restrict__ p, F& a, F& b) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
  a.v[0] = w0.x; a.v[1] = w0.y; a.v[2] = w0.z; a.v[3] = w0.w; a.v[4] = w1.x; a.v[5] = w1.y; a.v[6] = w1.z; a.v[7] = w1.w; a.v[8] = w2.x;
  b.v[0] = w2.z; b.v[1] = w2.w; b.v[2] = w3.x; b.v[3] = w3.y; b.v[4] = w3.z; b.v[5] = w3.w; b.v[6] = w4.x; b.v[7] = w4.y; b.v[8] = w4.z;
}
template <class F>
__device__ __forceinline__ void store_words20(uint32_t* __restrict__ p, const F& a, const F& b) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]); q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
  q[2] = make_uint4(a.v[8], 0u, b.v[0], b.v[1]); q[3] = make_uint4(b.v[2], b.v[3], b.v[4], b.v[5]);
  q[4] = make_uint4(b.v[6], b.v[7], b.v[8], 0u);
}
template <class F>
__device__ __forceinline__ Affine<F> load_affine(const uint32_t* __restrict__ bases, uint32_t idx) {
  Affine<F> q; load_words20(bases + (size_t)AFFINE_WORDS * idx, q.x, q.y); return q;
}
template <class F>
__device__ __forceinline__ void store_xyzz(uint32_t* __restrict__ base, size_t idx, const XYZZ<F>& p) {
  uint32_t* d = base + (size_t)XYZZ_WORDS * idx;
  store_words20(d, p.X, p.Y); store_words20(d + AFFINE_WORDS, p.ZZ, p.ZZZ);
}
template <class F>
__device__ __forceinline__ XYZZ<F> load_xyzz(const uint32_t* __restrict__ base, size_t idx) {
  const uint32_t* d = base + (size_t)XYZZ_WORDS * idx;
  XYZZ<F> p; load_words20(d, p.X, p.Y); load_words20(d + AFFINE_WORDS, p.ZZ, p.ZZZ); return p;
}

}  // namespace vz
