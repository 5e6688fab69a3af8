#pragma once
#include "ckgen.hpp"
#include "keccak.hpp"

namespace vz {

template <class F>
VZ_HD F fp_pow(const F& a, const uint32_t* e) {
  F acc = F::one();
  bool started = false;
  for (int i = 255; i >= 0; i--) {
    if (started) acc = F::sqr(acc);
    if ((e[i >> 5] >> (i & 31)) & 1) { acc = started ? F::mul(acc, a) : a; started = true; }
  }
  return started ? acc : F::one();
}

// returns true and writes the root if a is a square
template <class F>
VZ_HD bool fp_sqrt(const F& a, const SqrtParams& sp, F* root) {
  if (a.is_zero()) { *root = a; return true; }
  F zz; for (int i = 0; i < 8; i++) zz.v[i] = sp.z[i];
  F x = fp_pow(a, sp.q1h);
  F b = fp_pow(a, sp.q);
  int m = sp.s;
  const F one = F::one();
  while (!b.eq(one)) {
    int i = 0; F t = b;
    while (!t.eq(one)) { t = F::sqr(t); i++; if (i >= m) return false; }  // non-residue
    F g = zz;
    for (int k = 0; k < m - i - 1; k++) g = F::sqr(g);
    x = F::mul(x, g);
    zz = F::sqr(g);
    b = F::mul(b, zz);
    m = i;
  }
  *root = x;
  return true;
}

template <class F>
SqrtParams sqrt_params() {
  SqrtParams sp;
  uint32_t pm1[8]; for (int i = 0; i < 8; i++) pm1[i] = F::Params::MOD.w[i];
  pm1[0] -= 1;  // p odd
  int s = 0; uint32_t q[8]; for (int i = 0; i < 8; i++) q[i] = pm1[i];
  while (!(q[0] & 1)) { for (int i = 0; i < 8; i++) q[i] = (q[i] >> 1) | (i < 7 ? q[i + 1] << 31 : 0); s++; }
  sp.s = s;
  for (int i = 0; i < 8; i++) sp.q[i] = q[i];
  uint32_t h[8]; uint64_t c = 1;  // (q+1)/2
  for (int i = 0; i < 8; i++) { c += q[i]; h[i] = (uint32_t)c; c >>= 32; }
  for (int i = 0; i < 8; i++) sp.q1h[i] = (h[i] >> 1) | (i < 7 ? h[i + 1] << 31 : 0);
  uint32_t half[8];  // (p-1)/2
  for (int i = 0; i < 8; i++) half[i] = (pm1[i] >> 1) | (i < 7 ? pm1[i + 1] << 31 : 0);
  F g = F::one();
  for (;;) {
    g = F::add(g, F::one());
    F e = fp_pow(g, half);
    if (!e.eq(F::one())) break;  // Euler criterion: -1 => non-residue
  }
  F z = fp_pow(g, q);
  for (int i = 0; i < 8; i++) sp.z[i] = z.v[i];
  return sp;
}

// generator `idx` of the key labelled `label` (ckgen.hpp: try-and-increment over SHAKE256), Montgomery coordinates.  Host and device.
template <class F>
VZ_HD void ckgen_point(const CkLabel& label, const SqrtParams& sp, int b_small, uint64_t idx, F* xo, F* yo) {
  uint8_t msg[80];
  for (int i = 0; i < label.len; i++) msg[i] = label.bytes[i];
  for (int i = 0; i < 8; i++) msg[label.len + i] = (uint8_t)(idx >> (8 * i));
  F bcoef = F::zero();
  { F o = F::one(); int bm = b_small < 0 ? -b_small : b_small; for (int i = 0; i < bm; i++) bcoef = F::add(bcoef, o); if (b_small < 0) bcoef = F::neg(bcoef); }
  for (uint32_t ctr = 0;; ctr++) {
    for (int i = 0; i < 4; i++) msg[label.len + 8 + i] = (uint8_t)(ctr >> (8 * i));
    uint64_t h[4];
    shake256_32(msg, label.len + 12, h);
    const uint32_t sign = (uint32_t)(h[3] >> 63);
    F x;
    for (int i = 0; i < 4; i++) { x.v[2 * i] = (uint32_t)h[i]; x.v[2 * i + 1] = (uint32_t)(h[i] >> 32); }
    const int bits = F::Params::BITS;
    if (bits < 256) x.v[7] &= (bits % 32) ? ((1u << (bits % 32)) - 1) : 0xffffffffu;
    bool ge = true;  // x >= p ?
    for (int i = 7; i >= 0; i--) { if (x.v[i] != F::Params::MOD.w[i]) { ge = x.v[i] > F::Params::MOD.w[i]; break; } }
    if (ge) continue;
    F xm = F::to_mont(x);
    F rhs = F::add(F::mul(F::sqr(xm), xm), bcoef);
    F y;
    if (!fp_sqrt(rhs, sp, &y)) continue;
    F yc = F::from_mont(y);
    if ((yc.v[0] & 1) != sign) y = F::neg(y);
    *xo = xm; *yo = y;
    return;
  }
}

template <class F>
__global__ void __launch_bounds__(256) k_ckgen(CkLabel label, SqrtParams sp, int b_small, size_t first, size_t n, uint32_t* __restrict__ out) {
  size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= n) return;
  F xm, y;
  ckgen_point<F>(label, sp, b_small, first + t, &xm, &y);
  uint4* o = reinterpret_cast<uint4*>(out + 16 * t);
  o[0] = make_uint4(xm.v[0], xm.v[1], xm.v[2], xm.v[3]); o[1] = make_uint4(xm.v[4], xm.v[5], xm.v[6], xm.v[7]);
  o[2] = make_uint4(y.v[0], y.v[1], y.v[2], y.v[3]); o[3] = make_uint4(y.v[4], y.v[5], y.v[6], y.v[7]);
}

template <class F>
hipError_t ckgen_run(hipStream_t stream, const CkLabel& label, int b_small, size_t first, size_t n, uint32_t* d_out) {
  static const SqrtParams sp = sqrt_params<F>();
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_ckgen<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, label, sp, b_small, first, n, d_out);
  return hipGetLastError();
}

}  // namespace vz
