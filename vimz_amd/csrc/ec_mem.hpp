// Loads / stores of curve data in its resident form.  Affine points (keys, window tables): 64 B — x, y as 256-bit integers, one sector, four
// 16-byte accesses, limbs cut out after the load.  XYZZ accumulators: coordinates of COORD_WORDS (10) words, 9 used (fp29.hpp), 160 B.
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
// the 64 bytes of a point as they come out of memory: a gather keeps THESE in flight (16 registers) and cuts the limbs out when the point is used
struct RawAffine { uint4 w0, w1, w2, w3; };
__device__ __forceinline__ RawAffine load_affine_raw(const uint32_t* __restrict__ bases, size_t idx) {
  const uint4* q = reinterpret_cast<const uint4*>(bases + (size_t)AFFINE_WORDS * idx);
  RawAffine r; r.w0 = q[0]; r.w1 = q[1]; r.w2 = q[2]; r.w3 = q[3]; return r;
}
template <class F>
__device__ __forceinline__ Affine<F> affine_of_raw(const RawAffine& a) {
  const uint32_t x[8] = {a.w0.x, a.w0.y, a.w0.z, a.w0.w, a.w1.x, a.w1.y, a.w1.z, a.w1.w}, y[8] = {a.w2.x, a.w2.y, a.w2.z, a.w2.w, a.w3.x, a.w3.y, a.w3.z, a.w3.w};
  Affine<F> r; r.x = F::pack(x); r.y = F::pack(y); return r;
}
template <class F>
__device__ __forceinline__ Affine<F> load_affine_at(const uint32_t* __restrict__ p) { return affine_of_raw<F>(load_affine_raw(p, 0)); }
template <class F>
__device__ __forceinline__ Affine<F> load_affine(const uint32_t* __restrict__ bases, size_t idx) { return load_affine_at<F>(bases + (size_t)AFFINE_WORDS * idx); }
// (coordinates canonical — below p, so below 2^256: what to_affine and the key generators produce)
template <class F>
__device__ __forceinline__ void store_affine_at(uint32_t* __restrict__ p, const Affine<F>& a) {
  uint32_t x[8], y[8]; a.x.unpack(x); a.y.unpack(y);
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(x[0], x[1], x[2], x[3]); q[1] = make_uint4(x[4], x[5], x[6], x[7]);
  q[2] = make_uint4(y[0], y[1], y[2], y[3]); q[3] = make_uint4(y[4], y[5], y[6], y[7]);
}
template <class F>
__device__ __forceinline__ void store_affine(uint32_t* __restrict__ bases, size_t idx, const Affine<F>& a) { store_affine_at<F>(bases + (size_t)AFFINE_WORDS * idx, a); }
constexpr int XYZZ_HALF_WORDS = 2 * COORD_WORDS;
template <class F>
__device__ __forceinline__ void store_xyzz(uint32_t* __restrict__ base, size_t idx, const XYZZ<F>& p) {
  uint32_t* d = base + (size_t)XYZZ_WORDS * idx;
  store_words20(d, p.X, p.Y); store_words20(d + XYZZ_HALF_WORDS, p.ZZ, p.ZZZ);
}
template <class F>
__device__ __forceinline__ XYZZ<F> load_xyzz(const uint32_t* __restrict__ base, size_t idx) {
  const uint32_t* d = base + (size_t)XYZZ_WORDS * idx;
  XYZZ<F> p; load_words20(d, p.X, p.Y); load_words20(d + XYZZ_HALF_WORDS, p.ZZ, p.ZZZ); return p;
}

// ---- four lanes, one full addition ---------------------------------------------------------------------------------------------------
// The tails of a bucket reduction (suffix scans, trees) are chains of DEPENDENT full additions with few of them per level: one lane
// per addition leaves three of a CU's four SIMDs idle and takes 14 sequential field multiplications (≈7 µs).  The 14 products of
// add-2008-s fall into four dependency levels of at most four independent products:
//     U1 = X1·ZZ2, U2 = X2·ZZ1, S1 = Y1·ZZZ2, S2 = Y2·ZZZ1   |   PP = P², RR = R², Z12 = ZZ1·ZZ2, Z123 = ZZZ1·ZZZ2
//     PPP = P·PP, Q = U1·PP, ZZ3 = Z12·PP                     |   Ya = R·(Q − X3), Yb = S1·PPP, ZZZ3 = Z123·PPP
// so a quad of consecutive lanes (q = lane & 3) computes one addition in four multiplication levels, exchanging 9-word field
// elements inside the quad by DPP moves (63 words in all).  Operands and result live in LDS (sh[]): quad_add_compute reads them and returns what
// this lane will write; the caller puts a barrier between it and quad_add_store (other quads may still read the destination).
// Identity operands and the doubling / cancellation case (P ≡ 0; lane 0 of the quad then runs the scalar formula) keep add_full's
// semantics.  Same bounds as add_full (ec.hpp).  Every lane of the wave must call both functions (shuffles are wave-wide).
template <class F> struct QuadRes { F c0, c1; XYZZ<F> full; uint32_t mode; };       // mode 0: nothing to store; 1: copy of b; 2: sum; 3: lane 0 holds `full`
// lane K of every quad to the quad's four lanes: a DPP quad_perm move (one VALU instruction per word; __shfl with a runtime lane goes
// through ds_bpermute — the LDS crossbar and its latency, exposed with one wave per SIMD)
template <int K, class F>
__device__ __forceinline__ F quad_bcast(const F& v) {
  F r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.v[i], K * 0x55, 0xf, 0xf, true);
  return r;
}
// The rare case of a four-lane addition (equal x: doubling or cancellation) is a FUNCTION: inlined, its 3.5 k instructions sat in every one of the dozens of
// unrolled tree levels of k_msm_small / k_combine / k_reduce (74 k -> 42 k, 73 k -> 35 k, 52 k -> 36 k instructions of code).  Its result goes through a local
// of the rare branch: handing it `&out.full` put the whole QuadRes — the common path's values — into scratch, 22 % slower (profiles/r05_segments_queues_sweep.txt).
// With the local: one chain +1.5 %, three segments unchanged.
template <class F>
__device__ __noinline__ void quad_add_rare(const XYZZ<F>* A, const XYZZ<F>* B, XYZZ<F>* out) { XYZZ<F> a = *A; add_full(a, *B); *out = a; }
template <class F>
__device__ __forceinline__ QuadRes<F> quad_add_compute(const XYZZ<F>* __restrict__ sh, uint32_t ia, uint32_t ib, bool active) {
  const int q = (int)(threadIdx.x & 3u);
  QuadRes<F> out; out.mode = 0;
  const XYZZ<F>& A = sh[ia]; const XYZZ<F>& B = sh[ib];
  const bool a_id = A.ZZ.is_zero(), b_id = B.ZZ.is_zero();
  // (operands are selected per lane, then ONE multiplication is issued per level: a select between four products would make every
  //  lane compute all four)
  auto sel = [](bool c, const F& x, const F& y) { F r; for (int i = 0; i < 9; i++) r.v[i] = c ? x.v[i] : y.v[i]; return r; };
  // level 1: U1 = X1·ZZ2 | U2 = X2·ZZ1 | S1 = Y1·ZZZ2 | S2 = Y2·ZZZ1
  const F* px = q == 0 ? &A.X : q == 1 ? &B.X : q == 2 ? &A.Y : &B.Y;
  const F* py = q == 0 ? &B.ZZ : q == 1 ? &A.ZZ : q == 2 ? &B.ZZZ : &A.ZZZ;
  const F m1 = F::mul(*px, *py);
  F o1;
#pragma unroll
  for (int i = 0; i < 9; i++) o1.v[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m1.v[i], 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
  // q0,q1: P = U2 − U1;  q2,q3: R = S2 − S1   (the even lane of a pair holds the "1" operand)
  const F lo = sel(q & 1, o1, m1), hi = sel(q & 1, m1, o1);
  const F PR = F::template sub<2>(hi, lo);
  const int pz = __builtin_amdgcn_update_dpp(0, (q == 0 && PR.is_zero_mod()) ? 1 : 0, 0x00, 0xf, 0xf, true);
  // level 2: PP = P² | Z12 = ZZ1·ZZ2 | RR = R² | Z123 = ZZZ1·ZZZ2
  const F* pa = q == 1 ? &A.ZZ : &A.ZZZ; const F* pb = q == 1 ? &B.ZZ : &B.ZZZ;
  const F za = *pa, zb = *pb;
  const F m2 = F::mul(sel(q & 1, za, PR), sel(q & 1, zb, PR));
  const F PP = quad_bcast<0>(m2);
  const F Z12 = quad_bcast<1>(m2);
  // level 3: PPP = P·PP | Q = U1·PP | — | ZZ3 = Z12·PP
  const F m3 = F::mul(sel(q == 0, PR, sel(q == 1, lo, Z12)), PP);
  const F PPP = quad_bcast<0>(m3);
  const F Qv = quad_bcast<1>(m3);
  const F S1 = quad_bcast<2>(lo);              // lane 2's `lo` is S1
  // X3 and Q − X3 (meaningful on lane 2, whose m2 is RR)
  const F X3 = F::template sub<4>(m2, F::add(PPP, F::dbl(Qv)));
  const F D = F::template sub<6>(Qv, X3);
  // level 4: Yb = S1·PPP | — | Ya = R·(Q − X3) | ZZZ3 = Z123·PPP
  const F m4 = F::mul(sel(q == 0, S1, sel(q == 2, PR, m2)), sel(q == 2, D, PPP));
  const F Yb = quad_bcast<0>(m4);
  if (!active || b_id) return out;
  if (a_id) { out.mode = 1; out.c0 = q == 0 ? B.X : q == 1 ? B.Y : q == 2 ? B.ZZ : B.ZZZ; return out; }
  if (pz) {                                          // same x: doubling or cancellation — rare, the scalar formula on lane 0
    out.mode = 3;
    if (q == 0) { XYZZ<F> tmp; quad_add_rare<F>(&A, &B, &tmp); out.full = tmp; }
    return out;
  }
  out.mode = 2;
  if (q == 2) { out.c0 = X3; out.c1 = F::template sub<2>(m4, Yb); }
  if (q == 3) { out.c0 = m3; out.c1 = m4; }
  return out;
}
template <class F>
__device__ __forceinline__ void quad_add_store(XYZZ<F>* __restrict__ sh, uint32_t ia, const QuadRes<F>& r) {
  const int q = (int)(threadIdx.x & 3u);
  XYZZ<F>& A = sh[ia];
  if (r.mode == 1) { if (q == 0) A.X = r.c0; else if (q == 1) A.Y = r.c0; else if (q == 2) A.ZZ = r.c0; else A.ZZZ = r.c0; }
  else if (r.mode == 2) { if (q == 2) { A.X = r.c0; A.Y = r.c1; } else if (q == 3) { A.ZZ = r.c0; A.ZZZ = r.c1; } }
  else if (r.mode == 3 && q == 0) A = r.full;
}
// One level of `count` (<= 64) additions sh[dst(e)] += sh[src(e)] by a 256-thread workgroup, four lanes each.  A wave none of whose
// quads has an addition at this level skips the arithmetic (the shuffles are wave-wide, the barriers workgroup-wide): the last five
// levels of a 256-leaf tree issue on one wave instead of four — the tails are issue-bound per SIMD and, with three proofs sharing
// the GPU, every slot a waiting wave does not burn goes to another kernel.
template <class F, class Fd, class Fs>
__device__ __forceinline__ void quad_level(XYZZ<F>* __restrict__ sh, uint32_t count, Fd dst, Fs src) {
  const uint32_t e = threadIdx.x >> 2;
  const bool active = e < count;
  QuadRes<F> r; r.mode = 0;
  uint32_t ia = 0;
  // (the wave index through readfirstlane: a condition the compiler KNOWS to be scalar becomes a branch; a per-lane one was
  //  flattened into predicated code that the idle waves still issued — same instruction count as before)
  const uint32_t wave0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
  if (wave0 < 4u * count) {                            // this wave holds at least one active quad
    ia = active ? dst(e) : 0u;
    const uint32_t ib = active ? src(e) : 0u;
    r = quad_add_compute<F>(sh, ia, ib, active);
  }
  __syncthreads();
  quad_add_store<F>(sh, ia, r);
  __syncthreads();
}

// sh[0] = sum of sh[0..256) by a 256-thread workgroup: eight levels, every addition a four-lane one (two rounds for the 128 pairs
// of the first level).  Ends with a barrier.
template <class F>
__device__ __forceinline__ void quad_tree256(XYZZ<F>* __restrict__ sh) {
  quad_level<F>(sh, 64, [](uint32_t e) { return e; }, [](uint32_t e) { return e + 128; });
  quad_level<F>(sh, 64, [](uint32_t e) { return e + 64; }, [](uint32_t e) { return e + 192; });
  for (uint32_t d = 64; d > 0; d >>= 1) quad_level<F>(sh, d, [](uint32_t e) { return e; }, [d](uint32_t e) { return e + d; });
}

}  // namespace vz
