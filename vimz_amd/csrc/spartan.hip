// CompressedSNARK::{prove, verify} for the Nova IVC object (reference call sites
// vimz/src/nova_snark_backend/mod.rs:52-67: `CompressedSNARK::<_, _, _, _, S<G1>, S<G2>>::{setup, prove, verify}` with
// nova_scotia::S = nova-snark 0.23.0's `spartan::snark::RelaxedR1CSSNARK<G, ipa_pc::EvaluationEngine<G>>`).  README.md:196 counts
// this step into the total proof time; after folding takes seconds it is the larger part of it (≈13 s on the reference's CPU).
//
// nova-snark is not vendored, so — like the augmented circuits — this is our own statement of the same
// construction, NOT byte-compatible with the crate's proof object: Spartan's two sum-checks for relaxed R1CS followed by
// inner-product-argument openings of the committed vectors, made non-interactive over a SHA3-256 transcript.  An independent
// verifier written against this description lives in tests/_spartan.py (the oracle side); the product's own verifier is below.
//
// For one relaxed instance  U = (comm_W, comm_E, u, X0, X1)  with witness vectors Z (n wires: u at wire 0, X at the last two) and
// E (m rows) over shape (A, B, C), padded to M = 2^s rows and N = 2^t columns:
//   1. tau in F^s from the transcript;  outer sum-check (degree 3, s rounds, top variable first) of
//        sum_x eq(tau, x) · (Az(x)·Bz(x) − u·Cz(x) − E(x)) = 0          -> point rx, claims va, vb, vc, ve (= Az, Bz, Cz, E at rx)
//   2. rho from the transcript;  inner sum-check (degree 2, t rounds) of
//        va + rho·vb + rho²·vc = sum_y M(y)·Z(y),   M(y) = sum_x eq(rx, x)·(A + rho·B + rho²·C)(x, y)      -> point ry
//      the verifier evaluates M(ry) from the sparse matrices itself and needs Z(ry) = W'(ry) + u·eq(ry,0) + X0·eq(ry,n−2) + X1·eq(ry,n−1),
//      W' = Z with the three public entries zeroed — the vector comm_W commits to (bases: wire i ↦ ck[i−1]);
//   3. inner-product arguments:  <W', eq(ry,·)> = eval_W under comm_W  and  <E, eq(rx,·)> = ve under comm_E.
// IPA (Bulletproofs, inverse-free folding): P = <a,G> + <a,b>·U';  per round  L = <a_L,G_R> + <a_L,b_R>·U',  R = <a_R,G_L> + <a_R,b_L>·U',
// challenge x,  a' = x·a_L + a_R,  b' = b_L + x·b_R,  G' = G_L + x·G_R,  P' = R + x·P + x²·L;  final check P = a·G + a·b·U'.
// All challenges are 128 bits.  The compressed proof of an IVC = the two running instances, the last fresh secondary instance and
// one such argument for each of the three (the fresh one with u = 1, E = 0), plus (steps, z_0, z_n) for the hash checks.
//
// GPU work: (A,B,C)·Z and the transposed products by the SpMV kernels of the fold (r1cs_ops.hpp), the sum-check rounds by
// reduction + bind kernels over vectors resident in HBM (one 96-byte read-back per round), the IPA rounds by the Pippenger MSM
// of the fold (msm.hpp) and a per-point double-and-add kernel for the generator folding.
#include "ivc_internal.hpp"
#include "ec_mem.hpp"
#include "proof_io.hpp"
#include "merge_internal.hpp"
#ifdef VIMZ_TESTING
#include "../../include/vimz_hip_testing.h"
#endif

namespace {

// ---- kernels -------------------------------------------------------------------------------------------------------------------------
template <class F, int K>
__device__ __forceinline__ void block_reduce_store(F (&acc)[K], uint32_t* __restrict__ partial) {
  __shared__ uint32_t sh[K * 256 * 8];
  const uint32_t t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < K; k++)
#pragma unroll
    for (int w = 0; w < 8; w++) sh[(k * 256 + t) * 8 + w] = acc[k].v[w];
  __syncthreads();
  for (uint32_t d = 128; d > 0; d >>= 1) {
    if (t < d) {
#pragma unroll
      for (int k = 0; k < K; k++) {
        F a, b;
#pragma unroll
        for (int w = 0; w < 8; w++) { a.v[w] = sh[(k * 256 + t) * 8 + w]; b.v[w] = sh[(k * 256 + t + d) * 8 + w]; }
        a = F::add(a, b);
#pragma unroll
        for (int w = 0; w < 8; w++) sh[(k * 256 + t) * 8 + w] = a.v[w];
      }
    }
    __syncthreads();
  }
  if (t == 0)
#pragma unroll
    for (int k = 0; k < K; k++) { F a; for (int w = 0; w < 8; w++) a.v[w] = sh[(k * 256) * 8 + w]; store_fe(partial, (size_t)blockIdx.x * K + k, a); }
}

// out[k] = sum over blocks of partial[blk][k]   (one workgroup)
template <class F, int K>
__global__ void __launch_bounds__(256) k_sum_partials(const uint32_t* __restrict__ partial, uint32_t nblocks, uint32_t* __restrict__ out) {
  F acc[K];
  for (int k = 0; k < K; k++) acc[k] = F::zero();
  for (uint32_t b = threadIdx.x; b < nblocks; b += 256)
    for (int k = 0; k < K; k++) acc[k] = F::add(acc[k], load_fe<F>(partial, (size_t)b * K + k));
  block_reduce_store<F, K>(acc, out);
}

// eq table doubling: the newly processed variable becomes the top bit of the index
template <class F>
__global__ void __launch_bounds__(256) k_eq_step(uint32_t* __restrict__ tab, size_t len, F tau) {
  VZ_GRID_STRIDE(i, len) { const F lo = load_fe<F>(tab, i), hi = F::mul(lo, tau); store_fe(tab, i + len, hi); store_fe(tab, i, F::sub(lo, hi)); }
}

// outer round: evaluations at 0, 2, 3 of the round polynomial (pairs (i, i + half): the top remaining variable is bound)
template <class F>
__global__ void __launch_bounds__(256) k_sc_outer(const uint32_t* __restrict__ eq, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                  const uint32_t* __restrict__ c, const uint32_t* __restrict__ e, size_t half, F u, uint32_t* __restrict__ partial) {
  F acc[3] = {F::zero(), F::zero(), F::zero()};
  VZ_GRID_STRIDE(i, half) {
    F q0 = load_fe<F>(eq, i), a0 = load_fe<F>(a, i), b0 = load_fe<F>(b, i), c0 = load_fe<F>(c, i), e0 = e ? load_fe<F>(e, i) : F::zero();
    const F q1 = load_fe<F>(eq, i + half), a1 = load_fe<F>(a, i + half), b1 = load_fe<F>(b, i + half), c1 = load_fe<F>(c, i + half), e1 = e ? load_fe<F>(e, i + half) : F::zero();
    const F dq = F::sub(q1, q0), da = F::sub(a1, a0), db = F::sub(b1, b0), dc = F::sub(c1, c0), de = F::sub(e1, e0);
    acc[0] = F::add(acc[0], F::mul(q0, F::sub(F::sub(F::mul(a0, b0), F::mul(u, c0)), e0)));
    q0 = F::add(q1, dq); a0 = F::add(a1, da); b0 = F::add(b1, db); c0 = F::add(c1, dc); e0 = F::add(e1, de);      // t = 2
    acc[1] = F::add(acc[1], F::mul(q0, F::sub(F::sub(F::mul(a0, b0), F::mul(u, c0)), e0)));
    q0 = F::add(q0, dq); a0 = F::add(a0, da); b0 = F::add(b0, db); c0 = F::add(c0, dc); e0 = F::add(e0, de);      // t = 3
    acc[2] = F::add(acc[2], F::mul(q0, F::sub(F::sub(F::mul(a0, b0), F::mul(u, c0)), e0)));
  }
  block_reduce_store<F, 3>(acc, partial);
}
// inner round: evaluations at 0 and 2 of m(t)·z(t)
template <class F>
__global__ void __launch_bounds__(256) k_sc_inner(const uint32_t* __restrict__ m, const uint32_t* __restrict__ z, size_t half, uint32_t* __restrict__ partial) {
  F acc[2] = {F::zero(), F::zero()};
  VZ_GRID_STRIDE(i, half) {
    const F m0 = load_fe<F>(m, i), z0 = load_fe<F>(z, i), m1 = load_fe<F>(m, i + half), z1 = load_fe<F>(z, i + half);
    acc[0] = F::add(acc[0], F::mul(m0, z0));
    acc[1] = F::add(acc[1], F::mul(F::add(m1, F::sub(m1, m0)), F::add(z1, F::sub(z1, z0))));
  }
  block_reduce_store<F, 2>(acc, partial);
}
struct BindSet { uint32_t* p[5]; int n; };
template <class F>
__global__ void __launch_bounds__(256) k_bind(BindSet s, size_t half, F r) {
  for (int v = 0; v < s.n; v++) {
    uint32_t* x = s.p[v];
    if (!x) continue;
    VZ_GRID_STRIDE(i, half) { const F lo = load_fe<F>(x, i), hi = load_fe<F>(x, i + half); store_fe(x, i, F::add(lo, F::mul(r, F::sub(hi, lo)))); }
  }
}
// <a_L, b_R> and <a_R, b_L>
template <class F>
__global__ void __launch_bounds__(256) k_dot2(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t half, uint32_t* __restrict__ partial) {
  F acc[2] = {F::zero(), F::zero()};
  VZ_GRID_STRIDE(i, half) {
    acc[0] = F::add(acc[0], F::mul(load_fe<F>(a, i), load_fe<F>(b, i + half)));
    acc[1] = F::add(acc[1], F::mul(load_fe<F>(a, i + half), load_fe<F>(b, i)));
  }
  block_reduce_store<F, 2>(acc, partial);
}
template <class F>
__global__ void __launch_bounds__(256) k_ipa_fold_vec(uint32_t* __restrict__ a, uint32_t* __restrict__ b, size_t half, F x) {
  VZ_GRID_STRIDE(i, half) {
    store_fe(a, i, F::add(F::mul(x, load_fe<F>(a, i)), load_fe<F>(a, i + half)));
    store_fe(b, i, F::add(load_fe<F>(b, i), F::mul(x, load_fe<F>(b, i + half))));
  }
}
// G[i] <- G[i] + x·G[i + half]   (x: 128 bits; affine points in the resident 9x29 form, identity = (0,0))
template <class G>
__global__ void __launch_bounds__(256) k_ipa_fold_bases(uint32_t* __restrict__ bases, size_t half, uint4 xw) {
  const uint32_t x[4] = {xw.x, xw.y, xw.z, xw.w};
  VZ_GRID_STRIDE(i, half) {
    const Affine<G> lo = load_affine<G>(bases, (uint32_t)i), hi = load_affine<G>(bases, (uint32_t)(i + half));
    XYZZ<G> acc = XYZZ<G>::identity();
    for (int bit = 127; bit >= 0; bit--) {
      acc = dbl(acc);
      if ((x[bit >> 5] >> (bit & 31)) & 1u) add_mixed(acc, hi);
    }
    add_mixed(acc, lo);
    const Affine<G> r = to_affine(acc);
    store_affine<G>(bases, i, r);
  }
}
// out = x + r·y + r2·z
template <class F>
__global__ void __launch_bounds__(256) k_combine3(size_t n, const uint32_t* __restrict__ x, const uint32_t* __restrict__ y, const uint32_t* __restrict__ z, F r, F r2, uint32_t* __restrict__ out) {
  VZ_GRID_STRIDE(i, n) store_fe(out, i, F::add(load_fe<F>(x, i), F::add(F::mul(r, load_fe<F>(y, i)), F::mul(r2, load_fe<F>(z, i)))));
}
// out[0..N): the committed part of Z (public entries zeroed), zero-padded
template <class F>
__global__ void __launch_bounds__(256) k_committed_part(const uint32_t* __restrict__ Z, size_t nw, size_t N, uint32_t* __restrict__ out) {
  VZ_GRID_STRIDE(i, N) store_fe(out, i, (i == 0 || i + 2 >= nw) ? F::zero() : load_fe<F>(Z, i));
}

// b of the W opening: eq(ry, ·) with the public slots (wire 0 = u, the last two wires = X0, X1) and the padding above the last wire
// zeroed.  Those positions of the opened vector are NOT part of what comm_W may say about Z: their generators are live and honest
// commitments never use them, so without the mask a prover could hide a correction of (u, X0, X1) there and prove the relation for
// a z whose public entries differ from the instance's.
template <class F>
__global__ void __launch_bounds__(256) k_mask_public(uint32_t* __restrict__ b, size_t nw, size_t N) {
  VZ_GRID_STRIDE(i, N) if (i == 0 || i + 2 >= nw) store_fe(b, i, F::zero());
}

// ---- per-side resident data -----------------------------------------------------------------------------------------------------------
struct SideDev {
  CsrDev A{}, B{}, C{}; const uint32_t* dict = nullptr; const uint32_t* items = nullptr; uint32_t n_long = 0, n_med = 0;     // the shape
  CsrDev At{}, Bt{}, Ct{}; const uint32_t* items_t = nullptr; uint32_t n_long_t = 0, n_med_t = 0;                            // its transpose
  uint32_t n_w = 0, n_c = 0, s = 0, t = 0;
  size_t M = 0, N = 0;
  uint32_t *eq = nullptr, *va = nullptr, *vb = nullptr, *vc = nullptr, *ve = nullptr, *vm = nullptr, *vz = nullptr;   // max(M, N) elements each
  uint32_t* G = nullptr;        // max(M, N) affine points, folded in place by the IPA
  uint32_t* partial = nullptr;  // reduction scratch
  uint32_t* pin = nullptr;      // pinned read-back (a few elements)
};
struct SpartanCache { SideDev side[2]; std::vector<void*> owned; MsmWorkspace ws; G1Aff ipa_u1; G2Aff ipa_u2; bool ready = false; };

constexpr unsigned RED_BLOCKS = 512;

uint32_t ceil_log2(size_t x) { uint32_t k = 0; while (((size_t)1 << k) < x) k++; return k; }

template <class F>
hipError_t build_transpose(const cb::Csr* M3[3], uint32_t n_rows, uint32_t n_cols, SideDev& S, std::vector<void*>& owned) {
  std::vector<uint32_t> items;
  std::vector<std::vector<uint32_t>> cnts(3);
  for (int m = 0; m < 3; m++) {
    const cb::Csr& Mx = *M3[m];
    std::vector<uint32_t> ptr(n_cols + 1, 0), col(Mx.col.size()), coef(Mx.col.size());
    for (uint32_t c : Mx.col) ptr[c + 1]++;
    for (uint32_t c = 0; c < n_cols; c++) ptr[c + 1] += ptr[c];
    std::vector<uint32_t> cur(ptr.begin(), ptr.end() - 1);
    for (uint32_t r = 0; r < n_rows; r++)
      for (uint32_t k = Mx.row_ptr[r]; k < Mx.row_ptr[r + 1]; k++) { const uint32_t pos = cur[Mx.col[k]]++; col[pos] = r; coef[pos] = Mx.coef[k]; }
    for (uint32_t c = 0; c < n_cols; c++) if (ptr[c + 1] - ptr[c] > SPMV_LONG) items.push_back(((uint32_t)m << 30) | c);
    cnts[m].resize(n_cols);
    for (uint32_t c = 0; c < n_cols; c++) cnts[m][c] = ptr[c + 1] - ptr[c];
    CsrDev& D = m == 0 ? S.At : m == 1 ? S.Bt : S.Ct;
    hipError_t e;
    if ((e = upload(ptr, &D.row_ptr)) != hipSuccess) return e; owned.push_back((void*)D.row_ptr);
    if ((e = upload(col, &D.col)) != hipSuccess) return e; if (D.col) owned.push_back((void*)D.col);
    if ((e = upload(coef, &D.coef)) != hipSuccess) return e; if (D.coef) owned.push_back((void*)D.coef);
  }
  S.n_med_t = spmv_sort_items(items, [&](uint32_t it) { return cnts[it >> 30][it & 0x3fffffffu]; });
  S.n_long_t = (uint32_t)items.size();
  hipError_t e = upload(items, &S.items_t);
  if (S.items_t) owned.push_back((void*)S.items_t);
  return e;
}

template <class F>
void spmv_any(hipStream_t s, const CsrDev& A, const CsrDev& B, const CsrDev& C, const uint32_t* dict, const uint32_t* items, uint32_t n_long, uint32_t n_med,
              size_t nrows, const uint32_t* z, uint32_t* az, uint32_t* bz, uint32_t* cz) {
  hipLaunchKernelGGL(k_spmv3<F>, dim3(stream_grid(3 * nrows)), dim3(256), 0, s, A, B, C, dict, nrows, z, az, bz, cz);
  if (n_long) hipLaunchKernelGGL(k_spmv_long<F>, dim3(spmv_long_blocks(n_long, n_med)), dim3(256), 0, s, A, B, C, dict, items, n_long, n_med, z, az, bz, cz);
}

void spartan_release(vimz_ivc* v) {
  auto* c = (SpartanCache*)v->spartan_cache;
  if (!c) return;
  if (v->ctx) {
    std::lock_guard<std::mutex> g(v->ctx->mu);
    hipSetDevice(v->ctx->device);
    hipStreamSynchronize(v->ctx->stream);
    for (void* d : c->owned) hipFree(d);
    for (auto& s : c->side) if (s.pin) hipHostFree(s.pin);
    c->ws.release();
  }
  delete c;
  v->spartan_cache = nullptr;
}

// an independent generator for the inner-product term (nobody knows its discrete log with respect to the key)
template <class C>
int derive_ipa_u(vimz_ctx* ctx, int curve, Affine<typename C::Base>* out) {
  vimz_bases* b = nullptr;
  int rc = vimz_bases_generate(ctx, curve, "vimz-ipa-u", 10, 1, &b);
  if (rc) return rc;
  uint64_t xy[8];
  rc = vimz_bases_download(ctx, b, 0, xy, 1, VIMZ_FORM_MONTGOMERY);
  vimz_bases_free(ctx, b);
  if (rc) return rc;
  memcpy(out->x.v, xy, 32); memcpy(out->y.v, xy + 4, 32);
  return VIMZ_OK;
}

// CompressedSNARK::setup: transposed shapes and scratch on the device (caller does NOT hold the lock)
int spartan_setup(vimz_ivc* v, SpartanCache** out) {
  vimz_ctx* ctx = v->ctx;
  if (v->spartan_cache) { *out = (SpartanCache*)v->spartan_cache; return VIMZ_OK; }
  std::unique_ptr<SpartanCache> c(new SpartanCache());
  int rc;
  if ((rc = derive_ipa_u<BnG1>(ctx, VIMZ_CURVE_BN254_G1, &c->ipa_u1))) return rc;
  if ((rc = derive_ipa_u<Grumpkin>(ctx, VIMZ_CURVE_GRUMPKIN, &c->ipa_u2))) return rc;
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  vimz_prover* p = v->pri;
  const cb::Builder& b1 = v->circ1->build->b; const cb::BuilderT<Fq>& b2 = v->c2.b;
  for (int side = 0; side < 2; side++) {
    SideDev& S = c->side[side];
    if (side == 0) {
      S.A = p->A; S.B = p->B; S.C = p->C; S.dict = p->dict; S.n_w = p->n_wires; S.n_c = p->n_c;
      const cb::Csr* M3[3] = {&b1.A, &b1.B, &b1.C};
      P_TRY(build_transpose<Fe>(M3, S.n_c, S.n_w, S, c->owned));
    } else {
      S.A = v->sec.A; S.B = v->sec.B; S.C = v->sec.C; S.dict = v->sec.dict; S.items = v->sec.long_items; S.n_long = v->sec.n_long; S.n_med = v->sec.n_med;
      S.n_w = v->sec.n_w; S.n_c = v->sec.n_c;
      const cb::Csr* M3[3] = {&b2.A, &b2.B, &b2.C};
      P_TRY(build_transpose<Fq>(M3, S.n_c, S.n_w, S, c->owned));
    }
    S.s = ceil_log2(S.n_c); S.t = ceil_log2(S.n_w); S.M = (size_t)1 << S.s; S.N = (size_t)1 << S.t;
    const size_t big = std::max(S.M, S.N);
    if ((side == 0 ? v->ck1->n : v->ck2->n) < big) return vz_fail(ctx, VIMZ_ERR_INVALID, "compress: the commitment key is shorter than the padded shape (needs next_pow2(max(wires, constraints)) generators)");
    uint32_t** vecs[] = {&S.eq, &S.va, &S.vb, &S.vc, &S.ve, &S.vm, &S.vz};
    for (auto pv : vecs) { P_TRY(hipMalloc((void**)pv, 32 * big)); c->owned.push_back(*pv); }
    P_TRY(hipMalloc((void**)&S.G, 4 * (size_t)AFFINE_WORDS * big)); c->owned.push_back(S.G);
    P_TRY(hipMalloc((void**)&S.partial, 32 * 3 * (RED_BLOCKS + 1))); c->owned.push_back(S.partial);
    P_TRY(hipHostMalloc((void**)&S.pin, 32 * 8));
  }
  *out = c.get();
  v->spartan_cache = c.release();
  v->spartan_free = spartan_release;
  return VIMZ_OK;
}

template <class F>
F interp_cubic(const F& s0, const F& s1, const F& s2, const F& s3, const F& r) {     // Lagrange through t = 0, 1, 2, 3
  const F one = F::one(), two = F::dbl(one), three = F::add(two, one);
  const F inv2 = F::pow_pm2(two), inv6 = F::pow_pm2(F::mul(two, three));
  const F r1 = F::sub(r, one), r2 = F::sub(r, two), r3 = F::sub(r, three);
  const F l0 = F::neg(F::mul(F::mul(F::mul(r1, r2), r3), inv6));                  // (r−1)(r−2)(r−3)/(−6)
  const F l1 = F::mul(F::mul(F::mul(r, r2), r3), inv2);                           // r(r−2)(r−3)/2
  const F l2 = F::neg(F::mul(F::mul(F::mul(r, r1), r3), inv2));                   // r(r−1)(r−3)/(−2)
  const F l3 = F::mul(F::mul(F::mul(r, r1), r2), inv6);                           // r(r−1)(r−2)/6
  return F::add(F::add(F::mul(l0, s0), F::mul(l1, s1)), F::add(F::mul(l2, s2), F::mul(l3, s3)));
}
template <class F>
F interp_quad(const F& s0, const F& s1, const F& s2, const F& r) {                  // through t = 0, 1, 2
  const F one = F::one(), two = F::dbl(one), inv2 = F::pow_pm2(two);
  const F r1 = F::sub(r, one), r2 = F::sub(r, two);
  const F l0 = F::mul(F::mul(r1, r2), inv2), l1 = F::neg(F::mul(r, r2)), l2 = F::mul(F::mul(r, r1), inv2);
  return F::add(F::add(F::mul(l0, s0), F::mul(l1, s1)), F::mul(l2, s2));
}

// One relaxed instance of one side, as the prover and the verifier see it.
template <class F, class FS>
struct Instance { Affine<FS> cW, cE; F u, X0, X1; bool has_E; };

template <class F, class FS>
void absorb_instance(Transcript& tr, const F& digest, const Instance<F, FS>& I, uint32_t side_tag) {
  const uint64_t tag = side_tag; tr.absorb_words("side", &tag, 1);
  tr.absorb_fe("digest", digest);
  tr.absorb_fe("cWx", I.cW.x); tr.absorb_fe("cWy", I.cW.y); tr.absorb_fe("cEx", I.cE.x); tr.absorb_fe("cEy", I.cE.y);
  tr.absorb_fe("u", I.u); tr.absorb_fe("X0", I.X0); tr.absorb_fe("X1", I.X1);
}

// ---- prover ----------------------------------------------------------------------------------------------------------------------------
template <class F, class C>
int ipa_prove(vimz_ctx* ctx, SpartanCache& cache, SideDev& S, hipStream_t s, Transcript& tr, const Affine<typename C::Base>& Ugen, uint32_t* a, uint32_t* b,
              uint32_t rounds, const Affine<typename C::Base>& comm, const F& claim, Writer& out) {
  typedef typename C::Base FS;
  typedef typename C::Coord G;
  tr.absorb_fe("ipaP.x", comm.x); tr.absorb_fe("ipaP.y", comm.y); tr.absorb_fe("ipaC", claim);
  uint32_t r0[4]; tr.challenge(r0);
  const Affine<FS> Up = to_affine(host_mul<FS>(Ugen, r0, 128));
  size_t len = (size_t)1 << rounds;
  for (uint32_t j = 0; j < rounds; j++) {
    const size_t half = len >> 1;
    const unsigned gb = (unsigned)std::min<size_t>(RED_BLOCKS, (half + 255) / 256);
    hipLaunchKernelGGL(k_dot2<F>, dim3(gb), dim3(256), 0, s, (const uint32_t*)a, (const uint32_t*)b, half, S.partial);
    hipLaunchKernelGGL((k_sum_partials<F, 2>), dim3(1), dim3(256), 0, s, (const uint32_t*)S.partial, gb, S.partial + 8 * 3 * RED_BLOCKS);
    P_TRY(hipMemcpyAsync(S.pin, S.partial + 8 * 3 * RED_BLOCKS, 64, hipMemcpyDeviceToHost, s));
    Affine<FS> Lg, Rg;
    P_TRY(msm_run<C>(s, cache.ws, S.G + (size_t)AFFINE_WORDS * half, a, half, 1, 0, &Lg, nullptr, nullptr, 0, nullptr));          // <a_L, G_R>
    P_TRY(msm_run<C>(s, cache.ws, S.G, a + 8 * half, half, 1, 0, &Rg, nullptr, nullptr, 0, nullptr));                             // <a_R, G_L>
    F cl, cr; memcpy(cl.v, S.pin, 32); memcpy(cr.v, S.pin + 8, 32);              // (the MSMs synchronised the stream)
    XYZZ<FS> L = from_affine(Lg); { XYZZ<FS> t = host_mul_fe<FS, F>(Up, cl); add_full(L, t); }
    XYZZ<FS> R = from_affine(Rg); { XYZZ<FS> t = host_mul_fe<FS, F>(Up, cr); add_full(R, t); }
    const Affine<FS> La = to_affine(L), Ra = to_affine(R);
    out.point(La); out.point(Ra);
    tr.absorb_fe("L.x", La.x); tr.absorb_fe("L.y", La.y); tr.absorb_fe("R.x", Ra.x); tr.absorb_fe("R.y", Ra.y);
    uint32_t xw[4]; tr.challenge(xw);
    F xc = F::zero(); for (int k = 0; k < 4; k++) xc.v[k] = xw[k];
    const F x = F::to_mont(xc);
    hipLaunchKernelGGL(k_ipa_fold_vec<F>, dim3(stream_grid(half)), dim3(256), 0, s, a, b, half, x);
    hipLaunchKernelGGL(k_ipa_fold_bases<G>, dim3(stream_grid(half)), dim3(256), 0, s, S.G, half, make_uint4(xw[0], xw[1], xw[2], xw[3]));
    P_TRY(hipGetLastError());
    len = half;
  }
  F afin;
  P_TRY(hipMemcpyAsync(S.pin, a, 32, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  memcpy(afin.v, S.pin, 32);
  out.fe(afin);
  tr.absorb_fe("a", afin);
  return VIMZ_OK;
}

// bases of the W opening: index i (wire i) -> ck[i-1], index 0 -> ck[N-1]; of the E opening: ck[0..M)
hipError_t load_bases(hipStream_t s, uint32_t* G, const vimz_bases* ck, size_t n, bool shifted) {
  if (!shifted) return hipMemcpyAsync(G, ck->d, 4 * (size_t)AFFINE_WORDS * n, hipMemcpyDeviceToDevice, s);
  hipError_t e = hipMemcpyAsync(G + AFFINE_WORDS, ck->d, 4 * (size_t)AFFINE_WORDS * (n - 1), hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return e;
  return hipMemcpyAsync(G, ck->d + (size_t)AFFINE_WORDS * (n - 1), 4 * (size_t)AFFINE_WORDS, hipMemcpyDeviceToDevice, s);
}

template <class F, class C, class Fwd>
int spartan_prove(vimz_ctx* ctx, SpartanCache& cache, SideDev& S, hipStream_t s, const vimz_bases* ck, const Affine<typename C::Base>& Ugen, const F& digest,
                  const Instance<F, typename C::Base>& I, const uint32_t* Z, const uint32_t* E, uint32_t side_tag, Transcript& tr, Writer& out, Fwd fwd,
                  bool forge_public_slot = false) {
  const size_t M = S.M, N = S.N, big = std::max(M, N);
  absorb_instance(tr, digest, I, side_tag);
  // vectors: (A,B,C)·Z padded to M, E padded to M
  P_TRY(hipMemsetAsync(S.va, 0, 32 * big, s)); P_TRY(hipMemsetAsync(S.vb, 0, 32 * big, s)); P_TRY(hipMemsetAsync(S.vc, 0, 32 * big, s)); P_TRY(hipMemsetAsync(S.ve, 0, 32 * big, s));
  fwd(Z, S.va, S.vb, S.vc);
  if (E) P_TRY(hipMemcpyAsync(S.ve, E, 32 * (size_t)S.n_c, hipMemcpyDeviceToDevice, s));
  // eq(tau, ·)
  std::vector<F> tau(S.s);
  for (auto& x : tau) x = tr.challenge_fe<F>();
  { const F one = F::one(); P_TRY(hipMemcpyAsync(S.eq, one.v, 32, hipMemcpyHostToDevice, s)); P_TRY(hipStreamSynchronize(s)); }
  for (uint32_t k = S.s; k-- > 0;) hipLaunchKernelGGL(k_eq_step<F>, dim3(stream_grid((size_t)1 << (S.s - 1 - k))), dim3(256), 0, s, S.eq, (size_t)1 << (S.s - 1 - k), tau[k]);
  // outer sum-check
  F claim = F::zero();
  std::vector<F> rx(S.s);
  size_t len = M;
  for (uint32_t j = 0; j < S.s; j++) {
    const size_t half = len >> 1;
    const unsigned gb = (unsigned)std::min<size_t>(RED_BLOCKS, (half + 255) / 256);
    hipLaunchKernelGGL(k_sc_outer<F>, dim3(gb), dim3(256), 0, s, (const uint32_t*)S.eq, (const uint32_t*)S.va, (const uint32_t*)S.vb, (const uint32_t*)S.vc,
                       E ? (const uint32_t*)S.ve : (const uint32_t*)nullptr, half, I.u, S.partial);
    hipLaunchKernelGGL((k_sum_partials<F, 3>), dim3(1), dim3(256), 0, s, (const uint32_t*)S.partial, gb, S.partial + 8 * 3 * RED_BLOCKS);
    P_TRY(hipMemcpyAsync(S.pin, S.partial + 8 * 3 * RED_BLOCKS, 96, hipMemcpyDeviceToHost, s));
    P_TRY(hipStreamSynchronize(s));
    F s0, s2, s3; memcpy(s0.v, S.pin, 32); memcpy(s2.v, S.pin + 8, 32); memcpy(s3.v, S.pin + 16, 32);
    out.fe(s0); out.fe(s2); out.fe(s3);
    tr.absorb_fe("o0", s0); tr.absorb_fe("o2", s2); tr.absorb_fe("o3", s3);
    const F r = tr.challenge_fe<F>();
    rx[j] = r;
    claim = interp_cubic(s0, F::sub(claim, s0), s2, s3, r);
    BindSet bs{{S.eq, S.va, S.vb, S.vc, E ? S.ve : nullptr}, 5};
    hipLaunchKernelGGL(k_bind<F>, dim3(stream_grid(half)), dim3(256), 0, s, bs, half, r);
    len = half;
  }
  F cl[4];
  { uint32_t* src[4] = {S.va, S.vb, S.vc, S.ve};
    for (int k = 0; k < 4; k++) P_TRY(hipMemcpyAsync(S.pin + 8 * k, src[k], 32, hipMemcpyDeviceToHost, s));
    P_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < 4; k++) memcpy(cl[k].v, S.pin + 8 * k, 32);
    if (!E) cl[3] = F::zero(); }
  for (int k = 0; k < 4; k++) { out.fe(cl[k]); tr.absorb_fe("claim", cl[k]); }
  const F rho = tr.challenge_fe<F>(), rho2 = F::sqr(rho);
  // eq(rx, ·) over the rows, then M = A^T e + rho B^T e + rho^2 C^T e over the columns
  { const F one = F::one(); P_TRY(hipMemcpyAsync(S.eq, one.v, 32, hipMemcpyHostToDevice, s)); P_TRY(hipStreamSynchronize(s)); }
  for (uint32_t k = S.s; k-- > 0;) hipLaunchKernelGGL(k_eq_step<F>, dim3(stream_grid((size_t)1 << (S.s - 1 - k))), dim3(256), 0, s, S.eq, (size_t)1 << (S.s - 1 - k), rx[k]);
  spmv_any<F>(s, S.At, S.Bt, S.Ct, S.dict, S.items_t, S.n_long_t, S.n_med_t, S.n_w, S.eq, S.va, S.vb, S.vc);
  P_TRY(hipMemsetAsync(S.vm, 0, 32 * big, s)); P_TRY(hipMemsetAsync(S.vz, 0, 32 * big, s));
  hipLaunchKernelGGL(k_combine3<F>, dim3(stream_grid(S.n_w)), dim3(256), 0, s, (size_t)S.n_w, (const uint32_t*)S.va, (const uint32_t*)S.vb, (const uint32_t*)S.vc, rho, rho2, S.vm);
  P_TRY(hipMemcpyAsync(S.vz, Z, 32 * (size_t)S.n_w, hipMemcpyDeviceToDevice, s));
  claim = F::add(cl[0], F::add(F::mul(rho, cl[1]), F::mul(rho2, cl[2])));
  std::vector<F> ry(S.t);
  len = N;
  for (uint32_t j = 0; j < S.t; j++) {
    const size_t half = len >> 1;
    const unsigned gb = (unsigned)std::min<size_t>(RED_BLOCKS, (half + 255) / 256);
    hipLaunchKernelGGL(k_sc_inner<F>, dim3(gb), dim3(256), 0, s, (const uint32_t*)S.vm, (const uint32_t*)S.vz, half, S.partial);
    hipLaunchKernelGGL((k_sum_partials<F, 2>), dim3(1), dim3(256), 0, s, (const uint32_t*)S.partial, gb, S.partial + 8 * 3 * RED_BLOCKS);
    P_TRY(hipMemcpyAsync(S.pin, S.partial + 8 * 3 * RED_BLOCKS, 64, hipMemcpyDeviceToHost, s));
    P_TRY(hipStreamSynchronize(s));
    F s0, s2; memcpy(s0.v, S.pin, 32); memcpy(s2.v, S.pin + 8, 32);
    out.fe(s0); out.fe(s2);
    tr.absorb_fe("i0", s0); tr.absorb_fe("i2", s2);
    const F r = tr.challenge_fe<F>();
    ry[j] = r;
    claim = interp_quad(s0, F::sub(claim, s0), s2, r);
    BindSet bs{{S.vm, S.vz, nullptr, nullptr, nullptr}, 2};
    hipLaunchKernelGGL(k_bind<F>, dim3(stream_grid(half)), dim3(256), 0, s, bs, half, r);
    len = half;
  }
  // Z(ry) is what the bound vector holds; the committed part's evaluation is Z(ry) minus the three public entries' share
  F zr;
  P_TRY(hipMemcpyAsync(S.pin, S.vz, 32, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  memcpy(zr.v, S.pin, 32);
  auto eq_at = [&](const std::vector<F>& pt, size_t idx) {          // eq(pt, idx), pt[0] = the top bit
    F acc = F::one();
    const size_t k = pt.size();
    for (size_t q = 0; q < k; q++) { const bool bit = (idx >> (k - 1 - q)) & 1; acc = F::mul(acc, bit ? pt[q] : F::sub(F::one(), pt[q])); }
    return acc;
  };
  const F evalW = F::sub(zr, F::add(F::mul(I.u, eq_at(ry, 0)), F::add(F::mul(I.X0, eq_at(ry, S.n_w - 2)), F::mul(I.X1, eq_at(ry, S.n_w - 1)))));
  out.fe(evalW);
  tr.absorb_fe("evalW", evalW);
  // opening of the committed part of Z at ry:  a = W' (into va), b = eq(ry, ·) (into vb), bases shifted by one wire
  int rc;
  hipLaunchKernelGGL(k_committed_part<F>, dim3(stream_grid(N)), dim3(256), 0, s, Z, (size_t)S.n_w, N, S.va);
  { const F one = F::one(); P_TRY(hipMemcpyAsync(S.vb, one.v, 32, hipMemcpyHostToDevice, s)); P_TRY(hipStreamSynchronize(s)); }
  for (uint32_t k = S.t; k-- > 0;) hipLaunchKernelGGL(k_eq_step<F>, dim3(stream_grid((size_t)1 << (S.t - 1 - k))), dim3(256), 0, s, S.vb, (size_t)1 << (S.t - 1 - k), ry[k]);
  hipLaunchKernelGGL(k_mask_public<F>, dim3(stream_grid(N)), dim3(256), 0, s, S.vb, (size_t)S.n_w, N);
  if (forge_public_slot) {      // test hook (vimz_ivc_compress): the opened vector carries X0 − X0' = −1 in the slot of wire n−2
    const F m1 = F::neg(F::one());
    P_TRY(hipMemcpyAsync(S.va + 8 * (size_t)(S.n_w - 2), m1.v, 32, hipMemcpyHostToDevice, s)); P_TRY(hipStreamSynchronize(s));
  }
  P_TRY(load_bases(s, S.G, ck, N, true));
  if ((rc = ipa_prove<F, C>(ctx, cache, S, s, tr, Ugen, S.va, S.vb, S.t, I.cW, evalW, out))) return rc;
  // opening of E at rx
  if (E) {
    P_TRY(hipMemsetAsync(S.va, 0, 32 * M, s));
    P_TRY(hipMemcpyAsync(S.va, E, 32 * (size_t)S.n_c, hipMemcpyDeviceToDevice, s));
    { const F one = F::one(); P_TRY(hipMemcpyAsync(S.vb, one.v, 32, hipMemcpyHostToDevice, s)); P_TRY(hipStreamSynchronize(s)); }
    for (uint32_t k = S.s; k-- > 0;) hipLaunchKernelGGL(k_eq_step<F>, dim3(stream_grid((size_t)1 << (S.s - 1 - k))), dim3(256), 0, s, S.vb, (size_t)1 << (S.s - 1 - k), rx[k]);
    P_TRY(load_bases(s, S.G, ck, M, false));
    if ((rc = ipa_prove<F, C>(ctx, cache, S, s, tr, Ugen, S.va, S.vb, S.s, I.cE, cl[3], out))) return rc;
  }
  P_TRY(hipGetLastError());
  return VIMZ_OK;
}

// ---- verifier (host arithmetic; one MSM per opening on the GPU) ----------------------------------------------------------------------------
template <class F>
std::vector<F> eq_table(const std::vector<F>& pt) {       // eq(pt, ·), pt[0] = the top bit of the index
  const size_t k = pt.size();
  std::vector<F> tab((size_t)1 << k);
  tab[0] = F::one();
  size_t len = 1;
  for (size_t q = k; q-- > 0;) { for (size_t i = 0; i < len; i++) { const F hi = F::mul(tab[i], pt[q]); tab[i + len] = hi; tab[i] = F::sub(tab[i], hi); } len <<= 1; }
  return tab;
}

template <class F, class C>
bool ipa_verify(vimz_ctx* ctx, SpartanCache& cache, SideDev& S, hipStream_t s, Transcript& tr, const Affine<typename C::Base>& Ugen, const vimz_bases* ck, bool shifted,
                const std::vector<F>& b, uint32_t rounds, const Affine<typename C::Base>& comm, const F& claim, Reader& in) {
  typedef typename C::Base FS;
  tr.absorb_fe("ipaP.x", comm.x); tr.absorb_fe("ipaP.y", comm.y); tr.absorb_fe("ipaC", claim);
  uint32_t r0[4]; tr.challenge(r0);
  const Affine<FS> Up = to_affine(host_mul<FS>(Ugen, r0, 128));
  XYZZ<FS> P = from_affine(comm); { XYZZ<FS> t = host_mul_fe<FS, F>(Up, claim); add_full(P, t); }
  std::vector<F> xs(rounds);
  for (uint32_t j = 0; j < rounds; j++) {
    const Affine<FS> L = in.point<FS>(), R = in.point<FS>();
    if (!in.ok) return false;
    tr.absorb_fe("L.x", L.x); tr.absorb_fe("L.y", L.y); tr.absorb_fe("R.x", R.x); tr.absorb_fe("R.y", R.y);
    uint32_t xw[4]; tr.challenge(xw);
    F xc = F::zero(); for (int k = 0; k < 4; k++) xc.v[k] = xw[k];
    xs[j] = F::to_mont(xc);
    // P <- R + x·P + x²·L
    const Affine<FS> Pa = to_affine(P);
    XYZZ<FS> acc = from_affine(R);
    { XYZZ<FS> t = host_mul<FS>(Pa, xw, 128); add_full(acc, t); }
    { XYZZ<FS> t = host_mul_fe<FS, F>(L, F::sqr(xs[j])); add_full(acc, t); }
    P = acc;
  }
  const F afin = in.fe<F>();
  if (!in.ok) return false;
  tr.absorb_fe("a", afin);
  // s_i = product of the challenges of the rounds in which index i sat in the right half (round 0 = top bit)
  const size_t n = (size_t)1 << rounds;
  std::vector<F> sv(n);
  sv[0] = F::one();
  size_t len = 1;
  for (uint32_t q = rounds; q-- > 0;) { for (size_t i = 0; i < len; i++) sv[i + len] = F::mul(sv[i], xs[q]); len <<= 1; }
  F bfin = F::zero();
  for (size_t i = 0; i < n; i++) bfin = F::add(bfin, F::mul(sv[i], b[i]));
  Affine<FS> Gfin;
  if (hipMemcpyAsync(S.va, sv.data(), 32 * n, hipMemcpyHostToDevice, s) != hipSuccess) return false;
  if (load_bases(s, S.G, ck, n, shifted) != hipSuccess) return false;
  if (msm_run<C>(s, cache.ws, S.G, S.va, n, 1, 0, &Gfin, nullptr, nullptr, 0, nullptr) != hipSuccess) return false;
  XYZZ<FS> rhs = host_mul_fe<FS, F>(Gfin, afin);
  { XYZZ<FS> t = host_mul_fe<FS, F>(Up, F::mul(afin, bfin)); add_full(rhs, t); }
  const Affine<FS> lhs_a = to_affine(P), rhs_a = to_affine(rhs);
  return lhs_a.x.eq(rhs_a.x) && lhs_a.y.eq(rhs_a.y);
}

template <class F, class C, class B>
bool spartan_verify(vimz_ctx* ctx, SpartanCache& cache, SideDev& S, hipStream_t s, const vimz_bases* ck, const Affine<typename C::Base>& Ugen, const F& digest,
                    const Instance<F, typename C::Base>& I, const B& shape /* host builder: A, B, C as CSR + dict */, uint32_t side_tag, Transcript& tr, Reader& in) {
  absorb_instance(tr, digest, I, side_tag);
  std::vector<F> tau(S.s);
  for (auto& x : tau) x = tr.challenge_fe<F>();
  F claim = F::zero();
  std::vector<F> rx(S.s);
  for (uint32_t j = 0; j < S.s; j++) {
    const F s0 = in.fe<F>(), s2 = in.fe<F>(), s3 = in.fe<F>();
    if (!in.ok) return false;
    tr.absorb_fe("o0", s0); tr.absorb_fe("o2", s2); tr.absorb_fe("o3", s3);
    rx[j] = tr.challenge_fe<F>();
    claim = interp_cubic(s0, F::sub(claim, s0), s2, s3, rx[j]);
  }
  F cl[4];
  for (int k = 0; k < 4; k++) cl[k] = in.fe<F>();
  if (!in.ok) return false;
  if (!I.has_E && !cl[3].is_zero()) return false;
  {
    F e = F::one();
    for (uint32_t k = 0; k < S.s; k++) e = F::mul(e, F::add(F::mul(tau[k], rx[k]), F::mul(F::sub(F::one(), tau[k]), F::sub(F::one(), rx[k]))));
    const F want = F::mul(e, F::sub(F::sub(F::mul(cl[0], cl[1]), F::mul(I.u, cl[2])), cl[3]));
    if (!want.eq(claim)) return false;
  }
  for (int k = 0; k < 4; k++) tr.absorb_fe("claim", cl[k]);
  const F rho = tr.challenge_fe<F>(), rho2 = F::sqr(rho);
  claim = F::add(cl[0], F::add(F::mul(rho, cl[1]), F::mul(rho2, cl[2])));
  std::vector<F> ry(S.t);
  for (uint32_t j = 0; j < S.t; j++) {
    const F s0 = in.fe<F>(), s2 = in.fe<F>();
    if (!in.ok) return false;
    tr.absorb_fe("i0", s0); tr.absorb_fe("i2", s2);
    ry[j] = tr.challenge_fe<F>();
    claim = interp_quad(s0, F::sub(claim, s0), s2, ry[j]);
  }
  const F evalW = in.fe<F>();
  if (!in.ok) return false;
  tr.absorb_fe("evalW", evalW);
  const std::vector<F> ex = eq_table(rx), ey = eq_table(ry);
  // M(ry) = sum over the non-zeros of (A + rho B + rho^2 C) of eq(rx, row) · eq(ry, col) · value
  F vm = F::zero();
  {
    const cb::Csr* Ms[3] = {&shape.A, &shape.B, &shape.C};
    const F w[3] = {F::one(), rho, rho2};
    for (int m = 0; m < 3; m++) {
      F acc = F::zero();
      for (uint32_t r = 0; r + 1 < Ms[m]->row_ptr.size(); r++) {
        F row = F::zero();
        for (uint32_t k = Ms[m]->row_ptr[r]; k < Ms[m]->row_ptr[r + 1]; k++) row = F::add(row, F::mul(shape.dict[Ms[m]->coef[k]], ey[Ms[m]->col[k]]));
        acc = F::add(acc, F::mul(row, ex[r]));
      }
      vm = F::add(vm, F::mul(w[m], acc));
    }
  }
  const F vz = F::add(evalW, F::add(F::mul(I.u, ey[0]), F::add(F::mul(I.X0, ey[S.n_w - 2]), F::mul(I.X1, ey[S.n_w - 1]))));
  if (!F::mul(vm, vz).eq(claim)) return false;
  {   // the W opening is over the committed positions only: b = eq(ry, ·) with the public slots and the padding zeroed (k_mask_public)
    std::vector<F> eyW(ey);
    for (size_t i = 0; i < eyW.size(); i++) if (i == 0 || i + 2 >= S.n_w) eyW[i] = F::zero();
    if (!ipa_verify<F, C>(ctx, cache, S, s, tr, Ugen, ck, true, eyW, S.t, I.cW, evalW, in)) return false;
  }
  if (I.has_E && !ipa_verify<F, C>(ctx, cache, S, s, tr, Ugen, ck, false, ex, S.s, I.cE, cl[3], in)) return false;
  return true;
}

const uint64_t CSNARK_MAGIC = 0x314e5343565aull;   // "ZVCSN1"

struct Loaded {          // the instances a compressed proof is about, as both provers and verifiers hold them
  RelaxedInst<Fq> U1; RelaxedInst<Fe> U2; FreshInst<Fe> u2;
  Fe u1_run; Fq u2_run;
};

void bind_statement(Transcript& tr, uint64_t steps, const std::vector<Fe>& z0, const std::vector<Fe>& zn) {
  tr.absorb_words("steps", &steps, 1);
  for (auto& z : z0) tr.absorb_fe("z0", z);
  for (auto& z : zn) tr.absorb_fe("zn", z);
}

template <class F> F u256_to_fe(const U256w& x) { return from_u256<F>(x); }

Instance<Fe, Fq> inst_primary(const Loaded& L) { Instance<Fe, Fq> I; I.cW = L.U1.W; I.cE = L.U1.E; I.u = L.u1_run; I.X0 = u256_to_fe<Fe>(L.U1.X0); I.X1 = u256_to_fe<Fe>(L.U1.X1); I.has_E = true; return I; }
Instance<Fq, Fe> inst_secondary(const Loaded& L) { Instance<Fq, Fe> I; I.cW = L.U2.W; I.cE = L.U2.E; I.u = L.u2_run; I.X0 = u256_to_fe<Fq>(L.U2.X0); I.X1 = u256_to_fe<Fq>(L.U2.X1); I.has_E = true; return I; }
Instance<Fq, Fe> inst_fresh(const Loaded& L) {
  Instance<Fq, Fe> I; I.cW = L.u2.W; I.cE.x = Fe::zero(); I.cE.y = Fe::zero(); I.u = Fq::one();
  I.X0 = cross_field<Fq>(L.u2.x0); I.X1 = cross_field<Fq>(L.u2.x1); I.has_E = false; return I;
}

}  // namespace

extern "C" {

// words of a compressed proof for this IVC's shapes
size_t vimz_ivc_compressed_size(vimz_ivc* v) {
  if (!v) return 0;
  const uint32_t s1 = ceil_log2(v->pri->n_c), t1 = ceil_log2(v->pri->n_wires), s2 = ceil_log2(v->sec.n_c), t2 = ceil_log2(v->sec.n_w);
  auto snark = [](size_t s, size_t t, bool e) { return 4 * (3 * s + 4 + 2 * t + 1 + (4 * t + 1) * 2 / 2 + (e ? 4 * s + 1 : 0)) + 4 * (4 * t); };
  (void)snark;
  auto words = [](size_t s, size_t t, bool e) { return 4 * (3 * s + 4 + 2 * t + 1) + 4 * (4 * t + 1) + (e ? 4 * (4 * s + 1) : 0); };
  return 8 * (8 + 4 * 2 * (size_t)v->pri->len_z + 4 * (7 + 7 + 4) + words(s1, t1, true) + words(s2, t2, true) + words(s2, t2, false));
}

// CompressedSNARK::prove (mod.rs:56-59): blob receives vimz_ivc_compressed_size(v) bytes.  seconds (optional): {setup, prove}.
#ifdef VIMZ_TESTING
static std::atomic<bool> g_forge_public_slot{false};
void vimz_test_forge_public_slot(int on) { g_forge_public_slot.store(on != 0); }
#endif
int vimz_ivc_compress(vimz_ivc* v, uint8_t* blob, size_t cap, double seconds[2]) {
  if (!v || !blob) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = v->ctx;
  if (cap < vimz_ivc_compressed_size(v)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_compress: buffer too small");
  if (v->i == 0) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_compress: nothing has been folded");
  double t0 = now_s();
  SpartanCache* cache = nullptr;
  int rc = spartan_setup(v, &cache);
  if (rc) return rc;
  const double t_setup = now_s() - t0;
  t0 = now_s();
  // Test hook for the negative test of the public-slot binding (tests/test_gpu_compress.py): a cheating prover that claims x0 + 1 for
  // the last fresh instance and hides the difference in the generator of that wire's slot.  Exists only in libvimz_hip_testing.so
  // (vimz_test_forge_public_slot, include/vimz_hip_testing.h); the product library has no such prover.
#ifdef VIMZ_TESTING
  const bool forge = g_forge_public_slot.load();
#else
  const bool forge = false;
#endif
  G2Aff forge_gen; forge_gen.x = forge_gen.y = Fe::zero();
  if (forge) {
    uint64_t xy[8];
    if ((rc = vimz_bases_download(ctx, v->ck2, v->sec.n_w - 3, xy, 1, VIMZ_FORM_MONTGOMERY))) return rc;
    memcpy(forge_gen.x.v, xy, 32); memcpy(forge_gen.y.v, xy + 4, 32);
  }
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  P_TRY(hipStreamSynchronize(s));
  vimz_prover* p = v->pri;
  Loaded L; L.U1 = v->U1; L.U2 = v->U2; L.u2 = v->u2; L.u1_run = v->u1_run; L.u2_run = v->u2_run;
  if (forge) {      // comm' = comm − ck[n−3] commits to the same W with −1 in the slot of wire n−2;  x0' = x0 + 1
    G2 a = from_affine(L.u2.W); G2Aff neg = forge_gen; neg.y = Fe::neg(neg.y); add_mixed(a, neg); L.u2.W = to_affine(a);
    L.u2.x0 = Fe::add(L.u2.x0, Fe::one());
  }
  Writer out;
  out.word(CSNARK_MAGIC); out.word(v->i); out.word(p->len_z); out.word(cache->side[0].s); out.word(cache->side[0].t); out.word(cache->side[1].s); out.word(cache->side[1].t); out.word(0);
  for (auto& z : v->z0) out.fe(z);
  for (auto& z : p->z_cur) out.fe(z);
  out.point(L.U1.W); out.point(L.U1.E); out.fe(L.U1.u); out.u256(L.U1.X0); out.u256(L.U1.X1);
  out.point(L.U2.W); out.point(L.U2.E); out.fe(L.U2.u); out.u256(L.U2.X0); out.u256(L.U2.X1);
  out.point(L.u2.W); out.fe(L.u2.x0); out.fe(L.u2.x1);
  Transcript tr("vimz-compressed-snark-v1");
  bind_statement(tr, v->i, v->z0, p->z_cur);
  tr.absorb_words("inst", out.w.data() + 8 + 8 * p->len_z, 4 * (7 + 7 + 4));
  auto fwd1 = [&](const uint32_t* z, uint32_t* az, uint32_t* bz, uint32_t* cz) { launch_spmv(p, s, z, az, bz, cz, 0); };
  auto fwd2 = [&](const uint32_t* z, uint32_t* az, uint32_t* bz, uint32_t* cz) { sec_spmv<Fq>(v->sec, s, z, az, bz, cz); };
  if ((rc = spartan_prove<Fe, BnG1>(ctx, *cache, cache->side[0], s, v->ck1, cache->ipa_u1, v->c1->digest, inst_primary(L), p->Zrun, p->E, 1, tr, out, fwd1))) return rc;
  if ((rc = spartan_prove<Fq, Grumpkin>(ctx, *cache, cache->side[1], s, v->ck2, cache->ipa_u2, v->c2.digest, inst_secondary(L), v->sec.Zrun, v->sec.E, 2, tr, out, fwd2))) return rc;
  if ((rc = spartan_prove<Fq, Grumpkin>(ctx, *cache, cache->side[1], s, v->ck2, cache->ipa_u2, v->c2.digest, inst_fresh(L), v->sec.z2, nullptr, 3, tr, out, fwd2, forge))) return rc;
  if (8 * out.w.size() != vimz_ivc_compressed_size(v)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_compress: internal size mismatch");
  memcpy(blob, out.w.data(), 8 * out.w.size());
  if (seconds) { seconds[0] = t_setup; seconds[1] = now_s() - t0; }
  return VIMZ_OK;
}

// CompressedSNARK::verify(vk, num_steps, z0_primary, z0_secondary) (mod.rs:63-67).  `v` supplies the verifier key: an IVC object
// created for the same step circuit and commitment keys (its folding state is neither read nor changed).
// result: 0 = accepted; bit 0 / 1 the primary / secondary chain hash; bit 2 / 3 / 4 the argument for the primary running /
// secondary running / last fresh secondary instance; bit 12 the statement (steps, z0) differs; bit 13 malformed proof.
int vimz_ivc_verify_compressed(vimz_ivc* v, const uint8_t* blob, size_t len, uint64_t num_steps, const uint64_t* z0, uint32_t* result) {
  if (!v || !blob || !z0 || !result) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = v->ctx;
  SpartanCache* cache = nullptr;
  int rc = spartan_setup(v, &cache);
  if (rc) return rc;
  if (len != vimz_ivc_compressed_size(v) || (len & 7)) { *result = 8192; return VIMZ_OK; }
  std::vector<uint64_t> words(len / 8); memcpy(words.data(), blob, len);
  Reader in{words.data(), words.size()};
  vimz_prover* p = v->pri;
  uint32_t res = 0;
  const uint64_t magic = in.word(), steps = in.word(), lz = in.word(), s1 = in.word(), t1 = in.word(), s2 = in.word(), t2 = in.word(); in.word();
  if (magic != CSNARK_MAGIC || lz != p->len_z || s1 != cache->side[0].s || t1 != cache->side[0].t || s2 != cache->side[1].s || t2 != cache->side[1].t) { *result = 8192; return VIMZ_OK; }
  std::vector<Fe> z0p(lz), zn(lz), z0c(lz);
  for (auto& z : z0p) z = in.fe<Fe>();
  for (auto& z : zn) z = in.fe<Fe>();
  for (uint32_t k = 0; k < lz; k++) { Fe c; memcpy(c.v, z0 + 4 * k, 32); if (!c.is_reduced()) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_verify_compressed: z0 element not below the modulus"); z0c[k] = Fe::to_mont(c); }
  if (steps != num_steps || steps == 0) res |= 4096;
  for (uint32_t k = 0; k < lz; k++) if (!z0c[k].eq(z0p[k])) res |= 4096;
  Loaded L;
  L.U1.W = in.point<Fq>(); L.U1.E = in.point<Fq>(); L.U1.u = in.fe<Fq>(); L.U1.X0 = in.u256(); L.U1.X1 = in.u256();
  L.U2.W = in.point<Fe>(); L.U2.E = in.point<Fe>(); L.U2.u = in.fe<Fe>(); L.U2.X0 = in.u256(); L.U2.X1 = in.u256();
  L.u2.W = in.point<Fe>(); L.u2.x0 = in.fe<Fe>(); L.u2.x1 = in.fe<Fe>();
  if (!in.ok) { *result = 8192; return VIMZ_OK; }
  L.u1_run = cross_field<Fe>(L.U1.u); L.u2_run = cross_field<Fq>(L.U2.u);      // u is a small integer: the same in both fields
  // the two chain hashes, recomputed from the verifier's own digests and the CLAIMED z0
  {
    Fe h1 = instance_hash_native<BnFr>(v->c1->digest, steps, z0c, zn, L.U2);
    if (!h1.eq(L.u2.x0)) res |= 1;
    std::vector<Fq> zq = {Fq::zero()};
    Fq h2 = instance_hash_native<BnFq>(v->c2.digest, steps, zq, zq, L.U1);
    if (!cross_field<Fe>(h2).eq(L.u2.x1)) res |= 2;
  }
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  Transcript tr("vimz-compressed-snark-v1");
  bind_statement(tr, steps, z0p, zn);
  tr.absorb_words("inst", words.data() + 8 + 8 * lz, 4 * (7 + 7 + 4));
  const cb::Builder& b1 = v->circ1->build->b; const cb::BuilderT<Fq>& b2 = v->c2.b;
  bool ok1 = false, ok2 = false, ok3 = false;
  try {
    ok1 = spartan_verify<Fe, BnG1>(ctx, *cache, cache->side[0], s, v->ck1, cache->ipa_u1, v->c1->digest, inst_primary(L), b1, 1, tr, in);
    ok2 = ok1 && spartan_verify<Fq, Grumpkin>(ctx, *cache, cache->side[1], s, v->ck2, cache->ipa_u2, v->c2.digest, inst_secondary(L), b2, 2, tr, in);
    ok3 = ok2 && spartan_verify<Fq, Grumpkin>(ctx, *cache, cache->side[1], s, v->ck2, cache->ipa_u2, v->c2.digest, inst_fresh(L), b2, 3, tr, in);
  } catch (const std::exception& e) { return vz_fail(ctx, VIMZ_ERR_INVALID, e.what()); }
  if (!ok1) res |= 4;
  if (ok1 && !ok2) res |= 8;
  if (ok1 && ok2 && !ok3) res |= 16;
  if (!in.ok || (ok3 && in.pos != in.n)) res |= 8192;
  *result = res;
  return VIMZ_OK;
}

// ---- CompressedSNARK for a merged proof (merge.hip): one argument for the folded primary instance, one for the folded secondary one.
// Blob: {magic, words of the records} ‖ records (segments' statements and instances, the fold tree's cross-term commitments) ‖ the two
// arguments.  The verifier replays the records (every segment's hash checks, adjacency, the folds) and verifies the arguments for the
// instances IT computed.
static const uint64_t CMERGED_MAGIC = 0x31474d43565aull;   // "ZVCMG1"

static size_t arg_words(size_t s, size_t t, bool e) { return 4 * (3 * s + 4 + 2 * t + 1) + 4 * (4 * t + 1) + (e ? 4 * (4 * s + 1) : 0); }

size_t vimz_ivc_merged_compressed_size(const vimz_ivc_merged* m) {
  if (!m || !m->vk) return 0;
  const vimz_ivc* v = m->vk;
  const uint32_t s1 = ceil_log2(v->pri->n_c), t1 = ceil_log2(v->pri->n_wires), s2 = ceil_log2(v->sec.n_c), t2 = ceil_log2(v->sec.n_w);
  return 8 * (2 + records_words(m) + arg_words(s1, t1, true) + arg_words(s2, t2, true));
}

int vimz_ivc_merged_compress(vimz_ivc_merged* m, uint8_t* blob, size_t cap, double seconds[2]) {
  if (!m || !blob || !m->vk || m->broken) return VIMZ_ERR_INVALID;
  vimz_ivc* v = m->vk; vimz_ctx* ctx = v->ctx;
  if (cap < vimz_ivc_merged_compressed_size(m)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_compress: buffer too small");
  double t0 = now_s();
  SpartanCache* cache = nullptr;
  int rc = spartan_setup(v, &cache);
  if (rc) return rc;
  const double t_setup = now_s() - t0;
  t0 = now_s();
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  P_TRY(hipStreamSynchronize(s));
  vimz_prover* p = v->pri;
  Writer out;
  out.word(CMERGED_MAGIC); out.word(records_words(m));
  write_records(m, out);
  Transcript tr("vimz-merged-compressed-v1");
  tr.absorb_words("records", out.w.data() + 2, out.w.size() - 2);
  Instance<Fe, Fq> I1; I1.cW = m->acc.P.cW; I1.cE = m->acc.P.cE; I1.u = m->acc.P.u; I1.X0 = m->acc.P.X0; I1.X1 = m->acc.P.X1; I1.has_E = true;
  Instance<Fq, Fe> I2; I2.cW = m->acc.Q.cW; I2.cE = m->acc.Q.cE; I2.u = m->acc.Q.u; I2.X0 = m->acc.Q.X0; I2.X1 = m->acc.Q.X1; I2.has_E = true;
  auto fwd1 = [&](const uint32_t* z, uint32_t* az, uint32_t* bz, uint32_t* cz) { launch_spmv(p, s, z, az, bz, cz, 0); };
  auto fwd2 = [&](const uint32_t* z, uint32_t* az, uint32_t* bz, uint32_t* cz) { sec_spmv<Fq>(v->sec, s, z, az, bz, cz); };
  if ((rc = spartan_prove<Fe, BnG1>(ctx, *cache, cache->side[0], s, v->ck1, cache->ipa_u1, v->c1->digest, I1, m->Zp, m->Ep, 1, tr, out, fwd1))) return rc;
  if ((rc = spartan_prove<Fq, Grumpkin>(ctx, *cache, cache->side[1], s, v->ck2, cache->ipa_u2, v->c2.digest, I2, m->Zq, m->Eq, 2, tr, out, fwd2))) return rc;
  if (8 * out.w.size() != vimz_ivc_merged_compressed_size(m)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_merged_compress: internal size mismatch");
  memcpy(blob, out.w.data(), 8 * out.w.size());
  if (seconds) { seconds[0] = t_setup; seconds[1] = now_s() - t0; }
  return VIMZ_OK;
}

// verify(vk, num_steps, z0) of a compressed merged proof.  result: 0 = accepted; bit 0 / 1 a segment's primary / secondary chain hash;
// bit 2 / 3 the argument for the folded primary / secondary instance; bit 12 statement (steps, z0, adjacency); bit 13 malformed.
int vimz_ivc_verify_merged_compressed(vimz_ivc* v, const uint8_t* blob, size_t len, uint64_t num_steps, const uint64_t* z0, uint32_t* result) {
  if (!v || !blob || !z0 || !result) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = v->ctx;
  SpartanCache* cache = nullptr;
  int rc = spartan_setup(v, &cache);
  if (rc) return rc;
  if ((len & 7) || len < 16) { *result = 8192; return VIMZ_OK; }
  std::vector<uint64_t> words(len / 8); memcpy(words.data(), blob, len);
  Reader in{words.data(), words.size()};
  const uint64_t magic = in.word(), rw = in.word();
  if (magic != CMERGED_MAGIC || rw > words.size() - 2) { *result = 8192; return VIMZ_OK; }
  std::vector<MSeg> segs; std::vector<MOp> ops;
  if (!read_records(in, v, segs, ops) || in.pos != 2 + rw) { *result = 8192; return VIMZ_OK; }
  const uint32_t s1 = cache->side[0].s, t1 = cache->side[0].t, s2 = cache->side[1].s, t2 = cache->side[1].t;
  if (words.size() != 2 + rw + arg_words(s1, t1, true) + arg_words(s2, t2, true)) { *result = 8192; return VIMZ_OK; }
  uint32_t res = 0;
  MAcc R;
  if (!merged_replay(v, segs, ops, &R, &res)) { *result = res | 8192; return VIMZ_OK; }
  vimz_prover* p = v->pri;
  if (R.n != num_steps) res |= 4096;
  for (uint32_t k = 0; k < p->len_z; k++) {
    Fe c; memcpy(c.v, z0 + 4 * k, 32);
    if (!c.is_reduced()) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_ivc_verify_merged_compressed: z0 element not below the modulus");
    if (!Fe::to_mont(c).eq(R.zs[k])) res |= 4096;
  }
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  Transcript tr("vimz-merged-compressed-v1");
  tr.absorb_words("records", words.data() + 2, rw);
  Instance<Fe, Fq> I1; I1.cW = R.P.cW; I1.cE = R.P.cE; I1.u = R.P.u; I1.X0 = R.P.X0; I1.X1 = R.P.X1; I1.has_E = true;
  Instance<Fq, Fe> I2; I2.cW = R.Q.cW; I2.cE = R.Q.cE; I2.u = R.Q.u; I2.X0 = R.Q.X0; I2.X1 = R.Q.X1; I2.has_E = true;
  const cb::Builder& b1 = v->circ1->build->b; const cb::BuilderT<Fq>& b2 = v->c2.b;
  bool ok1 = false, ok2 = false;
  try {
    ok1 = spartan_verify<Fe, BnG1>(ctx, *cache, cache->side[0], s, v->ck1, cache->ipa_u1, v->c1->digest, I1, b1, 1, tr, in);
    ok2 = ok1 && spartan_verify<Fq, Grumpkin>(ctx, *cache, cache->side[1], s, v->ck2, cache->ipa_u2, v->c2.digest, I2, b2, 2, tr, in);
  } catch (const std::exception& e) { return vz_fail(ctx, VIMZ_ERR_INVALID, e.what()); }
  if (!ok1) res |= 4;
  if (ok1 && !ok2) res |= 8;
  if (!in.ok || (ok2 && in.pos != in.n)) res |= 8192;
  *result = res;
  return VIMZ_OK;
}

}  // extern "C"
