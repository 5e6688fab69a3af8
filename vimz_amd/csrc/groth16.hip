// vimz_decider_*: the decider of the Nova + CycleFold path — the Groth16 proof the reference's Sonobe backend makes of its final fold before
// the proof goes on chain (`DeciderEth<.., Groth16<Bn254>, ..>`, vimz/src/sonobe_backend/decider.rs:13-21; `Decider::preprocess` / `prove`,
// mod.rs:72-78; eight of the 25 calldata words, solidity.rs:13-27, contracts/*Verifier.sol:785-810).  Groth16 over BN254 is a published
// protocol (Groth, EUROCRYPT 2016; verification equation as the EVM's pairing precompile evaluates it); the circuit it proves here is ours
// (aug/decider.hpp), as is the deterministic TEST setup — the toxic waste is derived from a caller's seed, where Sonobe draws its keys from
// `StdRng::from_seed([41; 32])` (mod.rs:54) inside crates that are not vendored: proofs are checked by the oracle-side pairing
// (tests/_pairing.py), not by the reference's contracts.  Parity unpinned, like every layer above the step relation.
//
// On the GPU: the radix-2 NTTs over BN254 Fr that turn (A·z, B·z, C·z) into the quotient polynomial h, the three G1 multi-scalar
// multiplications of a proof over the key's queries (the Pippenger of msm.hpp), the G2 one (per-point double-and-add over Fq2 and a host sum:
// one MSM of ~10^6 points per proof, not a hot loop), and the fixed-base multiplications that make the keys.  On the host: the circuit
// (synthesis, witness, its sparse products — 10^6 rows of a few terms), the QAP evaluation at the trapdoor, the final point arithmetic.
#include <thread>
#include "cyclefold_internal.hpp"
#include "decider_view.hpp"
#include "aug/decider.hpp"
#include "vecops_api.hpp"

namespace {

// ---- Fq2 = Fq[u] / (u² + 1): coordinates of G2 (the twist y² = x³ + 3/(9 + u)) ------------------------------------------------------------
struct Fq2 {
  Fq c0, c1;
  static constexpr bool LAZY = false;
  static VZ_HD Fq2 zero() { Fq2 r; r.c0 = Fq::zero(); r.c1 = Fq::zero(); return r; }
  static VZ_HD Fq2 one() { Fq2 r; r.c0 = Fq::one(); r.c1 = Fq::zero(); return r; }
  VZ_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  VZ_HD bool is_zero_mod() const { return is_zero(); }
  VZ_HD bool eq(const Fq2& b) const { return c0.eq(b.c0) && c1.eq(b.c1); }
  VZ_HD Fq2 canon() const { return *this; }
  static VZ_HD Fq2 add(const Fq2& a, const Fq2& b) { Fq2 r; r.c0 = Fq::add(a.c0, b.c0); r.c1 = Fq::add(a.c1, b.c1); return r; }
  static VZ_HD Fq2 sub(const Fq2& a, const Fq2& b) { Fq2 r; r.c0 = Fq::sub(a.c0, b.c0); r.c1 = Fq::sub(a.c1, b.c1); return r; }
  template <int K> static VZ_HD Fq2 sub(const Fq2& a, const Fq2& b) { return sub(a, b); }
  static VZ_HD Fq2 neg(const Fq2& a) { return sub(zero(), a); }
  static VZ_HD Fq2 dbl(const Fq2& a) { return add(a, a); }
  static VZ_HD Fq2 mul(const Fq2& a, const Fq2& b) {      // Karatsuba: three base-field products
    const Fq t0 = Fq::mul(a.c0, b.c0), t1 = Fq::mul(a.c1, b.c1);
    const Fq t2 = Fq::mul(Fq::add(a.c0, a.c1), Fq::add(b.c0, b.c1));
    Fq2 r; r.c0 = Fq::sub(t0, t1); r.c1 = Fq::sub(Fq::sub(t2, t0), t1); return r;
  }
  static VZ_HD Fq2 sqr(const Fq2& a) {
    const Fq t = Fq::mul(a.c0, a.c1);
    Fq2 r; r.c0 = Fq::mul(Fq::add(a.c0, a.c1), Fq::sub(a.c0, a.c1)); r.c1 = Fq::dbl(t); return r;
  }
  static VZ_HD Fq2 pow_pm2(const Fq2& a) {      // the inverse (0 -> 0): conj(a) / (c0² + c1²)
    const Fq n = Fq::pow_pm2(Fq::add(Fq::sqr(a.c0), Fq::sqr(a.c1)));
    Fq2 r; r.c0 = Fq::mul(a.c0, n); r.c1 = Fq::neg(Fq::mul(a.c1, n)); return r;
  }
};
typedef Affine<Fq2> G2PAff;      // a point of BN254 G2
typedef XYZZ<Fq2> G2P;

Fq fq_from_dec_limbs(const uint64_t w[4]) { Fq c; memcpy(c.v, w, 32); return Fq::to_mont(c); }
G2PAff g2_generator() {
  // (10857046999023057135944570762232829481370756359578518086990519993285655852781 + 11559732032986387107991004021392285783925812861821192530917403151452391805634 u,
  //   8495653923123431417604973247489272438418190587263600148770280649306958101930 +  4082367875863433681332203403145435568316851327593401208105741076214120093531 u)
  static const uint64_t x0[4] = {0x46debd5cd992f6edull, 0x674322d4f75edaddull, 0x426a00665e5c4479ull, 0x1800deef121f1e76ull};
  static const uint64_t x1[4] = {0x97e485b7aef312c2ull, 0xf1aa493335a9e712ull, 0x7260bfb731fb5d25ull, 0x198e9393920d483aull};
  static const uint64_t y0[4] = {0x4ce6cc0166fa7daaull, 0xe3d1e7690c43d37bull, 0x4aab71808dcb408full, 0x12c85ea5db8c6debull};
  static const uint64_t y1[4] = {0x55acdadcd122975bull, 0xbc4b313370b38ef3ull, 0xec9e99ad690c3395ull, 0x090689d0585ff075ull};
  G2PAff g; g.x.c0 = fq_from_dec_limbs(x0); g.x.c1 = fq_from_dec_limbs(x1); g.y.c0 = fq_from_dec_limbs(y0); g.y.c1 = fq_from_dec_limbs(y1);
  return g;
}
G1Aff g1_generator() { G1Aff g; g.x = Fq::one(); g.y = Fq::dbl(Fq::one()); return g; }
bool g2_on_curve(const G2PAff& p) {      // y² = x³ + 3/(9 + u)
  if (aff_is_identity(p)) return true;
  Fq2 nine_u; nine_u.c0 = cb::f_from_u64<Fq>(9); nine_u.c1 = Fq::one();
  Fq2 three = Fq2::zero(); three.c0 = cb::f_from_u64<Fq>(3);
  const Fq2 b = Fq2::mul(three, Fq2::pow_pm2(nine_u));
  return Fq2::sqr(p.y).eq(Fq2::add(Fq2::mul(Fq2::sqr(p.x), p.x), b));
}

// ---- host helpers over Fr -----------------------------------------------------------------------------------------------------------------
Fe fr_pow(Fe base, const uint32_t e[8]) {
  Fe acc = Fe::one();
  for (int i = 255; i >= 0; i--) { acc = Fe::sqr(acc); if ((e[i >> 5] >> (i & 31)) & 1u) acc = Fe::mul(acc, base); }
  return acc;
}
Fe fr_pow_u64(Fe base, uint64_t e) { uint32_t w[8] = {(uint32_t)e, (uint32_t)(e >> 32), 0, 0, 0, 0, 0, 0}; return fr_pow(base, w); }
Fe fr_from_hash(const void* seed, size_t n, const char* tag) {      // an element of Fr from SHA3-256(seed ‖ tag): 248 bits, never zero in practice
  Sha3 h; h.update(seed, n); h.update(tag, strlen(tag));
  uint8_t d[32]; h.finish(d); d[31] = 0;
  Fe c; memcpy(c.v, d, 32);
  return Fe::to_mont(c);
}
// a primitive 2^k-th root of unity of Fr: 5 generates the multiplicative group, r - 1 = 2^28 · odd
Fe fr_root_of_unity(int k) {
  uint32_t e[8]; uint64_t br = 1;
  for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)BnFr::MOD.w[i] - br; e[i] = (uint32_t)d; br = (d >> 32) & 1; }      // r - 1
  for (int s = 0; s < k; s++) { for (int i = 0; i < 8; i++) e[i] = (e[i] >> 1) | (i < 7 ? e[i + 1] << 31 : 0u); }                 // >> k (exact: k <= 28)
  return fr_pow(cb::f_from_u64<Fe>(5), e);
}

// ---- kernels --------------------------------------------------------------------------------------------------------------------------------
// out[i] = s_i · G for a fixed base G: T[w][d] = d · 2^(4w) · G (64 windows of 16 multiples) turns a multiplication into at most 64 mixed additions
template <class F>
__global__ void __launch_bounds__(128) k_fixed_mul(const uint32_t* __restrict__ scalars /* canonical words */, size_t n, const Affine<F>* __restrict__ table, Affine<F>* __restrict__ out) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  for (int k = 0; k < 8; k++) s[k] = scalars[8 * i + k];
  XYZZ<F> acc = XYZZ<F>::identity();
  for (int w = 0; w < 64; w++) {
    const uint32_t d = (s[w >> 3] >> (4 * (w & 7))) & 15u;
    if (d) { const Affine<F> q = table[16 * w + d]; add_mixed(acc, q); }
  }
  out[i] = to_affine(acc);
}
// partial[t] = Σ_{i ≡ t (mod T)} s_i · P_i by double-and-add per point (scalars in Montgomery form; zero scalars and identity points cost nothing)
template <class F>
__global__ void __launch_bounds__(128) k_msm_naive(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ scalars, size_t n, XYZZ<F>* __restrict__ partial) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, T = (size_t)gridDim.x * blockDim.x;
  XYZZ<F> acc = XYZZ<F>::identity();
  for (size_t i = t; i < n; i += T) {
    Fr sm; for (int k = 0; k < 8; k++) sm.v[k] = scalars[8 * i + k];
    if (sm.is_zero()) continue;
    const Affine<F> P = bases[i];
    if (aff_is_identity(P)) continue;
    const Fr s = Fr::from_mont(sm);
    int top = 255;
    while (top > 0 && !((s.v[top >> 5] >> (top & 31)) & 1u)) top--;
    XYZZ<F> r = from_affine(P);
    for (int b = top - 1; b >= 0; b--) { r = dbl(r); if ((s.v[b >> 5] >> (b & 31)) & 1u) add_mixed(r, P); }
    add_full(acc, r);
  }
  partial[t] = acc;
}
// tw[k] = base^k
__global__ void k_pow_table(uint32_t* __restrict__ out, size_t n, Fr base) {
  const size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (k >= n) return;
  Fr acc = Fr::one(), b = base;
  for (size_t e = k; e; e >>= 1) { if (e & 1) acc = Fr::mul(acc, b); b = Fr::sqr(b); }
  store_fe(out, k, acc);
}
__global__ void k_bitrev(uint32_t* __restrict__ a, uint32_t n, int logn) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = __brev(i) >> (32 - logn);
  if (i < j) { const Fr x = load_fe<Fr>(a, i), y = load_fe<Fr>(a, j); store_fe(a, i, y); store_fe(a, j, x); }
}
// one radix-2 stage (decimation in time over bit-reversed input): butterflies of span `half`, twiddle ω^(j · n / (2 half))
__global__ void k_ntt_stage(uint32_t* __restrict__ a, uint32_t n, uint32_t half, const uint32_t* __restrict__ tw) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n / 2) return;
  const uint32_t j = t & (half - 1), i0 = ((t - j) << 1) + j, i1 = i0 + half;
  const Fr w = load_fe<Fr>(tw, (size_t)j * (n / (2 * half)));
  const Fr u = load_fe<Fr>(a, i0), v = Fr::mul(load_fe<Fr>(a, i1), w);
  store_fe(a, i0, Fr::add(u, v)); store_fe(a, i1, Fr::sub(u, v));
}
// a[i] *= s · g^i   (coset shift and its inverse; the 1/n of an inverse transform rides in s)
__global__ void k_scale_pow(uint32_t* __restrict__ a, uint32_t n, Fr s, Fr g) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr acc = s, b = g;
  for (uint32_t e = i; e; e >>= 1) { if (e & 1) acc = Fr::mul(acc, b); b = Fr::sqr(b); }
  store_fe(a, i, Fr::mul(load_fe<Fr>(a, i), acc));
}
// a[i] = (a[i]·b[i] − c[i]) · zinv
__global__ void k_quotient(uint32_t* __restrict__ a, const uint32_t* __restrict__ b, const uint32_t* __restrict__ c, uint32_t n, Fr zinv) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  store_fe(a, i, Fr::mul(Fr::sub(Fr::mul(load_fe<Fr>(a, i), load_fe<Fr>(b, i)), load_fe<Fr>(c, i)), zinv));
}

// ---- the key --------------------------------------------------------------------------------------------------------------------------------
struct G16Key {
  uint32_t m = 0, n_pub = 0, n_c = 0, n = 0; int logn = 0;      // wires, public inputs (without the constant), constraints, domain
  Fe omega, omega_inv, n_inv, coset, coset_inv, zinv;           // domain constants (coset generator 5: Z(5·ω^i) = 5^n − 1)
  uint32_t *tw = nullptr, *tw_inv = nullptr;                    // ω^k, ω^-k for k < n/2
  vimz_bases *a_q = nullptr, *b1_q = nullptr, *l_q = nullptr, *h_q = nullptr;      // G1 queries (a, b: m points; l: private wires; h: n − 1)
  G2PAff* b2_q = nullptr;                                       // G2 query (m points, device)
  G1Aff alpha1, beta1, delta1; G2PAff beta2, gamma2, delta2;
  std::vector<G1Aff> ic;                                        // n_pub + 1 points
};
}  // namespace

struct vimz_decider {
  vimz_cf* vk = nullptr; vimz_ctx* ctx = nullptr;
  aug::DeciderCircuit circ;
  G16Key key;
  std::vector<uint8_t> seed;
  double setup_s[4] = {0, 0, 0, 0};      // circuit synthesis, QAP evaluation at the trapdoor (host), key points (GPU), total
};

namespace {

template <class F> std::vector<Affine<F>> fixed_table(const Affine<F>& g) {      // T[w][d] = d · 2^(4w) · g on the host
  std::vector<Affine<F>> T(64 * 16);
  XYZZ<F> base = from_affine(g);
  for (int w = 0; w < 64; w++) {
    const Affine<F> b = to_affine(base);
    XYZZ<F> acc = XYZZ<F>::identity();
    T[16 * w].x = F::zero(); T[16 * w].y = F::zero();
    for (int d = 1; d < 16; d++) { add_mixed(acc, b); T[16 * w + d] = to_affine(acc); }
    for (int k = 0; k < 4; k++) base = dbl(base);
  }
  return T;
}
template <class F> XYZZ<F> host_mul_fr(const Affine<F>& p, const Fe& k_mont) {
  const Fe c = Fe::from_mont(k_mont);
  XYZZ<F> acc = XYZZ<F>::identity();
  for (int i = 255; i >= 0; i--) { acc = dbl(acc); if ((c.v[i >> 5] >> (i & 31)) & 1u) add_mixed(acc, p); }
  return acc;
}

// scalars (Montgomery, host) -> s_i · G as device affine points (standard Montgomery coordinates)
template <class F>
hipError_t fixed_base_batch(hipStream_t s, const std::vector<Fe>& sc, const Affine<F>* d_table, Affine<F>** d_out) {
  const size_t n = sc.size();
  std::vector<uint32_t> canon(8 * std::max<size_t>(n, 1));
  for (size_t i = 0; i < n; i++) { const Fe c = Fe::from_mont(sc[i]); memcpy(&canon[8 * i], c.v, 32); }
  uint32_t* d_sc = nullptr;
  hipError_t e = hipMalloc((void**)&d_sc, 32 * std::max<size_t>(n, 1)); if (e != hipSuccess) return e;
  e = hipMalloc((void**)d_out, sizeof(Affine<F>) * std::max<size_t>(n, 1)); if (e != hipSuccess) { hipFree(d_sc); return e; }
  e = hipMemcpyAsync(d_sc, canon.data(), 32 * n, hipMemcpyHostToDevice, s);
  if (e == hipSuccess && n) hipLaunchKernelGGL(k_fixed_mul<F>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, d_sc, n, d_table, *d_out);
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(d_sc);
  return e;
}
// device std-form G1 points -> a commitment key in the MSM's resident form
int bases_from_device(vimz_ctx* ctx, const G1Aff* d_pts, size_t n, vimz_bases** out) {
  vimz_bases* b = new vimz_bases(); b->curve = VIMZ_CURVE_BN254_G1; b->n = n; b->d = nullptr;
  if (n) {
    hipError_t e = hipMalloc(&b->d, 4 * (size_t)AFFINE_WORDS * n);
    if (e != hipSuccess) { delete b; return vz_fail(ctx, VIMZ_ERR_HIP, "decider: hipMalloc(query)", e); }
    launch_points_to_internal<Fq>(ctx->stream, (const uint32_t*)d_pts, 0, b->d, n);
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { hipFree(b->d); delete b; return vz_fail(ctx, VIMZ_ERR_HIP, "decider: query conversion", e); }
  }
  *out = b;
  return VIMZ_OK;
}

hipError_t ntt(hipStream_t s, uint32_t* d, const G16Key& K, bool inverse) {
  hipLaunchKernelGGL(k_bitrev, dim3((K.n + 255) / 256), dim3(256), 0, s, d, K.n, K.logn);
  for (uint32_t half = 1; half < K.n; half <<= 1)
    hipLaunchKernelGGL(k_ntt_stage, dim3((K.n / 2 + 255) / 256), dim3(256), 0, s, d, K.n, half, (const uint32_t*)(inverse ? K.tw_inv : K.tw));
  return hipGetLastError();
}

void free_key(vimz_ctx* ctx, G16Key& K) {
  (void)ctx;      // (not vimz_bases_free: it takes the context's lock, which callers on an error path hold)
  for (vimz_bases** b : {&K.a_q, &K.b1_q, &K.l_q, &K.h_q}) if (*b) {
    if ((*b)->d) hipFree((*b)->d);
    if ((*b)->tables) hipFree((*b)->tables);
    for (auto& t : (*b)->small) { if (t.rows) hipFree(t.rows); if (t.mult) hipFree(t.mult); }
    delete *b; *b = nullptr;
  }
  if (K.b2_q) hipFree(K.b2_q); K.b2_q = nullptr;
  if (K.tw) hipFree(K.tw); if (K.tw_inv) hipFree(K.tw_inv); K.tw = K.tw_inv = nullptr;
}

// (A,B,C)·z of a builder's CSR on the host threads
void host_spmv3(const cb::BuilderT<Fe>& b, const std::vector<Fe>& z, std::vector<Fe>* out /* [3] of n_c */) {
  const uint32_t nc = b.n_constraints();
  const cb::Csr* Ms[3] = {&b.A, &b.B, &b.C};
  for (int m = 0; m < 3; m++) out[m].assign(nc, Fe::zero());
  const unsigned T = std::max(1u, std::min(16u, usable_cpus()));
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; t++) th.emplace_back([&, t] {
    const uint32_t lo = (uint32_t)((uint64_t)nc * t / T), hi = (uint32_t)((uint64_t)nc * (t + 1) / T);
    for (int m = 0; m < 3; m++) for (uint32_t r = lo; r < hi; r++) {
      Fe acc = Fe::zero();
      for (uint32_t k = Ms[m]->row_ptr[r]; k < Ms[m]->row_ptr[r + 1]; k++) acc = Fe::add(acc, Fe::mul(b.dict[Ms[m]->coef[k]], z[Ms[m]->col[k]]));
      out[m][r] = acc;
    }
  });
  for (auto& x : th) x.join();
}

}  // namespace

extern "C" {

void vimz_decider_free(vimz_decider* d) {
  if (!d) return;
  if (d->ctx) {
    std::unique_lock<std::mutex> g(d->ctx->mu);
    hipSetDevice(d->ctx->device);
    hipStreamSynchronize(d->ctx->stream);
    g.unlock();
    free_key(d->ctx, d->key);
  }
  delete d;
}

// Decider::preprocess (vimz/src/sonobe_backend/mod.rs:72-75): the decider circuit for this prover's shapes and a Groth16 key pair from a
// deterministic TEST setup (the trapdoor is derived from `seed`: anyone who knows the seed can forge — what a ceremony is for).
// seconds (optional) = {circuit synthesis, QAP evaluation at the trapdoor, key points on the GPU, total}.
int vimz_decider_setup(vimz_cf* v, const uint8_t* seed, size_t seed_len, vimz_decider** out, double seconds[4]) {
  if (!v || !out || (!seed && seed_len)) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = v->ctx;
  const double t_all = now_s();
  std::unique_ptr<vimz_decider> d(new vimz_decider());
  d->vk = v; d->ctx = ctx; d->seed.assign(seed, seed + seed_len);
  try {
    const cb::BuilderT<Fe>& main = v->circ->build->b;
    d->circ.finish(main, v->c1->len_z);
  } catch (const std::exception& e) { return vz_fail(ctx, VIMZ_ERR_INVALID, e.what()); }
  const double t_syn = now_s();
  G16Key& K = d->key;
  const cb::BuilderT<Fe>& b = d->circ.b;
  K.m = b.n_wires; K.n_pub = d->circ.n_public; K.n_c = b.n_constraints();
  K.n = 1; K.logn = 0;
  while (K.n < K.n_c + K.n_pub + 1) { K.n <<= 1; K.logn++; }
  if (K.logn > 26) return vz_fail(ctx, VIMZ_ERR_INVALID, "decider: circuit too large for the domain");
  // trapdoor
  const Fe tau = fr_from_hash(seed, seed_len, "vimz-decider-tau"), alpha = fr_from_hash(seed, seed_len, "vimz-decider-alpha"), beta = fr_from_hash(seed, seed_len, "vimz-decider-beta"),
           gamma = fr_from_hash(seed, seed_len, "vimz-decider-gamma"), delta = fr_from_hash(seed, seed_len, "vimz-decider-delta");
  const Fe gamma_inv = Fe::pow_pm2(gamma), delta_inv = Fe::pow_pm2(delta);
  K.omega = fr_root_of_unity(K.logn); K.omega_inv = Fe::pow_pm2(K.omega);
  K.n_inv = Fe::pow_pm2(cb::f_from_u64<Fe>(K.n));
  K.coset = cb::f_from_u64<Fe>(5); K.coset_inv = Fe::pow_pm2(K.coset);
  K.zinv = Fe::pow_pm2(Fe::sub(fr_pow_u64(K.coset, K.n), Fe::one()));
  { Fe chk = fr_pow_u64(K.omega, K.n / 2); if (K.n > 1 && !Fe::add(chk, Fe::one()).is_zero()) return vz_fail(ctx, VIMZ_ERR_INVALID, "decider: root of unity"); }
  // Lagrange basis at tau: L_j(tau) = Z(tau)/n · ω^j / (tau − ω^j)   (batch inversion)
  const Fe z_tau = Fe::sub(fr_pow_u64(tau, K.n), Fe::one());
  std::vector<Fe> L(K.n), den(K.n);
  { Fe w = Fe::one(); for (uint32_t j = 0; j < K.n; j++) { den[j] = Fe::sub(tau, w); L[j] = w; w = Fe::mul(w, K.omega); } }
  { std::vector<Fe> pre(K.n); Fe run = Fe::one();
    for (uint32_t j = 0; j < K.n; j++) { pre[j] = run; run = Fe::mul(run, den[j]); }
    Fe inv = Fe::pow_pm2(run);
    for (uint32_t j = K.n; j-- > 0;) { const Fe dj = den[j]; den[j] = Fe::mul(inv, pre[j]); inv = Fe::mul(inv, dj); } }
  { const Fe c = Fe::mul(z_tau, K.n_inv); for (uint32_t j = 0; j < K.n; j++) L[j] = Fe::mul(Fe::mul(L[j], den[j]), c); }
  // u_i = A_i(tau), v_i = B_i(tau), w_i = C_i(tau); the constant and the public inputs get a row of their own (a = z_i, b = c = 0)
  std::vector<Fe> uvw[3];
  for (int q = 0; q < 3; q++) uvw[q].assign(K.m, Fe::zero());
  { const cb::Csr* Ms[3] = {&b.A, &b.B, &b.C};
    std::vector<std::thread> th;
    for (int q = 0; q < 3; q++) th.emplace_back([&, q] {
      for (uint32_t r = 0; r < K.n_c; r++) for (uint32_t k = Ms[q]->row_ptr[r]; k < Ms[q]->row_ptr[r + 1]; k++) {
        Fe& dst = uvw[q][Ms[q]->col[k]]; dst = Fe::add(dst, Fe::mul(b.dict[Ms[q]->coef[k]], L[r]));
      }
    });
    for (auto& x : th) x.join();
    for (uint32_t i = 0; i <= K.n_pub; i++) uvw[0][i] = Fe::add(uvw[0][i], L[K.n_c + i]); }
  std::vector<Fe> lq(K.m - K.n_pub - 1), icq(K.n_pub + 1), hq(K.n - 1);
  for (uint32_t i = 0; i < K.m; i++) {
    const Fe k = Fe::add(Fe::add(Fe::mul(beta, uvw[0][i]), Fe::mul(alpha, uvw[1][i])), uvw[2][i]);
    if (i <= K.n_pub) icq[i] = Fe::mul(k, gamma_inv); else lq[i - K.n_pub - 1] = Fe::mul(k, delta_inv);
  }
  { Fe t = Fe::mul(z_tau, delta_inv); for (uint32_t j = 0; j + 1 < K.n; j++) { hq[j] = t; t = Fe::mul(t, tau); } }
  const double t_qap = now_s();
  // key points on the GPU
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const G1Aff g1 = g1_generator(); const G2PAff g2 = g2_generator();
  if (!g2_on_curve(g2)) return vz_fail(ctx, VIMZ_ERR_INVALID, "decider: G2 generator constant");
  const std::vector<G1Aff> T1 = fixed_table<Fq>(g1); const std::vector<G2PAff> T2 = fixed_table<Fq2>(g2);
  G1Aff* dT1 = nullptr; G2PAff* dT2 = nullptr;
  P_TRY(hipMalloc((void**)&dT1, sizeof(G1Aff) * T1.size())); P_TRY(hipMalloc((void**)&dT2, sizeof(G2PAff) * T2.size()));
  P_TRY(hipMemcpy(dT1, T1.data(), sizeof(G1Aff) * T1.size(), hipMemcpyHostToDevice)); P_TRY(hipMemcpy(dT2, T2.data(), sizeof(G2PAff) * T2.size(), hipMemcpyHostToDevice));
  struct Tables { G1Aff* a; G2PAff* b; ~Tables() { hipFree(a); hipFree(b); } } tables{dT1, dT2};
  auto g1_query = [&](const std::vector<Fe>& sc, vimz_bases** outb) -> int {
    G1Aff* pts = nullptr;
    hipError_t e = fixed_base_batch<Fq>(s, sc, dT1, &pts);
    if (e != hipSuccess) { if (pts) hipFree(pts); return vz_fail(ctx, VIMZ_ERR_HIP, "decider: key points", e); }
    const int rc = bases_from_device(ctx, pts, sc.size(), outb);
    hipFree(pts);
    return rc;
  };
  int rc;
  if ((rc = g1_query(uvw[0], &K.a_q)) || (rc = g1_query(uvw[1], &K.b1_q)) || (rc = g1_query(lq, &K.l_q)) || (rc = g1_query(hq, &K.h_q))) { free_key(ctx, K); return rc; }
  { hipError_t e = fixed_base_batch<Fq2>(s, uvw[1], dT2, &K.b2_q); if (e != hipSuccess) { free_key(ctx, K); return vz_fail(ctx, VIMZ_ERR_HIP, "decider: G2 key points", e); } }
  // domain tables
  P_TRY(hipMalloc((void**)&K.tw, 32 * (size_t)std::max<uint32_t>(K.n / 2, 1))); P_TRY(hipMalloc((void**)&K.tw_inv, 32 * (size_t)std::max<uint32_t>(K.n / 2, 1)));
  { Fr w, wi; memcpy(w.v, K.omega.v, 32); memcpy(wi.v, K.omega_inv.v, 32);
    hipLaunchKernelGGL(k_pow_table, dim3((K.n / 2 + 255) / 256), dim3(256), 0, s, K.tw, (size_t)K.n / 2, w);
    hipLaunchKernelGGL(k_pow_table, dim3((K.n / 2 + 255) / 256), dim3(256), 0, s, K.tw_inv, (size_t)K.n / 2, wi);
    P_TRY(hipGetLastError()); P_TRY(hipStreamSynchronize(s)); }
  // the few points of the verifying key, on the host
  K.alpha1 = to_affine(host_mul_fr<Fq>(g1, alpha)); K.beta1 = to_affine(host_mul_fr<Fq>(g1, beta)); K.delta1 = to_affine(host_mul_fr<Fq>(g1, delta));
  K.beta2 = to_affine(host_mul_fr<Fq2>(g2, beta)); K.gamma2 = to_affine(host_mul_fr<Fq2>(g2, gamma)); K.delta2 = to_affine(host_mul_fr<Fq2>(g2, delta));
  K.ic.resize(K.n_pub + 1);
  for (uint32_t i = 0; i <= K.n_pub; i++) K.ic[i] = to_affine(host_mul_fr<Fq>(g1, icq[i]));
  const double t_end = now_s();
  d->setup_s[0] = t_syn - t_all; d->setup_s[1] = t_qap - t_syn; d->setup_s[2] = t_end - t_qap; d->setup_s[3] = t_end - t_all;
  if (seconds) memcpy(seconds, d->setup_s, sizeof(d->setup_s));
  *out = d.release();
  return VIMZ_OK;
}

// info = {constraints, wires, public inputs, domain size, non-zeros of A, B, C, 0}
int vimz_decider_info(const vimz_decider* d, uint64_t info[8]) {
  if (!d || !info) return VIMZ_ERR_INVALID;
  const cb::BuilderT<Fe>& b = d->circ.b;
  info[0] = d->key.n_c; info[1] = d->key.m; info[2] = d->key.n_pub; info[3] = d->key.n; info[4] = b.A.col.size(); info[5] = b.B.col.size(); info[6] = b.C.col.size(); info[7] = 0;
  return VIMZ_OK;
}

// The verifying key as canonical little-endian words: alpha (G1: x, y), beta, gamma, delta (G2: x.c0, x.c1, y.c0, y.c1), the number of IC
// points, then the IC points (G1).  Returns the byte size (copies when buf is large enough).
int64_t vimz_decider_vk(const vimz_decider* d, void* buf, size_t cap) {
  if (!d) return VIMZ_ERR_INVALID;
  Writer w;
  const G16Key& K = d->key;
  w.point(K.alpha1);
  for (const G2PAff* p : {&K.beta2, &K.gamma2, &K.delta2}) { w.fe(p->x.c0); w.fe(p->x.c1); w.fe(p->y.c0); w.fe(p->y.c1); }
  w.word(K.ic.size());
  for (auto& p : K.ic) w.point(p);
  const size_t bytes = 8 * w.w.size();
  if (buf && cap >= bytes) memcpy(buf, w.w.data(), bytes);
  return (int64_t)bytes;
}

// Decider::prove (mod.rs:76-78) for the final fold a merged proof of ONE segment holds.  kzg = {c_W, c_E, e_W, e_E} (canonical): the challenges
// and evaluations of the two KZG openings the same calldata carries (vimz_cf_merged_kzg_open).  public_out: n_public canonical elements
// (i, z_0, z_i, h_inst); proof_out: A.x, A.y, B.x.c0, B.x.c1, B.y.c0, B.y.c1, C.x, C.y (canonical).
// seconds (optional) = {witness + sparse products (host), NTTs, MSMs, total}.
int vimz_decider_prove(vimz_decider* d, vimz_cf_merged* m, const uint64_t kzg[16], uint64_t* public_out, uint64_t proof_out[32], double seconds[4]) {
  if (!d || !m || !kzg || !public_out || !proof_out) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = d->ctx; vimz_cf* v = d->vk;
  const G16Key& K = d->key;
  const double t_all = now_s();
  CfDeciderView V;
  int rc = vz_cf_merged_decider_view(m, &V);
  if (rc) return rc;
  if (V.vk != v) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_prove: the merged proof belongs to another prover than the decider was set up for");
  const size_t nw = v->pri->n_wires, nc = v->pri->n_c;
  std::vector<Fe> Wf(nw), Ef(nc);
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    P_TRY(hipSetDevice(ctx->device));
    P_TRY(hipMemcpyAsync(Wf.data(), V.Zp, 32 * nw, hipMemcpyDeviceToHost, ctx->stream));
    P_TRY(hipMemcpyAsync(Ef.data(), V.Ep, 32 * nc, hipMemcpyDeviceToHost, ctx->stream));
    P_TRY(hipStreamSynchronize(ctx->stream));
  }
  aug::DeciderIn in;
  in.digest = v->c1->digest; in.i = V.n; in.z0 = V.zs; in.zi = V.ze;
  in.U = V.U; in.u = V.u; in.cfU = V.cfU; memcpy(in.r_low, V.r, 16);
  in.cmT = nn_point(V.cmT); in.Wn = nn_point(V.cW); in.En = nn_point(V.cE);
  Fe k4[4];
  for (int q = 0; q < 4; q++) { Fe c; memcpy(c.v, kzg + 4 * q, 32); if (!c.is_reduced()) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_prove: a KZG word is not below the modulus"); k4[q] = Fe::to_mont(c); }
  in.cW = k4[0]; in.cE = k4[1]; in.eW = k4[2]; in.eE = k4[3];
  in.Wf = Wf.data() + 1; in.Ef = Ef.data();
  std::vector<Fe> z;
  bool bad = false;
  try { z = d->circ.witness(v->circ->build->b, in, &bad); } catch (const std::exception& e) { return vz_fail(ctx, VIMZ_ERR_INVALID, e.what()); }
  if (bad) return vz_fail(ctx, VIMZ_ERR_UNSAT, "vimz_decider_prove: the proof does not satisfy the decider's statement (hashes, relaxed relation or KZG evaluations)");
  std::vector<Fe> abc[3];
  host_spmv3(d->circ.b, z, abc);
  for (int q = 0; q < 3; q++) abc[q].resize(K.n, Fe::zero());
  for (uint32_t i = 0; i <= K.n_pub; i++) abc[0][K.n_c + i] = z[i];
  const double t_wit = now_s();
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  // h = (a·b − c) / Z on the coset 5·H
  uint32_t* dv[3] = {nullptr, nullptr, nullptr}; uint32_t* dz = nullptr; G2P* dpart = nullptr;
  struct Free { uint32_t** v; uint32_t** z; G2P** p; ~Free() { for (int q = 0; q < 3; q++) if (v[q]) hipFree(v[q]); if (*z) hipFree(*z); if (*p) hipFree(*p); } } fr{dv, &dz, &dpart};
  for (int q = 0; q < 3; q++) { P_TRY(hipMalloc((void**)&dv[q], 32 * (size_t)K.n)); P_TRY(hipMemcpyAsync(dv[q], abc[q].data(), 32 * (size_t)K.n, hipMemcpyHostToDevice, s)); }
  P_TRY(hipMalloc((void**)&dz, 32 * (size_t)K.m)); P_TRY(hipMemcpyAsync(dz, z.data(), 32 * (size_t)K.m, hipMemcpyHostToDevice, s));
  auto asFr = [](const Fe& x) { Fr r; memcpy(r.v, x.v, 32); return r; };
  const unsigned gb = (K.n + 255) / 256;
  for (int q = 0; q < 3; q++) {
    P_TRY(ntt(s, dv[q], K, true));
    hipLaunchKernelGGL(k_scale_pow, dim3(gb), dim3(256), 0, s, dv[q], K.n, asFr(K.n_inv), asFr(K.coset));
    P_TRY(ntt(s, dv[q], K, false));
  }
  hipLaunchKernelGGL(k_quotient, dim3(gb), dim3(256), 0, s, dv[0], (const uint32_t*)dv[1], (const uint32_t*)dv[2], K.n, asFr(K.zinv));
  P_TRY(ntt(s, dv[0], K, true));
  hipLaunchKernelGGL(k_scale_pow, dim3(gb), dim3(256), 0, s, dv[0], K.n, asFr(K.n_inv), asFr(K.coset_inv));
  P_TRY(hipGetLastError()); P_TRY(hipStreamSynchronize(s));
  const double t_ntt = now_s();
  // the multi-scalar multiplications
  uint64_t pt[8];
  auto g1_msm = [&](const vimz_bases* q, const uint32_t* sc, size_t n, G1Aff* outp) -> int {
    if (!n) { outp->x = Fq::zero(); outp->y = Fq::zero(); return VIMZ_OK; }
    const int r2 = vz_msm_device(ctx, q, 0, sc, n, 1, 0, pt, VIMZ_FORM_MONTGOMERY);
    if (!r2) { memcpy(outp->x.v, pt, 32); memcpy(outp->y.v, pt + 4, 32); }
    return r2;
  };
  G1Aff sa, sb1, sl, sh;
  if ((rc = g1_msm(K.a_q, dz, K.m, &sa)) || (rc = g1_msm(K.b1_q, dz, K.m, &sb1)) || (rc = g1_msm(K.l_q, dz + 8 * (size_t)(K.n_pub + 1), K.m - K.n_pub - 1, &sl)) ||
      (rc = g1_msm(K.h_q, dv[0], K.n - 1, &sh))) return rc;
  const unsigned PT = 128 * 256;
  P_TRY(hipMalloc((void**)&dpart, sizeof(G2P) * PT));
  hipLaunchKernelGGL(k_msm_naive<Fq2>, dim3(PT / 128), dim3(128), 0, s, (const G2PAff*)K.b2_q, (const uint32_t*)dz, (size_t)K.m, dpart);
  P_TRY(hipGetLastError());
  std::vector<G2P> part(PT);
  P_TRY(hipMemcpyAsync(part.data(), dpart, sizeof(G2P) * PT, hipMemcpyDeviceToHost, s));
  P_TRY(hipStreamSynchronize(s));
  G2P sb2 = G2P::identity();
  for (auto& p : part) add_full(sb2, p);
  const double t_msm = now_s();
  // A = alpha + Σ z_i a_i + r·delta;  B = beta + Σ z_i b_i + s·delta;  C = Σ_priv z_i l_i + Σ h_j hq_j + s·A + r·B1 − r·s·delta
  Sha3 hs; hs.update(d->seed.data(), d->seed.size()); hs.update("vimz-decider-rs", 15); hs.update(z.data(), 32 * (size_t)(K.n_pub + 1));
  uint8_t dg[32]; hs.finish(dg);
  const Fe r = fr_from_hash(dg, 32, "r"), sr = fr_from_hash(dg, 32, "s");
  G1 A = from_affine(K.alpha1); add_mixed(A, sa); { G1 t = host_mul_fr<Fq>(K.delta1, r); add_full(A, t); }
  G1 B1 = from_affine(K.beta1); add_mixed(B1, sb1); { G1 t = host_mul_fr<Fq>(K.delta1, sr); add_full(B1, t); }
  G2P B2 = from_affine(K.beta2); add_full(B2, sb2); { G2P t = host_mul_fr<Fq2>(K.delta2, sr); add_full(B2, t); }
  const G1Aff Aa = to_affine(A), B1a = to_affine(B1);
  G1 C = from_affine(sl); add_mixed(C, sh);
  { G1 t = host_mul_fr<Fq>(Aa, sr); add_full(C, t); }
  { G1 t = host_mul_fr<Fq>(B1a, r); add_full(C, t); }
  { G1 t = host_mul_fr<Fq>(K.delta1, Fe::mul(r, sr)); if (!t.is_identity()) t.Y = Fq::neg(t.Y); add_full(C, t); }
  const G1Aff Ca = to_affine(C); const G2PAff Ba = to_affine(B2);
  auto put = [&](uint64_t* dst, const Fq& mont) { const Fq c = Fq::from_mont(mont); memcpy(dst, c.v, 32); };
  put(proof_out, Aa.x); put(proof_out + 4, Aa.y);
  put(proof_out + 8, Ba.x.c0); put(proof_out + 12, Ba.x.c1); put(proof_out + 16, Ba.y.c0); put(proof_out + 20, Ba.y.c1);
  put(proof_out + 24, Ca.x); put(proof_out + 28, Ca.y);
  for (uint32_t i = 0; i < K.n_pub; i++) { const Fe c = Fe::from_mont(z[1 + i]); memcpy(public_out + 4 * i, c.v, 32); }
  if (seconds) { seconds[0] = t_wit - t_all; seconds[1] = t_ntt - t_wit; seconds[2] = t_msm - t_ntt; seconds[3] = now_s() - t_all; }
  return VIMZ_OK;
}

}  // extern "C"
