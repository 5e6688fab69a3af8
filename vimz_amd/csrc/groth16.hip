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
#include <sys/random.h>
#include <functional>
#include <thread>
#include "cyclefold_internal.hpp"
#include "decider_view.hpp"
#include "aug/decider.hpp"
#include "pairing.hpp"
#include "vecops_api.hpp"

namespace {

// Secrets are wiped when they go out of scope: the trapdoor, everything derived from it that still determines it (powers of tau, the
// Lagrange basis at tau, the query scalars), a proof's blinders.
void wipe(void* p, size_t n) { explicit_bzero(p, n); }
template <class T> void wipe(std::vector<T>& v) { if (!v.empty()) explicit_bzero(v.data(), v.size() * sizeof(T)); }

// ---- Fq2 = Fq[u] / (u² + 1): coordinates of G2 (the twist y² = x³ + 3/(9 + u)) — pairing.hpp ------------------------------------------------
using vz::pairing::Fq2;
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
using vz::pairing::g2_on_curve;

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
// The G2 multi-scalar multiplication of a proof, Σ z_i·[b_i]_2 over the wires whose query point is not the identity, BIT PLANE BY BIT PLANE:
//     Σ_i s_i·P_i = Σ_j 2^j·S_j,   S_j = Σ_{i: bit j of s_i} P_i
// — no doublings on the device (a per-point double-and-add spends two thirds of its additions on them and ran 2.0 s for the full decider's 0.7 M
// full-size scalars: a quarter of the threads held all the work), no sorting: plane j is a filtered sum.  A wave walks 64 consecutive scalars at a time,
// queues the indices whose bit j is set in LDS (ballot + prefix count: the compaction of msm.hpp's k_ones_dense) and adds 64 queued bases at a time, one per
// lane.  blockIdx.y = the plane; partial[plane][thread]; k_g2_plane_tree folds a plane's partials; the host adds the 254 plane sums by Horner.
constexpr uint32_t G2_PLANE_THREADS = 2048, G2_PLANES = 254;
__global__ void __launch_bounds__(256) k_g2_canon(const uint32_t* __restrict__ scalars_mont, const uint32_t* __restrict__ idx, uint32_t n, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  store_fe(out, i, Fr::from_mont(load_fe<Fr>(scalars_mont, idx[i])));
}
__global__ void __launch_bounds__(256) k_g2_planes(const Affine<Fq2>* __restrict__ bases, const uint32_t* __restrict__ canon, uint32_t n, XYZZ<Fq2>* __restrict__ partial) {
  __shared__ uint32_t queue[4][128];
  const uint32_t plane = blockIdx.y, word = plane >> 5, bit = plane & 31u;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t wave = t >> 6, nwaves = G2_PLANE_THREADS / 64;
  volatile uint32_t* q = queue[wv];
  XYZZ<Fq2> acc = XYZZ<Fq2>::identity();
  uint32_t cnt = 0;                                     // (wave-uniform)
  for (uint32_t base = wave * 64; base < n; base += nwaves * 64) {
    const uint32_t i = base + lane;
    const bool on = i < n && ((canon[8 * (size_t)i + word] >> bit) & 1u);
    const uint64_t m = __ballot(on);
    if (on) q[cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
    cnt += (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
    if (cnt >= 64) {
      cnt -= 64;
      const uint32_t ix = q[cnt + lane];
      __builtin_amdgcn_wave_barrier();
      const Affine<Fq2> pt = bases[ix];
      add_mixed(acc, pt);
    }
  }
  if (lane < cnt) { const Affine<Fq2> pt = bases[q[lane]]; add_mixed(acc, pt); }
  partial[(size_t)plane * G2_PLANE_THREADS + t] = acc;
}
__global__ void __launch_bounds__(128) k_g2_plane_tree(const XYZZ<Fq2>* __restrict__ partial, XYZZ<Fq2>* __restrict__ out) {
  __shared__ XYZZ<Fq2> sh[128];
  const uint32_t plane = blockIdx.x, t = threadIdx.x;
  XYZZ<Fq2> acc = XYZZ<Fq2>::identity();
  for (uint32_t k = t; k < G2_PLANE_THREADS; k += 128) add_full(acc, partial[(size_t)plane * G2_PLANE_THREADS + k]);
  sh[t] = acc;
  __syncthreads();
  for (uint32_t h = 64; h >= 1; h >>= 1) {
    if (t < h) { XYZZ<Fq2> a = sh[t]; add_full(a, sh[t + h]); sh[t] = a; }
    __syncthreads();
  }
  if (t == 0) out[plane] = sh[0];
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
  G2PAff* b2_q = nullptr; uint32_t* b2_idx = nullptr; uint32_t n_b2 = 0;      // G2 query: the n_b2 wires whose point is not the identity (device: points, wire numbers)
  G1Aff alpha1, beta1, delta1; G2PAff beta2, gamma2, delta2;
  std::vector<G1Aff> ic;                                        // n_pub + 1 points
};
void free_key_raw(G16Key& K) {      // (not vimz_bases_free: it takes the context's lock, which callers on an error path hold)
  for (vimz_bases** b : {&K.a_q, &K.b1_q, &K.l_q, &K.h_q}) if (*b) {
    if ((*b)->d) hipFree((*b)->d);
    if ((*b)->tables) hipFree((*b)->tables);
    for (auto& t : (*b)->small) { if (t.rows) hipFree(t.rows); if (t.mult) hipFree(t.mult); }
    delete *b; *b = nullptr;
  }
  if (K.b2_q) hipFree(K.b2_q); K.b2_q = nullptr;
  if (K.b2_idx) hipFree(K.b2_idx); K.b2_idx = nullptr;
  if (K.tw) hipFree(K.tw); if (K.tw_inv) hipFree(K.tw_inv); K.tw = K.tw_inv = nullptr;
}
}  // namespace

struct vimz_decider {
  vimz_cf* vk = nullptr; vimz_ctx* ctx = nullptr;
  aug::DeciderCircuit circ;
  G16Key key;
  G2PAff kzg_vk;                         // [tau]G2 of the SRS the prover's ck_main is made of (identity: not given — KZG checks are refused)
  double setup_s[4] = {0, 0, 0, 0};      // circuit synthesis, QAP evaluation at the trapdoor (host), key points (GPU), total
  ~vimz_decider() { if (ctx) { hipSetDevice(ctx->device); free_key_raw(key); } }      // (error paths of the setup end here too)
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
  wipe(canon);
  return e;
}
// device std-form G1 points -> a commitment key in the MSM's resident form
int bases_from_device(vimz_ctx* ctx, const G1Aff* d_pts, size_t n, vimz_bases** out, int canonical = 0) {
  vimz_bases* b = new vimz_bases(); b->curve = VIMZ_CURVE_BN254_G1; b->n = n; b->d = nullptr;
  if (n) {
    hipError_t e = hipMalloc(&b->d, 4 * (size_t)AFFINE_WORDS * n);
    if (e != hipSuccess) { delete b; return vz_fail(ctx, VIMZ_ERR_HIP, "decider: hipMalloc(query)", e); }
    launch_points_to_internal<Fq>(ctx->stream, (const uint32_t*)d_pts, canonical, b->d, n);
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


// (A,B,C)·z of a builder's CSR on the host threads
template <class FF>
void host_spmv3(const cb::BuilderT<FF>& b, const std::vector<FF>& z, std::vector<FF>* out /* [3] of n_c */) {
  typedef FF Fe;
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


namespace {

// ---- randomness: the toxic waste of a locally run setup and a proof's blinding scalars come from the OS -------------------------------------------
bool os_random(void* buf, size_t n) {
  uint8_t* p = (uint8_t*)buf;
  while (n) { const ssize_t k = getrandom(p, n, 0); if (k <= 0) return false; p += k; n -= (size_t)k; }
  return true;
}
struct Trapdoor { Fe tau, alpha, beta, gamma, delta; ~Trapdoor() { wipe(this, sizeof(*this)); } };
// a uniform non-zero element of Fr by rejection (254-bit draws below the modulus: accepted with probability 0.76)
bool fr_random(Fe* out) {
  for (;;) {
    uint8_t raw[32];
    if (!os_random(raw, sizeof(raw))) return false;
    raw[31] &= 0x3f;
    Fe c; memcpy(c.v, raw, 32);
    wipe(raw, sizeof(raw));
    if (!c.is_reduced() || c.is_zero()) continue;
    *out = Fe::to_mont(c);
    wipe(&c, sizeof(c));
    return true;
  }
}
bool trapdoor_random(Trapdoor& t) {
  Fe* dst[5] = {&t.tau, &t.alpha, &t.beta, &t.gamma, &t.delta};
  for (int k = 0; k < 5; k++) if (!fr_random(dst[k])) return false;
  return true;
}
Trapdoor trapdoor_seeded(const uint8_t* seed, size_t n) {
  Trapdoor t;
  t.tau = fr_from_hash(seed, n, "vimz-decider-tau"); t.alpha = fr_from_hash(seed, n, "vimz-decider-alpha"); t.beta = fr_from_hash(seed, n, "vimz-decider-beta");
  t.gamma = fr_from_hash(seed, n, "vimz-decider-gamma"); t.delta = fr_from_hash(seed, n, "vimz-decider-delta");
  return t;
}
void put_fq(uint64_t* dst, const Fq& mont) { const Fq c = Fq::from_mont(mont); memcpy(dst, c.v, 32); }
bool get_fq(const uint64_t* src, Fq* out) { Fq c; memcpy(c.v, src, 32); if (!c.is_reduced()) return false; *out = Fq::to_mont(c); return true; }
void put_g2(uint64_t* dst, const G2PAff& p) { put_fq(dst, p.x.c0); put_fq(dst + 4, p.x.c1); put_fq(dst + 8, p.y.c0); put_fq(dst + 12, p.y.c1); }
bool get_g2(const uint64_t* src, G2PAff* p) { return get_fq(src, &p->x.c0) && get_fq(src + 4, &p->x.c1) && get_fq(src + 8, &p->y.c0) && get_fq(src + 12, &p->y.c1); }
bool get_g1(const uint64_t* src, G1Aff* p) { return get_fq(src, &p->x) && get_fq(src + 4, &p->y); }
U256w u256_of(const uint64_t* w) { U256w r; memcpy(r.w, w, 32); return r; }

// KZG::setup: srs[i] = [tau^i]G1 for i < n (resident MSM form), vk = [tau]G2
int kzg_setup_impl(vimz_ctx* ctx, const Fe& tau, size_t n, vimz_bases** srs_out, uint64_t vk_g2_out[16]) {
  std::vector<Fe> pw(n);
  struct Wipe { std::vector<Fe>& v; ~Wipe() { wipe(v); } } wp{pw};
  { Fe t = Fe::one(); for (size_t i = 0; i < n; i++) { pw[i] = t; t = Fe::mul(t, tau); } }
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  const std::vector<G1Aff> T1 = fixed_table<Fq>(g1_generator());
  G1Aff* dT1 = nullptr;
  P_TRY(hipMalloc((void**)&dT1, sizeof(G1Aff) * T1.size()));
  struct Tab { G1Aff* a; ~Tab() { hipFree(a); } } tab{dT1};
  P_TRY(hipMemcpy(dT1, T1.data(), sizeof(G1Aff) * T1.size(), hipMemcpyHostToDevice));
  G1Aff* pts = nullptr;
  hipError_t e = fixed_base_batch<Fq>(ctx->stream, pw, dT1, &pts);
  if (e != hipSuccess) { if (pts) hipFree(pts); return vz_fail(ctx, VIMZ_ERR_HIP, "vimz_kzg_setup: powers of tau", e); }
  const int rc = bases_from_device(ctx, pts, n, srs_out);
  hipFree(pts);
  if (rc) return rc;
  put_g2(vk_g2_out, to_affine(host_mul_fr<Fq2>(g2_generator(), tau)));
  return VIMZ_OK;
}

// the decider circuit for this prover's shapes: light (checks 1-4) or full (also the CycleFold instance's commitments and relation: the first
// generators of the prover's CycleFold commitment key become constants of the circuit)
int decider_circuit_build(vimz_cf* v, bool light, aug::DeciderCircuit& circ) {
  vimz_ctx* ctx = v->ctx;
  const cb::BuilderT<Fe>& main = v->circ->build->b;
  std::vector<G2Aff> gens;
  if (!light) {
    const cb::BuilderT<Fq>& cfb = v->cf.b;
    const size_t need = std::max<size_t>(cfb.n_wires - 1 - aug::CF_IO, cfb.n_constraints());
    if (!v->ck2 || v->ck2->n < need) return vz_fail(ctx, VIMZ_ERR_INVALID, "decider: the CycleFold commitment key is shorter than the vectors it commits to");
    gens.resize(need);
    // (a stream of its own and NOT the context's lock: the key is immutable, and the prover whose context this is may be in the middle of a fold that holds the
    //  lock for seconds — the set-up's host part is meant to run under that fold: tools/e2e.py)
    P_TRY(hipSetDevice(ctx->device));
    hipStream_t ts = nullptr;
    P_TRY(hipStreamCreateWithFlags(&ts, hipStreamNonBlocking));
    uint32_t* tmp = nullptr;
    struct FreeTmp { uint32_t** q; hipStream_t* s; ~FreeTmp() { if (*q) hipFree(*q); hipStreamDestroy(*s); } } ft{&tmp, &ts};
    P_TRY(hipMalloc((void**)&tmp, 64 * need));
    launch_points_from_internal<Fe>(ts, v->ck2->d, 0, tmp, need);
    P_TRY(hipGetLastError());
    P_TRY(hipMemcpyAsync(gens.data(), tmp, 64 * need, hipMemcpyDeviceToHost, ts));
    P_TRY(hipStreamSynchronize(ts));
  }
  try {
    if (light) circ.finish(main, v->c1->len_z);
    else circ.finish(main, v->c1->len_z, &v->cf.b, gens.data(), (uint32_t)gens.size());
  } catch (const std::exception& e) { return vz_fail(ctx, VIMZ_ERR_INVALID, e.what()); }
  return VIMZ_OK;
}
// [tau]G2 given with a prover whose main key is not the SRS it belongs to makes every later KZG check fail far from its cause:
// e(srs[1], G2) == e(G1, [tau]G2) once, at set-up / load.  (srs[0] = G1 for an SRS made by vimz_kzg_setup; any other key fails here.)
int kzg_vk_matches_srs(vimz_cf* v, const G2PAff& vk, const char* who) {
  vimz_ctx* ctx = v->ctx;
  G1Aff p01[2];
  if (!v->ck1 || v->ck1->n < 2) return vz_fail(ctx, VIMZ_ERR_INVALID, "decider: the main commitment key is too short to be an SRS");
  {      // (a stream of its own, not the context's lock: see decider_circuit_build)
    P_TRY(hipSetDevice(ctx->device));
    hipStream_t ts = nullptr;
    P_TRY(hipStreamCreateWithFlags(&ts, hipStreamNonBlocking));
    uint32_t* tmp = nullptr;
    struct FreeTmp { uint32_t** q; hipStream_t* s; ~FreeTmp() { if (*q) hipFree(*q); hipStreamDestroy(*s); } } ft{&tmp, &ts};
    P_TRY(hipMalloc((void**)&tmp, 128));
    launch_points_from_internal<Fq>(ts, v->ck1->d, 0, tmp, 2);
    P_TRY(hipGetLastError());
    P_TRY(hipMemcpyAsync(p01, tmp, 128, hipMemcpyDeviceToHost, ts));
    P_TRY(hipStreamSynchronize(ts));
  }
  const G1Aff g1 = g1_generator();
  G1Aff ng = g1; ng.y = Fq::neg(g1.y);
  if (!vz::pairing::consts().ok) return vz_fail(ctx, VIMZ_ERR_INVALID, "decider: pairing constants");
  if (!p01[0].x.eq(g1.x) || !p01[0].y.eq(g1.y) || !vz::pairing::g1_on_curve(p01[1]) || !vz::pairing::product_is_one({{p01[1], g2_generator()}, {ng, vk}})) {
    std::string m = std::string(who) + ": the KZG verifying key is not [tau]G2 of the SRS the prover commits with";
    return vz_fail(ctx, VIMZ_ERR_INVALID, m.c_str());
  }
  return VIMZ_OK;
}

int decider_setup_impl(vimz_cf* v, const uint64_t kzg_vk_g2[16], const Trapdoor& td, bool light, vimz_decider** out, double seconds[4]) {
  vimz_ctx* ctx = v->ctx;
  const double t_all = now_s();
  std::unique_ptr<vimz_decider> d(new vimz_decider());
  d->vk = v; d->ctx = ctx;
  d->kzg_vk.x = d->kzg_vk.y = Fq2::zero();
  if (kzg_vk_g2) {
    if (!get_g2(kzg_vk_g2, &d->kzg_vk) || !vz::pairing::g2_on_curve(d->kzg_vk) || !vz::pairing::g2_in_subgroup(d->kzg_vk))
      return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_setup: the KZG verifying key is not a point of G2");
    const int rcv = kzg_vk_matches_srs(v, d->kzg_vk, "vimz_decider_setup");
    if (rcv) return rcv;
  }
  { const int rcb = decider_circuit_build(v, light, d->circ); if (rcb) return rcb; }
  const double t_syn = now_s();
  G16Key& K = d->key;
  const cb::BuilderT<Fe>& b = d->circ.b;
  K.m = b.n_wires; K.n_pub = d->circ.n_public; K.n_c = b.n_constraints();
  K.n = 1; K.logn = 0;
  while (K.n < K.n_c + K.n_pub + 1) { K.n <<= 1; K.logn++; }
  if (K.logn > 26) return vz_fail(ctx, VIMZ_ERR_INVALID, "decider: circuit too large for the domain");
  const Fe tau = td.tau, alpha = td.alpha, beta = td.beta, gamma = td.gamma, delta = td.delta;
  const Fe gamma_inv = Fe::pow_pm2(gamma), delta_inv = Fe::pow_pm2(delta);
  K.omega = fr_root_of_unity(K.logn); K.omega_inv = Fe::pow_pm2(K.omega);
  K.n_inv = Fe::pow_pm2(cb::f_from_u64<Fe>(K.n));
  K.coset = cb::f_from_u64<Fe>(5); K.coset_inv = Fe::pow_pm2(K.coset);
  K.zinv = Fe::pow_pm2(Fe::sub(fr_pow_u64(K.coset, K.n), Fe::one()));
  { Fe chk = fr_pow_u64(K.omega, K.n / 2); if (K.n > 1 && !Fe::add(chk, Fe::one()).is_zero()) return vz_fail(ctx, VIMZ_ERR_INVALID, "decider: root of unity"); }
  // Lagrange basis at tau: L_j(tau) = Z(tau)/n · ω^j / (tau − ω^j)   (batch inversion)
  const Fe z_tau = Fe::sub(fr_pow_u64(tau, K.n), Fe::one());
  if (z_tau.is_zero()) return vz_fail(ctx, VIMZ_ERR_INVALID, "decider: tau lies in the domain");
  std::vector<Fe> L(K.n), den(K.n);
  std::vector<Fe> uvw[3];
  std::vector<Fe> lq, icq, hq;
  struct WipeAll { std::vector<Fe>*v[8]; ~WipeAll() { for (auto* x : v) wipe(*x); } } wa{{&L, &den, &uvw[0], &uvw[1], &uvw[2], &lq, &icq, &hq}};
  // (all of this on the host's threads, in chunks: 4 M elements at the full decider's size — 0.8 s on one thread per phase)
  const unsigned TH = std::max(1u, std::min(16u, usable_cpus()));
  auto parallel = [&](uint64_t n, const std::function<void(uint64_t, uint64_t)>& f) {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < TH; t++) { const uint64_t lo = n * t / TH, hi = n * (t + 1) / TH; if (lo < hi) th.emplace_back([&f, lo, hi] { f(lo, hi); }); }
    for (auto& x : th) x.join();
  };
  { const Fe c = Fe::mul(z_tau, K.n_inv);
    parallel(K.n, [&](uint64_t lo, uint64_t hi) {      // ω^j, tau − ω^j, one batched inversion per chunk
      Fe w = fr_pow_u64(K.omega, lo);
      for (uint64_t j = lo; j < hi; j++) { den[j] = Fe::sub(tau, w); L[j] = w; w = Fe::mul(w, K.omega); }
      std::vector<Fe> pre(hi - lo); Fe run = Fe::one();
      for (uint64_t j = lo; j < hi; j++) { pre[j - lo] = run; run = Fe::mul(run, den[j]); }
      Fe inv = Fe::pow_pm2(run);
      for (uint64_t j = hi; j-- > lo;) { const Fe dj = den[j]; den[j] = Fe::mul(inv, pre[j - lo]); inv = Fe::mul(inv, dj); }
      for (uint64_t j = lo; j < hi; j++) L[j] = Fe::mul(Fe::mul(L[j], den[j]), c);
      wipe(pre);
    }); }
  // u_i = A_i(tau), v_i = B_i(tau), w_i = C_i(tau); the constant and the public inputs get a row of their own (a = z_i, b = c = 0).
  // Every thread walks all the non-zeros and accumulates the wires of ITS range: no shared accumulator, no per-thread copy of three 130 MB vectors
  for (int q = 0; q < 3; q++) uvw[q].assign(K.m, Fe::zero());
  { const cb::Csr* Ms[3] = {&b.A, &b.B, &b.C};
    parallel(K.m, [&](uint64_t wlo, uint64_t whi) {
      for (int q = 0; q < 3; q++) {
        const cb::Csr& M = *Ms[q];
        for (uint32_t r = 0; r < K.n_c; r++) for (uint32_t k = M.row_ptr[r]; k < M.row_ptr[r + 1]; k++) {
          const uint32_t col = M.col[k];
          if (col < wlo || col >= whi) continue;
          Fe& dst = uvw[q][col]; dst = Fe::add(dst, Fe::mul(b.dict[M.coef[k]], L[r]));
        }
      }
    });
    for (uint32_t i = 0; i <= K.n_pub; i++) uvw[0][i] = Fe::add(uvw[0][i], L[K.n_c + i]); }
  lq.assign(K.m - K.n_pub - 1, Fe::zero()); icq.assign(K.n_pub + 1, Fe::zero()); hq.assign(K.n - 1, Fe::zero());
  parallel(K.m, [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; i++) {
      const Fe k = Fe::add(Fe::add(Fe::mul(beta, uvw[0][i]), Fe::mul(alpha, uvw[1][i])), uvw[2][i]);
      if (i <= K.n_pub) icq[i] = Fe::mul(k, gamma_inv); else lq[i - K.n_pub - 1] = Fe::mul(k, delta_inv);
    }
  });
  { const Fe t0 = Fe::mul(z_tau, delta_inv);
    parallel(K.n - 1, [&](uint64_t lo, uint64_t hi) { Fe t = Fe::mul(t0, fr_pow_u64(tau, lo)); for (uint64_t j = lo; j < hi; j++) { hq[j] = t; t = Fe::mul(t, tau); } }); }
  const double t_qap = now_s();
  // key points on the GPU (any failure below: ~vimz_decider releases what was built)
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const G1Aff g1 = g1_generator(); const G2PAff g2 = g2_generator();
  if (!g2_on_curve(g2)) return vz_fail(ctx, VIMZ_ERR_INVALID, "decider: G2 generator constant");
  const std::vector<G1Aff> T1 = fixed_table<Fq>(g1); const std::vector<G2PAff> T2 = fixed_table<Fq2>(g2);
  struct Tables { G1Aff* a = nullptr; G2PAff* b = nullptr; ~Tables() { if (a) hipFree(a); if (b) hipFree(b); } } tables;
  P_TRY(hipMalloc((void**)&tables.a, sizeof(G1Aff) * T1.size())); P_TRY(hipMalloc((void**)&tables.b, sizeof(G2PAff) * T2.size()));
  P_TRY(hipMemcpy(tables.a, T1.data(), sizeof(G1Aff) * T1.size(), hipMemcpyHostToDevice)); P_TRY(hipMemcpy(tables.b, T2.data(), sizeof(G2PAff) * T2.size(), hipMemcpyHostToDevice));
  auto g1_query = [&](const std::vector<Fe>& sc, vimz_bases** outb) -> int {
    G1Aff* pts = nullptr;
    hipError_t e = fixed_base_batch<Fq>(s, sc, tables.a, &pts);
    if (e != hipSuccess) { if (pts) hipFree(pts); return vz_fail(ctx, VIMZ_ERR_HIP, "decider: key points", e); }
    const int rc = bases_from_device(ctx, pts, sc.size(), outb);
    hipFree(pts);
    return rc;
  };
  int rc;
  if ((rc = g1_query(uvw[0], &K.a_q)) || (rc = g1_query(uvw[1], &K.b1_q)) || (rc = g1_query(lq, &K.l_q)) || (rc = g1_query(hq, &K.h_q))) return rc;
  { // the G2 query holds only the wires that occur in B (the others' points are the identity)
    std::vector<uint32_t> ix; std::vector<Fe> sc;
    for (uint32_t i = 0; i < K.m; i++) if (!uvw[1][i].is_zero()) { ix.push_back(i); sc.push_back(uvw[1][i]); }
    K.n_b2 = (uint32_t)ix.size();
    hipError_t e = fixed_base_batch<Fq2>(s, sc, tables.b, &K.b2_q);
    wipe(sc);
    if (e != hipSuccess) return vz_fail(ctx, VIMZ_ERR_HIP, "decider: G2 key points", e);
    P_TRY(hipMalloc((void**)&K.b2_idx, 4 * std::max<size_t>(ix.size(), 1)));
    P_TRY(hipMemcpy(K.b2_idx, ix.data(), 4 * ix.size(), hipMemcpyHostToDevice)); }
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

// ---- the verifier: what contracts/*Verifier.sol::verifyNovaProof does with the 25 words (ContrastVerifier.sol:685-783) ---------------------------------
struct VerifierKey {
  Fe pp_hash; uint32_t len_z = 0;
  G1Aff alpha; G2PAff beta, gamma, delta; std::vector<G1Aff> ic;
  G1Aff kzg_g1; G2PAff kzg_g2, kzg_vk;
};
enum { DV_STEPS = 1, DV_KZG_W = 2, DV_KZG_E = 4, DV_GROTH16 = 8, DV_MALFORMED = 16 };
G1Aff g1_lin(const G1Aff& a, const uint64_t* k256, const G1Aff& b) {      // a + k·b
  G1 acc = aff_is_identity(a) ? G1::identity() : from_affine(a);
  if (!aff_is_identity(b)) { G1 t = host_mul<Fq>(b, (const uint32_t*)k256, 256); add_full(acc, t); }
  return to_affine(acc);
}
G1Aff g1_negate(const G1Aff& p) { G1Aff r = p; if (!aff_is_identity(p)) r.y = Fq::neg(p.y); return r; }
// KZG10Verifier.check (:167-189): e(pi, VK) · e(x·(−pi) − c + y·G_1, G_2) == 1
bool kzg_check(const VerifierKey& K, const G1Aff& c, const G1Aff& pi, const uint64_t* x, const uint64_t* y) {
  G1 acc = host_mul<Fq>(g1_negate(pi), (const uint32_t*)x, 256);
  { const G1Aff nc = g1_negate(c); if (!aff_is_identity(nc)) add_mixed(acc, nc); }
  { G1 t = host_mul<Fq>(K.kzg_g1, (const uint32_t*)y, 256); add_full(acc, t); }
  return vz::pairing::product_is_one({{pi, K.kzg_vk}, {to_affine(acc), K.kzg_g2}});
}
// words: the 25 calldata words as 4 little-endian limbs each.  Returns a bit set of DV_* (0 = accepted)
uint32_t verify_words(const VerifierKey& K, uint64_t steps, const uint64_t* z0, const uint64_t* zi, const uint64_t* w) {
  using namespace vz::pairing;
  if (!consts().ok) return DV_MALFORMED;
  uint32_t res = 0;
  if (steps < 2) res |= DV_STEPS;                                                    // :697
  G1Aff P[8]; static const int at[8] = {0, 2, 4, 6, 9, 15, 21, 23};              // U_i.cmW, U_i.cmE, u_i.cmW, cmT, A, C, proof_W, proof_E
  for (int k = 0; k < 8; k++) if (!get_g1(w + 4 * at[k], &P[k]) || !g1_on_curve(P[k])) return res | DV_MALFORMED;
  G2PAff B;      // calldata: x imaginary, x real, y imaginary, y real
  if (!get_fq(w + 4 * 11, &B.x.c1) || !get_fq(w + 4 * 12, &B.x.c0) || !get_fq(w + 4 * 13, &B.y.c1) || !get_fq(w + 4 * 14, &B.y.c0) || !g2_on_curve(B) || !g2_in_subgroup(B)) return res | DV_MALFORMED;
  Fe pub_el; std::vector<Fe> pub;
  auto push_canon = [&](const uint64_t* c) { Fe x; memcpy(x.v, c, 32); if (!x.is_reduced()) return false; pub.push_back(Fe::to_mont(x)); return true; };      // checkField
  pub.push_back(K.pp_hash); pub.push_back(cb::f_from_u64<Fe>(steps));
  for (uint32_t k = 0; k < K.len_z; k++) if (!push_canon(z0 + 4 * k)) return res | DV_GROTH16;
  for (uint32_t k = 0; k < K.len_z; k++) if (!push_canon(zi + 4 * k)) return res | DV_GROTH16;
  const uint64_t* r = w + 4 * 8;
  const G1Aff cmW = g1_lin(P[0], r, P[2]), cmE = g1_lin(P[1], r, P[3]);              // :712-713, :735-736
  auto push_limbs = [&](const Fq& coord) { const Fq c = Fq::from_mont(coord); U256w u; memcpy(u.w, c.v, 32); uint64_t l[aug::DEC_LIMBS]; aug::decider_limbs55(u, l); for (int k = 0; k < aug::DEC_LIMBS; k++) pub.push_back(cb::f_from_u64<Fe>(l[k])); };
  push_limbs(cmW.x); push_limbs(cmW.y); push_limbs(cmE.x); push_limbs(cmE.y);
  for (int k = 17; k <= 20; k++) if (!push_canon(w + 4 * k)) res |= DV_GROTH16;
  if (res & DV_GROTH16) return res;
  push_limbs(P[3].x); push_limbs(P[3].y);
  if (pub.size() + 1 != K.ic.size()) return res | DV_MALFORMED;
  if (!kzg_check(K, cmW, P[6], w + 4 * 17, w + 4 * 19)) res |= DV_KZG_W;             // :725-730
  if (!kzg_check(K, cmE, P[7], w + 4 * 18, w + 4 * 20)) res |= DV_KZG_E;             // :748-753
  // Groth16 (:386-620): vk_x = IC_0 + Σ pub_k·IC_{k+1};  e(−A, B)·e(alpha, beta)·e(vk_x, gamma)·e(C, delta) == 1
  G1 vkx = from_affine(K.ic[0]);
  if (aff_is_identity(K.ic[0])) vkx = G1::identity();
  for (size_t k = 0; k < pub.size(); k++) { if (pub[k].is_zero() || aff_is_identity(K.ic[k + 1])) continue; G1 t = host_mul_fr<Fq>(K.ic[k + 1], pub[k]); add_full(vkx, t); }
  if (!product_is_one({{g1_negate(P[4]), B}, {K.alpha, K.beta}, {to_affine(vkx), K.gamma}, {P[5], K.delta}})) res |= DV_GROTH16;
  return res;
}
// key blob (vimz_decider_vk): pp_hash, len_z, alpha (G1), beta, gamma, delta (G2: x.c0, x.c1, y.c0, y.c1), n_ic, IC points, KZG G_1 (G1), G_2, VK (G2)
bool parse_key(const uint64_t* w, size_t n, VerifierKey* K) {
  using namespace vz::pairing;
  size_t pos = 0;
  auto need = [&](size_t k) { return pos + k <= n; };
  if (!need(5)) return false;
  { Fe c; memcpy(c.v, w, 32); if (!c.is_reduced()) return false; K->pp_hash = Fe::to_mont(c); } pos += 4;
  K->len_z = (uint32_t)w[pos++];
  if (!need(8 + 48 + 1) || !get_g1(w + pos, &K->alpha)) return false; pos += 8;
  for (G2PAff* p : {&K->beta, &K->gamma, &K->delta}) { if (!get_g2(w + pos, p)) return false; pos += 16; }
  const uint64_t n_ic = w[pos++];
  if (n_ic != 1 + (uint64_t)aug::decider_n_public(K->len_z) || !need(8 * n_ic + 8 + 32)) return false;
  K->ic.resize(n_ic);
  for (auto& p : K->ic) { if (!get_g1(w + pos, &p)) return false; pos += 8; }
  if (!get_g1(w + pos, &K->kzg_g1)) return false; pos += 8;
  if (!get_g2(w + pos, &K->kzg_g2) || !get_g2(w + pos + 16, &K->kzg_vk)) return false; pos += 32;
  if (pos != n) return false;
  if (!g1_on_curve(K->alpha) || !g1_on_curve(K->kzg_g1)) return false;
  for (auto& p : K->ic) if (!g1_on_curve(p)) return false;
  for (const G2PAff* p : {&K->beta, &K->gamma, &K->delta, &K->kzg_g2, &K->kzg_vk}) if (!g2_on_curve(*p) || !g2_in_subgroup(*p)) return false;
  return true;
}
VerifierKey key_of(const vimz_decider* d) {
  VerifierKey K; const G16Key& G = d->key;
  K.pp_hash = d->vk->c1->digest; K.len_z = d->circ.len_z;
  K.alpha = G.alpha1; K.beta = G.beta2; K.gamma = G.gamma2; K.delta = G.delta2; K.ic = G.ic;
  K.kzg_g1 = g1_generator(); K.kzg_g2 = g2_generator(); K.kzg_vk = d->kzg_vk;
  return K;
}

}  // namespace

extern "C" {

void vimz_decider_free(vimz_decider* d) {
  if (!d) return;
  if (d->ctx) {
    std::unique_lock<std::mutex> g(d->ctx->mu);
    hipSetDevice(d->ctx->device);
    hipStreamSynchronize(d->ctx->stream);
  }
  delete d;
}

// KZG::setup (Sonobe's `KZG::setup(rng, n)` inside vimz/src/sonobe_backend/folding.rs:36-48 `prepare_folding`): tau from the OS, used and forgotten
int vimz_kzg_setup(vimz_ctx* ctx, size_t n, vimz_bases** srs_out, uint64_t vk_g2_out[16]) {
  if (!ctx || !srs_out || !vk_g2_out || !n) return VIMZ_ERR_INVALID;
  Trapdoor t;
  if (!trapdoor_random(t)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_kzg_setup: no randomness from the OS");
  return kzg_setup_impl(ctx, t.tau, n, srs_out, vk_g2_out);
}

// Decider::preprocess (vimz/src/sonobe_backend/mod.rs:72-75): the decider circuit for this prover's shapes and a Groth16 key pair.  The toxic
// waste is drawn from the OS's randomness, used and forgotten: a LOCALLY TRUSTED setup (whoever runs it could have kept it) — what the
// reference's own `StdRng::from_seed([41; 32])` (mod.rs:54) is, too; a deployment would import keys from a ceremony instead.
// kzg_vk_g2 (optional): [tau]G2 of the SRS the prover's ck_main is made of (vimz_kzg_setup) — needed by vimz_decider_verify and part of vimz_decider_vk.
// seconds (optional) = {circuit synthesis, QAP evaluation at the trapdoor, key points on the GPU, total}.
// light != 0: the shrunken circuit of the reference's opt-in `light-test` feature (vimz/Cargo.toml:56-59; checks 1-4 of aug/decider.hpp: the CycleFold
// instance is bound by its hash only); light == 0: the full decider, as `decider.rs:13-21` instantiates it (≈ 2.75 M constraints more: set-up in seconds).
int vimz_decider_setup(vimz_cf* v, const uint64_t kzg_vk_g2[16], int light, vimz_decider** out, double seconds[4]) {
  if (!v || !out) return VIMZ_ERR_INVALID;
  Trapdoor t;
  if (!trapdoor_random(t)) return vz_fail(v->ctx, VIMZ_ERR_INVALID, "vimz_decider_setup: no randomness from the OS");
  return decider_setup_impl(v, kzg_vk_g2, t, light != 0, out, seconds);
}
#ifdef VIMZ_TESTING
// Host only, no GPU: the decider circuit (aug/decider.hpp) over the Nova + CycleFold recursion of the trivial step circuit with made-up commitments — the
// run of vimz_cf_selfcheck, with the running relaxed witness kept by folding on the host — then the decider's final fold, its witness, and the checks:
// result bit 0 a step's F' witness is bad, 1 the folded scalars differ from F''s, 2 the decider's witness generator flags the honest input, 3 its witness
// violates a row of the decider's R1CS, 4 the public inputs are not the contract's layout (pp_hash, i, z_0, z_i, 4 x 5 limbs, c_W, c_E, e_W, e_E, 2 x 5 limbs),
// 5 a wrong KZG evaluation / a wrong challenge / a changed cmT limb is NOT flagged, 6 a changed public input leaves every row satisfied.
// full != 0: the FULL decider (aug/decider_cf.hpp) — the CycleFold instances of every step are real here (witnesses of the CycleFold circuit for the step's
// two folds, committed under a made-up key G_k = (k + 1)·G, folded on the host into a running witness and error vector): 7 the host's running CycleFold
// instance differs from the one F' folds in-circuit, 8 a running CycleFold witness / error vector that violates its relation, or a commitment that does not
// open to it, is NOT flagged by the witness generator, 9 such a witness leaves every row of the decider satisfied.
// counts = {decider constraints, wires, public inputs, main constraints}.
int vimz_decider_selfcheck(int steps, int full, uint32_t* result, uint64_t counts[4]) {
  if (!result || steps < 2 || steps > 32) return VIMZ_ERR_INVALID;
  try {
    CfCircuit cf; cf.finish();
    cb::BuilderT<Fe> b;
    b.len_z = 1; b.n_priv = 0; b.n_wires = 3;
    b.enforce(cb::LCT<Fe>::constant(Fe::one()), cb::LCT<Fe>::wire(2), cb::LCT<Fe>::wire(1));
    b.n_linear = 1;
    CfMainCircuit c1(b); c1.use_worker = false; c1.finish(cf);
    const uint32_t nw = b.n_wires, nc = b.n_constraints();
    uint32_t res = 0;
    const G1Aff g1 = CycleSide<BnFq>::G(); const G2Aff g2 = CycleSide<BnFr>::G();
    auto fake1 = [&](uint64_t k) { const uint32_t w[2] = {(uint32_t)k, (uint32_t)(k >> 32)}; return to_affine(scalar_mul(g1, w, 64)); };
    auto fake2 = [&](uint64_t k) { const uint32_t w[2] = {(uint32_t)k, (uint32_t)(k >> 32)}; return to_affine(host_mul<Fe>(g2, w, 64)); };
    std::vector<Fe> z0 = {cb::f_from_u64<Fe>(7)};
    CfMainRelaxed U = CfMainRelaxed::zero(); G1Aff UW = g1_identity(), UE = g1_identity();
    CfMainFresh u = CfMainFresh::zero(); G1Aff uW = g1_identity();
    CfRelaxed cfU = CfRelaxed::zero();
    std::vector<Fe> Zrun(nw, Fe::zero()), Erun(nc, Fe::zero()), z_last;
    // full: the running CycleFold pair on the host, committed under G_k = (k + 1)·G (a made-up key: Σ v_k·G_k is one multiplication by Σ (k + 1)·v_k)
    const uint32_t cnw = cf.b.n_wires, cnc = cf.b.n_constraints(), cnW = cnw - 1 - aug::CF_IO;
    std::vector<Fq> cfZ(cnw, Fq::zero()), cfE(cnc, Fq::zero());
    G2Aff cf_cmW = g2_identity(), cf_cmE = g2_identity();
    auto cf_commit = [&](const Fq* v, size_t n) {
      Fq s = Fq::zero();
      for (size_t k = 0; k < n; k++) s = Fq::add(s, Fq::mul(v[k], cb::f_from_u64<Fq>(k + 1)));
      const Fq c = Fq::from_mont(s);
      return to_affine(host_mul<Fe>(g2, c.v, 256));
    };
    auto g2_fold = [&](const G2Aff& P, const uint32_t low[4], const G2Aff& Q) {
      const uint32_t k[5] = {low[0], low[1], low[2], low[3], 1u};
      G2 a = aff_is_identity(P) ? G2::identity() : from_affine(P);
      if (!aff_is_identity(Q)) { G2 t = host_mul<Fe>(Q, k, 129); add_full(a, t); }
      return to_affine(a);
    };
    // one CycleFold instance (r, P1, P2 -> P3 = P1 + r·P2): witness, commitments, and — once the caller has the challenge — the fold into the running pair
    struct CfInst { std::vector<Fq> z, T; G2Aff cmW, cmT; };
    auto cf_instance = [&](const uint32_t r_low[4], const G1Aff& P1, const G1Aff& P2, CfInst& o) {
      bool bad = false;
      cf.witness(r_low, P1, P2, o.z, &bad);
      if (bad) res |= 1;
      std::vector<Fq> p1[3], p2[3];
      host_spmv3(cf.b, cfZ, p1); host_spmv3(cf.b, o.z, p2);
      o.T.resize(cnc);
      for (uint32_t k = 0; k < cnc; k++) o.T[k] = Fq::sub(Fq::sub(Fq::add(Fq::mul(p1[0][k], p2[1][k]), Fq::mul(p2[0][k], p1[1][k])), Fq::mul(cfZ[0], p2[2][k])), p1[2][k]);
      o.cmW = cf_commit(o.z.data() + 1, cnW); o.cmT = cf_commit(o.T.data(), cnc);
    };
    auto cf_fold_in = [&](const CfInst& o, const uint32_t r_low[4]) {
      const Fq r = rho_element<Fq>(r_low);
      for (uint32_t k = 0; k < cnw; k++) cfZ[k] = Fq::add(cfZ[k], Fq::mul(r, o.z[k]));
      for (uint32_t k = 0; k < cnc; k++) cfE[k] = Fq::add(cfE[k], Fq::mul(r, o.T[k]));
      cf_cmW = g2_fold(cf_cmW, r_low, o.cmW); cf_cmE = g2_fold(cf_cmE, r_low, o.cmT);
    };
    // x1 += r·x2 over the main shape, with the cross term folded into E:  returns nothing, updates Zrun / Erun
    auto fold_in = [&](const std::vector<Fe>& z2, const Fe& r) {
      std::vector<Fe> p1[3], p2[3];
      host_spmv3(b, Zrun, p1); host_spmv3(b, z2, p2);
      const Fe u1 = Zrun[0];
      for (uint32_t k = 0; k < nc; k++) {
        const Fe T = Fe::sub(Fe::sub(Fe::add(Fe::mul(p1[0][k], p2[1][k]), Fe::mul(p2[0][k], p1[1][k])), Fe::mul(u1, p2[2][k])), p1[2][k]);      // u_2 = 1
        Erun[k] = Fe::add(Erun[k], Fe::mul(r, T));
      }
      for (uint32_t w = 0; w < nw; w++) Zrun[w] = Fe::add(Zrun[w], Fe::mul(r, z2[w]));
    };
    for (int i = 0; i < steps; i++) {
      CfMainIn in = CfMainIn::zero();
      in.digest = c1.digest; in.i = (uint64_t)i; in.z0 = z0; in.U = U; in.u = u; in.cfU = cfU;
      CfChallenges ch; ch.h_U = cf_hash_main(c1.digest, i, z0, z0.data(), U); ch.h_cf = cf_hash_cf(c1.digest, cfU);
      G1Aff Wn = g1_identity(), En = g1_identity();
      if (i > 0) {
        const G1Aff cT = i > 1 ? fake1(0x1000 + i) : g1_identity();
        in.T = nn_point(cT);
        cf_challenge_main(ch, u, in.T);
        Wn = g1_fold(UW, ch.r, uW); En = g1_fold(UE, ch.r, cT);
        in.Wn = nn_point(Wn); in.En = nn_point(En);
        if (full) {
          CfInst i1, i2;
          cf_instance(ch.r, UW, uW, i1);
          in.cf1W = i1.cmW; in.cf1T = i1.cmT;
          cf_challenge_cf1(ch, in.cf1W, in.Wn, in.cf1T);
          cf_fold_in(i1, ch.r1);
          cf_instance(ch.r, UE, cT, i2);
          in.cf2W = i2.cmW; in.cf2T = i2.cmT;
          cf_challenge_cf2(ch, in.cf2W, in.En, in.cf2T);
          cf_fold_in(i2, ch.r2);
        } else {
          in.cf1W = fake2(0x2000 + i); in.cf1T = i > 1 ? fake2(0x3000 + i) : g2_identity();
          in.cf2W = fake2(0x4000 + i); in.cf2T = fake2(0x5000 + i);
        }
        fold_in(z_last, rho_element<Fe>(ch.r));      // what this step's F' verifies: U_i = U_{i-1} (+) u_{i-1}
      }
      std::vector<Fe> augw; bool bad = false;
      CfMainOut o = c1.witness(in, z0.data(), z0.data(), augw, &bad);
      if (bad) res |= 1;
      if (full && i > 0) {      // the instance F' folded in-circuit is the one the host keeps
        const CfRelaxed& n = o.cfU_new;
        if (!n.W.x.eq(cf_cmW.x) || !n.W.y.eq(cf_cmW.y) || !n.E.x.eq(cf_cmE.x) || !n.E.y.eq(cf_cmE.y) || !cross_field<Fe>(cfZ[0]).eq(n.u)) res |= 128;
        for (int k = 0; k < aug::CF_IO; k++) if (memcmp(to_u256(cfZ[cnw - aug::CF_IO + k]).w, n.x[k].w, 32)) res |= 128;
      }
      std::vector<Fe> z = {Fe::one(), z0[0], z0[0]};
      z.insert(z.end(), augw.begin(), augw.end());
      if (i > 0 && (!o.U_new.u.eq(Zrun[0]) || !o.U_new.x0.eq(Zrun[nw - 2]) || !o.U_new.x1.eq(Zrun[nw - 1]))) res |= 2;
      U = o.U_new; UW = i > 0 ? Wn : g1_identity(); UE = i > 0 ? En : g1_identity(); cfU = o.cfU_new;
      uW = fake1(0x6000 + i); u.W = nn_point(uW); u.x0 = o.x0; u.x1 = o.x1;
      z_last = z;
    }
    // the decider's final fold U_{n+1} = U_n (+) u_n, on the host
    aug::DeciderIn in;
    in.digest = c1.digest; in.i = (uint64_t)steps; in.z0 = z0; in.zi = z0; in.U = U; in.u = u; in.cfU = cfU;
    const G1Aff cmT = fake1(0x7777);
    in.cmT = nn_point(cmT);
    uint32_t r_low[4]; aug::decider_challenge(u, in.cmT, r_low);
    fold_in(z_last, rho_element<Fe>(r_low));
    in.Wn = nn_point(g1_fold(UW, r_low, uW)); in.En = nn_point(g1_fold(UE, r_low, cmT));
    const Fe cW = aug::decider_kzg_challenge(in.digest, in.Wn), cE = aug::decider_kzg_challenge(in.digest, in.En);
    auto horner = [](const Fe* v, size_t n, const Fe& c) { Fe acc = Fe::zero(); for (size_t j = n; j-- > 0;) acc = Fe::add(Fe::mul(acc, c), v[j]); return acc; };
    in.eW = horner(Zrun.data() + 1, nw - 3, cW); in.eE = horner(Erun.data(), nc, cE);
    in.Wf = Zrun.data() + 1; in.Ef = Erun.data();
    aug::DeciderCircuit dc;
    if (full) {
      std::vector<G2Aff> gens(std::max(cnW, cnc));
      G2 acc = from_affine(g2);
      for (auto& p : gens) { p = to_affine(acc); add_mixed(acc, g2); }
      dc.finish(b, 1, &cf.b, gens.data(), (uint32_t)gens.size());
      in.full.W = cfZ.data() + 1; in.full.E = cfE.data();
    } else dc.finish(b, 1);
    if (counts) { counts[0] = dc.b.n_constraints(); counts[1] = dc.b.n_wires; counts[2] = dc.n_public; counts[3] = nc; }
    bool bad = false;
    std::vector<Fe> zd = dc.witness(b, in, &bad);
    if (bad) res |= 4;
    auto violated = [&](const std::vector<Fe>& zz) { std::vector<Fe> p[3]; host_spmv3(dc.b, zz, p); for (uint32_t k = 0; k < dc.b.n_constraints(); k++) if (!Fe::mul(p[0][k], p[1][k]).eq(p[2][k])) return true; return false; };
    if (violated(zd)) res |= 8;
    // the public inputs, in the contract's order
    { std::vector<Fe> want = {in.digest, cb::f_from_u64<Fe>((uint64_t)steps), z0[0], z0[0]};
      for (const U256w* c : {&in.Wn.x, &in.Wn.y, &in.En.x, &in.En.y}) { uint64_t l[aug::DEC_LIMBS]; aug::decider_limbs55(*c, l); for (int k = 0; k < aug::DEC_LIMBS; k++) want.push_back(cb::f_from_u64<Fe>(l[k])); }
      want.push_back(cW); want.push_back(cE); want.push_back(in.eW); want.push_back(in.eE);
      for (const U256w* c : {&in.cmT.x, &in.cmT.y}) { uint64_t l[aug::DEC_LIMBS]; aug::decider_limbs55(*c, l); for (int k = 0; k < aug::DEC_LIMBS; k++) want.push_back(cb::f_from_u64<Fe>(l[k])); }
      if (want.size() != dc.n_public || dc.n_public != aug::decider_n_public(1)) res |= 16;
      else for (size_t k = 0; k < want.size(); k++) if (!zd[1 + k].eq(want[k])) res |= 16; }
    // negatives: the witness generator flags a wrong evaluation and a cross-term commitment other than the one the challenge was derived from
    { aug::DeciderIn t = in; t.eW = Fe::add(t.eW, Fe::one()); bool b2 = false; dc.witness(b, t, &b2); if (!b2) res |= 32; }
    { aug::DeciderIn t = in; t.cmT = nn_point(fake1(0x7778)); bool b2 = false; dc.witness(b, t, &b2); if (!b2) res |= 32; }      // (another r: the folded scalars no longer match W')
    // ... and every public input matters to some row
    for (uint32_t k = 0; k < dc.n_public; k++) { std::vector<Fe> zz = zd; zz[1 + k] = Fe::add(zz[1 + k], Fe::one()); if (!violated(zz)) res |= 64; }
    if (full) {
      // negatives of checks 5 and 6: a changed witness element (the commitment no longer opens, a row no longer holds), a changed error element,
      // and a witness changed TOGETHER with its commitment (opens, but the relation fails: check 6 alone)
      auto flagged = [&](const std::vector<Fq>& Zc, const std::vector<Fq>& Ec) {
        aug::DeciderIn t = in; t.full.W = Zc.data() + 1; t.full.E = Ec.data();
        bool b2 = false; std::vector<Fe> zz = dc.witness(b, t, &b2);
        return std::make_pair(b2, violated(zz));
      };
      { std::vector<Fq> Zc = cfZ; Zc[5] = Fq::add(Zc[5], Fq::one()); auto f = flagged(Zc, cfE); if (!f.first) res |= 256; if (!f.second) res |= 512; }
      { std::vector<Fq> Ec = cfE; Ec[3] = Fq::add(Ec[3], Fq::one()); auto f = flagged(cfZ, Ec); if (!f.first) res |= 256; if (!f.second) res |= 512; }
      // (under G_k = (k + 1)·G, adding b + 1 to element a and taking a + 1 off element b leaves the commitment as it was: only the relation can object)
      { std::vector<Fq> Zc = cfZ; Zc[1 + 4] = Fq::add(Zc[1 + 4], cb::f_from_u64<Fq>(10)); Zc[1 + 9] = Fq::sub(Zc[1 + 9], cb::f_from_u64<Fq>(5));
        if (!(cf_commit(Zc.data() + 1, cnW).x.eq(cf_cmW.x))) res |= 256;
        auto f = flagged(Zc, cfE); if (!f.first) res |= 256; if (!f.second) res |= 512; }
    }
    *result = res;
    return VIMZ_OK;
  } catch (const std::exception& e) { return vz_fail(nullptr, VIMZ_ERR_INVALID, e.what()); }
}
// deterministic TEST setups: the trapdoor is derived from `seed` — anyone who knows the seed can forge.  Not in the product library.
int vimz_testing_kzg_setup_seeded(vimz_ctx* ctx, const uint8_t* seed, size_t seed_len, size_t n, vimz_bases** srs_out, uint64_t vk_g2_out[16]) {
  if (!ctx || !srs_out || !vk_g2_out || !n || (!seed && seed_len)) return VIMZ_ERR_INVALID;
  return kzg_setup_impl(ctx, fr_from_hash(seed, seed_len, "vimz-kzg-tau"), n, srs_out, vk_g2_out);
}
int vimz_testing_decider_setup_seeded(vimz_cf* v, const uint64_t kzg_vk_g2[16], int light, const uint8_t* seed, size_t seed_len, vimz_decider** out, double seconds[4]) {
  if (!v || !out || (!seed && seed_len)) return VIMZ_ERR_INVALID;
  return decider_setup_impl(v, kzg_vk_g2, trapdoor_seeded(seed, seed_len), light != 0, out, seconds);
}
#endif

// info = {constraints, wires, public inputs, domain size, non-zeros of A, B, C, rows of checks 5 and 6 (0: the light decider)}
int vimz_decider_info(const vimz_decider* d, uint64_t info[8]) {
  if (!d || !info) return VIMZ_ERR_INVALID;
  const cb::BuilderT<Fe>& b = d->circ.b;
  info[0] = d->key.n_c; info[1] = d->key.m; info[2] = d->key.n_pub; info[3] = d->key.n; info[4] = b.A.col.size(); info[5] = b.B.col.size(); info[6] = b.C.col.size(); info[7] = d->circ.full ? d->key.n_c - d->circ.light_constraints : 0;
  return VIMZ_OK;
}

// The verifying key — everything a contract generated for this circuit would hold as constants (contracts/ContrastVerifier.sol:140-160 KZG G_1 / G_2 / VK,
// :238-… alpha…delta and IC_k, :703 the public-parameter hash) — as canonical little-endian words: pp_hash (4), len_z (1), alpha (G1: x, y), beta,
// gamma, delta (G2: x.c0, x.c1, y.c0, y.c1), the number of IC points, the IC points (G1), then KZG G_1 (G1), G_2, VK (G2; zeros when the decider was
// set up without one).  Returns the byte size (copies when buf is large enough).
int64_t vimz_decider_vk(const vimz_decider* d, void* buf, size_t cap) {
  if (!d) return VIMZ_ERR_INVALID;
  Writer w;
  const G16Key& K = d->key;
  w.fe(d->vk->c1->digest); w.word(d->circ.len_z);
  w.point(K.alpha1);
  auto g2w = [&](const G2PAff& p) { w.fe(p.x.c0); w.fe(p.x.c1); w.fe(p.y.c0); w.fe(p.y.c1); };
  for (const G2PAff* p : {&K.beta2, &K.gamma2, &K.delta2}) g2w(*p);
  w.word(K.ic.size());
  for (auto& p : K.ic) w.point(p);
  w.point(g1_generator()); g2w(g2_generator()); g2w(d->kzg_vk);
  const size_t bytes = 8 * w.w.size();
  if (buf && cap >= bytes) memcpy(buf, w.w.data(), bytes);
  return (int64_t)bytes;
}

// ---- the key at rest: vimz_decider_key_save / _load -------------------------------------------------------------------------------------------------------
// A Groth16 key pair for the decider circuit as bytes, so that a set-up is made ONCE per circuit and shape (0.4 s instead of 0.8 s per run at contrast HD) and —
// the point of ADVICE r4 — so that keys made ELSEWHERE (a ceremony's, converted to this layout) can be used: the library then never sees a trapdoor.
// Layout (little-endian u64 words; curve points canonical, G1 (x, y), G2 (x.c0, x.c1, y.c0, y.c1)): magic, m, n_pub, n_c, n, len_z, mode (0 full, 1 light), pp_hash (4), KZG [tau]G2 (16),
// alpha1, beta1, delta1 (8 each), beta2, gamma2, delta2 (16 each), IC (8 x (n_pub + 1)), then the queries a (8 m), b1 (8 m), l (8 (m - n_pub - 1)), h (8 (n - 1)), b2 (16 m).
// A loaded key is TRUSTED like any common reference string: its points are range- and curve-checked for the verifying part, the queries are taken as they are
// (a wrong query makes proofs that do not verify, nothing worse).
static const uint64_t G16_KEY_MAGIC = 0x3259454b36314756ull;      // "VG16KEY2"
static const size_t G16_KEY_HEADER = 7;
int64_t vimz_decider_key_save(vimz_decider* d, void* buf, size_t cap) {
  if (!d) return VIMZ_ERR_INVALID;
  const G16Key& K = d->key; vimz_ctx* ctx = d->ctx;
  const size_t nq[4] = {K.m, K.m, (size_t)K.m - K.n_pub - 1, (size_t)K.n - 1};
  const size_t words = G16_KEY_HEADER + 4 + 16 + 24 + 48 + 8 * (size_t)(K.n_pub + 1) + 8 * (nq[0] + nq[1] + nq[2] + nq[3]) + 16 * (size_t)K.m;
  if (!buf || cap < 8 * words) return (int64_t)(8 * words);
  uint64_t* w = (uint64_t*)buf; size_t pos = 0;
  w[pos++] = G16_KEY_MAGIC; w[pos++] = K.m; w[pos++] = K.n_pub; w[pos++] = K.n_c; w[pos++] = K.n; w[pos++] = d->circ.len_z; w[pos++] = d->circ.full ? 0 : 1;
  { const Fe c = Fe::from_mont(d->vk->c1->digest); memcpy(w + pos, c.v, 32); pos += 4; }
  put_g2(w + pos, d->kzg_vk); pos += 16;
  for (const G1Aff* p : {&K.alpha1, &K.beta1, &K.delta1}) { put_fq(w + pos, p->x); put_fq(w + pos + 4, p->y); pos += 8; }
  for (const G2PAff* p : {&K.beta2, &K.gamma2, &K.delta2}) { put_g2(w + pos, *p); pos += 16; }
  for (auto& p : K.ic) { put_fq(w + pos, p.x); put_fq(w + pos + 4, p.y); pos += 8; }
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const vimz_bases* qs[4] = {K.a_q, K.b1_q, K.l_q, K.h_q};
  uint32_t* tmp = nullptr;
  P_TRY(hipMalloc((void**)&tmp, 64 * (size_t)std::max<size_t>(K.m, K.n)));
  struct FreeTmp { uint32_t* q; ~FreeTmp() { hipFree(q); } } ft{tmp};
  for (int q = 0; q < 4; q++) {
    if (nq[q]) {
      launch_points_from_internal<Fq>(s, qs[q]->d, 1, tmp, nq[q]);
      P_TRY(hipGetLastError());
      P_TRY(hipMemcpyAsync(w + pos, tmp, 64 * nq[q], hipMemcpyDeviceToHost, s));
      P_TRY(hipStreamSynchronize(s));
    }
    pos += 8 * nq[q];
  }
  { std::vector<G2PAff> b2c(K.n_b2); std::vector<uint32_t> ix(K.n_b2);
    P_TRY(hipMemcpyAsync(b2c.data(), K.b2_q, sizeof(G2PAff) * (size_t)K.n_b2, hipMemcpyDeviceToHost, s));
    P_TRY(hipMemcpyAsync(ix.data(), K.b2_idx, 4 * (size_t)K.n_b2, hipMemcpyDeviceToHost, s));
    P_TRY(hipStreamSynchronize(s));
    memset(w + pos, 0, 128 * (size_t)K.m);      // (the identity: (0, 0))
    for (uint32_t k = 0; k < K.n_b2; k++) put_g2(w + pos + 16 * (size_t)ix[k], b2c[k]);
    pos += 16 * (size_t)K.m; }
  return pos == words ? (int64_t)(8 * words) : (int64_t)VIMZ_ERR_INVALID;
}
// prover: as for vimz_decider_setup (shapes, keys, context).  The key must be for exactly this prover's circuits (sizes and public-parameter hash are checked).
int vimz_decider_key_load(vimz_cf* v, const void* buf, size_t len, vimz_decider** out) {
  if (!v || !buf || !out || (len & 7)) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = v->ctx;
  const uint64_t* w = (const uint64_t*)buf; const size_t nwords = len / 8;
  if (nwords < G16_KEY_HEADER + 4 + 16 + 24 + 48 || w[0] != G16_KEY_MAGIC || w[6] > 1) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_key_load: not a decider key");
  std::unique_ptr<vimz_decider> d(new vimz_decider());
  d->vk = v; d->ctx = ctx;
  { const int rcb = decider_circuit_build(v, w[6] == 1, d->circ); if (rcb) return rcb; }
  G16Key& K = d->key;
  const cb::BuilderT<Fe>& b = d->circ.b;
  K.m = b.n_wires; K.n_pub = d->circ.n_public; K.n_c = b.n_constraints();
  K.n = 1; K.logn = 0;
  while (K.n < K.n_c + K.n_pub + 1) { K.n <<= 1; K.logn++; }
  size_t pos = 1;
  if (w[pos] != K.m || w[pos + 1] != K.n_pub || w[pos + 2] != K.n_c || w[pos + 3] != K.n || w[pos + 4] != d->circ.len_z)
    return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_key_load: the key is for a decider circuit of other sizes");
  pos += 6;
  { Fe c; memcpy(c.v, w + pos, 32); if (!c.is_reduced() || !Fe::to_mont(c).eq(v->c1->digest)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_key_load: the key is for other circuits (public-parameter hash)"); pos += 4; }
  const size_t nq[4] = {K.m, K.m, (size_t)K.m - K.n_pub - 1, (size_t)K.n - 1};
  const size_t words = G16_KEY_HEADER + 4 + 16 + 24 + 48 + 8 * (size_t)(K.n_pub + 1) + 8 * (nq[0] + nq[1] + nq[2] + nq[3]) + 16 * (size_t)K.m;
  if (nwords != words) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_key_load: wrong length");
  using vz::pairing::g1_on_curve; using vz::pairing::g2_in_subgroup;
  if (!get_g2(w + pos, &d->kzg_vk) || !g2_on_curve(d->kzg_vk) || !g2_in_subgroup(d->kzg_vk)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_key_load: KZG verifying key"); pos += 16;
  if (!aff_is_identity(d->kzg_vk)) { const int rcv = kzg_vk_matches_srs(v, d->kzg_vk, "vimz_decider_key_load"); if (rcv) return rcv; }
  for (G1Aff* p : {&K.alpha1, &K.beta1, &K.delta1}) { if (!get_g1(w + pos, p) || !g1_on_curve(*p)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_key_load: a G1 key point"); pos += 8; }
  for (G2PAff* p : {&K.beta2, &K.gamma2, &K.delta2}) { if (!get_g2(w + pos, p) || !g2_on_curve(*p) || !g2_in_subgroup(*p)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_key_load: a G2 key point"); pos += 16; }
  K.ic.resize(K.n_pub + 1);
  for (auto& p : K.ic) { if (!get_g1(w + pos, &p) || !g1_on_curve(p)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_key_load: an IC point"); pos += 8; }
  // domain constants (as in the set-up)
  K.omega = fr_root_of_unity(K.logn); K.omega_inv = Fe::pow_pm2(K.omega);
  K.n_inv = Fe::pow_pm2(cb::f_from_u64<Fe>(K.n));
  K.coset = cb::f_from_u64<Fe>(5); K.coset_inv = Fe::pow_pm2(K.coset);
  K.zinv = Fe::pow_pm2(Fe::sub(fr_pow_u64(K.coset, K.n), Fe::one()));
  std::lock_guard<std::mutex> g(ctx->mu);
  P_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  uint32_t* tmp = nullptr;
  P_TRY(hipMalloc((void**)&tmp, 64 * (size_t)std::max<size_t>(K.m, K.n)));
  struct FreeTmp { uint32_t* q; ~FreeTmp() { hipFree(q); } } ft{tmp};
  vimz_bases** qs[4] = {&K.a_q, &K.b1_q, &K.l_q, &K.h_q};
  for (int q = 0; q < 4; q++) {
    if (nq[q]) { P_TRY(hipMemcpyAsync(tmp, w + pos, 64 * nq[q], hipMemcpyHostToDevice, s)); P_TRY(hipStreamSynchronize(s)); }
    const int rc = bases_from_device(ctx, (const G1Aff*)tmp, nq[q], qs[q], 1);
    if (rc) return rc;
    pos += 8 * nq[q];
  }
  { std::vector<G2PAff> b2; std::vector<uint32_t> ix;
    for (uint32_t i = 0; i < K.m; i++) {
      G2PAff pt;
      if (!get_g2(w + pos, &pt)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_key_load: a G2 query coordinate is not below the modulus");
      pos += 16;
      if (!aff_is_identity(pt)) { b2.push_back(pt); ix.push_back(i); }
    }
    K.n_b2 = (uint32_t)ix.size();
    P_TRY(hipMalloc((void**)&K.b2_q, sizeof(G2PAff) * std::max<size_t>(b2.size(), 1)));
    P_TRY(hipMemcpy(K.b2_q, b2.data(), sizeof(G2PAff) * b2.size(), hipMemcpyHostToDevice));
    P_TRY(hipMalloc((void**)&K.b2_idx, 4 * std::max<size_t>(ix.size(), 1)));
    P_TRY(hipMemcpy(K.b2_idx, ix.data(), 4 * ix.size(), hipMemcpyHostToDevice)); }
  P_TRY(hipMalloc((void**)&K.tw, 32 * (size_t)std::max<uint32_t>(K.n / 2, 1))); P_TRY(hipMalloc((void**)&K.tw_inv, 32 * (size_t)std::max<uint32_t>(K.n / 2, 1)));
  { Fr tw, twi; memcpy(tw.v, K.omega.v, 32); memcpy(twi.v, K.omega_inv.v, 32);
    hipLaunchKernelGGL(k_pow_table, dim3((K.n / 2 + 255) / 256), dim3(256), 0, s, K.tw, (size_t)K.n / 2, tw);
    hipLaunchKernelGGL(k_pow_table, dim3((K.n / 2 + 255) / 256), dim3(256), 0, s, K.tw_inv, (size_t)K.n / 2, twi);
    P_TRY(hipGetLastError()); P_TRY(hipStreamSynchronize(s)); }
  *out = d.release();
  return VIMZ_OK;
}

// Decider::prove (mod.rs:76-78) for the IVC proof `ivc` holds after i >= 1 steps (left unchanged): the final fold U_{i+1} = NIFS(U_i, u_i) on the GPU
// (cross term, its commitment, the challenge of aug/decider.hpp, the folded witness and error vector), the KZG openings of U_{i+1}'s two commitments at
// the challenges that follow from them, and the Groth16 proof of the decider circuit — ALL 25 calldata words (contracts/*Verifier.sol:785-810;
// vimz_amd/calldata.py names them), canonical, 4 little-endian limbs each.  public_out: the info[2] public inputs (canonical), as the contract
// assembles them from the words.  VIMZ_ERR_UNSAT when the IVC proof does not satisfy the decider's statement.
// seconds (optional) = {final fold + KZG openings (GPU), witness + sparse products (host), NTTs, MSMs, total, 0}.
int vimz_decider_prove(vimz_decider* d, vimz_cf* ivc, uint64_t* public_out, uint64_t words_out[100], double seconds[6]) {
  if (!d || !ivc || !public_out || !words_out) return VIMZ_ERR_INVALID;
  vimz_ctx* ctx = d->ctx; vimz_cf* v = d->vk;
  const G16Key& K = d->key;
  const double t_all = now_s();
  if (ivc->ctx != ctx || ivc->pri->n_wires != v->pri->n_wires || ivc->pri->n_c != v->pri->n_c || !ivc->c1->digest.eq(v->c1->digest) || ivc->ck1 != v->ck1)
    return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_prove: the IVC proof belongs to other shapes / keys than the decider was set up for");
  if (ivc->broken) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_prove: the IVC failed in the middle of a step");
  if (ivc->i == 0) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_prove: the IVC has no steps");
  vimz_prover* p = ivc->pri;
  const size_t nw = p->n_wires, nc = p->n_c;
  std::vector<Fe> Wf(nw), Ef(nc);
  aug::DeciderIn in;
  G1Aff cmT, cWn, cEn;
  uint32_t r_low[4];
  uint64_t kzg_out[2][12];      // eval (4), proof (8)
  Fe cW, cE;
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    P_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    for (hipStream_t q : {ivc->s2, ivc->s3, ivc->s4}) if (q) P_TRY(hipStreamSynchronize(q));
    P_TRY(hipStreamSynchronize(s));
    uint32_t* dev = nullptr;
    P_TRY(hipMalloc((void**)&dev, 32 * (nw + 2 * nc)));
    struct FreeDev { uint32_t* q; ~FreeDev() { hipFree(q); } } fd{dev};
    uint32_t *Wd = dev, *Ed = dev + 8 * nw, *Td = Ed + 8 * nc;
    P_TRY(hipMemcpyAsync(Wd, p->Zrun, 32 * nw, hipMemcpyDeviceToDevice, s));
    P_TRY(hipMemcpyAsync(Ed, p->E, 32 * nc, hipMemcpyDeviceToDevice, s));
    // cross term of (U_i, W_i) and the last instance of F' (strict: u = 1), its commitment
    hipLaunchKernelGGL(k_cross_term<Fr>, dim3(stream_grid(nc)), dim3(256), 0, s, nc, p->AZ, p->BZ, p->CZ, ivc->u_run, ivc->azl, ivc->bzl, ivc->czl, Fe::one(), Td);
    P_TRY(hipGetLastError());
    uint64_t pt[8];
    int rc = vz_msm_device(ctx, p->ck, 0, Td, nc, 1, 0, pt, VIMZ_FORM_MONTGOMERY);
    if (rc) return rc;
    memcpy(cmT.x.v, pt, 32); memcpy(cmT.y.v, pt + 4, 32);
    // the challenge, as the decider circuit derives it
    aug::decider_challenge(ivc->u, nn_point(cmT), r_low);
    const Fe rho = cf_r_element_fr(r_low);
    Fold5 f; for (int k = 0; k < 5; k++) { f.x1[k] = nullptr; f.x2[k] = nullptr; f.n[k] = 0; }
    f.x1[0] = Wd; f.x2[0] = ivc->Zl; f.n[0] = nw;
    f.x1[1] = Ed; f.x2[1] = Td; f.n[1] = nc;
    hipLaunchKernelGGL(k_fold5<Fr>, dim3(1024), dim3(256), 0, s, f, rho);
    P_TRY(hipGetLastError());
    cWn = g1_fold(ivc->UW, r_low, ivc->uW); cEn = g1_fold(ivc->UE, r_low, cmT);
    in.digest = v->c1->digest;
    in.Wn = nn_point(cWn); in.En = nn_point(cEn); in.cmT = nn_point(cmT);
    cW = aug::decider_kzg_challenge(in.digest, in.Wn); cE = aug::decider_kzg_challenge(in.digest, in.En);
    uint64_t ch[4];
    { const Fe c = Fe::from_mont(cW); memcpy(ch, c.v, 32); }
    if ((rc = vz_kzg_open_device(ctx, p->ck, 0, VIMZ_FIELD_BN254_FR, Wd + 8, nw - 3, ch, VIMZ_FORM_CANONICAL, kzg_out[0], kzg_out[0] + 4))) return rc;
    { const Fe c = Fe::from_mont(cE); memcpy(ch, c.v, 32); }
    if ((rc = vz_kzg_open_device(ctx, p->ck, 0, VIMZ_FIELD_BN254_FR, Ed, nc, ch, VIMZ_FORM_CANONICAL, kzg_out[1], kzg_out[1] + 4))) return rc;
    P_TRY(hipMemcpyAsync(Wf.data(), Wd, 32 * nw, hipMemcpyDeviceToHost, s));
    P_TRY(hipMemcpyAsync(Ef.data(), Ed, 32 * nc, hipMemcpyDeviceToHost, s));
    P_TRY(hipStreamSynchronize(s));
  }
  const double t_fold = now_s();
  in.i = ivc->i; in.z0 = ivc->z0; in.zi = p->z_cur;
  in.U = ivc->U; in.u = ivc->u; in.cfU = ivc->cfU;
  { Fe c; memcpy(c.v, kzg_out[0], 32); in.eW = Fe::to_mont(c); memcpy(c.v, kzg_out[1], 32); in.eE = Fe::to_mont(c); }
  in.Wf = Wf.data() + 1; in.Ef = Ef.data();
  std::vector<Fq> cfZ, cfE;
  if (d->circ.full) {      // checks 5 and 6: the running CycleFold pair's witness and error vector
    const SecDev& S = ivc->sec;
    if (ivc->ck2 != v->ck2 || S.n_w != v->cf.b.n_wires || S.n_c != v->cf.b.n_constraints())
      return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_prove: the IVC proof's CycleFold key / shape is not the one the full decider was set up for");
    cfZ.resize(S.n_w); cfE.resize(S.n_c);
    std::lock_guard<std::mutex> g(ctx->mu);
    P_TRY(hipSetDevice(ctx->device));
    P_TRY(hipMemcpyAsync(cfZ.data(), S.Zrun, 32 * (size_t)S.n_w, hipMemcpyDeviceToHost, ctx->stream));
    P_TRY(hipMemcpyAsync(cfE.data(), S.E, 32 * (size_t)S.n_c, hipMemcpyDeviceToHost, ctx->stream));
    P_TRY(hipStreamSynchronize(ctx->stream));
    in.full.W = cfZ.data() + 1; in.full.E = cfE.data();
  }
  // (the main relation's rows are a third of the witness's time when evaluated one by one inside the circuit: their products come from the host's threads)
  std::vector<Fe> pre_abc[3];
  host_spmv3(v->circ->build->b, Wf, pre_abc);
  in.pre_az = pre_abc[0].data(); in.pre_bz = pre_abc[1].data(); in.pre_cz = pre_abc[2].data(); in.pre_z = Wf.data();
  std::vector<Fe> z;
  bool bad = false;
  try { z = d->circ.witness(v->circ->build->b, in, &bad); } catch (const std::exception& e) { return vz_fail(ctx, VIMZ_ERR_INVALID, e.what()); }
  if (bad) return vz_fail(ctx, VIMZ_ERR_UNSAT, "vimz_decider_prove: the IVC proof does not satisfy the decider's statement (hashes, relaxed relations, CycleFold commitments or KZG evaluations)");
  std::vector<Fe> abc[3];
  host_spmv3(d->circ.b, z, abc);
  for (int q = 0; q < 3; q++) abc[q].resize(K.n, Fe::zero());
  for (uint32_t i = 0; i <= K.n_pub; i++) abc[0][K.n_c + i] = z[i];
  const double t_wit = now_s();
  int rc;
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
  // (the wires' MSMs take unit scalars out first: a third of the full decider's wires are bits)
  auto g1_msm = [&](const vimz_bases* q, const uint32_t* sc, size_t n, G1Aff* outp, int split_ones) -> int {
    if (!n) { outp->x = Fq::zero(); outp->y = Fq::zero(); return VIMZ_OK; }
    const int r2 = vz_msm_device(ctx, q, 0, sc, n, 1, 0, pt, VIMZ_FORM_MONTGOMERY, split_ones);
    if (!r2) { memcpy(outp->x.v, pt, 32); memcpy(outp->y.v, pt + 4, 32); }
    return r2;
  };
  G1Aff sa, sb1, sl, sh;
  const int so = d->circ.full ? 1 : 0;
  static const bool dbg_t = getenv("VIMZ_DEBUG_TIMING") != nullptr;
  double tm0 = now_s();
  auto lap = [&](const char* what) { if (dbg_t) { const double t = now_s(); fprintf(stderr, "[decider] %s %.1f ms\n", what, 1e3 * (t - tm0)); tm0 = t; } };
  if ((rc = g1_msm(K.a_q, dz, K.m, &sa, so))) return rc; lap("msm a");
  if ((rc = g1_msm(K.b1_q, dz, K.m, &sb1, so))) return rc; lap("msm b1");
  if ((rc = g1_msm(K.l_q, dz + 8 * (size_t)(K.n_pub + 1), K.m - K.n_pub - 1, &sl, so))) return rc; lap("msm l");
  if ((rc = g1_msm(K.h_q, dv[0], K.n - 1, &sh, 0))) return rc; lap("msm h");
  // the G2 one, bit plane by bit plane (k_g2_planes): dv[1] (the quotient's inputs are spent) holds the canonical scalars of the query's wires
  G2P sb2 = G2P::identity();
  if (K.n_b2) {
    if ((size_t)K.n_b2 > (size_t)K.n) return vz_fail(ctx, VIMZ_ERR_INVALID, "decider: the G2 query is longer than the domain");
    P_TRY(hipMalloc((void**)&dpart, sizeof(G2P) * ((size_t)G2_PLANES * G2_PLANE_THREADS + G2_PLANES)));
    G2P* dsum = dpart + (size_t)G2_PLANES * G2_PLANE_THREADS;
    hipLaunchKernelGGL(k_g2_canon, dim3((K.n_b2 + 255) / 256), dim3(256), 0, s, (const uint32_t*)dz, (const uint32_t*)K.b2_idx, K.n_b2, dv[1]);
    hipLaunchKernelGGL(k_g2_planes, dim3(G2_PLANE_THREADS / 256, G2_PLANES), dim3(256), 0, s, (const G2PAff*)K.b2_q, (const uint32_t*)dv[1], K.n_b2, dpart);
    hipLaunchKernelGGL(k_g2_plane_tree, dim3(G2_PLANES), dim3(128), 0, s, (const G2P*)dpart, dsum);
    P_TRY(hipGetLastError());
    std::vector<G2P> planes(G2_PLANES);
    P_TRY(hipMemcpyAsync(planes.data(), dsum, sizeof(G2P) * G2_PLANES, hipMemcpyDeviceToHost, s));
    P_TRY(hipStreamSynchronize(s));
    for (int j = (int)G2_PLANES - 1; j >= 0; j--) { sb2 = dbl(sb2); add_full(sb2, planes[j]); }
  }
  lap("msm b2 (G2)");
  const double t_msm = now_s();
  // A = alpha + Σ z_i a_i + r·delta;  B = beta + Σ z_i b_i + s·delta;  C = Σ_priv z_i l_i + Σ h_j hq_j + s·A + r·B1 − r·s·delta;  r, s fresh from the OS (zero knowledge)
  Fe r, sr;
  struct WipeRs { Fe *a, *b; ~WipeRs() { wipe(a, sizeof(Fe)); wipe(b, sizeof(Fe)); } } wrs{&r, &sr};
  if (!fr_random(&r) || !fr_random(&sr)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_decider_prove: no randomness from the OS");
  G1 A = from_affine(K.alpha1); add_mixed(A, sa); { G1 t = host_mul_fr<Fq>(K.delta1, r); add_full(A, t); }
  G1 B1 = from_affine(K.beta1); add_mixed(B1, sb1); { G1 t = host_mul_fr<Fq>(K.delta1, sr); add_full(B1, t); }
  G2P B2 = from_affine(K.beta2); add_full(B2, sb2); { G2P t = host_mul_fr<Fq2>(K.delta2, sr); add_full(B2, t); }
  const G1Aff Aa = to_affine(A), B1a = to_affine(B1);
  G1 C = from_affine(sl); add_mixed(C, sh);
  { G1 t = host_mul_fr<Fq>(Aa, sr); add_full(C, t); }
  { G1 t = host_mul_fr<Fq>(B1a, r); add_full(C, t); }
  { G1 t = host_mul_fr<Fq>(K.delta1, Fe::mul(r, sr)); if (!t.is_identity()) t.Y = Fq::neg(t.Y); add_full(C, t); }
  const G1Aff Ca = to_affine(C); const G2PAff Ba = to_affine(B2);
  // the 25 words
  uint64_t* w = words_out;
  memset(w, 0, 800);
  put_fq(w + 0, ivc->UW.x); put_fq(w + 4, ivc->UW.y); put_fq(w + 8, ivc->UE.x); put_fq(w + 12, ivc->UE.y);
  put_fq(w + 16, ivc->uW.x); put_fq(w + 20, ivc->uW.y); put_fq(w + 24, cmT.x); put_fq(w + 28, cmT.y);
  w[32] = (uint64_t)r_low[0] | ((uint64_t)r_low[1] << 32); w[33] = (uint64_t)r_low[2] | ((uint64_t)r_low[3] << 32); w[34] = 1;      // r = 2^128 + low
  put_fq(w + 36, Aa.x); put_fq(w + 40, Aa.y);
  put_fq(w + 44, Ba.x.c1); put_fq(w + 48, Ba.x.c0); put_fq(w + 52, Ba.y.c1); put_fq(w + 56, Ba.y.c0);      // imaginary parts first, as the EVM's precompile takes G2
  put_fq(w + 60, Ca.x); put_fq(w + 64, Ca.y);
  { const Fe c = Fe::from_mont(cW); memcpy(w + 68, c.v, 32); } { const Fe c = Fe::from_mont(cE); memcpy(w + 72, c.v, 32); }
  memcpy(w + 76, kzg_out[0], 32); memcpy(w + 80, kzg_out[1], 32);
  memcpy(w + 84, kzg_out[0] + 4, 64); memcpy(w + 92, kzg_out[1] + 4, 64);
  for (uint32_t i = 0; i < K.n_pub; i++) { const Fe c = Fe::from_mont(z[1 + i]); memcpy(public_out + 4 * i, c.v, 32); }
  if (seconds) { seconds[0] = t_fold - t_all; seconds[1] = t_wit - t_fold; seconds[2] = t_ntt - t_wit; seconds[3] = t_msm - t_ntt; seconds[4] = now_s() - t_all; seconds[5] = 0; }
  return VIMZ_OK;
}

// Decider::verify (reached from `verify_final_proof`, vimz/src/sonobe_backend/decider.rs:31-50, mod.rs:80): what the contract generated for this key
// does with (steps, z_0, z_i, 25 words) — contracts/ContrastVerifier.sol:685-783 — on the host: steps >= 2, the two folded commitments, their
// KZG openings, the Groth16 proof for the public inputs the words imply.  *result = 0 accepted; else a bit set: 1 fewer than two steps, 2 KZG
// opening of cmW, 4 of cmE, 8 Groth16, 16 a word pair is not a curve point (the contract's precompile would revert).  Host only.
int vimz_decider_verify(const vimz_decider* d, uint64_t steps, const uint64_t* z0, const uint64_t* zi, const uint64_t words[100], uint32_t* result) {
  if (!d || !z0 || !zi || !words || !result) return VIMZ_ERR_INVALID;
  if (aff_is_identity(d->kzg_vk)) return vz_fail(d->ctx, VIMZ_ERR_INVALID, "vimz_decider_verify: the decider was set up without the SRS's verifying key");
  *result = verify_words(key_of(d), steps, z0, zi, words);
  return VIMZ_OK;
}
// the same with the key as bytes (vimz_decider_vk's layout; also how a contract's constants are fed in: tests/test_decider_verify_host.py runs the
// reference's six committed proofs through this with the constants of contracts/*Verifier.sol).  No context, no GPU.
int vimz_decider_verify_key(const uint64_t* key_words, size_t n_key_words, uint64_t steps, const uint64_t* z0, const uint64_t* zi, uint32_t len_z,
                            const uint64_t words[100], uint32_t* result) {
  if (!key_words || !z0 || !zi || !words || !result) return VIMZ_ERR_INVALID;
  VerifierKey K;
  if (!parse_key(key_words, n_key_words, &K) || K.len_z != len_z) return VIMZ_ERR_INVALID;
  *result = verify_words(K, steps, z0, zi, words);
  return VIMZ_OK;
}

}  // extern "C"
