// The SpMV / cross-term seam of SURVEY.md §8b row 2 for a CALLER-SUPPLIED shape: what a patched nova-snark 0.23.0 binds in place of
//   R1CSShape::multiply_vec(&self, z) -> (Az, Bz, Cz)                      (sparse_matrix_vec_product over COO triplets, rayon)
//   R1CSShape::commit_T(ck, U1, W1, U2, W2) -> (T, comm_T)                 (two multiply_vec, the cross term, CE::commit(ck, T))
// reached from NIFS::prove and is_sat* inside RecursiveSNARK::{prove_step, verify} (vimz/src/nova_snark_backend/folding.rs:35-41,53-55).
// The matrices arrive exactly as nova-snark holds them — three lists of (row, col, value) triplets in any order — and are turned
// into the resident form the fold kernels use: CSR with a coefficient dictionary (8 B per non-zero; R1CS coefficients repeat
// massively: ±1, powers of two, a few hundred Poseidon constants), plus the list of long rows for the wave-per-row kernel.
// Independent of the step-circuit builder and of any prover object: z, Az, Bz, Cz, T are vimz_vec handles, ck a vimz_bases.
#include <hip/hip_runtime.h>
#include <array>
#include <cstring>
#include <memory>
#include <new>
#include <unordered_map>
#include <vector>

#include "internal.hpp"
#include "r1cs_ops.hpp"

using namespace vz;

struct vimz_r1cs {
  int field = 0;
  size_t nrows = 0, ncols = 0, nnz[3] = {0, 0, 0};
  CsrDev M[3] = {};
  uint32_t* dict = nullptr; size_t ndict = 0;
  uint32_t* long_items = nullptr; uint32_t n_long = 0, n_med = 0;
  uint32_t* scratch = nullptr;          // 6 x nrows elements: (A,B,C)·z1 and (A,B,C)·z2 of commit_T
  std::vector<void*> owned;
};

#define R_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) return vz_fail(ctx, VIMZ_ERR_HIP, #x, _e); } while (0)

namespace {

struct KeyHash {
  size_t operator()(const std::array<uint64_t, 4>& k) const { return (size_t)(k[0] * 0x9e3779b97f4a7c15ull ^ (k[1] + 0x7f4a7c15ull) * 0xff51afd7ed558ccdull ^ k[2] * 0xc4ceb9fe1a85ec53ull ^ k[3]); }
};

template <class F>
int upload_shape(vimz_ctx* ctx, vimz_r1cs* S, const uint32_t* const rows[3], const uint32_t* const cols[3], const uint64_t* const vals[3], const size_t nnz[3], int form) {
  const size_t nr = S->nrows;
  std::unordered_map<std::array<uint64_t, 4>, uint32_t, KeyHash> index;
  std::vector<F> dict;
  std::vector<uint32_t> items;
  for (int m = 0; m < 3; m++) {
    std::vector<uint32_t> row_ptr(nr + 1, 0), col(nnz[m]), coef(nnz[m]);
    for (size_t k = 0; k < nnz[m]; k++) {
      if (rows[m][k] >= nr || cols[m][k] >= S->ncols) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_r1cs_upload: triplet outside the shape");
      row_ptr[rows[m][k] + 1]++;
    }
    for (size_t r = 0; r < nr; r++) row_ptr[r + 1] += row_ptr[r];
    std::vector<uint32_t> cur(row_ptr.begin(), row_ptr.end() - 1);
    for (size_t k = 0; k < nnz[m]; k++) {          // counting sort by row; triplets of a row keep their order
      std::array<uint64_t, 4> key; memcpy(key.data(), vals[m] + 4 * k, 32);
      auto it = index.find(key);
      uint32_t id;
      if (it == index.end()) {
        F v; memcpy(v.v, key.data(), 32);
        if (!v.is_reduced()) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_r1cs_upload: coefficient not below the modulus");
        id = (uint32_t)dict.size();
        dict.push_back(form == VIMZ_FORM_CANONICAL ? F::to_mont(v) : v);
        index.emplace(key, id);
      } else id = it->second;
      const uint32_t pos = cur[rows[m][k]]++;
      col[pos] = cols[m][k]; coef[pos] = id;
    }
    for (size_t r = 0; r < nr; r++) if (row_ptr[r + 1] - row_ptr[r] > SPMV_LONG) items.push_back(((uint32_t)m << 30) | (uint32_t)r);
    auto up = [&](const std::vector<uint32_t>& v, const uint32_t** dst) -> hipError_t {
      uint32_t* d = nullptr;
      hipError_t e = hipMalloc((void**)&d, 4 * std::max<size_t>(v.size(), 1));
      if (e != hipSuccess) return e;
      S->owned.push_back(d);
      *dst = d;
      return v.empty() ? hipSuccess : hipMemcpy(d, v.data(), 4 * v.size(), hipMemcpyHostToDevice);
    };
    R_TRY(up(row_ptr, &S->M[m].row_ptr)); R_TRY(up(col, &S->M[m].col)); R_TRY(up(coef, &S->M[m].coef));
    S->nnz[m] = nnz[m];
  }
  // the long items, those of <= SPMV_MED terms first (16 lanes each), the rest a wave each
  {
    std::vector<std::vector<uint32_t>> cnt(3, std::vector<uint32_t>(nr, 0));
    for (int m = 0; m < 3; m++) for (size_t k = 0; k < nnz[m]; k++) cnt[m][rows[m][k]]++;
    S->n_med = spmv_sort_items(items, [&](uint32_t it) { return cnt[it >> 30][it & 0x3fffffffu]; });
  }
  S->n_long = (uint32_t)items.size();
  if (!items.empty()) {
    R_TRY(hipMalloc((void**)&S->long_items, 4 * items.size())); S->owned.push_back(S->long_items);
    R_TRY(hipMemcpy(S->long_items, items.data(), 4 * items.size(), hipMemcpyHostToDevice));
  }
  S->ndict = dict.size();
  const std::vector<F> dd = dict_for_device(dict);      // (both forms of every coefficient: r1cs_ops.hpp)
  R_TRY(hipMalloc((void**)&S->dict, 32 * std::max<size_t>(dd.size(), 1))); S->owned.push_back(S->dict);
  if (!dd.empty()) R_TRY(hipMemcpy(S->dict, dd.data(), 32 * dd.size(), hipMemcpyHostToDevice));
  R_TRY(hipMalloc((void**)&S->scratch, 32 * 6 * std::max<size_t>(nr, 1))); S->owned.push_back(S->scratch);
  return VIMZ_OK;
}

template <class F>
void spmv3(const vimz_r1cs* S, hipStream_t s, const uint32_t* z, uint32_t* az, uint32_t* bz, uint32_t* cz) {
  hipLaunchKernelGGL(k_spmv3<F>, dim3(stream_grid(3 * S->nrows)), dim3(256), 0, s, S->M[0], S->M[1], S->M[2], (const uint32_t*)S->dict, S->nrows, z, az, bz, cz);
  if (S->n_long)
    hipLaunchKernelGGL(k_spmv_long<F>, dim3(spmv_long_blocks(S->n_long, S->n_med)), dim3(256), 0, s, S->M[0], S->M[1], S->M[2], (const uint32_t*)S->dict,
                       (const uint32_t*)S->long_items, S->n_long, S->n_med, z, az, bz, cz);
}

template <class Fn>
int field_dispatch(int field, Fn fn) {
  switch (field) {
    case VIMZ_FIELD_BN254_FR: return fn(Fp<BnFr>());
    case VIMZ_FIELD_BN254_FQ: return fn(Fp<BnFq>());
    case VIMZ_FIELD_PALLAS_FP: return fn(Fp<PallasFp>());
    case VIMZ_FIELD_VESTA_FQ: return fn(Fp<VestaFq>());
  }
  return VIMZ_ERR_INVALID;
}
int curve_scalar_field(int curve) {
  switch (curve) {
    case VIMZ_CURVE_BN254_G1: return VIMZ_FIELD_BN254_FR;
    case VIMZ_CURVE_GRUMPKIN: return VIMZ_FIELD_BN254_FQ;
    case VIMZ_CURVE_PALLAS: return VIMZ_FIELD_VESTA_FQ;
    default: return VIMZ_FIELD_PALLAS_FP;
  }
}

}  // namespace

extern "C" {

void vimz_r1cs_free(vimz_ctx* ctx, vimz_r1cs* S) {
  if (!S) return;
  if (ctx) {
    std::lock_guard<std::mutex> g(ctx->mu);
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (void* d : S->owned) hipFree(d);
  }
  delete S;
}

int vimz_r1cs_upload(vimz_ctx* ctx, int field, size_t nrows, size_t ncols, const vimz_coo* A, const vimz_coo* B, const vimz_coo* C, int form, vimz_r1cs** out) {
  if (!ctx || !A || !B || !C || !out || field < 0 || field > 3 || nrows == 0 || ncols == 0 || nrows >= (1u << 30) || ncols >= (1ull << 32))
    return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_r1cs_upload: bad argument");
  const vimz_coo* Ms[3] = {A, B, C};
  const uint32_t* rows[3]; const uint32_t* cols[3]; const uint64_t* vals[3]; size_t nnz[3];
  for (int m = 0; m < 3; m++) {
    rows[m] = Ms[m]->row; cols[m] = Ms[m]->col; vals[m] = Ms[m]->val; nnz[m] = Ms[m]->nnz;
    if (nnz[m] && (!rows[m] || !cols[m] || !vals[m])) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_r1cs_upload: NULL triplet array");
    if (nnz[m] >= (1ull << 32)) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_r1cs_upload: too many non-zeros");
  }
  std::unique_ptr<vimz_r1cs> S(new (std::nothrow) vimz_r1cs());
  if (!S) return vz_fail(ctx, VIMZ_ERR_INVALID, "out of host memory");
  S->field = field; S->nrows = nrows; S->ncols = ncols;
  std::unique_lock<std::mutex> g(ctx->mu);
  R_TRY(hipSetDevice(ctx->device));
  int rc;
  try {
    rc = field_dispatch(field, [&](auto f) { typedef decltype(f) F; return upload_shape<F>(ctx, S.get(), rows, cols, vals, nnz, form); });
  } catch (const std::exception& e) { rc = vz_fail(ctx, VIMZ_ERR_INVALID, e.what()); }
  if (rc) { for (void* d : S->owned) hipFree(d); return rc; }
  *out = S.release();
  return VIMZ_OK;
}

int vimz_r1cs_info(const vimz_r1cs* S, uint64_t info[8]) {
  if (!S || !info) return VIMZ_ERR_INVALID;
  info[0] = S->nrows; info[1] = S->ncols; info[2] = S->nnz[0]; info[3] = S->nnz[1]; info[4] = S->nnz[2]; info[5] = S->ndict; info[6] = S->n_long; info[7] = (uint64_t)S->field;
  return VIMZ_OK;
}

int vimz_spmv3(vimz_ctx* ctx, const vimz_r1cs* S, const vimz_vec* z, vimz_vec* az, vimz_vec* bz, vimz_vec* cz) {
  if (!ctx || !S || !z || !az || !bz || !cz) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_spmv3: bad argument");
  if (z->field != S->field || az->field != S->field || bz->field != S->field || cz->field != S->field) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_spmv3: field mismatch");
  if (z->n < S->ncols || az->n < S->nrows || bz->n < S->nrows || cz->n < S->nrows) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_spmv3: vector shorter than the shape");
  std::lock_guard<std::mutex> g(ctx->mu);
  R_TRY(hipSetDevice(ctx->device));
  field_dispatch(S->field, [&](auto f) { typedef decltype(f) F; spmv3<F>(S, ctx->stream, z->d, az->d, bz->d, cz->d); return VIMZ_OK; });
  R_TRY(hipGetLastError());
  R_TRY(hipStreamSynchronize(ctx->stream));
  return VIMZ_OK;
}

int vimz_commit_T(vimz_ctx* ctx, const vimz_r1cs* S, const vimz_bases* ck, const vimz_vec* z1, const uint64_t u1[4], const vimz_vec* z2, const uint64_t u2[4], int form,
                  vimz_vec* T_out, uint64_t comm_T[8], int out_form) {
  if (!ctx || !S || !ck || !z1 || !z2 || !u1 || !u2 || !T_out || !comm_T) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_commit_T: bad argument");
  if (curve_scalar_field(ck->curve) != S->field || z1->field != S->field || z2->field != S->field || T_out->field != S->field)
    return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_commit_T: the key's scalar field, the shape and the vectors must agree");
  if (z1->n < S->ncols || z2->n < S->ncols || T_out->n < S->nrows || ck->n < S->nrows) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_commit_T: vector or key shorter than the shape");
  std::lock_guard<std::mutex> g(ctx->mu);
  R_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t nr = S->nrows;
  uint32_t* w = S->scratch;
  int rc = field_dispatch(S->field, [&](auto f) {
    typedef decltype(f) F;
    F a, b; memcpy(a.v, u1, 32); memcpy(b.v, u2, 32);
    if (!a.is_reduced() || !b.is_reduced()) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_commit_T: u not below the modulus");
    if (form == VIMZ_FORM_CANONICAL) { a = F::to_mont(a); b = F::to_mont(b); }
    spmv3<F>(S, s, z1->d, w, w + 8 * nr, w + 16 * nr);
    spmv3<F>(S, s, z2->d, w + 24 * nr, w + 32 * nr, w + 40 * nr);
    hipLaunchKernelGGL(k_cross_term<F>, dim3(stream_grid(nr)), dim3(256), 0, s, nr, (const uint32_t*)w, (const uint32_t*)(w + 8 * nr), (const uint32_t*)(w + 16 * nr), a,
                       (const uint32_t*)(w + 24 * nr), (const uint32_t*)(w + 32 * nr), (const uint32_t*)(w + 40 * nr), b, T_out->d);
    return VIMZ_OK;
  });
  if (rc) return rc;
  R_TRY(hipGetLastError());
  return vz_msm_device(ctx, ck, 0, T_out->d, nr, 1, 0, comm_T, out_form);
}

// is_sat_relaxed of a resident assignment: the number of rows with (A·z)∘(B·z) != u·(C·z) + E and the first of them.
int vimz_r1cs_check_relaxed(vimz_ctx* ctx, const vimz_r1cs* S, const vimz_vec* z, const uint64_t u[4], int form, const vimz_vec* E, uint64_t* bad_rows, uint64_t* first_bad) {
  if (!ctx || !S || !z || !u || !bad_rows) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_r1cs_check_relaxed: bad argument");
  if (z->field != S->field || z->n < S->ncols || (E && (E->field != S->field || E->n < S->nrows)))
    return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_r1cs_check_relaxed: field or length of a vector does not fit the shape");
  std::lock_guard<std::mutex> g(ctx->mu);
  R_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t nr = S->nrows;
  uint32_t* w = S->scratch;
  uint32_t* bad_d = w + 8 * 3 * nr;              // (the second half of the scratch is free here)
  const uint32_t init[2] = {0, 0xffffffffu};
  R_TRY(hipMemcpyAsync(bad_d, init, 8, hipMemcpyHostToDevice, s));
  int rc = field_dispatch(S->field, [&](auto f) {
    typedef decltype(f) F;
    F a; memcpy(a.v, u, 32);
    if (!a.is_reduced()) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_r1cs_check_relaxed: u not below the modulus");
    if (form == VIMZ_FORM_CANONICAL) a = F::to_mont(a);
    spmv3<F>(S, s, z->d, w, w + 8 * nr, w + 16 * nr);
    hipLaunchKernelGGL(k_check_relaxed<F>, dim3(stream_grid(nr)), dim3(256), 0, s, nr, (const uint32_t*)w, (const uint32_t*)(w + 8 * nr), (const uint32_t*)(w + 16 * nr), a,
                       E ? (const uint32_t*)E->d : (const uint32_t*)nullptr, bad_d);
    return VIMZ_OK;
  });
  if (rc) return rc;
  R_TRY(hipGetLastError());
  uint32_t bad[2];
  R_TRY(hipMemcpyAsync(bad, bad_d, 8, hipMemcpyDeviceToHost, s));
  R_TRY(hipStreamSynchronize(s));
  *bad_rows = bad[0];
  if (first_bad) *first_bad = bad[0] ? bad[1] : ~0ull;
  return VIMZ_OK;
}

// x1 <- x1 + r * x2 over the first n elements: RelaxedR1CSWitness::fold (W and E) / the fold of any resident vector.
int vimz_vec_axpy(vimz_ctx* ctx, vimz_vec* x1, const uint64_t r[4], int form, const vimz_vec* x2, size_t n) {
  if (!ctx || !x1 || !x2 || !r) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_vec_axpy: bad argument");
  if (x1->field != x2->field || n > x1->n || n > x2->n) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_vec_axpy: fields differ or a vector is shorter than n");
  if (x1 == x2) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_vec_axpy: x1 and x2 must be different vectors");
  if (!n) return VIMZ_OK;
  std::lock_guard<std::mutex> g(ctx->mu);
  R_TRY(hipSetDevice(ctx->device));
  int rc = field_dispatch(x1->field, [&](auto f) {
    typedef decltype(f) F;
    F a; memcpy(a.v, r, 32);
    if (!a.is_reduced()) return vz_fail(ctx, VIMZ_ERR_INVALID, "vimz_vec_axpy: r not below the modulus");
    if (form == VIMZ_FORM_CANONICAL) a = F::to_mont(a);
    hipLaunchKernelGGL(k_axpy_inplace<F>, dim3(stream_grid(n)), dim3(256), 0, ctx->stream, n, x1->d, a, (const uint32_t*)x2->d);
    return VIMZ_OK;
  });
  if (rc) return rc;
  R_TRY(hipGetLastError());
  return VIMZ_OK;
}

}  // extern "C"
