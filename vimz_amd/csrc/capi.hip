// C ABI (include/vimz_hip.h) over the gfx950 kernels.  One translation unit: the kernels are templates
// over the four curves / fields and are instantiated from here.
#include <hip/hip_runtime.h>
#include <mutex>
#include <new>
#include <string>
#include <cstring>
#include <cstdio>

#include "internal.hpp"
#define HIP_TRY(c, x) do { hipError_t _e = (x); if (_e != hipSuccess) return vz::vz_fail(c, VIMZ_ERR_HIP, #x, _e); } while (0)
#include "vecops_api.hpp"
#include "ckgen.hpp"

using namespace vz;

static thread_local std::string g_err_noctx;

namespace vz {
int vz_fail(vimz_ctx* c, int code, const char* what, hipError_t e) {
  std::string m = what;
  if (e != hipSuccess) { m += ": "; m += hipGetErrorString(e); }
  if (c) c->err = m; else g_err_noctx = m;
  return code;
}
}  // namespace vz
static int fail(vimz_ctx* c, int code, const char* what, hipError_t e = hipSuccess) { return vz::vz_fail(c, code, what, e); }

namespace vz {
int vz_ensure_scratch(vimz_ctx* c, size_t bytes) {
  if (bytes <= c->scratch_bytes) return VIMZ_OK;
  if (c->scratch) hipFree(c->scratch);
  c->scratch = nullptr; c->scratch_bytes = 0;
  HIP_TRY(c, hipMalloc(&c->scratch, bytes));
  c->scratch_bytes = bytes;
  return VIMZ_OK;
}
}  // namespace vz
static int ensure_scratch(vimz_ctx* c, size_t bytes) { return vz::vz_ensure_scratch(c, bytes); }

template <class Fn>
static int field_dispatch(int field, Fn fn) {
  switch (field) {
    case VIMZ_FIELD_BN254_FR: return fn(Fp<BnFr>());
    case VIMZ_FIELD_BN254_FQ: return fn(Fp<BnFq>());
    case VIMZ_FIELD_PALLAS_FP: return fn(Fp<PallasFp>());
    case VIMZ_FIELD_VESTA_FQ: return fn(Fp<VestaFq>());
  }
  return VIMZ_ERR_INVALID;
}
template <class Fn>
static int curve_dispatch(int curve, Fn fn) {
  switch (curve) {
    case VIMZ_CURVE_BN254_G1: return fn(BnG1());
    case VIMZ_CURVE_GRUMPKIN: return fn(Grumpkin());
    case VIMZ_CURVE_PALLAS: return fn(Pallas());
    case VIMZ_CURVE_VESTA: return fn(Vesta());
  }
  return VIMZ_ERR_INVALID;
}
static int curve_base_field(int curve) {
  switch (curve) {
    case VIMZ_CURVE_BN254_G1: return VIMZ_FIELD_BN254_FQ;
    case VIMZ_CURVE_GRUMPKIN: return VIMZ_FIELD_BN254_FR;
    case VIMZ_CURVE_PALLAS: return VIMZ_FIELD_PALLAS_FP;
    default: return VIMZ_FIELD_VESTA_FQ;
  }
}
static int curve_scalar_field(int curve) {
  switch (curve) {
    case VIMZ_CURVE_BN254_G1: return VIMZ_FIELD_BN254_FR;
    case VIMZ_CURVE_GRUMPKIN: return VIMZ_FIELD_BN254_FQ;
    case VIMZ_CURVE_PALLAS: return VIMZ_FIELD_VESTA_FQ;
    default: return VIMZ_FIELD_PALLAS_FP;
  }
}

namespace vz {
int vz_msm_device(vimz_ctx* c, const vimz_bases* bases, size_t base_offset, const uint32_t* d_scalars, size_t n,
                      int scalars_mont, int window_bits, uint64_t out_xy[8], int out_form, int split_ones) {
  return curve_dispatch(bases->curve, [&](auto cv) {
    typedef decltype(cv) C;
    typedef typename C::Base F;
    Affine<F> r;
    BaseTables tb = bases->tb(base_offset);
    bool have_tb = bases->tables != nullptr;
    if (n <= MSM_SMALL_MAX && window_bits <= 0 && !split_ones) {      // a fixed slice with small-MSM tables (vimz_bases_precompute(7), or an IVC's slices)
      vimz_bases* bm = const_cast<vimz_bases*>(bases);
      std::lock_guard<std::mutex> g(bm->small_mu);
      for (auto& t : bm->small)
        if (t.offset <= base_offset && base_offset + n <= t.offset + t.n) { tb = BaseTables{t.rows, t.n, base_offset - t.offset, t.c, t.K, 0, t.mult}; have_tb = true; break; }
    }
    hipError_t e = msm_run<C>(c->stream, c->msm_ws, bases->d + (size_t)AFFINE_WORDS * base_offset, d_scalars, n, scalars_mont, window_bits, &r,
                              &c->last_msm, c->profiling ? c->ev : nullptr, split_ones, have_tb ? &tb : nullptr);
    if (e != hipSuccess) return vz_fail(c, VIMZ_ERR_HIP, "msm", e);
    if (c->profiling && n) {
      for (int i = 0; i < 6; i++) c->msm_tot_ms[i] += c->last_msm.ms[i];
      c->msm_tot_calls++; c->msm_tot_points += n; c->msm_tot_entries += c->last_msm.entries;
    }
    if (out_form == VIMZ_FORM_CANONICAL) { r.x = F::from_mont(r.x); r.y = F::from_mont(r.y); }
    memcpy(out_xy, r.x.v, 32); memcpy(out_xy + 4, r.y.v, 32);
    return VIMZ_OK;
  });
}

// core of vimz_kzg_open on a device-resident vector of Montgomery elements (caller holds c->mu and has set the device)
int vz_kzg_open_device(vimz_ctx* c, const vimz_bases* srs, size_t base_offset, int field, const uint32_t* d_vec, size_t n, const uint64_t z[4], int form,
                       uint64_t eval_out[4], uint64_t proof_xy[8]) {
  if (n == 0 || base_offset + n > srs->n) return vz_fail(c, VIMZ_ERR_INVALID, "kzg open: the SRS is shorter than the vector");
  std::vector<uint32_t> host(8 * n);
  hipError_t e;
  if ((e = hipMemcpyAsync(host.data(), d_vec, 32 * n, hipMemcpyDeviceToHost, c->stream)) != hipSuccess || (e = hipStreamSynchronize(c->stream)) != hipSuccess)
    return vz_fail(c, VIMZ_ERR_HIP, "kzg open: download", e);
  int rc = field_dispatch(field, [&](auto f) {
    typedef decltype(f) F;
    F zz; memcpy(zz.v, z, 32);
    if (!zz.is_reduced()) return vz_fail(c, VIMZ_ERR_INVALID, "kzg open: z not below the modulus");
    if (form == VIMZ_FORM_CANONICAL) zz = F::to_mont(zz);
    F* a = reinterpret_cast<F*>(host.data());       // Montgomery, as resident
    F carry = F::zero();                              // q_i, walking down; the last carry is p(z)
    for (size_t i = n; i-- > 0;) { const F cur = F::add(a[i], F::mul(zz, carry)); a[i] = carry; carry = cur; }      // a[i] <- q_i  (q_{n-1} = 0)
    const F ev = form == VIMZ_FORM_CANONICAL ? F::from_mont(carry) : carry;
    memcpy(eval_out, ev.v, 32);
    return VIMZ_OK;
  });
  if (rc) return rc;
  if (n == 1) { memset(proof_xy, 0, 64); return VIMZ_OK; }      // a constant: the quotient is zero, its commitment the identity
  rc = vz_ensure_scratch(c, 32 * (n - 1)); if (rc) return rc;
  if ((e = hipMemcpyAsync(c->scratch, host.data(), 32 * (n - 1), hipMemcpyHostToDevice, c->stream)) != hipSuccess) return vz_fail(c, VIMZ_ERR_HIP, "kzg open: upload", e);
  return vz_msm_device(c, srs, base_offset, (const uint32_t*)c->scratch, n - 1, 1, 0, proof_xy, form);
}
int vz_small_tables(vimz_ctx* c, vimz_bases* b, size_t offset, size_t n, bool with_mult, BaseTables* out) {
  *out = BaseTables{nullptr, 0, 0, 0, 0};
  if (!n || n > MSM_SMALL_MAX || offset + n > b->n) return VIMZ_OK;
  std::lock_guard<std::mutex> g(b->small_mu);
  vimz_small_tables* t = nullptr;
  for (auto& e : b->small) if (e.offset == offset && e.n == n) t = &e;
  return curve_dispatch(b->curve, [&](auto cv) {
    typedef decltype(cv) C;
    const int K = (C::Scalar::Params::BITS + SMALL_C) / SMALL_C;
    hipError_t e = hipSuccess;
    if (!t) {
      vimz_small_tables nt; nt.offset = offset; nt.n = n; nt.c = SMALL_C; nt.K = K;
      if ((e = hipMalloc((void**)&nt.rows, 4 * (size_t)AFFINE_WORDS * n * K)) != hipSuccess) return vz_fail(c, VIMZ_ERR_HIP, "small-MSM window tables", e);
      if ((e = build_tables<C>(c->stream, b->d + (size_t)AFFINE_WORDS * offset, n, SMALL_C, K, nt.rows)) != hipSuccess) { hipFree(nt.rows); return vz_fail(c, VIMZ_ERR_HIP, "small-MSM window tables", e); }
      b->small.push_back(nt);
      t = &b->small.back();
    }
    if (with_mult && !t->mult) {
      const size_t nm = (size_t)1 << (SMALL_C - 1);
      if ((e = hipMalloc((void**)&t->mult, 4 * (size_t)AFFINE_WORDS * n * K * nm)) != hipSuccess) return vz_fail(c, VIMZ_ERR_HIP, "small-MSM multiples tables (1.5 GB per 7.7 k points)", e);
      if ((e = build_multiples<C>(c->stream, t->rows, n, SMALL_C, K, t->mult)) != hipSuccess) { hipFree(t->mult); t->mult = nullptr; return vz_fail(c, VIMZ_ERR_HIP, "small-MSM multiples tables", e); }
    }
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return vz_fail(c, VIMZ_ERR_HIP, "small-MSM tables", e);
    out->d = t->rows; out->n_total = n; out->offset = 0; out->c = SMALL_C; out->K = K; out->mult = with_mult ? t->mult : nullptr;
    return VIMZ_OK;
  });
}
}  // namespace vz
static int msm_device(vimz_ctx* c, const vimz_bases* bases, size_t base_offset, const uint32_t* d_scalars, size_t n,
                      int scalars_mont, int window_bits, uint64_t out_xy[8], int out_form) {
  return vz::vz_msm_device(c, bases, base_offset, d_scalars, n, scalars_mont, window_bits, out_xy, out_form);
}

// Input pipeline: packing of pixels into field elements on the device — ten pixels per element, pixel k in
// bits [24k, 24k + 24) with R in the low byte (a grey value alone in the low byte), rows padded with zero pixels: what
// pyvimz/pyvimz/img/ops.py:4-33 (compress_by_rows) and :36-70 (compress_by_blocks, 40 x 40 blocks -> 160 elements) produce as hex
// strings.  One thread per packed element; the output is canonical little-endian limbs, i.e. the 30 pixel bytes followed by two zeros.
static __global__ void __launch_bounds__(256) k_pack_pixels(const uint8_t* __restrict__ px, uint32_t h, uint32_t w, uint32_t ch, uint32_t block,
                                                            uint32_t n_out, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  uint32_t row, col0, run;       // first pixel of this element and how many pixels of the row / block row are left
  if (!block) {
    const uint32_t per = (w + 9) / 10;
    row = i / per; col0 = (i % per) * 10; run = w - col0;
  } else {
    const uint32_t bw = (w + block - 1) / block, per_row = (block + 9) / 10, per_blk = block * per_row;
    const uint32_t b = i / per_blk, e = i % per_blk, br = b / bw, bc = b % bw, r = e / per_row, c = e % per_row;
    row = br * block + r; col0 = bc * block + c * 10;
    const uint32_t end = min(w, (bc + 1) * block);
    run = row < h && col0 < end ? end - col0 : 0;
  }
  uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint32_t np = run < 10 ? run : 10;
  for (uint32_t k = 0; k < np; k++)
    for (uint32_t c = 0; c < 3; c++) {
      const uint32_t v = c < ch ? px[((size_t)row * w + col0 + k) * ch + c] : 0u;
      const uint32_t byte = 3 * k + c;
      words[byte >> 2] |= v << (8 * (byte & 3));
    }
  uint4* o = reinterpret_cast<uint4*>(out + 8 * (size_t)i);
  o[0] = make_uint4(words[0], words[1], words[2], words[3]);
  o[1] = make_uint4(words[4], words[5], words[6], words[7]);
}

extern "C" {

const char* vimz_version(void) { return "vimz-hip 0.1 (gfx950)"; }

int vimz_ctx_create(int device, vimz_ctx** out) {
  if (!out) return fail(nullptr, VIMZ_ERR_INVALID, "out is NULL");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) return fail(nullptr, VIMZ_ERR_NO_DEVICE, "no HIP device visible", e);
  if (device < 0 || device >= count) return fail(nullptr, VIMZ_ERR_INVALID, "device index out of range");
  vimz_ctx* c = new (std::nothrow) vimz_ctx();
  if (!c) return fail(nullptr, VIMZ_ERR_INVALID, "out of host memory");
  c->device = device;
  // the context's stream carries the latency-critical sequential chain of a fold: give it the highest priority so the
  // batch producer (a second stream inside the prover) cannot delay it
  int prio_lo = 0, prio_hi = 0;
  hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi)) != hipSuccess ||
      (e = hipEventCreate(&c->t0)) != hipSuccess || (e = hipEventCreate(&c->t1)) != hipSuccess) {
    delete c; return fail(nullptr, VIMZ_ERR_HIP, "context setup", e);
  }
  for (int i = 0; i < 7; i++) if ((e = hipEventCreate(&c->ev[i])) != hipSuccess) { delete c; return fail(nullptr, VIMZ_ERR_HIP, "event", e); }
  *out = c;
  return VIMZ_OK;
}

void vimz_ctx_destroy(vimz_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  c->msm_ws.release();
  if (c->scratch) hipFree(c->scratch);
  for (int i = 0; i < 7; i++) if (c->ev[i]) hipEventDestroy(c->ev[i]);
  hipEventDestroy(c->t0); hipEventDestroy(c->t1);
  for (auto& ps : c->spare_streams) hipStreamDestroy(ps.second);
  hipStreamDestroy(c->stream);
  delete c;
}

const char* vimz_last_error(const vimz_ctx* c) { return c ? c->err.c_str() : g_err_noctx.c_str(); }

int vimz_device_info(vimz_ctx* c, char* name, size_t name_len, int* cus, uint64_t* hbm_bytes) {
  if (!c) return VIMZ_ERR_INVALID;
  hipDeviceProp_t p;
  HIP_TRY(c, hipGetDeviceProperties(&p, c->device));
  if (name && name_len) { snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName); }
  if (cus) *cus = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
  return VIMZ_OK;
}

size_t vimz_pack_count(size_t height, size_t width, int block) {
  if (!block) return height * ((width + 9) / 10);
  const size_t b = (size_t)block;
  return ((height + b - 1) / b) * ((width + b - 1) / b) * b * ((b + 9) / 10);
}
int vimz_pack_pixels(vimz_ctx* c, const uint8_t* pixels, size_t height, size_t width, int channels, int block, uint64_t* out) {
  // (each dimension is bounded before anything is multiplied: the kernel indexes with 32-bit products)
  if (!c || !pixels || !out || !height || !width || (channels != 1 && channels != 3) || block < 0 || block > 4096 || height > (1ull << 20) || width > (1ull << 20) ||
      height * width > (1ull << 30))
    return fail(c, VIMZ_ERR_INVALID, "vimz_pack_pixels: bad argument (at most 2^20 per dimension, 2^30 pixels, blocks of at most 4096)");
  const size_t n = vimz_pack_count(height, width, block), in_bytes = height * width * (size_t)channels;
  if (n >= (1ull << 31)) return fail(c, VIMZ_ERR_INVALID, "vimz_pack_pixels: output too large");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = ensure_scratch(c, 32 * n + in_bytes + 64);
  if (rc) return rc;
  uint8_t* d_in = (uint8_t*)c->scratch + 32 * n;
  HIP_TRY(c, hipMemcpyAsync(d_in, pixels, in_bytes, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_pack_pixels, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t*)d_in, (uint32_t)height, (uint32_t)width,
                     (uint32_t)channels, (uint32_t)block, (uint32_t)n, (uint32_t*)c->scratch);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(out, c->scratch, 32 * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VIMZ_OK;
}

int vimz_sync(vimz_ctx* c) {
  if (!c) return VIMZ_ERR_INVALID;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VIMZ_OK;
}

// A marker in a profiler's kernel trace: an empty kernel named k_trace_marker with `id` workgroups on the context's stream, waited for.
// bench.py brackets its timed region with markers 1 and 2 so that tools/trace_busy.py measures the fold itself, not the set-up,
// the compression and the extras around it.
namespace vz { __global__ void k_trace_marker() {} }
int vimz_trace_marker(vimz_ctx* c, int id) {
  if (!c || id < 1 || id > 1024) return VIMZ_ERR_INVALID;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(vz::k_trace_marker, dim3((unsigned)id), dim3(64), 0, c->stream);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VIMZ_OK;
}

int vimz_timer_start(vimz_ctx* c) {
  if (!c) return VIMZ_ERR_INVALID;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipEventRecord(c->t0, c->stream));
  return VIMZ_OK;
}
int vimz_timer_stop(vimz_ctx* c, float* ms) {
  if (!c || !ms) return VIMZ_ERR_INVALID;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipEventRecord(c->t1, c->stream));
  HIP_TRY(c, hipEventSynchronize(c->t1));
  HIP_TRY(c, hipEventElapsedTime(ms, c->t0, c->t1));
  return VIMZ_OK;
}
int vimz_set_profiling(vimz_ctx* c, int enabled) { if (!c) return VIMZ_ERR_INVALID; c->profiling = enabled != 0; return VIMZ_OK; }
int vimz_msm_last_profile(vimz_ctx* c, float ms[6], uint32_t info[4]) {
  if (!c) return VIMZ_ERR_INVALID;
  if (ms) memcpy(ms, c->last_msm.ms, sizeof(float) * 6);
  if (info) { info[0] = c->last_msm.c; info[1] = c->last_msm.K; info[2] = c->last_msm.subs; info[3] = c->last_msm.entries; }
  return VIMZ_OK;
}

int vimz_msm_profile_totals(vimz_ctx* c, double ms[6], uint64_t counts[3], int reset) {
  if (!c) return VIMZ_ERR_INVALID;
  std::lock_guard<std::mutex> g(c->mu);
  if (ms) memcpy(ms, c->msm_tot_ms, sizeof(double) * 6);
  if (counts) { counts[0] = c->msm_tot_calls; counts[1] = c->msm_tot_points; counts[2] = c->msm_tot_entries; }
  if (reset) { memset(c->msm_tot_ms, 0, sizeof(c->msm_tot_ms)); c->msm_tot_calls = c->msm_tot_points = c->msm_tot_entries = 0; }
  return VIMZ_OK;
}

// ---- commitment key ----------------------------------------------------------------------------------
int vimz_bases_upload(vimz_ctx* c, int curve, const uint64_t* xy, size_t n, int form, vimz_bases** out) {
  if (!c || !out || (!xy && n) || curve < 0 || curve > 3) return fail(c, VIMZ_ERR_INVALID, "vimz_bases_upload: bad argument");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  vimz_bases* b = new vimz_bases(); b->curve = curve; b->n = n; b->d = nullptr;
  if (n) {
    int rc = ensure_scratch(c, 64 * n); if (rc) { delete b; return rc; }
    hipError_t e = hipMalloc(&b->d, 4 * (size_t)AFFINE_WORDS * n);
    if (e != hipSuccess) { delete b; return fail(c, VIMZ_ERR_HIP, "hipMalloc(bases)", e); }
    e = hipMemcpyAsync(c->scratch, xy, 64 * n, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { hipFree(b->d); delete b; return fail(c, VIMZ_ERR_HIP, "copy bases", e); }
    // resident form: 9 x 29-bit Montgomery coordinates (fp29.hpp)
    field_dispatch(curve_base_field(curve), [&](auto f) { typedef decltype(f) F; launch_points_to_internal<F>(c->stream, (const uint32_t*)c->scratch, form == VIMZ_FORM_CANONICAL, b->d, n); return VIMZ_OK; });
    e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { hipFree(b->d); delete b; return fail(c, VIMZ_ERR_HIP, "bases sync", e); }
  }
  *out = b;
  return VIMZ_OK;
}
int vimz_bases_generate(vimz_ctx* c, int curve, const char* label, size_t label_len, size_t n, vimz_bases** out) {
  if (!c || !out || (!label && label_len) || label_len > 64 || curve < 0 || curve > 3) return fail(c, VIMZ_ERR_INVALID, "vimz_bases_generate: bad argument");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  vimz_bases* b = new vimz_bases(); b->curve = curve; b->n = n; b->d = nullptr;
  if (n) {
    int rcs = ensure_scratch(c, 64 * n); if (rcs) { delete b; return rcs; }
    hipError_t e = hipMalloc(&b->d, 4 * (size_t)AFFINE_WORDS * n);
    if (e != hipSuccess) { delete b; return fail(c, VIMZ_ERR_HIP, "hipMalloc(bases)", e); }
    CkLabel L; memset(&L, 0, sizeof(L)); memcpy(L.bytes, label, label_len); L.len = (int)label_len;
    static const int B_COEF[4] = {3, -17, 5, 5};
    int rc = field_dispatch(curve_base_field(curve), [&](auto f) {
      typedef decltype(f) F;
      hipError_t e2 = ckgen_run<F>(c->stream, L, B_COEF[curve], 0, n, (uint32_t*)c->scratch);
      if (e2 == hipSuccess) { launch_points_to_internal<F>(c->stream, (const uint32_t*)c->scratch, 0, b->d, n); e2 = hipStreamSynchronize(c->stream); }
      return e2 == hipSuccess ? VIMZ_OK : fail(c, VIMZ_ERR_HIP, "ckgen", e2);
    });
    if (rc) { hipFree(b->d); delete b; return rc; }
  }
  *out = b;
  return VIMZ_OK;
}
int vimz_bases_download(vimz_ctx* c, const vimz_bases* b, size_t offset, uint64_t* xy, size_t n, int form) {
  if (!c || !b || (!xy && n) || offset + n > b->n) return fail(c, VIMZ_ERR_INVALID, "vimz_bases_download: bad argument");
  if (!n) return VIMZ_OK;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = ensure_scratch(c, 64 * n); if (rc) return rc;
  field_dispatch(curve_base_field(b->curve), [&](auto f) { typedef decltype(f) F; launch_points_from_internal<F>(c->stream, b->d + (size_t)AFFINE_WORDS * offset, form == VIMZ_FORM_CANONICAL, (uint32_t*)c->scratch, n); return VIMZ_OK; });
  HIP_TRY(c, hipMemcpyAsync(xy, c->scratch, 64 * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VIMZ_OK;
}
size_t vimz_bases_len(const vimz_bases* b) { return b ? b->n : 0; }
void vimz_bases_free(vimz_ctx* c, vimz_bases* b) {
  if (!b) return;
  if (c) {
    std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device); hipStreamSynchronize(c->stream);
    if (b->d) hipFree(b->d);
    if (b->tables) hipFree(b->tables);
    for (auto& t : b->small) { if (t.rows) hipFree(t.rows); if (t.mult) hipFree(t.mult); }
  }
  delete b;
}
// Window tables T_j[i] = 2^(c j) P_i for the whole key (c = window_bits, 0 -> 16): K x the key's size in HBM
// (0.67 GB for 2^19 points).  Afterwards MSMs over this key with window_bits = 0 use one shared bucket set.
int vimz_bases_precompute(vimz_ctx* c, vimz_bases* b, int window_bits) {
  if (!c || !b) return fail(c, VIMZ_ERR_INVALID, "vimz_bases_precompute: bad argument");
  const int cw = window_bits > 0 ? window_bits : 16;
  if (cw == SMALL_C) {       // the small-MSM form: rows 2^(7w)·P_i and every multiple of them (keys of at most MSM_SMALL_MAX points)
    if (b->n > MSM_SMALL_MAX) return fail(c, VIMZ_ERR_INVALID, "vimz_bases_precompute: window_bits 7 (tables of multiples) is for keys of at most 30720 points");
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(c, hipSetDevice(c->device));
    BaseTables t;
    return vz_small_tables(c, b, 0, b->n, true, &t);
  }
  if (cw < 10 || cw > 16) return fail(c, VIMZ_ERR_INVALID, "vimz_bases_precompute: window_bits must be 7 or in [10, 16]");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (b->tables) { hipFree(b->tables); b->tables = nullptr; }
  if (!b->n) return VIMZ_OK;
  return curve_dispatch(b->curve, [&](auto cv) {
    typedef decltype(cv) C;
    const int K = (C::Scalar::Params::BITS + 1 + cw - 1) / cw;
    hipError_t e = hipMalloc(&b->tables, 4 * (size_t)AFFINE_WORDS * b->n * K);
    if (e != hipSuccess) return fail(c, VIMZ_ERR_HIP, "hipMalloc(window tables)", e);
    e = build_tables<C>(c->stream, b->d, b->n, cw, K, b->tables);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { hipFree(b->tables); b->tables = nullptr; return fail(c, VIMZ_ERR_HIP, "build_tables", e); }
    b->table_c = cw; b->table_K = K;
    return VIMZ_OK;
  });
}

// ---- device vectors ----------------------------------------------------------------------------------
int vimz_vec_alloc(vimz_ctx* c, int field, size_t n, vimz_vec** out) {
  if (!c || !out || field < 0 || field > 3) return fail(c, VIMZ_ERR_INVALID, "vimz_vec_alloc: bad argument");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  vimz_vec* v = new vimz_vec{field, n, nullptr};
  if (n) {
    hipError_t e = hipMalloc(&v->d, 32 * n);
    if (e != hipSuccess) { delete v; return fail(c, VIMZ_ERR_HIP, "hipMalloc(vec)", e); }
    e = hipMemsetAsync(v->d, 0, 32 * n, c->stream);
    if (e != hipSuccess) { hipFree(v->d); delete v; return fail(c, VIMZ_ERR_HIP, "memset(vec)", e); }
  }
  *out = v;
  return VIMZ_OK;
}
int vimz_vec_upload(vimz_ctx* c, vimz_vec* v, size_t offset, const uint64_t* host, size_t n, int form) {
  if (!c || !v || (!host && n) || offset + n > v->n) return fail(c, VIMZ_ERR_INVALID, "vimz_vec_upload: bad argument");
  if (!n) return VIMZ_OK;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  uint32_t* dst = v->d + 8 * offset;
  HIP_TRY(c, hipMemcpyAsync(dst, host, 32 * n, hipMemcpyHostToDevice, c->stream));
  if (form == VIMZ_FORM_CANONICAL)
    field_dispatch(v->field, [&](auto f) { typedef decltype(f) F; launch_to_mont<F>(c->stream, dst, n); return VIMZ_OK; });
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // host buffer may be reused by the caller
  return VIMZ_OK;
}
int vimz_vec_download(vimz_ctx* c, const vimz_vec* v, size_t offset, uint64_t* host, size_t n, int form) {
  if (!c || !v || (!host && n) || offset + n > v->n) return fail(c, VIMZ_ERR_INVALID, "vimz_vec_download: bad argument");
  if (!n) return VIMZ_OK;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t* src = v->d + 8 * offset;
  if (form == VIMZ_FORM_CANONICAL) {
    int rc = ensure_scratch(c, 32 * n); if (rc) return rc;
    field_dispatch(v->field, [&](auto f) { typedef decltype(f) F; launch_from_mont<F>(c->stream, src, (uint32_t*)c->scratch, n); return VIMZ_OK; });
    src = (const uint32_t*)c->scratch;
  }
  HIP_TRY(c, hipMemcpyAsync(host, src, 32 * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VIMZ_OK;
}
size_t vimz_vec_len(const vimz_vec* v) { return v ? v->n : 0; }
void vimz_vec_free(vimz_ctx* c, vimz_vec* v) {
  if (!v) return;
  if (c) { std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device); hipStreamSynchronize(c->stream); if (v->d) hipFree(v->d); }
  delete v;
}

// ---- MSM ---------------------------------------------------------------------------------------------
int vimz_msm(vimz_ctx* c, const vimz_bases* bases, const uint64_t* scalars, size_t n, int form, int window_bits,
             uint64_t out_xy[8], int out_form) {
  if (!c || !bases || !out_xy || (!scalars && n) || n > bases->n) return fail(c, VIMZ_ERR_INVALID, "vimz_msm: bad argument");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n) {
    int rc = ensure_scratch(c, 32 * n); if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->scratch, scalars, 32 * n, hipMemcpyHostToDevice, c->stream));
  }
  return msm_device(c, bases, 0, (const uint32_t*)c->scratch, n, form == VIMZ_FORM_MONTGOMERY, window_bits, out_xy, out_form);
}

int vimz_msm_vec(vimz_ctx* c, const vimz_bases* bases, size_t base_offset, const vimz_vec* v, size_t offset, size_t n,
                 int window_bits, uint64_t out_xy[8], int out_form) {
  if (!c || !bases || !v || !out_xy || offset + n > v->n || base_offset + n > bases->n)
    return fail(c, VIMZ_ERR_INVALID, "vimz_msm_vec: bad argument");
  if (v->field != curve_scalar_field(bases->curve)) return fail(c, VIMZ_ERR_INVALID, "vimz_msm_vec: vector is not over the curve's scalar field");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  return msm_device(c, bases, base_offset, v->d + 8 * offset, n, 1, window_bits, out_xy, out_form);
}

int vimz_msm_vec_ex(vimz_ctx* c, const vimz_bases* bases, size_t base_offset, const vimz_vec* v, size_t offset, size_t n,
                    int window_bits, int flags, uint64_t out_xy[8], int out_form) {
  if (!c || !bases || !v || !out_xy || offset + n > v->n || base_offset + n > bases->n)
    return fail(c, VIMZ_ERR_INVALID, "vimz_msm_vec_ex: bad argument");
  if (v->field != curve_scalar_field(bases->curve)) return fail(c, VIMZ_ERR_INVALID, "vimz_msm_vec_ex: vector is not over the curve's scalar field");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  return vz::vz_msm_device(c, bases, base_offset, v->d + 8 * offset, n, 1, window_bits, out_xy, out_form, (flags & VIMZ_MSM_SPLIT_ONES) ? 1 : 0);
}

// KZG opening of a committed vector at a point (the `KZG::prove` the Sonobe decider calls for the final commitments, reached from
// vimz/src/sonobe_backend/decider.rs:13-21): with the SRS's G1 powers as bases, comm = Σ v_i·[τ^i]G commits to p(X) = Σ v_i X^i;
// the opening at z is  eval = p(z)  and  proof = commit((p(X) − p(z)) / (X − z)) — the quotient's coefficients by synthetic division
// (q_{n-2} = v_{n-1}, q_{i-1} = v_i + z·q_i; a serial recurrence of n multiplications: host, ≈ 10 ms at 3·10^5), then the same MSM as the
// commitment, one point shorter.  The pairing check e(proof, [τ − z]H) = e(comm − eval·G, H) is the on-chain verifier's (no G2 arithmetic here).
int vimz_kzg_open(vimz_ctx* c, const vimz_bases* srs, size_t base_offset, const vimz_vec* v, size_t offset, size_t n, const uint64_t z[4], int form,
                  uint64_t eval_out[4], uint64_t proof_xy[8]) {
  if (!c || !srs || !v || !z || !eval_out || !proof_xy || n == 0 || offset + n > v->n || base_offset + n > srs->n) return fail(c, VIMZ_ERR_INVALID, "vimz_kzg_open: bad argument");
  if (v->field != curve_scalar_field(srs->curve)) return fail(c, VIMZ_ERR_INVALID, "vimz_kzg_open: vector is not over the curve's scalar field");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  return vz::vz_kzg_open_device(c, srs, base_offset, v->field, v->d + 8 * offset, n, z, form, eval_out, proof_xy);
}

// ---- probes ------------------------------------------------------------------------------------------
int vimz_field_op(vimz_ctx* c, int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  if (!c || !a || !out || op < 0 || op > 3 || (op != 3 && !b)) return fail(c, VIMZ_ERR_INVALID, "vimz_field_op: bad argument");
  if (!n) return VIMZ_OK;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = ensure_scratch(c, 96 * n); if (rc) return rc;
  uint32_t* da = (uint32_t*)c->scratch; uint32_t* db = da + 8 * n; uint32_t* dout = db + 8 * n;
  HIP_TRY(c, hipMemcpyAsync(da, a, 32 * n, hipMemcpyHostToDevice, c->stream));
  if (b) HIP_TRY(c, hipMemcpyAsync(db, b, 32 * n, hipMemcpyHostToDevice, c->stream));
  rc = field_dispatch(field, [&](auto f) { typedef decltype(f) F; launch_field_probe<F>(c->stream, op, da, db, dout, n); return VIMZ_OK; });
  if (rc) return fail(c, rc, "vimz_field_op: bad field");
  HIP_TRY(c, hipMemcpyAsync(out, dout, 32 * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VIMZ_OK;
}

int vimz_curve_add(vimz_ctx* c, int curve, const uint64_t* p, const uint64_t* q, uint64_t* out, size_t n) {
  if (!c || !p || !q || !out) return fail(c, VIMZ_ERR_INVALID, "vimz_curve_add: bad argument");
  if (!n) return VIMZ_OK;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = ensure_scratch(c, 192 * n); if (rc) return rc;
  uint32_t* dp = (uint32_t*)c->scratch; uint32_t* dq = dp + 16 * n; uint32_t* dout = dq + 16 * n;
  HIP_TRY(c, hipMemcpyAsync(dp, p, 64 * n, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(dq, q, 64 * n, hipMemcpyHostToDevice, c->stream));
  rc = curve_dispatch(curve, [&](auto cv) { typedef typename decltype(cv)::Base F; launch_curve_add_probe<F>(c->stream, dp, dq, dout, n); return VIMZ_OK; });
  if (rc) return fail(c, rc, "vimz_curve_add: bad curve");
  HIP_TRY(c, hipMemcpyAsync(out, dout, 64 * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VIMZ_OK;
}

}  // extern "C"
