// Internal definitions shared by the translation units behind the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/vimz_hip.h"
#include "ec.hpp"
#include "msm_api.hpp"

struct vimz_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  hipEvent_t ev[7] = {};
  bool profiling = false;
  vz::MsmWorkspace msm_ws;
  vz::MsmStats last_msm = {};
  double msm_tot_ms[6] = {};            // sums over every profiled MSM since the last reset
  uint64_t msm_tot_calls = 0, msm_tot_points = 0, msm_tot_entries = 0;
  std::mutex mu;
  std::string err;
  void* scratch = nullptr;  // device staging for host-scalar MSM / probes
  size_t scratch_bytes = 0;
  // Streams of provers that were freed, kept for the provers created next (vz_stream_acquire / vz_stream_release): no stream is created and
  // destroyed per fold call (the state-chain helper's) or per prover, and a context's later provers sit on the streams its first ones had.
  // (Not the cure for the slow folds of provers created after others had folded on the context — that was a priority inversion between
  // the streams, prover_internal.hpp: wait_row_flag — but kept: no stream creation per fold call.)
  std::mutex stream_mu;
  std::vector<std::pair<int, hipStream_t>> spare_streams;     // (priority, stream)
};
// a non-blocking stream of the given priority on c's device (caller has set the device): a recycled one if there is one
static inline hipError_t vz_stream_acquire(vimz_ctx* c, int priority, hipStream_t* out) {
  {
    std::lock_guard<std::mutex> g(c->stream_mu);
    for (size_t i = 0; i < c->spare_streams.size(); i++)
      if (c->spare_streams[i].first == priority) { *out = c->spare_streams[i].second; c->spare_streams.erase(c->spare_streams.begin() + i); return hipSuccess; }
  }
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, priority);
}
// hand a stream back (synchronised first); destroyed with the context
static inline void vz_stream_release(vimz_ctx* c, int priority, hipStream_t s) {
  if (!s) return;
  hipStreamSynchronize(s);
  std::lock_guard<std::mutex> g(c->stream_mu);
  c->spare_streams.emplace_back(priority, s);
}
// Tables of a fixed slice [offset, offset + n) of a key for the fused small MSMs (the four per-step MSMs of an IVC run over fixed
// slices): rows 2^(7w)·P_i and every multiple m·2^(7w)·P_i, m = 1..64 (msm_api.hpp: BaseTables).  Built once per key and slice,
// shared by every IVC created over the key, released with the key.
struct vimz_small_tables { size_t offset = 0, n = 0; uint32_t* rows = nullptr; uint32_t* mult = nullptr; int c = 0, K = 0; };
struct vimz_bases {
  int curve; size_t n; uint32_t* d;
  std::mutex small_mu; std::vector<vimz_small_tables> small;
  uint32_t* tables = nullptr; int table_c = 0, table_K = 0;   // optional window tables (vimz_bases_precompute)
  vz::BaseTables tb(size_t offset) const { return vz::BaseTables{tables, n, offset, table_c, table_K, table_c == 11}; }
};
struct vimz_vec { int field; size_t n; uint32_t* d; };


namespace vz {
int vz_fail(vimz_ctx* c, int code, const char* what, hipError_t e = hipSuccess);
int vz_ensure_scratch(vimz_ctx* c, size_t bytes);
// tables of the slice [offset, offset + n) of `b` for the fused small MSMs (built on first use on the context's stream; caller holds
// c->mu and has set the device).  with_mult: also every multiple (64 x the rows' footprint).  out->d == nullptr: n too large.
int vz_small_tables(vimz_ctx* c, vimz_bases* b, size_t offset, size_t n, bool with_mult, BaseTables* out);
// MSM over device-resident scalars on the context's stream (caller holds c->mu and has set the device)
int vz_msm_device(vimz_ctx* c, const vimz_bases* bases, size_t base_offset, const uint32_t* d_scalars, size_t n,
                  int scalars_mont, int window_bits, uint64_t out_xy[8], int out_form, int split_ones = 0);
// KZG opening (vimz_kzg_open) of a device-resident vector of Montgomery elements of field `field` (caller holds c->mu and has set the device)
int vz_kzg_open_device(vimz_ctx* c, const vimz_bases* srs, size_t base_offset, int field, const uint32_t* d_vec, size_t n, const uint64_t z[4], int form,
                       uint64_t eval_out[4], uint64_t proof_xy[8]);
}
