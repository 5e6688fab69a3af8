// C ABI for the step-circuit builder (host-only translation unit, built with g++).
#include <cstring>
#include <memory>
#include <string>
#include "../../include/vimz_hip.h"
#include "circuit/circuits.hpp"
#include "circuit_handle.hpp"

using namespace vz;
using namespace vz::cb;

static thread_local std::string g_circuit_err;

extern "C" {

const char* vimz_circuit_last_error(void) { return g_circuit_err.c_str(); }

int vimz_circuit_build(int transformation, int width, int width2, int rows_in, int rows_out, int crop_height, vimz_circuit** out) {
  if (!out || width <= 0) { g_circuit_err = "vimz_circuit_build: bad argument"; return VIMZ_ERR_INVALID; }
  try {
    StepShape S{width, width2, rows_in, rows_out, crop_height};
    auto cbuild = build_step_circuit(transformation, S);
    auto* h = new vimz_circuit();
    h->transformation = transformation; h->shape = S;
    h->build = std::move(cbuild);
    *out = h;
    return VIMZ_OK;
  } catch (const std::exception& e) {
    g_circuit_err = e.what();
    return VIMZ_ERR_INVALID;
  }
}

void vimz_circuit_free(vimz_circuit* c) { delete c; }

int vimz_circuit_info(const vimz_circuit* c, uint64_t info[VIMZ_CIRCUIT_INFO_LEN]) {
  if (!c || !info) return VIMZ_ERR_INVALID;
  const Builder& b = c->build->b;
  memset(info, 0, sizeof(uint64_t) * VIMZ_CIRCUIT_INFO_LEN);
  info[0] = b.n_wires; info[1] = b.n_constraints(); info[2] = b.n_linear; info[3] = b.len_z; info[4] = b.n_priv;
  info[5] = b.A.col.size(); info[6] = b.B.col.size(); info[7] = b.C.col.size(); info[8] = b.dict.size();
  info[9] = b.decomp.size(); info[10] = b.lane_groups.size(); info[11] = b.lane_instr.size(); info[12] = b.lane_rows.size();
  info[13] = b.jobs.size(); info[14] = b.chains.size(); info[15] = b.fops.size();
  return VIMZ_OK;
}

// Copies one table of the circuit into `buf` (returns the byte size needed when buf is NULL or too small).
int64_t vimz_circuit_export(const vimz_circuit* c, int what, void* buf, size_t cap) {
  if (!c) return VIMZ_ERR_INVALID;
  const Builder& b = c->build->b;
  const void* src = nullptr; size_t bytes = 0;
  std::vector<uint32_t> tmp;
  auto vec = [&](const auto& v) { src = v.data(); bytes = v.size() * sizeof(v[0]); };
  switch (what) {
    case VIMZ_CX_A_ROWPTR: vec(b.A.row_ptr); break;
    case VIMZ_CX_A_COL: vec(b.A.col); break;
    case VIMZ_CX_A_COEF: vec(b.A.coef); break;
    case VIMZ_CX_B_ROWPTR: vec(b.B.row_ptr); break;
    case VIMZ_CX_B_COL: vec(b.B.col); break;
    case VIMZ_CX_B_COEF: vec(b.B.coef); break;
    case VIMZ_CX_C_ROWPTR: vec(b.C.row_ptr); break;
    case VIMZ_CX_C_COL: vec(b.C.col); break;
    case VIMZ_CX_C_COEF: vec(b.C.coef); break;
    case VIMZ_CX_DICT_MONT: vec(b.dict); break;
    case VIMZ_CX_DICT_CANON: {
      bytes = b.dict.size() * 32;
      if (buf && cap >= bytes) { Fe* o = (Fe*)buf; for (size_t i = 0; i < b.dict.size(); i++) o[i] = Fe::from_mont(b.dict[i]); }
      return (int64_t)bytes;
    }
    case VIMZ_CX_DECOMP: vec(b.decomp); break;
    case VIMZ_CX_LANE_GROUPS: vec(b.lane_groups); break;
    case VIMZ_CX_LANE_INSTR: vec(b.lane_instr); break;
    case VIMZ_CX_LANE_ROWS: vec(b.lane_rows); break;
    case VIMZ_CX_JOBS: vec(b.jobs); break;
    case VIMZ_CX_CHAINS: vec(b.chains); break;
    case VIMZ_CX_FOPS: vec(b.fops); break;
    case VIMZ_CX_ZOUT: vec(b.zout); break;
    default: return VIMZ_ERR_INVALID;
  }
  if (buf && cap >= bytes && bytes) memcpy(buf, src, bytes);
  return (int64_t)bytes;
}

}  // extern "C"
