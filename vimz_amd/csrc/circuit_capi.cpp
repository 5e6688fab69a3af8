// C ABI for the step-circuit builder (host-only translation unit, built with g++).
#include <cstring>
#include <memory>
#include <string>
#include <algorithm>
#include "../../include/vimz_hip.h"
#include "circuit/circuits.hpp"
#include "circuit_handle.hpp"

using namespace vz;
using namespace vz::cb;

static thread_local std::string g_circuit_err;

namespace {
struct Reader {
  const uint8_t* p; size_t len, pos = 0; bool ok = true;
  template <class T> T get() { T v{}; if (pos + sizeof(T) > len) { ok = false; return v; } memcpy(&v, p + pos, sizeof(T)); pos += sizeof(T); return v; }
  const uint8_t* bytes(size_t n) { if (pos + n > len) { ok = false; return nullptr; } const uint8_t* r = p + pos; pos += n; return r; }
};
const uint8_t BN254_FR_LE[32] = {0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
                                 0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};
}  // namespace

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
    case VIMZ_CX_LC_TERMS: vec(b.lc_terms); break;
    default: return VIMZ_ERR_INVALID;
  }
  if (buf && cap >= bytes && bytes) memcpy(buf, src, bytes);
  return (int64_t)bytes;
}

// ---- iden3 binary formats: what circom writes and nova_scotia::circom::reader::load_r1cs reads
//      (vimz/src/nova_snark_backend/folding.rs:22).  A circom-built circuit loaded this way has no witness program: its
//      witnesses come from circom's own generator (.wtns) and are folded with vimz_prover_fold_witness. -----------------

int vimz_circuit_load_r1cs(const uint8_t* data, size_t len, vimz_circuit** out) {
  if (!data || !out) { g_circuit_err = "vimz_circuit_load_r1cs: bad argument"; return VIMZ_ERR_INVALID; }
  Reader R{data, len};
  const uint8_t* magic = R.bytes(4);
  if (!magic || memcmp(magic, "r1cs", 4) != 0 || R.get<uint32_t>() != 1) { g_circuit_err = "not an iden3 r1cs v1 file"; return VIMZ_ERR_INVALID; }
  const uint32_t nsec = R.get<uint32_t>();
  size_t hdr_pos = 0, con_pos = 0, con_size = 0;
  for (uint32_t i = 0; i < nsec && R.ok; i++) {
    const uint32_t type = R.get<uint32_t>(); const uint64_t size = R.get<uint64_t>();
    if (type == 1) hdr_pos = R.pos; else if (type == 2) { con_pos = R.pos; con_size = size; }
    R.bytes(size);
  }
  if (!R.ok || !hdr_pos || !con_pos) { g_circuit_err = "r1cs: missing header or constraint section"; return VIMZ_ERR_INVALID; }
  R.pos = hdr_pos;
  const uint32_t fs = R.get<uint32_t>();
  const uint8_t* prime = R.bytes(32);
  if (fs != 32 || !prime || memcmp(prime, BN254_FR_LE, 32) != 0) { g_circuit_err = "r1cs: field is not BN254 Fr"; return VIMZ_ERR_INVALID; }
  const uint32_t n_wires = R.get<uint32_t>(), n_out = R.get<uint32_t>(), n_in = R.get<uint32_t>(), n_prv = R.get<uint32_t>();
  R.get<uint64_t>();  // labels
  const uint32_t n_con = R.get<uint32_t>();
  if (!R.ok || n_out != n_in || 1 + n_out + n_in + n_prv > n_wires) { g_circuit_err = "r1cs: step circuits need as many public outputs as public inputs"; return VIMZ_ERR_INVALID; }
  auto h = std::make_unique<vimz_circuit>();
  h->transformation = -1; h->shape = StepShape{0, 0, 0, 0, 0};
  h->build = std::make_unique<CircuitBuild>();
  Builder& b = h->build->b;
  b.n_wires = n_wires; b.len_z = n_out; b.n_priv = n_prv;
  R.pos = con_pos;
  const size_t con_end = con_pos + con_size;
  for (uint32_t k = 0; k < n_con; k++) {
    Csr* M[3] = {&b.A, &b.B, &b.C};
    for (int m = 0; m < 3; m++) {
      const uint32_t nt = R.get<uint32_t>();
      LC lc; lc.t.reserve(nt);
      for (uint32_t t = 0; t < nt && R.ok; t++) {
        const uint32_t w = R.get<uint32_t>();
        const uint8_t* c = R.bytes(32);
        if (!c || w >= n_wires) { R.ok = false; break; }
        Fe x; memcpy(x.v, c, 32);
        lc.t.push_back({w, Fe::to_mont(x)});
      }
      std::sort(lc.t.begin(), lc.t.end(), [](const Term& a, const Term& c2) { return a.w < c2.w; });
      b.push_row(*M[m], lc);
    }
    if (!R.ok || R.pos > con_end) { g_circuit_err = "r1cs: truncated constraint section"; return VIMZ_ERR_INVALID; }
  }
  *out = h.release();
  return VIMZ_OK;
}

// .wtns -> n_out canonical elements (returns the witness length through n_out; copies when out != NULL and cap is enough)
int vimz_wtns_load(const uint8_t* data, size_t len, uint64_t* out, size_t cap_elems, size_t* n_out) {
  if (!data || !n_out) return VIMZ_ERR_INVALID;
  Reader R{data, len};
  const uint8_t* magic = R.bytes(4);
  if (!magic || memcmp(magic, "wtns", 4) != 0) { g_circuit_err = "not an iden3 wtns file"; return VIMZ_ERR_INVALID; }
  R.get<uint32_t>();  // version
  const uint32_t nsec = R.get<uint32_t>();
  uint32_t nw = 0; const uint8_t* vals = nullptr;
  for (uint32_t i = 0; i < nsec && R.ok; i++) {
    const uint32_t type = R.get<uint32_t>(); const uint64_t size = R.get<uint64_t>();
    const size_t start = R.pos;
    if (type == 1) {
      const uint32_t fs = R.get<uint32_t>(); const uint8_t* prime = R.bytes(32);
      if (fs != 32 || !prime || memcmp(prime, BN254_FR_LE, 32) != 0) { g_circuit_err = "wtns: field is not BN254 Fr"; return VIMZ_ERR_INVALID; }
      nw = R.get<uint32_t>();
    } else if (type == 2) vals = data + R.pos;
    R.pos = start; R.bytes(size);
  }
  if (!R.ok || !vals || !nw || (size_t)(vals - data) + 32 * (size_t)nw > len) { g_circuit_err = "wtns: malformed"; return VIMZ_ERR_INVALID; }
  *n_out = nw;
  if (out && cap_elems >= nw) memcpy(out, vals, 32 * (size_t)nw);
  return VIMZ_OK;
}

}  // extern "C"

// ---- the augmented verifier circuits on their own (host-only parity hooks; the IVC prover is in ivc.hip) ---------------------
#include "aug/export.hpp"
struct vimz_augcircuit {
  int side;
  vz::aug::AugCircuit<vz::BnFr> c1;
  vz::aug::AugCircuit<vz::BnFq> c2;
};
extern "C" {
int vimz_augcircuit_build(int side, vimz_augcircuit** out) {
  if (!out || (side != 0 && side != 1)) { g_circuit_err = "vimz_augcircuit_build: bad argument"; return VIMZ_ERR_INVALID; }
  try {
    auto h = std::make_unique<vimz_augcircuit>();
    h->side = side;
    if (side == 0) { h->c1.init_trivial_step(); h->c1.finish(true); }
    else { h->c2.init_trivial_step(); h->c2.finish(false); }
    *out = h.release();
    return VIMZ_OK;
  } catch (const std::exception& e) { g_circuit_err = e.what(); return VIMZ_ERR_INVALID; }
}
void vimz_augcircuit_free(vimz_augcircuit* c) { delete c; }
int64_t vimz_augcircuit_export(const vimz_augcircuit* c, int what, void* buf, size_t cap) {
  if (!c) return VIMZ_ERR_INVALID;
  return c->side == 0 ? vz::aug::export_r1cs(c->c1, what, buf, cap) : vz::aug::export_r1cs(c->c2, what, buf, cap);
}
int vimz_augcircuit_witness(const vimz_augcircuit* c, const uint64_t* inputs, uint64_t* wires_out, uint64_t* outputs) {
  if (!c || !inputs) return VIMZ_ERR_INVALID;
  try {
    return c->side == 0 ? vz::aug::witness_flat(c->c1, inputs, wires_out, outputs) : vz::aug::witness_flat(c->c2, inputs, wires_out, outputs);
  } catch (const std::exception& e) { g_circuit_err = e.what(); return VIMZ_ERR_INVALID; }
}
}  // extern "C"
