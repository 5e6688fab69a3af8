// The merged proof of several row segments (merge.hip; spartan.hip compresses it): ONE verifiable object out of S Nova IVC proofs of
// contiguous row segments — the "host-side sequential final fold" of BASELINE.json's north_star applied to IVC proofs, so that
// segments folded concurrently (on one GPU or on several) end in the single proof object the reference's fold_input returns
// (vimz/src/nova_snark_backend/folding.rs:27-43).
//
// Protocol (ours, like the augmented circuits: nova-snark has nothing of the kind).  Out-of-circuit NIFS, the same
// final fold CompressedSNARK::prove performs on the last fresh secondary instance, extended to S segments and made a tree:
//   record j   = (n_j, z_start_j, z_end_j, U1_j, U2_j, u2_j, T_j): an IVC proof's statement and instances; T_j = comm of the cross term
//                of (U2_j, u2_j), which the IVC holds anyway for its next step.
//   Leaf(j)    : checks  u2_j.x0 = H1(digest1, n_j, z_start_j, z_end_j, U2_j)  and  u2_j.x1 = H2(digest2, n_j, 0, 0, U1_j);
//                h = SHA3-256("vimz-merge-leaf-v1" ‖ digest1 ‖ digest2 ‖ len_z ‖ record j),  r = chal(h, 'q');
//                P = U1_j (as a relaxed instance over Fr),  Q = U2_j + r·u2_j  (NIFS with E += r·T_j).
//   Node(A,B)  : requires A.z_end = B.z_start;  h = SHA3-256("vimz-merge-node-v1" ‖ A.h ‖ B.h ‖ T_p ‖ T_q),  r_p = chal(h,'p'), r_q = chal(h,'q');
//                P = A.P + r_p·B.P,  Q = A.Q + r_q·B.Q   (NIFS for two relaxed instances:  W += r·W',  E += r·T + r²·E',  u += r·u',  X += r·X');
//                n = A.n + B.n, z_start = A.z_start, z_end = B.z_end.
//   chal(h, t) = the first 16 bytes of SHA3-256(h ‖ t) as a little-endian integer.
// Every challenge is derived after the cross-term commitment it multiplies and binds, through the hash chain, every instance below it.
// The verifier replays the tree over the records (host arithmetic only), obtains (n, z_start, z_end, P, Q) and then checks ONE primary
// and ONE secondary relaxed instance against their witnesses — or one compressed argument for each (vimz_ivc_merged_compress).
#pragma once
#include "ivc_internal.hpp"
#include "proof_io.hpp"

struct MInstP { G1Aff cW, cE; Fe u, X0, X1; };      // relaxed instance of the primary circuit (scalars in BN254 Fr, commitments on G1)
struct MInstQ { G2Aff cW, cE; Fq u, X0, X1; };      // of the secondary circuit (BN254 Fq, Grumpkin)
struct MSeg { uint64_t n = 0; std::vector<Fe> zs, ze; RelaxedInst<Fq> U1; RelaxedInst<Fe> U2; FreshInst<Fe> u2; G2Aff T; };
struct MOp { uint32_t kind = 0, leaf = 0; G1Aff Tp; G2Aff Tq; };      // kind 0: push Leaf(leaf);  kind 1: pop B, pop A, push Node(A, B; Tp, Tq)
struct MAcc { uint8_t h[32] = {}; uint64_t n = 0; std::vector<Fe> zs, ze; MInstP P; MInstQ Q; };

struct vimz_ivc_merged {
  vimz_ivc* vk = nullptr;                 // shapes, keys, context (must outlive this object)
  std::vector<MSeg> segs; std::vector<MOp> ops;
  MAcc acc;                               // what the ops evaluate to
  uint32_t* dev = nullptr;                // one allocation: the folded witnesses and running products of both sides, scratch
  uint32_t *Zp = nullptr, *Ep = nullptr, *AZp = nullptr, *BZp = nullptr, *CZp = nullptr;
  uint32_t *Zq = nullptr, *Eq = nullptr, *AZq = nullptr, *BZq = nullptr, *CZq = nullptr;
  uint32_t* leaf_q[5] = {};               // Z, E, AZ, BZ, CZ of an incoming segment's folded secondary instance
  uint32_t* Tq = nullptr;                 // secondary cross term
  void* pin = nullptr;                    // pinned window sums of the secondary cross-term commitment
  double seconds[4] = {};                 // leaf work, node GPU wait, node host, total
  bool broken = false;                    // a merge failed after its folds were queued: the vectors no longer match the records
};

namespace {
// ---- the statement part of a merged proof as words: header, segment records, ops (canonical little-endian) ------------------------------
const uint64_t MERGED_MAGIC = 0x3147524d5a56ull;      // "VZMRG1"

void write_records(const vimz_ivc_merged* m, Writer& w) {
  const vimz_ivc* vk = m->vk;
  w.word(MERGED_MAGIC); w.word(m->segs.size()); w.word(m->ops.size()); w.word(vk->pri->len_z);
  w.word(vk->pri->n_wires); w.word(vk->pri->n_c); w.word(vk->sec.n_w); w.word(vk->sec.n_c);
  for (auto& s : m->segs) {
    w.word(s.n);
    for (auto& z : s.zs) w.fe(z);
    for (auto& z : s.ze) w.fe(z);
    w.point(s.U1.W); w.point(s.U1.E); w.fe(s.U1.u); w.u256(s.U1.X0); w.u256(s.U1.X1);
    w.point(s.U2.W); w.point(s.U2.E); w.fe(s.U2.u); w.u256(s.U2.X0); w.u256(s.U2.X1);
    w.point(s.u2.W); w.fe(s.u2.x0); w.fe(s.u2.x1);
    w.point(s.T);
  }
  for (auto& o : m->ops) {
    w.word(o.kind); w.word(o.leaf);
    if (o.kind == 1) { w.point(o.Tp); w.point(o.Tq); }
  }
}
size_t records_words(const vimz_ivc_merged* m) {
  const size_t lz = m->vk->pri->len_z;
  size_t n = 8 + m->segs.size() * (1 + 4 * (2 * lz + 7 + 7 + 4 + 2));
  for (auto& o : m->ops) n += 2 + (o.kind == 1 ? 16 : 0);
  return n;
}


// parse what write_records wrote, from an untrusted source: range checks on every element, curve checks on every point
bool read_records(Reader& in, const vimz_ivc* vk, std::vector<MSeg>& segs, std::vector<MOp>& ops) {
  const vimz_prover* p = vk->pri;
  const size_t lz = p->len_z;
  const uint64_t magic = in.word(), S = in.word(), n_ops = in.word(), lzb = in.word(), nw1 = in.word(), nc1 = in.word(), nw2 = in.word(), nc2 = in.word();
  if (!in.ok || magic != MERGED_MAGIC || lzb != lz || nw1 != p->n_wires || nc1 != p->n_c || nw2 != vk->sec.n_w || nc2 != vk->sec.n_c || S == 0 || S > 4096 || n_ops != 2 * S - 1) return false;
  segs.assign(S, MSeg()); ops.assign(n_ops, MOp());
  for (auto& s : segs) {
    s.n = in.word();
    s.zs.resize(lz); s.ze.resize(lz);
    for (auto& z : s.zs) z = in.fe<Fe>();
    for (auto& z : s.ze) z = in.fe<Fe>();
    s.U1.W = in.point<Fq>(); s.U1.E = in.point<Fq>(); s.U1.u = in.fe<Fq>(); s.U1.X0 = in.u256(); s.U1.X1 = in.u256();
    s.U2.W = in.point<Fe>(); s.U2.E = in.point<Fe>(); s.U2.u = in.fe<Fe>(); s.U2.X0 = in.u256(); s.U2.X1 = in.u256();
    s.u2.W = in.point<Fe>(); s.u2.x0 = in.fe<Fe>(); s.u2.x1 = in.fe<Fe>();
    s.T = in.point<Fe>();
    if (!in.ok) return false;
  }
  for (auto& o : ops) {
    o.kind = (uint32_t)in.word(); o.leaf = (uint32_t)in.word();
    o.Tp.x = o.Tp.y = Fq::zero(); o.Tq.x = o.Tq.y = Fe::zero();
    if (o.kind == 1) { o.Tp = in.point<Fq>(); o.Tq = in.point<Fe>(); } else if (o.kind != 0) in.ok = false;
    if (!in.ok) return false;
  }
  return true;
}
}  // namespace

namespace vz {
// Evaluate the ops over the records with host arithmetic.  flags: bit 0 / 1 a segment's primary / secondary chain hash differs;
// bit 12 segments not adjacent; bit 13 malformed op sequence.  Returns false only when nothing could be evaluated (bit 13).
bool merged_replay(const vimz_ivc* vk, const std::vector<MSeg>& segs, const std::vector<MOp>& ops, MAcc* out, uint32_t* flags);
}
