// The VIMz step circuits, width-parametrised (the reference fixes width 128 in circuits/nova_snark/*.circom
// `component main`; 4K/8K need 384/768).  Each function mirrors one Circom template:
//   hash        circuits/nova_snark/hash_step.circom:6-16
//   grayscale   circuits/src/grayscale_step.circom:8-66
//   contrast    circuits/src/contrast_step.circom:10-98
//   brightness  circuits/src/brightness_step.circom:7-105
//   blur        circuits/src/blur_step.circom:6-73        (+ utils/convolution_step.circom:10-48)
//   sharpness   circuits/src/sharpness_step.circom:6-103
//   resize      circuits/src/resize_step.circom:10-112
//   redact      circuits/src/redact_step.circom:7-26
// State updates: circuits/src/utils/state.circom:11-79.  IVC state layouts: vimz/src/transformation.rs:25-50.
//   crop        circuits/src/crop_step.circom:9-120 (literal; its cropped-row hash chain is phase 2: see the T_CROP case)
#pragma once
#include "gadgets.hpp"

namespace vz {
namespace cb {

enum Transformation { T_BLUR = 0, T_BRIGHTNESS, T_CONTRAST, T_CROP, T_GRAYSCALE, T_HASH, T_REDACT, T_RESIZE, T_SHARPNESS };

struct StepShape { int width, width2, rows_in, rows_out, crop_height; };

inline int ivc_state_len(int t) {
  switch (t) {
    case T_BLUR: case T_SHARPNESS: return 4;
    case T_BRIGHTNESS: case T_CONTRAST: case T_CROP: return 3;
    case T_GRAYSCALE: case T_REDACT: case T_RESIZE: return 2;
    default: return 1;
  }
}
inline int step_input_width(int t, const StepShape& s) {
  switch (t) {
    case T_BLUR: case T_SHARPNESS: return 4 * s.width;
    case T_BRIGHTNESS: case T_CONTRAST: case T_GRAYSCALE: return 2 * s.width;
    case T_CROP: case T_HASH: return s.width;
    case T_REDACT: return s.width + 1;
    default: return s.rows_in * s.width + s.rows_out * s.width2;
  }
}

struct CircuitBuild {
  Builder b;
  Gadgets g;
  uint32_t out0 = 1, in0 = 0, priv0 = 0;
  CircuitBuild() : g(b) {}
  uint32_t out_wire(int i) const { return out0 + i; }
  FV zin(int i) const { return FV{LC::wire(in0 + i), ValRef{REF_ZIN, (uint32_t)i}}; }
  std::vector<FV> row_fv(uint32_t src, int len) const { std::vector<FV> v; for (int i = 0; i < len; i++) v.push_back(fv_wire(src + i)); return v; }

  // HeadTailHasher(w)(head, row) -> out wire: phase-A chain for the row, phase-B pair hash
  void head_tail_to_output(const FV& head, uint32_t row_src, int w, int out_idx) {
    g.begin_chain(0);
    FV rh = g.array_hash(row_fv(row_src, w));
    g.begin_chain(1);
    g.pair_hash(head, rh, out_wire(out_idx));
    b.zout[out_idx] = ZOut{ValRef{REF_JOB, (uint32_t)b.jobs.size() - 1}, 0};
  }
  void passthrough_output(int idx) {  // step_out[idx] <== step_in[idx]  (kept as the one linear constraint)
    b.enforce(LC::constant(Fe::one()), LC::wire(in0 + idx), LC::wire(out_wire(idx)));
    b.n_linear++;
    b.zout[idx] = ZOut{ValRef{REF_ZIN, (uint32_t)idx}, 0};
  }
};

inline std::unique_ptr<CircuitBuild> build_step_circuit(int t, const StepShape& S) {
  auto cbp = std::make_unique<CircuitBuild>();
  CircuitBuild& c = *cbp;
  Builder& b = c.b; Gadgets& g = c.g;
  typedef Gadgets::LaneCtx LaneCtx;
  typedef Gadgets::LV LV;
  const int w = S.width;
  b.len_z = (uint32_t)ivc_state_len(t);
  b.n_priv = (uint32_t)step_input_width(t, S);
  c.out0 = b.alloc(b.len_z);   // wires 1..len_z
  c.in0 = b.alloc(b.len_z);
  c.priv0 = b.alloc(b.n_priv);
  b.zout.assign(b.len_z, ZOut{ValRef{REF_CONST_ZERO, 0}, 0});
  const uint32_t P = c.priv0;

  switch (t) {
    case T_HASH: {
      c.head_tail_to_output(c.zin(0), P, w, 0);
      break;
    }
    case T_GRAYSCALE: {
      auto d0 = g.decompress_row(P, w), d1 = g.decompress_row(P + w, w);
      g.lane_group(10 * w, 10 * w, 1, {d0, d1}, [&](LaneCtx& L) {
        LV r = L.byte(0, 0, 0), gg = L.byte(0, 0, 1), bb = L.byte(0, 0, 2);
        LV inter = L.add(L.add(L.muli(r, 299), L.muli(gg, 587)), L.muli(bb, 114));
        LV gray = L.muli(L.byte(1, 0, 0), 1000);
        LV k = L.imm(1000);
        L.assert_less_eq(18, L.sub(inter, gray), k);
        L.assert_less_eq(18, L.sub(gray, inter), k);
      });
      c.head_tail_to_output(c.zin(0), P, w, 0);
      c.head_tail_to_output(c.zin(1), P + w, w, 1);
      break;
    }
    case T_CONTRAST: case T_BRIGHTNESS: {
      auto d0 = g.decompress_row(P, w), d1 = g.decompress_row(P + w, w);
      g.lane_group(30 * w, 10 * w, 3, {d0, d1}, [&](LaneCtx& L) {
        LV o = L.byte(0, 0, 3), tr = L.byte(1, 0, 3), f = L.zin(2);
        LV adj[4];
        for (int k = 0; k < 4; k++) {  // the quadratic expression is assigned to four signals: four constraints
          if (t == T_CONTRAST) adj[k] = L.addi(L.mul(L.addi(o, -128), f), 1280);
          else adj[k] = L.mul(f, o);
        }
        LV zero = L.imm(0), top = L.imm(2550), ten = L.imm(10);
        LV neg = L.less_eq(13, adj[0], L.sub(zero, adj[1]));   // adjusted <= -adjusted
        LV big = L.less_eq(13, top, adj[2]);                   // 2550 <= adjusted
        LV gt = L.mux(big, adj[3], top);
        LV fin = L.mux(neg, gt, zero);
        LV t10 = L.muli(tr, 10);
        L.assert_less_eq(13, L.sub(fin, t10), ten);
        L.assert_less_eq(13, L.sub(t10, fin), ten);
      });
      c.head_tail_to_output(c.zin(0), P, w, 0);
      c.head_tail_to_output(c.zin(1), P + w, w, 1);
      c.passthrough_output(2);
      break;
    }
    case T_BLUR: case T_SHARPNESS: {
      std::vector<Gadgets::Decomp> d;
      for (int k = 0; k < 4; k++) d.push_back(g.decompress_row(P + k * w, w));
      g.lane_group(30 * w, 10 * w, 3, d, [&](LaneCtx& L) {
        LV tr = L.byte(3, 0, 3);
        if (t == T_BLUR) {
          LV conv = L.imm(0);
          for (int m = 0; m < 3; m++) for (int n = 0; n < 3; n++) conv = L.add(conv, L.byte(m, n - 1, 3));
          LV t9 = L.muli(tr, 9), nine = L.imm(9);
          L.assert_less_eq(13, L.sub(conv, t9), nine);
          L.assert_less_eq(13, L.sub(t9, conv), nine);
        } else {
          LV conv = L.muli(L.byte(1, 0, 3), 5);
          conv = L.sub(conv, L.byte(0, 0, 3)); conv = L.sub(conv, L.byte(1, -1, 3));
          conv = L.sub(conv, L.byte(1, 1, 3)); conv = L.sub(conv, L.byte(2, 0, 3));
          LV zero = L.imm(0), top = L.imm(255), one = L.imm(1);
          LV neg = L.less_eq(12, conv, L.sub(zero, conv));
          LV big = L.less_eq(12, top, conv);
          LV gt = L.mux(big, conv, top);
          LV fin = L.mux(neg, gt, zero);
          L.assert_less_eq(9, L.sub(fin, tr), one);
          L.assert_less_eq(9, L.sub(tr, fin), one);
        }
      });
      // UpdateIVCStateConv(3, w): z = [orig, tran, common0, common1]
      FV rh[3];
      for (int i = 0; i < 3; i++) {
        g.begin_chain(0);
        rh[i] = g.array_hash(c.row_fv(P + i * w, w), i >= 1 ? c.out_wire(1 + i) : 0);  // new.common[i-1] <== row_hash[i]
        if (i >= 1) b.zout[1 + i] = ZOut{ValRef{REF_JOB, (uint32_t)b.jobs.size() - 1}, 0};
      }
      for (int i = 0; i < 2; i++) {
        // fresh = IsZero(old.common[i]);  old.common[i] === row_hash[i] * (1 - fresh)
        const uint32_t inv = b.alloc(2), out = inv + 1;
        FV in = c.zin(2 + i);
        b.enforce(in.lc, LC::wire(inv), LC::constant(Fe::one()) - LC::wire(out));   // out <== -in*inv + 1
        b.enforce(in.lc, LC::wire(out), LC());                                      // in * out === 0
        FieldOp f; memset(&f, 0, sizeof(f)); f.op = FOP_ISZERO; f.wire = inv; f.a = in.ref;
        b.fops.push_back(f);
        b.enforce(rh[i].lc, LC::constant(Fe::one()) - LC::wire(out), in.lc);
      }
      g.begin_chain(1);
      g.pair_hash(c.zin(0), rh[1], c.out_wire(0));
      b.zout[0] = ZOut{ValRef{REF_JOB, (uint32_t)b.jobs.size() - 1}, 0};
      c.head_tail_to_output(c.zin(1), P + 3 * w, w, 1);
      break;
    }
    case T_RESIZE: {
      const int w2 = S.width2, ri = S.rows_in, ro = S.rows_out;
      std::vector<Gadgets::Decomp> d;
      for (int k = 0; k < ri; k++) d.push_back(g.decompress_row(P + k * w, w));
      for (int k = 0; k < ro; k++) d.push_back(g.decompress_row(P + ri * w + k * w2, w2));
      for (int i = 0; i < ro; i++) {
        g.lane_group(30 * w2, 10 * w2, 3, d, [&](LaneCtx& L) {
          LV a = L.add(L.byte(i, 0, 3, 2), L.byte(i, 1, 3, 2));
          LV bb = L.add(L.byte(i + 1, 0, 3, 2), L.byte(i + 1, 1, 3, 2));
          LV tr = L.byte(ri + i, 0, 3);
          if (ri == 3) {  // reference 3 -> 2 relation
            const int wt = (i % 2 == 0) ? 2 : 1;
            LV summ = L.add(L.muli(a, wt), L.muli(bb, 3 - wt));
            LV t6 = L.muli(tr, 6), six = L.imm(6);
            L.assert_less_eq(12, L.sub(summ, t6), six);
            L.assert_less_eq(12, L.sub(t6, summ), six);
          } else {        // 2 -> 1 extension for 4K/8K: |a+b+c+d - 4t| <= 4
            LV summ = L.add(a, bb);
            LV t4 = L.muli(tr, 4), four = L.imm(4);
            L.assert_less_eq(12, L.sub(summ, t4), four);
            L.assert_less_eq(12, L.sub(t4, summ), four);
          }
        });
      }
      // hash chains: orig rows then resized rows, each row = ArrayHasher (phase A) + PairHasher (phase B)
      std::vector<FV> rh;
      for (int k = 0; k < ri; k++) { g.begin_chain(0); rh.push_back(g.array_hash(c.row_fv(P + k * w, w))); }
      for (int k = 0; k < ro; k++) { g.begin_chain(0); rh.push_back(g.array_hash(c.row_fv(P + ri * w + k * w2, w2))); }
      g.begin_chain(1);
      FV h = c.zin(0);
      for (int k = 0; k < ri; k++) h = g.pair_hash(h, rh[k], k == ri - 1 ? c.out_wire(0) : 0);
      b.zout[0] = ZOut{ValRef{REF_JOB, (uint32_t)b.jobs.size() - 1}, 0};
      g.begin_chain(1);
      h = c.zin(1);
      for (int k = 0; k < ro; k++) h = g.pair_hash(h, rh[ri + k], k == ro - 1 ? c.out_wire(1) : 0);
      b.zout[1] = ZOut{ValRef{REF_JOB, (uint32_t)b.jobs.size() - 1}, 0};
      break;
    }
    case T_REDACT: {
      g.begin_chain(0);
      FV bh = g.array_hash(c.row_fv(P, w));
      g.begin_chain(1);
      FV c0 = g.pair_hash(c.zin(1), bh);
      g.begin_chain(1);
      FV c1 = g.pair_hash(c.zin(1), fv_zero());
      g.begin_chain(1);
      g.pair_hash(c.zin(0), bh, c.out_wire(0));
      b.zout[0] = ZOut{ValRef{REF_JOB, (uint32_t)b.jobs.size() - 1}, 0};
      // selector: out = (c1 - c0) * s + c0, out is step_out.tran_hash itself
      FV s = fv_wire(P + w);
      b.enforce(c1.lc - c0.lc, s.lc, LC::wire(c.out_wire(1)) - c0.lc);
      FieldOp f; memset(&f, 0, sizeof(f)); f.op = FOP_MUX; f.wire = c.out_wire(1); f.bound = 1; f.a = s.ref; f.b = c0.ref; f.c = c1.ref;
      b.fops.push_back(f);
      b.zout[1] = ZOut{ValRef{REF_FOP, (uint32_t)b.fops.size() - 1}, 0};
      break;
    }
    case T_CROP: {
      // CropHash(widthOrig = w, widthCrop = S.width2, heightCrop = S.crop_height), followed literally
      // (circuits/src/crop_step.circom:9-83, MultiplexerCrop :85-120; SURVEY.md F6: x = info bits 0-11, y = 12-23,
      // row_index = 24-35, and `step_out.info <== step_in.info + 1`).  The cropped row's hash depends on step_in only through
      // `info`, which is predictable (info_0 + i): its chain is phase 2 (after the early field ops, before the state hashes), so
      // the prover can compute it for all rows ahead of the IVC state chain (prover_internal.hpp, fold_prepare).
      const int wc = S.width2, H = S.crop_height, W10 = 10 * w, C10 = 10 * wc;
      auto d0 = g.decompress_row(P, w);
      // --- CropInfoDecompressor + the row-range test: one lane
      LC x_lc, y_lc, ri_lc, s_lc; uint32_t s_wire = 0;
      g.lane_group(1, 1, 1, {}, [&](LaneCtx& L) {
        LV info = L.zin(2);
        std::vector<LC> bits = L.num2bits(info, 36);
        LC xl, yl, rl;
        if (!L.counting) for (int i = 0; i < 12; i++) { xl = xl + bits[i].scaled(fe_pow2(i)); yl = yl + bits[12 + i].scaled(fe_pow2(i)); rl = rl + bits[24 + i].scaled(fe_pow2(i)); }
        LV y = L.andi(L.shri(info, 12), 0xfff, yl), ri = L.andi(L.shri(info, 24), 0xfff, rl);
        LV gte = L.less_eq(12, y, ri);                                   // GreaterEqThan(12)(row_index, y)
        LV lt = L.less_eq(12, ri, L.addi(y, H - 1));                     // LessThan(12)(row_index, y + heightCrop)
        LV sel = L.mul(gte, lt);                                         // selector.s <== gte.out * lt.out
        if (!L.counting) { x_lc = xl; y_lc = yl; ri_lc = rl; s_lc = sel.lc; s_wire = sel.lc.t[0].w; }
      });
      // --- Decoder(W10): out[k] <-- (x == k); out[k]*(x - k) === 0; sum(out) === 1 substitutes out[0]
      std::vector<LC> dec(W10);
      g.lane_group(W10 - 1, W10 - 1, 1, {}, [&](LaneCtx& L) {
        LV info = L.zin(2);
        LV xv = L.andi(info, 0xfff, x_lc);
        LV k = L.lane_index(1);
        LV o = L.eq_hint(xv, k);
        L.enforce(o.lc, xv.lc - k.lc, LC());
        if (!L.counting) dec[L.lane + 1] = o.lc;
      });
      { LC sum; for (int k = 1; k < W10; k++) sum.add_in_place(dec[k]); dec[0] = LC::constant(Fe::one()) - sum; b.enforce(dec[0], x_lc, LC()); }
      // --- EscalarProduct rows: out[h] = sum_k inp[k+h] * dec[k]   (one constraint per non-constant product)
      std::vector<LC> mux_out(C10);
      for (int h = 0; h < C10; h++) {
        g.lane_group(W10 - h, W10 - h, 1, {d0}, [&](LaneCtx& L) {
          LV inp = L.byte(0, h, 0);
          LV info = L.zin(2);
          LV xv = L.andi(info, 0xfff, LC());
          LV k = L.lane_index(0);
          LV dv = L.eq_value(xv, k, L.counting ? LC() : dec[L.lane]);
          LV aux = L.mul(inp, dv);
          if (!L.counting) mux_out[h].add_in_place(aux.lc);
        });
      }
      // --- CompressorCrop: Num2Bits(24) of every selected value
      g.lane_group(C10, C10, 1, {d0}, [&](LaneCtx& L) {
        LV info = L.zin(2);
        LV xv = L.andi(info, 0xfff, LC());
        LV v = L.byte_rel(0, xv, L.counting ? LC() : mux_out[L.lane]);
        L.num2bits(v, 24);
      });
      // --- hash of the cropped row, conditional chaining, state update
      std::vector<FV> cropped;
      for (int i = 0; i < wc; i++) {
        LC lc;
        for (int j = 0; j < 10; j++) lc = lc + mux_out[10 * i + j].scaled(fe_pow2(24 * j));
        FieldOp f; memset(&f, 0, sizeof(f)); f.op = FOP_LC; f.early = 1;
        f.a = ValRef{0, (uint32_t)b.lc_terms.size()}; f.b = ValRef{0, (uint32_t)lc.t.size()};
        for (auto& tm : lc.t) b.lc_terms.push_back(LcTerm{tm.w, b.coef_id(tm.c)});
        b.fops.push_back(f);
        cropped.push_back(FV{lc, ValRef{REF_FOP, (uint32_t)b.fops.size() - 1}});
      }
      g.begin_chain(2);
      FV th = g.array_hash(cropped);
      g.begin_chain(1);
      FV c1 = g.pair_hash(c.zin(1), th);
      FV c0 = c.zin(1);
      b.enforce(c1.lc - c0.lc, s_lc, LC::wire(c.out_wire(1)) - c0.lc);      // Mux1, out is step_out.base.tran_hash
      // the selector as an (early) field-op value, so that the host's state chain can read it from the ahead-of-time pass
      ValRef s_ref;
      { FieldOp f; memset(&f, 0, sizeof(f)); f.op = FOP_LC; f.early = 1; f.a = ValRef{0, (uint32_t)b.lc_terms.size()}; f.b = ValRef{0, 1};
        b.lc_terms.push_back(LcTerm{s_wire, b.coef_id(Fe::one())}); b.fops.push_back(f); s_ref = ValRef{REF_FOP, (uint32_t)b.fops.size() - 1}; }
      { FieldOp f; memset(&f, 0, sizeof(f)); f.op = FOP_MUX; f.wire = c.out_wire(1); f.bound = 1; f.a = s_ref; f.b = c0.ref; f.c = c1.ref;
        b.fops.push_back(f); b.zout[1] = ZOut{ValRef{REF_FOP, (uint32_t)b.fops.size() - 1}, 0}; }
      c.head_tail_to_output(c.zin(0), P, w, 0);
      b.enforce(LC::constant(Fe::one()), LC::wire(c.in0 + 2) + LC::constant(Fe::one()), LC::wire(c.out_wire(2)));   // info + 1 (linear)
      b.n_linear++;
      b.zout[2] = ZOut{ValRef{REF_ZIN, 2}, 1};
      (void)y_lc; (void)ri_lc;
      break;
    }
    default:
      throw std::runtime_error("unknown transformation");
  }
  group_boolean_rows_first(b);
  return cbp;
}

}  // namespace cb
}  // namespace vz
