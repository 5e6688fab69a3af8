// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).
//
// Semantic restatement of the nine VIMz step circuits: given the IVC state z_i and the
// per-step private inputs, compute z_{i+1} and whether every constraint of the step
// relation is satisfiable (i.e. whether the reference's witness generator would succeed
// and the R1CS would hold).  Sources followed, line by line:
//   circuits/src/contrast_step.circom:10-98      ContrastHash / ContrastChecker
//   circuits/src/brightness_step.circom:7-105    BrightnessHash / BrightnessChecker
//   circuits/src/grayscale_step.circom:8-66      GrayScaleHash / GrayscaleChecker
//   circuits/src/blur_step.circom:6-73           Blur / ConvolveBlur
//   circuits/src/sharpness_step.circom:6-103     Sharpen / ConvolveSharpen
//   circuits/src/utils/convolution_step.circom:10-48   UnwrapAndExtend (zero extension)
//   circuits/src/resize_step.circom:10-112       ResizeHash
//   circuits/src/crop_step.circom:9-83 + :85-120 CropHash / MultiplexerCrop (literal)
//   circuits/src/redact_step.circom:7-26         RedactHash
//   circuits/nova_snark/hash_step.circom:6-16    NovaHash
//   circuits/src/utils/state.circom:11-79        IVC state updates
//   circuits/src/utils/pixels.circom:6-141       (de)compressors
// circomlib comparators/mux semantics: SURVEY.md Appendix B.
// State layouts (z): vimz/src/transformation.rs:25-50.
#pragma once
#include <vector>
#include "poseidon.hpp"

namespace orc {

enum Transformation { T_BLUR = 0, T_BRIGHTNESS, T_CONTRAST, T_CROP, T_GRAYSCALE, T_HASH, T_REDACT, T_RESIZE, T_SHARPNESS };

struct StepShape {  // geometry of one step; HD defaults from circuits/nova_snark/*.circom `component main`
  int width;        // packed elements per original row (128 HD, 384 4K, 768 8K)
  int width2;       // resize: packed elements per resized row; crop: widthCrop
  int rows_in;      // resize: rowCountOrig (3 or 2)
  int rows_out;     // resize: rowCountResized (2 or 1)
  int crop_height;  // crop: heightCrop
};

static inline int ivc_state_len(int t) {  // transformation.rs:43-50
  switch (t) {
    case T_BLUR: case T_SHARPNESS: return 4;
    case T_BRIGHTNESS: case T_CONTRAST: case T_CROP: return 3;
    case T_GRAYSCALE: case T_REDACT: case T_RESIZE: return 2;
    default: return 1;
  }
}
static inline int step_input_width(int t, const StepShape& s) {  // transformation.rs:53-66 generalised in width
  switch (t) {
    case T_BLUR: case T_SHARPNESS: return 4 * s.width;
    case T_BRIGHTNESS: case T_CONTRAST: case T_GRAYSCALE: return 2 * s.width;
    case T_CROP: case T_HASH: return s.width;
    case T_REDACT: return s.width + 1;  // width = 160 block elements
    default: return s.rows_in * s.width + s.rows_out * s.width2;
  }
}

typedef __int128 i128;

struct Packed {  // canonical 256-bit integer of a packed element
  u64 l[4];
  bool fits240() const { return (l[3] >> 48) == 0; }
  unsigned byte(int k) const { return (unsigned)((l[k / 8] >> (8 * (k % 8))) & 0xff); }  // k-th byte
};

// LessEqThan(n)(a,b) of circomlib: Num2Bits(n+1)(a + 2^n - (b+1)); out = 1 - bit n.
static inline int less_eq(int n, i128 a, i128 b, bool& valid) {
  i128 v = a + ((i128)1 << n) - (b + 1);
  if (v < 0 || v >= ((i128)1 << (n + 1))) { valid = false; return 0; }
  return (v >> n) & 1 ? 0 : 1;
}
static inline int less_than(int n, i128 a, i128 b, bool& valid) { return less_eq(n, a, b - 1, valid); }

struct StepOut { bool ok; std::vector<BnFr> z; };

// in: canonical 4-limb integers, step_input_width of them, in the reference's flattened order
// (row_orig rows first, then row_tran rows; redact: block then indicator).
static inline StepOut step_eval(int t, const StepShape& S, const std::vector<BnFr>& z, const u64* in) {
  StepOut R; R.ok = true; R.z = z;
  const int w = S.width;
  auto P = [&](int idx) { Packed p; memcpy(p.l, in + 4 * idx, 32); return p; };
  auto F = [&](int idx) { return BnFr::from_canonical(in + 4 * idx); };
  auto row_fe = [&](int start, int len) { std::vector<BnFr> v(len); for (int i = 0; i < len; i++) v[i] = F(start + i); return v; };
  auto need240 = [&](int start, int len) { for (int i = 0; i < len; i++) if (!P(start + i).fits240()) R.ok = false; };
  // pixel (row start index, pixel x, colour)
  auto px = [&](int start, int x, int c) -> i128 { return (i128)P(start + x / 10).byte((x % 10) * 3 + c); };

  switch (t) {
    case T_HASH: {
      auto r = row_fe(0, w);
      R.z[0] = head_tail_hash(z[0], r.data(), w);
      break;
    }
    case T_CONTRAST: case T_BRIGHTNESS: case T_GRAYSCALE: {
      need240(0, 2 * w);
      u64 fc[4] = {0, 0, 0, 0};
      if (t != T_GRAYSCALE) z[2].to_canonical(fc);
      i128 f = (i128)fc[0];
      if (t != T_GRAYSCALE && (fc[1] | fc[2] | fc[3] || fc[0] >> 40)) R.ok = false;  // beyond comparator range anyway
      for (int x = 0; x < 10 * w && R.ok; x++) {
        if (t == T_GRAYSCALE) {
          i128 inter = 299 * px(0, x, 0) + 587 * px(0, x, 1) + 114 * px(0, x, 2);
          i128 g = px(w, x, 0);  // DecompressorGray: low byte of each 24-bit slot
          if (!less_eq(18, inter - 1000 * g, 1000, R.ok)) R.ok = false;
          if (!less_eq(18, 1000 * g - inter, 1000, R.ok)) R.ok = false;
          continue;
        }
        for (int c = 0; c < 3; c++) {
          i128 o = px(0, x, c), tr = px(w, x, c);
          i128 adj = (t == T_CONTRAST) ? (o - 128) * f + 1280 : f * o;
          int neg = less_eq(13, adj, -adj, R.ok);     // adj <= 0
          int big = less_eq(13, 2550, adj, R.ok);     // 2550 <= adj
          i128 fin = neg ? 0 : (big ? 2550 : adj);
          if (!less_eq(13, fin - 10 * tr, 10, R.ok)) R.ok = false;
          if (!less_eq(13, 10 * tr - fin, 10, R.ok)) R.ok = false;
        }
      }
      auto ro = row_fe(0, w), rt = row_fe(w, w);
      R.z[0] = head_tail_hash(z[0], ro.data(), w);
      R.z[1] = head_tail_hash(z[1], rt.data(), w);
      break;
    }
    case T_BLUR: case T_SHARPNESS: {
      need240(0, 4 * w);
      const int W10 = 10 * w;
      auto o = [&](int m, int xe, int c) -> i128 {  // extended coordinate: xe in [0, W10+2), pixel = xe-1
        int x = xe - 1;
        if (x < 0 || x >= W10) return 0;
        return px(m * w, x, c);
      };
      for (int x = 0; x < W10 && R.ok; x++) for (int c = 0; c < 3; c++) {
        i128 tr = px(3 * w, x, c);
        if (t == T_BLUR) {
          i128 conv = 0;
          for (int m = 0; m < 3; m++) for (int n = 0; n < 3; n++) conv += o(m, x + n, c);
          if (!less_eq(13, conv - 9 * tr, 9, R.ok)) R.ok = false;
          if (!less_eq(13, 9 * tr - conv, 9, R.ok)) R.ok = false;
        } else {
          i128 conv = 5 * o(1, x + 1, c) - o(0, x + 1, c) - o(1, x, c) - o(1, x + 2, c) - o(2, x + 1, c);
          int neg = less_eq(12, conv, -conv, R.ok);
          int big = less_eq(12, 255, conv, R.ok);
          i128 fin = neg ? 0 : (big ? 255 : conv);
          if (!less_eq(9, fin - tr, 1, R.ok)) R.ok = false;
          if (!less_eq(9, tr - fin, 1, R.ok)) R.ok = false;
        }
      }
      BnFr rh[3];
      for (int i = 0; i < 3; i++) { auto r = row_fe(i * w, w); rh[i] = array_hash(r.data(), w); }
      for (int i = 0; i < 2; i++) {  // old.common[i] === row_hash[i] * (1 - IsZero(old.common[i]))
        if (!z[2 + i].is_zero() && z[2 + i] != rh[i]) R.ok = false;
      }
      auto rt = row_fe(3 * w, w);
      R.z[0] = pair_hash(z[0], rh[1]);
      R.z[1] = head_tail_hash(z[1], rt.data(), w);
      R.z[2] = rh[1]; R.z[3] = rh[2];
      break;
    }
    case T_RESIZE: {
      const int w2 = S.width2, ri = S.rows_in, ro = S.rows_out;
      need240(0, ri * w + ro * w2);
      const int tbase = ri * w;
      for (int i = 0; i < ro && R.ok; i++) for (int j = 0; j < 10 * w2; j++) for (int c = 0; c < 3; c++) {
        i128 a = px(i * w, 2 * j, c) + px(i * w, 2 * j + 1, c);
        i128 b = px((i + 1) * w, 2 * j, c) + px((i + 1) * w, 2 * j + 1, c);
        i128 tr = px(tbase + i * w2, j, c);
        if (ri == 3) {  // reference 3->2 relation (resize_step.circom:78-101)
          i128 wt = (i % 2 == 0) ? 2 : 1;
          i128 summ = a * wt + b * (3 - wt);
          if (!less_eq(12, summ - 6 * tr, 6, R.ok)) R.ok = false;
          if (!less_eq(12, 6 * tr - summ, 6, R.ok)) R.ok = false;
        } else {        // 2->1 extension for 4K/8K (pyvimz transformations.py:130-145): |a+b+c+d - 4t| <= 4
          i128 summ = a + b;
          if (!less_eq(12, summ - 4 * tr, 4, R.ok)) R.ok = false;
          if (!less_eq(12, 4 * tr - summ, 4, R.ok)) R.ok = false;
        }
      }
      BnFr h = z[0];
      for (int i = 0; i < ri; i++) { auto r = row_fe(i * w, w); h = pair_hash(h, array_hash(r.data(), w)); }
      R.z[0] = h;
      h = z[1];
      for (int i = 0; i < ro; i++) { auto r = row_fe(tbase + i * w2, w2); h = pair_hash(h, array_hash(r.data(), w2)); }
      R.z[1] = h;
      break;
    }
    case T_REDACT: {
      auto blk = row_fe(0, w);
      BnFr bh = array_hash(blk.data(), w);
      BnFr red = F(w);
      BnFr c0 = pair_hash(z[1], bh), c1 = pair_hash(z[1], BnFr::zero());
      R.z[0] = pair_hash(z[0], bh);
      R.z[1] = (c1 - c0) * red + c0;  // Mux1, selector NOT constrained boolean (literal)
      break;
    }
    case T_CROP: {
      need240(0, w);
      u64 info[4]; z[2].to_canonical(info);
      if (info[1] | info[2] | info[3] || info[0] >> 36) { R.ok = false; break; }  // Num2Bits(36)
      unsigned x = info[0] & 0xfff, y = (info[0] >> 12) & 0xfff, row_index = (info[0] >> 24) & 0xfff;
      const int W10 = 10 * w, wc = S.width2;
      if ((int)x >= W10) R.ok = false;  // Decoder success === 1
      std::vector<BnFr> cropped(wc);
      for (int i = 0; i < wc; i++) {
        u64 pk[4] = {0, 0, 0, 0};
        for (int j = 0; j < 10; j++) {
          int src = (int)x + i * 10 + j;
          u64 v = (src < W10) ? (u64)P(src / 10).byte((src % 10) * 3) : 0;  // DecompressorCrop = low byte only
          int bit = 24 * j;
          pk[bit / 64] |= v << (bit % 64);
          if (bit % 64 > 56) pk[bit / 64 + 1] |= v >> (64 - bit % 64);
        }
        cropped[i] = BnFr::from_canonical(pk);
      }
      BnFr th = array_hash(cropped.data(), wc);
      bool v = true;
      int ge = less_than(12, y, (i128)row_index + 1, v);            // GreaterEqThan(12)(row_index, y)
      int lt = less_than(12, row_index, (i128)y + S.crop_height, v);  // LessThan(12)
      if (!v) R.ok = false;
      auto ro = row_fe(0, w);
      R.z[0] = head_tail_hash(z[0], ro.data(), w);
      if (ge && lt) R.z[1] = pair_hash(z[1], th);
      R.z[2] = z[2] + BnFr::one();
      break;
    }
  }
  return R;
}

}  // namespace orc
