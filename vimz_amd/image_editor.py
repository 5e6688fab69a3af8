"""Input tooling: image -> packed rows -> VIMz input JSON.

Host-side mirror of the reference's Python input generator (pyvimz/pyvimz/image_editor.py:71-150,
pyvimz/pyvimz/img/ops.py:4-105, pyvimz/pyvimz/img/transformations.py:6-147), restated with integer
numpy arithmetic so that it is bit-exact against the reference's outputs (pinned in
tests/test_oracle_golden.py against the reference's committed PNGs/hashes and against fixtures minted
by importing pyvimz, see tests/golden/make_fixtures.py).

Packing (ops.py:17-31, circuits/src/utils/pixels.circom:15-27): 10 pixels per field element, pixel k
in bits [24k, 24k+24), R in the low byte — i.e. the 30 little-endian bytes of an element are exactly
the raw R,G,B bytes of its 10 pixels.  Grayscale: the value sits in the low byte of each 24-bit slot.
Rows are returned as uint64 limb arrays of shape (rows, width, 4) (canonical little-endian), the
layout the C ABI (include/vimz_hip.h) takes; `rows_to_hex` renders the reference's "0x…" strings.
"""
import json

import numpy as np

PACKING_FACTOR = 10  # vimz/src/lib.rs:10


def compress_by_rows(image):
    """image: (H, W, 3) or (H, W) uint8 -> (H, ceil(W/10), 4) uint64 limbs (ops.py:4-33)."""
    img = np.asarray(image, dtype=np.uint8)
    h, w = img.shape[:2]
    n = (w + PACKING_FACTOR - 1) // PACKING_FACTOR
    buf = np.zeros((h, n * PACKING_FACTOR, 3), dtype=np.uint8)
    if img.ndim == 2:
        buf[:, :w, 0] = img
    else:
        buf[:, :w, :] = img[:, :, :3]
    out = np.zeros((h, n, 32), dtype=np.uint8)
    out[:, :, :30] = buf.reshape(h, n, 30)
    return out.view("<u8").reshape(h, n, 4)


def compress_by_blocks(image, block=40):
    """(H, W[,3]) -> (blocks, block*block/10, 4) limbs, blocks row-major (ops.py:36-70)."""
    img = np.asarray(image, dtype=np.uint8)
    h, w = img.shape[:2]
    blocks = []
    for br in range(0, h, block):
        for bc in range(0, w, block):
            sub = img[br:br + block, bc:bc + block]
            blocks.append(compress_by_rows(sub).reshape(-1, 4))
    return np.stack(blocks)


def rows_to_hex(rows):
    """(R, W, 4) limbs -> list of list of '0x…' strings, formatted as the reference does (ops.py:21-27)."""
    rows = np.asarray(rows, dtype=np.uint64)
    out = []
    for r in rows:
        line = []
        for e in r:
            v = int(e[0]) | int(e[1]) << 64 | int(e[2]) << 128 | int(e[3]) << 192
            line.append("0x" + format(v, "060x"))
        out.append(line)
    return out


def hex_to_rows(rows_hex):
    """list of list of hex strings -> (R, W, 4) limbs (vimz/src/input.rs:95-105)."""
    R = len(rows_hex)
    W = len(rows_hex[0]) if R else 0
    out = np.zeros((R, W, 4), dtype=np.uint64)
    for i, r in enumerate(rows_hex):
        for j, s in enumerate(r):
            v = int(s, 16)
            for k in range(4):
                out[i, j, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


# ---------------------------------------------------------------- transformations (integer restatements)

def convert_to_grayscale(img):
    """Pillow's 'L' conversion (transformations.py:40-41): (19595 R + 38470 G + 7471 B + 32768) >> 16."""
    a = np.asarray(img, dtype=np.uint32)
    return ((19595 * a[..., 0] + 38470 * a[..., 1] + 7471 * a[..., 2] + 32768) >> 16).astype(np.uint8)


def adjust_contrast(img, factor):
    """transformations.py:44-56, same float64 expression: ((c - 128.0) * f + 128.0).clip(0, 255) truncated to uint8.
    (Equals clip((c-128)*f10 + 1280, 0, 2550) // 10 on the reference's samples; the float form is kept so every
    factor rounds exactly as the reference does.)"""
    a = np.asarray(img)[..., :3].astype(np.float64)
    mean = float(int(128 * 1000)) / 1000
    return ((a - mean) * factor + mean).clip(0, 255).astype(np.uint8)


def adjust_brightness(img, factor):
    """transformations.py:59-63: clip(float(c) * f, 0, 255) truncated to uint8 (float64, as the reference)."""
    a = np.asarray(img)[..., :3].astype(float)
    return np.clip(a * factor, 0, 255).astype(np.uint8)


def _conv3(chan, kernel, weight):
    """ops.py:73-105 conv2d with zero padding, floor division, clamp to [0,255]."""
    h, w = chan.shape
    p = np.zeros((h + 2, w + 2), dtype=np.int64)
    p[1:-1, 1:-1] = chan
    acc = np.zeros((h, w), dtype=np.int64)
    for m in range(3):
        for n in range(3):
            if kernel[m][n]:
                acc += kernel[m][n] * p[m:m + h, n:n + w]
    return np.clip(acc // weight, 0, 255).astype(np.uint8)


def blur_image(img):
    a = np.asarray(img, dtype=np.int64)[..., :3]
    k = [[1, 1, 1], [1, 1, 1], [1, 1, 1]]
    return np.dstack([_conv3(a[..., c], k, 9) for c in range(3)])


def sharpen_image(img):
    a = np.asarray(img, dtype=np.int64)[..., :3]
    k = [[0, -1, 0], [-1, 5, -1], [0, -1, 0]]
    return np.dstack([_conv3(a[..., c], k, 1) for c in range(3)])


def crop_image(img, x, y, new_w, new_h):
    return np.asarray(img)[y:y + new_h, x:x + new_w]


def resize_image(img, new_h, new_w):
    """transformations.py:97-147.  HD->SD: rows 3->2 with weights (2/3,1/3)/(1/3,2/3), result/2 truncated to uint8;
    otherwise 2x2 box with weight 1/2 then /2.  The reference computes in float64; a*w+b*w+c*(1-w)+d*(1-w) is
    reproduced with the same float expression so the truncation matches exactly."""
    a = np.asarray(img)[..., :3].astype(np.float64)
    h, w, _ = a.shape
    xr, yr = float(w) / float(new_w), float(h) / float(new_h)
    ii = np.arange(new_h); jj = np.arange(new_w)
    yl = (ii * yr).astype(np.int64); xl = (jj * xr).astype(np.int64)
    A = a[yl][:, xl]; B = a[yl][:, xl + 1]; Cc = a[yl + 1][:, xl]; D = a[yl + 1][:, xl + 1]
    if h == 720:
        wt = np.where(ii % 2 == 0, 2.0, 1.0) / 3
        wt = wt[:, None, None]
        summ = A * wt + B * wt + Cc * (1 - wt) + D * (1 - wt)
    else:
        wt = 0.5
        summ = A * wt + B * wt + Cc * wt + D * wt
    return (summ / 2).astype(np.uint8)


def random_image_redaction(img, block=40):
    """Checkerboard redaction (transformations.py:71-94): block (by,bx) is zeroed when by+bx is odd."""
    a = np.array(img)
    h, w = a.shape[:2]
    ind = []
    out = a.copy()
    for by in range(h // block):
        for bx in range(w // block):
            red = (by + bx) % 2 == 1
            ind.append(1 if red else 0)
            if red:
                out[by * block:(by + 1) * block, bx * block:(bx + 1) * block] = 0
    return out, ind


SIZE_MAP = {"sd": (640, 480), "hd": (1280, 720), "fhd": (1920, 1080)}


def build_input(operation, image, factor=None, x=None, y=None, crop_size=None, resize_to=None):
    """Mirror of image_editor.main (image_editor.py:71-150): returns the VIMzInput as a dict of limb arrays
    {"original": (R,W,4), "transformed": (R',W',4) or None, "factor"|"info"|"redact": ...}."""
    img = np.asarray(image)[..., :3]
    out = {"original": compress_by_rows(img), "transformed": None}
    if operation == "hash":
        pass
    elif operation == "grayscale":
        out["transformed"] = compress_by_rows(convert_to_grayscale(img))
    elif operation in ("brightness", "contrast"):
        fn = adjust_brightness if operation == "brightness" else adjust_contrast
        out["transformed"] = compress_by_rows(fn(img, factor))
        out["factor"] = int(factor * 10)
    elif operation in ("sharpness", "blur"):
        fn = sharpen_image if operation == "sharpness" else blur_image
        zeros = np.zeros((1, out["original"].shape[1], 4), dtype=np.uint64)
        out["original"] = np.concatenate([zeros, out["original"], zeros])
        out["transformed"] = compress_by_rows(fn(img))
    elif operation == "crop":
        w, h = SIZE_MAP[crop_size.lower()]
        out["transformed"] = compress_by_rows(crop_image(img, x, y, w, h))
        out["info"] = x * 2 ** 24 + y * 2 ** 12
    elif operation == "redact":
        out["original"] = compress_by_blocks(img)
        tr, ind = random_image_redaction(img)
        out["redact"] = ind
        out["transformed"] = compress_by_blocks(tr)
    elif operation == "resize":
        w, h = resize_to
        out["transformed"] = compress_by_rows(resize_image(img, h, w))
    else:
        raise ValueError(f"unknown operation {operation}")
    return out


def dump_json(inp, path):
    """Write the reference's JSON schema (vimz/src/input.rs:9-62)."""
    d = {"original": rows_to_hex(inp["original"])}
    if inp.get("transformed") is not None:
        d["transformed"] = rows_to_hex(inp["transformed"])
    for k in ("factor", "info"):
        if k in inp:
            d[k] = int(inp[k])
    if "redact" in inp:
        d["redact"] = ["0x1" if r else "0x0" for r in inp["redact"]]
    with open(path, "w") as fp:
        json.dump(d, fp, indent=4)


def load_json(path):
    with open(path) as fp:
        d = json.load(fp)
    out = {"original": hex_to_rows(d["original"]), "transformed": hex_to_rows(d["transformed"]) if d.get("transformed") else None}
    for k in ("factor", "info"):
        if k in d:
            out[k] = int(d[k])
    if "redact" in d:
        out["redact"] = [int(s, 16) for s in d["redact"]]
    return out
