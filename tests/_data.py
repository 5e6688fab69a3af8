"""Shared helpers for tests: golden data access (tests/golden) and step-input assembly."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as fp:
        return json.load(fp)


def load_image(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, name + ".png")).convert("RGB"))


def rows10(op):
    with open(os.path.join(GOLDEN, f"rows10_{op}.json")) as fp:
        return json.load(fp)


def config_rows(transformation, resolution, n):
    """First `n` step inputs of a BASELINE.json configuration: the reference's sample image img2, upscaled by
    nearest neighbour to 4K (x3) / 8K (x6) — the reference ships no 4K/8K image — cut to the strip the first n steps
    read, run through our image editor and sliced with prepare_step_input (vimz/src/nova_snark_backend/input.rs:57-96).
    Returns (rows (n, n_priv, 4), z0)."""
    from vimz_amd import folding, image_editor as ie
    k = {"HD": 1, "4K": 3, "8K": 6}[resolution]
    a, b = folding.RATIO_TO_LOWER.get(resolution, (1, 1)) if transformation == "resize" else (1, 1)
    need = n * a + 4                                     # source rows read by steps 0..n-1 (conv steps look one row ahead)
    need += (-need) % (2 * k)
    src = load_image("img2")[: (need + k - 1) // k + 1]
    img = np.repeat(np.repeat(src, k, axis=0), k, axis=1)[:need]
    kw = {"factor": 1.4} if transformation in ("contrast", "brightness") else {}
    if transformation == "resize":
        kw["resize_to"] = (img.shape[1] * b // a, img.shape[0] * b // a)
    inp = ie.build_input(transformation, img, **kw)
    rows = np.stack([folding.prepare_step_input(i, transformation, inp, resolution) for i in range(n)])
    return rows, folding.ivc_initial_state(transformation, inp)
