#!/usr/bin/env python3
"""Mint the golden fixtures under tests/golden/ from the reference checkout.

Runs ONLY in the build container (needs /root/reference); the outputs are committed, this script
documents how they were made.  Nothing here travels to the GPU box except the data files it writes.

  kat.json            the reference's own committed known answers:
                        - marketplace/image-data/*.hash                       (image running hashes)
                        - marketplace/proofs/*.proof decoded (steps, z0, z_final) with the layout of
                          marketplace/vimz_marketplace_sdk/artifacts.py:19-42
                        - circuits/src/utils/decompress_input.json             (packing KAT)
                        - circuits/nova_snark/circuit_parameters.csv + redact_step.compile_log (sizes)
  img1.png, img2.png  the reference's sample images (marketplace/image-data; img1 == source_image/HD.png)
  rows10_<op>.json    first DEMO_STEPS=10 steps of input produced by IMPORTING the reference's Python
                      (pyvimz.img.ops.compress_by_rows + pyvimz.img.transformations.*) on those images.
"""
import csv
import json
import os
import shutil
import sys

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "pyvimz"))
from pyvimz.img.ops import compress_by_rows, compress_by_blocks  # noqa: E402
from pyvimz.img import transformations as T  # noqa: E402

DEMO_STEPS = 10


def decode_proof(raw):
    body = raw[4:]
    steps = int.from_bytes(body[0:32], "big")
    proof_len = 32 * 25
    state_len = (len(body) - proof_len - 32) // 2
    z0 = [int.from_bytes(body[s:s + 32], "big") for s in range(32, 32 + state_len, 32)]
    zf = [int.from_bytes(body[s:s + 32], "big") for s in range(32 + state_len, 32 + 2 * state_len, 32)]
    return {"selector": raw[:4].hex(), "steps": steps, "z0": [str(v) for v in z0], "z_final": [str(v) for v in zf],
            "proof_words": [str(int.from_bytes(body[s:s + 32], "big")) for s in range(len(body) - proof_len, len(body), 32)]}


def main():
    kat = {"hashes": {}, "proofs": {}}
    d = os.path.join(REF, "marketplace/image-data")
    for f in sorted(os.listdir(d)):
        if f.endswith(".hash"):
            kat["hashes"][f[:-5]] = open(os.path.join(d, f)).read().strip()
    d = os.path.join(REF, "marketplace/proofs")
    for f in sorted(os.listdir(d)):
        if f.endswith(".proof"):
            kat["proofs"][f[:-6]] = decode_proof(open(os.path.join(d, f), "rb").read())
    kat["decompress"] = json.load(open(os.path.join(REF, "circuits/src/utils/decompress_input.json")))
    sizes = {}
    with open(os.path.join(REF, "circuits/nova_snark/circuit_parameters.csv")) as fp:
        for row in csv.DictReader(fp):
            sizes[row["File Name"]] = {"constraints": int(row["Non-Linear Constraints"]), "wires": int(row["Wires"]),
                                       "public_inputs": int(row["Public Inputs"]), "private_inputs": int(row["Private Inputs"]),
                                       "public_outputs": int(row["Public Outputs"])}
    sizes["redact"] = {"constraints": 8758, "wires": 8903, "public_inputs": 2, "private_inputs": 161, "public_outputs": 2}
    kat["circuit_sizes"] = sizes
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)

    for name in ("img1", "img2"):
        shutil.copyfile(os.path.join(REF, "marketplace/image-data", name + ".png"), os.path.join(HERE, name + ".png"))
        os.chmod(os.path.join(HERE, name + ".png"), 0o644)

    img1 = Image.open(os.path.join(HERE, "img1.png")).convert("RGB")
    img2 = Image.open(os.path.join(HERE, "img2.png")).convert("RGB")
    N = DEMO_STEPS

    def dump(op, obj):
        json.dump(obj, open(os.path.join(HERE, f"rows10_{op}.json"), "w"))

    o1 = compress_by_rows(img1)
    dump("hash", {"image": "img1", "original": o1[:N]})
    dump("grayscale", {"image": "img1", "original": o1[:N], "transformed": compress_by_rows(T.convert_to_grayscale(img1))[:N]})
    o2 = compress_by_rows(img2)
    dump("contrast", {"image": "img2", "factor": int(1.4 * 10), "original": o2[:N],
                      "transformed": compress_by_rows(T.adjust_contrast(img2, 1.4))[:N]})
    dump("brightness", {"image": "img1", "factor": int(1.4 * 10), "original": o1[:N],
                        "transformed": compress_by_rows(T.adjust_brightness(img1, 1.4))[:N]})
    # conv2d in the reference is a per-pixel Python loop: run it on the top slab only; output rows 0..N-1 depend
    # on input rows -1..N, so a slab of N+4 rows reproduces the full-image result for them.
    slab = Image.fromarray(np.array(img1)[:N + 4])
    zeros = [["0x00"] * 128]
    bl, _ = T.blur_image(slab)
    dump("blur", {"image": "img1", "original": (zeros + o1)[:N + 2], "transformed": compress_by_rows(bl)[:N]})
    sh, _ = T.sharpen_image(slab)
    dump("sharpness", {"image": "img1", "original": (zeros + o1)[:N + 2], "transformed": compress_by_rows(sh)[:N]})
    # resize HD->SD: 3 original rows -> 2 resized rows per step (per-pixel Python loop in the reference, ~20 s).
    rs = T.resize_image(img1, 480, 640)  # full image: the reference derives its ratios from the image height
    dump("resize", {"image": "img1", "original": o1[:3 * N], "transformed": compress_by_rows(rs)[:2 * N]})
    dump("crop", {"image": "img1", "info": 200 * 2 ** 24 + 100 * 2 ** 12, "original": o1[:N]})
    blocks = compress_by_blocks(np.array(img1))
    tr, ind = T.random_image_redaction(img1)
    dump("redact", {"image": "img1", "original": blocks[:N], "redact": ind[:N], "transformed": compress_by_blocks(tr)[:N]})
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
