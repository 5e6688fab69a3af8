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
