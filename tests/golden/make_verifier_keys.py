#!/usr/bin/env python3
"""Mint tests/golden/verifier_keys.json: the CONSTANTS of the reference's on-chain verifiers (contracts/*Verifier.sol, identical to
marketplace/contracts/*Verifier.sol) — Groth16 verification key (alpha, beta, gamma, delta, IC_0..IC_n), the KZG10 verifier's G_1 / G_2 / VK,
the public-parameter hash `public_inputs[0]` and the state width of the entry point.  Data only: no Solidity text is copied; the verification
logic is restated in tests/_novadecider.py (citing ContrastVerifier.sol line by line).

Runs ONLY in the build container (needs /root/reference); the output is committed, this script documents how it was made."""
import json
import os
import re

REF = "/root/reference/contracts"
HERE = os.path.dirname(os.path.abspath(__file__))


def ints(text):
    return [int(x) for x in re.findall(r"\b\d{20,}\b", text)]


def array_after(src, decl):
    """The integers of the Solidity array literal that follows `decl` (up to the closing `];`)."""
    at = src.index(decl)
    return ints(src[at:src.index("];", at)])


def extract(path):
    src = open(path).read()
    const = {m.group(1): int(m.group(2)) for m in re.finditer(r"uint256 constant (\w+)\s*=\s*(\d+);", src)}
    n_ic = 0
    while f"IC{n_ic}x" in const:
        n_ic += 1
    len_z = int(re.search(r"uint256\[(\d+)\] calldata initial_state", src).group(1))
    n_pub = int(re.search(r"uint256\[(\d+)\] memory public_inputs;", src).group(1))
    assert n_ic == n_pub + 1 == 1 + 1 + 1 + 2 * len_z + 10 + 10 + 4 + 10, (path, n_ic, n_pub, len_z)
    assert int(re.search(r"uint(?:256)?\[(\d+)\] calldata _pubSignals", src).group(1)) == n_pub
    g1 = array_after(src, "uint256[2] G_1 =")
    g2 = array_after(src, "uint256[2][2] G_2 =")
    vk = array_after(src, "uint256[2][2] VK =")
    assert len(g1) == 2 and len(g2) == 4 and len(vk) == 4
    return {
        "len_z": len_z,
        "n_public_inputs": n_pub,
        "pp_hash": str(int(re.search(r"public_inputs\[0\] = (\d+);", src).group(1))),
        "field_r": str(const["r"]), "field_q": str(const["q"]),
        # Groth16: G2 constants keep the contract's own names — x1/y1 go to the pairing precompile FIRST (= imaginary parts)
        "groth16": {"alpha": [str(const["alphax"]), str(const["alphay"])],
                    "beta": {k: str(const["beta" + k]) for k in ("x1", "x2", "y1", "y2")},
                    "gamma": {k: str(const["gamma" + k]) for k in ("x1", "x2", "y1", "y2")},
                    "delta": {k: str(const["delta" + k]) for k in ("x1", "x2", "y1", "y2")},
                    "ic": [[str(const[f"IC{k}x"]), str(const[f"IC{k}y"])] for k in range(n_ic)]},
        # KZG10: arrays as written, [[a, b], [c, d]]; the contract's `pairing` sends [k][1] before [k][0] ("imaginary part first")
        "kzg": {"G_1": [str(v) for v in g1], "G_2": [[str(g2[0]), str(g2[1])], [str(g2[2]), str(g2[3])]],
                "VK": [[str(vk[0]), str(vk[1])], [str(vk[2]), str(vk[3])]]},
        "limb_bits": int(re.search(r"x >> \((\d+) \* i\)", src).group(1)),
        "limbs": int(re.search(r"uint256\[(\d+)\] memory limbs;", src).group(1)),
    }


def main():
    out = {}
    for f in sorted(os.listdir(REF)):
        if f.endswith("Verifier.sol"):
            out[f[:-len("Verifier.sol")].lower()] = extract(os.path.join(REF, f))
    json.dump(out, open(os.path.join(HERE, "verifier_keys.json"), "w"), indent=1)
    print({k: (v["len_z"], v["n_public_inputs"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
