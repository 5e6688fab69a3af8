#!/usr/bin/env python3
"""Folding steps/s of the seam-based NIFS accumulator (vimz_amd.nifs.RelaxedAccumulator) on a synthetic satisfiable circuit over a Pasta
scalar field: product rows of random earlier wires with dense 255-bit witnesses (so W and the cross term are dense scalars — the
worst case for both MSMs).  GPU box only.  usage: pasta_seam_bench.py [curve: pallas|vesta] [constraints] [steps]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vimz_amd import hip, nifs, _lib  # noqa: E402

Q = {_lib.CURVE_PALLAS: 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,      # scalar field of Pallas = base field of Vesta
     _lib.CURVE_VESTA: 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001}


def limbs(vals):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(4):
            out[i, k] = (v >> (64 * k)) & ((1 << 64) - 1)
    return out


def main():
    curve = {"pallas": _lib.CURVE_PALLAS, "vesta": _lib.CURVE_VESTA}[sys.argv[1] if len(sys.argv) > 1 else "pallas"]
    n_mul = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    q = Q[curve]
    rng = np.random.default_rng(7)
    n_in = 1024
    m = n_in + n_mul
    a_idx = np.array([rng.integers(0, k) for k in range(n_in, m)], dtype=np.int64)
    b_idx = np.array([rng.integers(0, k) for k in range(n_in, m)], dtype=np.int64)
    rows = np.arange(n_mul, dtype=np.uint32)
    one = np.tile(np.array([1, 0, 0, 0], dtype=np.uint64), (n_mul, 1))
    A = (rows, a_idx.astype(np.uint32), one)
    B = (rows, b_idx.astype(np.uint32), one)
    C = (rows, (np.arange(n_mul) + n_in).astype(np.uint32), one)
    # two public outputs: w_last * u = x0, w_last-1 * u = x1
    A = (np.concatenate([A[0], [n_mul, n_mul + 1]]).astype(np.uint32), np.concatenate([A[1], [m - 1, m - 2]]).astype(np.uint32), np.concatenate([A[2], one[:2]]))
    B = (np.concatenate([B[0], [n_mul, n_mul + 1]]).astype(np.uint32), np.concatenate([B[1], [m, m]]).astype(np.uint32), np.concatenate([B[2], one[:2]]))
    C = (np.concatenate([C[0], [n_mul, n_mul + 1]]).astype(np.uint32), np.concatenate([C[1], [m + 1, m + 2]]).astype(np.uint32), np.concatenate([C[2], one[:2]]))

    def witness():
        w = [int.from_bytes(rng.bytes(32), "little") % q for _ in range(n_in)] + [0] * n_mul
        for t in range(n_mul):
            w[n_in + t] = w[a_idx[t]] * w[b_idx[t]] % q
        return limbs(w), [w[m - 1], w[m - 2]]

    ctx = hip.Context(0)
    acc = nifs.RelaxedAccumulator(ctx, curve, n_mul + 2, m, 2, A, B, C, q)
    wit = [witness() for _ in range(3)]
    for w, X in wit[:2]:
        acc.fold(w, X)
    ctx.sync()
    t0 = time.time()
    for i in range(steps):
        acc.fold(*wit[i % 3])
    ctx.sync()
    dt = time.time() - t0
    ok = acc.verify()
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'pallas'}: {n_mul + 2} constraints, {m} witness wires: {steps} folds in {dt * 1e3:.1f} ms = {steps / dt:.1f} steps/s"
          f" ({dt / steps * 1e3:.2f} ms per fold incl. the witness upload), verify() = {ok}")
    acc.free(); ctx.close()


if __name__ == "__main__":
    main()
