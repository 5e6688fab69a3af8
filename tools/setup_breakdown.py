#!/usr/bin/env python3
"""Where "Prepare folding" goes (tools/e2e.py's span; the reference's `prepare_folding`, vimz/src/nova_snark_backend/folding.rs:20-25): contexts, step
circuit (host builder), commitment key (GPU), window tables (GPU), secondary key, one IVC per segment (verifier circuits' synthesis on the host, device
buffers, small-MSM tables).  usage: setup_breakdown.py <transformation> <resolution> [segments]"""
import json
import sys
import time

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vimz_amd import _lib, folding, hip  # noqa: E402
from vimz_amd.circuit import Circuit  # noqa: E402


def main():
    t, res = sys.argv[1], sys.argv[2]
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    out = {}
    t0 = time.time(); ctxs = [hip.Context(0) for _ in range(S)]; out["contexts"] = time.time() - t0
    t0 = time.time(); circuit = Circuit(t, *folding.default_shape(t, res)); out["step_circuit_host"] = time.time() - t0
    n = 1 << (max(circuit.n_wires, circuit.n_constraints) + folding.AUGMENTED_ROOM - 1).bit_length()
    t0 = time.time(); ck = ctxs[0].bases_generate(_lib.CURVE_BN254_G1, n, b"ck"); ctxs[0].sync(); out["ck_gpu"] = time.time() - t0
    t0 = time.time(); ck.precompute(15); ctxs[0].sync(); out["window_tables_gpu"] = time.time() - t0
    t0 = time.time(); ck2 = ctxs[0].bases_generate(_lib.CURVE_GRUMPKIN, folding.SECONDARY_KEY_LEN, b"ck-secondary"); out["ck_secondary"] = time.time() - t0
    batch = folding.default_batch(circuit)
    ivcs = []
    for k, c in enumerate(ctxs):
        t0 = time.time(); ivcs.append(hip.IVC(c, circuit, ck, ck2, max_batch=batch)); out[f"ivc_{k}"] = time.time() - t0
    out["total"] = sum(out.values())
    print(json.dumps({"config": f"{t}_step_{res}", "segments": S, "batch": batch, "seconds": out}))


if __name__ == "__main__":
    main()
