"""Readers for circom's artefacts in the iden3 binary formats — `.r1cs` (constraints) and `.wtns` (one witness) — for ANY prime, in Python:
what `load_r1cs` / `generate_witness_from_bin` hand to nova-scotia (vimz/src/nova_snark_backend/folding.rs:22, :35-41), feeding the
seam-based accumulator (`vimz_amd.nifs`).  (The native reader `vimz_circuit_load_r1cs` serves the BN254 provers.)

Column order: circom numbers the wires [1 | public outputs | public inputs | private]; nova-snark's R1CSShape orders an assignment
z = [W (private wires) | u | X (public wires)] — `to_nova_columns` maps one onto the other the way nova-scotia's CircomCircuit does."""
import struct

import numpy as np


def _sections(data, magic):
    if data[:4] != magic:
        raise ValueError(f"not a {magic.decode()} file")
    _version, nsec = struct.unpack_from("<II", data, 4)
    pos, out = 12, {}
    for _ in range(nsec):
        t, size = struct.unpack_from("<IQ", data, pos)
        pos += 12
        out[t] = data[pos:pos + size]
        pos += size
    return out


def read_r1cs(data):
    """-> dict(prime, n_wires, n_pub_out, n_pub_in, n_prv, n_constraints, A, B, C) with A, B, C as (rows u32, wires u32, values (nnz, 4) u64
    canonical) triplets in circom's wire numbering."""
    sec = _sections(bytes(data), b"r1cs")
    hdr = sec[1]
    fs = struct.unpack_from("<I", hdr, 0)[0]
    if fs != 32:
        raise ValueError("only 32-byte field elements are supported")
    prime = int.from_bytes(hdr[4:36], "little")
    n_wires, n_pub_out, n_pub_in, n_prv, _n_labels, ncon = struct.unpack_from("<IIIIQI", hdr, 36)
    body, pos = sec[2], 0
    mats = [([], [], []) for _ in range(3)]
    for k in range(ncon):
        for rows, wires, vals in mats:
            n = struct.unpack_from("<I", body, pos)[0]
            pos += 4
            for _ in range(n):
                w = struct.unpack_from("<I", body, pos)[0]
                rows.append(k); wires.append(w); vals.append(body[pos + 4:pos + 36])
                pos += 36
    def pack(m):
        rows, wires, vals = m
        v = np.frombuffer(b"".join(vals), dtype=np.uint64).reshape(-1, 4).copy() if vals else np.zeros((0, 4), dtype=np.uint64)
        return np.asarray(rows, dtype=np.uint32), np.asarray(wires, dtype=np.uint32), v
    A, B, C = (pack(m) for m in mats)
    return {"prime": prime, "n_wires": n_wires, "n_pub_out": n_pub_out, "n_pub_in": n_pub_in, "n_prv": n_prv, "n_constraints": ncon, "A": A, "B": B, "C": C}


def read_wtns(data):
    """-> (prime, (n_wires, 4) uint64 canonical values in circom's wire order; wire 0 is the constant 1)."""
    sec = _sections(bytes(data), b"wtns")
    hdr = sec[1]
    fs = struct.unpack_from("<I", hdr, 0)[0]
    if fs != 32:
        raise ValueError("only 32-byte field elements are supported")
    prime = int.from_bytes(hdr[4:36], "little")
    n = struct.unpack_from("<I", hdr, 36)[0]
    return prime, np.frombuffer(sec[2], dtype=np.uint64).reshape(n, 4).copy()


def to_nova_columns(r1cs):
    """The matrices with circom wire j moved to nova-snark's column: j = 0 -> the u slot, 1 <= j <= n_pub -> X, the rest -> W.
    -> (n_witness, n_public, A, B, C)."""
    n_pub = r1cs["n_pub_out"] + r1cs["n_pub_in"]
    n_w = r1cs["n_wires"] - 1 - n_pub
    def cols(wires):
        w = wires.astype(np.int64)
        return np.where(w == 0, n_w, np.where(w <= n_pub, n_w + w, w - n_pub - 1)).astype(np.uint32)
    out = [(rows, cols(wires), vals) for rows, wires, vals in (r1cs["A"], r1cs["B"], r1cs["C"])]
    return n_w, n_pub, out[0], out[1], out[2]


def split_witness(r1cs, wtns_values):
    """One `.wtns` -> (W (n_witness, 4), X list of ints) for RelaxedAccumulator.fold."""
    n_pub = r1cs["n_pub_out"] + r1cs["n_pub_in"]
    v = np.asarray(wtns_values, dtype=np.uint64).reshape(-1, 4)
    if v.shape[0] != r1cs["n_wires"] or any(int(x) for x in v[0] ^ np.array([1, 0, 0, 0], dtype=np.uint64)):
        raise ValueError("witness does not fit the circuit (length, or wire 0 is not 1)")
    X = [sum(int(l) << (64 * i) for i, l in enumerate(row)) for row in v[1:1 + n_pub]]
    return v[1 + n_pub:], X
