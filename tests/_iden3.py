"""TEST INFRASTRUCTURE: writers for the iden3 binary formats (`.r1cs`, `.wtns`), used to check the
product's native readers without circom.  Layout as published in the iden3 r1cs binary format spec."""
import struct

import numpy as np

PRIME = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def _section(t, payload):
    return struct.pack("<IQ", t, len(payload)) + payload


def write_r1cs(n_wires, n_pub_out, n_pub_in, n_prv, csr_abc, dict_canon, prime=PRIME):
    """csr_abc: three (row_ptr, col, coef) uint32 triples; dict_canon: (n,4) uint64 coefficient dictionary."""
    ncon = len(csr_abc[0][0]) - 1
    dict_bytes = [bytes(np.ascontiguousarray(d)) for d in dict_canon]
    hdr = struct.pack("<I", 32) + prime.to_bytes(32, "little") + struct.pack("<IIIIQI", n_wires, n_pub_out, n_pub_in, n_prv, n_wires, ncon)
    body = bytearray()
    for k in range(ncon):
        for rp, col, coef in csr_abc:
            lo, hi = int(rp[k]), int(rp[k + 1])
            body += struct.pack("<I", hi - lo)
            for j in range(lo, hi):
                body += struct.pack("<I", int(col[j])) + dict_bytes[int(coef[j])]
    labels = b"".join(struct.pack("<Q", i) for i in range(n_wires))
    return b"r1cs" + struct.pack("<II", 1, 3) + _section(1, hdr) + _section(2, bytes(body)) + _section(3, labels)


def write_wtns(witness_limbs, prime=PRIME):
    w = np.ascontiguousarray(witness_limbs, dtype=np.uint64).reshape(-1, 4)
    hdr = struct.pack("<I", 32) + prime.to_bytes(32, "little") + struct.pack("<I", w.shape[0])
    return b"wtns" + struct.pack("<II", 2, 2) + _section(1, hdr) + _section(2, w.tobytes())
