"""ctypes wrapper over oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Field elements are Python ints at this level; arrays cross the boundary as numpy uint64 (n, 4)
little-endian limbs in canonical (non-Montgomery) form.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

FIELD_BN_FR, FIELD_BN_FQ, FIELD_PALLAS_FP, FIELD_VESTA_FQ = 0, 1, 2, 3
CURVE_BN_G1, CURVE_GRUMPKIN, CURVE_PALLAS, CURVE_VESTA = 0, 1, 2, 3
# base field / scalar field of each curve
CURVE_BASE = {0: 1, 1: 0, 2: 2, 3: 3}
CURVE_SCALAR = {0: 0, 1: 1, 2: 3, 3: 2}
# generators
P_PALLAS = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
Q_VESTA = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
GENERATORS = {
    0: (1, 2),
    1: (1, 17631683881184975370165255887551781615748388533673675138860),
    2: (P_PALLAS - 1, 2),
    3: (Q_VESTA - 1, 2),
}

(T_BLUR, T_BRIGHTNESS, T_CONTRAST, T_CROP, T_GRAYSCALE, T_HASH, T_REDACT, T_RESIZE, T_SHARPNESS) = range(9)


def to_limbs(vals):
    """ints -> (n,4) uint64"""
    vals = list(vals)
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        v = int(v)
        for k in range(4):
            out[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def from_limbs(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [int(a[0]) | int(a[1]) << 64 | int(a[2]) << 128 | int(a[3]) << 192 for a in arr]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.orc_msm_mont_timed.restype = C.c_double
        lib.orc_first_unsat.restype = C.c_long
        self.modulus = {}
        for f in range(4):
            m = np.zeros(4, dtype=np.uint64)
            lib.orc_field_modulus(f, _p(m))
            self.modulus[f] = from_limbs(m)[0]

    # ---- fields
    def _bin(self, fn, fid, a, b):
        o = np.zeros(4, dtype=np.uint64)
        fn(fid, _p(to_limbs([a])), _p(to_limbs([b])), _p(o))
        return from_limbs(o)[0]

    def f_add(self, fid, a, b): return self._bin(self.lib.orc_f_add, fid, a, b)
    def f_sub(self, fid, a, b): return self._bin(self.lib.orc_f_sub, fid, a, b)
    def f_mul(self, fid, a, b): return self._bin(self.lib.orc_f_mul, fid, a, b)

    def f_inv(self, fid, a):
        o = np.zeros(4, dtype=np.uint64)
        self.lib.orc_f_inv(fid, _p(to_limbs([a])), _p(o))
        return from_limbs(o)[0]

    def to_mont(self, fid, limbs):
        limbs = np.ascontiguousarray(limbs, dtype=np.uint64)
        o = np.zeros_like(limbs)
        self.lib.orc_f_to_mont(fid, _p(limbs), _p(o), C.c_size_t(limbs.size // 4))
        return o

    def from_mont(self, fid, limbs):
        limbs = np.ascontiguousarray(limbs, dtype=np.uint64)
        o = np.zeros_like(limbs)
        self.lib.orc_f_from_mont(fid, _p(limbs), _p(o), C.c_size_t(limbs.size // 4))
        return o

    # ---- curves (points are (x, y) int tuples; identity = (0, 0))
    def on_curve(self, cid, pt):
        return bool(self.lib.orc_curve_on(cid, _p(to_limbs(pt))))

    def curve_add(self, cid, p, q):
        o = np.zeros(8, dtype=np.uint64)
        self.lib.orc_curve_add(cid, _p(to_limbs(p)), _p(to_limbs(q)), _p(o))
        return tuple(from_limbs(o))

    def curve_mul(self, cid, p, k):
        o = np.zeros(8, dtype=np.uint64)
        self.lib.orc_curve_mul(cid, _p(to_limbs(p)), _p(to_limbs([k])), _p(o))
        return tuple(from_limbs(o))

    def seq_bases(self, cid, n, start=0):
        """P_i = (start + i + 1) * G, as (n, 8) uint64 canonical affine."""
        o = np.zeros((n, 8), dtype=np.uint64)
        self.lib.orc_curve_seq_bases(cid, _p(to_limbs(GENERATORS[cid])), _p(to_limbs([start])), C.c_size_t(n), _p(o))
        return o

    def msm(self, cid, bases, scalars, threads=8):
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        n = scalars.size // 4
        o = np.zeros(8, dtype=np.uint64)
        self.lib.orc_msm(cid, _p(bases), _p(scalars), C.c_size_t(n), threads, _p(o))
        return tuple(from_limbs(o))

    def msm_mont_timed(self, cid, bases_mont, scalars, threads):
        n = scalars.size // 4
        o = np.zeros(8, dtype=np.uint64)
        secs = self.lib.orc_msm_mont_timed(cid, _p(bases_mont), _p(scalars), C.c_size_t(n), threads, _p(o))
        return secs, tuple(from_limbs(o))

    # ---- Nova IVC relation (oracle/nova.hpp)
    def nova_hash(self, fid, inputs):
        o = np.zeros(4, dtype=np.uint64)
        self.lib.orc_nova_hash(fid, _p(to_limbs(inputs)), len(inputs), _p(o))
        return from_limbs(o)[0]

    def nova_instance_hash(self, fid, digest, i, z0, z, U):
        """trunc250(H(digest, i, z0, z, U)); U = [W.x, W.y, E.x, E.y, u, X0, X1]."""
        assert len(z0) == len(z)
        o = np.zeros(4, dtype=np.uint64)
        self.lib.orc_nova_instance_hash(fid, _p(to_limbs([digest])), C.c_uint64(i), _p(to_limbs(z0)), _p(to_limbs(z)), len(z), _p(to_limbs(U)), _p(o))
        return from_limbs(o)[0]

    def nova_step(self, side, is_primary, digest, i, z0, z_i, z_next, U, u, T):
        """The augmented circuit's relation, natively: returns None if the incoming hash does not match (or the base case does
        not start from z0), else (U_new[7], rho, x1)."""
        Un, rho, x1 = np.zeros((7, 4), dtype=np.uint64), np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        ok = self.lib.orc_nova_step(side, int(is_primary), _p(to_limbs([digest])), C.c_uint64(i), _p(to_limbs(z0)), _p(to_limbs(z_i)), _p(to_limbs(z_next)), len(z_i),
                                    _p(to_limbs(U)), _p(to_limbs(u)), _p(to_limbs(T)), _p(Un), _p(rho), _p(x1))
        if not ok:
            return None
        return from_limbs(Un), from_limbs(rho)[0], from_limbs(x1)[0]

    def r1cs_check_relaxed(self, fid, tables, n_wires, z, u=1, E=None, threads=8):
        """First row where Az∘Bz != u·Cz + E over field fid (-1 if none).  tables: dict with {A,B,C}_{rowptr,col,coef} (uint32)
        and dict_canon ((n,4) uint64), as exported by the product."""
        self.lib.orc_r1cs_check_relaxed.restype = C.c_long
        PP = C.c_void_p * 3
        keep = [np.ascontiguousarray(tables[f"{m}_{k}"], dtype=np.uint32) for k in ("rowptr", "col", "coef") for m in "ABC"]
        rp, col, coef = PP(*[a.ctypes.data for a in keep[0:3]]), PP(*[a.ctypes.data for a in keep[3:6]]), PP(*[a.ctypes.data for a in keep[6:9]])
        d = np.ascontiguousarray(tables["dict_canon"], dtype=np.uint64)
        z = np.ascontiguousarray(z, dtype=np.uint64)
        Ep = _p(np.ascontiguousarray(E, dtype=np.uint64)) if E is not None else None
        nrows = len(keep[0]) - 1
        return self.lib.orc_r1cs_check_relaxed(fid, C.c_size_t(nrows), C.c_size_t(n_wires), rp, col, coef, _p(d), C.c_size_t(d.size // 4), _p(z),
                                               _p(to_limbs([u])), Ep, threads)

    # ---- Poseidon & hashers
    def poseidon(self, inputs):
        o = np.zeros(4, dtype=np.uint64)
        self.lib.orc_poseidon(_p(to_limbs(inputs)), len(inputs), _p(o))
        return from_limbs(o)[0]

    def poseidon_constants(self, t):
        rf, rp = C.c_int(), C.c_int()
        self.lib.orc_poseidon_constants(t, None, None, C.byref(rf), C.byref(rp))
        Cs = np.zeros(((rf.value + rp.value) * t, 4), dtype=np.uint64)
        Ms = np.zeros((t * t, 4), dtype=np.uint64)
        self.lib.orc_poseidon_constants(t, _p(Cs), _p(Ms), C.byref(rf), C.byref(rp))
        return Cs, Ms, rf.value, rp.value

    def array_hash(self, arr):
        o = np.zeros(4, dtype=np.uint64)
        self.lib.orc_array_hash(_p(to_limbs(arr)), len(arr), _p(o))
        return from_limbs(o)[0]

    def head_tail_hash(self, head, tail):
        o = np.zeros(4, dtype=np.uint64)
        self.lib.orc_head_tail_hash(_p(to_limbs([head])), _p(to_limbs(tail)), len(tail), _p(o))
        return from_limbs(o)[0]

    def image_hash(self, rows_limbs, nrows, width):
        rows_limbs = np.ascontiguousarray(rows_limbs, dtype=np.uint64)
        o = np.zeros(4, dtype=np.uint64)
        self.lib.orc_image_hash(_p(rows_limbs), nrows, width, _p(o))
        return from_limbs(o)[0]

    # ---- step semantics
    def step_eval(self, t, z_in, inputs_limbs, width=128, width2=64, rows_in=3, rows_out=2, crop_h=480):
        inputs_limbs = np.ascontiguousarray(inputs_limbs, dtype=np.uint64)
        L = self.lib.orc_ivc_state_len(t)
        assert len(z_in) == L
        need = self.lib.orc_step_input_width(t, width, width2, rows_in, rows_out, crop_h)
        assert inputs_limbs.size == need * 4, (inputs_limbs.size // 4, need)
        zo = np.zeros((L, 4), dtype=np.uint64)
        ok = self.lib.orc_step_eval(t, width, width2, rows_in, rows_out, crop_h, _p(to_limbs(z_in)), _p(inputs_limbs), _p(zo))
        return bool(ok), from_limbs(zo)

    # ---- relaxed R1CS algebra (arrays of (n,4) uint64 canonical)
    def spmv(self, fid, nrows, ncols, row_ptr, col, val, z, threads=8):
        out = np.zeros((nrows, 4), dtype=np.uint64)
        self.lib.orc_spmv(fid, C.c_size_t(nrows), C.c_size_t(ncols), _p(np.ascontiguousarray(row_ptr, dtype=np.uint32)),
                          _p(np.ascontiguousarray(col, dtype=np.uint32)), _p(np.ascontiguousarray(val, dtype=np.uint64)),
                          _p(np.ascontiguousarray(z, dtype=np.uint64)), _p(out), threads)
        return out

    def cross_term(self, fid, az1, bz1, cz1, u1, az2, bz2, cz2, u2, threads=1):
        n = az1.size // 4
        T = np.zeros((n, 4), dtype=np.uint64)
        c = lambda a: _p(np.ascontiguousarray(a, dtype=np.uint64))
        if threads > 1:
            self.lib.orc_cross_term_mt(fid, C.c_size_t(n), c(az1), c(bz1), c(cz1), _p(to_limbs([u1])), c(az2), c(bz2), c(cz2), _p(to_limbs([u2])), _p(T), threads)
        else:
            self.lib.orc_cross_term(fid, C.c_size_t(n), c(az1), c(bz1), c(cz1), _p(to_limbs([u1])), c(az2), c(bz2), c(cz2), _p(to_limbs([u2])), _p(T))
        return T

    def axpy(self, fid, a, r, b, threads=1):
        n = a.size // 4
        o = np.zeros((n, 4), dtype=np.uint64)
        args = (fid, C.c_size_t(n), _p(np.ascontiguousarray(a, dtype=np.uint64)), _p(to_limbs([r])), _p(np.ascontiguousarray(b, dtype=np.uint64)), _p(o))
        if threads > 1:
            self.lib.orc_axpy_mt(*args, threads)
        else:
            self.lib.orc_axpy(*args)
        return o

    def first_unsat(self, fid, az, bz, cz, u=1, E=None):
        n = az.size // 4
        c = lambda a: _p(np.ascontiguousarray(a, dtype=np.uint64))
        return self.lib.orc_first_unsat(fid, C.c_size_t(n), c(az), c(bz), c(cz), _p(to_limbs([u])), None if E is None else c(E))


CURVE_B = {0: 3, 1: -17, 2: 5, 3: 5}
BASE_MODULUS = {
    0: 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
    1: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
    2: P_PALLAS, 3: Q_VESTA,
}


def _sqrt_mod(a, p):
    """Tonelli-Shanks in Python big ints (None if a is a non-residue)."""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2; s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p; i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        t, r = t * c % p, r * b % p
    return r


def ck_derive(cid, label, i):
    """Python restatement of the product's commitment-key derivation (vimz_amd/csrc/ckgen.hpp):
    try-and-increment over SHAKE256(label || LE64(i) || LE32(ctr))."""
    import hashlib
    p = BASE_MODULUS[cid]
    bits = p.bit_length()
    ctr = 0
    while True:
        h = hashlib.shake_256(label + i.to_bytes(8, "little") + ctr.to_bytes(4, "little")).digest(32)
        ctr += 1
        v = int.from_bytes(h, "little")
        sign = v >> 255
        x = v & ((1 << bits) - 1)
        if x >= p:
            continue
        y = _sqrt_mod(x * x * x + CURVE_B[cid], p)
        if y is None:
            continue
        cands = (y, (p - y) % p)   # either root; the parity rule picks one
        y = cands[0] if (cands[0] & 1) == sign else cands[1]
        return x, y


def witness_execute(orc, circuit, z_in, priv_limbs):
    """Run the oracle's independent executor over the product-built witness program of `circuit`
    (vimz_amd.circuit.Circuit).  Returns (status, z (n_wires,4) canonical, z_out ints)."""
    lib = orc.lib
    names = ["DECOMP", "LANE_GROUPS", "LANE_INSTR", "LANE_ROWS", "JOBS", "CHAINS", "FOPS", "ZOUT"]
    if not hasattr(circuit, "_tables"):
        circuit._tables = {n: np.ascontiguousarray(circuit.export(n)) for n in names}
        for i, n in enumerate(names):
            sz = lib.orc_witness_struct_sizes(i)
            assert circuit._tables[n].size % sz == 0, (n, sz)
    T = circuit._tables
    sizes = np.array([circuit.n_wires, circuit.len_z, circuit.n_priv, circuit.n_decomp, circuit.n_lane_groups, circuit.n_jobs,
                      circuit.n_chains, circuit.n_fops], dtype=np.uint32)
    z = np.zeros((circuit.n_wires, 4), dtype=np.uint64)
    zo = np.zeros((circuit.len_z, 4), dtype=np.uint64)
    priv = np.ascontiguousarray(priv_limbs, dtype=np.uint64)
    assert priv.size == 4 * circuit.n_priv
    if "_lc" not in T:
        T["_lc"] = np.ascontiguousarray(circuit.export("LC_TERMS"))
        T["_dict"] = np.ascontiguousarray(circuit.export("DICT_CANON", np.uint64))
    st = lib.orc_witness_execute(_p(sizes), *[_p(T[n]) for n in names], _p(T["_lc"]), _p(T["_dict"]), _p(to_limbs(z_in)), _p(priv), _p(z), _p(zo))
    return st, z, from_limbs(zo)


def r1cs_check(orc, circuit, z, want_products=False, threads=8):
    """First row where Az∘Bz != Cz for the circuit's R1CS (-1 if satisfied), computed by the oracle."""
    lib = orc.lib
    lib.orc_r1cs_check.restype = C.c_long
    if not hasattr(circuit, "_csr"):
        circuit._csr = [tuple(np.ascontiguousarray(a) for a in circuit.csr(m)) for m in "ABC"]
        circuit._dict = np.ascontiguousarray(circuit.export("DICT_CANON", np.uint64))
    PP = C.c_void_p * 3
    rp = PP(*[a[0].ctypes.data for a in circuit._csr])
    col = PP(*[a[1].ctypes.data for a in circuit._csr])
    coef = PP(*[a[2].ctypes.data for a in circuit._csr])
    n = circuit.n_constraints
    outs = [np.zeros((n, 4), dtype=np.uint64) if want_products else None for _ in range(3)]
    z = np.ascontiguousarray(z, dtype=np.uint64)
    r = lib.orc_r1cs_check(C.c_size_t(n), C.c_size_t(circuit.n_wires), rp, col, coef, _p(circuit._dict), C.c_size_t(circuit.n_dict),
                           _p(z), *[(_p(o) if o is not None else None) for o in outs], threads)
    return (r, outs) if want_products else r


_cached = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load():
    global _cached
    if _cached is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".hpp", ".cpp"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            build()
        _cached = Oracle(C.CDLL(so))
    return _cached
