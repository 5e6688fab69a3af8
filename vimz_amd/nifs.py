"""Nova's NIFS accumulator for a CALLER-SUPPLIED R1CS shape, on any of the four curves (BN254 G1, Grumpkin, Pallas, Vesta), through the
C-ABI seam alone: `vimz_r1cs_upload` / `vimz_commit_T` / `vimz_msm_vec` / `vimz_vec_axpy` / `vimz_r1cs_check_relaxed` (include/vimz_hip.h).

This is what `RecursiveSNARK::prove_step` does per step with nova-snark's `NIFS::prove` (reached from vimz/src/nova_snark_backend/
folding.rs:35-41) when the step circuit's R1CS and witnesses come from elsewhere — e.g. a circom circuit compiled for the Pasta primes —
and only the heavy lifting is rerouted to the GPU.  The running witness W, the error vector E and the products stay resident; per step
the host sees two commitments and a challenge.  The transcript is SHA3-256 over the canonical encodings (the product's own; a caller
wiring this into nova-snark keeps nova-snark's transcript and only uses the calls below).

No CPU fallback: every vector operation is a device call."""
import hashlib

import numpy as np

from . import _lib as L
from . import hip


def _limbs(x):
    return np.array([(int(x) >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)


def _int(limbs):
    return sum(int(v) << (64 * i) for i, v in enumerate(np.asarray(limbs, dtype=np.uint64).reshape(-1)[:4]))


class RelaxedAccumulator:
    """Running relaxed instance (comm_W, comm_E, u, X) and its witness (W, E) for the shape (A, B, C) over `curve`'s scalar field.

    Assignment layout z = [W (n_witness) | u | X (n_public)], as R1CSShape::multiply_vec orders it in nova-snark; the matrices are COO
    triplets over that layout.  `ck`: a commitment key of at least max(n_witness, nrows) generators on `curve` (derived on the GPU if
    None)."""

    def __init__(self, ctx, curve, nrows, n_witness, n_public, A, B, C, modulus, ck=None, ck_label=b"ck"):
        self.ctx, self.curve, self.field = ctx, curve, L.CURVE_SCALAR_FIELD[curve]
        self.nrows, self.nw, self.nx, self.q = nrows, n_witness, n_public, int(modulus)
        if self.q != L.MODULUS[self.field]:      # (a circom file compiled for another prime would upload fine and fold mod the wrong field)
            raise ValueError(f"the shape's prime {self.q:#x} is not the scalar field of curve {curve}")
        self.ncols = n_witness + 1 + n_public
        self.shape = hip.R1CSShape(ctx, self.field, nrows, self.ncols, A, B, C)
        self.own_ck = ck is None
        self.ck = ck if ck is not None else ctx.bases_generate(curve, max(n_witness, nrows), ck_label)
        self.z = ctx.vec_alloc(self.field, self.ncols)          # zero instance: u = 0, X = 0, W = 0
        self.E = ctx.vec_alloc(self.field, nrows)
        self.u, self.X = 0, [0] * n_public
        self.comm_W = np.zeros(8, dtype=np.uint64)
        self.comm_E = np.zeros(8, dtype=np.uint64)
        self.steps = 0

    @classmethod
    def from_r1cs(cls, ctx, curve, r1cs, ck=None):
        """From a parsed circom `.r1cs` (vimz_amd.iden3.read_r1cs) whose prime is `curve`'s scalar field."""
        from . import iden3
        n_w, n_pub, A, B, Cm = iden3.to_nova_columns(r1cs)
        return cls(ctx, curve, r1cs["n_constraints"], n_w, n_pub, A, B, Cm, r1cs["prime"], ck=ck)

    # ---- host-side pieces ------------------------------------------------------------------------------------------------------------
    def _scalar_mul(self, point, k):
        """k·P for one affine point, as an MSM of one term over a throw-away key (vimz_msm)."""
        if not np.any(point):
            return np.zeros(8, dtype=np.uint64)
        B = self.ctx.bases_upload(self.curve, np.asarray(point, dtype=np.uint64).reshape(1, 8))
        try:
            return self.ctx.msm(B, _limbs(k).reshape(1, 4))
        finally:
            B.free()

    def _add(self, p, q):
        return self.ctx.curve_add(self.curve, np.asarray(p, dtype=np.uint64).reshape(1, 8), np.asarray(q, dtype=np.uint64).reshape(1, 8))[0]

    def challenge(self, comm_W2, X2, comm_T):
        h = hashlib.sha3_256(b"vimz-amd/nifs/1")
        for a in (self.comm_W, self.comm_E, _limbs(self.u), *[_limbs(x) for x in self.X], comm_W2, *[_limbs(x) for x in X2], comm_T):
            h.update(np.ascontiguousarray(a, dtype=np.uint64).tobytes())
        return (int.from_bytes(h.digest()[:16], "little") | 1 << 128) % self.q

    # ---- one folding step --------------------------------------------------------------------------------------------------------------
    def fold(self, witness, public):
        """Fold the fresh instance (witness: n_witness canonical elements as ints or (n, 4) limbs; public: n_public ints).  Returns
        (challenge, comm_W of the fresh instance, comm_T)."""
        w = np.asarray(witness, dtype=np.uint64).reshape(-1, 4) if isinstance(witness, np.ndarray) else np.stack([_limbs(x) for x in witness])
        assert w.shape[0] == self.nw and len(public) == self.nx
        z2 = self.ctx.vec_alloc(self.field, self.ncols)
        T = None
        try:
            z2.upload(w, 0)
            z2.upload(np.stack([_limbs(1)] + [_limbs(x) for x in public]), self.nw)
            comm_W2 = self.ctx.msm_vec(self.ck, z2, n=self.nw)
            T, comm_T = self.shape.commit_T(self.ck, self.z, self.u, z2, 1)
            r = self.challenge(comm_W2, public, comm_T)
            hip.vec_axpy(self.ctx, self.z, r, z2)              # W, u and X together: z holds all three
            hip.vec_axpy(self.ctx, self.E, r, T)
        finally:
            z2.free()
            if T is not None:
                T.free()
        self.u = (self.u + r) % self.q
        self.X = [(a + r * int(b)) % self.q for a, b in zip(self.X, public)]
        self.comm_W = self._add(self.comm_W, self._scalar_mul(comm_W2, r))
        self.comm_E = self._add(self.comm_E, self._scalar_mul(comm_T, r))
        self.steps += 1
        return r, comm_W2, comm_T

    # ---- RecursiveSNARK::verify's part for this instance: is_sat_relaxed ---------------------------------------------------------------------
    def verify(self):
        """0 if the running instance satisfies the relaxed relation and both commitments open to the resident W and E; else a bit mask
        (1: relation, 2: comm_W, 4: comm_E, 8: the scalars kept on the host differ from the resident ones)."""
        res = 0
        bad, _ = self.shape.check_relaxed(self.z, self.u, self.E)
        if bad:
            res |= 1
        if not np.array_equal(self.ctx.msm_vec(self.ck, self.z, n=self.nw), self.comm_W):
            res |= 2
        if not np.array_equal(self.ctx.msm_vec(self.ck, self.E, n=self.nrows), self.comm_E):
            res |= 4
        tail = self.z.download(self.nw, 1 + self.nx)
        if [_int(t) for t in tail] != [self.u] + self.X:
            res |= 8
        return res

    def instance(self):
        return {"comm_W": self.comm_W.copy(), "comm_E": self.comm_E.copy(), "u": self.u, "X": list(self.X), "steps": self.steps}

    def free(self):
        self.shape.free(); self.z.free(); self.E.free()
        if self.own_ck:
            self.ck.free()
