"""TEST INFRASTRUCTURE ONLY: a CPU stand-in for vimz_amd.hip.Prover built on the oracle, implementing the same
transcript and fold algebra.  Used (a) to re-derive GPU folds independently and (b) as the backend of the
world_size-2 gloo test of the sharding driver, which must run without a GPU."""
import pickle

import numpy as np

from tests._oracle import from_limbs, r1cs_check, to_limbs, witness_execute

R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
M128 = (1 << 128) - 1


def ro_point(pt):
    x, y = pt
    return [x & M128, (x >> 128) | ((y & 1) << 126)]


class OracleProver:
    def __init__(self, oracle, circuit, key):
        """key: (>= max(aux wires, constraints), 8) uint64 canonical affine generators."""
        self.o, self.c, self.key = oracle, circuit, key
        self.aux0 = 1 + 2 * circuit.len_z
        self.n_aux = circuit.n_wires - self.aux0
        self.reset([0] * circuit.len_z)

    def _commit(self, v, n):
        return self.o.msm(0, self.key[:n], v, threads=8)

    def reset(self, z0):
        c = self.c
        self.z0 = list(z0); self.z = list(z0); self.steps = 0
        self.ro = self.o.poseidon([0x56494d7a, c.n_constraints, c.n_wires, c.len_z, c.t, c.shape[0]])
        self.zdig = 0
        for v in z0:
            self.zdig = self.o.poseidon([self.zdig, v])
        self.Z = self.E = self.AZ = self.BZ = self.CZ = None
        self.u = 0; self.cW = (0, 0); self.cE = (0, 0)

    def state_chain(self, z_start, inputs):
        z = list(z_start); out = [to_limbs(z)]
        for inp in inputs:
            st, _, z = witness_execute(self.o, self.c, z, inp)
            out.append(to_limbs(z))
        return np.stack(out)

    def fold(self, inputs):
        o, c = self.o, self.c
        for inp in inputs:
            st, w, z_next = witness_execute(o, c, self.z, inp)
            bad, (a2, b2, c2) = r1cs_check(o, c, w, want_products=True)
            assert st == 0 and bad == -1
            for v in z_next:
                self.zdig = o.poseidon([self.zdig, v])
            cW2 = self._commit(w[self.aux0:], self.n_aux)
            if self.steps == 0:
                self.Z, self.AZ, self.BZ, self.CZ = w, a2, b2, c2
                self.E = np.zeros((c.n_constraints, 4), dtype=np.uint64)
                self.cW, self.u = cW2, 1
                self.ro = o.poseidon([self.ro] + ro_point(cW2) + [self.zdig])
            else:
                T = o.cross_term(0, self.AZ, self.BZ, self.CZ, self.u, a2, b2, c2, 1)
                cT = self._commit(T, c.n_constraints)
                self.ro = o.poseidon([self.ro] + ro_point(cW2) + ro_point(cT) + [self.zdig])
                r = self.ro & M128
                self.Z = o.axpy(0, self.Z, r, w); self.E = o.axpy(0, self.E, r, T)
                self.AZ = o.axpy(0, self.AZ, r, a2); self.BZ = o.axpy(0, self.BZ, r, b2); self.CZ = o.axpy(0, self.CZ, r, c2)
                self.cW = o.curve_add(0, self.cW, o.curve_mul(0, cW2, r))
                self.cE = o.curve_add(0, self.cE, o.curve_mul(0, cT, r))
                self.u = (self.u + r) % R_MOD
            self.z = z_next
            self.steps += 1

    def export(self):
        d = {k: getattr(self, k) for k in ("Z", "E", "AZ", "BZ", "CZ", "u", "cW", "cE", "ro", "zdig", "z", "z0", "steps")}
        return np.frombuffer(pickle.dumps(d), dtype=np.uint8)

    def merge(self, blob):
        o = self.o
        d = pickle.loads(np.asarray(blob, dtype=np.uint8).tobytes())
        if d["steps"] == 0:
            return
        if list(d["z0"]) != list(self.z):
            raise ValueError("merge: the incoming segment does not start at the state this accumulator ends in")
        if self.steps == 0:
            for k, v in d.items():
                setattr(self, k, v)
            return
        T = o.cross_term(0, self.AZ, self.BZ, self.CZ, self.u, d["AZ"], d["BZ"], d["CZ"], d["u"])
        cT = self._commit(T, self.c.n_constraints)
        self.ro = o.poseidon([self.ro, d["ro"]] + ro_point(cT))
        self.zdig = o.poseidon([self.zdig, d["zdig"]])
        r = self.ro & M128
        r2 = r * r % R_MOD
        self.Z = o.axpy(0, self.Z, r, d["Z"])
        self.E = o.axpy(0, o.axpy(0, self.E, r, T), r2, d["E"])
        self.AZ = o.axpy(0, self.AZ, r, d["AZ"]); self.BZ = o.axpy(0, self.BZ, r, d["BZ"]); self.CZ = o.axpy(0, self.CZ, r, d["CZ"])
        self.cW = o.curve_add(0, self.cW, o.curve_mul(0, d["cW"], r))
        self.cE = o.curve_add(0, o.curve_add(0, self.cE, o.curve_mul(0, cT, r)), o.curve_mul(0, d["cE"], r2))
        self.u = (self.u + r * d["u"]) % R_MOD
        self.steps += d["steps"]; self.z = d["z"]

    def verify(self):
        """0 = accepted; same flag bits as vimz_prover_verify."""
        o, c = self.o, self.c
        _, (az, bz, cz) = r1cs_check(o, c, self.Z, want_products=True)
        res = 0
        if o.first_unsat(0, az, bz, cz, u=self.u, E=self.E) != -1:
            res |= 1
        if self._commit(self.Z[self.aux0:], self.n_aux) != self.cW:
            res |= 2
        if self._commit(self.E, c.n_constraints) != self.cE:
            res |= 4
        if not (np.array_equal(az, self.AZ) and np.array_equal(bz, self.BZ) and np.array_equal(cz, self.CZ)):
            res |= 8
        return res

    def instance(self):
        return {"comm_W": to_limbs(self.cW).reshape(-1), "comm_E": to_limbs(self.cE).reshape(-1), "u": to_limbs([self.u])[0],
                "z": to_limbs(self.z), "steps": self.steps}
