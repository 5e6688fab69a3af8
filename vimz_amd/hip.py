"""Thin object wrapper over the C ABI (include/vimz_hip.h) for Python callers: tests, bench.py and the
host-side folding driver.  Arrays are numpy uint64 with 4 limbs per field element; nothing is computed here.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class DeviceVec:
    def __init__(self, ctx, handle, field, n):
        self.ctx, self.h, self.field, self.n = ctx, handle, field, n

    def upload(self, host, offset=0, form=L.FORM_CANONICAL):
        host = _u64(host)
        self.ctx._chk(self.ctx.lib.vimz_vec_upload(self.ctx.h, self.h, offset, _ptr(host), host.size // 4, form))
        return self

    def download(self, offset=0, n=None, form=L.FORM_CANONICAL):
        n = self.n - offset if n is None else n
        out = np.zeros((n, 4), dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.vimz_vec_download(self.ctx.h, self.h, offset, _ptr(out), n, form))
        return out

    def free(self):
        if self.h:
            self.ctx.lib.vimz_vec_free(self.ctx.h, self.h)
            self.h = None


class Bases:
    def __init__(self, ctx, handle, curve, n):
        self.ctx, self.h, self.curve, self.n = ctx, handle, curve, n

    def precompute(self, window_bits=0):
        """Window tables in HBM (vimz_bases_precompute): later MSMs over this key share one bucket set."""
        self.ctx._chk(self.ctx.lib.vimz_bases_precompute(self.ctx.h, self.h, window_bits))
        return self

    def download(self, offset=0, n=None, form=L.FORM_CANONICAL):
        n = self.n - offset if n is None else n
        out = np.zeros((n, 8), dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.vimz_bases_download(self.ctx.h, self.h, offset, _ptr(out), n, form))
        return out

    def free(self):
        if self.h:
            self.ctx.lib.vimz_bases_free(self.ctx.h, self.h)
            self.h = None


class Context:
    """One per GPU (vimz_ctx)."""

    def __init__(self, device=0):
        self.lib = L.lib()
        h = C.c_void_p()
        rc = self.lib.vimz_ctx_create(device, C.byref(h))
        if rc != L.OK:
            raise L.VimzError(rc, self.lib.vimz_last_error(None).decode())
        self.h = h
        self.device = device

    def _chk(self, rc):
        if rc != L.OK:
            raise L.VimzError(rc, self.lib.vimz_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.vimz_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, hbm = C.c_int(), C.c_uint64()
        self._chk(self.lib.vimz_device_info(self.h, name, 256, C.byref(cus), C.byref(hbm)))
        return {"name": name.value.decode(), "cus": cus.value, "hbm_bytes": hbm.value}

    def host_fingerprint(self):
        """vimz_host_fingerprint: what a benchmark line says about the HOST it was measured on."""
        out = (C.c_double * 4)()
        self.lib.vimz_host_fingerprint.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._chk(self.lib.vimz_host_fingerprint(self.h, out))
        model = "?"
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        return {"cpu_model": model, "us_per_host_poseidon_t9": out[0], "us_per_empty_launch_and_sync": out[1], "usable_cores": int(out[2]),
                "us_per_event_record_and_sync": out[3]}

    def trace_marker(self, ident):
        """vimz_trace_marker: an empty kernel of `ident` workgroups in the profiler's kernel trace (bench.py: 1 / 2 around the timed region)."""
        self.lib.vimz_trace_marker.argtypes = [C.c_void_p, C.c_int]
        self._chk(self.lib.vimz_trace_marker(self.h, int(ident)))

    def sync(self):
        self._chk(self.lib.vimz_sync(self.h))

    def timer_start(self):
        self._chk(self.lib.vimz_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.lib.vimz_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def set_profiling(self, on):
        self._chk(self.lib.vimz_set_profiling(self.h, 1 if on else 0))

    def msm_last_profile(self):
        ms = (C.c_float * 6)()
        info = (C.c_uint32 * 4)()
        self._chk(self.lib.vimz_msm_last_profile(self.h, ms, info))
        names = ["hist", "scan", "scatter", "accumulate", "combine", "reduce"]
        return {"ms": dict(zip(names, list(ms))), "window_bits": info[0], "windows": info[1], "sub_buckets": info[2], "entries": info[3]}

    def msm_profile_totals(self, reset=False):
        self.lib.vimz_msm_profile_totals.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
        ms = (C.c_double * 6)()
        cnt = (C.c_uint64 * 3)()
        self._chk(self.lib.vimz_msm_profile_totals(self.h, ms, cnt, 1 if reset else 0))
        names = ["hist", "scan", "scatter", "accumulate", "combine", "reduce"]
        return {"ms": dict(zip(names, list(ms))), "calls": int(cnt[0]), "points": int(cnt[1]), "entries": int(cnt[2])}

    # ---- commitment key / vectors
    def bases_upload(self, curve, xy, form=L.FORM_CANONICAL):
        xy = _u64(xy)
        n = xy.size // 8
        h = C.c_void_p()
        self._chk(self.lib.vimz_bases_upload(self.h, curve, _ptr(xy), n, form, C.byref(h)))
        return Bases(self, h, curve, n)

    def bases_generate(self, curve, n, label=b"ck"):
        h = C.c_void_p()
        self._chk(self.lib.vimz_bases_generate(self.h, curve, label, len(label), n, C.byref(h)))
        return Bases(self, h, curve, n)

    def vec_alloc(self, field, n):
        h = C.c_void_p()
        self._chk(self.lib.vimz_vec_alloc(self.h, field, n, C.byref(h)))
        return DeviceVec(self, h, field, n)

    def vec_from_host(self, field, host, form=L.FORM_CANONICAL):
        host = _u64(host)
        v = self.vec_alloc(field, host.size // 4)
        return v.upload(host, 0, form)

    # ---- input pipeline
    def pack_pixels(self, image, block=0):
        """compress_by_rows (block = 0) / compress_by_blocks (block = 40) of pyvimz on the device: (H, W[, 3]) uint8 ->
        (H, ceil(W/10), 4) or (blocks, block*block/10, 4) uint64 limbs."""
        img = np.ascontiguousarray(image, dtype=np.uint8)
        if img.ndim == 3 and img.shape[2] > 3:
            img = np.ascontiguousarray(img[:, :, :3])
        h, w = img.shape[:2]
        ch = 1 if img.ndim == 2 else 3
        lib = self.lib
        lib.vimz_pack_count.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
        lib.vimz_pack_count.restype = C.c_size_t
        lib.vimz_pack_pixels.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        n = lib.vimz_pack_count(h, w, block)
        out = np.zeros((n, 4), dtype=np.uint64)
        self._chk(lib.vimz_pack_pixels(self.h, _ptr(img), h, w, ch, block, _ptr(out)))
        if block:
            per = block * ((block + 9) // 10)
            return out.reshape(-1, per, 4)
        return out.reshape(h, -1, 4)

    # ---- MSM
    def msm(self, bases, scalars, form=L.FORM_CANONICAL, window_bits=0, out_form=L.FORM_CANONICAL):
        scalars = _u64(scalars)
        out = np.zeros(8, dtype=np.uint64)
        self._chk(self.lib.vimz_msm(self.h, bases.h, _ptr(scalars), scalars.size // 4, form, window_bits, _ptr(out), out_form))
        return out

    def msm_vec(self, bases, vec, n=None, offset=0, base_offset=0, window_bits=0, out_form=L.FORM_CANONICAL, split_ones=False):
        n = vec.n - offset if n is None else n
        out = np.zeros(8, dtype=np.uint64)
        if split_ones:
            self._chk(self.lib.vimz_msm_vec_ex(self.h, bases.h, base_offset, vec.h, offset, n, window_bits, 1, _ptr(out), out_form))
        else:
            self._chk(self.lib.vimz_msm_vec(self.h, bases.h, base_offset, vec.h, offset, n, window_bits, _ptr(out), out_form))
        return out

    def kzg_open(self, srs, vec, z, n=None, offset=0, base_offset=0):
        """vimz_kzg_open: (p(z), proof point) for p(X) = sum_i vec[offset + i] X^i over the SRS powers `srs`; canonical integers."""
        n = vec.n - offset if n is None else n
        lib = self.lib
        lib.vimz_kzg_open.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        ev, pr = np.zeros(4, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        self._chk(lib.vimz_kzg_open(self.h, srs.h, base_offset, vec.h, offset, n, _ptr(_zlimbs([z], 1)), L.FORM_CANONICAL, _ptr(ev), _ptr(pr)))
        ints = lambda a: sum(int(a[k]) << (64 * k) for k in range(4))
        return ints(ev), (ints(pr[:4]), ints(pr[4:]))

    # ---- probes
    def field_op(self, field, op, a, b=None):
        a = _u64(a)
        n = a.size // 4
        b = _u64(b) if b is not None else None
        out = np.zeros((n, 4), dtype=np.uint64)
        self._chk(self.lib.vimz_field_op(self.h, field, {"add": 0, "sub": 1, "mul": 2, "inv": 3}[op], _ptr(a), _ptr(b), _ptr(out), n))
        return out

    def curve_add(self, curve, p, q):
        p, q = _u64(p), _u64(q)
        n = p.size // 8
        out = np.zeros((n, 8), dtype=np.uint64)
        self._chk(self.lib.vimz_curve_add(self.h, curve, _ptr(p), _ptr(q), _ptr(out), n))
        return out


class Prover:
    """vimz_prover: GPU folding of one transformation's step circuit (mirrors `fold_input`,
    vimz/src/nova_snark_backend/folding.rs:27-43)."""
    PHASES = ["witness", "state_chain_host", "spmv", "msm_w", "cross_term", "msm_t", "ro_host", "fold", "host_ec"]

    def __init__(self, ctx, circuit, ck, max_batch=16):
        self.ctx, self.circuit, self.ck = ctx, circuit, ck
        lib = ctx.lib
        vp, sz = C.c_void_p, C.c_size_t
        lib.vimz_prover_create.argtypes = [vp, vp, vp, sz, C.POINTER(vp)]
        lib.vimz_prover_free.argtypes = [vp]
        lib.vimz_prover_free.restype = None
        lib.vimz_prover_reset.argtypes = [vp, vp]
        lib.vimz_prover_fold.argtypes = [vp, vp, sz]
        lib.vimz_prover_verify.argtypes = [vp, C.POINTER(C.c_uint32)]
        lib.vimz_prover_instance.argtypes = [vp, vp, vp, vp, vp, C.POINTER(C.c_uint64)]
        lib.vimz_prover_running.argtypes = [vp, vp, vp]
        lib.vimz_prover_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        lib.vimz_prover_witness.argtypes = [vp, vp, sz, vp, vp, vp]
        lib.vimz_prover_spmv.argtypes = [vp, vp, vp, vp, vp]
        h = vp()
        ctx._chk(lib.vimz_prover_create(ctx.h, circuit.h, ck.h, max_batch, C.byref(h)))
        self.h = h
        self.max_batch = max_batch

    def close(self):
        if self.h:
            self.ctx.lib.vimz_prover_free(self.h)
            self.h = None

    def reset(self, z0):
        z = np.zeros((self.circuit.len_z, 4), dtype=np.uint64)
        for i, v in enumerate(z0):
            for k in range(4):
                z[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
        self.ctx._chk(self.ctx.lib.vimz_prover_reset(self.h, _ptr(z)))

    def fold(self, step_inputs):
        """step_inputs: (nsteps, n_priv, 4) uint64 canonical."""
        a = _u64(step_inputs).reshape(-1, self.circuit.n_priv, 4)
        self.ctx._chk(self.ctx.lib.vimz_prover_fold(self.h, _ptr(a), a.shape[0]))

    def fold_witness(self, witnesses):
        """witnesses: (nsteps, n_wires, 4) uint64 canonical, iden3 wire order (e.g. parsed from circom's .wtns files)."""
        a = _u64(witnesses).reshape(-1, self.circuit.n_wires, 4)
        lib = self.ctx.lib
        lib.vimz_prover_fold_witness.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.ctx._chk(lib.vimz_prover_fold_witness(self.h, _ptr(a), a.shape[0]))

    def verify(self):
        r = C.c_uint32()
        self.ctx._chk(self.ctx.lib.vimz_prover_verify(self.h, C.byref(r)))
        return r.value

    def instance(self):
        cw, ce, u = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        z = np.zeros((self.circuit.len_z, 4), dtype=np.uint64)
        steps = C.c_uint64()
        self.ctx._chk(self.ctx.lib.vimz_prover_instance(self.h, _ptr(cw), _ptr(ce), _ptr(u), _ptr(z), C.byref(steps)))
        return {"comm_W": cw, "comm_E": ce, "u": u, "z": z, "steps": steps.value}

    def running(self):
        z = np.zeros((self.circuit.n_wires, 4), dtype=np.uint64)
        E = np.zeros((self.circuit.n_constraints, 4), dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.vimz_prover_running(self.h, _ptr(z), _ptr(E)))
        return z, E

    def profile(self):
        s = (C.c_double * 9)()
        n = (C.c_uint64 * 9)()
        self.ctx._chk(self.ctx.lib.vimz_prover_profile(self.h, s, n))
        return {k: {"seconds": s[i], "count": int(n[i])} for i, k in enumerate(self.PHASES)}

    def witness(self, inputs, want_wires=True):
        a = _u64(inputs).reshape(-1, self.circuit.n_priv, 4)
        rows = a.shape[0]
        zw = np.zeros((rows, self.circuit.n_wires, 4), dtype=np.uint64) if want_wires else None
        zs = np.zeros((rows + 1, self.circuit.len_z, 4), dtype=np.uint64)
        st = np.zeros(rows, dtype=np.uint32)
        self.ctx._chk(self.ctx.lib.vimz_prover_witness(self.h, _ptr(a), rows, _ptr(zw), _ptr(zs), _ptr(st)))
        return zw, zs, st

    def spmv(self, z):
        z = _u64(z)
        n = self.circuit.n_constraints
        out = [np.zeros((n, 4), dtype=np.uint64) for _ in range(3)]
        self.ctx._chk(self.ctx.lib.vimz_prover_spmv(self.h, _ptr(z), *[_ptr(o) for o in out]))
        return out

    # ---- multi-GPU support: state chain, export / merge of running instances
    def state_chain(self, z_start, inputs):
        a = _u64(inputs).reshape(-1, self.circuit.n_priv, 4)
        n = a.shape[0]
        z = np.zeros((self.circuit.len_z, 4), dtype=np.uint64)
        for i, v in enumerate(z_start):
            for k in range(4):
                z[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
        out = np.zeros((n + 1, self.circuit.len_z, 4), dtype=np.uint64)
        lib = self.ctx.lib
        lib.vimz_prover_state_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        self.ctx._chk(lib.vimz_prover_state_chain(self.h, _ptr(z), _ptr(a), n, _ptr(out)))
        return out

    def export(self):
        lib = self.ctx.lib
        lib.vimz_prover_export_size.argtypes = [C.c_void_p]
        lib.vimz_prover_export_size.restype = C.c_size_t
        lib.vimz_prover_export.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        n = lib.vimz_prover_export_size(self.h)
        buf = np.zeros(n, dtype=np.uint8)
        self.ctx._chk(lib.vimz_prover_export(self.h, _ptr(buf), n))
        return buf

    def merge_prover(self, other):
        """Final fold with another prover on the same GPU (vimz_prover_merge_prover): no host round trip."""
        lib = self.ctx.lib
        lib.vimz_prover_merge_prover.argtypes = [C.c_void_p, C.c_void_p]
        self.ctx._chk(lib.vimz_prover_merge_prover(self.h, other.h))

    def merge(self, blob):
        lib = self.ctx.lib
        lib.vimz_prover_merge.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self.ctx._chk(lib.vimz_prover_merge(self.h, _ptr(blob), blob.size))


# ---- Nova IVC (include/vimz_hip.h: vimz_ivc_*) --------------------------------------------------------------------------
CX_R1CS = {"A_rowptr": 0, "A_col": 1, "A_coef": 2, "B_rowptr": 3, "B_col": 4, "B_coef": 5, "C_rowptr": 6, "C_col": 7, "C_coef": 8,
           "dict_canon": 10}
IX_RUNNING_Z, IX_RUNNING_E, IX_FRESH_Z, IX_INSTANCE, IX_FRESH_INSTANCE, IX_PARAMS, IX_INFO, IX_LAST_STEP = 100, 101, 102, 103, 104, 105, 106, 107


def _export(fn, *args):
    """Two-call export protocol of the ABI: size query, then copy."""
    n = fn(*args, None, 0)
    if n < 0:
        raise L.VimzError(n, "export")
    buf = np.zeros((n + 7) // 8 + 1, dtype=np.uint64)
    got = fn(*args, _ptr(buf), n)
    if got != n:
        raise L.VimzError(got, "export")
    return buf.view(np.uint8)[:n]


def _r1cs_tables(fn, *args):
    out = {}
    for name, code in CX_R1CS.items():
        a = _export(fn, *args, code)
        out[name] = a.view(np.uint64).reshape(-1, 4) if name == "dict_canon" else a.view(np.uint32)
    return out


class IVC:
    """vimz_ivc: Nova IVC of one transformation's step circuit with the augmented verifier circuits on the BN254/Grumpkin cycle
    (RecursiveSNARK::new / prove_step / verify; reference entry vimz/src/nova_snark_backend/folding.rs:27-56)."""
    PHASES = ["verifier_circuit_primary_host", "verifier_circuit_secondary_host", "wait_secondary_msm", "wait_primary_msm",
              "upload_launch", "producer_wait", "secondary_gpu", "total"]

    def __init__(self, ctx, circuit, ck_primary, ck_secondary, max_batch=16):
        self.ctx, self.circuit = ctx, circuit
        lib = ctx.lib
        vp, sz = C.c_void_p, C.c_size_t
        lib.vimz_ivc_create.argtypes = [vp, vp, vp, vp, sz, C.POINTER(vp)]
        lib.vimz_ivc_free.argtypes = [vp]
        lib.vimz_ivc_free.restype = None
        lib.vimz_ivc_reset.argtypes = [vp, vp]
        lib.vimz_ivc_fold.argtypes = [vp, vp, sz]
        lib.vimz_ivc_fold_witness.argtypes = [vp, vp, sz]
        lib.vimz_ivc_verify.argtypes = [vp, C.c_uint64, vp, C.POINTER(C.c_uint32)]
        lib.vimz_ivc_info.argtypes = [vp, vp]
        lib.vimz_ivc_state.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
        lib.vimz_ivc_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        lib.vimz_ivc_export.argtypes = [vp, C.c_int, C.c_int, vp, sz]
        lib.vimz_ivc_export.restype = C.c_int64
        h = vp()
        ctx._chk(lib.vimz_ivc_create(ctx.h, circuit.h, ck_primary.h, ck_secondary.h, max_batch, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.ctx.lib.vimz_ivc_free(self.h)
            self.h = None

    def reset(self, z0):
        z = np.zeros((self.circuit.len_z, 4), dtype=np.uint64)
        for i, v in enumerate(z0):
            for k in range(4):
                z[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
        self.ctx._chk(self.ctx.lib.vimz_ivc_reset(self.h, _ptr(z)))

    def add_msm_helper(self, helper_ctx, ck_on_helper):
        """vimz_ivc_add_msm_helper: another GPU (context) takes a base range of every step's large MSM(T)."""
        lib = self.ctx.lib
        lib.vimz_ivc_add_msm_helper.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.ctx._chk(lib.vimz_ivc_add_msm_helper(self.h, helper_ctx.h, ck_on_helper.h))

    def fold(self, step_inputs):
        a = _u64(step_inputs).reshape(-1, self.circuit.n_priv, 4)
        self.ctx._chk(self.ctx.lib.vimz_ivc_fold(self.h, _ptr(a), a.shape[0]))

    def fold_witness(self, witnesses):
        a = _u64(witnesses).reshape(-1, self.circuit.n_wires, 4)
        self.ctx._chk(self.ctx.lib.vimz_ivc_fold_witness(self.h, _ptr(a), a.shape[0]))

    def verify(self, num_steps, z0):
        """RecursiveSNARK::verify(pp, num_steps, z0): 0 = accepted (bit 12: not a proof of `num_steps` steps from `z0`)."""
        z = np.zeros((self.circuit.len_z, 4), dtype=np.uint64)
        for i, v in enumerate(z0):
            for k in range(4):
                z[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
        r = C.c_uint32()
        self.ctx._chk(self.ctx.lib.vimz_ivc_verify(self.h, int(num_steps), _ptr(z), C.byref(r)))
        return r.value

    def info(self):
        a = np.zeros(12, dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.vimz_ivc_info(self.h, _ptr(a)))
        keys = ["steps", "primary_wires", "primary_constraints", "step_wires", "step_constraints", "secondary_wires", "secondary_constraints",
                "len_z", "verifier_wires", "primary_nnz", "secondary_nnz", "head_rows"]
        return {k: int(a[i]) for i, k in enumerate(keys)}

    def state(self):
        z = np.zeros((self.circuit.len_z, 4), dtype=np.uint64)
        steps = C.c_uint64()
        self.ctx._chk(self.ctx.lib.vimz_ivc_state(self.h, _ptr(z), C.byref(steps)))
        return [sum(int(z[i, k]) << (64 * k) for k in range(4)) for i in range(self.circuit.len_z)], steps.value

    def state_chain(self, z_start, inputs):
        a = _u64(inputs).reshape(-1, self.circuit.n_priv, 4)
        n = a.shape[0]
        z = np.zeros((self.circuit.len_z, 4), dtype=np.uint64)
        for i, v in enumerate(z_start):
            for k in range(4):
                z[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
        out = np.zeros((n + 1, self.circuit.len_z, 4), dtype=np.uint64)
        lib = self.ctx.lib
        lib.vimz_ivc_state_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        self.ctx._chk(lib.vimz_ivc_state_chain(self.h, _ptr(z), _ptr(a), n, _ptr(out)))
        return out

    def digest_stride(self):
        """Elements per row of row_digests' output; 0 when this circuit's digests depend on the state (crop)."""
        lib = self.ctx.lib
        lib.vimz_ivc_digest_stride.argtypes = [C.c_void_p]
        lib.vimz_ivc_digest_stride.restype = C.c_size_t
        return int(lib.vimz_ivc_digest_stride(self.h))

    def row_digests(self, inputs):
        """Part (1) of the state chain: the state-independent row hashes of `inputs` ((n, stride, 4) uint64, opaque).  Any GPU can
        compute them for any rows: the ranks of a sharded proof hash their own rows side by side."""
        a = _u64(inputs).reshape(-1, self.circuit.n_priv, 4)
        out = np.zeros((a.shape[0], self.digest_stride(), 4), dtype=np.uint64)
        lib = self.ctx.lib
        lib.vimz_ivc_row_digests.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        self.ctx._chk(lib.vimz_ivc_row_digests(self.h, _ptr(a), a.shape[0], _ptr(out)))
        return out

    def chain_from_digests(self, z_start, inputs, digests):
        """Part (2): the serial chain over rows whose digests are known (host only); returns (n + 1, len_z, 4) states."""
        a = _u64(inputs).reshape(-1, self.circuit.n_priv, 4)
        d = np.ascontiguousarray(digests, dtype=np.uint64).reshape(a.shape[0], -1, 4)
        out = np.zeros((a.shape[0] + 1, self.circuit.len_z, 4), dtype=np.uint64)
        lib = self.ctx.lib
        lib.vimz_ivc_chain_from_digests.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        self.ctx._chk(lib.vimz_ivc_chain_from_digests(self.h, _ptr(_zlimbs(z_start, self.circuit.len_z)), _ptr(a), _ptr(d), a.shape[0], _ptr(out)))
        return out

    def profile(self):
        s = (C.c_double * 8)()
        n = (C.c_uint64 * 8)()
        self.ctx._chk(self.ctx.lib.vimz_ivc_profile(self.h, s, n))
        return {k: (s[i], n[i]) for i, k in enumerate(self.PHASES)}

    def proof_export(self):
        """The proof (and resume state) as bytes: vimz_ivc_proof_export."""
        lib = self.ctx.lib
        lib.vimz_ivc_proof_size.argtypes = [C.c_void_p]
        lib.vimz_ivc_proof_size.restype = C.c_size_t
        lib.vimz_ivc_proof_export.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        n = lib.vimz_ivc_proof_size(self.h)
        buf = np.zeros(n, dtype=np.uint8)
        self.ctx._chk(lib.vimz_ivc_proof_export(self.h, _ptr(buf), n))
        return buf

    def proof_import(self, blob):
        lib = self.ctx.lib
        lib.vimz_ivc_proof_import.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        b = np.ascontiguousarray(blob, dtype=np.uint8)
        self.ctx._chk(lib.vimz_ivc_proof_import(self.h, _ptr(b), b.size))

    def compress(self):
        """CompressedSNARK::prove: returns (proof bytes as uint8 array, {"setup_s", "prove_s"})."""
        lib = self.ctx.lib
        lib.vimz_ivc_compressed_size.argtypes = [C.c_void_p]
        lib.vimz_ivc_compressed_size.restype = C.c_size_t
        lib.vimz_ivc_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_double)]
        n = lib.vimz_ivc_compressed_size(self.h)
        buf = np.zeros(n, dtype=np.uint8)
        sec = (C.c_double * 2)()
        self.ctx._chk(lib.vimz_ivc_compress(self.h, _ptr(buf), n, sec))
        return buf, {"setup_s": sec[0], "prove_s": sec[1]}

    def verify_compressed(self, blob, num_steps, z0):
        """CompressedSNARK::verify(vk, num_steps, z0): 0 = accepted.  This IVC object only supplies the verifier key."""
        lib = self.ctx.lib
        lib.vimz_ivc_verify_compressed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint32)]
        b = np.ascontiguousarray(blob, dtype=np.uint8)
        z = np.zeros((self.circuit.len_z, 4), dtype=np.uint64)
        for i, v in enumerate(z0):
            for k in range(4):
                z[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
        r = C.c_uint32()
        self.ctx._chk(lib.vimz_ivc_verify_compressed(self.h, _ptr(b), b.size, int(num_steps), _ptr(z), C.byref(r)))
        return r.value

    def export(self, side, what):
        """(n, 4) uint64 canonical elements."""
        return _export(self.ctx.lib.vimz_ivc_export, self.h, side, what).view(np.uint64).reshape(-1, 4)

    def r1cs(self, side):
        return _r1cs_tables(self.ctx.lib.vimz_ivc_export, self.h, side)


class CycleFoldIVC:
    """vimz_cf: Nova + CycleFold IVC of one transformation's step circuit (the reference's Sonobe backend: Folding::prove_step / verify,
    vimz/src/sonobe_backend/folding.rs:52-75).  ck_main on BN254 G1 (e.g. a KZG SRS's powers), ck_cyclefold on Grumpkin."""
    PHASES = ["cross_term_msm", "cyclefold_instances", "main_circuit_host", "fresh_instance", "producer_wait", "total", "wait_row_event", "wait_verifier_commitment"]

    def __init__(self, ctx, circuit, ck_main, ck_cyclefold, max_batch=16):
        self.ctx, self.circuit = ctx, circuit
        lib = ctx.lib
        vp, sz = C.c_void_p, C.c_size_t
        lib.vimz_cf_create.argtypes = [vp, vp, vp, vp, sz, C.POINTER(vp)]
        lib.vimz_cf_free.argtypes = [vp]
        lib.vimz_cf_free.restype = None
        lib.vimz_cf_reset.argtypes = [vp, vp]
        lib.vimz_cf_fold.argtypes = [vp, vp, sz]
        lib.vimz_cf_verify.argtypes = [vp, C.c_uint64, vp, C.POINTER(C.c_uint32)]
        lib.vimz_cf_info.argtypes = [vp, vp]
        lib.vimz_cf_state.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
        lib.vimz_cf_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        lib.vimz_cf_export.argtypes = [vp, C.c_int, C.c_int, vp, sz]
        lib.vimz_cf_export.restype = C.c_int64
        h = vp()
        ctx._chk(lib.vimz_cf_create(ctx.h, circuit.h, ck_main.h, ck_cyclefold.h, max_batch, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.ctx.lib.vimz_cf_free(self.h)
            self.h = None

    def reset(self, z0):
        self.ctx._chk(self.ctx.lib.vimz_cf_reset(self.h, _ptr(_zlimbs(z0, self.circuit.len_z))))

    def fold(self, step_inputs):
        a = _u64(step_inputs).reshape(-1, self.circuit.n_priv, 4)
        self.ctx._chk(self.ctx.lib.vimz_cf_fold(self.h, _ptr(a), a.shape[0]))

    def verify(self, num_steps, z0):
        r = C.c_uint32()
        self.ctx._chk(self.ctx.lib.vimz_cf_verify(self.h, int(num_steps), _ptr(_zlimbs(z0, self.circuit.len_z)), C.byref(r)))
        return r.value

    def info(self):
        a = np.zeros(12, dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.vimz_cf_info(self.h, _ptr(a)))
        keys = ["steps", "main_wires", "main_constraints", "step_wires", "step_constraints", "cyclefold_wires", "cyclefold_constraints",
                "len_z", "verifier_wires", "main_nnz", "cyclefold_nnz", "cyclefold_io"]
        return {k: int(a[i]) for i, k in enumerate(keys)}

    def state(self):
        z = np.zeros((self.circuit.len_z, 4), dtype=np.uint64)
        steps = C.c_uint64()
        self.ctx._chk(self.ctx.lib.vimz_cf_state(self.h, _ptr(z), C.byref(steps)))
        return [sum(int(z[i, k]) << (64 * k) for k in range(4)) for i in range(self.circuit.len_z)], steps.value

    def profile(self):
        s = (C.c_double * 8)()
        n = (C.c_uint64 * 8)()
        self.ctx._chk(self.ctx.lib.vimz_cf_profile(self.h, s, n))
        return {k: (s[i], n[i]) for i, k in enumerate(self.PHASES)}

    def export(self, side, what):
        return _export(self.ctx.lib.vimz_cf_export, self.h, side, what).view(np.uint64).reshape(-1, 4)

    def r1cs(self, side):
        return _r1cs_tables(self.ctx.lib.vimz_cf_export, self.h, side)

    def poke(self, which, index, value):
        """Test hook (vimz_cf_poke, include/vimz_hip_testing.h): exists only when the process runs on libvimz_hip_testing.so
        (VIMZ_HIP_LIBRARY=testing); the product library has no way to overwrite a prover's vectors."""
        lib = self.ctx.lib
        if not hasattr(lib, "vimz_cf_poke"):
            raise L.VimzError(L.ERR_INVALID, "vimz_cf_poke is a test hook: start the process with VIMZ_HIP_LIBRARY=testing")
        lib.vimz_cf_poke.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
        self.ctx._chk(lib.vimz_cf_poke(self.h, which, index, _ptr(_zlimbs([value], 1))))

    def state_chain(self, z_start, inputs):
        a = _u64(inputs).reshape(-1, self.circuit.n_priv, 4)
        out = np.zeros((a.shape[0] + 1, self.circuit.len_z, 4), dtype=np.uint64)
        lib = self.ctx.lib
        lib.vimz_cf_state_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        self.ctx._chk(lib.vimz_cf_state_chain(self.h, _ptr(_zlimbs(z_start, self.circuit.len_z)), _ptr(a), a.shape[0], _ptr(out)))
        return out

    def kzg_open(self, which, z):
        """vimz_cf_kzg_open: (eval, proof point) of the running main instance's comm_W (which = 0) or comm_E (1) at z."""
        lib = self.ctx.lib
        lib.vimz_cf_kzg_open.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        ev, pr = np.zeros(4, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        self.ctx._chk(lib.vimz_cf_kzg_open(self.h, which, _ptr(_zlimbs([z], 1)), _ptr(ev), _ptr(pr)))
        ints = lambda a: sum(int(a[k]) << (64 * k) for k in range(4))
        return ints(ev), (ints(pr[:4]), ints(pr[4:]))

    def digest_stride(self):
        lib = self.ctx.lib
        lib.vimz_cf_digest_stride.argtypes = [C.c_void_p]
        lib.vimz_cf_digest_stride.restype = C.c_size_t
        return int(lib.vimz_cf_digest_stride(self.h))

    def row_digests(self, inputs):
        a = _u64(inputs).reshape(-1, self.circuit.n_priv, 4)
        out = np.zeros((a.shape[0], self.digest_stride(), 4), dtype=np.uint64)
        lib = self.ctx.lib
        lib.vimz_cf_row_digests.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        self.ctx._chk(lib.vimz_cf_row_digests(self.h, _ptr(a), a.shape[0], _ptr(out)))
        return out

    def chain_from_digests(self, z_start, inputs, digests):
        a = _u64(inputs).reshape(-1, self.circuit.n_priv, 4)
        d = np.ascontiguousarray(digests, dtype=np.uint64).reshape(a.shape[0], -1, 4)
        out = np.zeros((a.shape[0] + 1, self.circuit.len_z, 4), dtype=np.uint64)
        lib = self.ctx.lib
        lib.vimz_cf_chain_from_digests.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        self.ctx._chk(lib.vimz_cf_chain_from_digests(self.h, _ptr(_zlimbs(z_start, self.circuit.len_z)), _ptr(a), _ptr(d), a.shape[0], _ptr(out)))
        return out

    def proof_export(self):
        """The proof (and resume state) as bytes: vimz_cf_proof_export."""
        lib = self.ctx.lib
        lib.vimz_cf_proof_size.argtypes = [C.c_void_p]
        lib.vimz_cf_proof_size.restype = C.c_size_t
        lib.vimz_cf_proof_export.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        n = lib.vimz_cf_proof_size(self.h)
        buf = np.zeros(n, dtype=np.uint8)
        self.ctx._chk(lib.vimz_cf_proof_export(self.h, _ptr(buf), n))
        return buf

    def proof_import(self, blob):
        lib = self.ctx.lib
        lib.vimz_cf_proof_import.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        b = np.ascontiguousarray(blob, dtype=np.uint8)
        self.ctx._chk(lib.vimz_cf_proof_import(self.h, _ptr(b), b.size))


def cyclefold_selfcheck_last_step(steps=4):
    """vimz_cf_selfcheck_last_step (host only): (digest, z_0, uint64 words of the last step's VIMZ_IX_LAST_STEP record)."""
    from . import _lib
    lib = _lib.testing_lib()
    lib.vimz_cf_selfcheck_last_step.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
    lib.vimz_cf_selfcheck_last_step.restype = C.c_int64
    n = lib.vimz_cf_selfcheck_last_step(int(steps), None, 0)
    if n < 0:
        raise _lib.VimzError(int(n), "vimz_cf_selfcheck_last_step")
    buf = np.zeros(n // 8, dtype=np.uint64)
    assert lib.vimz_cf_selfcheck_last_step(int(steps), _ptr(buf), n) == n
    ints = lambda a: sum(int(a[k]) << (64 * k) for k in range(4))
    return ints(buf[0:4]), ints(buf[4:8]), buf[8:]


def cyclefold_selfcheck_merge(segs_run0=3, segs_run1=2):
    """vimz_cf_selfcheck_merge (host only): (digest, record words, accumulator dict) of made-up segment records replayed by the library."""
    from . import _lib
    lib = _lib.testing_lib()
    lib.vimz_cf_selfcheck_merge.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.vimz_cf_selfcheck_merge.restype = C.c_int64
    n = lib.vimz_cf_selfcheck_merge(segs_run0, segs_run1, None, 0)
    if n < 0:
        raise _lib.VimzError(int(n), "vimz_cf_selfcheck_merge")
    buf = np.zeros(n // 8, dtype=np.uint64)
    assert lib.vimz_cf_selfcheck_merge(segs_run0, segs_run1, _ptr(buf), n) == n
    w = [int(x) for x in buf]
    el = lambda pos: sum(w[pos + k] << (64 * k) for k in range(4))
    dg, nrec = el(0), w[4]
    rec = buf[5:5 + nrec]
    pos = 5 + nrec
    acc = {"n": w[pos], "zs": [el(pos + 1)], "ze": [el(pos + 5)]}
    pos += 9
    vals = [el(pos + 4 * k) for k in range(7 + 12)]
    acc["P"] = [(vals[0], vals[1]), (vals[2], vals[3]), vals[4], vals[5], vals[6]]
    acc["Q"] = [(vals[7], vals[8]), (vals[9], vals[10]), vals[11], vals[12:19]]
    return dg, rec, acc


def kzg_setup(ctx, n, seed=None):
    """KZG::setup: (srs as Bases — use it as ck_main of a CycleFoldIVC —, vk_g2 = [tau]G2 as a (4, 4) uint64 array for Decider).  tau comes from the
    OS's randomness and is forgotten; `seed` (bytes) selects the deterministic TEST setup of libvimz_hip_testing.so instead."""
    vp = C.c_void_p
    h = vp()
    vk = np.zeros((4, 4), dtype=np.uint64)
    if seed is None:
        ctx.lib.vimz_kzg_setup.argtypes = [vp, C.c_size_t, C.POINTER(vp), vp]
        ctx._chk(ctx.lib.vimz_kzg_setup(ctx.h, n, C.byref(h), _ptr(vk)))
    else:
        fn = _seeded(ctx, "vimz_testing_kzg_setup_seeded")
        fn.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(vp), vp]
        ctx._chk(fn(ctx.h, bytes(seed), len(seed), n, C.byref(h), _ptr(vk)))
    return Bases(ctx, h, L.CURVE_BN254_G1, n), vk


def _seeded(ctx, name):
    if not hasattr(ctx.lib, name):
        raise RuntimeError(f"{name} exists only in libvimz_hip_testing.so (start the process with VIMZ_HIP_LIBRARY=testing): seeded setups are test hooks")
    return getattr(ctx.lib, name)


def set_head_rows(rows):
    """vimz_set_head_rows: rows of a short fold call whose Poseidon chains run on the host (0: all on the GPU; -1: the library's policy)."""
    lib = L.lib()
    lib.vimz_set_head_rows.argtypes = [C.c_long]
    lib.vimz_set_head_rows.restype = C.c_long
    return int(lib.vimz_set_head_rows(int(rows)))


class Decider:
    """vimz_decider: `Decider::preprocess` / `prove` / `verify` of the Sonobe backend (vimz/src/sonobe_backend/mod.rs:72-80) — the 25 calldata words of
    contracts/*Verifier.sol and their local verification.  prover: a CycleFoldIVC (shapes, keys, context; keep it open); kzg_vk: [tau]G2 of the SRS
    its ck_main is made of (kzg_setup), or None (no verify).  light=False: the FULL decider (decider.rs:13-21: the running CycleFold instance's commitments
    and relation are checked inside the circuit); light=True: the reference's opt-in `light-test` variant (vimz/Cargo.toml:56-59).  The Groth16 trapdoor is drawn from the OS's randomness and forgotten; `seed` selects
    the deterministic TEST setup of libvimz_hip_testing.so."""
    RESULT_BITS = {1: "fewer than two steps", 2: "KZG opening of cmW", 4: "KZG opening of cmE", 8: "Groth16", 16: "a word pair is not a curve point"}

    def __init__(self, prover, kzg_vk=None, seed=None, light=False):
        self.prover, self.ctx = prover, prover.ctx
        self.light = bool(light)
        lib = self.ctx.lib
        vp = C.c_void_p
        lib.vimz_decider_setup.argtypes = [vp, vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_double)]
        lib.vimz_decider_free.argtypes = [vp]
        lib.vimz_decider_free.restype = None
        lib.vimz_decider_info.argtypes = [vp, vp]
        lib.vimz_decider_vk.argtypes = [vp, vp, C.c_size_t]
        lib.vimz_decider_vk.restype = C.c_int64
        lib.vimz_decider_prove.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_double)]
        lib.vimz_decider_verify.argtypes = [vp, C.c_uint64, vp, vp, vp, C.POINTER(C.c_uint32)]
        h = vp()
        sec = (C.c_double * 4)()
        self._kzg_vk = None if kzg_vk is None else np.ascontiguousarray(kzg_vk, dtype=np.uint64)
        kp = None if self._kzg_vk is None else _ptr(self._kzg_vk)
        if seed is None:
            self.ctx._chk(lib.vimz_decider_setup(prover.h, kp, int(self.light), C.byref(h), sec))
        else:
            fn = _seeded(self.ctx, "vimz_testing_decider_setup_seeded")
            fn.argtypes = [vp, vp, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_double)]
            self.ctx._chk(fn(prover.h, kp, int(self.light), bytes(seed), len(seed), C.byref(h), sec))
        self.h = h
        self.setup_seconds = {"circuit_synthesis": sec[0], "qap_at_trapdoor_host": sec[1], "key_points_gpu": sec[2], "total": sec[3]}

    def close(self):
        if self.h:
            self.ctx.lib.vimz_decider_free(self.h)
            self.h = None

    def save_key(self):
        """vimz_decider_key_save: the key pair as bytes (uint8 array; 0.5 GB at contrast HD)."""
        lib = self.ctx.lib
        lib.vimz_decider_key_save.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.vimz_decider_key_save.restype = C.c_int64
        n = lib.vimz_decider_key_save(self.h, None, 0)
        if n < 0:
            self.ctx._chk(int(n))
        buf = np.zeros(n, dtype=np.uint8)
        got = lib.vimz_decider_key_save(self.h, _ptr(buf), n)
        if got != n:
            self.ctx._chk(int(got) if got < 0 else L.ERR_INVALID)
        return buf

    @classmethod
    def load_key(cls, prover, blob):
        """vimz_decider_key_load: a decider over `prover` (a CycleFoldIVC) from a saved — or externally made — key pair; no trapdoor is involved."""
        d = cls.__new__(cls)
        d.prover, d.ctx = prover, prover.ctx
        lib = d.ctx.lib
        vp = C.c_void_p
        lib.vimz_decider_key_load.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp)]
        lib.vimz_decider_free.argtypes = [vp]
        lib.vimz_decider_free.restype = None
        lib.vimz_decider_info.argtypes = [vp, vp]
        lib.vimz_decider_vk.argtypes = [vp, vp, C.c_size_t]
        lib.vimz_decider_vk.restype = C.c_int64
        lib.vimz_decider_prove.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_double)]
        lib.vimz_decider_verify.argtypes = [vp, C.c_uint64, vp, vp, vp, C.POINTER(C.c_uint32)]
        b = np.ascontiguousarray(blob, dtype=np.uint8)
        h = vp()
        d.ctx._chk(lib.vimz_decider_key_load(prover.h, _ptr(b), b.size, C.byref(h)))
        d.h = h
        d._kzg_vk = None
        d.setup_seconds = {"circuit_synthesis": 0.0, "qap_at_trapdoor_host": 0.0, "key_points_gpu": 0.0, "total": 0.0}
        return d

    def info(self):
        a = np.zeros(8, dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.vimz_decider_info(self.h, _ptr(a)))
        return dict(zip(["constraints", "wires", "public_inputs", "domain", "nnz_a", "nnz_b", "nnz_c", "cyclefold_rows"], (int(x) for x in a[:8])))

    def key_words(self):
        """vimz_decider_vk's words (uint64 array): what vimz_decider_verify_key takes."""
        return _export(self.ctx.lib.vimz_decider_vk, self.h).view(np.uint64)

    def verifying_key(self):
        """The constants of a contract generated for this circuit, as Python integers, in the shape tests/_novadecider.py's `verify` takes:
        {"len_z", "pp_hash", "groth16": {"alpha": G1, "beta", "gamma", "delta": G2, "ic": [G1 ...]}, "kzg": {"G_1": G1, "G_2": G2, "VK": G2}};
        G2 points ((x.c0, x.c1), (y.c0, y.c1)) = (real, imaginary)."""
        return parse_verifying_key(self.key_words())

    def prove(self, ivc=None):
        """Decider::prove for the IVC proof `ivc` (default: the prover the decider was set up over) holds.  Returns (25 words: ints, public inputs:
        ints, seconds dict)."""
        ivc = self.prover if ivc is None else ivc
        n_pub = self.info()["public_inputs"]
        pub = np.zeros((n_pub, 4), dtype=np.uint64)
        words = np.zeros((25, 4), dtype=np.uint64)
        sec = (C.c_double * 6)()
        self.ctx._chk(self.ctx.lib.vimz_decider_prove(self.h, ivc.h, _ptr(pub), _ptr(words), sec))
        ints = lambda a: [sum(int(a[i, q]) << (64 * q) for q in range(4)) for i in range(a.shape[0])]
        return ints(words), ints(pub), {"final_fold_and_kzg": sec[0], "witness_host": sec[1], "ntt": sec[2], "msm": sec[3], "total": sec[4]}

    def verify(self, steps, z0, z_i, words):
        """Decider::verify: 0 = accepted, else the bit set of RESULT_BITS."""
        z0a, zia, wa = _zlimbs([int(x) for x in z0], len(z0)), _zlimbs([int(x) for x in z_i], len(z_i)), _zlimbs([int(x) for x in words], 25)
        res = C.c_uint32(0)
        self.ctx._chk(self.ctx.lib.vimz_decider_verify(self.h, int(steps), _ptr(z0a), _ptr(zia), _ptr(wa), C.byref(res)))
        return int(res.value)


def parse_verifying_key(w):
    w = [int(x) for x in w]
    pos = [0]

    def el():
        v = sum(w[pos[0] + q] << (64 * q) for q in range(4))
        pos[0] += 4
        return v

    def g1():
        return (el(), el())

    def g2():
        return ((el(), el()), (el(), el()))

    pp = el()
    len_z = w[pos[0]]
    pos[0] += 1
    alpha, beta, gamma, delta = g1(), g2(), g2(), g2()
    n_ic = w[pos[0]]
    pos[0] += 1
    ic = [g1() for _ in range(n_ic)]
    kz = {"G_1": g1(), "G_2": g2(), "VK": g2()}
    assert pos[0] == len(w)
    return {"len_z": len_z, "pp_hash": pp, "groth16": {"alpha": alpha, "beta": beta, "gamma": gamma, "delta": delta, "ic": ic}, "kzg": kz}


def verifying_key_words(key):
    """The inverse of parse_verifying_key: a key in tests/_novadecider.py's shape as the words vimz_decider_verify_key takes."""
    out = []

    def el(v):
        out.extend((int(v) >> (64 * q)) & 0xFFFFFFFFFFFFFFFF for q in range(4))

    def g1(p):
        el(p[0]); el(p[1])

    def g2(p):
        el(p[0][0]); el(p[0][1]); el(p[1][0]); el(p[1][1])

    g = key["groth16"]
    el(key["pp_hash"]); out.append(int(key["len_z"]))
    g1(g["alpha"]); g2(g["beta"]); g2(g["gamma"]); g2(g["delta"])
    out.append(len(g["ic"]))
    for p in g["ic"]:
        g1(p)
    g1(key["kzg"]["G_1"]); g2(key["kzg"]["G_2"]); g2(key["kzg"]["VK"])
    return np.array(out, dtype=np.uint64)


def decider_verify_key(key_words, steps, z0, z_i, words):
    """vimz_decider_verify_key: host only, no context, no GPU.  Returns the result bits (0 = accepted); raises VimzError on a malformed key."""
    lib = L.lib()
    vp = C.c_void_p
    lib.vimz_decider_verify_key.argtypes = [vp, C.c_size_t, C.c_uint64, vp, vp, C.c_uint32, vp, C.POINTER(C.c_uint32)]
    kw = np.ascontiguousarray(key_words, dtype=np.uint64)
    z0a, zia, wa = _zlimbs([int(x) for x in z0], len(z0)), _zlimbs([int(x) for x in z_i], len(z_i)), _zlimbs([int(x) for x in words], 25)
    res = C.c_uint32(0)
    rc = lib.vimz_decider_verify_key(_ptr(kw), kw.size, int(steps), _ptr(z0a), _ptr(zia), len(z0), _ptr(wa), C.byref(res))
    if rc:
        raise L.VimzError(rc, "vimz_decider_verify_key: malformed key or arguments")
    return int(res.value)


class PendingFold:
    """vimz_ivc_pending: the segments' folds of a run of rows, begun before the state they start from is known (MergedProof.fold_segments_begin)."""

    def __init__(self, merged_cls, ivcs, step_inputs):
        self.merged_cls, self.vk, self.ctx = merged_cls, ivcs[0], ivcs[0].ctx
        lib = self.ctx.lib
        vp = C.c_void_p
        lib.vimz_ivc_fold_segments_begin.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(vp)]
        lib.vimz_ivc_pending_digests.argtypes = [vp, vp]
        lib.vimz_ivc_pending_start.argtypes = [vp, vp]
        lib.vimz_ivc_pending_finish.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_double)]
        self.rows = np.ascontiguousarray(_u64(step_inputs).reshape(-1, self.vk.circuit.n_priv, 4))      # (kept alive: the folds read it until finish)
        arr = (vp * len(ivcs))(*[v.h for v in ivcs])
        h = vp()
        self.ctx._chk(lib.vimz_ivc_fold_segments_begin(arr, len(ivcs), _ptr(self.rows), self.rows.shape[0], C.byref(h)))
        self.h = h

    def digests(self):
        """(rows, stride, 4) uint64: the rows' digests, as IVC.row_digests returns them.  Blocks until the folds' chain passes have produced them."""
        stride = self.vk.digest_stride()
        out = np.zeros((self.rows.shape[0], stride, 4), dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.vimz_ivc_pending_digests(self.h, _ptr(out)))
        return out

    def start(self, z_start):
        self.ctx._chk(self.ctx.lib.vimz_ivc_pending_start(self.h, _ptr(_zlimbs(z_start, self.vk.circuit.len_z))))

    def finish(self):
        """Joins the folds and merges: (MergedProof, {"merge_s", "total_s"})."""
        h, self.h = self.h, None
        mh = C.c_void_p()
        sec = (C.c_double * 3)()
        self.ctx._chk(self.ctx.lib.vimz_ivc_pending_finish(h, C.byref(mh), sec))
        m = self.merged_cls.__new__(self.merged_cls)
        self.merged_cls.__init__(m, _handle=mh, _vk=self.vk)
        return m, {"state_chain_s": 0.0, "merge_s": sec[1], "total_s": sec[2]}

    def cancel(self):
        if self.h:
            h, self.h = self.h, None
            self.ctx.lib.vimz_ivc_pending_finish(h, None, None)


def head_rows_policy(nsteps, segments=False):
    """vimz_head_rows_policy(_segments): rows of a lone fold call of nsteps rows (segments=True: of a proof of nsteps rows made as concurrent segments) whose
    Poseidon chains the library would evaluate on the host."""
    lib = L.lib()
    fn = lib.vimz_head_rows_policy_segments if segments else lib.vimz_head_rows_policy
    fn.argtypes = [C.c_size_t]
    fn.restype = C.c_size_t
    return int(fn(int(nsteps)))


class CycleFoldMerged:
    """vimz_cf_merged: ONE proof object out of the CycleFold proofs of contiguous row segments (vimz_cf_merge).  `first`: the prover of the
    first segment (left unchanged; supplies shapes, keys and context and must stay open)."""
    PHASES = ["cross_terms_and_commitments", "folds", "host", "total"]

    def __init__(self, first=None, _handle=None, _vk=None):
        self.vk = first if first is not None else _vk
        self.ctx = self.vk.ctx
        lib = self.ctx.lib
        vp, sz = C.c_void_p, C.c_size_t
        lib.vimz_cf_merge_merged.argtypes = [vp, vp]
        lib.vimz_cf_merged_size.argtypes = [vp]
        lib.vimz_cf_merged_size.restype = sz
        lib.vimz_cf_merged_save.argtypes = [vp, vp, sz]
        lib.vimz_cf_merged_load.argtypes = [vp, vp, sz, C.POINTER(vp)]
        lib.vimz_cf_merged_create.argtypes = [vp, C.POINTER(vp)]
        lib.vimz_cf_merged_free.argtypes = [vp]
        lib.vimz_cf_merged_free.restype = None
        lib.vimz_cf_merge.argtypes = [vp, vp]
        lib.vimz_cf_merged_verify.argtypes = [vp, C.c_uint64, vp, C.POINTER(C.c_uint32)]
        lib.vimz_cf_merged_info.argtypes = [vp, vp]
        lib.vimz_cf_merged_state.argtypes = [vp, vp, vp, C.POINTER(C.c_uint64)]
        lib.vimz_cf_merged_profile.argtypes = [vp, C.POINTER(C.c_double)]
        lib.vimz_cf_merged_records.argtypes = [vp, vp, sz]
        lib.vimz_cf_merged_records.restype = C.c_int64
        lib.vimz_cf_merged_export.argtypes = [vp, C.c_int, C.c_int, vp, sz]
        lib.vimz_cf_merged_export.restype = C.c_int64
        if _handle is not None:
            self.h = _handle
            return
        h = vp()
        self.ctx._chk(lib.vimz_cf_merged_create(first.h, C.byref(h)))
        self.h = h

    @classmethod
    def load(cls, vk, blob):
        """vimz_cf_merged_load: a merged proof from bytes, into the context of `vk` (a CycleFoldIVC for the same step circuit and keys)."""
        lib = vk.ctx.lib
        lib.vimz_cf_merged_load.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        b = np.ascontiguousarray(blob, dtype=np.uint8)
        if b.ctypes.data % 8:
            b = np.frombuffer(bytearray(b.tobytes()) + bytearray(8), dtype=np.uint8)[:b.size].copy()
        h = C.c_void_p()
        vk.ctx._chk(lib.vimz_cf_merged_load(vk.h, _ptr(b), b.size, C.byref(h)))
        return cls(_handle=h, _vk=vk)

    def save(self):
        """vimz_cf_merged_save: the object as bytes (records + folded vectors)."""
        n = self.ctx.lib.vimz_cf_merged_size(self.h)
        buf = np.zeros(n, dtype=np.uint8)
        self.ctx._chk(self.ctx.lib.vimz_cf_merged_save(self.h, _ptr(buf), n))
        return buf

    @classmethod
    def of(cls, provers):
        """The merged proof of segments proven by `provers`, in row order."""
        m = cls(provers[0])
        try:
            for v in provers[1:]:
                m.merge(v)
        except Exception:
            m.close()
            raise
        return m

    def close(self):
        if self.h:
            self.ctx.lib.vimz_cf_merged_free(self.h)
            self.h = None

    def merge(self, nxt):
        """Fold the next row segment's proof in (a CycleFoldIVC), or another merged object holding one run (vimz_cf_merge_merged)."""
        if isinstance(nxt, CycleFoldMerged):
            self.ctx._chk(self.ctx.lib.vimz_cf_merge_merged(self.h, nxt.h))
        else:
            self.ctx._chk(self.ctx.lib.vimz_cf_merge(self.h, nxt.h))

    def verify(self, num_steps, z0):
        r = C.c_uint32()
        self.ctx._chk(self.ctx.lib.vimz_cf_merged_verify(self.h, int(num_steps), _ptr(_zlimbs(z0, self.vk.circuit.len_z)), C.byref(r)))
        return r.value

    def info(self):
        a = np.zeros(8, dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.vimz_cf_merged_info(self.h, _ptr(a)))
        return dict(zip(["steps", "segments", "len_z", "main_wires", "main_constraints", "cyclefold_wires", "cyclefold_constraints", "broken"], (int(x) for x in a)))

    def state(self):
        """(z_start, z_end, steps)"""
        lz = self.vk.circuit.len_z
        zs, ze = np.zeros((lz, 4), dtype=np.uint64), np.zeros((lz, 4), dtype=np.uint64)
        steps = C.c_uint64()
        self.ctx._chk(self.ctx.lib.vimz_cf_merged_state(self.h, _ptr(zs), _ptr(ze), C.byref(steps)))
        ints = lambda a: [sum(int(a[i, k]) << (64 * k) for k in range(4)) for i in range(lz)]
        return ints(zs), ints(ze), steps.value

    def profile(self):
        s = (C.c_double * 4)()
        self.ctx._chk(self.ctx.lib.vimz_cf_merged_profile(self.h, s))
        return dict(zip(self.PHASES, s))

    def share(self):
        """vimz_cf_merged_share: the ticket (records + HIP IPC handle) another process of this node opens with CycleFoldMerged.open_shared."""
        lib = self.ctx.lib
        lib.vimz_cf_merged_share.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.vimz_cf_merged_share.restype = C.c_int64
        return _export(lib.vimz_cf_merged_share, self.h)

    @classmethod
    def open_shared(cls, vk, ticket):
        """vimz_cf_merged_open_shared: an object of this process's own out of another process's shared one (device-to-device copy)."""
        b = np.ascontiguousarray(np.frombuffer(bytes(ticket), dtype=np.uint8) if isinstance(ticket, (bytes, bytearray)) else ticket, dtype=np.uint8)
        lib = vk.ctx.lib
        lib.vimz_cf_merged_open_shared.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        vk.ctx._chk(lib.vimz_cf_merged_open_shared(vk.h, _ptr(b), b.size, C.byref(h)))
        return cls(_handle=h, _vk=vk)

    def kzg_open(self, which, z):
        """vimz_cf_merged_kzg_open: (eval, proof point) of the folded main instance's comm_W (which = 0) or comm_E (1) at z."""
        lib = self.ctx.lib
        lib.vimz_cf_merged_kzg_open.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        ev, pr = np.zeros(4, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        self.ctx._chk(lib.vimz_cf_merged_kzg_open(self.h, which, _ptr(_zlimbs([z], 1)), _ptr(ev), _ptr(pr)))
        ints = lambda a: sum(int(a[k]) << (64 * k) for k in range(4))
        return ints(ev), (ints(pr[:4]), ints(pr[4:]))

    def records(self):
        n = self.ctx.lib.vimz_cf_merged_records(self.h, None, 0)
        buf = np.zeros(n // 8, dtype=np.uint64)
        assert self.ctx.lib.vimz_cf_merged_records(self.h, _ptr(buf), n) == n
        return buf

    def export(self, side, what):
        return _export(self.ctx.lib.vimz_cf_merged_export, self.h, side, what).view(np.uint64).reshape(-1, 4)


def cyclefold_selfcheck(steps=4):
    """vimz_cf_selfcheck (host only, no GPU): (result bits, {"main_wires", "main_constraints", "cyclefold_wires", "cyclefold_constraints"})."""
    from . import _lib
    lib = _lib.testing_lib()
    lib.vimz_cf_selfcheck.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.c_void_p]
    r = C.c_uint32()
    counts = np.zeros(8, dtype=np.uint64)
    rc = lib.vimz_cf_selfcheck(int(steps), C.byref(r), _ptr(counts))
    if rc:
        raise _lib.VimzError(rc, "vimz_cf_selfcheck")
    return r.value, dict(zip(["main_wires", "main_constraints", "cyclefold_wires", "cyclefold_constraints", "main_flipped", "main_unnoticed", "cyclefold_flipped", "cyclefold_unnoticed"],
                             (int(x) for x in counts)))


def _zlimbs(z0, len_z):
    z = np.zeros((len_z, 4), dtype=np.uint64)
    for i, v in enumerate(z0):
        for k in range(4):
            z[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return z


class MergedProof:
    """vimz_ivc_merged: ONE proof object out of the IVC proofs of contiguous row segments (the host-side sequential final fold of
    BASELINE.json's north_star; the reference's fold_input returns one RecursiveSNARK, vimz/src/nova_snark_backend/folding.rs:27-43).
    `first`: the IVC of the first segment (left unchanged; supplies shapes, keys and context and must stay open)."""
    PHASES = ["leaf", "wait_cross_term_msm", "folds_and_host", "total"]

    def __init__(self, first=None, _handle=None, _vk=None):
        self.vk = first if first is not None else _vk
        self.ctx = self.vk.ctx
        lib = self.ctx.lib
        vp, sz = C.c_void_p, C.c_size_t
        lib.vimz_ivc_merged_create.argtypes = [vp, C.POINTER(vp)]
        lib.vimz_ivc_merged_free.argtypes = [vp]
        lib.vimz_ivc_merged_free.restype = None
        lib.vimz_ivc_merge.argtypes = [vp, vp]
        lib.vimz_ivc_merge_merged.argtypes = [vp, vp]
        lib.vimz_ivc_merged_verify.argtypes = [vp, C.c_uint64, vp, C.POINTER(C.c_uint32)]
        lib.vimz_ivc_merged_info.argtypes = [vp, vp]
        lib.vimz_ivc_merged_state.argtypes = [vp, vp, vp, C.POINTER(C.c_uint64)]
        lib.vimz_ivc_merged_profile.argtypes = [vp, C.POINTER(C.c_double)]
        lib.vimz_ivc_merged_size.argtypes = [vp]
        lib.vimz_ivc_merged_size.restype = sz
        lib.vimz_ivc_merged_save.argtypes = [vp, vp, sz]
        lib.vimz_ivc_merged_load.argtypes = [vp, vp, sz, C.POINTER(vp)]
        lib.vimz_ivc_merged_records.argtypes = [vp, vp, sz]
        lib.vimz_ivc_merged_records.restype = C.c_int64
        lib.vimz_ivc_merged_export.argtypes = [vp, C.c_int, C.c_int, vp, sz]
        lib.vimz_ivc_merged_export.restype = C.c_int64
        lib.vimz_ivc_merged_compressed_size.argtypes = [vp]
        lib.vimz_ivc_merged_compressed_size.restype = sz
        lib.vimz_ivc_merged_compress.argtypes = [vp, vp, sz, C.POINTER(C.c_double)]
        lib.vimz_ivc_verify_merged_compressed.argtypes = [vp, vp, sz, C.c_uint64, vp, C.POINTER(C.c_uint32)]
        if _handle is not None:
            self.h = _handle
        else:
            h = vp()
            self.ctx._chk(lib.vimz_ivc_merged_create(first.h, C.byref(h)))
            self.h = h

    @classmethod
    def of(cls, ivcs):
        """The merged proof of the segments' IVCs, in row order."""
        m = cls(ivcs[0])
        try:
            for v in ivcs[1:]:
                m.merge(v)
        except Exception:
            m.close()
            raise
        return m

    @classmethod
    def fold_segments(cls, ivcs, step_inputs, z0, digests=None):
        """vimz_ivc_fold_segments: fold_input in one call — the rows as len(ivcs) concurrent segments (own context each), merged.
        digests (optional): the rows' digests as vimz_ivc_row_digests returns them, when the caller has them already.
        Returns (MergedProof, {"state_chain_s", "merge_s", "total_s"}); its verifier key is ivcs[0]."""
        vk = ivcs[0]
        lib = vk.ctx.lib
        lib.vimz_ivc_fold_segments_dg.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_double)]
        a = _u64(step_inputs).reshape(-1, vk.circuit.n_priv, 4)
        arr = (C.c_void_p * len(ivcs))(*[v.h for v in ivcs])
        h = C.c_void_p()
        sec = (C.c_double * 3)()
        dg = None
        if digests is not None:
            dg = np.ascontiguousarray(digests, dtype=np.uint64)
            if dg.size != a.shape[0] * vk.digest_stride() * 4:
                raise ValueError("fold_segments: digests do not match the rows")
        vk.ctx._chk(lib.vimz_ivc_fold_segments_dg(arr, len(ivcs), _ptr(_zlimbs(z0, vk.circuit.len_z)), _ptr(a), a.shape[0], _ptr(dg) if dg is not None else None, C.byref(h), sec))
        m = cls.__new__(cls)
        cls.__init__(m, _handle=h, _vk=vk)
        return m, {"state_chain_s": sec[0], "merge_s": sec[1], "total_s": sec[2]}

    @classmethod
    def fold_segments_begin(cls, ivcs, step_inputs):
        """vimz_ivc_fold_segments_begin: fold_input for a run of rows whose start state is not known yet (a rank of a sharded proof).  Returns a
        PendingFold: .digests() blocks until the rows are hashed (by the folds' own chain passes), .start(z) provides the state, .finish() -> (MergedProof, seconds)."""
        return PendingFold(cls, ivcs, step_inputs)

    @classmethod
    def load(cls, vk, blob):
        """vimz_ivc_merged_load: `vk` = an IVC for the same step circuit and keys (its state is not touched)."""
        b = np.ascontiguousarray(blob, dtype=np.uint8)
        m = cls.__new__(cls)
        cls.__init__(m, _handle=C.c_void_p(0), _vk=vk)
        h = C.c_void_p()
        vk.ctx._chk(vk.ctx.lib.vimz_ivc_merged_load(vk.h, _ptr(b), b.size, C.byref(h)))
        m.h = h
        return m

    def share(self):
        """vimz_ivc_merged_share: the ticket (records + HIP IPC handle of the device allocation) another process of this node opens
        with MergedProof.open_shared; keep this object open until that process is done."""
        lib = self.ctx.lib
        lib.vimz_ivc_merged_share.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.vimz_ivc_merged_share.restype = C.c_int64
        return _export(lib.vimz_ivc_merged_share, self.h)

    @classmethod
    def open_shared(cls, vk, ticket):
        """vimz_ivc_merged_open_shared: an object of this process's own out of another process's shared one (device-to-device copy)."""
        b = np.ascontiguousarray(np.frombuffer(bytes(ticket), dtype=np.uint8) if isinstance(ticket, (bytes, bytearray)) else ticket, dtype=np.uint8)
        m = cls.__new__(cls)
        cls.__init__(m, _handle=C.c_void_p(0), _vk=vk)
        lib = vk.ctx.lib
        lib.vimz_ivc_merged_open_shared.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        vk.ctx._chk(lib.vimz_ivc_merged_open_shared(vk.h, _ptr(b), b.size, C.byref(h)))
        m.h = h
        return m

    def close(self):
        if self.h:
            self.ctx.lib.vimz_ivc_merged_free(self.h)
            self.h = None

    def merge(self, nxt):
        """Fold the next row segment in: an IVC (vimz_ivc_merge) or another merged proof (vimz_ivc_merge_merged)."""
        if isinstance(nxt, MergedProof):
            self.ctx._chk(self.ctx.lib.vimz_ivc_merge_merged(self.h, nxt.h))
        else:
            self.ctx._chk(self.ctx.lib.vimz_ivc_merge(self.h, nxt.h))

    def verify(self, num_steps, z0):
        r = C.c_uint32()
        self.ctx._chk(self.ctx.lib.vimz_ivc_merged_verify(self.h, int(num_steps), _ptr(_zlimbs(z0, self.vk.circuit.len_z)), C.byref(r)))
        return r.value

    def info(self):
        a = np.zeros(8, dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.vimz_ivc_merged_info(self.h, _ptr(a)))
        keys = ["steps", "segments", "ops", "len_z", "primary_wires", "primary_constraints", "secondary_wires", "secondary_constraints"]
        return {k: int(a[i]) for i, k in enumerate(keys)}

    def state(self):
        """(z_start, z_end, steps)"""
        lz = self.vk.circuit.len_z
        a, b = np.zeros((lz, 4), dtype=np.uint64), np.zeros((lz, 4), dtype=np.uint64)
        n = C.c_uint64()
        self.ctx._chk(self.ctx.lib.vimz_ivc_merged_state(self.h, _ptr(a), _ptr(b), C.byref(n)))
        ints = lambda z: [sum(int(z[i, k]) << (64 * k) for k in range(4)) for i in range(lz)]
        return ints(a), ints(b), n.value

    def profile(self):
        s = (C.c_double * 4)()
        self.ctx._chk(self.ctx.lib.vimz_ivc_merged_profile(self.h, s))
        return {k: s[i] for i, k in enumerate(self.PHASES)}

    def save(self):
        n = self.ctx.lib.vimz_ivc_merged_size(self.h)
        buf = np.zeros(n, dtype=np.uint8)
        self.ctx._chk(self.ctx.lib.vimz_ivc_merged_save(self.h, _ptr(buf), n))
        return buf

    def records(self):
        """The statement part as uint64 words (header, segment records, ops): what an independent verifier replays."""
        return _export(self.ctx.lib.vimz_ivc_merged_records, self.h).view(np.uint64)

    def export(self, side, what):
        return _export(self.ctx.lib.vimz_ivc_merged_export, self.h, side, what).view(np.uint64).reshape(-1, 4)

    def compress(self):
        lib = self.ctx.lib
        n = lib.vimz_ivc_merged_compressed_size(self.h)
        buf = np.zeros(n, dtype=np.uint8)
        sec = (C.c_double * 2)()
        self.ctx._chk(lib.vimz_ivc_merged_compress(self.h, _ptr(buf), n, sec))
        return buf, {"setup_s": sec[0], "prove_s": sec[1]}

    @staticmethod
    def verify_compressed(vk, blob, num_steps, z0):
        lib = vk.ctx.lib
        lib.vimz_ivc_verify_merged_compressed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint32)]
        b = np.ascontiguousarray(blob, dtype=np.uint8)
        r = C.c_uint32()
        vk.ctx._chk(lib.vimz_ivc_verify_merged_compressed(vk.h, _ptr(b), b.size, int(num_steps), _ptr(_zlimbs(z0, vk.circuit.len_z)), C.byref(r)))
        return r.value


class AugCircuit:
    """vimz_augcircuit: one side's verifier circuit over a trivial step circuit — host-only hook for the circuit's own tests."""

    def __init__(self, side):
        lib = L.lib()
        vp = C.c_void_p
        lib.vimz_augcircuit_build.argtypes = [C.c_int, C.POINTER(vp)]
        lib.vimz_augcircuit_free.argtypes = [vp]
        lib.vimz_augcircuit_free.restype = None
        lib.vimz_augcircuit_export.argtypes = [vp, C.c_int, vp, C.c_size_t]
        lib.vimz_augcircuit_export.restype = C.c_int64
        lib.vimz_augcircuit_witness.argtypes = [vp, vp, vp, vp]
        self.lib, self.side = lib, side
        h = vp()
        rc = lib.vimz_augcircuit_build(side, C.byref(h))
        if rc:
            raise L.VimzError(rc, "vimz_augcircuit_build")
        self.h = h
        info = _export(lib.vimz_augcircuit_export, self.h, IX_INFO).view(np.uint64)
        self.n_wires, self.n_constraints = int(info[0]), int(info[1])

    def close(self):
        if self.h:
            self.lib.vimz_augcircuit_free(self.h)
            self.h = None

    def r1cs(self):
        return _r1cs_tables(self.lib.vimz_augcircuit_export, self.h)

    def witness(self, inputs):
        """inputs: 17 integers (digest, i, z0, z, U[7], u[4], T[2]).  Returns (wires[n_wires] ints as (n,4) u64, outputs: list of 11 ints)."""
        a = np.zeros((17, 4), dtype=np.uint64)
        assert len(inputs) == 17
        for i, v in enumerate(inputs):
            for k in range(4):
                a[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
        wires = np.zeros((self.n_wires, 4), dtype=np.uint64)
        out = np.zeros((11, 4), dtype=np.uint64)
        rc = self.lib.vimz_augcircuit_witness(self.h, _ptr(a), _ptr(wires), _ptr(out))
        if rc:
            raise L.VimzError(rc, "vimz_augcircuit_witness")
        return wires, [sum(int(out[i, k]) << (64 * k) for k in range(4)) for i in range(11)]


class _Coo(C.Structure):
    _fields_ = [("row", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p), ("nnz", C.c_size_t)]


def vec_axpy(ctx, x1, r, x2, n=None, form=L.FORM_CANONICAL):
    """x1 <- x1 + r * x2 (vimz_vec_axpy): RelaxedR1CSWitness::fold of a resident vector; r an int (canonical) or 4 limbs in `form`."""
    lib = ctx.lib
    lib.vimz_vec_axpy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    rl = _u64([(int(r) >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]) if isinstance(r, int) else _u64(r)
    ctx._chk(lib.vimz_vec_axpy(ctx.h, x1.h, rl.ctypes.data, form, x2.h, x1.n if n is None else n))


class R1CSShape:
    """vimz_r1cs: a caller-supplied R1CS shape resident on the GPU — the `R1CSShape` of nova-snark 0.23.0 behind the C ABI
    (multiply_vec -> spmv3, commit_T -> commit_T).  Matrices are COO triplets (row, col, value) in any order, like the crate's."""

    def __init__(self, ctx, field, nrows, ncols, A, B, Cm, form=L.FORM_CANONICAL):
        """A, B, Cm: (rows uint32[nnz], cols uint32[nnz], vals (nnz, 4) uint64)."""
        self.ctx, self.field, self.nrows, self.ncols = ctx, field, nrows, ncols
        lib = ctx.lib
        vp = C.c_void_p
        lib.vimz_r1cs_upload.argtypes = [vp, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(_Coo), C.POINTER(_Coo), C.POINTER(_Coo), C.c_int, C.POINTER(vp)]
        lib.vimz_r1cs_free.argtypes = [vp, vp]
        lib.vimz_r1cs_free.restype = None
        lib.vimz_spmv3.argtypes = [vp, vp, vp, vp, vp, vp]
        lib.vimz_commit_T.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int]
        keep, coos = [], []
        for rows, cols, vals in (A, B, Cm):
            r, c = np.ascontiguousarray(rows, dtype=np.uint32), np.ascontiguousarray(cols, dtype=np.uint32)
            v = _u64(vals)
            keep += [r, c, v]
            coos.append(_Coo(r.ctypes.data, c.ctypes.data, v.ctypes.data, r.size))
        h = vp()
        ctx._chk(lib.vimz_r1cs_upload(ctx.h, field, nrows, ncols, C.byref(coos[0]), C.byref(coos[1]), C.byref(coos[2]), form, C.byref(h)))
        self.h = h

    def check_relaxed(self, z, u, E=None, form=L.FORM_CANONICAL):
        """is_sat_relaxed (vimz_r1cs_check_relaxed): (number of unsatisfied rows, first of them or None)."""
        lib = self.ctx.lib
        lib.vimz_r1cs_check_relaxed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        ul = _u64([(int(u) >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]) if isinstance(u, int) else _u64(u)
        bad, first = C.c_uint64(), C.c_uint64()
        self.ctx._chk(lib.vimz_r1cs_check_relaxed(self.ctx.h, self.h, z.h, ul.ctypes.data, form, E.h if E is not None else None, C.byref(bad), C.byref(first)))
        return int(bad.value), (int(first.value) if bad.value else None)

    def free(self):
        if self.h:
            self.ctx.lib.vimz_r1cs_free(self.ctx.h, self.h)
            self.h = None

    def multiply_vec(self, z):
        """R1CSShape::multiply_vec: z a DeviceVec of ncols elements -> (Az, Bz, Cz) as new DeviceVecs."""
        out = [self.ctx.vec_alloc(self.field, self.nrows) for _ in range(3)]
        self.ctx._chk(self.ctx.lib.vimz_spmv3(self.ctx.h, self.h, z.h, out[0].h, out[1].h, out[2].h))
        return out

    def commit_T(self, ck, z1, u1, z2, u2=1):
        """R1CSShape::commit_T: returns (T DeviceVec of nrows elements, comm_T as (8,) uint64 canonical affine)."""
        T = self.ctx.vec_alloc(self.field, self.nrows)
        lim = lambda x: np.array([(int(x) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)], dtype=np.uint64)
        a, b = lim(u1), lim(u2)
        out = np.zeros(8, dtype=np.uint64)
        try:
            self.ctx._chk(self.ctx.lib.vimz_commit_T(self.ctx.h, self.h, ck.h, z1.h, _ptr(a), z2.h, _ptr(b), L.FORM_CANONICAL, T.h, _ptr(out), L.FORM_CANONICAL))
        except Exception:
            T.free()
            raise
        return T, out
