"""Step circuits: R1CS shape + witness program built by the native builder (include/vimz_hip.h,
`vimz_circuit_*`).  Plays the role of `load_r1cs(config.circuit_file())` in
vimz/src/nova_snark_backend/folding.rs:22.  Host-only — works without a GPU."""
import ctypes as C

import numpy as np

from . import _lib as L

TRANSFORMATIONS = ["blur", "brightness", "contrast", "crop", "grayscale", "hash", "redact", "resize", "sharpness"]
T_ID = {n: i for i, n in enumerate(TRANSFORMATIONS)}

CX = dict(A_ROWPTR=0, A_COL=1, A_COEF=2, B_ROWPTR=3, B_COL=4, B_COEF=5, C_ROWPTR=6, C_COL=7, C_COEF=8, DICT_MONT=9,
          DICT_CANON=10, DECOMP=11, LANE_GROUPS=12, LANE_INSTR=13, LANE_ROWS=14, JOBS=15, CHAINS=16, FOPS=17, ZOUT=18, LC_TERMS=19)

# packed elements per row at each resolution (vimz/src/transformation.rs:93-101 rows; width = pixels / 10)
RESOLUTION_WIDTH = {"SD": 64, "HD": 128, "FHD": 192, "4K": 384, "8K": 768}


def default_shape(transformation, resolution="HD"):
    """(width, width2, rows_in, rows_out, crop_height) as instantiated by circuits/nova_snark/*.circom at HD,
    scaled in width for the other resolutions (ratio_to_lower: vimz/src/transformation.rs:115-123)."""
    w = RESOLUTION_WIDTH[resolution]
    if transformation == "redact":
        return (160, 0, 0, 0, 0)
    if transformation == "resize":
        lower = {"HD": "SD", "FHD": "HD", "4K": "FHD", "8K": "4K"}[resolution]
        ri, ro = (3, 2) if resolution in ("HD", "FHD") else (2, 1)
        return (w, RESOLUTION_WIDTH[lower], ri, ro, 0)
    if transformation == "crop":
        return (w, 64, 0, 0, 480)
    return (w, 0, 0, 0, 0)


class Circuit:
    def __init__(self, transformation, width=128, width2=64, rows_in=3, rows_out=2, crop_height=480):
        self.lib = L.lib()
        self.lib.vimz_circuit_last_error.restype = C.c_char_p
        self.lib.vimz_circuit_export.restype = C.c_int64
        self.lib.vimz_circuit_export.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        self.lib.vimz_circuit_free.argtypes = [C.c_void_p]
        self.lib.vimz_circuit_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        self.transformation = transformation
        self.t = T_ID[transformation]
        self.shape = (width, width2, rows_in, rows_out, crop_height)
        h = C.c_void_p()
        rc = self.lib.vimz_circuit_build(self.t, width, width2, rows_in, rows_out, crop_height, C.byref(h))
        if rc != L.OK:
            raise L.VimzError(rc, self.lib.vimz_circuit_last_error().decode())
        self.h = h
        info = (C.c_uint64 * 16)()
        self.lib.vimz_circuit_info(self.h, info)
        (self.n_wires, self.n_constraints, self.n_linear, self.len_z, self.n_priv, self.nnz_a, self.nnz_b, self.nnz_c,
         self.n_dict, self.n_decomp, self.n_lane_groups, self.n_lane_instr, self.n_lane_rows, self.n_jobs, self.n_chains,
         self.n_fops) = [int(x) for x in info]

    def prepare_ivc(self):
        """vimz_circuit_prepare_ivc: synthesise the augmented circuits of a Nova IVC over this circuit now (host only; vimz_ivc_create otherwise does it on first use)."""
        self.lib.vimz_circuit_prepare_ivc.argtypes = [C.c_void_p]
        rc = self.lib.vimz_circuit_prepare_ivc(self.h)
        if rc != L.OK:
            raise L.VimzError(rc, "vimz_circuit_prepare_ivc")

    @classmethod
    def for_resolution(cls, transformation, resolution="HD"):
        return cls(transformation, *default_shape(transformation, resolution))

    @classmethod
    def from_r1cs(cls, data):
        """Load a circom-built iden3 `.r1cs` (bytes) — the analogue of load_r1cs (folding.rs:22).  The circuit has no
        witness program: fold it with Prover.fold_witness."""
        self = cls.__new__(cls)
        self.lib = L.lib()
        self.lib.vimz_circuit_last_error.restype = C.c_char_p
        self.lib.vimz_circuit_export.restype = C.c_int64
        self.lib.vimz_circuit_export.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        self.lib.vimz_circuit_free.argtypes = [C.c_void_p]
        self.lib.vimz_circuit_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        self.lib.vimz_circuit_load_r1cs.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        self.transformation, self.t, self.shape = None, -1, (0, 0, 0, 0, 0)
        h = C.c_void_p()
        rc = self.lib.vimz_circuit_load_r1cs(bytes(data), len(data), C.byref(h))
        if rc != L.OK:
            raise L.VimzError(rc, self.lib.vimz_circuit_last_error().decode())
        self.h = h
        info = (C.c_uint64 * 16)()
        self.lib.vimz_circuit_info(self.h, info)
        (self.n_wires, self.n_constraints, self.n_linear, self.len_z, self.n_priv, self.nnz_a, self.nnz_b, self.nnz_c,
         self.n_dict, self.n_decomp, self.n_lane_groups, self.n_lane_instr, self.n_lane_rows, self.n_jobs, self.n_chains,
         self.n_fops) = [int(x) for x in info]
        return self

    def export(self, what, dtype=np.uint8):
        n = self.lib.vimz_circuit_export(self.h, CX[what], None, 0)
        if n < 0:
            raise L.VimzError(int(n), "vimz_circuit_export")
        buf = np.zeros(max(int(n), 1), dtype=np.uint8)
        self.lib.vimz_circuit_export(self.h, CX[what], buf.ctypes.data_as(C.c_void_p), int(n))
        return buf[:int(n)].view(dtype)

    def csr(self, m):
        """(row_ptr, col, coef) uint32 arrays of matrix m in 'A','B','C'."""
        return (self.export(f"{m}_ROWPTR", np.uint32), self.export(f"{m}_COL", np.uint32), self.export(f"{m}_COEF", np.uint32))

    def close(self):
        if self.h:
            self.lib.vimz_circuit_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def wtns_load(data):
    """Parse an iden3 `.wtns` (bytes) -> (n, 4) uint64 canonical witness."""
    lib = L.lib()
    lib.vimz_wtns_load.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    n = C.c_size_t()
    rc = lib.vimz_wtns_load(bytes(data), len(data), None, 0, C.byref(n))
    if rc != L.OK:
        raise L.VimzError(rc, "vimz_wtns_load")
    out = np.zeros((n.value, 4), dtype=np.uint64)
    lib.vimz_wtns_load(bytes(data), len(data), out.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
    return out
