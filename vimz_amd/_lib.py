"""Loader for the gfx950 extension (vimz_amd/libvimz_hip.so), the C ABI declared in include/vimz_hip.h.

There is NO CPU fallback: if the shared library is missing this raises, and every entry point returns
an error (surfaced as VimzError) when no MI355X is visible.
"""
import ctypes as C
import os

# Hardware queues of the HIP runtime: the default of 4 makes the streams of several concurrent provers of one process wait for each
# other (three segments: 785 steps/s against 885 with 8).  Read by the runtime when it initialises, so it only takes
# effect if this module is imported before anything touches the GPU; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libvimz_hip.so")

OK = 0
ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_UNSAT = -1, -2, -3, -4
FORM_CANONICAL, FORM_MONTGOMERY = 0, 1
CURVE_BN254_G1, CURVE_GRUMPKIN, CURVE_PALLAS, CURVE_VESTA = 0, 1, 2, 3
FIELD_BN254_FR, FIELD_BN254_FQ, FIELD_PALLAS_FP, FIELD_VESTA_FQ = 0, 1, 2, 3
CURVE_SCALAR_FIELD = {0: 0, 1: 1, 2: 3, 3: 2}
CURVE_BASE_FIELD = {0: 1, 1: 0, 2: 2, 3: 3}
# the four field moduli (BN254 Fr, BN254 Fq, Pallas Fp, Vesta Fq), indexed by FIELD_*
MODULUS = {
    0: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    1: 21888242871839275222246405745257275088696311157297823662689037894645226208583,
    2: 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
    3: 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,
}


class VimzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vimz_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libvimz_hip.so (built by `make -C vimz_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: the HIP extension has not been built (run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C vimz_amd/csrc`). There is no CPU fallback.")
        L = C.CDLL(SO_PATH)
        vp, sz, i, u64p = C.c_void_p, C.c_size_t, C.c_int, C.c_void_p
        L.vimz_version.restype = C.c_char_p
        L.vimz_last_error.restype = C.c_char_p
        L.vimz_last_error.argtypes = [vp]
        L.vimz_ctx_create.argtypes = [i, C.POINTER(vp)]
        L.vimz_ctx_destroy.argtypes = [vp]
        L.vimz_ctx_destroy.restype = None
        L.vimz_device_info.argtypes = [vp, C.c_char_p, sz, C.POINTER(i), C.POINTER(C.c_uint64)]
        L.vimz_sync.argtypes = [vp]
        L.vimz_timer_start.argtypes = [vp]
        L.vimz_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
        L.vimz_set_profiling.argtypes = [vp, i]
        L.vimz_msm_last_profile.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.vimz_bases_upload.argtypes = [vp, i, u64p, sz, i, C.POINTER(vp)]
        L.vimz_bases_generate.argtypes = [vp, i, C.c_char_p, sz, sz, C.POINTER(vp)]
        L.vimz_bases_download.argtypes = [vp, vp, sz, u64p, sz, i]
        L.vimz_bases_precompute.argtypes = [vp, vp, i]
        L.vimz_bases_len.argtypes = [vp]
        L.vimz_bases_len.restype = sz
        L.vimz_bases_free.argtypes = [vp, vp]
        L.vimz_bases_free.restype = None
        L.vimz_vec_alloc.argtypes = [vp, i, sz, C.POINTER(vp)]
        L.vimz_vec_upload.argtypes = [vp, vp, sz, u64p, sz, i]
        L.vimz_vec_download.argtypes = [vp, vp, sz, u64p, sz, i]
        L.vimz_vec_len.argtypes = [vp]
        L.vimz_vec_len.restype = sz
        L.vimz_vec_free.argtypes = [vp, vp]
        L.vimz_vec_free.restype = None
        L.vimz_msm.argtypes = [vp, vp, u64p, sz, i, i, u64p, i]
        L.vimz_msm_vec.argtypes = [vp, vp, sz, vp, sz, sz, i, u64p, i]
        L.vimz_msm_vec_ex.argtypes = [vp, vp, sz, vp, sz, sz, i, i, u64p, i]
        L.vimz_field_op.argtypes = [vp, i, i, u64p, u64p, u64p, sz]
        L.vimz_curve_add.argtypes = [vp, i, u64p, u64p, u64p, sz]
        _lib = L
    return _lib
