"""Loader for the gfx950 extension (vimz_amd/libvimz_hip.so), the C ABI declared in include/vimz_hip.h.

There is NO CPU fallback: if the shared library is missing this raises, and every entry point returns
an error (surfaced as VimzError) when no MI355X is visible.
"""
import ctypes as C
import os

# Hardware queues of the HIP runtime.  Its default of 4 makes the streams of several concurrent provers of ONE process wait for each
# other (three segments: 785 steps/s against 885 with 8; more than 8 changes nothing) — but the GPU has about eight in all: two
# processes with 8 each on one GPU collapse to 63 steps/s, four with 2 each to 67, while 4 each is fine for two and for four
# processes (profiles/r03_hw_queues.txt).  So: 8 when this process has its GPU to itself, the runtime's 4 when ranks share one, as
# far as a launcher's environment tells (LOCAL_WORLD_SIZE / visible devices); callers that know
# better set GPU_MAX_HW_QUEUES themselves (an explicit setting wins).  The runtime reads the variable when it initialises: this only
# takes effect if the module is imported before anything touches the GPU.
def default_hw_queues(n_devices=None):
    local = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
    if n_devices is None:
        vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", ""))
        n_devices = len([x for x in vis.split(",") if x.strip()]) if vis else None
    if n_devices is None:
        try:
            n_devices = len([d for d in os.listdir("/sys/class/kfd/kfd/topology/nodes")
                             if int(open(f"/sys/class/kfd/kfd/topology/nodes/{d}/simd_count").read() or 0) > 0])
        except (OSError, ValueError):
            n_devices = 1
    per_gpu = -(-local // max(1, n_devices))
    return 8 if per_gpu <= 1 else 4


HW_QUEUES_DEFAULTED = "GPU_MAX_HW_QUEUES" not in os.environ      # (a launcher that starts ranks drops a defaulted value so that each rank decides for itself)
os.environ.setdefault("GPU_MAX_HW_QUEUES", str(default_hw_queues()))

HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_SO_PATH = os.path.join(HERE, "libvimz_hip.so")
# libvimz_hip_testing.so: the same library with the test hooks of include/vimz_hip_testing.h (built from the same objects; the product library
# has none of them).  A test that needs a hook on a prover object runs in a process started with VIMZ_HIP_LIBRARY=testing; the host-only
# hooks are reached through testing_lib() beside the product library.
TESTING_SO_PATH = os.path.join(HERE, "libvimz_hip_testing.so")
SO_PATH = TESTING_SO_PATH if os.environ.get("VIMZ_HIP_LIBRARY") == "testing" else PRODUCT_SO_PATH

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
_testing = None


def testing_lib():
    """libvimz_hip_testing.so, for its GPU-free hooks (self-checks of the circuits, helper-thread test): test infrastructure."""
    global _testing
    if _testing is None:
        if not os.path.exists(TESTING_SO_PATH):
            raise ImportError(f"{TESTING_SO_PATH} is missing: run `make -C vimz_amd/csrc` (or __graft_entry__.build())")
        _testing = lib() if SO_PATH == TESTING_SO_PATH else C.CDLL(TESTING_SO_PATH)
    return _testing


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
