"""The C-ABI shared library loads and exports every symbol include/vimz_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="vimz_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vimz_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    for must in ("vimz_ctx_create", "vimz_msm", "vimz_bases_upload", "vimz_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from vimz_amd import _lib
    if not os.path.exists(_lib.PRODUCT_SO_PATH) or not os.path.exists(_lib.TESTING_SO_PATH):
        import __graft_entry__ as g
        g.build()
    L = ctypes.CDLL(_lib.PRODUCT_SO_PATH)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, f"symbols declared in include/vimz_hip.h but not exported: {missing}"


def test_test_hooks_are_not_in_the_product_library():
    """VERDICT r3 / ADVICE r3: vimz_cf_poke, the self-checks and the forging prover are declared in include/vimz_hip_testing.h and compiled
    only under -DVIMZ_TESTING: libvimz_hip.so exports none of them (and reads no VIMZ_TEST_* environment), libvimz_hip_testing.so exports
    every product symbol and every hook."""
    from vimz_amd import _lib
    hooks = sorted(set(declared_symbols("vimz_hip_testing.h")) - set(declared_symbols()))
    assert {"vimz_cf_poke", "vimz_cf_selfcheck", "vimz_worker_selftest", "vimz_test_forge_public_slot", "vimz_strict_bits_selfcheck"} <= set(hooks)
    P = ctypes.CDLL(_lib.PRODUCT_SO_PATH)
    leaked = [s for s in hooks if hasattr(P, s)]
    assert not leaked, f"test hooks exported by the product library: {leaked}"
    assert b"VIMZ_TEST_" not in open(_lib.PRODUCT_SO_PATH, "rb").read()
    T = ctypes.CDLL(_lib.TESTING_SO_PATH)
    missing = [s for s in declared_symbols() + hooks if not hasattr(T, s)]
    assert not missing, missing


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product fails loudly instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vimz_amd import hip, _lib
    with pytest.raises(_lib.VimzError) as e:
        hip.Context(0)
    assert e.value.code == _lib.ERR_NO_DEVICE


def test_bench_sizes_its_cpu_baseline_by_the_usable_cores():
    """bench.py's CPU baseline uses as many threads as the process may really run (affinity mask capped by the cgroup quota): at least
    one, never more than the machine shows."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_a_truncated_compressed_proof_is_an_invalid_argument():
    """a blob shorter than its 8-byte header raises VimzError like every other malformed input (not numpy's ValueError)."""
    from vimz_amd import _lib, folding
    for blob in (b"", b"\x01", b"\x56\x5a\x43\x4d\x47\x31\x00"):
        with pytest.raises(_lib.VimzError) as e:
            folding.verify_compressed_proof(None, blob, 1, [0])
        assert e.value.code == _lib.ERR_INVALID
