"""The GPU-free host code of the library under ThreadSanitizer and AddressSanitizer + UBSan (SURVEY.md §5 "Race detection /
sanitizers"): `make -C vimz_amd/csrc sanitize` instruments the HOST passes of the translation units that hold the verifier circuits' witness
generators with their helper threads, the Nova + CycleFold recursion, the merge transcript replay and the circuit builder / loaders, and
tests/native/host_sanitize.cpp drives them — two concurrent witness generators with their own helper threads, 20 000 wake-ups of a sleeping
helper, the recursion with helpers on, malformed .r1cs / .wtns bytes.  No GPU is touched."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vimz_amd", "csrc")


@pytest.fixture(scope="module")
def built():
    if not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    r = subprocess.run(["make", "-s", "-j", "4", "-C", CSRC, "sanitize"], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return os.path.join(CSRC, "build")


@pytest.mark.parametrize("which", ["tsan", "asan"])
def test_host_code_under_sanitizers(built, which):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=0 abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1 halt_on_error=1")
    r = subprocess.run([os.path.join(built, f"host_{which}")], env=env, capture_output=True, text=True, timeout=900)
    tail = r.stdout[-1500:] + r.stderr[-4000:]
    assert r.returncode == 0 and "host_sanitize ok" in r.stdout, tail
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
