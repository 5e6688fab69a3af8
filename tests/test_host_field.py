"""Host-side field arithmetic of the library (the verifier circuits' witnesses are computed on the host with it)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lazy_dot_product_equals_sum_of_products(tmp_path):
    exe = tmp_path / "fp_dot_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "vimz_amd", "csrc"), "-o", str(exe),
                           os.path.join(ROOT, "tests", "native", "fp_dot_check.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    for f in ("BnFr", "BnFq", "PallasFp", "VestaFq"):
        assert f"{f}: 0 mismatches" in out.stdout
