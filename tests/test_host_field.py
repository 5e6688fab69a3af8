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


def test_lazily_reduced_coordinate_field_and_curve_formulas(tmp_path):
    """vimz_amd/csrc/fp29.hpp + ec.hpp compiled for the host: 9x29-bit lazy arithmetic == the canonical 8x32 arithmetic on every
    representative the formulas admit, and 12 000 random curve operations per curve (mixed / full additions, doublings, the
    doubling and cancellation branches) stay inside the documented bounds and equal the canonical computation."""
    exe = tmp_path / "fp29_lazy_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "vimz_amd", "csrc"), "-o", str(exe),
                           os.path.join(ROOT, "tests", "native", "fp29_lazy_check.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    for f in ("BnFr", "BnFq", "PallasFp", "VestaFq"):
        assert f"{f}: field 0 mismatches" in out.stdout
    for c in ("BnG1", "Grumpkin", "Pallas", "Vesta"):
        assert f"{c}: curve 0 mismatches" in out.stdout
