"""The header-only C++ façade (reference class names) and the TNLP adaptor compile with g++ and drive
the C ABI; without a GPU the value passes return false with the 'no CPU fallback' message."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_facade_and_tnlp_adaptor(built, tmp_path):
    exe = str(tmp_path / "facade")
    csrc = os.path.join(ROOT, "lpopc_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp_facade_test.cpp"), "-o", exe,
                           "-L", csrc, "-lrpm_hip", "-Wl,-rpath," + csrc])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "n=85 m=66 nnz_jac=1602" in r.stdout                 # SURVEY App. C, Bryson-Denham 1x20
    assert "eval_g -> true" in r.stdout or "no CPU fallback" in r.stdout
    assert "group eval_g -> true" in r.stdout or ("group eval_g -> false" in r.stdout and "rank 0" in r.stdout)
