"""The page-lock registry (lpopc_amd/csrc/rpm_pin.cpp -> librpm_pin.so) on the CPU: its bookkeeping against a mock of the
runtime's hipHostRegister / hipHostUnregister (tests/native/pin_registry_test.cpp), and the library as built — it loads, it
exports what rpm_pin.h declares, and without a device a request is refused, counted and explained, never silent."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lpopc_amd", "csrc")


import pytest


@pytest.mark.parametrize("sanitize", [[], ["-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"]], ids=["plain", "asan_ubsan"])
def test_registry_bookkeeping_against_a_mock_runtime(tmp_path, sanitize):
    """(the second build runs the same 20 000 random acquisitions / releases under AddressSanitizer and UBSan: CPU only, the pool's
    GPU boxes do not run sanitizers)"""
    exe = str(tmp_path / "pin_registry_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + sanitize +
                          [os.path.join(ROOT, "tests", "native", "pin_registry_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr


def test_pin_library_loads_and_refuses_loudly_without_a_device(built):
    import torch  # noqa: F401  (one HIP runtime per process: the copy torch maps, see lpopc_amd/engine.py)
    L = C.CDLL(os.path.join(CSRC, "librpm_pin.so"))
    for name in ("rpm_pin_acquire", "rpm_pin_release_owner", "rpm_pin_counter", "rpm_pin_held", "rpm_pin_last_error"):
        assert hasattr(L, name), name
    # the product library resolves the same instance (DT_NEEDED, found next to it)
    out = subprocess.run(["readelf", "-d", os.path.join(CSRC, "librpm_hip.so")], capture_output=True, text=True).stdout
    assert "librpm_pin.so" in out and "$ORIGIN" in out
    if torch.cuda.is_available():
        return
    L.rpm_pin_acquire.restype = C.c_void_p
    L.rpm_pin_acquire.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    L.rpm_pin_counter.restype = C.c_long
    buf = (C.c_double * 32768)()
    owner = C.c_int()
    before = L.rpm_pin_counter(1)
    assert L.rpm_pin_acquire(C.addressof(owner), C.addressof(buf), C.sizeof(buf), 8, 0) is None
    assert L.rpm_pin_counter(1) == before + 1 and L.rpm_pin_counter(100) == 0
    msg = C.create_string_buffer(256)
    L.rpm_pin_last_error(msg, 256)
    assert b"hipHostRegister" in msg.value
