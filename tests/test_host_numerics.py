"""CPU: the product's host/device numeric headers (glibc atan2f restatement, x87 80-bit emulation)
compiled for the host and compared with this machine's libm / native long double."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hn(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hn") / "host_numerics.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests/support/host_numerics.cpp")])
    L = C.CDLL(so)
    L.check_atanf.restype = C.c_long; L.check_atanf.argtypes = [C.c_long, C.POINTER(C.c_long)]
    L.check_atan2f.restype = C.c_long; L.check_atan2f.argtypes = [C.c_long, C.c_uint64, C.POINTER(C.c_uint32)]
    L.check_x87.restype = C.c_long; L.check_x87.argtypes = [C.c_long, C.c_uint64, C.POINTER(C.c_double)]
    return L


def test_atanf_matches_host_libm(hn):
    fb = C.c_long(0)
    assert hn.check_atanf(41, C.byref(fb)) == 0, hex(fb.value)       # every 41st of all 2^32 floats


def test_atan2f_matches_host_libm(hn):
    fb = (C.c_uint32 * 2)()
    assert hn.check_atan2f(20_000_000, 2024, fb) == 0, (hex(fb[0]), hex(fb[1]))


def test_x87_emulation_matches_long_double(hn):
    fb = (C.c_double * 2)()
    assert hn.check_x87(10_000_000, 99, fb) == 0, (fb[0], fb[1])
