"""Builds and runs tests/cpp/test_friendly.cpp: the C++ mirror of the reference's friendly-layer tests over
include/m4ri_friendly.hpp and libm4ri_hip.so (g++ only: the header needs no HIP toolchain)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "m4ri-rust_amd", "lib")


@pytest.fixture(scope="module")
def exe(built, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "test_friendly")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_friendly.cpp"), "-o", out,
                           "-L", LIBDIR, "-lm4ri_hip", "-Wl,-rpath," + LIBDIR])
    return out


def test_cpp_host_part(exe):
    assert subprocess.run([exe, "host"], capture_output=True, text=True).returncode == 0


@pytest.mark.gpu
def test_cpp_mul_part(exe):
    r = subprocess.run([exe, "mul"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_cpp_mul_panics_without_gpu(exe):
    import m4ri_rust_amd  # noqa: F401
    from m4ri_rust_amd import device
    if device.device_count() > 0:
        pytest.skip("GPU present")
    r = subprocess.run([exe, "mul"], capture_output=True, text=True)
    assert r.returncode != 0 and "Multiplication failed" in (r.stderr + r.stdout)
