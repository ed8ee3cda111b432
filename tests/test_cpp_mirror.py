"""C++ host mirror (include/ndrustfft.hpp) running the reference's own tests, restated in
tests/cpp/test_ndrustfft.cpp.  gpu: against the real library; CPU: it must build, refuse to plan
without a device, and (host logic check) pass when linked against the emulated build."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "ndrustfft_amd", "csrc"), "-s", "-j4"])
    subprocess.check_call(["make", "-C", CPP, "-s"])
    return os.path.join(CPP, "test_ndrustfft")


def test_builds_and_refuses_without_device():
    exe = _build()
    import ctypes
    from ndrustfft_amd import _lib
    if _lib.default().c.ndfft_device_count() > 0:
        pytest.skip("a GPU is visible")
    r = subprocess.run([exe, "--expect-no-device"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no CPU fallback" in r.stdout


def test_host_logic_against_emulated_library(tmp_path):
    emul = os.path.join(ROOT, "tests", "emul")
    subprocess.check_call(["make", "-C", emul, "-s", "-j4"])
    exe = str(tmp_path / "test_ndrustfft_emul")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(CPP, "test_ndrustfft.cpp"),
                           "-o", exe, "-L" + os.path.join(emul, "_build"), "-lndfft_emul",
                           "-Wl,-rpath," + os.path.join(emul, "_build")])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "test result: ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_reference_tests_in_cpp_on_gpu():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0 and "test result: ok." in r.stdout and "; 0 failed" in r.stdout, r.stdout + r.stderr
