"""GPU: run the C++ facade test program (the reference's own three pixel tests + rectangle, C++ spelling)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_cpp_facade_runs_reference_tests(tmp_path):
    exe = tmp_path / "facade_test"
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"),
                           "-o", str(exe), "-L", os.path.join(ROOT, "rustcv_amd"), "-lrustcv_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "rustcv_amd"), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], text=True)
    assert "all passed" in out
