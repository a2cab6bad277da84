"""Runs the C++ host-mirror test (tests/cpp/test_host_mirror.cpp: plonky2-named C++ classes over the C-ABI, compared with
the oracle) on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_host_mirror_against_oracle():
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])  # (incremental: a binary older than its sources is rebuilt)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


def test_cpp_host_mirror_builds():
    """the header-only mirror compiles against include/glprover.h with a plain C++17 compiler (no HIP needed)"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "test_host_mirror"))
