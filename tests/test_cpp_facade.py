"""The C++ host façade (include/spg_graph_wrapper.hpp) compiles against the C ABI and behaves like the
reference's call sites: decimation.h functions + IsometryXd on CPU, the 3-pose SE3 example of
src/test_marginalize_se3.cpp on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sparsifyposegraph_amd")


def build(tmp_path):
    exe = str(tmp_path / "facade_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "facade_demo.cpp"),
                           "-L" + PKG, "-lspg_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_facade_host_side(tmp_path):
    out = subprocess.run([build(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host ok" in out.stdout


@pytest.mark.gpu
def test_facade_marginalize_se3_chain(tmp_path):
    out = subprocess.run([build(tmp_path), "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gpu ok" in out.stdout
