"""The C++ host façade (include/spg_graph_wrapper.hpp, include/spg_evaluate.hpp) compiles against the C ABI and
behaves like the reference's call sites: decimation.h functions, IsometryXd, the job-line grammar on the CPU; the
3-pose SE3 example of src/test_marginalize_se3.cpp, the full GraphWrapper virtual set and evaluate()
(src/evaluate.cpp:32-221) on the GPU; the multi-rank driver from C++ (tests/cpp/ranks_demo.cpp)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sparsifyposegraph_amd")


def build(tmp_path):
    exe = str(tmp_path / "facade_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "facade_demo.cpp"),
                           "-L" + PKG, "-lspg_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def build_ranks(tmp_path):
    exe = str(tmp_path / "ranks_demo")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "ranks_demo.cpp"), "-L" + PKG, "-lspg_hip", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_facade_host_side(tmp_path):
    out = subprocess.run([build(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host ok" in out.stdout


def test_ranks_demo_compiles(tmp_path):
    out = subprocess.run([build_ranks(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 2 and "usage" in out.stderr


@pytest.mark.gpu
def test_facade_full_interface(tmp_path):
    out = subprocess.run([build(tmp_path), "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gpu ok" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("job", ["se2 X online tree global 2 10", "glc X cluster tree global 2 10 kld 10", "se2 X cluster tree global 2 10 chi2 10"])
def test_cpp_evaluate_matches_python_harness(tmp_path, job, hip_ctx):
    """evaluate() in C++ (spg_evaluate.hpp) and the Python harness (evaluate.py) drive the same library through the
    same call sequence: the .kld series must agree, and the result files must be where the reference puts them."""
    from sparsifyposegraph_amd import g2o_io
    from sparsifyposegraph_amd.evaluate import EvaluateInfo, evaluate
    from sparsifyposegraph_amd.graph import (DecimateOptions, GraphWrapperHIP, SparsityOptions, clusterDecimate, onlineDecimate)
    from tests import util
    g, which, opts, *_ = util.load_golden("manhattan_nfr_tree")
    sub, _ = util.prefix_graph(g, which, 60)
    path = str(tmp_path / "m60.g2o")
    g2o_io.write_g2o(path, sub)
    words = job.split()
    out = subprocess.run([build(tmp_path), "evaluate", path, str(tmp_path / "res"), job], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got = [(int(l.split()[1]), float(l.split()[2])) for l in out.stdout.splitlines() if l.startswith("kld ")]
    stem = [l.split()[1] for l in out.stdout.splitlines() if l.startswith("stem ")][0]
    alg = {"se2": "se2", "glc": "glc"}[words[0]]
    assert stem == str(tmp_path / "res" / words[2] / "2" / "m60" / f"{alg}_tree_g")
    lines = open(stem + ".kld").read().split("\n")
    assert [int(l.split()[0]) for l in lines if l] == [i for i, _ in got]
    txt = open(stem + ".txt").read()
    assert txt.startswith(alg.upper() + " Tree\n    baseline:     nodes = 59; edges = ") and "fillin" in txt and "    last " in txt
    # the same job through the Python harness: the source graph optimised at load, as the C++ program does
    full = GraphWrapperHIP.load(path, ctx=hip_ctx)
    full.optimize()
    ids, poses = full.vertices()
    sub2 = dict(sub, poses=poses)
    dec = {"online": (onlineDecimate, DecimateOptions(2, 100)), "cluster": (clusterDecimate, DecimateOptions(2, 10))}[words[2]]
    info = EvaluateInfo(dec[0], dec[1], SparsityOptions(SparsityOptions.Tree, linPoint=SparsityOptions.Global),
                        "glc" if words[0] == "glc" else "nfr", kldPeriod=10, useChi2=("chi2" in words))
    ref, *_ = evaluate(sub2, info, lambda glc: GraphWrapperHIP(ctx=hip_ctx, pose_dim=3, useGLC=glc), full)
    assert [i for i, _ in got] == [i for i, _ in ref]
    for (i, a), (_, b) in zip(got, ref):
        assert a == pytest.approx(b, rel=1e-7, abs=1e-8), (i, a, b)


def _write_small_sphere(tmp_path):
    from sparsifyposegraph_amd import g2o_io
    path = str(tmp_path / "s600.g2o")
    g2o_io.write_g2o(path, g2o_io.synth_sphere(n_poses=600, ring=30))
    return path


@pytest.mark.gpu
def test_cpp_ranks_rccl_world_size_1(tmp_path):
    """The built-in exchange (ncclAllGather bound from librccl) from a C++ caller. One GPU on the test box: RCCL
    refuses two ranks on one device, so the collective itself is exercised at world size 1."""
    exe, path = build_ranks(tmp_path), _write_small_sphere(tmp_path)
    out = subprocess.run([exe, "rccl", "0", "1", path, str(tmp_path / "id")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ncclAllGather" in out.stdout and "rank 0 ok" in out.stdout


@pytest.mark.gpu
def test_cpp_ranks_world_size_2_sharing_one_gpu(tmp_path):
    """spg_ctx_create_ranks + spg_graph_marginalize_ranks from two C++ processes sharing the test box's one MI355X,
    every batch sharded; the exchange is the caller's (shared-memory staging) because RCCL needs one GPU per rank."""
    exe, path = build_ranks(tmp_path), _write_small_sphere(tmp_path)
    rdv = str(tmp_path / "shm")
    procs = [subprocess.Popen([exe, "shm", str(r), "2", path, rdv], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o and " 0 exchanged" not in o
