"""The header-only C++ facade (include/MeshFEMHip/LinearElasticity.hh) compiles with plain g++
against the C ABI and behaves like the reference's Simulator: std::runtime_error without a device
(no fallback), a correct cantilever on the GPU."""
import os
import subprocess

import pytest

import meshfem_amd as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "facade_cantilever")


EXE2 = os.path.join(ROOT, "tests", "cpp", "facade_scalar_and_constraints")


def _build(name="facade_cantilever"):
    src = os.path.join(ROOT, "tests", "cpp", name + ".cc")
    libdir = os.path.dirname(M.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o",
                           os.path.join(ROOT, "tests", "cpp", name),
                           "-L", libdir, "-lmeshfem_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])


def test_facade_compiles_and_throws_without_device():
    _build()
    r = subprocess.run([EXE, "-1"], capture_output=True, text=True)
    assert r.returncode == 3 and "runtime_error" in r.stdout and "no CPU fallback" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_facade_cantilever_on_gpu():
    _build()
    r = subprocess.run([EXE, "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "tip deflection" in r.stdout


def test_facade_scalar_and_constraints_compiles():
    _build("facade_scalar_and_constraints")
    r = subprocess.run([EXE2, "-1"], capture_output=True, text=True)
    assert r.returncode == 3 and "runtime_error" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_facade_scalar_and_constraints_on_gpu():
    _build("facade_scalar_and_constraints")
    r = subprocess.run([EXE2, "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "poisson" in r.stdout and "free body" in r.stdout and "generic SPSDSystem" in r.stdout


EXE3 = os.path.join(ROOT, "tests", "cpp", "facade_homogenization")


def test_facade_homogenization_compiles():
    _build("facade_homogenization")
    r = subprocess.run([EXE3, "-1"], capture_output=True, text=True)
    assert r.returncode == 3 and "runtime_error" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_facade_homogenization_on_gpu():
    _build("facade_homogenization")
    r = subprocess.run([EXE3, "0"], capture_output=True, text=True)
    assert r.returncode == 0 and "homogenization ok" in r.stdout, r.stdout + r.stderr


EXE_C = os.path.join(ROOT, "tests", "cpp", "abi_plain_c")


def _build_c():
    libdir = os.path.dirname(M.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "abi_plain_c.c"), "-o", EXE_C, "-L", libdir, "-lmeshfem_hip", "-lm",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])


def test_plain_c_client_compiles_and_fails_loudly_without_device():
    _build_c()
    r = subprocess.run([EXE_C, "-1"], capture_output=True, text=True)
    assert r.returncode == 3 and "2 elements, 5 nodes" in r.stdout and "no CPU fallback" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_plain_c_client_on_gpu():
    _build_c()
    r = subprocess.run([EXE_C, "0"], capture_output=True, text=True)
    assert r.returncode == 0 and "plain C client ok" in r.stdout, r.stdout + r.stderr


EXE4 = os.path.join(ROOT, "tests", "cpp", "facade_distributed")


def test_facade_distributed_compiles():
    _build("facade_distributed")
    r = subprocess.run([EXE4, "-1"], capture_output=True, text=True)
    assert r.returncode == 3 and "runtime_error" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_facade_distributed_on_gpu():
    """No Python / torch in this process: the library finds RCCL in the ROCm installation by itself."""
    _build("facade_distributed")
    r = subprocess.run([EXE4, "0"], capture_output=True, text=True)
    assert r.returncode == 0 and "distributed facade ok" in r.stdout and "RCCL" in r.stdout, r.stdout + r.stderr


EXE5 = os.path.join(ROOT, "tests", "cpp", "abi_solver_only_swap")


def test_solver_only_swap_compiles_and_fails_loudly_without_device():
    """INTEGRATION.md section 1b as a program: a host simulator with its own node numbering, per-element tensors, DoF map and Dirichlet
    variables hands them to the C ABI (mfh_mesh_set ... mfh_solve)."""
    _build("abi_solver_only_swap")
    r = subprocess.run([EXE5, "-1"], capture_output=True, text=True)
    assert r.returncode == 3 and "18 elements, 63 nodes" in r.stdout and "no CPU fallback" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_solver_only_swap_on_gpu():
    _build("abi_solver_only_swap")
    r = subprocess.run([EXE5, "0"], capture_output=True, text=True)
    assert r.returncode == 0 and "solver-only swap ok" in r.stdout, r.stdout + r.stderr
