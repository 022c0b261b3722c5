"""The compiled pybind11 modules of meshfem_amd/pybind (mesh, tensors, sparse_matrices, periodic_homogenization: the reference's
extension-module names and signatures) -- checked in an interpreter of their own (tests/pybind_checks.py), because the
module names `mesh`, `tensors`, ... are too generic to put on the path of the whole test process."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(what):
    r = subprocess.run([sys.executable, os.path.join(HERE, "pybind_checks.py"), what], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_compiled_modules_host_side():
    _run("cpu")


@pytest.mark.gpu
def test_compiled_modules_on_the_device():
    _run("gpu")
