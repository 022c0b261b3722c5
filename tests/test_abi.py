"""CPU tests of the drop-in boundary: the shared object loads, exports every symbol declared in
include/meshfem_hip.h, and fails loudly (never silently falls back) when there is no device."""
import ctypes
import os
import re

import pytest

import meshfem_amd as M
from meshfem_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(headers=("meshfem_hip.h", "meshfem_hip_extras.h")):
    txt = ""
    for h in headers:
        with open(os.path.join(ROOT, "include", h)) as f:
            txt += f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    # functions only: `mfh_status (*name)(...)` is the return type of a callback typedef
    return sorted(set(re.findall(r"\b(mfh_[a-z_A-Z0-9]+)\s*\(", txt)) - {"mfh_status"})


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(M.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 50
    for s in syms:
        assert hasattr(lib, s), "libmeshfem_hip.so does not export %s" % s
    # the Python binding declares exactly the headers' symbols
    assert sorted(_lib.PROTOTYPES) == syms
    # the drop-in boundary itself (meshfem_hip.h) carries no kernel timers, test hooks or device-pointer building blocks:
    # those live in meshfem_hip_extras.h
    boundary = _declared_symbols(("meshfem_hip.h",))
    assert not [s for s in boundary if s.startswith(("mfh_time_", "mfh_debug_", "mfh_dev_tl", "mfh_dev_pcg", "mfh_dev_spmv", "mfh_dev_precond",
                                                     "mfh_dev_dots", "mfh_dev_mask", "mfh_dev_set", "mfh_tl_partitioned"))]
    assert "mfh_solve" in boundary and "mfh_assemble" in boundary and "mfh_dist_solve" in boundary


def test_no_cuda_shims_or_fallback_paths_in_product():
    bad = re.compile(r"__HIP_PLATFORM_AMD__|cuda_runtime|hipify|import\s+oracle|from\s+oracle", re.I)
    for d, _, files in os.walk(os.path.join(ROOT, "meshfem_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".hh", ".h")):
                with open(os.path.join(d, fn)) as f:
                    assert not bad.search(f.read()), fn


def test_no_device_fails_loudly(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    with pytest.raises(M.MeshFEMHipError):
        M.Context(0)


def test_host_only_context_refuses_numeric_work():
    import numpy as np
    from meshfem_amd import grid
    V, T = grid.grid_tet_mesh(1, 1, 1)
    c = M.Context(-1)
    c.mesh_build(T, V, 1)
    for call in (lambda: c.assemble(), lambda: c.element_stiffness(), lambda: c.solve(np.zeros(3 * c.n_dof)),
                 lambda: c.apply_K(np.zeros(3 * c.n_dof)), lambda: c.elem_volumes()):
        with pytest.raises(M.MeshFEMHipError) as ei:
            call()
        assert ei.value.code == _lib.ERR_HIP
    assert b"gfx950" in _lib.load().mfh_version()


def test_public_header_is_plain_c99(tmp_path):
    """The drop-in boundary is a C ABI: include/meshfem_hip.h must compile as C99 (no C++-isms, no torch / HIP types)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "c_abi_check.c"
    src.write_text('#include "meshfem_hip.h"\n#include "meshfem_hip_extras.h"\nint main(void) { mfh_ctx* c = 0; (void)c; return (int)MFH_OK; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_default_library_has_no_experiment_variants():
    """The shipped library contains the product kernels only: no ablation / racy variant of the assembly kernel can be
    selected (the timing experiments of round 1 are documented in profiles/r01_assembly_optimisation_log.md, not compiled
    in), and no kernel symbol carries an extra experiment template argument."""
    c = M.Context(-1)
    with pytest.raises(M.MeshFEMHipError, match="unknown option"):
        c.set_option("debug_variant", 1)
    c.close()
    import subprocess
    out = subprocess.run(["nm", "-C", M.LIB_PATH], capture_output=True, text=True).stdout
    names = set(re.findall(r"k_assemble_gather<[^>]*>", out))
    assert names, "assembly kernel stubs not found"
    for n in names:
        # <DIM, DEG, MAT, UPPER, DET>: UPPER only names the launches on the upper-triangle storage (option matrix_storage), same code; DET is
        # the product option "deterministic" (waves add in list order) as an instantiation of its own, so that the default kernels carry
        # no barrier inside the contribution loop
        assert re.fullmatch(r"k_assemble_gather<\d, \d, \d, (true|false), (true|false)>", n), "unexpected template arguments: " + n
    src = open(os.path.join(ROOT, "meshfem_amd", "csrc", "mfh_kernels.hip")).read()
    assert "racy" not in src and "DBG" not in src
