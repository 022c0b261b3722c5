import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import ctypes
        from meshfem_amd import _lib
        lib = _lib.load()
        h = ctypes.c_void_p()
        if lib.mfh_create(0, ctypes.byref(h)) != 0:
            return False
        lib.mfh_destroy(h)
        return True
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu` are skipped (not failed) on a box without a HIP device."""
    if not any("gpu" in it.keywords for it in items) or _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def gpu_available():
    return _have_gpu()


@pytest.fixture(scope="session", autouse=True)
def _build_lib_once():
    # the in-tree .so is rebuilt only when a source is newer (hipcc cross-compiles without a GPU)
    from meshfem_amd import build
    if build.needs_build():
        build.build_lib(verbose=False)
    yield
