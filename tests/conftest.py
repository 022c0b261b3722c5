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


_RENDEZVOUS_ERRORS = ("Address already in use", "EADDRINUSE", "Connection refused", "Connection reset", "failed to connect", "connect() timed out",
                      "Socket Timeout", "The client socket has timed out", "The server socket has failed to listen")


@pytest.fixture(scope="session", autouse=True)
def _spawn_retries_a_lost_port():
    """The multi-rank tests pick a free port, close it and hand it to their ranks (torch.multiprocessing.spawn(worker, args=(world, port, ...))): on a
    busy box another process can take the port in between, and the rendezvous of one rank fails. Such a failure -- and only such a failure -- is
    retried once on a new port; anything a worker raises on its own goes through unchanged."""
    try:
        import torch.multiprocessing as mp
    except Exception:   # noqa: BLE001
        yield
        return
    import socket
    original = mp.spawn

    def spawn(fn, args=(), nprocs=1, join=True, **kw):
        try:
            return original(fn, args=args, nprocs=nprocs, join=join, **kw)
        except Exception as e:   # noqa: BLE001
            if not join or len(args) < 2 or not isinstance(args[1], int) or not any(m in str(e) for m in _RENDEZVOUS_ERRORS):
                raise
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            print("spawn: rendezvous on port %d failed (%s); once more on port %d" % (args[1], str(e).splitlines()[-1][:120], port), flush=True)
            return original(fn, args=(args[0], port) + tuple(args[2:]), nprocs=nprocs, join=join, **kw)

    mp.spawn = spawn
    yield
    mp.spawn = original
