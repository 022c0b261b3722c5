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


# the test files that start several ranks (torch.multiprocessing.spawn): they run LAST, so that a lost rendezvous or a slow multi-rank test cannot
# keep the single-process parity tests from being reported under `-x`
_MULTI_RANK_FILES = ("test_gpu_distributed.py", "test_gpu_distributed_multigrid.py", "test_gpu_config4_partitioned.py", "test_gpu_fuzz_scatter.py",
                     "test_gpu_peer_transport.py")


def pytest_collection_modifyitems(config, items):
    """Multi-rank test files last; tests marked `gpu` are skipped (not failed) on a box without a HIP device."""
    items.sort(key=lambda it: 1 if os.path.basename(str(it.fspath)) in _MULTI_RANK_FILES else 0)       # (stable: the order inside the two groups stays)
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


# (bind / listen failures of the rendezvous port only -- "connection reset / refused / timed out" also appear when a rank crashes or hangs mid-test,
# and such a failure must not be retried into a pass)
_RENDEZVOUS_ERRORS = ("Address already in use", "EADDRINUSE", "The server socket has failed to listen", "failed to listen")
SPAWN_RETRIES = []          # every retry on record (printed in the session summary by pytest_terminal_summary)


def pytest_terminal_summary(terminalreporter):
    if SPAWN_RETRIES:
        terminalreporter.write_sep("=", "multi-rank spawns retried on a new rendezvous port: %d" % len(SPAWN_RETRIES))
        for line in SPAWN_RETRIES:
            terminalreporter.write_line("  " + line)


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
        import os
        import warnings
        # the ranks of a test share ONE device: every rank's arena bounds its free memory to its share (mfh_pool.cpp)
        saved = os.environ.get("MFH_DEVICE_SHARERS")
        os.environ["MFH_DEVICE_SHARERS"] = str(max(1, int(nprocs)))
        try:
            try:
                return original(fn, args=args, nprocs=nprocs, join=join, **kw)
            except Exception as e:   # noqa: BLE001
                if not join or len(args) < 2 or not isinstance(args[1], int) or not any(m in str(e) for m in _RENDEZVOUS_ERRORS):
                    raise
                s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
                msg = "spawn of %s: rendezvous port %d could not be bound (%s); once more on port %d" % (getattr(fn, "__name__", "?"), args[1], str(e).splitlines()[-1][:120], port)
                SPAWN_RETRIES.append(msg)
                warnings.warn(msg)
                return original(fn, args=(args[0], port) + tuple(args[2:]), nprocs=nprocs, join=join, **kw)
        finally:
            if saved is None:
                os.environ.pop("MFH_DEVICE_SHARERS", None)
            else:
                os.environ["MFH_DEVICE_SHARERS"] = saved

    mp.spawn = spawn
    yield
    mp.spawn = original
