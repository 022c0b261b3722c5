"""Row chunks of the assembly kernel (symbolic phase, host side): the greedy scan -- every chunk takes as many whole rows as fit into
chunk_slots slots -- is run by the host threads over ranges of rows and stitched; the result must be the plain sequential scan's, whatever the
range size (down to ranges shorter than a chunk), with forced chunk ends ("breaks": the cuts the symbolic phase adds on meshes of 2^25 elements
and more, mfh_symbolic_gpu.hip) and with rows of any length up to the chunk size."""
import ctypes as C

import numpy as np
import pytest


def _chunks(lib, row_ptr, slots, breaks, grain, threads):
    n = len(row_ptr) - 1
    out = np.empty(n + 2, dtype=np.int32)
    n_out = C.c_int64()
    br = np.ascontiguousarray(breaks, dtype=np.int64)
    st = lib.mfh_debug_row_chunks(n, row_ptr.ctypes.data, slots, len(br), br.ctypes.data if len(br) else None, grain, threads, out.ctypes.data, len(out),
                                  C.byref(n_out))
    assert st == 0
    return out[: n_out.value].copy()


def _reference(row_ptr, slots, breaks):
    n, out, r, bs = len(row_ptr) - 1, [0], 0, set(int(b) for b in breaks)
    while r < n:
        r2 = r + 1
        while r2 < n and r2 not in bs and row_ptr[r2 + 1] - row_ptr[r] <= slots:
            r2 += 1
        out.append(r2)
        r = r2
    return np.array(out, dtype=np.int32)


@pytest.mark.parametrize("seed", range(6))
def test_threaded_scan_equals_the_sequential_one(seed):
    from meshfem_amd._lib import load
    lib = load()
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2000, 40000))
    slots = int(rng.choice([64, 128, 256]))
    lens = rng.integers(1, [8, 30, slots][seed % 3] + 1, size=n)
    if seed % 2:
        lens[rng.integers(0, n, size=20)] = slots          # rows that fill a chunk on their own
    row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    breaks = np.sort(rng.choice(np.arange(1, n), size=[0, 3, 40][seed % 3], replace=False))
    want = _reference(row_ptr, slots, breaks)
    seq = _chunks(lib, row_ptr, slots, breaks, 1 << 18, 1)
    assert np.array_equal(seq, want)
    for grain, threads in ((1 << 18, 0), (1000, 8), (97, 8), (5, 64), (1, 3)):
        got = _chunks(lib, row_ptr, slots, breaks, grain, threads)
        assert np.array_equal(got, want), (grain, threads)
    # properties the kernel relies on: whole rows, at most `slots` slots, a chunk ends at every break
    size = row_ptr[want[1:]] - row_ptr[want[:-1]]
    assert size.max() <= slots and want[-1] == n and np.all(np.diff(want) > 0) and np.all(np.isin(breaks, want))
