"""Randomised check of the general-mesh distributed path: a Delaunay mesh of random points that only rank 0 holds, scattered over 2-3 ranks
sharing the GPU (distributed.scatter_mesh), solved with the partitioned multigrid and block-Jacobi PCG, against the single-context solve of the
whole mesh (nodes matched by position).   python scripts/fuzz_scatter.py [first seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_scatter_util import run

if __name__ == "__main__":
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    bad = 0
    for seed in range(s0, s0 + cnt):
        try:
            ok, line = run(seed)
        except Exception as e:   # noqa: BLE001
            ok, line = False, "seed %d: %s: %s" % (seed, type(e).__name__, str(e)[:300])
        print("ok " if ok else "BAD", line, flush=True)
        bad += not ok
    print("failures:", bad)
