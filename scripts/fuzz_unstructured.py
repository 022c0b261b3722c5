"""Randomised self-consistency check on unstructured meshes (tests/fuzz_unstructured_util.py): many seeds.
    python scripts/fuzz_unstructured.py [first seed] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_unstructured_util import check

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for seed in range(s0, s0 + cnt):
    try:
        ok, line = check(seed)
    except Exception as e:   # noqa: BLE001
        ok, line = False, "seed %d: %s: %s" % (seed, type(e).__name__, str(e)[:200])
    print("ok " if ok else "BAD", line, flush=True)
    bad += not ok
print("failures:", bad)
