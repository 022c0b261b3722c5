"""Per-kernel AND per-launch-size summary of a rocprofv3 kernel trace (csv): the aggregate-level kernels of the multigrid hierarchy (k_st_*) run on
several levels with one name -- the grid size tells the levels apart, which the --stats table cannot.   python scripts/trace_by_level.py <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: [0, 0.0])
for r in rows:
    name = re.sub(r"\(.*", "", r.get("Kernel_Name", ""))
    name = re.sub(r"^void ", "", name).replace("mfh::k::", "")
    dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6      # ms
    grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    key = (name, grid if name.startswith(("k_st_", "k_tl_", "k_mg_")) else "*")
    a = acc[key]
    a[0] += 1
    a[1] += dur
total = sum(v[1] for v in acc.values())
print("%-78s %10s %8s %12s %10s" % ("kernel", "grid", "calls", "total ms", "avg us"))
for (name, grid), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%-78s %10s %8d %12.3f %10.2f" % (name[:78], grid, n, t, t / n * 1e3))
print("total %.3f ms in %d launches" % (total, len(rows)))
