"""Print the top kernels of a rocprofv3 results database (rocprofv3 --kernel-trace --stats -d DIR -o NAME).
    python scripts/prof_top.py DIR/NAME_results.db [n]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
print("%-90s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit %d" % n):
    print("%-90s %8d %12.1f %10.1f %6.2f" % (name[:90], calls, total / 1e3, avg / 1e3, pct))
