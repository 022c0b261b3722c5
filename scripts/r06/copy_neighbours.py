"""Which kernels sit around the __amd_rocclr_copyBuffer launches of a rocprofv3 kernel trace (who issues the copies?).   python scripts/r06/copy_neighbours.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("void mfh::k::", "").replace("mfh::k::", "").split("(")[0][:40]
pairs = collections.Counter()
sizes = collections.Counter()
for i, r in enumerate(rows):
    if "copyBuffer" in r["Kernel_Name"]:
        prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
        pairs[(prev, nxt)] += 1
        sizes[(r.get("Grid_Size_X") or r.get("Grid_Size"), int((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000))] += 1
for (a, b), n in pairs.most_common(12):
    print("%5d  after %-42s before %s" % (n, a, b))
print("grid size, duration us:", sizes.most_common(8))
