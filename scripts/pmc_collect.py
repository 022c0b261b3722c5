"""Run ON THE GPU BOX: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE -- one counter per run, no trace domains)
over scripts/pmc_probe.py, calibrated on k_axpby's known byte count (gfx950 reports ~1/2 of the coalesced read bytes in
FETCH_SIZE, see /opt/skills/guides/MI355X_MICROARCH.md), reduced to HBM bytes per launch per kernel.
Writes gpurun_out/pmc_traffic_n<N>.json (copy it to profiles/ to have bench.py attach `roofline.traffic`)."""
import csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
slab = sys.argv[2] if len(sys.argv) > 2 else ""        # "slab:WORLD:RANK" (see pmc_probe.py)
out_dir = os.path.join(ROOT, "gpurun_out", "pmc")
os.makedirs(out_dir, exist_ok=True)
raw, meta = {}, None
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(out_dir, counter)
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
           os.path.join(ROOT, "scripts", "pmc_probe.py"), str(n)] + ([slab] if slab else [])
    r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=900)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            meta = json.loads(line)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = {}
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"]
        for key in ("k_geometry", "k_assemble_gather", "k_axpby", "k_spmv<", "k_mf_cluster", "k_mf_rows"):
            if key in name:
                a = acc.setdefault(key.rstrip("<"), [0, 0.0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    raw[counter] = {k: dict(launches=v[0], avg_KB=v[1] / v[0]) for k, v in acc.items()}
    os.replace(f, os.path.join(out_dir, "%s_n%d_counter_collection.csv" % (counter, n)))
true_b = 8.0 * meta["calib_axpby_doubles"]
cal_f = raw["FETCH_SIZE"]["k_axpby"]["avg_KB"] * 1024 / true_b
cal_w = raw["WRITE_SIZE"]["k_axpby"]["avg_KB"] * 1024 / true_b
res = dict(workload=("%d^3 grid, %d P2 tets" % (n, meta["elems"]) + (" (upper-triangle storage)" if meta.get("storage") == "upper" else "")) if not slab else ("%s of a %s grid, %d local P2 tets" % (slab, meta["global_grid"], meta["elems"])), meta=meta, raw=raw,
           calibration=dict(kernel="k_axpby(b=0): reads 8n, writes 8n bytes, n=%d" % meta["calib_axpby_doubles"],
                            fetch_reported_over_true=cal_f, write_reported_over_true=cal_w,
                            note="gfx950 FETCH_SIZE reports ~1/2 of coalesced read bytes (MI355X_MICROARCH.md, HBM section); corrected by the measured factor",
                            gather_patterns="scripts/probe/gather_probe.hip under rocprofv3 --pmc FETCH_SIZE (profiles/r04_gather_probe_fetch_size.csv): 64.0 B counted per "
                                            "128-byte line for the coalesced stream, 66.0 / 66.6 / 67.3 B per line when a lane reads 8 / 24 / 128 bytes of a line of its own "
                                            "(lines in random order), 64.0 B with the lines in order -- the counter counts LINES x 64 B whatever the width of the access, so the "
                                            "streaming factor holds for the record gathers of k_assemble_gather to within 5 % (it over-reports them by that much)"))
# keys = kernel names without template arguments (those changed between rounds; r01 profiles carry the r01 spellings)
names = {"k_assemble_gather": "k_assemble_gather", "k_spmv": "k_spmv", "k_mf_cluster": "k_mf_cluster", "k_mf_rows": "k_mf_rows"}
for k, full in names.items():
    if k not in raw["FETCH_SIZE"]:
        continue
    fb = raw["FETCH_SIZE"][k]["avg_KB"] * 1024 / cal_f
    wb = raw["WRITE_SIZE"][k]["avg_KB"] * 1024 / cal_w
    res[full] = dict(fetch_bytes=fb, write_bytes=wb, traffic_bytes=fb + wb)
tag = ("_" + slab.replace(":", "_")) if slab else ("_upper" if os.environ.get("PMC_UPPER_STORAGE") == "1" else "")
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "pmc_traffic_n%d%s.json" % (n, tag)), "w"), indent=1)
print(json.dumps({k: res[k] for k in names.values() if k in res}, indent=1))
print("calibration", cal_f, cal_w)
