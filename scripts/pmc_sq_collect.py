"""Run ON THE GPU BOX: SQ counter passes (one rocprofv3 --pmc run per set, no trace domains) over scripts/pmc_probe.py;
prints per-kernel averages for the kernels named on the command line (default: the matrix-free operator)."""
import csv, glob, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
keys = sys.argv[2:] or ["k_mf_cluster", "k_mf_rows", "k_assemble_gather"]
SETS = [["SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"],
        ["SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_VMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_WAVES"]]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for si, cs in enumerate(SETS):
    d = os.path.join(ROOT, "gpurun_out", "pmc_sq", "set%d" % si)
    cmd = ["rocprofv3", "--pmc"] + cs + ["--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "scripts", "pmc_probe.py"), str(n)]
    subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=900)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    for row in csv.DictReader(open(f)):
        for k in keys:
            if k in row["Kernel_Name"]:
                a = acc[k][row["Counter_Name"]]
                a[0] += 1; a[1] += float(row["Counter_Value"])
                acc[k]["_regs"] = [1, float(row["VGPR_Count"])]; acc[k]["_lds"] = [1, float(row["LDS_Block_Size"])]
    os.replace(f, os.path.join(ROOT, "gpurun_out", "pmc_sq", "sq_set%d_n%d_counter_collection.csv" % (si, n)))
for k in keys:
    print(k, {c: v[1] / v[0] for c, v in sorted(acc[k].items())})
