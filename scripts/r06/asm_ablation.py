"""Upper bound of a row-lane decomposition of k_assemble_gather, measured before building it: timing-only builds in which the
loads a row-lane kernel would share between the contributions of one (element, row node) -- the row node's two support gradients, the volume, the Lame
parameters: 9 of the 15 doubles of element record per contribution -- are simply not made (variant 1), and in which no element record is read at all
(variant 2). configs[2], one process per sample; then SQ / TCP counters of the default and of variant 1 (rocprofv3 --pmc, one set per run).
    python scripts/r06/asm_ablation.py            (child: ... child [pmc])"""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import meshfem_amd as M
    from meshfem_amd import grid
    n = 60
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    c.material_isotropic(200.0, 0.35)
    c.symbolic(False)
    c.assemble()
    if len(sys.argv) > 2:            # under rocprofv3 --pmc: a few launches are enough
        c.time_assembly_kernel(M.ASSEMBLE_GATHER, 4)
    else:
        t = [c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20) for _ in range(3)]
        print("kernel_ms %s" % " ".join("%.3f" % v for v in t), flush=True)
    sys.exit(0)

VARIANTS = [("default", None), ("no row-node loads (row-lane upper bound)", "abl1"), ("no element record at all", "abl2")]


def env_for(tag):
    env = dict(os.environ)
    if tag:
        env["MESHFEM_HIP_LIB"] = os.path.join(ROOT, "meshfem_amd", "variants", "libmeshfem_hip_%s.so" % tag)
    return env


for rep in range(0 if os.environ.get("ABL_SKIP_TIMING") else 3):
    for name, tag in VARIANTS:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env_for(tag), capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("kernel_ms")]
        print("%-44s %s" % (name, line[0] if line else "FAILED " + out.stderr[-300:].replace("\n", " | ")), flush=True)

SETS = [["SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"],
        ["SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_VMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_WAVES"],
        ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"], ["TCP_PENDING_STALL_CYCLES_sum", "TCP_TA_DATA_STALL_CYCLES_sum"]]
# (a first version asked for seven TCP counters in one pass: rocprofv3 never returned -- the TCP sets are small and have a short timeout)
for name, tag in VARIANTS[:2]:
    acc = collections.defaultdict(lambda: [0, 0.0])
    for si, cs in enumerate(SETS):
        d = os.path.join(ROOT, "gpurun_out", "r06_abl_pmc", (tag or "default") + "_set%d" % si)
        cmd = ["rocprofv3", "--pmc"] + cs + ["--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "child", "pmc"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(env_for(tag), TMPDIR="/tmp"), timeout=900 if si < 2 else 150)
        except subprocess.TimeoutExpired:
            print("%s set %d (%s): rocprofv3 did not return" % (name, si, " ".join(cs)), flush=True)
            continue
        fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not fs:
            print("%s set %d: no counter file (%s)" % (name, si, r.stderr[-200:].replace("\n", " | ")), flush=True)
            continue
        for row in csv.DictReader(open(fs[0])):
            if "k_assemble_gather" in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]
                a[0] += 1; a[1] += float(row["Counter_Value"])
                acc["_VGPRs"] = [1, float(row["VGPR_Count"])]
    avg = {k: v[1] / v[0] for k, v in sorted(acc.items())}
    print("counters per launch, %s:" % name, {k: (round(v) if v > 100 else round(v, 3)) for k, v in avg.items()}, flush=True)
    if avg.get("SQ_WAVE_CYCLES"):
        w = avg["SQ_WAVE_CYCLES"]
        print("   VALU-busy %.3f of wave cycles, waiting %.3f, LDS-active %.3f, VMEM-active %.3f; LDS bank-conflict share of LDS index cycles %.3f; VALU instructions per wave %.0f" % (
            avg.get("SQ_ACTIVE_INST_VALU", 0) / w, avg.get("SQ_WAIT_ANY", 0) / w, avg.get("SQ_ACTIVE_INST_LDS", 0) / w, avg.get("SQ_ACTIVE_INST_VMEM", 0) / w,
            avg.get("SQ_LDS_BANK_CONFLICT", 0) / max(avg.get("SQ_LDS_IDX_ACTIVE", 1), 1), avg.get("SQ_INSTS_VALU", 0) / max(avg.get("SQ_WAVES", 1), 1)), flush=True)
