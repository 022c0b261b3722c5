"""Run ON THE GPU BOX: where do the latency-bound kernels wait? One rocprofv3 --pmc pass per counter set (no trace domains)
over scripts/pmc_probe.py; per-kernel averages of SQ / TA / TCP / TCC counters -> gpurun_out/pmc_deep_n<N>.json.
    python scripts/pmc_deep.py [grid] [kernel substrings ...]"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
keys = sys.argv[2:] or ["k_mf_cluster", "k_mf_rows", "k_assemble_gather", "k_spmv<"]
SETS = [
    ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"],   # (GRBM_* counters hang the collection)
    ["TA_BUSY_avr", "TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_TOTAL_WAVEFRONTS_sum",
     "TA_FLAT_READ_WAVEFRONTS_sum", "TA_FLAT_WRITE_WAVEFRONTS_sum"],
    ["TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_GATE_EN1_sum",
     "TCP_TA_TCP_STATE_READ_sum"],
    ["TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_LEVEL_sum", "TCC_BUSY_avr", "TCC_TAG_STALL_sum"],
    ["SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INST_CYCLES_VMEM_RD", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL",
     "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"],
    ["SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_INST_LEVEL_LDS"],
]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for si, cs in enumerate(SETS):
    d = os.path.join(ROOT, "gpurun_out", "pmc_deep", "set%d" % si)
    cmd = ["rocprofv3", "--pmc"] + cs + ["--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "scripts", "pmc_probe.py"), str(n)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=150)
    except subprocess.TimeoutExpired:
        print("set", si, "timed out", flush=True)
        continue
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        print("set", si, "failed:", r.stderr[-400:])
        continue
    for row in csv.DictReader(open(fs[0])):
        for k in keys:
            if k in row["Kernel_Name"]:
                a = acc[k][row["Counter_Name"]]
                a[0] += 1; a[1] += float(row["Counter_Value"])
                acc[k]["_VGPRs"] = [1, float(row["VGPR_Count"])]; acc[k]["_LDS_bytes"] = [1, float(row["LDS_Block_Size"])]
out = {k: {c: v[1] / v[0] for c, v in sorted(acc[k].items())} for k in keys}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "pmc_deep_n%d.json" % n), "w"), indent=1)
for k in keys:
    print(k, json.dumps(out[k]))
