"""Register / scratch usage of the kernels of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage; CPU only).
    python scripts/kernel_regs.py <file.hip> <name substring> [extra hipcc flags ...]"""
import re
import subprocess
import sys

src, pat, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Rpass-analysis=kernel-resource-usage",
       "-c", src, "-o", "/tmp/_regs.o"] + flags
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = dict(name=subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip())
        continue
    if cur is None:
        continue
    for key in ("VGPRs:", "AGPRs:", "ScratchSize [bytes/lane]:", "Occupancy [waves/SIMD]:", "VGPRs Spill:", "LDS Size"):
        if key in line:
            cur[key] = line.split(key)[1].split("[")[0].strip()
    if "LDS Size" in line:
        if pat in cur["name"]:
            print("%-70s VGPRs %s  scratch %s  spill %s  occupancy %s" % (cur["name"].split("(")[0][-70:], cur.get("VGPRs:"), cur.get("ScratchSize [bytes/lane]:"),
                                                                    cur.get("VGPRs Spill:"), cur.get("Occupancy [waves/SIMD]:")))
        cur = None
