"""Order in which the lanes of a wave walk the 9 components of their block in the LDS accumulation of k_assemble_gather (MFH_ASM_ROT builds): same order for
all lanes (shipped) / even and odd lanes in opposite orders / three rotations by lane % 3. configs[2], one process per sample.
    python scripts/r06/rot_probe.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = os.path.join(ROOT, "scripts", "r06", "tile_probe.py")
for rep in range(3):
    for name, tag in (("same order (shipped)", None), ("even / odd lanes opposite", "rot2"), ("three rotations", "rot3")):
        env = dict(os.environ)
        if tag:
            env["MESHFEM_HIP_LIB"] = os.path.join(ROOT, "meshfem_amd", "variants", "libmeshfem_hip_%s.so" % tag)
        out = subprocess.run([sys.executable, child, "child"], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("kernel_ms")]
        print("%-28s %s" % (name, line[0] if line else "FAILED " + out.stderr[-200:]), flush=True)
