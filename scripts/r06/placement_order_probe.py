"""Where the K values end up, by what the process did before: the assembly kernel's time at configs[2] (HIP events, 20 launches).
python scripts/r06/placement_order_probe.py <mode>
  plain            no torch in the process, no reservation (the arena takes segments on demand)
  reserve          no torch, mfh_device_reserve_for first
  torch_reserve    torch touches the device first, then the reservation
  reserve_torch    the reservation first, then torch touches the device (bench.py)
  torch_plain      torch touches the device, no reservation"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
mode = sys.argv[1]
import meshfem_amd as M
from meshfem_amd import grid
if mode in ("reserve", "reserve_torch"):
    M.device_reserve_for(3, 2, 24 * 60 ** 3)
if "torch" in mode:
    import torch
    torch.cuda.set_device(0)
    x = torch.zeros(1 << 20, device="cuda:0")
    torch.cuda.synchronize()
if mode == "torch_reserve":
    M.device_reserve_for(3, 2, 24 * 60 ** 3)
n = 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
T = np.ascontiguousarray(T, dtype=np.int32)
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.symbolic(False)
c.set_option("reembed", 1)
for _ in range(3):
    c.assemble()
c.dev_sync()
print("%-14s kernel %.4f ms" % (mode, c.time_assembly_kernel(M.ASSEMBLE_GATHER, 20)), flush=True)
c.close()
