import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
import numpy as np
import meshfem_amd as M
from meshfem_amd import mesh_io, homogenization as H
from oracle import meshfem_oracle as O
GOLD = os.path.join(ROOT, "tests", "golden")
V, E, _ = mesh_io.load_msh(os.path.join(GOLD, "meshes", "cube_cross.msh"))
base = O.ElasticityTensor.isotropic(3, 200.0, 0.35)
for rtol in (1e-8, 1e-9, 1e-10, 3e-11, 1e-11):
    try:
        res = H.homogenize(V[:, :3], E, 2, Cbase=base.D, rtol=rtol, preconditioner=M.PRECOND_MULTIGRID)
        print("MFH_OPTIONS=%-18s rtol %g ok" % (os.environ.get("MFH_OPTIONS", ""), rtol), res["Ch"][0, 0], res["iterations"], flush=True)
    except Exception as e:
        print("MFH_OPTIONS=%-18s rtol %g FAILED" % (os.environ.get("MFH_OPTIONS", ""), rtol), str(e)[:100], flush=True)
