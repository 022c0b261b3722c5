cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "two_level" 2>&1 | tail -5
MFH_TL_TIMING=1 timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -8
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid
n=60
V,T=grid.grid_tet_mesh(n,n,n,[0,0,0],[1,1,1])
c=M.Context(0); c.mesh_build(T,V,2); c.material_isotropic(200.,0.35)
c.bc_dirichlet_box([-1e-9,-9,-9],[1e-9,9,9],[0,0,0]); c.bc_neumann_box([1-1e-9,-9,-9],[1+1e-9,9,9],[0,-1,0])
c.set_preconditioner(M.PRECOND_TWO_LEVEL)
u=c.sim_solve(rtol=1e-8); print(c.last_info["iterations"], c.last_info["solve_ms"], c.precond_info())
PY
