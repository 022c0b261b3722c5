cd $GRAFT_REPO_ROOT
MFH_TL_TIMING=1 timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -12
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid
n=60
V,T=grid.grid_tet_mesh(n,n,n,[0,0,0],[1,1,1])
c=M.Context(0); c.mesh_build(T,V,2); c.material_isotropic(200.,0.35)
c.bc_dirichlet_box([-1e-9,-9,-9],[1e-9,9,9],[0,0,0]); c.bc_neumann_box([1-1e-9,-9,-9],[1+1e-9,9,9],[0,-1,0])
c.set_preconditioner(M.PRECOND_TWO_LEVEL)
u=c.sim_solve(rtol=1e-8); print(c.last_info, c.precond_info())
PY
