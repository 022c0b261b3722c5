cd $GRAFT_REPO_ROOT
for MODE in none torch_warm; do
timeout 600 python - $MODE <<'PY'
import sys; sys.path.insert(0,'.')
mode=sys.argv[1]
import torch
import meshfem_amd as M
from meshfem_amd import grid
if mode=='torch_warm':
    x=torch.empty(int(40e9),dtype=torch.uint8,device='cuda'); x.fill_(1); torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
V,T=grid.grid_tet_mesh(60,60,60,[0,0,0],[1,1,1])
c=M.Context(0); c.mesh_build(T,V,2); c.material_isotropic(200,0.35); c.assemble()
print(mode,'first ctx: asm ms',round(c.time_assembly_kernel(M.ASSEMBLE_GATHER,3),3),'spmv ms',round(c.time_spmv_kernel(10),3), round(c.time_spmv_kernel(10),3), flush=True)
PY
done
