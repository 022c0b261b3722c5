cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -5
MFH_BENCH_FORCE_DISTRIBUTED=1 timeout 600 python bench.py --steps 5 --warmup 1 --grid 30 2>&1 | tail -3
timeout 300 python - <<'PY'
import sys; sys.path.insert(0,'.')
import meshfem_amd as M
from meshfem_amd import grid
V,T=grid.grid_tet_mesh(40,40,40,[0,0,0],[1,1,1])
for order in (1,0):
  for slots in (256,512):
    c=M.Context(0); c.set_option('contrib_order',order); c.set_option('chunk_slots',slots); c.mesh_build(T,V,2); c.material_isotropic(200,0.35); c.symbolic(False)
    out=[]
    for dbg in (0,1,2):
        c.set_option('debug_variant',dbg); out.append(round(c.time_assembly_kernel(M.ASSEMBLE_GATHER,5),3))
    print('order',order,'slots',slots,'ms normal/racy-rmw/store-only',out, flush=True)
    c.close()
PY
cd /tmp && export TMPDIR=/tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$CNT -o pmc -- python $R/scripts/pmc_probe.py 40 > $R/gpurun_out/pmc_$CNT.json 2> $R/gpurun_out/pmc_$CNT.err
  echo "pmc $CNT rc=$?"; cat $R/gpurun_out/pmc_$CNT.json; ls $R/gpurun_out/pmc_$CNT | head
done
