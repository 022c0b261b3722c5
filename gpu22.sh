cd /tmp && export TMPDIR=/tmp
cat > /tmp/probe.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import meshfem_amd as M
from meshfem_amd import grid
n=60
V,T=grid.grid_tet_mesh(n,n,n,[0,0,0],[1,1,1])
c=M.Context(0); c.mesh_build(T,V,2); c.material_isotropic(200.,0.35); c.assemble()
c.set_option("matrix_free",1)
for rows,pairs in ((256,2048),(512,4096),(1024,8192),(128,1024)):
    c.set_option("mf_chunk_rows",rows); c.set_option("mf_chunk_pairs",pairs)
    print("rows",rows,"pairs",pairs,"total ms", c.time_spmv_kernel(10), flush=True)
PY
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mf -o mf2 -- python /tmp/probe.py 2>&1 | grep "total ms"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/prof_mf/mf2_kernel_trace.csv")) if "k_mf_rows" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in rows]
print("k_mf_rows launches", len(d)); 
for i in range(0,len(d),11): print([round(x,3) for x in d[i:i+11]][-3:])
f=[ (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in csv.DictReader(open("gpurun_out/prof_mf/mf2_kernel_trace.csv")) if "k_mf_forces" in r["Kernel_Name"]]
print("forces", sum(f)/len(f))
PY
rm -f gpurun_out/prof_mf/*kernel_trace.csv
