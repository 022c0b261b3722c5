O=gpurun_out/r06n
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for i in 1 2; do
echo "3 waves (160 VGPRs):"; timeout 600 python scripts/r06/op_materials.py 44 2>/dev/null | grep general
echo "2 waves (204 VGPRs):"; MESHFEM_HIP_LIB=$R/meshfem_amd/variants/libmeshfem_hip_gen2w.so timeout 600 python scripts/r06/op_materials.py 44 2>/dev/null | grep general
done > $O/general_ab.txt 2>&1
cat $O/general_ab.txt
