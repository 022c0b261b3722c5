O=gpurun_out/r06p
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
MFH_MG_TIMING=1 MFH_SOLVE_TIMING=1 MFH_SYM_TIMING=1 timeout 600 python scripts/hom_profile.py 44 > $O/hom_profile.log 2>&1 < /dev/null
grep -n "rep 1" -A26 $O/hom_profile.log | cut -c1-170
grep "multigrid setup\|mfh solve\|symbolic\]" $O/hom_profile.log | tail -42
