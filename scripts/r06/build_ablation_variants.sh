# timing-only builds of the library for the row-lane upper-bound experiment (round 6, docs/design/04_2 (xiii)); the default build comes last
set -e
cd "$(dirname "$0")/../.."
mkdir -p meshfem_amd/variants
for v in "abl1:-DMFH_ASM_ABLATE=1" "abl2:-DMFH_ASM_ABLATE=2"; do
  name=${v%%:*}; flags=${v#*:}
  MFH_CXXFLAGS="$flags" python -m meshfem_amd.build --force > /dev/null
  cp meshfem_amd/libmeshfem_hip.so meshfem_amd/variants/libmeshfem_hip_$name.so
done
python -m meshfem_amd.build --force > /dev/null
ls -la meshfem_amd/variants
