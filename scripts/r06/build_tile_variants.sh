# timing-only builds of the library with other tile geometries of K's value array (round 6, docs/design/04_2 (xii)); the default build comes last
set -e
cd "$(dirname "$0")/../.."
for v in "t32:-DMFH_TILE_LOG=5" "t128:-DMFH_TILE_LOG=7" "pad16:-DMFH_TILE_PAD=16" "pad272:-DMFH_TILE_PAD=272"; do
  name=${v%%:*}; flags=${v#*:}
  MFH_CXXFLAGS="$flags" python -m meshfem_amd.build --force > /dev/null
  cp meshfem_amd/libmeshfem_hip.so meshfem_amd/variants/libmeshfem_hip_$name.so
done
python -m meshfem_amd.build --force > /dev/null
ls -la meshfem_amd/variants
