set -x
for o in mg_over_correction=1.0 mg_over_correction=1.3 mg_steps_coarse=2,mg_ratio_coarse=0.15 mg_steps_fine=2 mg_coarse_cycles=2 mg_steps_agg=3,mg_ratio_agg=0.1 mg_agg_target=16 mg_agg_target=64; do
  echo "== $o"; MFH_OPTIONS=$o timeout 300 python scripts/config4_homogenization.py 44 --only-mg 2>&1 | grep "^multigrid" | cut -c1-330
done
