set -x
timeout 600 python scripts/config4_homogenization.py 44 --skip-bj 2>&1 | grep -v amdgpu | cut -c1-700
