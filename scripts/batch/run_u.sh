set -x
timeout 400 python -m pytest tests/test_gpu_multigrid.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python scripts/config4_homogenization.py 44 --skip-bj 2>&1 | grep -v amdgpu | cut -c1-600
