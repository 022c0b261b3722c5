O=gpurun_out/r06m
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_unstructured.py tests/test_periodic_conditions.py tests/test_homogenization_analytic.py -x -q -m gpu -k "anisotropic or tensor or general or fuzz or periodic or configs3" > $O/tests.log 2>&1 < /dev/null
tail -3 $O/tests.log
timeout 600 python scripts/r06/op_materials.py 44 > $O/op_materials.txt 2>&1 < /dev/null
cat $O/op_materials.txt
timeout 600 python bench.py --leg config3 > $O/leg_config3.json 2> $O/leg_config3.err < /dev/null
python -c "
import json; d=json.load(open('$O/leg_config3.json')); print(d['wall_s'])"
