set -x
python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python bench.py --gpus 2 --ranks-per-gpu-ok --grid 24 --steps 3 --warmup 1 --no-cpu > gpurun_out/r03_n2_weak.json 2> gpurun_out/r03_n2_weak.err; tail -3 gpurun_out/r03_n2_weak.err; python -c "
import json; d=json.loads(open('gpurun_out/r03_n2_weak.json').read().strip().splitlines()[-1]); print(d['scaling'], d['value'], d['config']['workload'], d['pcg'].get('iterations'), d['pcg'].get('transports_tried'), d['pcg'].get('error'))"
python bench.py --gpus 2 --ranks-per-gpu-ok --scaling strong --grid 23 --steps 3 --warmup 1 --no-cpu > gpurun_out/r03_n2_strong.json 2> gpurun_out/r03_n2_strong.err; tail -3 gpurun_out/r03_n2_strong.err; python -c "
import json; d=json.loads(open('gpurun_out/r03_n2_strong.json').read().strip().splitlines()[-1]); print(d['scaling'], d['value'], d['config']['workload'], d['pcg'].get('iterations'), d['pcg'].get('solve_s'), d['pcg'].get('error'))"
python bench.py --gpus 1 --scaling strong --grid 23 --steps 3 --warmup 1 --no-cpu > gpurun_out/r03_n1_strong.json 2> gpurun_out/r03_n1_strong.err; tail -3 gpurun_out/r03_n1_strong.err; python -c "
import json; d=json.loads(open('gpurun_out/r03_n1_strong.json').read().strip().splitlines()[-1]); print(d['scaling'], d['value'], d['config']['workload'], d['pcg'].get('iterations'), d['pcg'].get('solve_s'))"
