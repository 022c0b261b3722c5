# Does the RELATIVE placement of a context's buffers inside one reserved segment set the assembly kernel's time?
# (reservation: 3.27 ms on every box; separate hipMallocs: 3.08-3.21.)  Granularity / stagger of the large class varied through the experiment knobs.
run() {
  timeout 300 python bench.py --no-strong-n1 --no-cpu --no-solve --no-orderings --no-config3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('$1 kernel %.3f step %.3f'%(d['roofline']['kernel_ms'],d['ms_per_step']))"
}
unset MFH_BENCH_NO_RESERVE MFH_ARENA_GRAN_KB MFH_ARENA_STAGGER_KB
run "reserve gran=2048 stagger=0"
MFH_BENCH_NO_RESERVE=1 run "separate hipMallocs"
for gs in "4 0" "4 4" "4 68" "4 260" "4 1028" "64 192" "4 12" "4 516" "4 4100" "2048 2048"; do
  set -- $gs
  MFH_ARENA_GRAN_KB=$1 MFH_ARENA_STAGGER_KB=$2 run "reserve gran=$1 stagger=$2"
done
MFH_BENCH_NO_RESERVE=1 run "separate hipMallocs"
run "reserve gran=2048 stagger=0"
