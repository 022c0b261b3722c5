# option placement_trials through MFH_OPTIONS on the default bench leg: kernel time and first assembly, alternating 0 / 2 / 3 trials
run() {
  timeout 400 python bench.py --no-strong-n1 --no-cpu --no-orderings --no-config3 --no-solve 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
print('$1: kernel %.3f step %.3f first assembly %.1f ms' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['setup']['first_assembly_ms']))"
}
for i in 1 2 3 4; do
  for t in 0 2 3; do MFH_OPTIONS="placement_trials=$t" run "placement_trials $t"; done
done
