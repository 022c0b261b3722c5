# The arena's segments from hipMalloc / hipExtMallocWithFlags(hipDeviceMallocContiguous) / the virtual-memory API: headline kernel, operator, solves.
run() {
  timeout 400 python bench.py --no-strong-n1 --no-cpu --no-orderings --no-config3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); p=d['pcg']; m=d['pcg_multigrid']
print('$1: kernel %.3f step %.3f | operator %.4f bj-iteration %.3f | multigrid %d its %.1f ms | sym %.3f' % (d['roofline']['kernel_ms'], d['ms_per_step'], p['matrix_free']['kernels_ms'], p['ms_per_iteration'], m['iterations'], m['solve_ms'], d['setup']['symbolic_s']))"
}
export MFH_BENCH_NO_RESERVE=1
for i in 1 2; do
  for k in plain contiguous vmm; do MFH_ARENA_ALLOC=$k run "separate, $k"; done
done
unset MFH_BENCH_NO_RESERVE
for k in plain contiguous vmm; do MFH_ARENA_ALLOC=$k run "one reserved segment, $k"; done
