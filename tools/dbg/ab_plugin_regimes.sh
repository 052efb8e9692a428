#!/bin/bash
# as ab_plugin20.sh with the extra legs of the line: settled window, replicas' median, cold start, switch window, post-reset
cfg=$1; rounds=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export KMPC_DEBUG=1 KMPC_FORCE_PLUGIN=1 KMPC_KERNEL_CACHE=/tmp/kmpc_ab_cache KMPC_BENCH_NO_F32_LEG=1
for r in $(seq 1 $rounds); do
  for f in "$@"; do
    KMPC_PLUGIN_FLAGS="$f" python3 "$root/bench.py" --config $cfg --cpu-seconds 0 --no-probe --batch 4096 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
    print('%-40s kernel %.4f frac %.4f | replicas med %.4f | cold %.4f | switch %.4f | post-reset %.4f (%.2f solves)' % ('[$f]', r['avg_kernel_ms'], r['frac'], r['replicas']['frac_median'], r['cold_start_frac'], r['switch_window_frac'], r['post_reset_frac'], r['post_reset_newton_solves_per_step']))
except Exception as e:
    print('%-40s failed: %s' % ('[$f]', e))"
  done
done
