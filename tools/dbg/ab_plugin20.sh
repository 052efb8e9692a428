#!/bin/bash
# as ab_plugin.sh, the driver's window only (K = 20, W = 5)
cfg=$1; rounds=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export KMPC_DEBUG=1 KMPC_FORCE_PLUGIN=1 KMPC_KERNEL_CACHE=/tmp/kmpc_ab_cache
for r in $(seq 1 $rounds); do
  for f in "$@"; do
    KMPC_PLUGIN_FLAGS="$f" python3 "$root/bench.py" --config $cfg --cpu-seconds 0 --no-extras --no-probe --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
    print('%-44s K=%-3d %7.2f M  kernel %.4f ms frac %.4f newton %.3f status %s' % ('[$f]', d['steps'], d['value']/1e6, r['avg_kernel_ms'], r['frac'], r['newton_solves_per_step'], d['config']['worst_qp_status']))
except Exception as e:
    print('%-44s failed: %s' % ('[$f]', e))"
  done
done
