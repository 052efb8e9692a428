#!/bin/bash
# Dev tool (GPU box): A/B/... of compile-time settings of the fused roll-out kernel on ONE box, alternating -- every setting is a roll-out
# plug-in compiled on the box from the shipped sources with extra flags (KMPC_FORCE_PLUGIN, KMPC_PLUGIN_FLAGS; 2 s per kernel).
#   tools/dbg/ab_plugin.sh <config> <rounds> "<flags A>" "<flags B>" ...      (an empty string = the sources as they are)
cfg=$1; rounds=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export KMPC_DEBUG=1 KMPC_FORCE_PLUGIN=1 KMPC_KERNEL_CACHE=/tmp/kmpc_ab_cache
for r in $(seq 1 $rounds); do
  for f in "$@"; do
    for kw in "20 5" "200 20"; do
      k=${kw% *}; w=${kw#* }
      KMPC_PLUGIN_FLAGS="$f" python3 "$root/bench.py" --config $cfg --cpu-seconds 0 --no-extras --no-probe --steps $k --warmup $w 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-28s K=%-3d %7.2f M  kernel %.4f ms frac %.4f newton %.3f status %s' % ('[$f]', d['steps'], d['value']/1e6, r['avg_kernel_ms'], r['frac'], r['newton_solves_per_step'], d['config']['worst_qp_status']))"
    done
  done
done
