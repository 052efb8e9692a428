#!/bin/bash
# Dev tool (GPU box): several builds of the library on ONE box, alternating, on the driver's window of cfg2 with all its legs
#   REPS=2 tools/dbg/ab_libs.sh <libA.so> <libB.so> ...
root=${GRAFT_REPO_ROOT:-/root/repo}
export KMPC_DEBUG=1
for rep in $(seq 1 ${REPS:-2}); do for l in "$@"; do
KMPC_LIB=$root/koopman-online-updated-mpc_amd/$l python3 $root/bench.py --cpu-seconds 0 --config ${CFG:-cfg2} --no-probe --steps 20 --warmup 5 --replicas 3 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print("%-26s K=%-3d %.1f M kernel %.4f frac %.3f | cold %.3f post %.3f switch %.3f | replicas %s" % (sys.argv[1], d["steps"], d["value"]/1e6, r["avg_kernel_ms"], r["frac"], r.get("cold_start_frac",0), r.get("post_reset_frac",0), r.get("switch_window_frac") or 0, (r.get("replicas") or {}).get("kernel_ms")))' "$l"
done; done
