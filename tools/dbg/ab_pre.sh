#!/bin/bash
# Dev tool (GPU box): the driver's window of cfg2 (all legs, three replicas) under several environment settings of ONE library,
# alternating:   LIB=libkoopmpc_dev.so tools/dbg/ab_pre.sh KMPC_PLACE_TAIL=5 KMPC_PLACE_TAIL=3 "X=1"
# (round 5 used it with a KMPC_BENCH_PRE switch of bench.py -- the length of the launch in front of the timed one -- that is gone again:
#  the launch in front is the W warm-up steps, as the driver runs it; profiles/r5_cfg2_placement_window.txt)
root=${GRAFT_REPO_ROOT:-/root/repo}
export KMPC_DEBUG=1
for rep in 1 2; do for v in "$@"; do
env $v KMPC_LIB=$root/koopman-online-updated-mpc_amd/${LIB:-libkoopmpc_devbase.so} python3 $root/bench.py --cpu-seconds 0 --config cfg2 --no-probe --steps 20 --warmup 5 --replicas 3 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print("%-44s K=%-3d %.1f M kernel %.4f frac %.3f | cold %.3f post %.3f switch %.3f | replicas %s" % (sys.argv[1], d["steps"], d["value"]/1e6, r["avg_kernel_ms"], r["frac"], r.get("cold_start_frac",0), r.get("post_reset_frac",0), r.get("switch_window_frac") or 0, (r.get("replicas") or {}).get("kernel_ms")))' "$v"
done; done
