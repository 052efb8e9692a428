#!/bin/bash
# Dev tool (GPU box): A/B/... of builds of the library on ONE box, alternating; cfg2 on the driver's window (K = 20) and the default one
#   REPS=2 ARGS="--no-extras" [CFG=cfg5] tools/dbg/ab2.sh <libA.so> <libB.so> ...
root=${GRAFT_REPO_ROOT:-/root/repo}
export KMPC_DEBUG=1  # (the library reads its measurement switches only with this set)
for rep in $(seq 1 ${REPS:-2}); do
for lib in "$@"; do
for w in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do
KMPC_LIB=$root/koopman-online-updated-mpc_amd/$lib python3 $root/bench.py --cpu-seconds 0 --config ${CFG:-cfg2} --no-probe $w $ARGS 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print("%-24s K=%-3d %.1f M  kernel %.4f ms frac %.3f | cold %.1f M (%.3f) post-reset %.1f M | status %d" % (sys.argv[1], d["steps"], d["value"]/1e6, r["avg_kernel_ms"], r["frac"], r.get("cold_start_value",0)/1e6, r.get("cold_start_frac",0), r.get("post_reset_value",0)/1e6, d["config"]["worst_qp_status"]))' $lib
done; done; done
