#!/bin/bash
# Dev tool (GPU box): ONE library under several environment settings, alternating (cfg2, K = 20 and K = 200)
#   REPS=3 tools/dbg/ab_env.sh <lib.so> <VAR=value> [<VAR=value> ...]      ("X=1" = no switch)
root=${GRAFT_REPO_ROOT:-/root/repo}
export KMPC_DEBUG=1
lib=$1; shift
for rep in $(seq 1 ${REPS:-3}); do
for v in "$@"; do
for w in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do
env $v KMPC_LIB=$root/koopman-online-updated-mpc_amd/$lib python3 $root/bench.py --cpu-seconds 0 --config ${CFG:-cfg2} --no-probe --no-extras $w 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print("%-34s K=%-3d %.1f M  kernel %.4f ms frac %.3f" % (sys.argv[1], d["steps"], d["value"]/1e6, r["avg_kernel_ms"], r["frac"]))' "$v"
done; done; done
