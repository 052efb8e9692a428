#!/bin/bash
# Dev tool (GPU box): cfg2 with the dev library -- default window and the driver's arguments, warm and cold legs
root=${GRAFT_REPO_ROOT:-/root/repo}
export KMPC_LIB=$root/koopman-online-updated-mpc_amd/libkoopmpc_dev.so
for a in "" "--steps 20 --warmup 5"; do
python3 $root/bench.py --cpu-seconds 0 --config cfg2 --batch 4096 $a 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("steps %d: %.1f M frac %.3f kernel %.3f ms | cold %.1f M (%.2f solves) | post-reset %.1f M" % (d["steps"], d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_kernel_ms"], d["roofline"]["cold_start_value"]/1e6, d["roofline"].get("newton_solves_per_step", 0), d["roofline"]["post_reset_value"]/1e6))'
done
