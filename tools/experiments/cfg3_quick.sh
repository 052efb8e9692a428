#!/bin/bash
# Dev tool (GPU box): cfg3 (and cfg2) with the dev library
root=${GRAFT_REPO_ROOT:-/root/repo}
export KMPC_LIB=$root/koopman-online-updated-mpc_amd/libkoopmpc_dev.so
for c in cfg3 cfg2; do
python3 $root/bench.py --cpu-seconds 0 --config $c --no-extras --batch $([ $c = cfg3 ] && echo 16384 || echo 4096) 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("%s: %.1f M frac %.3f kernel %.3f ms newton %s" % (d["metric"][-28:], d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_kernel_ms"], d["config"]["qp"][-60:]))'
done
