#!/bin/bash
# Dev tool (GPU box): A/B of two builds of the library on ONE box, alternating (box-to-box spread is +-3 %)
#   tools/dbg/ab.sh <libA.so> <libB.so> [configs...]
root=${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; shift 2
cfgs=${@:-cfg2 cfg3}
for rep in 1 2; do
for lib in $A $B; do
for c in $cfgs; do
KMPC_LIB=$root/koopman-online-updated-mpc_amd/$lib python3 $root/bench.py --cpu-seconds 0 --config $c --no-extras --batch $(python3 -c "import sys; sys.path.insert(0,'$root'); import bench; print(bench.CONFIGS['$c']['B'])") 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("%-22s %-30s %.1f M  kernel %.3f ms" % (sys.argv[1], d["metric"][-28:], d["value"]/1e6, d["roofline"]["avg_kernel_ms"]))' $lib
done; done; done
