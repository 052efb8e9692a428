#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel durations of tools/dbg/lift_gram_timing.py for a few batch sizes
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/gpurun_out/lift_gram_prof; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for B in "$@"; do
  rm -rf "$out/trace"
  rocprofv3 --kernel-trace --stats -d "$out/trace" -o t --output-format csv -- python3 "$root/tools/dbg/lift_gram_timing.py" $B > "$out/run_$B.log" 2>&1
  f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
  echo "== B = $B (KMPC_LIFT_GRAM_CT1=${KMPC_LIFT_GRAM_CT1:-})"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "lift_coop" in n or "gram_reduce" in n:
        print("   %-50s calls %6s avg %9.2f us min %9.2f max %9.2f" % (n.replace("void kmpc::", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
rm -rf "$out/trace"
