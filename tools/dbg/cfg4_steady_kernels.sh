#!/bin/bash
# Dev tool (GPU box): per-kernel durations of the LAST 150 steps of a cfg4 run (the settled shared-model step) from a rocprofv3 kernel trace
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/gpurun_out/cfg4_steady; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$out/trace" -o bench --output-format csv -- python3 "$root/bench.py" --config cfg4 --cpu-seconds 0 --no-extras --no-probe --steps 200 --warmup 20 > "$out/bench.json" 2> "$out/trace.log"
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]; n = n[:n.index("(")] if "(" in n else n
    n = n.replace("void kmpc::", "").replace("kmpc::", "")
    per.setdefault(n, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
tot = 0.0
for n, v in per.items():
    if len(v) < 300: continue
    v.sort(); last = [d for _, d in v[-150:]]
    m = sum(last) / len(last) / 1e3
    tot += m
    print("%-60s last 150 launches: mean %.2f us  min %.2f  max %.2f   (all %d launches: %.2f)" % (n[:60], m, min(last) / 1e3, max(last) / 1e3, len(v), sum(d for _, d in v) / len(v) / 1e3))
print("sum of the means %.2f us" % tot)
PY
rm -rf "$out/trace"
