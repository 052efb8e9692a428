#!/bin/bash
# Dev tool (GPU box): rocprofv3 passes over the default bench command; summaries land in gpurun_out/prof_<tag>/.
#   kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md).
tag=${1:-r1e}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -o bench --output-format csv -- python3 "$root/bench.py" --cpu-seconds 0 > "$out/bench_trace.json" 2> "$out/trace.log"
rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o bench --output-format csv -- python3 "$root/bench.py" --cpu-seconds 0 > "$out/bench_fetch.json" 2> "$out/fetch.log"
rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o bench --output-format csv -- python3 "$root/bench.py" --cpu-seconds 0 > "$out/bench_write.json" 2> "$out/write.log"
find "$out" -name "*.csv" | head -20
python3 - "$out" <<'PY'
import csv, glob, sys, os, json
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    for row in list(csv.DictReader(open(f)))[:8]:
        print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
res = {}
for which in ("fetch", "write"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % which, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "rollout_kernel" in r["Kernel_Name"]]
        vals = [float(r["Counter_Value"]) for r in rows]
        print("==", which, "rollout_kernel dispatches:", len(vals), "last values", vals[-4:])
        res[which] = vals
json.dump(res, open(out + "/pmc_rollout.json", "w"))
PY
