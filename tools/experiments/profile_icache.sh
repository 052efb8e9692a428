#!/bin/bash
# Dev tool (GPU box): instruction-cache counters of the roll-out kernel.  usage: profile_icache.sh <tag> <python script> [args]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQC_TC_INST_REQ -d "$out/pmcI" -o run --output-format csv -- python3 "$@" > "$out/run.log" 2>&1
python3 - "$out" > "$out/icache_summary.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/pmcI/**/*counter_collection.csv", recursive=True):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "rollout_kernel" in r["Kernel_Name"]:
            disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = list(disp)[-2:]
    for k in sorted({k for i in ids for k in disp[i]}):
        print("%-28s %s" % (k, [disp[i].get(k) for i in ids]))
PY
cat "$out/icache_summary.txt"
