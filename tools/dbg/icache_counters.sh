#!/bin/bash
# Dev tool (GPU box): instruction-cache counters of the dominant kernel of a bench configuration (last launches).
#   tools/dbg/icache_counters.sh <config> [bench args]
cfg=${1:-cfg2}; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/gpurun_out/icache_$cfg; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d "$out/pmc" -o run --output-format csv -- python3 "$root/bench.py" --config $cfg --cpu-seconds 0 --no-extras --no-probe --steps 20 --warmup 5 "$@" > "$out/run.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "rollout_kernel" in r["Kernel_Name"] or "step_kernel" in r["Kernel_Name"]:
            disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = list(disp)[-3:]
    for k in sorted({k for i in ids for k in disp[i]}):
        print("%-28s %s" % (k, [disp[i].get(k) for i in ids]))
PY
rm -rf "$out/pmc"
