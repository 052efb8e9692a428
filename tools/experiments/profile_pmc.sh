#!/bin/bash
# Dev tool (GPU box): two --pmc passes (issue / wait counters) over the default bench command; per-launch values of
# the roll-out kernel are summarised into gpurun_out/prof_<tag>/pmc_summary.txt
tag=${1:-r1f}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES -d "$out/pmc3" -o bench --output-format csv -- python3 "$root/bench.py" --cpu-seconds 0 --spin-seconds 0.2 > "$out/bench_pmc3.json" 2> "$out/pmc3.log"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$out/pmc4" -o bench --output-format csv -- python3 "$root/bench.py" --cpu-seconds 0 --spin-seconds 0.2 > "$out/bench_pmc4.json" 2> "$out/pmc4.log"
python3 - "$out" > "$out/pmc_summary.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for which in ("pmc3", "pmc4"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % which, recursive=True):
        disp = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if "rollout_kernel" in r["Kernel_Name"]:
                disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        ids = list(disp)
        # the 200-step launches: those whose SQ_WAVES / GRBM value is the maximum kind; take the last three
        last = ids[-3:]
        keys = sorted({k for i in last for k in disp[i]})
        print("== %s: roll-out kernel, last three launches of the process (200 steps x 4096 trajectories each)" % which)
        for k in keys:
            print("%-24s %s" % (k, [disp[i].get(k) for i in last]))
PY
cat "$out/pmc_summary.txt"
