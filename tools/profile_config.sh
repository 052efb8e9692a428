#!/bin/bash
# Dev tool (GPU box): rocprofv3 passes over one bench configuration; summaries land in gpurun_out/prof_<tag>/.
#   tools/profile_config.sh <config> <tag> [extra bench.py arguments]
# passes: kernel trace + stats | FETCH_SIZE | WRITE_SIZE | issue / wait counters | MFMA counters  (counters in passes of their
# own, kernel trace only -- MI355X_MICROARCH.md; the CPU baseline is never run under the profiler)
cfg=${1:-cfg2}; tag=${2:-r2}; shift 2
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && export KMPC_DEBUG=1
args="--config $cfg --cpu-seconds 0 --no-extras --no-probe $*"
rocprofv3 --kernel-trace --stats -d "$out/trace" -o bench --output-format csv -- python3 "$root/bench.py" $args > "$out/bench_trace.json" 2> "$out/trace.log"
rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o bench --output-format csv -- python3 "$root/bench.py" $args > "$out/bench_fetch.json" 2> "$out/fetch.log"
rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o bench --output-format csv -- python3 "$root/bench.py" $args > "$out/bench_write.json" 2> "$out/write.log"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d "$out/sq" -o bench --output-format csv -- python3 "$root/bench.py" $args > "$out/bench_sq.json" 2> "$out/sq.log"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d "$out/mfma" -o bench --output-format csv -- python3 "$root/bench.py" $args > "$out/bench_mfma.json" 2> "$out/mfma.log"
cd "$root"
python3 - "$out" <<'PY' > "$out/summary.txt"
import csv, glob, sys, collections, json
out = sys.argv[1]
def short(n):
    n = n.replace("void kmpc::", "").replace("kmpc::", "")
    return n[:n.index("(")] if "(" in n else n
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats)")
    for row in list(csv.DictReader(open(f)))[:10]:
        print("%-64s calls %6s  avg %12.1f ns  %6s %%" % (short(row["Name"])[:64], row["Calls"], float(row["AverageNs"]), row["Percentage"]))
    import shutil; shutil.copy(f, out + "/kernel_stats.csv")
for which in ("fetch", "write", "sq", "mfma"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % which, recursive=True):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            d = per.setdefault(k, collections.OrderedDict())
            d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        print("== %s pass: mean counter value per launch (number of launches)" % which)
        for k, d in per.items():
            if not any(x in k for x in ("rollout", "step_kernel", "step_qp_kernel", "lift", "gram_kernel", "gram_reduce", "shared_model", "shared_fast")):
                continue
            print("  %s" % k[:90])
            for c, v in d.items():
                tail = v[-max(1, len(v) // 3):]  # the last third of the launches: past set-up and spin-up
                print("      %-30s %16.1f   (%d launches, last %d averaged)" % (c, sum(tail) / len(tail), len(v), len(tail)))
for name in ("trace", "fetch", "write", "sq", "mfma"):
    try:
        line = [l for l in open(out + "/bench_%s.json" % name) if l.startswith("{")][-1]
        d = json.loads(line)
        print("== bench line under the %s pass: value %.4g %s, %.2f us/step, roofline frac %.4f, kernel %.4f ms" % (name, d["value"], d["unit"], d["ms_per_step"] * 1e3, d["roofline"]["frac"], d["roofline"]["avg_kernel_ms"]))
    except Exception as e:
        print("== bench line under the %s pass: not available (%s)" % (name, e))
PY
# the raw traces are large (gpurun merges at most 64 MiB back): keep the summaries unless KEEP_RAW is set
if [ -z "$KEEP_RAW" ]; then rm -rf "$out/trace" "$out/fetch" "$out/write" "$out/sq" "$out/mfma"; fi
cat "$out/summary.txt"
