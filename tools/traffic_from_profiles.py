"""Dev tool: profiles/traffic.json entries from the PMC passes of tools/profile_config.sh.
    python tools/traffic_from_profiles.py <config> <gpurun_out/prof_dir> <profiles/summary name>
HBM bytes per trajectory-step of the dominant kernel = (FETCH_SIZE x c + WRITE_SIZE) KiB per launch / (trajectories x steps
per launch).  c = 2 for the register-state roll-outs (round 3: the state is read as 16-byte-per-lane coalesced loads, for which
gfx950's FETCH_SIZE reports exactly half of the bytes, MI355X_MICROARCH.md "HBM"); c = 1.593 for the kernels that still read
8 bytes per lane (round 1's own-pattern calibration, profiles/r1_traffic.json).  The counters are the mean over the last third
of the kernel's launches (summary.txt)."""
import json, os, re, sys
cfg, d, summary_name = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def bench_line(name):
    return json.loads([l for l in open(os.path.join(d, "bench_%s.json" % name)) if l.startswith("{")][-1])
bf, bw = bench_line("fetch"), bench_line("write")
fused = "rollout_kernel" in bf["roofline"]["kernel"]
whole = bf["roofline"]["kernel"].startswith("whole shared-model step")  # cfg4: every launch of the step counts
dom = "rollout_kernel" if fused else ("" if whole else ("step_qp_kernel" if "step_qp_kernel" in bf["roofline"]["kernel"] else "step_kernel"))
vals, cur, sec = {}, None, None
for l in open(os.path.join(d, "summary.txt")):
    if l.startswith("== "):
        sec = l.split()[1]
    elif l.startswith("  ") and not l.startswith("      "):
        cur = l.strip()
    elif l.startswith("      ") and cur and cur.startswith(dom):
        p = l.split()
        vals[(sec, p[0])] = vals.get((sec, p[0]), 0.0) + float(p[1])  # (summed over the step's kernels when `whole`)
B = int(re.search(r"(\d+) trajectories per GPU x", bf["config"]["workload"]).group(1))
L, N = [int(x) for x in re.search(r"N=(\d+), (\d+)-dim", bf["metric"]).groups()][::-1]
units = B * bf["roofline"]["steps_per_launch"]
fetch_raw = vals[("fetch", "FETCH_SIZE")] * 1024.0 / units
write = vals[("write", "WRITE_SIZE")] * 1024.0 / (B * bw["roofline"]["steps_per_launch"])
wide = fused and (L + 2 <= 32) and "y = C x" in bf["config"]["workload"] or (fused and cfg in ("cfg2", "cfg3", "cfg3-L20"))
cal = 2.0 if wide else 1.593
entry = {"L": L, "N": N, "B": B, "bytes_per_trajectory_step": fetch_raw * cal + write,
         "fetch_bytes_per_trajectory_step_raw": fetch_raw, "write_bytes_per_trajectory_step": write, "fetch_calibration": cal,
         "source": "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x %s (%s) + WRITE, %s, "
                   "last third of its launches" % (summary_name, cal, "16-byte-per-lane coalesced reads: gfx950 reports half the bytes, MI355X_MICROARCH.md" if wide
                                                   else "own-pattern calibration of round 1: 8-byte-per-lane coalesced reads report 0.628 of the bytes",
                                                   "summed over all kernels of the step" if whole else "dominant kernel " + dom)}
tp = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tp))
key = "%s:f64:%s" % (cfg, "fused" if fused else "steps")
print(key, "was", tj.get(key, {}).get("bytes_per_trajectory_step"), "now", entry["bytes_per_trajectory_step"], "algorithmic", bf["roofline"]["algorithmic_bytes_per_trajectory_step"])
tj[key] = entry
json.dump(tj, open(tp, "w"), indent=1)
