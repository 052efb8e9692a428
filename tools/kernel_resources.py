#!/usr/bin/env python3
"""Register / spill / LDS figures of every kernel in a gfx950 code object (from its AMDGPU metadata note).

    llvm-objdump --offloading csrc/rollout_kernel.o      # writes <obj>.0.hipv4-amdgcn-amd-amdhsa--gfx950
    python tools/kernel_resources.py <code object> [name filter]
"""
import re
import subprocess
import sys

out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", sys.argv[1]], capture_output=True, text=True).stdout
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = {}
rows = []
for ln in out.splitlines():
    m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", ln)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "agpr_count" and cur:
        rows.append(cur)
        cur = {}
    cur[k] = v
if cur:
    rows.append(cur)
for r in rows:
    if "name" not in r or flt not in r["name"]:
        continue
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void kmpc::", "")
    print("%-58s vgpr %3s spill %3s | sgpr %3s spill %3s | scratch %4s B | lds %s" % (
        name[:58], r.get("vgpr_count"), r.get("vgpr_spill_count"), r.get("sgpr_count"), r.get("sgpr_spill_count"),
        r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size")))
