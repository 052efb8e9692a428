#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: last `n` dispatches of kernels matching a name."""
import csv, sys, collections
path, pat, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = list(csv.DictReader(open(path)))
disp = collections.OrderedDict()
for r in rows:
    if pat in r["Kernel_Name"]:
        disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        disp[r["Dispatch_Id"]]["_grid"] = r.get("Grid_Size", "")
        disp[r["Dispatch_Id"]]["_lds"] = r.get("LDS_Block_Size", "")
        disp[r["Dispatch_Id"]]["_vgpr"] = r.get("VGPR_Count", "")
ids = list(disp)[-n:]
keys = sorted({k for i in ids for k in disp[i]})
for k in keys:
    vals = [disp[i].get(k) for i in ids]
    print("%-28s %s" % (k, vals))
