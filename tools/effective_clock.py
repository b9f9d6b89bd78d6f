#!/usr/bin/env python
"""Effective shader clock of a kernel from a `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` run:
GRBM_GUI_ACTIVE (busy cycles of the graphics block) / kernel duration, per dispatch (MI355X_MICROARCH.md, "DVFS give-back").
  python tools/effective_clock.py <rocprof output dir> [kernel-name-substring]"""
import csv
import glob
import json
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "terrain_tile_kernel"
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if pat in r["Kernel_Name"]}
acc = {}
for r in csv.DictReader(open(cc)):
    if pat in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        acc.setdefault(r["Dispatch_Id"], []).append(float(r["Counter_Value"]))
rows = []
for k, v in acc.items():
    if k in dur:
        rows.append({"dispatch": k, "instances": len(v), "gui_active_max": max(v), "gui_active_sum": sum(v), "duration_ns": dur[k],
                     "clock_GHz_max": max(v) / dur[k], "clock_GHz_mean_instance": sum(v) / len(v) / dur[k]})
print(json.dumps({"kernel": pat, "dispatches": rows}, indent=1))
