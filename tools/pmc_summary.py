"""Median per-dispatch counter values of the terrain kernel from rocprofv3 --pmc output directories (measurement tool)."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "terrain_tile_kernel"
res = {}
for d in sorted(glob.glob(out + "/*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = {}
        for row in csv.DictReader(open(f)):
            if pat not in row.get("Kernel_Name", ""):
                continue
            per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for c, dd in per.items():
            v = sorted(dd.values())
            res.setdefault(os.path.basename(d), {})[c] = v[len(v) // 2]
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        ds = []
        for row in csv.DictReader(open(f)):
            if pat in row.get("Kernel_Name", ""):
                ds.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
        if ds:
            ds.sort()
            res.setdefault(os.path.basename(d), {})["_median_ms"] = ds[len(ds) // 2]
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
