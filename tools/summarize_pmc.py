#!/usr/bin/env python
"""Summarise the rocprofv3 counter CSVs of tools/profile_bench.sh into one JSON (per-dispatch means of every counter for
the kernels whose name contains a pattern) and copy the --stats kernel summary next to it.

  python tools/summarize_pmc.py gpurun_out/prof_<tag> profiles/<tag>_bench_terrain [kernel-name-substring] [pixels-per-launch]
"""
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else "terrain_tile_kernel"
pixels = int(sys.argv[4]) if len(sys.argv) > 4 else 40000 * 40000
out = {}
for f in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pat not in row.get("Kernel_Name", ""):
                continue
            key = (row["Counter_Name"], row["Dispatch_Id"])
            per[key] = per.get(key, 0.0) + float(row["Counter_Value"])  # (summed over XCDs / dimensions of one dispatch)
    names = sorted({k[0] for k in per})
    for n in names:
        vals = [v for (c, _), v in per.items() if c == n]
        out[n] = {"dispatches": len(vals), "mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals)}
out["_kernel_pattern"] = pat
out["_pixels_per_launch"] = pixels
json.dump(out, open(dst + "_pmc.json", "w"), indent=1)
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], dst + "_kernel_stats.csv")
print(json.dumps({k: (v["mean"] if isinstance(v, dict) else v) for k, v in out.items()}, indent=1))
