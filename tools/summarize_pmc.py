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
# several comma-separated patterns = the kernels that together make one step (round 3: streaming strips + frame tiles): the
# per-dispatch means of each are ADDED ("mean" = bytes / instructions per step); "per_kernel" keeps them apart
pats = (sys.argv[3] if len(sys.argv) > 3 else "terrain_tile_kernel").split(",")
pat = pats[0]
pixels = int(sys.argv[4]) if len(sys.argv) > 4 else 40000 * 40000
out = {}
for f in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            hit = [p_ for p_ in pats if p_ in row.get("Kernel_Name", "")]
            if not hit:
                continue
            key = (row["Counter_Name"], hit[0], row["Dispatch_Id"])
            per[key] = per.get(key, 0.0) + float(row["Counter_Value"])  # (summed over XCDs / dimensions of one dispatch)
    for n in sorted({k[0] for k in per}):
        entry = {"mean": 0.0, "per_kernel": {}}
        for p_ in pats:
            vals = [v for (c, kp, _), v in per.items() if c == n and kp == p_]
            if vals:
                sv = sorted(vals)
                entry["per_kernel"][p_] = {"dispatches": len(vals), "mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals),
                                           "median": sv[len(sv) // 2] if len(sv) % 2 else 0.5 * (sv[len(sv) // 2 - 1] + sv[len(sv) // 2])}
                entry["mean"] += sum(vals) / len(vals)
        entry["dispatches"] = max(v["dispatches"] for v in entry["per_kernel"].values())
        out[n] = entry
out["_kernel_pattern"] = ",".join(pats)
out["_pixels_per_launch"] = pixels
json.dump(out, open(dst + "_pmc.json", "w"), indent=1)
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], dst + "_kernel_stats.csv")
# rocprofv3's --stats averages every dispatch of a kernel, the cold first one included (the first launch of a 70 GB write set
# runs 10-15 % long), so its mean can exceed the bench's ms_per_step of the timed steps.  From the kernel trace of the same run:
# per-kernel MEDIAN and the mean WITHOUT each kernel's first dispatch -- the numbers to hold against bench.py's kernel_ms.
traces = glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True)
if traces:
    durs = {}
    with open(traces[0]) as fh:
        for row in csv.DictReader(fh):
            try:
                durs.setdefault(row["Kernel_Name"], []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
            except (KeyError, ValueError):
                pass
    med = {}
    for name, v in durs.items():
        v.sort()
        d = [x[1] for x in v]
        warm = d[1:] if len(d) > 1 else d
        sw = sorted(warm)
        med[name[:160]] = {"calls": len(d), "first_us": d[0] / 1e3, "median_us": (sw[len(sw) // 2] if len(sw) % 2 else 0.5 * (sw[len(sw) // 2 - 1] + sw[len(sw) // 2])) / 1e3,
                           "mean_without_first_us": sum(warm) / len(warm) / 1e3, "min_us": min(d) / 1e3, "max_us": max(d) / 1e3,
                           "total_ms": sum(d) / 1e6}
    top = dict(sorted(med.items(), key=lambda kv: -kv[1]["total_ms"])[:60])
    json.dump(top, open(dst + "_kernel_medians.json", "w"), indent=1)
print(json.dumps({k: (v["mean"] if isinstance(v, dict) else v) for k, v in out.items()}, indent=1))
