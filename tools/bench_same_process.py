#!/usr/bin/env python
"""The bench line and the rocprofv3 kernel trace OF THE SAME PROCESS, side by side.

  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o bench -- \
        python $REPO/bench.py --steps K --warmup W --no-cpu-baseline --no-secondary --no-end-to-end > <dir>/bench.log
  python tools/bench_same_process.py <dir> profiles/<tag>_bench_same_process.json

bench.py launches, in this order: W warm-up + K timed steps on the library's scattered planes, then (XDEM_BENCH_AB, default on)
W + K on torch.empty planes (a one-launch spot check of the timed planes in between).  A step = one `terrain_strip_kernel` dispatch (raster interior) + one `terrain_tile_kernel`
dispatch (frame of edge tiles).  This tool takes the kernel trace, keeps the dispatches of those two kernels in start order,
cuts them into the four groups and compares the mean duration of the TIMED dispatches with the `kernel_ms` /
`kernel_ms_caller_planes` of the JSON line the very same process printed (HIP events on the launch stream).  Every dispatch
duration is kept in the output, not just the mean: the question is whether the profile reproduces the line."""
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
line = None
for l in open(os.path.join(src, "bench.log")):
    if l.startswith("{"):
        line = json.loads(l)
assert line is not None, "no JSON line in bench.log"
K, W = line["steps"], line["warmup"]
traces = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
assert traces, "no kernel trace"
rows = []
with open(traces[0]) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"]
        if "terrain_strip_kernel" in name or "terrain_tile_kernel" in name:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "strip" if "terrain_strip_kernel" in name else "tile", name))
rows.sort()
strips = [(s, e) for s, e, k, _ in rows if k == "strip"]
tiles = [(s, e) for s, e, k, _ in rows if k == "tile"]
names = sorted({n[:140] for _, _, _, n in rows})
roof = line["roofline"]
ab = "caller_planes" in roof or "other_planes" in roof
# lines since the allocator's calibration (round 6, last session): 3 probe launches on each placement come first, and the timed planes are
# whichever placement the calibration kept -- the group names say which
planes = roof.get("planes") or {}
main = planes.get("backing") or "scattered"
other = "scattered" if main == "torch" else "torch"
cal = planes.get("calibration_ms")
per = int(planes.get("calibration_launches_per_candidate") or 3)
calib = ([(f"calibration_{i}_{k}", per) for i, (k, _) in enumerate(cal)] if isinstance(cal, list) else
         [("calibration_scattered", 3), ("calibration_torch", 3)]) if cal else []
# (bench.py checks a 64-row crop of the timed planes with one more small launch right behind the first timed group: one strip + one tile dispatch)
spot = [("spot_check", 1)] if roof.get("output_spot_check") else []
groups = calib + [(main + "_warmup", W), (main + "_timed", K)] + spot + ([(other + "_warmup", W), (other + "_timed", K)] if ab else [])
assert len(strips) == len(tiles) == sum(n for _, n in groups), (len(strips), len(tiles), groups)
out = {"bench_line": {"steps": K, "warmup": W, "ms_per_step": line["ms_per_step"], "kernel_ms": line["roofline"]["kernel_ms"],
                      "kernel_ms_min": line["roofline"]["kernel_ms_min"], "kernel_ms_max": line["roofline"]["kernel_ms_max"],
                      "frac": line["roofline"]["frac"], "planes": planes or {"backing": "scattered"},
                      "kernel_ms_other_planes": (roof.get("other_planes") or roof.get("caller_planes") or {}).get("kernel_ms"),
                      "kernel_ms_caller_planes": line["roofline"].get("kernel_ms_caller_planes"),
                      "frac_caller_planes": line["roofline"].get("frac_caller_planes"), "frac_scattered_planes": roof.get("frac_scattered_planes")},
       "kernels": names, "groups": {}}
i = 0
for gname, n in groups:
    s_us = [(e - s) / 1e3 for s, e in strips[i:i + n]]
    t_us = [(e - s) / 1e3 for s, e in tiles[i:i + n]]
    # a step under the HIP events = from the start of the first kernel to the end of the second (they are queued back to back)
    span = [(max(strips[j][1], tiles[j][1]) - min(strips[j][0], tiles[j][0])) / 1e3 for j in range(i, i + n)]
    out["groups"][gname] = {"strip_kernel_us": [round(x, 1) for x in s_us], "tile_kernel_us": [round(x, 1) for x in t_us],
                            "step_span_us": [round(x, 1) for x in span],
                            "strip_mean_ms": round(sum(s_us) / n / 1e3, 4), "tile_mean_ms": round(sum(t_us) / n / 1e3, 4),
                            "step_span_mean_ms": round(sum(span) / n / 1e3, 4)}
    i += n
px = line["roofline"]["pixels_per_launch"]
cmp_ = {}
for gname, key in ((main + "_timed", "kernel_ms"), (other + "_timed", "kernel_ms_other_planes")):
    if gname in out["groups"] and out["bench_line"].get(key):
        g = out["groups"][gname]
        ksum = g["strip_mean_ms"] + g["tile_mean_ms"]
        cmp_[gname] = {"rocprof_strip_plus_tile_ms": round(ksum, 4), "rocprof_step_span_ms": g["step_span_mean_ms"],
                       "bench_" + key: out["bench_line"][key],
                       "rocprof_kernels_over_bench": round(ksum / out["bench_line"][key], 4),
                       "rocprof_span_over_bench": round(g["step_span_mean_ms"] / out["bench_line"][key], 4),
                       "frac_from_rocprof_kernels": round(48.0 * px / (ksum * 1e-3) / 1e9 / 8000.0, 4)}
out["comparison"] = cmp_
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({"bench_line": out["bench_line"], "comparison": cmp_}, indent=1))
