"""Dispatch sequence from a rocprofv3 --kernel-trace CSV: the last N dispatches in start order with duration and the idle gap
since the previous dispatch ended.  python tools/trace_sequence.py <dir-or-csv> [N] [name-filter]"""
import csv
import glob
import os
import re
import sys

src = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 80
flt = sys.argv[3] if len(sys.argv) > 3 else ""
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if flt:
    rows = [r for r in rows if flt in r[2]]
rows = rows[-n_last:]
prev_end = None
tot = gap_tot = 0.0
for s, e, name in rows:
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("xd::", "").replace("(anonymous namespace)::", "")[:64]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{short:64s} {((e - s) / 1e3):9.1f} us   gap {gap:7.1f} us")
    tot += (e - s) / 1e3
    gap_tot += max(gap, 0.0)
    prev_end = e
print(f"-- {len(rows)} dispatches, {tot:.1f} us busy, {gap_tot:.1f} us of gaps, {(rows[-1][1] - rows[0][0]) / 1e3:.1f} us first start to last end")
