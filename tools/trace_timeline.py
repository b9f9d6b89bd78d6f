"""Print the tail of a rocprofv3 --kernel-trace CSV as a timeline (measurement tool): python tools/trace_timeline.py <csv> [n_last]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -70:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = t0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)[:90]
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f} us  {name}")
    prev_end = e
