"""Per-kernel medians of rocprofv3 --pmc passes (measurement tool): duration, effective clock, SQ counter ratios.
  python tools/pmc_by_kernel.py <dir holding one sub-directory per rocprofv3 pass> [name filter]
Kernels are keyed by their (shortened) full name, so template variants stay apart."""
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "terrain_"


def short(n):
    n = n.replace("xd::", "")
    n = re.sub(r"\(.*$", "", n)
    return n.replace("void ", "").strip()


res = {}
for d in sorted(glob.glob(out + "/*")):
    if not os.path.isdir(d):
        continue
    dur = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if flt in row.get("Kernel_Name", ""):
                dur[row["Dispatch_Id"]] = (short(row["Kernel_Name"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    per = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if flt not in row.get("Kernel_Name", ""):
                continue
            k = (short(row["Kernel_Name"]), row["Counter_Name"])
            per.setdefault(k, {}).setdefault(row["Dispatch_Id"], 0.0)
            per[k][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for (kn, c), dd in per.items():
        v = sorted(dd.values())
        res.setdefault(kn, {})[c] = v[len(v) // 2]
    byk = {}
    for did, (kn, ns) in dur.items():
        byk.setdefault(kn, []).append(ns / 1e6)
    for kn, v in byk.items():
        v.sort()
        res.setdefault(kn, {}).setdefault("_ms_" + os.path.basename(d), v[len(v) // 2])
for kn, r in res.items():
    ms = [v for k, v in r.items() if k.startswith("_ms_")]
    if "GRBM_GUI_ACTIVE" in r and ms:
        msg = r.get("_ms_grbm", ms[0])
        r["clock_GHz"] = round(r["GRBM_GUI_ACTIVE"] / 8 / (msg * 1e6), 3)
    if "SQ_WAVE_CYCLES" in r:
        wc = r["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in r:
                r[c + "/WAVE_CYCLES"] = round(r[c] / wc, 4)
    if "SQ_INSTS_VALU" in r and "SQ_ACTIVE_INST_VALU" in r:
        r["quad_cycles_per_valu_inst"] = round(r["SQ_ACTIVE_INST_VALU"] / r["SQ_INSTS_VALU"], 3)
json.dump(res, open(out + "/by_kernel.json", "w"), indent=1)
print(json.dumps(res, indent=1))
