#!/bin/bash
# SQ / GRBM counters of selected terrain-kernel variants (run through gpurun).  One --pmc pass per counter set, <= 8 SQ counters.
set -x
TAG=${1:-r02b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $OUT/counters_list.txt 2>&1
for V in L2/store0/rows32 F64/store0/rows32 L2/store1/rows16; do
  N=$(echo $V | tr '/' '_')
  CMD="python $GRAFT_REPO_ROOT/tools/variant_bench.py --size 40000 --reps 3 --rounds 1 --only $V"
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/sqa_$N -o v -- $CMD > $OUT/sqa_$N.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/sqb_$N -o v -- $CMD > $OUT/sqb_$N.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm_$N -o v -- $CMD > $OUT/grbm_$N.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, sys, json
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/" + (sys.argv[1] if len(sys.argv) > 1 else "r02b")
res = {}
for d in sorted(glob.glob(out + "/*_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = {}
        for row in csv.DictReader(open(f)):
            if "terrain_tile_kernel" not in row.get("Kernel_Name", ""): continue
            per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for c, dd in per.items():
            v = sorted(dd.values())
            res.setdefault(os.path.basename(d), {})[c] = v[len(v) // 2]
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        ds = []
        for row in csv.DictReader(open(f)):
            if "terrain_tile_kernel" in row.get("Kernel_Name", ""):
                ds.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
        if ds:
            ds.sort(); res.setdefault(os.path.basename(d), {})["_median_ms"] = ds[len(ds) // 2]
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
