"""TCC / EA / TLB counters of the headline terrain launch on physically contiguous planes against the library's scattered
backing (round 4).  Runs on the GPU box:  python tools/backing_pmc.py gpurun_out/r04_backing
Separate `rocprofv3 --pmc` passes (never combined with a trace domain other than --kernel-trace), each under `timeout`; the
counter names are taken from what this rocprofv3 lists, so an unknown name costs nothing.  Writes <out>/backing_pmc.json."""
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04_backing")
os.makedirs(out_dir, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")

avail = ""
for cmd in (["rocprofv3", "--list-avail"], ["rocprofv3", "-L"], ["rocprofv3-avail", "list"]):
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd="/tmp", env=env)
        if len(p.stdout) + len(p.stderr) > len(avail):
            avail = p.stdout + p.stderr
    except Exception as e:  # noqa: BLE001
        print("list failed", cmd, e)
open(os.path.join(out_dir, "avail.txt"), "w").write(avail)
names = set(re.findall(r"\b((?:TCC|TCP|TCA|GRBM|SQ|TA|TD|GL2C|CPC|CPF|SPI|UTCL2|ATC|MC|EA)_[A-Za-z0-9_\[\]]+)", avail)) | set(re.findall(r"\b(FETCH_SIZE|WRITE_SIZE|WRITE_REQ_32B|L2CacheHit|MemUnitStalled|WriteUnitStalled|MemWrites32B|VALUBusy|MemUnitBusy)\b", avail))
print(len(names), "counter names listed")
open(os.path.join(out_dir, "tcc_names.txt"), "w").write("\n".join(sorted(n for n in names if n.startswith(("TCC", "TCP", "TCA", "UTCL2", "ATC")))))

WISH = [
    ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_WRREQ_IO_CREDIT_STALL_sum"],
    ["TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum", "TCC_TOO_MANY_EA_WRREQS_STALL_sum", "TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_WRREQ_64B_sum"],
    ["TCC_TAG_STALL_sum", "TCC_BUSY_sum", "TCC_CYCLE_sum", "TCC_REQ_sum"],
    ["TCC_EA0_WRREQ_DRAM_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_LEVEL_sum"],
    ["TCC_EA0_WRREQ", "TCC_EA0_WRREQ_STALL"],          # raw: one value per TCC instance -> channel balance
    ["TCC_WRITE_sum", "TCC_WRITEBACK_sum", "TCC_HIT_sum", "TCC_MISS_sum"],
    ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_PENDING_STALL_CYCLES_sum"],
    ["TCP_TCC_WRITE_REQ_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_NC_WRITE_REQ_sum", "TCP_TCC_UC_WRITE_REQ_sum"],
    ["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_INST_ANY", "SQ_INSTS_VMEM_WR", "SQ_WAIT_ANY"],
    ["WRITE_SIZE"], ["FETCH_SIZE"],
]
groups = []
for g in WISH:
    have = [c for c in g if c in names] if names else g
    if have:
        groups.append(have)
print("groups:", groups)

result = {"_what": "per-dispatch counters of terrain_strip_kernel at 40000^2 (11 planes), warm dispatches only (the first of each run dropped); "
                   "contiguous = hipExtMallocWithFlags(hipDeviceMallocContiguous) planes, scattered = 8 MiB pieces in pseudo-random order",
          "backings": {}}
for backing in ("contiguous", "scattered"):
    res = {}
    for gi, g in enumerate(groups):
        d = os.path.join(out_dir, f"{backing}_g{gi}")
        log = os.path.join(out_dir, f"{backing}_g{gi}.log")
        cmd = ["timeout", "230", "rocprofv3", "--pmc", *g, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "x", "--",
               sys.executable, os.path.join(ROOT, "tools", "backing_run.py"), backing, "5"]
        with open(log, "w") as fh:
            rc = subprocess.run(cmd, stdout=fh, stderr=subprocess.STDOUT, cwd="/tmp", env=env).returncode
        tail = open(log).read()[-300:]
        m = re.search(r"median ([0-9.]+) ms", tail)
        res.setdefault("_kernel_ms_under_pmc", {})["+".join(g)[:60]] = float(m.group(1)) if m else None
        if rc != 0:
            print("pass failed", backing, g, rc, tail[-200:])
        per = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if "terrain_strip_kernel" not in row.get("Kernel_Name", ""):
                        continue
                    per.setdefault((row["Counter_Name"], int(row["Dispatch_Id"])), []).append(float(row["Counter_Value"]))
        for c in sorted({k[0] for k in per}):
            disp = sorted(k[1] for k in per if k[0] == c)
            warm = disp[1:] if len(disp) > 1 else disp
            tot = sorted(sum(per[(c, d_)]) for d_ in warm)
            entry = {"dispatches": len(warm), "median_per_dispatch": tot[len(tot) // 2], "min": tot[0], "max": tot[-1]}
            inst = per[(c, warm[len(warm) // 2])]
            if len(inst) > 1:   # one row per instance / dimension: keep the spread (channel balance)
                entry["instances"] = len(inst)
                entry["instance_min"], entry["instance_max"] = min(inst), max(inst)
                entry["instance_values"] = inst if len(inst) <= 160 else None
            res[c] = entry
        for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True):   # keep gpurun_out small
            if os.path.getsize(f) > (2 << 20):
                os.remove(f)
    result["backings"][backing] = res
    json.dump(result, open(os.path.join(out_dir, "backing_pmc.json"), "w"), indent=1)
for c in sorted(set(result["backings"]["contiguous"]) & set(result["backings"]["scattered"])):
    if c.startswith("_"):
        continue
    a, b = result["backings"]["contiguous"][c]["median_per_dispatch"], result["backings"]["scattered"][c]["median_per_dispatch"]
    print(f"{c:44s} contiguous {a:16.1f} scattered {b:16.1f} ratio {a / b if b else float('nan'):8.3f}")
print(json.dumps({k: v["_kernel_ms_under_pmc"] for k, v in result["backings"].items()}, indent=1))
