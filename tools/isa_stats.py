"""Per-kernel register / instruction statistics from hipcc -S output (measurement tool).
  hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S xdem_amd/csrc/terrain_ff.hip -o /tmp/t.s
  python tools/isa_stats.py /tmp/t.s [filter]
Counts are static over the whole kernel body (loops and cold paths included); `hot` restricts to the largest basic block
run (heuristic: the longest stretch between labels), which for the specialised terrain kernels is one output row."""
import re
import subprocess
import sys

src = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
# function bodies: "<name>:" ... ".end_amdhsa_kernel"
for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\t\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel(.*?); Occupancy: (\d+)", src, re.S):
    name, body, _, after = m.group(1), m.group(2), m.group(3), m.group(4) + "; Occupancy: " + m.group(5)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem:
        continue
    info = {k: re.search(rf"; {k}: (\d+)", after) for k in ("NumVgprs", "NumSgprs", "Occupancy", "LDSByteSize", "ScratchSize")}
    info = {k: (v.group(1) if v else "?") for k, v in info.items()}
    blocks = re.split(r"\n\.LBB\w+:.*", body)
    def count(text):
        ins = [l.strip().split()[0] for l in text.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = {"valu64": 0, "valu32": 0, "trans": 0, "cvt": 0, "ds": 0, "vmem": 0, "salu": 0, "other": 0}
        for i in ins:
            if i.startswith("v_"):
                if re.search(r"_(rsq|sqrt|rcp|exp|log|sin|cos)_", i):
                    c["trans"] += 1
                elif i.startswith("v_cvt"):
                    c["cvt"] += 1
                elif "f64" in i or "b64" in i or "i64" in i or "u64" in i:
                    c["valu64"] += 1
                else:
                    c["valu32"] += 1
            elif i.startswith("ds_"):
                c["ds"] += 1
            elif i.startswith(("global_", "buffer_", "flat_")):
                c["vmem"] += 1
            elif i.startswith("s_"):
                c["salu"] += 1
            else:
                c["other"] += 1
        return c, len(ins)
    tot, n = count(body)
    # blocks of the outermost loop with the most instructions (the row march): labels carry "in Loop: Header=BBx_y Depth=1"
    loops = {}
    for bm in re.finditer(r"\n(\.LBB\w+):([^\n]*)\n(.*?)(?=\n\.LBB\w+:|\Z)", body, re.S):
        hm = re.search(r"Header=(BB\w+) Depth=1", bm.group(2)) or re.search(r"=>This Loop Header: Depth=1", bm.group(2))
        if hm:
            key = hm.group(1) if hm.lastindex else bm.group(1)[2:]
            loops.setdefault(key, []).append(bm.group(3))
    if loops:
        main = max(loops.values(), key=lambda bl: sum(len(b) for b in bl))
        lc, ln = count("\n".join(main))
        print(f"   main loop ins {ln}: {lc}")
        print(f"   ~VALU cycles per loop trip {4 * lc['valu64'] + 2 * lc['valu32'] + 4 * lc['cvt'] + 8 * lc['trans']}")
    hot = max(blocks, key=lambda b: len(b))
    hc, hn = count(hot)
    cyc = lambda c: 4 * c["valu64"] + 2 * c["valu32"] + 4 * c["cvt"] + 8 * c["trans"]
    print(dem.replace("xd::", "").replace("(TileArgs<float, float>)", ""))
    print(f"   regs {info}  total ins {n}: {tot}  ~VALU cycles total {cyc(tot)} (/5 = {cyc(tot)/5:.0f})")
    print(f"   largest block ins {hn}: {hc}  ~VALU cycles {cyc(hc)}")
