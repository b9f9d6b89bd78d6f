"""ISA ledger of the headline streaming terrain kernel by ABLATION: the kernel is compiled for the full 11-attribute set and for the
set minus one group of planes at a time (compile-time masks: whatever only the dropped planes need folds away); the difference of
the per-row instruction counts -- and of their issue cycles by the measured table of tools/ubench.hip (profiles/r03_ubench.txt) --
is what that group costs.  What remains when only one cheap plane is left is the stencil + the row's fixed work.
  python tools/isa_ledger.py [--fit 2] [--dir 0]        (hipcc on PATH; no GPU needed)"""
import argparse, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BITS = {"slope": 1, "aspect": 2, "hillshade": 4, "profile": 16, "tangential": 32, "planform": 64, "flowline": 128, "max": 256, "min": 512, "tpi": 1024, "tri": 2048}
FULL = 4087
# cycles per wave instruction and SIMD at the nominal clock (profiles/r03_ubench.txt)
COST = [(r"v_rsq_f64|v_rcp_f64|v_sqrt_f64", 17.2), (r"v_rsq_f32|v_sqrt_f32|v_rcp_f32", 8.5), (r"v_fma_f64", 5.74), (r"v_mul_f64", 5.31),
        (r"v_fmac_f64", 4.68), (r"v_add_f64|v_max_f64|v_min_f64", 4.86), (r"v_cmp_\w+_f64", 4.65), (r"v_mov_b64", 4.48), (r"v_cvt_", 4.6),
        (r"v_pk_", 4.9), (r"v_cndmask|v_cmp_|v_med3|v_max_|v_min_|v_bfi|v_and_or|v_mad_u32_u24|v_bitop3|v_lshl_add|v_perm", 4.7),
        (r"v_", 2.9)]


def compile_mask(mask, fit, dir_, tmp):
    out = os.path.join(tmp, f"strip_{mask}.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", f"-DXD_MASK={mask}u", f"-DXD_FIT={fit}",
                           f"-DXD_DIR={dir_}", "-I" + os.path.join(ROOT, "xdem_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "strip_isa.hip"), "-o", out], stderr=subprocess.DEVNULL)
    src = open(out).read()
    m = re.search(r"\n(_Z\w+):[^\n]*\n(.*?)\n\t\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel", src, re.S)
    body, desc = m.group(2), m.group(3)
    vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
    scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1))
    # the row march = the blocks of the depth-1 loop with the most instructions; hot path only: blocks that hold a plane store or feed one
    loops = {}
    for bm in re.finditer(r"\n(\.LBB\w+):([^\n]*)\n(.*?)(?=\n\.LBB\w+:|\Z)", body, re.S):
        hm = re.search(r"Header=(BB\w+) Depth=1", bm.group(2)) or re.search(r"=>This Loop Header: Depth=1", bm.group(2))
        if hm:
            loops.setdefault(hm.group(1) if hm.lastindex else bm.group(1)[2:], []).append(bm.group(3))
    main = max(loops.values(), key=lambda bl: sum(len(b) for b in bl))
    # (the march is unrolled five rows deep; a row's hot path is ONE label-delimited stretch -- row partials, the "window complete"
    #  branch, the whole tail with its plane stores -- and the five longest stretches of the loop are those; what the loop holds
    #  besides them are the cold paths: ring refills, float64 hillshade, the reference-order sums)
    hot = sorted(main, key=len)[-5:]
    ins = [l.strip().split()[0] for b in hot for l in b.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    n_rows = 5.0
    cnt = {"valu": 0, "f64": 0, "cvt": 0, "trans": 0, "pk": 0, "salu": 0, "lds": 0, "vmem": 0}
    cyc = 0.0
    for i in ins:
        if i.startswith("v_"):
            cnt["valu"] += 1
            cyc += next(c for pat, c in COST if re.match(pat, i))
            if re.search(r"_(rsq|sqrt|rcp)_", i): cnt["trans"] += 1
            elif i.startswith("v_cvt"): cnt["cvt"] += 1
            elif i.startswith("v_pk_"): cnt["pk"] += 1
            elif "f64" in i or "b64" in i: cnt["f64"] += 1
        elif i.startswith("s_"): cnt["salu"] += 1
        elif i.startswith("ds_"): cnt["lds"] += 1
        elif i.startswith(("global_", "buffer_")): cnt["vmem"] += 1
    return {k: v / n_rows for k, v in cnt.items()} | {"cycles": cyc / n_rows, "vgpr": vgpr, "scratch": scratch}


# the launches bench.py times (secondary.terrain_sets + the headline): name -> (fit, directional, mask)
BENCH_SETS = {"headline: full 11, Florinsky, geometric curvatures": (2, 0, FULL),
              "slope": (2, 0, 1), "slope+aspect Horn (DEM.slope() / aspect() defaults of the reference's examples)": (0, 0, 3),
              "slope+aspect Florinsky": (2, 0, 3), "hillshade": (2, 0, 4), "slope+aspect+hillshade Florinsky": (2, 0, 7),
              "full 11, directional curvatures": (2, 1, FULL), "full 11, ZevenbergThorne fit (3x3)": (1, 0, FULL),
              "full 11, ZevenbergThorne fit, directional curvatures": (1, 1, FULL)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fit", type=int, default=2)
    ap.add_argument("--dir", type=int, default=0)
    ap.add_argument("--json", default=None, help="write the per-row issue cycles of bench.py's launches to this file (bench.py: issue_frac) and stop")
    a = ap.parse_args()
    if a.json:
        import json

        out = {"_what": "vector-instruction issue cycles per output row of 64 pixels (hot path of the streaming kernel's unrolled loop) at the per-"
                        "instruction costs of tools/ubench.hip (profiles/r03_ubench.txt, cycles per wave instruction and SIMD measured at the NOMINAL "
                        "2.4 GHz: divide by 2.4e9 for seconds), by tools/isa_ledger.py --json from the compiled kernels", "sets": {}}
        with tempfile.TemporaryDirectory() as tmp:
            for name, (fit, dr, mask) in BENCH_SETS.items():
                r = compile_mask(mask, fit, dr, tmp)
                out["sets"][name] = {"cycles_per_row": round(r["cycles"], 1), "vector_instructions_per_row": round(r["valu"], 1),
                                     "float64_class_per_row": round(r["f64"] + r["cvt"], 1), "vgprs": r["vgpr"]}
                print(name, out["sets"][name], flush=True)
        json.dump(out, open(a.json, "w"), indent=1)
        return
    groups = [("slope", ["slope"]), ("aspect", ["aspect"]), ("hillshade", ["hillshade"]), ("profile curvature", ["profile"]),
              ("tangential + planform curvature", ["tangential", "planform"]), ("flowline curvature", ["flowline"]), ("max + min curvature", ["max", "min"]),
              ("TPI", ["tpi"]), ("TRI", ["tri"]), ("all six curvatures", ["profile", "tangential", "planform", "flowline", "max", "min"]),
              ("slope + aspect + hillshade", ["slope", "aspect", "hillshade"]), ("TPI + TRI", ["tpi", "tri"])]
    with tempfile.TemporaryDirectory() as tmp:
        full = compile_mask(FULL, a.fit, a.dir, tmp)
        print(f"terrain_strip_kernel<fit {a.fit}, Spec<{FULL}, dir {a.dir}, lean tail>, 128 rows, 4 waves/SIMD>: per output row of 64 pixels (hot path of the unrolled five-row loop)")
        print(f"  FULL SET   vector {full['valu']:6.1f} (float64-class {full['f64']:.1f}, conversions {full['cvt']:.1f}, packed f32 {full['pk']:.1f}, transcendental {full['trans']:.1f})"
              f"  scalar {full['salu']:.1f}  LDS {full['lds']:.1f}  VMEM {full['vmem']:.1f}   ~{full['cycles']:.0f} issue cycles   VGPRs {full['vgpr']}  scratch {full['scratch']} B")
        print(f"  {'what the group costs (full set minus the set without it)':58s} {'vector':>7s} {'f64':>6s} {'cvt':>5s} {'cycles':>7s} {'share':>6s}")
        for name, planes in groups:
            m = FULL
            for p in planes:
                m &= ~BITS[p]
            r = compile_mask(m, a.fit, a.dir, tmp)
            print(f"  {name:58s} {full['valu'] - r['valu']:7.1f} {full['f64'] - r['f64']:6.1f} {full['cvt'] - r['cvt']:5.1f} {full['cycles'] - r['cycles']:7.0f} "
                  f"{(full['cycles'] - r['cycles']) / full['cycles']:6.1%}   (without: {r['vgpr']} VGPRs)")
        base = compile_mask(BITS["planform"], a.fit, a.dir, tmp)
        print(f"  {'stencil + one plane (planform curvature alone)':58s} {base['valu']:7.1f} {base['f64']:6.1f} {base['cvt']:5.1f} {base['cycles']:7.0f} {base['cycles'] / full['cycles']:6.1%}")
        s1 = compile_mask(BITS["slope"], a.fit, a.dir, tmp)
        print(f"  {'first-derivative stencil + slope alone':58s} {s1['valu']:7.1f} {s1['f64']:6.1f} {s1['cvt']:5.1f} {s1['cycles']:7.0f} {s1['cycles'] / full['cycles']:6.1%}")


if __name__ == "__main__":
    main()
