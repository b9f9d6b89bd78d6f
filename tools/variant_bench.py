"""A/B timing of terrain-kernel variants on one GPU, all in one process (measurement tool).

  python tools/variant_bench.py [--size 40000] [--reps 5] [--rounds 2]

Loads the measurement builds (make -C xdem_amd/csrc variants) side by side through ctypes and times the headline launch
(Florinsky, 11 attributes, float32) for every combination of tail level / float64 tail, tile height and store form,
interleaved over `rounds` so that clock drift of the box hits all variants alike.  Times are the library's own HIP events."""
import argparse
import ctypes
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=40000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default="", help="comma-separated substrings: run only the variants whose name contains one")
    a = ap.parse_args()
    import torch

    from xdem_amd.synth import fbm_torch

    n = a.size
    dem = fbm_torch(n, n, "cuda", seed=42)
    out = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    csrc = os.path.join(ROOT, "xdem_amd", "csrc")
    libs = {"R01": "libxdemhip_r01.so", "MIX": "libxdemhip_exp2.so", "NOSTORE": "libxdemhip_expnostore.so", "RAWRSQ": "libxdemhip_exprawrsq.so", "NOLOAD": "libxdemhip_expnoload.so", "PLAINSTORE": "libxdemhip_expplain.so", "PLAINLOAD": "libxdemhip_expsa.so", "NTSC0SC1": "libxdemhip_expsb.so", "SC0SC1": "libxdemhip_expsc.so"}
    variants = []
    for tag, fn in libs.items():
        path = os.path.join(csrc, fn)
        if not os.path.exists(path):
            print("missing", path)
            continue
        L = ctypes.CDLL(path)
        ctx = ctypes.c_void_p()
        L.xdemhip_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        assert L.xdemhip_create(0, ctypes.byref(ctx)) == 0
        L.xdemhip_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        L.xdemhip_last_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        L.xdemhip_synchronize.argtypes = [ctypes.c_void_p]
        L.xdemhip_terrain.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
            ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
            ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
        if tag == "R01":  # the round-1 library: in-session baseline (boxes of the pool clock differently)
            if not a.only or any(o in "R01" for o in a.only.split(",")):
                variants.append(("R01", L, ctx, None, None, None, 0))
            continue
        combos = (list(itertools.product([0, 1], [0, 1], [16, 24, 32])) + [(0, 0, 132), (0, 0, 232), (0, 0, 332)]) if tag == "MIX" else [(0, 0, 32), (0, 0, 16)]
        for math, store, rows in combos:
            name = f"{tag if math == 0 else tag + '-F64'}/store{store}/rows{rows}"
            if a.only and not any(o in name for o in a.only.split(",")):
                continue
            variants.append((name, L, ctx, math, store, rows, 0))
    planes = (ctypes.c_void_p * 11)(*[out[i].data_ptr() for i in range(11)])
    mask = 0xFFF & ~(1 << 3)
    times = {v[0]: [] for v in variants}

    def run(L, ctx, math, store, rows, pf=0):
        if math is not None:
            L.xdemhip_set_option(ctx, b"terrain_math", math)
            L.xdemhip_set_option(ctx, b"terrain_store", store)
            L.xdemhip_set_option(ctx, b"terrain_rows", rows)
        rc = L.xdemhip_terrain(ctx, dem.data_ptr(), 0, n, n, n, 0, 0, 10.0, 2, 0, mask, 0, 3, 45.0, 315.0, 1.0, 1, 0, planes, 1)
        assert rc == 0, rc
        ms = ctypes.c_float()
        assert L.xdemhip_last_kernel_ms(ctx, ctypes.byref(ms)) == 0
        return ms.value

    sums = {}
    for name, L, ctx, math, store, rows, pf in variants:  # warm-up + checksum of every variant
        run(L, ctx, math, store, rows, pf)
        torch.cuda.synchronize()
        sums[name] = [float(torch.nan_to_num(out[i, ::37, ::41]).double().sum()) for i in range(11)]
    base = sums[variants[0][0]]
    for r in range(a.rounds):
        for name, L, ctx, math, store, rows, pf in variants:
            for _ in range(a.reps):
                times[name].append(run(L, ctx, math, store, rows, pf))
    gb = 48.0 * n * n / 1e9
    print(f"{'variant':24s} {'min ms':>8s} {'median':>8s} {'Gpx/s':>8s} {'TB/s':>6s} {'frac':>6s}  checksum-vs-first")
    res = {}
    for name, *_ in variants:
        t = sorted(times[name])
        mn, med = t[0], t[len(t) // 2]
        same = sum(1 for x, y in zip(sums[name], base) if x == y)
        res[name] = {"min_ms": mn, "median_ms": med, "frac_of_8TBps": gb / med / 8.0, "planes_equal_checksum": same}
        print(f"{name:24s} {mn:8.3f} {med:8.3f} {n * n / med / 1e6:8.1f} {gb / med:6.2f} {gb / med / 8.0:6.3f}  {same}/11")
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"size": n, "results": res}, f, indent=1)


if __name__ == "__main__":
    main()
