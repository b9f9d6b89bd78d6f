"""A/B timing of the headline terrain launch between builds of the library, in one process (measurement tool).

  python tools/ab_libs.py [--size 40000] [--reps 6] [--rounds 3] name=path/to/lib.so [name=path ...]

Every library gets its own context; launches are interleaved over `rounds` so that the box's clock drift hits all builds
alike.  Times are the libraries' own HIP events (xdemhip_last_kernel_ms).  Also checks that the builds agree bit for bit
on a crop (or reports the largest ulp distance per plane)."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FULL_MASK = 4087  # slope .. TRI: the 11-attribute set of the headline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=40000)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--mask", type=int, default=FULL_MASK, help="attribute mask (1 slope, 2 aspect, 4 hillshade, ...; default the 11 of the headline)")
    ap.add_argument("--fit", type=int, default=2, help="0 Horn, 1 Zevenbergen-Thorne, 2 Florinsky")
    ap.add_argument("--curv", type=int, default=0, help="0 geometric, 1 directional curvatures")
    ap.add_argument("--planes", default="torch", help="torch (torch.empty = ordinary hipMalloc) | scattered (the product library's 8 MiB-piece backing) | both")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    import numpy as np
    import torch

    from xdem_amd.synth import fbm_torch

    n = a.size
    dem = fbm_torch(n, n, "cuda", seed=42)
    outs = {}
    if a.planes in ("torch", "both"):
        outs["torch"] = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    built = []
    for spec in a.libs:
        name, path = spec.split("=", 1)
        L = ctypes.CDLL(os.path.join(ROOT, path))
        ctx = ctypes.c_void_p()
        L.xdemhip_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        assert L.xdemhip_create(0, ctypes.byref(ctx)) == 0
        L.xdemhip_last_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        L.xdemhip_synchronize.argtypes = [ctypes.c_void_p]
        L.xdemhip_terrain.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
            ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
            ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
        built.append((name, L, ctx))
    if a.planes in ("scattered", "both"):
        # the scattered backing through the C-ABI of the LAST library given (never through xdem_amd._lib: that loader opens the product
        # library RTLD_GLOBAL, after which every build loaded here would resolve its internal symbols to the product's -- an A/B of a
        # library against itself, as session r06c found out)
        _, L, ctx = built[-1]
        L.xdemhip_device_alloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]
        ptr, gotc = ctypes.c_void_p(), ctypes.c_int()
        rc = L.xdemhip_device_alloc(ctx, 11 * n * n * 4, 8, ctypes.byref(ptr), ctypes.byref(gotc))
        assert rc == 0, rc

        class _Arr:   # torch view of the library's allocation (kept alive for the process: a measurement tool)
            __cuda_array_interface__ = {"shape": (11, n, n), "typestr": "<f4", "data": (ptr.value, False), "version": 2}

        outs["scattered"] = torch.as_tensor(_Arr(), device="cuda")
    out = next(iter(outs.values()))
    assert getattr(sys.modules.get("xdem_amd._lib"), "_lib", None) is None, "the product library is loaded RTLD_GLOBAL: this A/B would compare it with itself"
    npl = bin(a.mask).count("1")
    planes = (ctypes.c_void_p * 11)(*[out[i].data_ptr() for i in range(11)])

    def launch(L, ctx):
        rc = L.xdemhip_terrain(ctx, ctypes.c_void_p(dem.data_ptr()), 0, n, n, n, 0, 0, 10.0, a.fit, a.curv, a.mask, 0, 3, 45.0, 315.0,
                               1.0, 1, 0, planes, 1)
        assert rc == 0, rc
        L.xdemhip_synchronize(ctx)
        ms = ctypes.c_float()
        L.xdemhip_last_kernel_ms(ctx, ctypes.byref(ms))
        return float(ms.value)

    for pname, pl in outs.items():
        out = pl
        planes = (ctypes.c_void_p * 11)(*[out[i].data_ptr() for i in range(11)])
        times = {name: [] for name, _, _ in built}
        for name, L, ctx in built:
            launch(L, ctx)
        for _ in range(a.rounds):
            for name, L, ctx in built:
                for _ in range(a.reps):
                    times[name].append(launch(L, ctx))
        for name, t in times.items():
            t = sorted(t)
            print(f"[{pname:9s} planes] {name:12s} min {t[0]:7.3f}  median {t[len(t) // 2]:7.3f}  max {t[-1]:7.3f} ms", flush=True)
    # agreement on a crop (bit patterns; NaN == NaN)
    crop = slice(0, min(n, 4096))
    ref = None
    for name, L, ctx in built:
        launch(L, ctx)
        gf = out[:npl, crop, crop].cpu().numpy()
        got = gf.view(np.int32).astype(np.int64)
        got = np.where(got < 0, -(got & 0x7FFFFFFF), got)
        got[np.isnan(gf)] = 1 << 40   # every NaN is the same NaN (the sign of a NaN is not an output)
        if ref is None:
            ref = got
            continue
        d = np.abs(got - ref)
        print(f"{name} vs {built[0][0]}: max ulp distance per plane {d.reshape(npl, -1).max(axis=1).tolist()}", flush=True)


if __name__ == "__main__":
    main()
