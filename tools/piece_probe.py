"""Headline terrain launch on plane backings of different piece sizes / on a scattered input raster (measurement tool).
  python tools/piece_probe.py [--size 40000] [--reps 8] [--pieces 2,8,32,128] [--lib xdem_amd/csrc/libxdemhip.so]
Backings: torch.empty; the library's scattered range with XDEMHIP_SCATTER_PIECE_MB = each of --pieces (8 = the shipped form);
then the shipped planes once more with the INPUT raster in a scattered range too.  Two rounds, so that drift shows."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=40000)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--pieces", default="2,8,32,128")
    ap.add_argument("--lib", default="xdem_amd/csrc/libxdemhip.so")
    ap.add_argument("--fit", type=int, default=2)
    a = ap.parse_args()
    import torch
    from xdem_amd.synth import fbm_torch

    n = a.size
    L = ctypes.CDLL(os.path.join(ROOT, a.lib))
    ctx = ctypes.c_void_p()
    L.xdemhip_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    assert L.xdemhip_create(0, ctypes.byref(ctx)) == 0
    L.xdemhip_last_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    L.xdemhip_synchronize.argtypes = [ctypes.c_void_p]
    L.xdemhip_device_alloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]
    L.xdemhip_device_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.xdemhip_terrain.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
        ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
        ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
        ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    dem = fbm_torch(n, n, "cuda", seed=42)
    torch.cuda.synchronize()

    def scattered(nbytes, piece_mb):
        os.environ["XDEMHIP_SCATTER_PIECE_MB"] = str(piece_mb)
        p, g = ctypes.c_void_p(), ctypes.c_int()
        rc = L.xdemhip_device_alloc(ctx, nbytes, 8, ctypes.byref(p), ctypes.byref(g))
        assert rc == 0, rc
        return p

    def run(label, dem_ptr, base_ptr):
        planes = (ctypes.c_void_p * 11)(*[base_ptr + i * n * n * 4 for i in range(11)])
        ts = []
        for k in range(a.reps + 2):
            rc = L.xdemhip_terrain(ctx, ctypes.c_void_p(dem_ptr), 0, n, n, n, 0, 0, 10.0, a.fit, 0, 4087, 0, 3, 45.0, 315.0, 1.0, 1, 0, planes, 1)
            assert rc == 0, rc
            L.xdemhip_synchronize(ctx)
            ms = ctypes.c_float()
            L.xdemhip_last_kernel_ms(ctx, ctypes.byref(ms))
            if k >= 2:
                ts.append(float(ms.value))
        ts.sort()
        print(f"{label:44s} min {ts[0]:7.3f}  median {ts[len(ts) // 2]:7.3f}  max {ts[-1]:7.3f} ms", flush=True)

    pieces = [int(x) for x in a.pieces.split(",")]
    for rnd in range(2):
        out = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
        run(f"[round {rnd}] torch.empty planes", dem.data_ptr(), out.data_ptr())
        del out
        torch.cuda.empty_cache()
        for pm in pieces:
            p = scattered(11 * n * n * 4, pm)
            run(f"[round {rnd}] scattered planes, {pm} MiB pieces", dem.data_ptr(), p.value)
            if pm == 8 and rnd == 0:
                d2 = scattered(n * n * 4, 8)
                t = torch.as_tensor(type("A", (), {"__cuda_array_interface__": {"shape": (n, n), "typestr": "<f4", "data": (d2.value, False), "version": 2}})(), device="cuda")
                t.copy_(dem)
                torch.cuda.synchronize()
                run(f"[round {rnd}]   ... + scattered INPUT raster (8 MiB)", d2.value, p.value)
                del t
                L.xdemhip_device_free(ctx, d2)
            L.xdemhip_device_free(ctx, p)


if __name__ == "__main__":
    main()
