import sys, os, time, faulthandler
faulthandler.dump_traceback_later(45, repeat=True)
print('start', flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from xdem_amd import _lib, terrain
from xdem_amd.synth import fbm_numpy
FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
print('imports done', flush=True)
ctx = _lib.default_context()
print('ctx done', flush=True)
dem = fbm_numpy((1400, 2600), seed=3)
print('dem done', flush=True)
d = torch.from_numpy(dem).cuda()
print('cuda done', flush=True)
res = {}
for mode in [int(a) for a in sys.argv[1:]] or [0, 3, 2, 1]:
    ctx.set_option("terrain_stream", mode)
    print("mode", mode, "launch", flush=True)
    out = torch.full((11, 1400, 2600), -7.0, dtype=torch.float32, device="cuda")
    t = time.time()
    terrain.terrain_attributes_device(d, FULL, resolution=10.0, out=out)
    torch.cuda.synchronize()
    print("mode", mode, "done", round(time.time() - t, 3), flush=True)
    res[mode] = out.cpu().numpy()
if 0 in res:
    for m, v in res.items():
        if m == 0: continue
        same = (v == res[0]) | (np.isnan(v) & np.isnan(res[0]))
        untouched = v == -7.0
        print("mode", m, "equal to tiles:", float(same.mean()), "untouched:", float(untouched.mean()), "wrong:", float((~same & ~untouched).mean()))
        if m == 1 and not same.all():
            bad = np.argwhere(~same)
            print("first bad", bad[:5], "rows with bad", np.unique(bad[:, 1])[:20], "cols", np.unique(bad[:, 2])[:20])
