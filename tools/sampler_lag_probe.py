"""How fast do the amdgpu sysfs metrics follow the load?  (measurement tool)  Runs the headline terrain launch for `secs` seconds with
bench.GpuSampler at 20 ms and prints the time series (t since the first launch, sclk, power, busy) every 100 ms, then 2 s of idle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from xdem_amd import _lib, terrain
from xdem_amd.synth import fbm_torch

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
dem = fbm_torch(n, n, dev, seed=42)
out = terrain.alloc_planes(11, n, n, torch.float32, ctx, dev)
kw = dict(resolution=10.0, surface_fit="Florinsky", curv_method="geometric", ctx=ctx)
terrain.terrain_attributes_device(dem, bench.FULL, out=out, **kw)
torch.cuda.synchronize()
time.sleep(2.0)
s = bench.GpuSampler(0, period=0.02).start()
t0 = time.perf_counter()
k = 0
while time.perf_counter() - t0 < secs:
    for _ in range(8):
        terrain.terrain_attributes_device(dem, bench.FULL, out=out, **kw)
    torch.cuda.synchronize()
    k += 8
t1 = time.perf_counter()
time.sleep(2.0)
s.stop()
print(f"{k} launches in {t1 - t0:.2f} s = {(t1 - t0) / k * 1e3:.3f} ms each; files: {sorted(s.files)}")
last = -1
for t, v in s.samples:
    b = int((t - t0) * 10)
    if b != last:
        last = b
        print(f"t={t - t0:6.2f}s  sclk_hz={v.get('sclk_hz', 0) / 1e6:7.0f} MHz  sclk_dpm={v.get('sclk_dpm')}  power={v.get('power_in_uW', 0) / 1e6:6.0f} W  busy={v.get('busy_pct')}  T={v.get('temp2_mC', 0) / 1e3:.0f}/{v.get('temp3_mC', 0) / 1e3:.0f}")
