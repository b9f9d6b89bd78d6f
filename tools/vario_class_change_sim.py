"""How often does a wave of the variogram pair kernels change lag class, per ring and per order of the B points?  (CPU simulation on the
C5 geometry of SURVEY 8d, reading B: 9091-point centre disk x ten rings, 50 edges geomspace(sqrt 2, maxlag); measurement tool)

The run-length kernels (csrc/variogram.hip) keep a lane's class between consecutive B points; a wave-pair (64 A points of one wave against
one B point) costs 8 vector instructions when every lane stays in its class and ~18 more when any lane leaves it.  The A points of a
wave are 64 consecutive points of the Morton-ordered centre sample.  Orders of the B points compared:
  morton     Morton order of the lattice coordinates over the whole ring union (what PairSet uploads: the shipped form)
  ring+morton ring by ring, Morton inside a ring
  radial     ring by ring, by distance from the run's centre
  slices     ring by ring, thin annular slices (width = a quarter of the narrowest lag class the ring meets), by angle inside a slice
  polar      ring by ring, Morton order of (log r, angle) quantised to 10 bits each
  python tools/vario_class_change_sim.py [runs=2] [samples=9091]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xdem_amd import spatialstats as ss

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 9091
size, rings = 20000, 10
maxdist = math.sqrt(2.0) * (size - 1)
ratio = samples / (math.pi * maxdist**2 / math.sqrt(2.0) ** (2 * rings))
rng = np.random.default_rng(45)
centres = []
blocks = ss.equidistant_blocks_from_raster(None, 1.0, runs, samples, ratio, rng, values_of=lambda idx: np.zeros(idx.size, np.float32), shape=(size, size), centres_out=centres)
edges = np.geomspace(math.sqrt(2.0), maxdist, 50)
thr2 = edges**2                      # class k = [e_{k-1}, e_k): searchsorted(right) on d^2 against e^2 (vario_edge 0)
r0, radii = ss._equidistant_radii(samples, ratio, 1.0, maxdist, math.sqrt(2.0))
radii = np.asarray(radii)


def morton(x, y):
    def spread(v):
        v = v.astype(np.uint64) & np.uint64(0xFFFF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x33333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x55555555)
        return v
    return np.argsort(spread(x) | (spread(y) << np.uint64(1)), kind="stable")


def orders(xb, yb, cx, cy):
    r = np.hypot(xb - cx, yb - cy)
    th = np.arctan2(yb - cy, xb - cx)
    ring = np.clip(np.searchsorted(radii, r, side="right") - 1, 0, len(radii) - 2)
    out = {"morton": morton(xb, yb)}
    def by_ring(key_fn):
        o = []
        for k in range(len(radii) - 1):
            idx = np.flatnonzero(ring == k)
            if idx.size:
                o.append(idx[key_fn(idx, k)])
        return np.concatenate(o)
    out["ring+morton"] = by_ring(lambda idx, k: morton(xb[idx], yb[idx]))
    out["radial"] = by_ring(lambda idx, k: np.argsort(r[idx], kind="stable"))
    def slices(idx, k):
        lo, hi = radii[k], radii[k + 1]
        # narrowest class width among the classes whose edges lie in [lo - r0, hi + r0]
        e = edges[(edges >= max(lo - r0, 0)) & (edges <= hi + r0)]
        w = np.min(np.diff(e)) / 4 if e.size > 1 else (hi - lo)
        s = np.floor((r[idx] - lo) / max(w, 1.0)).astype(np.int64)
        return np.lexsort((th[idx], s))
    out["slices"] = by_ring(slices)
    def polar(idx, k):
        lr = np.log(np.maximum(r[idx], 1.0))
        q1 = ((lr - lr.min()) / max(np.ptp(lr), 1e-9) * 1023).astype(np.int64)
        q2 = ((th[idx] + math.pi) / (2 * math.pi) * 1023).astype(np.int64)
        return morton(q1, q2)
    out["polar"] = by_ring(polar)
    return out, ring


def wave_changes(xa, ya, xb, yb):
    """For every wave (64 consecutive A points) the boolean over consecutive B points: some lane's class differs from its class at the
    previous B point.  Returns (changes per B index summed over waves, waves)."""
    nb = xb.size
    tot = np.zeros(nb, dtype=np.int64)
    lanes = np.zeros(nb, dtype=np.int64)
    nw = 0
    for w0 in range(0, xa.size - 63, 64):
        ax, ay = xa[w0:w0 + 64, None], ya[w0:w0 + 64, None]
        ch = np.zeros(nb, dtype=bool)
        for j0 in range(0, nb, 8192):
            j1 = min(nb, j0 + 8192 + 1)
            d2 = (ax - xb[None, j0:j1]) ** 2 + (ay - yb[None, j0:j1]) ** 2
            cls = np.searchsorted(thr2, d2, side="right")
            diff = cls[:, 1:] != cls[:, :-1]
            ch[j0 + 1:j1] |= diff.any(axis=0)
            lanes[j0 + 1:j1] += diff.sum(axis=0)
        tot += ch
        nw += 1
    return tot, lanes, nw


t0 = time.time()
acc = {}
for bi, blk in enumerate(blocks):
    xa, ya, _, xb, yb, _ = blk
    oa = morton(xa, ya)
    xa, ya = xa[oa], ya[oa]
    cx, cy = centres[bi]
    ords, ring = orders(xb, yb, float(cx), float(cy))
    for name, o in ords.items():
        tot, lanes, nw = wave_changes(xa, ya, xb[o], yb[o])
        rg = ring[o]
        for k in range(len(radii) - 1):
            m = rg == k
            a = acc.setdefault((name, k), [0, 0, 0])
            a[0] += int(tot[m].sum()); a[1] += int(m.sum()) * nw; a[2] += int(lanes[m].sum())
    print(f"run {bi}: A {xa.size} points, B {xb.size} points, {time.time() - t0:.0f} s", flush=True)

names = ["morton", "ring+morton", "radial", "slices", "polar"]
print(f"\nC5 reading B geometry, {len(blocks)} runs, {samples} points per sample, disk radius r0 = {r0:.0f} px, 50 classes (ratio {edges[1] / edges[0]:.3f} per class)")
print("fraction of WAVE-pairs in which some lane changes its lag class (lane-pairs that change in brackets), by ring of the B point:")
print(f"{'ring':>4s} {'radii (px)':>16s} {'share of pairs':>14s} " + " ".join(f"{n:>18s}" for n in names))
tot_pairs = sum(acc[("morton", k)][1] for k in range(len(radii) - 1))
overall = {n: [0, 0, 0] for n in names}
for k in range(len(radii) - 1):
    p = acc[("morton", k)][1]
    row = []
    for n in names:
        c, q, l = acc[(n, k)]
        overall[n][0] += c; overall[n][1] += q; overall[n][2] += l
        row.append(f"{c / max(q, 1):8.3f} ({l / max(q * 64, 1):6.4f})")
    print(f"{k:4d} {radii[k]:7.0f}-{radii[k + 1]:7.0f} {p / tot_pairs:14.3f} " + " ".join(f"{r:>18s}" for r in row))
print(f"{'all':>4s} {'':16s} {1.0:14.3f} " + " ".join(f"{overall[n][0] / overall[n][1]:8.3f} ({overall[n][2] / (overall[n][1] * 64):6.4f})".rjust(18) for n in names))
print("\nvector instructions per wave-pair of the counting pass at 8 (in class) + 18 (class change) and the speed-up over the shipped order:")
base = 8 + 18 * overall["morton"][0] / overall["morton"][1]
for n in names:
    v = 8 + 18 * overall[n][0] / overall[n][1]
    print(f"  {n:12s} {v:5.2f} per pair   x{base / v:.3f}")
