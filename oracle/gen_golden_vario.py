"""Golden vectors for the variogram path recorded from the reference's own host-preparation functions (the pairwise
part lives in scikit-gstat, absent here): the equidistant sampling-parameter table (spatialstats.py:1104-1183) and the
circular / ring masks (880-937).  Container-only; called by oracle/gen_golden.py."""
import os

import numpy as np


def main(ref, out_dir: str) -> None:
    ss = ref.spatialstats
    rec = {}
    rows = []
    for subsample in (10, 100, 1000, 10000, 10**7):
        for shape in ((100, 100), (1000, 1000), (539, 985), (20000, 20000)):
            for gsd in (1.0, 20.0):
                extent = (0.0, (shape[0] - 1) * gsd, 0.0, (shape[1] - 1) * gsd)
                try:
                    runs, samples, ratio = ss._choose_cdist_equidistant_sampling_parameters(
                        subsample=subsample, extent=extent, shape=shape)
                except ValueError:
                    runs, samples, ratio = -1, -1, np.nan
                rows.append((subsample, shape[0], shape[1], gsd, runs, samples, ratio))
    rec["T7|params"] = np.array(rows, dtype=np.float64)
    rec["T7|circ"] = ss._create_circular_mask((30, 41), center=(12, 20), radius=9.5)
    rec["T7|ring"] = ss._create_ring_mask((30, 41), center=(12, 20), in_radius=4.0, out_radius=11.0)
    np.savez_compressed(os.path.join(out_dir, "vario_golden.npz"), **rec)
    print("variogram fixtures written:", len(rows), "parameter rows")
