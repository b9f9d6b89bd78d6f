"""xdem_amd -- MI355X-native drop-in for xDEM's three dense-array hot paths.

* :mod:`xdem_amd.terrain`      fused stencil engine behind ``get_terrain_attribute`` / ``slope`` / ... (+ ``terrain.surfit`` /
  ``terrain.window`` / ``terrain.freq``: the reference's engine-boundary functions under their own names)
* :mod:`xdem_amd.coreg`        Nuth-Kaab inner loop behind ``NuthKaab.fit``
* :mod:`xdem_amd.spatialstats` pairwise lag binning behind ``sample_empirical_variogram``
* :mod:`xdem_amd.dist`         row-block / pair-set sharding over the GPUs of a node (torch.distributed / RCCL)

All compute goes through ``libxdemhip.so`` (hand-written HIP for gfx950, C-ABI in ``include/xdemhip.h``);
there is no CPU fallback.
"""
from . import _lib  # noqa: F401
from . import terrain  # noqa: F401
from . import coreg, spatialstats  # noqa: F401
from . import freq, surfit, window  # noqa: F401  (the reference's engine-boundary modules: xdem.terrain.surfit / .window / .freq)
from .dem import DEM  # noqa: F401
from .terrain import get_terrain_attribute  # noqa: F401

terrain.surfit, terrain.window, terrain.freq = surfit, window, freq   # reachable as upstream spells them: xdem_amd.terrain.surfit._get_surface_attributes

__version__ = "0.1.0"
