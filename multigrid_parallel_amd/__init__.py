"""multigrid_parallel_amd -- MI355X-native 3D geometric multigrid V-cycle behind the C call surface of
knram06/multigrid_parallel's mg_3d.h.

The product is the C-ABI shared library ``lib/libmg3d.so`` (hand-written HIP for gfx950 + C host side,
sources in ``csrc/``, ABI in ``include/mg3d.h``, drop-in headers ``include/mg_3d.h`` / ``postprocess.h``).
This Python package is only the thin ctypes mirror used by the tests and by bench.py.
"""
from .binding import Solver, Solver32, DistSolver, DistSolver32, EsParams, Mg3dError, lib, lib_path  # noqa: F401
