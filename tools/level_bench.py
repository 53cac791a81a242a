#!/usr/bin/env python3
"""Per-level timing of the launches a V-cycle makes below the finest level (four passes, residual + restriction,
two passes): python tools/level_bench.py [c L].  No tuning knob is set, so MG3D_LIB_PATH=<other build> gives an A/B."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_U

c, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (9, 7)
reps = int(os.environ.get("REPS", "200"))
s = M.Solver(c, L, 2)
rng = np.random.default_rng(12345)


def timeit(fn):
    for _ in range(5):
        fn()
    s.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        s.sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


for lev in range(L - 1, 0, -1):
    n = (c - 1) * 2 ** lev + 1
    s.upload(MG3D_U, lev, rng.uniform(-1, 1, n ** 3))
    s.upload(MG3D_D, lev, rng.uniform(-1, 1, n ** 3))
    if lev == L - 1:
        globals()["reps"] = max(5, reps // 10)
    else:
        globals()["reps"] = int(os.environ.get("REPS", "200"))
    t4 = timeit(lambda: s.smooth(lev, 0, 2))
    t2 = timeit(lambda: s.smooth(lev, 0, 1))
    tr = timeit(lambda: s.smooth_restrict(lev, 0))
    print(f"level {lev} ({n}^3): four passes {t4 * 1e3:7.4f} ms   two passes {t2 * 1e3:7.4f} ms   residual+restriction {tr * 1e3:7.4f} ms", flush=True)
