#!/usr/bin/env python3
"""Micro-benchmark of the fused sweep kernel shapes on one level (default 513^3, seeded uniform(-1,1) fields,
SURVEY 8(d)).  Usage: python tools/sweep_bench.py [N-args c L] ; prints ms and algorithmic GB/s per shape."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_U

c, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (9, 7)
reps = int(os.environ.get("REPS", "5"))
s = M.Solver(c, L, 2)
N = s.N
lev = L - 1
rng = np.random.default_rng(12345)
s.upload(MG3D_U, lev, rng.uniform(-1, 1, N ** 3))
s.upload(MG3D_D, lev, rng.uniform(-1, 1, N ** 3))
n = N ** 3


def timeit(fn, reps=reps):
    fn()
    s.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    s.sync()
    return (time.perf_counter() - t0) / reps


cases = {
    "S4RES": (lambda: s.smooth_residual(lev, 0, 2, True, False), (6 + 3) * n * 8,
              ["6,4,1", "6,4,2", "4,4,2", "4,8,1", "2,8,2"]),
    "S4NORM": (lambda: s.smooth_residual(lev, 1, 2, False, False), (6 + 2) * n * 8,
               ["6,4,1", "6,4,2", "4,4,2", "4,8,1", "2,8,2"]),
    "S4": (lambda: s.smooth(lev, 0, 2), 6 * n * 8, ["8,4,1", "8,4,2", "6,4,2", "4,8,1"]),
    "S2RES": (lambda: s.smooth_residual(lev, 0, 1, True, False), (3 + 3) * n * 8, ["4,8,1", "8,4,2", "6,4,2"]),
    "S2": (lambda: s.smooth(lev, 0, 1), 3 * n * 8, ["6,8,1", "8,4,2", "4,8,1"]),
    "S0RES": (lambda: s.residual(lev, True, False), 3 * n * 8, ["4,8,1", "8,4,2", "4,8,2", "4,8,3", "2,8,4"]),
    "S0RST": (lambda: s.smooth_restrict(lev, 0), 4.125 * n * 8, ["4,8,1", "4,8,2"]),
    "S2RST": (lambda: s.smooth_restrict(lev, 1), 7.125 * n * 8, ["4,8,1"]),
    "S0NORM": (lambda: s.residual(lev, False, False), 2 * n * 8, ["4,8,1", "8,4,2", "4,8,2", "4,8,3", "2,8,4"]),
    "S2NORM": (lambda: s.smooth_residual(lev, 1, 1, False, False), 5 * n * 8, ["4,8,1"]),
}
only = os.environ.get("ONLY")
cfg_override = os.environ.get("CFGS")
for name, (fn, alg, cfgs) in cases.items():
    if only and name not in only.split(","):
        continue
    for cfg in (cfg_override.split(";") if cfg_override else cfgs):
        rj, nw, pf = (int(x) for x in cfg.split(","))
        for key, v in (("sweep_rj", rj), ("sweep_nw", nw), ("sweep_pf", pf)):  # the options API: nothing reads the environment per launch
            s.set_option(key, v)
        for ci in os.environ.get("CIS", "0").split(","):
            s.set_option("sweep_ci", int(ci))
            t = timeit(fn)
            print(f"{name:6s} cfg {cfg:6s} CI {ci:>3s}: {t * 1e3:8.3f} ms   algorithmic {alg / t / 1e9:8.1f} GB/s", flush=True)
if os.environ.get("NO_UNFUSED"):
    sys.exit(0)
os.environ["MG3D_NO_FUSE"] = "1"
s2 = M.Solver(c, L, 2)
s2.upload(MG3D_U, lev, rng.uniform(-1, 1, N ** 3))
s2.upload(MG3D_D, lev, rng.uniform(-1, 1, N ** 3))
s, sold = s2, s
t = timeit(lambda: s.smooth(lev, 0, 2))
print(f"unfused 4 colour passes: {t * 1e3:8.3f} ms   algorithmic {6 * n * 8 / t / 1e9:8.1f} GB/s")
t = timeit(lambda: s.residual(lev, True, False))
print(f"unfused residual+store : {t * 1e3:8.3f} ms   algorithmic {3 * n * 8 / t / 1e9:8.1f} GB/s")
