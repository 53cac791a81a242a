#!/usr/bin/env python3
"""Coarse direct solve in isolation: back-to-back launches (warm caches) against launches that each follow a
pass over a large array (cold caches, the situation inside a V-cycle).  Usage: lu_bench.py [coarse_pts]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_U

c = int(sys.argv[1]) if len(sys.argv) > 1 else 9
rng = np.random.default_rng(1)
for faces in ("random faces (full system)", "zero faces (reduced system, as in a V-cycle)"):
  with M.Solver(c, 6, 2) as s:      # levels c .. 257^3 (c = 9): the top level serves as the cache flusher
    s.setup_test_problem()
    rhs = rng.uniform(-1, 1, (c, c, c))
    if faces.startswith("zero"):
        inner = rhs[1:-1, 1:-1, 1:-1].copy()
        rhs[:] = 0.0
        rhs[1:-1, 1:-1, 1:-1] = inner
    s.upload(MG3D_D, 0, rhs.reshape(-1))
    for _ in range(5):
        s.coarse_solve()
    s.sync()
    t0 = time.perf_counter()
    for _ in range(200):
        s.coarse_solve()
    s.sync()
    warm = (time.perf_counter() - t0) / 200
    top = s.num_levels - 1
    def flush():
        s.smooth(top, 0, 2)
    for _ in range(3):
        flush()
    s.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        flush()
    s.sync()
    tf = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(50):
        flush(); s.coarse_solve()
    s.sync()
    cold = (time.perf_counter() - t0) / 50 - tf
  print(f"coarse solve {c}^3, {faces}: back-to-back {warm * 1e6:.1f} us per launch, after a 257^3 sweep {cold * 1e6:.1f} us (sweep alone {tf * 1e6:.1f} us)")
