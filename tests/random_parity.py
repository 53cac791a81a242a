#!/usr/bin/env python3
"""Randomised parity sweep (run on the GPU box): many (coarse points, levels, sweeps) combinations, a few cycles each,
every grid value of the finest level compared with the oracle bit for bit.  Usage: python tests/random_parity.py [count seed]  (a checker script: it lives under tests/ because it calls the oracle)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_U, MG3D_D
import _oracle as O

count, seed = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (60, 1)
rng = np.random.default_rng(seed)
O.lib().orc_set_threads(8)
bad = 0
t0 = time.time()
for n in range(count):
    while True:
        c, L, nu = int(rng.integers(3, 14)), int(rng.integers(2, 7)), int(rng.integers(1, 4))
        nu = int(os.environ.get("NU", nu))  # NU=2 NMIN=66 CYCLES=4 MG3D_CARRY_MIN=66: the carried-cycle schedule
        N = (c - 1) * (1 << (L - 1)) + 1
        if max(9, int(os.environ.get("NMIN", "9"))) <= N <= 161 and c ** 3 <= 1400:
            break
    cycles = int(os.environ.get("CYCLES", "2"))
    want_norms, want_u, _, _ = O.run_problem(c, L, nu, cycles)
    with M.Solver(c, L, nu) as s:
        s.setup_test_problem()
        norms = s.vcycles(cycles)
        u = s.download(MG3D_U, L - 1)
    ok = np.array_equal(u, want_u) and np.allclose(norms, want_norms, rtol=1e-9, atol=0)
    bad += not ok
    print(f"{n:3d}  c={c:2d} L={L} nu={nu}  N={N:3d}  {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{count - bad} of {count} configurations bit-identical, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
