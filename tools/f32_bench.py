#!/usr/bin/env python3
"""BASELINE configs[4] on ONE GPU: 1025^3 (args 9 8 2) Poisson in binary32, damped-Jacobi V(2,2) cycles after an
F-cycle (FMG) start.  Usage: f32_bench.py [coarse levels nu cycles]; prints ms per V-cycle, algorithmic GB/s
(SURVEY 8(d) as totalled there: [3(nu1+nu2)+8] n w + 3 n_c w per level, w = 4) and the F-cycle start's time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_parallel_amd as M

c, L, nu, cycles = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (9, 8, 2, 10)))
N = (c - 1) * (1 << (L - 1)) + 1
alg = 0
for l in range(1, L):
    n, nc = ((c - 1) * (1 << l) + 1) ** 3, ((c - 1) * (1 << (l - 1)) + 1) ** 3
    alg += (3 * 2 * nu + 8) * n * 4 + 3 * nc * 4
alg += (c ** 6 + 2 * c ** 3) * 8
with M.Solver32(c, L, nu) as s:
    s.setup_test_problem(fmg=False)
    s.vcycles(2)
    s.sync()
    t0 = time.perf_counter()
    norms = s.vcycles(cycles)
    s.sync()
    dt = (time.perf_counter() - t0) / cycles
    t0 = time.perf_counter()
    s.setup_test_problem(fmg=True)
    s.sync()
    tf = time.perf_counter() - t0
    after = s.vcycles(3)
print(f"{N}^3 fp32 Jacobi V({nu},{nu}): {dt * 1e3:.2f} ms per cycle = {1 / dt:.1f} V-cycles/s, algorithmic "
      f"{alg / 1e9:.2f} GB per cycle -> {alg / dt / 1e9:.0f} GB/s ({alg / dt / 8e12:.2f} of 8 TB/s); "
      f"F-cycle start {tf * 1e3:.1f} ms; norms from zero guess {norms[0]:.4g} -> {norms[-1]:.4g}; "
      f"after the F-cycle start {after[0]:.4g} {after[1]:.4g} {after[2]:.4g}")
