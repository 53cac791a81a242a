#!/usr/bin/env python3
"""Smoother-only benchmark modelled on the reference's test_rb_gs_3d.c:56-101 (the driver behind its only
published numbers, red_black_gs_scalability.txt): per iteration one pre-smoother sweep (red, black), one
post-smoother sweep (black, red) and the residual norm.  Usage: rb_gs_bench.py [N] [iterations]
Default 513^3 x 50 iterations, no convergence test; prints ms/iteration and smoother HBM GB/s
(algorithmic: 2 RB sweeps x 3*n*w + norm 2*n*w per iteration)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_U, P

N = int(sys.argv[1]) if len(sys.argv) > 1 else 513
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
h = 1.0 / (N - 1)
u = np.zeros(N ** 3)
M.lib().mg3d_fill_boundary_host(P(u), N, h)
with M.Solver(N, 1, 1) as s:
    s.upload(MG3D_U, 0, u)
    s.upload(MG3D_D, 0, np.zeros(N ** 3))
    init = s.residual(0, store=False)
    for _ in range(3):
        s.smooth(0, 0, 1); s.smooth(0, 1, 1)
    s.sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        s.smooth(0, 0, 1)
        s.smooth(0, 1, 1)
        nrm = s.residual(0, store=False, want_norm=False)
    s.sync()
    dt = (time.perf_counter() - t0) / iters
    nrm = s.residual(0, store=False)
n = N ** 3
print(f"N={N}: {dt * 1e3:.3f} ms per iteration (pre+post sweep + norm), smoother+norm algorithmic "
      f"{(6 + 2) * n * 8 / dt / 1e9:.1f} GB/s, {2 * (N - 2) ** 3 / dt / 1e9:.2f} G point-updates/s "
      f"(reference 8 threads: 0.408 G/s at 50^3); residual {init:.6g} -> {nrm:.6g}")

# ---- the fp32 / damped-Jacobi variant of the same protocol (last sentence): per iteration two Jacobi
# sweeps (what a pre + a post sweep are to the red-black smoother) and the residual norm, on the top level of a
# hierarchy whose finest grid is the smallest (2^k (c-1) + 1) >= N  (the variant's contexts are hierarchies)
if os.environ.get("RB_GS_F32", "1") == "1":
    c = 9
    L = 1
    while (c - 1) * (1 << (L - 1)) + 1 < N:
        L += 1
    with M.Solver32(c, L, 2) as s:
        top = L - 1
        Nf = s.level_n(top)
        s.fill_boundary(MG3D_U, top)
        init = s.residual(top, store=False)
        for _ in range(3):
            s.smooth(top, 2)
        s.sync()
        t0 = time.perf_counter()
        for _ in range(iters):
            s.smooth(top, 2)
            s.residual(top, store=False, want_norm=False)
        s.sync()
        dt = (time.perf_counter() - t0) / iters
        nrm = s.residual(top, store=False)
    n = Nf ** 3
    print(f"fp32 Jacobi N={Nf}: {dt * 1e3:.3f} ms per iteration (2 sweeps + norm), compulsory {(3 + 2) * n * 4 / dt / 1e9:.1f} GB/s "
          f"(SURVEY credit {(6 + 2) * n * 4 / dt / 1e9:.1f}), {2 * (Nf - 2) ** 3 / dt / 1e9:.2f} G point-updates/s; "
          f"residual {init:.6g} -> {nrm:.6g}")
