#!/usr/bin/env python3
"""PCIe-inclusive rate of a 16-cycle 513^3 solve: what the Solver* facade does around the device-resident loop --
upload the finest u and d (pageable host arrays), 16 V-cycles with the norm read back each cycle, download u."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_U, P

c, L, nu, cycles = 9, 7, 2, 16
with M.Solver(c, L, nu) as s:
    s.get_details()
    N, h = s.N, s.h
    full = np.zeros(N ** 3)
    M.lib().mg3d_fill_boundary_host(P(full), N, h)
    s.upload(MG3D_U, L - 1, full); s.upload(MG3D_D, L - 1, full); s.lin_solve(); s.sync()  # warm
    t0 = time.perf_counter()
    s.upload(MG3D_U, L - 1, full)
    s.upload(MG3D_D, L - 1, full)
    t1 = time.perf_counter()
    norms = [s.lin_solve() for _ in range(cycles)]
    t2 = time.perf_counter()
    u = s.download(MG3D_U, L - 1)
    t3 = time.perf_counter()
print(f"upload u,d {1e3 * (t1 - t0):.1f} ms, {cycles} cycles (norm read back each) {1e3 * (t2 - t1):.1f} ms, download u "
      f"{1e3 * (t3 - t2):.1f} ms -> PCIe-inclusive {cycles / (t3 - t0):.1f} V-cycles/s, device-resident "
      f"{cycles / (t2 - t1):.1f} V-cycles/s")
