#!/usr/bin/env python3
"""PCIe-inclusive rate of a 16-cycle 513^3 solve: what the Solver* facade does around the device-resident loop --
upload the finest u and d, 16 V-cycles with the norm read back each cycle, download u.  Once with pageable host
arrays (plain numpy), once with the page-locked arrays the facade gets from mg3d_host_alloc."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_U, P

import ctypes as C

c, L, nu, cycles = 9, 7, 2, 16


def pinned(n):
    p = C.c_void_p()
    assert M.lib().mg3d_host_alloc(n * 8, C.byref(p)) == 0
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n,)), p


for kind in ("pageable", "page-locked"):
  with M.Solver(c, L, nu) as s:
    s.get_details()
    N, h = s.N, s.h
    if kind == "pageable":
        full, out, handles = np.zeros(N ** 3), np.zeros(N ** 3), []
    else:
        (full, h1), (out, h2) = pinned(N ** 3), pinned(N ** 3)
        handles = [h1, h2]
    M.lib().mg3d_fill_boundary_host(P(full), N, h)
    lib, hnd = M.lib(), s._h
    lib.mg3d_upload(hnd, MG3D_U, L - 1, P(full)); lib.mg3d_upload(hnd, MG3D_D, L - 1, P(full)); s.lin_solve(); s.sync()  # warm
    t0 = time.perf_counter()
    lib.mg3d_upload(hnd, MG3D_U, L - 1, P(full))
    lib.mg3d_upload(hnd, MG3D_D, L - 1, P(full))
    t1 = time.perf_counter()
    norms = [s.lin_solve() for _ in range(cycles)]
    t2 = time.perf_counter()
    lib.mg3d_download(hnd, MG3D_U, L - 1, P(out))
    t3 = time.perf_counter()
    for hh in handles:
        lib.mg3d_host_free(hh)
  print(f"{kind:11s}: upload u,d {1e3 * (t1 - t0):.1f} ms, {cycles} cycles (norm read back each) {1e3 * (t2 - t1):.1f} ms, download u "
      f"{1e3 * (t3 - t2):.1f} ms -> PCIe-inclusive {cycles / (t3 - t0):.1f} V-cycles/s, device-resident "
      f"{cycles / (t2 - t1):.1f} V-cycles/s")
