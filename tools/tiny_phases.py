#!/usr/bin/env python3
"""Where the single-workgroup coarse cycle launch (tiny_cycle_kernel) spends its time: phase stamps of the last launch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_parallel_amd as M
with M.Solver(9, 5, 2) as s:
    s.setup_test_problem()
    s.vcycles(5)
    st = (C.c_longlong * 16)()
    M.lib().mg3d_debug_tiny_stamps(st)
names = ["load d", "pre-smooth", "residual", "restriction + vote", "solve set-up", "solve", "gather x", "store x + prolong", "post-smooth", "store u"]
for i, n in enumerate(names):
    print(f"{n:26s} {(st[i + 1] - st[i]) / 100.0:7.2f} us")
print(f"{'total':26s} {(st[10] - st[0]) / 100.0:7.2f} us")
