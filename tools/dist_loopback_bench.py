#!/usr/bin/env python3
"""Loopback rehearsal of the i-slab schedule on ONE GPU: P virtual ranks of the 513^3 problem in one process.
Reports time per V-cycle (all ranks' work serialised on one device: single-GPU time x (1 + halo redundancy)
+ the device copies that stand in for the RCCL exchanges) and checks the norms against the single-domain run."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_parallel_amd as M

c, L, nu = 9, 7, 2
steps = 10
with M.Solver(c, L, nu) as s:
    s.setup_test_problem(); s.vcycles(2); s.sync()
    t0 = time.perf_counter(); ref = s.vcycles(steps); t1 = time.perf_counter()
print(f"single domain        : {(t1 - t0) / steps * 1e3:7.3f} ms/cycle")
for P in [int(x) for x in (sys.argv[1:] or ["2", "4", "8"])]:
    with M.DistSolver(c, L, nu, nranks=P) as d:
        d.setup_test_problem(); d.vcycles(2); d.sync()
        t0 = time.perf_counter(); n = d.vcycles(steps); t1 = time.perf_counter()
    ok = np.allclose(n, ref, rtol=1e-10)
    print(f"loopback {P} ranks      : {(t1 - t0) / steps * 1e3:7.3f} ms/cycle  first distributed level {d.first_level}, halo {d.halo}, norms match: {ok}")

# ---- the fp32 / Jacobi variant (BASELINE configs[4]): 1025^3 on P virtual ranks (MG3D_LOOPBACK_F32=0 skips it)
if os.environ.get("MG3D_LOOPBACK_F32", "1") == "1":
    c, L, nu = 9, 8, 2
    with M.Solver32(c, L, nu) as s:
        s.setup_test_problem(fmg=False); s.vcycles(2); s.sync()
        t0 = time.perf_counter(); ref = s.vcycles(steps); t1 = time.perf_counter()
    print(f"fp32 1025^3 single domain : {(t1 - t0) / steps * 1e3:7.3f} ms/cycle")
    for P in [int(x) for x in (sys.argv[1:] or ["2", "4", "8"])]:
        with M.DistSolver32(c, L, nu, nranks=P) as d:
            d.setup_test_problem(fmg=False); d.vcycles(2); d.sync()
            t0 = time.perf_counter(); n = d.vcycles(steps); t1 = time.perf_counter()
        print(f"fp32 loopback {P} ranks    : {(t1 - t0) / steps * 1e3:7.3f} ms/cycle  first distributed level {d.first_level}, "
              f"halo {d.halo}, norms match: {np.allclose(n, ref, rtol=1e-10)}")
