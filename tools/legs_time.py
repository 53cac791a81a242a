"""Finest-level launch times of a schedule at 513^3 (LEVELS=n for another size): MG3D_LEGS=1 (default) one launch per leg,
MG3D_LEGS=0 the carried cycles; with MG3D_LIB_PATH an A/B against another build of the library (tools/build_variant.sh)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_parallel_amd as M
L = int(os.environ.get("LEVELS", "7"))
os.environ.setdefault("MG3D_LEGS", "1")
with M.Solver(9, L, 2) as s:
    s.setup_test_problem()
    s.vcycles(3)
    t0 = time.perf_counter(); n = s.vcycles(20); t = time.perf_counter() - t0
    print(f"legs={os.environ['MG3D_LEGS']} env {[(k, v) for k, v in os.environ.items() if k.startswith('MG3D_SWEEP')]}: {t / 20 * 1e3:.3f} ms per cycle, last norm {n[-1]:.6e}", flush=True)
    s.timing_enable(3)
    s.vcycles(8)
    for (lvl, kn), (cnt, sec) in sorted(s.kernel_times().items()):
        if lvl == L - 1 and cnt:
            print(f"   {kn:18s} {cnt:3d} x {sec / cnt * 1e3:.4f} ms")
