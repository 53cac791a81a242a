"""scratch probe (round 4): one-launch-per-leg schedule against the plain one and the oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _oracle as O
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_U

os.environ["MG3D_CARRY_MIN"] = "66"
os.environ["MG3D_LEGS_MIN"] = "66"
ok = True
for c, L, calls in [(9, 5, (5,)), (5, 6, (1, 2, 3)), (17, 4, (4,))]:
    res = []
    for legs, nocarry in (("1", "0"), ("0", "1")):
        os.environ["MG3D_LEGS"] = legs
        os.environ["MG3D_NO_CARRY"] = nocarry
        with M.Solver(c, L, 2) as s:
            s.setup_test_problem()
            norms = []
            for k in calls:
                norms += list(s.vcycles(k))
            res.append((np.array(norms), [s.download(MG3D_U, l) for l in range(L)], [s.download(MG3D_D, l) for l in range(L - 1)]))
    for i, (a, b) in enumerate(zip(res[0][1] + res[0][2], res[1][1] + res[1][2])):
        if not np.array_equal(a, b):
            ok = False
            print(f"{c} {L} batch: field {i} differs: {np.sum(a != b)} of {a.size}, max {np.max(np.abs(a - b))}")
    rel = np.max(np.abs(res[0][0] / res[1][0] - 1))
    print(f"{c} {L} {calls}: batch fields {'identical' if ok else 'DIFFER'}; norms rel {rel:.3e}", flush=True)
    if rel > 1e-12:
        ok = False
    # single calls interleaved with downloads
    logs = []
    for legs, nocarry in (("1", "0"), ("0", "1")):
        os.environ["MG3D_LEGS"] = legs
        os.environ["MG3D_NO_CARRY"] = nocarry
        log = []
        with M.Solver(c, L, 2) as s:
            s.setup_test_problem()
            log.append(s.vcycle()); log.append(s.vcycle())
            log.append(s.download(MG3D_U, L - 1))
            log.append(s.download(MG3D_D, L - 2))
            log.append(s.vcycle()); log.append(s.vcycle()); log.append(s.vcycle())
            log += list(s.vcycles(2))
            log.append(s.vcycle())
            log.append(s.download(MG3D_U, L - 1)); log.append(s.download(MG3D_U, L - 2)); log.append(s.download(MG3D_D, L - 2))
        logs.append(log)
    for i, (a, b) in enumerate(zip(*logs)):
        if isinstance(a, np.ndarray):
            if not np.array_equal(a, b):
                ok = False
                print(f"single: step {i} differs: {np.sum(a != b)} of {a.size}")
        elif abs(a / b - 1) > 1e-12:
            ok = False
            print(f"single: step {i} norm {a} vs {b}")
    print(f"{c} {L}: single-call sequence checked", flush=True)
print("ALL OK" if ok else "FAILED")
if os.environ.get("BIG"):
    os.environ.pop("MG3D_CARRY_MIN"); os.environ.pop("MG3D_LEGS_MIN")
    for legs in ("1", "0"):
        os.environ["MG3D_LEGS"] = legs
        os.environ["MG3D_NO_CARRY"] = "0"
        with M.Solver(9, 7, 2) as s:
            s.setup_test_problem()
            s.vcycles(3)
            t0 = time.perf_counter(); n = s.vcycles(20); t = time.perf_counter() - t0
            print(f"513^3 legs={legs}: {t / 20 * 1e3:.3f} ms per cycle, last norm {n[-1]:.6e}", flush=True)
            s.timing_enable(3)
            s.vcycles(8)
            for (lvl, kn), (cnt, sec) in sorted(s.kernel_times().items()):
                if lvl == 6 and cnt:
                    print(f"   {kn:18s} {cnt:3d} x {sec / cnt * 1e3:.4f} ms")
