"""ms per V(2,2) cycle of the two schedules of consecutive cycles by problem size: `legs` (one launch per leg on the finest level) against
`carried` (three launches), same process, same box.  python tools/legs_by_size.py [c,L ...]   (default: a ladder from 129^3 to 1025^3)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_parallel_amd as M

cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(5, 6), (7, 6), (9, 6), (7, 7), (9, 7), (11, 7), (7, 8), (9, 8)]
for c, L in cases:
    N = (c - 1) * 2 ** (L - 1) + 1
    reps = 40 if N < 300 else 20 if N < 700 else 8
    out = []
    for legs in (0, 1, 0, 1):
        with M.Solver(c, L, 2) as s:
            s.set_option("legs", legs)
            s.set_option("legs_min", 0)
            s.setup_test_problem()
            s.vcycles(3)
            t0 = time.perf_counter(); n = s.vcycles(reps); t = time.perf_counter() - t0
            out.append((t / reps * 1e3, n[-1]))
    assert all(abs(o[1] - out[0][1]) <= 1e-13 * out[0][1] for o in out), out  # the norm's partial sums are grouped per schedule
    print(f"{N:5d}^3 (c={c}, L={L})  carried {out[0][0]:8.3f} {out[2][0]:8.3f}   legs {out[1][0]:8.3f} {out[3][0]:8.3f} ms per cycle", flush=True)
