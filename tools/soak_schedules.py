"""Soak of the three finest-level schedules at 513^3 (run on the GPU box): a random problem, then many rounds of batch calls
(vcycles(k), k random) and single calls (vcycle()) interleaved with downloads, in each schedule; after every round the
finest u of the schedules must be identical bit for bit (hash), the norms equal to 1e-12.  Catches what a short test
cannot: a rare hazard in the LDS rings / run-ahead buffers would show as a hash mismatch after hundreds of cycles."""
import hashlib, sys, time
import numpy as np
sys.path.insert(0, '.')
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_U

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N, top = 513, 6
rng = np.random.default_rng(2026)
d = rng.uniform(-1, 1, N ** 3)
u0 = rng.uniform(-1, 1, N ** 3)
sols = {}
for name, opts in (("legs", {"legs": 1}), ("carried", {"legs": 0, "carry": 1}), ("plain", {"legs": 0, "carry": 0})):
    s = M.Solver(9, 7, 2)
    for k, v in opts.items():
        s.set_option(k, v)
    s.get_details()
    s.upload(MG3D_D, top, d)
    s.upload(MG3D_U, top, u0)
    sols[name] = s
del d, u0
plan = np.random.default_rng(7)
t0 = time.time()
total = 0
for r in range(rounds):
    kind = int(plan.integers(0, 3))
    k = int(plan.integers(1, 40))
    out = {}
    for name, s in sols.items():
        if kind == 0:
            norms = list(s.vcycles(k))
        elif kind == 1:
            norms = [s.vcycle() for _ in range(min(k, 12))]
        else:
            norms = [s.vcycle() for _ in range(3)] + list(s.vcycles(k)) + [s.vcycle()]
        u = s.download(MG3D_U, top)
        out[name] = (np.array(norms), hashlib.sha256(u.tobytes()).hexdigest(), hashlib.sha256(s.download(MG3D_D, top - 1).tobytes()).hexdigest())
        del u
    total += len(out["legs"][0])
    ref = out["plain"]
    for name in ("legs", "carried"):
        assert out[name][1] == ref[1], f"round {r}: u of {name} differs from plain"
        assert out[name][2] == ref[2], f"round {r}: coarse d of {name} differs from plain"
        np.testing.assert_allclose(out[name][0], ref[0], rtol=1e-12)
    print(f"round {r:2d} kind {kind} k {k:2d}: {len(ref[0])} cycles, last norm {ref[0][-1]:.6e}  identical in all three schedules", flush=True)
print(f"soak ok: {total} cycles per schedule in {time.time() - t0:.1f} s")
