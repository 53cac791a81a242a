import sys, time, numpy as np
sys.path.insert(0, '.')
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_U
with M.Solver(9, 7, 2) as s:
    s.setup_test_problem()
    t0 = time.time()
    n = s.vcycles(1500)
    print("1500 cycles in %.2f s, %.1f V-cycles/s" % (time.time() - t0, 1500 / (time.time() - t0)))
    print("norm[15], norm[100], norm[-1]:", n[15], n[100], n[-1], "max after 30:", n[30:].max())
    assert n[30:].max() < 5e-6 and np.isfinite(n).all()
    u = s.download(MG3D_U, 6).reshape(513, 513, 513)
    x = np.arange(513) / 512.0
    err = 0.0
    for i in range(0, 513, 16):
        err = max(err, np.abs(u[i] - (x[i] ** 2 - 2 * x[:, None] ** 2 + x[None, :] ** 2)).max())
    print("max |u - u*| sampled:", err)
    assert err < 1e-10
print("soak ok")
