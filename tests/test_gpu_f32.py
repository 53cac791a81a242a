"""The single-precision / damped-Jacobi / F-cycle variant (BASELINE configs[4]) against its CPU restatement.

PARITY UNPINNED: the reference has no fp32 arithmetic, no Jacobi smoother and only a commented-out FMG start, so
oracle/mg3d_oracle_f32.c is a statement of intent, not a pinned copy of reference behaviour.  What these tests do
establish: the HIP kernels compute exactly that statement (every grid value bit for bit, norms to the summation
order), the cycle converges, and the binary32 solution agrees with the pinned double-precision path to binary32
accuracy."""
import numpy as np
import pytest

import _oracle as O
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_R, MG3D_U

pytestmark = pytest.mark.gpu
OMEGA = 6.0 / 7.0


def rnd(n, seed):
    return np.random.default_rng(seed).uniform(-1, 1, n ** 3).astype(np.float32)


@pytest.mark.parametrize("c,L", [(3, 3), (5, 3), (9, 3), (3, 5), (5, 4)])
def test_operators_match_the_restatement(c, L):
    lib = O.lib()
    with M.Solver32(c, L, 2, OMEGA) as s:
        top = L - 1
        N, Nc = s.level_n(top), s.level_n(top - 1)
        h = np.float32(1.0 / (N - 1))
        u, d = rnd(N, 1), rnd(N, 2)
        s.upload(MG3D_U, top, u)
        s.upload(MG3D_D, top, d)
        # smoother: 1, 2 and 3 sweeps (buffer parity)
        for iters in (1, 2, 3):
            s.upload(MG3D_U, top, u)
            s.smooth(top, iters)
            want, scratch = u.copy(), np.zeros_like(u)
            lib.orc32_smooth(O.PF(want), O.PF(d), O.PF(scratch), N, h, np.float32(OMEGA), iters)
            assert np.array_equal(s.download(MG3D_U, top), want), f"{iters} sweeps"
        # residual + norm
        r0 = rnd(N, 3)
        s.upload(MG3D_R, top, r0)
        got_norm = s.residual(top, store=True)
        want_r = r0.copy()
        want_norm = lib.orc32_residual(O.PF(want), O.PF(d), N, h, O.PF(want_r))
        assert np.array_equal(s.download(MG3D_R, top), want_r)  # boundary of r untouched
        assert got_norm == pytest.approx(want_norm, rel=1e-12)
        # restriction
        s.restrict(top)
        want_dc = np.zeros(Nc ** 3, dtype=np.float32)
        lib.orc32_restrict(O.PF(want_r), N, O.PF(want_dc), Nc)
        assert np.array_equal(s.download(MG3D_D, top - 1), want_dc)
        # prolongation
        ec = rnd(Nc, 4)
        s.upload(MG3D_U, top - 1, ec)
        s.prolong(top)
        lib.orc32_prolong(O.PF(ec), Nc, O.PF(want), N)
        assert np.array_equal(s.download(MG3D_U, top), want)
        # boundary fill
        s.zero(MG3D_U, top)
        s.fill_boundary(MG3D_U, top)
        want_b = np.zeros(N ** 3, dtype=np.float32)
        lib.orc32_fill_boundary(O.PF(want_b), N, 1.0 / (N - 1))
        assert np.array_equal(s.download(MG3D_U, top), want_b)
        # coarsest solve through double
        n0 = c ** 3
        b0 = rnd(c, 5)
        s.upload(MG3D_D, 0, b0)
        s.coarse_solve()
        LU = np.zeros(n0 * n0)
        lib.orc_coarse_matrix(O.P(LU), c, (1.0 / (N - 1)) * (1 << (L - 1)))
        lib.orc_lu_factor(O.P(LU), n0)
        want_x = np.zeros(n0, dtype=np.float32)
        lib.orc32_coarse_solve(O.P(LU), n0, O.PF(b0), O.PF(want_x))
        assert np.array_equal(s.download(MG3D_U, 0), want_x)


@pytest.mark.parametrize("c,L,nu,fmg", [(5, 4, 2, False), (5, 4, 2, True), (3, 5, 1, False), (9, 3, 3, True), (9, 4, 2, False)])
def test_cycles_match_the_restatement(c, L, nu, fmg):
    cycles = 6
    N = (c - 1) * (1 << (L - 1)) + 1
    want_norms, want_u = np.zeros(cycles), np.zeros(N ** 3, dtype=np.float32)
    O.lib().orc32_run_problem(c, L, nu, OMEGA, cycles, 1 if fmg else 0, O.P(want_norms), O.PF(want_u))
    with M.Solver32(c, L, nu, OMEGA) as s:
        s.setup_test_problem(fmg=fmg)
        norms = s.vcycles(cycles)
        u = s.download(MG3D_U, L - 1)
    assert np.array_equal(u, want_u)
    np.testing.assert_allclose(norms, want_norms, rtol=1e-12)
    # it converges: from a zero guess by orders of magnitude, from the F-cycle start down to the binary32 floor
    assert norms[-1] < norms[0] * (0.5 if fmg else 0.1)


def test_f32_solution_agrees_with_the_pinned_double_path():
    """65^3: after enough cycles both paths sit at their discretisation/rounding floor; the binary32 solution is
    the double one to a few hundred binary32 ulps of the solution's magnitude (|u| <= 2)."""
    c, L, nu = 5, 5, 2
    with M.Solver(c, L, nu) as d64:
        d64.setup_test_problem()
        d64.vcycles(12)
        u64 = d64.download(MG3D_U, L - 1)
    with M.Solver32(c, L, nu, OMEGA) as s:
        s.setup_test_problem(fmg=True)
        n = s.vcycles(25)
        u32 = s.download(MG3D_U, L - 1)
    assert np.abs(u32.astype(np.float64) - u64).max() < 2e-4
    assert n[-1] < 2 * n.min()  # parked at the binary32 floor, not drifting


def jacobi_np(u, d, N, omega=OMEGA):
    U, D = u.reshape(N, N, N), d.reshape(N, N, N)
    h = np.float32(1.0 / (N - 1))
    ssum = U[:-2, 1:-1, 1:-1] + U[2:, 1:-1, 1:-1]
    ssum = ssum + U[1:-1, :-2, 1:-1]
    ssum = ssum + U[1:-1, 2:, 1:-1]
    ssum = ssum + U[1:-1, 1:-1, :-2]
    ssum = ssum + U[1:-1, 1:-1, 2:]
    ssum = ssum - (h * h) * D[1:-1, 1:-1, 1:-1]
    gs = np.float32(1.0) / np.float32(6.0) * ssum
    want = U.copy()
    want[1:-1, 1:-1, 1:-1] = U[1:-1, 1:-1, 1:-1] + np.float32(omega) * (gs - U[1:-1, 1:-1, 1:-1])
    return want.reshape(-1)


@pytest.mark.parametrize("c,L", [(9, 6), (5, 7), (3, 8)])
def test_large_level_shapes(c, L):
    """257^3 (three hierarchies): partial last vectors and tiles of the single-sweep and the paired-sweep kernels (one, two,
    three and four sweeps: pairs, pair + single) against numpy slices."""
    with M.Solver32(c, L, 2, OMEGA) as s:
        top = L - 1
        N = s.level_n(top)
        u, d = rnd(N, 11), rnd(N, 12)
        s.upload(MG3D_D, top, d)
        want = u
        for iters in (1, 2, 3, 4):
            s.upload(MG3D_U, top, u)
            s.smooth(top, iters)
            want = jacobi_np(want, d, N)
            assert np.array_equal(s.download(MG3D_U, top), want), f"{iters} sweeps at {N}^3"


@pytest.mark.parametrize("knob", ["MG3D_F32_NO_PAIRS", "MG3D_F32_NO_FUSE", "MG3D_F32_NO_CARRY"])
def test_fused_and_separate_launches_agree(monkeypatch, knob):
    """MG3D_F32_NO_PAIRS=1 runs every sweep as its own launch, MG3D_F32_NO_FUSE=1 stores r and restricts it in a
    second launch and takes the norm in a launch of its own: same bits as the two-sweeps-per-launch, the
    residual+restriction and the sweeps+norm kernels.  MG3D_F32_NO_CARRY=1: every cycle forms its own norm as a third
    stage of its last launch; by default all cycles of a call but the last leave it to a tap on the next cycle's first
    sweep (a Jacobi sweep gathers the six neighbours of the field it starts from anyway)."""
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv(knob, flag)
        with M.Solver32(9, 5, 2, OMEGA) as s:
            s.setup_test_problem()
            res.append((s.vcycles(4), s.download(MG3D_U, 4)))
    assert np.array_equal(res[0][1], res[1][1])
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-12)  # the fused norm sums in another order


def test_fcycle_start_keeps_the_interpolated_guess():
    """The reference's vcycle zeroes u[q] on entry below the finest level (mg_dirichlet_analytic.c:698-700), so its FMG
    start (:771-806) keeps the interpolated guess only on the finest level; the fp64 path reproduces that (pinned).
    This variant deliberately zeroes only the COARSER level before descending: the guess survives on every level (at
    1025^3 the first V-cycle behind the start already sits on the binary32 floor, tests/test_gpu_fullsize.py).  Pinned here: (a) the residual right after
    the start, before any V-cycle, is orders of magnitude below the zero-guess residual; (b) the restatement does the
    same (bit parity above)."""
    c, L, nu = 5, 5, 2
    with M.Solver32(c, L, nu, OMEGA) as s:
        s.setup_test_problem(fmg=False)
        zero_guess = s.residual(L - 1, store=False)
        s.setup_test_problem(fmg=True)
        after_start = s.residual(L - 1, store=False)
        n = s.vcycles(6)
    assert after_start < 1e-3 * zero_guess  # 65^3: 88 against 3e5; a wiped guess would leave the zero-guess residual
    assert n[0] < 0.2 * after_start and n[-1] < n[0]  # and the cycles carry on from there (65^3: 88 -> 9.4 -> ... -> 0.64)
