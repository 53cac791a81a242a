"""Parity of the HIP path (through the C ABI of libmg3d.so) against the oracle and the golden vectors.
Bar: every grid value bit-identical (integer-exact comparison of fp64); residual norms to rel 1e-12
(the only quantity whose summation order differs: sequential on the CPU, two-stage tree on the GPU)."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_R, MG3D_U, P, check

pytestmark = pytest.mark.gpu


def norm_rtol(N):
    """Tolerance on a residual norm of an N^3 level.  The grid values are bit-identical, so the only
    difference is the order in which the (N-2)^3 non-negative squares are added: the reference adds them
    sequentially (first-order error bound (n-1)*2^-53 relative to the sum, half of that on the sqrt), the
    GPU by a balanced tree (error ~ log2(n)*2^-53).  The bound below is the sequential sum's."""
    return max(1e-13, 0.5 * (N - 2) ** 3 * 2.0 ** -53)


NORM_RTOL = norm_rtol(33)  # levels up to 33^3: 2e-12
# The GPU's OWN reduction (wave shuffle -> LDS -> per-block partial sums -> fold) is held to this against the exactly
# rounded sum of the same squares (O.exact_residual_norm), whatever the size: a dropped or doubled partial sum of 1e-9
# relative weight would pass norm_rtol(513) = 7e-9 -- it does not pass this.
EXACT_NORM_RTOL = 1e-13


def assert_norm_exact(s, level, got_norm):
    """`got_norm` is the residual norm the library returned for the state now on `level`: compare with the exactly
    rounded sum over the oracle's residual field of the downloaded (bit-identical) u and d."""
    u, d = s.download(MG3D_U, level), s.download(MG3D_D, level)
    want = O.exact_residual_norm(u, d, s.level_n(level), s.level_h(level))
    assert got_norm == pytest.approx(want, rel=EXACT_NORM_RTOL), (got_norm, want)


G = np.load(os.path.join(O.GOLDEN, "operators.npz"))
V = np.load(os.path.join(O.GOLDEN, "vcycle.npz"))


def rnd(n, seed):
    return np.random.default_rng(seed).uniform(-1, 1, n)


# ------------------------------------------------------------------ operators, host-pointer forms
@pytest.mark.parametrize("N", [3, 4, 5, 9, 17, 33, 50, 65])
@pytest.mark.parametrize("post,iters", [(0, 1), (0, 2), (1, 1), (1, 3)])
def test_smoother_matches_oracle(N, post, iters):
    h = 1.0 / (N - 1)
    v, d = rnd(N ** 3, 10 + N), rnd(N ** 3, 20 + N)
    want = v.copy()
    (O.lib().orc_post_smooth if post else O.lib().orc_pre_smooth)(O.P(want), O.P(d), N, h, iters)
    got = v.copy()
    check(M.lib().mg3d_host_smooth(P(got), P(d), N, h, iters, post))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("N", [5, 9, 17, 33])
def test_smoother_matches_golden(N):
    for name, post, it in (("pre1", 0, 1), ("pre2", 0, 2), ("post1", 1, 1), ("post3", 1, 3)):
        v = G[f"sm_v0_{N}"].copy()
        check(M.lib().mg3d_host_smooth(P(v), P(G[f"sm_d0_{N}"]), N, 1.0 / (N - 1), it, post))
        assert np.array_equal(v, G[f"sm_{name}_{N}"]), name


@pytest.mark.parametrize("N", [3, 4, 5, 9, 17, 33, 50, 65, 129])
def test_residual_matches_oracle(N):
    h = 1.0 / (N - 1)
    v, d = rnd(N ** 3, 30 + N), rnd(N ** 3, 40 + N)
    want = np.full(N ** 3, 3.25)
    got = want.copy()  # boundary entries of res must survive untouched (mg_3d.h:824-825)
    O.lib().orc_set_threads(1)
    wn = O.lib().orc_residual(O.P(v), O.P(d), N, h, O.P(want))
    gn = C.c_double(0)
    check(M.lib().mg3d_host_residual(P(v), P(d), N, h, P(got), C.byref(gn)))
    assert np.array_equal(got, want)
    assert gn.value == pytest.approx(wn, rel=norm_rtol(N))
    # against the correctly rounded sum of the (bit-identical) squares: the tree sum is far tighter
    import math
    exact = math.sqrt(math.fsum((got.reshape(N, N, N)[1:-1, 1:-1, 1:-1].ravel() ** 2).tolist()))
    assert gn.value == pytest.approx(exact, rel=1e-14)
    gn2 = C.c_double(0)
    check(M.lib().mg3d_host_residual(P(v), P(d), N, h, None, C.byref(gn2)))
    assert gn2.value == gn.value  # deterministic reduction: same value with and without the store


@pytest.mark.parametrize("N", [5, 9, 17, 33])
def test_residual_matches_golden(N):
    res = np.zeros(N ** 3)
    gn = C.c_double(0)
    check(M.lib().mg3d_host_residual(P(G[f"sm_v0_{N}"]), P(G[f"sm_d0_{N}"]), N, 1.0 / (N - 1), P(res), C.byref(gn)))
    assert np.array_equal(res, G[f"res_r_{N}"])
    assert gn.value == pytest.approx(G[f"res_norm_{N}"][0], rel=NORM_RTOL)


@pytest.mark.parametrize("Nc", [2, 3, 5, 9, 17, 33, 65])
def test_restrict_matches_oracle(Nc):
    Nf = 2 * Nc - 1
    r = rnd(Nf ** 3, 50 + Nc)  # non-zero boundary exercises the injection faces
    want, got = np.full(Nc ** 3, 9.0), np.full(Nc ** 3, -9.0)
    O.lib().orc_restrict(O.P(r), Nf, O.P(want), Nc)
    check(M.lib().mg3d_host_restrict(P(r), Nf, P(got), Nc))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("Nc", [3, 5, 9, 17])
def test_restrict_matches_golden(Nc):
    Nf = 2 * Nc - 1
    got = np.zeros(Nc ** 3)
    check(M.lib().mg3d_host_restrict(P(G[f"rs_r_{Nf}"]), Nf, P(got), Nc))
    assert np.array_equal(got, G[f"rs_dc_{Nc}"])


@pytest.mark.parametrize("Nc", [2, 3, 5, 9, 17, 33, 65])
def test_prolong_matches_oracle(Nc):
    Nf = 2 * Nc - 1
    ec, ef = rnd(Nc ** 3, 60 + Nc), rnd(Nf ** 3, 70 + Nc)
    want, got = ef.copy(), ef.copy()
    O.lib().orc_prolong(O.P(ec), Nc, O.P(want), Nf)
    check(M.lib().mg3d_host_prolong(P(ec), Nc, P(got), Nf))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("Nc", [3, 5, 9, 17])
def test_prolong_matches_golden(Nc):
    Nf = 2 * Nc - 1
    got = G[f"pr_ef0_{Nf}"].copy()
    check(M.lib().mg3d_host_prolong(P(G[f"pr_ec_{Nc}"]), Nc, P(got), Nf))
    assert np.array_equal(got, G[f"pr_ef_{Nf}"])


@pytest.mark.parametrize("c", [3, 5, 9])
def test_lu_solve_matches_golden_and_oracle(c):
    n = c ** 3
    h = 0.125 if c == 9 else 1.0 / (c - 1) / 7.0
    A = np.zeros(n * n)
    M.lib().mg3d_coarse_matrix(P(A), c, h)
    M.lib().mg3d_lu_factor(P(A), n)
    A_or = np.zeros(n * n)
    O.lib().orc_coarse_matrix(O.P(A_or), c, h)
    O.lib().orc_lu_factor(O.P(A_or), n)
    assert np.array_equal(A, A_or)  # host factorisation (band-skipping) == literal dense sweep
    b = G[f"lu_b_{c}"]
    x = np.zeros(n)
    check(M.lib().mg3d_host_lu_solve(P(A), n, P(b), P(x)))
    assert np.array_equal(x, G[f"lu_x_{c}"])


@pytest.mark.parametrize("c", [5, 9])
@pytest.mark.parametrize("kind", ["tiny", "huge", "zeros", "negzero", "mixed"])
def test_lu_solve_numerators_outside_the_fast_division_window(c, kind):
    """The back substitution divides through a host reciprocal with two FMA refinements (lu_div) while the
    numerator lies in a wide exponent window, and falls back to the ordinary division chunk-wise otherwise.
    Right-hand sides that leave the window (denormals, 1e200, signed zeros) must still match the oracle bit
    for bit, signs of zero included."""
    n = c ** 3
    h = 1.0 / (c - 1)
    A = np.zeros(n * n)
    O.lib().orc_coarse_matrix(O.P(A), c, h)
    O.lib().orc_lu_factor(O.P(A), n)
    rng = np.random.default_rng(c)
    b = rng.uniform(-1, 1, n)
    if kind == "tiny":
        b *= 1e-308
    elif kind == "huge":
        b *= 1e200
    elif kind == "zeros":
        b[rng.random(n) < 0.7] = 0.0
    elif kind == "negzero":
        b[rng.random(n) < 0.7] = -0.0
    else:
        sel = rng.integers(0, 5, n)
        b = np.where(sel == 0, b * 1e-310, np.where(sel == 1, b * 1e180, np.where(sel == 2, -0.0, np.where(sel == 3, 0.0, b))))
    want, got = np.zeros(n), np.zeros(n)
    O.lib().orc_lu_solve(O.P(A), n, O.P(b), O.P(want))
    check(M.lib().mg3d_host_lu_solve(P(A), n, P(b), P(got)))
    assert np.array_equal(got, want)
    assert np.array_equal(np.signbit(got), np.signbit(want))


@pytest.mark.parametrize("c", [5, 9])
@pytest.mark.parametrize("kind", ["zero_faces", "negzero_faces", "mixed_zero_faces", "one_face_entry", "all_zero", "random_identity_rows"])
def test_lu_solve_reduced_system_path(c, kind, monkeypatch):
    """The direct solve drops the factor's identity rows (the boundary rows of constructCoarseMatrixA, mg_3d.h:179-185)
    when every right-hand side entry on them is +-0 -- what a V-cycle hands it -- and takes the full system otherwise.
    Both routes against the oracle, bit for bit and sign of zero for sign of zero, and against each other
    (MG3D_LU_REDUCED=0: no reduced factor is built)."""
    n = c ** 3
    A = np.zeros(n * n)
    O.lib().orc_coarse_matrix(O.P(A), c, 1.0 / (c - 1))
    O.lib().orc_lu_factor(O.P(A), n)
    rng = np.random.default_rng(17 * c)
    b = rng.uniform(-1, 1, (c, c, c))
    face = np.ones((c, c, c), dtype=bool)
    face[1:-1, 1:-1, 1:-1] = False
    if kind == "zero_faces":
        b[face] = 0.0
    elif kind == "negzero_faces":
        b[face] = -0.0
    elif kind == "mixed_zero_faces":
        b[face] = np.where(rng.random(int(face.sum())) < 0.5, 0.0, -0.0)
        b[1:-1, 1:-1, 1:-1][rng.random((c - 2,) * 3) < 0.3] = 0.0
    elif kind == "one_face_entry":
        b[face] = 0.0
        b[c - 1, c // 2, c // 2] = 1e-300  # one entry on an identity row that is not a zero: the full system
    elif kind == "all_zero":
        b[:] = 0.0
        b[face] = -0.0
    b = b.reshape(-1).copy()
    LU = A
    if kind == "random_identity_rows":  # a generic banded factor with identity rows sprinkled in, zero entries on them
        bw = 9
        Ad = np.zeros((n, n))
        ident = rng.random(n) < 0.55
        for i in range(n):
            if ident[i]:
                Ad[i, i] = 1.0
                continue
            lo, hi = max(0, i - bw), min(n, i + bw + 1)
            Ad[i, lo:hi] = rng.uniform(-1, 1, hi - lo)
            Ad[i, i] = 2.0 * bw + 2.0
        LU = Ad.reshape(-1).copy()
        O.lib().orc_lu_factor(O.P(LU), n)
        b[ident] = np.where(rng.random(int(ident.sum())) < 0.5, 0.0, -0.0)
    want, got, full = np.zeros(n), np.zeros(n), np.zeros(n)
    O.lib().orc_lu_solve(O.P(LU), n, O.P(b), O.P(want))
    check(M.lib().mg3d_host_lu_solve(P(LU), n, P(b), P(got)))
    monkeypatch.setenv("MG3D_LU_REDUCED", "0")
    check(M.lib().mg3d_host_lu_solve(P(LU), n, P(b), P(full)))
    for x in (got, full):
        assert np.array_equal(x, want)
        assert np.array_equal(np.signbit(x), np.signbit(want))


def test_lu_solve_dense_random_matrix():
    # a full (non-banded) diagonally dominant factor: exercises the wide-band block kernel
    n = 200
    rng = np.random.default_rng(7)
    A = rng.uniform(-1, 1, (n, n)) + n * np.eye(n)
    LU = A.reshape(-1).copy()
    O.lib().orc_lu_factor(O.P(LU), n)
    b = rng.uniform(-1, 1, n)
    want, got = np.zeros(n), np.zeros(n)
    O.lib().orc_lu_solve(O.P(LU), n, O.P(b), O.P(want))
    check(M.lib().mg3d_host_lu_solve(P(LU), n, P(b), P(got)))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,bw", [(64, 5), (65, 63), (200, 64), (300, 65), (257, 100), (511, 128), (640, 127), (130, 129)])
def test_lu_solve_random_banded_matrix(n, bw):
    """Generic banded factors (not the Poisson operator): unknown counts that are not multiples of the 64-step
    chunk, half bandwidths on either side of the one-/two-rows-per-lane and the single-wave/block-kernel boundaries
    (64, 128).  Asymmetric band content, random right-hand side with a few exact zeros."""
    rng = np.random.default_rng(n * 131 + bw)
    A = np.zeros((n, n))
    for i in range(n):
        lo, hi = max(0, i - bw), min(n, i + bw + 1)
        A[i, lo:hi] = rng.uniform(-1, 1, hi - lo)
        A[i, i] = 2.0 * bw + 1.0 + rng.uniform(0, 1)
    LU = A.reshape(-1).copy()
    O.lib().orc_lu_factor(O.P(LU), n)
    b = rng.uniform(-1, 1, n)
    b[rng.random(n) < 0.1] = 0.0
    want, got = np.zeros(n), np.zeros(n)
    O.lib().orc_lu_solve(O.P(LU), n, O.P(b), O.P(want))
    check(M.lib().mg3d_host_lu_solve(P(LU), n, P(b), P(got)))
    assert np.array_equal(got, want)


# ------------------------------------------------------------------ whole V-cycles
@pytest.mark.parametrize("c,L,nu", [(3, 3, 1), (3, 5, 2), (5, 3, 3), (9, 2, 2), (5, 5, 2), (9, 5, 2), (3, 2, 0)])
def test_vcycle_history_and_solution_bit_exact(c, L, nu):
    cycles = 15 if f"norms_{c}_{L}_{nu}" not in V else len(V[f"norms_{c}_{L}_{nu}"])
    O.lib().orc_set_threads(1)
    want_norms, want_u, want_init, _ = O.run_problem(c, L, nu, cycles)
    with M.Solver(c, L, nu) as s:
        s.setup_test_problem()
        rt = norm_rtol(s.N)
        assert s.get_initial_residual() == pytest.approx(want_init, rel=rt)
        got = np.array([s.lin_solve() for _ in range(cycles)])
        u = s.download(MG3D_U, L - 1)
        assert_norm_exact(s, L - 1, got[-1])  # the GPU reduction itself, to 1e-13 (the bound above is the reference's)
    np.testing.assert_allclose(got, want_norms, rtol=rt, atol=0)
    assert np.array_equal(u, want_u)
    key = f"{c}_{L}_{nu}"
    if f"norms_{key}" in V:  # and against the compiled reference itself
        np.testing.assert_allclose(got, V[f"norms_{key}"], rtol=rt, atol=0)
        if f"u_{key}" in V:
            assert np.array_equal(u, V[f"u_{key}"])
        else:
            assert np.array_equal(u[::97], V[f"usample_{key}"])


def test_vcycles_batch_equals_single_calls():
    with M.Solver(5, 4, 2) as a, M.Solver(5, 4, 2) as b:
        a.setup_test_problem()
        b.setup_test_problem()
        one = np.array([a.lin_solve() for _ in range(6)])
        many = b.vcycles(6)
        assert np.array_equal(one, many)
        assert np.array_equal(a.download(MG3D_U, 3), b.download(MG3D_U, 3))


def test_intermediate_levels_match_oracle_after_one_cycle():
    """Every level of u, d, r after one V-cycle (the reference's arrays are all observable)."""
    c, L, nu = 5, 4, 2
    H = O.Hierarchy(c, L)
    N, h = H.N[-1], 1.0 / (H.N[-1] - 1)
    O.lib().orc_fill_boundary(O.P(H.d[-1]), N, h)
    O.lib().orc_fill_boundary(O.P(H.u[-1]), N, h)
    n0 = c ** 3
    LU = np.zeros(n0 * n0)
    O.lib().orc_coarse_matrix(O.P(LU), c, h * (1 << (L - 1)))
    O.lib().orc_lu_factor(O.P(LU), n0)
    O.lib().orc_set_threads(1)
    wn = O.lib().orc_vcycle(H.ptrs(H.u), H.ptrs(H.d), H.ptrs(H.r), h, L - 1, L, nu, N, O.P(LU))
    with M.Solver(c, L, nu) as s:
        s.set_keep_residual(True)  # r is a reference-visible array; by default it is restricted on the fly
        s.setup_test_problem()
        gn = s.lin_solve()
        assert gn == pytest.approx(wn, rel=NORM_RTOL)
        for l in range(L):
            assert np.array_equal(s.download(MG3D_U, l), H.u[l]), f"u level {l}"
            assert np.array_equal(s.download(MG3D_D, l), H.d[l]), f"d level {l}"
            assert np.array_equal(s.download(MG3D_R, l), H.r[l]), f"r level {l}"


@pytest.mark.parametrize("c,L,nu", [(5, 5, 2), (3, 4, 2)])
def test_legacy_host_vcycle_dirichlet_protocol(c, L, nu):
    """test_mg_3d_dirichlet.c: caller-owned host hierarchies, LU built with the FINEST h (:40)."""
    key = f"dir_{c}_{L}_{nu}"
    ref = V[f"norms_{key}"]
    H = O.Hierarchy(c, L)
    N, h = H.N[-1], 1.0 / (H.N[-1] - 1)
    n0 = c ** 3
    LU = np.zeros(n0 * n0)
    M.lib().mg3d_coarse_matrix(P(LU), c, h)
    M.lib().mg3d_lu_factor(P(LU), n0)
    M.lib().mg3d_fill_boundary_host(P(H.u[-1]), N, h)
    init = C.c_double(0)
    check(M.lib().mg3d_host_residual(P(H.u[-1]), P(H.d[-1]), N, h, None, C.byref(init)))
    assert init.value == pytest.approx(V[f"init_{key}"][0], rel=norm_rtol(N))
    got = []
    for _ in range(len(ref)):
        nrm = C.c_double(0)
        check(M.lib().mg3d_host_vcycle(H.ptrs(H.u), H.ptrs(H.d), H.ptrs(H.r), h, L - 1, L, nu, N, P(LU), C.byref(nrm), None, None))
        got.append(nrm.value)
    np.testing.assert_allclose(got, ref, rtol=norm_rtol(N), atol=0)
    assert np.array_equal(H.u[-1][::97], V[f"usample_{key}"])


def test_timing_table_counts():
    with M.Solver(5, 3, 2) as s:
        s.setup_test_problem()
        s.timing_enable(True)
        for _ in range(3):
            s.lin_solve()
        t = s.timing()
        assert t[(2, "Smoother1")][0] == 3 and t[(1, "CalcResidual2")][0] == 3
        assert t[(0, "Recurse, Direct Solve")][0] == 3
        assert t[(2, "Smoother1")][1] > 0


def test_error_paths():
    L = M.lib()
    h = C.c_void_p()
    assert L.mg3d_ctx_create(2, 3, 2, 1.0, C.byref(h)) == 1
    with M.Solver(5, 2, 1) as s:
        with pytest.raises(M.Mg3dError) as e:
            s.lin_solve()  # no LU set
        assert e.value.code == 5
        with pytest.raises(M.Mg3dError):
            s.restrict(0)
    a = np.zeros(27)
    assert L.mg3d_host_restrict(P(a), 3, P(a), 3) == 1


def test_unfused_kernel_set_gives_identical_results(monkeypatch):
    """MG3D_NO_FUSE=1 selects the one-launch-per-colour-pass kernels; both kernel sets must agree bit for bit."""
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("MG3D_NO_FUSE", flag)
        with M.Solver(5, 5, 3) as s:  # nu = 3: the fused path chains a 4-pass and a 2-pass launch
            s.set_keep_residual(True)
            s.setup_test_problem()
            norms = s.vcycles(5)
            outs.append((norms, s.download(MG3D_U, 4), s.download(MG3D_R, 3), s.download(MG3D_D, 2)))
    assert np.array_equal(outs[0][0], outs[1][0]) or np.allclose(outs[0][0], outs[1][0], rtol=1e-12, atol=0)
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("c,L,nu", [(5, 5, 2), (9, 4, 1), (5, 4, 3), (3, 6, 2)])
def test_on_the_fly_restriction_equals_materialised_residual(c, L, nu):
    """Default mode (residual restricted on the fly, r never stored) against keep_residual mode: u, d on every
    level and the norms must be identical bit for bit."""
    res = []
    for keep in (False, True):
        with M.Solver(c, L, nu) as s:
            s.set_keep_residual(keep)
            s.setup_test_problem()
            norms = s.vcycles(4)
            res.append((norms, [s.download(MG3D_U, l) for l in range(L)], [s.download(MG3D_D, l) for l in range(L)]))
    assert np.array_equal(res[0][0], res[1][0])
    for l in range(L):
        assert np.array_equal(res[0][1][l], res[1][1][l]), f"u level {l}"
        assert np.array_equal(res[0][2][l], res[1][2][l]), f"d level {l}"


@pytest.mark.parametrize("c,L,nu", [(5, 5, 2), (9, 4, 1), (5, 4, 3), (3, 6, 2), (9, 5, 2)])
def test_prolongation_fused_into_smoother_equals_separate_launch(c, L, nu, monkeypatch):
    """Option fuse_up_max (here through its environment override, read when the context is created) folds
    prolongateAndCorrectError into the post-smoother's four-pass launch on every level (the default since the
    prolonging four-row shape runs without scratch, round 4); with 0 it is its own kernel below the top level.  Same bits
    either way (and both equal the oracle, see the history tests)."""
    import subprocess, sys, json
    outs = []
    for flag in ("0", "1"):
        code = (f"import os,sys,hashlib,json; sys.path.insert(0,{os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r});"
                f"import numpy as np, multigrid_parallel_amd as M;"
                f"s=M.Solver({c},{L},{nu}); s.setup_test_problem(); n=s.vcycles(5);"
                f"u=s.download(0,{L - 1}); print(json.dumps([list(n), hashlib.sha256(u.tobytes()).hexdigest()]))")
        env = dict(os.environ, MG3D_FUSE_UP_MAX="100000" if flag == "1" else "0")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1]


def test_full_size_513_known_answer_history():
    """BASELINE's headline size (args 9 7 2 = 513^3, V(2,2)): the residual history printed by the unmodified
    reference (six digits), the stopping cycle, and the final error against the analytic solution."""
    known = [3.86147e+07, 4.68671e+06, 602953, 80775.4, 11126.2, 1563.03, 222.942, 32.2171, 4.71335, 0.698137,
             0.104742, 0.0159273, 0.00245617, 0.00038423, 6.09488e-05, 9.8336e-06]
    with M.Solver(9, 7, 2) as s:
        s.setup_test_problem()
        init = s.get_initial_residual()
        norms = s.vcycles(16)
        u = s.download(MG3D_U, 6)
    for got, want in zip(norms, known):
        assert got == pytest.approx(want, rel=2e-5)  # 6 printed digits; late cycles sit on the rounding floor
    assert norms[15] <= 1e-8 * init < norms[14]      # test_mg_3d.c:31,40 stops after cycle 16
    N = 513
    g = np.arange(N) / (N - 1)
    err = 0.0
    for i in range(N):  # plane by plane: keeps the temporary small
        exact = g[i] ** 2 - 2 * g[:, None] ** 2 + g[None, :] ** 2
        err += float(((u[i * N * N:(i + 1) * N * N].reshape(N, N) - exact) ** 2).sum())
    assert np.sqrt(err) == pytest.approx(4.48843e-09, rel=1e-3)


def test_smoother_only_loop_50_cubed():
    """The smoother-only loop of test_rb_gs_3d.c:56-101 on 50^3 (N not of the form 2^k+1): one pre- and one
    post-smoother sweep plus the residual norm per iteration, relative tolerance 1e-6.
    Known answer: the reference's CURRENT operators (compiled unmodified, oracle/_ref) stop at iteration 1303
    with 'Residual Norm: 0.28129  ResidRatio: 0.991804'.  The published note red_black_gs_scalability.txt:6-7
    (652 / 0.280135 / 0.983675) was produced by an older state of the code and is not reproducible from the
    reference tree (its driver no longer compiles; 0.991804^2 = 0.983675: that build evidently did twice the
    smoothing per iteration).  The final field is also compared bit for bit with the oracle."""
    N = 50
    h = 1.0 / (N - 1)
    u = np.zeros(N ** 3)
    d = np.zeros(N ** 3)
    M.lib().mg3d_fill_boundary_host(P(u), N, h)  # test_rb_gs_3d.c:34
    with M.Solver(N, 1, 1) as s:  # a one-level context of 50^3 points
        s.upload(MG3D_U, 0, u)
        s.upload(MG3D_D, 0, d)
        init = s.residual(0, store=False)  # :41
        cmp_norm = init * 1e-6
        norms = []
        while True:
            s.smooth(0, 0, 1)  # preSmoother(u,d,N,h,1)  :70
            s.smooth(0, 1, 1)  # postSmoother(u,d,N,h,1) :71
            norms.append(s.residual(0, store=False))
            if not norms[-1] > cmp_norm or len(norms) > 2000:
                break
        got_u = s.download(MG3D_U, 0)
    assert len(norms) == 1303
    assert float(f"{norms[-1]:.6g}") == 0.28129
    assert float(f"{norms[-1] / norms[-2]:.6g}") == 0.991804
    assert float(f"{(norms[-1] / norms[-2]) ** 2:.6g}") == 0.983675
    O.lib().orc_set_threads(4)
    want = u.copy()
    for _ in range(1303):
        O.lib().orc_pre_smooth(O.P(want), O.P(d), N, h, 1)
        O.lib().orc_post_smooth(O.P(want), O.P(d), N, h, 1)
    assert np.array_equal(got_u, want)


@pytest.mark.parametrize("c,L,nu", [(5, 4, 2), (3, 5, 1), (9, 4, 2)])
def test_fmg_initialize_matches_oracle_and_golden(c, L, nu):
    H = O.Hierarchy(c, L)
    N, h = H.N[-1], 1.0 / (H.N[-1] - 1)
    n0 = c ** 3
    LU = np.zeros(n0 * n0)
    O.lib().orc_coarse_matrix(O.P(LU), c, h * (1 << (L - 1)))
    O.lib().orc_lu_factor(O.P(LU), n0)
    O.lib().orc_fill_boundary(O.P(H.d[-1]), N, h)
    O.lib().orc_fill_boundary(O.P(H.u[-1]), N, h)
    O.lib().orc_set_threads(1)
    O.lib().orc_fmg_initialize(H.ptrs(H.u), H.ptrs(H.d), H.ptrs(H.r), c, L, nu, 1.0, O.P(LU))
    with M.Solver(c, L, nu) as s:
        s.setup_test_problem()
        s.fmg_initialize()
        for l in range(L):
            assert np.array_equal(s.download(MG3D_U, l), H.u[l]), f"u level {l}"
        norms = s.vcycles(3)
        u = s.download(MG3D_U, L - 1)
    want = [O.lib().orc_vcycle(H.ptrs(H.u), H.ptrs(H.d), H.ptrs(H.r), h, L - 1, L, nu, N, O.P(LU)) for _ in range(3)]
    np.testing.assert_allclose(norms, want, rtol=norm_rtol(N))
    assert np.array_equal(u, H.u[-1])
    key = f"fmg_{c}_{L}_{nu}"
    if f"u_{key}" in V:
        assert np.array_equal(u, V[f"u_{key}"])


@pytest.mark.parametrize("N", [5, 9, 33])
def test_device_boundary_fill_matches_host(N):
    with M.Solver(N, 1, 1) as s:
        v = rnd(N ** 3, N)
        s.upload(MG3D_U, 0, v)
        s.fill_boundary(MG3D_U, 0)
        want = v.copy()
        M.lib().mg3d_fill_boundary_host(P(want), N, 1.0 / (N - 1))
        assert np.array_equal(s.download(MG3D_U, 0), want)


@pytest.mark.parametrize("N", [1, 2, 3])
def test_degenerate_sizes_are_no_ops_or_match_oracle(N):
    """Grids with an empty interior (N = 1, 2) leave every array untouched and give a zero norm, as the reference's
    loops (1 .. N-2) do; N = 3 has a single interior point."""
    h = 0.5
    v, d = rnd(N ** 3, 5), rnd(N ** 3, 6)
    got, want = v.copy(), v.copy()
    check(M.lib().mg3d_host_smooth(P(got), P(d), N, h, 2, 0))
    O.lib().orc_pre_smooth(O.P(want), O.P(d), N, h, 2)
    assert np.array_equal(got, want)
    res_g, res_w = np.full(N ** 3, 1.5), np.full(N ** 3, 1.5)
    gn = C.c_double(-1)
    check(M.lib().mg3d_host_residual(P(got), P(d), N, h, P(res_g), C.byref(gn)))
    O.lib().orc_set_threads(1)
    wn = O.lib().orc_residual(O.P(want), O.P(d), N, h, O.P(res_w))
    assert np.array_equal(res_g, res_w) and gn.value == pytest.approx(wn, rel=1e-14, abs=0)
    if N < 3:
        assert gn.value == 0.0 and np.array_equal(got, v)


def test_zero_smoothing_iterations_and_single_level():
    """nu = 0 (the V-cycle degenerates to residual / restrict / solve / prolong) and a one-level hierarchy
    (vcycle at q = 0 is the direct solve and returns 0, mg_3d.h:1262-1277)."""
    O.lib().orc_set_threads(1)
    want_norms, want_u, _, _ = O.run_problem(5, 3, 0, 3)
    with M.Solver(5, 3, 0) as s:
        s.setup_test_problem()
        got = s.vcycles(3)
        assert np.array_equal(s.download(MG3D_U, 2), want_u)
    np.testing.assert_allclose(got, want_norms, rtol=1e-12)
    with M.Solver(5, 1, 2) as s:
        s.get_details()
        b = rnd(125, 9)
        s.upload(MG3D_D, 0, b)
        assert s.vcycle(0) == 0.0
        LU = np.zeros(125 * 125)
        O.lib().orc_coarse_matrix(O.P(LU), 5, s.h)
        O.lib().orc_lu_factor(O.P(LU), 125)
        x = np.zeros(125)
        O.lib().orc_lu_solve(O.P(LU), 125, O.P(b), O.P(x))
        assert np.array_equal(s.download(MG3D_U, 0), x)


def test_host_vcycle_started_below_the_finest_level():
    """vcycle(u,f,res,h,q,...) with q < numLevels-1 zeroes its own guess first (mg_3d.h:1254-1260), which is how the
    FMG driver calls it; caller-owned hierarchies, every level copied back."""
    c, L, nu, q = 5, 4, 2, 2
    H, G_ = O.Hierarchy(c, L), O.Hierarchy(c, L)
    Nq = H.N[q]
    hq = 1.0 / (H.N[-1] - 1) * (1 << (L - 1 - q))
    n0 = c ** 3
    LU = np.zeros(n0 * n0)
    O.lib().orc_coarse_matrix(O.P(LU), c, hq * (1 << q))
    O.lib().orc_lu_factor(O.P(LU), n0)
    for X in (H, G_):
        X.u[q][:] = rnd(Nq ** 3, 1)  # must be wiped by the cycle
        X.d[q][:] = rnd(Nq ** 3, 2)
    O.lib().orc_set_threads(1)
    wn = O.lib().orc_vcycle(H.ptrs(H.u), H.ptrs(H.d), H.ptrs(H.r), hq, q, L, nu, Nq, O.P(LU))
    gn = C.c_double(0)
    check(M.lib().mg3d_host_vcycle(G_.ptrs(G_.u), G_.ptrs(G_.d), G_.ptrs(G_.r), hq, q, L, nu, Nq, P(LU), C.byref(gn),
                                   None, None))
    assert gn.value == pytest.approx(wn, rel=norm_rtol(Nq))
    for l in range(q + 1):
        assert np.array_equal(G_.u[l], H.u[l]) and np.array_equal(G_.d[l], H.d[l]) and np.array_equal(G_.r[l], H.r[l])


def test_full_size_513_bit_exact_against_oracle(monkeypatch):
    """BASELINE's headline size, three V(2,2) cycles in each of the three schedules -- one launch per leg (the default at
    this size since round 4), carried cycles (option legs = 0: the second cycle both continues the first and runs ahead
    into the third), plain (carry = 0 too): the whole 513^3 solution vector (1.08 GB) is bit-identical to the oracle's
    (OpenMP over the host cores; the oracle's grid values do not depend on the thread count) every time."""
    O.lib().orc_set_threads(min(16, os.cpu_count() or 1))
    want_norms, want_u, _, _ = O.run_problem(9, 7, 2, 3)
    with M.Solver(9, 7, 2) as s:
        s.setup_test_problem()
        s.timing_enable(3)
        got = s.vcycles(3)
        kt = {kn: n for (lvl, kn), (n, _) in s.kernel_times().items() if lvl == 6}
        assert kt.get("leg_up") == 3 and kt.get("leg_down") == 2 and "sweep4+norm" not in kt, kt
        s.timing_enable(0)
        u = s.download(MG3D_U, 6)
        assert_norm_exact(s, 6, got[-1])  # 133 M squares: the GPU's tree sum against the exactly rounded one, 1e-13
    assert np.array_equal(u, want_u)
    np.testing.assert_allclose(got, want_norms, rtol=norm_rtol(513))
    O.lib().orc_set_threads(1)
    with M.Solver(9, 7, 2) as s:
        s.set_option("legs", 0)
        s.setup_test_problem()
        s.timing_enable(3)
        carried = s.vcycles(3)
        assert {kn: n for (lvl, kn), (n, _) in s.kernel_times().items() if lvl == 6}.get("sweep4+norm") == 2
        s.timing_enable(0)
        assert np.array_equal(s.download(MG3D_U, 6), want_u)
        assert_norm_exact(s, 6, carried[-1])
    np.testing.assert_allclose(carried, want_norms, rtol=norm_rtol(513))
    with M.Solver(9, 7, 2) as s:
        s.set_option("legs", 0)
        s.set_option("carry", 0)
        s.setup_test_problem()
        plain = s.vcycles(3)
        assert np.array_equal(s.download(MG3D_U, 6), want_u)
    np.testing.assert_allclose(plain, want_norms, rtol=norm_rtol(513))


def test_unknown_sweep_shape_falls_back_to_the_default(monkeypatch):
    """MG3D_SWEEP_CFG naming a shape that was not compiled must not refuse the launch (a refused launch used to leave the
    buffers swapped and the norm at 0): the default shape runs, same bits."""
    c, L, nu = 5, 4, 2
    with M.Solver(c, L, nu) as s:
        s.setup_test_problem()
        want_n = s.vcycles(3)
        want_u = s.download(MG3D_U, L - 1)
    monkeypatch.setenv("MG3D_SWEEP_CFG", "9,9,9")
    with M.Solver(c, L, nu) as s:
        s.setup_test_problem()
        got_n = s.vcycles(3)
        assert np.array_equal(s.download(MG3D_U, L - 1), want_u)
    assert np.array_equal(got_n, want_n) and got_n[-1] > 0


@pytest.mark.parametrize("c,L,nu", [(9, 4, 2), (5, 5, 2), (3, 6, 3), (9, 3, 1)])
def test_single_workgroup_level_equals_the_generic_kernels(monkeypatch, c, L, nu):
    """The level above the coarsest one runs LDS-resident in one workgroup (csrc/mg3d_tiny.hip: two launches instead of
    five); MG3D_NO_TINY=1 keeps the plane-marching kernels.  Same bits on every level, and both equal the oracle."""
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("MG3D_NO_TINY", flag)
        with M.Solver(c, L, nu) as s:
            s.setup_test_problem()
            norms = s.vcycles(4)
            res.append((norms, [s.download(MG3D_U, l) for l in range(L)], [s.download(MG3D_D, l) for l in range(L - 1)]))
    assert np.array_equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert np.array_equal(a, b)
    want_norms, want_u, _, _ = O.run_problem(c, L, nu, 4)
    assert np.array_equal(res[0][1][-1], want_u)
    np.testing.assert_allclose(res[0][0], want_norms, rtol=norm_rtol((c - 1) * (1 << (L - 1)) + 1))


@pytest.mark.parametrize("dirty_r", [False, True])
@pytest.mark.parametrize("c,L,nu", [(9, 4, 2), (9, 3, 1), (5, 5, 2), (5, 3, 3), (3, 6, 2), (9, 5, 2)])
def test_bottom_of_the_cycle_in_one_launch(monkeypatch, c, L, nu, dirty_r):
    """Level 1 down, the direct solve on level 0 and level 1 up as ONE single-workgroup launch (tiny_cycle_kernel, with the
    reduced factor inside) against the three launches it replaces (MG3D_NO_TINY_CYCLE=1) and the oracle: u and d of every
    level.  dirty_r: non-zero values on the FACES of r on level 1, written from outside the cycle -- the restriction
    injects them into the coarse right-hand side (mg_3d.h:879-958), the reduced system no longer applies and the launch
    takes its full-system route (single-wave substitution); the reference sees exactly such values too."""
    N1 = 2 * c - 1
    rng = np.random.default_rng(c * 100 + L)
    r1 = rng.uniform(-1, 1, (N1, N1, N1))
    r1[1:-1, 1:-1, 1:-1] = 0.0
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("MG3D_NO_TINY_CYCLE", flag)
        with M.Solver(c, L, nu) as s:
            s.setup_test_problem()
            if dirty_r:
                s.upload(MG3D_R, 1, r1.reshape(-1))
            norms = s.vcycles(3)
            res.append((norms, [s.download(MG3D_U, l) for l in range(L)], [s.download(MG3D_D, l) for l in range(L - 1)]))
    assert np.array_equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert np.array_equal(a, b) and np.array_equal(np.signbit(a), np.signbit(b))
    H = O.Hierarchy(c, L)
    N, h = H.N[-1], 1.0 / (H.N[-1] - 1)
    LU = np.zeros(c ** 6)
    O.lib().orc_coarse_matrix(O.P(LU), c, h * (1 << (L - 1)))
    O.lib().orc_lu_factor(O.P(LU), c ** 3)
    O.lib().orc_fill_boundary(O.P(H.d[-1]), N, h)
    O.lib().orc_fill_boundary(O.P(H.u[-1]), N, h)
    if dirty_r:
        H.r[1][:] = r1.reshape(-1)
    O.lib().orc_set_threads(1)
    want = [O.lib().orc_vcycle(H.ptrs(H.u), H.ptrs(H.d), H.ptrs(H.r), h, L - 1, L, nu, N, O.P(LU)) for _ in range(3)]
    np.testing.assert_allclose(res[0][0], want, rtol=norm_rtol(N))
    for l in range(L):
        assert np.array_equal(res[0][1][l], H.u[l]), f"u level {l}"
    for l in range(L - 1):
        assert np.array_equal(res[0][2][l], H.d[l]), f"d level {l}"
    if dirty_r:
        assert np.any(H.d[0] != 0) and np.any(H.d[0].reshape(c, c, c)[0] != 0)  # the injected faces really are there


@pytest.mark.parametrize("c,L,calls", [(9, 5, (5,)), (5, 6, (1, 2, 3)), (3, 7, (4, 1)), (17, 4, (6,)), (9, 6, (3,))])
def test_carried_cycles_equal_the_plain_schedule_and_the_oracle(monkeypatch, c, L, calls):
    """mg3d_vcycles on a V(2,2) problem of more than 65^3 points: every cycle but the last of a call ends with the launch
    that also begins the next one (two post-smoothing passes, the norm tapped half-way, the next cycle's pre-smoothing
    passes -- whose first, red, pass is the identity behind the previous cycle's last red pass), and that next cycle's
    down-leg on the top level is ONE launch (one pass + residual + restriction).  Against the plain schedule
    (MG3D_NO_CARRY=1): u and d of every level bit for bit, also across several calls (a call never ends in the carried
    state); against the oracle: u of the finest level bit for bit, the history to summation-order accuracy."""
    res = []
    monkeypatch.setenv("MG3D_CARRY_MIN", "66")  # (default: from 257^3 up -- the last case; the 129^3 ones are the cheap ones)
    for flag in ("0", "1"):
        monkeypatch.setenv("MG3D_NO_CARRY", flag)
        with M.Solver(c, L, 2) as s:
            s.setup_test_problem()
            norms = []
            for k in calls:
                norms += list(s.vcycles(k))
            res.append((np.array(norms), [s.download(MG3D_U, l) for l in range(L)], [s.download(MG3D_D, l) for l in range(L - 1)]))
    N = (c - 1) * (1 << (L - 1)) + 1
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=norm_rtol(N))
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert np.array_equal(a, b) and np.array_equal(np.signbit(a), np.signbit(b))
    want_norms, want_u, _, _ = O.run_problem(c, L, 2, sum(calls))
    assert np.array_equal(res[0][1][-1], want_u)
    np.testing.assert_allclose(res[0][0], want_norms, rtol=norm_rtol(N))


def test_single_cycle_calls_run_ahead_and_other_calls_put_the_result_back(monkeypatch):
    """mg3d_vcycle (one cycle per call, the reference's solve loop) ends with the launch that also begins the next cycle.
    Whatever comes between two cycles -- reading u, a new right-hand side, a smoothing sweep, a residual, a norm, a
    changed sweep count, a raw device pointer -- must see and continue from the finished cycle's own u: the whole
    interleaved sequence, step by step, bit for bit against the plain schedule (MG3D_NO_CARRY=1)."""
    c, L = 9, 5
    N = (c - 1) * (1 << (L - 1)) + 1
    monkeypatch.setenv("MG3D_CARRY_MIN", "66")
    rng = np.random.default_rng(77)
    d2 = rng.uniform(-1, 1, N ** 3)
    logs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("MG3D_NO_CARRY", flag)
        log = []
        with M.Solver(c, L, 2) as s:
            top = L - 1
            s.setup_test_problem()
            s.timing_enable(1)
            log.append(s.vcycle())
            log.append(s.vcycle())
            log.append(s.download(MG3D_U, top))          # the finished cycle's own u: made from the launch's input
            log.append(s.vcycle())
            s.upload(MG3D_D, top, d2)                    # a new right-hand side: the passes run ahead are void
            log.append(s.vcycle())
            log.append(s.vcycle())
            s.smooth(top, 0, 1)
            log.append(s.vcycle())
            log.append(s.residual(top, True, True))
            log.append(s.download(MG3D_R, top))
            log.append(s.vcycle())
            log.append(s.l2norm(MG3D_U, top))
            log.append(s.vcycle())
            log += list(s.vcycles(3))                    # a batch call behind single ones
            log.append(s.vcycle())
            if flag == "0":                              # the switch thrown while a cycle is carried: it is finished first
                s.set_option("carry", 0)
            log.append(s.vcycle())
            if flag == "0":
                s.set_option("carry", 1)
            log.append(s.vcycle())
            log.append(s.vcycle(top - 1))                # a cycle from a lower level
            log.append(s.vcycle())
            s.fmg_initialize()
            log.append(s.vcycle())
            log.append(s.vcycle())
            kt = {kn: n for (lvl, kn), (n, _) in s.kernel_times().items() if lvl == top}
            log.append(s.download(MG3D_U, top))
            log.append(s.download(MG3D_U, top - 1))
            log.append(s.download(MG3D_D, top - 1))
        logs.append(log)
        if flag == "0":
            assert kt.get("sweep4+norm", 0) >= 12 and kt.get("sweep1+restrict", 0) >= 6, kt
        else:
            assert "sweep4+norm" not in kt and "sweep1+restrict" not in kt, kt
    assert len(logs[0]) == len(logs[1])
    for i, (a, b) in enumerate(zip(logs[0], logs[1])):
        if isinstance(a, np.ndarray):
            assert np.array_equal(a, b), f"step {i}"
        else:
            np.testing.assert_allclose(a, b, rtol=norm_rtol(N), err_msg=f"step {i}")


def test_carried_cycles_are_taken_and_counted(monkeypatch):
    """The kernel timers name the launches: with carrying, K cycles make K-1 tap launches and K-1 one-pass restricting
    launches on the top level, one ordinary start and one ordinary end."""
    monkeypatch.setenv("MG3D_CARRY_MIN", "66")
    with M.Solver(9, 5, 2) as s:
        s.setup_test_problem()
        s.timing_enable(1)
        s.vcycles(5)
        kt = {kn: n for (lvl, kn), (n, _) in s.kernel_times().items() if lvl == 4}
        s.timing_enable(0)
    monkeypatch.delenv("MG3D_CARRY_MIN")
    with M.Solver(9, 5, 2) as s:  # default threshold: a 129^3 problem runs the plain schedule
        s.setup_test_problem()
        s.timing_enable(1)
        s.vcycles(3)
        assert not any(kn == "sweep4+norm" for (lvl, kn) in s.kernel_times())
    assert kt.get("sweep4+norm") == 4 and kt.get("sweep1+restrict") == 4, kt
    assert kt.get("sweep4") == 1 and kt.get("sweep2+residual") == 1 and kt.get("sweep2") == 5 and kt.get("residual") == 1, kt


@pytest.mark.parametrize("c,L", [(9, 4), (5, 5), (3, 6), (9, 6)])
def test_one_sweep_down_leg_two_launches_equal_the_fused_shape(monkeypatch, c, L):
    """V(1,1): two colour passes + residual + restriction run as two launches (MG3D_FUSE_RST2=0; the default below 130
    points per side) or as one (=1; the default from there up: no scratch since the restriction parks its r pairs in
    LDS).  Same bits, and both equal the oracle."""
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("MG3D_FUSE_RST2", flag)
        with M.Solver(c, L, 1) as s:
            s.setup_test_problem()
            norms = s.vcycles(4)
            res.append((norms, [s.download(MG3D_U, l) for l in range(L)], [s.download(MG3D_D, l) for l in range(L - 1)]))
    assert np.array_equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert np.array_equal(a, b)
    want_norms, want_u, _, _ = O.run_problem(c, L, 1, 4)
    assert np.array_equal(res[0][1][-1], want_u)
    np.testing.assert_allclose(res[0][0], want_norms, rtol=norm_rtol((c - 1) * (1 << (L - 1)) + 1))


def test_measured_chunk_length_changes_no_bit(monkeypatch):
    """The first launch of a sweep shape on a level times a few chunk lengths and keeps the fastest (MG3D_SWEEP_TUNE=0:
    the cost model's choice; MG3D_SWEEP_CI: a fixed one).  Chunking is a work distribution only: same grid values (the norm is a sum of per-block partial sums).""" 
    c, L, nu = 9, 5, 2  # 129^3: large enough for the measurement to run
    res = []
    for env in ({"MG3D_SWEEP_TUNE": "0"}, {}, {"MG3D_SWEEP_CI": "5"}, {"MG3D_SWEEP_CI": "23", "MG3D_SWEEP_TUNE": "0"}, {}):
        for k in ("MG3D_SWEEP_TUNE", "MG3D_SWEEP_CI"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with M.Solver(c, L, nu) as s:
            s.setup_test_problem()
            norms = s.vcycles(3)
            res.append((norms, s.download(MG3D_U, L - 1), s.download(MG3D_D, L - 2)))
    for r in res[1:]:
        assert np.array_equal(r[1], res[0][1]) and np.array_equal(r[2], res[0][2])
        np.testing.assert_allclose(r[0], res[0][0], rtol=1e-13)  # one partial sum per block: the grouping differs
    # launches that form a norm are not measured: with or without the measurement the norms are the same bits
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[1][0], res[4][0])
    want_norms, want_u, _, _ = O.run_problem(c, L, nu, 3)
    assert np.array_equal(res[0][1], want_u)


@pytest.mark.parametrize("c,L,nu", [(9, 4, 2), (5, 5, 2), (3, 6, 2), (7, 4, 2), (9, 5, 2), (6, 4, 3)])
@pytest.mark.parametrize("small,legs", [("0", "0"), ("65", "0"), ("0", "65"), ("1000", "1000")])
def test_small_level_policy_changes_no_bit(monkeypatch, c, L, nu, small, legs):
    """Levels of at most 65^3 points run the two-rows-per-thread shapes with two planes in flight (MG3D_SMALL_MAX); for
    V(2,2) each leg of the cycle can run as ONE launch (four passes + residual + restriction; prolongation + four passes:
    MG3D_FUSE_LEG_MAX, the default up to round 2, opt-in since).  Either policy off, both off, or both forced onto every
    level: the same bits as the default, and the default equals the oracle."""
    res = []
    for env in ({}, {"MG3D_SMALL_MAX": small, "MG3D_FUSE_LEG_MAX": legs}):
        for k in ("MG3D_SMALL_MAX", "MG3D_FUSE_LEG_MAX"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with M.Solver(c, L, nu) as s:
            s.setup_test_problem()
            norms = s.vcycles(3)
            res.append((norms, [s.download(MG3D_U, l) for l in range(L)], [s.download(MG3D_D, l) for l in range(L - 1)]))
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-13)
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert np.array_equal(a, b)
    want_norms, want_u, _, _ = O.run_problem(c, L, nu, 3)
    assert np.array_equal(res[0][1][-1], want_u)


def test_two_host_threads_solve_concurrently():
    """Two contexts driven from two host threads at once (ctypes releases the GIL): the sweep launcher's first-use
    measurement and its remembered choices are process-wide and mutex-protected; both solves give the oracle's bits."""
    import threading

    c, L, nu = 9, 5, 2  # 129^3: the levels large enough for the measurement to run
    want_norms, want_u, _, _ = O.run_problem(c, L, nu, 3)
    out = {}

    def work(tag):
        with M.Solver(c, L, nu) as s:
            s.setup_test_problem()
            n = s.vcycles(3)
            out[tag] = (n, s.download(MG3D_U, L - 1))

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive()
    for t in range(2):
        assert np.array_equal(out[t][1], want_u)
        np.testing.assert_allclose(out[t][0], want_norms, rtol=norm_rtol((c - 1) * (1 << (L - 1)) + 1))


def test_options_are_per_context_and_set_through_the_api(monkeypatch):
    """mg3d_ctx_set_option / _get_option: the launch policy belongs to a context -- defaults, the environment as an override
    when the context is created, the API afterwards.  Two contexts of one process differ; the environment is not read
    again after creation; unknown keys are refused; the grid values do not depend on any of it."""
    monkeypatch.delenv("MG3D_NO_CARRY", raising=False)
    monkeypatch.setenv("MG3D_CARRY_MIN", "66")
    with M.Solver(9, 5, 2) as a, M.Solver(9, 5, 2) as b:
        assert a.get_option("carry") == 1 and a.get_option("carry_min") == 66
        assert a.get_option("legs") == 1 and a.get_option("legs_min") == 160  # (129^3 < 160: the carried cycles run here)
        assert set(a.options()) >= {"carry", "carry_min", "legs", "legs_min", "tiny", "tiny_cycle", "lu_reduced", "fuse_rst2",
                                    "small_max", "fuse_leg_max", "fuse_up_max", "sweep_tune", "sweep_ci"}
        monkeypatch.setenv("MG3D_NO_CARRY", "1")  # after creation: nobody reads it any more
        b.set_option("carry", 0)
        b.set_option("small_max", 0)
        b.set_option("tiny_cycle", 0)
        for s in (a, b):
            s.setup_test_problem()
            s.timing_enable(1)
        na, nb = a.vcycles(4), b.vcycles(4)
        ka = {kn for (lvl, kn), (n, _) in a.kernel_times().items() if lvl == 4}
        kb = {kn for (lvl, kn), (n, _) in b.kernel_times().items() if lvl == 4}
        assert "sweep4+norm" in ka and "sweep4+norm" not in kb, (ka, kb)
        for l in range(5):
            assert np.array_equal(a.download(MG3D_U, l), b.download(MG3D_U, l)), l
        np.testing.assert_allclose(na, nb, rtol=1e-12)
        with pytest.raises(M.Mg3dError):
            a.set_option("no_such_option", 1)
        with pytest.raises(M.Mg3dError):
            a.get_option("no_such_option")
    with M.Solver(9, 5, 2) as c:  # created with MG3D_NO_CARRY=1 in the environment: the override is taken at creation
        assert c.get_option("carry") == 0
