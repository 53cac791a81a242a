"""Pins the oracle (oracle/mg3d_oracle.c) bit-for-bit to golden vectors produced by the compiled,
unmodified reference (generator: oracle/gen_golden.py) and to the reference's printed known-answer
histories.  CPU only."""
import hashlib
import os

import numpy as np
import pytest

import _oracle as O

G = np.load(os.path.join(O.GOLDEN, "operators.npz"))
V = np.load(os.path.join(O.GOLDEN, "vcycle.npz"))


@pytest.fixture(autouse=True)
def _one_thread():
    # reduction order of the norm equals the reference's sequential sum only with one thread
    O.lib().orc_set_threads(1)


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


@pytest.mark.parametrize("N", [5, 9, 17, 33])
@pytest.mark.parametrize("name,post,it", [("pre1", 0, 1), ("pre2", 0, 2), ("post1", 1, 1), ("post3", 1, 3)])
def test_smoother_bit_exact(N, name, post, it):
    v = G[f"sm_v0_{N}"].copy()
    d = G[f"sm_d0_{N}"]
    (O.lib().orc_post_smooth if post else O.lib().orc_pre_smooth)(O.P(v), O.P(d), N, 1.0 / (N - 1), it)
    assert np.array_equal(v, G[f"sm_{name}_{N}"])


@pytest.mark.parametrize("N", [5, 9, 17, 33])
def test_residual_bit_exact(N):
    v, d = G[f"sm_v0_{N}"], G[f"sm_d0_{N}"]
    res = np.zeros(N ** 3)
    nrm = O.lib().orc_residual(O.P(v), O.P(d), N, 1.0 / (N - 1), O.P(res))
    assert np.array_equal(res, G[f"res_r_{N}"])
    # boundary of res is never written (mg_3d.h:824-825)
    r3 = res.reshape(N, N, N)
    assert not r3[0].any() and not r3[:, 0].any() and not r3[:, :, 0].any()
    assert nrm == G[f"res_norm_{N}"][0]
    assert O.lib().orc_residual(O.P(v), O.P(d), N, 1.0 / (N - 1), None) == G[f"res_norm_{N}"][1]
    assert O.lib().orc_l2norm(O.P(d), N ** 3) == G[f"l2_{N}"][0]


@pytest.mark.parametrize("Nc", [3, 5, 9, 17])
def test_restrict_bit_exact(Nc):
    Nf = 2 * Nc - 1
    dc = np.full(Nc ** 3, 7.0)
    O.lib().orc_restrict(O.P(G[f"rs_r_{Nf}"]), Nf, O.P(dc), Nc)
    assert np.array_equal(dc, G[f"rs_dc_{Nc}"])


@pytest.mark.parametrize("Nc", [3, 5, 9, 17])
def test_prolong_bit_exact(Nc):
    Nf = 2 * Nc - 1
    ef = G[f"pr_ef0_{Nf}"].copy()
    O.lib().orc_prolong(O.P(G[f"pr_ec_{Nc}"]), Nc, O.P(ef), Nf)
    assert np.array_equal(ef, G[f"pr_ef_{Nf}"])


@pytest.mark.parametrize("N", [3, 5, 9])
def test_boundary_fill_bit_exact(N):
    v = G[f"bc_v0_{N}"].copy()
    O.lib().orc_fill_boundary(O.P(v), N, float(G[f"bc_h_{N}"][0]))
    assert np.array_equal(v, G[f"bc_v_{N}"])


@pytest.mark.parametrize("N", [3, 5])
def test_coarse_matrix_and_lu_bit_exact(N):
    n = N ** 3
    A = np.zeros(n * n)
    O.lib().orc_coarse_matrix(O.P(A), N, float(G[f"cm_h_{N}"][0]))
    assert np.array_equal(A, G[f"cm_A_{N}"])
    O.lib().orc_lu_factor(O.P(A), n)
    assert np.array_equal(A, G[f"lu_LU_{N}"])
    x = np.zeros(n)
    O.lib().orc_lu_solve(O.P(A), n, O.P(G[f"lu_b_{N}"]), O.P(x))
    assert np.array_equal(x, G[f"lu_x_{N}"])


@pytest.mark.parametrize("N", [3, 5, 9])
def test_banded_lu_factor_equals_the_dense_sweep_byte_for_byte(N):
    """orc_lu_factor_banded (used for the coarse grids the dense O(n^3) sweep cannot reach, c = 17 and 33) against the
    pinned orc_lu_factor: same bytes, signed zeros included; and against the golden factors of the compiled reference."""
    n = N ** 3
    A = np.zeros(n * n)
    O.lib().orc_coarse_matrix(O.P(A), N, 1.0 / (N - 1))
    B = A.copy()
    O.lib().orc_lu_factor(O.P(A), n)
    O.lib().orc_set_threads(3)
    O.lib().orc_lu_factor_banded(O.P(B), n)
    assert A.tobytes() == B.tobytes()
    if N == 9:
        assert np.array_equal(sha(B), G["lu_sha_9"])


@pytest.mark.parametrize("n,bl,bu,seed", [(60, 5, 9, 1), (97, 20, 3, 2), (130, 1, 1, 3), (64, 63, 63, 4), (75, 0, 7, 5)])
def test_banded_lu_factor_random_bands(n, bl, bu, seed):
    """random diagonally dominant band matrices with holes inside the band (exact zeros that fill in)"""
    rng = np.random.default_rng(seed)
    A = np.zeros((n, n))
    for i in range(n):
        for j in range(max(0, i - bl), min(n, i + bu + 1)):
            if i == j or rng.uniform() < 0.6:
                A[i, j] = rng.uniform(-1, 1)
        A[i, i] = (-1) ** i * (np.abs(A[i]).sum() + 1.0)
    A = A.reshape(-1).copy()
    B = A.copy()
    O.lib().orc_lu_factor(O.P(A), n)
    O.lib().orc_lu_factor_banded(O.P(B), n)
    assert A.tobytes() == B.tobytes()


def test_lu_c9_bit_exact():
    n = 729
    A = np.zeros(n * n)
    O.lib().orc_coarse_matrix(O.P(A), 9, 0.125)
    O.lib().orc_lu_factor(O.P(A), n)
    assert np.array_equal(sha(A), G["lu_sha_9"])
    x = np.zeros(n)
    O.lib().orc_lu_solve(O.P(A), n, O.P(G["lu_b_9"]), O.P(x))
    assert np.array_equal(x, G["lu_x_9"])


@pytest.mark.parametrize("c,L,nu", [(3, 3, 1), (3, 5, 2), (5, 3, 3), (9, 2, 2), (5, 5, 2), (9, 5, 2)])
def test_vcycle_history_bit_exact(c, L, nu):
    key = f"{c}_{L}_{nu}"
    ref = V[f"norms_{key}"]
    norms, u, init, _ = O.run_problem(c, L, nu, len(ref))
    assert init == V[f"init_{key}"][0]
    # the reference's driver squares the returned sqrt and takes the sqrt again (test_mg_3d.c:53-59):
    # at most 1 ulp away from the plain sqrt the oracle returns
    np.testing.assert_allclose(norms, ref, rtol=4e-16, atol=0)
    assert np.array_equal(sha(u), V[f"usha_{key}"])  # the solution vector is bit-identical
    if f"u_{key}" in V:
        assert np.array_equal(u, V[f"u_{key}"])
    else:
        assert np.array_equal(u[::97], V[f"usample_{key}"])


@pytest.mark.parametrize("c,L,nu", [(5, 5, 2), (3, 4, 2)])
def test_dirichlet_driver_history_bit_exact(c, L, nu):
    key = f"dir_{c}_{L}_{nu}"
    ref = V[f"norms_{key}"]
    norms, u, init, _ = O.run_problem(c, L, nu, len(ref), mode=1)
    assert init == V[f"init_{key}"][0]
    assert np.array_equal(norms, ref)
    assert np.array_equal(sha(u), V[f"usha_{key}"])


# Known-answer histories printed by the unmodified reference (6 printed digits)
KNOWN = {
    (5, 5, 2): [74651.9, 9198.35, 1219.39, 170.177, 24.6618, 3.68103, 0.563252, 0.0880884, 0.0140466,
                0.00227868, 0.000375223, 6.25855e-05, 1.05534e-05, 1.79591e-06, 3.0789e-07],
    (9, 5, 2): [600893, 73400.9, 9566.66, 1305, 183.942, 26.5851, 3.92421, 0.590481, 0.0904885, 0.014113,
                0.00223841, 0.000360659, 5.89564e-05, 9.7633e-06, 1.63505e-06],
}
KNOWN_DIRICHLET_5_5_2 = [74831.4, 9392.75, 1372.13, 265.208, 69.895, 21.3226, 6.76706, 2.16709, 0.695417, 0.223269]


@pytest.mark.parametrize("args", sorted(KNOWN))
def test_known_answer_histories(args):
    norms, u, init, _ = O.run_problem(*args, len(KNOWN[args]))
    for got, want in zip(norms, KNOWN[args]):
        assert float(f"{got:.6g}") == pytest.approx(want, rel=1e-12)
    # stopping rule of test_mg_3d.c:31,40: first norm <= 1e-8 * ||d|| is reached at the last listed cycle
    assert norms[-1] <= 1e-8 * init < norms[-2]
    # final error against the analytic solution
    c, L, _ = args
    N = (c - 1) * (1 << (L - 1)) + 1
    g = np.arange(N) / (N - 1)
    exact = g[:, None, None] ** 2 - 2 * g[None, :, None] ** 2 + g[None, None, :] ** 2
    err = np.sqrt(((u.reshape(N, N, N) - exact) ** 2).sum())
    assert err == pytest.approx({(5, 5, 2): 1.60434e-09, (9, 5, 2): 1.85423e-09}[args], rel=2e-3)


def test_known_answer_dirichlet_driver():
    norms, _, _, _ = O.run_problem(5, 5, 2, 10, mode=1)
    for got, want in zip(norms, KNOWN_DIRICHLET_5_5_2):
        assert float(f"{got:.6g}") == pytest.approx(want, rel=1e-12)


def test_thread_count_invariance():
    # red_black_gs_scalability.txt:6-7 -- results do not depend on the thread count (6 digits)
    O.lib().orc_set_threads(4)
    n4, u4, _, _ = O.run_problem(5, 4, 2, 6)
    O.lib().orc_set_threads(1)
    n1, u1, _, _ = O.run_problem(5, 4, 2, 6)
    assert np.array_equal(u1, u4)
    np.testing.assert_allclose(n1, n4, rtol=1e-13)


@pytest.mark.parametrize("c,L,nu", [(5, 4, 2), (3, 5, 1)])
def test_fmg_initialize_bit_exact(c, L, nu):
    """SolverFMGInitialize (spec mg_dirichlet_analytic.c:771-806) replayed with the reference's own operators."""
    key = f"fmg_{c}_{L}_{nu}"
    H = O.Hierarchy(c, L)
    N, h = H.N[-1], 1.0 / (H.N[-1] - 1)
    n0 = c ** 3
    LU = np.zeros(n0 * n0)
    O.lib().orc_coarse_matrix(O.P(LU), c, h * (1 << (L - 1)))
    O.lib().orc_lu_factor(O.P(LU), n0)
    O.lib().orc_fill_boundary(O.P(H.d[-1]), N, h)
    O.lib().orc_fill_boundary(O.P(H.u[-1]), N, h)
    O.lib().orc_fmg_initialize(H.ptrs(H.u), H.ptrs(H.d), H.ptrs(H.r), c, L, nu, 1.0, O.P(LU))
    assert np.array_equal(H.u[-1], V[f"u0_{key}"])
    norms = [O.lib().orc_vcycle(H.ptrs(H.u), H.ptrs(H.d), H.ptrs(H.r), h, L - 1, L, nu, N, O.P(LU)) for _ in range(3)]
    assert np.array_equal(np.array(norms), V[f"norms_{key}"])
    assert np.array_equal(H.u[-1], V[f"u_{key}"])


def test_f32_restatement_is_consistent_with_the_pinned_double_oracle():
    """oracle/mg3d_oracle_f32.c is parity-unpinned (no reference implementation of fp32 / Jacobi / F-cycle).  What
    can be checked on the CPU: with omega = 1 and one sweep its Jacobi update is the binary32 rounding of the
    double seven-point average; its transfer operators are the binary32 versions of the pinned double ones; its
    F-cycle + V-cycles reach the double solution to binary32 accuracy."""
    import ctypes as C
    lib = O.lib()
    N, Nc = 9, 5
    rng = np.random.default_rng(3)
    u64, d64 = rng.uniform(-1, 1, N ** 3), rng.uniform(-1, 1, N ** 3)
    u32, d32 = u64.astype(np.float32), d64.astype(np.float32)
    h = 1.0 / (N - 1)
    out = np.zeros_like(u32)
    lib.orc32_jacobi(O.PF(u32), O.PF(d32), O.PF(out), N, C.c_float(h), C.c_float(1.0))
    U = u32.astype(np.float64).reshape(N, N, N)
    D = d32.astype(np.float64).reshape(N, N, N)
    avg = (U[:-2, 1:-1, 1:-1] + U[2:, 1:-1, 1:-1] + U[1:-1, :-2, 1:-1] + U[1:-1, 2:, 1:-1] + U[1:-1, 1:-1, :-2]
           + U[1:-1, 1:-1, 2:] - h * h * D[1:-1, 1:-1, 1:-1]) / 6
    got = out.reshape(N, N, N)
    assert np.abs(got[1:-1, 1:-1, 1:-1] - avg).max() < 4e-7
    assert np.array_equal(got[0], u32.reshape(N, N, N)[0])  # boundary copied
    # restriction / prolongation against the double oracle on binary32-representable data
    r64 = u32.astype(np.float64)
    dc64, dc32 = np.zeros(Nc ** 3), np.zeros(Nc ** 3, dtype=np.float32)
    lib.orc_restrict(O.P(r64), N, O.P(dc64), Nc)
    lib.orc32_restrict(O.PF(u32), N, O.PF(dc32), Nc)
    assert np.abs(dc32 - dc64).max() < 1e-6
    ef64, ef32 = d32.astype(np.float64), d32.copy()
    ec32 = rng.uniform(-1, 1, Nc ** 3).astype(np.float32)
    lib.orc_prolong(O.P(ec32.astype(np.float64)), Nc, O.P(ef64), N)
    lib.orc32_prolong(O.PF(ec32), Nc, O.PF(ef32), N)
    assert np.abs(ef32 - ef64).max() < 1e-6
    # whole solve: 33^3, F-cycle start + 12 V(2,2) cycles with omega = 6/7
    c, L = 5, 4
    Nf = 33
    norms, uf = np.zeros(12), np.zeros(Nf ** 3, dtype=np.float32)
    lib.orc32_run_problem(c, L, 2, 6.0 / 7.0, 12, 1, O.P(norms), O.PF(uf))
    _, u_ref, _, _ = O.run_problem(c, L, 2, 12)
    assert np.abs(uf - u_ref).max() < 1e-4
    assert norms[-1] < 1.0
