"""Property tests of the HIP operators against the oracle on random sizes and data (hypothesis): ragged sizes
(even N, N not of the form 2^k+1), every smoothing count, random fields including special values' neighbours.
Bit-exact comparison of every array; norms to the summation-order tolerance."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import _oracle as O
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import P, check

pytestmark = pytest.mark.gpu
SET = dict(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])


def field(n, seed, scale):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, n) * scale
    a[rng.integers(0, n, max(1, n // 50))] = 0.0  # exact zeros (signed-zero paths)
    return a


@settings(**SET)
@given(N=st.integers(3, 41), iters=st.integers(0, 5), post=st.booleans(), seed=st.integers(0, 2 ** 31),
       scale=st.sampled_from([1e-300, 1e-8, 1.0, 1e6, 1e150]))
def test_smoother_and_residual_random(N, iters, post, seed, scale):
    h = 1.0 / (N - 1)
    v, d = field(N ** 3, seed, scale), field(N ** 3, seed + 1, scale)
    want, got = v.copy(), v.copy()
    (O.lib().orc_post_smooth if post else O.lib().orc_pre_smooth)(O.P(want), O.P(d), N, h, iters)
    check(M.lib().mg3d_host_smooth(P(got), P(d), N, h, iters, int(post)))
    assert np.array_equal(got, want)
    rw, rg = np.zeros(N ** 3), np.zeros(N ** 3)
    O.lib().orc_set_threads(1)
    wn = O.lib().orc_residual(O.P(want), O.P(d), N, h, O.P(rw))
    gn = C.c_double(0)
    check(M.lib().mg3d_host_residual(P(got), P(d), N, h, P(rg), C.byref(gn)))
    assert np.array_equal(rg, rw)
    if np.isfinite(wn):
        assert gn.value == pytest.approx(wn, rel=1e-11)


@settings(**SET)
@given(Nc=st.integers(2, 21), seed=st.integers(0, 2 ** 31), scale=st.sampled_from([1e-200, 1.0, 1e100]))
def test_grid_transfer_random(Nc, seed, scale):
    Nf = 2 * Nc - 1
    r = field(Nf ** 3, seed, scale)
    want, got = np.zeros(Nc ** 3), np.ones(Nc ** 3)
    O.lib().orc_restrict(O.P(r), Nf, O.P(want), Nc)
    check(M.lib().mg3d_host_restrict(P(r), Nf, P(got), Nc))
    assert np.array_equal(got, want)
    ec, ef = field(Nc ** 3, seed + 2, scale), field(Nf ** 3, seed + 3, scale)
    w2, g2 = ef.copy(), ef.copy()
    O.lib().orc_prolong(O.P(ec), Nc, O.P(w2), Nf)
    check(M.lib().mg3d_host_prolong(P(ec), Nc, P(g2), Nf))
    assert np.array_equal(g2, w2)


@settings(max_examples=12, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(c=st.sampled_from([3, 5, 9]), L=st.integers(2, 4), nu=st.integers(0, 3), seed=st.integers(0, 2 ** 31),
       keep=st.booleans())
def test_vcycle_random_rhs(c, L, nu, seed, keep):
    """A V-cycle on a random right-hand side and random initial guess (not the harmonic test problem)."""
    H = O.Hierarchy(c, L)
    N, h = H.N[-1], 1.0 / (H.N[-1] - 1)
    H.u[-1][:] = field(N ** 3, seed, 1.0)
    H.d[-1][:] = field(N ** 3, seed + 1, 1e3)
    u0, d0 = H.u[-1].copy(), H.d[-1].copy()
    n0 = c ** 3
    LU = np.zeros(n0 * n0)
    O.lib().orc_coarse_matrix(O.P(LU), c, h * (1 << (L - 1)))
    O.lib().orc_lu_factor(O.P(LU), n0)
    O.lib().orc_set_threads(1)
    want = [O.lib().orc_vcycle(H.ptrs(H.u), H.ptrs(H.d), H.ptrs(H.r), h, L - 1, L, nu, N, O.P(LU)) for _ in range(2)]
    with M.Solver(c, L, nu) as s:
        s.set_keep_residual(keep)
        s.get_details()
        s.upload(0, L - 1, u0)
        s.upload(1, L - 1, d0)
        got = s.vcycles(2)
        assert np.array_equal(s.download(0, L - 1), H.u[-1])
        for l in range(L - 1):
            assert np.array_equal(s.download(1, l), H.d[l])
    np.testing.assert_allclose(got, want, rtol=1e-11)


@pytest.mark.parametrize("N,iters,post", [(66, 2, False), (131, 2, True), (200, 1, False), (258, 2, True), (301, 2, False),
                                          (301, 3, True)])
def test_ragged_multi_tile_sizes(N, iters, post):
    """Sizes that are neither 2^k+1 nor a single tile: several j- and k-tiles with ragged last tiles, several i-chunks
    (equal, with a short tail, one or many rounds of blocks), both k-tilings, the XCD grouping of tile columns.  Smoother
    passes (4-pass and 2-pass launches) and the stored residual, bit for bit; the norm to the summation order."""
    h = 1.0 / (N - 1)
    v, d = field(N ** 3, 1000 + N, 1.0), field(N ** 3, 2000 + N, 1e2)
    want, got = v.copy(), v.copy()
    O.lib().orc_set_threads(O.lib().orc_max_threads())
    (O.lib().orc_post_smooth if post else O.lib().orc_pre_smooth)(O.P(want), O.P(d), N, h, iters)
    check(M.lib().mg3d_host_smooth(P(got), P(d), N, h, iters, int(post)))
    assert np.array_equal(got, want)
    rw, rg = np.zeros(N ** 3), np.zeros(N ** 3)
    wn = O.lib().orc_residual(O.P(want), O.P(d), N, h, O.P(rw))
    gn = C.c_double(0)
    check(M.lib().mg3d_host_residual(P(got), P(d), N, h, P(rg), C.byref(gn)))
    assert np.array_equal(rg, rw)
    assert gn.value == pytest.approx(wn, rel=max(1e-11, 0.5 * (N - 2) ** 3 * 2.0 ** -53))
    O.lib().orc_set_threads(1)
