"""The numpy slab operators (tests/_slab_numpy.py) equal the oracle on a full cube (ig0 = 0, ni = N): this pins the
helper the multi-process schedule test relies on."""
import numpy as np
import pytest

import _oracle as O
import _slab_numpy as S


@pytest.mark.parametrize("N", [5, 9, 17])
def test_numpy_operators_match_oracle(N):
    rng = np.random.default_rng(N)
    h = 1.0 / (N - 1)
    u, d = rng.uniform(-1, 1, (N, N, N)), rng.uniform(-1, 1, (N, N, N))
    want = u.copy().reshape(-1)
    O.lib().orc_set_threads(1)
    O.lib().orc_pre_smooth(O.P(want), O.P(d.reshape(-1)), N, h, 2)
    got = u.copy()
    S.smooth(got, d, h, 2, False, 0, N)
    assert np.array_equal(got.reshape(-1), want)
    O.lib().orc_post_smooth(O.P(want), O.P(d.reshape(-1)), N, h, 1)
    S.smooth(got, d, h, 1, True, 0, N)
    assert np.array_equal(got.reshape(-1), want)
    r_want, r_got = np.zeros(N ** 3), np.zeros((N, N, N))
    nrm = O.lib().orc_residual(O.P(want), O.P(d.reshape(-1)), N, h, O.P(r_want))
    ss = S.residual(got, d, h, r_got, 0, N)
    assert np.array_equal(r_got.reshape(-1), r_want)
    assert np.sqrt(ss) == pytest.approx(nrm, rel=1e-13)
    Nc = (N + 1) // 2
    rr = rng.uniform(-1, 1, (N, N, N))
    dc_want, dc_got = np.zeros(Nc ** 3), np.zeros((Nc, Nc, Nc))
    O.lib().orc_restrict(O.P(rr.reshape(-1)), N, O.P(dc_want), Nc)
    S.restrict_planes(rr, 0, N, dc_got, 0, Nc, 0, Nc)
    assert np.array_equal(dc_got.reshape(-1), dc_want)
    ec = rng.uniform(-1, 1, (Nc, Nc, Nc))
    ef_want = u.copy().reshape(-1)
    ef_got = u.copy()
    O.lib().orc_prolong(O.P(ec.reshape(-1)), Nc, O.P(ef_want), N)
    S.prolong_planes(ec, 0, Nc, ef_got, 0, N, 0, N)
    assert np.array_equal(ef_got.reshape(-1), ef_want)
