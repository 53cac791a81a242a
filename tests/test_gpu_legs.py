"""One launch per leg on the top level (MG3D_LEGS=1; csrc/mg3d_ctx.hip "two launches per level", round 4): the up-leg is
prolongation + four passes in ONE launch, the down-leg three (behind another cycle) or four passes + residual + restriction
in ONE launch, and the residual norm has no stage of its own -- its red half falls out of the up-leg's last pass, its black
half out of the next down-leg's first.  Same bar as everywhere: every grid value bit-identical to the plain schedule and to
the oracle; the norm against the exactly rounded sum."""
import numpy as np
import pytest

import _oracle as O
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_R, MG3D_U

from test_gpu_parity import EXACT_NORM_RTOL, assert_norm_exact, norm_rtol

pytestmark = pytest.mark.gpu


def _legs(s, on):
    """the schedule under test through the options API (mg3d_ctx_set_option), from 129^3 up (default: 130 points a side)"""
    s.set_option("legs_min", 66)
    s.set_option("legs", 1 if on else 0)
    s.set_option("carry", 0)  # (the other schedule in these tests is the plain one: one launch per operator group)
    return s


@pytest.mark.parametrize("c,L,calls", [(9, 5, (5,)), (5, 6, (1, 2, 3)), (3, 7, (4, 1)), (17, 4, (6,)), (9, 6, (3,)), (11, 5, (3,))])
def test_one_launch_per_leg_equals_the_plain_schedule_and_the_oracle(monkeypatch, c, L, calls):
    """mg3d_vcycles: u and d of every level after the calls, bit for bit against the plain schedule (one launch per
    operator group, MG3D_NO_CARRY=1), u of the finest level against the oracle, every norm of the history against the plain
    schedule's and the oracle's, the last one against the exactly rounded sum.  (11, 5): 161^3, not a power of two + 1 in the
    tile arithmetic; (17, 4): a 17^3 coarsest grid.)"""
    res = []
    for on in (True, False):
        with M.Solver(c, L, 2) as s:
            _legs(s, on)
            s.setup_test_problem()
            norms = []
            for k in calls:
                norms += list(s.vcycles(k))
            if on:
                assert_norm_exact(s, L - 1, norms[-1])
            res.append((np.array(norms), [s.download(MG3D_U, l) for l in range(L)], [s.download(MG3D_D, l) for l in range(L - 1)]))
    N = (c - 1) * (1 << (L - 1)) + 1
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-12)
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert np.array_equal(a, b) and np.array_equal(np.signbit(a), np.signbit(b))
    want_norms, want_u, _, _ = O.run_problem(c, L, 2, sum(calls))
    assert np.array_equal(res[0][1][-1], want_u)
    np.testing.assert_allclose(res[0][0], want_norms, rtol=norm_rtol(N))


def test_every_norm_of_a_batch_is_the_exactly_rounded_one(monkeypatch):
    """The two halves of a norm come from two launches (the up-leg of cycle n, the down-leg of cycle n + 1) and are folded
    by a third: every cycle's value, not only the last, against the exactly rounded sum over the oracle's residual field
    of that cycle's u (taken from a second solver stepped one cycle at a time)."""
    c, L, K = 9, 5, 5
    with M.Solver(c, L, 2) as batch, M.Solver(c, L, 2) as step:
        _legs(batch, True)
        _legs(step, True)
        batch.setup_test_problem()
        step.setup_test_problem()
        got = batch.vcycles(K)
        d = step.download(MG3D_D, L - 1)
        for k in range(K):
            one = step.vcycle()
            u = step.download(MG3D_U, L - 1)  # (puts the finished cycle's u back: the down-leg run ahead is dropped)
            want = O.exact_residual_norm(u, d, step.level_n(L - 1), step.level_h(L - 1))
            assert got[k] == pytest.approx(want, rel=EXACT_NORM_RTOL), k
            assert one == pytest.approx(want, rel=EXACT_NORM_RTOL), k


def test_single_cycle_calls_run_the_next_down_leg_ahead_and_other_calls_swap_it_back(monkeypatch):
    """mg3d_vcycle (one cycle per call, the reference's solve loop): the call ends with the NEXT cycle's down-leg, run
    into the alt buffers so that the norm is complete.  Whatever comes between two cycles -- reading u or the coarser
    right-hand side, a new right-hand side, a smoothing sweep, a residual, a norm, a cycle from a lower level, FMG, a batch
    call, the switch thrown -- must see and continue from the finished cycle's own state: the interleaved sequence step
    by step against the plain schedule."""
    c, L = 9, 5
    N = (c - 1) * (1 << (L - 1)) + 1
    d2 = np.random.default_rng(77).uniform(-1, 1, N ** 3)
    logs = []
    for on in (True, False):
        log = []
        with M.Solver(c, L, 2) as s:
            _legs(s, on)
            top = L - 1
            s.setup_test_problem()
            s.timing_enable(1)
            log.append(s.vcycle())
            log.append(s.vcycle())
            log.append(s.download(MG3D_U, top))
            log.append(s.download(MG3D_D, top - 1))       # the coarser right-hand side of the FINISHED cycle
            log.append(s.vcycle())
            s.upload(MG3D_D, top, d2)                     # a new right-hand side: the down-leg run ahead is void
            log.append(s.vcycle())
            log.append(s.vcycle())
            s.smooth(top, 0, 1)
            log.append(s.vcycle())
            log.append(s.residual(top, True, True))
            log.append(s.download(MG3D_R, top))
            log.append(s.vcycle())
            log.append(s.l2norm(MG3D_U, top))
            log.append(s.vcycle())
            log += list(s.vcycles(3))                     # a batch call behind single ones continues from the state run ahead
            log.append(s.vcycle())
            if on:                                        # the switch thrown while a down-leg has run ahead
                s.set_option("legs", 0)
            log.append(s.vcycle())
            if on:
                s.set_option("legs", 1)
            log.append(s.vcycle())
            log.append(s.vcycle(top - 1))                 # a cycle from a lower level
            log.append(s.vcycle())
            s.fmg_initialize()
            log.append(s.vcycle())
            log.append(s.vcycle())
            kt = {kn: n for (lvl, kn), (n, _) in s.kernel_times().items() if lvl == top}
            log.append(s.download(MG3D_U, top))
            log.append(s.download(MG3D_U, top - 1))
            log.append(s.download(MG3D_D, top - 1))
        logs.append(log)
        if on:
            assert kt.get("leg_up", 0) >= 14 and kt.get("leg_down", 0) >= 14, kt
        else:
            assert "leg_up" not in kt and "leg_down" not in kt, kt
    assert len(logs[0]) == len(logs[1])
    for i, (a, b) in enumerate(zip(logs[0], logs[1])):
        if isinstance(a, np.ndarray):
            assert np.array_equal(a, b), f"step {i}"
        else:
            np.testing.assert_allclose(a, b, rtol=1e-12, err_msg=f"step {i}")


def test_leg_launches_are_taken_and_counted(monkeypatch):
    """K cycles of one call: K up-leg and K - 1 down-leg launches on the top level, the first cycle's ordinary down-leg, one
    norm-only launch (the last cycle's), nothing else there; off by default."""
    with M.Solver(9, 5, 2) as s:
        _legs(s, True)
        s.setup_test_problem()
        s.timing_enable(1)
        s.vcycles(5)
        kt = {kn: n for (lvl, kn), (n, _) in s.kernel_times().items() if lvl == 4}
    # the first cycle has no cycle in front of it: the ordinary down-leg (four passes; residual + restriction; the face
    # injection once); the last one forms its norm in a launch of its own
    assert kt == {"leg_up": 5, "leg_down": 4, "sweep4": 1, "residual": 2, "restrict": 1}, kt
    with M.Solver(9, 5, 2) as s:
        s.setup_test_problem()
        s.timing_enable(1)
        s.vcycles(3)
        assert not any(kn.startswith("leg_") for (lvl, kn) in s.kernel_times())


def test_right_hand_side_and_scale_do_not_matter(monkeypatch):
    """Random right-hand side and start, values down to denormals: the schedule changes no bit."""
    c, L = 9, 5
    N = (c - 1) * (1 << (L - 1)) + 1
    rng = np.random.default_rng(5)
    for scale in (1.0, 1e-300, 1e150):
        d, u0 = rng.uniform(-1, 1, N ** 3) * scale, rng.uniform(-1, 1, N ** 3) * scale
        res = []
        for on in (True, False):
            with M.Solver(c, L, 2) as s:
                _legs(s, on)
                s.get_details()
                s.upload(MG3D_D, L - 1, d)
                s.upload(MG3D_U, L - 1, u0)
                n = s.vcycles(3)
                res.append((n, s.download(MG3D_U, L - 1), s.download(MG3D_D, L - 2)))
        assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2]), scale
        np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-12)


def test_next_call_continues_behind_the_last_cycle_unless_the_state_was_touched(monkeypatch):
    """A cycle's first red pass is the identity behind the red pass that ended the cycle before it -- also across two mg3d_vcycles
    calls, as long as nothing has touched u or d of the top level in between: the second call's first cycle then takes the
    one-launch down-leg too (no four-pass launch, no residual + restriction launch).  An upload in between ends that: the next
    cycle runs all four pre-smoothing passes -- on a u whose red points no longer are what a red pass would make them -- and
    equals the plain schedule started from the same state, bit for bit."""
    c, L = 9, 5
    top = L - 1
    with M.Solver(c, L, 2) as s:
        _legs(s, True)
        s.setup_test_problem()
        s.vcycles(2)
        s.timing_enable(1)
        n2 = s.vcycles(3)
        kt = {kn: n for (lvl, kn), (n, _) in s.kernel_times().items() if lvl == top}
        assert kt == {"leg_up": 3, "leg_down": 3, "residual": 1}, kt  # (residual: the last cycle's norm-only launch)
        want_norms, want_u, _, _ = O.run_problem(c, L, 2, 5)
        assert np.array_equal(s.download(MG3D_U, top), want_u)
        np.testing.assert_allclose(n2, want_norms[2:], rtol=norm_rtol(s.N))
        # a download in between reads only: the next call still continues behind the last cycle
        s.timing_reset()
        s.vcycles(1)
        kt = {kn: n for (lvl, kn), (n, _) in s.kernel_times().items() if lvl == top}
        assert kt == {"leg_up": 1, "leg_down": 1, "residual": 1}, kt
        assert np.array_equal(s.download(MG3D_U, top), O.run_problem(c, L, 2, 6)[1])
        # touch the state: every interior point moved a little, red ones included
        u = s.download(MG3D_U, top)
        N = s.N
        u3 = u.reshape(N, N, N)
        rng = np.random.default_rng(5)
        u3[1:-1, 1:-1, 1:-1] += 1e-3 * rng.standard_normal((N - 2, N - 2, N - 2))
        s.upload(MG3D_U, top, u)
        s.timing_reset()
        got_norms = s.vcycles(2)
        kt = {kn: n for (lvl, kn), (n, _) in s.kernel_times().items() if lvl == top}
        assert kt.get("sweep4", 0) == 1 and kt.get("leg_down", 0) == 1, kt
        got_u = s.download(MG3D_U, top)
        d = s.download(MG3D_D, top)
    with M.Solver(c, L, 2) as p:
        _legs(p, False)
        p.setup_test_problem()
        p.upload(MG3D_U, top, u)
        p.upload(MG3D_D, top, d)
        plain_norms = p.vcycles(2)
        assert np.array_equal(p.download(MG3D_U, top), got_u)
    np.testing.assert_allclose(got_norms, plain_norms, rtol=1e-12)
