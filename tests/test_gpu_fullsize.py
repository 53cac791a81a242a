"""BASELINE.json configs[3] and configs[4] at their FULL workloads on one GPU.

configs[3] -- 513^3 ("512^3") on i-slabs of 8 ranks (also 2 and 4): all ranks virtual on this GPU through the loopback
transport (device copies instead of RCCL, the same schedule code); the assembled solution must be bit-identical to the
single-domain HIP result, which tests/test_gpu_parity.py pins to the oracle at the same size.
configs[4] -- 1025^3 fp32 / damped Jacobi / F-cycle start (PARITY UNPINNED: no reference implementation exists):
size-independent properties at the full size -- the F-cycle start lands on the binary32 floor, V-cycles stay there, the
solution is the analytic x^2-2y^2+z^2 (which the 7-point stencil reproduces exactly) to binary32 accuracy, and the
paired/fused launches give the same bits as one launch per operator."""
import hashlib

import numpy as np
import pytest

import os

import _oracle as O
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_R, MG3D_U
from test_gpu_parity import norm_rtol

pytestmark = pytest.mark.gpu

_single = {}


def single_513(keep_r=False):
    key = bool(keep_r)
    if key not in _single:
        with M.Solver(9, 7, 2) as s:
            s.set_keep_residual(keep_r)
            s.setup_test_problem()
            norms = s.vcycles(2)
            out = {"norms": norms, "u": s.download(MG3D_U, 6)}
            if keep_r:
                for lvl in (4, 5, 6):
                    out[("r", lvl)] = s.download(MG3D_R, lvl)
                    out[("u", lvl)] = s.download(MG3D_U, lvl)
                    if lvl < 6:
                        out[("d", lvl)] = s.download(MG3D_D, lvl)
        _single[key] = out
    return _single[key]


@pytest.mark.parametrize("P", [8, 4, 2])
def test_configs3_513_cubed_on_slabs(P):
    """`9 7 2` on P slabs, two V(2,2) cycles: every value of the 1.08 GB solution equals the single-domain one."""
    want = single_513()
    with M.DistSolver(9, 7, 2, nranks=P) as d:
        assert d.halo == 6 and d.first_level == {8: 4, 4: 3, 2: 2}[P]  # levels >= 129^3 / 65^3 / 33^3 distributed
        d.setup_test_problem()
        norms = d.vcycles(2)
        u = d.download(MG3D_U, 6)
        if P == 8:  # the slab path's own reduction at full size against the exactly rounded sum of the same 133 M squares
            assert norms[-1] == pytest.approx(O.exact_residual_norm(u, d.download(MG3D_D, 6), 513, d.h), rel=1e-13)
    assert np.array_equal(u, want["u"])
    np.testing.assert_allclose(norms, want["norms"], rtol=norm_rtol(513), atol=0)
    np.testing.assert_allclose(norms, [3.86147e+07, 4.68671e+06], rtol=2e-6)  # the reference's printed history


def test_configs3_513_cubed_8_slabs_every_distributed_level():
    """r kept (reference-visible residual arrays): u, d and r of the three distributed levels, owned planes of all 8
    ranks assembled, equal the single-domain arrays after two cycles."""
    want = single_513(keep_r=True)
    with M.DistSolver(9, 7, 2, nranks=8) as d:
        d.set_keep_residual(True)
        d.setup_test_problem()
        norms = d.vcycles(2)
        np.testing.assert_allclose(norms, want["norms"], rtol=norm_rtol(513), atol=0)
        for lvl in (4, 5, 6):
            assert np.array_equal(d.download(MG3D_U, lvl), want[("u", lvl)]), f"u level {lvl}"
            assert np.array_equal(d.download(MG3D_R, lvl), want[("r", lvl)]), f"r level {lvl}"
            if lvl < 6:
                assert np.array_equal(d.download(MG3D_D, lvl), want[("d", lvl)]), f"d level {lvl}"


@pytest.mark.parametrize("c,L", [(3, 9), (17, 6), (33, 2), (17, 2)])
def test_admissible_coarse_grids_bit_exact_against_oracle(c, L):
    """The coarse grids mg_3d.h:123,163 admits beyond the 9^3 of the headline: `3 9 2` (513^3 through NINE levels, SURVEY
    8(d)'s alternative spelling), `17 6 2` (513^3 over a 17^3 coarse grid: 4913 unknowns, half-band 289 -- the wide-band
    solve kernel), and two-level cycles over 33^3 (35937 unknowns, half-band 1089, the largest the reference's
    assert(n*n < INT_MAX) lets through) and 17^3.  Two V(2,2) cycles, the whole solution vector against the oracle."""
    nu = 2
    N = (c - 1) * (1 << (L - 1)) + 1
    O.lib().orc_set_threads(min(16, os.cpu_count() or 1))
    want_norms, want_u, want_init, _ = O.run_problem(c, L, nu, 2)
    O.lib().orc_set_threads(1)
    with M.Solver(c, L, nu) as s:
        s.setup_test_problem()
        init = s.get_initial_residual()
        got = s.vcycles(2)
        u = s.download(MG3D_U, L - 1)
    assert init == pytest.approx(want_init, rel=norm_rtol(N))  # tree sum on the device, sequential sum in the oracle
    assert np.array_equal(u, want_u)
    np.testing.assert_allclose(got, want_norms, rtol=norm_rtol(N), atol=0)
    if (c, L) == (3, 9):
        np.testing.assert_allclose(got, [3.86147e+07, 4.68671e+06], rtol=3e-2)  # same problem as `9 7 2`, other hierarchy


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8)).hexdigest()


def test_configs4_1025_cubed_fp32_fcycle(monkeypatch):
    """`9 8 2` = 1025^3, binary32, damped Jacobi (omega 6/7): F-cycle start + 4 V(2,2) cycles."""
    c, L, nu = 9, 8, 2
    N = 1025
    with M.Solver32(c, L, nu) as s:
        s.setup_test_problem(fmg=False)
        cold = s.vcycles(2)  # from a zero guess: the first cycles are far above the floor
        s.setup_test_problem(fmg=True)
        norms = s.vcycles(4)
        u = s.download(MG3D_U, L - 1)
    digest, n_ref = _digest(u), norms.copy()
    # the F-cycle start replaces thousands of x of residual reduction: its first V-cycle already ends at the floor
    assert norms[0] < 1e-2 * cold[0]
    assert norms.max() < 4 * norms.min()  # parked at the binary32 floor, neither diverging nor still converging
    # solution against the analytic one, plane by plane (the array is 4.3 GB; u* in double)
    U = u.reshape(N, N, N)
    x = np.arange(N) / (N - 1.0)
    yz = -2.0 * x[:, None] ** 2 + x[None, :] ** 2  # [j, k]
    worst = 0.0
    for i in range(0, N, 8):
        worst = max(worst, float(np.abs(U[i].astype(np.float64) - (x[i] ** 2 + yz)).max()))
    # boundary values are exact binary32 roundings of u*; interior: rounding floor of a 1025^3 binary32 solve
    print("fp32 1025^3: norms", norms, "cold", cold, "max |u - u*|", worst)
    assert worst < 2e-3, worst
    assert np.abs(U[0].astype(np.float64) - (x[0] ** 2 + yz)).max() < 3e-7  # a face: u* rounded to binary32
    del U, u
    # one launch per sweep / per operator instead of the paired and fused kernels: same bits
    for knob in ("MG3D_F32_NO_PAIRS", "MG3D_F32_NO_FUSE"):
        monkeypatch.setenv(knob, "1")
        with M.Solver32(c, L, nu) as s:
            s.setup_test_problem(fmg=True)
            n2 = s.vcycles(4)
            assert _digest(s.download(MG3D_U, L - 1)) == digest, knob
        np.testing.assert_allclose(n2, n_ref, rtol=1e-12)
        monkeypatch.delenv(knob)


def test_configs4_1025_cubed_fp32_on_8_slabs():
    """configs[4] at its workload on 8 virtual ranks (loopback): F-cycle start + 2 V(2,2) cycles at 1025^3; the
    assembled 4.3 GB solution has the digest of the single-domain one, norms to the summation order."""
    c, L, nu = 9, 8, 2
    with M.Solver32(c, L, nu) as s:
        s.setup_test_problem(fmg=True)
        want_n = s.vcycles(2)
        want = _digest(s.download(MG3D_U, L - 1))
    with M.DistSolver32(c, L, nu, nranks=8) as d:
        assert d.halo == 4 and d.first_level == 4  # levels >= 129^3 distributed (16 planes per rank there)
        d.setup_test_problem(fmg=True)
        norms = d.vcycles(2)
        got = _digest(d.download(MG3D_U, L - 1))
    assert got == want
    np.testing.assert_allclose(norms, want_n, rtol=1e-12, atol=0)
