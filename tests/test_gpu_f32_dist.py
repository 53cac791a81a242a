"""The single-precision / damped-Jacobi / F-cycle variant on i-slabs (BASELINE configs[4]: 1025^3 on 8 GPUs), verified on
ONE GPU through the loopback transport: all ranks virtual in this process, device copies instead of RCCL, the same
schedule code.  PARITY UNPINNED like the variant itself; what these tests establish is that the decomposition changes no
bit: every owned plane of the assembled solution equals the single-domain `Solver32` result (which
tests/test_gpu_f32.py compares with its CPU restatement), norms to the summation order."""
import numpy as np
import pytest

import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_U

pytestmark = pytest.mark.gpu


def single(c, L, nu, cycles, fmg):
    with M.Solver32(c, L, nu) as s:
        s.setup_test_problem(fmg=fmg)
        return s.vcycles(cycles), s.download(MG3D_U, L - 1)


@pytest.mark.parametrize("fmg", [False, True])
@pytest.mark.parametrize("c,L,nu,P", [(5, 5, 2, 2), (5, 5, 2, 4), (9, 5, 2, 8), (9, 5, 2, 3), (5, 6, 2, 8), (5, 5, 1, 4), (5, 5, 3, 2),
                                      (9, 4, 4, 2)])
def test_fp32_slabs_match_single_domain(monkeypatch, c, L, nu, P, fmg):
    monkeypatch.setenv("MG3D_SLAB_MIN_PLANES", "8")  # thin slabs: as many distributed levels as possible
    cycles = 4
    want_norms, want_u = single(c, L, nu, cycles, fmg)
    with M.DistSolver32(c, L, nu, nranks=P) as d:
        assert 1 <= d.first_level < L and d.halo == nu + 2
        d.setup_test_problem(fmg=fmg)
        norms = d.vcycles(cycles)
        u = d.download(MG3D_U, L - 1)
    assert np.array_equal(u, want_u)
    np.testing.assert_allclose(norms, want_norms, rtol=1e-12, atol=0)


def test_fp32_slabs_default_threshold_and_unfused(monkeypatch):
    """Default replication threshold (16 planes per rank) and the one-launch-per-operator kernels on slabs."""
    want_norms, want_u = single(9, 5, 2, 3, True)
    for knob in (None, "MG3D_F32_NO_PAIRS", "MG3D_F32_NO_FUSE"):
        if knob:
            monkeypatch.setenv(knob, "1")
        with M.DistSolver32(9, 5, 2, nranks=4) as d:
            d.setup_test_problem(fmg=True)
            norms = d.vcycles(3)
            assert np.array_equal(d.download(MG3D_U, 4), want_u), knob
        np.testing.assert_allclose(norms, want_norms, rtol=1e-12, atol=0)
        if knob:
            monkeypatch.delenv(knob)


def test_fp32_slabs_random_rhs(monkeypatch):
    """Seeded random right-hand side and initial guess (no symmetry to hide an index slip), 3 ranks, uneven slabs."""
    monkeypatch.setenv("MG3D_SLAB_MIN_PLANES", "8")
    c, L, nu, P = 9, 4, 2, 3
    N = (c - 1) * (1 << (L - 1)) + 1
    rng = np.random.default_rng(7)
    u0 = rng.uniform(-1, 1, N ** 3).astype(np.float32)
    d0 = rng.uniform(-1, 1, N ** 3).astype(np.float32)
    with M.Solver32(c, L, nu) as s:
        s.upload(MG3D_U, L - 1, u0)
        s.upload(MG3D_D, L - 1, d0)
        want_n = s.vcycles(3)
        want_u = s.download(MG3D_U, L - 1)
    with M.DistSolver32(c, L, nu, nranks=P) as d:
        d.upload(MG3D_U, L - 1, u0)
        d.upload(MG3D_D, L - 1, d0)
        norms = d.vcycles(3)
        assert np.array_equal(d.download(MG3D_U, L - 1), want_u)
    np.testing.assert_allclose(norms, want_n, rtol=1e-12, atol=0)
