"""The i-slab (multi-GPU) V-cycle, verified on ONE GPU through the loopback transport: all ranks are virtual,
live in this process and exchange halos by device copies, running the same schedule code as the RCCL path.
Every owned plane of the assembled solution must be bit-identical to the single-domain HIP result (which is
itself bit-identical to the oracle, tests/test_gpu_parity.py)."""
import numpy as np
import pytest

import _oracle as O
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import MG3D_D, MG3D_R, MG3D_U

EXACT_NORM_RTOL = 1e-13  # the slab path's reduction (per-rank partial sums, all-gather, sum in rank order) against the exactly rounded sum


def exact_norm(d, L):
    """exactly rounded residual norm of the assembled finest-level state of a DistSolver (tests/_oracle.py)"""
    N = d.N
    return O.exact_residual_norm(d.download(MG3D_U, L - 1), d.download(MG3D_D, L - 1), N, d.h)

pytestmark = pytest.mark.gpu


def single(c, L, nu, cycles):
    with M.Solver(c, L, nu) as s:
        s.setup_test_problem()
        norms = s.vcycles(cycles)
        return norms, s.download(MG3D_U, L - 1)


@pytest.mark.parametrize("min_planes", [8, 16])
@pytest.mark.parametrize("c,L,nu,P", [(5, 5, 2, 2), (5, 5, 2, 4), (5, 5, 2, 8), (9, 5, 2, 8), (9, 5, 2, 2), (5, 5, 1, 4),
                                      (5, 5, 3, 2), (3, 6, 2, 3), (9, 4, 2, 4), (9, 6, 2, 4), (9, 6, 1, 4), (9, 6, 3, 2)])
def test_slab_vcycles_match_single_domain(monkeypatch, c, L, nu, P, min_planes):
    """min_planes 8: thin slabs, as many distributed levels as possible; 16: the default replication threshold."""
    if ((c - 1) << (L - 1)) // P < max(min_planes, 2 * nu + 2):
        pytest.skip("no level gives every rank that many planes")
    monkeypatch.setenv("MG3D_SLAB_MIN_PLANES", str(min_planes))
    cycles = 6
    want_norms, want_u = single(c, L, nu, cycles)
    with M.DistSolver(c, L, nu, nranks=P) as d:
        assert 1 <= d.first_level < L and d.halo == 2 * nu + 2
        d.setup_test_problem()
        norms = d.vcycles(cycles)
        u = d.download(MG3D_U, L - 1)
        assert norms[-1] == pytest.approx(exact_norm(d, L), rel=EXACT_NORM_RTOL)
    assert np.array_equal(u, want_u)
    np.testing.assert_allclose(norms, want_norms, rtol=1e-11, atol=0)


@pytest.mark.parametrize("c,L,P,min_planes", [(9, 5, 2, 16), (9, 5, 8, 16), (9, 5, 4, 8), (5, 6, 4, 8), (3, 7, 3, 16), (9, 6, 4, 16),
                                              (9, 6, 8, 8)])
def test_slab_carried_cycles(monkeypatch, c, L, P, min_planes):
    """Carried cycles on slabs (V(2,2), top level > 65^3; by default from 257^3 up -- the last two cases): every cycle but
    the last of a call ends with the four-pass launch that taps the norm and begins the next cycle, the exchange behind it
    refreshes three halo planes (plan variant `policy | 2`), the next down-leg is one launch.  Equal to the single domain
    and to the slab path's plain schedule bit for bit, also over several calls."""
    monkeypatch.setenv("MG3D_SLAB_MIN_PLANES", str(min_planes))
    monkeypatch.setenv("MG3D_CARRY_MIN", "66")
    monkeypatch.setenv("MG3D_LEGS", "0")  # (from 160 points per side the one-launch legs would take the carried cycles' place)
    want_norms, want_u = single(c, L, 2, 7)
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("MG3D_NO_CARRY", flag)
        with M.DistSolver(c, L, 2, nranks=P) as d:
            d.setup_test_problem()
            norms = list(d.vcycles(4)) + list(d.vcycles(1)) + list(d.vcycles(2))
            assert d.carried_cycles() == (4 if flag == "0" else 0)  # 3 + 0 + 1
            assert norms[-1] == pytest.approx(exact_norm(d, L), rel=EXACT_NORM_RTOL)
            res.append((np.array(norms), d.download(MG3D_U, L - 1), [d.download(MG3D_U, l) for l in range(d.first_level, L - 1)]))
    assert np.array_equal(res[0][1], want_u) and np.array_equal(res[1][1], want_u)
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a, b)
    np.testing.assert_allclose(res[0][0], want_norms, rtol=1e-11, atol=0)
    np.testing.assert_allclose(res[1][0], want_norms, rtol=1e-11, atol=0)


@pytest.mark.parametrize("overlap", ["0", "1"])
@pytest.mark.parametrize("c,L,P,min_planes", [(9, 5, 2, 16), (9, 5, 8, 16), (9, 5, 4, 8), (5, 6, 4, 8), (3, 7, 3, 16), (9, 6, 4, 16),
                                              (9, 6, 8, 8), (9, 6, 2, 16),
                                              # 80: only 257^3 is distributed, the replicated hierarchy below it has a 129^3 top level
                                              # that qualifies for the one-launch legs itself -- and whose u and d the slab path
                                              # rewrites every cycle (the context must not take it for a cycle continuing behind one)
                                              (9, 6, 2, 80)])
def test_slab_one_launch_per_leg(monkeypatch, c, L, P, min_planes, overlap):
    """One launch per leg on slabs (V(2,2); by default from 160 points per side -- the last three cases): every cycle but the last
    of a batch ends with the one-launch up-leg over the owned planes (edge windows first when the exchanges have their own stream
    and the slab is thick enough), the exchange behind it brings five halo planes (plan variant `policy | 4`), the next cycle's
    down-leg is one launch that also completes the norm (`policy | 8`).  Equal to the single domain and to the slab path's plain
    schedule bit for bit, also over several calls."""
    monkeypatch.setenv("MG3D_SLAB_MIN_PLANES", str(min_planes))
    monkeypatch.setenv("MG3D_LEGS_MIN", "66")
    monkeypatch.setenv("MG3D_NO_OVERLAP", "0" if overlap == "1" else "1")
    want_norms, want_u = single(c, L, 2, 7)
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MG3D_LEGS", flag)
        monkeypatch.setenv("MG3D_NO_CARRY", "1")
        with M.DistSolver(c, L, 2, nranks=P) as d:
            d.setup_test_problem()
            norms = list(d.vcycles(4)) + list(d.vcycles(1)) + list(d.vcycles(2))
            assert d.legs_cycles() == (4 if flag == "1" else 0) and d.carried_cycles() == 0  # 3 + 0 + 1
            assert norms[-1] == pytest.approx(exact_norm(d, L), rel=EXACT_NORM_RTOL)
            res.append((np.array(norms), d.download(MG3D_U, L - 1), [d.download(MG3D_U, l) for l in range(d.first_level, L - 1)]))
    assert np.array_equal(res[0][1], want_u) and np.array_equal(res[1][1], want_u)
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a, b)
    np.testing.assert_allclose(res[0][0], want_norms, rtol=1e-11, atol=0)
    np.testing.assert_allclose(res[1][0], want_norms, rtol=1e-11, atol=0)
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-13, atol=0)


@pytest.mark.parametrize("fuse", ["0", "1"])
def test_slab_one_sweep_cycles_both_down_leg_routes(monkeypatch, fuse):
    """V(1,1) on slabs: the down-leg's two passes + residual + restriction as two launches (default) or as the one-launch
    shape (MG3D_FUSE_RST2=1) -- both bit-identical to the single domain."""
    monkeypatch.setenv("MG3D_SLAB_MIN_PLANES", "8")
    monkeypatch.setenv("MG3D_FUSE_RST2", fuse)
    c, L, nu, P = 9, 5, 1, 4
    want_norms, want_u = single(c, L, nu, 5)
    with M.DistSolver(c, L, nu, nranks=P) as d:
        d.setup_test_problem()
        norms = d.vcycles(5)
        assert np.array_equal(d.download(MG3D_U, L - 1), want_u)
    np.testing.assert_allclose(norms, want_norms, rtol=1e-11, atol=0)


def test_slab_intermediate_levels_match_single_domain(monkeypatch):
    monkeypatch.setenv("MG3D_SLAB_MIN_PLANES", "8")
    c, L, nu, P = 9, 5, 2, 4
    with M.Solver(c, L, nu) as s, M.DistSolver(c, L, nu, nranks=P) as d:
        s.set_keep_residual(True)
        d.set_keep_residual(True)
        s.setup_test_problem()
        d.setup_test_problem()
        s.vcycles(2)
        d.vcycles(2)
        for lvl in range(d.first_level, L):
            assert np.array_equal(d.download(MG3D_U, lvl), s.download(MG3D_U, lvl)), f"u level {lvl}"
            if lvl < L - 1:
                assert np.array_equal(d.download(MG3D_D, lvl), s.download(MG3D_D, lvl)), f"d level {lvl}"
            assert np.array_equal(d.download(MG3D_R, lvl), s.download(MG3D_R, lvl)), f"r level {lvl}"
        for lvl in range(d.first_level):  # replicated levels
            assert np.array_equal(d.download(MG3D_U, lvl), s.download(MG3D_U, lvl)), f"u level {lvl}"
            assert np.array_equal(d.download(MG3D_D, lvl), s.download(MG3D_D, lvl)), f"d level {lvl}"


def test_single_rank_rccl_communicator():
    """nranks = 1 through the RCCL code path (unique id, ncclCommInitRank is skipped for one rank)."""
    uid = M.DistSolver.unique_id()
    assert len(uid) == 128
    want_norms, want_u = single(5, 4, 2, 3)
    with M.DistSolver(5, 4, 2, rank=0, nranks=1, unique_id=uid) as d:
        d.setup_test_problem()
        norms = d.vcycles(3)
        assert np.array_equal(d.download(MG3D_U, 3), want_u)
    np.testing.assert_allclose(norms, want_norms, rtol=1e-11)


def test_single_rank_builds_real_communicators(monkeypatch):
    """MG3D_FORCE_COMM=1: the 128-byte unique id travels through ctypes into ncclCommInitRank, every exchange of the cycle
    runs as a self-addressed grouped send/receive (and the norm as a real all-gather) on the ONE communicator, driven from
    the communication stream behind events of the compute stream (round 4: the default for RCCL too), and the communicator
    is destroyed again -- everything of the multi-process set-up that one GPU can run (RCCL refuses two ranks on one
    device)."""
    monkeypatch.setenv("MG3D_FORCE_COMM", "1")
    uid = M.DistSolver.unique_id()
    assert len(uid) == 128 and any(uid)
    want_norms, want_u = single(5, 4, 2, 3)
    with M.DistSolver(5, 4, 2, rank=0, nranks=1, unique_id=uid) as d:
        assert d.comm_info()[:2] == (1, True)  # ncclCommCount of the real communicator; exchanges on the communication stream
        d.setup_test_problem()
        norms = d.vcycles(3)
        assert np.array_equal(d.download(MG3D_U, 3), want_u)
    np.testing.assert_allclose(norms, want_norms, rtol=1e-11)
    monkeypatch.setenv("MG3D_NO_OVERLAP", "1")
    uid = M.DistSolver.unique_id()  # a unique id serves one communicator
    with M.DistSolver(5, 4, 2, rank=0, nranks=1, unique_id=uid) as d:  # every exchange on the compute stream
        assert d.comm_info()[:2] == (1, False)
        d.setup_test_problem()
        assert np.array_equal(d.vcycles(3), norms)


def test_overlap_and_sequential_exchange_agree(monkeypatch):
    """MG3D_NO_OVERLAP=1 keeps every halo exchange on the compute stream; by default all of them are issued on the
    communication stream (one communicator) and the large u exchanges run underneath the coarser levels, the norm kernel
    and the interior of the launch that follows them."""
    for carry_min in ("130", "66"):  # plain schedule / carried cycles (whose last u exchange is the three-plane one)
        monkeypatch.setenv("MG3D_CARRY_MIN", carry_min)
        res = []
        for flag in ("0", "1"):
            monkeypatch.setenv("MG3D_NO_OVERLAP", flag)
            with M.DistSolver(9, 5, 2, nranks=2) as d:
                d.setup_test_problem()
                res.append((d.vcycles(5), d.download(MG3D_U, 4)))
                assert d.carried_cycles() == (4 if carry_min == "66" else 0)
        # (with the exchanges on the communication stream the stage's last launch makes its edge windows first, as a launch of
        # their own: the same squares in another grouping of per-block partial sums)
        np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-13)
        assert np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("c,L,nu,P,min_planes", [(5, 5, 2, 4, 8), (9, 5, 2, 8, 16), (3, 6, 2, 3, 8), (9, 5, 1, 2, 16)])
def test_coarse_levels_on_rank_0_only(monkeypatch, c, L, nu, P, min_planes):
    """MG3D_COARSE_GATHER=1 (the north-star's wording: "the coarsest level's direct solve is gathered to rank 0"): the
    restricted right-hand side travels to rank 0, which alone runs the small levels and the gauss_elim.h solve and
    broadcasts the correction.  Same bits as the default (every rank runs them redundantly behind one all-gather)."""
    monkeypatch.setenv("MG3D_SLAB_MIN_PLANES", str(min_planes))
    want_norms, want_u = single(c, L, nu, 4)
    monkeypatch.setenv("MG3D_COARSE_GATHER", "1")
    with M.DistSolver(c, L, nu, nranks=P) as d:
        d.setup_test_problem()
        norms = d.vcycles(4)
        assert np.array_equal(d.download(MG3D_U, L - 1), want_u)
    np.testing.assert_allclose(norms, want_norms, rtol=1e-11, atol=0)


def test_slab_phase_timers(monkeypatch):
    """mg3d_dist_timing_*: the per-phase split bench.py's N > 1 line reports per rank"""
    with M.DistSolver(9, 5, 2, nranks=4) as d:
        d.setup_test_problem()
        d.vcycles(2)
        d.timing_enable(True)
        norms = d.vcycles(5)
        t = d.timing()
        d.timing_enable(False)
        again = d.vcycles(1)
    assert t["cycles"] == 5 and t["cycle_ms"] > 0
    assert t["exchange_ms"] > 0 and t["replicated_ms"] > 0 and t["exchange_overlapped_ms"] > 0  # loopback overlaps by default
    assert 0 < t["kernels_ms"] < t["cycle_ms"]
    assert again[0] < norms[-1]
