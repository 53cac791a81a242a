"""World-size-2 (and 3) run of the i-slab V-cycle SCHEDULE on CPU: real processes, torch.distributed `gloo`
send/recv for the halo planes, broadcast for the replicated right-hand side, all_gather for the norm.

What is under test is the host logic the multi-GPU path rests on: the partition (`mg3d_slab_*` from libmg3d.so,
pure host arithmetic), the halo depth H = 2*nu+2, and the EXCHANGE PLAN of csrc/mg3d_dist.hip (`mg3d_dist_plan`: the
very list of sends / receives / broadcasts its RCCL transport issues, phase by phase) -- every transfer below is read
from that plan, none is computed here; the order of the compute steps mirrors dist_enqueue_vcycle.  The per-slab arithmetic is the numpy statement of the
operators (tests/_slab_numpy.py, pinned to the oracle); the replicated levels run the oracle's V-cycle.  The
assembled solution must equal the single-domain oracle bit for bit.  No GPU, no HIP compute."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _oracle as O
import _plan as PL
import _slab_numpy as S
import multigrid_parallel_amd as M


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def owned(c, L, P, H, level, r):
    lo, hi = C.c_int(0), C.c_int(0)
    assert M.lib().mg3d_slab_owned(c, L, P, H, level, r, C.byref(lo), C.byref(hi)) == 0
    return lo.value, hi.value


class Slab:
    def __init__(self, c, L, P, H, level, r):
        self.N = (c - 1) * (1 << level) + 1
        self.glo, self.ghi = owned(c, L, P, H, level, r)
        self.h_lo = H if r > 0 else 0
        self.h_hi = H if r < P - 1 else 0
        self.ig0 = self.glo - self.h_lo
        self.ni = self.ghi - self.glo + self.h_lo + self.h_hi
        self.own_lo, self.own_hi = self.h_lo, self.h_lo + self.ghi - self.glo
        self.u, self.d, self.r = (np.zeros((self.ni, self.N, self.N)) for _ in range(3))


class PlanRunner:
    """Executes the PRODUCT's exchange plan (mg3d_dist_plan -- the list the RCCL and loopback transports of
    csrc/mg3d_dist.hip execute) over gloo: this test holds no plane arithmetic of its own for the transfers."""

    def __init__(self, c, L, P, nu, r, policy=0):
        self.plain = PL.phases(c, L, P, nu, r, 0, policy)
        self.carried = PL.phases(c, L, P, nu, r, 0, policy | 2)  # a cycle that ends ahead into the next one
        # one launch per leg: a cycle that ends with the one-launch up-leg (4), one that follows such a cycle (8), both (12)
        self.legs = {(i, o): PL.phases(c, L, P, nu, r, 0, policy | (8 if i else 0) | (4 if o else 0)) for i in (0, 1) for o in (0, 1)}
        self.ph = self.plain
        self.r, self.cur = r, 0

    def start_cycle(self, carry_out=False, legs_in=False, legs_out=False):
        self.cur = 0
        self.ph = self.legs[(int(legs_in), int(legs_out))] if (legs_in or legs_out) else self.carried if carry_out else self.plain

    def run(self, kind, level, array_of, norm_part=None):
        """next phase of the cycle; array_of(field, level) -> the (planes, N, N) array the entries index"""
        es = self.ph.get(self.cur, [])
        self.cur += 1
        assert all(e.kind == kind and e.level == level for e in es), \
            f"schedule and plan out of step: phase {self.cur - 1} is {[(PL.KIND_NAMES[e.kind], e.level) for e in es][:1]}, schedule wants {PL.KIND_NAMES[kind]} {level}"
        reqs, landing, gathered = [], [], None
        for e in es:
            if e.op == PL.ALLGATHER:
                parts = [torch.zeros(1, dtype=torch.float64) for _ in range(dist.get_world_size())]
                dist.all_gather(parts, torch.tensor([norm_part], dtype=torch.float64))
                gathered = [float(p) for p in parts]
                continue
            a = array_of(e.field, e.level)
            pitch = (a.shape[2] + 15) // 16 * 16
            assert e.plane_elems == pitch * a.shape[1]  # the device layout's padded plane
            view = a[e.offset:e.offset + e.count]
            assert view.shape[0] == e.count, "entry reaches outside the array"
            if e.op == PL.SEND:
                reqs.append(dist.isend(torch.from_numpy(view.copy()), e.peer))
            elif e.op == PL.RECV:
                t = torch.empty(view.shape, dtype=torch.float64)
                reqs.append(dist.irecv(t, e.peer))
                landing.append((t, view))
            else:  # broadcast in place, root = peer
                t = torch.from_numpy(view.copy())
                dist.broadcast(t, src=e.peer)
                view[:] = t.numpy()
        for q in reqs:
            q.wait()
        for t, view in landing:
            view[:] = t.numpy()
        return gathered


def worker(r, P, port, c, L, nu, cycles, out_path, policy=0, carry=False, legs=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=r, world_size=P)
    lib = M.lib()
    H = lib.mg3d_slab_halo(nu)
    ld = lib.mg3d_slab_first_level(c, L, P, H)
    Nf = (c - 1) * (1 << (L - 1)) + 1
    hs = [1.0 / (Nf - 1) * (1 << (L - 1 - l)) for l in range(L)]
    lv = {l: Slab(c, L, P, H, l, r) for l in range(ld, L)}
    # replicated hierarchy 0..ld-1 (oracle layout) + LU with the reference's coarse spacing (mg_3d.h:287)
    Hc = O.Hierarchy(c, ld)
    n0 = c ** 3
    LU = np.zeros(n0 * n0)
    O.lib().orc_coarse_matrix(O.P(LU), c, hs[0])
    O.lib().orc_lu_factor(O.P(LU), n0)
    O.lib().orc_set_threads(1)
    # test_mg_3d.c problem on the full grid, each rank keeps its slab
    full = np.zeros(Nf ** 3)
    O.lib().orc_fill_boundary(O.P(full), Nf, hs[L - 1])
    f3 = full.reshape(Nf, Nf, Nf)
    top = lv[L - 1]
    top.u[:] = f3[top.ig0:top.ig0 + top.ni]
    top.d[:] = f3[top.ig0:top.ig0 + top.ni]
    norms = []
    plan = PlanRunner(c, L, P, nu, r, policy)
    Ncr = Hc.N[ld - 1]

    def array_of(field, level):
        if level >= ld:
            return getattr(lv[level], "ud"[field])
        return (Hc.u if field == 0 else Hc.d)[level].reshape(Ncr, Ncr, Ncr)

    carried = False  # u of the top level already holds the next cycle's first three pre-smoothing passes
    pending = None   # one launch per leg: the previous cycle's sum of squares, reduced behind this cycle's down-leg
    for cyc in range(cycles):
        # carried cycles (V(2,2); csrc/mg3d_dist.hip dist_enqueue_vcycle): every cycle but the last ends ahead
        carry_in, carry_out = carried, carry and nu == 2 and cyc + 1 < cycles
        # one launch per leg: every cycle but the last ends with the one-launch up-leg, whose norm the next cycle completes
        legs_in, legs_out = pending is not None, legs and nu == 2 and cyc + 1 < cycles
        plan.start_cycle(carry_out, legs_in, legs_out)
        for l in range(L - 1, ld - 1, -1):  # ---- down
            sl = lv[l]
            if l < L - 1:
                sl.u[:] = 0.0
            if l == L - 1 and legs_in:
                # the down-leg in one launch: the cycle's first red pass is the identity behind the previous cycle's last one
                # (checked here, skipped on the GPU); black, red, black are left.  It reads five planes either side.
                before = sl.u.copy()
                S.colour_pass(sl.u, sl.d, hs[l], 1, sl.ig0, sl.N)
                assert np.array_equal(sl.u[sl.own_lo:sl.own_hi], before[sl.own_lo:sl.own_hi]), "red behind red: the identity"
                sl.u[:] = before
                for colour in (0, 1, 0):
                    S.colour_pass(sl.u, sl.d, hs[l], colour, sl.ig0, sl.N)
            elif l == L - 1 and carry_in:
                S.colour_pass(sl.u, sl.d, hs[l], 0, sl.ig0, sl.N)  # the one pre-smoothing pass that is left: black
            else:
                S.smooth(sl.u, sl.d, hs[l], nu, False, sl.ig0, sl.N)
            S.residual(sl.u, sl.d, hs[l], sl.r, sl.ig0, sl.N)
            if l - 1 >= ld:
                sc = lv[l - 1]
                S.restrict_planes(sl.r, sl.ig0, sl.N, sc.d, sc.ig0, sc.N, sc.own_lo, sc.own_hi)
            else:
                dc = Hc.d[ld - 1].reshape(Ncr, Ncr, Ncr)
                flo, fhi = owned(c, L, P, H, ld, r)
                S.restrict_planes(sl.r, sl.ig0, sl.N, dc, 0, Ncr, 0 if r == 0 else flo // 2, Ncr if r == P - 1 else fhi // 2)
            if l == L - 1 and legs_in:  # the previous cycle's norm, completed by the launch above
                parts = plan.run(PL.NORM, L - 1, array_of, norm_part=pending)
                norms.append(float(np.sqrt(sum(parts))))
                pending = None
            # first what the coarser level waits for ...
            if l - 1 >= ld:
                plan.run(PL.HALO_D, l - 1, array_of)
            else:
                plan.run(PL.RHS_GATHER if policy else PL.RHS_ALLGATHER, ld - 1, array_of)
            # ... then the halos of u, final on this level until the way up (on the GPU: issued behind the exchange above on
            # the one communication stream, running underneath the coarser levels)
            plan.run(PL.HALO_U_DOWN, l, array_of)
        # ---- coarse levels: identical on every rank -- or on rank 0 alone, whose correction is then broadcast
        if not policy or r == 0:
            Hc.u[ld - 1][:] = 0.0
            O.lib().orc_vcycle(Hc.ptrs(Hc.u), Hc.ptrs(Hc.d), Hc.ptrs(Hc.r), hs[ld - 1], ld - 1, L, nu, Hc.N[ld - 1], O.P(LU))
        else:
            Hc.u[ld - 1][:] = np.nan  # must be overwritten by the broadcast
        if policy:
            plan.run(PL.CORR_BCAST, ld - 1, array_of)
        for l in range(ld, L):  # ---- up
            sl = lv[l]
            # the correction is applied to every local plane, halos included: both operands have exact halos
            if l - 1 >= ld:
                sc = lv[l - 1]
                plan.run(PL.HALO_U_UP, l - 1, array_of)
                S.prolong_planes(sc.u, sc.ig0, sc.N, sl.u, sl.ig0, sl.N, 0, sl.ni)
            else:
                S.prolong_planes(Hc.u[ld - 1].reshape(Ncr, Ncr, Ncr), 0, Ncr, sl.u, sl.ig0, sl.N, 0, sl.ni)
            S.smooth(sl.u, sl.d, hs[l], nu, True, sl.ig0, sl.N)  # no exchange: uses up 2*nu of the H halo planes
        # on the GPU the exchange below runs underneath the norm kernel, which reads the first halo plane:
        # that plane is left as the post-smoother produced it (exact), planes 2..H are refreshed
        top_before = top.u.copy()
        if legs_out:
            # the one-launch up-leg wrote the owned planes only; five halo planes either side come by exchange for the next
            # cycle's down-leg (the sixth stays stale until the exchange behind that launch).  No NORM phase: the sum of
            # squares over the owned planes waits for the next cycle.
            pending = S.residual(top_before, top.d, hs[L - 1], None, top.ig0, top.N, top.own_lo, top.own_hi)
            top.u[:top.own_lo] = np.nan
            top.u[top.own_hi:] = np.nan
            plan.run(PL.HALO_U_NEXT, L - 1, array_of)
            if top.own_lo:
                assert np.isnan(top.u[:top.own_lo - 5]).all() and not np.isnan(top.u[top.own_lo - 5:top.own_lo]).any()
            assert plan.cur == len(plan.ph), "the cycle used every phase of the plan"
            stale = np.isnan(top.u)
            top.u[stale] = top_before[stale]  # (stale, not poisoned: the next down-leg must not depend on them)
            continue
        if carry_out:
            # the launch that took the norm went on: the next cycle's pre-smoothing passes red (the identity behind the
            # post-smoother's last red pass: checked here, skipped on the GPU), black, red -- every halo plane is used up,
            # the plan's exchange brings three back, which the one pass + residual + restriction left of that down-leg read
            ss = S.residual(top_before, top.d, hs[L - 1], None, top.ig0, top.N, top.own_lo, top.own_hi)
            S.colour_pass(top.u, top.d, hs[L - 1], 1, top.ig0, top.N)
            assert np.array_equal(top.u[top.own_lo:top.own_hi], top_before[top.own_lo:top.own_hi]), "red behind red: the identity"
            S.colour_pass(top.u, top.d, hs[L - 1], 0, top.ig0, top.N)
            S.colour_pass(top.u, top.d, hs[L - 1], 1, top.ig0, top.N)
            plan.run(PL.HALO_U_NEXT, L - 1, array_of)
        else:
            plan.run(PL.HALO_U_NEXT, L - 1, array_of)
            ss = S.residual(top_before, top.d, hs[L - 1], None, top.ig0, top.N, top.own_lo, top.own_hi)
            assert ss == S.residual(top.u, top.d, hs[L - 1], None, top.ig0, top.N, top.own_lo, top.own_hi)
        carried = carry_out
        parts = plan.run(PL.NORM, L - 1, array_of, norm_part=ss)
        assert plan.cur == len(plan.ph), "the cycle used every phase of the plan"
        norms.append(float(np.sqrt(sum(parts))))
    # assemble the owned planes on rank 0
    mine = torch.from_numpy(top.u[top.own_lo:top.own_hi].copy())
    if r == 0:
        u = np.zeros((Nf, Nf, Nf))
        u[top.glo:top.ghi] = mine.numpy()
        for q in range(1, P):
            lo, hi = owned(c, L, P, H, L - 1, q)
            t = torch.empty((hi - lo, Nf, Nf), dtype=torch.float64)
            dist.recv(t, q)
            u[lo:hi] = t.numpy()
        np.savez(out_path, u=u.reshape(-1), norms=np.array(norms))
    else:
        dist.send(mine, 0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("c,L,nu,P,min_planes,policy,carry", [
    (5, 5, 2, 2, 16, 0, False), (5, 5, 1, 2, 8, 0, False), (3, 6, 2, 3, 8, 0, False), (3, 6, 2, 3, 16, 0, False),
    (5, 5, 2, 2, 16, 1, False), (3, 6, 2, 3, 8, 1, False),
    # carried cycles: the schedule of V(2,2) cycles that end ahead into the next one, its plan variant (policy | 2)
    (5, 5, 2, 2, 16, 0, True), (3, 6, 2, 3, 8, 0, True), (3, 6, 2, 3, 8, 1, True),
    # one launch per leg: the up-leg writes the owned planes only, five halo planes by exchange, the norm completed by the next
    # cycle's down-leg (plan variants policy | 4, | 8, | 12)
    (5, 5, 2, 2, 16, 0, "legs"), (3, 6, 2, 3, 8, 0, "legs"), (3, 6, 2, 3, 8, 1, "legs")])
def test_slab_schedule_over_gloo(tmp_path, monkeypatch, c, L, nu, P, min_planes, policy, carry):
    monkeypatch.setenv("MG3D_SLAB_MIN_PLANES", str(min_planes))  # 8: thin slabs, three distributed levels
    cycles = 4 if carry else 3
    out = str(tmp_path / "res.npz")
    mp.spawn(worker, args=(P, free_port(), c, L, nu, cycles, out, policy, carry is True, carry == "legs"), nprocs=P, join=True)
    got = np.load(out)
    O.lib().orc_set_threads(1)
    want_norms, want_u, _, _ = O.run_problem(c, L, nu, cycles)
    assert np.array_equal(got["u"], want_u)
    np.testing.assert_allclose(got["norms"], want_norms, rtol=1e-12)


def test_partition_properties():
    """Cuts are nested (coarse plane ic and fine plane 2*ic share an owner), cover the level, and leave every rank
    at least a halo's worth of planes."""
    lib = M.lib()
    os.environ.pop("MG3D_SLAB_MIN_PLANES", None)
    for (c, L, nu, P) in [(9, 7, 2, 8), (9, 7, 2, 4), (9, 7, 2, 2), (5, 6, 3, 3), (9, 8, 2, 8), (3, 7, 1, 5)]:
        H = lib.mg3d_slab_halo(nu)
        assert H == 2 * nu + 2
        ld = lib.mg3d_slab_first_level(c, L, P, H)
        assert 1 <= ld < L
        for l in range(ld, L):
            N = (c - 1) * (1 << l) + 1
            prev = 0
            for r in range(P):
                lo, hi = owned(c, L, P, H, l, r)
                assert lo == prev and hi - lo >= H
                prev = hi
                if l > ld:
                    clo, chi = owned(c, L, P, H, l - 1, r)
                    assert lo == 2 * clo and (hi == 2 * chi or (r == P - 1 and hi == N))
                else:
                    assert lo % 2 == 0
            assert prev == N
    assert lib.mg3d_slab_first_level(9, 7, 8, 6) == 4  # 513^3 on 8 GPUs: 129^3 and up distributed (>= 16 planes per rank)
    assert owned(9, 7, 8, 6, 6, 3) == (192, 256)
