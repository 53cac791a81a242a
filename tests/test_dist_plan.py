"""The exchange plan the multi-GPU transports execute (mg3d_dist_plan, csrc/mg3d_dist.hip), checked WITHOUT a GPU for
every rank of P = 2, 3, 4, 8: the RCCL transport issues exactly these entries (one ncclGroup per phase), so
  * a send without its receive, or with another count, is the hang the first real multi-GPU run would die of;
  * an offset outside the slab, or on other global planes than the peer's, is silent corruption.
What a phase replaces in the reference: the implicit barrier that ends an orphaned `omp for` (mg_3d.h:658-702, 807-842,
961-995, 1007-1145), after which every thread sees its neighbours' planes."""
import ctypes as C

import pytest

import _plan as PL
import multigrid_parallel_amd as M

CONFIGS = [(9, 7, 2), (9, 8, 2), (5, 6, 3), (3, 7, 1), (9, 5, 1), (5, 7, 2)]


def owned(c, L, P, H, level, r):
    lo, hi = C.c_int(0), C.c_int(0)
    assert M.lib().mg3d_slab_owned(c, L, P, H, level, r, C.byref(lo), C.byref(hi)) == 0
    return lo.value, hi.value


def slab(c, L, P, H, level, r):
    glo, ghi = owned(c, L, P, H, level, r)
    h_lo, h_hi = (H if r > 0 else 0), (H if r < P - 1 else 0)
    return dict(ig0=glo - h_lo, ni=ghi - glo + h_lo + h_hi, own_lo=h_lo, own_hi=h_lo + ghi - glo)


# bit 0: coarse levels on rank 0; bit 1: the cycle is carried into the next; bits 2 / 3: one launch per leg (the cycle ends with the
# one-launch up-leg / follows such a cycle)
@pytest.mark.parametrize("policy", [0, 1, 2, 3, 4, 8, 12, 13])
@pytest.mark.parametrize("overlap", [0, 1])
@pytest.mark.parametrize("P", [2, 3, 4, 8])
@pytest.mark.parametrize("c,L,nu", CONFIGS)
def test_every_send_has_its_receive(c, L, nu, P, overlap, policy):
    lib = M.lib()
    H = lib.mg3d_slab_halo(nu)
    ld = lib.mg3d_slab_first_level(c, L, P, H)
    if ld >= L:
        pytest.skip("no level gives every rank enough planes")
    plans = [PL.entries(c, L, P, nu, r, overlap, policy) for r in range(P)]
    nph = {max(e.phase for e in p) + 1 for p in plans}
    assert len(nph) == 1, "every rank walks the same number of phases"
    nph = nph.pop()
    Nc = (c - 1) * (1 << (ld - 1)) + 1
    sent = recvd = 0
    for ph in range(nph):
        per = [[e for e in p if e.phase == ph] for p in plans]
        kinds = {(e.kind, e.level, e.stream) for es in per for e in es}
        assert len(kinds) == 1, f"phase {ph}: all ranks agree on what it is and on which stream it runs"
        kind, level, stream = kinds.pop()
        # overlap off: everything on the compute stream (0).  On: everything on the communication stream with the one
        # communicator -- 2 = overlapped (the u halos), 1 = the compute stream joins at once (critical path)
        assert stream == ((2 if kind in (PL.HALO_U_DOWN, PL.HALO_U_NEXT) else 1) if overlap else 0)
        for r in range(P):
            for e in per[r]:
                assert e.count > 0 and e.plane_elems > 0
                if e.op == PL.SEND:
                    sent += 1
                    m = [x for x in per[e.peer] if x.op == PL.RECV and x.peer == r and x.field == e.field and x.level == e.level]
                    assert len(m) == 1, f"phase {ph}: send {r}->{e.peer} has exactly one receive"
                    m = m[0]
                    assert (m.count, m.plane_elems) == (e.count, e.plane_elems)
                    if e.level >= ld:  # slab to slab: in bounds, same global planes, owned -> halo
                        a, b = slab(c, L, P, H, e.level, r), slab(c, L, P, H, e.level, e.peer)
                        assert 0 <= e.offset and e.offset + e.count <= a["ni"]
                        assert 0 <= m.offset and m.offset + m.count <= b["ni"]
                        assert a["ig0"] + e.offset == b["ig0"] + m.offset
                        assert a["own_lo"] <= e.offset and e.offset + e.count <= a["own_hi"], "only owned planes are sent"
                        assert m.offset + m.count <= b["own_lo"] or m.offset >= b["own_hi"], "only halo planes are overwritten"
                        assert abs(e.peer - r) == 1
                    else:  # replicated arrays: global plane indices on both sides
                        assert e.offset == m.offset and 0 <= e.offset and e.offset + e.count <= Nc
                elif e.op == PL.RECV:
                    recvd += 1
                    m = [x for x in per[e.peer] if x.op == PL.SEND and x.peer == r and x.field == e.field and x.level == e.level]
                    assert len(m) == 1, f"phase {ph}: receive {r}<-{e.peer} has exactly one send"
        if kind in (PL.RHS_ALLGATHER, PL.CORR_BCAST):
            lists = [[(e.op, e.peer, e.field, e.level, e.offset, e.count, e.plane_elems) for e in es] for es in per]
            assert all(x == lists[0] for x in lists), "every rank issues the same broadcasts in the same order"
            assert all(op == PL.BCAST for op, *_ in lists[0])
            if kind == PL.RHS_ALLGATHER:
                assert [x[1] for x in lists[0]] == list(range(P))
                cover = 0
                for _, root, _, _, off, cnt, _ in lists[0]:
                    assert off == cover  # the owners' ranges tile the coarse level in rank order
                    flo, fhi = owned(c, L, P, H, ld, root)
                    assert off == (0 if root == 0 else flo // 2) and off + cnt == (Nc if root == P - 1 else fhi // 2)
                    cover += cnt
                assert cover == Nc
            else:
                assert lists[0] == [(PL.BCAST, 0, 0, ld - 1, 0, Nc, lists[0][0][6])]
        if kind == PL.RHS_GATHER:
            got = sorted((e.offset, e.offset + e.count) for e in per[0])
            flo, fhi = owned(c, L, P, H, ld, 0)
            cover = fhi // 2  # rank 0 restricted these itself
            for a, b in got:
                assert a == cover
                cover = b
            assert cover == Nc and all(e.op == PL.RECV for e in per[0])
        if kind == PL.NORM:
            legs_in, legs_out = nu == 2 and bool(policy & 8), nu == 2 and bool(policy & 4)
            # at the end of the cycle -- unless the next cycle completes it (bit 2); first of all behind such a cycle (bit 3)
            assert ph in ([0] if legs_in else []) + ([] if legs_out else [nph - 1])
            assert [[(e.op, e.offset, e.count) for e in es] for es in per] == [[(PL.ALLGATHER, r, 1)] for r in range(P)]
    assert sent == recvd > 0


@pytest.mark.parametrize("c,L,nu,P", [(9, 7, 2, 8), (5, 6, 3, 3)])
def test_phase_sequence_of_a_cycle(c, L, nu, P):
    """the order dist_enqueue_vcycle walks: per distributed level the coarser right-hand side (what the next level waits
    for), then the level's u halos (for the way up: they travel underneath the coarser levels -- one communicator, one
    in-order stream, so the critical exchange is issued first); the coarse levels; per level the correction's halos; the
    finest u for the next cycle; the norm"""
    lib = M.lib()
    H = lib.mg3d_slab_halo(nu)
    ld = lib.mg3d_slab_first_level(c, L, P, H)
    seq = []
    for e in PL.entries(c, L, P, nu, 1, 0, 0):
        if not seq or seq[-1][0] != e.phase:
            seq.append((e.phase, e.kind, e.level))
    want = []
    for l in range(L - 1, ld - 1, -1):
        want.append((PL.HALO_D, l - 1) if l - 1 >= ld else (PL.RHS_ALLGATHER, ld - 1))
        want.append((PL.HALO_U_DOWN, l))
    for l in range(ld + 1, L):
        want.append((PL.HALO_U_UP, l - 1))
    want.append((PL.HALO_U_NEXT, L - 1))
    want.append((PL.NORM, L - 1))
    assert [(k, l) for _, k, l in seq] == want
    assert [p for p, _, _ in seq] == list(range(len(want)))
    # halo depth: H planes, except the end-of-cycle refresh that leaves the nearest plane alone
    for e in PL.entries(c, L, P, nu, 1, 0, 0):
        if e.kind in (PL.HALO_U_DOWN, PL.HALO_D, PL.HALO_U_UP):
            assert e.count == H
        if e.kind == PL.HALO_U_NEXT:
            assert e.count == H - 1


def test_plan_rejects_bad_arguments():
    f = M.lib().mg3d_dist_plan
    assert f(9, 7, 8, 2, 8, 0, 0, None, 0) < 0   # rank out of range
    assert f(9, 7, 0, 2, 0, 0, 0, None, 0) < 0
    assert f(3, 3, 8, 2, 0, 0, 0, None, 0) < 0   # nothing can be distributed


@pytest.mark.parametrize("P", [2, 3, 4, 8])
@pytest.mark.parametrize("c,L,nu", [(9, 8, 2), (9, 6, 2), (5, 6, 1), (3, 7, 3)])
def test_fp32_variant_every_send_has_its_receive(c, L, nu, P):
    """mg3d32_dist_plan (BASELINE configs[4]: fp32 / Jacobi on slabs, H = nu + 2, 32-float row pitch): the V-cycle from
    every distributed level q -- the F-cycle start runs one from each -- with and without the norm phase.  The RCCL
    branches of csrc/mg3d_f32_dist.hip issue exactly these entries; they have never run on two physical GPUs, so the
    pairing of (own_hi - H, own_hi) with the upper neighbour's lower halo etc. is established here."""
    lib = M.lib()
    H = lib.mg3d32_slab_halo(nu)
    assert H == nu + 2
    ld = lib.mg3d_slab_first_level(c, L, P, H)
    if ld >= L:
        pytest.skip("no level gives every rank enough planes")
    Nc = (c - 1) * (1 << (ld - 1)) + 1
    for q in range(ld, L):
        for want_norm in (0, 1):
            plans = [PL.entries(c, L, P, nu, r, q, want_norm, fn="mg3d32_dist_plan") for r in range(P)]
            nph = {max(e.phase for e in p) + 1 for p in plans}
            assert len(nph) == 1
            nph = nph.pop()
            seq = []
            for ph in range(nph):
                per = [[e for e in p if e.phase == ph] for p in plans]
                kinds = {(e.kind, e.level) for es in per for e in es}
                assert len(kinds) == 1
                seq.append(kinds.pop())
                for r in range(P):
                    for e in per[r]:
                        assert e.stream == 0 and e.count > 0
                        if e.op == PL.SEND:
                            m = [x for x in per[e.peer] if x.op == PL.RECV and x.peer == r and x.field == e.field and x.level == e.level]
                            assert len(m) == 1 and (m[0].count, m[0].plane_elems) == (e.count, e.plane_elems)
                            a, b = slab(c, L, P, H, e.level, r), slab(c, L, P, H, e.level, e.peer)
                            N = (c - 1) * (1 << e.level) + 1
                            assert e.plane_elems == (N + 31) // 32 * 32 * N and e.count == H
                            assert a["ig0"] + e.offset == b["ig0"] + m[0].offset
                            assert a["own_lo"] <= e.offset and e.offset + e.count <= a["own_hi"]
                            assert 0 <= m[0].offset and m[0].offset + m[0].count <= b["ni"]
                            assert m[0].offset + m[0].count <= b["own_lo"] or m[0].offset >= b["own_hi"]
                        elif e.op == PL.RECV:
                            assert len([x for x in per[e.peer] if x.op == PL.SEND and x.peer == r and x.field == e.field and x.level == e.level]) == 1
                if seq[-1][0] == PL.RHS_ALLGATHER:
                    lists = [[(e.peer, e.offset, e.count) for e in es] for es in per]
                    assert all(x == lists[0] for x in lists) and [x[0] for x in lists[0]] == list(range(P))
                    cover = 0
                    for root, off, cnt in lists[0]:
                        flo, fhi = owned(c, L, P, H, ld, root)
                        assert off == cover == (0 if root == 0 else flo // 2) and off + cnt == (Nc if root == P - 1 else fhi // 2)
                        cover += cnt
                    assert cover == Nc
            want = [(PL.HALO_U_NEXT, q)]
            for l in range(q, ld - 1, -1):
                want.append((PL.HALO_U_DOWN, l))
                want.append((PL.HALO_D, l - 1) if l - 1 >= ld else (PL.RHS_ALLGATHER, ld - 1))
            want += [(PL.HALO_U_UP, l - 1) for l in range(ld + 1, q + 1)]
            if want_norm:
                want.append((PL.NORM, q))
            assert seq == want


@pytest.mark.parametrize("P", [2, 3, 4, 8])
@pytest.mark.parametrize("c,L,nu", CONFIGS)
def test_carried_cycle_plan_differs_only_in_the_last_u_exchange(c, L, nu, P):
    """policy bit 1 (a V(2,2) cycle that ends ahead into the next one, csrc/mg3d_ctx.hip "carried cycles"): the exchange
    behind the cycle's last launch refreshes halo planes 1..3 (that launch has used up all of them; the next cycle's
    one-pass + residual + restriction launch reads three either side) instead of 2..H.  Nothing else changes, and for any
    other sweep count the bit changes nothing."""
    lib = M.lib()
    H = lib.mg3d_slab_halo(nu)
    ld = lib.mg3d_slab_first_level(c, L, P, H)
    if ld >= L:
        pytest.skip("no level gives every rank enough planes")
    for r in range(P):
        plain, carried = PL.entries(c, L, P, nu, r, 0, 0), PL.entries(c, L, P, nu, r, 0, 2)
        if nu != 2:
            assert plain == carried
            continue
        assert len(plain) == len(carried)
        sl = slab(c, L, P, H, L - 1, r)
        for a, b in zip(plain, carried):
            if a.kind != PL.HALO_U_NEXT:
                assert a == b
                continue
            assert (a.phase, a.op, a.peer, a.field, a.level, a.plane_elems) == (b.phase, b.op, b.peer, b.field, b.level, b.plane_elems)
            assert a.count == H - 1 and b.count == 3
            if b.op == PL.SEND:  # the three owned planes next to the boundary the peer sits behind
                assert b.offset == (sl["own_hi"] - 3 if b.peer == r + 1 else sl["own_lo"])
            else:  # the three halo planes next to the owned ones
                assert b.offset == (sl["own_hi"] if b.peer == r + 1 else sl["own_lo"] - 3)


@pytest.mark.parametrize("P", [2, 3, 4, 8])
@pytest.mark.parametrize("c,L,nu", CONFIGS)
def test_one_launch_per_leg_plans(c, L, nu, P):
    """policy bits 2 and 3 (V(2,2) cycles with one launch per leg on the finest level, csrc/mg3d_dist.hip): a cycle that ends with
    the one-launch up-leg (4) exchanges halo planes 1..5 of the finest u -- that launch writes the owned planes only, the next
    cycle's one-launch down-leg reads five either side -- and has no NORM phase of its own; the cycle behind it (8) opens with the
    NORM phase of its predecessor.  Everything else is the plain plan, and for any other sweep count the bits change nothing."""
    lib = M.lib()
    H = lib.mg3d_slab_halo(nu)
    ld = lib.mg3d_slab_first_level(c, L, P, H)
    if ld >= L:
        pytest.skip("no level gives every rank enough planes")
    for r in range(P):
        plain = PL.entries(c, L, P, nu, r, 0, 0)
        if nu != 2:
            assert all(PL.entries(c, L, P, nu, r, 0, pol) == plain for pol in (4, 8, 12))
            continue
        sl = slab(c, L, P, H, L - 1, r)
        body = [e for e in plain if e.kind != PL.NORM]
        norm = [e for e in plain if e.kind == PL.NORM]
        for pol in (4, 8, 12):
            got = PL.entries(c, L, P, nu, r, 0, pol)
            shift = 1 if pol & 8 else 0
            g_norm = [e for e in got if e.kind == PL.NORM]
            g_body = [e for e in got if e.kind != PL.NORM]
            want_norm_phases = ([0] if pol & 8 else []) + ([] if pol & 4 else [max(e.phase for e in plain) + shift])
            assert [e.phase for e in g_norm] == want_norm_phases
            assert all(e._replace(phase=0) == norm[0]._replace(phase=0) for e in g_norm)
            assert len(g_body) == len(body)
            for a, b in zip(body, g_body):
                assert b.phase == a.phase + shift
                if a.kind != PL.HALO_U_NEXT or not pol & 4:
                    assert a._replace(phase=0) == b._replace(phase=0)
                    continue
                assert (a.op, a.peer, a.field, a.level, a.plane_elems, a.stream) == (b.op, b.peer, b.field, b.level, b.plane_elems, b.stream)
                assert a.count == H - 1 and b.count == 5
                if b.op == PL.SEND:  # the five owned planes next to the boundary the peer sits behind
                    assert b.offset == (sl["own_hi"] - 5 if b.peer == r + 1 else sl["own_lo"])
                else:  # the five halo planes next to the owned ones
                    assert b.offset == (sl["own_hi"] if b.peer == r + 1 else sl["own_lo"] - 5)
