"""ctypes view of the product's exchange plan (mg3d_dist_plan / mg3d32_dist_plan, include/mg3d.h): the list of transfers the
library's own transports execute.  Test infrastructure only."""
import ctypes as C
from collections import namedtuple

import multigrid_parallel_amd as M

HALO_U_DOWN, HALO_D, RHS_ALLGATHER, RHS_GATHER, CORR_BCAST, HALO_U_UP, HALO_U_NEXT, NORM = range(8)
SEND, RECV, BCAST, ALLGATHER = range(4)
KIND_NAMES = ["HALO_U_DOWN", "HALO_D", "RHS_ALLGATHER", "RHS_GATHER", "CORR_BCAST", "HALO_U_UP", "HALO_U_NEXT", "NORM"]


class Xfer(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("phase", "kind", "op", "peer", "field", "level", "offset", "count")] + \
               [("plane_elems", C.c_longlong), ("stream", C.c_int)]


E = namedtuple("E", "phase kind op peer field level offset count plane_elems stream")


def entries(c, L, P, nu, rank, overlap=0, policy=0, fn="mg3d_dist_plan"):
    f = getattr(M.lib(), fn)
    n = f(c, L, P, nu, rank, overlap, policy, None, 0)
    assert n >= 0, f"{fn} failed: {n}"
    buf = (Xfer * max(n, 1))()
    assert f(c, L, P, nu, rank, overlap, policy, C.cast(buf, C.c_void_p), n) == n
    return [E(*(getattr(buf[i], k) for k in E._fields)) for i in range(n)]


def phases(c, L, P, nu, rank, overlap=0, policy=0, fn="mg3d_dist_plan"):
    """entries grouped by phase number: {phase: [E, ...]}"""
    out = {}
    for e in entries(c, L, P, nu, rank, overlap, policy, fn):
        out.setdefault(e.phase, []).append(e)
    return out
