"""ctypes binding of oracle/liboracle.so (the CPU checker).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
dp = C.POINTER(C.c_double)
fp = C.POINTER(C.c_float)


def P(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(dp)


def PF(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(fp)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L.orc_bc_func.restype = C.c_double
        L.orc_bc_func.argtypes = [C.c_double] * 3
        L.orc_fill_boundary.argtypes = [dp, C.c_int, C.c_double]
        L.orc_coarse_matrix.argtypes = [dp, C.c_int, C.c_double]
        L.orc_lu_factor.argtypes = [dp, C.c_int]
        L.orc_lu_factor_banded.argtypes = [dp, C.c_int]
        L.orc_lu_factor_banded.restype = None
        L.orc_lu_solve.argtypes = [dp, C.c_int, dp, dp]
        L.orc_smooth_color.argtypes = [dp, dp, C.c_int, C.c_double, C.c_int]
        L.orc_pre_smooth.argtypes = [dp, dp, C.c_int, C.c_double, C.c_int]
        L.orc_post_smooth.argtypes = [dp, dp, C.c_int, C.c_double, C.c_int]
        L.orc_residual.restype = C.c_double
        L.orc_residual.argtypes = [dp, dp, C.c_int, C.c_double, dp]
        L.orc_restrict.argtypes = [dp, C.c_int, dp, C.c_int]
        L.orc_prolong.argtypes = [dp, C.c_int, dp, C.c_int]
        L.orc_l2norm.restype = C.c_double
        L.orc_l2norm.argtypes = [dp, C.c_long]
        L.orc_vcycle.restype = C.c_double
        L.orc_vcycle.argtypes = [C.POINTER(dp)] * 3 + [C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, dp]
        L.orc_fmg_initialize.argtypes = [C.POINTER(dp)] * 3 + [C.c_int, C.c_int, C.c_int, C.c_double, dp]
        L.orc_run_problem.restype = C.c_double
        L.orc_run_problem.argtypes = [C.c_int] * 5 + [dp, dp, dp]
        # single precision / Jacobi / F-cycle variant (oracle/mg3d_oracle_f32.c, parity unpinned)
        L.orc32_fill_boundary.argtypes = [fp, C.c_int, C.c_double]
        L.orc32_jacobi.argtypes = [fp, fp, fp, C.c_int, C.c_float, C.c_float]
        L.orc32_smooth.argtypes = [fp, fp, fp, C.c_int, C.c_float, C.c_float, C.c_int]
        L.orc32_residual.restype = C.c_double
        L.orc32_residual.argtypes = [fp, fp, C.c_int, C.c_float, fp]
        L.orc32_restrict.argtypes = [fp, C.c_int, fp, C.c_int]
        L.orc32_prolong.argtypes = [fp, C.c_int, fp, C.c_int]
        L.orc32_coarse_solve.argtypes = [dp, C.c_int, fp, fp]
        L.orc32_run_problem.restype = C.c_double
        L.orc32_run_problem.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, dp, fp]
        # the mixed-boundary ("electrospray") problem (oracle/mg3d_oracle_es.c, parity unpinned)
        L.orc_es_fill.argtypes = [dp, C.c_int, C.c_double, C.c_void_p, C.c_double]
        L.orc_es_smooth.argtypes = [dp, dp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p]
        L.orc_es_ghost_all.argtypes = [dp, C.c_int, C.c_double, C.c_void_p]
        L.orc_es_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, dp, dp, dp]
        L.orc_es_dirichlet_x0.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]
        L.orc_es_dirichlet_xl.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]
        # one thread unless a test asks for more: on a box whose core count exceeds this process's CPU share (the GPU
        # box: 256 cores, 16 granted) the OpenMP default makes every parallel region of a small problem take milliseconds
        L.orc_set_threads(1)
        _lib = L
    return _lib


def level_sizes(c, L):
    return [(c - 1) * (1 << l) + 1 for l in range(L)]


class Hierarchy:
    """Three level hierarchies u, d, r as the reference allocates them (mg_3d.h:30-48)."""

    def __init__(self, c, L):
        self.c, self.L = c, L
        self.N = level_sizes(c, L)
        self.u = [np.zeros(n ** 3) for n in self.N]
        self.d = [np.zeros(n ** 3) for n in self.N]
        self.r = [np.zeros(n ** 3) for n in self.N]

    def ptrs(self, xs):
        return (dp * self.L)(*[P(a) for a in xs])


def run_problem(c, L, nu, cycles, mode=0, want_u=True):
    N = level_sizes(c, L)[-1]
    norms = np.zeros(cycles)
    u = np.zeros(N ** 3) if want_u else None
    init = C.c_double(0)
    secs = lib().orc_run_problem(c, L, nu, cycles, mode, P(norms), P(u) if want_u else None, C.byref(init))
    return norms, u, init.value, secs


def exact_residual_norm(u, d, N, h):
    """sqrt of the EXACTLY ROUNDED sum of the squared residuals of (u, d): the residual field from the oracle
    (mg_3d.h:819-821; bit-identical to the GPU's diffs whenever u and d are), each square rounded to double as both
    sides do, the sum in extended precision by numpy's pairwise summation (error ~ log2(n) 2^-64: nothing at the 1e-13 the
    GPU reduction is held to).  The oracle's own return value is the reference's SEQUENTIAL sum and carries that sum's
    rounding error (up to n 2^-53): it pins the reference's number, this pins the GPU's reduction."""
    res = np.zeros(N ** 3)
    lib().orc_residual(P(np.ascontiguousarray(u)), P(np.ascontiguousarray(d)), N, h, P(res))
    sq = res * res
    total = np.longdouble(0)
    step = 1 << 24
    for a in range(0, sq.size, step):  # chunked: the extended-precision copy of a 513^3 field would be 2 GB
        total += np.sum(sq[a:a + step].astype(np.longdouble))
    return float(np.sqrt(total))


class EsParams(C.Structure):
    """orc_es_params (same layout as the product's mg3d_es_params); defaults = mg_3d_bkup.c:12-18"""
    _fields_ = [("length", C.c_double), ("capillary_radius", C.c_double), ("extractor_inner", C.c_double),
                ("extractor_outer", C.c_double), ("capillary_voltage", C.c_double), ("extractor_voltage", C.c_double)]

    def __init__(self):
        super().__init__(3e-4, 1.326e-5, 1e-4, 1.4e-4, 0.0, -1350.0)


def es_run(c, L, nu, cycles, params=None):
    p = params or EsParams()
    N = level_sizes(c, L)[-1]
    norms, u, init = np.zeros(cycles), np.zeros(N ** 3), C.c_double(0)
    lib().orc_es_run(c, L, nu, cycles, C.byref(p), P(norms), P(u), C.cast(C.byref(init), dp))
    return norms, u, init.value
