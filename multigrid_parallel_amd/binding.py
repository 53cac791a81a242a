"""ctypes mirror of include/mg3d.h plus a `Solver` class that follows the reference's Solver* facade
(mg_3d.h:107-144, 275-293, 1412-1467).  No compute happens here; every call goes into libmg3d.so and
raises `Mg3dError` when the library (or a GPU) is missing -- there is no CPU fallback."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
dp = C.POINTER(C.c_double)
fp = C.POINTER(C.c_float)

MG3D_U, MG3D_D, MG3D_R = 0, 1, 2
STAGES = 7
_ERR = {1: "bad argument", 2: "no device", 3: "HIP error", 4: "allocation failed", 5: "bad state"}


class Mg3dError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mg3d error {code} ({_ERR.get(code, '?')}): {msg}")
        self.code = code


def lib_path():
    # MG3D_LIB_PATH: an alternative build of the same library (A/B measurements of kernel variants on one box)
    return os.environ.get("MG3D_LIB_PATH") or os.path.join(_HERE, "lib", "libmg3d.so")


_lib = None

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against include/mg3d.h
SIGNATURES = {
    "mg3d_last_error": (C.c_char_p, []),
    "mg3d_stage_name": (C.c_char_p, [C.c_int]),
    "mg3d_device_count": (C.c_int, []),
    "mg3d_ctx_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_void_p)]),
    "mg3d_ctx_destroy": (C.c_int, [C.c_void_p]),
    "mg3d_ctx_num_levels": (C.c_int, [C.c_void_p]),
    "mg3d_ctx_level_n": (C.c_int, [C.c_void_p, C.c_int]),
    "mg3d_ctx_level_h": (C.c_double, [C.c_void_p, C.c_int]),
    "mg3d_ctx_set_smooth_iters": (C.c_int, [C.c_void_p, C.c_int]),
    "mg3d_ctx_set_keep_residual": (C.c_int, [C.c_void_p, C.c_int]),
    "mg3d_ctx_build_coarse": (C.c_int, [C.c_void_p, C.c_double]),
    "mg3d_ctx_set_lu": (C.c_int, [C.c_void_p, dp]),
    "mg3d_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_int, dp]),
    "mg3d_download": (C.c_int, [C.c_void_p, C.c_int, C.c_int, dp]),
    "mg3d_zero": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mg3d_sync": (C.c_int, [C.c_void_p]),
    "mg3d_device_view": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                   C.POINTER(C.c_long)]),
    "mg3d_smooth": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mg3d_residual": (C.c_int, [C.c_void_p, C.c_int, C.c_int, dp]),
    "mg3d_smooth_residual": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, dp]),
    "mg3d_smooth_restrict": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mg3d_restrict": (C.c_int, [C.c_void_p, C.c_int]),
    "mg3d_prolong": (C.c_int, [C.c_void_p, C.c_int]),
    "mg3d_coarse_solve": (C.c_int, [C.c_void_p]),
    "mg3d_l2norm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, dp]),
    "mg3d_vcycle": (C.c_int, [C.c_void_p, C.c_int, dp]),
    "mg3d_vcycles": (C.c_int, [C.c_void_p, C.c_int, dp]),
    "mg3d_fmg_initialize": (C.c_int, [C.c_void_p]),
    "mg3d_fill_boundary": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mg3d_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "mg3d_timing_reset": (C.c_int, [C.c_void_p]),
    "mg3d_timing_get": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), dp]),
    "mg3d_kernel_name": (C.c_char_p, [C.c_int]),
    "mg3d_kernel_time_get": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), dp]),
    "mg3d_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mg3d_dist_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                   C.POINTER(C.c_void_p)]),
    "mg3d_dist_destroy": (C.c_int, [C.c_void_p]),
    "mg3d_dist_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mg3d_dist_first_level": (C.c_int, [C.c_void_p]),
    "mg3d_dist_halo": (C.c_int, [C.c_void_p]),
    "mg3d_dist_carried_cycles": (C.c_int, [C.c_void_p]),
    "mg3d_dist_legs_cycles": (C.c_int, [C.c_void_p]),
    "mg3d_dist_set_keep_residual": (C.c_int, [C.c_void_p, C.c_int]),
    "mg3d_dist_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "mg3d_ctx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "mg3d_ctx_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "mg3d_option_name": (C.c_char_p, [C.c_int]),
    "mg3d32_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "mg3d32_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "mg3d32_kernel_name": (C.c_char_p, [C.c_int]),
    "mg3d32_kernel_time_get": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), dp]),
    "mg3d_dist_build_coarse": (C.c_int, [C.c_void_p, C.c_double]),
    "mg3d_dist_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_int, dp]),
    "mg3d_dist_download": (C.c_int, [C.c_void_p, C.c_int, C.c_int, dp]),
    "mg3d_dist_vcycles": (C.c_int, [C.c_void_p, C.c_int, dp]),
    "mg3d_dist_sync": (C.c_int, [C.c_void_p]),
    "mg3d_slab_halo": (C.c_int, [C.c_int]),
    "mg3d_slab_first_level": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mg3d_slab_owned": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int)]),
    "mg3d_debug_tiny_stamps": (C.c_int, [C.POINTER(C.c_longlong)]),
    "mg3d_dist_plan": (C.c_int, [C.c_int] * 7 + [C.c_void_p, C.c_int]),
    "mg3d32_dist_plan": (C.c_int, [C.c_int] * 7 + [C.c_void_p, C.c_int]),
    "mg3d_dist_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "mg3d_dist_timing_get": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "mg3d_host_smooth": (C.c_int, [dp, dp, C.c_int, C.c_double, C.c_int, C.c_int]),
    "mg3d_host_residual": (C.c_int, [dp, dp, C.c_int, C.c_double, dp, dp]),
    "mg3d_host_restrict": (C.c_int, [dp, C.c_int, dp, C.c_int]),
    "mg3d_host_prolong": (C.c_int, [dp, C.c_int, dp, C.c_int]),
    "mg3d_host_lu_solve": (C.c_int, [dp, C.c_int, dp, dp]),
    "mg3d_host_vcycle": (C.c_int, [C.POINTER(dp), C.POINTER(dp), C.POINTER(dp), C.c_double, C.c_int, C.c_int, C.c_int,
                                   C.c_int, dp, dp, C.POINTER(C.c_int), dp]),
    "mg3d_bc_func": (C.c_double, [C.c_double, C.c_double, C.c_double]),
    "mg3d_fill_boundary_host": (None, [dp, C.c_int, C.c_double]),
    "mg3d_coarse_matrix": (None, [dp, C.c_int, C.c_double]),
    "mg3d_lu_factor": (None, [dp, C.c_int]),
    "mg3d_l2norm_host": (C.c_double, [dp, C.c_long]),
    "mg3d_smooth_edges_host": (None, [dp, C.c_int]),
    "mg3d_write_vtk": (C.c_int, [C.c_char_p, dp, C.c_double, C.c_int]),
    "mg3d_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "mg3d_host_free": (C.c_int, [C.c_void_p]),
    # single precision / damped Jacobi / F-cycle variant (parity unpinned)
    "mg3d32_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]),
    "mg3d32_destroy": (C.c_int, [C.c_void_p]),
    "mg3d32_level_n": (C.c_int, [C.c_void_p, C.c_int]),
    "mg3d32_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_int, fp]),
    "mg3d32_download": (C.c_int, [C.c_void_p, C.c_int, C.c_int, fp]),
    "mg3d32_zero": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mg3d32_sync": (C.c_int, [C.c_void_p]),
    "mg3d32_fill_boundary": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mg3d32_smooth": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mg3d32_residual": (C.c_int, [C.c_void_p, C.c_int, C.c_int, dp]),
    "mg3d32_restrict": (C.c_int, [C.c_void_p, C.c_int]),
    "mg3d32_prolong": (C.c_int, [C.c_void_p, C.c_int]),
    "mg3d32_coarse_solve": (C.c_int, [C.c_void_p]),
    "mg3d32_vcycles": (C.c_int, [C.c_void_p, C.c_int, dp]),
    "mg3d32_fmg_initialize": (C.c_int, [C.c_void_p]),
    "mg3d_es_default_params": (C.c_int, [C.c_void_p]),
    "mg3d_es_coarse_matrix": (None, [dp, C.c_int, C.c_double, C.c_void_p]),
    "mg3d_es_setup": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mg3d_es_smooth": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mg3d_es_vcycles": (C.c_int, [C.c_void_p, C.c_int, dp]),
    "mg3d32_slab_halo": (C.c_int, [C.c_int]),
    "mg3d32_dist_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p,
                                     C.c_int, C.POINTER(C.c_void_p)]),
    "mg3d32_dist_destroy": (C.c_int, [C.c_void_p]),
    "mg3d32_dist_first_level": (C.c_int, [C.c_void_p]),
    "mg3d32_dist_halo": (C.c_int, [C.c_void_p]),
    "mg3d32_dist_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mg3d32_dist_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_int, fp]),
    "mg3d32_dist_download": (C.c_int, [C.c_void_p, C.c_int, C.c_int, fp]),
    "mg3d32_dist_zero": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mg3d32_dist_fill_boundary": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mg3d32_dist_vcycles": (C.c_int, [C.c_void_p, C.c_int, dp]),
    "mg3d32_dist_fmg_initialize": (C.c_int, [C.c_void_p]),
    "mg3d32_dist_sync": (C.c_int, [C.c_void_p]),
}


class EsParams(C.Structure):
    """mg3d_es_params: the mixed-boundary problem of mg_3d_bkup.c:12-18 (defaults = the reference's #defines)."""
    _fields_ = [("length", C.c_double), ("capillary_radius", C.c_double), ("extractor_inner_radius", C.c_double),
                ("extractor_outer_radius", C.c_double), ("capillary_voltage", C.c_double), ("extractor_voltage", C.c_double)]

    @staticmethod
    def default():
        p = EsParams()
        check(lib().mg3d_es_default_params(C.byref(p)))
        return p


def lib():
    """Load libmg3d.so (built in-tree by `make -C multigrid_parallel_amd/csrc`)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise Mg3dError(2, f"{path} not built; run __graft_entry__.build() (there is no CPU fallback)")
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name, None)
            if fn is None and os.environ.get("MG3D_LIB_PATH"):
                continue  # an older build used as an A/B reference may lack newer entry points
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def P(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], "need contiguous float64"
    return a.ctypes.data_as(dp)


def check(rc):
    if rc != 0:
        raise Mg3dError(rc, lib().mg3d_last_error().decode(errors="replace"))


class Solver:
    """Device-resident solver context.  Method names follow the reference facade:

    Solver(c, L, nu)                     ~ SolverInitialize(argv = c, L, nu)      mg_3d.h:107
    get_details()                        ~ SolverGetDetails (N, h; builds the LU)   mg_3d.h:275
    setup_boundary_conditions()          ~ SolverSetupBoundaryConditions            mg_3d.h:1412
    get_initial_residual()               ~ SolverGetInitialResidual                 mg_3d.h:1430
    lin_solve()                          ~ SolverLinSolve (one V-cycle, its norm)   mg_3d.h:1415
    get_residual()                       ~ SolverGetResidual                        mg_3d.h:1425
    finalize()                           ~ SolverFinalize                           mg_3d.h:1452
    """

    def __init__(self, coarse_pts, num_levels, smooth_iters, grid_length=1.0):
        self._h = C.c_void_p()
        self.L = lib()
        check(self.L.mg3d_ctx_create(coarse_pts, num_levels, smooth_iters, grid_length, C.byref(self._h)))
        self.c, self.num_levels, self.nu = coarse_pts, num_levels, smooth_iters
        self.N = self.L.mg3d_ctx_level_n(self._h, num_levels - 1)
        self.h = self.L.mg3d_ctx_level_h(self._h, num_levels - 1)

    # -- lifetime
    def finalize(self):
        if self._h:
            check(self.L.mg3d_ctx_destroy(self._h))
            self._h = C.c_void_p()

    close = finalize

    def __del__(self):
        try:
            self.finalize()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.finalize()

    def set_keep_residual(self, keep=True):
        check(self.L.mg3d_ctx_set_keep_residual(self._h, int(keep)))

    def set_option(self, key, value):
        """mg3d_ctx_set_option: launch / schedule policy by key (include/mg3d.h has the table)."""
        check(self.L.mg3d_ctx_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int(0)
        check(self.L.mg3d_ctx_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def options(self):
        out, i = {}, 0
        while self.L.mg3d_option_name(i):
            k = self.L.mg3d_option_name(i).decode()
            out[k] = self.get_option(k)
            i += 1
        return out

    # -- geometry
    def level_n(self, level):
        return self.L.mg3d_ctx_level_n(self._h, level)

    def level_h(self, level):
        return self.L.mg3d_ctx_level_h(self._h, level)

    # -- facade
    def get_details(self, coarse_h=None):
        """Builds + factors the coarsest operator with spacing h*2^(L-1) (mg_3d.h:287) unless given."""
        ch = self.h * (1 << (self.num_levels - 1)) if coarse_h is None else coarse_h
        check(self.L.mg3d_ctx_build_coarse(self._h, ch))
        return self.N, self.h

    def set_lu(self, LU):
        check(self.L.mg3d_ctx_set_lu(self._h, P(LU)))

    def setup_boundary_conditions(self, field=MG3D_D, level=None):
        level = self.num_levels - 1 if level is None else level
        n = self.level_n(level)
        a = self.download(field, level)
        self.L.mg3d_fill_boundary_host(P(a), n, self.level_h(level))
        self.upload(field, level, a)

    def setup_test_problem(self):
        """test_mg_3d.c:11-29: BC values on the faces of d and of u, interior zero."""
        self.get_details()
        self.zero(MG3D_U, self.num_levels - 1)
        self.zero(MG3D_D, self.num_levels - 1)
        self.setup_boundary_conditions(MG3D_D)
        self.setup_boundary_conditions(MG3D_U)

    def get_initial_residual(self):
        return self.l2norm(MG3D_D, self.num_levels - 1)

    def lin_solve(self):
        return self.vcycle(self.num_levels - 1)

    def get_residual(self):
        return self.residual(self.num_levels - 1, store=False)

    # -- data
    def upload(self, field, level, host):
        n = self.level_n(level)
        host = np.ascontiguousarray(host, dtype=np.float64).reshape(-1)
        assert host.size == n ** 3
        check(self.L.mg3d_upload(self._h, field, level, P(host)))

    def download(self, field, level):
        n = self.level_n(level)
        out = np.empty(n ** 3)
        check(self.L.mg3d_download(self._h, field, level, P(out)))
        return out

    def zero(self, field, level):
        check(self.L.mg3d_zero(self._h, field, level))

    def sync(self):
        check(self.L.mg3d_sync(self._h))

    # -- operators
    def smooth(self, level, post, iters):
        check(self.L.mg3d_smooth(self._h, level, int(post), iters))

    def residual(self, level, store=True, want_norm=True):
        nrm = C.c_double(0)
        check(self.L.mg3d_residual(self._h, level, int(store), C.byref(nrm) if want_norm else None))
        return nrm.value

    def smooth_residual(self, level, post, iters, store=True, want_norm=True):
        nrm = C.c_double(0)
        check(self.L.mg3d_smooth_residual(self._h, level, int(post), iters, int(store),
                                          C.byref(nrm) if want_norm else None))
        return nrm.value

    def smooth_restrict(self, level, iters):
        check(self.L.mg3d_smooth_restrict(self._h, level, iters))

    def restrict(self, level):
        check(self.L.mg3d_restrict(self._h, level))

    def prolong(self, level):
        check(self.L.mg3d_prolong(self._h, level))

    def coarse_solve(self):
        check(self.L.mg3d_coarse_solve(self._h))

    def l2norm(self, field, level):
        nrm = C.c_double(0)
        check(self.L.mg3d_l2norm(self._h, field, level, C.byref(nrm)))
        return nrm.value

    def vcycle(self, level=None, want_norm=True):
        level = self.num_levels - 1 if level is None else level
        nrm = C.c_double(0)
        check(self.L.mg3d_vcycle(self._h, level, C.byref(nrm) if want_norm else None))
        return nrm.value

    def vcycles(self, count):
        norms = np.zeros(count)
        check(self.L.mg3d_vcycles(self._h, count, P(norms)))
        return norms

    def fmg_initialize(self):
        check(self.L.mg3d_fmg_initialize(self._h))

    # -- the mixed-boundary ("electrospray") problem, csrc/mg3d_es.hip
    def es_setup(self, params=None):
        self.es = params or EsParams.default()
        check(self.L.mg3d_es_setup(self._h, C.byref(self.es)))
        return self.es

    def es_smooth(self, level, post, iters):
        check(self.L.mg3d_es_smooth(self._h, level, int(post), iters))

    def es_vcycles(self, count):
        norms = np.zeros(count)
        check(self.L.mg3d_es_vcycles(self._h, count, P(norms)))
        return norms

    def fill_boundary(self, field, level):
        check(self.L.mg3d_fill_boundary(self._h, field, level))

    # -- timing
    def timing_enable(self, on=True):
        check(self.L.mg3d_timing_enable(self._h, int(on)))

    def timing_reset(self):
        check(self.L.mg3d_timing_reset(self._h))

    def kernel_times(self):
        out = {}
        for l in range(self.num_levels):
            for k in range(64):  # up to MG3D_NUM_KERNELS: the library names the ones it has
                if self.L.mg3d_kernel_name(k) == b"?":
                    break
                calls, secs = C.c_int(0), C.c_double(0)
                check(self.L.mg3d_kernel_time_get(self._h, l, k, C.byref(calls), C.byref(secs)))
                if calls.value:
                    out[(l, self.L.mg3d_kernel_name(k).decode())] = (calls.value, secs.value)
        return out

    def timing(self):
        out = {}
        for l in range(self.num_levels):
            for s in range(STAGES):
                calls, secs = C.c_int(0), C.c_double(0)
                check(self.L.mg3d_timing_get(self._h, l, s, C.byref(calls), C.byref(secs)))
                out[(l, self.L.mg3d_stage_name(s).decode())] = (calls.value, secs.value)
        return out


class DistSolver:
    """V-cycle on i-slabs: rank `rank` of `nranks` (one process per GPU, RCCL), or -- unique_id=None -- all ranks
    virtual in this process on one GPU (loopback transport; used to verify the decomposition)."""

    def __init__(self, coarse_pts, num_levels, smooth_iters, rank=0, nranks=1, unique_id=None, device=0,
                 grid_length=1.0):
        self.L = lib()
        self._h = C.c_void_p()
        uid = None if unique_id is None else C.c_char_p(bytes(unique_id))
        check(self.L.mg3d_dist_create(coarse_pts, num_levels, smooth_iters, grid_length, rank, nranks, uid, device,
                                      C.byref(self._h)))
        self.c, self.num_levels, self.nu, self.rank, self.nranks = coarse_pts, num_levels, smooth_iters, rank, nranks
        self.N = (coarse_pts - 1) * (1 << (num_levels - 1)) + 1
        self.h = grid_length / (self.N - 1)
        self.first_level = self.L.mg3d_dist_first_level(self._h)
        self.halo = self.L.mg3d_dist_halo(self._h)

    def carried_cycles(self):
        return self.L.mg3d_dist_carried_cycles(self._h)

    def legs_cycles(self):
        return self.L.mg3d_dist_legs_cycles(self._h)

    def comm_info(self):
        """(ranks in the RCCL communicator, overlap on/off, HIP device)"""
        n, ov, dev = C.c_int(0), C.c_int(0), C.c_int(0)
        check(self.L.mg3d_dist_comm_info(self._h, C.byref(n), C.byref(ov), C.byref(dev)))
        return n.value, bool(ov.value), dev.value

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(lib().mg3d_comm_unique_id(buf))
        return buf.raw

    def close(self):
        if self._h:
            check(self.L.mg3d_dist_destroy(self._h))
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def level_n(self, level):
        return (self.c - 1) * (1 << level) + 1

    def set_keep_residual(self, keep=True):
        check(self.L.mg3d_dist_set_keep_residual(self._h, int(keep)))

    def set_option(self, key, value):
        check(self.L.mg3d_dist_set_option(self._h, key.encode(), int(value)))

    def setup_test_problem(self):
        """test_mg_3d.c:11-29 on the full grid; every rank takes its slab."""
        check(self.L.mg3d_dist_build_coarse(self._h, self.h * (1 << (self.num_levels - 1))))
        full = np.zeros(self.N ** 3)
        self.L.mg3d_fill_boundary_host(P(full), self.N, self.h)
        fin = self.num_levels - 1
        self.upload(MG3D_D, fin, full)
        self.upload(MG3D_U, fin, full)
        return float(np.sqrt((full * full).sum()))

    def upload(self, field, level, host_full):
        host_full = np.ascontiguousarray(host_full, dtype=np.float64).reshape(-1)
        assert host_full.size == self.level_n(level) ** 3
        check(self.L.mg3d_dist_upload(self._h, field, level, P(host_full)))

    def download(self, field, level, out=None):
        n = self.level_n(level)
        out = np.zeros(n ** 3) if out is None else out
        check(self.L.mg3d_dist_download(self._h, field, level, P(out)))
        return out

    def vcycles(self, count):
        norms = np.zeros(count)
        check(self.L.mg3d_dist_vcycles(self._h, count, P(norms)))
        return norms

    def sync(self):
        check(self.L.mg3d_dist_sync(self._h))

    def timing_enable(self, on=True):
        check(self.L.mg3d_dist_timing_enable(self._h, int(on)))

    def timing(self):
        """per-cycle means since timing_enable(True): whole cycle, exchanges the compute stream waits for at once
        (critical path), exchanges that run overlapped (the u halos, underneath the launches in between), the
        replicated / rank-0 coarse levels; kernels on the distributed levels are what is left of the cycle"""
        ms = (C.c_double * 4)()
        n = C.c_int(0)
        check(self.L.mg3d_dist_timing_get(self._h, ms, C.byref(n)))
        k = max(1, n.value)
        return {"cycles": n.value, "cycle_ms": ms[0] / k, "exchange_ms": ms[1] / k, "exchange_overlapped_ms": ms[2] / k,
                "replicated_ms": ms[3] / k, "kernels_ms": (ms[0] - ms[1] - ms[3]) / k}


def PF(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], "need contiguous float32"
    return a.ctypes.data_as(fp)


class Solver32:
    """The single-precision / damped-Jacobi / F-cycle variant (BASELINE configs[4]; parity unpinned: semantics in
    csrc/mg3d_f32.hip, restated in plain C by the test infrastructure).  Same field and level numbering as `Solver`."""

    def __init__(self, coarse_pts, num_levels, smooth_iters, omega=6.0 / 7.0, grid_length=1.0):
        self.L = lib()
        self._h = C.c_void_p()
        check(self.L.mg3d32_create(coarse_pts, num_levels, smooth_iters, omega, grid_length, C.byref(self._h)))
        self.c, self.num_levels, self.nu, self.omega = coarse_pts, num_levels, smooth_iters, omega
        self.N = (coarse_pts - 1) * (1 << (num_levels - 1)) + 1

    def close(self):
        if self._h:
            self.L.mg3d32_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, key, value):
        """mg3d32_set_option: "pairs", "fuse", "carry" (1 on, 0 off)."""
        check(self.L.mg3d32_set_option(self._h, key.encode(), int(value)))

    def timing_enable(self, on=True):
        check(self.L.mg3d32_timing_enable(self._h, int(on)))

    def kernel_times(self):
        """{kernel name: (launches, seconds)} of the finest level's launches since timing_enable(True)"""
        out, k = {}, 0
        while self.L.mg3d32_kernel_name(k):
            n, sec = C.c_int(0), C.c_double(0)
            check(self.L.mg3d32_kernel_time_get(self._h, k, C.byref(n), C.byref(sec)))
            if n.value:
                out[self.L.mg3d32_kernel_name(k).decode()] = (n.value, sec.value)
            k += 1
        return out

    def level_n(self, level):
        return self.L.mg3d32_level_n(self._h, level)

    def upload(self, field, level, host):
        n = self.level_n(level)
        assert host.size == n ** 3
        check(self.L.mg3d32_upload(self._h, field, level, PF(np.ascontiguousarray(host, dtype=np.float32))))

    def download(self, field, level):
        out = np.empty(self.level_n(level) ** 3, dtype=np.float32)
        check(self.L.mg3d32_download(self._h, field, level, PF(out)))
        return out

    def zero(self, field, level):
        check(self.L.mg3d32_zero(self._h, field, level))

    def sync(self):
        check(self.L.mg3d32_sync(self._h))

    def fill_boundary(self, field, level):
        check(self.L.mg3d32_fill_boundary(self._h, field, level))

    def smooth(self, level, iters):
        check(self.L.mg3d32_smooth(self._h, level, iters))

    def residual(self, level, store=True, want_norm=True):
        n = C.c_double(0)
        check(self.L.mg3d32_residual(self._h, level, 1 if store else 0, C.byref(n) if want_norm else None))
        return n.value

    def restrict(self, level):
        check(self.L.mg3d32_restrict(self._h, level))

    def prolong(self, level):
        check(self.L.mg3d32_prolong(self._h, level))

    def coarse_solve(self):
        check(self.L.mg3d32_coarse_solve(self._h))

    def setup_test_problem(self, fmg=False):
        """test_mg_3d.c:11-29: boundary values on the faces of the finest u and d; with fmg=True on the faces of d
        on every level (the F-cycle start of mg_dirichlet_analytic.c:771-806 reads them), then that start."""
        top = self.num_levels - 1
        for l in range(self.num_levels):  # a clean hierarchy, as after the reference's calloc (mg_3d.h:44)
            for f in (MG3D_U, MG3D_D, MG3D_R):
                self.zero(f, l)
        if fmg:
            for l in range(self.num_levels):
                self.fill_boundary(MG3D_D, l)
            check(self.L.mg3d32_fmg_initialize(self._h))
        else:
            self.fill_boundary(MG3D_U, top)
            self.fill_boundary(MG3D_D, top)

    def vcycles(self, count):
        norms = np.zeros(count)
        check(self.L.mg3d32_vcycles(self._h, count, P(norms)))
        return norms


class DistSolver32:
    """`Solver32` on i-slabs: rank `rank` of `nranks` (one process per GPU, RCCL), or -- unique_id=None -- all ranks
    virtual in this process on one GPU (loopback transport)."""

    def __init__(self, coarse_pts, num_levels, smooth_iters, omega=6.0 / 7.0, rank=0, nranks=1, unique_id=None, device=0,
                 grid_length=1.0):
        self.L = lib()
        self._h = C.c_void_p()
        uid = None if unique_id is None else C.c_char_p(bytes(unique_id))
        check(self.L.mg3d32_dist_create(coarse_pts, num_levels, smooth_iters, omega, grid_length, rank, nranks, uid,
                                        device, C.byref(self._h)))
        self.c, self.num_levels, self.nu, self.rank, self.nranks = coarse_pts, num_levels, smooth_iters, rank, nranks
        self.N = (coarse_pts - 1) * (1 << (num_levels - 1)) + 1
        self.first_level = self.L.mg3d32_dist_first_level(self._h)
        self.halo = self.L.mg3d32_dist_halo(self._h)

    def close(self):
        if self._h:
            check(self.L.mg3d32_dist_destroy(self._h))
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def level_n(self, level):
        return (self.c - 1) * (1 << level) + 1

    def comm_info(self):
        n, dev = C.c_int(0), C.c_int(0)
        check(self.L.mg3d32_dist_comm_info(self._h, C.byref(n), C.byref(dev)))
        return n.value, dev.value

    def upload(self, field, level, host_full):
        host_full = np.ascontiguousarray(host_full, dtype=np.float32).reshape(-1)
        assert host_full.size == self.level_n(level) ** 3
        check(self.L.mg3d32_dist_upload(self._h, field, level, PF(host_full)))

    def download(self, field, level, out=None):
        out = np.zeros(self.level_n(level) ** 3, dtype=np.float32) if out is None else out
        check(self.L.mg3d32_dist_download(self._h, field, level, PF(out)))
        return out

    def zero(self, field, level):
        check(self.L.mg3d32_dist_zero(self._h, field, level))

    def fill_boundary(self, field, level):
        check(self.L.mg3d32_dist_fill_boundary(self._h, field, level))

    def setup_test_problem(self, fmg=False):
        """As Solver32.setup_test_problem, every rank on its slab."""
        top = self.num_levels - 1
        for l in range(self.num_levels):
            for f in (MG3D_U, MG3D_D, MG3D_R):
                self.zero(f, l)
        if fmg:
            for l in range(self.num_levels):
                self.fill_boundary(MG3D_D, l)
            check(self.L.mg3d32_dist_fmg_initialize(self._h))
        else:
            self.fill_boundary(MG3D_U, top)
            self.fill_boundary(MG3D_D, top)

    def vcycles(self, count):
        norms = np.zeros(count)
        check(self.L.mg3d32_dist_vcycles(self._h, count, P(norms)))
        return norms

    def sync(self):
        check(self.L.mg3d32_dist_sync(self._h))
