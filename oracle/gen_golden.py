#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference, compiled by oracle/Makefile
into oracle/_ref/libmg3d_ref.so (the reference's operators are external functions of its
header, so ctypes can call them one by one).

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference to have been
compiled):  make -C oracle && python oracle/gen_golden.py
The fixtures are data (seeded inputs + the reference's outputs); no reference source is stored.

Reference entry points called (file:line in /root/reference):
  preSmoother mg_3d.h:640, postSmoother :711, calculateResidual :794, restrictResidual :844,
  prolongateAndCorrectError :1000, setupBoundaryConditions :1147, constructCoarseMatrixA :147,
  convertToLU_InPlace gauss_elim.h:9, solveWithLU gauss_elim.h:31, vcycle mg_3d.h:1242,
  Solver* mg_3d.h:107-144,275-293,1412-1467 (through ref_shim.c:ref_run_problem).
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
SEED = 12345

dp = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(dp)


def load():
    lib = C.CDLL(os.path.join(HERE, "_ref", "libmg3d_ref.so"))
    lib.calculateResidual.restype = C.c_double
    lib.calculateResidual.argtypes = [dp, dp, C.c_int, C.c_double, dp]
    for f in (lib.preSmoother, lib.postSmoother):
        f.restype = None
        f.argtypes = [dp, dp, C.c_int, C.c_double, C.c_int]
    lib.restrictResidual.restype = None
    lib.restrictResidual.argtypes = [dp, C.c_int, dp, C.c_int]
    lib.prolongateAndCorrectError.restype = None
    lib.prolongateAndCorrectError.argtypes = [dp, C.c_int, dp, C.c_int]
    lib.setupBoundaryConditions.restype = None
    lib.setupBoundaryConditions.argtypes = [dp, C.c_int, C.c_double]
    lib.constructCoarseMatrixA.restype = None
    lib.constructCoarseMatrixA.argtypes = [dp, C.c_int, C.c_double]
    lib.convertToLU_InPlace.restype = None
    lib.convertToLU_InPlace.argtypes = [dp, C.c_int]
    lib.solveWithLU.restype = None
    lib.solveWithLU.argtypes = [dp, C.c_int, dp, dp]
    lib.vcycle.restype = C.c_double
    lib.vcycle.argtypes = [C.POINTER(dp), C.POINTER(dp), C.POINTER(dp), C.c_double, C.c_int, C.c_int, C.c_int,
                           C.c_int, dp]
    lib.SolverInitialize.restype = None
    lib.SolverInitialize.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    lib.SolverFinalize.restype = None
    lib.ref_run_problem.restype = C.c_double
    lib.ref_run_problem.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, dp]
    lib.GetL2NormOfVector.restype = C.c_double
    lib.GetL2NormOfVector.argtypes = [dp, C.c_int]
    return lib


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    os.environ["OMP_NUM_THREADS"] = "1"
    lib = load()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(SEED)
    ops = {}

    # ---- smoother / residual on seeded uniform(-1,1) fields ------------------------------
    for N in (5, 9, 17, 33):
        h = 1.0 / (N - 1)
        v0 = rng.uniform(-1, 1, N ** 3)
        d0 = rng.uniform(-1, 1, N ** 3)
        ops[f"sm_v0_{N}"] = v0
        ops[f"sm_d0_{N}"] = d0
        for name, fn, it in (("pre1", lib.preSmoother, 1), ("pre2", lib.preSmoother, 2),
                             ("post1", lib.postSmoother, 1), ("post3", lib.postSmoother, 3)):
            v = v0.copy()
            fn(P(v), P(d0), N, h, it)
            ops[f"sm_{name}_{N}"] = v
        res = np.zeros(N ** 3)
        nrm = lib.calculateResidual(P(v0), P(d0), N, h, P(res))
        ops[f"res_r_{N}"] = res
        ops[f"res_norm_{N}"] = np.array([nrm, lib.calculateResidual(P(v0), P(d0), N, h, None)])
        ops[f"l2_{N}"] = np.array([lib.GetL2NormOfVector(P(d0), N ** 3)])

    # ---- grid transfer -------------------------------------------------------------------
    for Nc in (3, 5, 9, 17):
        Nf = 2 * Nc - 1
        r = rng.uniform(-1, 1, Nf ** 3)  # boundary of r deliberately non-zero: exercises injection
        dc = rng.uniform(-1, 1, Nc ** 3)  # must be fully overwritten
        ops[f"rs_r_{Nf}"] = r
        lib.restrictResidual(P(r), Nf, P(dc), Nc)
        ops[f"rs_dc_{Nc}"] = dc
        ec = rng.uniform(-1, 1, Nc ** 3)  # coarse boundary non-zero: exercises fine-boundary update
        ef = rng.uniform(-1, 1, Nf ** 3)
        ops[f"pr_ec_{Nc}"] = ec
        ops[f"pr_ef0_{Nf}"] = ef.copy()
        lib.prolongateAndCorrectError(P(ec), Nc, P(ef), Nf)
        ops[f"pr_ef_{Nf}"] = ef

    # ---- boundary fill, coarse matrix, LU ------------------------------------------------
    for N in (3, 5, 9):
        h = 0.125 if N == 9 else 1.0 / (N - 1) / 3.0
        v = rng.uniform(-1, 1, N ** 3)
        ops[f"bc_v0_{N}"] = v.copy()
        lib.setupBoundaryConditions(P(v), N, h)
        ops[f"bc_v_{N}"] = v
        ops[f"bc_h_{N}"] = np.array([h])
    for N in (3, 5):
        n = N ** 3
        h = 1.0 / (N - 1) / 7.0
        A = np.zeros(n * n)
        lib.constructCoarseMatrixA(P(A), N, h)
        ops[f"cm_A_{N}"] = A.copy()
        ops[f"cm_h_{N}"] = np.array([h])
        lib.convertToLU_InPlace(P(A), n)
        ops[f"lu_LU_{N}"] = A.copy()
        b = rng.uniform(-1, 1, n)
        x = np.zeros(n)
        lib.solveWithLU(P(A), n, P(b), P(x))
        ops[f"lu_b_{N}"] = b
        ops[f"lu_x_{N}"] = x
    # c = 9: only the solve vector and a digest of the 4.25 MB factor
    n = 729
    A = np.zeros(n * n)
    lib.constructCoarseMatrixA(P(A), 9, 0.125)
    lib.convertToLU_InPlace(P(A), n)
    b = rng.uniform(-1, 1, n)
    x = np.zeros(n)
    lib.solveWithLU(P(A), n, P(b), P(x))
    ops["lu_b_9"] = b
    ops["lu_x_9"] = x
    ops["lu_sha_9"] = np.frombuffer(bytes.fromhex(digest(A)), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "operators.npz"), **ops)

    # ---- whole V-cycle histories through the Solver* API (test_mg_3d.c protocol) ------------
    vc = {}
    for (c, L, nu, cycles, keep_u) in ((3, 3, 1, 6, True), (3, 5, 2, 15, True), (5, 3, 3, 8, True),
                                       (9, 2, 2, 10, True), (5, 5, 2, 15, False), (9, 5, 2, 15, False)):
        N = (c - 1) * (1 << (L - 1)) + 1
        norms = np.zeros(cycles)
        u = np.zeros(N ** 3)
        init = C.c_double(0)
        lib.ref_run_problem(c, L, nu, cycles, P(norms), P(u), C.byref(init))
        key = f"{c}_{L}_{nu}"
        vc[f"norms_{key}"] = norms
        vc[f"init_{key}"] = np.array([init.value])
        vc[f"usha_{key}"] = np.frombuffer(bytes.fromhex(digest(u)), dtype=np.uint8)
        if keep_u:
            vc[f"u_{key}"] = u
        else:
            vc[f"usample_{key}"] = u[::97].copy()
        print(key, "N=", N, "last norm", norms[-1])

    # ---- test_mg_3d_dirichlet.c protocol (legacy driver; fine-h coarse matrix quirk, :40) ----
    # Its four stale call sites do not compile against the current headers, so the protocol is
    # replayed here call by call against the reference library (9-arg vcycle with h = GRID_LENGTH/(N-1)).
    for (c, L, nu, cycles) in ((5, 5, 2, 10), (3, 4, 2, 8)):
        argv = (C.c_char_p * 4)(b"ref", str(c).encode(), str(L).encode(), str(nu).encode())
        lib.SolverInitialize(4, argv)  # allocates the global tInfo that vcycle() writes into
        N = (c - 1) * (1 << (L - 1)) + 1
        h = 1.0 / (N - 1)
        lv = [np.zeros(((c - 1) * (1 << l) + 1) ** 3) for l in range(L)]
        lf = [np.zeros_like(a) for a in lv]
        lr = [np.zeros_like(a) for a in lv]
        arr = lambda xs: (dp * L)(*[P(a) for a in xs])
        n0 = c ** 3
        A = np.zeros(n0 * n0)
        lib.constructCoarseMatrixA(P(A), c, h)  # test_mg_3d_dirichlet.c:40 (finest h)
        lib.convertToLU_InPlace(P(A), n0)
        lib.setupBoundaryConditions(P(lv[L - 1]), N, h)  # :44
        init = lib.calculateResidual(P(lv[L - 1]), P(lf[L - 1]), N, h, None)  # :51
        U, F, R = arr(lv), arr(lf), arr(lr)
        norms = np.array([lib.vcycle(U, F, R, h, L - 1, L, nu, N, P(A)) for _ in range(cycles)])
        key = f"dir_{c}_{L}_{nu}"
        vc[f"norms_{key}"] = norms
        vc[f"init_{key}"] = np.array([init])
        vc[f"usha_{key}"] = np.frombuffer(bytes.fromhex(digest(lv[L - 1])), dtype=np.uint8)
        vc[f"usample_{key}"] = lv[L - 1][::97].copy()
        print(key, norms)
    # ---- SolverFMGInitialize (spec: mg_dirichlet_analytic.c:771-806 / commented mg_3d.h:1364-1404) replayed with
    # the reference's own operators on the test_mg_3d.c problem, followed by three V-cycles
    for (c, L, nu) in ((5, 4, 2), (3, 5, 1)):
        argv = (C.c_char_p * 4)(b"ref", str(c).encode(), str(L).encode(), str(nu).encode())
        lib.SolverInitialize(4, argv)
        N = (c - 1) * (1 << (L - 1)) + 1
        h = 1.0 / (N - 1)
        lv = [np.zeros(((c - 1) * (1 << l) + 1) ** 3) for l in range(L)]
        lf = [np.zeros_like(a) for a in lv]
        lr = [np.zeros_like(a) for a in lv]
        arr = lambda xs: (dp * L)(*[P(a) for a in xs])
        n0 = c ** 3
        A = np.zeros(n0 * n0)
        lib.constructCoarseMatrixA(P(A), c, h * (1 << (L - 1)))
        lib.convertToLU_InPlace(P(A), n0)
        lib.setupBoundaryConditions(P(lf[L - 1]), N, h)
        lib.setupBoundaryConditions(P(lv[L - 1]), N, h)
        U, F, R = arr(lv), arr(lf), arr(lr)
        Nl, hl = c, 1.0 / (c - 1)
        lib.setupBoundaryConditions(P(lv[0]), Nl, hl)           # :780
        lib.solveWithLU(P(A), n0, P(lf[0]), P(lv[0]))            # :783
        for l in range(1, L):
            Nc, Nl, hl = Nl, 2 * Nl - 1, hl * 0.5
            lib.prolongateAndCorrectError(P(lv[l - 1]), Nc, P(lv[l]), Nl)   # :795
            lib.setupBoundaryConditions(P(lv[l]), Nl, hl)                    # :798
            lv[l - 1][:] = 0.0                                               # :801
            lib.vcycle(U, F, R, hl, l, L, nu, Nl, P(A))                      # :804
        key = f"fmg_{c}_{L}_{nu}"
        vc[f"u0_{key}"] = lv[L - 1].copy()
        vc[f"norms_{key}"] = np.array([lib.vcycle(U, F, R, h, L - 1, L, nu, N, P(A)) for _ in range(3)])
        vc[f"u_{key}"] = lv[L - 1].copy()
        print(key, vc[f"norms_{key}"])
    np.savez_compressed(os.path.join(OUT, "vcycle.npz"), **vc)
    for f in ("operators.npz", "vcycle.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    sys.exit(main())
