"""The drop-in boundary: the reference's own drivers, unchanged, against include/mg_3d.h + libmg3d.so.

CPU (here): oracle/Makefile compiles /root/reference/test_mg_3d.c and test_mg_3d_dirichlet.c untouched
against the repo's headers (link gate), and without a GPU they fail loudly rather than falling back.
GPU box: the binaries built here travel in oracle/_ref/ and are run; their printed residual histories
must equal the reference's known answers."""
import os
import re
import subprocess

import numpy as np
import pytest

import _oracle as O

ROOT = O.ROOT
REF = "/root/reference"
BIN1 = os.path.join(ROOT, "oracle", "_ref", "dropin_test_mg_3d")
BIN2 = os.path.join(ROOT, "oracle", "_ref", "dropin_test_mg_3d_dirichlet")
BIN3 = os.path.join(ROOT, "oracle", "_ref", "dropin_test_rb_gs_3d")  # built with -DMG3D_LEGACY_TIMINGINFO
BIN4 = os.path.join(ROOT, "oracle", "_ref", "dropin_test_lu")
HAVE_GPU = os.path.exists("/dev/kfd")


def run(cmd, cwd, threads=None):
    env = dict(os.environ)
    if threads:
        env["OMP_NUM_THREADS"] = str(threads)
    return subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree only exists in the build container")
def test_reference_drivers_compile_and_link_unchanged(tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "dropin"], check=True, capture_output=True)
    assert all(os.path.exists(b) for b in (BIN1, BIN2, BIN3, BIN4))
    out = subprocess.run(["ldd", BIN1], capture_output=True, text=True).stdout
    assert "libmg3d.so" in out and "multigrid_parallel_amd/lib" in out
    # argument handling is the reference's: usage + exit 1 (mg_3d.h:109-113), abort on non-pow2 (:123)
    r = run([BIN1], tmp_path)
    assert r.returncode == 1 and r.stdout.startswith("Usage:")
    r = run([BIN1, "4", "3", "2"], tmp_path)
    assert r.returncode == -6 or r.returncode == 134
    for b in (BIN2, BIN3, BIN4):
        r = run([b], tmp_path)
        assert r.returncode == 1 and r.stdout.startswith("Usage:")
    # the legacy TimingInfo generation also serves the dirichlet driver (its tInfo[level][stage] shape)
    with open(os.path.join(REF, "test_mg_3d_dirichlet.c")) as src:
        subprocess.run(["gcc", "-x", "c", "-fopenmp", "-w", "-DMG3D_LEGACY_TIMINGINFO", "-fsyntax-only",
                        "-I" + os.path.join(ROOT, "include"), "-"], stdin=src, check=True, cwd=tmp_path)


@pytest.mark.skipif(HAVE_GPU or not os.path.exists(BIN1), reason="needs the CPU-only container")
def test_drivers_fail_loudly_without_gpu(tmp_path):
    for b in (BIN1, BIN2):
        r = run([b, "5", "3", "2"], tmp_path)
        assert r.returncode == 1
        assert "no CPU fallback" in r.stderr
    for b in (BIN3, BIN4):
        r = run([b, "5"], tmp_path)
        assert r.returncode == 1
        assert "no CPU fallback" in r.stderr


def history(stdout):
    return [float(m) for m in re.findall(r"Residual Norm:\s*(\S+)", stdout)]


KNOWN_5_5_2 = [74651.9, 9198.35, 1219.39, 170.177, 24.6618, 3.68103, 0.563252, 0.0880884, 0.0140466, 0.00227868,
               0.000375223, 6.25855e-05, 1.05534e-05, 1.79591e-06, 3.0789e-07]
KNOWN_DIR_5_5_2 = [74831.4, 9392.75, 1372.13, 265.208, 69.895, 21.3226, 6.76706, 2.16709, 0.695417, 0.223269]


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 4])
def test_reference_test_mg_3d_runs_on_gpu(tmp_path, threads):
    if not os.path.exists(BIN1):
        pytest.skip("oracle/_ref/dropin_test_mg_3d was not built (no reference tree at build time)")
    r = run([BIN1, "5", "5", "2"], tmp_path, threads)
    assert r.returncode == 0, r.stderr
    assert history(r.stdout) == pytest.approx(KNOWN_5_5_2, rel=2e-6)  # 6 printed digits
    assert f"Max threads: {threads}" in r.stdout
    err = float(re.search(r"Error norm:\s*(\S+)", r.stdout).group(1))
    assert err == pytest.approx(1.60434e-09, rel=1e-5)
    assert "LEVEL 4" in r.stdout and re.search(r"Smoother1\s+15\s", r.stdout)
    assert os.path.getsize(tmp_path / "diff2.vtk") > 65 ** 3 * 10


@pytest.mark.gpu
def test_reference_test_mg_3d_dirichlet_runs_on_gpu(tmp_path):
    if not os.path.exists(BIN2):
        pytest.skip("oracle/_ref/dropin_test_mg_3d_dirichlet was not built")
    r = run([BIN2, "5", "5", "2"], tmp_path, 1)
    assert r.returncode == 0, r.stderr
    assert history(r.stdout) == pytest.approx(KNOWN_DIR_5_5_2, rel=2e-6)
    assert "Error norm: 0.000000" in r.stdout  # the driver's `errNorm = diff*diff` overwrite (:90)
    assert re.search(r"CalcResidual2\s+10\s", r.stdout)


@pytest.mark.gpu
def test_api_surface_driver(tmp_path):
    """tests/c/api_surface.c (ours) compiled on the box against include/, compared with the oracle."""
    exe = tmp_path / "api_surface"
    subprocess.run(["gcc", "-O2", "-fopenmp", "-I" + os.path.join(ROOT, "include"), "-o", str(exe),
                    os.path.join(ROOT, "tests", "c", "api_surface.c"), "-L" + os.path.join(ROOT, "multigrid_parallel_amd", "lib"),
                    "-Wl,-rpath," + os.path.join(ROOT, "multigrid_parallel_amd", "lib"), "-lmg3d", "-lm"], check=True)
    r = run([str(exe)], tmp_path, 2)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    val = lambda tag: [[float(x) for x in m.split()] for m in re.findall(rf"^{tag} (.*)$", out, flags=re.M)]

    # replay the same sequence with the oracle
    lib = O.lib()
    lib.orc_set_threads(1)
    st = [12345.0]

    def rnd():
        st[0] = (st[0] * 16807.0) % 2147483647.0
        return 2.0 * (st[0] / 2147483647.0) - 1.0

    N, Nc = 9, 5
    h = 1.0 / (N - 1)
    v, f = np.zeros(N ** 3), np.zeros(N ** 3)
    for p in range(N ** 3):
        v[p] = rnd()
        f[p] = rnd()
    res, dc = np.zeros(N ** 3), np.zeros(Nc ** 3)
    lib.orc_pre_smooth(O.P(v), O.P(f), N, h, 2)
    lib.orc_post_smooth(O.P(v), O.P(f), N, h, 1)
    nrm = lib.orc_residual(O.P(v), O.P(f), N, h, O.P(res))
    lib.orc_restrict(O.P(res), N, O.P(dc), Nc)
    lib.orc_prolong(O.P(dc), Nc, O.P(v), N)
    sv = 0.0
    for p in range(N ** 3):
        sv += v[p] * (1 + p % 7)
    sd = 0.0
    for p in range(Nc ** 3):
        sd += dc[p] * (1 + p % 5)
    got = val("OPS")[0]
    assert got[0] == pytest.approx(nrm, rel=1e-13)
    assert got[1] == sv and got[2] == sd and got[3] == lib.orc_l2norm(O.P(f), N ** 3)

    # 9-argument vcycle on caller-owned hierarchies, correct coarse spacing
    c, L, nu = 3, 3, 2
    H = O.Hierarchy(c, L)
    Nf, hf = H.N[-1], 1.0 / (H.N[-1] - 1)
    LU = np.zeros(27 * 27)
    lib.orc_coarse_matrix(O.P(LU), c, hf * 4)
    lib.orc_lu_factor(O.P(LU), 27)
    lib.orc_fill_boundary(O.P(H.u[-1]), Nf, hf)
    want = [lib.orc_vcycle(H.ptrs(H.u), H.ptrs(H.d), H.ptrs(H.r), hf, L - 1, L, nu, Nf, O.P(LU)) for _ in range(4)]
    assert [x[0] for x in val("VC9")] == pytest.approx(want, rel=1e-13)
    su = 0.0
    for p in range(Nf ** 3):
        su += H.u[-1][p] * (1 + p % 11)
    assert val("VC9U")[0][0] == su
    assert val("TIMED")[0] == [4.0, 4.0]
    assert re.search(r"beta\s+3\s", out)

    # Solver facade
    norms, _, init, _ = O.run_problem(5, 3, 2, 4)
    lin = [x[0] for x in val("LIN")]
    assert val("INIT")[0][0] == init
    assert lin == pytest.approx(list(norms), rel=1e-13)
    assert val("RES3")[0][0] == pytest.approx(norms[2], rel=1e-13)
    assert re.search(r"LEVEL 2\n.*\n\s+Smoother1\s+1\s", out)  # reset before the 4th cycle


@pytest.mark.gpu
def test_facade_sees_host_writes_between_solves(tmp_path):
    """tests/c/host_dirty.c: rhs[] and grid[] are changed through the raw pointers of SolverGetDetails after
    SolverGetResidual / SolverResetTimingInfo / SolverSyncHost and announced after the fact with SolverMarkHostDirty
    (write-then-call order: the flag-only call must not clobber the write); the following cycles must start from those
    writes, as they do in the reference (mg_3d.h:278-279 hands out the solver's own arrays)."""
    exe = tmp_path / "host_dirty"
    subprocess.run(["gcc", "-O2", "-fopenmp", "-I" + os.path.join(ROOT, "include"), "-o", str(exe),
                    os.path.join(ROOT, "tests", "c", "host_dirty.c"), "-L" + os.path.join(ROOT, "multigrid_parallel_amd", "lib"),
                    "-Wl,-rpath," + os.path.join(ROOT, "multigrid_parallel_amd", "lib"), "-lmg3d", "-lm"], check=True)
    r = run([str(exe)], tmp_path, 2)
    assert r.returncode == 0, r.stderr
    val = lambda tag: [float(m) for m in re.findall(rf"^{tag} (\S+)$", r.stdout, flags=re.M)]
    lib = O.lib()
    lib.orc_set_threads(1)
    c, L, nu = 5, 3, 2
    H = O.Hierarchy(c, L)
    N, h = H.N[-1], 1.0 / (H.N[-1] - 1)
    LU = np.zeros(c ** 6)
    lib.orc_coarse_matrix(O.P(LU), c, h * (1 << (L - 1)))
    lib.orc_lu_factor(O.P(LU), c ** 3)
    lib.orc_fill_boundary(O.P(H.d[-1]), N, h)
    lib.orc_fill_boundary(O.P(H.u[-1]), N, h)
    cyc = lambda: lib.orc_vcycle(H.ptrs(H.u), H.ptrs(H.d), H.ptrs(H.r), h, L - 1, L, nu, N, O.P(LU))
    A = [cyc() for _ in range(3)]
    R = lib.orc_residual(O.P(H.u[-1]), O.P(H.d[-1]), N, h, None)
    mid = (N * N + N + 1) * (N // 2)
    keep = [a.copy() for a in H.u], [a.copy() for a in H.d]
    unchanged = cyc()  # what the fourth cycle would return had the caller not written anything
    for dst, src in zip(H.u + H.d, keep[0] + keep[1]):
        dst[:] = src
    H.d[-1][mid] = 250.0
    H.u[-1][mid + 1] += 0.125
    B = [cyc() for _ in range(3)]
    H.d[-1][mid - N] = -125.0
    Cc = [cyc() for _ in range(2)]
    H.u[-1][mid - 1] -= 0.25
    D = [cyc()]
    H.d[-1][mid + N] = 60.0  # written BEFORE SolverMarkHostDirty(): a flag-only call must not overwrite it
    E = [cyc()]
    H.u[-1][mid + 2] += 0.5
    H.d[-1][mid + 2] = -30.0
    F = [cyc()]
    assert val("A") == pytest.approx(A, rel=1e-13) and val("R") == pytest.approx([R], rel=1e-13)
    assert val("B") == pytest.approx(B, rel=1e-13)
    assert abs(B[0] - unchanged) > 1e-3 * unchanged  # the writes really entered the solve
    assert val("C") == pytest.approx(Cc, rel=1e-13) and val("D") == pytest.approx(D, rel=1e-13)
    assert val("E") == pytest.approx(E, rel=1e-13) and val("F") == pytest.approx(F, rel=1e-13)
    su = 0.0
    for p in range(N ** 3):
        su += H.u[-1][p] * (1 + p % 13)
    assert val("U")[0] == su


@pytest.mark.gpu
@pytest.mark.parametrize("c,L,nu", [(5, 4, 2), (3, 5, 1)])  # the configurations tests/golden/vcycle.npz holds fmg_* for
def test_reference_test_mg_3d_fmg_start(tmp_path, c, L, nu):
    """MG3D_USE_FMG=1: the unchanged test_mg_3d.c starts from the F-cycle guess of mg_dirichlet_analytic.c:771-806 (what
    that monolith's fifth argument `useFMG` selects, :70-80, :984-988).  The first printed norms are the `fmg_*` golden
    sequence, generated by replaying the reference's own operators (oracle/gen_golden.py)."""
    if not os.path.exists(BIN1):
        pytest.skip("oracle/_ref/dropin_test_mg_3d was not built")
    V = np.load(os.path.join(ROOT, "tests", "golden", "vcycle.npz"))
    want = V[f"norms_fmg_{c}_{L}_{nu}"]
    env = dict(os.environ, OMP_NUM_THREADS="2", MG3D_USE_FMG="1")
    r = subprocess.run([BIN1, str(c), str(L), str(nu)], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "Doing FMG Initialization....done" in r.stdout
    got = history(r.stdout)
    assert got[:len(want)] == pytest.approx(list(want), rel=6e-6)  # %g: 6 significant digits
    plain = subprocess.run([BIN1, str(c), str(L), str(nu)], cwd=tmp_path, env=dict(os.environ, OMP_NUM_THREADS="2"),
                           capture_output=True, text=True, timeout=600)
    assert "FMG" not in plain.stdout
    assert history(plain.stdout)[0] > 1.5 * got[0]  # the start really changed the run


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["default", "carried", "legs"])
def test_reference_test_mg_3d_129_cubed(tmp_path, monkeypatch, schedule):
    """BASELINE configs[1]: `9 5 2` = 129^3, V(2,2), through the unchanged reference driver.  "carried": the solve
    loop's cycles run ahead into each other (carried cycles, the default from 130 to 449 points per side); "legs": one
    launch per leg with the next cycle's down-leg run ahead behind every SolverLinSolve (the default from 160 points per
    side) -- same history, same error norm: the final SolverGetDetails / error check sees the finished cycle's own u."""
    if not os.path.exists(BIN1):
        pytest.skip("oracle/_ref/dropin_test_mg_3d was not built")
    if schedule == "carried":
        monkeypatch.setenv("MG3D_CARRY_MIN", "66")
    if schedule == "legs":
        monkeypatch.setenv("MG3D_LEGS_MIN", "66")
    known = [600893, 73400.9, 9566.66, 1305, 183.942, 26.5851, 3.92421, 0.590481, 0.0904885, 0.014113, 0.00223841,
             0.000360659, 5.89564e-05, 9.7633e-06, 1.63505e-06]
    r = run([BIN1, "9", "5", "2"], tmp_path, 2)
    assert r.returncode == 0, r.stderr
    assert history(r.stdout) == pytest.approx(known, rel=2e-6)
    assert float(re.search(r"Error norm:\s*(\S+)", r.stdout).group(1)) == pytest.approx(1.85423e-09, rel=1e-5)


def rb_gs_history(N, tol=1e-6):
    """test_rb_gs_3d.c:56-101 replayed with the oracle: one pre + one post sweep and the norm per iteration."""
    lib = O.lib()
    h = 1.0 / (N - 1)
    u, d = np.zeros(N ** 3), np.zeros(N ** 3)
    lib.orc_fill_boundary(O.P(u), N, h)
    init = lib.orc_residual(O.P(u), O.P(d), N, h, None)
    out, nrm = [], 1e9
    while nrm > init * tol:
        lib.orc_pre_smooth(O.P(u), O.P(d), N, h, 1)
        lib.orc_post_smooth(O.P(u), O.P(d), N, h, 1)
        old, nrm = nrm, lib.orc_residual(O.P(u), O.P(d), N, h, None)
        out.append((nrm, nrm / old))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("N,threads", [(18, 1), (18, 4), (33, 3)])
def test_reference_test_rb_gs_3d_runs_on_gpu(tmp_path, N, threads):
    """The smoother-only driver, unchanged: every thread of the team calls preSmoother /
    postSmoother / calculateResidual (test_rb_gs_3d.c:70-80), N is not of the form 2^k+1."""
    if not os.path.exists(BIN3):
        pytest.skip("oracle/_ref/dropin_test_rb_gs_3d was not built")
    r = run([BIN3, str(N)], tmp_path, threads)
    assert r.returncode == 0, r.stderr
    want = rb_gs_history(N)
    got = re.findall(r"^\s*(\d+)\s+Residual Norm:\s*(\S+)\s+ResidRatio:\s*(\S+)", r.stdout, flags=re.M)
    assert [int(g[0]) for g in got] == list(range(1, len(want) + 1))
    assert [float(g[1]) for g in got] == pytest.approx([w[0] for w in want], rel=6e-6)  # %20g: 6 significant digits
    assert [float(g[2]) for g in got[1:]] == pytest.approx([w[1] for w in want[1:]], rel=6e-6)
    assert f"Max OMP threads: {threads}" in r.stdout
    assert f"Number of calls: {len(want)}" in r.stdout


@pytest.mark.gpu
def test_reference_test_rb_gs_3d_50_cubed(tmp_path):
    """The size of the reference's only published numbers (red_black_gs_scalability.txt): 50^3, tol 1e-6.
    The unmodified operators of the current tree need 1303 iterations (0.28129 / 0.991804), see DESIGN.md."""
    if not os.path.exists(BIN3):
        pytest.skip("oracle/_ref/dropin_test_rb_gs_3d was not built")
    r = run([BIN3, "50"], tmp_path, 2)
    assert r.returncode == 0, r.stderr
    last = re.findall(r"^\s*(\d+)\s+Residual Norm:\s*(\S+)\s+ResidRatio:\s*(\S+)", r.stdout, flags=re.M)[-1]
    assert (int(last[0]), last[1], last[2]) == (1303, "0.28129", "0.991804")


@pytest.mark.gpu
@pytest.mark.parametrize("N", [3, 5, 9])
def test_reference_test_lu_runs_on_gpu(tmp_path, N):
    """test_lu.c unchanged: constructCoarseMatrixA + convertToLU_InPlace on the host, solveWithLU on the GPU,
    result written as VTK (9 significant digits)."""
    if not os.path.exists(BIN4):
        pytest.skip("oracle/_ref/dropin_test_lu was not built")
    r = run([BIN4, str(N)], tmp_path, 1)
    assert r.returncode == 0, r.stderr
    assert "Time taken for LU solve:" in r.stdout
    lib = O.lib()
    n, h = N ** 3, 1.0 / (N - 1)
    A, b, x = np.zeros(n * n), np.zeros(n), np.zeros(n)
    lib.orc_coarse_matrix(O.P(A), N, h)
    lib.orc_lu_factor(O.P(A), n)
    lib.orc_fill_boundary(O.P(b), N, h)
    lib.orc_lu_solve(O.P(A), n, O.P(b), O.P(x))
    text = (tmp_path / "output.vtk").read_text()
    body = text[text.index("LOOKUP_TABLE"):].split("\n", 1)[1].split()
    got = np.array([float(t) for t in body[:n]])
    assert np.allclose(got, x, rtol=2e-8, atol=1e-300)  # "%10.8e", values in memory order (postprocess.h:37-44)
