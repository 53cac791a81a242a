"""CPU-side checks of the drop-in boundary: libmg3d.so loads, exports every symbol include/mg3d.h declares,
the ctypes table covers the same set, host-only helpers match the oracle, and compute entry points fail
loudly (MG3D_ERR_NO_DEVICE) instead of falling back when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import _oracle as O
import multigrid_parallel_amd as M
from multigrid_parallel_amd.binding import SIGNATURES, P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mg3d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mg3d(?:32)?_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    L = C.CDLL(M.lib_path())
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/mg3d.h but not exported"
    assert sorted(SIGNATURES) == names


def test_no_oracle_in_product():
    """The product must not link, load or reference anything under oracle/."""
    import subprocess
    out = subprocess.run(["ldd", M.lib_path()], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "multigrid_parallel_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".c", ".h", "Makefile")):
                assert "oracle" not in open(os.path.join(root, f)).read().lower(), f
    for f in os.listdir(os.path.join(ROOT, "include")):
        assert "oracle" not in open(os.path.join(ROOT, "include", f)).read().lower(), f


def test_host_helpers_match_oracle():
    L = M.lib()
    for N in (3, 5, 9):
        h = 1.0 / (N - 1) / 3.0
        rng = np.random.default_rng(N)
        a = rng.uniform(-1, 1, N ** 3)
        b = a.copy()
        L.mg3d_fill_boundary_host(P(a), N, h)
        O.lib().orc_fill_boundary(O.P(b), N, h)
        assert np.array_equal(a, b)
        assert L.mg3d_l2norm_host(P(a), N ** 3) == O.lib().orc_l2norm(O.P(a), N ** 3)
    assert L.mg3d_bc_func(0.3, 0.7, 0.9) == O.lib().orc_bc_func(0.3, 0.7, 0.9)
    for N in (3, 5):
        n = N ** 3
        A, B = np.zeros(n * n), np.zeros(n * n)
        L.mg3d_coarse_matrix(P(A), N, 0.37)
        O.lib().orc_coarse_matrix(O.P(B), N, 0.37)
        assert np.array_equal(A, B)
        L.mg3d_lu_factor(P(A), n)
        O.lib().orc_lu_factor(O.P(B), n)
        assert np.array_equal(A, B)


def test_lu_factor_band_skip_equals_dense_sweep_c9():
    n = 729
    A, B = np.zeros(n * n), np.zeros(n * n)
    M.lib().mg3d_coarse_matrix(P(A), 9, 0.125)
    O.lib().orc_coarse_matrix(O.P(B), 9, 0.125)
    M.lib().mg3d_lu_factor(P(A), n)
    O.lib().orc_lu_factor(O.P(B), n)
    assert A.tobytes() == B.tobytes()


def test_lu_factor_wide_band_c17_equals_oracle_byte_for_byte():
    """c = 17 (4913 unknowns, half-band 289): the product's band-limited host factor against the oracle's, bytes (signed
    zeros outside the band included); both are pinned to the dense sweep at c <= 9."""
    n = 17 ** 3
    A, B = np.zeros(n * n), np.zeros(n * n)
    M.lib().mg3d_coarse_matrix(P(A), 17, 0.0625)
    O.lib().orc_coarse_matrix(O.P(B), 17, 0.0625)
    M.lib().mg3d_lu_factor(P(A), n)
    O.lib().orc_set_threads(4)
    O.lib().orc_lu_factor_banded(O.P(B), n)
    O.lib().orc_set_threads(1)
    assert A.tobytes() == B.tobytes()


def test_vtk_writer_format(tmp_path):
    N, h = 3, 0.5
    g = np.arange(27, dtype=np.float64) / 7
    p = tmp_path / "o.vtk"
    assert M.lib().mg3d_write_vtk(str(p).encode(), P(g), h, N) == 0
    lines = p.read_text().split("\n")
    assert lines[:6] == ["# vtk DataFile Version 2.0", "Potential data", "ASCII", "DATASET STRUCTURED_GRID",
                         "DIMENSIONS 3 3 3", "POINTS 27 float"]
    assert lines[6] == "0.00000000e+00 0.00000000e+00 0.00000000e+00"
    assert lines[7] == "0.00000000e+00 0.00000000e+00 5.00000000e-01"
    assert lines[6 + 27] == "" and lines[6 + 28] == "POINT_DATA 27"
    assert lines[6 + 31] == "0.00000000e+00" and lines[6 + 32] == "%10.8e" % (1 / 7)


def test_edge_smoothing_matches_reference_formulas():
    N = 5
    u = np.random.default_rng(3).uniform(-1, 1, (N, N, N))
    w = u.copy().reshape(-1)
    M.lib().mg3d_smooth_edges_host(P(w), N)
    w = w.reshape(N, N, N)
    # an edge along j at i=0,k=0 (mg_3d.h:312-317) and the (0,0,0) corner (mg_3d.h:397-399)
    for j in range(1, N - 1):
        assert w[0, j, 0] == 0.5 * (u[0, j, 1] + u[1, j, 0])
    assert w[0, 0, 0] == (1. / 3) * (w[0, 0, 1] + w[0, 1, 0] + w[1, 0, 0])
    assert np.array_equal(w[1:-1, 1:-1, :], u[1:-1, 1:-1, :])


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_compute_fails_loudly_without_gpu():
    L = M.lib()
    assert L.mg3d_device_count() == 0
    h = C.c_void_p()
    assert L.mg3d_ctx_create(5, 3, 2, 1.0, C.byref(h)) == 2  # MG3D_ERR_NO_DEVICE
    assert b"no CPU fallback" in L.mg3d_last_error()
    a = np.zeros(27)
    assert L.mg3d_host_smooth(P(a), P(a), 3, 0.5, 1, 0) == 2
    with pytest.raises(M.Mg3dError):
        M.Solver(5, 3, 2)


def test_lu_division_sequence_is_exact(tmp_path):
    """The twice-refined quotient of csrc/mg3d_kernels.hip:lu_div equals the IEEE quotient bit for bit
    (tests/c/div_check.c: hard significands, quotients beside rounding boundaries, random operands)."""
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c", "div_check.c")
    exe = str(tmp_path / "div_check")
    with open("/proc/cpuinfo") as f:
        hw_fma = " fma " in f.read()
    flags = ["-mfma"] if hw_fma else []
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", *flags, "-o", exe, src, "-lm"], check=True)
    r = subprocess.run([exe, "20000000" if hw_fma else "300000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "mismatches 0" in r.stdout


def test_fp32_slab_partition_is_the_fp64_one_with_its_own_halo():
    """The fp32 / Jacobi slab path (csrc/mg3d_f32_dist.hip) uses the same partition functions with H = nu + 2: owned
    ranges tile every distributed level, cuts are even on the first distributed level and double per finer level
    (coarse plane ic and fine plane 2 ic share an owner).  Pure host arithmetic, no GPU."""
    L = M.lib()
    for nu in (1, 2, 3):
        assert L.mg3d32_slab_halo(nu) == nu + 2 and L.mg3d_slab_halo(nu) == 2 * nu + 2
    c, levels, nu = 9, 8, 2
    H = L.mg3d32_slab_halo(nu)
    for P_ in (2, 3, 4, 8):
        ld = L.mg3d_slab_first_level(c, levels, P_, H)
        assert 1 <= ld < levels
        prev = None
        for lvl in range(ld, levels):
            N = (c - 1) * (1 << lvl) + 1
            cuts = []
            for r in range(P_):
                lo, hi = C.c_int(0), C.c_int(0)
                assert L.mg3d_slab_owned(c, levels, P_, H, lvl, r, C.byref(lo), C.byref(hi)) == 0
                cuts.append((lo.value, hi.value))
            assert cuts[0][0] == 0 and cuts[-1][1] == N
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(P_ - 1))
            assert all(hi - lo >= max(16, H) for lo, hi in cuts)
            assert all(lo % 2 == 0 for lo, _ in cuts)
            if prev is not None:
                assert [lo for lo, _ in cuts] == [2 * lo for lo, _ in prev]
            prev = cuts


def test_bench_self_launch_explains_missing_gpus():
    """`python bench.py --gpus N` without a launcher becomes the launcher (torch.distributed.run as a child process);
    with fewer than N visible GPUs it must say so and exit 2 -- not a Python usage error, not a hang."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert "needs 64 visible GPUs" in r.stderr and "one rank per GPU" in r.stderr
    assert r.stdout.strip() == ""


def test_option_keys_match_the_documented_tables():
    """mg3d_option_name enumerates the launch / schedule options (no GPU needed); the table in include/mg3d.h and the one in
    INTEGRATION.md name exactly these keys, and no product source reads the environment outside context creation."""
    L = M.lib()
    keys, i = [], 0
    while L.mg3d_option_name(i):
        keys.append(L.mg3d_option_name(i).decode())
        i += 1
    assert len(keys) == len(set(keys)) >= 15 and "carry" in keys and "legs" in keys
    hdr = open(os.path.join(ROOT, "include", "mg3d.h")).read()
    block = hdr[hdr.index(" *   key             default"):hdr.index("int mg3d_ctx_set_option")]
    doc = re.findall(r"^ \*   ([a-z_0-9/]+)\s+-?\d", block, flags=re.M)
    expand = lambda names: sorted(k for n in names for k in (["sweep_rj", "sweep_nw", "sweep_pf"] if n == "sweep_rj/nw/pf" else [n]))
    assert expand(doc) == sorted(keys), (expand(doc), sorted(keys))
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = integ[integ.index("| key (`mg3d_ctx_set_option`)"):integ.index("Read at creation only, no key")]
    tkeys = []
    for row in table.splitlines()[2:]:
        cell = row.split("|")[1] if row.count("|") > 3 else ""
        if "fp32" in cell:
            continue
        tkeys += re.findall(r"`([a-z_0-9]+)`", cell)
    assert sorted(tkeys) == sorted(keys), (sorted(tkeys), sorted(keys))
    # getenv only where a context / handle is created (or in host-side planning that has no handle)
    allowed = {"mg3d_ctx.hip": {"mg3d_options_init", "ctx_new"}, "mg3d_dist.hip": {"mg3d_dist_create", "mg3d_slab_first_level"},
               "mg3d_f32.hip": {"mg3d32_create_slabs"}}
    csrc = os.path.join(ROOT, "multigrid_parallel_amd", "csrc")
    for f in os.listdir(csrc):
        if not f.endswith((".hip", ".h", ".c")):
            continue
        src = open(os.path.join(csrc, f)).read()
        for m in re.finditer(r"getenv\(", src):
            head = src[:m.start()]
            fn = re.findall(r"^(?:extern \"C\" |static )?[a-zA-Z_][\w \*]*?\b(\w+)\([^;{]*\)\s*\n?\{", head, flags=re.M)
            assert f in allowed and fn and fn[-1] in allowed[f], f"getenv in {f} inside {fn[-1] if fn else '?'}"
