#!/usr/bin/env python3
"""bench.py -- V-cycles/s of the MI355X multigrid V-cycle on the reference's own problem.

  python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one V(nu,nu) cycle of the 3D Poisson problem test_mg_3d.c sets up (unit cube, RHS 0,
Dirichlet u = x^2-2y^2+z^2 on the six faces, zero initial guess), device resident, through the C ABI
of libmg3d.so (mg3d_vcycles).  Default workload: arguments `9 7 2` = 513^3 ("512^3"), V(2,2), fp64 --
BASELINE.json configs[2].  Prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     : dominant kernel, compulsory bytes per launch (inputs read once + outputs written once) / its
                 average duration measured with HIP events on the library's stream inside the timed region, against
                 8 TB/s HBM3E; the same for every finest-level launch; SURVEY 8(d)'s credit reported separately
  cpu_baseline : the reference CPU path (oracle/_ref, kind "reference") or our CPU restatement
                 (oracle/, kind "port") timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_cycle(c, L, nu, w=8):
    """SURVEY.md 8(d), summed from its per-operator figures: RB sweep 3n, residual+store 3n, residual norm
    2n, restriction n+n_c, coarse zeroing n_c, prolong+correct n_c+2n  =>  per level
    [3(nu1+nu2)+8]*n*w + 3*n_c*w, plus (n0^2+2 n0)*w for the LU solve.  (513^3, V(2,2): 25.18 GB, the figure
    BASELINE.md quotes; SURVEY's bracketed "+7" is an off-by-one against its own component list.)"""
    tot = 0
    for l in range(1, L):
        n = ((c - 1) * (1 << l) + 1) ** 3
        nc = ((c - 1) * (1 << (l - 1)) + 1) ** 3
        tot += (3 * (nu + nu) + 8) * n * w + 3 * nc * w
    n0 = c ** 3
    return tot + (n0 * n0 + 2 * n0) * w


def compulsory_bytes_per_cycle(c, L, w=8):
    """What one V-cycle must move if every leg of every level streams its fields exactly once: down-leg reads u, d,
    writes u and the coarse right-hand side; up-leg reads u, d and the coarse correction, writes u; plus the LU factors
    and two vectors on the coarsest level.  The zero coarse guess is never read (it is not stored)."""
    tot = 0
    for l in range(1, L):
        n = ((c - 1) * (1 << l) + 1) ** 3
        nc = ((c - 1) * (1 << (l - 1)) + 1) ** 3
        down = (3 if l == L - 1 else 2) * n * w + nc * w
        up = 3 * n * w + nc * w
        tot += down + up
    n0 = c ** 3
    return tot + (n0 * n0 + 2 * n0) * w


def kernel_source_hash():
    """sha256 over the kernel sources a counter measurement belongs to (profiles/pmc_traffic.json records it)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "multigrid_parallel_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.c")) +
                    [os.path.join(d, "Makefile")]):  # (the Makefile: compiler flags are part of what was measured)
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def host_cores():
    """Host cores this job may use: affinity, capped by the cgroup CPU quota and by the GPU box's per-GPU
    CPU share (16; MG3D_CPU_THREADS overrides)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("MG3D_CPU_THREADS", "16"))))


def cpu_child(args):
    """Runs in a fresh process so OMP_NUM_THREADS is honoured and no GPU runtime is loaded."""
    import ctypes as C
    import numpy as np
    cores = int(os.environ["OMP_NUM_THREADS"])
    ref = os.path.join(ROOT, "oracle", "_ref", "libmg3d_ref.so")
    port = os.path.join(ROOT, "oracle", "liboracle.so")
    dp = C.POINTER(C.c_double)
    cycles = args.cpu_cycles
    norms = np.zeros(cycles + 1)
    if os.path.exists(ref) and not args.cpu_port:
        lib = C.CDLL(ref)
        lib.ref_run_problem.restype = C.c_double
        lib.ref_run_problem.argtypes = [C.c_int] * 4 + [dp, dp, dp]
        # 1 untimed warm-up cycle is part of the same loop; time = (cycles+1) cycles, report per cycle
        secs = lib.ref_run_problem(args.coarse, args.levels, args.nu, cycles + 1, norms.ctypes.data_as(dp), None, None)
        kind = "reference"
    else:
        if not os.path.exists(port):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True,
                           stdout=subprocess.DEVNULL)
        lib = C.CDLL(port)
        lib.orc_run_problem.restype = C.c_double
        lib.orc_run_problem.argtypes = [C.c_int] * 5 + [dp, dp, dp]
        secs = lib.orc_run_problem(args.coarse, args.levels, args.nu, cycles + 1, 0, norms.ctypes.data_as(dp), None,
                                   None)
        kind = "port"
    N = (args.coarse - 1) * (1 << (args.levels - 1)) + 1
    per = secs / (cycles + 1)
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    print(json.dumps({"value": 1.0 / per, "unit": "V-cycles/s", "cores": cores, "kind": kind, "cpu": model,
                      "sample": f"{cycles + 1} consecutive V({args.nu},{args.nu}) cycles of the same {N}^3 problem "
                                f"(args {args.coarse} {args.levels} {args.nu}), OpenMP on {cores} host cores (this GPU's CPU share), "
                                f"omp_get_wtime around the cycle loop as test_mg_3d.c:36,68",
                      "host_cores_total": os.cpu_count(),
                      "seconds_per_cycle": per, "last_norm": float(norms[cycles])}))


def run_cpu_baseline(args):
    env = dict(os.environ)
    cores = host_cores()
    env.update(OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="close", OMP_PLACES="cores")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-child", "--coarse", str(args.coarse), "--levels",
           str(args.levels), "--smooth-iters", str(args.nu), "--cpu-cycles", str(args.cpu_cycles)]
    if args.cpu_port:
        cmd.append("--cpu-port")
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        res = json.loads(out.stdout.strip().splitlines()[-1])
        if cores > 8:  # also at 8 threads, the thread count of the reference's own scaling note
            env8 = dict(env, OMP_NUM_THREADS="8")
            cmd8 = [x for x in cmd]
            cmd8[cmd8.index("--cpu-cycles") + 1] = str(max(2, args.cpu_cycles // 2))
            out8 = subprocess.run(cmd8, env=env8, capture_output=True, text=True, timeout=900)
            res["value_8_threads"] = json.loads(out8.stdout.strip().splitlines()[-1])["value"]
        return res
    except Exception as e:  # the baseline is reported, never required
        return {"value": None, "unit": "V-cycles/s", "cores": cores, "kind": "port", "sample": f"failed: {e}"}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run` (one rank per GPU,
    rendezvous on 127.0.0.1) as a CHILD process before anything here touches the GPU, relay its output (rank 0
    prints the JSON line) and exit with its return code."""
    import socket
    import torch  # device_count() does not initialise the GPU on this image
    have = torch.cuda.device_count()
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs (one rank per GPU; RCCL refuses two ranks "
              f"on one device), this box has {have}", file=sys.stderr)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def f32_roofline(launches_tab, comp, alg, per, world, steps):
    """dominant finest-level launch (by total time) with its compulsory bytes / mean duration / 8 TB/s, the other launches
    beside it; without kernel timers (slabs) the whole cycle against its compulsory bytes"""
    whole = {"vcycle_compulsory_gb": comp / 1e9, "vcycle_frac_of_hbm_peak": comp / per / 1e9 / (HBM_PEAK_GBS * world),
             "vcycle_frac_survey_credit": alg / per / 1e9 / (HBM_PEAK_GBS * world)}
    if not launches_tab:
        return dict({"bound": "hbm", "kernel": "whole V-cycle (compulsory bytes: every leg streams its fields once, w = 4)",
                     "achieved": comp / per / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                     "frac": comp / per / 1e9 / (HBM_PEAK_GBS * world), "traffic": None}, **whole)
    dom = max(launches_tab, key=lambda r: r["ms"] * r["launches"])
    ach = dom["compulsory_bytes"] / (dom["ms"] * 1e-3) / 1e9
    return dict({"bound": "hbm", "kernel": f"fp32 finest level: {dom['what']} [timer {dom['kernel']}]", "achieved": ach,
                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                 "bytes_per_launch": dom["compulsory_bytes"], "bytes_definition": "compulsory: inputs read once + outputs written once",
                 "avg_launch_ms": dom["ms"], "launches_timed": dom["launches"], "finest_level_launches": launches_tab,
                 "finest_level_ms_per_cycle": sum(r["ms"] * r["launches"] for r in launches_tab) / max(1, steps)},
                **whole)


def f32_line(args):
    """One JSON line for the fp32 / damped-Jacobi / F-cycle variant (BASELINE configs[4]; parity unpinned).  Not the
    headline metric.  --gpus N > 1 (under torch.distributed.run, or self-launched): the same problem on N i-slabs."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("MG3D_BENCH_FORCE_DIST") == "1"
    if world > 1 or (force_dist and "RANK" in os.environ):
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import multigrid_parallel_amd as M
    c, nu = args.coarse, args.nu
    L = args.levels if args.levels != 7 else 8
    N = (c - 1) * (1 << (L - 1)) + 1
    alg = sum((3 * 2 * nu + 8) * ((c - 1) * (1 << l) + 1) ** 3 * 4 + 3 * ((c - 1) * (1 << (l - 1)) + 1) ** 3 * 4
              for l in range(1, L)) + (c ** 6 + 2 * c ** 3) * 8
    comp = compulsory_bytes_per_cycle(c, L, w=4) - (c ** 6 + 2 * c ** 3) * 4 + (c ** 6 + 2 * c ** 3) * 8  # LU in double
    if world > 1 or force_dist:
        uid = [M.DistSolver.unique_id() if rank == 0 else None]
        if dist.is_initialized():
            dist.broadcast_object_list(uid, src=0)
        s = M.DistSolver32(c, L, nu, rank=rank, nranks=world, unique_id=uid[0], device=local_rank)
        par = f"{world} GPUs, i-slabs, halo {s.halo} planes, levels >= {s.first_level} distributed, RCCL send/recv"
    else:
        s = M.Solver32(c, L, nu)
        par = "1 GPU"

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        s.sync()
        torch.cuda.synchronize()

    s.setup_test_problem(fmg=False)
    s.vcycles(args.warmup)
    barrier()
    t0 = time.perf_counter()
    norms = s.vcycles(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    # per-launch roofline of the finest level (single domain): the same cycles once more with the kernel timers on (event
    # pairs on the library's stream; kept out of `value`'s timed region: a marker packet is ~5 us of idle queue)
    launches_tab = []
    if world == 1 and not force_dist:
        s.timing_enable(True)
        s.vcycles(args.steps)
        s.timing_enable(False)
        n_f, n_c, w = N ** 3, (((N - 1) // 2) + 1) ** 3, 4
        kinds = {  # kernel timer -> (what one launch does, compulsory bytes: inputs read once + outputs written once)
            "pair": ("two damped-Jacobi sweeps", 3 * n_f * w),
            "pair+tap": ("two sweeps + the previous cycle's residual norm tapped from the first sweep's sums", 3 * n_f * w),
            "prolong+pair": ("prolongation + two sweeps", 3 * n_f * w + n_c * w),
            "prolong+pair+norm": ("prolongation + two sweeps + residual norm (third stage)", 3 * n_f * w + n_c * w),
            "pair+norm": ("two sweeps + residual norm", 3 * n_f * w),
            "residual+restrict": ("residual + full-weighting restriction, r never stored", 2 * n_f * w + n_c * w),
            "residual": ("residual norm", 2 * n_f * w),
            "prolong": ("prolongation", 2 * n_f * w + n_c * w),
            "sweep1": ("one sweep", 3 * n_f * w),
        }
        for kn, (calls, secs) in sorted(s.kernel_times().items()):
            what, comp_l = kinds[kn]
            ms = secs * 1e3 / calls
            launches_tab.append({"kernel": kn, "what": what, "launches": calls, "ms": ms, "compulsory_bytes": comp_l,
                                 "frac": comp_l / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per = elapsed / args.steps
    barrier()
    t0 = time.perf_counter()
    s.setup_test_problem(fmg=True)
    barrier()
    t_fmg = time.perf_counter() - t0
    after = s.vcycles(1)
    slabs_ok = None
    if (world > 1 or force_dist) and not args.no_verify:
        # untimed self-check: the owned planes of every rank against a single-domain run of the same sequence on rank 0
        import ctypes as C
        import hashlib
        import numpy as np
        from multigrid_parallel_amd.binding import MG3D_U, lib as _lib
        full = s.download(MG3D_U, L - 1)
        lo, hi = C.c_int(0), C.c_int(0)
        _lib().mg3d_slab_owned(c, L, world, s.halo, L - 1, rank, C.byref(lo), C.byref(hi))
        mine = (lo.value, hi.value, hashlib.sha256(full.reshape(N, N * N)[lo.value:hi.value].tobytes()).hexdigest())
        del full
        parts = [mine]
        if dist.is_initialized():
            parts = [None] * world
            dist.all_gather_object(parts, mine)
        if rank == 0:
            with M.Solver32(c, L, nu) as one:
                one.setup_test_problem(fmg=True)
                ref_after = one.vcycles(1)
                ref = one.download(MG3D_U, L - 1).reshape(N, N * N)
            slabs_ok = all(hashlib.sha256(ref[a:b].tobytes()).hexdigest() == h for a, b, h in parts) and \
                bool(np.allclose(after, ref_after, rtol=1e-9, atol=0))
            del ref
    if rank == 0:
        print(json.dumps({
            "metric": f"V-cycles/sec, {N}^3 Poisson, fp32, damped-Jacobi V({nu},{nu}) after an F-cycle start (parity unpinned)",
            "value": 1.0 / per, "unit": "V-cycles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{N}^3 Poisson (args {c} {L} {nu}), Dirichlet x^2-2y^2+z^2, omega 6/7, device-resident",
                       "coarse_pts": c, "levels": L, "smooth_iters": nu, "parallelism": par},
            "fcycle_start_ms": t_fmg * 1e3, "first_norm": float(norms[0]), "last_norm": float(norms[-1]),
            "norm_after_fcycle_start": float(after[0]), "slabs_bit_identical_to_single_domain": slabs_ok,
            "parity": "unpinned: the reference has no fp32 / Jacobi / F-cycle path; bit-identical to the builder's restatement only",
            "roofline": f32_roofline(launches_tab, comp, alg, per, world, args.steps),
            "cpu_baseline": None}))
    s.close()
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--coarse", type=int, default=9)
    ap.add_argument("--levels", type=int, default=7)
    ap.add_argument("--smooth-iters", dest="nu", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="N > 1: skip the untimed comparison of the assembled slabs with a single-domain run")
    ap.add_argument("--cpu-cycles", type=int, default=8)
    ap.add_argument("--cpu-port", action="store_true", help="time oracle/ (port) even if oracle/_ref exists")
    ap.add_argument("--cpu-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--breakdown", action="store_true", help="print the per-level stage table to stderr")
    ap.add_argument("--timing-mode", type=int, default=6, help=argparse.SUPPRESS)  # A/B of the timer cost (0 = none)
    ap.add_argument("--no-alt-schedules", action="store_true",
                    help="skip the extra timed regions (the other two schedules) reported beside `value`")
    ap.add_argument("--f32", action="store_true",
                    help="BASELINE configs[4] on one GPU instead of the headline: fp32, damped Jacobi, F-cycle start "
                         "(parity unpinned); default size 9 8 2 = 1025^3")
    args = ap.parse_args()
    if args.cpu_child:
        return cpu_child(args)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return self_launch(args)  # plain `python bench.py --gpus N`: become the launcher, one child rank per GPU
    if args.f32:
        return f32_line(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    import torch  # first: libmg3d then shares torch's HIP runtime (same SONAME)
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible (the product has no CPU path)")
    if "MG3D_BENCH_DEVICE" in os.environ:  # rehearsal of several ranks on one GPU (RCCL permitting)
        local_rank = int(os.environ["MG3D_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("MG3D_BENCH_FORCE_DIST") == "1"  # rehearsal: the N > 1 code path with one rank
    if world > 1 or (force_dist and "RANK" in os.environ):
        backend = os.environ.get("MG3D_BENCH_TORCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import multigrid_parallel_amd as M
    from multigrid_parallel_amd.binding import MG3D_U

    c, L, nu = args.coarse, args.levels, args.nu
    N = (c - 1) * (1 << (L - 1)) + 1

    def barrier(obj):
        if dist.is_initialized():
            dist.barrier()
        obj.sync()
        torch.cuda.synchronize()

    if world > 1 or force_dist:
        # i-slab decomposition: one rank per GPU, the library's own RCCL communicator for the plane
        # exchanges (unique id distributed through torch.distributed), strong scaling of the same problem
        uid = [M.DistSolver.unique_id() if rank == 0 else None]
        if dist.is_initialized():
            dist.broadcast_object_list(uid, src=0)
        solver = M.DistSolver(c, L, nu, rank=rank, nranks=world, unique_id=uid[0], device=local_rank)
        solver.setup_test_problem()
        solver.vcycles(1)  # priming: the sweep launcher measures its chunk lengths on first use; the problem is set up again
        init = solver.setup_test_problem()
        warm_norms = solver.vcycles(args.warmup)
        solver.timing_enable(True)  # per-phase event pairs on the library's streams (no host stall inside the batch)
        barrier(solver)
        t0 = time.perf_counter()
        norms = solver.vcycles(args.steps)
        barrier(solver)
        elapsed = time.perf_counter() - t0
        phase_ms = dict(solver.timing(), rank=rank)
        solver.timing_enable(False)
        per_rank = [phase_ms]
        if dist.is_initialized():
            per_rank = [None] * world
            dist.all_gather_object(per_rank, phase_ms)
        if dist.is_initialized():
            t = torch.tensor([elapsed], dtype=torch.float64,
                             device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # Self-check of the slab decomposition, outside the timed region (--no-verify skips it): every rank hashes the
        # planes it owns; rank 0 runs the same number of cycles on a single domain and hashes the same plane ranges.
        # On a real multi-GPU run this is the first evidence that the RCCL exchanges move the right planes.
        slabs_ok = None
        if not args.no_verify:
            import ctypes as C
            import hashlib
            import numpy as np
            from multigrid_parallel_amd.binding import lib as _lib
            full = solver.download(MG3D_U, L - 1)
            lo, hi = C.c_int(0), C.c_int(0)
            _lib().mg3d_slab_owned(c, L, world, solver.halo, L - 1, rank, C.byref(lo), C.byref(hi))
            mine = (rank, lo.value, hi.value, hashlib.sha256(full.reshape(N, N * N)[lo.value:hi.value].tobytes()).hexdigest())
            del full
            parts = [mine]
            if dist.is_initialized():
                parts = [None] * world
                dist.all_gather_object(parts, mine)
            if rank == 0:
                with M.Solver(c, L, nu) as one:
                    one.setup_test_problem()
                    ref_norms = one.vcycles(args.warmup + args.steps)
                    ref = one.download(MG3D_U, L - 1).reshape(N, N * N)
                slabs_ok = all(hashlib.sha256(ref[a:b].tobytes()).hexdigest() == h for _, a, b, h in parts) and \
                    bool(np.allclose(list(warm_norms) + list(norms), ref_norms, rtol=1e-9, atol=0))
                del ref
        rccl_ranks, overlap, dev = solver.comm_info()
        placement = [(rank, dev, torch.cuda.get_device_properties(dev).name)]
        if dist.is_initialized():
            gathered = [None] * world
            dist.all_gather_object(gathered, placement[0])
            placement = gathered
        # the reference's own printed history of this problem (SURVEY 6.3, 513^3 `9 7 2`, six printed digits): cycles 4 and 5
        known = {4: 80775.4, 5: 11126.2} if (c, L, nu) == (9, 7, 2) else {}
        hist = list(warm_norms) + list(norms)
        known_ok = all(abs(hist[k - 1] - v) <= 1e-5 * v for k, v in known.items() if k - 1 < len(hist)) if known else None
        if rank == 0:
            alg = algorithmic_bytes_per_cycle(c, L, nu)
            comp_cycle = compulsory_bytes_per_cycle(c, L)
            per_step = elapsed / args.steps
            print(json.dumps({
                "metric": "V-cycles/sec, 513^3 ('512^3') Poisson, V(2,2), fp64" if (c, L, nu) == (9, 7, 2)
                else f"V-cycles/sec, {N}^3 Poisson, V({nu},{nu}), fp64",
                "value": args.steps / elapsed, "unit": "V-cycles/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": per_step * 1e3, "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"{N}^3 Poisson (args {c} {L} {nu}), Dirichlet x^2-2y^2+z^2, V({nu},{nu}), "
                                       f"device-resident, test_mg_3d.c problem", "coarse_pts": c, "levels": L,
                           "smooth_iters": nu, "parallelism": f"{world} GPUs, i-slabs, halo {solver.halo} planes, "
                                                              f"levels >= {solver.first_level} distributed, RCCL send/recv"},
                "vcycle_compulsory_gb": comp_cycle / 1e9, "vcycle_compulsory_gbs": comp_cycle / per_step / 1e9,
                "vcycle_frac_of_hbm_peak": comp_cycle / per_step / 1e9 / (HBM_PEAK_GBS * world),
                "vcycle_survey_credit_gb": alg / 1e9, "vcycle_survey_credit_gbs": alg / per_step / 1e9,
                "vcycle_frac_survey_credit": alg / per_step / 1e9 / (HBM_PEAK_GBS * world),
                "first_norm": float(norms[0]), "last_norm": float(norms[-1]), "initial_rhs_norm": init,
                "rccl_ranks": rccl_ranks, "halo_overlap": overlap,
                "ranks": [{"rank": r, "device": d, "name": n} for r, d, n in placement],
                # where each rank's cycle went (ms per cycle, event pairs inside the timed region): kernels on its slabs,
                # exchanges the compute stream waits for at once, exchanges overlapped (u halos), coarse levels
                "per_rank_ms": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in pr.items()} for pr in per_rank],
                "coarse_policy": "rank 0 only (MG3D_COARSE_GATHER=1)" if os.environ.get("MG3D_COARSE_GATHER") == "1"
                else "replicated on every rank",
                "history_matches_reference": known_ok,
                "slabs_bit_identical_to_single_domain": slabs_ok,
                # no per-kernel timers on the slab path: the whole cycle against the aggregate HBM peak (the
                # per-kernel roofline and the CPU baseline belong to the N = 1 line)
                "roofline": {"bound": "hbm", "kernel": "whole V-cycle, all ranks (compulsory bytes: every leg streams "
                                                       "its fields once)",
                             "achieved": comp_cycle / per_step / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                             "frac": comp_cycle / per_step / 1e9 / (HBM_PEAK_GBS * world), "traffic": None,
                             "frac_survey_credit": alg / per_step / 1e9 / (HBM_PEAK_GBS * world)},
                "cpu_baseline": None}))
        solver.close()
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    solver = M.Solver(c, L, nu)
    # priming: the sweep launcher times a few chunk lengths the first time a kernel shape meets a level
    # (csrc/mg3d_sweep.hip); one cycle, then the problem is set up afresh -- kept out of the warm-up so that --warmup 0
    # is valid and the residual history starts at the initial guess
    solver.setup_test_problem()
    solver.vcycles(3)  # three: the launches consecutive cycles share (carried cycles) meet the level here, not in the timed region
    solver.setup_test_problem()
    init = solver.get_initial_residual()

    warm_norms = solver.vcycles(args.warmup)
    solver.timing_reset()
    # event pairs around the finest level's launches only, on every 4th cycle of the timed region (mode 6 = 4 + 2: 8 marker
    # packets per sampled cycle, ~5 us of idle queue each, no host stall; mode 3 = every cycle).  --breakdown: every stage
    # and kernel of every level
    solver.timing_enable(1 if args.breakdown else args.timing_mode)
    barrier(solver)
    t0 = time.perf_counter()
    norms = solver.vcycles(args.steps)
    barrier(solver)
    elapsed = time.perf_counter() - t0
    solver.timing_enable(0)

    tm = solver.timing()
    kt = solver.kernel_times()
    # the same K cycles with every cycle run on its own (MG3D_NO_CARRY=1: no launch shared between consecutive cycles,
    # csrc/mg3d_ctx.hip "carried cycles"), no timers: reported beside `value`; both schedules give the same bits
    plain = legs = carried = None
    if world == 1 and not args.no_alt_schedules:
        def timed_with(opts, what):
            old = {k: solver.get_option(k) for k in opts}
            for k, v in opts.items():
                solver.set_option(k, v)  # the options API (mg3d_ctx_set_option): no environment on any launch path
            try:
                solver.vcycles(3)
                barrier(solver)
                t1 = time.perf_counter()
                solver.vcycles(args.steps)
                barrier(solver)
                el1 = time.perf_counter() - t1
                return {"value": args.steps / el1, "ms_per_step": el1 / args.steps * 1e3, "what": what}
            finally:
                for k, v in old.items():
                    solver.set_option(k, v)
        N_min_legs, N_min_carry = solver.get_option("legs_min"), solver.get_option("carry_min")
        is_legs = bool(solver.get_option("legs")) and N >= N_min_legs and nu == 2
        is_carried = not is_legs and bool(solver.get_option("carry")) and N >= N_min_carry and nu == 2
        # the other schedules beside the configured one (same bits, see tests/test_gpu_parity.py, tests/test_gpu_legs.py)
        if is_legs or is_carried:
            plain = timed_with({"legs": 0, "carry": 0}, "options legs = 0, carry = 0: four launches per cycle on the finest level, none "
                                                        "shared between cycles")
        if not is_carried:
            carried = timed_with({"legs": 0, "carry": 1, "carry_min": min(N_min_carry, N)},
                                 "options legs = 0, carry = 1: consecutive cycles share a launch (three memory-bound launches per "
                                 "cycle on the finest level, 10.0 GB compulsory; the default below the option legs_min)")
        if not is_legs:
            legs = timed_with({"legs": 1, "legs_min": min(N_min_legs, N)},
                              "option legs = 1: ONE launch per leg on the finest level -- prolongation + four passes, three passes + "
                              "residual + restriction, the norm's halves taken from either side; 6.75 GB compulsory per cycle "
                              "instead of 10.0, both launches at two waves per SIMD (the default from the option legs_min, 160 points per side)")
    fin = L - 1
    n_f = N ** 3

    # ---- roofline: every finest-level launch of a cycle against what it MUST move.
    # "compulsory" bytes of a launch = its inputs read once + its outputs written once (n = points of the finest
    # level, n_c of the next coarser, w = 8): a smoothing launch 3nw whatever number of colour passes it fuses,
    # residual + restriction 2nw + n_c w, prolongation + smoothing 3nw + n_c w.  `frac` = compulsory bytes / time /
    # 8 TB/s can never exceed 1.  SURVEY 8(d) credits a fused launch with the bytes of the separate passes it
    # replaces (1.5 nw per colour pass); that figure is reported as *_survey_credit, never as HBM bandwidth.
    # `traffic` = HBM-side bytes from rocprofv3 counters (tools/pmc_traffic.py), only when measured on these sources.
    n_c = (((N - 1) // 2) + 1) ** 3
    w = 8
    kinds = {  # kernel timer name -> (description, compulsory bytes, SURVEY-credited bytes)
        "sweep4": ("4 colour passes (pre-smoother)", 3 * n_f * w, 6 * n_f * w),
        "residual": ("residual + full-weighting restriction, r never stored", 2 * n_f * w + n_c * w, 3 * n_f * w + (n_f + n_c) * w),
        "sweep2": ("prolongation + 2 colour passes (post-smoother, first half)", 3 * n_f * w + n_c * w, (n_c + 2 * n_f) * w + 3 * n_f * w),
        "sweep2+residual": ("2 colour passes + residual norm (post-smoother, second half)", 3 * n_f * w, 3 * n_f * w + 2 * n_f * w),
        # carried cycles (csrc/mg3d_ctx.hip): consecutive cycles of one mg3d_vcycles call share a launch
        "sweep4+norm": ("2 post-smoothing passes + residual norm of the cycle + the next cycle's first pre-smoothing passes "
                        "(its first red pass is the identity)", 3 * n_f * w, 6 * n_f * w + 2 * n_f * w),
        "sweep1+restrict": ("last pre-smoothing pass + residual + full-weighting restriction, r never stored",
                            3 * n_f * w + n_c * w, 1.5 * n_f * w + 3 * n_f * w + (n_f + n_c) * w),
        # one launch per leg (option legs)
        "leg_down": ("3 (behind another cycle; else 4) pre-smoothing passes + residual + full-weighting restriction in ONE launch",
                     3 * n_f * w + n_c * w, 4.5 * n_f * w + 3 * n_f * w + (n_f + n_c) * w),
        "leg_up": ("prolongation + 4 post-smoothing passes (+ the red half of the norm) in ONE launch",
                   3 * n_f * w + n_c * w, (n_c + 2 * n_f) * w + 6 * n_f * w),
        "colour_pass": ("one colour pass", 3 * n_f * w, 1.5 * n_f * w),
        "prolong": ("prolongation", 2 * n_f * w + n_c * w, (n_c + 2 * n_f) * w),
        "restrict": ("face injection of the restriction", 0, 0),
    }
    pmc, pmc_note = {}, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            rec = json.load(open(pmc_path))
            if rec.get("source_sha256") == kernel_source_hash():
                pmc = rec.get("bytes_per_launch", {})
            else:
                pmc_note = "profiles/pmc_traffic.json was measured on other kernel sources: ignored (re-run tools/pmc_traffic.py)"
        except Exception as e:
            pmc_note = f"profiles/pmc_traffic.json unreadable: {e}"
    launches_tab = []
    for (lvl, kn), (calls, secs) in sorted(kt.items()):
        if lvl != fin or kn not in kinds or not calls:
            continue
        desc, comp, credit = kinds[kn]
        ms = secs * 1e3 / calls
        row = {"kernel": kn, "what": desc, "launches": calls, "ms": ms, "compulsory_bytes": comp,
               "frac": comp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None,
               "survey_credit_bytes": credit,
               "frac_survey_credit": credit / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None,
               "counter_bytes": pmc.get(kn),
               "frac_counter": (pmc[kn] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kn in pmc and ms > 0 else None}
        launches_tab.append(row)
    # dominant = most time per STEADY-STATE cycle.  The timers sample every (mode - 2)-th cycle starting with the call's first one,
    # which has no cycle in front of it and takes the ordinary down-leg: in a sample of five the one-launch down-leg then shows four
    # launches against the up-leg's five although it runs in 19 of the 20 timed cycles -- a kernel seen in all sampled cycles but
    # one counts as once per cycle
    tm_ = 1 if args.breakdown else args.timing_mode
    sampled_ = args.steps if tm_ < 4 else len(range(0, args.steps, tm_ - 2))
    per_cycle = lambda r: max(1.0, r["launches"] / sampled_) if r["launches"] >= sampled_ - 1 else r["launches"] / sampled_
    dom = max((r for r in launches_tab if r["compulsory_bytes"]), key=lambda r: r["ms"] * per_cycle(r), default=None)
    if dom is None:
        dom = {"kernel": "none", "what": "no finest-level kernel timed", "ms": 0.0, "launches": 0, "compulsory_bytes": 0,
               "survey_credit_bytes": 0, "counter_bytes": None}
    achieved = dom["compulsory_bytes"] / (dom["ms"] * 1e-3) / 1e9 if dom["ms"] > 0 else 0.0
    roof = {"bound": "hbm", "kernel": f"sweep_kernel, finest level: {dom['what']} [timer {dom['kernel']}]",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": dom["counter_bytes"],
            "frac_definition": "compulsory bytes of the launch (inputs read once + outputs written once) / mean launch "
                               "duration (HIP event pairs on the library's stream, this run's timed region) / 8 TB/s; "
                               "<= 1 by construction.  frac_survey_credit: the same time against SURVEY 8(d)'s credited "
                               "bytes (1.5 n w per fused colour pass), may exceed 1 and is not a bandwidth",
            "traffic_source": "HBM-side bytes per launch from rocprofv3 PMC passes committed as profiles/pmc_traffic.json "
                              "(tools/pmc_traffic.py; used only when its source hash equals these kernel sources), "
                              "not measured in this run",
            "bytes_per_launch": dom["compulsory_bytes"], "bytes_definition": "compulsory: inputs read once + outputs written once",
            "avg_launch_ms": dom["ms"], "launches_timed": dom["launches"],
            "achieved_survey_credit": dom["survey_credit_bytes"] / (dom["ms"] * 1e-3) / 1e9 if dom["ms"] > 0 else 0.0,
            "frac_survey_credit": dom["survey_credit_bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if dom["ms"] > 0 else 0.0,
            "frac_counter": (dom["counter_bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if dom["counter_bytes"] and dom["ms"] > 0 else None,
            "finest_level_launches": launches_tab}
    if pmc_note:
        roof["traffic_note"] = pmc_note
    if any(r["kernel"] == "leg_up" for r in launches_tab):
        roof["schedule_note"] = ("one launch per leg (default from 160 points per side): the finest level streams through the chip twice "
                                 "per cycle instead of three times (6.75 instead of 10.0 GB compulsory, 7.8 instead of 11.4 GB HBM-side). "
                                 "Both launches run at two waves per SIMD since the prolongation is applied at the end of the previous "
                                 "step (up-leg) and the wave-edge rows keep one LDS copy, which makes room to park two slots of the d "
                                 "window (down-leg); the first version of round 4 needed one wave per SIMD and was bound by instruction "
                                 "issue (0.88 + 1.03 ms against 0.72 + 0.85 now).  See carried_schedule / plain_schedule for the same "
                                 "bits with three / four launches per cycle")
    # cycles of the timed region that carried the markers: all of them (--breakdown, mode 3) or every (mode - 2)-th
    tmode = 1 if args.breakdown else args.timing_mode
    sampled = args.steps if tmode < 4 else len(range(0, args.steps, tmode - 2))
    roof["cycles_timed"] = sampled
    finest_ms = sum(r["ms"] * r["launches"] for r in launches_tab) / max(1, sampled)
    finest_counter = sum(r["counter_bytes"] * r["launches"] for r in launches_tab if r["counter_bytes"]) / max(1, sampled) \
        if launches_tab and all(r["counter_bytes"] is not None or not r["compulsory_bytes"] for r in launches_tab) else None
    roof["finest_level_ms_per_cycle"] = finest_ms
    # the metric's second half, "smoother HBM GB/s": always the pre-smoother's four-pass launch (compulsory bytes / time)
    s4 = max((r for r in launches_tab if r["kernel"] in ("sweep4", "sweep4+norm") and r["ms"] > 0),
             key=lambda r: r["launches"], default=None)
    smoother_gbs = s4["compulsory_bytes"] / (s4["ms"] * 1e-3) / 1e9 if s4 else None
    roof["finest_level_counter_bytes_per_cycle"] = finest_counter

    # on-box ceiling of the memory system: device-to-device copy of 1 GiB (read + write), best of 5
    copy_gbs = None
    try:
        src = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
        dst = torch.empty_like(src)
        best = 1e9
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dst.copy_(src)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        copy_gbs = 2 * src.numel() * 8 / (best * 1e-3) / 1e9
        del src, dst
    except Exception:
        pass
    roof["measured_copy_gbs"] = copy_gbs
    roof["frac_of_measured_copy"] = (achieved / copy_gbs) if copy_gbs else None
    if dom["counter_bytes"] and copy_gbs and dom["ms"] > 0:
        roof["frac_counter_of_measured_copy"] = dom["counter_bytes"] / (dom["ms"] * 1e-3) / 1e9 / copy_gbs
    hist = list(warm_norms) + list(norms)
    to_tol = next((i + 1 for i, x in enumerate(hist) if x <= 1e-8 * init), None)

    if rank == 0:
        alg = algorithmic_bytes_per_cycle(c, L, nu)
        comp_cycle = compulsory_bytes_per_cycle(c, L)
        per_step = elapsed / args.steps
        line = {
            "metric": "V-cycles/sec, 513^3 ('512^3') Poisson, V(2,2), fp64" if (c, L, nu) == (9, 7, 2)
            else f"V-cycles/sec, {N}^3 Poisson, V({nu},{nu}), fp64",
            "value": args.steps / elapsed, "unit": "V-cycles/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": per_step * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{N}^3 Poisson (args {c} {L} {nu}), Dirichlet x^2-2y^2+z^2, V({nu},{nu}), "
                                   f"device-resident, test_mg_3d.c problem", "coarse_pts": c, "levels": L,
                       "smooth_iters": nu, "parallelism": f"{world} GPU" + ("" if world == 1 else " i-slabs")},
            # whole cycle: compulsory bytes (every level: each leg reads u, d (+ the coarse correction) and writes u
            # (+ the coarse rhs) once) against 8 TB/s; and the SURVEY 8(d) credit (25.18 GB at 513^3), labelled as such
            "vcycle_compulsory_gb": comp_cycle / 1e9, "vcycle_compulsory_gbs": comp_cycle / per_step / 1e9,
            "vcycle_frac_of_hbm_peak": comp_cycle / per_step / 1e9 / HBM_PEAK_GBS,
            "vcycle_survey_credit_gb": alg / 1e9, "vcycle_survey_credit_gbs": alg / per_step / 1e9,
            "vcycle_frac_survey_credit": alg / per_step / 1e9 / HBM_PEAK_GBS,
            # the metric's second half: the smoother kernel alone (finest-level four-pass launch), physical bytes
            "smoother_hbm_gbs": smoother_gbs,
            "first_norm": float(norms[0]), "last_norm": float(norms[-1]), "initial_rhs_norm": init,
            "cycles_to_1e-8": to_tol,  # test_mg_3d.c stopping rule; the reference needs 16 at 513^3
            # consecutive cycles of the timed call share a launch on the finest level (identical results)
            "schedule": "one launch per leg" if any(r["kernel"] == "leg_up" for r in launches_tab) else
                        "carried cycles" if any(r["kernel"] == "sweep4+norm" for r in launches_tab) else "plain",
            "plain_schedule": plain,
            "carried_schedule": carried,
            "legs_schedule": legs,
            "roofline": roof,
        }
        if args.breakdown:
            for (lvl, st), (calls, secs) in sorted(tm.items()):
                if calls:
                    print(f"level {lvl} {st:24s} calls {calls:5d}  {secs * 1e3 / calls:9.4f} ms/call", file=sys.stderr)
            for (lvl, kn), (calls, secs) in sorted(kt.items()):
                print(f"kernel level {lvl} {kn:18s} launches {calls:5d}  {secs * 1e3 / calls:9.4f} ms/launch", file=sys.stderr)
        if not args.no_cpu_baseline and world == 1:
            solver.finalize()
            line["cpu_baseline"] = run_cpu_baseline(args)
        print(json.dumps(line))


if __name__ == "__main__":
    main()
