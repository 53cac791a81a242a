#!/usr/bin/env python3
"""HBM-side bytes per finest-level launch from rocprofv3 counters (run on the GPU box):
two --pmc passes of `bench.py --steps 5 --warmup 1 --no-cpu-baseline` (FETCH_SIZE; WRITE_SIZE -- the guide: not in one
pass), gfx950 correction read bytes = 2 x FETCH_SIZE for 16-byte-per-lane loads (MI355X_MICROARCH.md, HBM), per kernel
name the finest level's steady-state dispatch group (most launches among the large-traffic groups), mean per launch.  Writes
gpurun_out/pmc_traffic.json {source_sha256, bytes_per_launch: {timer name: bytes}} and a readable table
gpurun_out/<tag>_pmc_traffic_finest_level.txt (gpurun only brings gpurun_out/ back): copy both to profiles/.  bench.py
only uses profiles/pmc_traffic.json while the kernel sources still hash to source_sha256.
Usage: python3 tools/pmc_traffic.py [tag]"""
import csv, glob, json, os, subprocess, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
cmd = ["python3", os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline"]
for name, ctr in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"])):
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *ctr, "--output-format", "csv", "-d", os.path.join(out, name),
                    "-o", "p", "--", *cmd], cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

# (TCC_EA0_RDREQ_DRAM was tried as a way to tell Infinity-Cache hits from HBM reads: it reports 100 % of the read requests
# for every kernel -- it names the address space, not the level that served the request)

def load(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            g = int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0)
            acc[row["Kernel_Name"]][(g, row["Counter_Name"])].append(float(row["Counter_Value"]))
    return acc


A, B = load("pmc_fetch"), load("pmc_write")
# (most specific first: the one-launch-per-leg shapes share their first template arguments with the four-row ones; bench.py's
# default command runs the configured schedule, then option carry = 0, then option legs = 1, so all of them appear)
timer_of = {"sweep_kernel<4, 0, 4, 8, 1, true, true, 0, 4": "leg_up", "sweep_kernel<3, 2, 4, 8,": "leg_down",
            "sweep_kernel<4, 3,": "sweep4+norm", "sweep_kernel<1, 2,": "sweep1+restrict", "sweep_kernel<4, 0, 4, 8, 1, false": "sweep4",
            "sweep_kernel<0, 2,": "residual", "sweep_kernel<2, 1,": "sweep2+residual", "sweep_kernel<2, 0,": "sweep2",
            "prolong_cell_kernel": "prolong"}
res, lines = {}, []
for k in sorted(A, key=lambda k: -max(sum(v) / len(v) for (g, c), v in A[k].items() if c == "FETCH_SIZE")):
    # the finest level's launches (at least half the kernel's largest value -- grids of different levels can coincide)
    # and among their grids the one launched most often: the steady state; the other grids are the chunk lengths the
    # sweep launcher tries the first time a shape meets a level
    def steady(table, ctr, grid=None):
        vals = {g: v for (g, c), v in table.items() if c == ctr and (grid is None or g == grid)}
        if not vals:
            return grid, 0.0, 0
        top = max(max(v) for v in vals.values())
        kept = {g: [x for x in v if x >= 0.5 * top] for g, v in vals.items()}
        g = max((g for g, v in kept.items() if v), key=lambda g: (len(kept[g]), sum(kept[g])))
        return g, sum(kept[g]) / len(kept[g]), len(kept[g])
    g, fetch_kb, n = steady(A[k], "FETCH_SIZE")
    wsel = B.get(k, {})
    mean = lambda key: steady(wsel, key[1], key[0])[1]
    rd, wr = 2 * fetch_kb * 1024, mean((g, "WRITE_SIZE")) * 1024
    lines.append(f"{k[:66]:66s} | grid {g:9d} | n={n:3d} | read {rd / 1e9:6.3f} GB (2 x FETCH_SIZE) | written {wr / 1e9:6.3f} GB | "
                 f"L2 hit {mean((g, 'TCC_HIT_sum')):.4g} miss {mean((g, 'TCC_MISS_sum')):.4g} | HBM-side bytes/launch {(rd + wr) / 1e9:6.3f} GB")
    for pat, t in timer_of.items():
        if pat in k and t not in res:
            res[t] = rd + wr
json.dump({"source_sha256": bench.kernel_source_hash(), "command": " ".join(cmd[1:]), "bytes_per_launch": res},
          open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
hdr = (f"rocprofv3 --pmc passes (separate runs: FETCH_SIZE ; WRITE_SIZE TCC_HIT_sum TCC_MISS_sum) of\\n  {' '.join(cmd)}\\n"
       "Per kernel: the finest level's steady-state dispatch group (most launches), mean per launch.  gfx950 correction:\\n"
       "read bytes = 2 x FETCH_SIZE (16-byte-per-lane loads; MI355X_MICROARCH.md, HBM).\\n\\n").replace("\\n", "\n")
open(os.path.join(out, f"{tag}_pmc_traffic_finest_level.txt"), "w").write(hdr + "\n".join(lines) + "\n")
print("\n".join(lines))
