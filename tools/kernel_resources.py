#!/usr/bin/env python3
"""Per-kernel register/LDS/occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: tools/kernel_resources.py multigrid_parallel_amd/csrc/mg3d_sweep.hip [filter]"""
import os, re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
       "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "multigrid_parallel_amd", "csrc"), "-c", src,
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]  # extra hipcc flags after the filter
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} spill {r.get('VGPR Spill','?'):>3s} "
          f"scratch {r.get('ScratchSize','?'):>5s} occ {r.get('Occupancy','?'):>2s} LDS {r.get('LDS Size','?'):>6s}")
