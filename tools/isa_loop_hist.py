#!/usr/bin/env python3
"""Instruction histogram of the biggest loop of one kernel of a .hip file (static count; the sweep kernel's plane loop
holds two steps).  Usage: tools/isa_loop_hist.py file.hip 'mangled-substring' """
import collections, os, re, subprocess, sys
src, pat = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = "/tmp/isa_loop_hist.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "multigrid_parallel_amd", "csrc"), "-S",
                "--cuda-device-only", "-o", asm, src], check=True, capture_output=True)
lines = open(asm).read().split("\n")
names = [m.group(1) for l in lines for m in [re.match(r"^(_Z\w+):", l)] if m and pat in m.group(1)]
for nm in names:
    start = next(i for i, l in enumerate(lines) if l.startswith(nm + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    b = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(b) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(b):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    if not loops:
        continue
    a, c = max(loops, key=lambda x: x[1] - x[0])
    cnt = collections.Counter()
    for l in b[a:c]:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if m:
            cnt[m.group(1)] += 1
    tot = sum(cnt.values())
    salu = sum(v for k, v in cnt.items() if k.startswith("s_"))
    valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
    name = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
    print(f"{name}: loop of {tot} instructions, SALU {salu}, VALU {valu}, "
          f"v_mov {sum(v for k, v in cnt.items() if k.startswith('v_mov'))}, "
          f"readlane/writelane {cnt['v_readlane_b32'] + cnt['v_writelane_b32']}, f64 {sum(v for k, v in cnt.items() if 'f64' in k)}")
    if len(sys.argv) > 3:
        print("   " + ", ".join(f"{k}:{v}" for k, v in cnt.most_common(40)))
