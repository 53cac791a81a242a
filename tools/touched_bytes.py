#!/usr/bin/env python3
"""Bytes the tiles of a fused sweep launch TOUCH (halo rows / columns and warm-up planes included), from the launcher's
own tiling rules (csrc/mg3d_sweep.hip: SweepShape, launch_sweep) -- to put beside the compulsory bytes (every point once)
and the counter bytes (what leaves L2 for the fabric).  Host arithmetic only.
  python tools/touched_bytes.py [N] [name:S,RES,CI ...]      default: the four finest-level launches of `9 7 2`"""
import sys

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 513
RJ, NW, W = 4, 8, 8
shapes = [a for a in sys.argv[1:] if ":" in a] or ["4 passes:4,0,231", "residual+restriction:0,2,103", "residual+restriction:0,2,65",
                                                     "prolongation+2 passes:2,0,103", "2 passes+norm:2,1,103"]
n = N ** 3
nc = ((N + 1) // 2) ** 3
print(f"N = {N}: compulsory field = {n * W / 1e9:.3f} GB")
for sh in shapes:
    name, rest = sh.split(":")
    S, RES, CI = (int(x) for x in rest.split(","))
    ST = S + ((1 if S > 0 else 2) if RES else 0)
    HJ = S + (1 if RES else 0) + (1 if RES == 2 else 0)
    HK = (HJ + 1) & ~1
    HI = HJ
    TJ = NW * RJ
    VJ = TJ - 2 * HJ
    ntj = (N + VJ - 1) // VJ
    tiles_for = lambda vk: 1 if N <= 128 else (N - 128 + vk - 1) // vk + 1
    vk_t, vk_l = 128 - 2 * HK, 112
    vk = vk_l if HK <= 8 and tiles_for(vk_l) <= tiles_for(vk_t) else vk_t
    ntk = tiles_for(vk)
    area = 0  # points of one plane touched by all tiles (rows x columns, clipped to the grid)
    for tj in range(ntj):
        j0, j1 = max(0, tj * VJ - HJ), min(N, tj * VJ - HJ + TJ)
        for tk in range(ntk):
            k0, k1 = tk * vk, min(N, tk * vk + 128)
            k1 += k1 & 1 if k1 < N else 0  # 16-byte pairs
            area += max(0, j1 - j0) * max(0, k1 - k0)
    planes = 0
    off = 0
    while off < N:
        ln = min(CI, N - off)
        lo = off - HI - 1  # warm-up planes (one more where the colour phase needs it)
        hi = off + ln - 1 + ST + (2 if RES == 2 else 0) + 1
        planes += min(N, hi) - max(0, lo)
        off += ln
    reads = 2 * area * planes * W
    writes = (n if S > 0 else 0) * W + (nc * W if RES == 2 else 0)
    comp = (3 if S > 0 else 2) * n * W + (nc * W if RES == 2 else 0) + (nc * W if "prolong" in name else 0)
    print(f"{name:26s} S={S} RES={RES} CI={CI:3d}: tiles {ntj}x{ntk}, {TJ}x128 own {VJ}x{128 - 2 * (128 - vk) // 2}; "
          f"touched reads {reads / 1e9:6.3f} GB + writes {writes / 1e9:5.3f} GB = {(reads + writes) / 1e9:6.3f} GB "
          f"= {(reads + writes) / comp:4.2f} x compulsory {comp / 1e9:5.3f} GB")
