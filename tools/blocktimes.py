"""Lock-step of the blocks of one sweep launch: wall clock of every block every 32 steps (measurement build:
tools/build_variant.sh bt -DMG3D_DEBUG_BLOCKTIMES [-DMG3D_DEBUG_BT_COND='(...)'] ; MG3D_LIB_PATH=.../libmg3d_bt.so python tools/blocktimes.py)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_parallel_amd as M
lib = ctypes.CDLL(os.environ["MG3D_LIB_PATH"])
with M.Solver(9, 7, 2) as s:
    s.setup_test_problem()
    s.vcycles(6)
    buf = np.zeros((8, 1024), dtype=np.uint64)
    assert lib.mg3d_debug_blocktimes(buf.ctypes.data_as(ctypes.c_void_p)) == 0
nb = int((buf[0] != 0).sum())
t = buf[:, :nb].astype(np.float64) / 100.0  # us (100 MHz)
t0 = t[0].min()
print(f"{nb} blocks; times in us relative to the first block's start")
for k in range(8):
    if not t[k].any():
        continue
    x = t[k][t[k] > 0] - t0
    print(f"step {32 * k:3d}: blocks {len(x):4d}  min {x.min():8.1f}  median {np.median(x):8.1f}  max {x.max():8.1f}  spread(5-95%) {np.percentile(x, 95) - np.percentile(x, 5):7.1f}")
# neighbours in j inside one chunk layer (tile fastest, ntk = 5): |t[b] - t[b + 5]| for the first layer (the XCD renumbering permutes blocks: report by block id only)
step = 128 // 32
x = t[step][:110]
print("first 110 blocks at step 128: std", x.std())
# per-tile step time of the first two chunk layers (xcd_remap == 2 mapping of mg3d_sweep_kernel.h)
T, NTK = 110, 5
def tile_of(vb):
    ch, tl = divmod(vb, T)
    r0 = (ch * T) & 7
    x = (r0 + tl) & 7
    o = 0
    for y in range(x):
        first = (y - r0 + 8) & 7
        o += (T - first + 7) >> 3 if first < T else 0
    return ch, o + (tl - ((x - r0 + 8) & 7)) // 8
rate = (t[7] - t[1]) / 192.0
for ch in range(2):
    grid = np.zeros((22, 5)); xcd = np.zeros((22, 5), dtype=int)
    for vb in range(ch * T, (ch + 1) * T):
        c, tl = tile_of(vb)
        grid[tl % 22, tl // 22] = rate[vb]; xcd[tl % 22, tl // 22] = vb & 7  # tiles are numbered j-fastest (MG3D_TILE_J_FASTEST)
    print(f"chunk layer {ch}: us per step by tile (rows tj, columns tk) | XCD group")
    for j in range(22):
        print("  " + " ".join(f"{v:5.2f}" for v in grid[j]) + "   | " + " ".join(str(v) for v in xcd[j]))
