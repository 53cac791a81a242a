/* mg3d_sweep_kernel.h -- the fused, temporally blocked red-black Gauss-Seidel kernel (device code; included by
 * mg3d_sweep.hip, which holds the description of the pipeline and the launchers).  Not installed. */
#ifndef MG3D_SWEEP_KERNEL_H
#define MG3D_SWEEP_KERNEL_H

#include "mg3d_internal.h"

#include <type_traits>

#define WAVE 64

struct SweepArgs {
    Geom g;
    const double *vin;
    const double *d;
    double *vout;     /* may equal nullptr when S == 0 */
    double *r;        /* residual store (interior only) or nullptr */
    double *partials; /* one partial sum of diff^2 per block, or nullptr */
    double hSq, sixth, invHsq;
    int c1;         /* colour of the first pass: 1 red, 0 black */
    int ntj, ntk;   /* tiles in j, k */
    int vk, hk;     /* k-tiling: tile tk covers columns [vk*tk, vk*tk + 128) and owns those at least hk from its
                       edges (a tile edge on the global boundary needs no halo); vk = 128 - 2*hk */
    int CI;         /* planes per i-chunk (lock-step chunks: block -> (tile column, chunk), tile fastest) */
    int edge;       /* > 0: this launch produces only the first and the last `edge` planes of [i_lo, i_hi) -- two chunks; the
                       rest is another launch's (slab path: the planes a halo exchange sends are made first, the exchange
                       then runs underneath the launch that makes the interior) */
    int i_lo, i_hi;     /* local output planes this launch produces */
    int acc_lo, acc_hi; /* local planes whose diff^2 enter the norm (owned planes of a slab) */
    /* fused prolongation (PRO): the level's input is vin + P(ec) (mg_3d.h:1000-1145); gce = geometry of ec */
    const double *ec;
    Geom gce;
    /* fused restriction (RES == 2): coarse geometry, coarse right-hand side, local coarse planes to write */
    Geom gc;
    double *dc;
    int ic_lo, ic_hi;
    int xcd_remap;  /* 1: blocks of one XCD group (blockIdx % 8) take consecutive shares of the work; 2: per chunk layer */
};

/* RES: 0 = smoothing only, 1 = + residual (r store and/or norm), 2 = + residual AND full-weighting
 * restriction of it into the coarse right-hand side (mg_3d.h:961-995) -- r never travels to HBM;
 * 3 = the residual NORM of the state half-way through the passes (after pass S/2), see "tap" in the kernel.
 * The 27-point restriction stencil reaches one fine point beyond the coarse point's centre, so its halo
 * and warm-up are one deeper. */
template <int S, int RES> struct SweepShape {
    /* With S > 0 the residual of the colour updated last falls out of stage S itself (same neighbour
     * sum), the other colour needs one more gather: ST = S + 1.  A pure residual (S == 0) needs both. */
    static constexpr bool TAIL = RES == 1 || RES == 2; /* a residual BEHIND the passes (the tap costs no stage) */
    static constexpr int ST = S + (TAIL ? (S > 0 ? 1 : 2) : 0); /* pipeline stages */
    /* halo rows.  Where the restriction rides along: S + 2 (a pass each, the residual, the coarse row's reach of one fine
     * row).  A tile's first row then has the parity of HJ; the coarse rows are centred on the thread's even rows when it is
     * even and on its odd rows when it is odd (CO in the kernel) -- rounding HJ up to an even number instead (rounds 2-3)
     * cost a tile row of two owned rows: 20 instead of 22 of 32 for three passes + residual + restriction */
    static constexpr int HJ = RES == 2 ? S + 2 : S + (TAIL ? 1 : 0);
    static constexpr int HK = (HJ + 1) & ~1;     /* halo columns, even so pairs stay aligned */
    static constexpr int HI = S + (TAIL ? 1 : 0) + (RES == 2 ? 1 : 0); /* warm-up planes */
};

/* One-lane shifts across the whole wave as DPP moves (v_mov_b32_dpp wave_shr:1 / wave_shl:1, two per
 * double): VALU-rate, no LDS crossbar.  Lane 0 / lane 63 have no source and read 0 (tile halo, never used). */
template <int CTRL> __device__ __forceinline__ double dpp_move(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    /* bound_ctrl: the lane without a source (0 or 63, a tile-halo lane) reads 0; no copy of the old value */
    const int rlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    const int rhi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)rhi << 32) | (unsigned int)rlo);
}
__device__ __forceinline__ double lane_from_left(double x) /* lane l receives lane l-1 */
{
#ifdef MG3D_NO_DPP
    return __shfl_up(x, 1, WAVE);
#else
    return dpp_move<0x138>(x); /* wave_shr:1 */
#endif
}
__device__ __forceinline__ double lane_from_right(double x) /* lane l receives lane l+1 */
{
#ifdef MG3D_NO_DPP
    return __shfl_down(x, 1, WAVE);
#else
    return dpp_move<0x130>(x); /* wave_shl:1 */
#endif
}

/* Cache policy of the streams (compile-time MG3D_NT bits: 1 = u loads, 4 = d loads, 2 = u stores non-temporal, 8 = in
 * the pure residual launches (S = 0) the loads of the rows no other tile column reads; default 2).
 * The output is not read again before the next launch, a gigabyte later: written non-temporally it does not push the
 * halo rows the neighbouring tile columns are about to re-read out of L2 / the Infinity Cache (513^3: 265 -> 274
 * V-cycles/s).  Non-temporal LOADS lose exactly those halo re-reads (231 V-cycles/s); write-through stores
 * (`sc1`, `sc0 sc1`, `sc1 nt` by inline asm) measured 258-263.  Loading only the read-once rows non-temporally
 * helped the residual + restriction launch while it kept two planes in flight (0.64 -> 0.59 ms) and hurt the smoothing
 * launches (0.72 -> 0.88); with one plane in flight it no longer does (513^3: 0.566 against 0.554 ms without, 257^3:
 * 0.088 against 0.079), so bit 8 is off again. */
typedef double v2d __attribute__((ext_vector_type(2)));
#ifndef MG3D_NT
#define MG3D_NT 2
#endif
/* pointers that went through an opaque scalar (MG3D_OPAQUE_BASE) have lost the compiler's "this is global memory":
 * they carry the address space explicitly, or the accesses become flat_load / flat_store */
#define MG3D_GLOBAL __attribute__((address_space(1)))
typedef const char MG3D_GLOBAL *gcbytes;
typedef char MG3D_GLOBAL *gbytes;
/* a thread's 32-bit byte offset inside a plane, opaque at the point of use: its zero-extension to 64 bits is loop
 * invariant, and once hoisted out of the plane loop the access is `64-bit lane address` (a register pair per row kept
 * across the loop and a 64-bit add per access) instead of the saddr form `SGPR base + 32-bit lane offset` */
__device__ __forceinline__ unsigned lane_off(unsigned off)
{
    asm volatile("" : "+v"(off));
    return off;
}
__device__ __forceinline__ long long lane_off(long long off) { return off; }
template <int BIT> __device__ __forceinline__ double2 ld_stream(gcbytes p)
{
    if constexpr ((MG3D_NT & BIT) != 0) {
        const v2d x = __builtin_nontemporal_load(reinterpret_cast<const v2d MG3D_GLOBAL *>(p));
        return make_double2(x.x, x.y);
    } else {
        const v2d x = *reinterpret_cast<const v2d MG3D_GLOBAL *>(p);
        return make_double2(x.x, x.y);
    }
}
__device__ __forceinline__ void st_stream(gbytes p, double2 o)
{
    v2d x;
    x.x = o.x;
    x.y = o.y;
#if (MG3D_NT & 2)
    __builtin_nontemporal_store(x, reinterpret_cast<v2d MG3D_GLOBAL *>(p));
#else
    *reinterpret_cast<v2d MG3D_GLOBAL *>(p) = x;
#endif
}
template <int BIT> __device__ __forceinline__ double2 ld_stream(const double *p)
{
    if constexpr ((MG3D_NT & BIT) != 0) {
        const v2d x = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p));
        return make_double2(x.x, x.y);
    } else {
        return *reinterpret_cast<const double2 *>(p);
    }
}
__device__ __forceinline__ void st_stream(double *p, double2 o)
{
#if (MG3D_NT & 2)
    v2d x;
    x.x = o.x;
    x.y = o.y;
    __builtin_nontemporal_store(x, reinterpret_cast<v2d *>(p));
#else
    *reinterpret_cast<double2 *>(p) = o;
#endif
}

/* DP: the DP oldest slots of the d window live in LDS instead of VGPRs (own rows only: written once, read by the thread
 * that wrote them, no barrier) -- what lets the down-leg of the cycle (a five-stage window) stay on chip at two waves per SIMD.
 * TAP >= 0: the residual NORM of the state between colour pass TAP and pass TAP + 1 without a stage of its own (see "tap"
 * below; RES == 3 is TAP = S / 2).  TAP = S: only the colour the last pass has updated; TAP = 0: only the colour the
 * first pass is about to update -- the two halves of one norm formed by two consecutive launches. */
#ifndef MG3D_KERNEL_ATTR
#define MG3D_KERNEL_ATTR
#endif
#ifndef MG3D_STAGE_MAJOR
#define MG3D_STAGE_MAJOR 1 /* rows of a one-wave-per-SIMD thread whose stage chains are interleaved in the source (1: row-major).  Same-box A/B at 513^3, round 4: 1 -> 2.450, 2 -> 2.466, 4 -> 2.811 ms per cycle (the scheduler re-orders either way; wider groups only add AGPR traffic) */
#endif
#ifdef MG3D_DEBUG_BLOCKTIMES /* measurement builds only (tools/blocktimes.py): wall clock of every block of the chosen shape at five points of its march */
__device__ unsigned long long g_dbg_bt[8][1024];
#ifndef MG3D_DEBUG_BT_COND
#define MG3D_DEBUG_BT_COND (PRO && S == 4 && TAP == 4)
#endif
#endif
template <int S, int RES, int RJ, int NW, int PF, bool PRO, bool RST, int DP, int TAP, int C1K = -1>
__global__ void __launch_bounds__(NW *WAVE) MG3D_KERNEL_ATTR sweep_kernel(SweepArgs a)
{
    using Sh = SweepShape<S, RES>;
    constexpr int ST = Sh::ST, HJ = Sh::HJ, HI = Sh::HI;
    constexpr int TAPQ = RES == 3 ? S / 2 : TAP;
    constexpr bool HASTAP = TAPQ >= 0;
    constexpr bool NORM = RES == 1 || HASTAP; /* this shape accumulates diff^2 */
    static_assert(DP >= 0 && DP < ST, "at least one slot of the d window stays in registers");
    static_assert(TAPQ <= S, "tap behind the last pass");
    constexpr int KV = ST - DP; /* slots of the d window kept in VGPRs */
    constexpr int TJ = NW * RJ, VJ = TJ - 2 * HJ;
    static_assert(RJ % 2 == 0, "RJ must be even (row parity of a wave's first row)");
    static_assert(VJ > 0 && ST >= 1, "tile too small");
    constexpr int STX = ST > 0 ? ST : 1;

    /* edge rows exchanged between waves: [parity][wave][top/bottom][stage][lane].  Two copies, written at the end of a step
     * and read at the top of the next, cost one barrier a step.  EXS: ONE copy and a second barrier behind the reads at the top
     * of the step (every wave has just left the first one: what it waits for is the LDS latency of its own reads) -- at eight
     * waves that frees 32 + 8 KB of LDS, what the restricting down-leg needs to park two slots of its d window and fit two
     * waves per SIMD */
#ifndef MG3D_EX_SINGLE
#define MG3D_EX_SINGLE(S_, RES_, NW_, DP_) ((RES_) == 2 && (NW_) == 8 && (DP_) > 0)
#endif
    constexpr bool EXS = MG3D_EX_SINGLE(S, RES, NW, DP);
    constexpr int EXB = EXS ? 1 : 2;
    __shared__ double ex[EXB][NW][2][STX][WAVE];
    __shared__ double red[NW];
    __shared__ double2 rex[RES == 2 ? EXB : 1][RES == 2 ? NW : 1][RES == 2 ? WAVE : 1]; /* r pair of a wave's last (CO = 0) / first (CO = 1) row */
    /* RES == 2: the coarse rows a thread completes are centred on its rows 2c + CO -- the even rows of the LEVEL */
    constexpr int CO = RES == 2 ? (HJ & 1) : 0;
    /* PRO: three consecutive coarse planes of the tile's coarse footprint, [plane % 3][row][col] */
    constexpr int CRW = PRO ? (NW * RJ) / 2 + 2 : 1, CCW = PRO ? WAVE + 2 : 1;
    __shared__ double cpl[PRO ? 3 : 1][CRW][CCW];
    /* RES == 2 behind colour passes: the r pairs a thread hands from one step's rows to the next step's restriction are
     * written once and read once a whole step later -- parked in LDS (own rows only: no hazard, no second buffer) they
     * free 4 x RJ VGPRs of a shape that otherwise spills inside the plane loop */
    constexpr bool RPARK = RES == 2 && S > 0 && RJ >= 4;
    __shared__ double2 rpark[RPARK ? NW * RJ : 1][RPARK ? WAVE : 1];
    __shared__ double rkpark[RPARK ? NW * RJ : 1][RPARK ? WAVE : 1]; /* likewise the diff a row keeps for one step (rkeep) */
    /* DP > 0: ring of the DP oldest planes of the d window, [ring slot][row][column][lane] */
    __shared__ double dpk[DP > 0 ? DP : 1][DP > 0 ? NW * RJ : 1][DP > 0 ? 2 : 1][DP > 0 ? WAVE : 1];

    const Geom &g = a.g;
    /* the wave index through readfirstlane: the compiler then knows that everything derived from it (the row
     * flags below) is wave-uniform and keeps it in scalar registers and scalar branches instead of 64-bit lane
     * masks -- the residual variants of this kernel were bound by the CU's one scalar ALU, not by memory */
#ifndef MG3D_UNIFORM_W
#define MG3D_UNIFORM_W 2 /* 0: never, 1: residual variants only, 2: every variant */
#endif
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = (MG3D_UNIFORM_W == 2 || (MG3D_UNIFORM_W == 1 && RES != 0)) ? __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE)
                                                                             : (int)(threadIdx.x / WAVE);
    /* this block's work: one segment -- tile column t_lin, planes off .. off + len of the output range */
    const int nout = a.i_hi - a.i_lo;
    int vb = blockIdx.x;
    if (a.xcd_remap == 1) {
        /* hardware deals blocks round-robin over the 8 XCDs (b % 8 names the XCD group, never which XCD); give
         * each group a contiguous run of shares.  Speed only. */
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, x = vb & 7, idx = vb >> 3;
        vb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
    }
    int t_lin = 0, off = 0, len = 0;
    {
        /* block -> (tile column, i-chunk), tile fastest: all tile columns of a chunk march through the same planes at
         * the same time, so a neighbour's halo rows are still in the Infinity Cache / L2 */
        const int T = a.ntj * a.ntk;
        const int ch = vb / T;
        int tl = vb - ch * T;
        if (a.xcd_remap == 2) {
            /* several rounds of blocks: inside every chunk's layer of T blocks, the blocks of one XCD group take a
             * contiguous run of tile columns (renumbering the whole grid would scatter the first round over all
             * chunks and break the lock-step).  Speed only. */
            const int r0 = (ch * T) & 7, x = (r0 + tl) & 7;
            int o = 0;
            for (int y = 0; y < x; y++) {
                const int first = (y - r0 + 8) & 7;
                o += first < T ? (T - first + 7) >> 3 : 0;
            }
            tl = o + (tl - ((x - r0 + 8) & 7)) / 8;
        }
        t_lin = tl;
        if (a.edge > 0) {
            off = ch == 0 ? 0 : nout - a.edge;
            len = a.edge;
        } else {
            off = ch * a.CI;
            len = min(a.CI, nout - off);
        }
    }
    double acc = 0.;

    /* tiles numbered j-fastest: a run of consecutive tiles -- what one XCD group works on (xcd_remap) -- is then a strip of
     * j-neighbours, whose shared halo is 8 of 32 rows, instead of k-neighbours (8 of 128 columns) with a j-neighbour five tiles
     * away.  Same-box A/B: up-leg 0.733 -> 0.718, four passes 0.673 -> 0.648 ms at 513^3; 1025^3 14.8-15.3 -> 14.2 ms per cycle */
#ifndef MG3D_TILE_J_FASTEST
#define MG3D_TILE_J_FASTEST 1
#endif
    const int tk = MG3D_TILE_J_FASTEST ? t_lin / a.ntj : t_lin % a.ntk, tj = MG3D_TILE_J_FASTEST ? t_lin % a.ntj : t_lin / a.ntk;

    const int jt0 = tj * VJ - HJ, kt0 = tk * a.vk;
    const int own_klo = tk == 0 ? 0 : kt0 + a.hk, own_khi = tk == a.ntk - 1 ? g.nk : kt0 + 2 * WAVE - a.hk;
    const int jrow0 = jt0 + w * RJ;
    const int kA = kt0 + 2 * lane; /* column 0 of the pair; column 1 = kA + 1 */
    const int i_out0 = a.i_lo + off, i_out1 = i_out0 + len;
    /* start plane: HI warm-up planes, one more if needed so that the column active at
     * local step p in row rr is (p + rr) & 1 */
    int i_s = i_out0 - HI;
    /* C1K >= 0: the launcher fixes the first pass's colour at compile time, which makes every plane parity inside the step
     * a constant (the restriction's weights and its store-or-accumulate branch).  Tried for the one-launch down-leg
     * (round 4): the specialised step spills 464 bytes where the runtime parity has none -- not used by any launcher */
    const int c1v = C1K >= 0 ? C1K : a.c1;
    i_s -= (g.ig0 + i_s + jt0 + 1 + c1v) & 1;
    /* the fused restriction finishes a coarse plane one fine plane after its centre, one step late */
    const int nsteps = (i_out1 - 1 + ST) - i_s + 1 + (RES == 2 ? 2 : 0);

    /* a plane in bytes fits 32 bits (the largest level the context admits, 2049^3: 34 MB a plane), so a plane's base is
     * ONE 32 x 32 -> 64-bit scalar product, not the 64 x 64-bit one `long long * int` compiles to (eight scalar
     * instructions a product, four products a step) */
    const unsigned plane_bytes = (unsigned)(g.plane * (long long)sizeof(double));
    /* plane ranges of this segment as (first plane, last - first); an empty range never matches */
    const int upd_first = max(1, 1 - g.ig0), upd_last = min(g.ni - 2, g.N - 2 - g.ig0);
    const int upd_lo = upd_last >= upd_first ? upd_first : 0x3fffffff;
    const unsigned upd_span = upd_last >= upd_first ? (unsigned)(upd_last - upd_first) : 0u;
    const int nrm_first = max(i_out0, a.acc_lo), nrm_last = min(i_out1, a.acc_hi) - 1;
    const bool nrm_any = a.partials != nullptr && nrm_last >= nrm_first;
    const int nrm_lo = nrm_any ? nrm_first : 0x3fffffff;
    const unsigned nrm_span = nrm_any ? (unsigned)(nrm_last - nrm_first) : 0u;

    /* RES == 2: fine planes qq (odd global index) behind which a coarse plane is complete AND to be stored:
     * centre qq - 1 in [i_out0, i_out1), coarse plane (ig0 + qq - 1) / 2 in [1, Nc - 2] and in [ic_lo, ic_hi) locally */
    const int rst_first = max(max(i_out0 + 1, 3 - g.ig0), 2 * (a.ic_lo + a.gc.ig0) + 1 - g.ig0);
    const int rst_last = min(min(i_out1, 2 * a.gc.N - 3 - g.ig0), 2 * (a.ic_hi - 1 + a.gc.ig0) + 1 - g.ig0);
    const int rst_lo = rst_last >= rst_first ? rst_first : 0x3fffffff;
    const unsigned rst_span = rst_last >= rst_first ? (unsigned)(rst_last - rst_first) : 0u;

    /* per-row / per-column masks */
    bool row_in[RJ], row_upd[RJ], row_own[RJ], row_once[RJ];
    /* byte offset of the thread's pair inside a plane: 32 bits (a plane is < 4 GB), so that an access is
     * `uniform 64-bit plane base (SGPRs) + per-lane 32-bit offset` -- the saddr form of global_load / global_store: half
     * the address registers and no 64-bit vector add per access */
#ifndef MG3D_EDGE_UNCOND
#define MG3D_EDGE_UNCOND 1 /* 1: the wave-edge LDS rows are read without a test in every shape (with MG3D_DLAG = 3 the
                              * four-pass shape has the registers for it: 252 VGPRs, no scratch; 257^3 0.120 -> 0.108 ms,
                              * 129^3 0.033 -> 0.028; with MG3D_DLAG = 2 it spills and loses 20 %) */
#endif
#ifndef MG3D_DLAG
#define MG3D_DLAG 3 /* bit 0: the four-pass smoothing shape, bit 1: every other shape -- d trails u by one plane (load_plane); same-box A/B at 513^3: residual + restriction 0.551 -> 0.532 ms, prolongation + 2 passes 0.697 -> 0.679; the four-pass shape needs it to run without scratch once its edge rows are read unconditionally */
#endif
#ifndef MG3D_ADDR32
#define MG3D_ADDR32 3 /* same bits: 32-bit per-lane offsets instead of 64-bit ones */
#endif
    constexpr int SHAPE_BIT = (S == 4 && (RES == 0 || RES == 3)) ? 0 : 1;
    constexpr int DLAG = (MG3D_DLAG >> SHAPE_BIT) & 1;
    static_assert(DP == 0 || DLAG == 1, "the parked d window assumes d trails u by one plane");
    constexpr bool A32 = ((MG3D_ADDR32 >> SHAPE_BIT) & 1) != 0;
    /* a row's offset = a wave-uniform part (row, scalar registers) + ONE per-lane part (column) for all rows: RJ - 1 VGPRs
     * less than an offset per row */
    unsigned row_base[RJ];
    static_assert(A32, "32-bit offsets inside a plane");
#pragma unroll
    for (int rr = 0; rr < RJ; rr++) {
        const int j = jrow0 + rr;
        row_in[rr] = j >= 0 && j < g.nj;
        row_upd[rr] = j >= 1 && j <= g.nj - 2;
        row_own[rr] = row_in[rr] && j >= tj * VJ && j < (tj + 1) * VJ;
        /* rows no other tile column reads (the outer HJ owned rows are the neighbours' halo) */
        row_once[rr] = row_in[rr] && j >= tj * VJ + HJ && j < (tj + 1) * VJ - HJ;
        /* loads are UNCONDITIONAL (load_plane): rows, columns and planes outside the level are clamped onto it.  What
         * they deliver there is never used -- a point of the level only reads neighbours inside the level, and the points
         * on its faces are passed through, not computed -- so clamping replaces a guard (two scalar ANDs, an EXEC save, a
         * branch and a restore per row and field: half of the step's scalar instructions) by nothing */
        const int jc = j < 0 ? 0 : (j >= g.nj ? g.nj - 1 : j), kc = 0;
        (void)kc;
        row_base[rr] = (unsigned)((long long)g.pitch * jc * (long long)sizeof(double));
    }
    const unsigned col_off = (unsigned)((kA > g.pitch - 2 ? g.pitch - 2 : kA) * (int)sizeof(double));
    const bool col_in[2] = {kA >= 0 && kA < g.nk, kA + 1 >= 0 && kA + 1 < g.nk};
    const bool col_upd[2] = {kA >= 1 && kA <= g.nk - 2, kA + 1 >= 1 && kA + 1 <= g.nk - 2};
    const bool pair_own = kA >= own_klo && kA < own_khi && col_in[0];
    /* lane masks used inside the plane loop (the row flags are wave-uniform scalars) */
    const bool own_upd[2] = {pair_own && col_upd[0], pair_own && col_upd[1]};
    const bool own_both = own_upd[0] && own_upd[1], own_only0 = own_upd[0] && !col_upd[1],
               own_only1 = own_upd[1] && !col_upd[0];
    /* (tried in round 4 for the one-wave-per-SIMD shapes: the column masks as 64-bit lane masks in scalar registers and
     * the select as two v_cndmask_b32 by inline asm instead of the AND + compare + two selects on byte-per-lane
     * booleans the compiler emits -- the scalar registers it takes are spilled to VGPR lanes, v_readlane / v_writelane
     * 117 -> 327 per three steps, 4933 -> 5127 instructions: not kept) */
    const bool k_edge_tile = tk == 0 || tk == a.ntk - 1; /* only there a pair can have one updatable column */
    /* the per-lane flags the plane loop tests, as bits of ONE register (a flag each costs a VGPR: with the scalar registers
     * full the compiler keeps a lane's boolean as a 0 / 1 dword); taken through an opaque copy at the top of every step so
     * that `flags & bit` is not hoisted out of the loop into one register per bit again */
    enum { LM_UPD0 = 1, LM_UPD1 = 2, LM_OWN0 = 4, LM_OWN1 = 8, LM_PAIR = 16, LM_CCOL = 32, LM_BOTH = 64, LM_ONLY0 = 128, LM_ONLY1 = 256 };
#define LM(bit) ((lm & (unsigned)(bit)) != 0u)
    /* RES == 2: the coarse points this thread completes -- rows centred on its even rows, its even column */
    bool crow_ok[RJ / 2];
    unsigned dc_row[RJ / 2]; /* wave-uniform: the coarse row's byte offset inside a coarse plane; the lane's part is its column */
    const unsigned dc_col = (unsigned)((kA >> 1) * (int)sizeof(double));
    const bool ccol_ok = pair_own && (kA >> 1) >= 1 && (kA >> 1) <= a.gc.nk - 2;
    const unsigned lmask = (col_upd[0] ? LM_UPD0 : 0) | (col_upd[1] ? LM_UPD1 : 0) | (own_upd[0] ? LM_OWN0 : 0) | (own_upd[1] ? LM_OWN1 : 0) |
                           (pair_own ? LM_PAIR : 0) | (ccol_ok ? LM_CCOL : 0) | (own_both ? LM_BOTH : 0) | (own_only0 ? LM_ONLY0 : 0) |
                           (own_only1 ? LM_ONLY1 : 0);
#pragma unroll
    for (int c = 0; c < RJ / 2; c++) {
        const int jc = (jrow0 + 2 * c + CO) >> 1;
        crow_ok[c] = row_own[2 * c + CO] && jc >= 1 && jc <= a.gc.nj - 2;
        dc_row[c] = (unsigned)((long long)a.gc.pitch * (jc < 0 ? 0 : jc) * (long long)sizeof(double));
    }

    /* pipeline state (see header).  last[rr][s][c]: newest stage-s output of column c */
    double last[RJ][STX][2], in_prev[RJ][2], dring[RJ][ST + 1][2], rkeep[RJ];
    double2 cur_v[RJ], nxt_v[PF][RJ], nxt_d[PF][RJ]; /* planes in flight from HBM: PF ahead */
#pragma unroll
    for (int rr = 0; rr < RJ; rr++) {
#pragma unroll
        for (int s = 0; s < STX; s++)
            last[rr][s][0] = last[rr][s][1] = 0.;
#pragma unroll
        for (int s = 0; s <= ST; s++)
            dring[rr][s][0] = dring[rr][s][1] = 0.;
        in_prev[rr][0] = in_prev[rr][1] = 0.;
        rkeep[rr] = 0.;
        cur_v[rr] = make_double2(0., 0.);
    }
    double2 rlag[RJ];           /* RES == 2: r pairs of the plane finished by the previous step */
    double racc[RJ / 2];        /* running 27-point sums, one per coarse row centred in this thread's rows */
#pragma unroll
    for (int rr = 0; rr < RJ; rr++) {
        rlag[rr] = make_double2(0., 0.);
        if constexpr (RPARK)
            rpark[w * RJ + rr][lane] = make_double2(0., 0.), rkpark[w * RJ + rr][lane] = 0.;
    }
#pragma unroll
    for (int c = 0; c < RJ / 2; c++)
        racc[c] = 0.;

    /* u of plane i and d of plane i - 1: no stage reads d of the plane that has just arrived (stage s works on plane
     * i - s, s >= 1), so its load trails u's by one step and lands straight in the first slot of the d window -- one
     * slot (2 x RJ doubles: 16 VGPRs of a register file that every shape fills) less than loading both together */
    auto load_plane = [&](int i, double2(&vv)[RJ], double2(&dd)[RJ]) {
#ifdef MG3D_EXPERIMENT_SAME_PLANE /* timing experiment only (wrong results): every load hits the same, cached, plane */
        const int iu = i_s < 0 ? 0 : i_s, id = iu;
        (void)i;
#else
        const int iu = i < 0 ? 0 : (i >= g.ni ? g.ni - 1 : i), id = i - DLAG < 0 ? 0 : (i - DLAG >= g.ni ? g.ni - 1 : i - DLAG);
#endif
        /* plane bases in bytes, uniform: one scalar 64-bit product per plane, not one re-materialised per row */
        long long pbase = (long long)((unsigned long long)plane_bytes * (unsigned)iu), pbase_d = (long long)((unsigned long long)plane_bytes * (unsigned)id);
        asm volatile("" : "+s"(pbase), "+s"(pbase_d));
        /* the plane's base ADDRESS opaque, not only its offset: otherwise `field + plane offset + row offset` is
         * re-associated into a loop-invariant 64-bit `field + row offset` per row and field (2 VGPRs each, kept across the
         * plane loop) plus a 64-bit vector add per access, instead of the saddr form `SGPR base + 32-bit lane offset` */
        unsigned long long ub_ = reinterpret_cast<unsigned long long>(a.vin) + (unsigned long long)pbase,
                           db_ = reinterpret_cast<unsigned long long>(a.d) + (unsigned long long)pbase_d;
        asm volatile("" : "+s"(ub_), "+s"(db_));
        const gcbytes ubase = reinterpret_cast<gcbytes>(ub_), dbase = reinterpret_cast<gcbytes>(db_);
        /* vin == NULL: the input field is identically zero (a coarse level's initial guess, mg_3d.h:1258-1259) --
         * neither zeroed in memory beforehand nor read */
        const bool have_u = a.vin != nullptr; /* uniform */
        /* (one uniform test for the whole plane, not one per row: the rows' loads stay a straight line) */
        if (have_u) {
#pragma unroll
            for (int rr = 0; rr < RJ; rr++) {
                const gcbytes pu = ubase + row_base[rr] + lane_off(col_off);
#if (MG3D_NT & 8)
                if (S == 0 && row_once[rr]) /* wave-uniform */
                    vv[rr] = ld_stream<8>(pu);
                else
#endif
                    vv[rr] = ld_stream<1>(pu);
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < RJ; rr++)
                vv[rr] = make_double2(0., 0.);
        }
#pragma unroll
        for (int rr = 0; rr < RJ; rr++) {
            const gcbytes pd = dbase + row_base[rr] + lane_off(col_off);
#if (MG3D_NT & 8)
            if (S == 0 && row_once[rr])
                dd[rr] = ld_stream<8>(pd);
            else
#endif
                dd[rr] = ld_stream<4>(pd);
        }
    };

    /* ---- PRO: coarse planes staged in LDS.  Plane c (local coarse index) lives in slot c mod 3. */
    const int jcb = jt0 >> 1, kcb = kt0 >> 1; /* coarse origin of the tile (jt0, kt0 are even) */
    auto coarse_of = [&](int i) { /* lower coarse parent plane (local) of fine local plane i */
        const int ig = g.ig0 + i, oi = ig & 1;
        return (ig - oi) / 2 - a.gce.ig0;
    };
    auto slot_of = [](int c) { return ((c % 3) + 3) % 3; };
    constexpr int CPT = PRO ? (CRW * CCW + NW * WAVE - 1) / (NW * WAVE) : 1; /* staged values per thread */
    auto coarse_fetch = [&](int c, double(&buf)[CPT]) {
        /* unconditional, from clamped indices (as load_plane): a fine point of the level has its parents inside the
         * coarse level, what is staged for positions outside it is never used */
        const int cc = c < 0 ? 0 : (c >= a.gce.ni ? a.gce.ni - 1 : c);
        const long long cbase = a.gce.plane * cc;
#pragma unroll
        for (int t = 0; t < CPT; t++) {
            int idx = threadIdx.x + t * NW * WAVE;
            idx = idx < CRW * CCW ? idx : CRW * CCW - 1;
            const int row = idx / CCW, col = idx - row * CCW;
            int jc = jcb + row, kc = kcb + col;
            jc = jc < 0 ? 0 : (jc >= a.gce.nj ? a.gce.nj - 1 : jc);
            kc = kc < 0 ? 0 : (kc >= a.gce.nk ? a.gce.nk - 1 : kc);
            buf[t] = a.ec[cbase + (long long)a.gce.pitch * jc + kc];
        }
    };
    auto coarse_put = [&](int c, const double(&buf)[CPT]) {
        const int sl = slot_of(c);
#pragma unroll
        for (int t = 0; t < CPT; t++) {
            const int idx = threadIdx.x + t * NW * WAVE;
            if (idx < CRW * CCW)
                (&cpl[sl][0][0])[idx] = buf[t];
        }
    };
    int have_hi = 0; /* highest coarse plane staged so far */
    /* v_in = u + P(ec) of fine local plane ip (global parity OI), added into the plane's registers: parents summed in the
     * reference's order per parity class (see prolong_kernel) */
    auto prolong_into = [&](double2(&vv)[RJ], int ip, auto oi_c) {
        constexpr int oi = decltype(oi_c)::value;
        const int s0 = slot_of(coarse_of(ip)), s1 = slot_of(coarse_of(ip) + 1);
#pragma unroll
        for (int rr = 0; rr < RJ; rr++) {
            const int oj = rr & 1; /* jrow0 is even */
            const int lr = (w * RJ + rr) >> 1;
            const double e000 = cpl[s0][lr][lane], e001 = cpl[s0][lr][lane + 1];
            double t0, t1;
            if (!oi && !oj) {
                t0 = e000;
                t1 = (e000 + e001) * 0.5;
            } else if (!oi) {
                const double e010 = cpl[s0][lr + 1][lane], e011 = cpl[s0][lr + 1][lane + 1];
                t0 = (e000 + e010) * 0.5;
                t1 = (((e000 + e010) + e001) + e011) * 0.25;
            } else if (!oj) {
                const double e100 = cpl[s1][lr][lane], e101 = cpl[s1][lr][lane + 1];
                t0 = (e000 + e100) * 0.5;
                t1 = (((e000 + e100) + e001) + e101) * 0.25;
            } else {
                const double e010 = cpl[s0][lr + 1][lane], e011 = cpl[s0][lr + 1][lane + 1];
                const double e100 = cpl[s1][lr][lane], e101 = cpl[s1][lr][lane + 1];
                const double e110 = cpl[s1][lr + 1][lane], e111 = cpl[s1][lr + 1][lane + 1];
                t0 = (((e000 + e010) + e100) + e110) * 0.25;
                double t = e000 + e001;
                t = t + e010;
                t = t + e011;
                t = t + e100;
                t = t + e101;
                t = t + e110;
                t = t + e111;
                t1 = t * 0.125;
            }
            vv[rr].x += t0;
            vv[rr].y += t1;
        }
    };
    /* MG3D_PRO_LATE: a plane receives its prolongation at the END of the step before the one that consumes it (behind the
     * step's stores, in the registers the load has landed in), not at the top of its own step -- where the plane in use,
     * the plane in flight, the whole window and the parents of a row are all live at once: that peak is what kept the
     * prolonging four-pass shape from two waves per SIMD (256 VGPRs + 68 bytes of scratch).  The coarse planes are staged
     * one step earlier for it (three primed instead of two). */
#ifndef MG3D_PRO_LATE
#define MG3D_PRO_LATE 1
#endif
    static_assert(!PRO || PF <= 2, "PRO: one or two planes in flight");
    if constexpr (PRO) {
        double buf[CPT];
        const int c0 = coarse_of(i_s);
#pragma unroll
        for (int c = 0; c < (MG3D_PRO_LATE ? 3 : 2); c++) {
            coarse_fetch(c0 + c, buf);
            coarse_put(c0 + c, buf);
        }
        have_hi = c0 + (MG3D_PRO_LATE ? 2 : 1);
        __syncthreads();
    }

    /* prime the prefetch queue with planes i_s .. i_s+PF-1 */
#pragma unroll
    for (int f = 0; f < PF; f++)
        load_plane(i_s + f, nxt_v[f], nxt_d[f]);
    if constexpr (PRO && MG3D_PRO_LATE != 0)
        prolong_into(nxt_v[0], i_s, std::integral_constant<int, 1>{}); /* plane i_s is odd in the step parity's sense (see the step) */

    int ring = 0; /* DP > 0: LDS ring slot this step's parked plane goes to (= step number mod DP) */
    auto step = [&](int pl, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        const int i = i_s + pl; /* local plane just arrived */
        const int par = pl & 1;
        unsigned lm = lmask;
        asm volatile("" : "+v"(lm));
        /* "this column is updatable" as two lane masks formed once a step (the other flags are tested where they are used: as
         * step-wide masks they cost more scalar registers than the shapes at the limit have): up-leg -1 % */
        const bool upd_col[2] = {(lm & (unsigned)LM_UPD0) != 0u, (lm & (unsigned)LM_UPD1) != 0u};
        /* current plane <- head of the prefetch queue, then request plane i+PF */
        if constexpr (PF == 2) {
            /* two planes in flight as a RING indexed by the step's parity (a template argument), not a queue that shifts:
             * shifting copies the registers of the plane still in flight, and a copy has to wait for its load -- with
             * the queue a "second plane in flight" was never in flight for more than one step */
#pragma unroll
            for (int rr = 0; rr < RJ; rr++) {
                cur_v[rr] = nxt_v[PAR][rr];
                dring[rr][0][0] = nxt_d[PAR][rr].x;
                dring[rr][0][1] = nxt_d[PAR][rr].y;
            }
            load_plane(i + PF, nxt_v[PAR], nxt_d[PAR]);
        } else {
#pragma unroll
            for (int rr = 0; rr < RJ; rr++) {
                cur_v[rr] = nxt_v[0][rr];
                dring[rr][0][0] = nxt_d[0][rr].x;
                dring[rr][0][1] = nxt_d[0][rr].y;
#pragma unroll
                for (int f = 0; f + 1 < PF; f++) {
                    nxt_v[f][rr] = nxt_v[f + 1][rr];
                    nxt_d[f][rr] = nxt_d[f + 1][rr];
                }
            }
            load_plane(i + PF, nxt_v[PF - 1], nxt_d[PF - 1]);
        }
        double cbuf[CPT];
        bool stage_new = false;
        if constexpr (PRO) {
            /* parity of a plane's global index: the segment starts where (ig0 + i_s + jt0 + 1 + c1) is even, a PRO tile
             * starts on an even row and PRO launches are post-smoothers (c1 = 0, the launchers refuse anything else), so it
             * is the step's parity -- known at compile time: the parent ladder is straight-line code */
            static_assert(!PRO || (HJ % 2 == 0 && VJ % 2 == 0), "PRO: tiles start on even rows");
            if constexpr (MG3D_PRO_LATE != 0) {
                /* the END of this step prolongs plane i + 1, the end of the next one plane i + 2: its parents (up to coarse
                 * plane coarse_of(i + 2) + 1) are fetched now and published before this step's barrier */
                stage_new = coarse_of(i + 2) + 1 > have_hi;
                if (stage_new)
                    coarse_fetch(have_hi + 1, cbuf);
            } else {
                /* the next step needs coarse planes up to coarse_of(i+1)+1: fetch one now, publish before the barrier */
                stage_new = coarse_of(i + 1) + 1 > have_hi;
                if (stage_new)
                    coarse_fetch(have_hi + 1, cbuf);
                prolong_into(cur_v, i, std::integral_constant<int, (1 + PAR) & 1>{});
            }
        }

        /* rows of the neighbouring waves, written at the end of the previous step */
        double e_top[STX], e_bot[STX];
#pragma unroll
        for (int s = 0; s < STX; s++) {
            /* the first / last wave has no neighbour: its outermost row is the tile's outermost halo row (or lies outside
             * the grid), whose results are never used -- it reads its own row instead of branching around the read.
             * Not in the two shapes that sit at 256 VGPRs: there the eight unconditional reads at the top of the step
             * lengthen live ranges into scratch spills inside the plane loop (measured 0.68 -> 0.94 ms at 513^3). */
            if constexpr (MG3D_EDGE_UNCOND || (S < 4 && (RES != 2 || PF == 1))) {
                e_top[s] = ex[EXS ? 0 : par ^ 1][w > 0 ? w - 1 : 0][1][s][lane];
                e_bot[s] = ex[EXS ? 0 : par ^ 1][w < NW - 1 ? w + 1 : NW - 1][0][s][lane];
            } else {
                e_top[s] = (w > 0) ? ex[EXS ? 0 : par ^ 1][w - 1][1][s][lane] : 0.;
                e_bot[s] = (w < NW - 1) ? ex[EXS ? 0 : par ^ 1][w + 1][0][s][lane] : 0.;
            }
        }
        /* the one row of another wave a thread's restriction needs (RES == 2): CO = 0 the last row of the wave above (first
         * coarse row), CO = 1 the first row of the wave below (last coarse row).  The outermost wave reads its own: a halo
         * row's sum, never stored */
        double2 nbr = make_double2(0., 0.);
        if constexpr (RES == 2)
            nbr = rex[EXS ? 0 : par ^ 1][CO == 0 ? (w > 0 ? w - 1 : 0) : (w < NW - 1 ? w + 1 : NW - 1)][lane];
        if constexpr (EXS)
            __syncthreads(); /* every wave holds its neighbours' rows: the single copy may be overwritten at the end of this step */
        /* store planes, bases in bytes */
        /* (planes outside the level: any product will do, the stores are guarded by v_ok / r_ok) */
        long long vbase = (long long)((unsigned long long)plane_bytes * (unsigned)(i - S)), rbase = (long long)((unsigned long long)plane_bytes * (unsigned)(i - ST));
        asm volatile("" : "+s"(vbase), "+s"(rbase));
        unsigned long long vb_ = reinterpret_cast<unsigned long long>(a.vout) + (unsigned long long)vbase,
                           rb_ = reinterpret_cast<unsigned long long>(a.r) + (unsigned long long)rbase;
        asm volatile("" : "+s"(vb_), "+s"(rb_));
        const gbytes voutb = reinterpret_cast<gbytes>(vb_), routb = reinterpret_cast<gbytes>(rb_);
        /* which planes may be updated (global boundary planes / slab halos are not), which enter the norm, which are
         * stored: each a range of planes fixed per segment (upd_lo .. below), tested with ONE unsigned compare
         * (q - lo <= span) instead of four signed ones and the branches a short-circuit turns them into */
        bool pl_upd[STX + 1], acc_ok[STX + 1];
#pragma unroll
        for (int s = 1; s <= ST; s++) {
            const int q = i - s;
            pl_upd[s] = (unsigned)(q - upd_lo) <= upd_span;
            acc_ok[s] = NORM ? (unsigned)(q - nrm_lo) <= nrm_span : false; /* planes that enter the norm */
        }
        /* this step's store planes lie in the output range (wave-uniform, once per step, not once per row) */
        const bool v_ok = (unsigned)(i - S - i_out0) < (unsigned)len;
        const bool r_ok = RES == 1 ? (a.r != nullptr) & ((unsigned)(i - ST - i_out0) < (unsigned)len) & pl_upd[ST] : false;
/* wave-uniform tests joined without short-circuit where that is free (one scalar AND instead of a branch per operand);
 * the restriction shape spills with it (scratch 36 -> 96 bytes, 0.59 -> 1.0 ms), the pure smoothers gain nothing */
#define MG3D_AND(x, y) ((RES == 1 || RES == 3 || (HASTAP && RES != 2)) ? ((x) & (y)) : ((x) && (y)))

        if constexpr (RES == 2) {
            /* (BEFORE this step's rows: they overwrite rlag in place -- a second set of r pairs would be 4 x RJ VGPRs)
             * Full weighting of plane qq = i-ST-1 (its r pairs are in rlag; the row above this wave's first
             * row was published by the wave above at the end of the previous step).  Coarse row centres sit
             * on this thread's rows rr = CO, CO + 2, ..; coarse column = this lane's even column kA.  The
             * reference adds the 27 products r*w in the order ti, tj, tk (mg_3d.h:980-988): planes arrive
             * in ti order, and inside a plane the nine terms below are tj-major, tk-minor. */
            const int qq = i - ST - 1, qg = g.ig0 + qq;
            /* (ig0 + i_s + jt0 + 1 + c1) is even, jt0 has the parity of HJ, the step's that of PAR */
            const bool odd = C1K >= 0 ? ((HJ + 1 + C1K + PAR + ST + 1) & 1) != 0 : (qg & 1) != 0;
            const double wi = odd ? 0.25 : 0.5;
#pragma unroll
            for (int c = 0; c < RJ / 2; c++) {
                double2 r0, r1, r2;
                constexpr int RLO = -1 + CO; /* rows 2c + RLO .. 2c + RLO + 2 */
                const bool lo_out = 2 * c + RLO < 0, hi_out = 2 * c + RLO + 2 > RJ - 1; /* compile time after unrolling */
                if constexpr (RPARK) {
                    r0 = lo_out ? nbr : rpark[w * RJ + (lo_out ? 0 : 2 * c + RLO)][lane];
                    r1 = rpark[w * RJ + 2 * c + RLO + 1][lane];
                    r2 = hi_out ? nbr : rpark[w * RJ + (hi_out ? 0 : 2 * c + RLO + 2)][lane];
                } else {
                    r0 = lo_out ? nbr : rlag[lo_out ? 0 : 2 * c + RLO];
                    r1 = rlag[2 * c + RLO + 1];
                    r2 = hi_out ? nbr : rlag[hi_out ? 0 : 2 * c + RLO + 2];
                }
                const double l0 = lane_from_left(r0.y), l1 = lane_from_left(r1.y), l2 = lane_from_left(r2.y);
                const double p[9] = {l0 * (wi * 0.25 * 0.25), r0.x * (wi * 0.25 * 0.5), r0.y * (wi * 0.25 * 0.25),
                                     l1 * (wi * 0.5 * 0.25),  r1.x * (wi * 0.5 * 0.5),  r1.y * (wi * 0.5 * 0.25),
                                     l2 * (wi * 0.25 * 0.25), r2.x * (wi * 0.25 * 0.5), r2.y * (wi * 0.25 * 0.25)};
                double run = racc[c];
#pragma unroll
                for (int t = 0; t < 9; t++)
                    run = run + p[t];
                if (odd) {
                    /* qq is the ti = 2 plane of coarse plane (qg-1)/2 and the ti = 0 plane of (qg+1)/2.  The coarse plane
                     * is stored when its centre plane qq - 1 lies in this segment's output range, it is an interior
                     * plane of the coarse level and one this launch is to write: one range of qq (rst_lo, rst_span) */
                    if (((unsigned)(qq - rst_lo) <= rst_span) & crow_ok[c]) { /* wave-uniform */
                        const int icl = ((qg - 1) >> 1) - a.gc.ig0;
                        if (LM(LM_CCOL)) {
                            const unsigned long long cb_ = reinterpret_cast<unsigned long long>(a.dc) +
                                                           (unsigned long long)(a.gc.plane * icl) * sizeof(double) + dc_row[c];
                            *reinterpret_cast<double MG3D_GLOBAL *>(reinterpret_cast<gbytes>(cb_) + lane_off(dc_col)) = run;
                        }
                    }
                    double fresh = 0.;
#pragma unroll
                    for (int t = 0; t < 9; t++)
                        fresh = fresh + p[t];
                    racc[c] = fresh;
                } else {
                    racc[c] = run;
                }
            }
        }
        /* The rows' stage chains.  ROW-major (every stage of a row, then the next row) is the order the two-waves-per-SIMD
         * shapes were tuned in: the other wave fills the gaps of a dependent chain.  The one-wave-per-SIMD shapes (eight rows a
         * thread) run STAGE-major -- stage s of all eight rows, then stage s + 1: consecutive instructions then belong to
         * independent rows, where row-major issued five dependent fp64 adds back to back with nobody to hide their latency
         * behind (SQ counters of the one-launch legs, round 4: 25 % of the wave's cycles issue stalls, 22 % waits). */
        constexpr int ILV = (NW == 4 && RJ == 8 && RJ % MG3D_STAGE_MAJOR == 0) ? MG3D_STAGE_MAJOR : 1;
        /* ILV rows at a time, stage-major inside the group (no lambdas: wrapping the bodies into closures made the tapped
         * four-pass shape -- 241 VGPRs -- spill 44 bytes) */
#pragma unroll
        for (int r0 = 0; r0 < RJ; r0 += ILV) {
            double nwG[ILV][STX + 1], diffsG[ILV][2];
#pragma unroll
            for (int g = 0; g < ILV; g++) {
                const int X = (PAR + r0 + g) & 1;
                nwG[g][0] = X ? cur_v[r0 + g].y : cur_v[r0 + g].x;
                diffsG[g][0] = diffsG[g][1] = 0.;
            }
#pragma unroll
            for (int s = 1; s <= ST; s++) {
#pragma unroll
                for (int g = 0; g < ILV; g++) {
                    const int rr = r0 + g;
                    const int X = (PAR + rr) & 1; /* active column of this row at this step */
                        const double up = last[rr][s - 1][X]; /* plane q-1: two steps old */
                        const double dn = nwG[g][s - 1];          /* plane q+1: this step */
                        const double jm = (rr > 0) ? last[rr - 1][s - 1][X] : e_top[s - 1];
                        const double jp = (rr < RJ - 1) ? last[rr + 1][s - 1][X] : e_bot[s - 1];
                        double km, kp;
                        if (X == 0) {
                            km = lane_from_left(last[rr][s - 1][1]);
                            kp = last[rr][s - 1][1];
                        } else {
                            km = last[rr][s - 1][0];
                            kp = lane_from_right(last[rr][s - 1][0]);
                        }
                        double dd; /* DLAG: slot 0 = plane i - 1 (load_plane) */
                        if constexpr (DP > 0) {
                            if (s - 1 < KV) {
                                dd = dring[rr][s - 1][X];
                            } else {
                                /* slot KV + m was parked at the end of step pl - 1 - m, into ring slot (pl - 1 - m) mod DP */
                                const int m = s - 1 - KV; /* compile-time after unrolling */
                                int idx = ring + (DP - 1 - m);
                                idx -= idx >= DP ? DP : 0;
                                dd = dpk[idx][w * RJ + rr][X][lane];
                            }
                        } else {
                            dd = dring[rr][s - DLAG][X];
                        }
                        const double center = (s == 1) ? in_prev[rr][X] : last[rr][s - 2][X];
                        double sum = up + dn;
                        sum = sum + jm;
                        sum = sum + jp;
                        sum = sum + km;
                        sum = sum + kp;
                        /* `&`, not `&&`: a short-circuit on the wave-uniform part turns every update into a scalar branch around
                         * it (78 branches a step); the select costs nothing and leaves one basic block to schedule */
                        const bool updu = row_upd[rr] & pl_upd[s]; /* wave-uniform part of "this point is updated" */
                        if (s <= S) {
        #ifdef MG3D_EXPERIMENT_DROP_FLOPS /* timing experiment only (wrong results): is the step bound by its fp64 operations? */
                            const double val = sum - dd;
        #else
                            const double val = a.sixth * (sum - a.hSq * dd); /* mg_3d.h:438-443 */
        #endif
                            if constexpr (RES == 2) /* (the restricting shapes sit at the register limit: the step-wide lane masks push them into scratch) */
                                nwG[g][s] = LM(updu ? (X ? LM_UPD1 : LM_UPD0) : 0) ? val : center;
                            else
                                nwG[g][s] = (updu & upd_col[X]) ? val : center;
                            if constexpr (HASTAP) {
                                /* The tap: the residual norm of the state BETWEEN pass TAPQ and pass TAPQ + 1 without a stage of its
                                 * own.  The colour pass TAPQ has just updated: its residual uses that pass's neighbour sum (as
                                 * below).  The other colour: pass TAPQ + 1 is about to update it from exactly the six neighbours
                                 * (all of the colour pass TAPQ + 1 leaves alone) and the centre (untouched by pass TAPQ) that the
                                 * residual of the tapped state is made of -- mg_3d.h:819-821 on the sum the update forms anyway.
                                 * TAPQ = S has only the first half, TAPQ = 0 only the second: two launches, one norm. */
                                if (s == TAPQ || s == TAPQ + 1) {
                                    const double diff = dd - a.invHsq * (sum - 6 * (s == TAPQ ? nwG[g][s] : center));
                                    if ((updu & row_own[rr]) & acc_ok[s])
                                        acc += LM(X ? LM_OWN1 : LM_OWN0) ? diff * diff : 0.;
                                }
                            }
                            if ((RES == 1 || RES == 2) && s == S) { /* residual of the point just updated: same six neighbours */
                                const double diff = dd - a.invHsq * (sum - 6 * nwG[g][s]); /* mg_3d.h:819-821 */
                                diffsG[g][0] = diff;
                                /* adding +0 leaves a sum of squares unchanged: a select, not a branch */
                                if constexpr (RES == 1)
                                    if (MG3D_AND(MG3D_AND(updu, row_own[rr]), acc_ok[s]))
                                        acc += LM(X ? LM_OWN1 : LM_OWN0) ? diff * diff : 0.;
                            }
                        } else {
                            const double diff = dd - a.invHsq * (sum - 6 * center); /* mg_3d.h:819-821 */
                            nwG[g][s] = center;
                            diffsG[g][S > 0 ? 1 : s - 1] = diff;
                            if constexpr (RES == 1)
                                if (MG3D_AND(MG3D_AND(updu, row_own[rr]), acc_ok[s]))
                                    acc += LM(X ? LM_OWN1 : LM_OWN0) ? diff * diff : 0.;
                        }

                }
            }
#pragma unroll
            for (int g = 0; g < ILV; g++) {
                const int rr = r0 + g;
                const int X = (PAR + rr) & 1;
                /* ---- stores: v' of plane i-S, r of plane i-S-2 (pairs complete at this step) */
                if constexpr (S > 0) {
                    if (MG3D_AND(v_ok, row_own[rr])) { /* wave-uniform */
                        const double other = last[rr][S - 1][X ^ 1]; /* finished one step ago */
                        double2 o;
                        o.x = X ? other : nwG[g][S];
                        o.y = X ? nwG[g][S] : other;
                        if (LM(LM_PAIR))
                            st_stream(voutb + row_base[rr] + lane_off(col_off), o);
                    }
                }
                if constexpr (RES == 2 || (RES == 1 && RST)) { /* RST = false: the norm only, r is not assembled */
                    /* column X: the residual-only stage now; column X^1: the previous step's diff */
                    double2 o;
                    const double kept = RPARK ? rkpark[w * RJ + rr][lane] : rkeep[rr];
                    o.x = X ? kept : diffsG[g][1];
                    o.y = X ? diffsG[g][1] : kept;
                    if constexpr (RES == 2) { /* read by the NEXT step's restriction */
                        if constexpr (RPARK)
                            rpark[w * RJ + rr][lane] = o;
                        else
                            rlag[rr] = o;
                        if (rr == (CO == 0 ? RJ - 1 : 0))
                            rex[EXS ? 0 : par][w][lane] = o;
                    }
                    if (RES == 1 && MG3D_AND(MG3D_AND(r_ok, row_own[rr]), row_upd[rr])) { /* wave-uniform */
                        double MG3D_GLOBAL *dst = reinterpret_cast<double MG3D_GLOBAL *>(routb + row_base[rr] + lane_off(col_off));
                        if (LM(LM_BOTH)) {
                            v2d x;
                            x.x = o.x;
                            x.y = o.y;
                            *reinterpret_cast<v2d MG3D_GLOBAL *>(dst) = x;
                        }
                        if (k_edge_tile) { /* boundary entries of r are never written (mg_3d.h:824-825) */
                            if (LM(LM_ONLY0))
                                dst[0] = o.x;
                            if (LM(LM_ONLY1))
                                dst[1] = o.y;
                        }
                    }
                    if constexpr (RPARK)
                        rkpark[w * RJ + rr][lane] = diffsG[g][0];
                    else
                        rkeep[rr] = diffsG[g][0];
                }
                /* ---- commit this row's new outputs */
                in_prev[rr][0] = cur_v[rr].x;
                in_prev[rr][1] = cur_v[rr].y;
    #pragma unroll
                for (int s = 0; s < ST; s++)
                    last[rr][s][X] = nwG[g][s];

            }
        }
        /* age the d window (slots 0 .. ST-1 = planes i-1 .. i-ST) */
        if constexpr (DP > 0) {
            /* the plane leaving the last register slot goes to the LDS ring (over the plane whose last reader ran above) */
#pragma unroll
            for (int rr = 0; rr < RJ; rr++) {
                dpk[ring][w * RJ + rr][0][lane] = dring[rr][KV - 1][0];
                dpk[ring][w * RJ + rr][1][lane] = dring[rr][KV - 1][1];
            }
            ring = ring + 1 == DP ? 0 : ring + 1;
        }
#pragma unroll
        for (int rr = 0; rr < RJ; rr++)
#pragma unroll
            for (int s = (DP > 0 ? KV - 1 : ST - DLAG); s >= 1; s--) {
                dring[rr][s][0] = dring[rr][s - 1][0];
                dring[rr][s][1] = dring[rr][s - 1][1];
            }
        /* publish this wave's edge rows for the next step */
#pragma unroll
        for (int s = 0; s < ST; s++) {
            ex[EXS ? 0 : par][w][0][s][lane] = last[0][s][(PAR + 0) & 1];
            ex[EXS ? 0 : par][w][1][s][lane] = last[RJ - 1][s][(PAR + RJ - 1) & 1];
        }
        if constexpr (PRO) {
            if constexpr (MG3D_PRO_LATE != 0)
                prolong_into(nxt_v[PF == 2 ? (PAR ^ 1) : 0], i + 1, std::integral_constant<int, PAR & 1>{});
            if (stage_new) {
                coarse_put(have_hi + 1, cbuf);
                have_hi++;
            }
        }
        __syncthreads();
    };

    int pl = 0;
    for (; pl + 1 < nsteps; pl += 2) {
#ifdef MG3D_DEBUG_BLOCKTIMES
        if constexpr (MG3D_DEBUG_BT_COND)
            if (threadIdx.x == 0 && blockIdx.x < 1024 && (pl & 31) == 0 && pl < 256)
                g_dbg_bt[pl >> 5][blockIdx.x] = wall_clock64();
#endif
        step(pl, std::integral_constant<int, 0>{});
        step(pl + 1, std::integral_constant<int, 1>{});
    }
    if (pl < nsteps)
        step(pl, std::integral_constant<int, 0>{});

    if (NORM && a.partials) {
#pragma unroll
        for (int off = WAVE / 2; off > 0; off >>= 1)
            acc += __shfl_down(acc, off, WAVE);
        if (lane == 0)
            red[w] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.;
            for (int x = 0; x < NW; x++)
                t += red[x];
            a.partials[blockIdx.x] = t;
        }
    }
}


#endif
