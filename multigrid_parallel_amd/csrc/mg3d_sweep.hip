/*
 * mg3d_sweep.hip -- the fused, temporally blocked red-black Gauss-Seidel kernel.
 *
 * One launch streams a level ONCE through the chip and applies S consecutive colour passes (S = 2*nu:
 * red,black,red,black for the pre-smoother, black,red,... for the post-smoother; mg_3d.h:640-781) and,
 * optionally, the residual of the result (mg_3d.h:794-842: r store and/or the L2 norm), optionally the
 * full-weighting restriction of that residual into the coarse right-hand side (mg_3d.h:961-995; r then
 * never travels to HBM), optionally the trilinear prolongation of the coarse correction into its input
 * (mg_3d.h:1000-1145):      reads v, d once  ->  writes v' (and r or d_coarse) once.
 * The separate-pass formulation moves 6*n*w bytes per RB sweep; this one moves 3*n*w for any number of
 * fused sweeps.
 *
 * Every grid value is bit-identical to the reference: a colour pass only reads the other colour, so the
 * result of a pass does not depend on traversal order, and each update evaluates the reference's
 * expression (mg_3d.h:438-443) with the same association and no FMA contraction.
 *
 * ---- pipeline --------------------------------------------------------------------------------------------
 * A block owns a (j,k) tile and marches along i (planes of NJ x NK points).  "Stage s" (s = 1..ST) is the
 * s-th colour pass (or, for s > S, the residual-only gather); at step p (plane p just loaded) stage s works
 * on plane p-s.  A point of plane q has colour (q+j+k)&1, hence column (j,k) is touched by stage s at step p
 * iff (p+j+k) == c1+1 (mod 2) -- independent of s: at each step exactly one column of every k-pair is
 * active, and it runs ALL stages (on planes p-1 .. p-ST).  Stage s needs from stage s-1:
 *      i-1, i+1 : the thread's own column, two steps ago / this step (registers)
 *      j-1, j+1 : the inactive column of the rows above/below, produced one step ago (registers: a thread
 *                 owns RJ consecutive rows; LDS for the rows of the neighbouring wave)
 *      k-1, k+1 : the pair partner (own register) and the neighbour lane's partner (one DPP move)
 * so the whole temporal window lives in registers; LDS only carries one row per wave edge and stage, and
 * there is ONE barrier per plane.
 *
 * Geometry: a wave covers 64 k-pairs = 128 consecutive k; NW waves are stacked in j, each thread holding RJ
 * rows: tile = (NW*RJ) x 128 points including a halo of H = S (+1 with residual, +2 with restriction)
 * points on every side that is recomputed redundantly.  Output goes to a second array (the halo makes an
 * in-place update racy between tiles).
 *
 * Work distribution: lock-step chunks -- block -> (tile column, i-chunk), tile fastest, so all tile columns of a chunk
 * march through the same planes at the same time and find their neighbours' halo rows in L2 / the Infinity Cache
 * (launch_sweep picks the chunk length).  The experiment it replaced -- one block per CU and equal shares of the linearised
 * (tile column, plane) space -- was 1.3-1.5x slower for want of that sharing (rounds 1-3 kept it behind a switch).
 * Scalar unit: every per-row / per-plane test is wave-uniform (the wave index goes through readfirstlane); they are
 * joined with `&` and selects rather than short-circuits wherever the register budget allows, because a branch per
 * row and stage (78 a step) cost the four-pass launch 9 % at 513^3 and 20 % on the levels below 129^3.
 * k-tiling: tiles start at multiples of 112 columns (128-byte aligned rows of 16 doubles) whenever that needs
 * no more tiles than the tightest packing, so a tile row is exactly eight cache lines.
 */
#include "mg3d_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <type_traits>
#include <mutex>
#include <vector>

#include "mg3d_sweep_kernel.h"

/* -------------------------------------------------------------------- launch */
/* the first-use measurement of chunk lengths blocks the host (hipEventSynchronize) in the middle of an enqueue: fine for
 * one process and its own stream, off by default once a process drives a real multi-rank RCCL communicator -- there every
 * rank's stream also waits for its neighbours, and a host that stops enqueueing is one more thing that has never run on
 * more than one GPU here (option sweep_tune = 1 / 0 overrides either way) */
static std::atomic<int> g_sweep_tune_default{1};
void k_sweep_set_tune_default(int on) { g_sweep_tune_default.store(on); }

static int current_device()
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev;
}

static int device_cus() /* of the CURRENT device (a process may drive several: mg3d_dist_create(device = ...)) */
{
    static std::mutex mu;
    static std::map<int, int> cus;
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(mu);
    auto it = cus.find(dev);
    if (it == cus.end()) {
        hipDeviceProp_t p;
        int n = hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 0;
        if (n <= 0)
            n = 256;
        it = cus.emplace(dev, n).first;
    }
    return it->second;
}

template <int S, int RES, int RJ, int NW, int PF, bool PRO = false, bool RST = true, int DP = 0, int TAP = -1, int C1K = -1>
static int launch_sweep(const mg3d_options &o, SweepArgs &a, int max_partials, hipStream_t s)
{
    using Sh = SweepShape<S, RES>;
    constexpr int VJ = NW * RJ - 2 * Sh::HJ;
    const Geom &g = a.g;
    a.ntj = (g.nj + VJ - 1) / VJ;
    /* k-tiling: tiles of 128 columns starting at multiples of vk = 128 - 2*hk, hk >= HK.  The first and the last
     * tile own their columns up to the global boundary (no halo needed there), so T tiles cover vk*(T-1) + 128
     * columns.  hk = 8 puts every tile on cache-line boundaries; it is taken whenever it costs no extra tile. */
    auto tiles_for = [&](int vk) { return g.nk <= 2 * WAVE ? 1 : (g.nk - 2 * WAVE + vk - 1) / vk + 1; };
    const int vk_tight = 2 * WAVE - 2 * Sh::HK, vk_line = 2 * WAVE - 16;
    a.vk = (Sh::HK <= 8 && tiles_for(vk_line) <= tiles_for(vk_tight)) ? vk_line : vk_tight;
    a.hk = (2 * WAVE - a.vk) / 2;
    a.ntk = tiles_for(a.vk);
    /* i-chunks, lock-step: block -> (tile column, chunk), tile fastest, so that all tile columns of a chunk march
     * through the same planes at the same time (a neighbour's halo rows are then still in the Infinity Cache / L2:
     * measured 1.3-1.5x faster than handing every CU an equal share of unsynchronised work).  Every shape needs > 128
     * VGPRs at 512 threads (one block per CU), so a launch runs in rounds of `ncu` blocks and costs about
     *      max( rounds * (CI + ovh),  blocks * (CI + ovh) / (sat * ncu) )   steps
     * -- the critical path of a CU, or, once about 80 % of the CUs stream, the memory system (513^3, four passes:
     * 2 chunks = 220 blocks, one round: 0.71 ms; 3 chunks = 330 blocks, two rounds: 0.95 ms; 4 chunks: 0.77 ms).
     * ovh = warm-up planes + pipeline drain per chunk.  Ties go to fewer chunks (fewer warm-up planes to read). */
    const int nout = a.i_hi - a.i_lo;
    const int ncu = device_cus();
    const long long T = (long long)a.ntj * a.ntk;
    if (a.edge > 0) { /* the two end windows of the range as two chunks of one launch (see SweepArgs::edge) */
        if (2 * a.edge > nout || (a.partials && 2 * T > max_partials))
            return -1;
        a.CI = a.edge;
        a.xcd_remap = 2 * T < 64 ? 0 : 2 * T <= ncu ? 1 : 2;
        hipLaunchKernelGGL((sweep_kernel<S, RES, RJ, NW, PF, PRO, RST, DP, TAP, C1K>), dim3((unsigned)(2 * T)), dim3(NW * WAVE), 0, s, a);
        return (int)(2 * T);
    }
    const int ovh = Sh::HI + Sh::ST + (RES == 2 ? 2 : 0) + 1;
    auto model_cost = [&](int ci) -> double { /* in steps; < 0: not launchable */
        const long long blocks = T * ((nout + ci - 1) / ci);
        if (a.partials && blocks > max_partials)
            return -1.;
        const double steps = (double)(ci + ovh);
        /* the four-pass launch saturates the memory system with ~80 % of the CUs streaming; the shorter pipelines
         * (two passes + residual, residual + restriction, prolongation + two passes) spend more of a step computing
         * and keep scaling to all of them */
        const double sat = S >= 4 && (RES == 0 || RES == 3) ? 0.8 : 1.0;
        const double crit = (double)((blocks + ncu - 1) / ncu) * steps, bw = (double)blocks * steps / (sat * ncu);
        return crit > bw ? crit : bw;
    };
    int best_ci = nout;
    double best_cost = 1e30;
    for (int nci = 1; nci <= 64 && nci <= nout; nci++) {
        const int ci = (nout + nci - 1) / nci;
        if (nci > 1 && ci < 2)
            break;
        const double cost = model_cost(ci);
        if (cost < 0)
            break;
        if (cost < best_cost * 0.97) {
            best_cost = cost;
            best_ci = ci;
        }
    }
    /* A short tail chunk: where a whole number of chunk layers leaves CUs idle (513^3, four passes: 110 tile columns,
     * two layers = 220 of 256 CUs), m full layers of x planes and a tail chunk that the left-over CUs work off in
     * ceil(T / left) turns, sized so that both finish together (231 + 231 + 51 planes: 240 steps instead of 266;
     * 0.675 against 0.700 ms).  It pays at 385^3 and 513^3 and costs 10-25 % at 257^3, 449^3 and 641^3 -- no rule in
     * T, the left-over CUs or the turns separates the two, so it is one more candidate for the measurement below and
     * the model's choice only for the size it was found on. */
    int tail_ci = 0;
    if (T < ncu && T * 8 > ncu) {
        const int m = ncu / (int)T, left = ncu - m * (int)T;
        if (left * 8 >= T) {
            const int turns = ((int)T + left - 1) / left;
            const int x = (turns * (nout + ovh) - ovh + m * turns) / (1 + m * turns);
            if (x > 0 && nout - m * x > 0 && (double)(x + ovh) < 0.97 * best_cost && model_cost(x) >= 0)
                tail_ci = x;
        }
    }
    a.CI = best_ci;
    /* (the tail chunk is the model's choice for the four-pass shapes at four turns -- the size it was found on -- and a
     * candidate of the measurement below everywhere else) */
    if (tail_ci && S >= 4 && (RES == 0 || RES == 3) && ((int)T + (ncu % (int)T) - 1) / (ncu % (int)T) >= 4)
        a.CI = tail_ci;
    bool forced = false;
    if (o.v[MG3D_OPT_SWEEP_CI] > 0) { /* a fixed chunk length (measurement only) */
        a.CI = o.v[MG3D_OPT_SWEEP_CI] < nout ? o.v[MG3D_OPT_SWEEP_CI] : nout;
        forced = true;
    }
    long long nb = 0;
    /* XCD grouping: the blocks of one XCD group (blockIdx % 8) take a contiguous run of tile columns, so that
     * neighbouring tile columns mostly share an L2 (a tenth to a quarter fewer bytes from the fabric).  One round of
     * blocks: renumber the whole grid (1); several rounds: inside every chunk's layer (2) -- renumbering the whole
     * grid would scatter the first round over all chunks and break the lock-step. */
    auto set_ci = [&](int ci) {
        a.CI = ci;
        nb = T * ((nout + ci - 1) / ci);
        /* (measured in isolation, grouping 0 sometimes wins by 2 %; inside the cycle it then loses 5 %: not tuned) */
        a.xcd_remap = nb < 64 ? 0 : nb <= ncu ? 1 : 2;
    };
    auto launch = [&]() {
        static_assert(C1K < 0 || !PRO || C1K == 0, "PRO launches are post-smoothers");
        if (C1K >= 0 && a.c1 != C1K)
            return;
        hipLaunchKernelGGL((sweep_kernel<S, RES, RJ, NW, PF, PRO, RST, DP, TAP, C1K>), dim3((unsigned)nb), dim3(NW * WAVE), 0, s, a);
    };
    /* Measured choice.  The model above ranks chunk lengths by steps; what a step costs depends on how many CUs stream
     * at once and on how well the chunks keep in lock-step, which it does not know.  The first launch of a shape on a
     * level geometry therefore times the model's best few candidates (the launch is idempotent: it reads u, d, writes
     * the other buffer; every chunking gives the same bits) and the fastest is remembered for the process.
     * Option sweep_tune = 0 keeps the model's choice. */
    const bool tune_on = o.v[MG3D_OPT_SWEEP_TUNE] >= 0 ? o.v[MG3D_OPT_SWEEP_TUNE] != 0 : g_sweep_tune_default.load() != 0;
    /* not for the launches that form the norm: its value depends (in the last bits) on how the points are grouped
     * into per-block partial sums, and a timing-dependent choice would make it differ from run to run */
    if (!forced && tune_on && !a.partials && T * nout >= 1024) {
        struct Key {
            int v[13];
            bool operator<(const Key &o) const { return memcmp(v, o.v, sizeof v) < 0; }
        };
        const Key key = {{g.ni, g.nj, g.nk, g.N, g.ig0 & 1, a.i_lo, a.i_hi, a.vin != nullptr, a.partials != nullptr,
                          a.r != nullptr, a.vk, max_partials, current_device()}};
        static std::mutex mu;
        static std::map<Key, int> tuned; /* chunk length */
        std::lock_guard<std::mutex> lock(mu);
        auto it = tuned.find(key);
        if (it == tuned.end()) {
            std::vector<int> cand;
            auto add = [&](int ci) {
                if (ci >= 1 && ci <= nout && model_cost(ci) >= 0 && std::find(cand.begin(), cand.end(), ci) == cand.end())
                    cand.push_back(ci);
            };
            add(a.CI);
            add(best_ci);
            add(tail_ci);
            for (int nci = 1; nci <= 64 && nci <= nout && cand.size() < 10; nci++) {
                const int ci = (nout + nci - 1) / nci;
                if (nci > 1 && ci < 2)
                    break;
                const double c = model_cost(ci);
                if (c >= 0 && c <= 1.35 * best_cost)
                    add(ci);
            }
            const int model_ci = a.CI;
            int pick = model_ci;
            hipEvent_t e0, e1;
            if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
                auto timed = [&]() { /* best of two after one launch that also warms the instruction cache */
                    float ms_min = 1e30f;
                    for (int rep = 0; rep < 3; rep++) {
                        float ms = 0.f;
                        (void)hipEventRecord(e0, s);
                        launch();
                        (void)hipEventRecord(e1, s);
                        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
                            ms = 1e30f;
                        if (rep > 0 && ms < ms_min)
                            ms_min = ms;
                    }
                    return ms_min;
                };
                float best_ms = 1e30f;
                for (int ci : cand) {
                    set_ci(ci);
                    const float ms = timed();
                    if (ms < best_ms) {
                        best_ms = ms;
                        pick = ci;
                    }
                }
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
            }
            it = tuned.emplace(key, pick).first;
            if (o.v[MG3D_OPT_SWEEP_TUNE_LOG])
                fprintf(stderr, "mg3d sweep<%d,%d,%d,%d,%d,%d> %dx%dx%d planes [%d,%d): %zu candidates, chunk %d (model %d)\n",
                        S, RES, RJ, NW, PF, (int)PRO, g.ni, g.nj, g.nk, a.i_lo, a.i_hi, cand.size(), pick, model_ci);
        }
        set_ci(it->second);
    } else {
        set_ci(a.CI);
    }
    if (a.partials && nb > max_partials)
        return -1;
    launch();
    return (int)nb;
}

/* tile shape per (S, residual): rows per thread RJ, waves NW, prefetch depth PF.  The defaults are the
 * measured best on MI355X; the options sweep_rj / sweep_nw / sweep_pf select another compiled shape. */
struct SweepCfg {
    int rj, nw, pf;
};

static SweepCfg opt_cfg(const mg3d_options &o, SweepCfg dflt)
{
    if (o.v[MG3D_OPT_SWEEP_RJ] > 0 && o.v[MG3D_OPT_SWEEP_NW] > 0)
        return SweepCfg{o.v[MG3D_OPT_SWEEP_RJ], o.v[MG3D_OPT_SWEEP_NW], o.v[MG3D_OPT_SWEEP_PF] > 0 ? o.v[MG3D_OPT_SWEEP_PF] : dflt.pf};
    return dflt;
}

/* the first shape listed is the default; a request that names no compiled shape falls back to it */
#define TRY(S_, RES_, RJ_, NW_, PF_)                                              \
    if (c.rj == RJ_ && c.nw == NW_ && c.pf == PF_)                                \
        return launch_sweep<S_, RES_, RJ_, NW_, PF_>(o, a, max_partials, s);
#define DFLT(S_, RES_, RJ_, NW_, PF_) return launch_sweep<S_, RES_, RJ_, NW_, PF_>(o, a, max_partials, s);

template <int S, int RES> static int dispatch(const mg3d_options &o, SweepArgs &a, SweepCfg c, int max_partials, hipStream_t s);

template <> int dispatch<4, 1>(const mg3d_options &o, SweepArgs &a, SweepCfg c, int max_partials, hipStream_t s)
{
    TRY(4, 1, 6, 4, 1) TRY(4, 1, 6, 4, 2) TRY(4, 1, 4, 4, 2) TRY(4, 1, 4, 8, 1) TRY(4, 1, 2, 8, 2)
    DFLT(4, 1, 6, 4, 2)
}
template <> int dispatch<4, 0>(const mg3d_options &o, SweepArgs &a, SweepCfg c, int max_partials, hipStream_t s)
{
    TRY(4, 0, 8, 4, 1) TRY(4, 0, 6, 4, 2) TRY(4, 0, 6, 4, 3) TRY(4, 0, 4, 8, 1) TRY(4, 0, 4, 8, 2)
    TRY(4, 0, 2, 8, 2) TRY(4, 0, 2, 8, 4) TRY(4, 0, 2, 16, 1)
    DFLT(4, 0, 4, 8, 1)
}
template <> int dispatch<2, 1>(const mg3d_options &o, SweepArgs &a, SweepCfg c, int max_partials, hipStream_t s)
{
    TRY(2, 1, 4, 8, 1) TRY(2, 1, 8, 4, 2) TRY(2, 1, 6, 4, 2) TRY(2, 1, 2, 16, 1) TRY(2, 1, 6, 8, 1)
    DFLT(2, 1, 4, 8, 1)
}
template <> int dispatch<2, 0>(const mg3d_options &o, SweepArgs &a, SweepCfg c, int max_partials, hipStream_t s)
{
    TRY(2, 0, 6, 8, 1) TRY(2, 0, 8, 4, 2) TRY(2, 0, 4, 8, 1) TRY(2, 0, 4, 8, 2) TRY(2, 0, 4, 8, 3)
    TRY(2, 0, 2, 8, 4) TRY(2, 0, 2, 16, 1)
    DFLT(2, 0, 4, 8, 1)
}
template <> int dispatch<0, 1>(const mg3d_options &o, SweepArgs &a, SweepCfg c, int max_partials, hipStream_t s)
{
    TRY(0, 1, 4, 8, 1) TRY(0, 1, 8, 4, 2) TRY(0, 1, 4, 8, 2) TRY(0, 1, 4, 8, 3) TRY(0, 1, 2, 8, 4) TRY(0, 1, 2, 16, 1)
    DFLT(0, 1, 4, 8, 1)
}

/* S colour passes starting with colour c1, optional residual.  Returns the number of
 * partial sums written (0 when no norm was requested), -1 if the shape is unsupported. */
template <> int dispatch<0, 2>(const mg3d_options &o, SweepArgs &a, SweepCfg c, int max_partials, hipStream_t s)
{
    TRY(0, 2, 4, 8, 1) TRY(0, 2, 4, 8, 2) TRY(0, 2, 2, 8, 2) TRY(0, 2, 2, 8, 4) TRY(0, 2, 2, 16, 1) TRY(0, 2, 6, 8, 1)
    DFLT(0, 2, 4, 8, 1)
}
template <> int dispatch<4, 2>(const mg3d_options &o, SweepArgs &a, SweepCfg c, int max_partials, hipStream_t s)
{
    /* the whole down-leg of a level in ONE launch.  The six-stage window only fits two rows per thread (16-row tiles
     * with a halo of 6: four owned rows): a quarter of the rows it computes are kept -- for the levels where a launch
     * is paid in pipeline steps and microseconds of launch latency, not in bytes */
    TRY(4, 2, 2, 8, 1) TRY(4, 2, 2, 8, 2)
    DFLT(4, 2, 2, 8, 2)
}
template <> int dispatch<2, 2>(const mg3d_options &o, SweepArgs &a, SweepCfg c, int max_partials, hipStream_t s)
{
    TRY(2, 2, 4, 8, 1)
    DFLT(2, 2, 4, 8, 1)
}
template <> int dispatch<1, 2>(const mg3d_options &o, SweepArgs &a, SweepCfg c, int max_partials, hipStream_t s)
{
    /* the LAST pre-smoothing pass + residual + restriction: the down-leg's only launch on the top level of a cycle whose
     * first three pre-smoothing passes rode on the previous cycle's last launch (k_sweep_tap) */
    TRY(1, 2, 4, 8, 1) TRY(1, 2, 4, 8, 2)
    DFLT(1, 2, 4, 8, 1)
}
template <> int dispatch<4, 3>(const mg3d_options &o, SweepArgs &a, SweepCfg c, int max_partials, hipStream_t s)
{
    TRY(4, 3, 4, 8, 1)
    DFLT(4, 3, 4, 8, 1)
}

bool k_sweep_fuse_rst2(const mg3d_options &o, int N) /* two passes + residual + restriction as ONE launch on a level of N points per side? */
{
    /* Since round 3 the shape has no scratch (the restriction overwrites its r pairs in place and parks them in LDS: 243
     * VGPRs; round 2: 156 bytes of scratch, 1.49 ms at 513^3 against 0.63 + 0.57 ms as two launches) and one launch
     * moves 3.4 GB instead of 5.5: V(1,1) at 513^3 2.70 -> 2.28 ms per cycle, 257^3 0.491 -> 0.460; at 129^3 and below
     * (0.180 -> 0.195 ms) the split launches' two-row shapes stay ahead.  Option fuse_rst2 = 1 / 0: always / never. */
    if (o.v[MG3D_OPT_FUSE_RST2] == 0 || o.v[MG3D_OPT_FUSE_RST2] == 1)
        return o.v[MG3D_OPT_FUSE_RST2] == 1;
    return N >= 130;
}

static int sweep_impl(const mg3d_options &o, const Geom &g, const double *vin, const double *d, double *vout, double *r, double *partials,
                      int max_partials, double h, int S, int c1, bool residual, hipStream_t s, int acc_lo, int acc_hi,
                      const Geom *gc, double *dc, int ic_lo, int ic_hi, const Geom *gce, const double *ec, int i_lo, int i_hi,
                      bool tap, int edge = 0)
{
    SweepArgs a;
    a.edge = edge;
    a.i_lo = i_lo >= 0 ? i_lo : 0;
    a.i_hi = i_hi >= 0 ? i_hi : g.ni;
    if (a.i_hi <= a.i_lo)
        return 0;
    a.ec = ec;
    a.gce = gce ? *gce : g;
    a.dc = dc;
    if (dc) {
        a.gc = *gc;
        a.ic_lo = ic_lo >= 0 ? ic_lo : 0;
        a.ic_hi = ic_hi >= 0 ? ic_hi : gc->ni;
    } else {
        a.gc = g;
        a.ic_lo = a.ic_hi = 0;
    }
    a.acc_lo = acc_lo;
    a.acc_hi = acc_hi < 0 ? g.ni : acc_hi;
    a.g = g;
    a.vin = vin;
    a.d = d;
    a.vout = vout;
    /* the restricting shapes (RES == 2) produce neither a stored r nor the norm: their callers never ask for either,
     * and without that code the residual + restriction launch has 60 fewer scalar instructions and 10 fewer branches a
     * step */
    if (dc && (r || partials))
        return -1;
    a.r = r;
    a.partials = partials;
    a.hSq = h * h;           /* mg_3d.h:644 */
    a.sixth = 1. / 6;        /* mg_3d.h:646 */
    a.invHsq = 1. / (h * h); /* mg_3d.h:797 */
    a.c1 = c1;
    if (tap) { /* four passes, the residual norm of the state after the second one into partials */
        if (S != 4 || dc || ec || r || !partials || residual)
            return -1;
        return dispatch<4, 3>(o, a, opt_cfg(o, {4, 8, 1}), max_partials, s);
    }
    if (ec) { /* prolongation fused into the input: 4- and 2-pass smoothing launches */
        if (dc || residual || (g.nj & 1) == 0 || c1 != 0) /* (c1: the kernel derives the plane parity from it, see PRO) */
            return -1;
        if (S == 4) /* small levels: two rows per thread (no spills, 8 owned rows of 16); else the opt-in four-row shape */
            return g.N <= o.v[MG3D_OPT_SMALL_MAX] ? launch_sweep<4, 0, 2, 8, 2, true>(o, a, max_partials, s)
                                              : launch_sweep<4, 0, 4, 8, 1, true>(o, a, max_partials, s);
        if (S == 2)
            return launch_sweep<2, 0, 4, 8, 1, true>(o, a, max_partials, s);
        return -1;
    }
    /* Levels of at most 65^3 points: a step costs a global-load latency (~1.4 us with one plane in flight), not
     * bandwidth, and only a few tiles exist anyway -- two rows per thread leave the registers for two planes in flight
     * (33^3: four passes 20 -> 13 us, residual + restriction 15 -> 12 us; at 257^3 the same shapes are 40 % slower). */
    const bool small = g.N <= o.v[MG3D_OPT_SMALL_MAX];
    if (dc && S == 0 && residual)
        return dispatch<0, 2>(o, a, opt_cfg(o, small ? SweepCfg{2, 8, 2} : SweepCfg{4, 8, 1}), max_partials, s);
    if (dc && S == 2 && residual)
        return dispatch<2, 2>(o, a, opt_cfg(o, {4, 8, 1}), max_partials, s);
    if (dc && S == 1 && residual)
        return dispatch<1, 2>(o, a, opt_cfg(o, {4, 8, 1}), max_partials, s);
    if (dc && S == 4 && residual)
        return dispatch<4, 2>(o, a, opt_cfg(o, {2, 8, 2}), max_partials, s);
    if (dc)
        return -1;
    if (S == 4 && residual)
        return dispatch<4, 1>(o, a, opt_cfg(o, {6, 4, 2}), max_partials, s);
    if (S == 4 && !residual) {
        /* from the zero guess on the levels between 66 and 300 points per side (129^3, 257^3: a step is paid in latency, the
         * launch has a CU per block and little else): two rows a thread, SIXTEEN waves -- the same 32-row tile at four waves
         * per SIMD (128 VGPRs, no scratch): 257^3 80 -> 71-74 us, 129^3 25 -> 22 us; 65^3 and below 11.5 -> 14 (the eight-wave
         * two-row shape stays), 513^3 425 -> 420 (the four-row shape stays); kernel trace, round 4 */
        const bool mid = vin == nullptr && g.N > 65 && g.N <= 300;
        return dispatch<4, 0>(o, a, opt_cfg(o, mid ? SweepCfg{2, 16, 1} : small ? SweepCfg{2, 8, 2} : SweepCfg{4, 8, 1}), max_partials, s);
    }
    if (S == 2 && residual) {
        /* the norm alone (the top level's second post-smoothing launch): a shape without the code that assembles r */
        const SweepCfg c = opt_cfg(o, {4, 8, 1});
        if (!r && c.rj == 4 && c.nw == 8 && (c.pf == 1 || c.pf == 2))
            return c.pf == 1 ? launch_sweep<2, 1, 4, 8, 1, false, false>(o, a, max_partials, s)
                             : launch_sweep<2, 1, 4, 8, 2, false, false>(o, a, max_partials, s);
        return dispatch<2, 1>(o, a, c, max_partials, s);
    }
    if (S == 2 && !residual)
        return dispatch<2, 0>(o, a, opt_cfg(o, small ? SweepCfg{2, 8, 4} : SweepCfg{4, 8, 1}), max_partials, s);
    if (S == 0 && residual)
        return dispatch<0, 1>(o, a, opt_cfg(o, {4, 8, 1}), max_partials, s);
    return -1;
}


/* ---------------------------------------------------------------------------------------------- one launch per leg
 * The legs of a V(2,2) cycle on a level as ONE launch each (mg3d_ctx.hip, "two launches per level"): the level streams
 * through the chip twice per cycle instead of three or four times.  The windows are five planes deep: they fit two waves
 * per SIMD (four rows per thread, eight waves) since the prolongation is applied at the end of the step before (MG3D_PRO_LATE),
 * the wave-edge rows keep one LDS copy (EXS) and two slots of the down-leg's d window are parked in LDS (DP); the first version
 * needed one wave per SIMD (eight rows, the 512-register budget) and was bound by instruction issue (MG3D_LEG_*_RJ = 8).
 *   down: S = 3 colour passes, black first -- the cycle's first red pass (mg_3d.h:657) is the identity behind the previous
 *         cycle's last red pass --, the residual (:1294) and its full-weighting restriction into the interior of the
 *         coarse right-hand side (:1310).  (S = 4, the leg of a cycle with none in front of it, needs a six-plane window:
 *         124 bytes of scratch at two waves per SIMD, 2.1 ms at one -- against 0.66 + 0.49 as two launches: not instantiated.)  partials != NULL: the sum of
 *         diff^2 of the INCOMING state over the colour the first pass updates -- the second half of the previous cycle's
 *         residual norm (:1354), see k_sweep_leg_up.
 *   up:   prolongation (:1331) folded into the loads, four post-smoothing passes black, red, black, red (:1341).
 *         partials != NULL: the sum of diff^2 of the RESULT over the colour the last pass has updated (red) -- the first
 *         half of the cycle's residual norm; the other half is formed by the next down-leg (or by a norm-only launch). */
/* the up-leg's shape: four rows a thread, eight waves (two waves per SIMD, two slots of the d window parked in LDS: 249 VGPRs,
 * no scratch since the prolongation is applied at the end of the previous step, MG3D_PRO_LATE) or eight rows, four waves (one
 * wave per SIMD, three slots parked: the shape of round 4's first version, bound by instruction issue) */
#ifndef MG3D_LEG_UP_RJ
#define MG3D_LEG_UP_RJ 4
#endif
#ifndef MG3D_LEG_DP_UP
#define MG3D_LEG_DP_UP (MG3D_LEG_UP_RJ == 4 ? 0 : 3) /* four rows: 253 VGPRs with nothing parked; same-box A/B 0.716 (0) / 0.740 (1) / 0.729-0.743 ms (2) */
#endif
/* the down-leg's shape: four rows a thread, eight waves (two waves per SIMD: two slots of the d window parked in LDS, which the
 * single copy of the wave-edge rows -- EXS in the kernel -- makes room for) or eight rows, four waves (one wave per SIMD) */
#ifndef MG3D_LEG_DOWN_RJ
#define MG3D_LEG_DOWN_RJ 4
#endif
#ifndef MG3D_LEG_DP_DOWN3
#define MG3D_LEG_DP_DOWN3 2
#endif

/* the leg launches take the chunk model's choice, never the first-use measurement: their norm-forming variants cannot be measured
 * (reproducible partial sums), and the variants without a norm half -- the first cycle of a call behind an earlier one, the last
 * cycle's up-leg -- would otherwise spend ~30 launches of measurement on a geometry whose model choice is the measured best
 * (MG3D_SWEEP_CI = 247 / 200 / 171 / 129 all slower, profiles/r04_one_launch_per_leg.txt) */
static mg3d_options leg_options(const mg3d_options &o)
{
    mg3d_options q = o;
    q.v[MG3D_OPT_SWEEP_TUNE] = 0;
    return q;
}

static void leg_args(SweepArgs &a, const Geom &g, const double *vin, const double *d, double *vout, double *partials, double h,
                     int c1, int i_lo, int i_hi, int acc_lo, int acc_hi)
{
    a.g = g;
    a.vin = vin;
    a.d = d;
    a.vout = vout;
    a.r = nullptr;
    a.partials = partials;
    a.hSq = h * h;           /* mg_3d.h:644 */
    a.sixth = 1. / 6;        /* mg_3d.h:646 */
    a.invHsq = 1. / (h * h); /* mg_3d.h:797 */
    a.c1 = c1;
    a.i_lo = i_lo >= 0 ? i_lo : 0;
    a.i_hi = i_hi >= 0 ? i_hi : g.ni;
    a.acc_lo = acc_lo;
    a.acc_hi = acc_hi < 0 ? g.ni : acc_hi;
    a.ec = nullptr;
    a.gce = g;
    a.gc = g;
    a.dc = nullptr;
    a.ic_lo = a.ic_hi = 0;
    a.edge = 0;
}

int k_sweep_leg_down(const mg3d_options &o, const Geom &g, const double *vin, const double *d, double *vout, const Geom &gc, double *dc, double h, int S,
                     double *partials, int max_partials, hipStream_t s, int acc_lo, int acc_hi, int ic_lo, int ic_hi, int i_lo,
                     int i_hi)
{
    const mg3d_options oq = leg_options(o);
    SweepArgs a;
    leg_args(a, g, vin, d, vout, partials, h, S == 4 ? 1 : 0, i_lo, i_hi, acc_lo, acc_hi);
    if (a.i_hi <= a.i_lo)
        return 0;
    a.gc = gc;
    a.dc = dc;
    a.ic_lo = ic_lo >= 0 ? ic_lo : 0;
    a.ic_hi = ic_hi >= 0 ? ic_hi : gc.ni;
#if MG3D_LEG_DOWN_RJ == 4
    if (S == 3 && partials)
        return launch_sweep<3, 2, 4, 8, 1, false, true, MG3D_LEG_DP_DOWN3, 0>(oq, a, max_partials, s);
    if (S == 3)
        return launch_sweep<3, 2, 4, 8, 1, false, true, MG3D_LEG_DP_DOWN3, -1>(oq, a, max_partials, s);
#else
    if (S == 3 && partials)
        return launch_sweep<3, 2, 8, 4, 1, false, true, MG3D_LEG_DP_DOWN3, 0>(oq, a, max_partials, s);
    if (S == 3)
        return launch_sweep<3, 2, 8, 4, 1, false, true, MG3D_LEG_DP_DOWN3, -1>(oq, a, max_partials, s);
#endif
    return -1;
}

int k_sweep_leg_up(const mg3d_options &o, const Geom &g, const double *vin, const double *d, double *vout, const Geom &gce, const double *ec, double h,
                   double *partials, int max_partials, hipStream_t s, int acc_lo, int acc_hi, int i_lo, int i_hi, int edge)
{
    if ((g.nj & 1) == 0)
        return -1;
    const mg3d_options oq = leg_options(o);
    SweepArgs a;
    leg_args(a, g, vin, d, vout, partials, h, 0, i_lo, i_hi, acc_lo, acc_hi);
    if (a.i_hi <= a.i_lo)
        return 0;
    a.ec = ec;
    a.gce = gce;
    a.edge = edge;
#if MG3D_LEG_UP_RJ == 4
    if (partials)
        return launch_sweep<4, 0, 4, 8, 1, true, true, MG3D_LEG_DP_UP, 4>(oq, a, max_partials, s);
    return launch_sweep<4, 0, 4, 8, 1, true, true, MG3D_LEG_DP_UP, -1>(oq, a, max_partials, s);
#else
    if (partials)
        return launch_sweep<4, 0, 8, 4, 1, true, true, MG3D_LEG_DP_UP, 4>(oq, a, max_partials, s);
    return launch_sweep<4, 0, 8, 4, 1, true, true, MG3D_LEG_DP_UP, -1>(oq, a, max_partials, s);
#endif
}

int k_sweep(const mg3d_options &o, const Geom &g, const double *vin, const double *d, double *vout, double *r, double *partials,
            int max_partials, double h, int S, int c1, bool residual, hipStream_t s, int acc_lo, int acc_hi,
            const Geom *gc, double *dc, int ic_lo, int ic_hi, const Geom *gce, const double *ec, int i_lo, int i_hi, int edge)
{
    return sweep_impl(o, g, vin, d, vout, r, partials, max_partials, h, S, c1, residual, s, acc_lo, acc_hi, gc, dc, ic_lo, ic_hi,
                      gce, ec, i_lo, i_hi, false, edge);
}

int k_sweep_tap(const mg3d_options &o, const Geom &g, const double *vin, const double *d, double *vout, double *partials, int max_partials,
                double h, int c1, hipStream_t s, int acc_lo, int acc_hi, int i_lo, int i_hi, int edge)
{
    return sweep_impl(o, g, vin, d, vout, nullptr, partials, max_partials, h, 4, c1, false, s, acc_lo, acc_hi, nullptr, nullptr,
                      -1, -1, nullptr, nullptr, i_lo, i_hi, true, edge);
}

#ifdef MG3D_DEBUG_BLOCKTIMES
extern "C" int mg3d_debug_blocktimes(unsigned long long *out) /* 8 x 1024 wall-clock samples (100 MHz ticks) of the last launch of the chosen shape */
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_bt), sizeof(unsigned long long) * 8 * 1024) == hipSuccess ? 0 : -1;
}
#endif
