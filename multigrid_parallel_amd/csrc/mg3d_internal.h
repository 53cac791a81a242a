/* mg3d_internal.h -- shared between mg3d_ctx.hip and mg3d_kernels.hip (not installed). */
#ifndef MG3D_INTERNAL_H
#define MG3D_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>

#include "mg3d.h"

/* Geometry of one (local) level as the kernels see it.
 * Device layout: idx = plane*i + pitch*j + k, pitch % 16 == 0 (128-byte rows),
 * plane = pitch*nj.  A single-GPU level has ni = nj = nk = N and ig0 = 0; an
 * i-slab of a distributed level has ni = owned planes + 2 (halo or physical
 * boundary on either side) and ig0 = global index of its local plane 0. */
struct Geom {
    int ni, nj, nk;
    int pitch;
    long long plane;
    int ig0; /* global i of local plane 0 (colour parity, grid-transfer alignment) */
    int N;   /* global points per side */
};

static inline int mg3d_pitch_for(int nk) { return (nk + 15) & ~15; }

struct LuBand {
    int n;        /* unknowns */
    int bw;       /* half bandwidth actually populated (max |i-j| with LU[i][j] != 0) */
    double *lcol; /* [n][bw]  lcol[j*bw+t] = LU[j+1+t][j]   (strictly lower, by column) */
    double *ucol; /* [n][bw]  ucol[j*bw+t] = LU[j-1-t][j]   (strictly upper, by column) */
    int npad;     /* n rounded up to a multiple of 64 (identity rows appended) */
    double *diag; /* [2*npad]: the diagonal, then RN(1/diagonal); 1 on the padding */
    int fast_div; /* every diagonal entry lies in lu_div()'s safe window: the reciprocals may be used */
    /* lane-rotated copies for the single-wave solve (bw <= 64*rot_r): entry [j][64*q + l] is the
     * factor of the row lane l accumulates at step j: forward row j+1+((l-j-1)&63)+64q, backward row
     * j-1-((j-1-l)&63)-64q; zero outside the band / matrix */
    int rot_r;    /* 0 when not built */
    double *lrot; /* [n][64*rot_r] */
    double *urot; /* [n][64*rot_r] */
    /* the same factors as ONE stream in the order the streamed solve consumes them: 2*npad/64 chunks of 64
     * steps (forward steps j = 0..npad-1, then backward steps j = npad-1..0) plus two spare chunks the
     * loaders may over-read; a step is 64 lanes x rot_r doubles, [lane][q] */
    int stream_ch; /* 64 when built, else 0 */
    double *stream;
    /* full factor only, when a reduced factor (the system without its identity rows) exists beside it: index of
     * unknown p in the reduced system, -1 for an identity row (mg3d_ctx.hip, install_lu) */
    int *in_map;
};

#define MG3D_MAX_PARTIALS 32768

/* Launch and schedule policy of ONE context: defaults, then the environment as an override read once when the context
 * is created (mg3d_options_init), afterwards only mg3d_ctx_set_option / mg3d_dist_set_option / mg3d32_set_option.
 * Nothing on a launch path reads the environment; two contexts of one process may differ.  Keys and environment names:
 * kOptionTable in mg3d_ctx.hip, the table in INTEGRATION.md. */
enum {
    MG3D_OPT_CARRY = 0,    /* consecutive V(2,2) cycles share a launch on the finest level ("carried cycles") */
    MG3D_OPT_CARRY_MIN,    /* ... from this many points per side (130) */
    MG3D_OPT_LEGS,         /* one launch per leg on the finest level instead (round 4; 1) */
    MG3D_OPT_LEGS_MIN,     /* ... from this many points per side (160: at 129^3 the carried cycles are 6 % faster, from 193^3 the legs win 3 - 24 %) */
    MG3D_OPT_TINY,         /* the level above the coarsest one in one workgroup */
    MG3D_OPT_TINY_CYCLE,   /* ... together with the direct solve in ONE launch */
    MG3D_OPT_LU_REDUCED,   /* install the factor without its identity rows beside the full one */
    MG3D_OPT_FUSE_RST2,    /* two passes + residual + restriction as one launch: -1 from 130 points per side, 0 never, 1 always */
    MG3D_OPT_SMALL_MAX,    /* largest level side that runs the two-rows-per-thread shapes (129) */
    MG3D_OPT_FUSE_LEG_MAX, /* largest level side whose legs run as one (two-row) launch each (0) */
    MG3D_OPT_FUSE_UP_MAX,  /* largest level side whose up-leg folds the prolongation into a four-pass launch (0) */
    MG3D_OPT_SWEEP_TUNE,   /* first-use measurement of chunk lengths: -1 on unless a multi-rank job, 0 off, 1 on */
    MG3D_OPT_SWEEP_TUNE_LOG,
    MG3D_OPT_SWEEP_CI,     /* > 0: planes per chunk of every fused sweep launch (measurement only) */
    MG3D_OPT_SWEEP_RJ,     /* > 0: another compiled tile shape (rows per thread, waves, planes in flight); unknown ones */
    MG3D_OPT_SWEEP_NW,     /*      fall back to the default */
    MG3D_OPT_SWEEP_PF,
    MG3D_OPT_COUNT
};
struct mg3d_options {
    int v[MG3D_OPT_COUNT];
};
void mg3d_options_init(mg3d_options *o);          /* defaults + environment */
int mg3d_option_index(const char *key);           /* -1: no such key */
const char *mg3d_option_key(int index);           /* NULL past the end */

/* launchers (mg3d_kernels.hip); all asynchronous on `s` */
void k_smooth_color(const Geom &g, double *v, const double *d, double hSq, int color, hipStream_t s);
/* writes partials (one per block) then reduces them, in a fixed order, into *sumsq_out */
void k_residual(const Geom &g, const double *v, const double *d, double invHsq, double *res, double *partials,
                double *sumsq_out, hipStream_t s);
void k_sumsq(const Geom &g, const double *a, double *partials, double *sumsq_out, hipStream_t s);
/* ic_lo/ic_hi, if_lo/if_hi: local plane range to produce; -1 = every local plane that is not a slab halo */
void k_restrict(const Geom &gf, const double *r, const Geom &gc, double *dc, hipStream_t s, int ic_lo = -1,
                int ic_hi = -1, bool faces_only = false /* injection on the coarse faces only */);
void k_prolong(const Geom &gc, const double *ec, const Geom &gf, double *ef, hipStream_t s, int if_lo = -1,
               int if_hi = -1);
/* BCFunc(i*h, j*h, k*h) = x*x - 2*y*y + z*z on the six faces of a field (mg_3d.h:89-90, 1147-1239) */
void k_fill_boundary(const Geom &g, double *v, double h, hipStream_t s);
/* folds np per-block partial sums, in a fixed order, into *out */
void k_fold(const double *partials, int np, double *out, hipStream_t s);
/* the same over two runs of partial sums (a norm whose halves two launches formed): *out = fold(pa) + fold(pb) */
void k_fold2(const double *pa, int na, const double *pb, int nb, double *out, hipStream_t s);
/* fused sweep (mg3d_sweep.hip): S colour passes starting with colour c1 (1 red, 0 black) from vin into
 * vout (vout != vin; ignored when S == 0), then optionally the residual of the result: r (may be NULL)
 * receives it on the interior, partials (may be NULL) one sum of diff^2 per block.  Returns the number
 * of partials written (>= 0) or -1 when the (S, residual) shape has no instantiation. */
void k_sweep_set_tune_default(int on); /* first-use chunk measurement on / off where the option says -1 */
bool k_sweep_fuse_rst2(const mg3d_options &o, int N); /* two passes + residual + restriction as ONE launch on a level of N points per side */
int k_sweep(const mg3d_options &o, const Geom &g, const double *vin, const double *d, double *vout, double *r, double *partials,
            int max_partials, double h, int S, int c1, bool residual, hipStream_t s, int acc_lo = 0,
            int acc_hi = -1 /* local planes entering the norm; default all */,
            const Geom *gc = nullptr, double *dc = nullptr /* non-NULL: also restrict the residual into the
            interior of the coarse right-hand side dc (S = 0 or 2 with residual only) */,
            int ic_lo = -1, int ic_hi = -1 /* local coarse planes to write; default all */,
            const Geom *gce = nullptr, const double *ec = nullptr /* non-NULL: the input is vin + P(ec), the
            trilinear prolongation of the coarse field ec (smoothing-only launches, S = 2 or 4) */,
            int i_lo = -1, int i_hi = -1 /* local output planes of this launch; default all.  Several launches
            with disjoint windows and the same vin/vout make up one sweep (overlap with halo exchange) */,
            int edge = 0 /* > 0: only the first and the last `edge` planes of [i_lo, i_hi), as one launch of two chunks (-1 when
            the range is shorter than 2 * edge): the planes a halo exchange sends first, the interior in a second launch */);
/* FOUR colour passes starting with colour c1 and, into partials, the residual norm of the state after the SECOND one
 * (the launch that ends one V-cycle -- its last two post-smoothing passes and its norm -- and begins the next: mg3d_ctx.hip,
 * "carried cycles").  Returns the number of partials written or -1. */
int k_sweep_tap(const mg3d_options &o, const Geom &g, const double *vin, const double *d, double *vout, double *partials, int max_partials,
                double h, int c1, hipStream_t s, int acc_lo = 0, int acc_hi = -1, int i_lo = -1, int i_hi = -1, int edge = 0);
/* One launch per leg of a V(2,2) cycle on a level (mg3d_sweep.hip, "one launch per leg").  down: S = 4 colour passes red
 * first, or S = 3 black first (behind another cycle), + residual + full-weighting restriction into the interior of dc;
 * partials (S = 3 only): sum of diff^2 of the INCOMING state over the colour the first pass updates.  up: the input is
 * vin + P(ec), four passes black first; partials: sum of diff^2 of the RESULT over the colour the last pass updated.
 * Return value as k_sweep. */
int k_sweep_leg_down(const mg3d_options &o, const Geom &g, const double *vin, const double *d, double *vout, const Geom &gc, double *dc, double h, int S,
                     double *partials, int max_partials, hipStream_t s, int acc_lo = 0, int acc_hi = -1, int ic_lo = -1,
                     int ic_hi = -1, int i_lo = -1, int i_hi = -1);
int k_sweep_leg_up(const mg3d_options &o, const Geom &g, const double *vin, const double *d, double *vout, const Geom &gce, const double *ec, double h,
                   double *partials, int max_partials, hipStream_t s, int acc_lo = 0, int acc_hi = -1, int i_lo = -1,
                   int i_hi = -1, int edge = 0);
/* mg3d_tiny.hip: the level above the coarsest one in one workgroup (LDS-resident), when it fits (N <= 17) */
bool k_tiny_fits(const Geom &g, const Geom &gc);
/* zero guess, `iters` x (red, black), residual, restriction (interior + face injection from r's boundary) into dc */
void k_tiny_down(const Geom &g, double *u, const double *d, const double *r, const Geom &gc, double *dc, double h, int iters,
                 hipStream_t s);
/* u += P(ec) at every point, then `iters` x (black, red) */
void k_tiny_up(const Geom &g, double *u, const double *d, const Geom &gc, const double *ec, double h, int iters, hipStream_t s);
/* the bottom of the cycle in one launch: tiny_down on level 1, the direct solve of level 0 (reduced factor lin, the full
 * one as its fall-back), tiny_up on level 1; dc / xc receive level 0's right-hand side and solution */
bool k_tiny_cycle_fits(const Geom &g, const Geom &gc, const LuBand &lu, const LuBand &lin);
void k_tiny_cycle(const Geom &g, double *u, const double *d, const double *r, const Geom &gc, double *dc, double *xc,
                  const LuBand &lu, const LuBand &lin, double h, int iters, hipStream_t s);
/* b and x are level-0 grids in the padded layout g0; work holds 2n doubles */
/* steps per chunk of the streamed solve for n unknowns and rot_r = R on the current device, 0 if it cannot run */
int mg3d_lu_stream_chunk(int n, int R);
void k_lu_solve(const LuBand &lu, const LuBand &lu_in /* reduced factor; n == 0: none */, const Geom &g0, const double *b_pad, double *x_pad, double *work, hipStream_t s);

#endif
