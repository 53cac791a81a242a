/* mg3d_f32_int.h -- the single-precision variant's context, shared by mg3d_f32.hip (kernels, single-domain driver)
 * and mg3d_f32_dist.hip (the same cycle on i-slabs of several GPUs).  Not installed. */
#ifndef MG3D_F32_INT_H
#define MG3D_F32_INT_H

#include <vector>

#include "mg3d_ctx.h"

enum { /* finest-level launches of the fp32 variant (mg3d32_kernel_name) */
    MG3D32_K_PAIR = 0,       /* two sweeps in one launch */
    MG3D32_K_PAIR_TAP,       /* ... with the previous cycle's residual norm tapped from the first sweep's sums */
    MG3D32_K_PRO_PAIR,       /* prolongation + two sweeps */
    MG3D32_K_PRO_PAIR_NORM,  /* prolongation + two sweeps + residual norm (third stage) */
    MG3D32_K_PAIR_NORM,      /* two sweeps + residual norm */
    MG3D32_K_RESIDUAL_RESTRICT, /* residual + full-weighting restriction, r never stored */
    MG3D32_K_RESIDUAL,       /* residual (norm and / or store) */
    MG3D32_K_PROLONG,        /* prolongation on its own */
    MG3D32_K_SWEEP1,         /* one sweep per launch */
    MG3D32_NUM_KERNELS
};

struct Level32 {
    Geom g; /* pitch and plane in floats; an i-slab has ni = owned + halo planes and ig0 = global index of plane 0 */
    double hd; /* spacing as the hierarchy defines it (double); h = (float)hd */
    float h, hSq, invHsq;
    size_t elems;
    float *f[3]; /* u, d, r */
    float *alt;  /* the smoother's second buffer */
    /* the planes this context OWNS, local indices (the whole level, 0 .. ni, on a single domain): norms are summed
     * over them, restriction produces the coarse planes under them */
    int own_lo, own_hi;
};

struct mg3d32_ctx {
    int c, L, iters;
    float omega;
    std::vector<Level32> lv;
    mg3d_ctx *coarse64; /* one-level double context: LU factors and the direct solve */
    hipStream_t stream;
    bool own_stream;
    double *partials, *sumsq, *h_sumsq;
    int sumsq_slots;
    /* MG3D_F32_NO_PAIRS=1 / MG3D_F32_NO_FUSE=1, read when the context is created: one launch per sweep / per
     * operator instead of the paired and fused kernels (same bits; tests/test_gpu_f32.py) */
    bool no_pairs, no_fuse, no_carry; /* launch policy: the environment at creation, mg3d32_set_option afterwards */
    /* kernel timers of the finest level's launches (mg3d32_timing_enable): hipEvent pairs recorded in-stream, resolved at
     * the next host synchronisation the entry point does anyway */
    bool timing;
    struct KTime {
        int calls;
        double seconds;
    } kt[MG3D32_NUM_KERNELS];
    struct Pending32 {
        int k;
        hipEvent_t a, b;
    };
    std::vector<Pending32> pending;
    std::vector<hipEvent_t> event_pool;
};

/* a context whose levels >= first_slab are i-slabs: owned global planes [glo[l], ghi[l]) plus `halo` planes on every
 * side that is not a physical boundary; `share` (may be NULL) = the stream to run on */
int mg3d32_create_slabs(int coarse_pts, int num_levels, int smooth_iters, double omega, double grid_length,
                        int first_slab, const int *glo, const int *ghi, int halo, hipStream_t share, mg3d32_ctx **out);

/* launchers on one context (asynchronous on its stream) */
bool e32_jacobi(mg3d32_ctx *ctx, int level, int iters, int norm_slot = -1, bool prolong_first = false, int tap_slot = -1);
void e32_residual(mg3d32_ctx *ctx, int level, bool store, int slot);
void e32_restrict(mg3d32_ctx *ctx, int level);
/* residual + restriction of the coarse planes [c_lo, c_hi) (local indices of the coarser level; -1: those under
 * the owned fine planes) */
void e32_residual_restrict(mg3d32_ctx *ctx, int level, int c_lo = -1, int c_hi = -1);
void e32_prolong(mg3d32_ctx *ctx, int level);
int e32_coarse_solve(mg3d32_ctx *ctx);
void e32_fill_boundary(mg3d32_ctx *ctx, int field, int level);
/* carry_out: the cycle's norm is left to the next cycle's first launch (which must pass tap_slot = this cycle's slot);
 * only where e32_can_carry() */
int e32_vcycle(mg3d32_ctx *ctx, int q, int slot, bool carry_out = false, int tap_slot = -1);
bool e32_can_carry(const mg3d32_ctx *ctx);

#endif
