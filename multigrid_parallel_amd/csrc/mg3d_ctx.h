/* mg3d_ctx.h -- solver context internals shared by mg3d_ctx.hip and mg3d_dist.hip (not installed). */
#ifndef MG3D_CTX_H
#define MG3D_CTX_H

#include <vector>

#include "mg3d_internal.h"

struct Level {
    Geom g;
    double h;
    size_t elems; /* doubles allocated per field */
    double *f[3]; /* u, d, r */
    double *alt;  /* second copy of u: the fused sweep writes out of place, then the two are swapped */
};

struct StageTimer {
    int calls;
    double seconds;
};

struct mg3d_ctx {
    int c, L, iters;
    double length;
    std::vector<Level> lv;
    hipStream_t stream;
    bool own_stream; /* false when a distributed driver shares one stream between contexts */
    LuBand lu;
    LuBand lu_in; /* the factor without its identity rows (n == 0: not built), see install_lu */
    bool have_lu;
    double *lu_work;  /* 2n doubles */
    double *partials; /* MG3D_MAX_PARTIALS doubles */
    double *sumsq;    /* device slots for squared norms */
    int sumsq_slots;
    double *h_sumsq;  /* pinned mirror */
    bool keep_r; /* materialise r on every level (reference-visible array) instead of restricting it on the fly */
    /* The faces of a coarse right-hand side are an injection of the fine residual's faces (mg_3d.h:879-958), and
     * calculateResidual never writes those (:824-825): they only change when somebody outside the cycle writes r of
     * level l or d of level l-1.  faces_dirty[l] says the injection l -> l-1 has to be redone (set at creation, by
     * upload / zero of those fields); faces_always[l] after a raw device pointer to one of them was handed out. */
    std::vector<char> faces_dirty, faces_always;
    /* the mixed-boundary problem of csrc/mg3d_es.hip, after mg3d_es_setup */
    bool have_es;
    mg3d_es_params es;
    /* carried cycles (mg3d_vcycles, see mg3d_enqueue_vcycle): u of the top level already holds the first three
     * pre-smoothing passes of the NEXT cycle; the finished cycle's own result is in the level's alt buffer */
    bool carried;
    /* one launch per leg on the top level (mg3d_enqueue_vcycle, "two launches per level"): 0 nothing outstanding;
     * 2 (inside mg3d_vcycles) the finished cycle's u is final, the second half of its norm rides on the next cycle's
     * down-leg (legs_slot, legs_npa: where it goes, how many partial sums the up-leg left); 3 (behind mg3d_vcycle) the
     * NEXT cycle's down-leg has already run, speculatively: u of the top level is three passes into it and the coarser
     * level's next right-hand side sits in that level's alt buffer; the finished cycle's own u is in the top level's alt */
    int legs_state, legs_slot, legs_npa;
    /* the last thing that happened to u of the top level is the red pass that ends a V(2,2) cycle and neither u nor d has been
     * touched since (mg3d_drop_carry clears it: every entry point that reads or writes level data calls that first): the next
     * cycle's first red pass is the identity also ACROSS calls, its down-leg can be the one launch (mg3d_enqueue_vcycle) */
    bool red_tail = false;
    bool raw_top; /* a raw device pointer to u or d of the top level was handed out (mg3d_device_view) */
    mg3d_options opt; /* launch / schedule policy (mg3d_options_init at creation, mg3d_ctx_set_option afterwards) */
    bool fused; /* fused sweep kernel (default) or one launch per colour pass (MG3D_NO_FUSE=1) */
    int timing; /* 0 off, 1 every level, 2 finest level only, 3 finest level's kernel timers only, 4 + k: 3 on every (k+2)-th cycle */
    int timing_phase; /* cycles since the last sampled one (timing >= 4) */
    std::vector<StageTimer> timers; /* [L][MG3D_NUM_STAGES] */
    /* stage timing never stalls the stream: event pairs are recorded in-stream and
     * resolved at the next host synchronisation the entry point does anyway */
    struct Pending {
        int slot; /* index into timers: stage timers first ([level][stage]), then kernel timers */
        hipEvent_t a, b;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
};

/* field `field` of `level` was written from outside the cycle (see faces_dirty) */
void mg3d_ctx_touched(mg3d_ctx *ctx, int field, int level, bool raw_pointer = false);
/* records a failure text for mg3d_last_error() and returns `code` */
int mg3d_fail(int code, const char *fmt, ...);
/* enqueue one V-cycle from level q of a (single-domain) context; squared norm of level q to sumsq[slot] */
/* carry_out: end the cycle with the launch that also starts the next one (only mg3d_vcycles asks, and never for the
 * last cycle of a call); ignored where mg3d_can_carry() says no */
/* carry_out: 0 the cycle ends the ordinary way; 1 another cycle of the same call follows; 2 the call ends here and runs
 * ahead speculatively (mg3d_vcycle) */
int mg3d_enqueue_vcycle(mg3d_ctx *ctx, int q, int slot, int carry_out = 0);
bool mg3d_can_carry(const mg3d_ctx *ctx, int q);
bool mg3d_can_legs(const mg3d_ctx *ctx, int q);
/* carried state -> the finished cycle's own u; a no-op (MG3D_OK) otherwise.  An error leaves the carried state in place. */
int mg3d_drop_carry(mg3d_ctx *ctx);
int mg3d_drop_carry_keep(mg3d_ctx *ctx); /* the same without clearing red_tail: mg3d_vcycle(s) themselves */

#endif
