/*
 * mg3d_ctx.hip -- solver context and V-cycle sequencing behind the C ABI (include/mg3d.h).
 *
 * The context is the device-resident counterpart of the reference's global
 * state (mg_3d.h:19-28): three level hierarchies u, d, r plus the factored
 * coarsest operator.  Everything is enqueued on one HIP stream; the only host
 * synchronisations are the ones an entry point's contract requires (returning
 * a norm, copying data back).
 */
#include "mg3d_ctx.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

/* ------------------------------------------------------------------ errors */
static thread_local char g_err[512] = "";

int mg3d_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define fail mg3d_fail

#define HIPCHK(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(MG3D_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,    \
                        __LINE__);                                                                         \
    } while (0)

#define CHK(call)              \
    do {                       \
        int rc_ = (call);      \
        if (rc_ != MG3D_OK)    \
            return rc_;        \
    } while (0)

extern "C" const char *mg3d_last_error(void) { return g_err; }

static const char *const kStageNames[MG3D_NUM_STAGES] = {"Smoother1",          "CalcResidual1", "Restrict Residual",
                                                         "Recurse, Direct Solve", "Prolongate&Correct", "Smoother2",
                                                         "CalcResidual2"}; /* mg_3d.h:136-137 */

static const char *const kKernelNames[MG3D_NUM_KERNELS] = {"sweep4", "sweep2", "sweep2+residual", "residual",
                                                           "restrict", "prolong", "coarse_solve", "colour_pass",
                                                           "sweep4+norm", "sweep1+restrict", "leg_down", "leg_up"};

extern "C" const char *mg3d_kernel_name(int k) { return (k >= 0 && k < MG3D_NUM_KERNELS) ? kKernelNames[k] : "?"; }

extern "C" const char *mg3d_stage_name(int stage)
{
    return (stage >= 0 && stage < MG3D_NUM_STAGES) ? kStageNames[stage] : "?";
}

extern "C" int mg3d_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

static int require_device(void)
{
    if (mg3d_device_count() <= 0)
        return fail(MG3D_ERR_NO_DEVICE, "no HIP device available: libmg3d has no CPU fallback");
    return MG3D_OK;
}

/* ----------------------------------------------------------------- context */
static hipEvent_t take_event(mg3d_ctx *ctx)
{
    hipEvent_t e = nullptr;
    if (!ctx->event_pool.empty()) {
        e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
        e = nullptr;
    }
    return e;
}

/* call only after the stream has been synchronised */
static void resolve_timers(mg3d_ctx *ctx)
{
    for (auto &p : ctx->pending) {
        float ms = 0.f;
        if (p.a && p.b && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            StageTimer &t = ctx->timers[(size_t)p.slot];
            t.calls++;
            t.seconds += ms * 1e-3;
        }
        if (p.a)
            ctx->event_pool.push_back(p.a);
        if (p.b)
            ctx->event_pool.push_back(p.b);
    }
    ctx->pending.clear();
}

/* scoped event pair: a stage of the reference's timing table, or (kernel = true) one kernel launch.
 * timing: 1 every level, 2 the finest level, 3 the finest level's kernel scopes only (what bench.py's roofline needs: 8
 * marker packets per cycle instead of 24; each costs ~5 us of idle queue, 0.09 against 0.04 ms of a 3.3 ms cycle.  Binding
 * the pair to the dispatch itself, hipExtLaunchKernelGGL, measured the same 0.04 ms as the 8 markers: not kept), 4 + k
 * (k >= 0): as 3, but only every (k + 2)-th full cycle carries the markers (a sample of the timed region). */
struct StageScope {
    mg3d_ctx *ctx;
    mg3d_ctx::Pending p;
    bool on;
    StageScope(mg3d_ctx *c, int l, int s, bool kernel = false) : ctx(c)
    {
        p.slot = kernel ? c->L * MG3D_NUM_STAGES + l * MG3D_NUM_KERNELS + s : l * MG3D_NUM_STAGES + s;
        p.a = p.b = nullptr;
        on = ctx->timing == 1 || (ctx->timing == 2 && l == ctx->L - 1) ||
             (ctx->timing >= 3 && kernel && l == ctx->L - 1 && ctx->timing_phase == 0);
        if (on && (p.a = take_event(ctx)))
            (void)hipEventRecord(p.a, ctx->stream);
    }
    ~StageScope()
    {
        if (!on)
            return;
        if ((p.b = take_event(ctx)))
            (void)hipEventRecord(p.b, ctx->stream);
        ctx->pending.push_back(p);
    }
};


static void free_band(LuBand &b)
{
    for (double *p : {b.lcol, b.ucol, b.diag, b.lrot, b.urot, b.stream})
        if (p)
            (void)hipFree(p);
    if (b.in_map)
        (void)hipFree(b.in_map);
    memset(&b, 0, sizeof b);
}

static void free_lu(mg3d_ctx *ctx)
{
    free_band(ctx->lu);
    free_band(ctx->lu_in);
    if (ctx->lu_work)
        (void)hipFree(ctx->lu_work);
    ctx->lu_work = nullptr;
    ctx->have_lu = false;
}

extern "C" int mg3d_ctx_destroy(mg3d_ctx *ctx)
{
    if (!ctx)
        return MG3D_OK;
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    for (auto &l : ctx->lv)
    {
        for (int k = 0; k < 3; k++)
            if (l.f[k])
                (void)hipFree(l.f[k]);
        if (l.alt)
            (void)hipFree(l.alt);
    }
    free_lu(ctx);
    if (ctx->partials)
        (void)hipFree(ctx->partials);
    if (ctx->sumsq)
        (void)hipFree(ctx->sumsq);
    if (ctx->h_sumsq)
        (void)hipHostFree(ctx->h_sumsq);
    resolve_timers(ctx);
    for (hipEvent_t e : ctx->event_pool)
        (void)hipEventDestroy(e);
    if (ctx->stream && ctx->own_stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MG3D_OK;
}

/* ------------------------------------------------------------------ options
 * One table: key, environment override (read once per context creation), how the environment value maps, default. */
struct OptionRow {
    const char *key, *env;
    int mode; /* 0: the value is the integer; 1: "1" means 0 (a NO_ switch), anything else 1; 2: "rj,nw,pf" */
    int dflt;
};
static const OptionRow kOptionTable[MG3D_OPT_COUNT] = {
    {"carry", "MG3D_NO_CARRY", 1, 1},
    {"carry_min", "MG3D_CARRY_MIN", 0, 130},
    {"legs", "MG3D_LEGS", 0, 1},
    {"legs_min", "MG3D_LEGS_MIN", 0, 160},
    {"tiny", "MG3D_NO_TINY", 1, 1},
    {"tiny_cycle", "MG3D_NO_TINY_CYCLE", 1, 1},
    {"lu_reduced", "MG3D_LU_REDUCED", 0, 1},
    {"fuse_rst2", "MG3D_FUSE_RST2", 0, -1},
    {"small_max", "MG3D_SMALL_MAX", 0, 129},
    {"fuse_leg_max", "MG3D_FUSE_LEG_MAX", 0, 0},
    {"fuse_up_max", "MG3D_FUSE_UP_MAX", 0, 1 << 20},
    {"sweep_tune", "MG3D_SWEEP_TUNE", 0, -1},
    {"sweep_tune_log", "MG3D_SWEEP_TUNE_LOG", 0, 0},
    {"sweep_ci", "MG3D_SWEEP_CI", 0, 0},
    {"sweep_rj", "MG3D_SWEEP_CFG", 2, 0},
    {"sweep_nw", nullptr, 0, 0},
    {"sweep_pf", nullptr, 0, 0},
};

void mg3d_options_init(mg3d_options *o)
{
    for (int i = 0; i < MG3D_OPT_COUNT; i++) {
        const OptionRow &r = kOptionTable[i];
        o->v[i] = r.dflt;
        const char *e = r.env ? getenv(r.env) : nullptr; /* context creation: the only place the environment is read */
        if (!e || !e[0])
            continue;
        if (r.mode == 0)
            o->v[i] = atoi(e);
        else if (r.mode == 1)
            o->v[i] = e[0] == '1' ? 0 : 1;
        else {
            int rj = 0, nw = 0, pf = 0;
            if (sscanf(e, "%d,%d,%d", &rj, &nw, &pf) >= 2) {
                o->v[MG3D_OPT_SWEEP_RJ] = rj;
                o->v[MG3D_OPT_SWEEP_NW] = nw;
                o->v[MG3D_OPT_SWEEP_PF] = pf;
            }
            i += 2; /* the two rows behind it were just set */
        }
    }
}

int mg3d_option_index(const char *key)
{
    if (key)
        for (int i = 0; i < MG3D_OPT_COUNT; i++)
            if (strcmp(key, kOptionTable[i].key) == 0)
                return i;
    return -1;
}

const char *mg3d_option_key(int index) { return index >= 0 && index < MG3D_OPT_COUNT ? kOptionTable[index].key : nullptr; }

extern "C" const char *mg3d_option_name(int index) { return mg3d_option_key(index); }

static mg3d_ctx *ctx_new(int L, int iters)
{
    mg3d_ctx *ctx = new mg3d_ctx();
    ctx->c = 0;
    ctx->length = 0.;
    ctx->L = L;
    ctx->iters = iters;
    ctx->have_lu = false;
    memset(&ctx->lu, 0, sizeof ctx->lu);
    memset(&ctx->lu_in, 0, sizeof ctx->lu_in);
    ctx->lu_work = nullptr;
    ctx->partials = ctx->sumsq = ctx->h_sumsq = nullptr;
    ctx->sumsq_slots = 0;
    ctx->stream = nullptr;
    ctx->own_stream = true;
    ctx->timing = 0;
    ctx->timing_phase = 0;
    ctx->timers.assign((size_t)L * (MG3D_NUM_STAGES + MG3D_NUM_KERNELS), StageTimer{0, 0.});
    ctx->lv.resize(L);
    for (auto &l : ctx->lv)
        l.f[0] = l.f[1] = l.f[2] = l.alt = nullptr;
    ctx->have_es = false;
    ctx->faces_dirty.assign(L, 1);
    ctx->faces_always.assign(L, 0);
    ctx->fused = true;
    ctx->carried = false;
    ctx->legs_state = ctx->legs_slot = ctx->legs_npa = 0;
    mg3d_options_init(&ctx->opt);
    ctx->raw_top = false;
    ctx->keep_r = false;
    if (const char *e = getenv("MG3D_KEEP_R"))
        ctx->keep_r = e[0] == '1';
    if (const char *e = getenv("MG3D_NO_FUSE"))
        ctx->fused = !(e[0] == '1');
    return ctx;
}

/* carried cycles (see mg3d_enqueue_vcycle): back to the finished cycle's own result.  The launch that carried the cycle
 * over took u behind the cycle's first two post-smoothing passes (still intact in the alt buffer) through its last two,
 * tapped the norm there and went on into the next cycle: the finished cycle's u itself was never written.  It is those
 * two passes (black, red) from alt -- one launch, paid only when somebody wants to see or change the state between two
 * cycles.  Every entry point that reads or writes level data, or changes what a cycle is, calls this first; only
 * mg3d_vcycle(s) continue from the carried state. */
int mg3d_drop_carry(mg3d_ctx *ctx)
{
    /* whoever calls this may go on to change u or d of the top level: the next cycle can no longer rely on its first red
     * pass being the identity (red_tail, mg3d_enqueue_vcycle); mg3d_vcycle(s) themselves use mg3d_drop_carry_keep */
    if (ctx)
        ctx->red_tail = false;
    return mg3d_drop_carry_keep(ctx);
}

int mg3d_drop_carry_keep(mg3d_ctx *ctx)
{
    if (ctx && ctx->legs_state != 0) {
        /* one launch per leg: behind mg3d_vcycle the next cycle's down-leg has run ahead into the alt buffers; the finished
         * cycle's own u is intact in the top level's alt, the coarser level's right-hand side was never touched -- swap
         * back, no launch.  (State 2 only exists inside mg3d_vcycles: u is final, the norm's second half is abandoned.) */
        if (ctx->legs_state == 3) {
            Level &l = ctx->lv[ctx->L - 1];
            double *t = l.f[MG3D_U];
            l.f[MG3D_U] = l.alt;
            l.alt = t;
        }
        ctx->legs_state = 0;
    }
    if (!ctx || !ctx->carried)
        return MG3D_OK;
    Level &l = ctx->lv[ctx->L - 1];
    const int np = k_sweep(ctx->opt, l.g, l.alt, l.f[MG3D_D], l.f[MG3D_U], nullptr, nullptr, MG3D_MAX_PARTIALS, l.h, 2, 0, false,
                           ctx->stream);
    /* a failure leaves the context where it was -- u three passes into the next cycle, `carried` still set: the caller
     * returns the error instead of going on with (and handing out) a state nobody asked for; a later call tries again */
    if (np < 0)
        return fail(MG3D_ERR_STATE, "carried cycle: the two passes that finish it could not be launched");
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(MG3D_ERR_HIP, "carried cycle: the two passes that finish it: %s", hipGetErrorString(e));
    ctx->carried = false;
    return MG3D_OK;
}

void mg3d_ctx_touched(mg3d_ctx *ctx, int field, int level, bool raw_pointer)
{
    /* a raw device pointer to u or d of the top level: the library no longer sees every write to them, so a single
     * mg3d_vcycle call does not run ahead into the next cycle any more (mg3d_vcycles still carries INSIDE a call) */
    if (raw_pointer && level == ctx->L - 1 && (field == MG3D_U || field == MG3D_D))
        ctx->raw_top = true;
    const int l = field == MG3D_R ? level : field == MG3D_D ? level + 1 : -1;
    if (l >= 1 && l < ctx->L) {
        ctx->faces_dirty[l] = 1;
        if (raw_pointer)
            ctx->faces_always[l] = 1;
    }
}

static int ctx_create_sizes(const int *n_per_level, const double *h_per_level, int L, int iters, mg3d_ctx **out)
{
    CHK(require_device());
    mg3d_ctx *ctx = ctx_new(L, iters);
#define CTXCHK(call)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            int rc_ = fail(e_ == hipErrorOutOfMemory ? MG3D_ERR_ALLOC : MG3D_ERR_HIP, "%s failed: %s", \
                           #call, hipGetErrorString(e_));                                                \
            mg3d_ctx_destroy(ctx);                                                                       \
            return rc_;                                                                                  \
        }                                                                                                \
    } while (0)
    CTXCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    for (int l = 0; l < L; l++) {
        Level &lev = ctx->lv[l];
        const int N = n_per_level[l];
        lev.g.N = N;
        lev.g.ni = lev.g.nj = lev.g.nk = N;
        lev.g.ig0 = 0;
        lev.g.pitch = mg3d_pitch_for(N);
        lev.g.plane = (long long)lev.g.pitch * N;
        lev.h = h_per_level[l];
        lev.elems = (size_t)lev.g.plane * N;
        for (int k = 0; k < 3; k++) {
            CTXCHK(hipMalloc(&lev.f[k], lev.elems * sizeof(double)));
            CTXCHK(hipMemsetAsync(lev.f[k], 0, lev.elems * sizeof(double), ctx->stream)); /* calloc, mg_3d.h:44 */
        }
        CTXCHK(hipMalloc(&lev.alt, lev.elems * sizeof(double)));
        CTXCHK(hipMemsetAsync(lev.alt, 0, lev.elems * sizeof(double), ctx->stream));
    }
    CTXCHK(hipMalloc(&ctx->partials, MG3D_MAX_PARTIALS * sizeof(double)));
    ctx->sumsq_slots = 1024;
    CTXCHK(hipMalloc(&ctx->sumsq, ctx->sumsq_slots * sizeof(double)));
    CTXCHK(hipHostMalloc(&ctx->h_sumsq, ctx->sumsq_slots * sizeof(double)));
    CTXCHK(hipStreamSynchronize(ctx->stream));
#undef CTXCHK
    *out = ctx;
    return MG3D_OK;
}

extern "C" int mg3d_ctx_create(int coarse_pts, int num_levels, int smooth_iters, double grid_length, mg3d_ctx **out)
{
    if (!out || coarse_pts < 3 || num_levels < 1 || num_levels > 24 || smooth_iters < 0)
        return fail(MG3D_ERR_ARG, "mg3d_ctx_create: bad arguments (c=%d L=%d iters=%d)", coarse_pts, num_levels,
                    smooth_iters);
    const long long finest = ((long long)(coarse_pts - 1) << (num_levels - 1)) + 1; /* mg_3d.h:126-127 */
    if (finest > 2049)
        return fail(MG3D_ERR_ARG, "mg3d_ctx_create: finest grid %lld^3 too large", finest);
    std::vector<int> n(num_levels);
    std::vector<double> h(num_levels);
    const double spacing = grid_length / (double)(finest - 1); /* mg_3d.h:143 */
    for (int l = 0; l < num_levels; l++) {
        n[l] = (coarse_pts - 1) * (1 << l) + 1; /* mg_3d.h:41 */
        h[l] = 0.;
    }
    /* spacing doubles per coarser level exactly as vcycle does (h_coarse = 2*h, mg_3d.h:1303) */
    h[num_levels - 1] = spacing;
    for (int l = num_levels - 2; l >= 0; l--)
        h[l] = 2 * h[l + 1];
    CHK(ctx_create_sizes(n.data(), h.data(), num_levels, smooth_iters, out));
    (*out)->c = coarse_pts;
    (*out)->length = grid_length;
    return MG3D_OK;
}

extern "C" int mg3d_ctx_num_levels(const mg3d_ctx *ctx) { return ctx ? ctx->L : 0; }
extern "C" int mg3d_ctx_level_n(const mg3d_ctx *ctx, int level)
{
    return (ctx && level >= 0 && level < ctx->L) ? ctx->lv[level].g.N : 0;
}
extern "C" double mg3d_ctx_level_h(const mg3d_ctx *ctx, int level)
{
    return (ctx && level >= 0 && level < ctx->L) ? ctx->lv[level].h : 0.;
}
extern "C" int mg3d_ctx_set_keep_residual(mg3d_ctx *ctx, int keep)
{
    CHK(mg3d_drop_carry(ctx));
    if (!ctx)
        return fail(MG3D_ERR_ARG, "mg3d_ctx_set_keep_residual: NULL context");
    ctx->keep_r = keep != 0;
    return MG3D_OK;
}

/* launch / schedule policy by key (the table in INTEGRATION.md): takes effect from the next call on; anything that
 * changes what a cycle is first finishes a cycle that has run ahead */
extern "C" int mg3d_ctx_set_option(mg3d_ctx *ctx, const char *key, int value)
{
    CHK(mg3d_drop_carry(ctx));
    if (!ctx)
        return fail(MG3D_ERR_ARG, "mg3d_ctx_set_option: NULL context");
    const int i = mg3d_option_index(key);
    if (i < 0)
        return fail(MG3D_ERR_ARG, "mg3d_ctx_set_option: no option \"%s\"", key ? key : "(null)");
    ctx->opt.v[i] = value;
    return MG3D_OK;
}

extern "C" int mg3d_ctx_get_option(const mg3d_ctx *ctx, const char *key, int *value)
{
    const int i = mg3d_option_index(key);
    if (!ctx || !value || i < 0)
        return fail(MG3D_ERR_ARG, "mg3d_ctx_get_option: NULL argument or no option \"%s\"", key ? key : "(null)");
    *value = ctx->opt.v[i];
    return MG3D_OK;
}

extern "C" int mg3d_ctx_set_smooth_iters(mg3d_ctx *ctx, int iters)
{
    CHK(mg3d_drop_carry(ctx));
    if (!ctx || iters < 0)
        return fail(MG3D_ERR_ARG, "mg3d_ctx_set_smooth_iters: bad arguments");
    ctx->iters = iters;
    return MG3D_OK;
}

/* --------------------------------------------------------------- coarse LU */
/* Banded, column-major copy of a row-major LU factor (entry (i, j) = at(i, j)) for the device solve:
 * bw = populated half bandwidth (max |i-j| with LU[i][j] != 0). */
template <class At> static int build_band(LuBand &out, long long n, At at)
{
    int bw = 1;
    for (long long i = 0; i < n; i++) {
        long long lo = 0, hi = n - 1;
        while (lo < i && at(i, lo) == 0.)
            lo++;
        while (hi > i && at(i, hi) == 0.)
            hi--;
        if (i - lo > bw)
            bw = (int)(i - lo);
        if (hi - i > bw)
            bw = (int)(hi - i);
    }
    std::vector<double> lcol((size_t)n * bw, 0.), ucol((size_t)n * bw, 0.), diag((size_t)2 * ((n + 63) / 64 * 64), 1.);
    int fast_div = 1;
    const long long npad = (n + 63) / 64 * 64;
    for (long long j = 0; j < n; j++) {
        diag[j] = at(j, j);
        /* reciprocal for lu_div(): correctly rounded by the host's IEEE division; 0 = "divide the ordinary way" */
        const double ad = fabs(diag[j]);
        if (!(ad >= 0x1p-460 && ad <= 0x1p460))
            fast_div = 0;
        diag[npad + j] = 1.0 / diag[j];
        for (int t = 0; t < bw; t++) {
            const long long il = j + 1 + t, iu = j - 1 - t;
            if (il < n)
                lcol[j * bw + t] = at(il, j);
            if (iu >= 0)
                ucol[j * bw + t] = at(iu, j);
        }
    }
    free_band(out);
    out.n = (int)n;
    out.bw = bw;
    out.fast_div = fast_div;
    out.npad = (int)npad;
    HIPCHK(hipMalloc(&out.lcol, lcol.size() * sizeof(double)));
    HIPCHK(hipMalloc(&out.ucol, ucol.size() * sizeof(double)));
    HIPCHK(hipMalloc(&out.diag, diag.size() * sizeof(double)));
    HIPCHK(hipMemcpy(out.lcol, lcol.data(), lcol.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(out.ucol, ucol.data(), ucol.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(out.diag, diag.data(), diag.size() * sizeof(double), hipMemcpyHostToDevice));
    /* narrow bands (coarse grids up to 11^3): lane-rotated copies for the single-wave kernel */
    const int R = (bw + 63) / 64;
    if (R <= 2 && 4 * (size_t)n * sizeof(double) <= 60000) {
        std::vector<double> lrot((size_t)n * 64 * R, 0.), urot((size_t)n * 64 * R, 0.);
        for (long long j = 0; j < n; j++)
            for (int q = 0; q < R; q++)
                for (int l = 0; l < 64; l++) {
                    const int tf = (int)((l - j - 1) & 63) + 64 * q, tb = (int)((j - 1 - l) & 63) + 64 * q;
                    const long long irow_f = j + 1 + tf, irow_b = j - 1 - tb;
                    if (tf < bw && irow_f < n)
                        lrot[(size_t)j * 64 * R + 64 * q + l] = at(irow_f, j);
                    if (tb < bw && irow_b >= 0)
                        urot[(size_t)j * 64 * R + 64 * q + l] = at(irow_b, j);
                }
        HIPCHK(hipMalloc(&out.lrot, lrot.size() * sizeof(double)));
        HIPCHK(hipMalloc(&out.urot, urot.size() * sizeof(double)));
        HIPCHK(hipMemcpy(out.lrot, lrot.data(), lrot.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(out.urot, urot.data(), urot.size() * sizeof(double), hipMemcpyHostToDevice));
        out.rot_r = R;
        /* stream for the loader/solver kernel (see lu_solve_stream_kernel) when its LDS ring fits */
        const int CH = mg3d_lu_stream_chunk((int)n, R);
        if (CH > 0) {
            const long long nch = npad / CH, step_d = 64 * R;
            std::vector<double> st((size_t)((2 * nch + 2) * CH * step_d), 0.);
            /* per lane the pair is stored as (factor for sum A, factor for sum B) of lu_stream_half: a lane
             * that has already finalised its unknown of the current 64-step chunk (or does so in this step)
             * carries its nearer row in B */
            for (long long j = 0; j < n; j++)
                for (int l = 0; l < 64; l++) {
                    const int of = (int)(j & 63), ob = (int)(j & 63);
                    const bool swf = R == 2 && l <= of, swb = R == 2 && l >= ob;
                    for (int q = 0; q < R; q++) {
                        const int qf = swf ? 1 - q : q, qb = swb ? 1 - q : q;
                        st[(size_t)(j * step_d + l * R + qf)] = lrot[(size_t)j * 64 * R + 64 * q + l];
                        st[(size_t)((npad + (npad - 1 - j)) * step_d + l * R + qb)] = urot[(size_t)j * 64 * R + 64 * q + l];
                    }
                }
            HIPCHK(hipMalloc(&out.stream, st.size() * sizeof(double)));
            HIPCHK(hipMemcpy(out.stream, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice));
            out.stream_ch = CH;
        }
    }
    return MG3D_OK;
}

static int install_lu(mg3d_ctx *ctx, const double *LU, long long n, size_t work_doubles)
{
    free_lu(ctx);
    CHK(build_band(ctx->lu, n, [&](long long i, long long j) { return LU[i * n + j]; }));
    HIPCHK(hipMalloc(&ctx->lu_work, work_doubles * sizeof(double)));
    /* The reduced system.  A row of the factor that is the identity row (constructCoarseMatrixA's boundary rows,
     * mg_3d.h:179-185; elimination leaves them alone) gives x[i] = b[i] in both substitutions, and when that b[i] is
     * +-0 -- every boundary entry of a V-cycle's coarse right-hand side: the injected faces of a residual that is never
     * written there, mg_3d.h:824-825, 879-958 -- its products with other rows' factors are +-0 and change no running sum
     * (sums start at +0 and never become -0).  What is left is the factor restricted to the other rows (9^3: 343 of 729
     * unknowns, half-band 49 instead of 81): the same values in the same order for every remaining term, i.e. the same
     * bits, in less than half the strictly sequential steps.  The solve kernel checks the right-hand side and takes
     * the full system whenever an identity row's entry is not a zero (the F-cycle start, host-pointer calls). */
    if (!ctx->opt.v[MG3D_OPT_LU_REDUCED] || !ctx->lu.stream_ch) /* (option lu_reduced, read per factor: tests compare both) */
        return MG3D_OK;
    std::vector<int> map((size_t)n, -1), rows;
    for (long long i = 0; i < n; i++) {
        bool ident = LU[i * n + i] == 1.;
        const long long lo = i - ctx->lu.bw < 0 ? 0 : i - ctx->lu.bw, hi = i + ctx->lu.bw >= n ? n - 1 : i + ctx->lu.bw;
        for (long long j = lo; j <= hi && ident; j++)
            ident = j == i || LU[i * n + j] == 0.;
        if (!ident) {
            map[(size_t)i] = (int)rows.size();
            rows.push_back((int)i);
        }
    }
    const long long ni = (long long)rows.size(), npad_in = (ni + 63) / 64 * 64;
    if (ni == 0 || ni == n || 2 * npad_in > ctx->lu.npad)
        return MG3D_OK; /* nothing to gain, or the reduced vectors do not fit beside the full right-hand side in LDS */
    CHK(build_band(ctx->lu_in, ni, [&](long long a, long long b) { return LU[(long long)rows[(size_t)a] * n + rows[(size_t)b]]; }));
    if (!ctx->lu_in.stream_ch) {
        free_band(ctx->lu_in);
        return MG3D_OK;
    }
    HIPCHK(hipMalloc(&ctx->lu.in_map, (size_t)n * sizeof(int)));
    HIPCHK(hipMemcpy(ctx->lu.in_map, map.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    return MG3D_OK;
}

extern "C" int mg3d_ctx_set_lu(mg3d_ctx *ctx, const double *LU)
{
    CHK(mg3d_drop_carry(ctx));
    if (!ctx || !LU)
        return fail(MG3D_ERR_ARG, "mg3d_ctx_set_lu: NULL argument");
    const int N0 = ctx->lv[0].g.N;
    const long long n = (long long)N0 * N0 * N0;
    if (n * n >= 2147483647LL) /* assert(totalNodes*totalNodes < INT_MAX), mg_3d.h:163 */
        return fail(MG3D_ERR_ARG, "mg3d_ctx_set_lu: coarse grid %d^3 too large for a dense factor", N0);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    CHK(install_lu(ctx, LU, n, 2 * (size_t)n));
    ctx->have_lu = true;
    ctx->have_es = false; /* whatever factor was loaded before (mg3d_es_setup re-arms it after its own call) */
    return MG3D_OK;
}

extern "C" int mg3d_ctx_build_coarse(mg3d_ctx *ctx, double h_coarse)
{
    CHK(mg3d_drop_carry(ctx));
    if (!ctx)
        return fail(MG3D_ERR_ARG, "mg3d_ctx_build_coarse: NULL context");
    const int N0 = ctx->lv[0].g.N;
    const long long n = (long long)N0 * N0 * N0;
    if (n * n >= 2147483647LL)
        return fail(MG3D_ERR_ARG, "mg3d_ctx_build_coarse: coarse grid %d^3 too large for a dense factor", N0);
    double *A = (double *)calloc((size_t)(n * n), sizeof(double)); /* mg_3d.h:283 */
    if (!A)
        return fail(MG3D_ERR_ALLOC, "mg3d_ctx_build_coarse: out of host memory");
    mg3d_coarse_matrix(A, N0, h_coarse); /* mg_3d.h:288 */
    mg3d_lu_factor(A, (int)n);           /* mg_3d.h:289 */
    const int rc = mg3d_ctx_set_lu(ctx, A);
    free(A);
    return rc;
}

/* ----------------------------------------------------------- data movement */
static int check_field_level(const mg3d_ctx *ctx, int field, int level, const char *who)
{
    if (!ctx || field < 0 || field > 2 || level < 0 || level >= ctx->L)
        return fail(MG3D_ERR_ARG, "%s: bad field/level (%d, %d)", who, field, level);
    return MG3D_OK;
}

extern "C" int mg3d_upload(mg3d_ctx *ctx, int field, int level, const double *host)
{
    CHK(mg3d_drop_carry(ctx));
    CHK(check_field_level(ctx, field, level, "mg3d_upload"));
    if (!host)
        return fail(MG3D_ERR_ARG, "mg3d_upload: NULL host pointer");
    const Level &l = ctx->lv[level];
    const int N = l.g.N;
    HIPCHK(hipMemcpy2DAsync(l.f[field], l.g.pitch * sizeof(double), host, N * sizeof(double), N * sizeof(double),
                            (size_t)N * N, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    mg3d_ctx_touched(ctx, field, level);
    return MG3D_OK;
}

extern "C" int mg3d_download(mg3d_ctx *ctx, int field, int level, double *host)
{
    CHK(mg3d_drop_carry_keep(ctx)); /* (reads only: the next cycle may still continue behind the last one, see red_tail) */
    CHK(check_field_level(ctx, field, level, "mg3d_download"));
    if (!host)
        return fail(MG3D_ERR_ARG, "mg3d_download: NULL host pointer");
    const Level &l = ctx->lv[level];
    const int N = l.g.N;
    HIPCHK(hipMemcpy2DAsync(host, N * sizeof(double), l.f[field], l.g.pitch * sizeof(double), N * sizeof(double),
                            (size_t)N * N, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MG3D_OK;
}

extern "C" int mg3d_zero(mg3d_ctx *ctx, int field, int level)
{
    CHK(mg3d_drop_carry(ctx));
    CHK(check_field_level(ctx, field, level, "mg3d_zero"));
    const Level &l = ctx->lv[level];
    HIPCHK(hipMemsetAsync(l.f[field], 0, l.elems * sizeof(double), ctx->stream));
    mg3d_ctx_touched(ctx, field, level);
    return MG3D_OK;
}

extern "C" int mg3d_sync(mg3d_ctx *ctx)
{
    if (!ctx)
        return fail(MG3D_ERR_ARG, "mg3d_sync: NULL context");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    resolve_timers(ctx);
    return MG3D_OK;
}

extern "C" int mg3d_device_view(mg3d_ctx *ctx, int field, int level, void **dev_ptr, int *pitch_doubles,
                                long *plane_doubles)
{
    CHK(mg3d_drop_carry(ctx));
    CHK(check_field_level(ctx, field, level, "mg3d_device_view"));
    const Level &l = ctx->lv[level];
    mg3d_ctx_touched(ctx, field, level, true);
    if (dev_ptr)
        *dev_ptr = l.f[field];
    if (pitch_doubles)
        *pitch_doubles = l.g.pitch;
    if (plane_doubles)
        *plane_doubles = (long)l.g.plane;
    return MG3D_OK;
}

/* ----------------------------------------------------------------- operators */
static int launch_ok(const char *who)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(MG3D_ERR_HIP, "%s: kernel launch failed: %s", who, hipGetErrorString(e));
    return MG3D_OK;
}

static int read_norm(mg3d_ctx *ctx, int slot, double *norm)
{
    if (!norm)
        return MG3D_OK;
    HIPCHK(hipMemcpyAsync(ctx->h_sumsq + slot, ctx->sumsq + slot, sizeof(double), hipMemcpyDeviceToHost,
                          ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    resolve_timers(ctx);
    *norm = sqrt(ctx->h_sumsq[slot]); /* mg_3d.h:841 */
    return MG3D_OK;
}

/* post-smoothing of 4 passes + norm as 2 + 2 passes (see enqueue_smooth_residual) */
static bool split_up_leg(int iters, int want_res) { return 2 * iters == 4 && want_res == 1; }

/* iters x (two colour passes), optionally followed by the residual of the result.
 * Fused path: chunks of 4 (or 2) passes per launch, each launch reading u and writing the
 * alternate buffer; the residual rides on the last launch.  want_res: 0 none, 1 norm only,
 * 2 store r (+ norm).  The squared norm goes to sumsq[slot]. */
static int enqueue_smooth_residual(mg3d_ctx *ctx, int level, int post, int iters, int want_res, int slot,
                                    Level *coarse = nullptr, const Level *pro = nullptr, bool zero_in = false,
                                    bool need_norm = true)
{
    /* need_norm = false: only r (or its restriction) is wanted -- the V-cycle drops the pre-smoothing norm
     * (mg_3d.h:1294 ignores calculateResidual's value) -- so no partial sums are produced or folded */
    /* zero_in: u of this level is to be taken as identically zero (the memset of mg_3d.h:1258-1259 folded
     * into the first launch: it neither reads u nor needs it zeroed); only valid when a launch with S > 0 follows */
    /* pro != NULL: the smoother's input is u + P(pro->u) (prolongateAndCorrectError, mg_3d.h:1331, folded
     * into the first launch's loads); the caller must have checked pro_fusable() */
    /* coarse != NULL (with want_res != 0): the residual is restricted on the fly into the interior of
     * coarse->d and never stored; the caller adds the face injection (k_restrict, faces_only) */
    Level &l = ctx->lv[level];
    hipStream_t s = ctx->stream;
    const int c1 = post ? 0 : 1; /* pre: red first (mg_3d.h:657); post: black first (mg_3d.h:728) */
    if (ctx->fused) {
        int passes = 2 * iters;
        bool done_res = want_res == 0;
        while (passes > 0 || !done_res) {
            /* Where the norm is wanted behind FOUR post-smoothing passes (the top level of a V(2,2) cycle) the stage
             * runs as 2 + 2 passes: the first launch takes the prolongation into its loads (the 2-pass shape has the
             * registers for it: 0.77 ms against 0.74 ms without), the second one the residual norm (0.90 ms) --
             * 1.67 ms instead of 0.56 (prolongation) + 0.82 (4 passes) + 0.47 (norm).  Below the top level no norm
             * is formed and prolongation + 4 passes in two launches stays cheaper. */
            const bool sp = split_up_leg(iters, want_res);
            const int S = (sp && post && passes >= 2) ? 2 : passes >= 4 ? 4 : passes; /* 4, 2 or 0 */
            const bool last = passes - S == 0;
            /* the residual rides on a 2-pass launch; behind 4 passes it gets its own launch (the 5-stage
             * window leaves too few registers for a tile with a useful interior: measured 1.9 ms fused
             * against 0.85 + 0.76 ms split on a 513^3 level) */
            /* two passes + residual + restriction (the down-leg of V(1,1), the tail of V(3,3)'s): one launch from 130
             * points per side up, two below (k_sweep_fuse_rst2) */
            /* small levels (<= MG3D_FUSE_LEG_MAX points per side): the whole down-leg -- four passes, residual,
             * restriction -- as one launch of the two-rows-per-thread shape: it wastes three quarters of its rows
             * and saves a launch where launches are paid in latency, not in bytes */
            const bool leg4 = S == 4 && coarse != nullptr && !need_norm && want_res != 0 && l.g.N <= ctx->opt.v[MG3D_OPT_FUSE_LEG_MAX];
            const bool res = last && want_res != 0 && (S != 4 || leg4) &&
                             !(S == 2 && coarse != nullptr && !k_sweep_fuse_rst2(ctx->opt, l.g.N));
            const bool rst = res && coarse != nullptr;
            const bool with_pro = pro != nullptr && passes == 2 * iters; /* first launch only */
            int np;
            {
                StageScope kt(ctx, level, S == 4 ? MG3D_K_SWEEP4 : S == 2 ? (res ? MG3D_K_SWEEP2_RES : MG3D_K_SWEEP2)
                                                                          : MG3D_K_RESIDUAL, true);
                np = k_sweep(ctx->opt, l.g, (zero_in && passes == 2 * iters) ? nullptr : l.f[MG3D_U], l.f[MG3D_D], l.alt,
                             (res && want_res == 2 && !rst) ? l.f[MG3D_R] : nullptr,
                             (res && need_norm) ? ctx->partials : nullptr,
                             MG3D_MAX_PARTIALS, l.h, S, c1, res, s, 0, -1, rst ? &coarse->g : nullptr,
                             rst ? coarse->f[MG3D_D] : nullptr, -1, -1, with_pro ? &pro->g : nullptr,
                             with_pro ? pro->f[MG3D_U] : nullptr);
            }
            if (np < 0) /* nothing was launched: no buffer swap, no fold of partial sums that were never written */
                return fail(MG3D_ERR_STATE, "fused sweep: no kernel for %d colour passes%s on level %d", S,
                            res ? " + residual" : "", level);
            if (S > 0) {
                double *t = l.f[MG3D_U];
                l.f[MG3D_U] = l.alt;
                l.alt = t;
            }
            if (res) {
                if (need_norm)
                    k_fold(ctx->partials, np, ctx->sumsq + slot, s);
                done_res = true;
            }
            passes -= S;
        }
        return MG3D_OK;
    }
    const double hSq = l.h * l.h; /* mg_3d.h:644 */
    for (int it = 0; it < 2 * iters; it++) {
        StageScope kt(ctx, level, MG3D_K_COLOUR_PASS, true);
        k_smooth_color(l.g, l.f[MG3D_U], l.f[MG3D_D], hSq, c1 ^ (it & 1), s);
    }
    if (want_res) {
        const double invHsq = 1. / (l.h * l.h); /* mg_3d.h:797 */
        StageScope kt(ctx, level, MG3D_K_RESIDUAL, true);
        k_residual(l.g, l.f[MG3D_U], l.f[MG3D_D], invHsq, want_res == 2 ? l.f[MG3D_R] : nullptr, ctx->partials,
                   ctx->sumsq + slot, s);
    }
    return MG3D_OK;
}

/* can the prolongation ride on the first smoothing launch?  (needs a smoothing-only first launch) */
static bool pro_fusable(const mg3d_ctx *ctx, int iters, int want_res, int level)
{
    /* small levels: the two-rows-per-thread four-pass shape has the registers for the prolongation (k_sweep) */
    const int up_max = ctx->opt.v[MG3D_OPT_FUSE_UP_MAX] > ctx->opt.v[MG3D_OPT_FUSE_LEG_MAX] ? ctx->opt.v[MG3D_OPT_FUSE_UP_MAX]
                                                                                             : ctx->opt.v[MG3D_OPT_FUSE_LEG_MAX];
    const bool small = ctx->lv[level].g.N <= up_max && 2 * iters == 4 && want_res == 0;
    /* On a 4-pass first launch of a level above small_max it used to be SLOWER (the four-row shape spilt; round 2 measured
     * 1.48 ms against 0.85 + 0.56 at 513^3).  Since the prolongation is applied at the end of the step before (MG3D_PRO_LATE in
     * the kernel) that shape has 248 VGPRs and no scratch: the 257^3 level of the 513^3 problem takes 0.12 instead of
     * 0.058 + 0.112 ms, the cycle 2.18 -> 2.13 ms -- fuse_up_max now defaults to every level.  The 2-pass first launch of
     * a split stage takes it almost for free. */
    if (!ctx->fused || iters < 1)
        return false;
    const bool sp = split_up_leg(iters, want_res);
    if (!sp && !small)
        return false;
    const int first = (2 * iters >= 4 && !sp) ? 4 : 2;
    const bool first_has_res = want_res != 0 && first == 2 && 2 * iters == 2;
    return !first_has_res;
}

static int enqueue_smooth(mg3d_ctx *ctx, int level, int post, int iters)
{
    return enqueue_smooth_residual(ctx, level, post, iters, 0, 0);
}

static int enqueue_residual(mg3d_ctx *ctx, int level, int store, int slot)
{
    return enqueue_smooth_residual(ctx, level, 0, 0, store ? 2 : 1, slot);
}

extern "C" int mg3d_smooth(mg3d_ctx *ctx, int level, int post, int iters)
{
    CHK(mg3d_drop_carry(ctx));
    CHK(check_field_level(ctx, 0, level, "mg3d_smooth"));
    if (iters < 0)
        return fail(MG3D_ERR_ARG, "mg3d_smooth: negative iteration count");
    CHK(enqueue_smooth(ctx, level, post, iters));
    return launch_ok("mg3d_smooth");
}

extern "C" int mg3d_residual(mg3d_ctx *ctx, int level, int store, double *norm)
{
    CHK(mg3d_drop_carry(ctx));
    CHK(check_field_level(ctx, 0, level, "mg3d_residual"));
    CHK(enqueue_residual(ctx, level, store, 0));
    CHK(launch_ok("mg3d_residual"));
    return read_norm(ctx, 0, norm);
}

extern "C" int mg3d_smooth_residual(mg3d_ctx *ctx, int level, int post, int iters, int store, double *norm)
{
    CHK(mg3d_drop_carry(ctx));
    CHK(check_field_level(ctx, 0, level, "mg3d_smooth_residual"));
    if (iters < 0)
        return fail(MG3D_ERR_ARG, "mg3d_smooth_residual: negative iteration count");
    CHK(enqueue_smooth_residual(ctx, level, post, iters, store ? 2 : 1, 0));
    CHK(launch_ok("mg3d_smooth_residual"));
    return read_norm(ctx, 0, norm);
}

extern "C" int mg3d_smooth_restrict(mg3d_ctx *ctx, int level, int iters)
{
    CHK(mg3d_drop_carry(ctx));
    CHK(check_field_level(ctx, 0, level, "mg3d_smooth_restrict"));
    if (level < 1 || iters < 0)
        return fail(MG3D_ERR_ARG, "mg3d_smooth_restrict: bad level/iteration count");
    Level &lev = ctx->lv[level], &lc = ctx->lv[level - 1];
    CHK(enqueue_smooth_residual(ctx, level, 0, iters, 2, ctx->sumsq_slots - 1, ctx->fused ? &lc : nullptr, nullptr, false,
                                /* need_norm: only without the fused restriction, whose shapes have no norm */ !ctx->fused));
    k_restrict(lev.g, lev.f[MG3D_R], lc.g, lc.f[MG3D_D], ctx->stream, -1, -1, ctx->fused);
    return launch_ok("mg3d_smooth_restrict");
}

extern "C" int mg3d_restrict(mg3d_ctx *ctx, int level)
{
    CHK(mg3d_drop_carry(ctx));
    CHK(check_field_level(ctx, 0, level, "mg3d_restrict"));
    if (level < 1)
        return fail(MG3D_ERR_ARG, "mg3d_restrict: level 0 has no coarser level");
    k_restrict(ctx->lv[level].g, ctx->lv[level].f[MG3D_R], ctx->lv[level - 1].g, ctx->lv[level - 1].f[MG3D_D],
               ctx->stream);
    return launch_ok("mg3d_restrict");
}

extern "C" int mg3d_prolong(mg3d_ctx *ctx, int level)
{
    CHK(mg3d_drop_carry(ctx));
    CHK(check_field_level(ctx, 0, level, "mg3d_prolong"));
    if (level < 1)
        return fail(MG3D_ERR_ARG, "mg3d_prolong: level 0 has no coarser level");
    k_prolong(ctx->lv[level - 1].g, ctx->lv[level - 1].f[MG3D_U], ctx->lv[level].g, ctx->lv[level].f[MG3D_U],
              ctx->stream);
    return launch_ok("mg3d_prolong");
}

extern "C" int mg3d_coarse_solve(mg3d_ctx *ctx)
{
    CHK(mg3d_drop_carry(ctx));
    if (!ctx)
        return fail(MG3D_ERR_ARG, "mg3d_coarse_solve: NULL context");
    if (!ctx->have_lu)
        return fail(MG3D_ERR_STATE, "mg3d_coarse_solve: no coarse LU set (mg3d_ctx_build_coarse / mg3d_ctx_set_lu)");
    if (ctx->have_es)
        return fail(MG3D_ERR_STATE, "mg3d_coarse_solve: the context holds the mixed-boundary factor of mg3d_es_setup");
    k_lu_solve(ctx->lu, ctx->lu_in, ctx->lv[0].g, ctx->lv[0].f[MG3D_D], ctx->lv[0].f[MG3D_U], ctx->lu_work, ctx->stream);
    return launch_ok("mg3d_coarse_solve");
}

extern "C" int mg3d_l2norm(mg3d_ctx *ctx, int field, int level, double *norm)
{
    CHK(mg3d_drop_carry_keep(ctx)); /* (reads only) */
    CHK(check_field_level(ctx, field, level, "mg3d_l2norm"));
    k_sumsq(ctx->lv[level].g, ctx->lv[level].f[field], ctx->partials, ctx->sumsq, ctx->stream);
    CHK(launch_ok("mg3d_l2norm"));
    return read_norm(ctx, 0, norm);
}

/* ------------------------------------------------------------------ V-cycle */
/* vcycle, mg_3d.h:1242-1362, unrolled: descend q..1, solve level 0, ascend 1..q.
 * The squared post-smoothing norm of level q goes to sumsq[slot]. */
/* Carried cycles.  Back to back, V-cycle n ends with post-smoothing passes black, red, black, red and the residual norm,
 * and cycle n+1 begins with pre-smoothing passes red, black, red, black on the same u and d (mg_3d.h:728 / :657).  That
 * first red pass recomputes every red point from black neighbours no pass has touched since the red pass before it: the
 * same operands in the same expression, the same bits -- it is the identity.  What is left is ONE alternating run
 *      [prolongation] B R | B R <norm of cycle n> B R | B [residual, restriction]
 * which three launches cover instead of four: prolongation + 2 passes (as before), four passes with the norm tapped
 * after the second (k_sweep_tap: no stage of its own), one pass + residual + restriction -- 9 n w + 2 n_c w instead of
 * 11 n w + 2 n_c w compulsory bytes per cycle on the top level.  The tap launch reads u of cycle n and writes the other
 * buffer; cycle n's own u (the tapped state) is never written -- mg3d_drop_carry makes it from that input when it is
 * wanted.  mg3d_vcycles never ends a call in the carried state (its last cycle ends the ordinary way); mg3d_vcycle does.  Only V(2,2) from the finest level of a context with
 * at least three levels, fused sweeps, r not kept; MG3D_NO_CARRY=1 switches it off (tests compare both). */
/* One launch per leg ("two launches per level", round 4).  The alternating run of colour passes between two cycles
 *      [prolongation] B R B R <norm of cycle n> (R = identity) B R B [residual, restriction]
 * is cut at the norm instead: the up-leg is ONE launch (prolongation + four passes, k_sweep_leg_up) and the down-leg is ONE
 * launch (three passes + residual + restriction, k_sweep_leg_down) -- 6 n w + 2 n_c w compulsory bytes per cycle on the top
 * level instead of 9 n w + 2 n_c w.  No launch has a stage for the norm: the residual of the points the up-leg's last pass
 * (red) has just updated falls out of that pass's neighbour sums, the residual of the black points out of the sums the
 * next down-leg's first pass (black) forms before it updates them -- two runs of partial sums, folded into one norm.  The
 * cycle's own u is the up-leg's output: nothing is speculative inside mg3d_vcycles.  Behind a single mg3d_vcycle call
 * the next cycle's down-leg runs at once, into the alt buffers (u of the top level, d of the level below), so that the
 * norm is complete when the call returns; whatever the caller does instead of another cycle swaps back
 * (mg3d_drop_carry: no launch).  Same conditions as the carried cycles.  Default from 160 points per side (option
 * legs_min): same-box A/B by size, round 4 (ms per cycle, carried / legs): 385^3 1.23 / 1.30, 513^3 2.53 / 2.45, 641^3 4.99 / 4.80,
 * 769^3 11.31 / 10.76, 1025^3 20.65 / 16.79 (profiles/r04_legs_by_size.txt); option legs = 0 keeps the carried cycles. */
bool mg3d_can_legs(const mg3d_ctx *ctx, int q)
{
    if (!ctx->opt.v[MG3D_OPT_LEGS])
        return false;
    const mg3d_ctx *c = ctx;
    /* the conditions of the carried cycles, except their own switch and threshold */
    return c->fused && !c->keep_r && !c->have_es && c->iters == 2 && q == c->L - 1 && q >= 2 && c->lv[q].g.N >= c->opt.v[MG3D_OPT_LEGS_MIN] &&
           c->lv[q].g.N > 65 && (c->lv[q].g.nj & 1) != 0;
}

bool mg3d_can_carry(const mg3d_ctx *ctx, int q)
{
    if (!ctx->opt.v[MG3D_OPT_CARRY])
        return false;
    /* from 257^3 up: there the launch saved is bytes (257^3: +5 %, 513^3: +17 %, 1025^3: +16 % V-cycles/s); at 129^3 a
     * launch is pipeline fill and the plain schedule's lighter launches are 1 % ahead.  MG3D_CARRY_MIN=<points per side>
     * moves the threshold (the tests run 129^3 problems); never at 65^3 and below (the two launches only exist in
     * the four-rows-per-thread shapes) */
    const int n_min = ctx->opt.v[MG3D_OPT_CARRY_MIN];
    return ctx->fused && !ctx->keep_r && !ctx->have_es && ctx->iters == 2 && q == ctx->L - 1 && q >= 2 &&
           ctx->lv[q].g.N >= n_min && ctx->lv[q].g.N > 65 && (ctx->lv[q].g.nj & 1) != 0 && split_up_leg(2, 1) &&
           pro_fusable(ctx, 2, 1, q);
}

int mg3d_enqueue_vcycle(mg3d_ctx *ctx, int q, int slot, int carry_out)
{
    if (!ctx->have_lu)
        return fail(MG3D_ERR_STATE, "mg3d_vcycle: no coarse LU set (mg3d_ctx_build_coarse / mg3d_ctx_set_lu)");
    if (ctx->have_es)
        return fail(MG3D_ERR_STATE, "mg3d_vcycle: the context holds the mixed-boundary factor of mg3d_es_setup "
                                    "(mg3d_es_vcycles, or load the Dirichlet factor again)");
    hipStream_t s = ctx->stream;
    const int L = ctx->L;
    struct PhaseTick { /* sampled kernel timers (timing >= 4): full cycles are counted off, also on an early return */
        mg3d_ctx *c;
        bool full;
        ~PhaseTick()
        {
            if (full && c->timing >= 4)
                c->timing_phase = (c->timing_phase + 1) % (c->timing - 2);
        }
    } tick{ctx, q == L - 1};
    /* level 1 below the top of the cycle, small enough for one workgroup's LDS: two launches instead of five */
    const bool no_tiny = !ctx->opt.v[MG3D_OPT_TINY];
    const bool tiny = !no_tiny && ctx->fused && !ctx->keep_r && q >= 2 && ctx->iters >= 1 && k_tiny_fits(ctx->lv[1].g, ctx->lv[0].g);
    /* ... and the whole bottom of the cycle (level 1 down, the direct solve, level 1 up) as ONE launch when the reduced
     * factor exists (mg3d_tiny.hip, tiny_cycle_kernel); MG3D_NO_TINY_CYCLE=1 keeps the three launches (tests compare) */
    const bool no_cyc = !ctx->opt.v[MG3D_OPT_TINY_CYCLE];
    const bool tiny_cyc = tiny && !no_cyc && k_tiny_cycle_fits(ctx->lv[1].g, ctx->lv[0].g, ctx->lu, ctx->lu_in);
    const bool can_legs = mg3d_can_legs(ctx, q);
    const bool can_carry = !can_legs && mg3d_can_carry(ctx, q);
    if ((ctx->carried && !can_carry) || (ctx->legs_state != 0 && !can_legs)) /* e.g. MG3D_NO_CARRY set between two calls: finish the carried cycle, go on plainly */
        CHK(mg3d_drop_carry_keep(ctx));
    const bool carry_in = ctx->carried;
    ctx->carried = false;
    const int legs_in = ctx->legs_state;
    ctx->legs_state = 0;
    /* red_tail: the last thing that happened to u of the top level was the red pass that ends a cycle, and nothing has touched
     * u or d since (every entry point that could clears the flag through mg3d_drop_carry; not once a raw pointer is out) --
     * this cycle's first red pass is the identity then, ACROSS calls as inside one: its down-leg is the one launch of three
     * passes + residual + restriction (no norm half: the finished cycle formed its norm itself) instead of four passes, then
     * residual + restriction (0.84 against 0.67 + 0.50 ms at 513^3) */
    const bool red_in = q == L - 1 && legs_in == 0 && !carry_in && ctx->red_tail && can_legs && !ctx->raw_top;
    ctx->red_tail = false;
    double *const part_a = ctx->partials, *const part_b = ctx->partials + MG3D_MAX_PARTIALS / 2;
    for (int l = q; l >= 1; l--) {
        Level &lev = ctx->lv[l];
        /* (a cycle with no cycle in front of it takes the ordinary down-leg below -- four passes, then residual +
         * restriction: the one-launch form of THAT leg needs a six-plane window, spills 180 bytes at eight rows per thread
         * and took 2.1 ms against 0.66 + 0.49: profiles/r04_bench_kernel_stats_note.txt) */
        if (l == q && can_legs && (legs_in != 0 || red_in)) {
            Level &lc = ctx->lv[l - 1];
            {
                StageScope t(ctx, l, MG3D_ST_SMOOTH1);
                if (legs_in == 3) {
                    /* this cycle's down-leg ran behind the previous mg3d_vcycle call: its u is the top level's u already,
                     * its restricted residual sits in the coarser level's alt buffer */
                    double *t2 = lc.f[MG3D_D];
                    lc.f[MG3D_D] = lc.alt;
                    lc.alt = t2;
                    ctx->faces_dirty[l] = 1; /* the two buffers take turns as d: inject the faces into this one */
                } else {
                    StageScope kt(ctx, l, MG3D_K_LEG_DOWN, true);
                    /* behind another cycle: black, red, black (the first red pass is the identity) and the black half of
                     * that cycle's norm, + residual + restriction (:1282 + :1294 + :1310) */
                    const int np = k_sweep_leg_down(ctx->opt, lev.g, lev.f[MG3D_U], lev.f[MG3D_D], lev.alt, lc.g, lc.f[MG3D_D], lev.h,
                                                    3, red_in ? nullptr : part_b, MG3D_MAX_PARTIALS / 2, s);
                    if (np < 0)
                        return fail(MG3D_ERR_STATE, "one launch per leg: no kernel for the down-leg");
                    double *t2 = lev.f[MG3D_U];
                    lev.f[MG3D_U] = lev.alt;
                    lev.alt = t2;
                    if (!red_in) /* (behind a cycle of this call: the black half of its norm) */
                        k_fold2(part_a, ctx->legs_npa, part_b, np, ctx->sumsq + ctx->legs_slot, s);
                }
            }
            { StageScope t(ctx, l, MG3D_ST_RESIDUAL1); }
            StageScope t(ctx, l, MG3D_ST_RESTRICT);
            if (ctx->faces_dirty[l] || ctx->faces_always[l]) {
                StageScope kt(ctx, l, MG3D_K_RESTRICT, true);
                k_restrict(lev.g, lev.f[MG3D_R], lc.g, lc.f[MG3D_D], s, -1, -1, true);
                ctx->faces_dirty[l] = 0;
            }
            continue;
        }
        if (l == q && carry_in) {
            { /* the one pre-smoothing pass that is left (black) + residual + restriction (:1282 + :1294 + :1310) */
                StageScope t(ctx, l, MG3D_ST_SMOOTH1);
                StageScope kt(ctx, l, MG3D_K_SWEEP1_RESTRICT, true);
                const int np = k_sweep(ctx->opt, lev.g, lev.f[MG3D_U], lev.f[MG3D_D], lev.alt, nullptr, nullptr, MG3D_MAX_PARTIALS, lev.h,
                                       1, 0, true, s, 0, -1, &ctx->lv[l - 1].g, ctx->lv[l - 1].f[MG3D_D]);
                if (np < 0)
                    return fail(MG3D_ERR_STATE, "carried cycle: no kernel for one pass + residual + restriction");
                double *t2 = lev.f[MG3D_U];
                lev.f[MG3D_U] = lev.alt;
                lev.alt = t2;
            }
            { StageScope t(ctx, l, MG3D_ST_RESIDUAL1); }
            StageScope t(ctx, l, MG3D_ST_RESTRICT);
            if (ctx->faces_dirty[l] || ctx->faces_always[l]) {
                StageScope kt(ctx, l, MG3D_K_RESTRICT, true);
                k_restrict(lev.g, lev.f[MG3D_R], ctx->lv[l - 1].g, ctx->lv[l - 1].f[MG3D_D], s, -1, -1, true);
                ctx->faces_dirty[l] = 0;
            }
            continue;
        }
        if (l == 1 && tiny_cyc) {
            {
                StageScope t(ctx, l, MG3D_ST_SMOOTH1);
                StageScope kt(ctx, l, MG3D_K_SWEEP4, true);
                k_tiny_cycle(lev.g, lev.f[MG3D_U], lev.f[MG3D_D], lev.f[MG3D_R], ctx->lv[0].g, ctx->lv[0].f[MG3D_D],
                             ctx->lv[0].f[MG3D_U], ctx->lu, ctx->lu_in, lev.h, ctx->iters, s); /* :1258 ... :1341 of levels 1, 0 */
            }
            { StageScope t(ctx, l, MG3D_ST_RESIDUAL1); } /* inside the launch above: counted, ~0 s */
            { StageScope t(ctx, l, MG3D_ST_RESTRICT); }
            ctx->faces_dirty[l] = 0;
            continue;
        }
        if (l == 1 && tiny) {
            {
                StageScope t(ctx, l, MG3D_ST_SMOOTH1);
                StageScope kt(ctx, l, MG3D_K_SWEEP4, true);
                k_tiny_down(lev.g, lev.f[MG3D_U], lev.f[MG3D_D], lev.f[MG3D_R], ctx->lv[0].g, ctx->lv[0].f[MG3D_D], lev.h,
                            ctx->iters, s); /* :1258 + :1282 + :1294 + :1310 */
            }
            { StageScope t(ctx, l, MG3D_ST_RESIDUAL1); } /* inside the launch above: counted, ~0 s */
            { StageScope t(ctx, l, MG3D_ST_RESTRICT); }
            ctx->faces_dirty[l] = 0; /* the launch injects the faces itself */
            continue;
        }
        /* :1258-1259: the zero initial guess of a coarser level; with the fused sweep the first launch simply
         * does not read u (and writes every plane of the other buffer), so no memset is needed */
        const bool zero_in = l < L - 1 && ctx->fused && ctx->iters > 0;
        if (l < L - 1 && !zero_in)
            (void)hipMemsetAsync(lev.f[MG3D_U], 0, lev.elems * sizeof(double), s);
        if (ctx->fused) { /* pre-smoother and residual in one pass over the level (:1282 + :1294) */
            {
                StageScope t(ctx, l, MG3D_ST_SMOOTH1);
                CHK(enqueue_smooth_residual(ctx, l, 0, ctx->iters, 2, ctx->sumsq_slots - 1,
                                            ctx->keep_r ? nullptr : &ctx->lv[l - 1], nullptr, zero_in, false));
            }
            StageScope t(ctx, l, MG3D_ST_RESIDUAL1); /* fused into the launch above: counted, ~0 s */
        } else {
            {
                StageScope t(ctx, l, MG3D_ST_SMOOTH1);
                CHK(enqueue_smooth(ctx, l, 0, ctx->iters)); /* :1282 */
            }
            StageScope t(ctx, l, MG3D_ST_RESIDUAL1);
            CHK(enqueue_residual(ctx, l, 1, ctx->sumsq_slots - 1)); /* :1294 (norm discarded) */
        }
        {
            StageScope t(ctx, l, MG3D_ST_RESTRICT);
            /* :1310; when the interior was restricted on the fly only the face injection (:879-958) is left, and
             * that only has something new to copy after r of this level or d of the coarser one was written from
             * outside the cycle (faces_dirty) */
            const bool faces_only = ctx->fused && !ctx->keep_r;
            if (!faces_only || ctx->faces_dirty[l] || ctx->faces_always[l]) {
                StageScope kt(ctx, l, MG3D_K_RESTRICT, true);
                k_restrict(lev.g, lev.f[MG3D_R], ctx->lv[l - 1].g, ctx->lv[l - 1].f[MG3D_D], s, -1, -1, faces_only);
                ctx->faces_dirty[l] = 0;
            }
        }
    }
    if (tiny_cyc) {
        StageScope t(ctx, 0, MG3D_ST_RECURSE); /* inside the launch above */
    } else {
        Level &l0 = ctx->lv[0];
        if (0 < L - 1)
            (void)hipMemsetAsync(l0.f[MG3D_U], 0, l0.elems * sizeof(double), s);
        StageScope t(ctx, 0, MG3D_ST_RECURSE);
        StageScope kt(ctx, 0, MG3D_K_COARSE_SOLVE, true);
        k_lu_solve(ctx->lu, ctx->lu_in, l0.g, l0.f[MG3D_D], l0.f[MG3D_U], ctx->lu_work, s); /* :1270 */
    }
    for (int l = 1; l <= q; l++) {
        Level &lev = ctx->lv[l];
        if (l == 1 && tiny_cyc) {
            { StageScope t(ctx, l, MG3D_ST_PROLONG); }
            { StageScope t(ctx, l, MG3D_ST_SMOOTH2); }
            { StageScope t(ctx, l, MG3D_ST_RESIDUAL2); }
            continue;
        }
        if (l == 1 && tiny) {
            { StageScope t(ctx, l, MG3D_ST_PROLONG); } /* inside the launch below */
            {
                StageScope t(ctx, l, MG3D_ST_SMOOTH2);
                StageScope kt(ctx, l, MG3D_K_SWEEP4, true);
                k_tiny_up(lev.g, lev.f[MG3D_U], lev.f[MG3D_D], ctx->lv[0].g, ctx->lv[0].f[MG3D_U], lev.h, ctx->iters,
                          s); /* :1331 + :1341 */
            }
            { StageScope t(ctx, l, MG3D_ST_RESIDUAL2); }
            continue;
        }
        if (l == q && can_legs) {
            Level &lc = ctx->lv[l - 1];
            { StageScope t(ctx, l, MG3D_ST_PROLONG); } /* :1331, folded into the launch below */
            {
                StageScope t(ctx, l, MG3D_ST_SMOOTH2);
                int npa;
                {
                    StageScope kt(ctx, l, MG3D_K_LEG_UP, true);
                    /* prolongation + black, red, black, red (:1331 + :1341); with another cycle behind it, the red half of
                     * the norm (:1354) from the last pass's sums */
                    npa = k_sweep_leg_up(ctx->opt, lev.g, lev.f[MG3D_U], lev.f[MG3D_D], lev.alt, lc.g, lc.f[MG3D_U], lev.h,
                                         carry_out ? part_a : nullptr, MG3D_MAX_PARTIALS / 2, s);
                }
                if (npa < 0)
                    return fail(MG3D_ERR_STATE, "one launch per leg: no kernel for the up-leg");
                double *t2 = lev.f[MG3D_U];
                lev.f[MG3D_U] = lev.alt;
                lev.alt = t2;
                if (carry_out == 1) { /* the next cycle of this mg3d_vcycles call completes the norm */
                    ctx->legs_state = 2;
                    ctx->legs_slot = slot;
                    ctx->legs_npa = npa;
                } else if (carry_out == 2) {
                    /* mg3d_vcycle: the next cycle's down-leg now, into the alt buffers (see mg3d_can_legs) */
                    StageScope kt(ctx, l, MG3D_K_LEG_DOWN, true);
                    const int npb = k_sweep_leg_down(ctx->opt, lev.g, lev.f[MG3D_U], lev.f[MG3D_D], lev.alt, lc.g, lc.alt, lev.h, 3, part_b,
                                                     MG3D_MAX_PARTIALS / 2, s);
                    if (npb < 0)
                        return fail(MG3D_ERR_STATE, "one launch per leg: no kernel for the down-leg");
                    double *t3 = lev.f[MG3D_U];
                    lev.f[MG3D_U] = lev.alt;
                    lev.alt = t3;
                    k_fold2(part_a, npa, part_b, npb, ctx->sumsq + slot, s);
                    ctx->legs_state = 3;
                }
            }
            {
                StageScope t(ctx, l, MG3D_ST_RESIDUAL2);
                if (!carry_out)
                    CHK(enqueue_residual(ctx, l, 0, slot)); /* :1354, the last cycle of a call: a launch of its own */
            }
            continue;
        }
        if (l == q && carry_out && can_carry) {
            { StageScope t(ctx, l, MG3D_ST_PROLONG); } /* :1331, folded into the launch below */
            {
                StageScope t(ctx, l, MG3D_ST_SMOOTH2);
                /* prolongation + the first two post-smoothing passes: the launch the ordinary cycle starts its up-leg with */
                CHK(enqueue_smooth_residual(ctx, l, 1, 1, 0, slot, nullptr, &ctx->lv[l - 1]));
                /* black, red (:1341, second half) <norm, :1354> black, red (:1282 of the next cycle, its first red pass
                 * being the identity) */
                StageScope kt(ctx, l, MG3D_K_SWEEP4_NORM, true);
                const int np = k_sweep_tap(ctx->opt, lev.g, lev.f[MG3D_U], lev.f[MG3D_D], lev.alt, ctx->partials, MG3D_MAX_PARTIALS,
                                           lev.h, 0, s);
                if (np < 0)
                    return fail(MG3D_ERR_STATE, "carried cycle: no kernel for four passes + norm tap");
                double *t2 = lev.f[MG3D_U];
                lev.f[MG3D_U] = lev.alt;
                lev.alt = t2;
                k_fold(ctx->partials, np, ctx->sumsq + slot, s);
                ctx->carried = true;
            }
            { StageScope t(ctx, l, MG3D_ST_RESIDUAL2); }
            continue;
        }
        const int want_norm = l == q ? 1 : 0;
        const bool pro = pro_fusable(ctx, ctx->iters, want_norm, l);
        {
            StageScope t(ctx, l, MG3D_ST_PROLONG); /* :1331; ~0 s when folded into the smoother's loads */
            if (!pro) {
                StageScope kt(ctx, l, MG3D_K_PROLONG, true);
                k_prolong(ctx->lv[l - 1].g, ctx->lv[l - 1].f[MG3D_U], lev.g, lev.f[MG3D_U], s);
            }
        }
        if (ctx->fused) { /* (prolongation,) post-smoother and residual norm (:1331 + :1341 + :1354) */
            {
                StageScope t(ctx, l, MG3D_ST_SMOOTH2);
                /* the norm of a level below the top one is computed and dropped by the reference (:1320
                 * ignores the recursive call's value): skip it, nothing observable changes */
                CHK(enqueue_smooth_residual(ctx, l, 1, ctx->iters, want_norm, slot, nullptr,
                                            pro ? &ctx->lv[l - 1] : nullptr));
            }
            StageScope t(ctx, l, MG3D_ST_RESIDUAL2); /* fused into the launch above: counted, ~0 s */
        } else {
            {
                StageScope t(ctx, l, MG3D_ST_SMOOTH2);
                CHK(enqueue_smooth(ctx, l, 1, ctx->iters)); /* :1341 */
            }
            StageScope t(ctx, l, MG3D_ST_RESIDUAL2);
            if (l == q)
                CHK(enqueue_residual(ctx, l, 0, slot)); /* :1354; below the top level the value is dropped (:1320) */
        }
    }
    /* a whole V(2,2) cycle from the top level has ended with its last red pass and nothing runs ahead: see red_in above */
    if (q == L - 1 && ctx->legs_state == 0 && !ctx->carried && ctx->iters == 2 && ctx->fused)
        ctx->red_tail = true;
    return launch_ok("mg3d_vcycle");
}

extern "C" int mg3d_vcycle(mg3d_ctx *ctx, int level, double *norm)
{
    CHK(check_field_level(ctx, 0, level, "mg3d_vcycle"));
    if (level == 0) { /* q == 0: direct solve, returns 0 (mg_3d.h:1262-1277) */
        CHK(mg3d_coarse_solve(ctx));
        if (norm)
            *norm = 0.;
        return MG3D_OK;
    }
    if (level != ctx->L - 1)
        CHK(mg3d_drop_carry(ctx)); /* (a cycle from a lower level rewrites the levels below the top one only -- but it ends whatever ran ahead) */
    /* One cycle per call is how the reference's solve loop runs (SolverLinSolve, mg_3d.h:1415-1420): the call ends with
     * the launch that also begins the NEXT cycle -- speculatively; whatever the caller does instead of another cycle
     * first puts the finished cycle's own u back (mg3d_drop_carry), and the norm returned is this cycle's either way.
     * Not once a raw pointer to u or d of the top level is out (mg3d_ctx_touched). */
    const int rc = mg3d_enqueue_vcycle(ctx, level, 0, ctx->raw_top ? 0 : 2);
    if (rc != MG3D_OK) {
        (void)mg3d_drop_carry(ctx); /* the first error is the one reported */
        return rc;
    }
    return read_norm(ctx, 0, norm);
}

extern "C" int mg3d_vcycles(mg3d_ctx *ctx, int count, double *norms)
{
    if (!ctx || count < 0)
        return fail(MG3D_ERR_ARG, "mg3d_vcycles: bad arguments");
    const int q = ctx->L - 1;
    if (q == 0) {
        for (int c = 0; c < count; c++) {
            CHK(mg3d_coarse_solve(ctx));
            if (norms)
                norms[c] = 0.;
        }
        return mg3d_sync(ctx);
    }
    const int batch = ctx->sumsq_slots - 1;
    if (count == 0) /* (behind a single mg3d_vcycle call the first cycle continues from the carried state) */
        CHK(mg3d_drop_carry_keep(ctx));
    struct Guard { /* an error return must not leave u of the top level a few passes into a cycle nobody asked for */
        mg3d_ctx *c;
        ~Guard() { (void)mg3d_drop_carry_keep(c); }
    } guard{ctx};
    for (int done = 0; done < count;) {
        const int nb = (count - done < batch) ? count - done : batch;
        for (int c = 0; c < nb; c++)
            /* every cycle but the last carries into the next; one launch per leg: not across a batch's norm read-back */
            CHK(mg3d_enqueue_vcycle(ctx, q, c, (done + c + 1 < count && (c + 1 < nb || !mg3d_can_legs(ctx, q))) ? 1 : 0));
        HIPCHK(hipMemcpyAsync(ctx->h_sumsq, ctx->sumsq, nb * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        resolve_timers(ctx);
        if (norms)
            for (int c = 0; c < nb; c++)
                norms[done + c] = sqrt(ctx->h_sumsq[c]);
        done += nb;
    }
    return MG3D_OK;
}

/* -------------------------------------------------------------------- FMG */
extern "C" int mg3d_fill_boundary(mg3d_ctx *ctx, int field, int level)
{
    CHK(mg3d_drop_carry(ctx));
    CHK(check_field_level(ctx, field, level, "mg3d_fill_boundary"));
    k_fill_boundary(ctx->lv[level].g, ctx->lv[level].f[field], ctx->lv[level].h, ctx->stream);
    mg3d_ctx_touched(ctx, field, level);
    return launch_ok("mg3d_fill_boundary");
}

/* SolverFMGInitialize, mg_dirichlet_analytic.c:771-806 */
extern "C" int mg3d_fmg_initialize(mg3d_ctx *ctx)
{
    CHK(mg3d_drop_carry(ctx));
    if (!ctx)
        return fail(MG3D_ERR_ARG, "mg3d_fmg_initialize: NULL context");
    if (!ctx->have_lu)
        return fail(MG3D_ERR_STATE, "mg3d_fmg_initialize: no coarse LU set");
    if (ctx->have_es)
        return fail(MG3D_ERR_STATE, "mg3d_fmg_initialize: the context holds the mixed-boundary factor of mg3d_es_setup");
    hipStream_t s = ctx->stream;
    k_fill_boundary(ctx->lv[0].g, ctx->lv[0].f[MG3D_U], ctx->lv[0].h, s);                                  /* :780 */
    k_lu_solve(ctx->lu, ctx->lu_in, ctx->lv[0].g, ctx->lv[0].f[MG3D_D], ctx->lv[0].f[MG3D_U], ctx->lu_work, s);         /* :783 */
    for (int l = 1; l < ctx->L; l++) {
        Level &lev = ctx->lv[l], &lc = ctx->lv[l - 1];
        k_prolong(lc.g, lc.f[MG3D_U], lev.g, lev.f[MG3D_U], s);                                              /* :795 */
        k_fill_boundary(lev.g, lev.f[MG3D_U], lev.h, s);                                                     /* :798 */
        (void)hipMemsetAsync(lc.f[MG3D_U], 0, lc.elems * sizeof(double), s);                                 /* :801 */
        CHK(mg3d_enqueue_vcycle(ctx, l, ctx->sumsq_slots - 1));                                              /* :804 */
    }
    return launch_ok("mg3d_fmg_initialize");
}

/* ------------------------------------------------------------------- timing */
extern "C" int mg3d_timing_enable(mg3d_ctx *ctx, int on)
{
    if (!ctx)
        return fail(MG3D_ERR_ARG, "mg3d_timing_enable: NULL context");
    ctx->timing = (on >= 1 && on <= 64) ? on : 0;
    ctx->timing_phase = 0;
    return MG3D_OK;
}

extern "C" int mg3d_timing_reset(mg3d_ctx *ctx) /* resetTimingInfo, timing_info.h:34-38 */
{
    if (!ctx)
        return fail(MG3D_ERR_ARG, "mg3d_timing_reset: NULL context");
    if (!ctx->pending.empty()) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        resolve_timers(ctx);
    }
    for (auto &t : ctx->timers)
        t = StageTimer{0, 0.};
    return MG3D_OK;
}

extern "C" int mg3d_kernel_time_get(mg3d_ctx *ctx, int level, int kernel, int *num_launches, double *seconds)
{
    if (!ctx || level < 0 || level >= ctx->L || kernel < 0 || kernel >= MG3D_NUM_KERNELS)
        return fail(MG3D_ERR_ARG, "mg3d_kernel_time_get: bad level/kernel");
    if (!ctx->pending.empty()) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        resolve_timers(ctx);
    }
    const StageTimer &t = ctx->timers[(size_t)ctx->L * MG3D_NUM_STAGES + (size_t)level * MG3D_NUM_KERNELS + kernel];
    if (num_launches)
        *num_launches = t.calls;
    if (seconds)
        *seconds = t.seconds;
    return MG3D_OK;
}

extern "C" int mg3d_timing_get(mg3d_ctx *ctx, int level, int stage, int *num_calls, double *seconds)
{
    if (!ctx || level < 0 || level >= ctx->L || stage < 0 || stage >= MG3D_NUM_STAGES)
        return fail(MG3D_ERR_ARG, "mg3d_timing_get: bad level/stage");
    if (!ctx->pending.empty()) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        resolve_timers(ctx);
    }
    const StageTimer &t = ctx->timers[(size_t)level * MG3D_NUM_STAGES + stage];
    if (num_calls)
        *num_calls = t.calls;
    if (seconds)
        *seconds = t.seconds;
    return MG3D_OK;
}

/* page-locked host arrays for the facade's finest u and d (mg_3d.h:275-293 hands their addresses to the caller,
 * and they cross PCIe around every solve: pageable memory downloads at 15 GB/s, page-locked at PCIe rate) */
extern "C" int mg3d_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0)
        return fail(MG3D_ERR_ARG, "mg3d_host_alloc: bad arguments");
    if (mg3d_device_count() <= 0)
        return fail(MG3D_ERR_NO_DEVICE, "no HIP device available: libmg3d has no CPU fallback");
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess)
        return fail(MG3D_ERR_ALLOC, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    memset(p, 0, bytes);
    *out = p;
    return MG3D_OK;
}

extern "C" int mg3d_host_free(void *p)
{
    if (p && hipHostFree(p) != hipSuccess)
        return fail(MG3D_ERR_HIP, "hipHostFree failed");
    return MG3D_OK;
}

/* ----------------------------------------------- host-pointer operator forms */
struct ScratchCtx { /* a one- or two-level context for a single host-pointer call */
    mg3d_ctx *ctx = nullptr;
    ~ScratchCtx() { mg3d_ctx_destroy(ctx); }
};

static int scratch_create(const int *n, const double *h, int L, ScratchCtx &sc)
{
    CHK(ctx_create_sizes(n, h, L, 0, &sc.ctx));
    sc.ctx->c = n[0];
    sc.ctx->length = 0.;
    return MG3D_OK;
}

extern "C" int mg3d_host_smooth(double *v, const double *d, int N, double h, int iters, int post)
{
    if (!v || !d || N < 1 || iters < 0)
        return fail(MG3D_ERR_ARG, "mg3d_host_smooth: bad arguments");
    ScratchCtx sc;
    CHK(scratch_create(&N, &h, 1, sc));
    CHK(mg3d_upload(sc.ctx, MG3D_U, 0, v));
    CHK(mg3d_upload(sc.ctx, MG3D_D, 0, d));
    CHK(mg3d_smooth(sc.ctx, 0, post, iters));
    return mg3d_download(sc.ctx, MG3D_U, 0, v);
}

extern "C" int mg3d_host_residual(const double *v, const double *d, int N, double h, double *res, double *norm)
{
    if (!v || !d || N < 1)
        return fail(MG3D_ERR_ARG, "mg3d_host_residual: bad arguments");
    ScratchCtx sc;
    CHK(scratch_create(&N, &h, 1, sc));
    CHK(mg3d_upload(sc.ctx, MG3D_U, 0, v));
    CHK(mg3d_upload(sc.ctx, MG3D_D, 0, d));
    if (res) /* only the interior of res is written (mg_3d.h:824-825): keep the caller's boundary values */
        CHK(mg3d_upload(sc.ctx, MG3D_R, 0, res));
    CHK(mg3d_residual(sc.ctx, 0, res != nullptr, norm));
    if (res)
        CHK(mg3d_download(sc.ctx, MG3D_R, 0, res));
    return MG3D_OK;
}

extern "C" int mg3d_host_restrict(const double *r, int Nf, double *dc, int Nc)
{
    if (!r || !dc || Nc < 1 || Nf != 2 * Nc - 1)
        return fail(MG3D_ERR_ARG, "mg3d_host_restrict: need Nf == 2*Nc-1 (got %d, %d)", Nf, Nc);
    const int n[2] = {Nc, Nf};
    const double h[2] = {2., 1.};
    ScratchCtx sc;
    CHK(scratch_create(n, h, 2, sc));
    CHK(mg3d_upload(sc.ctx, MG3D_R, 1, r));
    CHK(mg3d_restrict(sc.ctx, 1));
    return mg3d_download(sc.ctx, MG3D_D, 0, dc);
}

extern "C" int mg3d_host_prolong(const double *ec, int Nc, double *ef, int Nf)
{
    if (!ec || !ef || Nc < 1 || Nf != 2 * Nc - 1)
        return fail(MG3D_ERR_ARG, "mg3d_host_prolong: need Nf == 2*Nc-1 (got %d, %d)", Nf, Nc);
    const int n[2] = {Nc, Nf};
    const double h[2] = {2., 1.};
    ScratchCtx sc;
    CHK(scratch_create(n, h, 2, sc));
    CHK(mg3d_upload(sc.ctx, MG3D_U, 0, ec));
    CHK(mg3d_upload(sc.ctx, MG3D_U, 1, ef));
    CHK(mg3d_prolong(sc.ctx, 1));
    return mg3d_download(sc.ctx, MG3D_U, 1, ef);
}

extern "C" int mg3d_host_lu_solve(const double *LU, int n, const double *b, double *x)
{
    if (!LU || !b || !x || n < 1)
        return fail(MG3D_ERR_ARG, "mg3d_host_lu_solve: bad arguments");
    /* treat the n-vector as an n x 1 x 1 "grid" so the padded-layout solve kernel applies */
    CHK(require_device());
    ScratchCtx sc;
    mg3d_ctx *ctx = sc.ctx = ctx_new(1, 0);
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    CHK(install_lu(ctx, LU, n, 4 * (size_t)n));
    double *db = ctx->lu_work + 2 * (size_t)n, *dx = db + n;
    HIPCHK(hipMemcpy(db, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    Geom g;
    g.N = g.ni = n;
    g.nj = g.nk = 1;
    g.pitch = 1;
    g.plane = 1;
    g.ig0 = 0;
    k_lu_solve(ctx->lu, ctx->lu_in, g, db, dx, ctx->lu_work, ctx->stream);
    CHK(launch_ok("mg3d_host_lu_solve"));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(x, dx, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return MG3D_OK;
}

extern "C" int mg3d_host_vcycle(double **u, double **f, double **res, double h, int q, int num_levels, int iters,
                                int N, const double *LU, double *norm, int *stage_calls, double *stage_seconds)
{
    if (!u || !f || !res || !LU || q < 0 || q >= num_levels || N < 3 || iters < 0)
        return fail(MG3D_ERR_ARG, "mg3d_host_vcycle: bad arguments");
    /* level sizes below q: N_coarse = (N+1)/2 (mg_3d.h:1302), h_coarse = 2h (mg_3d.h:1303) */
    std::vector<int> n(q + 1);
    std::vector<double> hh(q + 1);
    n[q] = N;
    hh[q] = h;
    for (int l = q - 1; l >= 0; l--) {
        if (n[l + 1] < 3 || (n[l + 1] & 1) == 0)
            return fail(MG3D_ERR_ARG, "mg3d_host_vcycle: level %d has %d points per side, cannot coarsen", l + 1,
                        n[l + 1]);
        n[l] = (n[l + 1] + 1) / 2;
        hh[l] = 2 * hh[l + 1];
    }
    ScratchCtx sc;
    CHK(scratch_create(n.data(), hh.data(), q + 1, sc));
    mg3d_ctx *ctx = sc.ctx;
    /* the memset of mg_3d.h:1258 is skipped only on the caller's finest level (q == numLevels-1):
     * emulate by telling the context how many levels the caller's hierarchy has */
    ctx->iters = iters;
    ctx->keep_r = true; /* the caller owns res[] and may read it */
    ctx->timing = (stage_calls || stage_seconds) ? 1 : 0;
    CHK(mg3d_ctx_set_lu(ctx, LU));
    CHK(mg3d_upload(ctx, MG3D_U, q, u[q]));
    CHK(mg3d_upload(ctx, MG3D_D, q, f[q]));
    /* r is written on interiors only: start from the caller's arrays so untouched entries survive */
    for (int l = 1; l <= q; l++)
        CHK(mg3d_upload(ctx, MG3D_R, l, res[l]));
    if (q == 0) {
        {
            StageScope t(ctx, 0, MG3D_ST_RECURSE);
            CHK(mg3d_coarse_solve(ctx));
        }
        if (norm)
            *norm = 0.;
        CHK(mg3d_download(ctx, MG3D_U, 0, u[0]));
        if (stage_calls)
            stage_calls[MG3D_ST_RECURSE] += ctx->timers[MG3D_ST_RECURSE].calls;
        if (stage_seconds)
            stage_seconds[MG3D_ST_RECURSE] += ctx->timers[MG3D_ST_RECURSE].seconds;
        return MG3D_OK;
    }
    if (q < num_levels - 1) /* not the caller's finest level: its guess is zeroed first (:1258) */
        CHK(mg3d_zero(ctx, MG3D_U, q));
    CHK(mg3d_enqueue_vcycle(ctx, q, 0));
    double nrm = 0.;
    CHK(read_norm(ctx, 0, &nrm)); /* synchronises and resolves the stage timers */
    if (norm)
        *norm = nrm;
    for (size_t t = 0; t < (size_t)ctx->L * MG3D_NUM_STAGES; t++) {
        if (stage_calls)
            stage_calls[t] += ctx->timers[t].calls;
        if (stage_seconds)
            stage_seconds[t] += ctx->timers[t].seconds;
    }
    for (int l = 0; l <= q; l++) {
        CHK(mg3d_download(ctx, MG3D_U, l, u[l]));
        if (l < q)
            CHK(mg3d_download(ctx, MG3D_D, l, f[l]));
        if (l >= 1)
            CHK(mg3d_download(ctx, MG3D_R, l, res[l]));
    }
    return MG3D_OK;
}
