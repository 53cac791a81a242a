/*
 * mg3d_f32_dist.hip -- the single-precision / damped-Jacobi / F-cycle variant (BASELINE configs[4]: 1025^3 on 8 GPUs)
 * on i-slabs, one process per GPU.  PARITY UNPINNED like mg3d_f32.hip (no reference implementation exists); what is
 * established is that the slab decomposition changes no bit of the single-domain variant.
 *
 * Same partition and the same schedule as the double-precision slab path (mg3d_dist.hip; the reference's OpenMP path
 * splits every operator over i, mg_3d.h:658-659): rank r owns the global planes [b_l(r), b_l(r+1)) of every level with
 * at least 16 planes per rank, b_{l+1} = 2 b_l, smaller levels and the direct solve are replicated after one
 * all-gather of the restricted right-hand side.  A Jacobi sweep is out of place and uses up ONE halo plane per
 * sweep (a red-black sweep uses two), the residual one more, restriction reaches one fine plane beyond the owned
 * ones: H = nu + 2 halo planes per side.
 *
 * Per distributed level l and cycle:
 *   [u_l halos, top level only: refreshed at the start of the cycle -- below the top level the guess is zero]
 *   nu sweeps on every local plane -> residual + restriction of the OWNED coarse planes
 *   d_(l-1): halo exchange (or, into the first replicated level, the all-gather)
 *   u_l halos refreshed (the post-smoother will use them up again)
 *   ... coarser levels ...
 *   u_(l-1) halos refreshed -> prolongation folded into the first paired sweep on every local plane -> nu sweeps
 *   (+ the residual norm over the owned planes at the top level; per-rank sums gathered and added in rank order).
 * F-cycle start (mg_dirichlet_analytic.c:771-806) level by level with the same exchanges.
 * Transports: RCCL send/recv (ncclFloat) or loopback (virtual ranks of one process, device copies), as mg3d_dist.hip.
 */
#include "mg3d_f32_int.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <rccl/rccl.h>

#include "mg3d_plan.h"

#define fail mg3d_fail
#define HIPCHK(call)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(MG3D_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                      \
    } while (0)
#define NCCLCHK(call)                                                                                    \
    do {                                                                                                 \
        ncclResult_t e_ = (call);                                                                        \
        if (e_ != ncclSuccess)                                                                           \
            return fail(MG3D_ERR_HIP, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                       \
    } while (0)
#define CHK(call)           \
    do {                    \
        int rc_ = (call);   \
        if (rc_ != MG3D_OK) \
            return rc_;     \
    } while (0)

struct mg3d32_dist {
    int c, L, nu, P, ld, H;
    bool loopback;
    int device;
    ncclComm_t comm;
    bool have_comm;
    hipStream_t stream;             /* every operation of every local rank is ordered on this one stream */
    std::vector<mg3d32_ctx *> rs;   /* local ranks: all P (loopback) or this process's one */
    std::vector<int> rank_of;       /* their global rank numbers */
    double *gather, *d_norms, *h_norms;
    int norm_slots;
    /* the exchange plans of one V-cycle from level q, with / without the norm phase, per local rank (mg3d32_dist_plan):
     * plans[(q - ld) * 2 + want_norm][local rank]; the transports execute them entry by entry */
    std::vector<std::vector<Plan>> plans;
    const std::vector<Plan> *cur; /* the cycle being enqueued */
    int phase;
};

/* One V-cycle from distributed level q (mg3d32_dist_plan, include/mg3d.h): u_q halos first (they were last refreshed
 * before its owned planes changed), then per level u for the way up and the coarser right-hand side, the correction's
 * halos on the way up, the norm.  Host arithmetic only. */
static int build_plan32(Plan &pl, int c, int L, int P, int nu, int rank, int q, int want_norm)
{
    if (c < 3 || L < 2 || nu < 1 || P < 1 || rank < 0 || rank >= P || q >= L)
        return MG3D_ERR_ARG;
    PlanGeom G{c, L, P, nu, mg3d32_slab_halo(nu), 0, 32};
    G.ld = mg3d_slab_first_level(c, L, P, G.H);
    if (G.ld >= L || q < G.ld)
        return MG3D_ERR_ARG;
    pl = Plan();
    plan_halo(pl, G, MG3D_XK_HALO_U_NEXT, MG3D_U, q, rank, 0, 0);
    for (int l = q; l >= G.ld; l--) {
        plan_halo(pl, G, MG3D_XK_HALO_U_DOWN, MG3D_U, l, rank, 0, 0);
        if (l - 1 >= G.ld)
            plan_halo(pl, G, MG3D_XK_HALO_D, MG3D_D, l - 1, rank, 0, 0);
        else
            plan_rhs_allgather(pl, G);
    }
    for (int l = G.ld + 1; l <= q; l++)
        plan_halo(pl, G, MG3D_XK_HALO_U_UP, MG3D_U, l - 1, rank, 0, 0);
    if (want_norm)
        plan_norm(pl, G, q, rank);
    pl.begin.push_back((int)pl.e.size());
    return MG3D_OK;
}

/* the fp32 variant's plan: the V-cycle from level q (the F-cycle start runs them from every level); want_norm = 0
 * leaves the norm phase out (cycles inside the F-cycle start) */
extern "C" int mg3d32_dist_plan(int coarse_pts, int num_levels, int nranks, int smooth_iters, int rank, int q, int want_norm,
                                mg3d_xfer *out, int max_entries)
{
    Plan pl;
    const int rc = build_plan32(pl, coarse_pts, num_levels, nranks, smooth_iters, rank, q, want_norm);
    if (rc != MG3D_OK)
        return -rc;
    if (out)
        for (int i = 0; i < (int)pl.e.size() && i < max_entries; i++)
            out[i] = pl.e[(size_t)i];
    return (int)pl.e.size();
}

extern "C" int mg3d32_slab_halo(int smooth_iters) { return smooth_iters + 2; }

__global__ void sum32_in_order_kernel(const double *__restrict__ parts, int n, double *__restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.;
        for (int i = 0; i < n; i++)
            t += parts[i];
        *out = t;
    }
}

extern "C" int mg3d32_dist_destroy(mg3d32_dist *D)
{
    if (!D)
        return MG3D_OK;
    if (D->stream)
        (void)hipStreamSynchronize(D->stream);
    /* contexts that borrow the shared stream first, its owner (the first one) last */
    for (size_t i = D->rs.size(); i-- > 0;)
        if (D->rs[i])
            mg3d32_destroy(D->rs[i]);
    if (D->gather)
        (void)hipFree(D->gather);
    if (D->d_norms)
        (void)hipFree(D->d_norms);
    if (D->h_norms)
        (void)hipHostFree(D->h_norms);
    if (D->have_comm)
        (void)ncclCommDestroy(D->comm);
    delete D;
    return MG3D_OK;
}

extern "C" int mg3d32_dist_create(int coarse_pts, int num_levels, int smooth_iters, double omega, double grid_length,
                                  int rank, int nranks, const void *unique_id, int device, mg3d32_dist **out)
{
    if (!out || coarse_pts < 3 || num_levels < 2 || smooth_iters < 1 || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(MG3D_ERR_ARG, "mg3d32_dist_create: bad arguments");
    if (mg3d_device_count() <= 0)
        return fail(MG3D_ERR_NO_DEVICE, "no HIP device available: libmg3d has no CPU fallback");
    HIPCHK(hipSetDevice(device));
    mg3d32_dist *D = new mg3d32_dist();
    D->c = coarse_pts;
    D->L = num_levels;
    D->nu = smooth_iters;
    D->P = nranks;
    D->H = mg3d32_slab_halo(smooth_iters);
    D->ld = mg3d_slab_first_level(coarse_pts, num_levels, nranks, D->H);
    D->loopback = unique_id == nullptr;
    D->device = device;
    D->have_comm = false;
    D->stream = nullptr;
    D->gather = D->d_norms = D->h_norms = nullptr;
    if (D->ld >= num_levels) {
        delete D;
        return fail(MG3D_ERR_ARG, "mg3d32_dist_create: %d ranks leave no level with enough planes per rank", nranks);
    }
    const int first = D->loopback ? 0 : rank, last = D->loopback ? nranks : rank + 1;
    for (int r = first; r < last; r++) {
        std::vector<int> glo(num_levels, 0), ghi(num_levels, 0);
        for (int l = D->ld; l < num_levels; l++)
            mg3d_slab_owned(coarse_pts, num_levels, nranks, D->H, l, r, &glo[l], &ghi[l]);
        mg3d32_ctx *ctx = nullptr;
        const int rc = mg3d32_create_slabs(coarse_pts, num_levels, smooth_iters, omega, grid_length, D->ld, glo.data(),
                                           ghi.data(), D->H, D->stream, &ctx);
        if (rc != MG3D_OK) {
            mg3d32_dist_destroy(D);
            return rc;
        }
        if (!D->stream)
            D->stream = ctx->stream;
        D->rs.push_back(ctx);
        D->rank_of.push_back(r);
    }
#define DCHK(call)                                                                     \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            int rc_ = fail(e_ == hipErrorOutOfMemory ? MG3D_ERR_ALLOC : MG3D_ERR_HIP, \
                           "%s failed: %s", #call, hipGetErrorString(e_));             \
            mg3d32_dist_destroy(D);                                                    \
            return rc_;                                                                \
        }                                                                              \
    } while (0)
    D->norm_slots = 1024;
    DCHK(hipMalloc(&D->gather, sizeof(double) * nranks));
    DCHK(hipMalloc(&D->d_norms, sizeof(double) * D->norm_slots));
    DCHK(hipHostMalloc(&D->h_norms, sizeof(double) * D->norm_slots));
    DCHK(hipStreamSynchronize(D->stream));
#undef DCHK
    if (!D->loopback && nranks > 1) {
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof id);
        ncclResult_t e = ncclCommInitRank(&D->comm, nranks, id, rank);
        if (e != ncclSuccess) {
            const int rc = fail(MG3D_ERR_HIP, "ncclCommInitRank failed: %s", ncclGetErrorString(e));
            mg3d32_dist_destroy(D);
            return rc;
        }
        D->have_comm = true;
    }
    D->cur = nullptr;
    D->phase = 0;
    D->plans.resize((size_t)(num_levels - D->ld) * 2);
    for (int q = D->ld; q < num_levels; q++)
        for (int wn = 0; wn < 2; wn++) {
            std::vector<Plan> &v = D->plans[(size_t)(q - D->ld) * 2 + wn];
            v.resize(D->rs.size());
            for (size_t ri = 0; ri < D->rs.size(); ri++) {
                const int rc = build_plan32(v[ri], coarse_pts, num_levels, nranks, smooth_iters, D->rank_of[ri], q, wn);
                if (rc != MG3D_OK) {
                    mg3d32_dist_destroy(D);
                    return fail(rc, "mg3d32_dist_create: no exchange plan for rank %d of %d", D->rank_of[ri], nranks);
                }
            }
        }
    *out = D;
    return MG3D_OK;
}

extern "C" int mg3d32_dist_first_level(const mg3d32_dist *D) { return D ? D->ld : -1; }
extern "C" int mg3d32_dist_halo(const mg3d32_dist *D) { return D ? D->H : -1; }

extern "C" int mg3d32_dist_comm_info(const mg3d32_dist *D, int *rccl_ranks, int *device)
{
    if (!D)
        return fail(MG3D_ERR_ARG, "mg3d32_dist_comm_info: NULL");
    int n = 0;
    if (D->have_comm)
        NCCLCHK(ncclCommCount(D->comm, &n));
    if (rccl_ranks)
        *rccl_ranks = n;
    if (device)
        *device = D->device;
    return MG3D_OK;
}

/* ------------------------------------------------------------------------------------ data movement */
static int chk_field(mg3d32_dist *D, int field, int level, const char *who)
{
    if (!D || field < 0 || field > 2 || level < 0 || level >= D->L)
        return fail(MG3D_ERR_ARG, "%s: bad field/level", who);
    return MG3D_OK;
}

/* host is the FULL N^3 array; every local rank takes its slab, halos included */
extern "C" int mg3d32_dist_upload(mg3d32_dist *D, int field, int level, const float *host)
{
    CHK(chk_field(D, field, level, "mg3d32_dist_upload"));
    if (!host)
        return fail(MG3D_ERR_ARG, "mg3d32_dist_upload: NULL");
    for (auto *ctx : D->rs) {
        Level32 &l = ctx->lv[level];
        const int N = l.g.N;
        HIPCHK(hipMemcpy2DAsync(l.f[field], l.g.pitch * sizeof(float), host + (size_t)l.g.ig0 * N * N, N * sizeof(float),
                                N * sizeof(float), (size_t)l.g.ni * N, hipMemcpyHostToDevice, D->stream));
    }
    HIPCHK(hipStreamSynchronize(D->stream));
    return MG3D_OK;
}

/* writes the planes each local rank OWNS into the full host array (replicated levels: the whole level) */
extern "C" int mg3d32_dist_download(mg3d32_dist *D, int field, int level, float *host)
{
    CHK(chk_field(D, field, level, "mg3d32_dist_download"));
    if (!host)
        return fail(MG3D_ERR_ARG, "mg3d32_dist_download: NULL");
    for (auto *ctx : D->rs) {
        Level32 &l = ctx->lv[level];
        const int N = l.g.N;
        HIPCHK(hipMemcpy2DAsync(host + (size_t)(l.g.ig0 + l.own_lo) * N * N, N * sizeof(float),
                                l.f[field] + l.g.plane * l.own_lo, l.g.pitch * sizeof(float), N * sizeof(float),
                                (size_t)(l.own_hi - l.own_lo) * N, hipMemcpyDeviceToHost, D->stream));
    }
    HIPCHK(hipStreamSynchronize(D->stream));
    return MG3D_OK;
}

extern "C" int mg3d32_dist_zero(mg3d32_dist *D, int field, int level)
{
    CHK(chk_field(D, field, level, "mg3d32_dist_zero"));
    for (auto *ctx : D->rs)
        HIPCHK(hipMemsetAsync(ctx->lv[level].f[field], 0, ctx->lv[level].elems * sizeof(float), D->stream));
    return MG3D_OK;
}

extern "C" int mg3d32_dist_fill_boundary(mg3d32_dist *D, int field, int level)
{
    CHK(chk_field(D, field, level, "mg3d32_dist_fill_boundary"));
    for (auto *ctx : D->rs)
        e32_fill_boundary(ctx, field, level);
    return MG3D_OK;
}

extern "C" int mg3d32_dist_sync(mg3d32_dist *D)
{
    if (!D)
        return fail(MG3D_ERR_ARG, "mg3d32_dist_sync: NULL");
    HIPCHK(hipStreamSynchronize(D->stream));
    return MG3D_OK;
}

/* --------------------------------------------------------------------------------------- transport */
/* the next phase of the plan of the cycle being enqueued (mg3d_plan.h: plan_run); kind / level say where the schedule
 * believes it is */
static int run_phase32(mg3d32_dist *D, int kind, int level)
{
    const int ph = D->phase++;
    for (const Plan &pl : *D->cur)
        if (ph >= (int)pl.kind.size() || pl.kind[(size_t)ph] != kind || pl.level[(size_t)ph] != level)
            return fail(MG3D_ERR_STATE, "fp32 slab schedule and exchange plan out of step at phase %d (schedule: kind %d level %d)",
                        ph, kind, level);
    if (D->P == 1)
        return MG3D_OK;
    if (kind == MG3D_XK_NORM) { /* one double per rank: the all-gather is in doubles whatever the grid's precision */
        auto none = [&](size_t, const mg3d_xfer &) -> double * { return nullptr; };
        auto sumsq = [&](size_t ri) -> double * { return D->rs[ri]->sumsq; };
        return plan_run<double>(*D->cur, ph, D->loopback, D->comm, ncclDouble, D->stream, none, sumsq, D->gather);
    }
    auto base = [&](size_t ri, const mg3d_xfer &e) -> float * { return D->rs[ri]->lv[e.level].f[e.field]; };
    auto sumsq = [&](size_t ri) -> double * { return D->rs[ri]->sumsq; };
    return plan_run<float>(*D->cur, ph, D->loopback, D->comm, ncclFloat, D->stream, base, sumsq, D->gather);
}

/* the coarse planes of the first replicated level (ld-1) that rank r restricts into: those under its owned fine planes */
static void coarse_range(const mg3d32_dist *D, int r, int *lo, int *hi)
{
    int flo, fhi;
    mg3d_slab_owned(D->c, D->L, D->P, D->H, D->ld, r, &flo, &fhi);
    const int Nc = ((D->c - 1) << (D->ld - 1)) + 1;
    *lo = r == 0 ? 0 : flo / 2;
    *hi = r == D->P - 1 ? Nc : fhi / 2;
}

/* total = sum over ranks, in rank order (the same bits on every rank), of each rank's sumsq[0] */
static int reduce_norm32(mg3d32_dist *D, int q, int slot)
{
    hipStream_t s = D->stream;
    CHK(run_phase32(D, MG3D_XK_NORM, q));
    if (D->P == 1)
        HIPCHK(hipMemcpyAsync(D->gather, D->rs[0]->sumsq, sizeof(double), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(sum32_in_order_kernel, dim3(1), dim3(64), 0, s, D->gather, D->P, D->d_norms + slot);
    return MG3D_OK;
}

/* ----------------------------------------------------------------------------------------- V-cycle */
/* One V-cycle from level q (mg_3d.h:1242-1362 with the Jacobi smoother).  On entry the halos of d are exact on level q
 * and those of u_q at least as fresh as its owned planes' last change requires: they are refreshed here first.
 * norm_slot >= 0: the squared residual norm of level q (all ranks) lands in d_norms[norm_slot]. */
static int dist32_vcycle(mg3d32_dist *D, int q, int norm_slot)
{
    const int ld = D->ld;
    if (q < ld) { /* entirely replicated: every rank runs the ordinary cycle on identical data */
        for (auto *ctx : D->rs)
            CHK(e32_vcycle(ctx, q, ctx->sumsq_slots - 1));
        return MG3D_OK;
    }
    D->cur = &D->plans[(size_t)(q - ld) * 2 + (norm_slot >= 0 ? 1 : 0)];
    D->phase = 0;
    CHK(run_phase32(D, MG3D_XK_HALO_U_NEXT, q));
    for (int l = q; l >= ld; l--) {
        for (size_t ri = 0; ri < D->rs.size(); ri++) {
            mg3d32_ctx *ctx = D->rs[ri];
            e32_jacobi(ctx, l, ctx->iters); /* :1282 on every local plane; the outer nu halo planes go stale */
            if (l - 1 >= ld) {
                e32_residual_restrict(ctx, l); /* :1294 + :1310 into the owned coarse planes */
            } else {
                int lo, hi;
                coarse_range(D, D->rank_of[ri], &lo, &hi);
                e32_residual_restrict(ctx, l, lo, hi); /* this rank's part of the replicated coarse level */
            }
            Level32 &lc = ctx->lv[l - 1];
            HIPCHK(hipMemsetAsync(lc.f[MG3D_U], 0, lc.elems * sizeof(float), D->stream)); /* :1258 */
        }
        CHK(run_phase32(D, MG3D_XK_HALO_U_DOWN, l)); /* for the prolongation and the post-smoother on the way up */
        if (l - 1 >= ld)
            CHK(run_phase32(D, MG3D_XK_HALO_D, l - 1));
        else
            CHK(run_phase32(D, MG3D_XK_RHS_ALLGATHER, ld - 1));
    }
    for (auto *ctx : D->rs)
        CHK(e32_vcycle(ctx, ld - 1, ctx->sumsq_slots - 1)); /* replicated levels and the direct solve */
    for (int l = ld; l <= q; l++) {
        if (l - 1 >= ld)
            CHK(run_phase32(D, MG3D_XK_HALO_U_UP, l - 1)); /* the correction's halos: prolongation covers every local fine plane */
        const bool top = l == q && norm_slot >= 0;
        for (auto *ctx : D->rs) {
            /* :1331 + :1341 (+ :1354): prolongation folded into the first paired sweep; the norm over the owned planes */
            if (!e32_jacobi(ctx, l, ctx->iters, top ? 0 : -1, true) && top)
                e32_residual(ctx, l, false, 0);
        }
        if (top)
            CHK(reduce_norm32(D, q, norm_slot));
    }
    if (D->phase != (int)(*D->cur)[0].kind.size())
        return fail(MG3D_ERR_STATE, "fp32 slab schedule ended after %d of the plan's %d phases", D->phase,
                    (int)(*D->cur)[0].kind.size());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(MG3D_ERR_HIP, "mg3d32_dist: kernel launch failed: %s", hipGetErrorString(e));
    return MG3D_OK;
}

extern "C" int mg3d32_dist_vcycles(mg3d32_dist *D, int count, double *norms)
{
    if (!D || count < 0)
        return fail(MG3D_ERR_ARG, "mg3d32_dist_vcycles: bad arguments");
    for (int done = 0; done < count;) {
        const int nb = (count - done < D->norm_slots) ? count - done : D->norm_slots;
        for (int c = 0; c < nb; c++)
            CHK(dist32_vcycle(D, D->L - 1, c));
        HIPCHK(hipMemcpyAsync(D->h_norms, D->d_norms, nb * sizeof(double), hipMemcpyDeviceToHost, D->stream));
        HIPCHK(hipStreamSynchronize(D->stream));
        if (D->have_comm) { /* an RCCL failure that surfaced asynchronously: an error here, not a wrong number later */
            ncclResult_t ae = ncclSuccess;
            NCCLCHK(ncclCommGetAsyncError(D->comm, &ae));
            if (ae != ncclSuccess)
                return fail(MG3D_ERR_HIP, "RCCL reported an asynchronous error: %s", ncclGetErrorString(ae));
        }
        if (norms)
            for (int c = 0; c < nb; c++)
                norms[done + c] = sqrt(D->h_norms[c]);
        done += nb;
    }
    return MG3D_OK;
}

/* F-cycle start (FMG), mg_dirichlet_analytic.c:771-806: the right-hand sides d of ALL levels are the caller's */
extern "C" int mg3d32_dist_fmg_initialize(mg3d32_dist *D)
{
    if (!D)
        return fail(MG3D_ERR_ARG, "mg3d32_dist_fmg_initialize: NULL");
    for (auto *ctx : D->rs) {
        e32_fill_boundary(ctx, MG3D_U, 0); /* :780 */
        CHK(e32_coarse_solve(ctx));        /* :783 */
    }
    for (int l = 1; l < D->L; l++) {
        if (l - 1 >= D->ld) { /* the interpolated solution is formed on every local plane: the halos of its parent */
            std::vector<Plan> one(D->rs.size());
            for (size_t ri = 0; ri < D->rs.size(); ri++) {
                PlanGeom G{D->c, D->L, D->P, D->nu, D->H, D->ld, 32};
                plan_halo(one[ri], G, MG3D_XK_HALO_U_UP, MG3D_U, l - 1, D->rank_of[ri], 0, 0);
                one[ri].begin.push_back((int)one[ri].e.size());
            }
            D->cur = &one;
            D->phase = 0;
            CHK(run_phase32(D, MG3D_XK_HALO_U_UP, l - 1));
            D->cur = nullptr;
        }
        for (auto *ctx : D->rs) {
            e32_prolong(ctx, l);               /* :795 */
            e32_fill_boundary(ctx, MG3D_U, l); /* :798 */
            Level32 &lc = ctx->lv[l - 1];
            HIPCHK(hipMemsetAsync(lc.f[MG3D_U], 0, lc.elems * sizeof(float), D->stream)); /* :801 */
        }
        CHK(dist32_vcycle(D, l, -1)); /* :804 */
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(MG3D_ERR_HIP, "mg3d32_dist_fmg_initialize: kernel launch failed: %s", hipGetErrorString(e));
    return MG3D_OK;
}
