/*
 * mg3d_dist.hip -- the V-cycle on i-slabs of several GPUs (one process per GPU, RCCL over xGMI).
 *
 * Partition: the reference's OpenMP path already splits every operator over the slowest index i
 * (`#pragma omp for` over i, mg_3d.h:658-659); its implicit barriers become plane exchanges here.
 * Rank r owns the global planes [b_l(r), b_l(r+1)) of level l, with b_{l+1} = 2*b_l, so coarse plane ic
 * and fine plane 2*ic always have the same owner.  Levels too small to give every rank 16 planes are
 * REPLICATED: the restricted right-hand side is all-gathered once per cycle and every rank runs the
 * remaining levels (and the gauss_elim.h direct solve) redundantly on identical data -- what a gather to
 * rank 0 followed by a broadcast would deliver, bit for bit, with one collective instead of two.
 *
 * Halo: H = 2*nu + 2 planes on each side.  The fused sweep applies S = 2*nu colour passes per launch; a
 * slab end that is not refreshed between passes goes stale one plane per pass, so after a sweep the
 * local planes [S, ni-S) are exact, the residual on [S+1, ni-S-1) -- which covers the owned planes and
 * the one extra fine plane restriction needs (H = S+2).  Planes are contiguous in memory, so a halo is
 * one ncclSend/ncclRecv pair per neighbour, no packing.
 *
 * Exchanges per distributed level l and cycle:
 *   d_(l-1)  after restriction (small; on the compute stream, the coarser level needs it at once);
 *   u_l      after pre-smoothing + restriction.  Nothing touches u_l again until the prolongation on the
 *            way up, so this -- the large one -- runs on the communication stream (its own communicator)
 *            underneath all the smoothing of the coarser levels;
 *   u_(l-1)  after the coarser level's post-smoothing (an eighth of the volume): with fresh halos on the
 *            coarse correction AND on u_l the prolongation is applied to halo planes as well, and the
 *            post-smoother starts at once from 6 exact halo planes, one launch, no exchange;
 *   finest u after post-smoothing, for the next cycle's pre-smoother: on the communication stream,
 *            underneath the residual-norm kernel.  That kernel reads the first halo plane on either side,
 *            so the exchange leaves it alone (the local post-smoother already produced its exact value)
 *            and refreshes planes 2..H only.
 *
 * Transports: RCCL (ncclCommInitRank from a unique id the launcher distributes), or "loopback": all
 * ranks are virtual, live in this process on one GPU and exchange by device copies -- the same
 * schedule code, used to verify the decomposition bit for bit on a single-GPU box.
 */
#include "mg3d_ctx.h"

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

/* ------------------------------------------------------------------------------------------ plan */
extern "C" int mg3d_slab_halo(int smooth_iters) { return 2 * smooth_iters + 2; }

/* first distributed level: every rank must own at least max(16, halo) planes there.  Thinner slabs are launch-
 * latency bound either way (a 65^3 level costs a rank the same whether it sweeps 8+12 planes or all 65), so
 * replicating them is free in kernel time and saves three exchanges per level and cycle (loopback rehearsal, 8
 * ranks: 9.16 ms per cycle with 65^3 replicated against 9.22 distributed; 129^3 replicated as well: 9.61).
 * MG3D_SLAB_MIN_PLANES overrides the 16 (minimum 8): a knob for tuning on real xGMI. */
extern "C" int mg3d_slab_first_level(int coarse_pts, int num_levels, int nranks, int halo)
{
    const char *e = getenv("MG3D_SLAB_MIN_PLANES"); /* read per call: host-side planning only */
    const int v = e ? atoi(e) : 16;
    const int floor_planes = v > 8 ? v : 8;
    const int need = halo > floor_planes ? halo : floor_planes;
    for (int l = 1; l < num_levels; l++) {
        const long long n1 = ((long long)(coarse_pts - 1) << l);
        if (n1 / nranks >= need)
            return l;
    }
    return num_levels; /* nothing can be distributed */
}

/* owned planes [lo, hi) of `rank` on `level` (level >= first distributed level) */
extern "C" int mg3d_slab_owned(int coarse_pts, int num_levels, int nranks, int halo, int level, int rank, int *lo,
                               int *hi)
{
    const int ld = mg3d_slab_first_level(coarse_pts, num_levels, nranks, halo);
    if (level < ld || level >= num_levels || rank < 0 || rank >= nranks)
        return MG3D_ERR_ARG;
    const long long n1 = ((long long)(coarse_pts - 1) << ld); /* N-1 on level ld */
    auto cut = [&](int r) -> long long {
        if (r <= 0)
            return 0;
        const long long Nl = ((long long)(coarse_pts - 1) << level) + 1;
        if (r >= nranks)
            return Nl;
        /* even cut on the first distributed level (its coarse partner ld-1 is split at exact halves),
         * doubled per finer level: b_{l+1} = 2 b_l */
        return ((((n1 / 2) * r) / nranks) * 2) << (level - ld);
    };
    *lo = (int)cut(rank);
    *hi = (int)cut(rank + 1);
    return MG3D_OK;
}

static int build_plan(Plan &pl, int c, int L, int P, int nu, int rank, int overlap, int policy)
{
    if (c < 3 || L < 2 || nu < 1 || P < 1 || rank < 0 || rank >= P)
        return MG3D_ERR_ARG;
    PlanGeom G{c, L, P, nu, mg3d_slab_halo(nu), 0, 16};
    G.ld = mg3d_slab_first_level(c, L, P, G.H);
    if (G.ld >= L)
        return MG3D_ERR_ARG;
    pl = Plan();
    const int lc = G.ld - 1, Nc = ((c - 1) << lc) + 1;
    const long long pec = plan_plane_elems(G, Nc);
    const int cs = overlap ? 1 : 0;
    /* policy bit 3 (8): the cycle before this one ended with the one-launch up-leg (bit 2), which forms only the red half of
     * its residual norm; the black half falls out of this cycle's one-launch down-leg, and the norm is reduced behind it */
    if ((policy & 8) && nu == 2)
        plan_norm(pl, G, L - 1, rank);
    for (int l = L - 1; l >= G.ld; l--) { /* down */
        /* first what the coarser level waits for (its right-hand side), then the large u exchange that hides underneath the
         * coarser levels: with ONE communicator, driven from one in-order stream, the issue order is the execution order */
        if (l - 1 >= G.ld) {
            plan_halo(pl, G, MG3D_XK_HALO_D, MG3D_D, l - 1, rank, 0, 0);
        } else if (policy & 1) {
            const int ph = plan_open_phase(pl, MG3D_XK_RHS_GATHER, lc);
            for (int r = 1; r < P; r++) {
                int lo, hi;
                plan_coarse_range(G, r, &lo, &hi);
                if (rank == r)
                    pl.e.push_back(mg3d_xfer{ph, MG3D_XK_RHS_GATHER, MG3D_XOP_SEND, 0, MG3D_D, lc, lo, hi - lo, pec, 0});
                if (rank == 0)
                    pl.e.push_back(mg3d_xfer{ph, MG3D_XK_RHS_GATHER, MG3D_XOP_RECV, r, MG3D_D, lc, lo, hi - lo, pec, 0});
            }
        } else {
            plan_rhs_allgather(pl, G);
        }
        plan_halo(pl, G, MG3D_XK_HALO_U_DOWN, MG3D_U, l, rank, 0, cs);
    }
    if (policy & 1) {
        const int ph = plan_open_phase(pl, MG3D_XK_CORR_BCAST, lc);
        if (P > 1)
            pl.e.push_back(mg3d_xfer{ph, MG3D_XK_CORR_BCAST, MG3D_XOP_BCAST, 0, MG3D_U, lc, 0, Nc, pec, 0});
    }
    for (int l = G.ld; l < L; l++) { /* up */
        if (l - 1 >= G.ld)
            plan_halo(pl, G, MG3D_XK_HALO_U_UP, MG3D_U, l - 1, rank, 0, 0);
        if (l == L - 1) {
            /* policy bit 1: the cycle is carried into the next one (csrc/mg3d_ctx.hip "carried cycles", V(2,2) only): its
             * last launch has used up every halo plane and already holds three of the next cycle's pre-smoothing passes;
             * the one launch left of that down-leg (one pass + residual + restriction) reads three planes either side */
            /* policy bit 2 (4): one launch per leg (csrc/mg3d_ctx.hip, "one launch per leg"): the up-leg's one launch writes
             * the owned planes only; the next cycle's down-leg (three passes + residual + restriction) reads five either side */
            if ((policy & 4) && nu == 2)
                plan_halo(pl, G, MG3D_XK_HALO_U_NEXT, MG3D_U, l, rank, 0, cs, 5);
            else if ((policy & 2) && nu == 2)
                plan_halo(pl, G, MG3D_XK_HALO_U_NEXT, MG3D_U, l, rank, 0, cs, 3);
            else
                plan_halo(pl, G, MG3D_XK_HALO_U_NEXT, MG3D_U, l, rank, 1, cs);
        }
    }
    if (!((policy & 4) && nu == 2)) /* (bit 2: this cycle's norm is completed and reduced by the next cycle) */
        plan_norm(pl, G, L - 1, rank);
    pl.begin.push_back((int)pl.e.size());
    /* stream: 0 the compute stream (overlap off: every exchange sits where the schedule issues it); overlap on: EVERY
     * exchange is issued on the communication stream with the one communicator -- 1: the compute stream joins at once
     * (the exchange is on the critical path), 2: it joins when it next needs the field (the exchange runs underneath the
     * launches in between: the u halos on the way down, the finest u's halos for the next cycle) */
    for (auto &e : pl.e)
        e.stream = !overlap ? 0 : (e.kind == MG3D_XK_HALO_U_DOWN || e.kind == MG3D_XK_HALO_U_NEXT) ? 2 : 1;
    return MG3D_OK;
}

extern "C" int mg3d_dist_plan(int coarse_pts, int num_levels, int nranks, int smooth_iters, int rank, int overlap,
                              int policy, mg3d_xfer *out, int max_entries)
{
    Plan pl;
    const int rc = build_plan(pl, coarse_pts, num_levels, nranks, smooth_iters, rank, overlap, policy);
    if (rc != MG3D_OK)
        return rc < 0 ? rc : -rc;
    if (out)
        for (int i = 0; i < (int)pl.e.size() && i < max_entries; i++)
            out[i] = pl.e[(size_t)i];
    return (int)pl.e.size();
}

/* ------------------------------------------------------------------------------------- structures */
struct SlabLevel {
    Level lv;
    int glo, ghi;         /* owned global planes */
    int own_lo, own_hi;   /* the same in local plane indices */
    int h_lo, h_hi;       /* halo planes below / above (0 at a physical boundary) */
};

struct RankState {
    int legs_npa = 0; /* partial sums the one-launch up-leg has left in coarse->partials */
    int rank;
    std::vector<SlabLevel> dl; /* distributed levels ld .. L-1, index l - ld */
    mg3d_ctx *coarse;          /* replicated levels 0 .. ld-1 */
    double *gather;            /* P doubles: per-rank partial sums of squares */
};

struct mg3d_dist {
    int c, L, nu, P, ld, H;
    double length;
    bool loopback;
    int device;
    ncclComm_t comm;  /* compute-stream collectives and exchanges */
    bool have_comm;
    hipStream_t stream; /* every operation of every local rank is ordered on this one stream */
    hipStream_t comm_stream; /* the u halo exchanges that hide under coarser levels / the norm run here */
    hipEvent_t ev_ready, ev_now; /* compute -> communication stream ("the data is ready"), and back ("it has arrived") */
    std::vector<hipEvent_t> ev_u; /* per level: "the u halos of this level have arrived" */
    std::vector<char> u_pending;  /* per level: ev_u[l] has been recorded and not yet waited for */
    bool overlap; /* MG3D_NO_OVERLAP=1 keeps every exchange on the compute stream */
    std::vector<RankState> rs;
    double *h_norms; /* pinned */
    double *d_norms; /* device: squared norms per cycle */
    int norm_slots;
    std::vector<double> spacing; /* per level */
    double *selftest; /* MG3D_FORCE_COMM=1 on one rank: H planes of the finest level, target of the self-addressed receives */
    /* the exchange plan of one cycle for every local rank (mg3d_dist_plan): the transports below execute it entry by
     * entry and hold no plane arithmetic of their own */
    std::vector<Plan> plans;       /* one cycle on its own */
    std::vector<Plan> plans_carry; /* a cycle that ends ahead into the next one (policy bit 1) */
    std::vector<Plan> plans_legs[3]; /* one launch per leg: a cycle whose up-leg is one launch (policy | 4), one between two such
                                      * (| 12), the one that only completes its predecessor's norm (| 8) */
    bool legs_pending;             /* the last cycle's norm is half formed (red half in every rank's partials[0 .. legs_npa)) */
    bool red_tail;                 /* u of the finest level was last changed by the red pass that ends a V(2,2) cycle, all its halo
                                    * planes are exact and neither u nor d has been touched since (cleared by every entry point that
                                    * could): the next call's first cycle takes the one-launch down-leg too (csrc/mg3d_ctx.hip, red_tail) */
    bool legs_fixed, legs_on;      /* as carry_fixed / carry_on: agreed between the ranks of an RCCL job at creation */
    int n_legs;                    /* cycles whose up-leg ran as one launch (mg3d_dist_legs_cycles) */
    const std::vector<Plan> *cur;  /* the plan of the cycle being enqueued */
    bool carried;                  /* u of the top level holds three pre-smoothing passes of the next cycle */
    /* a call failed after a cycle of it had carried: u of the top level is three passes into a cycle nobody finished and
     * nothing here can put the finished cycle's own u back (the single-domain path has mg3d_drop_carry; on slabs that
     * would be two passes over the owned planes AND a full halo exchange -- on a handle whose transport may just have
     * failed).  Every later call that reads or continues from u fails loudly until u of the top level is uploaded again. */
    bool poisoned;
    /* carried cycles on or off, decided ONCE for a multi-rank RCCL job (mg3d_dist_create: the ranks agree by an
     * all-reduce) -- the two plans differ in the HALO_U_NEXT phase (3 planes against H - 1), and ranks that read
     * MG3D_NO_CARRY / MG3D_CARRY_MIN differently per cycle would post sends and receives of different sizes: a hang or
     * corrupted halos without a diagnostic.  Loopback and single-rank handles follow their options (mg3d_dist_set_option)
     * from cycle to cycle (the tests toggle them). */
    bool carry_fixed, carry_on;
    int n_carried;                 /* cycles that ended that way (mg3d_dist_carried_cycles) */
    int phase;  /* next phase of the cycle being enqueued */
    int policy; /* bit 0: coarse levels on rank 0 only (MG3D_COARSE_GATHER=1) */
    /* per-phase cost (mg3d_dist_timing_enable): event pairs, resolved at the next synchronisation */
    bool timing;
    struct Timed {
        int bucket; /* 0 cycle, 1 exchange on the compute stream, 2 exchange on the communication stream, 3 coarse levels */
        hipEvent_t a, b;
    };
    std::vector<Timed> timed;
    std::vector<hipEvent_t> ev_pool;
    double t_ms[4];
    int t_cycles;
};

static hipEvent_t dist_take_event(mg3d_dist *D)
{
    hipEvent_t e = nullptr;
    if (!D->ev_pool.empty()) {
        e = D->ev_pool.back();
        D->ev_pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
        e = nullptr;
    }
    return e;
}

struct DistScope { /* event pair around a piece of the cycle on stream s */
    mg3d_dist *D;
    mg3d_dist::Timed t;
    hipStream_t s;
    DistScope(mg3d_dist *d, int bucket, hipStream_t st) : D(d), s(st)
    {
        t.bucket = bucket;
        t.a = t.b = nullptr;
        if (D->timing && (t.a = dist_take_event(D)))
            (void)hipEventRecord(t.a, s);
    }
    ~DistScope()
    {
        if (!D->timing || !t.a)
            return;
        if ((t.b = dist_take_event(D)))
            (void)hipEventRecord(t.b, s);
        D->timed.push_back(t);
    }
};

/* call only after the streams have been synchronised */
static void dist_resolve_timers(mg3d_dist *D)
{
    for (auto &t : D->timed) {
        float ms = 0.f;
        if (t.a && t.b && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess)
            D->t_ms[t.bucket] += ms;
        if (t.a)
            D->ev_pool.push_back(t.a);
        if (t.b)
            D->ev_pool.push_back(t.b);
    }
    D->timed.clear();
}

static SlabLevel &SL(mg3d_dist *D, RankState &R, int l) { return R.dl[l - D->ld]; }

__global__ void sum_in_order_kernel(const double *__restrict__ parts, int n, double *__restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.;
        for (int i = 0; i < n; i++)
            t += parts[i];
        *out = t;
    }
}

static bool dist_carry_policy(mg3d_dist *D);
static bool dist_legs_policy(mg3d_dist *D);
static int dist_refuse_poisoned(const mg3d_dist *D, const char *who);

extern "C" int mg3d_comm_unique_id(void *out128)
{
    if (!out128)
        return fail(MG3D_ERR_ARG, "mg3d_comm_unique_id: NULL");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    memcpy(out128, &id, sizeof id);
    return MG3D_OK;
}

extern "C" int mg3d_dist_destroy(mg3d_dist *D)
{
    if (!D)
        return MG3D_OK;
    if (D->comm_stream)
        (void)hipStreamSynchronize(D->comm_stream);
    if (D->stream)
        (void)hipStreamSynchronize(D->stream);
    dist_resolve_timers(D);
    for (auto e : D->ev_pool)
        (void)hipEventDestroy(e);
    if (D->ev_ready)
        (void)hipEventDestroy(D->ev_ready);
    if (D->ev_now)
        (void)hipEventDestroy(D->ev_now);
    for (auto e : D->ev_u)
        if (e)
            (void)hipEventDestroy(e);
    if (D->comm_stream)
        (void)hipStreamDestroy(D->comm_stream);
    for (auto &R : D->rs) {
        for (auto &s : R.dl) {
            for (int k = 0; k < 3; k++)
                if (s.lv.f[k])
                    (void)hipFree(s.lv.f[k]);
            if (s.lv.alt)
                (void)hipFree(s.lv.alt);
        }
        if (R.gather)
            (void)hipFree(R.gather);
    }
    /* contexts that borrow the shared stream go first, its owner last */
    for (auto &R : D->rs)
        if (R.coarse && !R.coarse->own_stream) {
            mg3d_ctx_destroy(R.coarse);
            R.coarse = nullptr;
        }
    for (auto &R : D->rs)
        if (R.coarse)
            mg3d_ctx_destroy(R.coarse);
    if (D->selftest)
        (void)hipFree(D->selftest);
    if (D->d_norms)
        (void)hipFree(D->d_norms);
    if (D->h_norms)
        (void)hipHostFree(D->h_norms);
    if (D->have_comm)
        (void)ncclCommDestroy(D->comm);
    delete D;
    return MG3D_OK;
}

extern "C" int mg3d_dist_create(int coarse_pts, int num_levels, int smooth_iters, double grid_length, int rank,
                                int nranks, const void *unique_id, int device, mg3d_dist **out)
{
    if (!out || coarse_pts < 3 || num_levels < 2 || smooth_iters < 1 || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(MG3D_ERR_ARG, "mg3d_dist_create: bad arguments");
    if (mg3d_device_count() <= 0)
        return fail(MG3D_ERR_NO_DEVICE, "no HIP device available: libmg3d has no CPU fallback");
    HIPCHK(hipSetDevice(device));
    mg3d_dist *D = new mg3d_dist();
    D->c = coarse_pts;
    D->L = num_levels;
    D->nu = smooth_iters;
    D->P = nranks;
    D->length = grid_length;
    D->H = mg3d_slab_halo(smooth_iters);
    D->ld = mg3d_slab_first_level(coarse_pts, num_levels, nranks, D->H);
    D->loopback = unique_id == nullptr;
    D->device = device;
    D->have_comm = false;
    D->selftest = nullptr;
    D->stream = nullptr;
    D->h_norms = D->d_norms = nullptr;
    D->comm_stream = nullptr;
    D->ev_ready = D->ev_now = nullptr;
    D->phase = 0;
    D->timing = false;
    D->t_cycles = 0;
    for (double &x : D->t_ms)
        x = 0.;
    D->policy = (getenv("MG3D_COARSE_GATHER") && getenv("MG3D_COARSE_GATHER")[0] == '1') ? 1 : 0;
    /* Overlap of the large u exchanges with the launches in between (round 4: ONE communicator).  Every exchange is issued
     * on the communication stream -- the only stream that ever drives the communicator, so RCCL sees one in-order sequence
     * of operations, identical on all ranks -- behind an event of the compute stream ("the planes are ready"); the compute
     * stream joins by an event when it needs the planes: at once for the exchanges on the critical path, later for the u
     * halos, whose transfers then run underneath the coarser levels / the interior of the next launch.  Default for both
     * transports (rounds 2-3 needed a second communicator for this and kept it off for RCCL); MG3D_NO_OVERLAP=1 keeps
     * every exchange on the compute stream. */
    {
        const bool off = getenv("MG3D_NO_OVERLAP") && getenv("MG3D_NO_OVERLAP")[0] == '1';
        D->overlap = !off;
    }
    if (D->ld >= num_levels) {
        delete D;
        return fail(MG3D_ERR_ARG, "mg3d_dist_create: %d ranks leave no level with >= 8 planes per rank", nranks);
    }
    /* spacing per level: finest = length/(N-1), doubled per coarser level (mg_3d.h:143,1303) */
    D->spacing.resize(num_levels);
    const long long finest = ((long long)(coarse_pts - 1) << (num_levels - 1)) + 1;
    D->spacing[num_levels - 1] = grid_length / (double)(finest - 1);
    for (int l = num_levels - 2; l >= 0; l--)
        D->spacing[l] = 2 * D->spacing[l + 1];
#define DCHK(call)                                                                        \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            int rc_ = fail(e_ == hipErrorOutOfMemory ? MG3D_ERR_ALLOC : MG3D_ERR_HIP,    \
                           "%s failed: %s", #call, hipGetErrorString(e_));                \
            mg3d_dist_destroy(D);                                                         \
            return rc_;                                                                   \
        }                                                                                 \
    } while (0)
    const int first = D->loopback ? 0 : rank, last = D->loopback ? nranks : rank + 1;
    for (int r = first; r < last; r++) {
        RankState R;
        R.rank = r;
        R.coarse = nullptr;
        R.gather = nullptr;
        D->rs.push_back(R);
    }
    for (auto &R : D->rs) {
        /* replicated part: an ordinary single-domain context of the ld lowest levels */
        const int rc = mg3d_ctx_create(coarse_pts, D->ld, smooth_iters, 1.0, &R.coarse);
        if (rc != MG3D_OK) {
            mg3d_dist_destroy(D);
            return rc;
        }
        for (int l = 0; l < D->ld; l++)
            R.coarse->lv[l].h = D->spacing[l];
        if (!D->stream)
            D->stream = R.coarse->stream; /* shared by all local ranks: one ordered queue */
        else {
            (void)hipStreamDestroy(R.coarse->stream);
            R.coarse->stream = D->stream;
            R.coarse->own_stream = false;
        }
        DCHK(hipMalloc(&R.gather, sizeof(double) * nranks));
        R.dl.resize(num_levels - D->ld);
        for (int l = D->ld; l < num_levels; l++) {
            SlabLevel &s = R.dl[l - D->ld];
            for (int k = 0; k < 3; k++)
                s.lv.f[k] = nullptr;
            s.lv.alt = nullptr;
            mg3d_slab_owned(coarse_pts, num_levels, nranks, D->H, l, R.rank, &s.glo, &s.ghi);
            const int N = (coarse_pts - 1) * (1 << l) + 1;
            s.h_lo = R.rank > 0 ? D->H : 0;
            s.h_hi = R.rank < nranks - 1 ? D->H : 0;
            Geom &g = s.lv.g;
            g.N = N;
            g.nj = g.nk = N;
            g.ig0 = s.glo - s.h_lo;
            g.ni = (s.ghi - s.glo) + s.h_lo + s.h_hi;
            g.pitch = mg3d_pitch_for(N);
            g.plane = (long long)g.pitch * N;
            s.own_lo = s.h_lo;
            s.own_hi = s.h_lo + (s.ghi - s.glo);
            s.lv.h = D->spacing[l];
            s.lv.elems = (size_t)g.plane * g.ni;
            for (int k = 0; k < 3; k++) {
                DCHK(hipMalloc(&s.lv.f[k], s.lv.elems * sizeof(double)));
                DCHK(hipMemsetAsync(s.lv.f[k], 0, s.lv.elems * sizeof(double), D->stream));
            }
            DCHK(hipMalloc(&s.lv.alt, s.lv.elems * sizeof(double)));
            DCHK(hipMemsetAsync(s.lv.alt, 0, s.lv.elems * sizeof(double), D->stream));
        }
    }
    DCHK(hipStreamCreateWithFlags(&D->comm_stream, hipStreamNonBlocking));
    DCHK(hipEventCreateWithFlags(&D->ev_ready, hipEventDisableTiming));
    DCHK(hipEventCreateWithFlags(&D->ev_now, hipEventDisableTiming));
    D->ev_u.assign(num_levels, nullptr);
    D->u_pending.assign(num_levels, 0);
    for (int l = D->ld; l < num_levels; l++)
        DCHK(hipEventCreateWithFlags(&D->ev_u[l], hipEventDisableTiming));
    D->norm_slots = 1024;
    DCHK(hipMalloc(&D->d_norms, sizeof(double) * D->norm_slots));
    DCHK(hipHostMalloc(&D->h_norms, sizeof(double) * D->norm_slots));
    DCHK(hipStreamSynchronize(D->stream));
#undef DCHK
    /* MG3D_FORCE_COMM=1: build the communicators for a single rank too (lets a one-GPU box exercise the unique-id
     * marshalling, ncclCommInitRank and ncclCommSplit of the multi-process path) */
    const bool force_comm = getenv("MG3D_FORCE_COMM") && getenv("MG3D_FORCE_COMM")[0] == '1';
    if (!D->loopback && (nranks > 1 || force_comm)) {
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof id);
        ncclResult_t e = ncclCommInitRank(&D->comm, nranks, id, rank);
        if (e != ncclSuccess) {
            const int rc = fail(MG3D_ERR_HIP, "ncclCommInitRank failed: %s", ncclGetErrorString(e));
            mg3d_dist_destroy(D);
            return rc;
        }
        D->have_comm = true;
        if (nranks > 1)
            k_sweep_set_tune_default(0); /* no host-blocking first-use measurement while ranks wait for each other */
        if (nranks == 1) { /* forced: scratch for the self-addressed exchanges */
            const Geom &gt = D->rs[0].dl.back().lv.g;
            if (hipMalloc(&D->selftest, (size_t)D->H * gt.plane * sizeof(double)) != hipSuccess)
                D->selftest = nullptr;
        }
    }
    /* the plan every exchange below is read from; built after the overlap decision is final */
    D->plans.resize(D->rs.size());
    for (size_t ri = 0; ri < D->rs.size(); ri++) {
        const int rc = build_plan(D->plans[ri], coarse_pts, num_levels, nranks, smooth_iters, D->rs[ri].rank,
                                  D->overlap ? 1 : 0, D->policy);
        if (rc != MG3D_OK) {
            mg3d_dist_destroy(D);
            return fail(rc, "mg3d_dist_create: no exchange plan for rank %d of %d", D->rs[ri].rank, nranks);
        }
    }
    D->plans_carry.resize(D->rs.size());
    for (size_t ri = 0; ri < D->rs.size(); ri++)
        (void)build_plan(D->plans_carry[ri], coarse_pts, num_levels, nranks, smooth_iters, D->rs[ri].rank, D->overlap ? 1 : 0,
                         D->policy | 2); /* same arguments as above: cannot fail where that did not */
    static const int legs_bits[3] = {4, 12, 8};
    for (int v = 0; v < 3; v++) {
        D->plans_legs[v].resize(D->rs.size());
        for (size_t ri = 0; ri < D->rs.size(); ri++)
            (void)build_plan(D->plans_legs[v][ri], coarse_pts, num_levels, nranks, smooth_iters, D->rs[ri].rank, D->overlap ? 1 : 0,
                             D->policy | legs_bits[v]);
    }
    D->cur = &D->plans;
    D->carried = false;
    D->poisoned = false;
    D->n_carried = 0;
    D->carry_fixed = false;
    D->carry_on = false;
    D->legs_pending = false;
    D->red_tail = false;
    D->legs_fixed = false;
    D->legs_on = false;
    D->n_legs = 0;
    if (D->have_comm && nranks > 1) {
        /* every rank must pick the same plan variant: all-reduce MIN of "carried cycles are on here", "one launch per leg is" */
        int mine[2] = {dist_carry_policy(D) ? 1 : 0, dist_legs_policy(D) ? 1 : 0}, *dflag = nullptr;
        bool ok = hipMalloc(&dflag, sizeof(mine)) == hipSuccess &&
                  hipMemcpyAsync(dflag, mine, sizeof(mine), hipMemcpyHostToDevice, D->stream) == hipSuccess &&
                  ncclAllReduce(dflag, dflag, 2, ncclInt, ncclMin, D->comm, D->stream) == ncclSuccess &&
                  hipMemcpyAsync(mine, dflag, sizeof(mine), hipMemcpyDeviceToHost, D->stream) == hipSuccess &&
                  hipStreamSynchronize(D->stream) == hipSuccess;
        if (dflag)
            (void)hipFree(dflag);
        if (!ok) {
            mg3d_dist_destroy(D);
            return fail(MG3D_ERR_HIP, "mg3d_dist_create: the ranks could not agree on the cycle schedule (all-reduce failed)");
        }
        D->carry_fixed = true;
        D->carry_on = mine[0] != 0;
        D->legs_fixed = true;
        D->legs_on = mine[1] != 0;
    }
    *out = D;
    return MG3D_OK;
}

extern "C" int mg3d_dist_timing_enable(mg3d_dist *D, int on)
{
    if (!D)
        return fail(MG3D_ERR_ARG, "mg3d_dist_timing_enable: NULL");
    HIPCHK(hipStreamSynchronize(D->comm_stream));
    HIPCHK(hipStreamSynchronize(D->stream));
    dist_resolve_timers(D);
    D->timing = on != 0;
    if (on) {
        for (double &x : D->t_ms)
            x = 0.;
        D->t_cycles = 0;
    }
    return MG3D_OK;
}

extern "C" int mg3d_dist_timing_get(mg3d_dist *D, double ms[4], int *cycles)
{
    if (!D || !ms)
        return fail(MG3D_ERR_ARG, "mg3d_dist_timing_get: NULL");
    HIPCHK(hipStreamSynchronize(D->comm_stream));
    HIPCHK(hipStreamSynchronize(D->stream));
    dist_resolve_timers(D);
    for (int i = 0; i < 4; i++)
        ms[i] = D->t_ms[i];
    if (cycles)
        *cycles = D->t_cycles;
    return MG3D_OK;
}

extern "C" int mg3d_dist_comm_info(const mg3d_dist *D, int *rccl_ranks, int *overlap, int *device)
{
    if (!D)
        return fail(MG3D_ERR_ARG, "mg3d_dist_comm_info: NULL");
    int n = 0;
    if (D->have_comm)
        NCCLCHK(ncclCommCount(D->comm, &n));
    if (rccl_ranks)
        *rccl_ranks = n; /* 0: no RCCL communicator (loopback, or one rank) */
    if (overlap)
        *overlap = D->overlap ? 1 : 0;
    if (device)
        *device = D->device;
    return MG3D_OK;
}

extern "C" int mg3d_dist_first_level(const mg3d_dist *D) { return D ? D->ld : -1; }
extern "C" int mg3d_dist_halo(const mg3d_dist *D) { return D ? D->H : -1; }
extern "C" int mg3d_dist_carried_cycles(const mg3d_dist *D) { return D ? D->n_carried : -1; }
extern "C" int mg3d_dist_legs_cycles(const mg3d_dist *D) { return D ? D->n_legs : -1; }

/* launch / schedule policy by key (mg3d_ctx_set_option) for every local rank.  carry / carry_min of a multi-rank RCCL job
 * are fixed when the handle is created (the ranks agree there) and refuse to change. */
extern "C" int mg3d_dist_set_option(mg3d_dist *D, const char *key, int value)
{
    if (!D)
        return fail(MG3D_ERR_ARG, "mg3d_dist_set_option: NULL");
    D->red_tail = false;
    const int i = mg3d_option_index(key);
    if (i < 0)
        return fail(MG3D_ERR_ARG, "mg3d_dist_set_option: no option \"%s\"", key ? key : "(null)");
    if ((D->carry_fixed && (i == MG3D_OPT_CARRY || i == MG3D_OPT_CARRY_MIN)) || (D->legs_fixed && (i == MG3D_OPT_LEGS || i == MG3D_OPT_LEGS_MIN)))
        return fail(MG3D_ERR_STATE, "mg3d_dist_set_option: %s is agreed between the ranks at creation", key);
    if (D->carried || D->legs_pending)
        return fail(MG3D_ERR_STATE, "mg3d_dist_set_option: inside a carried cycle");
    for (auto &R : D->rs)
        CHK(mg3d_ctx_set_option(R.coarse, key, value));
    return MG3D_OK;
}

extern "C" int mg3d_dist_set_keep_residual(mg3d_dist *D, int keep)
{
    if (!D)
        return fail(MG3D_ERR_ARG, "mg3d_dist_set_keep_residual: NULL");
    D->red_tail = false;
    CHK(dist_refuse_poisoned(D, "mg3d_dist_set_keep_residual"));
    for (auto &R : D->rs)
        R.coarse->keep_r = keep != 0;
    return MG3D_OK;
}

extern "C" int mg3d_dist_build_coarse(mg3d_dist *D, double h_coarse)
{
    if (!D)
        return fail(MG3D_ERR_ARG, "mg3d_dist_build_coarse: NULL");
    D->red_tail = false;
    for (auto &R : D->rs)
        CHK(mg3d_ctx_build_coarse(R.coarse, h_coarse));
    return MG3D_OK;
}

/* ------------------------------------------------------------------------------------ data movement */
static int dist_field_ptr(mg3d_dist *D, RankState &R, int field, int level, double **ptr, Geom *g)
{
    if (field < 0 || field > 2 || level < 0 || level >= D->L)
        return fail(MG3D_ERR_ARG, "mg3d_dist: bad field/level (%d, %d)", field, level);
    if (level >= D->ld) {
        *ptr = SL(D, R, level).lv.f[field];
        *g = SL(D, R, level).lv.g;
    } else {
        *ptr = R.coarse->lv[level].f[field];
        *g = R.coarse->lv[level].g;
    }
    return MG3D_OK;
}

/* host is the FULL N^3 array (reference layout); every local rank takes its slab (halos included) */
extern "C" int mg3d_dist_upload(mg3d_dist *D, int field, int level, const double *host)
{
    if (!D || !host)
        return fail(MG3D_ERR_ARG, "mg3d_dist_upload: NULL");
    D->red_tail = false;
    if (field == MG3D_U && level == D->L - 1)
        D->poisoned = false; /* u of the finest level, halos included, is replaced: nothing of the failed call is left */
    for (auto &R : D->rs) {
        double *p;
        Geom g;
        CHK(dist_field_ptr(D, R, field, level, &p, &g));
        const int N = g.N;
        HIPCHK(hipMemcpy2DAsync(p, g.pitch * sizeof(double), host + (size_t)g.ig0 * N * N, N * sizeof(double),
                                N * sizeof(double), (size_t)g.ni * N, hipMemcpyHostToDevice, D->stream));
        if (level < D->ld)
            mg3d_ctx_touched(R.coarse, field, level);
    }
    HIPCHK(hipStreamSynchronize(D->stream));
    return MG3D_OK;
}

/* writes the planes each local rank OWNS into the full host array (other planes untouched) */
extern "C" int mg3d_dist_download(mg3d_dist *D, int field, int level, double *host)
{
    if (!D || !host)
        return fail(MG3D_ERR_ARG, "mg3d_dist_download: NULL");
    CHK(dist_refuse_poisoned(D, "mg3d_dist_download"));
    for (auto &R : D->rs) {
        double *p;
        Geom g;
        CHK(dist_field_ptr(D, R, field, level, &p, &g));
        const int N = g.N;
        int lo = 0, hi = g.ni, glo = 0;
        if (level >= D->ld) {
            SlabLevel &s = SL(D, R, level);
            lo = s.own_lo;
            hi = s.own_hi;
            glo = s.glo;
        }
        HIPCHK(hipMemcpy2DAsync(host + (size_t)glo * N * N, N * sizeof(double), p + g.plane * lo,
                                g.pitch * sizeof(double), N * sizeof(double), (size_t)(hi - lo) * N,
                                hipMemcpyDeviceToHost, D->stream));
    }
    HIPCHK(hipStreamSynchronize(D->stream));
    return MG3D_OK;
}

extern "C" int mg3d_dist_sync(mg3d_dist *D)
{
    if (!D)
        return fail(MG3D_ERR_ARG, "mg3d_dist_sync: NULL");
    HIPCHK(hipStreamSynchronize(D->stream));
    return MG3D_OK;
}

/* --------------------------------------------------------------------------------------- transport */
/* Executes the next phase of the cycle's plan for every local rank; `kind` / `level` say where the schedule believes it
 * is -- a mismatch means schedule and plan have come apart, and nothing is sent.  RCCL: one group of the rank's
 * sends / receives / broadcasts exactly as listed.  Loopback: every send is copied into the receive entry that names it
 * in the peer's plan of the same phase (counts must agree), every broadcast range from the root's array into all others. */
static int run_phase(mg3d_dist *D, int kind, int level, hipStream_t s, int bucket = 1 /* timer: 1 waited for at once, 2 overlapped */)
{
    const int ph = D->phase++;
    for (auto &pl : *D->cur)
        if (ph >= (int)pl.kind.size() || pl.kind[(size_t)ph] != kind || pl.level[(size_t)ph] != level)
            return fail(MG3D_ERR_STATE, "slab schedule and exchange plan out of step at phase %d (schedule: kind %d level %d)",
                        ph, kind, level);
    if (D->P == 1) {
        /* forced single-rank communicators (MG3D_FORCE_COMM=1): a grouped send/receive addressed to itself, landing in
         * a scratch buffer -- a one-GPU self-test of the calls, streams and events */
        const bool halo = kind == MG3D_XK_HALO_U_DOWN || kind == MG3D_XK_HALO_D || kind == MG3D_XK_HALO_U_UP ||
                          kind == MG3D_XK_HALO_U_NEXT;
        if (D->have_comm && D->selftest && halo) {
            SlabLevel &a = SL(D, D->rs[0], level);
            const int n = D->H - (kind == MG3D_XK_HALO_U_NEXT ? 1 : 0);
            const size_t cnt = (size_t)n * a.lv.g.plane;
            const int field = kind == MG3D_XK_HALO_D ? MG3D_D : MG3D_U;
            ncclComm_t comm = D->comm;
            NCCLCHK(ncclGroupStart());
            NCCLCHK(ncclSend(a.lv.f[field] + a.lv.g.plane * a.own_lo, cnt, ncclDouble, 0, comm, s));
            NCCLCHK(ncclRecv(D->selftest, cnt, ncclDouble, 0, comm, s));
            NCCLCHK(ncclGroupEnd());
        }
        return MG3D_OK;
    }
    DistScope timer(D, bucket, s);
    auto base = [&](size_t ri, const mg3d_xfer &e) -> double * {
        RankState &R = D->rs[ri];
        return e.level >= D->ld ? SL(D, R, e.level).lv.f[e.field] : R.coarse->lv[e.level].f[e.field];
    };
    auto sumsq = [&](size_t ri) -> double * { return D->rs[ri].coarse->sumsq; };
    return plan_run<double>(*D->cur, ph, D->loopback, D->comm, ncclDouble, s, base, sumsq, D->rs[0].gather);
}

/* One exchange phase.  Overlap on (the default): issued on the communication stream -- the one stream that drives the one
 * communicator -- behind everything queued on the compute stream so far; `defer` = false: the compute stream joins at once
 * (the exchange is on the critical path), true: it joins in await_u(l), i.e. the transfer runs underneath whatever the
 * compute stream does until then (u halos of level l only: one pending exchange per level). */
static int exchange(mg3d_dist *D, int kind, int l, bool defer)
{
    if (!D->overlap || (D->P == 1 && !D->selftest))
        return run_phase(D, kind, l, D->stream);
    HIPCHK(hipEventRecord(D->ev_ready, D->stream));
    HIPCHK(hipStreamWaitEvent(D->comm_stream, D->ev_ready, 0));
    CHK(run_phase(D, kind, l, D->comm_stream, defer ? 2 : 1));
    if (defer) {
        HIPCHK(hipEventRecord(D->ev_u[l], D->comm_stream));
        D->u_pending[l] = 1;
    } else {
        HIPCHK(hipEventRecord(D->ev_now, D->comm_stream));
        HIPCHK(hipStreamWaitEvent(D->stream, D->ev_now, 0));
    }
    return MG3D_OK;
}

/* Start refreshing the u halos of level l behind everything queued on the compute stream so far, without
 * holding the compute stream up; await_u() makes the compute stream wait for the arrival. */
static int start_u_exchange(mg3d_dist *D, int kind, int l) { return exchange(D, kind, l, true); }

static int await_u(mg3d_dist *D, int l)
{
    if (D->u_pending[l]) {
        HIPCHK(hipStreamWaitEvent(D->stream, D->ev_u[l], 0));
        D->u_pending[l] = 0;
    }
    return MG3D_OK;
}

/* total = sum over ranks (in rank order, so every rank gets the same bits) of each rank's sumsq[0] */
static int reduce_norm(mg3d_dist *D, int slot)
{
    hipStream_t s = D->stream;
    CHK(exchange(D, MG3D_XK_NORM, D->L - 1, false));
    RankState &R = D->rs[0];
    if (D->P == 1 || (!D->loopback && !D->have_comm)) {
        if (D->P == 1 && D->have_comm) { /* forced single-rank communicator: the collective itself (self-test), on the stream
                                          * that drives the communicator */
            hipStream_t xs = (D->overlap && D->selftest) ? D->comm_stream : s;
            if (xs != s) {
                HIPCHK(hipEventRecord(D->ev_ready, s));
                HIPCHK(hipStreamWaitEvent(xs, D->ev_ready, 0));
            }
            NCCLCHK(ncclAllGather(R.coarse->sumsq, R.gather, 1, ncclDouble, D->comm, xs));
            if (xs != s) {
                HIPCHK(hipEventRecord(D->ev_now, xs));
                HIPCHK(hipStreamWaitEvent(s, D->ev_now, 0));
            }
        } else
            HIPCHK(hipMemcpyAsync(R.gather, R.coarse->sumsq, sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(sum_in_order_kernel, dim3(1), dim3(64), 0, s, R.gather, D->P, D->d_norms + slot);
    return MG3D_OK;
}

/* ----------------------------------------------------------------------------------------- V-cycle */
struct RestrictTarget { /* where a rank's restricted residual goes: coarse geometry, array, local planes */
    const Geom *gc;
    double *dc;
    int lo, hi;
};

/* One smoothing stage on distributed level l for every local rank: iters x two colour passes, optional
 * residual (want_res 2: r stored or restricted on the fly into tgt[], 1: norm only, over OWNED planes, into
 * the rank's coarse->sumsq[0]).  Same launch policy as the single-domain path.  On entry the halos of u and
 * d on this level are exact (H planes); nothing is exchanged in here. */
/* the top level's post-smoothing of a V(2,2) cycle runs as 2 + 2 passes: prolongation folded into the first launch,
 * the norm into the second (same policy and reasons as the single-domain path, csrc/mg3d_ctx.hip) */
static bool dist_split_up_leg(const mg3d_dist *D, int post, int want_res)
{
    return post && 2 * D->nu == 4 && want_res == 1;
}

struct ProlongSource { /* per local rank: the coarse correction a split up-leg folds into its first launch */
    const Geom *gc;
    const double *ec;
};

static int stage_smooth(mg3d_dist *D, int l, int post, int want_res, const RestrictTarget *tgt,
                        bool zero_in = false /* u is identically zero: the first launch does not read it */,
                        bool refresh_u = false /* start the exchange of the u halos (planes 2..H) as soon as u is final */,
                        const ProlongSource *pro = nullptr,
                        bool tap = false /* carried cycle: the stage's last launch is four passes with the norm tapped after
                                            the second -- the next cycle's first pre-smoothing passes ride on it */)
{
    hipStream_t s = D->stream;
    const int c1 = post ? 0 : 1;
    int passes = 2 * D->nu;
    bool done_res = want_res == 0, first = true;
    const bool sp = dist_split_up_leg(D, post, want_res);
    while (passes > 0 || !done_res) {
        const int S = (sp && passes >= 2) ? 2 : passes >= 4 ? 4 : passes;
        const bool last = passes - S == 0;
        /* two passes + residual + restriction: one launch from 130 points per side up, two below (k_sweep_fuse_rst2) */
        const bool res = last && want_res != 0 && S != 4 &&
                         !(S == 2 && tgt != nullptr && !k_sweep_fuse_rst2(D->rs[0].coarse->opt, SL(D, D->rs[0], l).lv.g.N));
        if (S == 0 && refresh_u) { /* u is final; the pure residual launch below reads owned +-1 only */
            CHK(start_u_exchange(D, MG3D_XK_HALO_U_NEXT, l));
            refresh_u = false;
        }
        /* Edge windows first (round 4).  The stage's last launch makes the planes the next cycle's halo exchange sends:
         * the first and last E output planes go first, as ONE launch of two chunks per rank; the exchange is issued behind
         * it on the communication stream and runs underneath the launch that makes the interior -- instead of sitting
         * between this launch and the next cycle's first one.  E = the window's margin beyond the owned planes + the planes
         * the exchange skips + those it sends (carried: 0 + 0 + 3; plain: 1 + 1 + (H - 1)).  Same bits: the chunking of a
         * sweep launch is a work distribution only, and the norm's per-block partial sums are folded from both launches. */
        if (last && refresh_u && S > 0 && D->overlap && D->P > 1 && (res || tap) && want_res == 1) {
            const int margin = tap ? 0 : 1, E = margin + (tap ? 0 : 1) + (tap ? 3 : D->H - 1);
            std::vector<int> wlo(D->rs.size()), whi(D->rs.size()), npa(D->rs.size());
            bool ok = true;
            for (size_t ri = 0; ri < D->rs.size(); ri++) {
                SlabLevel &sl = SL(D, D->rs[ri], l);
                wlo[ri] = sl.own_lo - margin < 0 ? 0 : sl.own_lo - margin;
                whi[ri] = sl.own_hi + margin > sl.lv.g.ni ? sl.lv.g.ni : sl.own_hi + margin;
                ok = ok && whi[ri] - wlo[ri] >= 2 * E + 2;
            }
            if (ok) {
                auto launch = [&](size_t ri, const double *vin, double *vout, double *part, int lo, int hi, int edge) -> int {
                    SlabLevel &sl = SL(D, D->rs[ri], l);
                    Level &lv = sl.lv;
                    mg3d_ctx *cx = D->rs[ri].coarse;
                    if (tap)
                        return k_sweep_tap(cx->opt, lv.g, vin, lv.f[MG3D_D], vout, part, MG3D_MAX_PARTIALS / 2, lv.h, c1, s, sl.own_lo,
                                           sl.own_hi, lo, hi, edge);
                    return k_sweep(cx->opt, lv.g, vin, lv.f[MG3D_D], vout, nullptr, part, MG3D_MAX_PARTIALS / 2, lv.h, S, c1, true, s,
                                   sl.own_lo, sl.own_hi, nullptr, nullptr, -1, -1, (pro && first) ? pro[ri].gc : nullptr,
                                   (pro && first) ? pro[ri].ec : nullptr, lo, hi, edge);
                };
                for (size_t ri = 0; ri < D->rs.size(); ri++) {
                    Level &lv = SL(D, D->rs[ri], l).lv;
                    npa[ri] = launch(ri, lv.f[MG3D_U], lv.alt, D->rs[ri].coarse->partials, wlo[ri], whi[ri], E);
                    if (npa[ri] < 0)
                        return fail(MG3D_ERR_STATE, "slab sweep: no kernel for the edge windows on level %d", l);
                }
                for (auto &R : D->rs) { /* the exchange sends from (and lands in) the NEW buffer */
                    Level &lv = SL(D, R, l).lv;
                    double *t = lv.f[MG3D_U];
                    lv.f[MG3D_U] = lv.alt;
                    lv.alt = t;
                }
                CHK(start_u_exchange(D, MG3D_XK_HALO_U_NEXT, l));
                refresh_u = false;
                for (size_t ri = 0; ri < D->rs.size(); ri++) {
                    Level &lv = SL(D, D->rs[ri], l).lv;
                    mg3d_ctx *cx = D->rs[ri].coarse;
                    const int npb = launch(ri, lv.alt, lv.f[MG3D_U], cx->partials + MG3D_MAX_PARTIALS / 2, wlo[ri] + E, whi[ri] - E, 0);
                    if (npb < 0)
                        return fail(MG3D_ERR_STATE, "slab sweep: no kernel for the interior window on level %d", l);
                    k_fold2(cx->partials, npa[ri], cx->partials + MG3D_MAX_PARTIALS / 2, npb, cx->sumsq, s);
                }
                done_res = true;
                passes -= S;
                first = false;
                continue;
            }
        }
        for (size_t ri = 0; ri < D->rs.size(); ri++) {
            RankState &R = D->rs[ri];
            SlabLevel &sl = SL(D, R, l);
            Level &lv = sl.lv;
            mg3d_ctx *cx = R.coarse;
            const bool rst = res && tgt != nullptr && tgt[ri].dc != nullptr;
            /* output planes of a smoothing launch: the owned planes plus what the next consumer reads beyond
             * them -- the residual/restriction behind a pre-smoother needs u on owned +-2, the top-level norm
             * owned +-1; the remaining halo planes are refreshed by an exchange before anything reads them
             * again.  Only the LAST smoothing launch of the stage may be trimmed: an earlier one feeds the next
             * launch's whole dependence cone. */
            const int margin = post ? 1 : 2;
            const bool trim = last;
            int w_lo = (!trim || sl.own_lo - margin < 0) ? 0 : sl.own_lo - margin;
            int w_hi = (!trim || sl.own_hi + margin > lv.g.ni) ? lv.g.ni : sl.own_hi + margin;
            /* a pure residual launch (S == 0) only has to produce what is consumed: the norm and the on-the-fly
             * restriction need the owned planes (their neighbours come from the pipeline's warm-up / drain), a
             * stored r one more plane on either side */
            if (S == 0) {
                const int pad = (want_res == 2 && !rst) ? 1 : 0;
                w_lo = sl.own_lo - pad < 0 ? 0 : sl.own_lo - pad;
                w_hi = sl.own_hi + pad > lv.g.ni ? lv.g.ni : sl.own_hi + pad;
            }
            if (tap && last) {
                /* output: the owned planes only -- the four passes use up the halo planes the first launch has left, and
                 * the next launch (one pass + residual + restriction) gets three fresh ones by exchange first */
                const int np = k_sweep_tap(cx->opt, lv.g, lv.f[MG3D_U], lv.f[MG3D_D], lv.alt, cx->partials, MG3D_MAX_PARTIALS, lv.h, c1, s,
                                           sl.own_lo, sl.own_hi, sl.own_lo, sl.own_hi);
                if (np < 0)
                    return fail(MG3D_ERR_STATE, "slab sweep: no kernel for four passes + norm tap on level %d", l);
                k_fold(cx->partials, np, cx->sumsq, s);
                continue;
            }
            const int np = k_sweep(cx->opt, lv.g, (zero_in && first) ? nullptr : lv.f[MG3D_U], lv.f[MG3D_D], lv.alt,
                                   (res && want_res == 2 && !rst) ? lv.f[MG3D_R] : nullptr,
                                   (res && want_res == 1) ? cx->partials : nullptr, /* the pre-smoothing norm is dropped (:1294) */
                                   MG3D_MAX_PARTIALS, lv.h, S, c1, res, s, sl.own_lo, sl.own_hi,
                                   rst ? tgt[ri].gc : nullptr, rst ? tgt[ri].dc : nullptr, rst ? tgt[ri].lo : -1,
                                   rst ? tgt[ri].hi : -1, (pro && first) ? pro[ri].gc : nullptr,
                                   (pro && first) ? pro[ri].ec : nullptr, w_lo, w_hi);
            if (np < 0) /* nothing was launched: no buffer swap, no fold */
                return fail(MG3D_ERR_STATE, "slab sweep: no kernel for %d colour passes%s on level %d", S,
                            res ? " + residual" : "", l);
            if (res && want_res == 1)
                k_fold(cx->partials, np, cx->sumsq, s);
        }
        if (S > 0)
            for (auto &R : D->rs) {
                Level &lv = SL(D, R, l).lv;
                double *t = lv.f[MG3D_U];
                lv.f[MG3D_U] = lv.alt;
                lv.alt = t;
            }
        if (res)
            done_res = true;
        passes -= S;
        first = false;
    }
    if (refresh_u) /* the last launch smoothed and took the norm in one go */
        CHK(start_u_exchange(D, MG3D_XK_HALO_U_NEXT, l));
    return MG3D_OK;
}

/* carried cycles on slabs (csrc/mg3d_ctx.hip has the argument): same conditions as the single-domain path */
static bool dist_carry_policy(mg3d_dist *D) /* what the options say (carry, carry_min), for this level geometry */
{
    const mg3d_options &o = D->rs[0].coarse->opt;
    const Geom &g = SL(D, D->rs[0], D->L - 1).lv.g;
    return o.v[MG3D_OPT_CARRY] != 0 && g.N >= o.v[MG3D_OPT_CARRY_MIN];
}

/* one launch per leg on slabs (csrc/mg3d_ctx.hip "one launch per leg"; options legs, legs_min): same conditions as the carried
 * cycles, which it replaces where both apply */
static bool dist_legs_policy(mg3d_dist *D)
{
    const mg3d_options &o = D->rs[0].coarse->opt;
    const Geom &g = SL(D, D->rs[0], D->L - 1).lv.g;
    return o.v[MG3D_OPT_LEGS] != 0 && g.N >= o.v[MG3D_OPT_LEGS_MIN];
}

static bool dist_can_legs(mg3d_dist *D)
{
    if (!(D->legs_fixed ? D->legs_on : dist_legs_policy(D)))
        return false;
    const Geom &g = SL(D, D->rs[0], D->L - 1).lv.g;
    return D->nu == 2 && D->H >= 5 && !D->rs[0].coarse->keep_r && g.N > 65 && (g.nj & 1) != 0;
}

static bool dist_can_carry(mg3d_dist *D)
{
    if (!(D->carry_fixed ? D->carry_on : dist_carry_policy(D)))
        return false;
    const Geom &g = SL(D, D->rs[0], D->L - 1).lv.g;
    return D->nu == 2 && !D->rs[0].coarse->keep_r && g.N > 65 && (g.nj & 1) != 0 && dist_split_up_leg(D, 1, 1);
}

static int dist_refuse_poisoned(const mg3d_dist *D, const char *who)
{
    if (D->poisoned)
        return fail(MG3D_ERR_STATE, "%s: an earlier mg3d_dist_vcycles call failed after one of its cycles had run ahead into the "
                                    "next: u of the finest level is mid-cycle; upload it again first", who);
    return MG3D_OK;
}

/* carry_out: another cycle of this call follows (it may be enqueued ahead into); legs_out: it follows in the same batch of norm
 * slots (its down-leg completes this cycle's norm into slot `slot`, which the batch then reads back) */
static int dist_enqueue_vcycle(mg3d_dist *D, int slot, bool carry_out = false, bool legs_out = false)
{
    hipStream_t s = D->stream;
    const int L = D->L, ld = D->ld;
    D->phase = 0;
    const bool can_legs = dist_can_legs(D), legs_in = D->legs_pending;
    const bool can = !can_legs && dist_can_carry(D), carry_in = D->carried;
    if ((carry_in && !can) || (legs_in && (!can_legs || slot < 1))) {
        D->carried = false;
        D->legs_pending = false;
        return fail(MG3D_ERR_STATE, "mg3d_dist_vcycles: carried state met a cycle that cannot continue it");
    }
    carry_out = carry_out && can;
    legs_out = legs_out && can_legs;
    D->cur = legs_out ? &D->plans_legs[legs_in ? 1 : 0] : legs_in ? &D->plans_legs[2] : carry_out ? &D->plans_carry : &D->plans;
    D->carried = false;
    D->legs_pending = false;
    /* behind a finished cycle of an earlier call (red_tail): the one-launch down-leg without a norm half */
    const bool red_in = !legs_in && !carry_in && can_legs && D->red_tail;
    D->red_tail = false;
    DistScope cycle_timer(D, 0, s);
    if (D->timing)
        D->t_cycles++;
    for (auto &R : D->rs)
        if (!R.coarse->have_lu)
            return fail(MG3D_ERR_STATE, "mg3d_dist_vcycles: no coarse LU set (mg3d_dist_build_coarse)");
    std::vector<RestrictTarget> tgt(D->rs.size());
    /* ---- down: distributed levels.  On entry the halos of d are exact on all levels (upload / the exchange
     * that follows each restriction below) and those of the finest u have been refreshed since it last
     * changed (upload, or the exchange started at the end of the previous cycle). */
    for (int l = L - 1; l >= ld; l--) {
        for (size_t ri = 0; ri < D->rs.size(); ri++) {
            RankState &R = D->rs[ri];
            SlabLevel &sl = SL(D, R, l);
            /* mg_3d.h:1258: zero guess below the finest level -- folded into the first sweep launch */
            /* owned coarse planes (plus the physical boundary planes at the ends of the domain) */
            RestrictTarget &t = tgt[ri];
            if (l - 1 >= ld) {
                SlabLevel &sc = SL(D, R, l - 1);
                t.gc = &sc.lv.g;
                t.dc = sc.lv.f[MG3D_D];
                t.lo = sc.own_lo;
                t.hi = sc.own_hi;
            } else {
                Level &lc = R.coarse->lv[ld - 1];
                t.gc = &lc.g;
                t.dc = lc.f[MG3D_D];
                t.lo = R.rank == 0 ? 0 : sl.glo / 2;
                t.hi = R.rank == D->P - 1 ? lc.g.N : sl.ghi / 2;
            }
        }
        const bool keep = D->rs[0].coarse->keep_r;
        std::vector<RestrictTarget> none(D->rs.size(), RestrictTarget{nullptr, nullptr, -1, -1});
        CHK(await_u(D, l));
        if (l == L - 1 && (legs_in || red_in)) {
            /* the down-leg as ONE launch over the owned planes: three passes (black first -- the cycle's first red pass is the
             * identity behind the previous cycle's last one), residual, restriction; it reads five planes either side, which
             * the previous cycle's last exchange refreshed.  On its way it forms the black half of the previous cycle's
             * residual norm, whose red half the one-launch up-leg left in partials[0 .. legs_npa): folded and reduced here. */
            for (size_t ri = 0; ri < D->rs.size(); ri++) {
                RankState &R = D->rs[ri];
                SlabLevel &sl = SL(D, R, l);
                Level &lv = sl.lv;
                mg3d_ctx *cx = R.coarse;
                double *part_b = red_in ? nullptr : cx->partials + MG3D_MAX_PARTIALS / 2;
                const int npb = k_sweep_leg_down(cx->opt, lv.g, lv.f[MG3D_U], lv.f[MG3D_D], lv.alt, *tgt[ri].gc, tgt[ri].dc, lv.h, 3, part_b,
                                                 MG3D_MAX_PARTIALS / 2, s, sl.own_lo, sl.own_hi, tgt[ri].lo, tgt[ri].hi, sl.own_lo, sl.own_hi);
                if (npb <= 0)
                    return fail(MG3D_ERR_STATE, "slab sweep: no kernel for the one-launch down-leg on level %d", l);
                if (!red_in)
                    k_fold2(cx->partials, R.legs_npa, part_b, npb, cx->sumsq, s);
                double *t = lv.f[MG3D_U];
                lv.f[MG3D_U] = lv.alt;
                lv.alt = t;
            }
            if (!red_in)
                CHK(reduce_norm(D, slot - 1));
        } else if (l == L - 1 && carry_in) {
            /* the one pre-smoothing pass that is left (black) + residual + restriction in one launch over the owned planes:
             * it reads three planes either side, which the previous cycle's last exchange refreshed */
            for (size_t ri = 0; ri < D->rs.size(); ri++) {
                SlabLevel &sl = SL(D, D->rs[ri], l);
                Level &lv = sl.lv;
                const int np = k_sweep(D->rs[ri].coarse->opt, lv.g, lv.f[MG3D_U], lv.f[MG3D_D], lv.alt, nullptr, nullptr, MG3D_MAX_PARTIALS, lv.h, 1, 0,
                                       true, s, sl.own_lo, sl.own_hi, tgt[ri].gc, tgt[ri].dc, tgt[ri].lo, tgt[ri].hi, nullptr,
                                       nullptr, sl.own_lo, sl.own_hi);
                if (np < 0)
                    return fail(MG3D_ERR_STATE, "slab sweep: no kernel for one pass + residual + restriction on level %d", l);
                double *t = lv.f[MG3D_U];
                lv.f[MG3D_U] = lv.alt;
                lv.alt = t;
            }
        } else
        /* :1282 + :1294 + :1310 (interior of the coarse rhs on the fly unless r is to be kept) */
        CHK(stage_smooth(D, l, 0, 2, keep ? none.data() : tgt.data(), l < L - 1));
        for (size_t ri = 0; ri < D->rs.size(); ri++) {
            SlabLevel &sl = SL(D, D->rs[ri], l);
            k_restrict(sl.lv.g, sl.lv.f[MG3D_R], *tgt[ri].gc, tgt[ri].dc, s, tgt[ri].lo, tgt[ri].hi, !keep);
        }
        /* the coarser level starts from its right-hand side at once: only owned planes were produced */
        if (l - 1 < ld)
            CHK(exchange(D, (D->policy & 1) ? MG3D_XK_RHS_GATHER : MG3D_XK_RHS_ALLGATHER, ld - 1, false));
        else
            CHK(exchange(D, MG3D_XK_HALO_D, l - 1, false));
        /* u of this level is final until the prolongation on the way up: refresh its halos underneath the
         * coarser levels (issued behind the right-hand side's exchange: one communicator, one in-order stream) */
        CHK(start_u_exchange(D, MG3D_XK_HALO_U_DOWN, l));
    }
    /* ---- replicated levels: the ordinary V-cycle from level ld-1 (its guess zeroed first, :1258), on every rank -- or,
     * with MG3D_COARSE_GATHER=1, on rank 0 alone, whose correction is then broadcast (same bits either way: the ranks
     * would have computed identical copies) */
    {
        DistScope timer(D, 3, s);
        for (auto &R : D->rs) {
            if ((D->policy & 1) && R.rank != 0)
                continue;
            Level &lc = R.coarse->lv[ld - 1];
            (void)hipMemsetAsync(lc.f[MG3D_U], 0, lc.elems * sizeof(double), s);
            R.coarse->red_tail = false; /* u and d of the context's top level were rewritten behind its back (here; the restriction) */
            if (ld - 1 == 0)
                CHK(mg3d_coarse_solve(R.coarse));
            else
                CHK(mg3d_enqueue_vcycle(R.coarse, ld - 1, R.coarse->sumsq_slots - 1));
        }
    }
    if (D->policy & 1)
        CHK(exchange(D, MG3D_XK_CORR_BCAST, ld - 1, false));
    /* ---- up */
    for (int l = ld; l < L; l++) {
        /* :1331 on every local plane, halos included: the correction's halos (post-smoothed on owned +-1 only)
         * are refreshed first, those of u have been under way since the pre-smoother */
        if (l - 1 >= ld)
            CHK(exchange(D, MG3D_XK_HALO_U_UP, l - 1, false));
        CHK(await_u(D, l));
        const int want = l == L - 1 ? 1 : 0;
        /* the prolongation rides on the post-smoother's first launch: the top level's split stage, and (option fuse_up_max, as
         * on a single domain) the four-pass launch of a V(2,2) cycle on the levels below it */
        const Geom &gl = SL(D, D->rs[0], l).lv.g;
        const bool fold = dist_split_up_leg(D, 1, want) ||
                          (D->nu == 2 && want == 0 && (gl.nj & 1) != 0 && gl.N <= D->rs[0].coarse->opt.v[MG3D_OPT_FUSE_UP_MAX]);
        std::vector<ProlongSource> pro(D->rs.size());
        for (size_t ri = 0; ri < D->rs.size(); ri++) {
            RankState &R = D->rs[ri];
            SlabLevel &sl = SL(D, R, l);
            Level &lc = l - 1 >= ld ? SL(D, R, l - 1).lv : R.coarse->lv[ld - 1];
            pro[ri] = ProlongSource{&lc.g, lc.f[MG3D_U]};
            if (!fold)
                k_prolong(lc.g, lc.f[MG3D_U], sl.lv.g, sl.lv.f[MG3D_U], s, 0, sl.lv.g.ni);
        }
        /* :1341 (+ :1354 at the top level): all H halo planes of u are exact here, the post-smoother uses up
         * 2*nu of them.  The next cycle's pre-smoother wants fresh halos on the finest u: that exchange starts
         * underneath the norm kernel, which reads the first halo plane on either side -- just produced
         * exactly by the post-smoother, so the exchange leaves that plane alone. */
        if (l == L - 1 && legs_out) {
            /* the up-leg as ONE launch: prolongation + four passes over the owned planes (it uses up four of the H halo planes of
             * u and of the correction), the red half of the norm from the last pass's own sums.  With a communication stream the
             * first and last five planes -- what the exchange for the next down-leg sends -- are made first, as one launch of two
             * chunks per rank; the exchange then runs underneath the launch that makes the interior. */
            const int E = 5;
            bool edge_first = D->overlap && D->P > 1;
            for (auto &R : D->rs)
                edge_first = edge_first && SL(D, R, l).own_hi - SL(D, R, l).own_lo >= 2 * E + 2;
            auto up = [&](size_t ri, const double *vin, double *vout, double *part, int maxp, int lo, int hi, int edge) -> int {
                SlabLevel &sl = SL(D, D->rs[ri], l);
                Level &lv = sl.lv;
                return k_sweep_leg_up(D->rs[ri].coarse->opt, lv.g, vin, lv.f[MG3D_D], vout, *pro[ri].gc, pro[ri].ec, lv.h, part, maxp, s,
                                      sl.own_lo, sl.own_hi, lo, hi, edge);
            };
            std::vector<int> n1(D->rs.size(), 0);
            for (size_t ri = 0; ri < D->rs.size(); ri++) {
                SlabLevel &sl = SL(D, D->rs[ri], l);
                n1[ri] = up(ri, sl.lv.f[MG3D_U], sl.lv.alt, D->rs[ri].coarse->partials, MG3D_MAX_PARTIALS / 4, sl.own_lo, sl.own_hi, edge_first ? E : 0);
                if (n1[ri] <= 0)
                    return fail(MG3D_ERR_STATE, "slab sweep: no kernel for the one-launch up-leg on level %d", l);
            }
            for (auto &R : D->rs) { /* the exchange sends from (and lands in) the NEW buffer */
                Level &lv = SL(D, R, l).lv;
                double *t = lv.f[MG3D_U];
                lv.f[MG3D_U] = lv.alt;
                lv.alt = t;
            }
            CHK(start_u_exchange(D, MG3D_XK_HALO_U_NEXT, l));
            for (size_t ri = 0; ri < D->rs.size(); ri++) {
                SlabLevel &sl = SL(D, D->rs[ri], l);
                int n2 = 0;
                if (edge_first) {
                    n2 = up(ri, sl.lv.alt, sl.lv.f[MG3D_U], D->rs[ri].coarse->partials + n1[ri], MG3D_MAX_PARTIALS / 2 - n1[ri], sl.own_lo + E,
                            sl.own_hi - E, 0);
                    if (n2 <= 0)
                        return fail(MG3D_ERR_STATE, "slab sweep: no kernel for the interior of the one-launch up-leg on level %d", l);
                }
                D->rs[ri].legs_npa = n1[ri] + n2;
            }
            D->legs_pending = true;
            D->n_legs++;
            continue;
        }
        CHK(stage_smooth(D, l, 1, want, nullptr, false, l == L - 1, fold ? pro.data() : nullptr, l == L - 1 && carry_out));
        if (l == L - 1 && carry_out) {
            D->carried = true;
            D->n_carried++;
        }
    }
    if (!legs_out)
        CHK(reduce_norm(D, slot));
    /* a whole V(2,2) cycle has ended the ordinary way: its last pass was red, the exchange behind it leaves all H halo planes
     * of u exact (plain plan: planes 2..H by exchange, plane 1 as the post-smoother made it) */
    if (!legs_out && !carry_out && D->nu == 2)
        D->red_tail = true;
    if (D->phase != (int)(*D->cur)[0].kind.size())
        return fail(MG3D_ERR_STATE, "slab schedule ended after %d of the plan's %d phases", D->phase, (int)(*D->cur)[0].kind.size());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(MG3D_ERR_HIP, "mg3d_dist_vcycles: kernel launch failed: %s", hipGetErrorString(e));
    return MG3D_OK;
}

/* after the last cycle the compute stream joins the exchange of the finest u halos that the cycle started */
static int dist_finish(mg3d_dist *D) { return await_u(D, D->L - 1); }

extern "C" int mg3d_dist_vcycles(mg3d_dist *D, int count, double *norms)
{
    if (!D || count < 0)
        return fail(MG3D_ERR_ARG, "mg3d_dist_vcycles: bad arguments");
    CHK(dist_refuse_poisoned(D, "mg3d_dist_vcycles"));
    const int carried_before = D->n_carried;
    /* any error return below: if a cycle of this call has carried, u of the finest level may be three passes into a cycle
     * that was never finished (also across a batch boundary, where D->carried is still set) -- see `poisoned` */
    struct Guard {
        mg3d_dist *d;
        int before;
        bool ok;
        ~Guard()
        {
            if (!ok && (d->carried || d->n_carried != before))
                d->poisoned = true;
            if (!ok)
                d->carried = false, d->legs_pending = false, d->red_tail = false; /* (a half-formed norm is simply lost: u itself is a finished cycle's) */
        }
    } guard{D, carried_before, false};
    for (int done = 0; done < count;) {
        const int nb = (count - done < D->norm_slots) ? count - done : D->norm_slots;
        for (int c = 0; c < nb; c++) {
            /* every cycle but the last of a call ends ahead into the next one (a call never ends in the carried state) */
            CHK(dist_enqueue_vcycle(D, c, done + c + 1 < count, c + 1 < nb));
        }
        CHK(dist_finish(D));
        HIPCHK(hipMemcpyAsync(D->h_norms, D->d_norms, nb * sizeof(double), hipMemcpyDeviceToHost, D->stream));
        HIPCHK(hipStreamSynchronize(D->stream));
        HIPCHK(hipStreamSynchronize(D->comm_stream));
        dist_resolve_timers(D);
        /* an RCCL failure that surfaced asynchronously (a peer died, a transport error): an error here, not a hang or a
         * wrong number later */
        for (ncclComm_t cm : {D->have_comm ? D->comm : nullptr})
            if (cm) {
                ncclResult_t ae = ncclSuccess;
                NCCLCHK(ncclCommGetAsyncError(cm, &ae));
                if (ae != ncclSuccess)
                    return fail(MG3D_ERR_HIP, "RCCL reported an asynchronous error: %s", ncclGetErrorString(ae));
            }
        if (norms)
            for (int c = 0; c < nb; c++)
                norms[done + c] = sqrt(D->h_norms[c]);
        done += nb;
    }
    guard.ok = true;
    return MG3D_OK;
}
