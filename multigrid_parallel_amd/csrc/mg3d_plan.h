/* mg3d_plan.h -- the exchange plan of the slab paths (include/mg3d.h: mg3d_dist_plan / mg3d32_dist_plan) and the one
 * executor both transports of both precisions run it through.  Shared by mg3d_dist.hip and mg3d_f32_dist.hip; not
 * installed.  The generator half is host arithmetic only. */
#ifndef MG3D_PLAN_H
#define MG3D_PLAN_H

#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "mg3d.h"

int mg3d_fail(int code, const char *fmt, ...);

struct PlanGeom {
    int c, L, P, nu, H, ld;
    int pitch_align; /* row pitch of the device layout in elements: 16 doubles / 32 floats (128-byte rows) */
};

struct Plan {
    std::vector<mg3d_xfer> e;
    std::vector<int> begin;       /* first entry of phase p; begin[nphases] = e.size() */
    std::vector<int> kind, level; /* what phase p is, also for a rank that takes no part in it */
};

static inline long long plan_plane_elems(const PlanGeom &G, int N)
{
    return (long long)((N + G.pitch_align - 1) / G.pitch_align * G.pitch_align) * N;
}

static inline void slab_local(const PlanGeom &G, int level, int rank, int *own_lo, int *own_hi, long long *plane_elems)
{
    int glo = 0, ghi = 0;
    mg3d_slab_owned(G.c, G.L, G.P, G.H, level, rank, &glo, &ghi);
    const int h_lo = rank > 0 ? G.H : 0;
    *own_lo = h_lo;
    *own_hi = h_lo + (ghi - glo);
    *plane_elems = plan_plane_elems(G, (G.c - 1) * (1 << level) + 1);
}

/* the coarse planes of the first replicated level (ld-1) that rank r restricts into: those under its owned fine planes */
static inline void plan_coarse_range(const PlanGeom &G, int r, int *lo, int *hi)
{
    int flo = 0, fhi = 0;
    mg3d_slab_owned(G.c, G.L, G.P, G.H, G.ld, r, &flo, &fhi);
    const int Nc = ((G.c - 1) << (G.ld - 1)) + 1;
    *lo = r == 0 ? 0 : flo / 2;
    *hi = r == G.P - 1 ? Nc : fhi / 2;
}

static inline int plan_open_phase(Plan &pl, int kind, int level)
{
    pl.begin.push_back((int)pl.e.size());
    pl.kind.push_back(kind);
    pl.level.push_back(level);
    return (int)pl.kind.size() - 1;
}

/* Halo planes skip+1 .. skip+n (counted from the slab's owned planes; n < 0: up to H) of `field` on distributed level l
 * from the neighbours' owned planes; skip = 0, n < 0 is the whole halo.  Halo plane t of the upper side is the upper
 * neighbour's t-th owned plane, of the lower side the lower neighbour's t-th from the top. */
static inline void plan_halo(Plan &pl, const PlanGeom &G, int kind, int field, int level, int rank, int skip, int stream,
                             int n = -1)
{
    const int phase = plan_open_phase(pl, kind, level);
    if (n < 0 || skip + n > G.H)
        n = G.H - skip;
    if (n <= 0)
        return;
    int lo, hi;
    long long pe;
    slab_local(G, level, rank, &lo, &hi, &pe);
    if (rank + 1 < G.P) {
        pl.e.push_back(mg3d_xfer{phase, kind, MG3D_XOP_SEND, rank + 1, field, level, hi - skip - n, n, pe, stream});
        pl.e.push_back(mg3d_xfer{phase, kind, MG3D_XOP_RECV, rank + 1, field, level, hi + skip, n, pe, stream});
    }
    if (rank > 0) {
        pl.e.push_back(mg3d_xfer{phase, kind, MG3D_XOP_SEND, rank - 1, field, level, lo + skip, n, pe, stream});
        pl.e.push_back(mg3d_xfer{phase, kind, MG3D_XOP_RECV, rank - 1, field, level, lo - skip - n, n, pe, stream});
    }
}

/* the d planes of the first replicated level every owner produced, to every rank: one broadcast per owner */
static inline void plan_rhs_allgather(Plan &pl, const PlanGeom &G)
{
    const int lc = G.ld - 1, Nc = ((G.c - 1) << lc) + 1;
    const int ph = plan_open_phase(pl, MG3D_XK_RHS_ALLGATHER, lc);
    if (G.P <= 1)
        return;
    for (int root = 0; root < G.P; root++) {
        int lo, hi;
        plan_coarse_range(G, root, &lo, &hi);
        pl.e.push_back(mg3d_xfer{ph, MG3D_XK_RHS_ALLGATHER, MG3D_XOP_BCAST, root, MG3D_D, lc, lo, hi - lo, plan_plane_elems(G, Nc), 0});
    }
}

static inline void plan_norm(Plan &pl, const PlanGeom &G, int level, int rank)
{
    const int ph = plan_open_phase(pl, MG3D_XK_NORM, level);
    if (G.P > 1)
        pl.e.push_back(mg3d_xfer{ph, MG3D_XK_NORM, MG3D_XOP_ALLGATHER, -1, -1, level, rank, 1, 1, 0});
}

#define PLAN_HIPCHK(call)                                                                                       \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess)                                                                                   \
            return mg3d_fail(MG3D_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define PLAN_NCCLCHK(call)                                                                                       \
    do {                                                                                                         \
        ncclResult_t e_ = (call);                                                                                \
        if (e_ != ncclSuccess)                                                                                   \
            return mg3d_fail(MG3D_ERR_HIP, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

/* Executes phase `ph` of the plans of all local ranks (loopback: plans[r] is rank r's; RCCL: plans[0] is this
 * process's) on stream s.  RCCL: one group of the rank's sends / receives / broadcasts / all-gather exactly as listed.
 * Loopback: every send is copied into the receive entry that names it in the peer's plan of the same phase (counts must
 * agree), every broadcast range from the root's array into all others, the norm parts into rank 0's gather buffer.
 * base(ri, entry) -> first element of the entry's (field, level) array on local rank index ri; sumsq(ri) -> that rank's
 * partial sum of squares; gather: P doubles (the all-gather's destination). */
template <class T, class Base, class SumSq>
static int plan_run(const std::vector<Plan> &plans, int ph, bool loopback, ncclComm_t comm, ncclDataType_t dt, hipStream_t s,
                    Base base, SumSq sumsq, double *gather)
{
    if (loopback) {
        for (size_t ri = 0; ri < plans.size(); ri++) {
            const Plan &pl = plans[ri];
            for (int i = pl.begin[(size_t)ph]; i < pl.begin[(size_t)ph + 1]; i++) {
                const mg3d_xfer &e = pl.e[(size_t)i];
                if (e.op == MG3D_XOP_SEND) {
                    const Plan &pp = plans[(size_t)e.peer];
                    const mg3d_xfer *m = nullptr;
                    for (int k = pp.begin[(size_t)ph]; k < pp.begin[(size_t)ph + 1]; k++) {
                        const mg3d_xfer &c = pp.e[(size_t)k];
                        if (c.op == MG3D_XOP_RECV && c.peer == (int)ri && c.field == e.field && c.level == e.level) {
                            if (m)
                                return mg3d_fail(MG3D_ERR_STATE, "exchange plan: two receives match one send (phase %d)", ph);
                            m = &c;
                        }
                    }
                    if (!m || m->count != e.count || m->plane_elems != e.plane_elems)
                        return mg3d_fail(MG3D_ERR_STATE, "exchange plan: send of rank %d to %d in phase %d has no receive of its size",
                                         (int)ri, e.peer, ph);
                    PLAN_HIPCHK(hipMemcpyAsync(base((size_t)e.peer, *m) + m->plane_elems * m->offset, base(ri, e) + e.plane_elems * e.offset,
                                               (size_t)e.count * e.plane_elems * sizeof(T), hipMemcpyDeviceToDevice, s));
                } else if (e.op == MG3D_XOP_BCAST && (int)ri == e.peer) {
                    for (size_t dst = 0; dst < plans.size(); dst++)
                        if (dst != ri)
                            PLAN_HIPCHK(hipMemcpyAsync(base(dst, e) + e.plane_elems * e.offset, base(ri, e) + e.plane_elems * e.offset,
                                                       (size_t)e.count * e.plane_elems * sizeof(T), hipMemcpyDeviceToDevice, s));
                } else if (e.op == MG3D_XOP_ALLGATHER && ri == 0) {
                    for (size_t r = 0; r < plans.size(); r++)
                        PLAN_HIPCHK(hipMemcpyAsync(gather + r, sumsq(r), sizeof(double), hipMemcpyDeviceToDevice, s));
                }
            }
        }
        return MG3D_OK;
    }
    const Plan &pl = plans[0];
    if (pl.begin[(size_t)ph] == pl.begin[(size_t)ph + 1])
        return MG3D_OK;
    PLAN_NCCLCHK(ncclGroupStart());
    /* a failure between ncclGroupStart and ncclGroupEnd must not leave the communicator inside an open group -- every later
     * call on it (the asynchronous-error check, teardown) would be queued and never issued: the process would hang
     * instead of reporting.  The first error is kept, the group is closed (its own result ignored), then it is returned. */
    ncclResult_t first = ncclSuccess;
    const char *what = "";
    auto in_group = [&](ncclResult_t r, const char *call) {
        if (r != ncclSuccess && first == ncclSuccess) {
            first = r;
            what = call;
        }
    };
    for (int i = pl.begin[(size_t)ph]; i < pl.begin[(size_t)ph + 1] && first == ncclSuccess; i++) {
        const mg3d_xfer &e = pl.e[(size_t)i];
        const size_t cnt = (size_t)e.count * e.plane_elems;
        if (e.op == MG3D_XOP_ALLGATHER) {
            in_group(ncclAllGather(sumsq(0), gather, cnt, ncclDouble, comm, s), "ncclAllGather");
            continue;
        }
        T *p = base(0, e) + e.plane_elems * e.offset;
        if (e.op == MG3D_XOP_SEND)
            in_group(ncclSend(p, cnt, dt, e.peer, comm, s), "ncclSend");
        else if (e.op == MG3D_XOP_RECV)
            in_group(ncclRecv(p, cnt, dt, e.peer, comm, s), "ncclRecv");
        else
            in_group(ncclBroadcast(p, p, cnt, dt, e.peer, comm, s), "ncclBroadcast");
    }
    if (first != ncclSuccess) {
        (void)ncclGroupEnd();
        return mg3d_fail(MG3D_ERR_HIP, "exchange plan, phase %d: %s: %s", ph, what, ncclGetErrorString(first));
    }
    PLAN_NCCLCHK(ncclGroupEnd());
    return MG3D_OK;
}

#endif
