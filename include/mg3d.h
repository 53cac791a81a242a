/*
 * mg3d.h -- C ABI of libmg3d.so, the MI355X (gfx950) implementation of the
 * reference's 3D geometric-multigrid V-cycle (knram06/multigrid_parallel,
 * mg_3d.h).  Plain C: opaque context, raw pointers, sizes, int status codes.
 * No C++ exceptions, no torch types cross this boundary.
 *
 * The reference has no FFI; its boundary is textual inclusion of mg_3d.h
 * (test_mg_3d.c:4-6).  include/mg_3d.h in this repo is the drop-in header
 * whose functions forward to the entry points below.  Every entry point cites
 * the reference definition it replaces (file:line under the reference tree).
 *
 * All grids are fp64, vertex centred, idx = N*N*i + N*j + k with k contiguous
 * on the HOST side (mg_3d.h:43-44).  On the device every level is kept in a
 * padded layout (k-pitch a multiple of 16 doubles); mg3d_upload/download
 * convert.  There is NO CPU fallback: every compute entry point returns
 * MG3D_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef MG3D_H
#define MG3D_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mg3d_ctx mg3d_ctx;

enum {
    MG3D_OK = 0,
    MG3D_ERR_ARG = 1,       /* bad argument (NULL, size, level out of range) */
    MG3D_ERR_NO_DEVICE = 2, /* no HIP device: the product has no CPU path */
    MG3D_ERR_HIP = 3,       /* a HIP runtime call failed; see mg3d_last_error() */
    MG3D_ERR_ALLOC = 4,
    MG3D_ERR_STATE = 5      /* e.g. V-cycle requested before the coarse LU was set */
};

/* fields of a level (mg_3d.h:26: double **u, **d, **r) */
enum { MG3D_U = 0, MG3D_D = 1, MG3D_R = 2 };

/* stages of the per-level timing table (mg_3d.h:136-137) */
enum {
    MG3D_ST_SMOOTH1 = 0, MG3D_ST_RESIDUAL1, MG3D_ST_RESTRICT, MG3D_ST_RECURSE,
    MG3D_ST_PROLONG, MG3D_ST_SMOOTH2, MG3D_ST_RESIDUAL2, MG3D_NUM_STAGES
};

const char *mg3d_last_error(void);     /* thread-local text of the last failure */
const char *mg3d_stage_name(int stage); /* "Smoother1", ... (mg_3d.h:136-137) */
int mg3d_device_count(void);            /* 0 when no usable device */

/* ------------------------------------------------------------------ context
 * Replaces SolverInitialize's allocation (mg_3d.h:107-144, 30-48): three
 * hierarchies u,d,r of num_levels levels, level l having
 * ((coarse_pts-1)*2^l+1)^3 points, all zero; spacing = grid_length/(N-1). */
int mg3d_ctx_create(int coarse_pts, int num_levels, int smooth_iters, double grid_length, mg3d_ctx **out);
int mg3d_ctx_destroy(mg3d_ctx *ctx); /* SolverFinalize, mg_3d.h:1452-1467 */
int mg3d_ctx_num_levels(const mg3d_ctx *ctx);
int mg3d_ctx_level_n(const mg3d_ctx *ctx, int level); /* points per side, mg_3d.h:41 */
double mg3d_ctx_level_h(const mg3d_ctx *ctx, int level);
int mg3d_ctx_set_smooth_iters(mg3d_ctx *ctx, int iters);
/* keep != 0: every V-cycle materialises the residual arrays r[level] as the reference does (mg_3d.h:1294).
 * Default 0: the residual is restricted into the coarse right-hand side on the fly and r is not written --
 * u, d and the returned norms are identical either way.  mg3d_host_vcycle always keeps r. */
int mg3d_ctx_set_keep_residual(mg3d_ctx *ctx, int keep);

/* Coarsest operator.  mg3d_ctx_build_coarse = constructCoarseMatrixA +
 * convertToLU_InPlace as SolverGetDetails does (mg_3d.h:282-289), with the
 * spacing the caller chooses (mg_3d.h:287 passes h*2^(L-1);
 * test_mg_3d_dirichlet.c:40 passes the finest h).  mg3d_ctx_set_lu installs a
 * caller-factored row-major LU (n = coarse_pts^3) instead. */
int mg3d_ctx_build_coarse(mg3d_ctx *ctx, double h_coarse);
int mg3d_ctx_set_lu(mg3d_ctx *ctx, const double *LU);

/* ------------------------------------------------------------ data movement
 * Host arrays are dense N^3 (reference layout). */
int mg3d_upload(mg3d_ctx *ctx, int field, int level, const double *host);
int mg3d_download(mg3d_ctx *ctx, int field, int level, double *host);
int mg3d_zero(mg3d_ctx *ctx, int field, int level); /* memset of mg_3d.h:1258-1259 */
int mg3d_sync(mg3d_ctx *ctx);
/* raw device view of a level for callers that share device memory (tests, bench) */
int mg3d_device_view(mg3d_ctx *ctx, int field, int level, void **dev_ptr, int *pitch_doubles, long *plane_doubles);

/* ------------------------------------------------ operators on device levels
 * mg3d_smooth     : preSmoother (post=0, mg_3d.h:640-709: iters x red,black)
 *                   postSmoother (post=1, mg_3d.h:711-781: iters x black,red)
 * mg3d_residual   : calculateResidual (mg_3d.h:794-842); store!=0 writes r on
 *                   the interior; *norm = sqrt(sum diff^2) (may be NULL)
 * mg3d_restrict   : restrictResidual r[level] -> d[level-1] (mg_3d.h:844-998)
 * mg3d_prolong    : prolongateAndCorrectError u[level-1] -> u[level] (mg_3d.h:1000-1145)
 * mg3d_coarse_solve: solveWithLU(LU, n, d[0], u[0]) (gauss_elim.h:31-60)
 * mg3d_l2norm     : GetL2NormOfVector over all N^3 entries (mg_3d.h:783-792) */
int mg3d_smooth(mg3d_ctx *ctx, int level, int post, int iters);
int mg3d_residual(mg3d_ctx *ctx, int level, int store, double *norm);
/* smoother followed by the residual of its result, as vcycle does back to back (mg_3d.h:1282+1294,
 * 1341+1354), in ONE pass over the level (fused sweep kernel) */
int mg3d_smooth_residual(mg3d_ctx *ctx, int level, int post, int iters, int store, double *norm);
/* smoother, residual and restriction of it into d[level-1] (mg_3d.h:1282+1294+1310) without storing r */
int mg3d_smooth_restrict(mg3d_ctx *ctx, int level, int iters);
int mg3d_restrict(mg3d_ctx *ctx, int level);
int mg3d_prolong(mg3d_ctx *ctx, int level);
int mg3d_coarse_solve(mg3d_ctx *ctx);
int mg3d_l2norm(mg3d_ctx *ctx, int field, int level, double *norm);

/* One V-cycle from `level` down (vcycle, mg_3d.h:1242-1362); *norm receives the
 * post-smoothing residual norm of `level` (the value SolverLinSolve returns,
 * mg_3d.h:1415-1420).  mg3d_vcycles runs `count` cycles from the finest level
 * back to back with a single host synchronisation at the end.
 * Consecutive V(2,2) cycles from a finest level of at least 257^3 share a launch ("carried cycles"): the
 * first red pre-smoothing pass of a cycle that follows another one is the identity, so a cycle's last two post-smoothing
 * passes, its norm and the next cycle's pre-smoothing passes run as one launch.  mg3d_vcycles does so inside a call;
 * mg3d_vcycle(finest level) ends ahead of itself, and every other entry point first restores the finished cycle's own u
 * (every grid value -- u, d, r of every level -- is that of separate cycles, bit for bit; the returned norm sums the same
 * squares in another grouping of per-block partial sums and agrees to the summation tolerance, <= 1e-13 relative against
 * the exactly rounded sum: tests/test_gpu_parity.py; MG3D_NO_CARRY=1 switches it off).
 * From 160 points per side (options legs, legs_min; round 4) the finest level runs ONE launch per leg instead --
 * prolongation + four passes, three passes + residual + restriction -- with the norm's two halves taken from the launches
 * on either side of it; behind a single mg3d_vcycle the next cycle's down-leg runs ahead into spare buffers and is swapped
 * back when anything else is asked (csrc/mg3d_ctx.hip, mg3d_can_legs).  Same bits; option legs = 0 keeps the carried cycles. */
int mg3d_vcycle(mg3d_ctx *ctx, int level, double *norm);
int mg3d_vcycles(mg3d_ctx *ctx, int count, double *norms);

/* Full-multigrid initialisation, SolverFMGInitialize (mg_dirichlet_analytic.c:771-806; commented copy
 * mg_3d.h:1364-1404): BCs on u[0], direct solve, then for every level prolong the coarser solution, impose the
 * Dirichlet values BCFunc on the six faces (on the device), zero the coarser level, one V-cycle from that level. */
int mg3d_fmg_initialize(mg3d_ctx *ctx);
/* setupBoundaryConditions (mg_3d.h:1147-1239) on a device-resident field */
int mg3d_fill_boundary(mg3d_ctx *ctx, int field, int level);

/* Launch and schedule policy of a context, by key.  Defaults, then the environment as an override read ONCE when the
 * context is created, afterwards only this call: nothing on a launch path reads the environment, two contexts of one
 * process may differ.  Grid values never depend on any of them (tests/test_gpu_parity.py, tests/test_gpu_legs.py compare).
 *   key             default  environment at creation   meaning
 *   carry           1        MG3D_NO_CARRY=1 -> 0      consecutive V(2,2) cycles share a launch on the finest level
 *   carry_min       130      MG3D_CARRY_MIN            ... from this many points per side
 *   legs            1        MG3D_LEGS                 one launch per leg on the finest level instead (see above)
 *   legs_min        160      MG3D_LEGS_MIN             ... from this many points per side (below: the carried cycles)
 *   tiny            1        MG3D_NO_TINY=1 -> 0       the level above the coarsest one in one workgroup
 *   tiny_cycle      1        MG3D_NO_TINY_CYCLE=1 -> 0 ... together with the direct solve in one launch
 *   lu_reduced      1        MG3D_LU_REDUCED           install the factor without its identity rows (read per factor)
 *   fuse_rst2       -1       MG3D_FUSE_RST2            2 passes + residual + restriction in one launch: -1 from 130 points
 *                                                      per side, 0 never, 1 always
 *   small_max       129      MG3D_SMALL_MAX            largest level side that runs the two-rows-per-thread shapes
 *   fuse_leg_max    0        MG3D_FUSE_LEG_MAX         largest level side whose legs run as one (two-row) launch each
 *   fuse_up_max     1048576  MG3D_FUSE_UP_MAX          largest level side whose up-leg folds the prolongation into 4 passes
 *                                                      (default: every level; 0: the prolongation is its own launch below the top)
 *   sweep_tune      -1       MG3D_SWEEP_TUNE           first-use measurement of chunk lengths: -1 on unless the process
 *                                                      drives a multi-rank RCCL job, 0 off, 1 on
 *   sweep_tune_log  0        MG3D_SWEEP_TUNE_LOG       print the measured choices
 *   sweep_ci        0        MG3D_SWEEP_CI             > 0: planes per chunk of every fused sweep launch (measurement)
 *   sweep_rj/nw/pf  0        MG3D_SWEEP_CFG="rj,nw,pf" another compiled tile shape (unknown ones fall back to the default)
 * mg3d_option_name(i) enumerates the keys (NULL past the end).  mg3d_dist_set_option forwards to every local rank (carry /
 * carry_min of a multi-rank job are agreed at creation and refuse to change); mg3d32_set_option knows "pairs", "fuse",
 * "carry" (MG3D_F32_NO_PAIRS / _NO_FUSE / _NO_CARRY at creation). */
int mg3d_ctx_set_option(mg3d_ctx *ctx, const char *key, int value);
int mg3d_ctx_get_option(const mg3d_ctx *ctx, const char *key, int *value);
const char *mg3d_option_name(int index);

/* per-stage timers (timing_info.h:6-47), filled from hipEvent pairs recorded in-stream (no stall).
 * on: 0 = off, 1 = every level, 2 = finest level only, 3 = the kernel timers of the finest level only (no stage
 * timers: a third of the marker packets, what bench.py's roofline object needs), 4 + k = as 3 on every (k+2)-th full
 * cycle only (a sample of the timed region: each marker costs ~5 us of idle queue). */
int mg3d_timing_enable(mg3d_ctx *ctx, int on);
int mg3d_timing_reset(mg3d_ctx *ctx);
int mg3d_timing_get(mg3d_ctx *ctx, int level, int stage, int *num_calls, double *seconds);
/* diagnostic: phase stamps (100 MHz clock) of the last single-workgroup coarse cycle launch (csrc/mg3d_tiny.hip):
 * start, d loaded, pre-smoothed, residual, restricted, solve start, solve end, correction ready, prolonged, post-smoothed, stored */
int mg3d_debug_tiny_stamps(long long *out16);
/* per-kernel timers, same mechanism, one event pair around each launch of the kernels below */
enum {
    MG3D_K_SWEEP4 = 0,   /* fused sweep, 4 colour passes */
    MG3D_K_SWEEP2,       /* fused sweep, 2 colour passes */
    MG3D_K_SWEEP2_RES,   /* fused sweep, 2 colour passes + residual */
    MG3D_K_RESIDUAL,     /* residual (+ r store) of the current field */
    MG3D_K_RESTRICT, MG3D_K_PROLONG, MG3D_K_COARSE_SOLVE, MG3D_K_COLOUR_PASS,
    MG3D_K_SWEEP4_NORM,     /* carried cycles (mg3d_vcycles): a cycle's last 2 post-smoothing passes, its residual norm and
                               the next cycle's first pre-smoothing passes in one launch */
    MG3D_K_SWEEP1_RESTRICT, /* carried cycles: the last pre-smoothing pass + residual + restriction */
    MG3D_K_LEG_DOWN,        /* one launch per leg: the pre-smoothing passes + residual + restriction */
    MG3D_K_LEG_UP,          /* one launch per leg: prolongation + the post-smoothing passes (+ half of the norm) */
    MG3D_NUM_KERNELS
};
const char *mg3d_kernel_name(int kernel);
int mg3d_kernel_time_get(mg3d_ctx *ctx, int level, int kernel, int *num_launches, double *seconds);

/* ------------------------------------------------------------ mixed boundary conditions ("electrospray")
 * The problem the reference was written for (mg_3d_bkup.c:12-18, 84-133, 739-778): Dirichlet patches on the two x
 * faces -- a capillary disc and an extractor annulus -- and zero-gradient walls everywhere else, imposed by copying a
 * freshly smoothed interior value onto the wall point behind it.  Carried by the live operators of mg_3d.h
 * (csrc/mg3d_es.hip; parity unpinned: the original neither compiles nor is order-independent).  The context must have
 * been created with grid_length == params->length. */
typedef struct mg3d_es_params {
    double length;                  /* side of the cube (GRID_LENGTH, mg_3d_bkup.c:12) */
    double capillary_radius;        /* Dirichlet disc on x = 0 (:14) */
    double extractor_inner_radius;  /* Dirichlet annulus on x = length (:15-16) */
    double extractor_outer_radius;
    double capillary_voltage;       /* (:17) */
    double extractor_voltage;       /* (:18) */
} mg3d_es_params;
int mg3d_es_default_params(mg3d_es_params *p); /* the reference's #defines */
/* host: the coarsest operator with zero-gradient rows on the walls (A zeroed by the caller, (N^3)^2 doubles) */
void mg3d_es_coarse_matrix(double *A, int N, double h, const mg3d_es_params *p);
int mg3d_es_setup(mg3d_ctx *ctx, const mg3d_es_params *p); /* zero fields, patches on the finest u, coarse LU */
int mg3d_es_smooth(mg3d_ctx *ctx, int level, int post, int iters); /* red-black passes + ghost copies */
int mg3d_es_vcycles(mg3d_ctx *ctx, int count, double *norms);

/* ------------------------------------------------------------ several GPUs (i-slabs)
 * The reference splits every operator over the slowest index i between OpenMP threads (mg_3d.h:658-659);
 * here rank r of nranks (one process per GPU) owns a contiguous range of i-planes of every level large
 * enough, exchanges halo planes with its two neighbours by RCCL send/recv, and the small levels plus the
 * direct solve are replicated after one all-gather of the restricted right-hand side.  See
 * csrc/mg3d_dist.hip.  unique_id: 128 bytes from mg3d_comm_unique_id() on rank 0, distributed by the
 * launcher; NULL selects the loopback transport (all ranks virtual, in this process, on `device`).
 * mg3d_dist_upload/download take the FULL N^3 host array; a rank reads its slab / writes its owned planes. */
typedef struct mg3d_dist mg3d_dist;
int mg3d_comm_unique_id(void *out128);
int mg3d_dist_create(int coarse_pts, int num_levels, int smooth_iters, double grid_length, int rank, int nranks,
                     const void *unique_id, int device, mg3d_dist **out);
int mg3d_dist_destroy(mg3d_dist *d);
/* ranks in the RCCL communicator (ncclCommCount; 0 without one), whether the exchanges run on the communication stream
 * (one communicator; the u halos then travel underneath the launches in between: the default, MG3D_NO_OVERLAP=1 at
 * creation keeps everything on the compute stream), the HIP device in use */
int mg3d_dist_comm_info(const mg3d_dist *d, int *rccl_ranks, int *overlap, int *device);
int mg3d_dist_first_level(const mg3d_dist *d); /* lowest distributed level */
int mg3d_dist_halo(const mg3d_dist *d);        /* halo planes per side */
int mg3d_dist_carried_cycles(const mg3d_dist *d); /* cycles since creation that ended ahead into the next one ("carried cycles") */
int mg3d_dist_legs_cycles(const mg3d_dist *d);    /* cycles since creation whose up-leg on the finest level ran as ONE launch (options legs, legs_min;
                                                   * plan policy bits 4 / 8 of mg3d_dist_plan): their norm is completed by the next cycle's down-leg */
int mg3d_dist_build_coarse(mg3d_dist *d, double h_coarse);
int mg3d_dist_set_keep_residual(mg3d_dist *d, int keep); /* as mg3d_ctx_set_keep_residual */
int mg3d_dist_set_option(mg3d_dist *d, const char *key, int value); /* as mg3d_ctx_set_option, for every local rank */
int mg3d_dist_upload(mg3d_dist *d, int field, int level, const double *host_full);
int mg3d_dist_download(mg3d_dist *d, int field, int level, double *host_full);
int mg3d_dist_vcycles(mg3d_dist *d, int count, double *norms);
int mg3d_dist_sync(mg3d_dist *d);
/* the partition itself (pure host arithmetic, usable without a GPU) */
int mg3d_slab_halo(int smooth_iters);
int mg3d_slab_first_level(int coarse_pts, int num_levels, int nranks, int halo);
int mg3d_slab_owned(int coarse_pts, int num_levels, int nranks, int halo, int level, int rank, int *lo, int *hi);

/* The exchange plan of ONE V-cycle for one rank (pure host arithmetic, usable without a GPU): every transfer the rank
 * takes part in, in issue order.  The library's own transports (RCCL and loopback) execute exactly this list -- they hold
 * no plane arithmetic of their own -- so a property checked on the plans of all ranks (every send has its receive with the
 * same count, offsets inside the slab and on the same global planes, every rank walks the same phases) is a property of
 * what runs on the GPUs.  What each phase stands for in the reference: the implicit barrier at the end of an orphaned
 * `omp for` (mg_3d.h:658-702, 807-842, 961-995, 1007-1145), after which every thread sees its neighbours' planes.
 * A phase = one ncclGroupStart/End (all ranks issue the same phases in the same order); offsets and counts are in planes
 * of `plane_elems` elements: LOCAL plane indices of the rank's slab for distributed levels, global ones for the
 * replicated level's arrays.  stream: 0 compute stream (overlap off); overlap on: every exchange is issued on the
 * communication stream, the only stream that drives the one communicator -- 1 the compute stream joins at once, 2 it joins
 * when it next needs the field (the transfer runs underneath the launches in between).  policy bit 0: the coarse levels are solved on rank 0 only (right-hand side gathered,
 * correction broadcast: MG3D_COARSE_GATHER=1) instead of replicated on every rank behind one all-gather.  Bit 1 (2): a V(2,2) cycle that
 * ends ahead into the next one ("carried cycles": HALO_U_NEXT brings three planes).  Bit 2 (4): a V(2,2) cycle whose up-leg on the
 * finest level is ONE launch (HALO_U_NEXT brings planes 1..5, no NORM phase at the end: the next cycle's one-launch down-leg
 * completes the norm); bit 3 (8): the cycle behind such a one (a NORM phase first, behind its down-leg).  8 alone: the last of a run. */
enum { MG3D_XK_HALO_U_DOWN = 0, /* u_l after pre-smoothing + restriction, for the prolongation on the way up */
       MG3D_XK_HALO_D,          /* d_(l-1) after restriction */
       MG3D_XK_RHS_ALLGATHER,   /* d of the first replicated level: one broadcast per owner */
       MG3D_XK_RHS_GATHER,      /* policy bit 0: the same planes to rank 0 only */
       MG3D_XK_CORR_BCAST,      /* policy bit 0: u of the first replicated level from rank 0 */
       MG3D_XK_HALO_U_UP,       /* u_(l-1), the coarse correction, before the prolongation into level l */
       MG3D_XK_HALO_U_NEXT,     /* finest u after post-smoothing, halo planes 2..H, for the next cycle */
       MG3D_XK_NORM,            /* per-rank sums of squares, all-gathered */
       MG3D_XK_COUNT };
enum { MG3D_XOP_SEND = 0, MG3D_XOP_RECV, MG3D_XOP_BCAST, MG3D_XOP_ALLGATHER };
typedef struct mg3d_xfer {
    int phase;       /* 0, 1, ... in issue order within the cycle */
    int kind;        /* MG3D_XK_* */
    int op;          /* MG3D_XOP_* */
    int peer;        /* send/recv: the other rank; broadcast: the root; all-gather: -1 */
    int field, level;
    int offset;      /* first plane (send: source, recv: destination, broadcast: both) */
    int count;       /* planes */
    long long plane_elems;
    int stream;
} mg3d_xfer;
/* returns the number of entries (out may be NULL to ask for it), or a negative MG3D_ERR_* */
int mg3d_dist_plan(int coarse_pts, int num_levels, int nranks, int smooth_iters, int rank, int overlap, int policy,
                   mg3d_xfer *out, int max_entries);
/* per-phase cost of the slab path, from event pairs on the streams the phases run on (on = 1; off by default):
 * ms[0] whole cycles, ms[1] exchanges the compute stream waits for at once, ms[2] exchanges that run overlapped,
 * ms[3] the replicated (or rank-0) coarse levels incl. the direct solve; kernels on the distributed levels =
 * ms[0] - ms[1] - ms[3].  Sums since the last enable; *cycles = cycles covered. */
int mg3d_dist_timing_enable(mg3d_dist *d, int on);
int mg3d_dist_timing_get(mg3d_dist *d, double ms[4], int *cycles);

/* ------------------------------------- host-pointer forms (reference signatures)
 * Same argument meaning as the reference functions; data is staged to the
 * device, computed there, and copied back.  They exist so that drivers that
 * call the operators directly (test_mg_3d_dirichlet.c:51,60) link unchanged. */
int mg3d_host_smooth(double *v, const double *d, int N, double h, int iters, int post);
int mg3d_host_residual(const double *v, const double *d, int N, double h, double *res, double *norm);
int mg3d_host_restrict(const double *r, int Nf, double *dc, int Nc);
int mg3d_host_prolong(const double *ec, int Nc, double *ef, int Nf);
int mg3d_host_lu_solve(const double *LU, int n, const double *b, double *x);
/* vcycle(u,f,res,h,q,numLevels,smootherIter,N,LU), mg_3d.h:1242: caller-owned
 * host hierarchies (allocGridLevels).  Levels 0..q are copied back after the cycle.
 * stage_calls / stage_seconds (optional, (q+1)*MG3D_NUM_STAGES entries, [level][stage])
 * are incremented by this cycle's per-stage event timings (tInfo of mg_3d.h:1279-1359). */
int mg3d_host_vcycle(double **u, double **f, double **res, double h, int q, int num_levels, int iters, int N,
                     const double *LU, double *norm, int *stage_calls, double *stage_seconds);

/* --------------------------------------------------- host-only helpers (no device)
 * mg3d_bc_func            : BCFunc (mg_3d.h:89-90)
 * mg3d_fill_boundary_host : setupBoundaryConditions (mg_3d.h:1147-1239)
 * mg3d_coarse_matrix      : constructCoarseMatrixA (mg_3d.h:147-273), A zeroed by caller
 * mg3d_lu_factor          : convertToLU_InPlace (gauss_elim.h:9-29)
 * mg3d_lu_solve_host      : NOT provided -- the solve runs on the device only
 * mg3d_l2norm_host        : GetL2NormOfVector (mg_3d.h:783-792)
 * mg3d_smooth_edges_host  : updateEdgeValues (mg_3d.h:304-430; cosmetic, never read by the stencil)
 * mg3d_write_vtk          : writeOutputData (postprocess.h:5-47) */
double mg3d_bc_func(double x, double y, double z);
void mg3d_fill_boundary_host(double *v, int N, double h);
void mg3d_coarse_matrix(double *A, int N, double h);
void mg3d_lu_factor(double *a, int n);
double mg3d_l2norm_host(const double *d, long n);
void mg3d_smooth_edges_host(double *u, int N);
int mg3d_write_vtk(const char *file_name, const double *grid, double h, int N);
/* zero-filled page-locked host memory (hipHostMalloc): arrays the facade moves across PCIe on every solve */
int mg3d_host_alloc(size_t bytes, void **out);
int mg3d_host_free(void *p);

/* ---- single precision / damped Jacobi / F-cycle variant (BASELINE configs[4]) --------------------------
 * PARITY UNPINNED: the reference has no fp32 arithmetic and no Jacobi smoother; its FMG start exists only as
 * mg_dirichlet_analytic.c:771-806 (commented copy mg_3d.h:1364-1404).  Semantics are defined by
 * csrc/mg3d_f32.hip and restated in plain C by the test infrastructure: binary32 storage and grid arithmetic with the
 * reference's association per operator, v' = v + omega*((1/6)(sum6 - h^2 d) - v) out of place, the reference's
 * double LU factors on the coarsest level (rhs widened, solution rounded), norms accumulated in double.
 * Fields and levels are numbered as above (MG3D_U/D/R, 0 = coarsest); host arrays are N^3 floats, k fastest. */
typedef struct mg3d32_ctx mg3d32_ctx;
int mg3d32_create(int coarse_pts, int num_levels, int smooth_iters, double omega, double grid_length,
                  mg3d32_ctx **out);
int mg3d32_destroy(mg3d32_ctx *ctx);
int mg3d32_set_option(mg3d32_ctx *ctx, const char *key, int value); /* "pairs", "fuse", "carry": see mg3d_ctx_set_option */
int mg3d32_level_n(const mg3d32_ctx *ctx, int level);
int mg3d32_upload(mg3d32_ctx *ctx, int field, int level, const float *host);
int mg3d32_download(mg3d32_ctx *ctx, int field, int level, float *host);
int mg3d32_zero(mg3d32_ctx *ctx, int field, int level);
int mg3d32_sync(mg3d32_ctx *ctx);
int mg3d32_fill_boundary(mg3d32_ctx *ctx, int field, int level);   /* BCFunc (mg_3d.h:89) on the six faces */
int mg3d32_smooth(mg3d32_ctx *ctx, int level, int iters);          /* `iters` damped-Jacobi sweeps */
int mg3d32_residual(mg3d32_ctx *ctx, int level, int store, double *norm); /* mg_3d.h:794-842 in binary32 */
int mg3d32_restrict(mg3d32_ctx *ctx, int level);                   /* r(level) -> d(level-1), mg_3d.h:844-998 */
int mg3d32_prolong(mg3d32_ctx *ctx, int level);                    /* u(level) += P u(level-1), mg_3d.h:1000-1145 */
int mg3d32_coarse_solve(mg3d32_ctx *ctx);                          /* gauss_elim.h:31-60 through double */
int mg3d32_vcycles(mg3d32_ctx *ctx, int count, double *norms);     /* mg_3d.h:1242-1362 with the Jacobi smoother */
/* F-cycle start, mg_dirichlet_analytic.c:771-806.  Unlike the reference's vcycle (which zeroes u[q] on entry below the
 * finest level, :698-700, and thereby discards the interpolated guess everywhere but on the finest level -- behaviour the
 * fp64 mg3d_fmg_initialize reproduces), this variant zeroes only the coarser level before descending: the interpolated
 * guess is kept on every level. */
int mg3d32_fmg_initialize(mg3d32_ctx *ctx);
/* kernel timers of the finest level's launches (event pairs in-stream, resolved by mg3d32_vcycles' own synchronisation):
 * mg3d32_kernel_name(k) is NULL past the last kernel */
int mg3d32_timing_enable(mg3d32_ctx *ctx, int on);
const char *mg3d32_kernel_name(int kernel);
int mg3d32_kernel_time_get(mg3d32_ctx *ctx, int kernel, int *num_launches, double *seconds);

/* The same variant on i-slabs of several GPUs (csrc/mg3d_f32_dist.hip; BASELINE configs[4]: 1025^3 on 8 GPUs): the
 * partition and schedule of mg3d_dist_* with H = smooth_iters + 2 halo planes (a Jacobi sweep uses up one plane per
 * sweep).  unique_id as for mg3d_dist_create; NULL = loopback (all ranks virtual in this process).  upload/download
 * take the FULL N^3 float array. */
typedef struct mg3d32_dist mg3d32_dist;
int mg3d32_slab_halo(int smooth_iters);
/* the exchange plan of this variant's V-cycle from distributed level q (the F-cycle start runs one from every level;
 * want_norm = 0 leaves the norm phase out), same entry format and the same executor as mg3d_dist_plan; element = float */
int mg3d32_dist_plan(int coarse_pts, int num_levels, int nranks, int smooth_iters, int rank, int q, int want_norm,
                     mg3d_xfer *out, int max_entries);
int mg3d32_dist_create(int coarse_pts, int num_levels, int smooth_iters, double omega, double grid_length, int rank,
                       int nranks, const void *unique_id, int device, mg3d32_dist **out);
int mg3d32_dist_destroy(mg3d32_dist *d);
int mg3d32_dist_first_level(const mg3d32_dist *d);
int mg3d32_dist_halo(const mg3d32_dist *d);
int mg3d32_dist_comm_info(const mg3d32_dist *d, int *rccl_ranks, int *device);
int mg3d32_dist_upload(mg3d32_dist *d, int field, int level, const float *host_full);
int mg3d32_dist_download(mg3d32_dist *d, int field, int level, float *host_full);
int mg3d32_dist_zero(mg3d32_dist *d, int field, int level);
int mg3d32_dist_fill_boundary(mg3d32_dist *d, int field, int level);
int mg3d32_dist_vcycles(mg3d32_dist *d, int count, double *norms);
int mg3d32_dist_fmg_initialize(mg3d32_dist *d); /* needs d of every level (boundary values), u zero */
int mg3d32_dist_sync(mg3d32_dist *d);

#ifdef __cplusplus
}
#endif
#endif /* MG3D_H */
