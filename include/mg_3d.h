/*
 * mg_3d.h -- drop-in replacement for the reference's mg_3d.h.
 *
 * Put this directory on the include path INSTEAD of the reference tree and link
 * against libmg3d.so:  test_mg_3d.c and test_mg_3d_dirichlet.c compile unchanged
 *   gcc -fopenmp -I<repo>/include test_mg_3d.c -L<repo>/multigrid_parallel_amd/lib -lmg3d -lm
 * Every function keeps the reference's name, argument meaning and error behaviour and
 * forwards through the C ABI of include/mg3d.h into hand-written HIP kernels (gfx950).
 * Nothing is computed on the CPU here except what the reference also does once at set-up
 * (boundary fill, coarse-matrix assembly and factorisation) and the 12-edge cosmetics.
 *
 * As in the reference, the including translation unit must `#define GRID_LENGTH` first
 * (test_mg_3d.c:4) and the header provides stdio/stdlib/math/string/assert/stdbool/omp.
 *
 * Host-pointer contract of the Solver* facade (reference: mg_3d.h:275-293 hands out raw
 * pointers to the finest u and d): the library keeps the hierarchy on the GPU while
 * SolverLinSolve is iterated.  Host arrays are pushed to the device on the first
 * SolverLinSolve after the caller could have written them, and pulled back by every other
 * Solver* call that follows a solve (SolverPrintTimingInfo, SolverGetResidual,
 * SolverSmoothenEdgeValues, SolverFinalize) -- in test_mg_3d.c that is line 74, before the
 * first host read at line 90.  Set MG3D_SYNC_EVERY_CYCLE=1 to copy back after every cycle.
 *
 * Both generations of the reference's own call sites are accepted (by argument count):
 *   vcycle(u,f,res,h,q,numLevels,smootherIter,N,LU)   mg_3d.h:1242            (current, 9 args)
 *   vcycle(u,f,res,  q,numLevels,smootherIter,N,LU)   test_mg_3d_dirichlet.c:60 (legacy, 8 args,
 *                                                      h = GRID_LENGTH/(N-1), mg_dirichlet_analytic.c:692-694)
 */
#ifndef MG_3D_H
#define MG_3D_H

#include <stdio.h>
#include <stdlib.h>
#include <assert.h>
#include <math.h>
#include <stdbool.h>
#include <limits.h>
#include <string.h>

#include <omp.h>

#include "gauss_elim.h"
#include "timing_info.h"
#include "mg3d.h"

/* ---- the reference's global solver state (mg_3d.h:19-28) ----------------------------
 * The reference defines these with EXTERNAL linkage inside its header, which only works for a program that is a
 * single translation unit (a second TU including mg_3d.h would fail to link).  Here they are `static`: every
 * including TU gets its own solver instance.  The reference's drivers are single-TU programs, so nothing observable
 * changes; a multi-TU program must keep all Solver* calls in one TU. */
static TimingInfo **tInfo = NULL;
static int coarseGridNum;
static int finestOneSideNum;
static int numLevels;
static int gsIterNum;
static double **u, **d, **r; /* host mirrors of the three hierarchies */
static double *A;            /* coarsest-level matrix, LU-factored in place */
static double spacing;

/* ---- state of this implementation ---------------------------------------------------- */
static mg3d_ctx *mg3d_solver_ctx_ = NULL;
static int mg3d_host_newer_ = 1;   /* host finest u/d written since the last upload */
static int mg3d_rhs_newer_ = 0;    /* host finest d alone written since the last upload (SolverMarkHostDirty) */
static int mg3d_device_newer_ = 0; /* device u newer than the host mirror */
static int mg3d_fmg_done_ = 0;     /* MG3D_USE_FMG=1: the F-cycle start has run */
static double mg3d_team_norm_ = 0.;
static int mg3d_pinned_top_ = 0;   /* finest u and d come from mg3d_host_alloc */

/* allocGridLevels, mg_3d.h:30-48: level i has ((N-1)*2^i+1)^3 zeroed doubles */
static inline void allocGridLevels(double ***lv, const int nLevels, const int N)
{
    *lv = (double **)malloc(sizeof(double *) * (size_t)nLevels);
    assert(*lv);
    for (int i = 0; i < nLevels; i++) {
        const size_t n = (size_t)(N - 1) * ((size_t)1 << i) + 1;
        (*lv)[i] = (double *)calloc(n * n * n, sizeof(double));
        assert((*lv)[i]);
    }
}

/* deAllocGridLevels, mg_3d.h:295-302 */
static inline void deAllocGridLevels(double ***lv, const int nLevels)
{
    for (int i = 0; i < nLevels; i++)
        free((*lv)[i]);
    free(*lv);
}

/* printGrid3D / printMatrix, mg_3d.h:51-87 (debug dumps) */
static inline void printGrid3D(const double *grid, const int N)
{
    for (int i = 0; i < N; i++) {
        printf("LEVEL %d\n", i);
        for (int k = N - 1; k >= 0; k--) {
            for (int j = 0; j < N; j++)
                printf("%10.5g ", grid[(size_t)N * N * i + (size_t)N * j + k]);
            printf("\n");
        }
        printf("\n");
    }
}

static inline void printMatrix(const double *mat, const int dim)
{
    for (int i = 0; i < dim; i++) {
        for (int j = 0; j < dim; j++)
            printf("%10.5lf ", mat[(size_t)dim * i + j]);
        printf("\n");
    }
}

static inline double BCFunc(double x, double y, double z) { return mg3d_bc_func(x, y, z); } /* mg_3d.h:89-90 */
static inline bool isPowerOfTwo(int x) { return (x & (x - 1)) == 0; }                        /* mg_3d.h:104-105 */

/* constructCoarseMatrixA, mg_3d.h:147-273 */
static inline void constructCoarseMatrixA(double *M, int N, const double h)
{
    assert((long long)N * N * N * ((long long)N * N * N) < INT_MAX); /* mg_3d.h:163 */
    mg3d_coarse_matrix(M, N, h);
}

/* setupBoundaryConditions, mg_3d.h:1147-1239 */
static inline void setupBoundaryConditions(double *v, int levelN, double h) { mg3d_fill_boundary_host(v, levelN, h); }

/* updateEdgeValues, mg_3d.h:304-430 */
static inline void updateEdgeValues(double *v, const int N) { mg3d_smooth_edges_host(v, N); }

/* GetL2NormOfVector, mg_3d.h:783-792 */
static inline double GetL2NormOfVector(const double *v, const int n) { return mg3d_l2norm_host(v, n); }

/* ---- grid operators on host arrays (mg_3d.h:640-1145), executed on the GPU -----------
 * In the reference these are orphaned `omp for` constructs: inside a parallel region every
 * thread of the team calls them together (mg_3d.h:1282-1341, test_rb_gs_3d.c:70-71) and the
 * implicit barrier of the last loop ends the call.  Here the master thread runs the GPU work
 * between two team barriers; a serial call (test_mg_3d_dirichlet.c) has a team of one. */
#define MG3D_TEAM_CALL_(expr, what) \
    do {                            \
        _Pragma("omp barrier")      \
        _Pragma("omp master")       \
        mg3d_die_((expr), (what));  \
        _Pragma("omp barrier")      \
    } while (0)

static inline void preSmoother(double *v, const double *rhs, const int N, const double h, const int smootherIter)
{
    MG3D_TEAM_CALL_(mg3d_host_smooth(v, rhs, N, h, smootherIter, 0), "preSmoother");
}

static inline void postSmoother(double *v, const double *rhs, const int N, const double h, const int smootherIter)
{
    MG3D_TEAM_CALL_(mg3d_host_smooth(v, rhs, N, h, smootherIter, 1), "postSmoother");
}

/* Called by every thread of an OpenMP team in the reference (orphaned `omp for`,
 * mg_3d.h:807) with the caller summing the squares of the returned partial norms
 * (test_mg_3d.c:53-59): the master thread does the work and returns the norm, the other
 * threads return 0.  A serial call (test_mg_3d_dirichlet.c:51) returns the full norm. */
static inline double calculateResidual(const double *v, const double *rhs, const int N, const double h, double *res)
{
    double nrm = 0.;
#pragma omp barrier
#pragma omp master
    mg3d_die_(mg3d_host_residual(v, rhs, N, h, res, &nrm), "calculateResidual");
#pragma omp barrier
    return nrm;
}

static inline void restrictResidual(const double *res, const int Nf, double *dc, const int Nc)
{
    MG3D_TEAM_CALL_(mg3d_host_restrict(res, Nf, dc, Nc), "restrictResidual");
}

static inline void prolongateAndCorrectError(const double *ec, const int Nc, double *ef, const int Nf)
{
    MG3D_TEAM_CALL_(mg3d_host_prolong(ec, Nc, ef, Nf), "prolongateAndCorrectError");
}

/* vcycle on caller-owned host hierarchies, mg_3d.h:1242-1362.  Team-safe like
 * calculateResidual.  Per-stage times are added to the global tInfo when it has been
 * allocated with at least q+1 levels (mg_3d.h:1279-1359). */
static inline double mg3d_vcycle9_(double **lu_, double **lf_, double **lres_, double h, int q, const int nLevels,
                                   const int smootherIter, int N, double *LU)
{
    double nrm = 0.;
#pragma omp barrier
#pragma omp master
    {
        int calls[24 * MG3D_NUM_STAGES];
        double secs[24 * MG3D_NUM_STAGES];
        memset(calls, 0, sizeof calls);
        memset(secs, 0, sizeof secs);
        const int timed = tInfo != NULL && q < 24;
        mg3d_die_(mg3d_host_vcycle(lu_, lf_, lres_, h, q, nLevels, smootherIter, N, LU, &nrm, timed ? calls : NULL,
                                   timed ? secs : NULL),
                  "vcycle");
        if (timed)
            for (int l = 0; l <= q; l++)
                for (int s = 0; s < MG3D_NUM_STAGES && s < MG3D_TI_NSTAGES_(tInfo, l); s++) {
                    MG3D_TI_CALLS_(tInfo, l, s) += calls[l * MG3D_NUM_STAGES + s];
                    MG3D_TI_TIME_(tInfo, l, s) += secs[l * MG3D_NUM_STAGES + s];
                }
    }
#pragma omp barrier
    return nrm;
}

static inline double mg3d_vcycle8_(double **lu_, double **lf_, double **lres_, int q, const int nLevels,
                                   const int smootherIter, int N, double *LU)
{
    return mg3d_vcycle9_(lu_, lf_, lres_, (double)(GRID_LENGTH) / (N - 1), q, nLevels, smootherIter, N, LU);
}

#define MG3D_PICK9_(a, b, c, d_, e, f, g, h_, i, name, ...) name
#define vcycle(...) MG3D_PICK9_(__VA_ARGS__, mg3d_vcycle9_, mg3d_vcycle8_, )(__VA_ARGS__)

/* ---- Solver facade (mg_3d.h:107-144, 275-293, 1412-1467) ----------------------------- */
/* Every Solver* call except SolverLinSolve goes through here: the device result (if newer) comes back
 * into the arrays SolverGetDetails handed out, and -- because the caller owns those arrays and may write
 * grid[] / rhs[] through the raw pointers at any time after such a call (the reference sees such writes,
 * mg_3d.h:278-279) -- the next SolverLinSolve uploads them again.  Writes between two consecutive
 * SolverLinSolve calls with no other Solver* call in between are not visible to a call counter; two calls
 * that are not in the reference cover them (or run with MG3D_SYNC_EVERY_CYCLE=1: download + upload around
 * every cycle):
 *   SolverSyncHost()       call BEFORE reading or writing grid[]: brings the device iterate into grid[] and hands
 *                          both arrays back to the caller (everything is uploaded again by the next SolverLinSolve);
 *   SolverMarkHostDirty()  call AFTER writing rhs[]: a flag only, it never copies anything back, so writes made
 *                          before it are never overwritten.  rhs[] is uploaded by the next SolverLinSolve; grid[] too
 *                          if the host copy is current (no solve since the last hand-back).  Writes to grid[] made
 *                          while the device iterate is newer than grid[] cannot be merged: SolverSyncHost() first. */
static inline void mg3d_pull_(void)
{
    if (mg3d_solver_ctx_ && mg3d_device_newer_) {
        mg3d_die_(mg3d_download(mg3d_solver_ctx_, MG3D_U, numLevels - 1, u[numLevels - 1]), "libmg3d download");
        mg3d_device_newer_ = 0;
    }
    mg3d_host_newer_ = 1;
}

static inline void SolverSyncHost(void) { mg3d_pull_(); }

static inline void SolverMarkHostDirty(void)
{
    if (mg3d_device_newer_)
        mg3d_rhs_newer_ = 1; /* the device iterate stays the authoritative u */
    else
        mg3d_host_newer_ = 1;
}

static inline void SolverInitialize(int argc, char **argv)
{
    if (argc != 4) {
        printf("Usage: %s <coarse grid points on one side> <number of levels> <gauss seidel iterations>\n", argv[0]);
        exit(1);
    }
    coarseGridNum = atoi(argv[1]);
    numLevels = atoi(argv[2]);
    gsIterNum = atoi(argv[3]);
    assert(isPowerOfTwo(coarseGridNum - 1)); /* mg_3d.h:123 */

    const int multFactor = 1 << (numLevels - 1);
    finestOneSideNum = ((coarseGridNum - 1) * multFactor) + 1;

    u = NULL;
    d = NULL;
    r = NULL;
    allocGridLevels(&u, numLevels, coarseGridNum);
    allocGridLevels(&d, numLevels, coarseGridNum);
    allocGridLevels(&r, numLevels, coarseGridNum);
    /* the finest u and d cross PCIe around every solve: page-locked memory from the library for those two */
    {
        const size_t nf = (size_t)finestOneSideNum * finestOneSideNum * finestOneSideNum * sizeof(double);
        void *pu = NULL, *pd = NULL;
        if (mg3d_host_alloc(nf, &pu) == 0 && mg3d_host_alloc(nf, &pd) == 0) {
            free(u[numLevels - 1]);
            free(d[numLevels - 1]);
            u[numLevels - 1] = (double *)pu;
            d[numLevels - 1] = (double *)pd;
            mg3d_pinned_top_ = 1;
        } else if (pu) {
            (void)mg3d_host_free(pu);
        }
    }
    allocTimingInfo(&tInfo, numLevels);
    spacing = (double)(GRID_LENGTH) / (finestOneSideNum - 1);

    mg3d_die_(mg3d_ctx_create(coarseGridNum, numLevels, gsIterNum, (double)(GRID_LENGTH), &mg3d_solver_ctx_),
              "SolverInitialize");
    mg3d_die_(mg3d_timing_enable(mg3d_solver_ctx_, 1), "SolverInitialize");
    mg3d_host_newer_ = 1;
    mg3d_rhs_newer_ = 0;
    mg3d_device_newer_ = 0;
    mg3d_fmg_done_ = 0;
}

static inline int SolverGetDetails(double **grid, double **rhs, double *h)
{
    (*grid) = u[numLevels - 1];
    (*rhs) = d[numLevels - 1];
    const int matDim = coarseGridNum * coarseGridNum * coarseGridNum;
    A = (double *)calloc((size_t)matDim * matDim, sizeof(double));
    const double coarseSpacing = (spacing * (1 << (numLevels - 1))); /* mg_3d.h:287 */
    constructCoarseMatrixA(A, coarseGridNum, coarseSpacing);
    convertToLU_InPlace(A, matDim);
    mg3d_die_(mg3d_ctx_set_lu(mg3d_solver_ctx_, A), "SolverGetDetails");
    *h = spacing;
    mg3d_host_newer_ = 1;
    return finestOneSideNum;
}

static inline void SolverSetupBoundaryConditions(void)
{
    mg3d_pull_();
    setupBoundaryConditions(d[numLevels - 1], finestOneSideNum, spacing);
    mg3d_host_newer_ = 1;
}

/* SolverFMGInitialize: live in mg_dirichlet_analytic.c:771-806, commented out in mg_3d.h:1364-1404 */
static inline void SolverFMGInitialize(void)
{
    const int fin = numLevels - 1;
    if (mg3d_host_newer_) {
        for (int l = 0; l < numLevels; l++) { /* FMG reads d on every level and accumulates into every u */
            mg3d_die_(mg3d_upload(mg3d_solver_ctx_, MG3D_U, l, u[l]), "SolverFMGInitialize");
            mg3d_die_(mg3d_upload(mg3d_solver_ctx_, MG3D_D, l, d[l]), "SolverFMGInitialize");
        }
        mg3d_host_newer_ = 0;
        mg3d_rhs_newer_ = 0;
    }
    if (mg3d_rhs_newer_) {
        mg3d_die_(mg3d_upload(mg3d_solver_ctx_, MG3D_D, fin, d[fin]), "SolverFMGInitialize");
        mg3d_rhs_newer_ = 0;
    }
    mg3d_die_(mg3d_fmg_initialize(mg3d_solver_ctx_), "SolverFMGInitialize");
    mg3d_device_newer_ = 1;
}

/* One V-cycle.  Entered by every thread of the caller's team (test_mg_3d.c:37-45); the
 * squares of the return values are summed by the caller, so the master returns the norm
 * and everybody else 0.  The two barriers keep the team together around the GPU work. */
static inline double SolverLinSolve(void)
{
    double ret = 0.;
#pragma omp barrier
#pragma omp master
    {
        const int fin = numLevels - 1;
        /* MG3D_USE_FMG=1: what the fifth argument `useFMG` of mg_dirichlet_analytic.c:70-80 selects -- the F-cycle
         * start of :984-988 once, in front of the first cycle (test_mg_3d.c has no such argument and stays unchanged).  It uploads every
         * level of the host hierarchies itself */
        if (!mg3d_fmg_done_) {
            const char *fmg = getenv("MG3D_USE_FMG");
            mg3d_fmg_done_ = 1;
            if (fmg && fmg[0] == '1') {
                printf("Doing FMG Initialization...."); /* mg_dirichlet_analytic.c:986-988 */
                SolverFMGInitialize();
                printf("done\n");
            }
        }
        if (mg3d_host_newer_) {
            mg3d_die_(mg3d_upload(mg3d_solver_ctx_, MG3D_U, fin, u[fin]), "SolverLinSolve");
            mg3d_die_(mg3d_upload(mg3d_solver_ctx_, MG3D_D, fin, d[fin]), "SolverLinSolve");
            mg3d_host_newer_ = 0;
            mg3d_rhs_newer_ = 0;
        } else if (mg3d_rhs_newer_) {
            mg3d_die_(mg3d_upload(mg3d_solver_ctx_, MG3D_D, fin, d[fin]), "SolverLinSolve");
            mg3d_rhs_newer_ = 0;
        }
        mg3d_die_(mg3d_vcycle(mg3d_solver_ctx_, fin, &mg3d_team_norm_), "SolverLinSolve");
        mg3d_device_newer_ = 1;
        const char *eager = getenv("MG3D_SYNC_EVERY_CYCLE");
        if (eager && eager[0] == '1')
            mg3d_pull_();
        ret = mg3d_team_norm_;
    }
#pragma omp barrier
    return ret;
}

static inline void SolverSmoothenEdgeValues(void)
{
    mg3d_pull_();
    updateEdgeValues(u[numLevels - 1], finestOneSideNum);
    mg3d_host_newer_ = 1;
}

static inline double SolverGetResidual(void)
{
    mg3d_pull_();
    return calculateResidual(u[numLevels - 1], d[numLevels - 1], finestOneSideNum, spacing, NULL);
}

static inline double SolverGetInitialResidual(void)
{
    mg3d_pull_();
    return GetL2NormOfVector(d[numLevels - 1], finestOneSideNum * finestOneSideNum * finestOneSideNum);
}

static inline void SolverResetTimingInfo(void)
{
    mg3d_pull_();
    mg3d_die_(mg3d_timing_reset(mg3d_solver_ctx_), "SolverResetTimingInfo");
    resetTimingInfo(tInfo, numLevels);
}

static inline void SolverPrintTimingInfo(void)
{
    mg3d_pull_(); /* the sync point between the solve loop and the caller's reads of `grid` */
    for (int l = 0; l < numLevels; l++) {
        for (int s = 0; s < MG3D_NUM_STAGES; s++)
            mg3d_die_(mg3d_timing_get(mg3d_solver_ctx_, l, s, &MG3D_TI_CALLS_(tInfo, l, s), &MG3D_TI_TIME_(tInfo, l, s)),
                      "SolverPrintTimingInfo");
        /* as the reference: "Recurse, Direct Solve" of level l is the time spent below l (mg_3d.h:1318-1325) */
    }
    for (int l = 1; l < numLevels; l++) {
        double below = 0.;
        for (int s = 0; s < MG3D_NUM_STAGES; s++)
            below += MG3D_TI_TIME_(tInfo, l - 1, s);
        MG3D_TI_TIME_(tInfo, l, MG3D_ST_RECURSE) = below;
        MG3D_TI_CALLS_(tInfo, l, MG3D_ST_RECURSE) = MG3D_TI_CALLS_(tInfo, l, MG3D_ST_SMOOTH1);
    }
    printTimingInfo(tInfo, numLevels);
}

static inline void SolverFinalize(void)
{
    mg3d_pull_();
    mg3d_die_(mg3d_ctx_destroy(mg3d_solver_ctx_), "SolverFinalize");
    mg3d_solver_ctx_ = NULL;
    deAllocTimingInfo(&tInfo, numLevels);
    free(A);
    if (mg3d_pinned_top_) { /* hand the two page-locked arrays back first; the rest is calloc'ed */
        (void)mg3d_host_free(u[numLevels - 1]);
        (void)mg3d_host_free(d[numLevels - 1]);
        u[numLevels - 1] = NULL;
        d[numLevels - 1] = NULL;
        mg3d_pinned_top_ = 0;
    }
    deAllocGridLevels(&u, numLevels);
    deAllocGridLevels(&d, numLevels);
    deAllocGridLevels(&r, numLevels);
}

#endif /* MG_3D_H */
