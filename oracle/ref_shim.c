/*
 * ref_shim.c -- builds the UNMODIFIED reference (mg_3d.h, gauss_elim.h,
 * timing_info.h, included from /root/reference via -I, never copied) into
 * oracle/_ref/libmg3d_ref.so.  TEST INFRASTRUCTURE ONLY.
 *
 * Every reference operator is a non-static definition inside the header, so
 * including it once exports preSmoother/postSmoother/calculateResidual/
 * restrictResidual/prolongateAndCorrectError/solveWithLU/convertToLU_InPlace/
 * constructCoarseMatrixA/setupBoundaryConditions/Solver* from this library.
 * The one function defined here, ref_run_problem(), is our own driver: it
 * calls the reference's Solver* API in the order test_mg_3d.c:11-68 does
 * (same OpenMP team structure), minus printf/VTK, for a fixed cycle count.
 */
#define GRID_LENGTH (1.)
#include "mg_3d.h"

double ref_run_problem(int c, int L, int iters, int cycles, double *norms, double *u_out, double *init_norm)
{
    char a0[] = "ref", a1[16], a2[16], a3[16];
    char *argv[4] = {a0, a1, a2, a3};
    snprintf(a1, sizeof a1, "%d", c);
    snprintf(a2, sizeof a2, "%d", L);
    snprintf(a3, sizeof a3, "%d", iters);
    SolverInitialize(4, argv);
    double *grid = NULL, *rhs = NULL, h;
    const int N = SolverGetDetails(&grid, &rhs, &h);
    SolverSetupBoundaryConditions();
    const double init = SolverGetInitialResidual();
    if (init_norm)
        *init_norm = init;
    setupBoundaryConditions(grid, N, h);
    const int nt = omp_get_max_threads();
    double *part = calloc((size_t)nt, sizeof(double));
    const double t0 = omp_get_wtime();
#pragma omp parallel
    {
        const int tid = omp_get_thread_num();
        for (int it = 0; it < cycles; it++) {
            part[tid] = SolverLinSolve();
#pragma omp barrier
#pragma omp single
            {
                double s = 0;
                for (int t = 0; t < nt; t++)
                    s += part[t] * part[t];
                if (norms)
                    norms[it] = sqrt(s);
            }
        }
    }
    const double t1 = omp_get_wtime();
    if (u_out)
        memcpy(u_out, grid, sizeof(double) * (size_t)N * N * N);
    free(part);
    SolverFinalize();
    return t1 - t0;
}

int ref_max_threads(void) { return omp_get_max_threads(); }
