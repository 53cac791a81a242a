/* Our own driver (not a reference file): exercises the parts of the drop-in surface of include/mg_3d.h that
 * the reference's two drivers do not reach -- the operator functions on host arrays, the current (9-argument)
 * vcycle, the 3-argument allocTimingInfo, SolverGetResidual / SolverSmoothenEdgeValues / SolverResetTimingInfo --
 * and prints values at %.17g for the Python test to compare with the oracle. */
#include <stdio.h>
#include <string.h>

#define GRID_LENGTH (1.)
#include "mg_3d.h"
#include "postprocess.h"

static double lcg_state = 12345.;
static double rnd(void)
{
    lcg_state = fmod(lcg_state * 16807., 2147483647.);
    return 2. * (lcg_state / 2147483647.) - 1.;
}

int main(int argc, char **argv)
{
    /* ---- operators on host arrays, N = 9 -> 5 */
    const int N = 9, Nc = 5;
    const double h = 1. / (N - 1);
    double *v = calloc(N * N * N, sizeof(double)), *f = calloc(N * N * N, sizeof(double));
    double *res = calloc(N * N * N, sizeof(double)), *dc = calloc(Nc * Nc * Nc, sizeof(double));
    for (int p = 0; p < N * N * N; p++) {
        v[p] = rnd();
        f[p] = rnd();
    }
    preSmoother(v, f, N, h, 2);
    postSmoother(v, f, N, h, 1);
    const double nrm = calculateResidual(v, f, N, h, res);
    restrictResidual(res, N, dc, Nc);
    prolongateAndCorrectError(dc, Nc, v, N);
    double sv = 0, sd = 0;
    for (int p = 0; p < N * N * N; p++)
        sv += v[p] * (1 + p % 7);
    for (int p = 0; p < Nc * Nc * Nc; p++)
        sd += dc[p] * (1 + p % 5);
    printf("OPS %.17g %.17g %.17g %.17g\n", nrm, sv, sd, GetL2NormOfVector(f, N * N * N));

    /* ---- current-generation timing + 9-argument vcycle on caller-owned hierarchies */
    const char *names[2] = {"alpha", "beta"};
    TimingInfo *one = NULL;
    allocTimingInfo(&one, (char **)names, 2);
    one->numCalls[1] = 3;
    printTimingInfo(one);
    resetTimingInfo(one);
    deAllocTimingInfo(&one);

    const int c = 3, L = 3, nu = 2, Nf = (c - 1) * 4 + 1;
    double **lu = NULL, **lf = NULL, **lr = NULL;
    allocGridLevels(&lu, L, c);
    allocGridLevels(&lf, L, c);
    allocGridLevels(&lr, L, c);
    const double hf = GRID_LENGTH / (Nf - 1);
    double *M = calloc(27 * 27, sizeof(double));
    constructCoarseMatrixA(M, c, hf * 4);
    convertToLU_InPlace(M, 27);
    setupBoundaryConditions(lu[L - 1], Nf, hf);
    allocTimingInfo(&tInfo, L);
    for (int it = 0; it < 4; it++)
        printf("VC9 %.17g\n", vcycle(lu, lf, lr, hf, L - 1, L, nu, Nf, M));
    printf("TIMED %d %d\n", tInfo[L - 1]->numCalls[0], tInfo[0]->numCalls[3]);
    deAllocTimingInfo(&tInfo, L);
    double su = 0;
    for (int p = 0; p < Nf * Nf * Nf; p++)
        su += lu[L - 1][p] * (1 + p % 11);
    printf("VC9U %.17g\n", su);

    /* ---- Solver facade extras */
    char *av[4] = {argv[0], "5", "3", "2"};
    SolverInitialize(4, av);
    double *grid, *rhs, hh;
    const int n = SolverGetDetails(&grid, &rhs, &hh);
    SolverSetupBoundaryConditions();
    setupBoundaryConditions(grid, n, hh);
    printf("INIT %.17g\n", SolverGetInitialResidual());
    printf("RES0 %.17g\n", SolverGetResidual());
    for (int it = 0; it < 3; it++)
        printf("LIN %.17g\n", SolverLinSolve());
    printf("RES3 %.17g\n", SolverGetResidual()); /* pulls u back, recomputes: must equal the last LIN */
    SolverSmoothenEdgeValues();
    printf("EDGE %.17g %.17g\n", grid[0], grid[n - 1]);
    SolverResetTimingInfo();
    printf("LIN %.17g\n", SolverLinSolve());
    SolverPrintTimingInfo();
    SolverFinalize();
    (void)argc;
    return 0;
}
