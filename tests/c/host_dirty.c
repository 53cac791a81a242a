/* host_dirty.c -- "solve, inspect, change rhs and grid through the raw pointers, solve again" through the
 * drop-in facade (include/mg_3d.h).  The reference hands out raw pointers (mg_3d.h:278-279) and sees every
 * write to them; the facade re-uploads after any Solver* call that gave the arrays back to the caller.
 * Between two consecutive SolverLinSolve calls: SolverSyncHost() BEFORE touching grid[], SolverMarkHostDirty()
 * AFTER writing rhs[] (both orders of "write" and "announce" are exercised).
 * Prints the norms; tests/test_dropin.py replays the same sequence with the oracle. */
#include <stdio.h>
#include <string.h>
#define GRID_LENGTH (1.)
#include "mg_3d.h"
#include "postprocess.h"

int main(void)
{
    char *argv[] = {"host_dirty", "5", "3", "2", NULL};
    SolverInitialize(4, argv);
    double *grid, *rhs, h;
    const int N = SolverGetDetails(&grid, &rhs, &h);
    SolverSetupBoundaryConditions();
    setupBoundaryConditions(grid, N, h);
    for (int c = 0; c < 3; c++)
        printf("A %.17g\n", SolverLinSolve());
    printf("R %.17g\n", SolverGetResidual()); /* hands grid[] back: the caller may write from here on */
    const int mid = (N * N + N + 1) * (N / 2);
    rhs[mid] = 250.0;        /* a point source */
    grid[mid + 1] += 0.125;  /* and a dent in the iterate */
    for (int c = 0; c < 3; c++)
        printf("B %.17g\n", SolverLinSolve());
    SolverResetTimingInfo(); /* another hand-back */
    rhs[mid - N] = -125.0;
    for (int c = 0; c < 2; c++)
        printf("C %.17g\n", SolverLinSolve());
    SolverSyncHost(); /* explicit hand-back BEFORE touching grid[] between two SolverLinSolve calls */
    grid[mid - 1] -= 0.25;
    printf("D %.17g\n", SolverLinSolve());
    rhs[mid + N] = 60.0;   /* write first ... */
    SolverMarkHostDirty(); /* ... announce afterwards: a flag only, the write above must survive it */
    printf("E %.17g\n", SolverLinSolve());
    SolverSyncHost();
    grid[mid + 2] += 0.5;
    rhs[mid + 2] = -30.0;
    SolverMarkHostDirty(); /* after a hand-back both arrays are the caller's: nothing may be lost either */
    printf("F %.17g\n", SolverLinSolve());
    SolverPrintTimingInfo();
    double s = 0.;
    for (int p = 0; p < N * N * N; p++)
        s += grid[p] * (1 + p % 13);
    printf("U %.17g\n", s);
    SolverFinalize();
    return 0;
}
