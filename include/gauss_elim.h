/*
 * gauss_elim.h -- drop-in for the reference's gauss_elim.h (same names, same argument
 * meaning), forwarding into libmg3d.so (include/mg3d.h).
 *   convertToLU_InPlace  gauss_elim.h:9-29   -> mg3d_lu_factor (host, once per solver)
 *   solveWithLU          gauss_elim.h:31-60  -> mg3d_host_lu_solve (HIP, banded, bit-identical)
 * gaussianElimination (gauss_elim.h:65-97) is used by the 1D demos only and is not provided.
 */
#ifndef GAUSS_ELIM_H
#define GAUSS_ELIM_H

#include <stdio.h>
#include <stdlib.h>

#include "mg3d.h"

static inline void mg3d_die_(int rc, const char *who)
{
    if (rc != MG3D_OK) {
        fprintf(stderr, "%s: libmg3d error %d: %s\n", who, rc, mg3d_last_error());
        exit(1);
    }
}

static inline void convertToLU_InPlace(double *a, int n) { mg3d_lu_factor(a, n); }

static inline void solveWithLU(const double *LU, const int n, const double *b, double *x)
{
    mg3d_die_(mg3d_host_lu_solve(LU, n, b, x), "solveWithLU");
}

#endif
