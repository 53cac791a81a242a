/*
 * mg3d_host.c -- host-only pieces of libmg3d.so: problem set-up, coarse
 * operator assembly + factorisation (done once per solver, mg_3d.h:282-289),
 * VTK output.  None of this is on the V-cycle hot path; the hot path is HIP
 * (mg3d_kernels.hip).  Compiled with -ffp-contract=off so that results are
 * bit-identical to the reference built by gcc -O2.
 */
#include "mg3d.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

/* BCFunc, mg_3d.h:89-90 */
double mg3d_bc_func(double x, double y, double z) { return x * x - 2 * y * y + z * z; }

/* setupBoundaryConditions, mg_3d.h:1147-1239: Dirichlet value BCFunc(i*h, j*h, k*h)
 * on every point that lies on one of the six faces. */
void mg3d_fill_boundary_host(double *v, int N, double h)
{
    const long NN = (long)N * N;
    for (int i = 0; i < N; i++) {
        const double x = i * h;
        for (int j = 0; j < N; j++) {
            const double y = j * h;
            double *row = v + NN * i + (long)N * j;
            if (i == 0 || i == N - 1 || j == 0 || j == N - 1) {
                for (int k = 0; k < N; k++)
                    row[k] = mg3d_bc_func(x, y, k * h);
            } else {
                row[0] = mg3d_bc_func(x, y, 0 * h);
                row[N - 1] = mg3d_bc_func(x, y, (N - 1) * h);
            }
        }
    }
}

/* constructCoarseMatrixA, mg_3d.h:147-273.  Row `p` of the dense n x n matrix:
 * identity on boundary nodes (:179-185), (1,1,1,1,1,1,-6)/h^2 on interior
 * nodes (:257-268).  A must be zero on entry (calloc, mg_3d.h:283). */
void mg3d_coarse_matrix(double *A, int N, double h)
{
    const long NN = (long)N * N, n = NN * N;
    const double hSq = h * h;
    const double invHsq = 1. / hSq;
    const double off = 1. * invHsq, diag = 6. * invHsq;
    long p = 0;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            for (int k = 0; k < N; k++, p++) {
                double *row = A + p * n;
                const int interior = i > 0 && i < N - 1 && j > 0 && j < N - 1 && k > 0 && k < N - 1;
                if (!interior) {
                    row[p] = 1.;
                    continue;
                }
                row[p - NN] = off;
                row[p + NN] = off;
                row[p - N] = off;
                row[p + N] = off;
                row[p - 1] = off;
                row[p + 1] = off;
                row[p] = -diag;
            }
}

/* The coarsest operator of the mixed-boundary problem (csrc/mg3d_es.hip): constructCoarseMatrixA (identity rows on
 * the boundary, mg_3d.h:179-185) except that a wall point -- a face point with an interior point in front of it that is
 * not part of a Dirichlet patch -- gets the row  x_wall - x_front = b_wall, the zero-gradient condition the smoother
 * imposes by its ghost copy (mg_3d_bkup.c:84-133).  The original pins every boundary point of the coarsest level
 * (mg_3d_bkup.c:490-494), which leaves the cycle with a convergence factor of 0.93-0.96. */
void mg3d_es_coarse_matrix(double *A, int N, double h, const mg3d_es_params *p)
{
    const long NN = (long)N * N, n = NN * N;
    mg3d_coarse_matrix(A, N, h);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            for (int k = 0; k < N; k++) {
                const int fi = i == 0 || i == N - 1, fj = j == 0 || j == N - 1, fk = k == 0 || k == N - 1;
                if (fi + fj + fk != 1)
                    continue; /* interior, or an edge / corner (never read by the stencil) */
                const long q = NN * i + (long)N * j + k;
                long front;
                if (fi) {
                    const double ty = j * h - p->length / 2., tz = k * h - p->length / 2., rr = ty * ty + tz * tz;
                    const int dirichlet = i == 0 ? rr <= p->capillary_radius * p->capillary_radius
                                                 : (rr > p->extractor_inner_radius * p->extractor_inner_radius &&
                                                    rr < p->extractor_outer_radius * p->extractor_outer_radius);
                    if (dirichlet)
                        continue;
                    front = i == 0 ? q + NN : q - NN;
                } else if (fj) {
                    front = j == 0 ? q + N : q - N;
                } else {
                    front = k == 0 ? q + 1 : q - 1;
                }
                A[q * n + front] = -1.;
            }
}

/* convertToLU_InPlace, gauss_elim.h:9-29: Doolittle, unit-lower, no pivoting,
 * row-major in place.  The multiplier z = a[k][i] / a[i][i] is formed as
 * a[k][i] * (1/a[i][i]) exactly as :17,:22 do.  The matrix is banded (half-width N^2)
 * and elimination without pivoting never fills outside the band, so only the
 * rows and columns inside it are visited: O(n * bw^2) instead of the dense O(n^3)
 * (c = 33: 4e10 instead of 1.5e13 operations).  What is skipped leaves every entry
 * as the dense loop would: a row whose multiplier is an exact zero changes nothing
 * (a -= 0*x), a zero of the pivot row likewise; the multiplier itself is stored as
 * the signed zero (+0) * (1/a[i][i]) the dense loop writes (:21-22), inside the band
 * by the loop below, outside it by the last pass. */
void mg3d_lu_factor(double *a, int n)
{
    int bwl = 0, bwu = 0; /* populated half-bandwidths of the input */
    for (int i = 0; i < n; i++) {
        const double *ri = a + (long)n * i;
        int lo = 0, hi = n - 1;
        while (lo < i && ri[lo] == 0.)
            lo++;
        while (hi > i && ri[hi] == 0.)
            hi--;
        if (i - lo > bwl)
            bwl = i - lo;
        if (hi - i > bwu)
            bwu = hi - i;
    }
    double *pinvs = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n - 1; i++) {
        const double *ri = a + (long)n * i;
        const double pinv = 1. / ri[i];
        if (pinvs)
            pinvs[i] = pinv;
        int last = i + bwu < n - 1 ? i + bwu : n - 1; /* last non-zero column of the pivot row */
        while (last > i && ri[last] == 0.)
            last--;
        const int klast = i + bwl < n - 1 ? i + bwl : n - 1;
        for (int k = i + 1; k <= klast; k++) {
            double *rk = a + (long)n * k;
            if (rk[i] == 0.) {
                rk[i] = rk[i] * pinv; /* keeps the sign of zero the reference would store */
                continue;
            }
            const double z = rk[i] * pinv;
            rk[i] = z;
            for (int j = i + 1; j <= last; j++)
                rk[j] -= z * ri[j];
        }
    }
    /* below the band: the entry is still the input's zero when the dense loop reaches it; it stores zero * pinv */
    if (pinvs) {
        for (int k = bwl + 1; k < n; k++) {
            double *rk = a + (long)n * k;
            for (int i = 0; i < k - bwl; i++)
                rk[i] = rk[i] * pinvs[i];
        }
        free(pinvs);
    }
}

/* GetL2NormOfVector, mg_3d.h:783-792 (sequential sum, all n entries) */
double mg3d_l2norm_host(const double *d, long n)
{
    double s = 0.;
    for (long i = 0; i < n; i++)
        s += d[i] * d[i];
    return sqrt(s);
}

/* updateEdgeValues, mg_3d.h:304-430: cosmetic averaging of the 12 edges
 * (0.5 * the two face neighbours) and then the 8 corners ((1./3) * the three
 * edge neighbours).  The stencil never reads these points. */
void mg3d_smooth_edges_host(double *u, int N)
{
    const long sI = (long)N * N, sJ = N, sK = 1;
    const long stride[3] = {sI, sJ, sK};
    /* An edge runs along axis `a`; the two other axes (b, c) sit at 0 or N-1.
     * Neighbour order inside the sum follows the reference: for edges along j
     * the k-neighbour comes first (:316), along k the j-neighbour (:330), along
     * i the j-neighbour (:372).  a+b is commutative, so only membership matters. */
    for (int a = 0; a < 3; a++) {
        const int b = (a + 1) % 3, c = (a + 2) % 3;
        for (int eb = 0; eb < 2; eb++)
            for (int ec = 0; ec < 2; ec++) {
                const long base = (eb ? (N - 1) * stride[b] : 0) + (ec ? (N - 1) * stride[c] : 0);
                const long nb = eb ? -stride[b] : stride[b], nc = ec ? -stride[c] : stride[c];
                for (int t = 1; t < N - 1; t++) {
                    const long p = base + t * stride[a];
                    u[p] = 0.5 * (u[p + nb] + u[p + nc]);
                }
            }
    }
    /* corners (:394-429): (1./3) * (k-neighbour + j-neighbour + i-neighbour), in that order */
    for (int ci = 0; ci < 2; ci++)
        for (int cj = 0; cj < 2; cj++)
            for (int ck = 0; ck < 2; ck++) {
                const long p = (ci ? (N - 1) * sI : 0) + (cj ? (N - 1) * sJ : 0) + (ck ? (N - 1) * sK : 0);
                const long di = ci ? -sI : sI, dj = cj ? -sJ : sJ, dk = ck ? -sK : sK;
                u[p] = (1. / 3) * (u[p + dk] + u[p + dj] + u[p + di]);
            }
}

/* writeOutputData, postprocess.h:5-47: ASCII legacy-VTK structured grid,
 * coordinates then one scalar per point, "%10.8e". */
int mg3d_write_vtk(const char *file_name, const double *grid, double h, int N)
{
    FILE *f = fopen(file_name, "w");
    if (!f)
        return MG3D_ERR_ARG;
    const long total = (long)N * N * N;
    fprintf(f,
            "# vtk DataFile Version 2.0\n"
            "Potential data\n"
            "ASCII\n"
            "DATASET STRUCTURED_GRID\n"
            "DIMENSIONS %d %d %d\n"
            "POINTS %ld float\n",
            N, N, N, total);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            for (int k = 0; k < N; k++)
                fprintf(f, "%10.8e %10.8e %10.8e\n", h * i, h * j, h * k);
    fprintf(f,
            "\n"
            "POINT_DATA %ld\n"
            "SCALARS data float 1\n"
            "LOOKUP_TABLE default\n",
            total);
    for (long p = 0; p < total; p++)
        fprintf(f, "%10.8e\n", grid[p]);
    fclose(f);
    return MG3D_OK;
}
