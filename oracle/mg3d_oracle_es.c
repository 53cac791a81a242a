/*
 * mg3d_oracle_es.c -- CPU statement of the mixed-boundary ("electrospray") problem of the reference's original
 * program, mg_3d_bkup.c, carried by the live operators of mg_3d.h.  TEST INFRASTRUCTURE ONLY (see mg3d_oracle.h).
 *
 * PARITY UNPINNED.  mg_3d_bkup.c does not compile against the current headers and its smoother is the
 * serial lexicographic Gauss-Seidel that the survey puts out of scope (its result depends on the traversal order);
 * no golden vector can be produced from the unmodified reference.  What is taken from it is the PROBLEM:
 *   - geometry and potentials, mg_3d_bkup.c:12-18: cube of side 3e-4, capillary disc (radius 1.326e-5, 0 V) on the
 *     face x = 0, extractor annulus (1e-4 .. 1.4e-4, -1350 V) on the face x = L, both centred in (y, z);
 *   - where the Dirichlet patches are, :739-778 (rr <= Rc^2 on x = 0; Ri^2 < rr < Ro^2 on x = L, rr from (j h - L/2,
 *     k h - L/2) with the level's own h), everything else on the six faces is a zero-gradient wall;
 *   - how the walls are imposed, :84-133: when an interior point next to a wall has been updated, its new value is
 *     copied onto the wall point behind it ("ghost copy"), on every level.
 * Everything else is the reference's live path: red-black passes in the order of mg_3d.h:640-781, the update of
 * :438-443, residual :794-842, restriction :844-998, prolongation :1000-1145, cycle :1242-1362, dense LU with
 * identity boundary rows.  With red-black ordering the ghost copy is order-independent: a wall point is written
 * only by the one interior point in front of it and read only by that point.
 * Two things the original leaves open are settled here (and in csrc/mg3d_es.hip, which this file restates):
 *   - prolongation adds the interpolated correction to every fine point, boundary included (:1000-1145); Dirichlet
 *     patch points are put back to their potential (finest level) or to zero (error equation) right after it;
 *   - the coarsest operator carries the zero-gradient condition on the walls as rows x_wall - x_front = b (the
 *     original pins every boundary point, mg_3d_bkup.c:490-494, which stalls the cycle at a factor of 0.93-0.96); the
 *     ghost copy is applied once behind the direct solve.
 */
#include <stdlib.h>
#include <string.h>

#include "mg3d_oracle.h"

typedef struct {
    double length, capillary_radius, extractor_inner, extractor_outer, capillary_voltage, extractor_voltage;
} orc_es_params;

#define IDX(i, j, k) (((size_t)(i) * N + (j)) * N + (k))

static double rr_of(const orc_es_params *p, double h, int j, int k)
{
    const double ty = j * h - p->length / 2., tz = k * h - p->length / 2.; /* mg_3d_bkup.c:98-100, 750-754 */
    return ty * ty + tz * tz;
}
int orc_es_dirichlet_x0(const orc_es_params *p, double h, int j, int k)
{
    return rr_of(p, h, j, k) <= p->capillary_radius * p->capillary_radius; /* :755 */
}
int orc_es_dirichlet_xl(const orc_es_params *p, double h, int j, int k)
{
    const double rr = rr_of(p, h, j, k); /* :771-773 */
    return rr > p->extractor_inner * p->extractor_inner && rr < p->extractor_outer * p->extractor_outer;
}

/* the Dirichlet patches of v: scale * potential (scale 1 on the finest level, 0 for the error equation) */
void orc_es_fill(double *v, int N, double h, const orc_es_params *p, double scale)
{
    for (int j = 0; j < N; j++)
        for (int k = 0; k < N; k++) {
            if (orc_es_dirichlet_x0(p, h, j, k))
                v[IDX(0, j, k)] = scale * p->capillary_voltage;
            if (orc_es_dirichlet_xl(p, h, j, k))
                v[IDX(N - 1, j, k)] = scale * p->extractor_voltage;
        }
}

/* ghost copy behind the interior point (i, j, k) whose value is val (mg_3d_bkup.c:84-133) */
static void ghost(double *v, int N, double h, const orc_es_params *p, int i, int j, int k, double val)
{
    if (i == 1 && !orc_es_dirichlet_x0(p, h, j, k))
        v[IDX(0, j, k)] = val;
    if (i == N - 2 && !orc_es_dirichlet_xl(p, h, j, k))
        v[IDX(N - 1, j, k)] = val;
    if (j == 1)
        v[IDX(i, 0, k)] = val;
    if (j == N - 2)
        v[IDX(i, N - 1, k)] = val;
    if (k == 1)
        v[IDX(i, j, 0)] = val;
    if (k == N - 2)
        v[IDX(i, j, N - 1)] = val;
}

void orc_es_smooth_color(double *v, const double *d, int N, double h, int color, const orc_es_params *p)
{
    const double hSq = h * h, sixth = 1. / 6;
    const size_t NN = (size_t)N * N;
    for (int i = 1; i < N - 1; i++)
        for (int j = 1; j < N - 1; j++)
            for (int k = 1; k < N - 1; k++) {
                if (((i + j + k) & 1) != color)
                    continue;
                const size_t q = IDX(i, j, k);
                double s = v[q - NN] + v[q + NN];
                s = s + v[q - N];
                s = s + v[q + N];
                s = s + v[q - 1];
                s = s + v[q + 1];
                s = s - hSq * d[q];
                v[q] = sixth * s; /* mg_3d.h:438-443 */
                ghost(v, N, h, p, i, j, k, v[q]);
            }
}

void orc_es_smooth(double *v, const double *d, int N, double h, int post, int iters, const orc_es_params *p)
{
    for (int s = 0; s < iters; s++) { /* pre: red, black (mg_3d.h:657-702); post: black, red (:728-773) */
        orc_es_smooth_color(v, d, N, h, post ? 0 : 1, p);
        orc_es_smooth_color(v, d, N, h, post ? 1 : 0, p);
    }
}

void orc_es_ghost_all(double *v, int N, double h, const orc_es_params *p)
{
    for (int i = 1; i < N - 1; i++)
        for (int j = 1; j < N - 1; j++)
            for (int k = 1; k < N - 1; k++)
                ghost(v, N, h, p, i, j, k, v[IDX(i, j, k)]);
}

double orc_es_vcycle(double **u, double **f, double **res, double h, int q, int numLevels, int iters, int N,
                     const double *LU, const orc_es_params *p)
{
    double *v = u[q];
    if (q < numLevels - 1)
        memset(v, 0, sizeof(double) * (size_t)N * N * N); /* mg_3d.h:1258-1259 */
    if (q == 0) {
        orc_lu_solve(LU, N * N * N, f[0], v); /* :1270 */
        orc_es_ghost_all(v, N, h, p);
        return 0.;
    }
    orc_es_smooth(v, f[q], N, h, 0, iters, p); /* :1282 */
    orc_residual(v, f[q], N, h, res[q]);       /* :1294 */
    const int Nc = (N + 1) / 2;
    orc_restrict(res[q], N, f[q - 1], Nc);     /* :1310 */
    orc_es_vcycle(u, f, res, 2 * h, q - 1, numLevels, iters, Nc, LU, p);
    orc_prolong(u[q - 1], Nc, v, N);           /* :1331 */
    orc_es_fill(v, N, h, p, q == numLevels - 1 ? 1. : 0.);
    orc_es_smooth(v, f[q], N, h, 1, iters, p); /* :1341 */
    return orc_residual(v, f[q], N, h, NULL);  /* :1354 */
}

/* coarsest operator: orc_coarse_matrix, with the zero-gradient row x_wall - x_front = b on every wall point */
void orc_es_coarse_matrix(double *A, int N, double h, const orc_es_params *p)
{
    const size_t NN = (size_t)N * N, n = NN * N;
    orc_coarse_matrix(A, N, h);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            for (int k = 0; k < N; k++) {
                const int fi = i == 0 || i == N - 1, fj = j == 0 || j == N - 1, fk = k == 0 || k == N - 1;
                if (fi + fj + fk != 1)
                    continue;
                const size_t q = IDX(i, j, k);
                size_t front;
                if (fi) {
                    if (i == 0 ? orc_es_dirichlet_x0(p, h, j, k) : orc_es_dirichlet_xl(p, h, j, k))
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

/* the whole problem: c, L, iters as the reference's argv; `cycles` V-cycles; norms[cycles]; u_out = finest u */
void orc_es_run(int c, int L, int iters, int cycles, const orc_es_params *p, double *norms, double *u_out, double *init_norm)
{
    double **u = (double **)malloc(sizeof(double *) * L), **f = (double **)malloc(sizeof(double *) * L),
           **r = (double **)malloc(sizeof(double *) * L);
    for (int l = 0; l < L; l++) {
        const size_t n = (size_t)(c - 1) * (1u << l) + 1;
        u[l] = (double *)calloc(n * n * n, sizeof(double));
        f[l] = (double *)calloc(n * n * n, sizeof(double));
        r[l] = (double *)calloc(n * n * n, sizeof(double));
    }
    const int N = (c - 1) * (1 << (L - 1)) + 1;
    const double h = p->length / (N - 1);
    const int n0 = c * c * c;
    double *A = (double *)calloc((size_t)n0 * n0, sizeof(double));
    orc_es_coarse_matrix(A, c, h * (1 << (L - 1)), p); /* spacing of the coarsest level, mg_3d.h:287 */
    orc_lu_factor(A, n0);
    orc_es_fill(u[L - 1], N, h, p, 1.);
    if (init_norm)
        *init_norm = orc_residual(u[L - 1], f[L - 1], N, h, NULL);
    for (int cy = 0; cy < cycles; cy++)
        norms[cy] = orc_es_vcycle(u, f, r, h, L - 1, L, iters, N, A, p);
    if (u_out)
        memcpy(u_out, u[L - 1], sizeof(double) * (size_t)N * N * N);
    for (int l = 0; l < L; l++) {
        free(u[l]);
        free(f[l]);
        free(r[l]);
    }
    free(u);
    free(f);
    free(r);
    free(A);
}
