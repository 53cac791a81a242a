/*
 * mg3d_oracle.c -- CPU restatement of the reference V-cycle (see header).
 * TEST INFRASTRUCTURE ONLY: never linked into or called from the product.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -fPIC -shared (oracle/Makefile).
 * All arithmetic is IEEE fp64, all expressions keep the reference's C
 * left-to-right association so results are bit-identical to the reference
 * compiled by gcc -O2 for baseline x86-64 (no FMA contraction).
 */
#include "mg3d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define IDX(N, i, j, k) ((long)(N) * (N) * (i) + (long)(N) * (j) + (k))

int orc_max_threads(void) { return omp_get_max_threads(); }
void orc_set_threads(int n) { omp_set_num_threads(n); }

/* mg_3d.h:89-90 */
double orc_bc_func(double x, double y, double z) { return x * x - 2 * y * y + z * z; }

/* mg_3d.h:1147-1239.  The reference walks the six faces one after the other;
 * the value at a point does not depend on which face loop wrote it last
 * (i*h, j*h, k*h are the same products), so one predicate loop is equivalent. */
void orc_fill_boundary(double *v, int N, double h)
{
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            const int edge_ij = (i == 0 || i == N - 1 || j == 0 || j == N - 1);
            if (edge_ij) {
                for (int k = 0; k < N; k++)
                    v[IDX(N, i, j, k)] = orc_bc_func(i * h, j * h, k * h);
            } else {
                v[IDX(N, i, j, 0)] = orc_bc_func(i * h, j * h, 0 * h);
                v[IDX(N, i, j, N - 1)] = orc_bc_func(i * h, j * h, (N - 1) * h);
            }
        }
}

/* mg_3d.h:147-273: identity rows on boundary nodes (:179-185), interior rows
 * (1,1,1,1,1,1,-6)/h^2 with oneCoeff = 1.*invHsq, sixCoeff = 6.*invHsq
 * (:155-159, 257-268). */
void orc_coarse_matrix(double *A, int N, double h)
{
    const long n = (long)N * N * N;
    const double hSq = h * h;
    const double invHsq = 1. / hSq;
    const double one = 1. * invHsq, six = 6. * invHsq;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            for (int k = 0; k < N; k++) {
                const long row = IDX(N, i, j, k);
                double *a = A + row * n;
                if (i == 0 || i == N - 1 || j == 0 || j == N - 1 || k == 0 || k == N - 1) {
                    a[row] = 1.;
                } else {
                    a[row - (long)N * N] = one;
                    a[row + (long)N * N] = one;
                    a[row - N] = one;
                    a[row + N] = one;
                    a[row - 1] = one;
                    a[row + 1] = one;
                    a[row] = -six;
                }
            }
}

/* gauss_elim.h:9-29: Doolittle, unit lower, no pivoting, in place, row major. */
void orc_lu_factor(double *a, int n)
{
    for (int i = 0; i < n - 1; i++) {
        const double *ri = a + (long)n * i;
        const double pinv = 1. / ri[i];
        for (int k = i + 1; k < n; k++) {
            double *rk = a + (long)n * k;
            const double z = rk[i] * pinv;
            rk[i] = z;
            for (int j = i + 1; j < n; j++)
                rk[j] -= z * ri[j];
        }
    }
}

/* The same elimination for coarse grids whose dense O(n^3) sweep is out of reach (c = 17: n = 4913, c = 33:
 * n = 35937, 1.5e13 operations): every operation of gauss_elim.h:9-29 that can change a value is performed, in the
 * reference's order per entry; the ones left out have an exact zero as a factor.  Rows are described by the first
 * non-zero column they start with (elimination without pivoting never extends a row to the left), pivot rows by
 * their current last non-zero column.  tests/test_oracle_golden.py pins this against orc_lu_factor (byte for byte,
 * signed zeros included) at c = 3, 5, 9 and on random banded matrices. */
void orc_lu_factor_banded(double *a, int n)
{
    int *first = (int *)malloc(sizeof(int) * (size_t)n), *lastc = (int *)malloc(sizeof(int) * (size_t)n);
    int *reach = (int *)malloc(sizeof(int) * (size_t)n); /* reach[i] = last row whose first non-zero column is <= i */
    double *pinv = (double *)malloc(sizeof(double) * (size_t)n);
    for (int k = 0; k < n; k++) {
        const double *r = a + (long)n * k;
        int f = 0, l = n - 1;
        while (f < k && r[f] == 0.)
            f++;
        while (l > k && r[l] == 0.)
            l--;
        first[k] = f;
        lastc[k] = l;
    }
    for (int i = 0; i < n; i++)
        reach[i] = i;
    for (int k = 0; k < n; k++) /* rows k with first[k] <= i for all i >= first[k] */
        if (reach[first[k]] < k)
            reach[first[k]] = k;
    for (int i = 1; i < n; i++)
        if (reach[i] < reach[i - 1])
            reach[i] = reach[i - 1];
    for (int i = 0; i < n - 1; i++) {
        const double *ri = a + (long)n * i;
        const double aii_inv = 1. / ri[i]; /* :17 */
        const int li = lastc[i], kend = reach[i];
        pinv[i] = aii_inv;
#pragma omp parallel for schedule(static) if (kend - i > 64)
        for (int k = i + 1; k <= kend; k++) {
            double *rk = a + (long)n * k;
            const double z = rk[i] * aii_inv; /* :21 */
            rk[i] = z;                        /* :22 */
            for (int j = i + 1; j <= li; j++)
                rk[j] -= z * ri[j];           /* :25 */
            if (lastc[k] < li && z != 0.)
                lastc[k] = li;
        }
    }
    /* rows never reached by pivot i (first[k] > i): the dense loop multiplies their untouched zero by 1/a[i][i] */
    for (int k = 1; k < n; k++) {
        double *rk = a + (long)n * k;
        for (int i = 0; i < k && i < n - 1; i++)
            if (k > reach[i])
                rk[i] = rk[i] * pinv[i];
    }
    free(first);
    free(lastc);
    free(reach);
    free(pinv);
}

/* gauss_elim.h:31-60: forward j ascending (:39-41), backward j descending
 * from n-1 (:54-55), divide by the diagonal (:57). */
void orc_lu_solve(const double *LU, int n, const double *b, double *x)
{
    for (int i = 0; i < n; i++) {
        const double *row = LU + (long)n * i;
        double sum = 0.;
        for (int j = 0; j < i; j++)
            sum += row[j] * x[j];
        x[i] = b[i] - sum;
    }
    for (int i = n - 1; i >= 0; i--) {
        const double *row = LU + (long)n * i;
        double sum = 0.;
        for (int j = n - 1; j > i; j--)
            sum += row[j] * x[j];
        x[i] = (x[i] - sum) / row[i];
    }
}

/* mg_3d.h:438-443 with kOffset of :669 (red) / :693 (black).
 * color 1 (red)  : first k = 1 + (i+j)%2     -> (i+j+k) odd
 * color 0 (black): first k = 1 + (i+j+1)%2   -> (i+j+k) even */
void orc_smooth_color(double *v, const double *d, int N, double h, int color)
{
    const double hSq = h * h;
    const double sixth = 1. / 6;
    const long NN = (long)N * N;
#pragma omp parallel for schedule(static)
    for (int i = 1; i < N - 1; i++)
        for (int j = 1; j < N - 1; j++) {
            const int k0 = 1 + (i + j + (color ? 0 : 1)) % 2;
            for (int k = k0; k < N - 1; k += 2) {
                const long p = IDX(N, i, j, k);
                v[p] = sixth * (v[p - NN] + v[p + NN] + v[p - N] + v[p + N] + v[p - 1] + v[p + 1] - hSq * d[p]);
            }
        }
}

void orc_pre_smooth(double *v, const double *d, int N, double h, int iters)
{
    for (int s = 0; s < iters; s++) {
        orc_smooth_color(v, d, N, h, 1);
        orc_smooth_color(v, d, N, h, 0);
    }
}

void orc_post_smooth(double *v, const double *d, int N, double h, int iters)
{
    for (int s = 0; s < iters; s++) {
        orc_smooth_color(v, d, N, h, 0);
        orc_smooth_color(v, d, N, h, 1);
    }
}

/* mg_3d.h:794-842.  diff as :819-821; res written on the interior only. */
double orc_residual(const double *v, const double *d, int N, double h, double *res)
{
    const double invHsq = 1. / (h * h);
    const long NN = (long)N * N;
    const int nt = omp_get_max_threads();
    double *part = (double *)calloc((size_t)nt, sizeof(double));
#pragma omp parallel
    {
        double ret = 0.;
#pragma omp for schedule(static)
        for (int i = 1; i < N - 1; i++)
            for (int j = 1; j < N - 1; j++)
                for (int k = 1; k < N - 1; k++) {
                    const long p = IDX(N, i, j, k);
                    const double diff =
                        d[p] - invHsq * (v[p - NN] + v[p + NN] + v[p - N] + v[p + N] + v[p - 1] + v[p + 1] - 6 * v[p]);
                    if (res)
                        res[p] = diff;
                    ret += diff * diff;
                }
        part[omp_get_thread_num()] = ret;
    }
    double tot = 0.;
    for (int t = 0; t < nt; t++)
        tot += part[t];
    free(part);
    return sqrt(tot);
}

/* mg_3d.h:844-998: injection on the six coarse faces (:879-958), 27-point
 * full weighting on the coarse interior accumulated ti -> tj -> tk from 0
 * (:973-989), weights (1/4,1/2,1/4)^3 (:851-872). */
void orc_restrict(const double *r, int Nf, double *dc, int Nc)
{
    static const double w1[3] = {0.25, 0.5, 0.25};
    double w[3][3][3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++)
            for (int c = 0; c < 3; c++)
                w[a][b][c] = w1[a] * w1[b] * w1[c]; /* exact powers of two */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < Nc; i++)
        for (int j = 0; j < Nc; j++)
            for (int k = 0; k < Nc; k++) {
                const int face = (i == 0 || i == Nc - 1 || j == 0 || j == Nc - 1 || k == 0 || k == Nc - 1);
                if (face) {
                    dc[IDX(Nc, i, j, k)] = r[IDX(Nf, 2 * i, 2 * j, 2 * k)];
                } else {
                    double val = 0.;
                    for (int ti = 0; ti < 3; ti++)
                        for (int tj = 0; tj < 3; tj++)
                            for (int tk = 0; tk < 3; tk++)
                                val += r[IDX(Nf, 2 * i - 1 + ti, 2 * j - 1 + tj, 2 * k - 1 + tk)] * w[ti][tj][tk];
                    dc[IDX(Nc, i, j, k)] = val;
                }
            }
}

/* mg_3d.h:1000-1145.  For each fine point the coarse parents are summed in
 * exactly the order the reference lists them, starting from retVal = 0.:
 *   odd,odd,odd  (:1028-1048): (0,0,0)(0,0,1)(0,1,0)(0,1,1)(1,0,0)(1,0,1)(1,1,0)(1,1,1) * 0.125
 *   i even       (:1064-1067): (jl,kl)(jl+1,kl)(jl,kl+1)(jl+1,kl+1)             * 0.25
 *   j even       (:1075-1078): (il,kl)(il+1,kl)(il,kl+1)(il+1,kl+1)             * 0.25
 *   k even       (:1085-1088): (il,jl)(il,jl+1)(il+1,jl)(il+1,jl+1)             * 0.25
 *   one odd axis (:1110-1133): low + high                                      * 0.5
 *   all even     (:1138)     : copy
 */
void orc_prolong(const double *ec, int Nc, double *ef, int Nf)
{
#define EC(a, b, c) ec[IDX(Nc, a, b, c)]
#pragma omp parallel for schedule(static)
    for (int i = 0; i < Nf; i++)
        for (int j = 0; j < Nf; j++)
            for (int k = 0; k < Nf; k++) {
                const int oi = i % 2, oj = j % 2, ok = k % 2;
                const int il = (i - oi) / 2, jl = (j - oj) / 2, kl = (k - ok) / 2;
                double t = 0.;
                switch (oi + oj + ok) {
                case 3:
                    t += EC(il, jl, kl);
                    t += EC(il, jl, kl + 1);
                    t += EC(il, jl + 1, kl);
                    t += EC(il, jl + 1, kl + 1);
                    t += EC(il + 1, jl, kl);
                    t += EC(il + 1, jl, kl + 1);
                    t += EC(il + 1, jl + 1, kl);
                    t += EC(il + 1, jl + 1, kl + 1);
                    t *= 0.125;
                    break;
                case 2:
                    if (!oi) {
                        t += EC(il, jl, kl);
                        t += EC(il, jl + 1, kl);
                        t += EC(il, jl, kl + 1);
                        t += EC(il, jl + 1, kl + 1);
                    } else if (!oj) {
                        t += EC(il, jl, kl);
                        t += EC(il + 1, jl, kl);
                        t += EC(il, jl, kl + 1);
                        t += EC(il + 1, jl, kl + 1);
                    } else {
                        t += EC(il, jl, kl);
                        t += EC(il, jl + 1, kl);
                        t += EC(il + 1, jl, kl);
                        t += EC(il + 1, jl + 1, kl);
                    }
                    t *= 0.25;
                    break;
                case 1:
                    t += EC(il, jl, kl);
                    t += EC(il + oi, jl + oj, kl + ok);
                    t *= 0.5;
                    break;
                default:
                    t = EC(il, jl, kl);
                }
                ef[IDX(Nf, i, j, k)] += t;
            }
#undef EC
}

/* mg_3d.h:783-792 */
double orc_l2norm(const double *d, long n)
{
    double ret = 0.;
    for (long i = 0; i < n; i++)
        ret += d[i] * d[i];
    return sqrt(ret);
}

/* mg_3d.h:1242-1362 */
double orc_vcycle(double **u, double **f, double **res, double h, int q, int numLevels, int iters, int N,
                  const double *LU)
{
    double *v = u[q];
    if (q < numLevels - 1)
        memset(v, 0, sizeof(double) * (size_t)N * N * N); /* :1258-1259 */
    if (q == 0) {
        orc_lu_solve(LU, N * N * N, f[0], v); /* :1270 */
        return 0.;
    }
    orc_pre_smooth(v, f[q], N, h, iters);        /* :1282 */
    orc_residual(v, f[q], N, h, res[q]);         /* :1294 */
    const int Nc = (N + 1) / 2;                  /* :1302 */
    orc_restrict(res[q], N, f[q - 1], Nc);       /* :1310 */
    orc_vcycle(u, f, res, 2 * h, q - 1, numLevels, iters, Nc, LU); /* :1303,1320 */
    orc_prolong(u[q - 1], Nc, v, N);             /* :1331 */
    orc_post_smooth(v, f[q], N, h, iters);       /* :1341 */
    return orc_residual(v, f[q], N, h, NULL);    /* :1354 */
}

void orc_fmg_initialize(double **u, double **d, double **r, int c, int numLevels, int iters, double grid_length,
                        const double *LU)
{
    int N = c;
    double h = grid_length / (c - 1);            /* mg_dirichlet_analytic.c:779 */
    orc_fill_boundary(u[0], N, h);               /* :780 */
    orc_lu_solve(LU, N * N * N, d[0], u[0]);     /* :783 */
    for (int l = 1; l < numLevels; l++) {
        const int Nc = N;
        N = 2 * N - 1;                           /* :790 */
        h = h * 0.5;                             /* :791 */
        orc_prolong(u[l - 1], Nc, u[l], N);      /* :795 */
        orc_fill_boundary(u[l], N, h);           /* :798 */
        memset(u[l - 1], 0, sizeof(double) * (size_t)Nc * Nc * Nc); /* :801 */
        orc_vcycle(u, d, r, h, l, numLevels, iters, N, LU);         /* :804 */
    }
}

static double **alloc_levels(int c, int L) /* mg_3d.h:30-48 */
{
    double **a = (double **)malloc(sizeof(double *) * (size_t)L);
    for (int l = 0; l < L; l++) {
        const size_t n = (size_t)(c - 1) * (1u << l) + 1;
        a[l] = (double *)calloc(n * n * n, sizeof(double));
    }
    return a;
}

static void free_levels(double **a, int L)
{
    for (int l = 0; l < L; l++)
        free(a[l]);
    free(a);
}

double orc_run_problem(int c, int L, int iters, int cycles, int coarse_h_mode, double *norms, double *u_out,
                       double *init_norm)
{
    const int N = (c - 1) * (1 << (L - 1)) + 1; /* mg_3d.h:126-127 */
    const double h = 1.0 / (N - 1);             /* GRID_LENGTH = 1, mg_3d.h:143 */
    double **u = alloc_levels(c, L), **d = alloc_levels(c, L), **r = alloc_levels(c, L);
    const size_t n0 = (size_t)c * c * c;
    double *A = (double *)calloc(n0 * n0, sizeof(double));
    orc_coarse_matrix(A, c, coarse_h_mode ? h : h * (1 << (L - 1))); /* mg_3d.h:287 | dirichlet:40 */
    if (c > 9) /* c = 17, 33 (admissible by mg_3d.h:163): the dense sweep needs 4e10 / 1.5e13 operations */
        orc_lu_factor_banded(A, (int)n0);
    else
        orc_lu_factor(A, (int)n0);
    double init = 0.;
    if (!coarse_h_mode) {
        orc_fill_boundary(d[L - 1], N, h);           /* test_mg_3d.c:17 */
        init = orc_l2norm(d[L - 1], (long)N * N * N); /* test_mg_3d.c:26 */
        orc_fill_boundary(u[L - 1], N, h);           /* test_mg_3d.c:29 */
    } else {
        orc_fill_boundary(u[L - 1], N, h);                     /* test_mg_3d_dirichlet.c:44 */
        init = orc_residual(u[L - 1], d[L - 1], N, h, NULL); /* test_mg_3d_dirichlet.c:51 */
    }
    if (init_norm)
        *init_norm = init;
    const double t0 = omp_get_wtime();
    for (int it = 0; it < cycles; it++) {
        const double nrm = orc_vcycle(u, d, r, h, L - 1, L, iters, N, A);
        if (norms)
            norms[it] = nrm;
    }
    const double t1 = omp_get_wtime();
    if (u_out)
        memcpy(u_out, u[L - 1], sizeof(double) * (size_t)N * N * N);
    free(A);
    free_levels(u, L);
    free_levels(d, L);
    free_levels(r, L);
    return t1 - t0;
}
