/*
 * mg3d_oracle_f32.c -- CPU restatement of the single-precision / damped-Jacobi / F-cycle variant.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 *
 * PARITY UNPINNED.  The reference (knram06/multigrid_parallel) has no fp32 arithmetic, no Jacobi smoother and
 * only a commented-out FMG start, so there is no reference output to pin this file against.  It states, in plain
 * C, exactly what csrc/mg3d_f32.hip is meant to compute, keeping the reference's association wherever the
 * reference has the operator in double:
 *   seven-point sum      mg_3d.h:438-443     ((((v[p-NN]+v[p+NN])+v[p-N])+v[p+N])+v[p-1])+v[p+1]
 *   residual             mg_3d.h:819-821     d[p] - invHsq*(sum6 - 6*v[p])
 *   restriction          mg_3d.h:844-998     faces injected, interior 27-point sum in ti,tj,tk order from 0
 *   prolongation         mg_3d.h:1000-1145   parent orders per parity class
 *   coarsest solve       gauss_elim.h:31-60  the reference's LU in double; rhs widened, result rounded
 *   V-cycle              mg_3d.h:1242-1362   with `iters` damped-Jacobi sweeps in place of the RB-GS smoothers
 *   F-cycle start        mg_dirichlet_analytic.c:771-806
 * All grid arithmetic is IEEE binary32 (compile with -ffp-contract=off; x86-64 SSE has no excess precision).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mg3d_oracle.h"

#define IDX(N, i, j, k) ((size_t)(N) * (N) * (i) + (size_t)(N) * (j) + (k))

void orc32_fill_boundary(float *v, int N, double h)
{
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            for (int k = 0; k < N; k++)
                if (i == 0 || i == N - 1 || j == 0 || j == N - 1 || k == 0 || k == N - 1)
                    v[IDX(N, i, j, k)] = (float)orc_bc_func(i * h, j * h, k * h);
}

static float sum6(const float *v, int N, size_t p)
{
    const size_t NN = (size_t)N * N;
    float s = v[p - NN] + v[p + NN];
    s = s + v[p - N];
    s = s + v[p + N];
    s = s + v[p - 1];
    s = s + v[p + 1];
    return s;
}

/* one sweep: vout = vin + omega*((1/6)(sum6 - h^2 d) - vin) on the interior, copy on the boundary */
void orc32_jacobi(const float *vin, const float *d, float *vout, int N, float h, float omega)
{
    const float hSq = h * h, sixth = 1.0f / 6.0f;
    memcpy(vout, vin, sizeof(float) * (size_t)N * N * N);
#pragma omp parallel for schedule(static)
    for (int i = 1; i < N - 1; i++)
        for (int j = 1; j < N - 1; j++)
            for (int k = 1; k < N - 1; k++) {
                const size_t p = IDX(N, i, j, k);
                const float s = sum6(vin, N, p) - hSq * d[p];
                const float gs = sixth * s;
                vout[p] = vin[p] + omega * (gs - vin[p]);
            }
}

/* `iters` sweeps; the result ends in v (scratch is the second buffer) */
void orc32_smooth(float *v, const float *d, float *scratch, int N, float h, float omega, int iters)
{
    float *a = v, *b = scratch;
    for (int it = 0; it < iters; it++) {
        orc32_jacobi(a, d, b, N, h, omega);
        float *t = a;
        a = b;
        b = t;
    }
    if (a != v)
        memcpy(v, a, sizeof(float) * (size_t)N * N * N);
}

double orc32_residual(const float *v, const float *d, int N, float h, float *res)
{
    const float invHsq = 1.0f / (h * h);
    double ret = 0.;
    for (int i = 1; i < N - 1; i++)
        for (int j = 1; j < N - 1; j++)
            for (int k = 1; k < N - 1; k++) {
                const size_t p = IDX(N, i, j, k);
                const float s = sum6(v, N, p) - 6 * v[p];
                const float diff = d[p] - invHsq * s;
                if (res)
                    res[p] = diff;
                ret += (double)diff * (double)diff;
            }
    return sqrt(ret);
}

void orc32_restrict(const float *r, int Nf, float *dc, int Nc)
{
    const float w1[3] = {0.25f, 0.5f, 0.25f};
    for (int ic = 0; ic < Nc; ic++)
        for (int jc = 0; jc < Nc; jc++)
            for (int kc = 0; kc < Nc; kc++) {
                const int face = ic == 0 || ic == Nc - 1 || jc == 0 || jc == Nc - 1 || kc == 0 || kc == Nc - 1;
                float val;
                if (face) {
                    val = r[IDX(Nf, 2 * ic, 2 * jc, 2 * kc)];
                } else {
                    val = 0.f;
                    for (int ti = 0; ti < 3; ti++)
                        for (int tj = 0; tj < 3; tj++)
                            for (int tk = 0; tk < 3; tk++) {
                                const float w = w1[ti] * w1[tj] * w1[tk];
                                val += r[IDX(Nf, 2 * ic - 1 + ti, 2 * jc - 1 + tj, 2 * kc - 1 + tk)] * w;
                            }
                }
                dc[IDX(Nc, ic, jc, kc)] = val;
            }
}

void orc32_prolong(const float *ec, int Nc, float *ef, int Nf)
{
#define C(a, b, c) ec[IDX(Nc, a, b, c)]
    for (int i = 0; i < Nf; i++)
        for (int j = 0; j < Nf; j++)
            for (int k = 0; k < Nf; k++) {
                const int oi = i & 1, oj = j & 1, ok = k & 1;
                const int il = (i - oi) / 2, jl = (j - oj) / 2, kl = (k - ok) / 2;
                float t;
                switch (oi + oj + ok) {
                case 3:
                    t = C(il, jl, kl) + C(il, jl, kl + 1);
                    t = t + C(il, jl + 1, kl);
                    t = t + C(il, jl + 1, kl + 1);
                    t = t + C(il + 1, jl, kl);
                    t = t + C(il + 1, jl, kl + 1);
                    t = t + C(il + 1, jl + 1, kl);
                    t = t + C(il + 1, jl + 1, kl + 1);
                    t = t * 0.125f;
                    break;
                case 2:
                    if (!oi)
                        t = ((C(il, jl, kl) + C(il, jl + 1, kl)) + C(il, jl, kl + 1)) + C(il, jl + 1, kl + 1);
                    else if (!oj)
                        t = ((C(il, jl, kl) + C(il + 1, jl, kl)) + C(il, jl, kl + 1)) + C(il + 1, jl, kl + 1);
                    else
                        t = ((C(il, jl, kl) + C(il, jl + 1, kl)) + C(il + 1, jl, kl)) + C(il + 1, jl + 1, kl);
                    t = t * 0.25f;
                    break;
                case 1:
                    t = (C(il, jl, kl) + C(il + oi, jl + oj, kl + ok)) * 0.5f;
                    break;
                default:
                    t = C(il, jl, kl);
                }
                ef[IDX(Nf, i, j, k)] += t;
            }
#undef C
}

void orc32_coarse_solve(const double *LU, int n, const float *b, float *x)
{
    double *bd = (double *)malloc(sizeof(double) * (size_t)n), *xd = (double *)malloc(sizeof(double) * (size_t)n);
    for (int p = 0; p < n; p++)
        bd[p] = (double)b[p];
    orc_lu_solve(LU, n, bd, xd);
    for (int p = 0; p < n; p++)
        x[p] = (float)xd[p];
    free(bd);
    free(xd);
}

/* hd = spacing of level q in double (the hierarchy halves it exactly); each level rounds its own to float */
double orc32_vcycle(float **u, float **d, float **r, float **scratch, double hd, int q, int iters, float omega, int N,
                    const double *LU)
{
    if (q == 0) {
        orc32_coarse_solve(LU, N * N * N, d[0], u[0]);
        return 0.;
    }
    const float h = (float)hd;
    const int Nc = (N + 1) / 2;
    orc32_smooth(u[q], d[q], scratch[q], N, h, omega, iters);
    orc32_residual(u[q], d[q], N, h, r[q]);
    orc32_restrict(r[q], N, d[q - 1], Nc);
    memset(u[q - 1], 0, sizeof(float) * (size_t)Nc * Nc * Nc);
    orc32_vcycle(u, d, r, scratch, 2 * hd, q - 1, iters, omega, Nc, LU);
    orc32_prolong(u[q - 1], Nc, u[q], N);
    orc32_smooth(u[q], d[q], scratch[q], N, h, omega, iters);
    return orc32_residual(u[q], d[q], N, h, NULL);
}

void orc32_fmg_initialize(float **u, float **d, float **r, float **scratch, int c, int numLevels, int iters,
                          float omega, double grid_length, const double *LU)
{
    int N = c;
    double h = grid_length / (c - 1);
    orc32_fill_boundary(u[0], N, h);
    orc32_coarse_solve(LU, N * N * N, d[0], u[0]);
    for (int l = 1; l < numLevels; l++) {
        const int Nc = N;
        N = 2 * N - 1;
        h = h * 0.5;
        orc32_prolong(u[l - 1], Nc, u[l], N);
        orc32_fill_boundary(u[l], N, h);
        memset(u[l - 1], 0, sizeof(float) * (size_t)Nc * Nc * Nc);
        orc32_vcycle(u, d, r, scratch, h, l, iters, omega, N, LU);
    }
}

/* The test problem of test_mg_3d.c:11-29 in binary32: boundary values on the faces of the finest u and d
 * (with use_fmg: on the faces of d on every level, the F-cycle start first), then `cycles` V-cycles.
 * norms[c] after each cycle, u_out = finest solution.  Returns the last norm. */
double orc32_run_problem(int c, int L, int iters, double omega, int cycles, int use_fmg, double *norms, float *u_out)
{
    const int Nf = (c - 1) * (1 << (L - 1)) + 1;
    const double hf = 1.0 / (Nf - 1);
    float **u = (float **)malloc(sizeof(float *) * L), **d = (float **)malloc(sizeof(float *) * L);
    float **r = (float **)malloc(sizeof(float *) * L), **s = (float **)malloc(sizeof(float *) * L);
    for (int l = 0; l < L; l++) {
        const size_t n = (size_t)(c - 1) * (1u << l) + 1;
        u[l] = (float *)calloc(n * n * n, sizeof(float));
        d[l] = (float *)calloc(n * n * n, sizeof(float));
        r[l] = (float *)calloc(n * n * n, sizeof(float));
        s[l] = (float *)calloc(n * n * n, sizeof(float));
    }
    const int n0 = c * c * c;
    double *LU = (double *)calloc((size_t)n0 * n0, sizeof(double));
    orc_coarse_matrix(LU, c, hf * (1 << (L - 1)));
    orc_lu_factor(LU, n0);
    if (use_fmg) {
        double h = 1.0 / (c - 1);
        for (int l = 0; l < L; l++, h *= 0.5)
            orc32_fill_boundary(d[l], (c - 1) * (1 << l) + 1, h);
        orc32_fmg_initialize(u, d, r, s, c, L, iters, (float)omega, 1.0, LU);
    } else {
        orc32_fill_boundary(u[L - 1], Nf, hf);
        orc32_fill_boundary(d[L - 1], Nf, hf);
    }
    double last = 0.;
    for (int cyc = 0; cyc < cycles; cyc++) {
        last = orc32_vcycle(u, d, r, s, hf, L - 1, iters, (float)omega, Nf, LU);
        if (norms)
            norms[cyc] = last;
    }
    if (u_out)
        memcpy(u_out, u[L - 1], sizeof(float) * (size_t)Nf * Nf * Nf);
    for (int l = 0; l < L; l++) {
        free(u[l]);
        free(d[l]);
        free(r[l]);
        free(s[l]);
    }
    free(u);
    free(d);
    free(r);
    free(s);
    free(LU);
    return last;
}
