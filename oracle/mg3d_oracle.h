/*
 * mg3d_oracle.h -- CPU restatement of the reference's 3D multigrid V-cycle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.
 * The product path (libmg3d.so, HIP) never links or calls it.
 *
 * Every function cites the reference lines (under /root/reference) whose
 * arithmetic it restates.  Association order of every floating-point
 * expression follows the reference literally; build with -ffp-contract=off.
 *
 * Parity status: PINNED.  oracle/Makefile builds oracle/_ref/ from the
 * unmodified reference sources; tests/golden/ holds vectors produced by that
 * build (generator: oracle/gen_golden.py) and tests/test_oracle_golden.py
 * checks this restatement bit-for-bit against them.
 *
 * Layout of every grid: idx = N*N*i + N*j + k, k contiguous (mg_3d.h:43-44).
 */
#ifndef MG3D_ORACLE_H
#define MG3D_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* mg_3d.h:89-90 */
double orc_bc_func(double x, double y, double z);
/* mg_3d.h:1147-1239 : Dirichlet values on the six faces of v */
void orc_fill_boundary(double *v, int N, double h);
/* mg_3d.h:147-273 : dense (N^3)x(N^3) coarse operator, A zeroed by caller */
void orc_coarse_matrix(double *A, int N, double h);
/* gauss_elim.h:9-29 */
void orc_lu_factor(double *a, int n);
/* the same factor, skipping operations with an exact zero factor (wide coarse grids; pinned to orc_lu_factor by the tests) */
void orc_lu_factor_banded(double *a, int n);
/* gauss_elim.h:31-60 */
void orc_lu_solve(const double *LU, int n, const double *b, double *x);

/* mg_3d.h:432-443 + 658-702: one colour pass; colour 1 = "red" ((i+j+k) odd),
 * colour 0 = "black" ((i+j+k) even) */
void orc_smooth_color(double *v, const double *d, int N, double h, int color);
/* mg_3d.h:640-709 : iters x (red, black) */
void orc_pre_smooth(double *v, const double *d, int N, double h, int iters);
/* mg_3d.h:711-781 : iters x (black, red) */
void orc_post_smooth(double *v, const double *d, int N, double h, int iters);
/* mg_3d.h:794-842 : returns sqrt(sum diff^2); res may be NULL.  The sum is
 * accumulated per i-plane chunk of the OpenMP static schedule and the chunk
 * partials are added in chunk order, so with 1 thread it is the reference's
 * sequential sum. */
double orc_residual(const double *v, const double *d, int N, double h, double *res);
/* mg_3d.h:844-998 */
void orc_restrict(const double *r, int Nf, double *dc, int Nc);
/* mg_3d.h:1000-1145 */
void orc_prolong(const double *ec, int Nc, double *ef, int Nf);
/* mg_3d.h:783-792 */
double orc_l2norm(const double *d, long n);

/* mg_3d.h:1242-1362 : one V-cycle from level q downwards.  u,f,res are arrays
 * of numLevels level pointers, level l having ((c-1)*2^l+1)^3 points
 * (mg_3d.h:41).  Returns the post-smoothing residual norm of level q. */
double orc_vcycle(double **u, double **f, double **res, double h, int q,
                  int numLevels, int iters, int N, const double *LU);

/* SolverFMGInitialize as specified by mg_dirichlet_analytic.c:771-806 (commented copy mg_3d.h:1364-1404):
 * BCs on u[0], direct solve, then per level: prolong the coarser solution into u[l], impose the BCs, zero
 * u[l-1], one V-cycle from level l.  (As in the reference, a V-cycle started below the finest level zeroes
 * its own guess first, mg_3d.h:1258.)  Every step is one of the pinned operators above. */
void orc_fmg_initialize(double **u, double **d, double **r, int c, int numLevels, int iters, double grid_length,
                        const double *LU);

/* Convenience driver used by tests and by bench.py's cpu_baseline leg:
 * sets up the reference's test problem (test_mg_3d.c:11-33: BC values into
 * the faces of both d and u on the finest level, interior zero), runs
 * `cycles` V-cycles and writes their norms to norms[].  If u_out is non-NULL
 * the finest-level solution (N^3 doubles) is copied there.  Returns seconds
 * spent in the cycle loop (omp_get_wtime around it, as test_mg_3d.c:36,68).
 * coarse_h_mode 0: coarse matrix built with h*2^(L-1) (mg_3d.h:287);
 *               1: built with the finest h (test_mg_3d_dirichlet.c:40 quirk),
 *                  and BC values written into u only (d stays 0, :44). */
double orc_run_problem(int c, int L, int iters, int cycles, int coarse_h_mode,
                       double *norms, double *u_out, double *init_norm);

int orc_max_threads(void);
void orc_set_threads(int n);

#ifdef __cplusplus
}
#endif

/* ---- single precision / damped Jacobi / F-cycle variant (mg3d_oracle_f32.c; PARITY UNPINNED, see its header) */
void orc32_fill_boundary(float *v, int N, double h);
void orc32_jacobi(const float *vin, const float *d, float *vout, int N, float h, float omega);
void orc32_smooth(float *v, const float *d, float *scratch, int N, float h, float omega, int iters);
double orc32_residual(const float *v, const float *d, int N, float h, float *res);
void orc32_restrict(const float *r, int Nf, float *dc, int Nc);
void orc32_prolong(const float *ec, int Nc, float *ef, int Nf);
void orc32_coarse_solve(const double *LU, int n, const float *b, float *x);
double orc32_vcycle(float **u, float **d, float **r, float **scratch, double hd, int q, int iters, float omega, int N,
                    const double *LU);
void orc32_fmg_initialize(float **u, float **d, float **r, float **scratch, int c, int numLevels, int iters,
                          float omega, double grid_length, const double *LU);
double orc32_run_problem(int c, int L, int iters, double omega, int cycles, int use_fmg, double *norms, float *u_out);

#endif
