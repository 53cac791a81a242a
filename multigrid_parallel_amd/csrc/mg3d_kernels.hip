/*
 * mg3d_kernels.hip -- gfx950 kernels of the multigrid V-cycle (baseline set).
 *
 * Arithmetic contract: every expression keeps the reference's association
 * (cited per kernel) and the file is compiled with -ffp-contract=off, so each
 * grid value is bit-identical to the reference CPU path; only the residual
 * norm's summation order differs (deterministic two-stage tree here).
 *
 * Layout: idx = plane*i + pitch*j + k (k contiguous, 128-byte aligned rows).
 * Colour of a point: (ig0 + i + j + k) & 1; 1 = red, 0 = black (mg_3d.h:669,693).
 */
#include "mg3d_internal.h"

#include <stdlib.h>

#include <map>
#include <mutex>

#define WAVE 64

__device__ __forceinline__ long long gidx(const Geom &g, int i, int j, int k)
{
    return g.plane * i + (long long)g.pitch * j + k;
}

/* ------------------------------------------------------------------ smoother
 * One colour pass of red-black Gauss-Seidel (smoothenAtIndex, mg_3d.h:438-443):
 *   v[p] = (1/6) * (((((((v[p-NN] + v[p+NN]) + v[p-N]) + v[p+N]) + v[p-1]) + v[p+1]) - hSq*d[p])
 * Each lane owns the k-pair (2m, 2m+1) of one row and updates the member whose
 * colour is being swept; boundary points (k = 0, nk-1) are never written. */
__global__ void __launch_bounds__(256) smooth_color_kernel(Geom g, double *__restrict__ v,
                                                           const double *__restrict__ d, double hSq, double sixth,
                                                           int color)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int i = 1 + blockIdx.z;
    if (j > g.nj - 2)
        return;
    const int k = 2 * m + ((color + g.ig0 + i + j) & 1);
    if (k < 1 || k > g.nk - 2)
        return;
    const long long p = gidx(g, i, j, k);
    double s = v[p - g.plane] + v[p + g.plane];
    s = s + v[p - g.pitch];
    s = s + v[p + g.pitch];
    s = s + v[p - 1];
    s = s + v[p + 1];
    s = s - hSq * d[p];
    v[p] = sixth * s;
}

void k_smooth_color(const Geom &g, double *v, const double *d, double hSq, int color, hipStream_t s)
{
    if (g.ni < 3 || g.nj < 3 || g.nk < 3)
        return;
    dim3 block(64, 4, 1);
    const int pairs = (g.nk + 1) / 2;
    dim3 grid((pairs + 63) / 64, (g.nj - 2 + 3) / 4, g.ni - 2);
    hipLaunchKernelGGL(smooth_color_kernel, grid, block, 0, s, g, v, d, hSq, 1. / 6, color);
}

/* ------------------------------------------------------------ boundary fill
 * setupBoundaryConditions, mg_3d.h:1147-1239: v = BCFunc(i*h, j*h, k*h) = ((x*x) - ((2*y)*y)) + (z*z) on
 * every point of the six faces (mg_3d.h:89-90; same operation order as the C expression, no contraction). */
__global__ void __launch_bounds__(256) fill_boundary_kernel(Geom g, double *__restrict__ v, double h)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int i = blockIdx.z;
    if (k >= g.nk || j >= g.nj)
        return;
    const int ig = g.ig0 + i;
    if (!(ig == 0 || ig == g.N - 1 || j == 0 || j == g.nj - 1 || k == 0 || k == g.nk - 1))
        return;
    const double x = ig * h, y = j * h, z = k * h;
    v[gidx(g, i, j, k)] = x * x - 2 * y * y + z * z;
}

void k_fill_boundary(const Geom &g, double *v, double h, hipStream_t s)
{
    dim3 grid((g.nk + 63) / 64, (g.nj + 3) / 4, g.ni);
    hipLaunchKernelGGL(fill_boundary_kernel, grid, dim3(64, 4, 1), 0, s, g, v, h);
}

/* ------------------------------------------------------------- block reduce */
__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1)
        x += __shfl_down(x, off, WAVE);
    return x;
}

/* sum over a 256-thread block, fixed order: lanes by shuffle tree, waves 0..3 sequentially */
__device__ __forceinline__ double block_sum_256(double x, double *lds4)
{
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    x = wave_sum(x);
    if ((tid & (WAVE - 1)) == 0)
        lds4[tid / WAVE] = x;
    __syncthreads();
    return ((lds4[0] + lds4[1]) + lds4[2]) + lds4[3];
}

/* final stage: one block folds np partials in a fixed order */
__global__ void __launch_bounds__(256) fold_partials_kernel(const double *__restrict__ partials, int np,
                                                            double *__restrict__ out)
{
    __shared__ double lds4[4];
    double acc = 0.;
    for (int t = threadIdx.x; t < np; t += 256)
        acc += partials[t];
    const double tot = block_sum_256(acc, lds4);
    if (threadIdx.x == 0)
        *out = tot;
}

/* ------------------------------------------------------------------ residual
 * calculateResidual, mg_3d.h:819-821:
 *   diff = d[p] - invHsq * (((((((v[p-NN]+v[p+NN])+v[p-N])+v[p+N])+v[p-1])+v[p+1]) - 6*v[p])
 * res (optional) is written on the interior only (mg_3d.h:824-825).
 * Each thread marches `chunk` planes in i keeping the i-1 / i / i+1 values of its
 * column in registers; diff^2 is reduced lane -> wave (__shfl_down) -> block. */
__global__ void __launch_bounds__(256) residual_kernel(Geom g, const double *__restrict__ v,
                                                       const double *__restrict__ d, double invHsq,
                                                       double *__restrict__ res, double *__restrict__ partials,
                                                       int chunk)
{
    __shared__ double lds4[4];
    const int k = 1 + blockIdx.x * 64 + threadIdx.x;
    const int j = 1 + blockIdx.y * 4 + threadIdx.y;
    const int i0 = 1 + blockIdx.z * chunk;
    int i1 = i0 + chunk;
    if (i1 > g.ni - 1)
        i1 = g.ni - 1;
    double acc = 0.;
    if (k <= g.nk - 2 && j <= g.nj - 2) {
        long long p = gidx(g, i0, j, k);
        double below = v[p - g.plane], here = v[p];
        for (int i = i0; i < i1; i++, p += g.plane) {
            const double above = v[p + g.plane];
            double s = below + above;
            s = s + v[p - g.pitch];
            s = s + v[p + g.pitch];
            s = s + v[p - 1];
            s = s + v[p + 1];
            s = s - 6 * here;
            const double diff = d[p] - invHsq * s;
            if (res)
                res[p] = diff;
            acc += diff * diff;
            below = here;
            here = above;
        }
    }
    const double tot = block_sum_256(acc, lds4);
    if (threadIdx.x == 0 && threadIdx.y == 0)
        partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tot;
}

__global__ void __launch_bounds__(256) fold2_partials_kernel(const double *__restrict__ pa, int na,
                                                             const double *__restrict__ pb, int nb, double *__restrict__ out)
{
    __shared__ double lds4[4];
    double acc = 0.;
    for (int t = threadIdx.x; t < na; t += 256)
        acc += pa[t];
    const double ta = block_sum_256(acc, lds4);
    __syncthreads();
    acc = 0.;
    for (int t = threadIdx.x; t < nb; t += 256)
        acc += pb[t];
    const double tb = block_sum_256(acc, lds4);
    if (threadIdx.x == 0)
        *out = ta + tb;
}

void k_fold2(const double *pa, int na, const double *pb, int nb, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(fold2_partials_kernel, dim3(1), dim3(256), 0, s, pa, na, pb, nb, out);
}

void k_fold(const double *partials, int np, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(256), 0, s, partials, np, out);
}

void k_residual(const Geom &g, const double *v, const double *d, double invHsq, double *res, double *partials,
                double *sumsq_out, hipStream_t s)
{
    if (g.ni < 3 || g.nj < 3 || g.nk < 3) {
        (void)hipMemsetAsync(sumsq_out, 0, sizeof(double), s);
        return;
    }
    const int gx = (g.nk - 2 + 63) / 64, gy = (g.nj - 2 + 3) / 4;
    int chunk = 16;
    while ((long long)gx * gy * ((g.ni - 2 + chunk - 1) / chunk) > MG3D_MAX_PARTIALS)
        chunk *= 2;
    const int gz = (g.ni - 2 + chunk - 1) / chunk;
    hipLaunchKernelGGL(residual_kernel, dim3(gx, gy, gz), dim3(64, 4, 1), 0, s, g, v, d, invHsq, res, partials,
                       chunk);
    hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(256), 0, s, partials, gx * gy * gz, sumsq_out);
}

/* GetL2NormOfVector (mg_3d.h:783-792) over every point of a level, boundary included */
__global__ void __launch_bounds__(256) sumsq_kernel(Geom g, const double *__restrict__ a,
                                                    double *__restrict__ partials)
{
    __shared__ double lds4[4];
    double acc = 0.;
    const long long rows = (long long)g.ni * g.nj;
    for (long long row = blockIdx.x * 4 + threadIdx.y; row < rows; row += (long long)gridDim.x * 4) {
        const double *r = a + (row / g.nj) * g.plane + (row % g.nj) * g.pitch;
        for (int k = threadIdx.x; k < g.nk; k += 64)
            acc += r[k] * r[k];
    }
    const double tot = block_sum_256(acc, lds4);
    if (threadIdx.x == 0 && threadIdx.y == 0)
        partials[blockIdx.x] = tot;
}

void k_sumsq(const Geom &g, const double *a, double *partials, double *sumsq_out, hipStream_t s)
{
    long long rows = (long long)g.ni * g.nj;
    int nb = (int)((rows + 3) / 4);
    if (nb > 2048)
        nb = 2048;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(64, 4, 1), 0, s, g, a, partials);
    hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(256), 0, s, partials, nb, sumsq_out);
}

/* --------------------------------------------------------------- restriction
 * restrictResidual, mg_3d.h:844-998.  Coarse faces: injection d_c = r(2i,2j,2k)
 * (:879-958).  Coarse interior: val = 0; val += r(2i-1+ti, 2j-1+tj, 2k-1+tk) * w[ti][tj][tk]
 * for ti, tj, tk = 0..2 in that nesting (:973-989), w = (1/4,1/2,1/4)^3.
 * "Face" in i means a PHYSICAL boundary plane (global index 0 or Nc-1); the
 * fine plane of coarse local plane ic is 2*(gc.ig0+ic) - gf.ig0. */
__global__ void __launch_bounds__(256) restrict_kernel(Geom gf, const double *__restrict__ r, Geom gc,
                                                       double *__restrict__ dc, int ic_lo, int ic_hi, int faces_only)
{
    const int kc = blockIdx.x * 64 + threadIdx.x;
    const int jc = blockIdx.y * 4 + threadIdx.y;
    const int ic = ic_lo + blockIdx.z;
    if (kc >= gc.nk || jc >= gc.nj || ic >= ic_hi)
        return;
    const int icg = gc.ig0 + ic;
    const int fi = 2 * icg - gf.ig0, fj = 2 * jc, fk = 2 * kc;
    const long long pf = gidx(gf, fi, fj, fk);
    const bool face = icg == 0 || icg == gc.N - 1 || jc == 0 || jc == gc.nj - 1 || kc == 0 || kc == gc.nk - 1;
    double val;
    if (face) {
        val = r[pf];
    } else if (faces_only) {
        return;
    } else {
        val = 0.;
#pragma unroll
        for (int ti = -1; ti <= 1; ti++)
#pragma unroll
            for (int tj = -1; tj <= 1; tj++)
#pragma unroll
                for (int tk = -1; tk <= 1; tk++) {
                    const double w = (ti ? 0.25 : 0.5) * (tj ? 0.25 : 0.5) * (tk ? 0.25 : 0.5);
                    val += r[pf + ti * gf.plane + tj * (long long)gf.pitch + tk] * w;
                }
    }
    dc[gidx(gc, ic, jc, kc)] = val;
}

/* injection on the six coarse faces only (mg_3d.h:879-958): one thread per face point.
 * blockIdx.z = face: 0/1 the physical i-faces, 2/3 the j-faces, 4/5 the k-faces */
__global__ void __launch_bounds__(256) restrict_faces_kernel(Geom gf, const double *__restrict__ r, Geom gc,
                                                             double *__restrict__ dc, int ic_lo, int ic_hi)
{
    const int b = blockIdx.x * 64 + threadIdx.x, a = blockIdx.y * 4 + threadIdx.y, f = blockIdx.z;
    int ic, jc, kc;
    if (f < 2) {
        const int icg = f == 0 ? 0 : gc.N - 1;
        ic = icg - gc.ig0;
        jc = a;
        kc = b;
        if (ic < ic_lo || ic >= ic_hi || jc >= gc.nj || kc >= gc.nk)
            return;
    } else if (f < 4) {
        ic = ic_lo + a;
        jc = f == 2 ? 0 : gc.nj - 1;
        kc = b;
        if (ic >= ic_hi || kc >= gc.nk)
            return;
    } else {
        ic = ic_lo + a;
        jc = b;
        kc = f == 4 ? 0 : gc.nk - 1;
        if (ic >= ic_hi || jc >= gc.nj)
            return;
    }
    const int fi = 2 * (gc.ig0 + ic) - gf.ig0;
    dc[gidx(gc, ic, jc, kc)] = r[gidx(gf, fi, 2 * jc, 2 * kc)];
}

void k_restrict(const Geom &gf, const double *r, const Geom &gc, double *dc, hipStream_t s, int ic_lo, int ic_hi,
                bool faces_only)
{
    if (faces_only) {
        const int lo = ic_lo >= 0 ? ic_lo : ((gc.ig0 == 0) ? 0 : 1);
        const int hi = ic_hi >= 0 ? ic_hi : ((gc.ig0 + gc.ni == gc.N) ? gc.ni : gc.ni - 1);
        if (hi <= lo)
            return;
        const int m = max(max(gc.nj, gc.nk), hi - lo);
        dim3 grid((m + 63) / 64, (m + 3) / 4, 6);
        hipLaunchKernelGGL(restrict_faces_kernel, grid, dim3(64, 4, 1), 0, s, gf, r, gc, dc, lo, hi);
        return;
    }

    /* local coarse planes written: physical boundary planes and owned planes; a
     * halo plane (local 0 / ni-1 that is not a physical boundary) is the neighbour's */
    const int lo = ic_lo >= 0 ? ic_lo : ((gc.ig0 == 0) ? 0 : 1);
    const int hi = ic_hi >= 0 ? ic_hi : ((gc.ig0 + gc.ni == gc.N) ? gc.ni : gc.ni - 1);
    if (hi <= lo)
        return;
    dim3 grid((gc.nk + 63) / 64, (gc.nj + 3) / 4, hi - lo);
    hipLaunchKernelGGL(restrict_kernel, grid, dim3(64, 4, 1), 0, s, gf, r, gc, dc, lo, hi, faces_only ? 1 : 0);
}

/* -------------------------------------------------------------- prolongation
 * prolongateAndCorrectError, mg_3d.h:1000-1145: ef[p] += P(ec) at EVERY fine point.
 * Parent order per parity class (o = odd flags of i,j,k; l = low coarse index):
 *   o=(1,1,1): ((((((c000+c001)+c010)+c011)+c100)+c101)+c110)+c111, *0.125   (:1028-1048)
 *   i even   : ((c(jl,kl)+c(jl+1,kl))+c(jl,kl+1))+c(jl+1,kl+1),       *0.25    (:1064-1067)
 *   j even   : ((c(il,kl)+c(il+1,kl))+c(il,kl+1))+c(il+1,kl+1),       *0.25    (:1075-1078)
 *   k even   : ((c(il,jl)+c(il,jl+1))+c(il+1,jl))+c(il+1,jl+1),       *0.25    (:1085-1088)
 *   one odd  : (low + high) * 0.5                                               (:1110-1133)
 *   none odd : copy                                                             (:1138)
 * The running sum starts from 0. (0. + x == x), as retVal does. */
__global__ void __launch_bounds__(256) prolong_kernel(Geom gc, const double *__restrict__ ec, Geom gf,
                                                      double *__restrict__ ef, int if_lo, int if_hi)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int i = if_lo + blockIdx.z;
    if (k >= gf.nk || j >= gf.nj || i >= if_hi)
        return;
    const int ig = gf.ig0 + i;
    const int oi = ig & 1, oj = j & 1, ok = k & 1;
    const int il = (ig - oi) / 2 - gc.ig0, jl = (j - oj) / 2, kl = (k - ok) / 2;
    const long long c0 = gidx(gc, il, jl, kl);
    const long long sI = gc.plane, sJ = gc.pitch, sK = 1;
    double t = 0.;
    switch (oi + oj + ok) {
    case 3:
        t += ec[c0];
        t += ec[c0 + sK];
        t += ec[c0 + sJ];
        t += ec[c0 + sJ + sK];
        t += ec[c0 + sI];
        t += ec[c0 + sI + sK];
        t += ec[c0 + sI + sJ];
        t += ec[c0 + sI + sJ + sK];
        t *= 0.125;
        break;
    case 2:
        if (!oi) {
            t += ec[c0];
            t += ec[c0 + sJ];
            t += ec[c0 + sK];
            t += ec[c0 + sJ + sK];
        } else if (!oj) {
            t += ec[c0];
            t += ec[c0 + sI];
            t += ec[c0 + sK];
            t += ec[c0 + sI + sK];
        } else {
            t += ec[c0];
            t += ec[c0 + sJ];
            t += ec[c0 + sI];
            t += ec[c0 + sI + sJ];
        }
        t *= 0.25;
        break;
    case 1:
        t += ec[c0];
        t += ec[c0 + oi * sI + oj * sJ + ok * sK];
        t *= 0.5;
        break;
    default:
        t = ec[c0];
    }
    const long long p = gidx(gf, i, j, k);
    ef[p] += t;
}

/* Cell-based form used by the V-cycle: a thread owns one coarse cell (jc, m) -- the 2 x 2 fine points
 * (2jc, 2jc+1) x (2m, 2m+1) of every fine plane -- and marches along i.  The eight coarse corners
 * E[a][b][c] = ec(il+a, jc+b, m+c) of the current cell stay in registers and are reused by the two fine
 * planes that share them; fine data moves as 16-byte k-pairs.  Same parent order as prolong_kernel. */
__global__ void __launch_bounds__(256) prolong_cell_kernel(Geom gc, const double *__restrict__ ec, Geom gf,
                                                           double *__restrict__ ef, int if_lo, int if_hi, int chunk)
{
    const int m = blockIdx.x * 64 + threadIdx.x;
    const int jc = blockIdx.y * 4 + threadIdx.y;
    const int k0 = 2 * m, j0 = 2 * jc;
    if (k0 >= gf.nk || j0 >= gf.nj)
        return;
    const int i_beg = if_lo + blockIdx.z * chunk;
    const int i_end = min(if_hi, i_beg + chunk);
    const bool row1 = j0 + 1 < gf.nj;
    const int m1 = min(m + 1, gc.nk - 1), jc1 = min(jc + 1, gc.nj - 1);
    const long long c00 = (long long)gc.pitch * jc + m, c01 = (long long)gc.pitch * jc + m1;
    const long long c10 = (long long)gc.pitch * jc1 + m, c11 = (long long)gc.pitch * jc1 + m1;
    double E0[2][2], E1[2][2]; /* coarse planes `have` and `have + 1`, [b][c] */
    int have = -0x40000000;
    auto load = [&](int il, double(&E)[2][2]) {
        const double *pl = ec + gc.plane * min(max(il, 0), gc.ni - 1);
        E[0][0] = pl[c00];
        E[0][1] = pl[c01];
        E[1][0] = pl[c10];
        E[1][1] = pl[c11];
    };
    for (int i = i_beg; i < i_end; i++) {
        const int ig = gf.ig0 + i, oi = ig & 1;
        const int il = (ig - oi) / 2 - gc.ig0;
        if (have != il) {
            if (have + 1 == il) {
                E0[0][0] = E1[0][0];
                E0[0][1] = E1[0][1];
                E0[1][0] = E1[1][0];
                E0[1][1] = E1[1][1];
            } else {
                load(il, E0);
            }
            load(il + 1, E1);
            have = il;
        }
        double t00, t01, t10, t11; /* [row parity][col parity] */
        if (!oi) {
            t00 = E0[0][0];
            t01 = (E0[0][0] + E0[0][1]) * 0.5;
            t10 = (E0[0][0] + E0[1][0]) * 0.5;
            t11 = (((E0[0][0] + E0[1][0]) + E0[0][1]) + E0[1][1]) * 0.25; /* i even: (jl,kl)(jl+1,kl)(jl,kl+1)(jl+1,kl+1) */
        } else {
            t00 = (E0[0][0] + E1[0][0]) * 0.5;
            t01 = (((E0[0][0] + E1[0][0]) + E0[0][1]) + E1[0][1]) * 0.25; /* j even: (il,kl)(il+1,kl)(il,kl+1)(il+1,kl+1) */
            t10 = (((E0[0][0] + E0[1][0]) + E1[0][0]) + E1[1][0]) * 0.25; /* k even: (il,jl)(il,jl+1)(il+1,jl)(il+1,jl+1) */
            double t = E0[0][0] + E0[0][1];
            t = t + E0[1][0];
            t = t + E0[1][1];
            t = t + E1[0][0];
            t = t + E1[0][1];
            t = t + E1[1][0];
            t = t + E1[1][1];
            t11 = t * 0.125;
        }
        double *row = ef + gf.plane * i + (long long)gf.pitch * j0 + k0;
        double2 a = *reinterpret_cast<double2 *>(row);
        a.x += t00;
        a.y += t01;
        *reinterpret_cast<double2 *>(row) = a;
        if (row1) {
            double2 b = *reinterpret_cast<double2 *>(row + gf.pitch);
            b.x += t10;
            b.y += t11;
            *reinterpret_cast<double2 *>(row + gf.pitch) = b;
        }
    }
}

void k_prolong(const Geom &gc, const double *ec, const Geom &gf, double *ef, hipStream_t s, int if_lo, int if_hi)
{
    const int lo = if_lo >= 0 ? if_lo : ((gf.ig0 == 0) ? 0 : 1);
    const int hi = if_hi >= 0 ? if_hi : ((gf.ig0 + gf.ni == gf.N) ? gf.ni : gf.ni - 1);
    if (hi <= lo)
        return;
    if (gf.nk == 2 * gc.nk - 1 && gf.nj == 2 * gc.nj - 1) {
        const int gx = ((gf.nk + 1) / 2 + 63) / 64, gy = ((gf.nj + 1) / 2 + 3) / 4;
        int chunk = 64; /* a few hundred to a few thousand blocks, like the sweep */
        while (chunk > 4 && (long long)gx * gy * ((hi - lo + chunk - 1) / chunk) < 2048)
            chunk /= 2;
        dim3 grid(gx, gy, (hi - lo + chunk - 1) / chunk);
        hipLaunchKernelGGL(prolong_cell_kernel, grid, dim3(64, 4, 1), 0, s, gc, ec, gf, ef, lo, hi, chunk);
        return;
    }
    dim3 grid((gf.nk + 63) / 64, (gf.nj + 3) / 4, hi - lo);
    hipLaunchKernelGGL(prolong_kernel, grid, dim3(64, 4, 1), 0, s, gc, ec, gf, ef, lo, hi);
}

/* ------------------------------------------------------- coarsest direct solve
 * solveWithLU, gauss_elim.h:31-60, on the banded factor.
 *   forward : z[i] = b[i] - sum_{j<i, ascending}  LU[i][j]*z[j]
 *   backward: x[i] = (z[i] - sum_{j>i, descending} LU[i][j]*x[j]) / LU[i][i]
 * Column-oriented: at step j the finished x[j] is broadcast and every row
 * inside the band adds its product.  Each row's sum therefore receives its
 * terms in exactly the reference's order (j ascending / descending), so the
 * result is bit-identical; only exact-zero factors outside the band are
 * skipped (adding +-0 never changes a sum that started from +0).
 *
 * One wave: lane l owns rows == l (mod 64); with bw <= 64*R a lane has at most
 * R rows in flight, their running sums live in registers, x[j] travels by
 * v_readlane.  No LDS traffic on the dependency chain, no barriers. */
#include "mg3d_lu_dev.h"

template <int R>
__global__ void __launch_bounds__(64) lu_solve_wave_kernel(LuBand lu, Geom g0, const double *__restrict__ b_pad,
                                                           double *__restrict__ x_pad)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x, n = lu.n;
    double *b = lds, *z = lds + n, *dg = lds + 2 * n; /* x overwrites b; dg = diagonal, then its reciprocals */
    const int NN = g0.nj * g0.nk;
    auto pad_of = [&](int p) -> long long {
        const int i = p / NN, rem = p - i * NN;
        const int j = rem / g0.nk, k = rem - j * g0.nk;
        return g0.plane * i + (long long)g0.pitch * j + k;
    };
    for (int p = lane; p < n; p += WAVE) {
        b[p] = b_pad[pad_of(p)];
        dg[p] = lu.diag[p];
        dg[n + p] = lu.diag[lu.npad + p];
    }
    __syncthreads();
    lu_wave_pass<R, true, false>(lu, lane, b, z, dg);
    __syncthreads();
    if (lu.fast_div)
        lu_wave_pass<R, false, true>(lu, lane, z, b, dg);
    else
        lu_wave_pass<R, false, false>(lu, lane, z, b, dg);
    __syncthreads();
    for (int p = lane; p < n; p += WAVE)
        x_pad[pad_of(p)] = b[p];
}

/* RI > 0: a reduced factor `lin` (the system without its identity rows, R = RI) exists beside the full one.  It is taken
 * when every identity row's right-hand side entry is +-0 -- then those unknowns are x = b and contribute +-0 to every
 * other row's running sums, which changes none of them (install_lu, mg3d_ctx.hip) -- else the full system. */
template <int R, int RI>
__global__ void __launch_bounds__(192) lu_solve_stream_kernel(LuBand lu, LuBand lin, Geom g0, const double *__restrict__ b_pad,
                                                             double *__restrict__ x_pad)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int CD = 64 * 64 * R; /* doubles per chunk of the full system's ring (the reduced one's is not larger) */
    const int tid = threadIdx.x, lane = tid & 63, n = lu.n, npad = lu.npad;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *ring = lds, *b = lds + 2 * CD, *z = b + npad, *dg = z + npad; /* x overwrites b; dg: diagonal, reciprocals */
    const int NN = g0.nj * g0.nk;
    auto pad_of = [&](int p) -> long long {
        const int i = p / NN, rem = p - i * NN;
        const int j = rem / g0.nk, k = rem - j * g0.nk;
        return g0.plane * i + (long long)g0.pitch * j + k;
    };
    int nonzero = 0;
    for (int p = tid; p < npad; p += 192) {
        const double v = p < n ? b_pad[pad_of(p)] : 0.;
        b[p] = v;
        if (RI > 0 && p < n)
            nonzero |= lu.in_map[p] < 0 && (__double_as_longlong(v) << 1) != 0ll;
    }
    if constexpr (RI > 0) {
        if (!__syncthreads_or(nonzero)) { /* workgroup-uniform */
            const int ni = lin.n, npi = lin.npad; /* 2 * npi <= npad (install_lu): both vectors fit into z's place */
            double *bi = z, *zi = z + npi;
            for (int p = tid; p < npi; p += 192) {
                bi[p] = 0.;
                dg[p] = lin.diag[p];
                dg[npi + p] = lin.diag[npi + p];
            }
            __syncthreads();
            for (int p = tid; p < n; p += 192) {
                const int q = lu.in_map[p];
                if (q >= 0)
                    bi[q] = b[p];
            }
            (void)ni;
            lu_stream_solve<RI>(lin, ring, bi, zi, dg, lane, wave);
            for (int p = tid; p < n; p += 192) {
                const int q = lu.in_map[p];
                x_pad[pad_of(p)] = q >= 0 ? bi[q] : b[p]; /* identity row: x = (b - (+0)) / 1 = b, the sign of a zero kept */
            }
            return;
        }
    }
    for (int p = tid; p < npad; p += 192) {
        dg[p] = lu.diag[p];
        dg[npad + p] = lu.diag[npad + p];
    }
    lu_stream_solve<R>(lu, ring, b, z, dg, lane, wave);
    for (int p = tid; p < n; p += 192)
        x_pad[pad_of(p)] = b[p];
}

static size_t lu_stream_lds(int npad, int R) { return sizeof(double) * (2 * (size_t)64 * 64 * R + 4 * (size_t)npad); }

int mg3d_lu_stream_chunk(int n, int R)
{
    if (R < 1 || R > 2)
        return 0;
    int dev = 0, max_lds = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess)
        return 0;
    return lu_stream_lds((n + 63) / 64 * 64, R) <= (size_t)max_lds ? 64 : 0;
}

template <int R, int RI>
static bool launch_lu_stream(const LuBand &lu, const LuBand &lin, const Geom &g0, const double *b_pad, double *x_pad,
                             hipStream_t s)
{
    const size_t lds = lu_stream_lds(lu.npad, R);
    /* a slot ring beyond 64 KB has to be asked for, per kernel and per device: exactly what is needed (the kernel's few
     * static bytes -- the workgroup vote -- count against the same 160 KB) */
    if (lds > 65536) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        static std::mutex mu;
        static std::map<int, size_t> granted; /* device -> bytes granted; 0 = refused */
        std::lock_guard<std::mutex> lock(mu);
        auto it = granted.find(dev);
        if (it == granted.end() || (it->second != 0 && it->second < lds)) {
            const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&lu_solve_stream_kernel<R, RI>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
            (void)hipGetLastError();
            granted[dev] = ok ? lds : 0;
            it = granted.find(dev);
        }
        if (it->second == 0)
            return false;
    }
    hipLaunchKernelGGL((lu_solve_stream_kernel<R, RI>), dim3(1), dim3(192), lds, s, lu, lin, g0, b_pad, x_pad);
    return true;
}

/* Generic fallback for wide bands (coarse grids beyond 9^3): one 1024-thread
 * block, running sums of the rows in flight kept in a global ring `acc[n]`,
 * one barrier per step.  Same per-row term order as above. */
__global__ void __launch_bounds__(1024) lu_solve_block_kernel(LuBand lu, Geom g0, const double *__restrict__ b_pad,
                                                              double *__restrict__ x_pad, double *__restrict__ z,
                                                              double *__restrict__ acc)
{
    const int n = lu.n, bw = lu.bw, tid = threadIdx.x;
    const int NN = g0.nj * g0.nk;
    auto pad_of = [&](int p) -> long long {
        const int i = p / NN, rem = p - i * NN;
        const int j = rem / g0.nk, k = rem - j * g0.nk;
        return g0.plane * i + (long long)g0.pitch * j + k;
    };
    __shared__ double bc;
    for (int i = tid; i < n; i += 1024)
        acc[i] = 0.;
    __syncthreads();
    for (int j = 0; j < n; j++) {
        if (tid == 0) {
            bc = b_pad[pad_of(j)] - acc[j];
            z[j] = bc;
        }
        __syncthreads();
        const double zj = bc;
        for (int t = tid; t < bw && j + 1 + t < n; t += 1024)
            acc[j + 1 + t] += lu.lcol[(long long)j * bw + t] * zj;
        __syncthreads();
    }
    for (int i = tid; i < n; i += 1024)
        acc[i] = 0.;
    __syncthreads();
    for (int j = n - 1; j >= 0; j--) {
        if (tid == 0) {
            bc = (z[j] - acc[j]) / lu.diag[j];
            x_pad[pad_of(j)] = bc;
        }
        __syncthreads();
        const double xj = bc;
        for (int t = tid; t < bw && j - 1 - t >= 0; t += 1024)
            acc[j - 1 - t] += lu.ucol[(long long)j * bw + t] * xj;
        __syncthreads();
    }
}

void k_lu_solve(const LuBand &lu, const LuBand &lu_in, const Geom &g0, const double *b_pad, double *x_pad, double *work,
                hipStream_t s)
{
    double *z = work, *acc = work + lu.n;
    const size_t lds = sizeof(double) * 4 * (size_t)lu.n;
    /* the reduced factor rides along when it exists (narrower band: one row per lane) */
    const bool red = lu.in_map && lu_in.n > 0 && lu_in.stream_ch == 64 && lu_in.rot_r == 1 && 2 * lu_in.npad <= lu.npad;
    if (lu.stream_ch == 64 && lu.rot_r == 2 && red && launch_lu_stream<2, 1>(lu, lu_in, g0, b_pad, x_pad, s))
        return;
    if (lu.stream_ch == 64 && lu.rot_r == 2 && launch_lu_stream<2, 0>(lu, lu, g0, b_pad, x_pad, s))
        return;
    if (lu.stream_ch == 64 && lu.rot_r == 1 && red && launch_lu_stream<1, 1>(lu, lu_in, g0, b_pad, x_pad, s))
        return;
    if (lu.stream_ch == 64 && lu.rot_r == 1 && launch_lu_stream<1, 0>(lu, lu, g0, b_pad, x_pad, s))
        return;
    if (lu.rot_r == 1)
        hipLaunchKernelGGL(lu_solve_wave_kernel<1>, dim3(1), dim3(64), lds, s, lu, g0, b_pad, x_pad);
    else if (lu.rot_r == 2)
        hipLaunchKernelGGL(lu_solve_wave_kernel<2>, dim3(1), dim3(64), lds, s, lu, g0, b_pad, x_pad);
    else
        hipLaunchKernelGGL(lu_solve_block_kernel, dim3(1), dim3(1024), 0, s, lu, g0, b_pad, x_pad, z, acc);
}
