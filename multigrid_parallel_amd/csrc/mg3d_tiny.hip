/*
 * mg3d_tiny.hip -- the level above the coarsest one (17^3 for the usual 9^3 coarse grid) in ONE workgroup.
 *
 * A visit of that level by the V-cycle (mg_3d.h:1242-1362) is eight dependent colour passes, a residual, a restriction
 * and a prolongation on 4913 points: as launches of the plane-marching sweep it costs five launches of 8-20 us each
 * (pipeline fill and drain, not work).  The whole level fits the LDS of one CU three times over (u, d, r: 39 KB each),
 * so two launches do it: `tiny_down` (zero guess, pre-smoothing, residual, full-weighting restriction into the coarsest
 * right-hand side) in front of the direct solve and `tiny_up` (prolongation, post-smoothing) behind it, every colour
 * pass one LDS sweep and one workgroup barrier.  Every expression is the one of the reference with its association
 * (update mg_3d.h:438-443, residual :819-821, restriction :961-995 with the ti, tj, tk order, face injection :879-958,
 * prolongation :1000-1145 with the parent order per parity class), no contraction: bit-identical to the generic
 * kernels (tests/test_gpu_parity.py compares both paths).
 */
#include "mg3d_internal.h"
#include "mg3d_lu_dev.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

#define TINY_MAX_N 17
#define TINY_THREADS 1024

__device__ __forceinline__ int lidx(int N, int i, int j, int k) { return (i * N + j) * N + k; }

/* one colour pass over the LDS copy (mg_3d.h:438-443, 658-702) */
__device__ __forceinline__ void tiny_pass(double *u, const double *d, int N, double hSq, double sixth, int color)
{
    const int M = N - 2, n = M * M * M;
    for (int t = threadIdx.x; t < n; t += TINY_THREADS) {
        const int i = 1 + t / (M * M), j = 1 + (t / M) % M, k = 1 + t % M;
        if (((i + j + k) & 1) != color)
            continue;
        const int p = lidx(N, i, j, k);
        double s = u[p - N * N] + u[p + N * N];
        s = s + u[p - N];
        s = s + u[p + N];
        s = s + u[p - 1];
        s = s + u[p + 1];
        s = s - hSq * d[p];
        u[p] = sixth * s;
    }
    __syncthreads();
}

/* zero guess (mg_3d.h:1258-1259), `iters` x (red, black) (:1282), residual (:1294), restriction into dc (:1310) */
__global__ void __launch_bounds__(TINY_THREADS) tiny_down_kernel(Geom g, double *__restrict__ u_out,
                                                                const double *__restrict__ d_in,
                                                                const double *__restrict__ r_in, Geom gc,
                                                                double *__restrict__ dc, double hSq, double sixth,
                                                                double invHsq, int iters)
{
    extern __shared__ double lds[];
    const int N = g.N, n = N * N * N, Nc = gc.N;
    double *u = lds, *d = lds + n, *r = lds + 2 * n;
    for (int t = threadIdx.x; t < n; t += TINY_THREADS) {
        const int i = t / (N * N), j = (t / N) % N, k = t % N;
        u[t] = 0.;
        d[t] = d_in[g.plane * i + (long long)g.pitch * j + k];
    }
    __syncthreads();
    for (int s = 0; s < iters; s++) {
        tiny_pass(u, d, N, hSq, sixth, 1);
        tiny_pass(u, d, N, hSq, sixth, 0);
    }
    /* residual on the interior (mg_3d.h:819-821); the boundary entries of r are whatever the level's r array holds
     * (nothing ever writes them, :824-825) -- only the face injection below reads them, straight from memory */
    {
        const int M = N - 2, m = M * M * M;
        for (int t = threadIdx.x; t < m; t += TINY_THREADS) {
            const int i = 1 + t / (M * M), j = 1 + (t / M) % M, k = 1 + t % M;
            const int p = lidx(N, i, j, k);
            double s = u[p - N * N] + u[p + N * N];
            s = s + u[p - N];
            s = s + u[p + N];
            s = s + u[p - 1];
            s = s + u[p + 1];
            s = s - 6 * u[p];
            r[p] = d[p] - invHsq * s;
        }
    }
    __syncthreads();
    /* restriction (mg_3d.h:844-998): 27-point sum in ti, tj, tk order from 0 on the interior, injection on the faces */
    for (int t = threadIdx.x; t < Nc * Nc * Nc; t += TINY_THREADS) {
        const int ic = t / (Nc * Nc), jc = (t / Nc) % Nc, kc = t % Nc;
        const bool face = ic == 0 || ic == Nc - 1 || jc == 0 || jc == Nc - 1 || kc == 0 || kc == Nc - 1;
        double val;
        if (face) {
            val = r_in[g.plane * (2 * ic) + (long long)g.pitch * (2 * jc) + 2 * kc];
        } else {
            val = 0.;
            const int pf = lidx(N, 2 * ic, 2 * jc, 2 * kc);
#pragma unroll
            for (int ti = -1; ti <= 1; ti++)
#pragma unroll
                for (int tj = -1; tj <= 1; tj++)
#pragma unroll
                    for (int tk = -1; tk <= 1; tk++) {
                        const double w = (ti ? 0.25 : 0.5) * (tj ? 0.25 : 0.5) * (tk ? 0.25 : 0.5);
                        val += r[pf + ti * N * N + tj * N + tk] * w;
                    }
        }
        dc[gc.plane * ic + (long long)gc.pitch * jc + kc] = val;
    }
    for (int t = threadIdx.x; t < n; t += TINY_THREADS) {
        const int i = t / (N * N), j = (t / N) % N, k = t % N;
        u_out[g.plane * i + (long long)g.pitch * j + k] = u[t];
    }
}

/* prolongation + correction at every fine point (mg_3d.h:1331 -> :1000-1145), `iters` x (black, red) (:1341) */
__global__ void __launch_bounds__(TINY_THREADS) tiny_up_kernel(Geom g, double *__restrict__ u_io,
                                                              const double *__restrict__ d_in, Geom gc,
                                                              const double *__restrict__ ec_in, double hSq, double sixth,
                                                              int iters)
{
    extern __shared__ double lds[];
    const int N = g.N, n = N * N * N, Nc = gc.N, nc = Nc * Nc * Nc;
    double *u = lds, *d = lds + n, *ec = lds + 2 * n;
    for (int t = threadIdx.x; t < n; t += TINY_THREADS) {
        const int i = t / (N * N), j = (t / N) % N, k = t % N;
        const long long q = g.plane * i + (long long)g.pitch * j + k;
        u[t] = u_io[q];
        d[t] = d_in[q];
    }
    for (int t = threadIdx.x; t < nc; t += TINY_THREADS) {
        const int i = t / (Nc * Nc), j = (t / Nc) % Nc, k = t % Nc;
        ec[t] = ec_in[gc.plane * i + (long long)gc.pitch * j + k];
    }
    __syncthreads();
    const int sI = Nc * Nc, sJ = Nc, sK = 1;
    for (int t = threadIdx.x; t < n; t += TINY_THREADS) {
        const int i = t / (N * N), j = (t / N) % N, k = t % N;
        const int oi = i & 1, oj = j & 1, ok = k & 1;
        const int c0 = (((i - oi) / 2) * Nc + (j - oj) / 2) * Nc + (k - ok) / 2;
        double x = 0.;
        switch (oi + oj + ok) {
        case 3:
            x += ec[c0];
            x += ec[c0 + sK];
            x += ec[c0 + sJ];
            x += ec[c0 + sJ + sK];
            x += ec[c0 + sI];
            x += ec[c0 + sI + sK];
            x += ec[c0 + sI + sJ];
            x += ec[c0 + sI + sJ + sK];
            x *= 0.125;
            break;
        case 2:
            if (!oi) {
                x += ec[c0];
                x += ec[c0 + sJ];
                x += ec[c0 + sK];
                x += ec[c0 + sJ + sK];
            } else if (!oj) {
                x += ec[c0];
                x += ec[c0 + sI];
                x += ec[c0 + sK];
                x += ec[c0 + sI + sK];
            } else {
                x += ec[c0];
                x += ec[c0 + sJ];
                x += ec[c0 + sI];
                x += ec[c0 + sI + sJ];
            }
            x *= 0.25;
            break;
        case 1:
            x += ec[c0];
            x += ec[c0 + oi * sI + oj * sJ + ok * sK];
            x *= 0.5;
            break;
        default:
            x = ec[c0];
        }
        u[t] += x;
    }
    __syncthreads();
    for (int s = 0; s < iters; s++) {
        tiny_pass(u, d, N, hSq, sixth, 0);
        tiny_pass(u, d, N, hSq, sixth, 1);
    }
    for (int t = threadIdx.x; t < n; t += TINY_THREADS) {
        const int i = t / (N * N), j = (t / N) % N, k = t % N;
        u_io[g.plane * i + (long long)g.pitch * j + k] = u[t];
    }
}

/* ---- the whole bottom of the cycle in ONE workgroup: tiny_down, the direct solve of the coarsest level and tiny_up
 * (mg_3d.h:1258-1310 on level 1, :1262-1277 on level 0, :1331-1341 on level 1).  Three dependent launches and a memset
 * (19 + 5 + 33 + 20 us in the kernel trace, most of it launch and first-touch latency) become one: u and d of level 1
 * stay in LDS across the solve, the coarse right-hand side never leaves the CU, the factors are streamed through the
 * LDS ring of lu_stream_solve (mg3d_lu_dev.h) by two of the workgroup's waves exactly as in the stand-alone solve.
 * The solve is the REDUCED one (install_lu, mg3d_ctx.hip: the factor without its identity rows) -- valid because the
 * faces of the coarse right-hand side are the injected faces of r, zeros unless somebody wrote r from outside the cycle;
 * if one of them is not a zero the full system is solved by the single-wave substitution with the factors read from
 * global memory (slow, correct, never seen in a V-cycle).  Every expression and order is the one of the three kernels it
 * replaces (tests compare both routes bit for bit: MG3D_NO_TINY_CYCLE=1).
 * A thread keeps the LDS indices and colours of the (at most TINY_PTS) interior points it owns in registers: the integer
 * divisions that decode them are paid once, not once per colour pass. */
#define TINY_CYC_THREADS 512
#define TINY_PTS 7 /* ceil(15^3 / 512) */

/* phase stamps of the last tiny_cycle_kernel launch (constant 100 MHz clock), written by thread 0: where the launch's
 * time goes (tools/tiny_phases.py through mg3d_debug_tiny_stamps) */
__device__ long long g_tiny_stamps[16];
#define TINY_STAMP(k)                                 \
    do {                                              \
        if (tid == 0)                                 \
            g_tiny_stamps[k] = (long long)wall_clock64(); \
    } while (0)

template <int RF>
__global__ void __launch_bounds__(TINY_CYC_THREADS) tiny_cycle_kernel(Geom g, double *__restrict__ u_io,
                                                                    const double *__restrict__ d_in,
                                                                    const double *__restrict__ r_in, Geom gc,
                                                                    double *__restrict__ dc, double *__restrict__ xc,
                                                                    LuBand lu, LuBand lin, double hSq, double sixth,
                                                                    double invHsq, int iters, int s_doubles)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int N = g.N, n = N * N * N, Nc = gc.N, nc = Nc * Nc * Nc, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *u = lds, *d = lds + n, *S = lds + 2 * n, *ec = S + s_doubles;
    double *r = S, *bfull = S + n; /* bfull: lu.npad entries */
    /* the interior points this thread owns */
    const int M = N - 2, m = M * M * M;
    int pt[TINY_PTS], col[TINY_PTS];
#pragma unroll
    for (int q = 0; q < TINY_PTS; q++) {
        const int t = tid + q * TINY_CYC_THREADS;
        if (t < m) {
            const int i = 1 + t / (M * M), j = 1 + (t / M) % M, k = 1 + t % M;
            pt[q] = lidx(N, i, j, k);
            col[q] = (i + j + k) & 1;
        } else {
            pt[q] = -1;
            col[q] = -1;
        }
    }
    auto pass = [&](int colour) { /* mg_3d.h:438-443, 658-702 */
#pragma unroll
        for (int q = 0; q < TINY_PTS; q++)
            if (col[q] == colour) {
                const int p = pt[q];
                double s = u[p - N * N] + u[p + N * N];
                s = s + u[p - N];
                s = s + u[p + N];
                s = s + u[p - 1];
                s = s + u[p + 1];
                s = s - hSq * d[p];
                u[p] = sixth * s;
            }
        __syncthreads();
    };
    /* ---- down (tiny_down_kernel): zero guess, pre-smoothing, residual, restriction */
    /* whole-field loops: a thread takes the points tid, tid + 512, ...; their (i, j, k) are decoded ONCE by division and
     * then advanced by carries -- a division by a run-time value costs ~35 instructions, and decoding every point in every
     * loop by three of them was a third of this kernel's time */
    TINY_STAMP(0);
    const int di = TINY_CYC_THREADS / (N * N), dj = (TINY_CYC_THREADS / N) % N, dk = TINY_CYC_THREADS % N;
    const int i0 = tid / (N * N), j0 = (tid / N) % N, k0 = tid % N;
    auto advance = [&](int &i, int &j, int &k) {
        k += dk;
        if (k >= N) {
            k -= N;
            j++;
        }
        j += dj;
        if (j >= N) {
            j -= N;
            i++;
        }
        i += di;
    };
    {
        int i = i0, j = j0, k = k0;
        for (int t = tid; t < n; t += TINY_CYC_THREADS) {
            u[t] = 0.;
            d[t] = d_in[g.plane * i + (long long)g.pitch * j + k];
            advance(i, j, k);
        }
    }
    __syncthreads();
    TINY_STAMP(1);
    for (int s = 0; s < iters; s++) {
        pass(1);
        pass(0);
    }
    TINY_STAMP(2);
#pragma unroll
    for (int q = 0; q < TINY_PTS; q++)
        if (pt[q] >= 0) { /* mg_3d.h:819-821 */
            const int p = pt[q];
            double s = u[p - N * N] + u[p + N * N];
            s = s + u[p - N];
            s = s + u[p + N];
            s = s + u[p - 1];
            s = s + u[p + 1];
            s = s - 6 * u[p];
            r[p] = d[p] - invHsq * s;
        }
    __syncthreads();
    TINY_STAMP(3);
    int nonzero = 0;
    for (int t = nc + tid; t < lu.npad; t += TINY_CYC_THREADS)
        bfull[t] = 0.; /* the padding rows of the factor */
    for (int t = tid; t < nc; t += TINY_CYC_THREADS) { /* two rounds: the divisions are cheap here */
        const int ic = t / (Nc * Nc), jc = (t / Nc) % Nc, kc = t % Nc;
        const bool face = ic == 0 || ic == Nc - 1 || jc == 0 || jc == Nc - 1 || kc == 0 || kc == Nc - 1;
        double val = 0.;
        if (face) { /* injection, mg_3d.h:879-958, from r's boundary entries as memory holds them */
            val = r_in[g.plane * (2 * ic) + (long long)g.pitch * (2 * jc) + 2 * kc];
            nonzero |= (__double_as_longlong(val) << 1) != 0ll;
            ec[t] = val; /* an identity row of the coarse operator: x = (b - (+0)) / 1 = b */
        } else {
            const int pf = lidx(N, 2 * ic, 2 * jc, 2 * kc);
#pragma unroll
            for (int ti = -1; ti <= 1; ti++)
#pragma unroll
                for (int tj = -1; tj <= 1; tj++)
#pragma unroll
                    for (int tk = -1; tk <= 1; tk++) {
                        const double w = (ti ? 0.25 : 0.5) * (tj ? 0.25 : 0.5) * (tk ? 0.25 : 0.5);
                        val += r[pf + ti * N * N + tj * N + tk] * w;
                    }
        }
        dc[gc.plane * ic + (long long)gc.pitch * jc + kc] = val;
        bfull[t] = val;
    }
    const int bad = __syncthreads_or(nonzero); /* workgroup-uniform; also: bfull, ec faces are complete */
    TINY_STAMP(4);
    /* ---- the direct solve (gauss_elim.h:31-60) */
    if (!bad) {
        const int npi = lin.npad;
        double *ring = S, *bi = S + 2 * 64 * 64, *zi = bi + npi, *dg = zi + npi;
        for (int p = tid; p < npi; p += TINY_CYC_THREADS) {
            bi[p] = 0.;
            dg[p] = lin.diag[p];
            dg[npi + p] = lin.diag[npi + p];
        }
        __syncthreads();
        for (int p = tid; p < nc; p += TINY_CYC_THREADS) {
            const int q = lu.in_map[p];
            if (q >= 0)
                bi[q] = bfull[p];
        }
        __syncthreads(); /* bfull lies where the ring is about to be filled */
        TINY_STAMP(5);
        lu_stream_solve<1>(lin, ring, bi, zi, dg, lane, wave);
        TINY_STAMP(6);
        for (int p = tid; p < nc; p += TINY_CYC_THREADS) {
            const int q = lu.in_map[p];
            if (q >= 0)
                ec[p] = bi[q];
        }
    } else {
        const int n0 = lu.n;
        double *z = S, *dg = S + n0; /* r is dead; bfull (the right-hand side, then the solution) lies behind both */
        for (int p = tid; p < n0; p += TINY_CYC_THREADS) {
            dg[p] = lu.diag[p];
            dg[n0 + p] = lu.diag[lu.npad + p];
        }
        __syncthreads();
        if (wave == 0)
            lu_wave_pass<RF, true, false>(lu, lane, bfull, z, dg);
        __syncthreads();
        if (wave == 0) {
            if (lu.fast_div)
                lu_wave_pass<RF, false, true>(lu, lane, z, bfull, dg);
            else
                lu_wave_pass<RF, false, false>(lu, lane, z, bfull, dg);
        }
        __syncthreads();
        for (int p = tid; p < nc; p += TINY_CYC_THREADS)
            ec[p] = bfull[p];
    }
    __syncthreads();
    TINY_STAMP(7);
    for (int t = tid; t < nc; t += TINY_CYC_THREADS) {
        const int ic = t / (Nc * Nc), jc = (t / Nc) % Nc, kc = t % Nc;
        xc[gc.plane * ic + (long long)gc.pitch * jc + kc] = ec[t];
    }
    /* ---- up (tiny_up_kernel): prolongation + correction at every fine point (mg_3d.h:1000-1145), post-smoothing */
    const int sI = Nc * Nc, sJ = Nc, sK = 1;
    int pi = i0, pj = j0, pk = k0;
    for (int t = tid; t < n; t += TINY_CYC_THREADS) {
        const int i = pi, j = pj, k = pk;
        advance(pi, pj, pk);
        const int oi = i & 1, oj = j & 1, ok = k & 1;
        const int c0 = (((i - oi) / 2) * Nc + (j - oj) / 2) * Nc + (k - ok) / 2;
        double x = 0.;
        switch (oi + oj + ok) {
        case 3:
            x += ec[c0];
            x += ec[c0 + sK];
            x += ec[c0 + sJ];
            x += ec[c0 + sJ + sK];
            x += ec[c0 + sI];
            x += ec[c0 + sI + sK];
            x += ec[c0 + sI + sJ];
            x += ec[c0 + sI + sJ + sK];
            x *= 0.125;
            break;
        case 2:
            if (!oi) {
                x += ec[c0];
                x += ec[c0 + sJ];
                x += ec[c0 + sK];
                x += ec[c0 + sJ + sK];
            } else if (!oj) {
                x += ec[c0];
                x += ec[c0 + sI];
                x += ec[c0 + sK];
                x += ec[c0 + sI + sK];
            } else {
                x += ec[c0];
                x += ec[c0 + sJ];
                x += ec[c0 + sI];
                x += ec[c0 + sI + sJ];
            }
            x *= 0.25;
            break;
        case 1:
            x += ec[c0];
            x += ec[c0 + oi * sI + oj * sJ + ok * sK];
            x *= 0.5;
            break;
        default:
            x = ec[c0];
        }
        u[t] += x;
    }
    __syncthreads();
    TINY_STAMP(8);
    for (int s = 0; s < iters; s++) {
        pass(0);
        pass(1);
    }
    TINY_STAMP(9);
    {
        int i = i0, j = j0, k = k0;
        for (int t = tid; t < n; t += TINY_CYC_THREADS) {
            u_io[g.plane * i + (long long)g.pitch * j + k] = u[t];
            advance(i, j, k);
        }
    }
    TINY_STAMP(10);
}

/* Dynamic LDS above the 64 KB a kernel gets without asking (118 KB at 17^3) is granted per kernel AND per device: asked
 * for once per (kernel, device), the answer remembered; a refusal (or a device whose opt-in limit is too small) sends the
 * level through the generic kernels instead of failing at launch. */
static bool tiny_lds_granted(const void *kernel, size_t lds)
{
    if (lds <= 65536)
        return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return false;
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> granted; /* bytes granted, 0 = refused */
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_pair(kernel, dev);
    auto it = granted.find(key);
    if (it == granted.end() || (it->second != 0 && it->second < lds)) {
        int optin = 0;
        bool ok = hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && (size_t)optin >= lds;
        if (!ok && optin == 0) /* attribute not reported: let the grant itself decide */
            ok = true;
        ok = ok && hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        (void)hipGetLastError();
        granted[key] = ok ? lds : 0;
        it = granted.find(key);
    }
    return it->second != 0;
}

static size_t tiny_down_lds(const Geom &g) { return sizeof(double) * 3 * (size_t)g.N * g.N * g.N; }
static size_t tiny_up_lds(const Geom &g, const Geom &gc)
{
    return sizeof(double) * (2 * (size_t)g.N * g.N * g.N + (size_t)gc.N * gc.N * gc.N);
}

bool k_tiny_fits(const Geom &g, const Geom &gc)
{
    const bool shape = g.ig0 == 0 && g.ni == g.N && g.nj == g.N && g.nk == g.N && g.N >= 3 && g.N <= TINY_MAX_N &&
                       gc.N == (g.N + 1) / 2 && gc.ig0 == 0 && gc.ni == gc.N;
    return shape && tiny_lds_granted((const void *)tiny_down_kernel, tiny_down_lds(g)) &&
           tiny_lds_granted((const void *)tiny_up_kernel, tiny_up_lds(g, gc));
}

void k_tiny_down(const Geom &g, double *u, const double *d, const double *r, const Geom &gc, double *dc, double h, int iters,
                 hipStream_t s)
{
    hipLaunchKernelGGL(tiny_down_kernel, dim3(1), dim3(TINY_THREADS), tiny_down_lds(g), s, g, u, d, r, gc, dc, h * h, 1. / 6,
                       1. / (h * h), iters);
}

void k_tiny_up(const Geom &g, double *u, const double *d, const Geom &gc, const double *ec, double h, int iters, hipStream_t s)
{
    hipLaunchKernelGGL(tiny_up_kernel, dim3(1), dim3(TINY_THREADS), tiny_up_lds(g, gc), s, g, u, d, gc, ec, h * h, 1. / 6, iters);
}

/* the S region of tiny_cycle_kernel in doubles: r and the full right-hand side, or the reduced solve's ring and vectors */
static int tiny_cycle_s_doubles(const Geom &g, const LuBand &lu, const LuBand &lin)
{
    const int n = g.N * g.N * g.N;
    const int a = n + lu.npad, b = 2 * 64 * 64 + 4 * lin.npad, c = 3 * lu.n;
    return std::max(a, std::max(b, c));
}

static size_t tiny_cycle_lds(const Geom &g, const Geom &gc, const LuBand &lu, const LuBand &lin)
{
    return sizeof(double) * (2 * (size_t)g.N * g.N * g.N + (size_t)tiny_cycle_s_doubles(g, lu, lin) + (size_t)gc.N * gc.N * gc.N);
}

bool k_tiny_cycle_fits(const Geom &g, const Geom &gc, const LuBand &lu, const LuBand &lin)
{
    if (!k_tiny_fits(g, gc) || !lu.in_map || lin.n <= 0 || lin.rot_r != 1 || lin.stream_ch != 64 || !lin.stream)
        return false;
    if ((lu.rot_r != 1 && lu.rot_r != 2) || !lu.lrot || !lu.urot || lu.n != gc.N * gc.N * gc.N)
        return false;
    const int M = g.N - 2;
    if (M * M * M > TINY_PTS * TINY_CYC_THREADS)
        return false;
    const size_t lds = tiny_cycle_lds(g, gc, lu, lin) + 512; /* + the workgroup vote's static bytes */
    const void *k = lu.rot_r == 2 ? (const void *)tiny_cycle_kernel<2> : (const void *)tiny_cycle_kernel<1>;
    int dev = 0, max_lds = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess)
        return false;
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && optin > max_lds)
        max_lds = optin;
    return lds <= (size_t)max_lds && tiny_lds_granted(k, lds - 512);
}

void k_tiny_cycle(const Geom &g, double *u, const double *d, const double *r, const Geom &gc, double *dc, double *xc,
                  const LuBand &lu, const LuBand &lin, double h, int iters, hipStream_t s)
{
    const size_t lds = tiny_cycle_lds(g, gc, lu, lin);
    const int sd = tiny_cycle_s_doubles(g, lu, lin);
    if (lu.rot_r == 2)
        hipLaunchKernelGGL(tiny_cycle_kernel<2>, dim3(1), dim3(TINY_CYC_THREADS), lds, s, g, u, d, r, gc, dc, xc, lu, lin, h * h,
                           1. / 6, 1. / (h * h), iters, sd);
    else
        hipLaunchKernelGGL(tiny_cycle_kernel<1>, dim3(1), dim3(TINY_CYC_THREADS), lds, s, g, u, d, r, gc, dc, xc, lu, lin, h * h,
                           1. / 6, 1. / (h * h), iters, sd);
}

extern "C" int mg3d_debug_tiny_stamps(long long *out16)
{
    if (!out16)
        return MG3D_ERR_ARG;
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tiny_stamps), 16 * sizeof(long long)) == hipSuccess ? MG3D_OK : MG3D_ERR_HIP;
}
