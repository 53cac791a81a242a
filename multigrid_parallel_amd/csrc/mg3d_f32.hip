/*
 * mg3d_f32.hip -- the single-precision / damped-Jacobi / F-cycle variant (BASELINE configs[4]).
 *
 * PARITY UNPINNED: the reference has no fp32 arithmetic, no Jacobi smoother and only a commented-out FMG
 * start (mg_3d.h:1364-1404, mg_dirichlet_analytic.c:771-806).  What this variant computes is therefore
 * defined here and restated in plain C by the test infrastructure, which the tests compare with
 * bit for bit:
 *   - storage and grid arithmetic in IEEE binary32, no contraction, the reference's association for the
 *     seven-point sum (mg_3d.h:438-443), the residual (:819-821), the 27-point restriction (:973-989) and
 *     the parent orders of the prolongation (:1028-1138);
 *   - smoother: weighted Jacobi, v' = v + omega * ((1/6) * (sum6 - h^2 d) - v), out of place;
 *   - coarsest level: the reference's LU factors (double), the right-hand side widened, the solution
 *     rounded back;
 *   - norms: squares of the binary32 residuals accumulated in double;
 *   - F-cycle: the FMG start of mg_dirichlet_analytic.c:771-806 (coarsest solve with boundary values, then per
 *     level: prolongate the solution, re-impose the boundary values, zero the coarse guess, one V-cycle).
 * First correct path: one launch per sweep (2 reads + 1 write per point and sweep), no temporal blocking.
 */
#include "mg3d_ctx.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define fail mg3d_fail
#define HIPCHK(call)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(MG3D_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                      \
    } while (0)
#define CHK(call)           \
    do {                    \
        int rc_ = (call);   \
        if (rc_ != MG3D_OK) \
            return rc_;     \
    } while (0)

#include "mg3d_f32_int.h"

static inline int pitch32(int nk) { return (nk + 31) & ~31; }

__device__ __forceinline__ long long gidx32(const Geom &g, int i, int j, int k)
{
    return g.plane * i + (long long)g.pitch * j + k;
}
/* local plane q may be updated: not a physical boundary plane, not the outermost plane of an i-slab (which has no
 * neighbour on one side).  On a single domain (ig0 = 0, ni = N) simply 1 <= q <= N-2. */
__device__ __forceinline__ bool plane_upd(const Geom &g, int q)
{
    return q >= 1 && q <= g.ni - 2 && g.ig0 + q >= 1 && g.ig0 + q <= g.N - 2;
}
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
/* the smoother's output is not read again before the next launch: stored non-temporally it does not push the halo
 * rows the neighbouring tiles are about to re-read out of the caches (measured on the fp64 sweep: +3.5 %) */
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_stream(float *p, float4 v)
{
#ifdef MG3D_F32_PLAIN_STORES
    st4(p, v);
#else
    v4f x;
    x.x = v.x;
    x.y = v.y;
    x.z = v.z;
    x.w = v.w;
    __builtin_nontemporal_store(x, reinterpret_cast<v4f *>(p));
#endif
}

/* sum of the six neighbours in the reference's order (mg_3d.h:438-443): ((((i- + i+) + j-) + j+) + k-) + k+ */
__device__ __forceinline__ float sum6(float im, float ip, float jm, float jp, float km, float kp)
{
    float s = im + ip;
    s = s + jm;
    s = s + jp;
    s = s + km;
    s = s + kp;
    return s;
}

/* One damped-Jacobi sweep, out of place.  A thread owns four consecutive k of one row and marches `chunk`
 * planes along i with its column's i-1 / i / i+1 values in registers; every point of the level is written
 * (boundary points are copied) so the two buffers can simply be exchanged. */
__global__ void __launch_bounds__(256) jacobi32_kernel(Geom g, const float *__restrict__ vin,
                                                       const float *__restrict__ d, float *__restrict__ vout,
                                                       float hSq, float sixth, float omega, int chunk)
{
    const int k0 = 4 * (blockIdx.x * 64 + threadIdx.x);
    const int j = blockIdx.y * 4 + threadIdx.y;
    if (k0 >= g.nk || j >= g.nj)
        return;
    const int i0 = blockIdx.z * chunk, i1 = min(i0 + chunk, g.ni);
    const bool jin = j >= 1 && j <= g.nj - 2;
    long long p = gidx32(g, i0, j, k0);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 below = i0 > 0 ? ld4(vin + p - g.plane) : zero, here = ld4(vin + p);
    for (int i = i0; i < i1; i++, p += g.plane) {
        const float4 above = i + 1 < g.ni ? ld4(vin + p + g.plane) : zero;
        float4 out = here;
        if (jin && plane_upd(g, i)) {
            const float4 jm = ld4(vin + p - g.pitch), jp = ld4(vin + p + g.pitch), dd = ld4(d + p);
            const float left = k0 > 0 ? vin[p - 1] : 0.f;
            const float right = k0 + 4 < g.nk ? vin[p + 4] : 0.f;
            const float hv[6] = {left, here.x, here.y, here.z, here.w, right};
            const float bl[4] = {below.x, below.y, below.z, below.w}, ab[4] = {above.x, above.y, above.z, above.w};
            const float jmv[4] = {jm.x, jm.y, jm.z, jm.w}, jpv[4] = {jp.x, jp.y, jp.z, jp.w};
            const float dv[4] = {dd.x, dd.y, dd.z, dd.w};
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int k = k0 + c;
                const float s = sum6(bl[c], ab[c], jmv[c], jpv[c], hv[c], hv[c + 2]) - hSq * dv[c];
                const float gs = sixth * s;
                o[c] = (k >= 1 && k <= g.nk - 2) ? hv[c + 1] + omega * (gs - hv[c + 1]) : hv[c + 1];
            }
            out = make_float4(o[0], o[1], o[2], o[3]);
        }
        st4_stream(vout + p, out);
        below = here;
        here = above;
    }
}

/* Two damped-Jacobi sweeps in one pass over the level (temporal blocking through LDS).
 * A 1024-thread block owns an extended tile of 16 rows x 256 columns (one wave per row, four k per lane) and
 * marches along i.  With input planes a-2, a-1, a of its column in registers a thread forms sweep 1 of plane
 * a-1 (row and column neighbours of that plane from LDS / the neighbouring lanes), then -- with sweep-1 planes
 * a-3, a-2, a-1 in registers -- sweep 2 of plane a-2.  The outermost ring of the tile is valid input only, the
 * next ring valid after sweep 1 only: 12 x 248 points per plane leave the block (1.38x redundant reads instead
 * of a second pass over HBM).  Input and sweep-1 planes are double-buffered in LDS: one barrier per plane. */
__device__ __forceinline__ float4 jacobi_pt4(float4 below, float4 above, float4 jm, float4 jp, float left, float4 here,
                                             float right, float4 dd, float hSq, float sixth, float omega, bool upd_plane_row,
                                             const bool (&kin)[4] /* column k0 + c is an interior column */)
{
    const float hv[6] = {left, here.x, here.y, here.z, here.w, right};
    const float bl[4] = {below.x, below.y, below.z, below.w}, ab[4] = {above.x, above.y, above.z, above.w};
    const float jmv[4] = {jm.x, jm.y, jm.z, jm.w}, jpv[4] = {jp.x, jp.y, jp.z, jp.w};
    const float dv[4] = {dd.x, dd.y, dd.z, dd.w};
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float s = sum6(bl[c], ab[c], jmv[c], jpv[c], hv[c], hv[c + 2]) - hSq * dv[c];
        const float gs = sixth * s;
        /* `&`: a select on (wave-uniform flag AND loop-invariant lane mask), not a branch per component */
        o[c] = (upd_plane_row & kin[c]) ? hv[c + 1] + omega * (gs - hv[c + 1]) : hv[c + 1];
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

constexpr int J2_ROWS = 16, J2_OUT_ROWS = 12, J2_OUT_COLS = 248;
constexpr int J2N_OUT_ROWS = 10; /* with the residual norm as a third stage the tile gives up one more ring of rows */

__device__ __forceinline__ double wave_sum32(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        x += __shfl_down(x, off, 64);
    return x;
}

/* NORM: a third stage forms the residual of the twice-smoothed field (plane a-3, from sweep-2 planes a-4, a-3, a-2)
 * and accumulates (double)diff^2 over the block's own points: the level's norm without another pass over HBM. */
/* PRO: the level's input is vin + P(ec) (prolongateAndCorrectError, mg_3d.h:1000-1145, parents in the reference's
 * order per parity class): the coarse planes of the tile's footprint are staged in LDS (ring of three, one new plane
 * every other step, fetched a step ahead) and every plane that enters the pipeline gets its correction on the way
 * in -- the prolongation costs neither a launch nor a pass over the fine level. */
constexpr int CP_ROWS = 10, CP_COLS = 130;

/* TAP: the residual norm of the INPUT field, formed where sweep 1 gathers the six neighbours of an input point anyway
 * (a Jacobi sweep reads only old values: mg_3d.h:819-821 on operands the step already holds, no stage of its own) --
 * the norm of the cycle before, when its last launch left it to this one ("carried cycles", e32_vcycle) */
template <bool NORM, bool PRO, bool TAP = false>
__global__ void __launch_bounds__(1024) jacobi32x2_kernel(Geom g, const float *__restrict__ vin,
                                                          const float *__restrict__ d, float *__restrict__ vout,
                                                          float hSq, float sixth, float omega, float invHsq,
                                                          double *__restrict__ partials, int chunk, Geom gc,
                                                          const float *__restrict__ ec, int acc_lo, int acc_hi)
{
    constexpr int OUT_ROWS = NORM ? J2N_OUT_ROWS : J2_OUT_ROWS, R0 = NORM ? 3 : 2;
    __shared__ float cpl[PRO ? 3 : 1][PRO ? CP_ROWS : 1][PRO ? CP_COLS : 1];
    /* one plane each of the input, the sweep-1 and (NORM) the sweep-2 field: 32 / 48 KB, two blocks per CU.
     * Per step: publish the input plane | barrier | every thread fetches the row neighbours it needs of all
     * three planes | barrier | compute the stages and publish their planes for the next step. */
    __shared__ float4 inp[J2_ROWS][64];
    __shared__ float4 s1b[J2_ROWS][64];
    __shared__ float4 s2b[NORM ? J2_ROWS : 1][NORM ? 64 : 1];
    __shared__ double red[16];
    /* a wave is one row of the tile (blockDim.x = 64): the row index through readfirstlane, so that everything derived
     * from it (row flags, the j parity of the prolongation) is a scalar and not a 64-bit lane mask (+0.7 %) */
    const int lane = threadIdx.x, r = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int j = (int)blockIdx.y * OUT_ROWS - R0 + r;
    const int k0 = (int)blockIdx.x * J2_OUT_COLS - 4 + 4 * lane;
    const int i0 = blockIdx.z * chunk, i1 = min(i0 + chunk, g.ni);
    const bool in_dom = j >= 0 && j < g.nj && k0 >= 0 && k0 < g.nk; /* pitch covers a partial last vector */
    const bool row_upd = j >= 1 && j <= g.nj - 2;
    const bool own = in_dom && r >= R0 && r < R0 + OUT_ROWS && lane >= 1 && lane <= 62;
    /* loop invariants of the plane loop, kept as scalars / lane masks: the planes that may be updated (plane_upd), the
     * interior columns of this lane's four, the columns that enter the norm.  The scalar unit is shared by the CU's
     * 32 waves of this kernel: per-plane tests re-evaluated in every step were a third of a step's time. */
    const int pu_lo = max(1, 1 - g.ig0), pu_hi = min(g.ni - 2, g.N - 2 - g.ig0);
    const bool kin[4] = {k0 >= 1 && k0 <= g.nk - 2, k0 + 1 >= 1 && k0 + 1 <= g.nk - 2, k0 + 2 >= 1 && k0 + 2 <= g.nk - 2,
                         k0 + 3 >= 1 && k0 + 3 <= g.nk - 2};
    const bool kacc[4] = {own && kin[0], own && kin[1], own && kin[2], own && kin[3]};
    const long long col = (long long)g.pitch * j + k0;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load = [&](const float *base, int i) { return (in_dom && i >= 0 && i < g.ni) ? ld4(base + g.plane * i + col) : zero; };
    const int rm = r > 0 ? r - 1 : r, rp = r < J2_ROWS - 1 ? r + 1 : r;
    /* registers: input planes a-2 (in_m), a-1 (in_c); sweep-1 planes a-3 (s_m), a-2 (s_c); d of planes a-2, a-3;
     * NORM: sweep-2 planes a-4 (o_m), a-3 (o_c) */
    const int a0 = i0 - (NORM ? 2 : 1), a1 = i1 + (NORM ? 2 : 1);
    /* ---- PRO: coarse staging */
    const int tid = r * 64 + lane;
    const int jt0 = (int)blockIdx.y * OUT_ROWS - R0, kt0 = (int)blockIdx.x * J2_OUT_COLS - 4;
    const int jcb = (jt0 - (jt0 & 1)) / 2, kcb = kt0 / 2; /* floor: jt0 may be odd and negative, kt0 is a multiple of 4 */
    auto slot_of = [](int c) { return ((c % 3) + 3) % 3; };
    /* the two staged values of this thread: in-plane offset and validity are loop invariants (lane masks / registers),
     * only the coarse plane index changes from fetch to fetch */
    bool cf_ok[2];
    long long cf_off[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int idx = tid + t * 1024, row = idx / CP_COLS, cc = idx - row * CP_COLS;
        const int jc = jcb + row, kc = kcb + cc;
        cf_ok[t] = PRO && idx < CP_ROWS * CP_COLS && jc >= 0 && jc < gc.nj && kc >= 0 && kc < gc.nk;
        cf_off[t] = (long long)gc.pitch * jc + kc;
    }
    auto coarse_fetch = [&](int c, float(&buf)[2]) {
        const bool cin = (c >= 0) & (c < gc.ni); /* wave-uniform */
        const long long pc = gc.plane * c;
#pragma unroll
        for (int t = 0; t < 2; t++)
            buf[t] = (cin & cf_ok[t]) ? ec[pc + cf_off[t]] : 0.f;
    };
    auto coarse_put = [&](int c, const float(&buf)[2]) {
        const int sl = slot_of(c);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int idx = tid + t * 1024;
            if (idx < CP_ROWS * CP_COLS)
                (&cpl[sl][0][0])[idx] = buf[t];
        }
    };
    /* v_in(plane i) = u + P(ec): the staged planes cover coarse_lo(i) and, for odd i, the one above.  Parity and
     * parents follow the GLOBAL plane index ig0 + i; cl is the local index of the lower coarse parent plane. */
    auto pro_apply = [&](float4 v, int i) { /* v = plane i of vin as loaded */
        if constexpr (PRO) {
            if (in_dom && i >= 0 && i < g.ni) {
                const int oi = (g.ig0 + i) & 1, oj = j & 1;
                const int cl = (g.ig0 + i - oi) / 2 - gc.ig0, lr = (j - oj) / 2 - jcb, lc = 2 * lane;
                const int s0 = slot_of(cl), s1 = slot_of(cl + 1);
                float E0[2][3], E1[2][3];
#pragma unroll
                for (int rr = 0; rr < 2; rr++)
#pragma unroll
                    for (int cc = 0; cc < 3; cc++) {
                        E0[rr][cc] = cpl[s0][lr + rr][lc + cc];
                        E1[rr][cc] = cpl[s1][lr + rr][lc + cc];
                    }
                /* one branch on the plane's (i, j) parity class, then the four components straight-line: k even takes
                 * the parents of column cc = c/2, k odd those of cc and cc + 1, each in the reference's order */
                float t[4];
                if (!oi && !oj) {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int cc = c >> 1;
                        t[c] = !(c & 1) ? E0[0][cc] : (E0[0][cc] + E0[0][cc + 1]) * 0.5f;
                    }
                } else if (!oi) {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int cc = c >> 1;
                        t[c] = !(c & 1) ? (E0[0][cc] + E0[1][cc]) * 0.5f
                                        : (((E0[0][cc] + E0[1][cc]) + E0[0][cc + 1]) + E0[1][cc + 1]) * 0.25f;
                    }
                } else if (!oj) {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int cc = c >> 1;
                        t[c] = !(c & 1) ? (E0[0][cc] + E1[0][cc]) * 0.5f
                                        : (((E0[0][cc] + E1[0][cc]) + E0[0][cc + 1]) + E1[0][cc + 1]) * 0.25f;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int cc = c >> 1;
                        if (!(c & 1)) {
                            t[c] = (((E0[0][cc] + E0[1][cc]) + E1[0][cc]) + E1[1][cc]) * 0.25f;
                        } else {
                            float x = E0[0][cc] + E0[0][cc + 1];
                            x = x + E0[1][cc];
                            x = x + E0[1][cc + 1];
                            x = x + E1[0][cc];
                            x = x + E1[0][cc + 1];
                            x = x + E1[1][cc];
                            x = x + E1[1][cc + 1];
                            t[c] = x * 0.125f;
                        }
                    }
                }
                v.x += t[0];
                v.y += t[1];
                v.z += t[2];
                v.w += t[3];
            }
        }
        return v;
    };
    int have_hi = 0;
    if constexpr (PRO) {
        const int gl = g.ig0 + a0 - 2;
        const int lo = (gl - (gl & 1)) / 2 - gc.ig0; /* floor(global plane / 2), as a local coarse index */
        float buf[2];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            coarse_fetch(lo + c, buf);
            coarse_put(lo + c, buf);
        }
        have_hi = lo + 2;
        __syncthreads();
    }
    auto load_in = [&](int i) { return pro_apply(load(vin, i), i); };
    float4 in_m = load_in(a0 - 2), in_c = load_in(a0 - 1), s_m = zero, s_c = zero, d2 = zero, d3 = zero;
    float4 o_m = zero, o_c = zero;
    double acc = 0.;
    s1b[r][lane] = zero;
    if constexpr (NORM)
        s2b[r][lane] = zero;
    /* The NORM / PRO variants need 73-103 VGPRs: one block per CU, nobody to hide a load behind.  They request the next
     * step's planes a step ahead (8 more registers, same occupancy); the plain pair runs two blocks per CU at 63 VGPRs
     * and must not grow. */
#ifndef MG3D_F32_TAP_PREF
#define MG3D_F32_TAP_PREF 0 /* same-box A/B at 1025^3: 11.71 ms per cycle without, 12.26 with (plain schedule 12.54) */
#endif
    constexpr bool PREF = NORM || PRO || (TAP && MG3D_F32_TAP_PREF != 0); /* (TAP: 79 VGPRs; held to 64 it spills and the pair takes 30 % longer) */
    float4 raw_next = PREF ? load(vin, a0) : zero, d_next = PREF ? load(d, a0 - 1) : zero;
    for (int a = a0; a <= a1; a++) {
        float4 raw, d1;
        if constexpr (PREF) {
            raw = raw_next;
            d1 = d_next;
            raw_next = load(vin, a + 1);
            d_next = load(d, a);
        } else {
            raw = load(vin, a);
            d1 = load(d, a - 1);
        }
        const float4 in_p = pro_apply(raw, a);
        float cbuf[2];
        bool stage_new = false;
        if constexpr (PRO) { /* plane a+1 (global G) needs coarse planes up to ceil(G/2) */
            stage_new = a + 1 >= 0 && (g.ig0 + a + 2) / 2 - gc.ig0 > have_hi; /* G >= 0: (G+1)/2 == ceil(G/2) */
            if (stage_new)
                coarse_fetch(have_hi + 1, cbuf);
        }
        inp[r][lane] = in_c;
        __syncthreads();
        if constexpr (PRO) {
            if (stage_new) {
                coarse_put(have_hi + 1, cbuf);
                have_hi++;
            }
        }
        const float4 ijm = inp[rm][lane], ijp = inp[rp][lane];   /* input plane a-1 */
        const float4 sjm = s1b[rm][lane], sjp = s1b[rp][lane];   /* sweep-1 plane a-2, published one step ago */
        float4 ojm = zero, ojp = zero;
        if constexpr (NORM) {
            ojm = s2b[rm][lane];                                 /* sweep-2 plane a-3 */
            ojp = s2b[rp][lane];
        }
        __syncthreads();
        /* sweep 1 of plane a-1 */
        float4 s_new;
        {
            const float left = __shfl_up(in_c.w, 1, 64), right = __shfl_down(in_c.x, 1, 64);
            const int q = a - 1;
            s_new = jacobi_pt4(in_m, in_p, ijm, ijp, left, in_c, right, d1, hSq, sixth, omega,
                               row_upd & (q >= pu_lo) & (q <= pu_hi), kin);
            if constexpr (TAP) {
                if (row_upd & (q >= i0) & (q < i1) & (q >= pu_lo) & (q <= pu_hi) & (q >= acc_lo) & (q < acc_hi)) { /* wave-uniform */
                    const float hv[6] = {left, in_c.x, in_c.y, in_c.z, in_c.w, right};
                    const float bl[4] = {in_m.x, in_m.y, in_m.z, in_m.w}, ab[4] = {in_p.x, in_p.y, in_p.z, in_p.w};
                    const float jmv[4] = {ijm.x, ijm.y, ijm.z, ijm.w}, jpv[4] = {ijp.x, ijp.y, ijp.z, ijp.w};
                    const float dv[4] = {d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const float ssum = sum6(bl[c], ab[c], jmv[c], jpv[c], hv[c], hv[c + 2]) - 6 * hv[c + 1];
                        const float diff = dv[c] - invHsq * ssum;
                        acc += kacc[c] ? (double)diff * (double)diff : 0.;
                    }
                }
            }
        }
        s1b[r][lane] = s_new;
        /* sweep 2 of plane a-2 */
        float4 o;
        {
            const int q = a - 2;
            const float left = __shfl_up(s_c.w, 1, 64), right = __shfl_down(s_c.x, 1, 64);
            o = jacobi_pt4(s_m, s_new, sjm, sjp, left, s_c, right, d2, hSq, sixth, omega,
                           row_upd & (q >= pu_lo) & (q <= pu_hi), kin);
            if ((q >= i0) & (q < i1)) /* wave-uniform */
                if (own)
                    st4_stream(vout + g.plane * q + col, o);
        }
        if constexpr (NORM) {
            s2b[r][lane] = o;
            /* residual of plane a-3 (mg_3d.h:819-821) */
            const int q = a - 3;
            const float left = __shfl_up(o_c.w, 1, 64), right = __shfl_down(o_c.x, 1, 64);
            /* wave-uniform: this row and plane enter the norm; which lanes / columns do is a loop-invariant mask, and
             * adding +0 leaves a sum of squares unchanged -- a select per component instead of a branch */
            if (row_upd & (q >= i0) & (q < i1) & (q >= pu_lo) & (q <= pu_hi) & (q >= acc_lo) & (q < acc_hi)) {
                const float hv[6] = {left, o_c.x, o_c.y, o_c.z, o_c.w, right};
                const float bl[4] = {o_m.x, o_m.y, o_m.z, o_m.w}, ab[4] = {o.x, o.y, o.z, o.w};
                const float jmv[4] = {ojm.x, ojm.y, ojm.z, ojm.w}, jpv[4] = {ojp.x, ojp.y, ojp.z, ojp.w};
                const float dv[4] = {d3.x, d3.y, d3.z, d3.w};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float ssum = sum6(bl[c], ab[c], jmv[c], jpv[c], hv[c], hv[c + 2]) - 6 * hv[c + 1];
                    const float diff = dv[c] - invHsq * ssum;
                    acc += kacc[c] ? (double)diff * (double)diff : 0.;
                }
            }
            o_m = o_c;
            o_c = o;
            d3 = d2;
        }
        in_m = in_c;
        in_c = in_p;
        s_m = s_c;
        s_c = s_new;
        d2 = d1;
    }
    if constexpr (NORM || TAP) {
        acc = wave_sum32(acc);
        if (lane == 0)
            red[r] = acc;
        __syncthreads();
        if (r == 0 && lane == 0) {
            double t = 0.;
            for (int x = 0; x < J2_ROWS; x++)
                t += red[x];
            partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
        }
    }
}


/* diff = d - invHsq * (sum6 - 6 v) (mg_3d.h:819-821) in binary32; res (optional) on the interior only;
 * per-block partial sums of (double)diff^2 in a fixed order */
__global__ void __launch_bounds__(256) residual32_kernel(Geom g, const float *__restrict__ v,
                                                         const float *__restrict__ d, float invHsq,
                                                         float *__restrict__ res, double *__restrict__ partials,
                                                         int chunk, int p_lo, int p_hi, int acc_lo, int acc_hi)
{
    /* planes [p_lo, p_hi): all of them updatable (the launcher clips); the norm takes those in [acc_lo, acc_hi) */
    __shared__ double lds4[4];
    const int k0 = 4 * (blockIdx.x * 64 + threadIdx.x);
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int i0 = p_lo + blockIdx.z * chunk, i1 = min(i0 + chunk, p_hi);
    double acc = 0.;
    if (k0 < g.nk && j >= 1 && j <= g.nj - 2) {
        long long p = gidx32(g, i0, j, k0);
        float4 below = ld4(v + p - g.plane), here = ld4(v + p);
        for (int i = i0; i < i1; i++, p += g.plane) {
            const float4 above = ld4(v + p + g.plane);
            const float4 jm = ld4(v + p - g.pitch), jp = ld4(v + p + g.pitch), dd = ld4(d + p);
            const float left = k0 > 0 ? v[p - 1] : 0.f;
            const float right = k0 + 4 < g.nk ? v[p + 4] : 0.f;
            const float hv[6] = {left, here.x, here.y, here.z, here.w, right};
            const float bl[4] = {below.x, below.y, below.z, below.w}, ab[4] = {above.x, above.y, above.z, above.w};
            const float jmv[4] = {jm.x, jm.y, jm.z, jm.w}, jpv[4] = {jp.x, jp.y, jp.z, jp.w};
            const float dv[4] = {dd.x, dd.y, dd.z, dd.w};
            float df[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int k = k0 + c;
                const float s = sum6(bl[c], ab[c], jmv[c], jpv[c], hv[c], hv[c + 2]) - 6 * hv[c + 1];
                df[c] = dv[c] - invHsq * s;
                if (k >= 1 && k <= g.nk - 2 && i >= acc_lo && i < acc_hi)
                    acc += (double)df[c] * (double)df[c];
            }
            if (res) { /* interior only (mg_3d.h:824-825): one 16-byte store unless the vector touches a face */
                if (k0 >= 1 && k0 + 3 <= g.nk - 2) {
                    st4(res + p, make_float4(df[0], df[1], df[2], df[3]));
                } else {
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        if (k0 + c >= 1 && k0 + c <= g.nk - 2)
                            res[p + c] = df[c];
                }
            }
            below = here;
            here = above;
        }
    }
    const int tid = threadIdx.y * 64 + threadIdx.x;
    acc = wave_sum32(acc);
    if ((tid & 63) == 0)
        lds4[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0)
        partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = ((lds4[0] + lds4[1]) + lds4[2]) + lds4[3];
}

/* restrictResidual (mg_3d.h:844-998) in binary32: faces by injection, interior by the 27-point sum in
 * ti, tj, tk order starting from 0 */
__global__ void __launch_bounds__(256) restrict32_kernel(Geom gf, const float *__restrict__ r, Geom gc,
                                                         float *__restrict__ dc, int c_lo)
{
    /* coarse plane ic (local) sits under the fine plane with the doubled GLOBAL index */
    const int kc = blockIdx.x * 64 + threadIdx.x, jc = blockIdx.y * 4 + threadIdx.y, ic = c_lo + blockIdx.z;
    if (kc >= gc.nk || jc >= gc.nj)
        return;
    const int icg = gc.ig0 + ic;
    const long long pf = gidx32(gf, 2 * icg - gf.ig0, 2 * jc, 2 * kc);
    const bool face = icg == 0 || icg == gc.N - 1 || jc == 0 || jc == gc.nj - 1 || kc == 0 || kc == gc.nk - 1;
    float val;
    if (face) {
        val = r[pf];
    } else {
        val = 0.f;
#pragma unroll
        for (int ti = -1; ti <= 1; ti++)
#pragma unroll
            for (int tj = -1; tj <= 1; tj++)
#pragma unroll
                for (int tk = -1; tk <= 1; tk++) {
                    const float w = (ti ? 0.25f : 0.5f) * (tj ? 0.25f : 0.5f) * (tk ? 0.25f : 0.5f);
                    val += r[pf + ti * gf.plane + tj * (long long)gf.pitch + tk] * w;
                }
    }
    dc[gidx32(gc, ic, jc, kc)] = val;
}

/* Residual and full-weighting restriction in one pass (r never travels to HBM).  Same tile and plane ring as the
 * paired sweep: stage 1 forms the residual of fine plane q from the input planes q-1, q, q+1 and publishes it in
 * LDS; stage 2, one thread per coarse point of the tile (6 x 124), adds that plane's nine weighted values to
 * the running sum of the coarse plane(s) it belongs to -- plane 2ic-1 starts coarse plane ic (ti = 0), plane 2ic
 * continues it, plane 2ic+1 completes and stores it -- which is the reference's ti, tj, tk order from 0
 * (mg_3d.h:973-989).  Coarse faces (injection) are left to restrict32_faces_kernel. */
__global__ void __launch_bounds__(1024) residual_restrict32_kernel(Geom g, const float *__restrict__ v,
                                                                   const float *__restrict__ d, float invHsq, Geom gc,
                                                                   float *__restrict__ dc, int cchunk, int c_lo, int c_hi)
{
    __shared__ float4 inp[2][J2_ROWS][64];
    /* r of the tile's plane, even and odd columns apart: the restriction reads tile columns 3+2k, 4+2k, 5+2k for
     * consecutive coarse points k -- stride 2 in one array (two-way bank conflicts on each of the nine reads: half
     * of the kernel's LDS cycles), stride 1 in [parity][column / 2] */
    __shared__ float rb[2][J2_ROWS][2][128];
    const int lane = threadIdx.x, r = threadIdx.y, tid = r * 64 + lane;
    const int jt0 = (int)blockIdx.y * J2_OUT_ROWS - 2, kt0 = (int)blockIdx.x * J2_OUT_COLS - 4;
    const int j = jt0 + r, k0 = kt0 + 4 * lane;
    /* coarse planes [c0, c1) of this block, local indices out of the launch's [c_lo, c_hi) (interior coarse planes
     * only); the fine plane under coarse plane c has the doubled GLOBAL index: local qf(c) = 2 (gc.ig0 + c) - g.ig0,
     * and planes qf(c0)-1 .. qf(c1)-1 are needed */
    const int c0 = c_lo + (int)blockIdx.z * cchunk, c1 = min(c0 + cchunk, c_hi);
    if (c0 >= c1)
        return;
    const bool in_dom = j >= 0 && j < g.nj && k0 >= 0 && k0 < g.nk;
    const bool row_int = j >= 1 && j <= g.nj - 2;
    const long long col = (long long)g.pitch * j + k0;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load = [&](const float *base, int i) { return (in_dom && i >= 0 && i < g.ni) ? ld4(base + g.plane * i + col) : zero; };
    const int rm = r > 0 ? r - 1 : r, rp = r < J2_ROWS - 1 ? r + 1 : r;
    /* stage 2 mapping: coarse point (jcl, kcl) of the tile, fine centre at tile row 2 + 2 jcl, tile column 4 + 2 kcl */
    const int jcl = tid / 124, kcl = tid - jcl * 124;
    const int jc = (jt0 + 2) / 2 + jcl, kc = (kt0 + 4) / 2 + kcl;
    const bool cown = tid < 6 * 124 && jc >= 1 && jc <= gc.nj - 2 && kc >= 1 && kc <= gc.nk - 2;
    const int fr = 2 + 2 * jcl, fc = 4 + 2 * kcl; /* centre in tile coordinates */
    float accA = 0.f, accB = 0.f; /* coarse plane being completed / the one after it */
    const int q0 = 2 * (gc.ig0 + c0) - g.ig0 - 1, q1 = 2 * (gc.ig0 + c1) - g.ig0 - 1; /* fine planes q0 .. q1 inclusive */
    float4 in_m = load(v, q0 - 1), in_c = load(v, q0);
    for (int q = q0; q <= q1; q++) {
        const float4 in_p = load(v, q + 1);
        const float4 dd = load(d, q);
        const int pb = q & 1;
        inp[pb][r][lane] = in_c;
        __syncthreads();
        {
            const float4 jm = inp[pb][rm][lane], jp = inp[pb][rp][lane];
            const float left = __shfl_up(in_c.w, 1, 64), right = __shfl_down(in_c.x, 1, 64);
            const float hv[6] = {left, in_c.x, in_c.y, in_c.z, in_c.w, right};
            const float bl[4] = {in_m.x, in_m.y, in_m.z, in_m.w}, ab[4] = {in_p.x, in_p.y, in_p.z, in_p.w};
            const float jmv[4] = {jm.x, jm.y, jm.z, jm.w}, jpv[4] = {jp.x, jp.y, jp.z, jp.w};
            const float dv[4] = {dd.x, dd.y, dd.z, dd.w};
            float df[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int k = k0 + c;
                const float s = sum6(bl[c], ab[c], jmv[c], jpv[c], hv[c], hv[c + 2]) - 6 * hv[c + 1];
                /* r is zero where the reference never writes it (boundary points, mg_3d.h:824-825) */
                df[c] = (row_int && k >= 1 && k <= g.nk - 2) ? dv[c] - invHsq * s : 0.f;
            }
            *reinterpret_cast<float2 *>(&rb[pb][r][0][2 * lane]) = make_float2(df[0], df[2]); /* columns 4 lane, 4 lane + 2 */
            *reinterpret_cast<float2 *>(&rb[pb][r][1][2 * lane]) = make_float2(df[1], df[3]); /* columns 4 lane + 1, + 3 */
        }
        __syncthreads();
        if (cown) {
            /* this plane's nine products, tj then tk ascending; the i-weight is 1/2 on the centre plane, else 1/4 */
            const int qg = g.ig0 + q; /* the i-weight and the coarse plane follow the global index */
            const float wi = (qg & 1) ? 0.25f : 0.5f;
            float p9[9];
#pragma unroll
            for (int tj = 0; tj < 3; tj++)
#pragma unroll
                for (int tk = 0; tk < 3; tk++) {
                    const float w = (wi * (tj == 1 ? 0.5f : 0.25f)) * (tk == 1 ? 0.5f : 0.25f);
                    /* tile column fc - 1 + tk, fc even: tk = 1 is even column fc, tk = 0 / 2 the odd ones beside it */
                    p9[tj * 3 + tk] = (tk == 1 ? rb[pb][fr - 1 + tj][0][fc >> 1]
                                               : rb[pb][fr - 1 + tj][1][(fc >> 1) - 1 + (tk >> 1)]) * w;
                }
            if (qg & 1) {
                /* completes coarse plane (qg-1)/2, starts coarse plane (qg+1)/2 */
#pragma unroll
                for (int t = 0; t < 9; t++)
                    accA += p9[t];
                const int ic = (qg - 1) / 2 - gc.ig0;
                if (ic >= c0 && ic < c1)
                    dc[gc.plane * ic + (long long)gc.pitch * jc + kc] = accA;
                accB = 0.f;
#pragma unroll
                for (int t = 0; t < 9; t++)
                    accB += p9[t];
                accA = accB;
            } else {
#pragma unroll
                for (int t = 0; t < 9; t++)
                    accA += p9[t];
            }
        }
        in_m = in_c;
        in_c = in_p;
    }
}

/* injection on the six coarse faces (mg_3d.h:879-958) from the stored r (whose boundary entries nothing writes) */
__global__ void __launch_bounds__(256) restrict32_faces_kernel(Geom gf, const float *__restrict__ r, Geom gc,
                                                               float *__restrict__ dc, int c_lo, int c_hi)
{
    /* the face points among the local coarse planes [c_lo, c_hi); i-faces only where the slab holds the physical one */
    const int b = blockIdx.x * 64 + threadIdx.x, a = blockIdx.y * 4 + threadIdx.y, f = blockIdx.z;
    const int m = gc.nj; /* nj == nk == N */
    if (a >= m || b >= m)
        return;
    int ic, jc, kc;
    if (f < 2) {
        ic = (f == 0 ? 0 : gc.N - 1) - gc.ig0;
        jc = a;
        kc = b;
    } else if (f < 4) {
        ic = a - gc.ig0; /* a runs over global planes */
        jc = f == 2 ? 0 : gc.nj - 1;
        kc = b;
    } else {
        ic = a - gc.ig0;
        jc = b;
        kc = f == 4 ? 0 : gc.nk - 1;
    }
    if (ic < c_lo || ic >= c_hi)
        return;
    dc[gidx32(gc, ic, jc, kc)] = r[gidx32(gf, 2 * (gc.ig0 + ic) - gf.ig0, 2 * jc, 2 * kc)];
}

/* prolongateAndCorrectError (mg_3d.h:1000-1145) in binary32, cell form: a thread owns the 2 x 2 fine points
 * above one coarse cell and marches along i; parents summed in the reference's order per parity class */
__global__ void __launch_bounds__(256) prolong32_kernel(Geom gc, const float *__restrict__ ec, Geom gf,
                                                        float *__restrict__ ef, int chunk)
{
    const int m = blockIdx.x * 64 + threadIdx.x, jc = blockIdx.y * 4 + threadIdx.y;
    const int k0 = 2 * m, j0 = 2 * jc;
    if (k0 >= gf.nk || j0 >= gf.nj)
        return;
    const int i_beg = blockIdx.z * chunk, i_end = min(gf.ni, i_beg + chunk);
    const bool row1 = j0 + 1 < gf.nj, col1 = k0 + 1 < gf.nk;
    const int m1 = min(m + 1, gc.nk - 1), jc1 = min(jc + 1, gc.nj - 1);
    const long long c00 = (long long)gc.pitch * jc + m, c01 = (long long)gc.pitch * jc + m1;
    const long long c10 = (long long)gc.pitch * jc1 + m, c11 = (long long)gc.pitch * jc1 + m1;
    float E0[2][2], E1[2][2];
    int have = -0x40000000;
    auto load = [&](int il, float(&E)[2][2]) {
        const float *pl = ec + gc.plane * min(max(il, 0), gc.ni - 1);
        E[0][0] = pl[c00];
        E[0][1] = pl[c01];
        E[1][0] = pl[c10];
        E[1][1] = pl[c11];
    };
    for (int i = i_beg; i < i_end; i++) {
        const int oi = (gf.ig0 + i) & 1, il = (gf.ig0 + i - oi) / 2 - gc.ig0; /* global parity, local coarse plane */
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
        float t00, t01, t10, t11;
        if (!oi) {
            t00 = E0[0][0];
            t01 = (E0[0][0] + E0[0][1]) * 0.5f;
            t10 = (E0[0][0] + E0[1][0]) * 0.5f;
            t11 = (((E0[0][0] + E0[1][0]) + E0[0][1]) + E0[1][1]) * 0.25f;
        } else {
            t00 = (E0[0][0] + E1[0][0]) * 0.5f;
            t01 = (((E0[0][0] + E1[0][0]) + E0[0][1]) + E1[0][1]) * 0.25f;
            t10 = (((E0[0][0] + E0[1][0]) + E1[0][0]) + E1[1][0]) * 0.25f;
            float t = E0[0][0] + E0[0][1];
            t = t + E0[1][0];
            t = t + E0[1][1];
            t = t + E1[0][0];
            t = t + E1[0][1];
            t = t + E1[1][0];
            t = t + E1[1][1];
            t11 = t * 0.125f;
        }
        float *row = ef + gf.plane * i + (long long)gf.pitch * j0 + k0;
        float2 a = *reinterpret_cast<float2 *>(row);
        a.x += t00;
        if (col1)
            a.y += t01;
        *reinterpret_cast<float2 *>(row) = a;
        if (row1) {
            float2 b = *reinterpret_cast<float2 *>(row + gf.pitch);
            b.x += t10;
            if (col1)
                b.y += t11;
            *reinterpret_cast<float2 *>(row + gf.pitch) = b;
        }
    }
}

/* boundary values x^2 - 2y^2 + z^2 (mg_3d.h:89-90) evaluated in double at (i h, j h, k h) and rounded once */
__global__ void __launch_bounds__(256) fill_boundary32_kernel(Geom g, float *__restrict__ v, double h)
{
    const int k = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, i = blockIdx.z;
    if (k >= g.nk || j >= g.nj)
        return;
    const int ig = g.ig0 + i;
    if (!(ig == 0 || ig == g.N - 1 || j == 0 || j == g.nj - 1 || k == 0 || k == g.nk - 1))
        return;
    const double x = ig * h, y = j * h, z = k * h;
    v[gidx32(g, i, j, k)] = (float)(x * x - 2 * y * y + z * z);
}

/* coarsest level <-> the double context of the direct solve */
__global__ void widen32_kernel(Geom g32, const float *__restrict__ a, Geom g64, double *__restrict__ b)
{
    const int k = threadIdx.x, j = blockIdx.x, i = blockIdx.y;
    if (k < g32.nk)
        b[g64.plane * i + (long long)g64.pitch * j + k] = (double)a[gidx32(g32, i, j, k)];
}
__global__ void narrow32_kernel(Geom g64, const double *__restrict__ a, Geom g32, float *__restrict__ b)
{
    const int k = threadIdx.x, j = blockIdx.x, i = blockIdx.y;
    if (k < g32.nk)
        b[gidx32(g32, i, j, k)] = (float)a[g64.plane * i + (long long)g64.pitch * j + k];
}

/* ------------------------------------------------------------------------------------------ context */
extern "C" int mg3d32_destroy(mg3d32_ctx *ctx)
{
    if (!ctx)
        return MG3D_OK;
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    for (auto &l : ctx->lv) {
        for (int k = 0; k < 3; k++)
            if (l.f[k])
                (void)hipFree(l.f[k]);
        if (l.alt)
            (void)hipFree(l.alt);
    }
    for (auto &p : ctx->pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (hipEvent_t e : ctx->event_pool)
        (void)hipEventDestroy(e);
    if (ctx->partials)
        (void)hipFree(ctx->partials);
    if (ctx->sumsq)
        (void)hipFree(ctx->sumsq);
    if (ctx->h_sumsq)
        (void)hipHostFree(ctx->h_sumsq);
    if (ctx->coarse64)
        mg3d_ctx_destroy(ctx->coarse64); /* owns the stream */
    delete ctx;
    return MG3D_OK;
}

extern "C" int mg3d32_create(int coarse_pts, int num_levels, int smooth_iters, double omega, double grid_length,
                             mg3d32_ctx **out)
{
    return mg3d32_create_slabs(coarse_pts, num_levels, smooth_iters, omega, grid_length, num_levels, nullptr, nullptr, 0,
                               nullptr, out);
}

int mg3d32_create_slabs(int coarse_pts, int num_levels, int smooth_iters, double omega, double grid_length,
                        int first_slab, const int *glo, const int *ghi, int halo, hipStream_t share, mg3d32_ctx **out)
{
    if (!out || coarse_pts < 3 || coarse_pts > 11 || num_levels < 1 || num_levels > 24 || smooth_iters < 0 ||
        !(omega > 0.) || !(grid_length > 0.))
        return fail(MG3D_ERR_ARG, "mg3d32_create: bad arguments");
    if (mg3d_device_count() <= 0)
        return fail(MG3D_ERR_NO_DEVICE, "no HIP device available: libmg3d has no CPU fallback");
    mg3d32_ctx *ctx = new mg3d32_ctx();
    ctx->c = coarse_pts;
    ctx->L = num_levels;
    ctx->iters = smooth_iters;
    ctx->omega = (float)omega;
    ctx->no_pairs = getenv("MG3D_F32_NO_PAIRS") && getenv("MG3D_F32_NO_PAIRS")[0] == '1';
    ctx->no_fuse = getenv("MG3D_F32_NO_FUSE") && getenv("MG3D_F32_NO_FUSE")[0] == '1';
    ctx->no_carry = getenv("MG3D_F32_NO_CARRY") && getenv("MG3D_F32_NO_CARRY")[0] == '1'; /* (creation: the only reads) */
    ctx->timing = false;
    for (auto &k : ctx->kt)
        k.calls = 0, k.seconds = 0.;
    ctx->coarse64 = nullptr;
    ctx->partials = ctx->sumsq = ctx->h_sumsq = nullptr;
    ctx->stream = nullptr;
    int rc = mg3d_ctx_create(coarse_pts, 1, smooth_iters, 1.0, &ctx->coarse64);
    if (rc != MG3D_OK) {
        delete ctx;
        return rc;
    }
    ctx->own_stream = true;
    if (share) { /* several contexts of one process (virtual ranks) run on one ordered stream */
        (void)hipStreamDestroy(ctx->coarse64->stream);
        ctx->coarse64->stream = share;
        ctx->coarse64->own_stream = false;
        ctx->own_stream = false;
    }
    ctx->stream = ctx->coarse64->stream;
    const long long finest = ((long long)(coarse_pts - 1) << (num_levels - 1)) + 1;
    std::vector<double> hs(num_levels);
    hs[num_levels - 1] = grid_length / (double)(finest - 1);
    for (int l = num_levels - 2; l >= 0; l--)
        hs[l] = 2 * hs[l + 1];
    /* the coarse operator is the reference's, with the coarsest spacing (mg_3d.h:287) */
    rc = mg3d_ctx_build_coarse(ctx->coarse64, hs[0]);
    if (rc != MG3D_OK) {
        mg3d32_destroy(ctx);
        return rc;
    }
#define C32(call)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            const int rc_ = fail(e_ == hipErrorOutOfMemory ? MG3D_ERR_ALLOC : MG3D_ERR_HIP,   \
                                 "%s failed: %s", #call, hipGetErrorString(e_));               \
            mg3d32_destroy(ctx);                                                               \
            return rc_;                                                                        \
        }                                                                                      \
    } while (0)
    ctx->lv.resize(num_levels);
    for (int l = 0; l < num_levels; l++) {
        Level32 &lev = ctx->lv[l];
        const int N = (coarse_pts - 1) * (1 << l) + 1;
        lev.g.ni = lev.g.nj = lev.g.nk = lev.g.N = N;
        lev.g.ig0 = 0;
        lev.own_lo = 0;
        lev.own_hi = N;
        if (l >= first_slab) { /* owned planes [glo, ghi) + `halo` planes towards every neighbour */
            const int h_lo = glo[l] > 0 ? halo : 0, h_hi = ghi[l] < N ? halo : 0;
            lev.g.ig0 = glo[l] - h_lo;
            lev.g.ni = (ghi[l] - glo[l]) + h_lo + h_hi;
            lev.own_lo = h_lo;
            lev.own_hi = h_lo + (ghi[l] - glo[l]);
        }
        lev.g.pitch = pitch32(N);
        lev.g.plane = (long long)lev.g.pitch * N;
        lev.hd = hs[l];
        lev.h = (float)hs[l];
        lev.hSq = lev.h * lev.h;
        lev.invHsq = 1.0f / (lev.h * lev.h);
        lev.elems = (size_t)lev.g.plane * lev.g.ni;
        for (int k = 0; k < 3; k++)
            lev.f[k] = nullptr;
        lev.alt = nullptr;
    }
    for (auto &lev : ctx->lv) {
        for (int k = 0; k < 3; k++) {
            C32(hipMalloc(&lev.f[k], lev.elems * sizeof(float)));
            C32(hipMemsetAsync(lev.f[k], 0, lev.elems * sizeof(float), ctx->stream));
        }
        C32(hipMalloc(&lev.alt, lev.elems * sizeof(float)));
        C32(hipMemsetAsync(lev.alt, 0, lev.elems * sizeof(float), ctx->stream));
    }
    ctx->sumsq_slots = 1024;
    C32(hipMalloc(&ctx->partials, sizeof(double) * MG3D_MAX_PARTIALS));
    C32(hipMalloc(&ctx->sumsq, sizeof(double) * ctx->sumsq_slots));
    C32(hipHostMalloc(&ctx->h_sumsq, sizeof(double) * ctx->sumsq_slots));
    C32(hipStreamSynchronize(ctx->stream));
#undef C32
    *out = ctx;
    return MG3D_OK;
}

/* launch policy by key: "pairs" (two sweeps per launch), "fuse" (prolongation / norm / restriction folded into the
 * sweeps' launches), "carry" (a cycle's norm tapped from the next cycle's first launch); 1 on (default), 0 off */
extern "C" int mg3d32_set_option(mg3d32_ctx *ctx, const char *key, int value)
{
    if (!ctx || !key)
        return fail(MG3D_ERR_ARG, "mg3d32_set_option: NULL argument");
    if (strcmp(key, "pairs") == 0)
        ctx->no_pairs = value == 0;
    else if (strcmp(key, "fuse") == 0)
        ctx->no_fuse = value == 0;
    else if (strcmp(key, "carry") == 0)
        ctx->no_carry = value == 0;
    else
        return fail(MG3D_ERR_ARG, "mg3d32_set_option: no option \"%s\"", key);
    return MG3D_OK;
}

extern "C" int mg3d32_level_n(const mg3d32_ctx *ctx, int level)
{
    return (ctx && level >= 0 && level < ctx->L) ? ctx->lv[level].g.N : -1;
}

static int check32(mg3d32_ctx *ctx, int field, int level, const char *who)
{
    if (!ctx || field < 0 || field > 2 || level < 0 || level >= ctx->L)
        return fail(MG3D_ERR_ARG, "%s: bad field/level", who);
    return MG3D_OK;
}

extern "C" int mg3d32_upload(mg3d32_ctx *ctx, int field, int level, const float *host)
{
    CHK(check32(ctx, field, level, "mg3d32_upload"));
    Level32 &l = ctx->lv[level];
    const int N = l.g.N;
    HIPCHK(hipMemcpy2DAsync(l.f[field], l.g.pitch * sizeof(float), host, N * sizeof(float), N * sizeof(float),
                            (size_t)N * N, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MG3D_OK;
}

extern "C" int mg3d32_download(mg3d32_ctx *ctx, int field, int level, float *host)
{
    CHK(check32(ctx, field, level, "mg3d32_download"));
    Level32 &l = ctx->lv[level];
    const int N = l.g.N;
    HIPCHK(hipMemcpy2DAsync(host, N * sizeof(float), l.f[field], l.g.pitch * sizeof(float), N * sizeof(float),
                            (size_t)N * N, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MG3D_OK;
}

extern "C" int mg3d32_sync(mg3d32_ctx *ctx)
{
    if (!ctx)
        return fail(MG3D_ERR_ARG, "mg3d32_sync: NULL");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MG3D_OK;
}

/* ---------------------------------------------------------------------------------------- kernel timers */
static const char *const kKernel32Names[MG3D32_NUM_KERNELS] = {"pair", "pair+tap", "prolong+pair", "prolong+pair+norm", "pair+norm",
                                                               "residual+restrict", "residual", "prolong", "sweep1"};
struct Scope32 { /* an event pair around the launches of one finest-level operator, when timing is on */
    mg3d32_ctx *c;
    int k;
    hipEvent_t a;
    Scope32(mg3d32_ctx *ctx, int level, int kernel) : c(ctx), k(kernel), a(nullptr)
    {
        if (!ctx->timing || level != ctx->L - 1)
            return;
        auto take = [&]() -> hipEvent_t {
            if (!c->event_pool.empty()) {
                hipEvent_t e = c->event_pool.back();
                c->event_pool.pop_back();
                return e;
            }
            hipEvent_t e = nullptr;
            return hipEventCreate(&e) == hipSuccess ? e : nullptr;
        };
        a = take();
        if (a)
            (void)hipEventRecord(a, c->stream);
    }
    ~Scope32()
    {
        if (!a)
            return;
        hipEvent_t b = nullptr;
        if (!c->event_pool.empty()) {
            b = c->event_pool.back();
            c->event_pool.pop_back();
        } else if (hipEventCreate(&b) != hipSuccess)
            b = nullptr;
        if (b) {
            (void)hipEventRecord(b, c->stream);
            c->pending.push_back({k, a, b});
        } else {
            c->event_pool.push_back(a);
        }
    }
};
static void resolve_timers32(mg3d32_ctx *ctx) /* behind a stream synchronisation */
{
    for (auto &p : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->kt[p.k].calls++;
            ctx->kt[p.k].seconds += ms * 1e-3;
        }
        ctx->event_pool.push_back(p.a);
        ctx->event_pool.push_back(p.b);
    }
    ctx->pending.clear();
}
extern "C" const char *mg3d32_kernel_name(int k) { return k >= 0 && k < MG3D32_NUM_KERNELS ? kKernel32Names[k] : nullptr; }
extern "C" int mg3d32_timing_enable(mg3d32_ctx *ctx, int on)
{
    if (!ctx)
        return fail(MG3D_ERR_ARG, "mg3d32_timing_enable: NULL");
    ctx->timing = on != 0;
    if (on)
        for (auto &k : ctx->kt)
            k.calls = 0, k.seconds = 0.;
    return MG3D_OK;
}
extern "C" int mg3d32_kernel_time_get(mg3d32_ctx *ctx, int kernel, int *launches, double *seconds)
{
    if (!ctx || kernel < 0 || kernel >= MG3D32_NUM_KERNELS)
        return fail(MG3D_ERR_ARG, "mg3d32_kernel_time_get: bad arguments");
    if (launches)
        *launches = ctx->kt[kernel].calls;
    if (seconds)
        *seconds = ctx->kt[kernel].seconds;
    return MG3D_OK;
}

/* ---------------------------------------------------------------------------------------- operators */
static int chunk_for(int planes, long long blocks_per_plane, long long want = 4096)
{
    int chunk = 64;
    while (chunk > 4 && blocks_per_plane * ((planes + chunk - 1) / chunk) < want)
        chunk /= 2;
    return chunk;
}


/* `iters` sweeps.  norm_slot >= 0: the caller wants ||d - A u|| of the result in sumsq[norm_slot]; returns true when
 * the last launch delivered it (paired sweep with the residual as third stage), false when a residual launch
 * still has to follow.  prolong_first: u += P(u of the next coarser level) before the first sweep (mg_3d.h:1331),
 * folded into the first paired launch when there is one. */
bool e32_jacobi(mg3d32_ctx *ctx, int level, int iters, int norm_slot, bool prolong_first, int tap_slot)
{
    Level32 &l = ctx->lv[level];
    const int gx = ((l.g.nk + 3) / 4 + 63) / 64, gy = (l.g.nj + 3) / 4;
    const int chunk = chunk_for(l.g.ni, (long long)gx * gy);
    int it = 0;
    bool normed = false;
    const bool no_pairs = ctx->no_pairs, no_fuse = ctx->no_fuse;
    auto swap = [&]() {
        float *t = l.f[MG3D_U];
        l.f[MG3D_U] = l.alt;
        l.alt = t;
    };
    const bool pairs = !no_pairs && l.g.N >= 33 && iters >= 2;
    if (prolong_first && !(pairs && !no_fuse))
        e32_prolong(ctx, level);
    if (pairs) { /* sweeps in pairs: one pass over HBM for two */
        const int px = (l.g.nk + J2_OUT_COLS - 1) / J2_OUT_COLS;
        for (; it + 2 <= iters; it += 2) {
            const bool with_norm = norm_slot >= 0 && !no_fuse && it + 2 == iters;
            const bool with_pro = prolong_first && !no_fuse && it == 0;
            /* tap_slot >= 0: ||d - A u|| of the field as it ENTERS this stage goes to sumsq[tap_slot] (the caller has
             * checked e32_can_carry: the first launch is a plain pair) */
            const bool with_tap = tap_slot >= 0 && it == 0 && !with_norm && !with_pro;
            const int py = (l.g.nj + (with_norm ? J2N_OUT_ROWS : J2_OUT_ROWS) - 1) / (with_norm ? J2N_OUT_ROWS : J2_OUT_ROWS);
            int ch = 128;
            while (ch > 8 && (long long)px * py * ((l.g.ni + ch - 1) / ch) < 1024)
                ch /= 2;
            while ((with_norm || with_tap) && (long long)px * py * ((l.g.ni + ch - 1) / ch) > MG3D_MAX_PARTIALS)
                ch *= 2;
            const dim3 grid(px, py, (l.g.ni + ch - 1) / ch), block(64, J2_ROWS, 1);
            const Geom gc = with_pro ? ctx->lv[level - 1].g : l.g;
            const float *ec = with_pro ? ctx->lv[level - 1].f[MG3D_U] : nullptr;
            double *part = (with_norm || with_tap) ? ctx->partials : nullptr;
            Scope32 kt(ctx, level, with_tap ? MG3D32_K_PAIR_TAP : with_norm && with_pro ? MG3D32_K_PRO_PAIR_NORM : with_norm ? MG3D32_K_PAIR_NORM
                                   : with_pro ? MG3D32_K_PRO_PAIR : MG3D32_K_PAIR);
#define J2_LAUNCH(NORM, PRO, ...)                                                                                   \
    hipLaunchKernelGGL((jacobi32x2_kernel<NORM, PRO, ##__VA_ARGS__>), grid, block, 0, ctx->stream, l.g, l.f[MG3D_U], l.f[MG3D_D], \
                       l.alt, l.hSq, 1.0f / 6.0f, ctx->omega, l.invHsq, part, ch, gc, ec, l.own_lo, l.own_hi)
            if (with_tap)
                J2_LAUNCH(false, false, true);
            else if (with_norm && with_pro)
                J2_LAUNCH(true, true);
            else if (with_norm)
                J2_LAUNCH(true, false);
            else if (with_pro)
                J2_LAUNCH(false, true);
            else
                J2_LAUNCH(false, false);
#undef J2_LAUNCH
            if (with_norm) {
                k_fold(ctx->partials, (int)(grid.x * grid.y * grid.z), ctx->sumsq + norm_slot, ctx->stream);
                normed = true;
            }
            if (with_tap)
                k_fold(ctx->partials, (int)(grid.x * grid.y * grid.z), ctx->sumsq + tap_slot, ctx->stream);
            swap();
        }
    }
    for (; it < iters; it++) {
        Scope32 kt(ctx, level, MG3D32_K_SWEEP1);
        hipLaunchKernelGGL(jacobi32_kernel, dim3(gx, gy, (l.g.ni + chunk - 1) / chunk), dim3(64, 4, 1), 0, ctx->stream,
                           l.g, l.f[MG3D_U], l.f[MG3D_D], l.alt, l.hSq, 1.0f / 6.0f, ctx->omega, chunk);
        swap();
        normed = false;
    }
    return normed;
}

void e32_residual(mg3d32_ctx *ctx, int level, bool store, int slot)
{
    Level32 &l = ctx->lv[level];
    /* planes to form: the owned ones plus one on either side (a stored r is restricted from there), clipped to the
     * updatable planes; the norm takes the owned ones */
    int p_lo = l.own_lo - 1 > 1 ? l.own_lo - 1 : 1, p_hi = l.own_hi + 1 < l.g.ni - 1 ? l.own_hi + 1 : l.g.ni - 1;
    if (p_hi > l.g.N - 1 - l.g.ig0)
        p_hi = l.g.N - 1 - l.g.ig0;
    if (l.g.N < 3 || p_hi <= p_lo) {
        (void)hipMemsetAsync(ctx->sumsq + slot, 0, sizeof(double), ctx->stream);
        return;
    }
    const int gx = ((l.g.nk + 3) / 4 + 63) / 64, gy = (l.g.nj + 3) / 4, np = p_hi - p_lo;
    int chunk = chunk_for(np, (long long)gx * gy);
    while ((long long)gx * gy * ((np + chunk - 1) / chunk) > MG3D_MAX_PARTIALS)
        chunk *= 2;
    const int gz = (np + chunk - 1) / chunk;
    Scope32 kt(ctx, level, MG3D32_K_RESIDUAL);
    hipLaunchKernelGGL(residual32_kernel, dim3(gx, gy, gz), dim3(64, 4, 1), 0, ctx->stream, l.g, l.f[MG3D_U],
                       l.f[MG3D_D], l.invHsq, store ? l.f[MG3D_R] : nullptr, ctx->partials, chunk, p_lo, p_hi, l.own_lo,
                       l.own_hi);
    k_fold(ctx->partials, gx * gy * gz, ctx->sumsq + slot, ctx->stream);
}

void e32_restrict(mg3d32_ctx *ctx, int level)
{
    Level32 &lf = ctx->lv[level], &lc = ctx->lv[level - 1];
    hipLaunchKernelGGL(restrict32_kernel, dim3((lc.g.nk + 63) / 64, (lc.g.nj + 3) / 4, lc.own_hi - lc.own_lo),
                       dim3(64, 4, 1), 0, ctx->stream, lf.g, lf.f[MG3D_R], lc.g, lc.f[MG3D_D], lc.own_lo);
}

void e32_residual_restrict(mg3d32_ctx *ctx, int level, int c_lo, int c_hi)
{
    Level32 &lf = ctx->lv[level], &lc = ctx->lv[level - 1];
    if (c_lo < 0) { /* the coarse planes this context owns */
        c_lo = lc.own_lo;
        c_hi = lc.own_hi;
    }
    /* the 27-point sum on the interior coarse planes, injection on the faces (mg_3d.h:879-958) */
    const int i_lo = c_lo > 1 - lc.g.ig0 ? c_lo : 1 - lc.g.ig0, i_hi = c_hi < lc.g.N - 1 - lc.g.ig0 ? c_hi : lc.g.N - 1 - lc.g.ig0;
    const int px = (lf.g.nk + J2_OUT_COLS - 1) / J2_OUT_COLS, py = (lf.g.nj + J2_OUT_ROWS - 1) / J2_OUT_ROWS;
    if (i_hi > i_lo) {
        const int nc = i_hi - i_lo;
        int cch = 64; /* coarse planes per block */
        while (cch > 4 && (long long)px * py * ((nc + cch - 1) / cch) < 1024)
            cch /= 2;
        Scope32 kt(ctx, level, MG3D32_K_RESIDUAL_RESTRICT);
        hipLaunchKernelGGL(residual_restrict32_kernel, dim3(px, py, (nc + cch - 1) / cch), dim3(64, J2_ROWS, 1), 0,
                           ctx->stream, lf.g, lf.f[MG3D_U], lf.f[MG3D_D], lf.invHsq, lc.g, lc.f[MG3D_D], cch, i_lo, i_hi);
    }
    const int m = lc.g.N;
    hipLaunchKernelGGL(restrict32_faces_kernel, dim3((m + 63) / 64, (m + 3) / 4, 6), dim3(64, 4, 1), 0, ctx->stream,
                       lf.g, lf.f[MG3D_R], lc.g, lc.f[MG3D_D], c_lo, c_hi);
}

void e32_prolong(mg3d32_ctx *ctx, int level)
{
    Level32 &lf = ctx->lv[level], &lc = ctx->lv[level - 1];
    const int gx = ((lf.g.nk + 1) / 2 + 63) / 64, gy = ((lf.g.nj + 1) / 2 + 3) / 4;
    const int chunk = chunk_for(lf.g.ni, (long long)gx * gy, 2048);
    Scope32 kt(ctx, level, MG3D32_K_PROLONG);
    hipLaunchKernelGGL(prolong32_kernel, dim3(gx, gy, (lf.g.ni + chunk - 1) / chunk), dim3(64, 4, 1), 0, ctx->stream,
                       lc.g, lc.f[MG3D_U], lf.g, lf.f[MG3D_U], chunk);
}

int e32_coarse_solve(mg3d32_ctx *ctx)
{
    Level32 &l0 = ctx->lv[0];
    Level &c0 = ctx->coarse64->lv[0];
    const dim3 grid(l0.g.nj, l0.g.ni);
    hipLaunchKernelGGL(widen32_kernel, grid, dim3(64), 0, ctx->stream, l0.g, l0.f[MG3D_D], c0.g, c0.f[MG3D_D]);
    CHK(mg3d_coarse_solve(ctx->coarse64));
    hipLaunchKernelGGL(narrow32_kernel, grid, dim3(64), 0, ctx->stream, c0.g, c0.f[MG3D_U], l0.g, l0.f[MG3D_U]);
    return MG3D_OK;
}

void e32_fill_boundary(mg3d32_ctx *ctx, int field, int level)
{
    Level32 &l = ctx->lv[level];
    hipLaunchKernelGGL(fill_boundary32_kernel, dim3((l.g.nk + 63) / 64, (l.g.nj + 3) / 4, l.g.ni), dim3(64, 4, 1), 0,
                       ctx->stream, l.g, l.f[field], l.hd);
}

static int launch_ok32(const char *who)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(MG3D_ERR_HIP, "%s: kernel launch failed: %s", who, hipGetErrorString(e));
    return MG3D_OK;
}

/* Carried cycles, the Jacobi form.  A Jacobi sweep reads only old values, so the first pre-smoothing sweep of a cycle
 * gathers, for every point, exactly the six neighbours the residual of the field it starts from is made of: the norm of
 * cycle n (:1354) is a tap on cycle n+1's first launch (jacobi32x2_kernel<.,.,TAP>) instead of a third stage on cycle n's
 * last one -- whose two-stage form (prolongation + two sweeps) keeps two more rows of its tile and one plane less in
 * flight.  Nothing about u changes; only the norm arrives one launch later, inside the same mg3d32_vcycles call (the
 * last cycle of a batch forms its own).  V(2,2) with the paired, fused launches only; mg3d32_set_option("carry", 0) (or MG3D_F32_NO_CARRY=1 when the context is created) switches it off. */
bool e32_can_carry(const mg3d32_ctx *ctx)
{
    if (ctx->no_carry)
        return false;
    return ctx->iters == 2 && !ctx->no_pairs && !ctx->no_fuse && ctx->L >= 2 && ctx->lv[ctx->L - 1].g.N >= 33;
}

/* the V-cycle of mg_3d.h:1242-1362 with the Jacobi smoother; the level's norm lands in sumsq[slot] */
int e32_vcycle(mg3d32_ctx *ctx, int q, int slot, bool carry_out, int tap_slot)
{
    if (q == 0)
        return e32_coarse_solve(ctx);
    e32_jacobi(ctx, q, ctx->iters, -1, false, tap_slot);  /* :1282 (+ the norm of the cycle before, :1354) */
    if (!ctx->no_fuse && ctx->lv[q].g.N >= 33) {
        e32_residual_restrict(ctx, q);                    /* :1294 + :1310, r not stored */
    } else {
        e32_residual(ctx, q, true, ctx->sumsq_slots - 1); /* :1294 (its norm is dropped) */
        e32_restrict(ctx, q);                             /* :1310 */
    }
    Level32 &lc = ctx->lv[q - 1];
    (void)hipMemsetAsync(lc.f[MG3D_U], 0, lc.elems * sizeof(float), ctx->stream); /* :1258 */
    CHK(e32_vcycle(ctx, q - 1, ctx->sumsq_slots - 1));    /* :1321 */
    /* :1331 + :1341 + :1354; below the top level the reference drops the norm (:1320), so it is not formed there */
    const bool top = q == ctx->L - 1;
    if (ctx->iters == 0)
        e32_prolong(ctx, q);
    if (top && carry_out) { /* the norm is the next cycle's first launch's to form */
        e32_jacobi(ctx, q, ctx->iters, -1, true);
        return MG3D_OK;
    }
    if (!e32_jacobi(ctx, q, ctx->iters, top ? slot : -1, ctx->iters > 0) && top)
        e32_residual(ctx, q, false, slot);
    return MG3D_OK;
}

extern "C" int mg3d32_smooth(mg3d32_ctx *ctx, int level, int iters)
{
    CHK(check32(ctx, 0, level, "mg3d32_smooth"));
    if (iters < 0)
        return fail(MG3D_ERR_ARG, "mg3d32_smooth: negative sweep count");
    if (ctx->lv[level].g.N >= 3)
        e32_jacobi(ctx, level, iters);
    return launch_ok32("mg3d32_smooth");
}

static int norm_out(mg3d32_ctx *ctx, int slot, double *norm)
{
    if (norm) {
        HIPCHK(hipMemcpyAsync(ctx->h_sumsq, ctx->sumsq + slot, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        *norm = sqrt(ctx->h_sumsq[0]);
    }
    return MG3D_OK;
}

extern "C" int mg3d32_residual(mg3d32_ctx *ctx, int level, int store, double *norm)
{
    CHK(check32(ctx, 0, level, "mg3d32_residual"));
    e32_residual(ctx, level, store != 0, 0);
    CHK(launch_ok32("mg3d32_residual"));
    return norm_out(ctx, 0, norm);
}

extern "C" int mg3d32_restrict(mg3d32_ctx *ctx, int level)
{
    CHK(check32(ctx, 0, level, "mg3d32_restrict"));
    if (level < 1)
        return fail(MG3D_ERR_ARG, "mg3d32_restrict: no coarser level");
    e32_restrict(ctx, level);
    return launch_ok32("mg3d32_restrict");
}

extern "C" int mg3d32_prolong(mg3d32_ctx *ctx, int level)
{
    CHK(check32(ctx, 0, level, "mg3d32_prolong"));
    if (level < 1)
        return fail(MG3D_ERR_ARG, "mg3d32_prolong: no coarser level");
    e32_prolong(ctx, level);
    return launch_ok32("mg3d32_prolong");
}

extern "C" int mg3d32_coarse_solve(mg3d32_ctx *ctx)
{
    if (!ctx)
        return fail(MG3D_ERR_ARG, "mg3d32_coarse_solve: NULL");
    CHK(e32_coarse_solve(ctx));
    return launch_ok32("mg3d32_coarse_solve");
}

extern "C" int mg3d32_fill_boundary(mg3d32_ctx *ctx, int field, int level)
{
    CHK(check32(ctx, field, level, "mg3d32_fill_boundary"));
    e32_fill_boundary(ctx, field, level);
    return launch_ok32("mg3d32_fill_boundary");
}

extern "C" int mg3d32_zero(mg3d32_ctx *ctx, int field, int level)
{
    CHK(check32(ctx, field, level, "mg3d32_zero"));
    HIPCHK(hipMemsetAsync(ctx->lv[level].f[field], 0, ctx->lv[level].elems * sizeof(float), ctx->stream));
    return MG3D_OK;
}

/* `count` V-cycles from the finest level, device resident; norms[c] = ||d - A u|| after cycle c */
extern "C" int mg3d32_vcycles(mg3d32_ctx *ctx, int count, double *norms)
{
    if (!ctx || count < 0)
        return fail(MG3D_ERR_ARG, "mg3d32_vcycles: bad arguments");
    const int slots = ctx->sumsq_slots - 1;
    for (int done = 0; done < count;) {
        const int nb = count - done < slots ? count - done : slots;
        const bool can = e32_can_carry(ctx);
        for (int c = 0; c < nb; c++) /* every cycle of the batch but its last leaves its norm to the next one's first launch */
            CHK(e32_vcycle(ctx, ctx->L - 1, c, can && c + 1 < nb, (can && c > 0) ? c - 1 : -1));
        CHK(launch_ok32("mg3d32_vcycles"));
        HIPCHK(hipMemcpyAsync(ctx->h_sumsq, ctx->sumsq, nb * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        resolve_timers32(ctx);
        if (norms)
            for (int c = 0; c < nb; c++)
                norms[done + c] = sqrt(ctx->h_sumsq[c]);
        done += nb;
    }
    return MG3D_OK;
}

/* F-cycle start (FMG), mg_dirichlet_analytic.c:771-806: the right-hand sides d of ALL levels are the caller's */
extern "C" int mg3d32_fmg_initialize(mg3d32_ctx *ctx)
{
    if (!ctx)
        return fail(MG3D_ERR_ARG, "mg3d32_fmg_initialize: NULL");
    e32_fill_boundary(ctx, MG3D_U, 0);     /* :780 */
    CHK(e32_coarse_solve(ctx));            /* :783 */
    for (int l = 1; l < ctx->L; l++) {
        e32_prolong(ctx, l);               /* :795 */
        e32_fill_boundary(ctx, MG3D_U, l); /* :798 */
        Level32 &lc = ctx->lv[l - 1];
        HIPCHK(hipMemsetAsync(lc.f[MG3D_U], 0, lc.elems * sizeof(float), ctx->stream)); /* :801 */
        CHK(e32_vcycle(ctx, l, ctx->sumsq_slots - 1));                                    /* :804 */
    }
    return launch_ok32("mg3d32_fmg_initialize");
}
