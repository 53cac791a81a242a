/* mg3d_lu_dev.h -- device-side pieces of the coarsest direct solve (gauss_elim.h:31-60), shared by the stand-alone solve
 * kernels (mg3d_kernels.hip) and the single-workgroup coarse cycle (mg3d_tiny.hip).  Not installed. */
#ifndef MG3D_LU_DEV_H
#define MG3D_LU_DEV_H

#include "mg3d_internal.h"

#ifndef WAVE
#define WAVE 64
#endif

__device__ __forceinline__ double readlane_f64(double x, int lane_uniform)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane_uniform);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane_uniform);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

/* x = num / d, correctly rounded, with the reciprocal work taken off the dependency chain.  The compiler's
 * own fp64 division (div_scale, rcp, two Newton steps, div_fmas, div_fixup) depends on the numerator from its
 * first instruction: a dozen dependent operations per back-substitution step.  Here r = RN(1/d) comes from the
 * host and the quotient is refined twice, q <- q + (num - d*q)*r with exact FMA residuals: after the first
 * step q is within one ulp, and for a faithful q and a correctly rounded reciprocal the second step delivers
 * RN(num/d) (Markstein's theorem; tests/c/div_check.c compares 10^8 operand pairs with the hardware quotient).
 * Numerators outside a wide safe exponent window (and zeros, infinities, NaNs) take the ordinary division
 * (`num` is wave-uniform, so that branch is too); a diagonal outside its window disables FAST altogether. */
template <bool FAST>
__device__ __forceinline__ double lu_div(double num, double d, double r)
{
    if (!FAST)
        return num / d;
    /* 2^-498 <= |num| < 2^499; the host vouches for |d| in [2^-460, 2^460] on the whole diagonal */
    const unsigned e = (unsigned)(__double_as_longlong(num) >> 52) & 0x7ffu;
    double q = num * r;
    double rem = __builtin_fma(-d, q, num);
    q = __builtin_fma(rem, r, q);
    rem = __builtin_fma(-d, q, num);
    q = __builtin_fma(rem, r, q);
    /* +0 (every boundary unknown of a V-cycle's coarse right-hand side) also comes out right: +-0 by sign of d */
    if (__builtin_expect(e - 525u >= 997u && __double_as_longlong(num) != 0ll, 0))
        q = num / d;
    return q;
}

/* One substitution pass of the single-wave solve.  Lane l owns the rows == l (mod 64); acc[0] is the
 * running sum of the row it finalises next, acc[1..] of the rows 64, 128, .. further on.  At step j
 * the owner lane (j & 63) turns its acc[0] into x[j]; v_readlane broadcasts it; every lane then adds
 * factor * x[j] to each of its sums (the owner first rotates its sums by one).  The factors come
 * pre-rotated so that lane l always reads element l of the step's column: no index arithmetic and no
 * predicates on the dependency chain  acc -> sub (-> div) -> readlane -> mul -> add.
 * Columns, rhs[j] and the diagonal are fetched U steps ahead (they do not depend on the solution). */
template <int R, bool FWD, bool FAST>
__device__ __forceinline__ void lu_wave_pass(const LuBand &lu, int lane, const double *rhs, double *out,
                                             const double *dg)
{
    constexpr int U = 8;
    const int n = lu.n;
    const double *cols = (FWD ? lu.lrot : lu.urot) + lane;
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; r++)
        acc[r] = 0.;
    double nxt[U][R], cur[U][R], rj[U], dj[U], rdj[U];
    auto fetch = [&](int step, double(&dst)[R]) {
        int j = FWD ? step : n - 1 - step;
        j = j < 0 ? 0 : (j >= n ? n - 1 : j); /* steps past the end: any valid column, never used */
#pragma unroll
        for (int r = 0; r < R; r++)
            dst[r] = cols[(long long)j * (64 * R) + 64 * r];
    };
#pragma unroll
    for (int u = 0; u < U; u++)
        fetch(u, nxt[u]);
    const int nch = (n + U - 1) / U;
    for (int c = 0; c < nch; c++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int r = 0; r < R; r++)
                cur[u][r] = nxt[u][r];
            const int step = c * U + u;
            const int j = FWD ? step : n - 1 - step;
            const bool in = step < n;
            rj[u] = in ? rhs[j] : 0.; /* uniform address: LDS broadcast */
            dj[u] = (!FWD && in) ? dg[j] : 1.;
            rdj[u] = (!FWD && in) ? dg[n + j] : 1.;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            fetch((c + 1) * U + u, nxt[u]);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int step = c * U + u;
            if (step >= n)
                break;
            const int j = FWD ? step : n - 1 - step;
            const int owner = j & 63;
            double xj = readlane_f64(rj[u] - acc[0], owner); /* gauss_elim.h:46 / :57 */
            if (!FWD)
                xj = lu_div<FAST>(xj, dj[u], rdj[u]);
            const bool own = lane == owner;
            if (own) /* one lane: 64 lanes storing to the same LDS word are serialised by the LDS */
                out[j] = xj;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const double base = own ? (r + 1 < R ? acc[r + 1] : 0.) : acc[r];
                acc[r] = base + cur[u][r] * xj; /* sum += LU[i][j]*x[j], gauss_elim.h:41,55 */
            }
        }
    }
}

/* Streamed variant of the single-wave solve: the shipped one for narrow bands.  Same substitution, same
 * dependency chain (sub -> readlane -> [divide] -> mul -> add); what changes is everything around the chain.
 * tools/lu_step_probe.hip prices a step of the plain pass at ~220 cycles of which the chain is 48: every LDS
 * or global access inside the step costs ~45 cycles of issue.  Here a step touches no memory at all:
 *   - the system is padded to a multiple of 64 unknowns (identity rows) and walked in chunks of 64 steps, in
 *     which every lane owns exactly one unknown: its right-hand side, diagonal and reciprocal sit in
 *     registers (one coalesced LDS read per chunk), its result leaves by one coalesced LDS write per chunk;
 *   - the factors of 32 steps at a time are read from LDS into registers before the steps run;
 *   - two loader waves of the same workgroup stream the factors -- stored in consumption order, forward
 *     steps then backward steps -- from HBM into a two-slot LDS ring, one chunk per slot, two chunks ahead
 *     (one in their registers, one in LDS); one workgroup barrier per chunk hands a filled slot to the solver
 *     wave and a used one back.  Three waves, one per SIMD: each may use the whole register file. */
/* lane LANE of `vec` <- the wave-uniform value / zero.  v_writelane_b32 by hand (this compiler has no builtin
 * for it).  The hazard recogniser does not look inside an asm block, so the block carries its own wait states
 * for a scalar source that a VALU instruction (v_readlane) has just written; the lane select is an inline
 * constant, which takes the instruction's other manual-wait-state rule out of play. */
template <int LANE>
__device__ __forceinline__ double writelane_f64(double vec, double uniform)
{
    const long long v = __double_as_longlong(vec), u = __double_as_longlong(uniform);
    int vlo = (int)(v & 0xffffffffll), vhi = (int)(v >> 32);
    const int ulo = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffll));
    const int uhi = __builtin_amdgcn_readfirstlane((int)(u >> 32));
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
        : "+v"(vlo), "+v"(vhi)
        : "s"(ulo), "s"(uhi), "n"(LANE));
    return __longlong_as_double(((long long)vhi << 32) | (unsigned int)vlo);
}

/* Given readlane(f(a, b)) the compiler rewrites it into f(readlane(a), readlane(b)) -- three VALU->SGPR
 * crossings on the dependency chain instead of one.  Passing the value through an empty asm keeps the
 * arithmetic per lane; the readlane itself stays the builtin, so the compiler still places the wait states the
 * instruction needs around it. */
__device__ __forceinline__ double opaque_f64(double x)
{
    const long long b = __double_as_longlong(x);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

/* lane LANE of `keep` <- lane LANE of `x`, nothing else touched: two moves under a one-lane EXEC mask (the wave
 * runs with all lanes on; no scalar register is involved, so no VALU<->SGPR crossing) */
template <int LANE>
__device__ __forceinline__ double keeplane_f64(double keep, double x)
{
    const long long k = __double_as_longlong(keep), b = __double_as_longlong(x);
    int klo = (int)(k & 0xffffffffll), khi = (int)(k >> 32);
    const int xlo = (int)(b & 0xffffffffll), xhi = (int)(b >> 32);
    asm("s_lshl_b64 exec, 1, %4\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\ts_mov_b64 exec, -1"
        : "+v"(klo), "+v"(khi)
        : "v"(xlo), "v"(xhi), "n"(LANE));
    return __longlong_as_double(((long long)khi << 32) | (unsigned int)klo);
}

template <int LANE>
__device__ __forceinline__ double zerolane_f64(double vec)
{
    const long long v = __double_as_longlong(vec);
    int vlo = (int)(v & 0xffffffffll), vhi = (int)(v >> 32);
    asm("v_writelane_b32 %0, 0, %2\n\tv_writelane_b32 %1, 0, %2" : "+v"(vlo), "+v"(vhi) : "n"(LANE));
    return __longlong_as_double(((long long)vhi << 32) | (unsigned int)vlo);
}

template <int R, bool FWD, bool FAST, int HALF, int U0>
__device__ __forceinline__ void lu_stream_steps(const double (&f)[32][R], double myrhs, double mydg, double myrdg,
                                                double &mine, double &mynum, double &A, double &B)
{
    if constexpr (U0 < 32) {
        constexpr int owner = FWD ? HALF * 32 + U0 : 63 - (HALF * 32 + U0); /* j & 63: chunks are 64-aligned in j */
        const double num = myrhs - A; /* gauss_elim.h:46 / :57; the owner's lane holds the real one */
        double xj;
        if (FWD) {
            xj = readlane_f64(num, owner);
        } else if (!FAST) {
            xj = readlane_f64(num, owner) / readlane_f64(mydg, owner);
        } else {
            /* lu_div()'s sequence, but every lane divides its own (mostly meaningless) numerator by its own
             * diagonal and only the quotient is broadcast: one VALU->SGPR crossing on the chain.  The owner's
             * numerator is kept; whether it was inside lu_div's window is checked once per chunk. */
            double q = num * myrdg;
            double rem = __builtin_fma(-mydg, q, num);
            q = __builtin_fma(rem, myrdg, q);
            rem = __builtin_fma(-mydg, q, num);
            q = __builtin_fma(rem, myrdg, q);
            xj = readlane_f64(opaque_f64(q), owner);
            mynum = keeplane_f64<owner>(mynum, num);
        }
        mine = writelane_f64<owner>(mine, xj);
        A = zerolane_f64<owner>(A);
        A = A + f[U0][0] * xj; /* sum += LU[i][j]*x[j], gauss_elim.h:41,55 */
        if (R == 2)
            B = B + f[U0][R - 1] * xj;
        lu_stream_steps<R, FWD, FAST, HALF, U0 + 1>(f, myrhs, mydg, myrdg, mine, mynum, A, B);
    }
}

/* 32 steps of a chunk.  A lane's two running sums are called A and B.  At the start of a chunk A belongs to
 * the row the lane finalises next and B to the row 64 further on; a lane finalises exactly once per chunk (at
 * the step whose unknown it owns), after which B is its next row and A starts from zero for the row 128 on.
 * The factor stream stores each step's pair already in (A's, B's) order for every lane, so the step is
 * select-free:  x = broadcast(rhs - A) [/ diagonal];  owner: result <- x, A <- 0;  A += fa*x;  B += fb*x.
 * At the end of the chunk every lane has switched, and the caller exchanges the names. */
template <int R, bool FWD, bool FAST, int HALF>
__device__ __forceinline__ void lu_stream_half(const double *slot, int lane, double myrhs, double mydg, double myrdg,
                                               double &mine, double &mynum, double &A, double &B)
{
    double f[32][R];
#pragma unroll
    for (int u = 0; u < 32; u++)
#pragma unroll
        for (int r = 0; r < R; r++)
            f[u][r] = slot[((HALF * 32 + u) * 64 + lane) * R + r];
    /* all 32 reads in flight before the first step: left to itself the scheduler sinks each one to its use,
     * behind a full lgkmcnt wait, and the step pays the LDS latency */
    __builtin_amdgcn_sched_barrier(0);
    lu_stream_steps<R, FWD, FAST, HALF, 0>(f, myrhs, mydg, myrdg, mine, mynum, A, B);
}

template <int R, bool FWD, bool FAST>
__device__ __forceinline__ void lu_stream_pass(const double *ring, int first_chunk, int nch, int npad, int lane,
                                               const double *rhs, double *out, const double *dg)
{
    constexpr int CD = 64 * 64 * R;
    double A = 0., B = 0.;
    for (int cc = 0; cc < nch; cc++) {
        const double *slot = ring + ((first_chunk + cc) & 1) * CD;
        const int jl = (FWD ? cc * 64 : npad - 64 * (cc + 1)) + lane; /* the unknown this lane owns in the chunk */
        const double myrhs = rhs[jl], mydg = FWD ? 1. : dg[jl], myrdg = FWD ? 1. : dg[npad + jl];
        double mine = 0., mynum = 1.;
        const double A0 = A, B0 = B;
        lu_stream_half<R, FWD, FAST, 0>(slot, lane, myrhs, mydg, myrdg, mine, mynum, A, B);
        lu_stream_half<R, FWD, FAST, 1>(slot, lane, myrhs, mydg, myrdg, mine, mynum, A, B);
        if (!FWD && FAST) {
            /* a numerator outside lu_div's window (never seen in a V-cycle: they are ordinary numbers or +0):
             * this chunk again, with the ordinary division */
            const unsigned e = (unsigned)(__double_as_longlong(mynum) >> 52) & 0x7ffu;
            const bool bad = e - 525u >= 997u && __double_as_longlong(mynum) != 0ll;
            if (__builtin_expect(__any(bad), 0)) {
                A = A0;
                B = B0;
                lu_stream_half<R, FWD, false, 0>(slot, lane, myrhs, mydg, myrdg, mine, mynum, A, B);
                lu_stream_half<R, FWD, false, 1>(slot, lane, myrhs, mydg, myrdg, mine, mynum, A, B);
            }
        }
        out[jl] = mine;
        if (R == 2) {
            const double t = A;
            A = B;
            B = t;
        }
        __syncthreads();
    }
}

/* The work of waves 0, 1, 2 once b[0..npad) and dg[0..2 npad) sit in LDS (no barrier after filling them yet): waves 1 and 2
 * stream the factors through the two-slot ring, wave 0 substitutes forward into z and backward into b.  Ends with a
 * workgroup barrier: b then holds the solution. */
template <int R>
__device__ __forceinline__ void lu_stream_solve(const LuBand &lu, double *ring, double *b, double *z, const double *dg,
                                                int lane, int wave)
{
    constexpr int CD = 64 * 64 * R;  /* doubles per chunk */
    constexpr int NV = CD / 2 / 128; /* 16-byte vectors per loader lane per chunk (two loader waves) */
    const int npad = lu.npad, nch = npad / 64, T = 2 * nch;
    if (wave == 1 || wave == 2) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const int at = (wave - 1) * 64 + lane;
        const v2d *src = reinterpret_cast<const v2d *>(lu.stream);
        v2d *dst = reinterpret_cast<v2d *>(ring);
        v2d regs[NV];
#pragma unroll
        for (int t = 0; t < NV; t++)
            regs[t] = src[t * 128 + at];
#pragma unroll
        for (int t = 0; t < NV; t++)
            dst[t * 128 + at] = regs[t];
#pragma unroll
        for (int t = 0; t < NV; t++)
            regs[t] = src[CD / 2 + t * 128 + at];
        __syncthreads();
        for (int c = 0; c < T; c++) {
            v2d *d2 = dst + ((c + 1) & 1) * (CD / 2);
#pragma unroll
            for (int t = 0; t < NV; t++)
                d2[t * 128 + at] = regs[t];
            const v2d *s2 = src + (long long)(c + 2) * (CD / 2);
#pragma unroll
            for (int t = 0; t < NV; t++)
                regs[t] = s2[t * 128 + at];
            __syncthreads();
        }
    } else if (wave == 0) {
        __syncthreads();
        lu_stream_pass<R, true, false>(ring, 0, nch, npad, lane, b, z, dg);
        if (lu.fast_div)
            lu_stream_pass<R, false, true>(ring, nch, nch, npad, lane, z, b, dg);
        else
            lu_stream_pass<R, false, false>(ring, nch, nch, npad, lane, z, b, dg);
    } else { /* further waves of a larger workgroup (mg3d_tiny.hip): they only keep the barrier count */
        for (int c = 0; c < T + 1; c++)
            __syncthreads();
    }
    __syncthreads();
}

#endif
