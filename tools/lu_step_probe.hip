// lu_step_probe.hip -- where do the cycles of one substitution step go?  The single-wave pass of
// csrc/mg3d_kernels.hip (R = 2 rows per lane, n = 729) on synthetic factors, with pieces switched off.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/lu_step_probe tools/lu_step_probe.hip && /tmp/lu_step_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ double rl(double x, int lane)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// bit 0: store x to LDS   bit 1: rotate sums on the owner   bit 2: factors from global (else constants)
// bit 3: rhs from LDS (else constant)   bit 4: step-count guard
template <int F> __global__ void __launch_bounds__(64) pass(const double *cols_g, double *out_g, long long *cyc, int n)
{
    constexpr int R = 2, U = 8;
    __shared__ double rhs[1024], out[1024];
    const int lane = threadIdx.x;
    for (int p = lane; p < 1024; p += 64) { rhs[p] = 1.0 + p * 1e-6; out[p] = 0.; }
    __syncthreads();
    const double *cols = cols_g + lane;
    double acc[R] = {0., 0.};
    double nxt[U][R], cur[U][R], rj[U];
    auto fetch = [&](int step, double(&dst)[R]) {
        int j = step >= n ? n - 1 : step;
        for (int r = 0; r < R; r++) dst[r] = (F & 4) ? cols[(long long)j * 128 + 64 * r] : 1e-3;
    };
    for (int u = 0; u < U; u++) fetch(u, nxt[u]);
    const long long t0 = __builtin_readcyclecounter();
    const int nch = (n + U - 1) / U;
    for (int c = 0; c < nch; c++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            for (int r = 0; r < R; r++) cur[u][r] = nxt[u][r];
            const int j = c * U + u;
            rj[u] = (F & 8) ? ((j < n) ? rhs[j] : 0.) : 1.0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) fetch((c + 1) * U + u, nxt[u]);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = c * U + u;
            if ((F & 16) && j >= n) break;
            const int owner = j & 63;
            const double xj = rl(rj[u] - acc[0], owner);
            const bool own = lane == owner;
            if (F & 1) { if (own) out[j] = xj; }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const double base = (F & 2) ? (own ? (r + 1 < R ? acc[r + 1] : 0.) : acc[r]) : acc[r];
                acc[r] = base + cur[u][r] * xj;
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out_g[lane] = acc[0] + acc[1] + out[lane];
    if (lane == 0) cyc[0] = t1 - t0;
}
template <int F> static void run(const char *what, const double *cols, double *out, long long *cyc, int n)
{
    long long h;
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(pass<F>, dim3(1), dim3(64), 0, 0, cols, out, cyc, n);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-62s %7.1f cycles/step  (%6.1f us per pass at 2.4 GHz)\n", what, (double)h / n, h / 2.4e3);
}
int main()
{
    const int n = 729;
    std::vector<double> hc((size_t)(n + 16) * 128, 1e-3);
    double *cols, *out; long long *cyc;
    hipMalloc(&cols, hc.size() * 8); hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    hipMemcpy(cols, hc.data(), hc.size() * 8, hipMemcpyHostToDevice);
    run<0>("chain only (sub, readlane, 2 mul, 2 add)", cols, out, cyc, n);
    run<2>("+ owner rotates its sums (4 cndmask)", cols, out, cyc, n);
    run<2 + 1>("+ owner stores x to LDS", cols, out, cyc, n);
    run<2 + 1 + 8>("+ rhs[j] read from LDS (broadcast)", cols, out, cyc, n);
    run<2 + 1 + 8 + 4>("+ factors fetched from global 8 steps ahead", cols, out, cyc, n);
    run<2 + 1 + 8 + 4 + 16>("+ step-count guard (the shipped pass)", cols, out, cyc, n);
    run<2 + 8 + 4 + 16>("shipped pass without the LDS store", cols, out, cyc, n);
    run<1 + 8 + 4 + 16>("shipped pass without the rotation", cols, out, cyc, n);
    run<2 + 1 + 8 + 16>("shipped pass with constant factors", cols, out, cyc, n);
    return 0;
}
