// chain_probe.hip -- dependent-chain latencies seen by ONE wave alone on the GPU (the coarse direct solve's
// situation): fp64 add / mul / fma, v_readlane round trip, LDS broadcast read, and the wave's clock rate.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/chain_probe tools/chain_probe.hip && /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define N 4096
__device__ __forceinline__ double rl(double x, int lane)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int MODE> __global__ void __launch_bounds__(64) chain(double *out, long long *cyc, double a, double b)
{
    double x = a + threadIdx.x * 1e-9, y = b;
    __shared__ double sh[64];
    sh[threadIdx.x] = b;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; i++) {
        if (MODE == 0) x = x + y;                                   // add
        if (MODE == 1) x = x * y;                                   // mul
        if (MODE == 2) x = __builtin_fma(x, y, y);                  // fma
        if (MODE == 3) x = rl(x, i & 63) + y;                       // readlane + add
        if (MODE == 4) { x = x * y; x = x + y; }                    // mul, add (unfused)
        if (MODE == 5) { x = rl(y - x, i & 63); x = x * y; x = x + y; } // one forward step's chain
        if (MODE == 6) x = x + sh[((int)x) & 63];                   // LDS read on chain + add + cvt
        if (MODE == 7) { double q = x * y; double r = __builtin_fma(-b, q, x); q = __builtin_fma(r, y, q); r = __builtin_fma(-b, q, x); x = __builtin_fma(r, y, q); } // lu_div chain
        if (MODE == 8) x = x / y;                                   // hardware division chain
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long w1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
template <int MODE> static void run(const char *name, int ops)
{
    double *out; long long *cyc, h[2];
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 0.9999999);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    int wc_khz = 0; hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
    const double ns = (double)h[1] / (wc_khz * 1e-6);               // wall clock ticks -> ns
    printf("%-34s %7.1f shader-clock cycles/iter  %7.1f ns/iter  (%d dependent ops)  => %.2f GHz  [launch %.1f us]\n", name,
           (double)h[0] / N, ns / N, ops, (double)h[0] / ns, ms * 1e3);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<0>("v_add_f64", 1); run<1>("v_mul_f64", 1); run<2>("v_fma_f64", 1); run<3>("readlane x2 + v_add_f64", 2);
    run<4>("v_mul_f64, v_add_f64", 2); run<5>("sub, readlane, mul, add (fwd step)", 4); run<6>("cvt + ds_read + add", 3);
    run<7>("mul + 4 fma (lu_div)", 5); run<8>("x / y (compiler)", 1);
    return 0;
}
