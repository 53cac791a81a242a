// slab_probe.hip -- does the Infinity Cache (256 MiB) pay for running the finest level's launches slab by slab?
// (1) raw read bandwidth of a streaming sum over working sets from 32 MB to 2 GB, repeated passes (resident vs not)
// (2) the real sweep kernels of libmg3d.so on i-windows of a 513^3 level:
//       full launches  A = 4 passes, B = residual + restriction
//       windowed       A(k) B(k) A(k+1) B(k+1) ... with W planes per window (B lags 3 planes)
//       warm           A on one window repeatedly (inputs resident)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I multigrid_parallel_amd/csrc -I include tools/slab_probe.hip \
//        -L multigrid_parallel_amd/lib -lmg3d -Wl,-rpath,$PWD/multigrid_parallel_amd/lib -o gpurun_out/slab_probe
#include "mg3d_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__global__ void __launch_bounds__(256) rd_sum(const double2 *__restrict__ a, size_t n2, int passes, double *out)
{
    double acc = 0;
    for (int p = 0; p < passes; p++)
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
            double2 x = a[i];
            acc += x.x + x.y;
        }
    if (acc == 1.2345e300)
        out[0] = acc;
}
__global__ void __launch_bounds__(256) cp(const double2 *__restrict__ a, double2 *__restrict__ o, size_t n2)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256)
        o[i] = a[i];
}

static Geom geom(int N)
{
    Geom g;
    g.ni = g.nj = g.nk = N;
    g.pitch = mg3d_pitch_for(N);
    g.plane = (long long)g.pitch * N;
    g.ig0 = 0;
    g.N = N;
    return g;
}

int main(int argc, char **argv)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto timed = [&](auto f, int reps) {
        f();
        CK(hipStreamSynchronize(s));
        float best = 1e9;
        for (int r = 0; r < reps; r++) {
            CK(hipEventRecord(e0, s));
            f();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best)
                best = ms;
        }
        return best;
    };
    const int N = 513;
    Geom g = geom(N), gc = geom(257);
    const size_t n = (size_t)g.plane * N, nc = (size_t)gc.plane * 257;
    double *u, *d, *alt, *dc, *partials;
    CK(hipMalloc(&u, n * 8));
    CK(hipMalloc(&d, n * 8));
    CK(hipMalloc(&alt, n * 8));
    CK(hipMalloc(&dc, nc * 8));
    CK(hipMalloc(&partials, MG3D_MAX_PARTIALS * 8));
    CK(hipMemset(u, 0, n * 8));
    CK(hipMemset(d, 0, n * 8));
    CK(hipMemset(alt, 0, n * 8));
    CK(hipMemset(dc, 0, nc * 8));
    /* ---- (1) raw read bandwidth by working set */
    for (size_t mb : {32, 64, 128, 192, 256, 384, 512, 1024}) {
        const size_t n2 = mb * 1024 * 1024 / 16;
        const int passes = (int)(4096 / mb) < 2 ? 2 : (int)(4096 / mb);
        float ms = timed([&] { hipLaunchKernelGGL(rd_sum, dim3(2048), dim3(256), 0, s, (const double2 *)u, n2, passes, partials); }, 3);
        printf("rd_sum  %5zu MB x %3d passes: %8.3f ms  %7.1f GB/s\n", mb, passes, ms, (double)mb * 1048576.0 * passes / ms / 1e6);
    }
    for (size_t mb : {32, 64, 128, 256, 1024}) { /* copy: back-to-back launches over the same buffers */
        const size_t n2 = mb * 1024 * 1024 / 16;
        float ms = timed([&] { for (int r = 0; r < 4; r++) hipLaunchKernelGGL(cp, dim3(2048), dim3(256), 0, s, (const double2 *)u, (double2 *)alt, n2); }, 3);
        printf("copy    %5zu MB x 4 launches: %8.3f ms  %7.1f GB/s (r+w)\n", mb, ms, 2.0 * mb * 1048576.0 * 4 / ms / 1e6);
    }
    /* ---- (2) sweep kernels on windows */
    const double h = 1.0 / (N - 1);
    auto A = [&](int lo, int hi) { k_sweep(g, u, d, alt, nullptr, nullptr, MG3D_MAX_PARTIALS, h, 4, 1, false, s, 0, -1, nullptr, nullptr, -1, -1, nullptr, nullptr, lo, hi); };
    auto B = [&](int lo, int hi) { k_sweep(g, alt, d, nullptr, nullptr, nullptr, MG3D_MAX_PARTIALS, h, 0, 1, true, s, 0, -1, &gc, dc, -1, -1, nullptr, nullptr, lo, hi); };
    auto C2 = [&](int lo, int hi) { k_sweep(g, u, d, alt, nullptr, nullptr, MG3D_MAX_PARTIALS, h, 2, 0, false, s, 0, -1, nullptr, nullptr, -1, -1, nullptr, nullptr, lo, hi); };
    auto D2 = [&](int lo, int hi) { k_sweep(g, alt, d, u, nullptr, partials, MG3D_MAX_PARTIALS, h, 2, 0, true, s, 0, -1, nullptr, nullptr, -1, -1, nullptr, nullptr, lo, hi); };
    float tA = timed([&] { A(0, N); }, 5), tB = timed([&] { B(0, N); }, 5);
    float tC = timed([&] { C2(0, N); }, 5), tD = timed([&] { D2(0, N); }, 5);
    printf("full: A(4 passes) %.3f ms  B(res+restr) %.3f ms  C(2 passes) %.3f ms  D(2 passes+norm) %.3f ms  A+B %.3f  C+D %.3f  all %.3f\n",
           tA, tB, tC, tD, tA + tB, tC + tD, tA + tB + tC + tD);
    for (int W : {8, 16, 24, 32, 48, 64}) {
        for (const char *ci : {"0", "8", "16", "32"}) {
            if (atoi(ci) > W)
                continue;
            setenv("MG3D_SWEEP_CI", ci, 1);
            auto clampw = [&](int x) { return x < 0 ? 0 : x > N ? N : x; };
            float tAw = timed([&] { for (int lo = 0; lo < N; lo += W) A(lo, clampw(lo + W)); }, 3);
            float tAB = timed([&] { /* B covers what A has finished minus 3 planes */
                int prev = 0;
                for (int lo = 0; lo < N; lo += W) {
                    A(lo, clampw(lo + W));
                    const int done = clampw(lo + W) >= N ? N : clampw(lo + W) - 3;
                    if (done > prev) { B(prev, done); prev = done; }
                }
            }, 3);
            float tCD = timed([&] {
                int prev = 0;
                for (int lo = 0; lo < N; lo += W) {
                    C2(lo, clampw(lo + W));
                    const int done = clampw(lo + W) >= N ? N : clampw(lo + W) - 3;
                    if (done > prev) { D2(prev, done); prev = done; }
                }
            }, 3);
            float tall = timed([&] { /* C D A' B' chained: each lags 3-5 planes behind the one before */
                int pd = 0, pa = 0, pb = 0;
                for (int lo = 0; lo < N; lo += W) {
                    const int hiC = clampw(lo + W);
                    C2(lo, hiC);
                    const int dD = hiC >= N ? N : hiC - 3;
                    if (dD > pd) { D2(pd, dD); pd = dD; }
                    const int dA = pd >= N ? N : pd - 5;
                    if (dA > pa) { A(pa, dA); pa = dA; }
                    const int dB = pa >= N ? N : pa - 3;
                    if (dB > pb) { B(pb, dB); pb = dB; }
                }
            }, 3);
            float tw = timed([&] { for (int r = 0; r < 8; r++) A(256, 256 + W); }, 3) / 8;
            printf("W %2d CI %2s: A windows %.3f ms | A,B interleaved %.3f | C,D interleaved %.3f | C,D,A,B chained %.3f | warm A per window %.4f ms -> x%d = %.3f\n",
                   W, ci, tAw, tAB, tCD, tall, tw, (N + W - 1) / W, tw * ((N + W - 1) / W));
            fflush(stdout);
        }
    }
    unsetenv("MG3D_SWEEP_CI");
    return 0;
}
