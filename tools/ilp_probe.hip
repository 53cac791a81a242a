// ilp_probe.hip -- what a dependent fp64 chain costs ONE wave alone on a SIMD (the situation of the one-wave-per-SIMD leg
// kernels, round 4): K independent add chains interleaved by inline asm (the compiler cannot re-serialise them), cycles per add.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/ilp_probe tools/ilp_probe.hip && /tmp/ilp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 2048
template <int K> __global__ void __launch_bounds__(64) chains(double *out, long long *cyc, double y)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    int z = threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; i++) {
        if (K == 1)
            asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1" : "+v"(x0) : "v"(y));
        if (K == 2)
            asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(x0), "+v"(x1) : "v"(y));
        if (K == 4)
            asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));
        if (K == 5) // dependent add, then an independent 32-bit move in between (does any other VALU op fill the gap?)
            asm volatile("v_add_f64 %0, %0, %2\n v_mov_b32 %1, %1\n v_add_f64 %0, %0, %2\n v_mov_b32 %1, %1\n v_add_f64 %0, %0, %2\n v_mov_b32 %1, %1\n v_add_f64 %0, %0, %2\n v_mov_b32 %1, %1"
                         : "+v"(x0), "+v"(z) : "v"(y));
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + z;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int K> static void run(const char *name, int adds)
{
    double *out; long long *cyc, h;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(chains<K>, dim3(1), dim3(64), 0, 0, out, cyc, 1e-9);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-52s %6.2f cycles per v_add_f64\n", name, (double)h / N / adds);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<1>("one chain (every add depends on the one before)", 4);
    run<2>("two independent chains interleaved", 4);
    run<4>("four independent chains interleaved", 4);
    run<5>("one chain, an independent v_mov_b32 between adds", 4);
    return 0;
}
