// Microbenchmark: latency of DEPENDENT f64 operations for a lone wave on gfx950 (s_memtime ticks).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(64) void k(double *out, double a, double b, int iters, long long *cyc)
{
    double x = a + threadIdx.x * 1e-9, y = b;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (OP == 0)
                x = __builtin_fma(x, y, a); // dependent FMA
            else if (OP == 1)
                x = x * y; // dependent mul
            else if (OP == 2)
                x = a / x + b; // dependent IEEE division (+ add)
            else if (OP == 3)
                x = __builtin_amdgcn_rcp(x) + b; // dependent v_rcp_f64 + add
            else if (OP == 4)
                x = x + y;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0)
        *cyc = t1 - t0;
}

template <int OP> void run(const char *name, int per)
{
    double *out;
    long long *cyc, h;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, 1.0000001, 0.9999999, 10, cyc);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, 1.0000001, 0.9999999, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %.1f ns per op (%d ops/iter-step), counter %.1f ticks per op\n", name,
           ms * 1e6 / (iters * 16.0 * per), per, (double)h / (iters * 16.0 * per));
}

int main()
{
    run<0>("dependent v_fma_f64", 1);
    run<1>("dependent v_mul_f64", 1);
    run<4>("dependent v_add_f64", 1);
    run<3>("dependent v_rcp_f64 + add", 1);
    run<2>("dependent f64 division + add", 1);
    return 0;
}
