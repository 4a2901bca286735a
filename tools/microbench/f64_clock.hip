// Microbenchmark: the CLOCK the chip holds under a long FP64 loop (gfx950).  f64_rate.hip converts times to cycles
// with the nominal 2.4 GHz; this one reads the shader clock (s_memtime) against the constant 100 MHz clock
// (s_memrealtime) inside the kernel, for launches as long as the vocoder kernel (tens of ms).
//   mode 0: v_fma_f64 only, NCH independent chains on pseudo-random operands
//   mode 1: the same with one broadcast ds_read_b128 per eight FMAs (the vocoder's ratio), the value folded in
//   mode 2: v_fma_f64 on all-zero operands (what the inert stage slot and idle lanes execute)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/f64_clock tools/microbench/f64_clock.hip && /tmp/f64_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ unsigned long long g_clk[4];

template <int NCH, int MODE>
__global__ __launch_bounds__(64) void k(double *out, double a, double b, int iters)
{
    __shared__ double2 tab[64];
    tab[threadIdx.x] = make_double2(1e-9 * threadIdx.x, 1e-10);
    __syncthreads();
    double acc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++)
        acc[i] = MODE == 2 ? 0.0 : (double)(threadIdx.x + i) * 0.37 + 0.11;
    if (MODE == 2) {
        a = 0.0;
        b = 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_clk[0] = clock64();
        g_clk[1] = wall_clock64();
    }
    const uint32_t la = (uint32_t)(uintptr_t)&tab[(threadIdx.x / 3) % 21];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            double bb = b;
            if (MODE == 1) {
                typedef double v2d __attribute__((ext_vector_type(2)));
                v2d q;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(la));
                bb = q.x;
            }
#pragma unroll
            for (int i = 0; i < NCH; i++)
                acc[i] = __builtin_fma(acc[i], a, bb);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_clk[2] = clock64();
        g_clk[3] = wall_clock64();
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NCH; i++)
        s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NCH, int MODE> void run(const char *name, int waves_per_simd, int iters)
{
    const int nblk = 256 * 4 * waves_per_simd;
    double *out;
    hipMalloc(&out, sizeof(double) * nblk * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NCH, MODE>), dim3(nblk), dim3(64), 0, 0, out, 0.99999913, 1.3e-7, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NCH, MODE>), dim3(nblk), dim3(64), 0, 0, out, 0.99999913, 1.3e-7, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[4];
    hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof c);
    const double us = (double)(c[3] - c[1]) / 100.0;
    const double ghz = (double)(c[2] - c[0]) / us / 1e3;
    const double instr = (double)iters * 8 * NCH;
    printf("%-28s waves/SIMD=%d: %7.2f ms, clock %.3f GHz, %.2f cycles per wave-instr per SIMD, %.1f TFLOP/s\n", name,
           waves_per_simd, ms, ghz, (double)(c[2] - c[0]) / (instr * waves_per_simd),
           (double)nblk * 64 * instr * 2 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main()
{
    // ~1.3 M instructions per wave ~ 5 ms (a burst), then ~ 60 ms
    run<16, 0>("fma burst", 2, 10000);
    run<16, 0>("fma long", 2, 120000);
    run<16, 0>("fma long", 2, 120000);
    run<16, 1>("fma + ds_read_b128 long", 2, 120000);
    run<16, 2>("fma on zeros long", 2, 120000);
    run<16, 0>("fma long", 1, 120000);
    run<16, 0>("fma long", 4, 60000);
    return 0;
}
