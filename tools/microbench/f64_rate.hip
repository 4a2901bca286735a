// Microbenchmark: sustained issue rate of v_fma_f64 / v_fma_f32 per SIMD on gfx950,
// with 1, 2, 4 waves per SIMD and NCHAIN independent accumulators per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int NCH>
__global__ __launch_bounds__(64) void k(T *out, T a, T b, int iters)
{
    T acc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++)
        acc[i] = (T)(threadIdx.x + i);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < NCH; i++)
                acc[i] = __builtin_fma(acc[i], a, b);
    }
    T s = 0;
#pragma unroll
    for (int i = 0; i < NCH; i++)
        s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <typename T, int NCH> void run(const char *name, int waves_per_simd)
{
    const int nblk = 256 * 4 * waves_per_simd;
    T *out;
    hipMalloc(&out, sizeof(T) * nblk * 64);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<T, NCH>), dim3(nblk), dim3(64), 0, 0, out, (T)1.0000001, (T)1e-9, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<T, NCH>), dim3(nblk), dim3(64), 0, 0, out, (T)1.0000001, (T)1e-9, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 8 * NCH;
    const double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
    const double tflops = (double)nblk * 64 * instr_per_wave * 2 / (ms * 1e-3) / 1e12;
    printf("%s chains=%d waves/SIMD=%d: %.3f ms, %.2f ns per wave-instr per SIMD (%.1f cyc @2.4GHz), %.1f TFLOP/s\n",
           name, NCH, waves_per_simd, ms, ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4, tflops);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<double, 8>("f64", 1); run<double, 16>("f64", 1); run<float, 16>("f32", 1); }
        if (w == 2) { run<double, 8>("f64", 2); run<double, 16>("f64", 2); run<float, 16>("f32", 2); }
        if (w == 4) { run<double, 8>("f64", 4); run<float, 16>("f32", 4); }
    }
    return 0;
}
