// Does hipExtStreamCreateWithCUMask work here, how do mask bits map to XCDs, and how much HBM
// bandwidth do N masked CUs pull?  (copy kernel, 2 GiB; per-block XCC id histogram)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>

__global__ void k_copy(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n, unsigned *xcc_hist)
{
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicAdd(&xcc_hist[xcc & 15], 1u);
    }
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b[i] = a[i];
}

int main()
{
    const size_t n = (1ull << 30) / sizeof(double2); // 1 GiB in, 1 GiB out
    double2 *a, *b;
    unsigned *hist;
    hipMalloc(&a, n * sizeof(double2));
    hipMalloc(&b, n * sizeof(double2));
    hipMalloc(&hist, 16 * 4);
    hipMemset(a, 1, n * sizeof(double2));
    struct M { const char *name; std::vector<uint32_t> m; };
    std::vector<M> masks;
    auto mk = [&](const char *name, auto pred) {
        std::vector<uint32_t> m(8, 0);
        for (int i = 0; i < 256; i++)
            if (pred(i))
                m[i / 32] |= 1u << (i % 32);
        masks.push_back({name, m});
    };
    mk("all 256", [](int) { return true; });
    mk("bits 0..63", [](int i) { return i < 64; });
    mk("bits i%4==0 (64)", [](int i) { return i % 4 == 0; });
    mk("bits 0..191", [](int i) { return i < 192; });
    mk("bits i%4!=0 (192)", [](int i) { return i % 4 != 0; });
    mk("bits i%8==0 (32)", [](int i) { return i % 8 == 0; });
    for (auto &mm : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mm.m.data());
        if (e != hipSuccess) {
            printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", mm.name, hipGetErrorString(e));
            continue;
        }
        hipMemsetAsync(hist, 0, 64, s);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, s, a, b, n, hist);
        hipMemsetAsync(hist, 0, 64, s);
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, s, a, b, n, hist);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned h[16];
        hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost);
        printf("%-20s %.3f ms  %.0f GB/s   blocks per XCC:", mm.name, ms, 2.0 * n * sizeof(double2) / ms / 1e6);
        for (int i = 0; i < 8; i++)
            printf(" %u", h[i]);
        printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
