// simd_map.hip -- which SIMD of its CU does wave w of a 256-thread workgroup run on?  (k_vocoder / k_vocoder_pair
// count on the four waves of a workgroup landing on the four SIMDs.)  HW_REG_HW_ID: SIMD_ID = bits 5:4, CU_ID = 11:8.
//   hipcc --offload-arch=gfx950 -O2 -o simd_map simd_map.hip && ./simd_map
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
__global__ void k(uint32_t *out)
{
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = hw | (xcc << 28);
    // keep the workgroup resident for a while so that the next ones go elsewhere
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 20000)
        ;
}
int main()
{
    for (int waves : {4, 8, 2}) {
        const int nwg = 64;
        uint32_t *d;
        (void)hipMalloc(&d, nwg * waves * 4);
        hipLaunchKernelGGL(k, dim3(nwg), dim3(64 * waves), 0, 0, d);
        (void)hipDeviceSynchronize();
        std::vector<uint32_t> h(nwg * waves);
        (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        int hist[8][4] = {};
        for (int g = 0; g < nwg; g++)
            for (int w = 0; w < waves; w++)
                hist[w][(h[g * waves + w] >> 4) & 3]++;
        printf("%d waves per workgroup, %d workgroups: SIMD of wave w (counts over SIMD 0..3)\n", waves, nwg);
        for (int w = 0; w < waves; w++)
            printf("  wave %d: %3d %3d %3d %3d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        printf("  first workgroups (xcc.cu.simd per wave):");
        for (int g = 0; g < 4; g++) {
            printf("  [");
            for (int w = 0; w < waves; w++)
                printf(" %u.%u.%u", h[g * waves + w] >> 28, (h[g * waves + w] >> 8) & 15, (h[g * waves + w] >> 4) & 3);
            printf(" ]");
        }
        printf("\n");
        (void)hipFree(d);
    }
    return 0;
}
