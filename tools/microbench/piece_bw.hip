// Microbenchmark: HBM bandwidth of the [row][frame] workspace layout as a function of the contiguous
// piece a workgroup touches per row.  The MLPG kernels move tiles of (35 rows) x (P bytes) of arrays whose
// rows are ~200 KB apart (one row = one dimension of one utterance): is 128- or 256-byte granularity
// what holds k_mlpg_build_mt / k_mlpg_fb_lds at 2.4-2.9 TB/s?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/piece_bw.hip -o /tmp/piece_bw && /tmp/piece_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kRows = 35;

// one workgroup per (tile of P bytes, utterance): rows r = 0..34, piece [col0, col0 + P) of each row
template <int MODE> // 0 read, 1 write, 2 copy (read A, write B)
__global__ __launch_bounds__(256) void k(const double *__restrict__ A, double *__restrict__ B, double *sink,
                                         uint64_t row_doubles, int piece_doubles, int tiles_per_row)
{
    const uint64_t utt = blockIdx.y;
    const uint64_t col0 = (uint64_t)blockIdx.x * (uint64_t)piece_doubles;
    const uint64_t base = utt * (uint64_t)kRows * row_doubles;
    double acc = 0.0;
    const int n = kRows * piece_doubles;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int r = e / piece_doubles, c = e - r * piece_doubles;
        const uint64_t o = base + (uint64_t)r * row_doubles + col0 + (uint64_t)c;
        if (MODE == 0)
            acc += A[o];
        else if (MODE == 1)
            B[o] = (double)e;
        else
            B[o] = A[o] + 1.0;
    }
    if (MODE == 0 && acc == 1.2345)
        sink[0] = acc;
}

template <int MODE> void run(const char *name, const double *A, double *B, double *sink, uint64_t row_doubles, int utts,
                             int piece_bytes)
{
    const int pd = piece_bytes / 8;
    const int tiles = (int)(row_doubles / pd);
    dim3 grid(tiles, utts), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, A, B, sink, row_doubles, pd, tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; i++)
        hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, A, B, sink, row_doubles, pd, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double bytes = (double)utts * kRows * (double)tiles * piece_bytes * (MODE == 2 ? 2 : 1);
    printf("%-5s piece %5d B: %7.3f ms  %6.2f TB/s\n", name, piece_bytes, ms, bytes / (ms * 1e-3) / 1e12);
}

int main()
{
    const uint64_t row_doubles = 25600; // ~ 25,546 frames
    const int utts = 256;
    const size_t bytes = (size_t)utts * kRows * row_doubles * 8; // 1.8 GB per array
    double *A, *B, *sink;
    hipMalloc(&A, bytes);
    hipMalloc(&B, bytes);
    hipMalloc(&sink, 8);
    hipMemset(A, 0, bytes);
    hipMemset(B, 0, bytes);
    for (int p : {128, 256, 512, 1024, 2048, 4096}) {
        run<0>("read", A, B, sink, row_doubles, utts, p);
        run<1>("write", A, B, sink, row_doubles, utts, p);
        run<2>("copy", A, B, sink, row_doubles, utts, p);
    }
    return 0;
}
