// Microbenchmark: what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for a KNOWN number of bytes, per access
// width -- the calibration MI355X_MICROARCH.md asks for before an absolute is trusted ("other access widths are
// uncalibrated").  Every kernel streams a 2 GiB buffer (past the 256 MiB Infinity Cache) exactly once:
//   rd4 / rd8 / rd16      global_load_dword / dwordx2 / dwordx4, lanes contiguous (4, 8, 16 B per lane)
//   rd8_rows              8 B per lane, 16-frame (128 B) pieces of rows 200 KB apart: the [dim][frame] workspace
//                         as the band solve's movers and the GV kernel read it
//   rd8_lane_stream       8 B per lane, every lane its own stream (lanes 1,920 B apart: one cache line per lane and
//                         instruction): the vocoder kernel's excitation reads
//   wr8 / wr16            stores of 8 / 16 B per lane
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/fetch_calib.hip -o tools/microbench/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o p --output-format csv -- tools/microbench/fetch_calib
// (tools/fetch_calib.sh runs both passes and prints reported / known per kernel.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr size_t kBytes = 2ull << 30;

template <typename T> __global__ __launch_bounds__(256) void rd(const T *__restrict__ a, size_t n, double *sink)
{
    T acc{};
    unsigned *ap = reinterpret_cast<unsigned *>(&acc);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = a[i];
        const unsigned *vp = reinterpret_cast<const unsigned *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++)
            ap[k] ^= vp[k];
    }
    if (ap[0] == 0x12345678u)
        sink[0] = 1.0;
}
__global__ __launch_bounds__(256) void rd8_rows(const double *__restrict__ a, size_t rows, size_t row_doubles, double *sink)
{
    // a workgroup takes 16-frame pieces: thread t -> row (t / 16) of a group of 16 rows, frame t % 16 of the piece
    double acc = 0.0;
    const size_t pieces = row_doubles / 16;
    for (size_t w = blockIdx.x; w < (rows / 16) * pieces; w += gridDim.x) {
        const size_t rg = w / pieces, pc = w % pieces;
        const size_t r = rg * 16 + threadIdx.x / 16;
        acc += a[r * row_doubles + pc * 16 + threadIdx.x % 16];
    }
    if (acc == 1.2345)
        sink[0] = acc;
}
__global__ __launch_bounds__(64) void rd8_lane_stream(const double *__restrict__ a, size_t per_lane, double *sink)
{
    // lane l of wave w streams doubles [ (w * 64 + l) * per_lane, + per_lane )
    const size_t id = (size_t)blockIdx.x * 64 + threadIdx.x;
    const double *p = a + id * per_lane;
    double acc = 0.0;
    for (size_t i = 0; i < per_lane; i++)
        acc += p[i];
    if (acc == 1.2345)
        sink[0] = acc;
}
template <typename T> __global__ __launch_bounds__(256) void wr(T *__restrict__ a, size_t n)
{
    T v{};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a[i] = v;
}

int main()
{
    void *buf;
    double *sink;
    if (hipMalloc(&buf, kBytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess)
        return 1;
    hipMemset(buf, 0, kBytes);
    hipDeviceSynchronize();
    const int grid = 256 * 16;
    hipLaunchKernelGGL(rd<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, kBytes / 4, sink);
    hipLaunchKernelGGL(rd<uint2>, dim3(grid), dim3(256), 0, 0, (const uint2 *)buf, kBytes / 8, sink);
    hipLaunchKernelGGL(rd<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4 *)buf, kBytes / 16, sink);
    {
        const size_t row_doubles = 25600, rows = kBytes / 8 / row_doubles / 16 * 16;
        hipLaunchKernelGGL(rd8_rows, dim3(grid), dim3(256), 0, 0, (const double *)buf, rows, row_doubles, sink);
        printf("rd8_rows known bytes %zu\n", rows * row_doubles * 8);
    }
    {
        const size_t lanes = 256 * 4 * 2 * 64, per_lane = kBytes / 8 / lanes; // two waves per SIMD, one pass
        hipLaunchKernelGGL(rd8_lane_stream, dim3(lanes / 64), dim3(64), 0, 0, (const double *)buf, per_lane, sink);
        printf("rd8_lane_stream known bytes %zu\n", lanes * per_lane * 8);
    }
    hipLaunchKernelGGL(wr<uint2>, dim3(grid), dim3(256), 0, 0, (uint2 *)buf, kBytes / 8);
    hipLaunchKernelGGL(wr<uint4>, dim3(grid), dim3(256), 0, 0, (uint4 *)buf, kBytes / 16);
    hipDeviceSynchronize();
    printf("stream kernels known bytes %zu\n", kBytes);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
