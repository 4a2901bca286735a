// jb_mlpg.hip -- parameter generation (MLPG + GV) kernels for gfx950.
//
// Restates, per (utterance, vector dimension), /root/reference/src/mlpg_adjust:
//   k_prep        A1/A2  Mask::create + boundary_distances   (mask.rs:20-82)
//   k_mlpg_build  A3/A4  per-dim parameter expansion + calc_wuw_and_wum
//                        (mod.rs:56-84, mlpg.rs:25-70) -- elementwise in time, so
//                        it is time-parallel: one thread per (frame, dim)
//   k_mlpg_solve  A5-A9  band LDL^T, substitutions, GV ascent, scatter
//                        (mlpg.rs:79-292, mask.rs:34-49) -- serial recurrences in
//                        time: one lane per (utterance, dim), lanes = adjacent dims
//                        so every load/store of the [frame][dim] workspace is
//                        coalesced.  All sums run in the reference's order, in
//                        f64, with FP contraction off: this kernel is bit-exact
//                        against the oracle except through libm-free paths only
//                        (sqrt and division are correctly rounded on gfx950).
// LF0 needs f64 + reference order: pulse positions are chaotic w.r.t. rounding of
// lf0.  The same code serves MCP and LPF.
//
// Compiled with -ffp-contract=off.
#include "jb_device.h"

namespace jb {

// --------------------------------------------------------------------------
// A1/A2: one lane per utterance; three short serial sweeps.
__global__ void k_prep(BatchDev bd, StreamDev sd, int si)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= bd.B)
        return;
    const UttDev u = bd.utt[b];
    const StreamStatesDev st = u.st[si];
    const uint64_t base = u.frame_off;
    uint32_t t = 0;
    for (uint32_t s = 0; s < u.S; s++) {
        // msd.unwrap_or(f64::MAX) > threshold  (model/mod.rs:113, mask.rs:24)
        double msd = st.msd ? st.msd[s] : 1.7976931348623157e308;
        uint8_t v = msd > st.msd_threshold;
        uint32_t d = u.dur[s];
        for (uint32_t i = 0; i < d; i++, t++) {
            sd.fstate[base + t] = s;
            sd.voiced[base + t] = v;
        }
    }
    const uint32_t T = t;
    uint32_t left = 0, k = 0, gl = 0;
    for (uint32_t f = 0; f < T; f++) {
        if (sd.voiced[base + f]) {
            uint32_t dl = f - left;
            sd.fl[base + f] = dl > 255 ? 255 : (uint8_t)dl;
            sd.vidx[base + k] = f;
            uint8_t sw = st.gv_switch ? st.gv_switch[sd.fstate[base + f]] : 0;
            sd.vsw[base + k] = sw;
            gl += sw;
            k++;
        } else {
            left = f + 1;
            sd.fl[base + f] = 0;
        }
    }
    if (T > 0) {
        uint32_t right = T - 1;
        for (uint32_t f = T; f-- > 0;) {
            if (sd.voiced[base + f]) {
                uint32_t dr = right - f;
                sd.fr[base + f] = dr > 255 ? 255 : (uint8_t)dr;
            } else {
                sd.fr[base + f] = 0;
                if (f == 0)
                    break;
                right = f - 1;
            }
        }
    }
    sd.Tv[b] = k;
    sd.gvlen[b] = gl;
}

// MeanVari::with_ivar (model/mean_vari.rs:21-31)
__device__ __forceinline__ double with_ivar(double vari)
{
    double av = fabs(vari);
    if (av > 1e19)
        return 0.0;
    if (av < 1e-19)
        return 1e38;
    return 1.0 / vari;
}

// --------------------------------------------------------------------------
// A3/A4: thread per (compacted frame k, dim m) of utterance blockIdx.y.
template <int BW>
__global__ void k_mlpg_build(BatchDev bd, StreamDev sd, int si)
{
    const int b = blockIdx.y;
    const UttDev u = bd.utt[b];
    const uint32_t Tv = sd.Tv[b];
    const int L = sd.L, W = sd.W;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (uint64_t)Tv * L)
        return;
    const uint32_t k = (uint32_t)(tid / L);
    const int m = (int)(tid % L);
    const StreamStatesDev st = u.st[si];
    const uint64_t base = u.frame_off;
    double wuw[BW];
#pragma unroll
    for (int j = 0; j < BW; j++)
        wuw[j] = 0.0;
    double wum = 0.0;
    for (int w = 0; w < W; w++) {
        const int ww = sd.win_width[w];
        const double *coef = sd.win_coef + sd.win_off[w];
        const int lw = ww / 2, rw = ww - lw - 1;
        for (int index = ww - 1; index >= 0; index--) {
            const double c = coef[index];
            if (c == 0.0)
                continue;
            const long idx = (long)k - ((long)index - (long)(ww / 2));
            if (idx < 0 || idx >= (long)Tv)
                continue;
            const uint32_t f = sd.vidx[base + idx];
            const uint32_t s = sd.fstate[base + f];
            const uint64_t pi = (uint64_t)s * (uint64_t)(W * L) + (uint64_t)(L * w + m);
            const double mean = st.mean[pi];
            double ivar = with_ivar(st.var[pi]);
            // dynamic windows touching an MSD boundary get ivar = 0 (mod.rs:69-80)
            if (w != 0 && ((int)sd.fl[base + f] < lw || (int)sd.fr[base + f] < rw))
                ivar = 0.0;
            const double wu = c * ivar;
            wum += wu * mean;
            for (int inner = ww - 1; inner >= index; inner--) {
                const double c2 = coef[inner];
                if (c2 == 0.0)
                    continue;
                const int j = inner - index;
                if ((uint64_t)k + (uint64_t)j >= Tv)
                    break;
#pragma unroll
                for (int jj = 0; jj < BW; jj++)
                    if (jj == j)
                        wuw[jj] += wu * c2;
            }
        }
    }
    const uint64_t o = (base + k) * (uint64_t)L + (uint64_t)m;
#pragma unroll
    for (int j = 0; j < BW; j++)
        sd.A[j][o] = wuw[j];
    sd.bvec[o] = wum;
}

// --------------------------------------------------------------------------
// A5-A9: lane per (utterance, dim).
template <int BW>
__global__ void k_mlpg_solve(BatchDev bd, StreamDev sd, int si)
{
    const int b = blockIdx.y;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int L = sd.L;
    if (m >= L)
        return;
    const UttDev u = bd.utt[b];
    const StreamStatesDev st = u.st[si];
    const uint32_t T = u.T;
    const uint32_t Tv = sd.Tv[b];
    const uint64_t base = u.frame_off;
    const uint64_t o0 = base * (uint64_t)L + (uint64_t)m;
#define IX(k) (o0 + (uint64_t)(k) * (uint64_t)L)

    if (Tv > 0) {
        // ---- A5 ldl_factorization + forward substitution (mlpg.rs:79-105) ----
        double P[BW][BW]; // P[i] = factored row t-i (i>=1)
        double G[BW];     // G[i] = g[t-i]
#pragma unroll
        for (int i = 0; i < BW; i++) {
            G[i] = 0.0;
#pragma unroll
            for (int j = 0; j < BW; j++)
                P[i][j] = 0.0;
        }
        for (uint32_t t = 0; t < Tv; t++) {
            double row[BW];
#pragma unroll
            for (int j = 0; j < BW; j++)
                row[j] = sd.A[j][IX(t)];
            double g = sd.bvec[IX(t)];
#pragma unroll
            for (int i = 1; i < BW; i++)
                if ((uint32_t)i <= t)
                    row[0] -= P[i][i] * P[i][i] * P[i][0];
#pragma unroll
            for (int i = 1; i < BW; i++) {
#pragma unroll
                for (int j = 1; j < BW - i; j++)
                    if ((uint32_t)j <= t)
                        row[i] -= P[j][j] * P[j][i + j] * P[j][0];
                row[i] /= row[0];
            }
#pragma unroll
            for (int i = 1; i < BW; i++)
                if ((uint32_t)i <= t)
                    g -= P[i][i] * G[i];
#pragma unroll
            for (int j = 0; j < BW; j++)
                sd.F[j][IX(t)] = row[j];
            sd.g[IX(t)] = g;
#pragma unroll
            for (int i = BW - 1; i >= 2; i--) {
                G[i] = G[i - 1];
#pragma unroll
                for (int j = 0; j < BW; j++)
                    P[i][j] = P[i - 1][j];
            }
            if constexpr (BW > 1) {
                G[1] = g;
#pragma unroll
                for (int j = 0; j < BW; j++)
                    P[1][j] = row[j];
            }
        }
        // ---- A6 backward substitution (mlpg.rs:106-113) ----
        {
            double Q[BW]; // Q[i] = par[t+i]
#pragma unroll
            for (int i = 0; i < BW; i++)
                Q[i] = 0.0;
            for (uint32_t t = Tv; t-- > 0;) {
                double p = sd.g[IX(t)] / sd.F[0][IX(t)];
#pragma unroll
                for (int i = 1; i < BW; i++)
                    if (t + (uint32_t)i < Tv)
                        p -= sd.F[i][IX(t)] * Q[i];
                sd.par[IX(t)] = p;
#pragma unroll
                for (int i = BW - 1; i >= 2; i--)
                    Q[i] = Q[i - 1];
                if constexpr (BW > 1)
                    Q[1] = p;
            }
        }
        // ---- A8 GV (mlpg.rs:145-292) ----
        const uint32_t gvlen = sd.gvlen[b];
        if (sd.use_gv && st.gv_mean && gvlen > 0) {
            const uint8_t *sw = sd.vsw + base;
            const double gv_mean = st.gv_mean[m] * st.gv_weight; // mlpg.rs:135-137
            const double gv_vari = st.gv_var[m];
            const double glen = (double)gvlen;
            double mean, vari;
            auto calc_gv = [&]() {
                double s = 0.0;
                for (uint32_t t = 0; t < Tv; t++)
                    if (sw[t])
                        s += sd.par[IX(t)];
                mean = s / glen;
                double v = 0.0;
                for (uint32_t t = 0; t < Tv; t++)
                    if (sw[t]) {
                        double p = sd.par[IX(t)];
                        v += (p - mean) * (p - mean);
                    }
                vari = v / glen;
            };
            // conv_gv (mlpg.rs:195-203)
            calc_gv();
            {
                const double ratio = sqrt(gv_mean / vari);
                for (uint32_t t = 0; t < Tv; t++)
                    if (sw[t])
                        sd.par[IX(t)] = ratio * (sd.par[IX(t)] - mean) + mean;
            }
            double step = 0.1, prev = 0.0; // STEPINIT
            const double length = (double)Tv;
            const double wgt = 1.0 / (double)((uint64_t)sd.W * (uint64_t)Tv);
            for (int it = 1; it <= 5; it++) { // GV_MAX_ITERATION
                calc_gv();
                const double gvobj = -0.5 * 1.0 * vari * gv_vari * (vari - 2.0 * gv_mean);
                // calc_hmmobj_derivative (mlpg.rs:204-229)
                for (uint32_t t = 0; t < Tv; t++) {
                    double g = sd.A[0][IX(t)] * sd.par[IX(t)];
#pragma unroll
                    for (int i = 1; i < BW; i++) {
                        if (t + (uint32_t)i < Tv)
                            g += sd.A[i][IX(t)] * sd.par[IX(t + i)];
                        if (t + 1 > (uint32_t)i)
                            g += sd.A[i][IX(t - i)] * sd.par[IX(t - i)];
                    }
                    sd.g[IX(t)] = g;
                }
                double hmmobj = 0.0;
                for (uint32_t t = 0; t < Tv; t++)
                    hmmobj += 1.0 * wgt * sd.par[IX(t)] * (sd.bvec[IX(t)] - 0.5 * sd.g[IX(t)]);
                const double obj = -(hmmobj + gvobj);
                if (it > 1) {
                    if (obj > prev)
                        step *= 0.5; // STEPDEC
                    else if (obj < prev)
                        step *= 1.2; // STEPINC
                }
                // next_step (mlpg.rs:230-258)
                const double dv = -2.0 * gv_vari * (vari - gv_mean) / length;
                const double ll = (double)((uint64_t)Tv * (uint64_t)Tv);
                const double lm1 = (double)(Tv - 1);
                for (uint32_t t = 0; t < Tv; t++) {
                    const double p = sd.par[IX(t)];
                    const double h = -1.0 * wgt * sd.A[0][IX(t)] -
                                     1.0 * 2.0 / ll *
                                         (lm1 * gv_vari * (vari - gv_mean) +
                                          2.0 * gv_vari * (p - mean) * (p - mean));
                    double next_g;
                    if (sw[t])
                        next_g = 1.0 / h *
                                 (1.0 * wgt * (-sd.g[IX(t)] + sd.bvec[IX(t)]) + 1.0 * dv * (p - mean));
                    else
                        next_g = 1.0 / h * (1.0 * wgt * (-sd.g[IX(t)] + sd.bvec[IX(t)]));
                    sd.par[IX(t)] = p + step * next_g;
                }
                prev = obj;
            }
        }
    }
    // ---- A9 scatter with NODATA (mask.rs:34-49, mod.rs:89-91) ----
    {
        uint32_t k = 0;
        for (uint32_t t = 0; t < T; t++) {
            double v = kNoData;
            if (sd.voiced[base + t]) {
                v = sd.par[IX(k)];
                k++;
            }
            sd.out[IX(t)] = v;
        }
    }
#undef IX
}

hipError_t launch_prep(const BatchDev &bd, const StreamDev &sd, int si, hipStream_t stream)
{
    if (bd.B == 0)
        return hipSuccess;
    dim3 grid((bd.B + 63) / 64), block(64);
    hipLaunchKernelGGL(k_prep, grid, block, 0, stream, bd, sd, si);
    return hipGetLastError();
}

template <int BW>
static hipError_t launch_mlpg_bw(const BatchDev &bd, const StreamDev &sd, int si, hipStream_t stream)
{
    const uint64_t work = (uint64_t)bd.maxT * (uint64_t)sd.L;
    if (work == 0 || bd.B == 0)
        return hipSuccess;
    {
        dim3 grid((unsigned)((work + 255) / 256), bd.B), block(256);
        hipLaunchKernelGGL(k_mlpg_build<BW>, grid, block, 0, stream, bd, sd, si);
    }
    {
        dim3 grid((sd.L + 63) / 64, bd.B), block(64);
        hipLaunchKernelGGL(k_mlpg_solve<BW>, grid, block, 0, stream, bd, sd, si);
    }
    return hipGetLastError();
}

hipError_t launch_mlpg(const BatchDev &bd, const StreamDev &sd, int si, hipStream_t stream)
{
    switch (sd.BW) {
    case 1:
        return launch_mlpg_bw<1>(bd, sd, si, stream);
    case 3:
        return launch_mlpg_bw<3>(bd, sd, si, stream);
    case 5:
        return launch_mlpg_bw<5>(bd, sd, si, stream);
    default:
        return hipErrorInvalidValue;
    }
}

} // namespace jb
