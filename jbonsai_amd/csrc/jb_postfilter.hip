// X1: mel-cepstral post-filter, beta > 0.
//
// Reference (per frame, inside Vocoder::synthesize, src/vocoder/mod.rs:116-118):
//   MelCepstrum::postfilter_mcp   src/vocoder/cepstrum.rs:23-37
//     b  = mc2b(c);  e1 = b2en(b)
//     b[1] -= beta*alpha*b[2];  b[k] *= 1+beta (k >= 2);  e2 = b2en(b)
//     b[0] += ln(e1/e2)/2;  c' = b2mc(b)          -> the vocoder then takes mc2b(c')
//   b2en  = sum(c2ir(freqt(b2mc(b), 575, -alpha), 576)^2)   src/vocoder/coefficients.rs:65-78
//   freqt                          src/vocoder/cepstrum.rs:153-173
//   c2ir                           src/vocoder/cepstrum.rs:175-186
//
// Two things about the reference's arithmetic shape this file:
//  * freqt feeds its input in ASCENDING index order (hts_engine's HTS_freqt walks c1[m1]..c1[0]),
//    so what it computes is not the textbook frequency transform of c -- it is followed as written.
//    It is linear in its input with a zero start state, so it is a constant [576][nmcp] operator of
//    the voice (alpha only).  k_pf_table builds that operator once by running the reference's own
//    recurrence on an impulse: column i is the state (nmcp-1-i) zero-input steps after the impulse
//    enters, bit for bit what the recurrence gives for the unit vector e_i.
//  * c2ir runs over the full 576-entry transformed cepstrum: n*ir[n] = sum_{k=1..n} k*g[k]*ir[n-k],
//    166 k multiply-adds per call, two calls per frame -- 0.7 Mflop per frame against 0.32 Mflop for
//    the whole MLSA filter of that frame.  With beta > 0 this kernel, not the vocoder, is the largest.
//
// k_postfilter: one wave per frame, both energies (e1, e2) computed side by side as two
// independent chains.  The convolution recurrence is blocked by 64: lane = n mod 64, nine
// accumulators per chain in registers.  Block B is finished by a 64-step triangular sweep (one ir
// value becomes final per step: times the correctly rounded 1/n, then broadcast), and then pushed into every later block as a Toeplitz product:
// ir[64B+l] comes from a register by readlane, k*g[k] from LDS with consecutive lanes on
// consecutive words.  The bound is LDS bandwidth: every multiply-add takes one 8-byte LDS operand
// per lane (512 B per wave instruction, 128 B/clk per CU).
//
// Rounding: the sums run in a different order from the reference's serial k loop and use FMAs;
// the energy sum itself keeps the reference's order.  Measured against the oracle the shift of
// b[0] agrees to ~1e-14 absolute (tests/test_gpu_postfilter.py).
#include "jb_device.h"

namespace jb {

constexpr int kIrLen = 576; // IRLENG (coefficients.rs:76)
constexpr int kIrBlk = kIrLen / 64;
#ifndef JB_PF_WAVES
#define JB_PF_WAVES 2
#endif
#ifndef JB_PF_BCAST
#define JB_PF_BCAST 1 // 1: finished block reaches the lanes through same-address LDS reads; 0: readlane
#endif
constexpr int kPfWaves = JB_PF_WAVES; // waves (= frames in flight) per workgroup

__device__ __forceinline__ double pf_readlane(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// One impulse through the reference's freqt recurrence (cepstrum.rs:156-170): after the impulse
// step the state is e_0; each further step is the zero-input map.  table[i][j], i = input index.
// rcp[n] = 1/n.  A single lane walks the 576-entry state in LDS (~20 k steps, once per batch).
__global__ __launch_bounds__(64) void k_pf_table(double *__restrict__ table, double *__restrict__ rcp, int nm,
                                                 double alpha)
{
    __shared__ double st[kIrLen];
    const int lane = threadIdx.x;
    for (int j = lane; j < kIrLen; j += 64) {
        st[j] = j == 0 ? 1.0 : 0.0;
        rcp[j] = j ? 1.0 / (double)j : 0.0;
    }
    __syncthreads();
    const double aa = 1.0 - alpha * alpha;
    for (int k = 0; k < nm; k++) {
        // state after k zero-input steps belongs to input index nm-1-k
        for (int j = lane; j < kIrLen; j += 64)
            table[(size_t)(nm - 1 - k) * kIrLen + j] = st[j];
        __syncthreads();
        if (lane == 0 && k + 1 < nm) {
            double fprev = st[0];          // f[0]
            double cur = 0.0 + alpha * st[0]; // self[i] == 0
            st[0] = cur;
            {
                const double f1 = st[1];
                const double n1 = aa * fprev + alpha * st[1];
                st[1] = n1;
                fprev = f1;
                cur = n1;
            }
            for (int j = 2; j < kIrLen; j++) {
                const double fj = st[j];
                const double nj = fprev + alpha * (fj - cur);
                st[j] = nj;
                fprev = fj;
                cur = nj;
            }
        }
        __syncthreads();
    }
}

// un-filtered bcoef of each utterance's first frame (Vocoder::synthesize's is_first branch)
__global__ void k_pf_first(BatchDev bd, VocDev vd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bd.B * vd.nmcp)
        return;
    const int b = i / vd.nmcp, k = i % vd.nmcp;
    if (bd.utt[b].T == 0)
        return;
    vd.bfirst[i] = vd.bcoef[bd.utt[b].frame_off * (uint64_t)vd.nmcp + (uint64_t)k];
}

// LDS per wave and chain: 64 zeros, then k*g[k] for k = 0..575.  The zeros are what lanes whose sum is
// already final multiply by in the triangular sweep (index 64 + lane - j <= 64, and kg[0] == 0).
constexpr int kPfPad = 64;
constexpr int kPfChain = kPfPad + kIrLen;
constexpr int kPfWaveLds = 2 * kPfChain + 2 * 64; // + the finished block of each chain, for broadcast reads

// LDS pointer whose loads stay single ds_read_b64: 2 LDS cycles per wave instruction (256 B/clk per CU);
// merged into ds_read2_b64 a pair costs 8 (128 B/clk).  volatile is what keeps them apart.
typedef const volatile __attribute__((address_space(3))) double *pf_lds_ptr;

__device__ __forceinline__ double pf_wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v += __shfl_xor(v, m);
    return v;
}

__global__ __launch_bounds__(64 * kPfWaves) void k_postfilter(VocDev vd, uint64_t nframes)
{
    extern __shared__ double pf_lds[]; // per wave: pad|kg0[576] | pad|kg1[576] | fin0[64] | fin1[64]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *kg0 = pf_lds + (size_t)w * kPfWaveLds + kPfPad;
    double *kg1 = kg0 + kPfChain;
    double *fl0 = kg1 + kIrLen, *fl1 = fl0 + 64;
    kg0[lane - kPfPad] = 0.0;
    kg1[lane - kPfPad] = 0.0;
    const int nm = vd.nmcp;
    const double alpha = vd.alpha, beta = vd.beta;
    const double *__restrict__ table = vd.pf_table;
    const double *__restrict__ rcp = vd.pf_rcp;
    const uint64_t stride = (uint64_t)gridDim.x * kPfWaves;
    for (uint64_t f = (uint64_t)blockIdx.x * kPfWaves + (uint64_t)w; f < nframes; f += stride) {
        double *bp = vd.bcoef + f * (uint64_t)nm;
        // ---- b (= mc2b(c), k_mc2b*) and the emphasised b' (cepstrum.rs:28-31) ----
        const double b0 = lane < nm ? bp[lane] : 0.0;
        const double b2 = pf_readlane(b0, 2);
        double b1 = b0;
        if (lane == 1)
            b1 = b0 - beta * alpha * b2;
        else if (lane >= 2)
            b1 = b0 * (1.0 + beta);
        // b2mc (coefficients.rs:65-73): c[i] = b[i] + alpha*b[i+1], last copied
        const double b0n = __shfl_down(b0, 1), b1n = __shfl_down(b1, 1);
        const double mc0 = lane < nm - 1 ? b0 + alpha * b0n : b0;
        const double mc1 = lane < nm - 1 ? b1 + alpha * b1n : b1;
        // ---- g = freqt(mc, 575, -alpha) as table * mc; lane holds j = 64*blk + lane ----
        double A0[kIrBlk], A1[kIrBlk];
#pragma unroll
        for (int r = 0; r < kIrBlk; r++)
            A0[r] = A1[r] = 0.0;
        for (int i = 0; i < nm; i++) {
            const double m0 = pf_readlane(mc0, i), m1 = pf_readlane(mc1, i);
            const double *row = table + (size_t)i * kIrLen + lane;
#pragma unroll
            for (int r = 0; r < kIrBlk; r++) {
                const double fv = row[64 * r];
                A0[r] = __builtin_fma(fv, m0, A0[r]);
                A1[r] = __builtin_fma(fv, m1, A1[r]);
            }
        }
        // ir[0] = exp(g[0]) (cepstrum.rs:177)
        const double x00 = exp(pf_readlane(A0[0], 0)), x10 = exp(pf_readlane(A1[0], 0));
        // k*g[k] (c2ir's first product, cepstrum.rs:181)
#pragma unroll
        for (int r = 0; r < kIrBlk; r++) {
            const double kf = (double)(64 * r + lane);
            kg0[64 * r + lane] = kf * A0[r];
            kg1[64 * r + lane] = kf * A1[r];
            A0[r] = A1[r] = 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- c2ir (cepstrum.rs:175-186), both chains ----
        double es0 = 0.0, es1 = 0.0; // per-lane partial sums of ir^2
#pragma unroll
        for (int B = 0; B < kIrBlk; B++) {
            // d / n as d * (1/n) with the correctly rounded reciprocal (within one ulp of the reference's
            // division), taken for every lane's current sum; only lane j's is final, and used, at step j.
            // Finished lanes keep adding x * 0.
            const double rcpv = rcp[64 * B + lane];
            const pf_lds_ptr t0 = (pf_lds_ptr)(kg0 + lane), t1 = (pf_lds_ptr)(kg1 + lane);
#pragma unroll 8
            for (int j = 0; j < 64; j++) {
                // (v_readlane, not ds_bpermute: the crossbar's latency sits on the serial chain, 349 vs 252 ms)
                double x0 = pf_readlane(A0[B] * rcpv, j), x1 = pf_readlane(A1[B] * rcpv, j);
                if (B == 0 && j == 0) {
                    x0 = x00;
                    x1 = x10;
                }
                A0[B] = __builtin_fma(x0, t0[-j], A0[B]);
                A1[B] = __builtin_fma(x1, t1[-j], A1[B]);
            }
            // final ir[64B + lane]: the same quotient the sweep broadcast for this lane
            double fin0 = A0[B] * rcpv, fin1 = A1[B] * rcpv;
            if (B == 0 && lane == 0) {
                fin0 = x00;
                fin1 = x10;
            }
            es0 += fin0 * fin0;
            es1 += fin1 * fin1;
            // Toeplitz push of block B into blocks r > B: k = 64(r-B) + lane - l in 1..575
            if (B + 1 < kIrBlk) {
                // ir[64B + l] reaches all lanes as a same-address LDS read (the VALU, not the LDS, is
                // the busier unit here; two readlanes per value would go to the VALU)
                fl0[lane] = fin0;
                fl1[lane] = fin1;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const pf_lds_ptr q0 = (pf_lds_ptr)fl0, q1 = (pf_lds_ptr)fl1;
                for (int l = 0; l < 64; l++) {
#if JB_PF_BCAST
                    const double u0 = q0[l], u1 = q1[l];
#else
                    const double u0 = pf_readlane(fin0, l), u1 = pf_readlane(fin1, l);
#endif
                    const pf_lds_ptr p0 = (pf_lds_ptr)(kg0 + lane - l), p1 = (pf_lds_ptr)(kg1 + lane - l);
#pragma unroll
                    for (int r = B + 1; r < kIrBlk; r++) {
                        A0[r] = __builtin_fma(u0, p0[64 * (r - B)], A0[r]);
                        A1[r] = __builtin_fma(u1, p1[64 * (r - B)], A1[r]);
                    }
                }
            }
        }
        // ---- gain correction (cepstrum.rs:33-34), b2mc, and the vocoder's own mc2b (mod.rs:118) ----
        const double e0 = pf_wave_sum(es0), e1 = pf_wave_sum(es1);
        const double shift = log(e0 / e1) / 2.0;
        if (lane == 0)
            b1 += shift;
        const double b1s = __shfl_down(b1, 1);
        const double mcf = lane < nm - 1 ? b1 + alpha * b1s : b1;
        double prev = pf_readlane(mcf, nm - 1);
        double outv = prev;
        if (alpha != 0.0) {
            for (int i = nm - 2; i >= 0; i--) {
                prev = pf_readlane(mcf, i) - alpha * prev;
                outv = lane == i ? prev : outv;
            }
        } else {
            outv = mcf;
        }
        if (lane < nm)
            bp[lane] = outv;
        // the next frame's kg stores must not pass this frame's LDS reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

hipError_t launch_pf_table(const VocDev &vd, hipStream_t stream)
{
    // postfilter_mcp's transform uses -alpha (coefficients.rs:76)
    hipLaunchKernelGGL(k_pf_table, dim3(1), dim3(64), 0, stream, vd.pf_table, vd.pf_rcp, vd.nmcp, -vd.alpha);
    return hipGetLastError();
}

hipError_t launch_postfilter(const BatchDev &bd, const VocDev &vd, uint64_t nframes, hipStream_t stream)
{
    if (nframes == 0 || bd.B == 0)
        return hipSuccess;
    const int n1 = bd.B * vd.nmcp;
    hipLaunchKernelGGL(k_pf_first, dim3((n1 + 255) / 256), dim3(256), 0, stream, bd, vd);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    const size_t lds = (size_t)kPfWaves * kPfWaveLds * sizeof(double); // 11 KB per wave
    // persistent grid: exactly the workgroups that are resident at once (a second, thinner round of
    // workgroups would run at a fraction of the occupancy the latency-bound sweep needs)
    static int per_cu = 0, ncu = 0;
    if (!per_cu) {
        int occ = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_postfilter, 64 * kPfWaves, lds) != hipSuccess || occ < 1)
            occ = 2;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return hipGetLastError();
        ncu = prop.multiProcessorCount;
        per_cu = occ;
    }
    uint64_t blocks = (nframes + kPfWaves - 1) / kPfWaves;
    if (blocks > (uint64_t)ncu * (uint64_t)per_cu)
        blocks = (uint64_t)ncu * (uint64_t)per_cu;
    hipLaunchKernelGGL(k_postfilter, dim3((unsigned)blocks), dim3(64 * kPfWaves), lds, stream, vd, nframes);
    return hipGetLastError();
}

} // namespace jb
