// jb_mglsa.hip -- Stage::NonZero of the vocoder: voices with GAMMA != 0, whose
// spectrum stream holds [gain, line spectral frequencies] and whose synthesis filter is the
// MGLSA cascade.  Restates /root/reference/src/vocoder:
//   k_stage_coef      per frame: postfilter_lsp, check_lsp_stability (lsp.rs:113-165),
//                     lsp2mgc = lsp2lpc -> ignorm -> x(-stage) -> mgc2mgc (lsp.rs:27-107,
//                     generalized.rs:6-38, cepstrum.rs:69-103), mc2b, gnorm, x gamma (mod.rs:92-106,148-156)
//   k_vocoder_mglsa   per sample: x *= c[0]; `stage` passes of dff (mglsa.rs:15-41); c += cinc (mod.rs:162-172)
// Off every BASELINE configuration (the nitech voice has no GAMMA option: stage 0) and PARITY
// UNPINNED -- no reference test reaches this branch; the CPU restatement it is tested against is held
// by identities only (tests/).  Built for completeness of the drop-in, not
// for speed: one thread per frame for the coefficients, one wave per time-chunk for the filter,
// riding the chunk / hand-off check / redo machinery of the Stage::Zero kernels.
//
// Filter mapping (lane = tap i, nmcp <= 64).  A dff pass updates d[i] += alpha * (d[i+1] - d[i-1]) in
// ascending i with d[i-1] already updated: d'[i] = (d[i] + alpha d[i+1]) - alpha d'[i-1], a first-order
// linear recurrence with the constant ratio -alpha -- a weighted scan over the wave (DPP row_shr
// 1/2/4/8, row_bcast 15/31: the sums are re-associated w.r.t. the reference, ~1e-14 relative).  The dot
// product with c[i+1] is a DPP sum; the stages are sequential (each subtracts its y from x).
#include <atomic>
#include "jb_device.h"

#include <algorithm>

namespace jb {

constexpr int kSgMaxN = 64; // nmcp <= 64 (check_voice: <= 61)

// ---- per-frame coefficients: the reference's loops, one thread per frame ------------------------
struct SgScratch {
    double lsp[kSgMaxN], mgc[kSgMaxN], lpc[kSgMaxN + 1], t[kSgMaxN + 1];
    double a0[kSgMaxN / 2 + 2], a1[kSgMaxN / 2 + 2], a2[kSgMaxN / 2 + 2];
    double b0[kSgMaxN / 2 + 2], b1[kSgMaxN / 2 + 2], b2[kSgMaxN / 2 + 2];
    double p[kSgMaxN / 2 + 2], q[kSgMaxN / 2 + 2];
};

// lsp2lpc (lsp.rs:27-94): ALL m entries of the buffer are taken as frequencies, as the reference does
__device__ void sg_lsp2lpc(const double *lsp, int m, double *out, SgScratch &w)
{
    const int mh1 = (m % 2 == 0) ? m / 2 : (m + 1) / 2, mh2 = (m % 2 == 0) ? m / 2 : (m - 1) / 2;
    for (int i = 0; i <= mh1; i++)
        w.a0[i] = w.a1[i] = w.a2[i] = 0.0;
    for (int i = 0; i <= mh2; i++)
        w.b0[i] = w.b1[i] = w.b2[i] = 0.0;
    for (int i = 0, k = 0; k < m; i++, k += 2)
        w.p[i] = -2.0 * cos(lsp[k]);
    for (int i = 0, k = 1; k < m; i++, k += 2)
        w.q[i] = -2.0 * cos(lsp[k]);
    double xff = 0.0, xf = 0.0;
    for (int i = 0; i <= m; i++)
        out[i] = 0.0;
    for (int k = 0; k <= m; k++) {
        const double xx = k == 0 ? 1.0 : 0.0;
        if (m % 2 == 1) {
            w.a0[0] = xx;
            w.b0[0] = xx - xff;
            xff = xf;
            xf = xx;
        } else {
            w.a0[0] = xx + xf;
            w.b0[0] = xx - xf;
            xf = xx;
        }
        for (int i = 0; i < mh1; i++) {
            w.a0[i + 1] = w.a0[i] + w.p[i] * w.a1[i] + w.a2[i];
            w.a2[i] = w.a1[i];
            w.a1[i] = w.a0[i];
        }
        for (int i = 0; i < mh2; i++) {
            w.b0[i + 1] = w.b0[i] + w.q[i] * w.b1[i] + w.b2[i];
            w.b2[i] = w.b1[i];
            w.b1[i] = w.b0[i];
        }
        if (k > 0)
            out[k - 1] = -0.5 * (w.a0[mh1] + w.b0[mh2]);
    }
    for (int i = m; i-- > 0;)
        out[i + 1] = -out[i];
    out[0] = 1.0;
}

// gnorm / ignorm (generalized.rs:6-38), gamma != 0
__device__ void sg_gnorm(double *c, int n, double gamma)
{
    const double k = 1.0 + gamma * c[0];
    c[0] = pow(k, 1.0 / gamma);
    for (int i = 1; i < n; i++)
        c[i] = c[i] / k;
}
__device__ void sg_ignorm(double *c, int n, double gamma)
{
    const double k = pow(c[0], gamma);
    c[0] = (k - 1.0) / gamma;
    for (int i = 1; i < n; i++)
        c[i] = c[i] * k;
}

// lsp2mgc (lsp.rs:96-107): lsp[n] -> mgc[n]; the alpha of the LPC equals the target alpha, so mgc2mgc
// (cepstrum.rs:94-102) is gnorm -> gc2gc -> ignorm with both gammas equal
__device__ void sg_lsp2mgc(const double *lsp, int n, int use_log_gain, int stage, double gamma, double *mgc,
                           SgScratch &w)
{
    sg_lsp2lpc(lsp, n, w.lpc, w);
    w.lpc[0] = use_log_gain ? exp(lsp[0]) : lsp[0];
    sg_ignorm(w.lpc, n + 1, gamma);
    for (int i = 1; i < n + 1; i++)
        w.lpc[i] *= -(double)stage;
    sg_gnorm(w.lpc, n + 1, gamma);
    // gc2gc (cepstrum.rs:69-92): c1 = lpc[n + 1] -> mgc[m2 + 1], m2 = n - 1
    const int n1 = n + 1, m2 = n - 1;
    mgc[0] = w.lpc[0];
    for (int i = 1; i <= m2; i++) {
        double ss1 = 0.0, ss2 = 0.0;
        const int kend = n1 < i ? n1 : i;
        for (int k = 1; k < kend; k++) {
            const int mk = i - k;
            const double cc = w.lpc[k] * mgc[mk];
            ss1 += (double)mk * cc;
            ss2 += (double)k * cc;
        }
        if (i < n1)
            mgc[i] = w.lpc[i] + (gamma * ss2 - gamma * ss1) / (double)i;
        else
            mgc[i] = (gamma * ss2 - gamma * ss1) / (double)i;
    }
    sg_ignorm(mgc, n, gamma);
}

__device__ double sg_lsp2en(const double *lsp, int n, int use_log_gain, int stage, double gamma, SgScratch &w)
{
    sg_lsp2mgc(lsp, n, use_log_gain, stage, gamma, w.mgc, w);
    double e = 0.0;
    for (int i = 0; i < n; i++)
        e += w.mgc[i] * w.mgc[i];
    return e;
}

// coefficients of one frame (mod.rs:92-106 / 148-156) into cc[n]
__device__ void sg_frame_coef(const double *spec, int n, double alpha, double beta, int use_log_gain, int stage,
                              bool filtered, double *cc, SgScratch &w)
{
    const double gamma = -1.0 / (double)stage; // stage.rs:31
    for (int i = 0; i < n; i++)
        w.lsp[i] = spec[i];
    if (filtered) {
        // postfilter_lsp (lsp.rs:113-139)
        if (beta > 0.0 && n > 2) {
            const double en1 = sg_lsp2en(w.lsp, n, use_log_gain, stage, gamma, w);
            for (int i = 0; i < n; i++) {
                if (i > 1 && i < n - 1) {
                    const double d1 = beta * (w.lsp[i + 1] - w.lsp[i]);
                    const double d2 = beta * (w.lsp[i] - w.lsp[i - 1]);
                    w.t[i] = w.lsp[i - 1] + d2 +
                             (d2 * d2 * ((w.lsp[i + 1] - w.lsp[i - 1]) - (d1 + d2))) / ((d2 * d2) + (d1 * d1));
                } else {
                    w.t[i] = w.lsp[i];
                }
            }
            for (int i = 0; i < n; i++)
                w.lsp[i] = w.t[i];
            const double en2 = sg_lsp2en(w.lsp, n, use_log_gain, stage, gamma, w);
            if (en1 != en2) {
                if (use_log_gain)
                    w.lsp[0] += 0.5 * log(en1 / en2);
                else
                    w.lsp[0] *= sqrt(en1 / en2);
            }
        }
        // check_lsp_stability (lsp.rs:141-165)
        const double PI = 3.14159265358979323846;
        const double mn = 0.25 * PI / (double)n;
        const int last = n - 1;
        for (int it = 0; it < 4; it++) {
            bool find = false;
            for (int j = 1; j < last; j++) {
                const double tmp = w.lsp[j + 1] - w.lsp[j];
                if (tmp < mn) {
                    w.lsp[j] -= 0.5 * (mn - tmp);
                    w.lsp[j + 1] += 0.5 * (mn - tmp);
                    find = true;
                }
            }
            if (w.lsp[1] < mn) {
                w.lsp[1] = mn;
                find = true;
            }
            if (w.lsp[last] > PI - mn) {
                w.lsp[last] = PI - mn;
                find = true;
            }
            if (!find)
                break;
        }
    }
    sg_lsp2mgc(w.lsp, n, use_log_gain, stage, gamma, w.mgc, w);
    // mc2b (cepstrum.rs:139-149)
    if (alpha != 0.0) {
        cc[n - 1] = w.mgc[n - 1];
        for (int i = n - 2; i >= 0; i--)
            cc[i] = w.mgc[i] - alpha * cc[i + 1];
    } else {
        for (int i = 0; i < n; i++)
            cc[i] = w.mgc[i];
    }
    sg_gnorm(cc, n, gamma);
    for (int i = 1; i < n; i++)
        cc[i] *= gamma;
}

__global__ __launch_bounds__(64) void k_stage_coef(BatchDev bd, VocDev vd)
{
    const int b = blockIdx.y;
    const UttDev *u = bd.utt + b;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= u->T)
        return;
    const int n = vd.nmcp;
    const uint64_t f = u->frame_off + t;
    SgScratch w;
    double cc[kSgMaxN];
    sg_frame_coef(vd.mcp + f * (uint64_t)n, n, vd.alpha, vd.beta_stage, vd.use_log_gain, vd.stage, true, cc, w);
    for (int i = 0; i < n; i++)
        vd.bcoef[f * (uint64_t)n + i] = cc[i];
    if (t == 0) {
        // the first frame starts from the coefficients of the un-filtered spectrum (mod.rs:92-106)
        sg_frame_coef(vd.mcp + f * (uint64_t)n, n, vd.alpha, vd.beta_stage, vd.use_log_gain, vd.stage, false, cc, w);
        for (int i = 0; i < n; i++)
            vd.bfirst[(uint64_t)b * (uint64_t)n + i] = cc[i];
    }
}

hipError_t launch_stage_coef(const BatchDev &bd, const VocDev &vd, hipStream_t stream)
{
    if (bd.B == 0 || bd.maxT == 0)
        return hipSuccess;
    dim3 grid((bd.maxT + 63) / 64, bd.B), block(64);
    hipLaunchKernelGGL(k_stage_coef, grid, block, 0, stream, bd, vd);
    return hipGetLastError();
}

// ---- the filter ---------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ double sg_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sg_readlane(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// state layout (doubles): d[stage][64], lane = tap
int mglsa_state_doubles(int stage) { return 64 * stage; }

// STAGE = 1..8: the delay lines of the stages in registers, four waves (items) per workgroup.  STAGE = 0: any number
// of stages (Stage::NonZero is generic in it, stage.rs:24-39), the delay lines in the wave's rows of dynamic LDS
// ([stage][64] doubles per wave, lane = tap: no lane reads another's row), `wpb` waves per workgroup.
constexpr int kSgRegStages = 8;
constexpr int kSgMaxStage = 256; // one wave per workgroup: 256 x 512 B = 128 KB of a CU's 160 KB
template <int STAGE>
__global__ __launch_bounds__(256) void k_vocoder_mglsa(BatchDev bd, VocDev vd, const VocWork *__restrict__ work,
                                                       uint32_t n_items, uint32_t wpb)
{
    extern __shared__ double sg_lds[];
    const uint32_t item = blockIdx.x * wpb + (threadIdx.x >> 6);
    if (item >= n_items)
        return;
    const VocWork wk = work[item];
    const int b = (int)wk.utt;
    const uint32_t T = bd.utt[b].T;
    const uint32_t t_begin = wk.t_start, t_out = wk.t_out;
    uint32_t t_end = wk.t_end;
    if (t_end > T)
        t_end = T;
    if (t_begin >= t_end)
        return;
    const int lane = threadIdx.x & 63;
    const uint64_t base = bd.utt[b].frame_off;
    const int n = vd.nmcp, fp = vd.fperiod, bs = vd.bs, nblk = vd.nblk;
    const double a = vd.alpha, na = -a, aa = 1.0 - a * a, vol = vd.volume;
    // scan weights: lane i folds lane i - s with (-a)^s where that lane exists (recurrence over taps 0..n-2)
    auto pw = [&](int s) {
        double r = 1.0;
        for (int k = 0; k < s; k++)
            r *= na;
        return r;
    };
    const int r16 = lane & 15;
    const double c1 = r16 >= 1 ? pw(1) : 0.0, c2 = r16 >= 2 ? pw(2) : 0.0, c4 = r16 >= 4 ? pw(4) : 0.0,
                 c8 = r16 >= 8 ? pw(8) : 0.0;
    // row_bcast15 hands lane 15 of a row to the next row: taken by rows 1 and 3 only (distance r16 + 1);
    // row_bcast31 hands lane 31 -- by then the sum over rows 0 and 1 -- to rows 2 and 3 (distance lane - 31);
    // row 3 has row 2's sum from the first step, row 2 takes both rows 0 and 1 from the second
    const int row = lane >> 4;
    const double cb15 = (row == 1 || row == 3) ? pw(r16 + 1) : 0.0, cb31 = lane >= 32 ? pw(lane - 31) : 0.0;
    const bool upd = lane >= 1 && lane <= n - 2; // taps the in-place update touches
    const bool dot = lane <= n - 2;              // taps of the dot product (with c[lane + 1])

    const int nstage = STAGE ? STAGE : vd.stage;
    constexpr int kUn = STAGE ? STAGE : 1; // (a run-time stage count: the stage loops stay loops)
    double dreg[STAGE ? STAGE : 1];
    double *const dl = sg_lds + (size_t)(threadIdx.x >> 6) * 64 * (size_t)(STAGE ? 0 : nstage) + lane;
#define SG_D(s) (*(STAGE ? &dreg[STAGE ? (s) : 0] : &dl[64 * (s)]))
#pragma unroll kUn
    for (int s = 0; s < nstage; s++)
        SG_D(s) = wk.load_state ? wk.load_state[64 * s + lane] : 0.0;
    auto save_state = [&](double *sp) {
#pragma unroll kUn
        for (int s = 0; s < nstage; s++)
            sp[64 * s + lane] = SG_D(s);
    };

    for (uint32_t t = t_begin; t < t_end; t++) {
        const uint64_t f = base + t;
        const bool emit = t >= t_out; // warm-up frames are computed but not stored
        if (t == t_out && t_out > t_begin && wk.save_warm)
            save_state(wk.save_warm);
        if (t == t_out + vd.ckpt_frames && wk.save_ckpt)
            save_state(wk.save_ckpt);
        if (vd.ckpt2_frames && t == t_out + vd.ckpt2_frames && wk.save_ckpt2)
            save_state(wk.save_ckpt2);
        // frame setup (mod.rs:148-161): c = previous frame's cc (first frame: the un-filtered one)
        const double *bcur = vd.bcoef + f * (uint64_t)n;
        const double *bprev = (t > 0) ? bcur - n : vd.bfirst + (uint64_t)b * (uint64_t)n;
        double ck = dot ? bprev[lane + 1] : 0.0; // c[lane + 1]
        const double ckt = dot ? bcur[lane + 1] : 0.0;
        const double ckinc = (ckt - ck) / (double)fp;
        double c0 = bprev[0];
        const double c0t = bcur[0], c0inc = (c0t - c0) / (double)fp;
        for (int q = 0; q < nblk; q++) {
            const uint64_t n0 = (uint64_t)t * (uint64_t)fp + (uint64_t)(q * bs);
            const int blen = min(bs, fp - q * bs); // (a shorter last block where bs does not divide the frame period)
            const double xin = lane < blen ? exc_block_ptr(vd, base, t, exc_code(vd, base, t), q)[q * bs + lane] : 0.0;
            double ob = 0.0;
            for (int i = 0; i < blen; i++) {
                double x = sg_readlane(xin, i) * c0; // x *= coefficients[0] (mod.rs:164)
#pragma unroll kUn
                for (int s = 0; s < nstage; s++) {
                    // dff (mglsa.rs:23-41)
                    const double ds = SG_D(s);
                    const double dn1 = sg_dpp<0x130>(ds); // d[i + 1] (wave_shl:1)
                    double e = upd ? fma(a, dn1, ds) : (lane == 0 ? ds : 0.0);
                    // d'[i] = e[i] - a d'[i - 1]: weighted inclusive scan
                    e = fma(c1, sg_dpp<0x111>(e), e);
                    e = fma(c2, sg_dpp<0x112>(e), e);
                    e = fma(c4, sg_dpp<0x114>(e), e);
                    e = fma(c8, sg_dpp<0x118>(e), e);
                    e = fma(cb15, sg_dpp<0x142>(e), e);
                    e = fma(cb31, sg_dpp<0x143>(e), e);
                    const double dnew = upd ? e : ds; // taps 0 and n - 1 (and idle lanes) keep their value
                    double y = dot ? dnew * ck : 0.0;
                    y += sg_dpp<0x111>(y);
                    y += sg_dpp<0x112>(y);
                    y += sg_dpp<0x114>(y);
                    y += sg_dpp<0x118>(y);
                    y += sg_dpp<0x142>(y);
                    y += sg_dpp<0x143>(y);
                    x -= sg_readlane(y, 63);
                    // shift by one tap, new head (mglsa.rs:36-40)
                    const double dm1 = sg_dpp<0x138>(dnew); // d[i - 1] (wave_shr:1)
                    SG_D(s) = lane == 0 ? fma(a, dnew, aa * x) : (lane <= n - 1 ? dm1 : 0.0);
                }
                ck += ckinc;
                c0 += c0inc;
                ob = (lane == i) ? x * vol : ob;
            }
            if (lane < blen && emit) {
                if (vd.pcm16) {
                    double v = fmin(ob, 32767.0);
                    v = fmax(v, -32768.0);
                    vd.pcm16[base * (uint64_t)fp + n0 + (uint64_t)lane] = (int16_t)(int)v;
                } else {
                    vd.pcm[base * (uint64_t)fp + n0 + (uint64_t)lane] = ob;
                }
            }
        }
    }
    if (wk.save_end)
        save_state(wk.save_end);
#undef SG_D
}

int mglsa_max_stage() { return kSgMaxStage; }

hipError_t launch_vocoder_mglsa(const BatchDev &bd, const VocDev &vd, const VocWork *work_dev, uint32_t n_items,
                                hipStream_t stream)
{
    if (n_items == 0)
        return hipSuccess;
    dim3 grid((n_items + 3) / 4), block(256);
    if (vd.stage > kSgRegStages) {
        if (vd.stage > kSgMaxStage)
            return hipErrorInvalidValue;
        // as many waves per workgroup as 128 KB of delay lines hold (stage 9..64: four)
        const size_t per_wave = (size_t)vd.stage * 64 * sizeof(double);
        uint32_t wpb = (uint32_t)std::min<size_t>(4, (128u << 10) / per_wave);
        if (wpb == 3)
            wpb = 2;
        // function attributes are per DEVICE and jb_multi drives several devices from one process: asked once per
        // device, and a failure is not remembered
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
            return hipErrorInvalidDevice;
        if (!attr_set[dev].load(std::memory_order_acquire)) {
            const hipError_t attr = hipFuncSetAttribute((const void *)k_vocoder_mglsa<0>,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
            if (attr != hipSuccess)
                return attr;
            attr_set[dev].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL(k_vocoder_mglsa<0>, dim3((n_items + wpb - 1) / wpb), dim3(64 * wpb), per_wave * wpb, stream, bd,
                           vd, work_dev, n_items, wpb);
        return hipGetLastError();
    }
    switch (vd.stage) {
#define JB_SG_CASE(S)                                                                                              \
    case S:                                                                                                        \
        hipLaunchKernelGGL(k_vocoder_mglsa<S>, grid, block, 0, stream, bd, vd, work_dev, n_items, 4u);             \
        break;
        JB_SG_CASE(1)
        JB_SG_CASE(2)
        JB_SG_CASE(3)
        JB_SG_CASE(4)
        JB_SG_CASE(5)
        JB_SG_CASE(6)
        JB_SG_CASE(7)
        JB_SG_CASE(8)
#undef JB_SG_CASE
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace jb
