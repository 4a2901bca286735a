// jb_vocoder.hip -- mixed excitation + MLSA cascade kernels for gfx950 (CDNA4).
//
// Restates /root/reference/src/vocoder (Stage::Zero) and src/speech.rs:
//   k_pitch     V2  pitch from lf0                  (vocoder/mod.rs:73-77)
//   k_mc2b      V2  mc2b                            (cepstrum.rs:139-149)
//   k_excite    V3+V5 mixed excitation (ring buffer as a feed-forward FIR) and gain
//   k_pulse     V3  pulse scheduler, per voiced run (excitation.rs:25-33,73-81,102-104)
//   k_vocoder   V3 (ring buffer as a 31-tap feed-forward FIR), V5 gain,
//               V6 df1, V7 df2/fir, V8 coefficient interpolation, V9 frame loop.
//
// k_vocoder design (one wave64 per utterance, the recursion is time-serial):
//   * the 5 Pade stages x (nmcp-1) all-pass taps of df2 live in registers,
//     lane = 12*stage + group, TPL consecutive taps per lane (60 lanes busy);
//   * the tap recursion rem' = d - a*rem has a constant ratio, so a lane's TPL
//     taps collapse to (loc, kappa=(-a)^TPL) and the carry across the 12 lanes of a
//     stage is a segmented weighted Kogge-Stone scan done with DPP row_shr 1/2/4/8
//     plus one row_bcast:15 step (a 12-lane segment straddles at most one 16-lane
//     DPP row boundary); per-lane coefficient registers encode segment ends;
//   * the dot product with c[2..] is the same scan with weights 1;
//   * a stage's output reaches the next stage's head lane with one wave_shr:1;
//   * the Pade combine broadcasts the five stage sums with v_readlane;
//   * excitation for a block of `bs` samples is computed lane-parallel
//     (lane = sample) into a VGPR and fed to the serial loop with v_readlane;
//     PCM is collected lane = sample and stored coalesced, bs*8 bytes per store.
// All arithmetic is f64; fused multiply-adds are written explicitly (the TU is
// compiled with -ffp-contract=off), sums are re-associated w.r.t. the reference
// (tolerance: see DESIGN.md; measured ~1e-13 relative).
#include "jb_device.h"

#include <cstdlib>
#include <cstring>

namespace jb {

// --------------------------------------------------------------------------
// V2a: pitch from lf0, thread per frame (vocoder/mod.rs:73-77).
__global__ void k_pitch(BatchDev bd, VocDev vd)
{
    const int b = blockIdx.y;
    const UttDev *u = bd.utt + b;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= u->T)
        return;
    const uint64_t f = u->frame_off + t;
    const double MAX_LF0 = 9.903487552536127, MIN_LF0 = 2.995732273553991; // constants.rs:4-6
    auto pitch_of = [&](double l) {
        if (l == kNoData)
            return 0.0;
        const double cl = l < MIN_LF0 ? MIN_LF0 : (l > MAX_LF0 ? MAX_LF0 : l);
        return (double)vd.fs / exp(cl);
    };
    const double p = pitch_of(vd.lf0[f]);
    const double prevp = t > 0 ? pitch_of(vd.lf0[f - 1]) : 0.0;
    vd.pitch[f] = p;
    // Excitation::start (excitation.rs:25-33) of a voiced frame: pitch_of_curr_point is the previous frame's pitch
    // (Excitation::end, :102-104) unless that frame was unvoiced; written here, coalesced, and not by the pulse
    // walk, whose lanes are one voiced run each (64 cache lines per store instruction)
    vd.cur_start[f] = prevp != 0.0 ? prevp : p;
    vd.pinc[f] = (prevp != 0.0 && p != 0.0) ? (p - prevp) / (double)vd.fperiod : 0.0;
}

// V2b: mc2b (cepstrum.rs:139-149).  64 frames per block staged through LDS so that both
// the read of mcp and the write of bcoef are coalesced; each thread then runs the
// recurrence b[i] = c[i] - alpha*b[i+1] of its own frame out of LDS (row stride 65: no
// bank conflicts for column access).
constexpr int kMc2bFrames = 64;
constexpr int kMc2bMax = 64; // nmcp <= 61

__global__ __launch_bounds__(kMc2bFrames) void k_mc2b(BatchDev bd, VocDev vd)
{
    const int b = blockIdx.y;
    const UttDev *u = bd.utt + b;
    const uint32_t t0 = blockIdx.x * kMc2bFrames;
    if (t0 >= u->T)
        return;
    const int n = vd.nmcp;
    const uint32_t nf = min((uint32_t)kMc2bFrames, u->T - t0);
    const uint64_t off = (u->frame_off + t0) * (uint64_t)n;
    const uint32_t tot = nf * (uint32_t)n;
    __shared__ double tile[kMc2bMax][kMc2bFrames + 1]; // [coef][frame]
    const int tid = threadIdx.x;
    for (uint32_t e = tid; e < tot; e += kMc2bFrames)
        tile[e % n][e / n] = vd.mcp[off + e];
    __syncthreads();
    if ((uint32_t)tid < nf && vd.alpha != 0.0) {
        double prev = tile[n - 1][tid];
        for (int i = n - 2; i >= 0; i--) {
            prev = tile[i][tid] - vd.alpha * prev;
            tile[i][tid] = prev;
        }
    }
    __syncthreads();
    for (uint32_t e = tid; e < tot; e += kMc2bFrames)
        vd.bcoef[off + e] = tile[e % n][e / n];
}

// --------------------------------------------------------------------------
// V3 pulse scheduler.  One lane per voiced run, walking its run (the counter is the only
// state carried across frames: at every
// frame start pitch_of_curr_point is reset to the previous frame's pitch by
// Excitation::end, excitation.rs:102-104).
// ---- one frame of the walk in closed form (round 4; CPU model and proof by exhaustion against the loop:
// tests/tools/pulse_closed_form.py, 0 mismatches of fires and carried counter in 300 k stress frames) ----
// Excitation::get's voiced branch (excitation.rs:73-81) runs, per sample,
//     counter += 1; fire = counter >= cur; if fire: counter -= cur; cur += inc
// and only `counter` is carried from frame to frame.  Two facts make a frame cost O(pulses), not O(samples):
//   * while cur stays in one binade, fl(cur + inc) = cur + d with d = RN_ulp(inc): cur_k = cur_0 + k*d exactly
//     (a tie -- inc = (p - p')/fperiod is one once in fperiod frames, p - p' being a multiple of the ulp --
//     rounds to even: from an even mantissa on, d is constant too);
//   * n increments of the counter are exact whenever c + n is representable (all c + k, k <= n, then are:
//     they share c's fractional bits and are no larger): one TwoSum tells.  Otherwise the walk stays below the
//     top of the counter's binade and takes the crossing as one rounded addition, as the loop does.
// The first fire of a stretch is estimated by a division and settled by evaluating both (exact) sides.  A
// sample whose cur cannot start a segment (tie from an odd mantissa, |d| out of range, denormal range) is
// taken the loop's way.  Round 2's event-driven form stopped at EVERY binade top of the counter (~9 per pitch
// period, 15 iterations per frame: slower than the loop on a wave in lockstep); here the exactness test makes
// the usual stretch the rest of the frame: 1.6 iterations per frame on the synthetic utterances.
__device__ __forceinline__ int pw_expo(double x) { return (int)(((unsigned long long)__double_as_longlong(x) >> 52) & 0x7ffull); }
__device__ __forceinline__ double pw_pow2e(int e) { return __longlong_as_double((long long)e << 52); } // biased exponent e
__device__ __forceinline__ bool pw_add_exact(double a, double b)
{
    const double s = a + b, bb = s - a;
    return (a - (s - bb)) + (b - bb) == 0.0;
}
__device__ __forceinline__ void pw_set(unsigned long long (&w)[4], int jj, int bs)
{
    int q = 0;
    while (jj >= bs) {
        jj -= bs;
        q++;
    }
    const unsigned long long bit = 1ull << jj;
    w[0] |= q == 0 ? bit : 0ull;
    w[1] |= q == 1 ? bit : 0ull;
    w[2] |= q == 2 ? bit : 0ull;
    w[3] |= q == 3 ? bit : 0ull;
}
__device__ __forceinline__ void pulse_frame_closed(double &counter, const double u0, const double inc, const int fp,
                                                   const int bs, unsigned long long (&w)[4])
{
    double c = counter, us = u0; // us: cur at sample j (exact)
    int j = 0;
    while (j < fp) {
        // ---- the cur-segment that starts at sample j: cur_{j+k} = us + k*d for k = 0..kk ----
        const int e = pw_expo(us);
        bool ok = us > 0.0 && e > 60;
        double d = 0.0;
        if (ok) {
            const double u1 = us + inc;
            d = u1 - us;
            const double hU = pw_pow2e(e - 53); // half an ulp of the binade
            const double r = fabs(inc - d);
            ok = pw_expo(u1) == e && d > -8.0 && d < 0.9 &&
                 (r < hU || (r == hU && (__double_as_longlong(us) & 1ll) == 0));
        }
        if (!ok) {
            c = c + 1.0;
            if (c >= us) {
                pw_set(w, j, bs);
                c = c - us;
            }
            us = us + inc;
            j++;
            continue;
        }
        const int rem = fp - 1 - j;
        const double top = pw_pow2e(e + 1), bot = pw_pow2e(e);
        int kk = rem;
        {
            const double ul = us + (double)rem * d;
            if (!(ul < top && ul >= bot)) {
                // the segment ends inside the frame: last k with cur_k still in the binade
                if (d > 0.0) {
                    const double q = (top - us) / d;
                    kk = q > (double)(rem + 2) ? rem : min(max((int)ceil(q) - 1, 0), rem);
                    while (kk > 0 && !(us + (double)kk * d < top))
                        kk--;
                    while (kk < rem && (us + (double)(kk + 1) * d < top))
                        kk++;
                } else {
                    const double q = (us - bot) / -d;
                    kk = q > (double)(rem + 2) ? rem : min(max((int)floor(q), 0), rem);
                    while (kk > 0 && !(us + (double)kk * d >= bot))
                        kk--;
                    while (kk < rem && (us + (double)(kk + 1) * d >= bot))
                        kk++;
                }
            }
        }
        const int jend = j + kk, js = j;
        const double rcp = 1.0 / (1.0 - d);
        // ---- the counter through the segment, a stretch of exact increments at a time ----
        while (j <= jend) {
            const int lim = jend - j + 1;
            const double uj = us + (double)(j - js) * d;
            int M;
            if (pw_add_exact(c, (double)lim))
                M = lim;
            else if (c >= 1.0)
                M = min((int)ceil(pw_pow2e(pw_expo(c) + 1) - c) - 1, lim);
            else
                M = 0;
            if (M <= 0) {
                c = c + 1.0;
                if (c >= uj) {
                    pw_set(w, j, bs);
                    c = c - uj;
                }
                j++;
                continue;
            }
            // first m in [0, M) with  c + (m + 1) >= uj + m*d  (both sides exact), M if none
            const double a = uj - c - 1.0;
            int m = 0;
            if (a > 0.0) {
                const double est = ceil(a * rcp);
                m = est < (double)M ? (int)est : M;
            }
            while (m > 0 && (c + (double)m) >= (uj + (double)(m - 1) * d))
                m--;
            while (m < M && !((c + (double)(m + 1)) >= (uj + (double)m * d)))
                m++;
            if (m < M) {
                c = (c + (double)(m + 1)) - (uj + (double)m * d);
                pw_set(w, j + m, bs);
                j += m + 1;
            } else {
                c = c + (double)M;
                j += M;
            }
        }
        // leave the segment: the update that crosses the binade is a rounded addition
        us = (us + (double)(jend - js) * d) + inc;
    }
    counter = c;
}

// One voiced run (entry r of utterance b's compact run list from k_prep_states).
__device__ __forceinline__ void pulse_run(const BatchDev &bd, const VocDev &vd, int b, uint32_t r)
{
    const UttDev u = bd.utt[b];
    const uint32_t t0 = vd.run_list[u.state_off + r];
    if (t0 >= u.T)
        return;
    const uint64_t base = u.frame_off;
    const int fp = vd.fperiod, bs = vd.bs, nblk = vd.nblk;
    if (vd.pitch[base + t0] == 0.0)
        return; // a zero-length voiced state followed by an unvoiced one
    if (t0 > 0 && vd.pitch[base + t0 - 1] != 0.0)
        return; // continuation of the previous list entry (zero-length states in between)
    double prevp = 0.0, counter = 0.0;
    if (nblk <= 4) {
        // frames in blocks of eight: the next block's pitch values are requested before this block is walked
        // (a frame is ~0.5 us of work now, less than the latency of its own load)
        constexpr int PB = 8;
        const double *pp = vd.pitch + base;
        const uint32_t T = u.T;
        double pb[PB], pn[PB];
#pragma unroll
        for (int k = 0; k < PB; k++)
            pb[k] = pp[min(t0 + (uint32_t)k, T - 1)];
        for (uint32_t tb = t0; tb < T; tb += PB) {
#pragma unroll
            for (int k = 0; k < PB; k++)
                pn[k] = pp[min(tb + (uint32_t)(PB + k), T - 1)];
#pragma unroll
            for (int k = 0; k < PB; k++) {
                const uint32_t t = tb + (uint32_t)k;
                const double p = t < T ? pb[k] : 0.0;
                if (p == 0.0)
                    return;
                double cur, inc;
                // Excitation::start (excitation.rs:25-33)
                if (prevp != 0.0) {
                    cur = prevp;
                    inc = (p - prevp) / (double)fp;
                } else {
                    inc = 0.0;
                    cur = p;
                    counter = p;
                }
                unsigned long long w[4] = {0ull, 0ull, 0ull, 0ull};
                pulse_frame_closed(counter, cur, inc, fp, bs, w);
                // (cur_start, pinc: k_pitch.)  The frame's mask words in as few stores as they fit: every lane is
                // another run, so every store instruction of the wave touches 64 cache lines
                unsigned long long *pm = vd.pmask + (base + t) * (uint64_t)nblk;
                if (nblk == 4) {
                    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                    u64x2 lo, hi;
                    lo.x = w[0]; lo.y = w[1]; hi.x = w[2]; hi.y = w[3];
                    *reinterpret_cast<u64x2 *>(pm) = lo;
                    *reinterpret_cast<u64x2 *>(pm + 2) = hi;
                } else {
                    for (int q = 0; q < nblk; q++)
                        pm[q] = w[q];
                }
                prevp = p; // Excitation::end
            }
#pragma unroll
            for (int k = 0; k < PB; k++)
                pb[k] = pn[k];
        }
        return;
    }
    for (uint32_t t = t0; t < u.T; t++) {
        const double p = vd.pitch[base + t];
        if (p == 0.0)
            break;
        double cur, inc;
        // Excitation::start (excitation.rs:25-33)
        if (prevp != 0.0) {
            cur = prevp;
            inc = (p - prevp) / (double)fp;
        } else {
            inc = 0.0;
            cur = p;
            counter = p;
        }
        for (int q = 0; q < nblk; q++) {
            // voiced branch of Excitation::get (excitation.rs:73-81), sample by sample (frame periods with more
            // than four mask words: the closed form above keeps a frame's words in registers); the pulse bits
            // are gathered in two 32-bit halves
            uint32_t mlo = 0, mhi = 0;
            const int blen = min(bs, fp - q * bs); // (the last block of a frame period that bs does not divide is shorter)
            const int b32 = blen < 32 ? blen : 32;
#pragma unroll 4
            for (int j = 0; j < b32; j++) {
                counter += 1.0;
                const bool fire = counter >= cur;
                counter = fire ? counter - cur : counter;
                mlo |= (uint32_t)fire << j;
                cur += inc;
            }
#pragma unroll 4
            for (int j = 32; j < blen; j++) {
                counter += 1.0;
                const bool fire = counter >= cur;
                counter = fire ? counter - cur : counter;
                mhi |= (uint32_t)fire << (j - 32);
                cur += inc;
            }
            vd.pmask[(base + t) * nblk + q] = ((unsigned long long)mhi << 32) | mlo;
        }
        prevp = p; // Excitation::end
    }
}

// Exclusive prefix of the per-utterance run counts (one block), and the counter reset.
__global__ __launch_bounds__(256) void k_run_scan(BatchDev bd, VocDev vd)
{
    __shared__ uint32_t part[256];
    const int tid = threadIdx.x;
    const int per = (bd.B + 255) / 256;
    uint32_t s = 0;
    for (int i = 0; i < per; i++) {
        const int b = tid * per + i;
        if (b < bd.B)
            s += vd.nruns[bd.order[b]];
    }
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int i = 0; i < 256; i++) {
            const uint32_t v = part[i];
            part[i] = acc;
            acc += v;
        }
        vd.run_base[bd.B] = acc;
        *vd.run_counter = 0;
    }
    __syncthreads();
    uint32_t acc = part[tid];
    for (int i = 0; i < per; i++) {
        const int b = tid * per + i;
        if (b < bd.B) {
            vd.run_base[b] = acc; // position b of the longest-first order
            acc += vd.nruns[bd.order[b]];
        }
    }
}

// Work-queue form: a fixed number of lanes, each taking the next run (utterances longest first)
// from an atomic counter until none is left, so that lanes stay busy and the walk occupies a few
// hundred waves instead of thousands.  Every lane reaches the exit condition (counter >= total).
__global__ __launch_bounds__(64) void k_pulse_queue(BatchDev bd, VocDev vd)
{
    const uint32_t total = vd.run_base[bd.B];
    for (;;) {
        // 64 runs per reservation: one atomic per wave and turn, not one per lane (they all hit one address).
        // The whole wave is here together every turn -- the reservation is read with readfirstlane --: a lane
        // without a run of its own skips the walk, nothing else.
        uint32_t idx0 = 0;
        if (threadIdx.x == 0)
            idx0 = atomicAdd(vd.run_counter, 64u);
        idx0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx0);
        if (idx0 >= total)
            return; // wave-uniform: every wave reaches this once the queue is empty
        const uint32_t idx = idx0 + threadIdx.x;
        if (idx < total) {
            // utterance position: last p with run_base[p] <= idx
            int lo = 0, hi = bd.B - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (vd.run_base[mid] <= idx)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            pulse_run(bd, vd, (int)bd.order[lo], idx - vd.run_base[lo]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// --------------------------------------------------------------------------
// V3 + V5: mixed excitation and gain, fully parallel over samples.
// The reference's 31-slot ring buffer (excitation.rs:43-100) is feed-forward:
//   x[n] = noise[n-15] + sum_k e[n-k] * lpf_{frame(n-k)}[k],
//   e[m] = voiced(frame(m)) ? pulse[m] - noise[m] : 0          
// The gain x *= exp(c0(n)) (vocoder/mod.rs:129-131) needs the MCP stream and is applied by
// the vocoder kernels, so this kernel depends on the LF0 and LPF streams only.
// Block = 256 consecutive samples of one utterance; e[] for the block and its history is
// staged in LDS, and so are the LPF taps of the (at most 4) frames the block touches.
constexpr int kExcBlock = 256;
constexpr int kExcHalo = 64; // >= nlpf-1
constexpr int kExcMaxFrames = (kExcBlock + kExcHalo) / 30 + 3;

__global__ __launch_bounds__(kExcBlock) void k_excite(BatchDev bd, VocDev vd)
{
    const int b = blockIdx.y;
    const UttDev *u = bd.utt + b;
    const int fp = vd.fperiod, bs = vd.bs, nblk = vd.nblk, nlpf = vd.nlpf;
    const uint64_t N = (uint64_t)u->T * (uint64_t)fp;
    const uint64_t n0 = (uint64_t)blockIdx.x * kExcBlock;
    if (n0 >= N)
        return;
    const uint64_t base = u->frame_off;
    const int tid = threadIdx.x;
    const int anti = (nlpf - 1) / 2;
    __shared__ double e[kExcBlock + kExcHalo];
    // taps of every frame the block and its history touch: (256 + 64) / fperiod + 2 frames; the
    // launch_excite sends frame periods below 30 samples to k_excite_any, hence 13 rows
    __shared__ double taps[kExcMaxFrames][64];
    __shared__ int anyv;
    // frames touched by samples [n0 - halo, n0 + 255]
    const long mfirst = (long)n0 - (long)kExcHalo;
    const uint32_t f_lo = mfirst > 0 ? (uint32_t)((uint64_t)mfirst / (uint64_t)fp) : 0u;
    if (tid == 0)
        anyv = 0;
    {
        const uint32_t f_hi = (uint32_t)((n0 + kExcBlock - 1) / (uint64_t)fp);
        const int nfr = (int)(f_hi - f_lo) + 1; // <= kExcMaxFrames for fperiod >= 30
        const int k = tid & 63;
        for (int fi = tid >> 6; fi < nfr && fi < kExcMaxFrames; fi += kExcBlock / 64) {
            const uint32_t fr = f_lo + (uint32_t)fi;
            taps[fi][k] = (fr < u->T && k < nlpf) ? vd.lpf[(base + fr) * (uint64_t)nlpf + k] : 0.0;
        }
    }
    __syncthreads();
    for (int j = tid; j < kExcBlock + kExcHalo; j += kExcBlock) {
        const long m = (long)n0 - kExcHalo + j;
        double ev = 0.0;
        if (m >= 0 && (uint64_t)m < N) {
            const uint32_t fr = (uint32_t)((uint64_t)m / (uint64_t)fp);
            const int i = (int)((uint64_t)m - (uint64_t)fr * (uint64_t)fp);
            const uint64_t f = base + fr;
            if (vd.pitch[f] != 0.0) {
                const unsigned long long pm = vd.pmask[f * (uint64_t)nblk + (uint64_t)(i / bs)];
                double pulse = 0.0;
                if ((pm >> (i % bs)) & 1ull)
                    pulse = sqrt(fma((double)i, vd.pinc[f], vd.cur_start[f]));
                ev = pulse - vd.noise[m];
                anyv = 1;
            }
        }
        e[j] = ev;
    }
    __syncthreads();
    const uint64_t n = n0 + (uint64_t)tid;
    if (n >= N)
        return;
    const uint32_t fr = (uint32_t)(n / (uint64_t)fp);
    const int i = (int)(n - (uint64_t)fr * (uint64_t)fp);
    double x = n >= (uint64_t)anti ? vd.noise[n - (uint64_t)anti] : 0.0;
    if (anyv) {
        // source sample n-k lies in this frame while k <= i, in the previous one after that
        // (nlpf-1 < fperiod); taps of the SOURCE sample's frame (excitation.rs:48-64)
        const double *tc = taps[fr - f_lo];
        const double *tp = taps[fr > f_lo ? fr - f_lo - 1 : 0];
        const int kmax = n + 1 < (uint64_t)nlpf ? (int)n + 1 : nlpf;
        const double *ep = e + kExcHalo + tid;
#pragma unroll 4
        for (int k = 0; k < kmax; k++)
            x = fma(ep[-k], k <= i ? tc[k] : tp[k], x);
    }
    if (vd.exc)
        vd.exc[base * (uint64_t)fp + n] = x;
    vd.xin[base * (uint64_t)fp + n] = x;
}

// The same sum for every shape k_excite's staging is not laid out for: more than 64 taps (its tap rows and its
// 64-sample history), a frame period below 30 samples (more than 13 frames under a block and its history) or below
// nlpf - 1 (a sample's window then reaches back over several frames).  The reference is generic in all three
// (Vocoder::new, vocoder/mod.rs:45-70; RingBuffer::new(nlpf), excitation.rs:113-123).  e[] of the block and its
// nlpf - 1 samples of history in dynamic LDS; the tap of source sample n - k is read where the LPF track has it
// (row of that sample's frame, L1 / L2 hits: a wave's lanes are consecutive samples), the frame of the source
// sample stepped back beside k.  Same order of additions as k_excite.
constexpr int kExcAnyMaxNlpf = 2047; // (256 + 2046) doubles of LDS per workgroup
__global__ __launch_bounds__(kExcBlock) void k_excite_any(BatchDev bd, VocDev vd)
{
    extern __shared__ double e_any[];
    const int b = blockIdx.y;
    const UttDev *u = bd.utt + b;
    const int fp = vd.fperiod, bs = vd.bs, nblk = vd.nblk, nlpf = vd.nlpf;
    const uint64_t N = (uint64_t)u->T * (uint64_t)fp;
    const uint64_t n0 = (uint64_t)blockIdx.x * kExcBlock;
    if (n0 >= N)
        return;
    const uint64_t base = u->frame_off;
    const int tid = threadIdx.x;
    const int anti = (nlpf - 1) / 2, halo = nlpf - 1;
    __shared__ int anyv;
    if (tid == 0)
        anyv = 0;
    __syncthreads();
    for (int j = tid; j < kExcBlock + halo; j += kExcBlock) {
        const long m = (long)n0 - halo + j;
        double ev = 0.0;
        if (m >= 0 && (uint64_t)m < N) {
            const uint32_t fr = (uint32_t)((uint64_t)m / (uint64_t)fp);
            const int i = (int)((uint64_t)m - (uint64_t)fr * (uint64_t)fp);
            const uint64_t f = base + fr;
            if (vd.pitch[f] != 0.0) {
                const unsigned long long pm = vd.pmask[f * (uint64_t)nblk + (uint64_t)(i / bs)];
                double pulse = 0.0;
                if ((pm >> (i % bs)) & 1ull)
                    pulse = sqrt(fma((double)i, vd.pinc[f], vd.cur_start[f]));
                ev = pulse - vd.noise[m];
                anyv = 1;
            }
        }
        e_any[j] = ev;
    }
    __syncthreads();
    const uint64_t n = n0 + (uint64_t)tid;
    if (n >= N)
        return;
    double x = n >= (uint64_t)anti ? vd.noise[n - (uint64_t)anti] : 0.0;
    if (anyv) {
        uint32_t fr = (uint32_t)(n / (uint64_t)fp);
        int i = (int)(n - (uint64_t)fr * (uint64_t)fp);
        const int kmax = n + 1 < (uint64_t)nlpf ? (int)n + 1 : nlpf;
        const double *ep = e_any + halo + tid;
        const double *tap = vd.lpf + (base + fr) * (uint64_t)nlpf; // taps of the SOURCE sample's frame (excitation.rs:48-64)
        for (int k = 0; k < kmax; k++) {
            x = fma(ep[-k], tap[k], x);
            if (i == 0) {
                i = fp - 1;
                tap -= nlpf;
            } else {
                i--;
            }
        }
    }
    if (vd.exc)
        vd.exc[base * (uint64_t)fp + n] = x;
    vd.xin[base * (uint64_t)fp + n] = x;
}
int excite_max_nlpf() { return kExcAnyMaxNlpf; }
// the shapes k_excite's LDS staging is laid out for (everything else: k_excite_any)
static bool excite_fits_staged(const VocDev &vd)
{
    return vd.fperiod >= 30 && vd.nlpf <= 64 && vd.nlpf - 1 <= vd.fperiod;
}

// nlpf == 0: the ring-buffer-less branch of Excitation::get (excitation.rs:87-100), reachable through
// Vocoder::synthesize alone (SpeechGenerator::new refuses an even LPF length).  Unvoiced samples take the next
// value of the noise stream -- which is drawn on unvoiced samples only, so the stream position of a frame is
// fperiod times the number of unvoiced frames before it --, voiced samples are the bare pulse, no delay.
__global__ __launch_bounds__(64) void k_uv_scan(BatchDev bd, VocDev vd)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const UttDev *u = bd.utt + b;
    const uint64_t base = u->frame_off;
    uint32_t acc = 0;
    for (uint32_t t0 = 0; t0 < u->T; t0 += 64) {
        const uint32_t t = t0 + (uint32_t)lane;
        const bool uv = t < u->T && vd.pitch[base + t] == 0.0;
        const unsigned long long m = __ballot(uv);
        if (t < u->T)
            vd.uv_before[base + t] = acc + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        acc += (uint32_t)__popcll(m);
    }
}
__global__ __launch_bounds__(256) void k_excite_nolpf(BatchDev bd, VocDev vd)
{
    const int b = blockIdx.y;
    const UttDev *u = bd.utt + b;
    const int fp = vd.fperiod, bs = vd.bs, nblk = vd.nblk;
    const uint64_t N = (uint64_t)u->T * (uint64_t)fp;
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N)
        return;
    const uint64_t base = u->frame_off;
    const uint32_t fr = (uint32_t)(n / (uint64_t)fp);
    const int i = (int)(n - (uint64_t)fr * (uint64_t)fp);
    const uint64_t f = base + fr;
    double x;
    if (vd.pitch[f] != 0.0) {
        const unsigned long long pm = vd.pmask[f * (uint64_t)nblk + (uint64_t)(i / bs)];
        x = ((pm >> (i % bs)) & 1ull) ? sqrt(fma((double)i, vd.pinc[f], vd.cur_start[f])) : 0.0;
    } else {
        x = vd.noise[(uint64_t)vd.uv_before[f] * (uint64_t)fp + (uint64_t)i];
    }
    if (vd.exc)
        vd.exc[base * (uint64_t)fp + n] = x;
    vd.xin[base * (uint64_t)fp + n] = x;
}

// Split excitation, ONE WAVE PER FRAME with FOUR CONSECUTIVE SAMPLES PER LANE (fperiod % 4 == 0,
// fperiod <= 256, nlpf <= 33).  k_excite above reads e[] and two tap sets from LDS for every
// (sample, tap): ~93 LDS reads per sample, which made it LDS-bound (21 ms at config 2).  Here a
// lane keeps the window e[4j-30 .. 4j+3] of its four outputs in registers and the taps of the
// frame are wave-uniform.
// The reference adds the terms of a sample in tap order k = 0..nlpf-1, and the source sample
// n-k lies in this frame for k <= i and in the previous one (previous frame's taps) after
// that.  Two passes in tap order reproduce that order bit for bit: the first over the frame's
// own e (previous-frame entries read as zero: x + 0*c == x), the second over the previous
// frame's e with its taps, executed only by the lanes that own the first 32 samples and only
// when that frame was voiced.
// A sample whose window holds no pulse does not depend on the LF0 track: the pulse-free pass computes
// EVERY sample that way from the MSD voiced flags and the LPF taps alone (it can start with the
// step); after the pulse walk, k_excite_fix adds the pulse terms.  Together they equal k_excite up to the
// order of the additions (~1e-16 relative).
constexpr int kExw = 4;           // samples per lane
constexpr int kExwHalo = 32;      // staged history, multiple of kExw, >= nlpf-1
constexpr int kExwQ = (256 + kExwHalo) / kExw; // LDS row pitch (q = (m + halo) / 4)

// The pulse-free pass of ONE frame by one wave.  NLPF is a template parameter (with a run-time tap count every
// tap is its own scalar load + branch: 14 ms).  A lane loads the four noise values of its own samples as one
// 32-byte piece and keeps them (the only LDS image is this frame's e, for the 30-sample history of a lane's
// window), the start values noise[n - 15] come straight from memory (the L1 has the lines), the previous
// frame's tail is loaded by the eight lanes that need it, and the taps of the frame are scalar loads issued
// WITH everything else at the top: a wave is a chain of memory round trips, so everything it will read is
// requested in one go.  (An earlier form staged 272 samples through three LDS images and spent two thirds of
// its VALU instructions outside its FMAs: tools/experiments/README.md.)
//   n0 = first sample of the frame in its utterance; tc / tpp = taps of the frame and of its predecessor
//   (wave-uniform pointers); vcur / vprev = voiced flags of the two (vprev false for an utterance's first frame).
// Returns the lane's four samples in x (own = the lane holds samples of the frame).
constexpr int kN4Group = 4; // taps per window group of pass 1
template <int NLPF>
__device__ __forceinline__ bool exc_pulse_free(const VocDev &vd, const int fp, const long n0,
                                               const double *__restrict__ tc, const double *__restrict__ tpp,
                                               const bool vcur, const bool vprev, const int lane,
                                               double (*ec)[kExwQ], double *ep, double *xs, double (&x)[kExw])
{
    static_assert(NLPF - 1 <= kExwHalo - 2, "history window too short");
    constexpr int kExwWin = NLPF - 1 + kExw;
    constexpr int anti = (NLPF - 1) / 2;
    constexpr int HQ = kExwHalo / kExw; // history columns of the image
    const int i0 = lane * kExw;
    const bool own = i0 < fp;
    // ---- every request of the wave ----
    double ck[NLPF]; // wave-uniform: scalar loads
#pragma unroll
    for (int k = 0; k < NLPF; k++)
        ck[k] = tc[k];
    const double tpv = lane < NLPF ? tpp[lane] : 0.0;
    const long nl = own ? n0 + i0 : n0; // (a lane past the frame reads the frame's first samples and keeps nothing)
    const double2 nva = *reinterpret_cast<const double2 *>(vd.noise + nl);
    const double2 nvb = *reinterpret_cast<const double2 *>(vd.noise + nl + 2);
    double xi[kExw];
#pragma unroll
    for (int r = 0; r < kExw; r++) {
        const long q = nl + r - anti;
        const double v = vd.noise[q < 0 ? 0 : q];
        xi[r] = q < 0 ? 0.0 : v;
    }
    // the previous frame's last 32 samples, four per lane on lanes 0..7 (zero before the utterance starts)
    const long nt = n0 - kExwHalo + i0;
    const bool has_tail = lane < HQ && nt >= 0;
    const long ntc = has_tail ? nt : n0;
    const double2 tla = *reinterpret_cast<const double2 *>(vd.noise + ntc);
    const double2 tlb = *reinterpret_cast<const double2 *>(vd.noise + ntc + 2);
    const double nv[kExw] = {nva.x, nva.y, nvb.x, nvb.y};
#pragma unroll
    for (int r = 0; r < kExw; r++)
        x[r] = xi[r];
    if (vcur) {
        // image of this frame's e = -noise: own samples at column HQ + lane, history columns zero
        if (lane < HQ) {
#pragma unroll
            for (int r = 0; r < kExw; r++)
                ec[r][lane] = 0.0;
        }
        if (own) {
#pragma unroll
            for (int r = 0; r < kExw; r++)
                ec[r][HQ + lane] = 0.0 - nv[r];
        }
    }
    if (vprev) {
        // previous frame's e tail, then zeros where this frame starts
        if (lane < HQ) {
            const double tl[kExw] = {tla.x, tla.y, tlb.x, tlb.y};
#pragma unroll
            for (int r = 0; r < kExw; r++)
                ep[i0 + r] = has_tail ? 0.0 - tl[r] : 0.0;
        } else if (lane < 2 * HQ) {
#pragma unroll
            for (int r = 0; r < kExw; r++)
                ep[i0 + r] = 0.0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (vcur && own) {
        // the window e[i0 - (NLPF-1) .. i0 + 3] a GROUP of taps at a time (taps k0..k1 touch eleven of its 34
        // values): with the whole window in registers the kernel needs 62 VGPRs and only one of its waves fits
        // a SIMD beside the resident GV kernel's two of 208; same order of the additions, so the same bits
        constexpr int G = kN4Group;
#pragma unroll
        for (int k0 = 0; k0 < NLPF; k0 += G) {
            const int k1 = k0 + G - 1 < NLPF - 1 ? k0 + G - 1 : NLPF - 1;
            const int lo = NLPF - 1 - k1; // window index of the oldest value the group touches
            double wg[G + kExw - 1];
#pragma unroll
            for (int c = 0; c < G + kExw - 1; c++) {
                const int idx = lo + c; // w[idx] = e[i0 - (NLPF-1) + idx]
                if (idx < kExwWin) {
                    if (idx >= NLPF - 1) {
                        wg[c] = 0.0 - nv[idx - (NLPF - 1)];
                    } else {
                        const int o = idx + kExwHalo - (NLPF - 1);
                        wg[c] = ec[o & (kExw - 1)][lane + (o >> 2)];
                    }
                } else {
                    wg[c] = 0.0;
                }
            }
#pragma unroll
            for (int k = k0; k <= k1; k++) {
#pragma unroll
                for (int r = 0; r < kExw; r++)
                    x[r] = fma(wg[NLPF - 1 + r - k - lo], ck[k], x[r]);
            }
        }
    }
    if (vprev) {
        // pass 2 (previous frame's sources and taps): only the first NLPF-1 samples have any; they are turned
        // to one sample per lane through LDS so that the pass costs NLPF-1 FMAs per wave.  ep[] continues
        // with zeros where this frame starts: x + 0*c == x, the order of the remaining terms is the tap order
        if (own && i0 < kExwHalo) {
#pragma unroll
            for (int r = 0; r < kExw; r++)
                xs[i0 + r] = x[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            double tk[NLPF];
#pragma unroll
            for (int k = 1; k < NLPF; k++) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(tpv), k);
                const int hi = __builtin_amdgcn_readlane(__double2hiint(tpv), k);
                tk[k] = __hiloint2double(hi, lo);
            }
            if (lane < NLPF - 1 && lane < fp) {
                double xv = xs[lane];
#pragma unroll
                for (int k = 1; k < NLPF; k++)
                    xv = fma(ep[lane - k + kExwHalo], tk[k], xv);
                xs[lane] = xv;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (own && i0 < kExwHalo) {
#pragma unroll
            for (int r = 0; r < kExw; r++)
                x[r] = xs[i0 + r];
        }
    }
    return own;
}

// Where does each frame's excitation come from (VocDev::exc_src)?  One wave per 64 consecutive frames of an
// utterance: lanes as (frame pair, tap) compare the LPF rows of frames fr0-1 .. fr0+63 with the canonical taps
// (row 0 of the batch) bit by bit, two frames per coalesced request, all requests in flight together; then
// lane = frame classifies: unvoiced behind unvoiced -> the noise stream; voiced behind voiced with both rows
// canonical -> the shared table (until the pulse pass finds a pulse reaching the frame); everything else -> the
// per-frame pass, whose work list (exc_gen) the wave appends its frames to (one atomic per wave).
template <int NLPF>
__global__ __launch_bounds__(256) void k_exc_classify(BatchDev bd, VocDev vd)
{
    const int b = blockIdx.y;
    const UttDev *u = bd.utt + b;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t T = u->T;
    const uint32_t fr0 = (blockIdx.x * 4u + (uint32_t)wv) * 64u;
    if (blockIdx.x * 256u >= T)
        return; // (the whole workgroup: a wave past the end stays for the workgroup's one list reservation)
    __shared__ uint32_t wcount[4], wbase;
    const uint64_t base = u->frame_off;
    unsigned long long canon = 0ull; // bit j: row of frame fr0 - 1 + j is canonical (j = 0..63), bit 64 below
    bool canon64 = false;
    if (!vd.exc_no_table && vd.lpf_canon) {
        // the LPF MLPG kept track (k_mlpg_static: same inputs as the first frame's, bit for bit): one byte per frame
        const long fa = (long)fr0 - 1 + lane;
        const bool ina = fa >= 0 && fa < (long)T;
        const uint8_t ca = vd.lpf_canon[base + (uint64_t)(ina ? fa : 0)];
        const long fb = (long)fr0 + 63;
        const uint8_t cb = vd.lpf_canon[base + (uint64_t)(fb < (long)T ? fb : 0)];
        canon = __ballot(ina && ca != 0);
        canon64 = fb < (long)T && cb != 0;
    } else if (!vd.exc_no_table) {
        const int half = lane >> 5, k = lane & 31;
        const bool tap = k < NLPF;
        const double cref = vd.lpf[tap ? k : 0];
        double v[33];
#pragma unroll
        for (int s = 0; s < 33; s++) {
            const long fr = (long)fr0 - 1 + 2 * s + half;
            const bool in = tap && fr >= 0 && fr < (long)T;
            v[s] = vd.lpf[in ? (base + (uint64_t)fr) * (uint64_t)NLPF + (uint64_t)k : (uint64_t)(tap ? k : 0)];
            if (!(fr >= 0 && fr < (long)T))
                v[s] = __longlong_as_double(~__double_as_longlong(cref)); // outside the utterance: not canonical
        }
        // (all 33 requests are out before the first comparison waits for one: the compiler otherwise sinks every
        // load to its use and the wave lives through 33 round trips)
#pragma unroll
        for (int s = 0; s < 33; s++)
            asm volatile("" : "+v"(v[s]));
#pragma unroll
        for (int s = 0; s < 33; s++) {
            const bool eq = !tap || __double_as_longlong(v[s]) == __double_as_longlong(cref);
            const unsigned long long m = __ballot(eq);
            const bool ea = (uint32_t)m == 0xFFFFFFFFu, eb = (uint32_t)(m >> 32) == 0xFFFFFFFFu;
            if (s < 32)
                canon |= ((unsigned long long)ea << (2 * s)) | ((unsigned long long)eb << (2 * s + 1));
            else
                canon64 = ea;
        }
    }
    const uint32_t fr = fr0 + (uint32_t)lane;
    const bool in = fr < T;
    const uint64_t f = base + (in ? fr : 0u); // (a lane past the utterance reads its first frame and keeps nothing)
    const bool vcur = in && vd.voiced[f] != 0;
    const bool vprev = in && fr > 0 && vd.voiced[f - 1] != 0;
    const bool ccur = lane == 63 ? canon64 : ((canon >> (lane + 1)) & 1ull) != 0;
    const bool cprev = ((canon >> lane) & 1ull) != 0;
    uint32_t code = 0;
    if (!vd.exc_no_table && !vcur && !vprev && fr >= 1)
        code = 1;
    else if (!vd.exc_no_table && vcur && vprev && ccur && cprev)
        code = 2;
    if (in)
        vd.exc_src[f] = (uint8_t)code;
    const bool gen = in && code == 0;
    const unsigned long long gm = __ballot(gen);
    // one reservation in the work list per WORKGROUP (with one per wave, 100 k atomics on one address were what
    // the kernel's 1.0 ms consisted of)
    if (lane == 0)
        wcount[wv] = (uint32_t)__popcll(gm);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        wbase = tot ? atomicAdd(vd.exc_gen_count, tot) : 0u;
    }
    __syncthreads();
    uint32_t slot0 = wbase;
    for (int w = 0; w < wv; w++)
        slot0 += wcount[w];
    if (gen) {
        const uint32_t rank = (uint32_t)__popcll(gm & ((1ull << lane) - 1ull));
        uint32_t *e = vd.exc_gen + 2 * (uint64_t)(slot0 + rank);
        e[0] = (uint32_t)b;
        e[1] = fr | (vcur ? 1u << 30 : 0u) | (vprev ? 1u << 31 : 0u);
    }
}

// The shared table: the pulse-free excitation of a voiced frame behind a voiced frame, both with the canonical
// taps, for every frame position 1 .. maxT-1 (a first frame has no predecessor: never read from the table).
template <int NLPF>
__global__ __launch_bounds__(256) void k_exc_table(BatchDev bd, VocDev vd)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t fr = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (uint32_t)wv));
    if (fr == 0 || fr >= bd.maxT)
        return;
    __shared__ double ec_s[4][kExw][kExwQ];
    __shared__ double ep_s[4][2 * kExwHalo];
    __shared__ double xs_s[4][kExwHalo];
    const int fp = vd.fperiod;
    const long n0 = (long)fr * (long)fp;
    double x[kExw];
    if (!exc_pulse_free<NLPF>(vd, fp, n0, vd.lpf, vd.lpf, true, true, lane, ec_s[wv], ep_s[wv], xs_s[wv], x))
        return;
    double *o = vd.exc_tab + (uint64_t)n0 + (uint64_t)(lane * kExw);
    *reinterpret_cast<double2 *>(o) = make_double2(x[0], x[1]);
    *reinterpret_cast<double2 *>(o + 2) = make_double2(x[2], x[3]);
}

// The per-frame pass over the work list of k_exc_classify: every wave of a fixed grid takes entries in
// turn (the next entry is requested while the current frame is worked on).
template <int NLPF>
__global__ __launch_bounds__(256) void k_exc_general(BatchDev bd, VocDev vd)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ double ec_s[4][kExw][kExwQ];
    __shared__ double ep_s[4][2 * kExwHalo];
    __shared__ double xs_s[4][kExwHalo];
    const uint32_t count = *vd.exc_gen_count;
    const uint32_t stride = gridDim.x * 4u;
    uint32_t i = blockIdx.x * 4u + (uint32_t)wv;
    if (i >= count)
        return;
    const int fp = vd.fperiod;
    uint32_t eb = vd.exc_gen[2 * (uint64_t)i], ef = vd.exc_gen[2 * (uint64_t)i + 1];
    for (; i < count; i += stride) {
        const int b = __builtin_amdgcn_readfirstlane((int)eb);
        const uint32_t efu = (uint32_t)__builtin_amdgcn_readfirstlane((int)ef);
        const uint32_t inext = i + stride < count ? i + stride : i;
        eb = vd.exc_gen[2 * (uint64_t)inext];
        ef = vd.exc_gen[2 * (uint64_t)inext + 1];
        const uint32_t fr = efu & 0x3FFFFFFFu;
        const bool vcur = (efu >> 30 & 1u) != 0, vprev = (efu >> 31) != 0;
        const uint64_t base = bd.utt[b].frame_off;
        const uint64_t f = base + fr;
        const uint64_t fprev = fr > 0 ? f - 1 : f;
        const long n0 = (long)fr * (long)fp;
        double x[kExw];
        const bool own = exc_pulse_free<NLPF>(vd, fp, n0, lpf_row(vd, f, NLPF), lpf_row(vd, fprev, NLPF),
                                              vcur, vprev, lane, ec_s[wv], ep_s[wv], xs_s[wv], x);
        if (own) {
            const uint64_t o = base * (uint64_t)fp + (uint64_t)n0 + (uint64_t)(lane * kExw);
            *reinterpret_cast<double2 *>(vd.xin + o) = make_double2(x[0], x[1]);
            *reinterpret_cast<double2 *>(vd.xin + o + 2) = make_double2(x[2], x[3]);
            if (vd.exc) {
                *reinterpret_cast<double2 *>(vd.exc + o) = make_double2(x[0], x[1]);
                *reinterpret_cast<double2 *>(vd.exc + o + 2) = make_double2(x[2], x[3]);
            }
        }
        // the LDS images are reused by the next frame
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Second half of the split excitation.  The pulse-free pass computed every sample with e = -noise in voiced
// frames; what is missing is the pulse term of the ring buffer's FIR (excitation.rs:48-64):
//     x[n] += a(g, p) * lpf_g[n - n_p]     for every pulse p of frame g with 0 <= n - n_p <= NLPF-1,
// a = sqrt(pitch_of_curr_point at the pulse) -- ADDED to what the first pass left (one rounding later than in
// tap order: ~1e-16 relative; the pulse positions and amplitudes are the exact ones).
// One wave per frame, the frame's samples four per lane as in the first pass, so every sample has exactly one
// owner: the wave of ITS frame applies the pulses of that frame and the tail of the previous frame's pulses
// that reaches into it, in ascending time.  Overlapping pulses (periods below NLPF samples: the lf0 clamp
// allows them) accumulate in registers, in order; no wave writes another frame's samples, so there is no
// ordering between waves to keep.  A frame costs ONE memory round trip whatever its pulse count (round 2's
// form recomputed the NLPF samples behind every pulse from a rebuilt window of e: NLPF^2 multiply-adds and a
// round trip per pulse, 3.0 ms alone on config 2 and the kernel the vocoder waited for).  Frames that no
// pulse reaches leave without touching memory; of the others only the lanes a pulse reaches load and store.
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

constexpr int kFixFrames = 8;

template <int NLPF>
__global__ __launch_bounds__(256) void k_excite_fix(BatchDev bd, VocDev vd)
{
    const int b = blockIdx.y;
    const UttDev *u = bd.utt + b;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t T = u->T;
    const uint64_t base = u->frame_off;
    const int fp = vd.fperiod, bs = vd.bs, nblk = vd.nblk;
    constexpr int H = NLPF - 1;
    __shared__ double tap_s[4][2][NLPF]; // taps of the previous frame [0] and of this one [1]
    double(*tap2)[NLPF] = tap_s[wv];
    // A wave walks kFixFrames frames, frame index = (block * kFixFrames + jf) * 4 + wave.  What decides whether a
    // frame has work -- the voiced flags and pulse mask words of the frame and of its predecessor -- is fetched
    // for all of them in ONE request: lanes 0..31 = (frame jf, word q) of the frame itself, lanes 32..63 the same
    // of the frame before it.  (The masks of unvoiced frames are never written: gated by the flags.)
    // The kernel's scalar unit is what it is bound by (one per CU: 182 scalar instructions per frame took 2.2 of
    // its 2.5 ms when every frame's eight words were fetched with readlane pairs and scanned, twice): so the words
    // are gated and trimmed in the lanes that loaded them -- a word counts if its frame is voiced, the previous
    // frame's only from the first sample whose pulse reaches into this frame -- and ONE ballot says which of the
    // 64 are non-zero; the per-frame loops then visit exactly the words that hold a pulse.
    static_assert(kFixFrames * 4 <= 32, "one lane per (frame, mask word), twice");
    const int s0 = lane * kExw; // this lane's samples s0 .. s0+3 of the frame
    const bool own = s0 < fp;
    // A frame that stood on the shared table gets pieces of a row of its own only for the BLOCKS (bs samples) a pulse
    // reaches; exc_src tells the vocoder which (a pulse's 31 samples lie in one or two of a frame's four blocks: the
    // rows were 17 GB written and read back per step of config 2).  Needs a lane's samples in one block and, for the
    // throughput kernel's pointer hand-over, blocks of an even number of samples; otherwise whole rows as before.
    const bool blk_rows = bs % kExw == 0 && bs % 2 == 0 && nblk <= 4;
    int myblk = 0;
    for (int q = 1; q < nblk; q++)
        myblk += s0 >= q * bs ? 1 : 0;
    const int tail0 = fp - H; // first sample of the previous frame whose pulse reaches into this one
    unsigned long long mword = 0ull;
    uint32_t vfl = 0, codev = 0; // codev: exc_src of frame jf on lane 4 * jf + 1
    uint32_t cnv = 0;            // lane 4 * jf + 2 (+ 32): the frame's (the frame before's) LPF row is the canonical one
    {
        const int half = lane >> 5, jf = (lane & 31) / 4, q = lane & 3;
        const long frl = (long)((blockIdx.x * (uint32_t)kFixFrames + (uint32_t)jf) * 4u + (uint32_t)wv) - half;
        if (frl >= 0 && frl < (long)T) {
            if (q == 0)
                vfl = vd.voiced[base + (uint64_t)frl];
            if (q == 1 && half == 0 && vd.exc_src)
                codev = vd.exc_src[base + (uint64_t)frl];
            if (q == 2 && vd.lpf_sparse)
                cnv = vd.lpf_canon[base + (uint64_t)frl];
            if (q < nblk)
                mword = vd.pmask[(base + (uint64_t)frl) * (uint64_t)nblk + (uint64_t)q];
        }
        // the flag of the quad's first lane for all four words of the frame (quad_perm [0,0,0,0])
        const uint32_t vq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vfl, 0x00, 0xf, 0xf, true);
        if (!vq)
            mword = 0ull;
        if (half) {
            const int qt = tail0 / bs, lo = tail0 - qt * bs;
            if (q < qt)
                mword = 0ull;
            else if (q == qt && lo > 0)
                mword &= ~0ull << lo;
        }
    }
    const unsigned long long nz = __ballot(mword != 0ull);
    // words of frame jf that hold a pulse reaching it: bits 0..3 = words of the previous frame, 4..7 = its own
    auto words_of = [&](int jf) -> uint32_t {
        return (uint32_t)((nz >> (32 + 4 * jf)) & 0xFull) | ((uint32_t)((nz >> (4 * jf)) & 0xFull) << 4);
    };
    // pulses that reach frame jf, in ascending time: fn(position relative to the frame's first sample, 0 = of the
    // previous frame / 1 = its own, sample index in the pulse's own frame); wave-uniform
    auto each_pulse = [&](int jf, uint32_t m8, auto fn) {
        while (m8) {
            const int idx = __builtin_ctz(m8);
            m8 &= m8 - 1u;
            const int ownf = idx >> 2, q = idx & 3;
            const int src = (ownf ? 0 : 32) + jf * 4 + q;
            const uint32_t wl = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mword, src);
            const uint32_t wh = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mword >> 32), src);
            unsigned long long w = ((unsigned long long)wh << 32) | wl;
            while (w) {
                const int j = __builtin_ctzll(w);
                w &= w - 1ull;
                fn(ownf ? q * bs + j : q * bs + j - fp, ownf, q * bs + j);
            }
        }
    };
    // Frames in groups of kFixGroup: FIRST everything the group reads from memory is requested (taps, amplitude
    // parameters, the samples of the lanes a pulse reaches), THEN the frames are worked off.  Frame after frame
    // with a load -> add -> store chain each, a wave lived through nine dependent round trips (26 us; the kernel
    // runs at full occupancy and was bound by exactly that).
#ifndef JB_FIX_GROUP
#define JB_FIX_GROUP 4
#endif
    constexpr int kFixGroup = JB_FIX_GROUP;
    static_assert(kFixFrames % kFixGroup == 0, "whole groups");
    for (int j0 = 0; j0 < kFixFrames; j0 += kFixGroup) {
        bool work[kFixGroup], touched[kFixGroup], from_tab[kFixGroup];
        uint32_t bmask[kFixGroup];
        double tv[kFixGroup], pv[kFixGroup], x[kFixGroup][kExw];
#pragma unroll
        for (int g = 0; g < kFixGroup; g++) {
            const int jf = j0 + g;
            const uint32_t fr = (uint32_t)__builtin_amdgcn_readfirstlane(
                (int)((blockIdx.x * (uint32_t)kFixFrames + (uint32_t)jf) * 4u + (uint32_t)wv));
            work[g] = touched[g] = from_tab[g] = false;
            bmask[g] = 0;
            tv[g] = pv[g] = 0.0;
#pragma unroll
            for (int r = 0; r < kExw; r++)
                x[g][r] = 0.0;
            if (fr >= T)
                continue;
            const uint64_t f = base + fr;
            const uint32_t m8 = words_of(jf);
            if (!m8)
                continue;
            const bool any_p = (m8 & 0xFu) != 0, any_c = (m8 >> 4) != 0;
            bool tch = false;
            each_pulse(jf, m8, [&](int p, int, int) { tch = tch || (s0 + kExw - 1 >= p && s0 <= p + H); });
            // a frame that stood on the shared table until now (exc_src 2) gets (pieces of) a row of its own: the lanes
            // of the blocks a pulse reaches take their samples from the table and write them with the pulse terms added
            from_tab[g] = (__builtin_amdgcn_readlane((int)codev, 4 * jf + 1) & 3) == 2;
            work[g] = true;
            bool in_blk = true;
            if (from_tab[g] && blk_rows) {
                uint32_t bm = 0;
                for (int q = 0; q < nblk; q++)
                    if (__ballot(tch && own && myblk == q))
                        bm |= 1u << q;
                bmask[g] = bm;
                in_blk = ((bm >> myblk) & 1u) != 0;
            }
            touched[g] = (tch || (from_tab[g] && in_blk)) && own;
            {
                // taps of the previous frame (lanes 0..31; only if one of its last H samples holds a pulse: one
                // frame in eight) and of this one (lanes 32..63), lane = tap
                const int hw = lane >> 5, k = lane & 31;
                // (a canonical row is not stored: row 0 of the batch holds the same bits)
                const uint32_t cn_c = (uint32_t)__builtin_amdgcn_readlane((int)cnv, 4 * jf + 2);
                const uint32_t cn_p = (uint32_t)__builtin_amdgcn_readlane((int)cnv, 32 + 4 * jf + 2);
                if (k < NLPF && (hw ? any_c : any_p))
                    tv[g] = vd.lpf[((hw ? cn_c : cn_p) ? 0ull : f - 1 + (uint64_t)hw) * (uint64_t)NLPF + (uint64_t)k];
                // amplitude parameters: lanes 0,1 = cur_start, pinc of the previous frame; 2,3 = of this one
                if (lane < 4 && ((lane >> 1) ? any_c : any_p)) {
                    const uint64_t ff = f - 1 + (uint64_t)(lane >> 1);
                    pv[g] = (lane & 1) ? vd.pinc[ff] : vd.cur_start[ff];
                }
            }
            if (touched[g]) {
                const double *src = from_tab[g] ? vd.exc_tab + ((uint64_t)fr * (uint64_t)fp + (uint64_t)s0)
                                                : vd.xin + (f * (uint64_t)fp + (uint64_t)s0);
                const double2 xa = *reinterpret_cast<const double2 *>(src);
                const double2 xb = *reinterpret_cast<const double2 *>(src + 2);
                x[g][0] = xa.x;
                x[g][1] = xa.y;
                x[g][2] = xb.x;
                x[g][3] = xb.y;
            }
        }
#pragma unroll
        for (int g = 0; g < kFixGroup; g++) {
            if (!work[g])
                continue;
            const int jf = j0 + g;
            const uint32_t fr = (uint32_t)__builtin_amdgcn_readfirstlane(
                (int)((blockIdx.x * (uint32_t)kFixFrames + (uint32_t)jf) * 4u + (uint32_t)wv));
            const uint64_t f = base + fr;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier(); // the previous frame's reads of tap2 are done
            if ((lane & 31) < NLPF)
                tap2[lane >> 5][lane & 31] = tv[g];
            const double cs_p = readlane_f64(pv[g], 0), pi_p = readlane_f64(pv[g], 1);
            const double cs_c = readlane_f64(pv[g], 2), pi_c = readlane_f64(pv[g], 3);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // the pulse terms, in ascending time
            each_pulse(jf, words_of(jf), [&](int p, int gsel, int i) {
                // voiced branch of Excitation::get (excitation.rs:73-81): pulse = sqrt(pitch_of_curr_point), which
                // has advanced by i increments when sample i of its frame is drawn
                const double amp = sqrt(gsel ? fma((double)i, pi_c, cs_c) : fma((double)i, pi_p, cs_p));
#pragma unroll
                for (int r = 0; r < kExw; r++) {
                    const int k = s0 + r - p;
                    const bool in = k >= 0 && k <= H;
                    const double tk = tap2[gsel][in ? k : 0];
                    if (in)
                        x[g][r] = fma(amp, tk, x[g][r]);
                }
            });
            if (touched[g]) {
                const uint64_t o = f * (uint64_t)fp + (uint64_t)s0;
                *reinterpret_cast<double2 *>(vd.xin + o) = make_double2(x[g][0], x[g][1]);
                *reinterpret_cast<double2 *>(vd.xin + o + 2) = make_double2(x[g][2], x[g][3]);
                if (vd.exc) {
                    *reinterpret_cast<double2 *>(vd.exc + o) = make_double2(x[g][0], x[g][1]);
                    *reinterpret_cast<double2 *>(vd.exc + o + 2) = make_double2(x[g][2], x[g][3]);
                }
            }
            if (from_tab[g] && lane == 0) // the vocoder reads the row (whole, or the blocks named in bits 4..7)
                vd.exc_src[f] = blk_rows ? (uint8_t)(2u | (bmask[g] << 4)) : (uint8_t)0;
        }
    } // groups of frames
}

// --------------------------------------------------------------------------
// DPP helpers (f64 moves as two 32-bit DPP movs; invalid source lanes read 0).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114,
              DPP_ROW_SHR8 = 0x118, DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142;


// PCM sink of the reference's callers (examples/is-bonsai/main.rs:44-48, hound 16-bit WAV):
// value.min(i16::MAX).max(i16::MIN) as i16 -- clamp, then truncate toward zero.
__device__ __forceinline__ int pcm_i16(double v)
{
    v = fmin(v, 32767.0);
    v = fmax(v, -32768.0);
    return (int)v;
}
__device__ __forceinline__ uint32_t pcm_i16x2(double a, double b)
{
    return ((uint32_t)pcm_i16(a) & 0xffffu) | ((uint32_t)pcm_i16(b) << 16);
}

struct ScanCoef {
    double c1, c2, c4, c8, cb;
};

// segmented weighted inclusive scan over the 12-lane stage segments
__device__ __forceinline__ double seg_scan(double v, const ScanCoef &k)
{
    v = fma(k.c1, dpp_f64<DPP_ROW_SHR1>(v), v);
    v = fma(k.c2, dpp_f64<DPP_ROW_SHR2>(v), v);
    v = fma(k.c4, dpp_f64<DPP_ROW_SHR4>(v), v);
    v = fma(k.c8, dpp_f64<DPP_ROW_SHR8>(v), v);
    v = fma(k.cb, dpp_f64<DPP_ROW_BCAST15>(v), v);
    return v;
}

__device__ __forceinline__ double ipow(double x, int n)
{
    double r = 1.0;
    for (int i = 0; i < n; i++)
        r *= x;
    return r;
}

// PPADE (src/vocoder/mlsa.rs:31)
__device__ __constant__ double kPPade[6] = {1.00000000000, 0.49993910000, 0.11070980000,
                                            0.01369984000, 0.00095648530, 0.00003041721};

// filter state layout (doubles): d[TPL][64] | ulane[64] | e11[6] | e12[6] | pad[4]
__host__ __device__ inline int voc_state_doubles(int tpl) { return 64 * tpl + 64 + 16; }

// Workgroups of four waves, one work item per wave: a workgroup's waves are spread over the four
// SIMDs of its CU, which one-wave workgroups are not (1008 one-wave workgroups ran at 0.44 us per
// sample against 0.25 for 512: the dispatcher had doubled them up on SIMDs).
template <int TPL>
__global__ __launch_bounds__(256) void k_vocoder(BatchDev bd, VocDev vd, const VocWork *__restrict__ work,
                                                 uint32_t n_items)
{
    const uint32_t item = blockIdx.x * 4u + (threadIdx.x >> 6);
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
    const int nmcp = vd.nmcp, fp = vd.fperiod, bs = vd.bs, nblk = vd.nblk;
    const int M = nmcp - 1; // live taps 1..M
    const double a = vd.alpha, iaa = 1.0 - a * a, vol = vd.volume;

    // ---- lane roles ----
    const int s = lane / kGroups, g = lane % kGroups;
    const bool active = lane < kPade * kGroups;
    const bool head = active && g == 0;
    const double kappa = ipow(-a, TPL);
    ScanCoef kc, ks; // weighted (carry) and unit (sum) coefficients
    {
        const int r16 = lane & 15;
        auto ok = [&](int sh) { return active && g >= sh && r16 >= sh; };
        kc.c1 = ok(1) ? ipow(kappa, 1) : 0.0;
        kc.c2 = ok(2) ? ipow(kappa, 2) : 0.0;
        kc.c4 = ok(4) ? ipow(kappa, 4) : 0.0;
        kc.c8 = ok(8) ? ipow(kappa, 8) : 0.0;
        const int rowstart = lane & ~15, segstart = s * kGroups;
        const bool straddle = active && rowstart > 0 && segstart < rowstart;
        kc.cb = straddle ? ipow(kappa, lane - rowstart + 1) : 0.0;
        ks.c1 = kc.c1 != 0.0 ? 1.0 : 0.0;
        ks.c2 = kc.c2 != 0.0 ? 1.0 : 0.0;
        ks.c4 = kc.c4 != 0.0 ? 1.0 : 0.0;
        ks.c8 = kc.c8 != 0.0 ? 1.0 : 0.0;
        ks.cb = straddle ? 1.0 : 0.0;
    }
    const double nh = (active && !head) ? 1.0 : 0.0;           // takes carry from lane-1
    const double hmask = (head && s > 0) ? 1.0 : 0.0;          // stage input from previous stage
    const double l0mask = (lane == 0) ? 1.0 : 0.0;             // stage-1 input = d22[0]
    const double Pl = active ? kPPade[s + 1] : 0.0;
    // tap indices of this lane: j = g*TPL + k + 1
    int tapj[TPL];
    bool dotv[TPL]; // participates in the c[2..] dot product
#pragma unroll
    for (int k = 0; k < TPL; k++) {
        tapj[k] = g * TPL + k + 1;
        dotv[k] = active && tapj[k] >= 2 && tapj[k] <= M;
    }

    // ---- state ----
    double d[TPL], cd[TPL], cdinc[TPL], ctgt[TPL];
#pragma unroll
    for (int k = 0; k < TPL; k++)
        d[k] = 0.0;
    double ulane = 0.0;
    double e11[6], e12[6];
#pragma unroll
    for (int i = 0; i < 6; i++)
        e11[i] = e12[i] = 0.0;
    if (wk.load_state) {
        const double *sp = wk.load_state;
#pragma unroll
        for (int k = 0; k < TPL; k++)
            d[k] = sp[64 * k + lane];
        ulane = sp[64 * TPL + lane];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            e11[i] = sp[64 * TPL + 64 + i];
            e12[i] = sp[64 * TPL + 70 + i];
        }
    }

    auto save_state = [&](double *sp) {
#pragma unroll
        for (int k = 0; k < TPL; k++)
            sp[64 * k + lane] = d[k];
        sp[64 * TPL + lane] = ulane;
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                sp[64 * TPL + 64 + i] = e11[i];
                sp[64 * TPL + 70 + i] = e12[i];
            }
        }
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
        // ---- frame setup (vocoder/mod.rs:116-125) ----
        // c at frame start = previous frame's cc exactly (mod.rs:140); first frame: c = cc.
        const double *bcur = vd.bcoef + f * (uint64_t)nmcp;
        // beta > 0: the first frame starts from the un-filtered mc2b(spectrum) (mod.rs:80-89)
        const double *bprev = (t > 0) ? bcur - nmcp : (vd.bfirst ? vd.bfirst + (uint64_t)b * (uint64_t)nmcp : bcur);
#pragma unroll
        for (int k = 0; k < TPL; k++) {
            double c0v = dotv[k] ? bprev[tapj[k]] : 0.0;
            double c1v = dotv[k] ? bcur[tapj[k]] : 0.0;
            cd[k] = c0v;
            ctgt[k] = c1v;
            cdinc[k] = (c1v - c0v) / (double)fp;
        }
        double c1 = bprev[1];
        const double c1inc = (bcur[1] - c1) / (double)fp;
        // V5 gain exp(c[0]) with the interpolated c[0] (mod.rs:129-131): exp once per frame,
        // then gain *= exp(cinc0) per sample (240 roundings ~ 2e-14 relative)
        double gain = exp(bprev[0]);
        const double gq = exp((bcur[0] - bprev[0]) / (double)fp);
        for (int q = 0; q < nblk; q++) {
            const int i0 = q * bs; // first sample of block within frame
            const uint64_t n0 = (uint64_t)t * (uint64_t)fp + (uint64_t)i0; // within utterance
            // excitation (gain applied) of this block, lane = sample (k_excite)
            const int blen = min(bs, fp - i0); // (a shorter last block where bs does not divide the frame period)
            const double xin = lane < blen ? exc_block_ptr(vd, base, t, exc_code(vd, base, t), q)[i0 + lane] : 0.0;
            // =========== Phase B: blen serial filter steps ===========
            double ob = 0.0;
            for (int i = 0; i < blen; i++) {
                double x = readlane_f64(xin, i) * gain;
                gain *= gq;
                // ---- V6 df1 (mlsa.rs:54-66), uniform across lanes ----
                {
                    double out = 0.0;
#pragma unroll
                    for (int ii = 5; ii >= 1; ii--) {
                        e11[ii] = fma(iaa, e12[ii - 1], a * e11[ii]);
                        e12[ii] = e11[ii] * c1;
                        const double v = e12[ii] * kPPade[ii];
                        x += (ii & 1) ? v : -v;
                        out += v;
                    }
                    e12[0] = x;
                    x += out;
                }
                // ---- V7 df2: five fir() calls at once ----
                // carry-free aggregate of this lane's taps
                double loc = d[0];
#pragma unroll
                for (int k = 1; k < TPL; k++)
                    loc = fma(-a, loc, d[k]);
                double I = fma(kappa, ulane, loc); // ulane != 0 only on head lanes
                I = seg_scan(I, kc);
                double r = fma(nh, dpp_f64<DPP_WAVE_SHR1>(I), ulane); // rem entering tap 0 of lane
#pragma unroll
                for (int k = 0; k < TPL; k++) {
                    // all-pass section: rem' = d - a*rem ; d' = (1-a^2)*rem + a*d = rem + a*rem'
                    const double rn = fma(-a, r, d[k]);
                    d[k] = fma(a, rn, r);
                    r = rn;
                }
                double yl = cd[0] * d[0];
#pragma unroll
                for (int k = 1; k < TPL; k++)
                    yl = fma(cd[k], d[k], yl);
                const double Y = seg_scan(yl, ks); // stage sums at lanes 12*s+11
                const double Yp = Y * Pl;
                const double v1 = readlane_f64(Yp, 11), v2 = readlane_f64(Yp, 23),
                             v3 = readlane_f64(Yp, 35), v4 = readlane_f64(Yp, 47),
                             v5 = readlane_f64(Yp, 59);
                // Pade combine in the reference's order (mlsa.rs:71-78)
                x += v5;
                x -= v4;
                x += v3;
                x -= v2;
                x += v1;
                double out = v5;
                out += v4;
                out += v3;
                out += v2;
                out += v1;
                // d22[0] = x feeds stage 1; d22[i] = y_i feeds stage i+1 (next sample)
                ulane = fma(hmask, dpp_f64<DPP_WAVE_SHR1>(Y), l0mask * x);
                x += out;
                // ---- V8 ----
#pragma unroll
                for (int k = 0; k < TPL; k++)
                    cd[k] += cdinc[k];
                c1 += c1inc;
                ob = (lane == i) ? x * vol : ob;
            }
            if (lane < blen && emit) {
                if (vd.pcm16)
                    vd.pcm16[base * (uint64_t)fp + n0 + (uint64_t)lane] = (int16_t)pcm_i16(ob);
                else
                    vd.pcm[base * (uint64_t)fp + n0 + (uint64_t)lane] = ob;
            }
        }
        (void)ctgt;
    }

    if (wk.save_end)
        save_state(wk.save_end);
}

// --------------------------------------------------------------------------
// The same recursion on TWO waves per work item, for launches that leave half of the device's SIMDs idle anyway
// (n_items <= 2 per CU: a single sentence, the frames a streaming generator serves first, a redo round).
// A lone wave runs k_vocoder at the issue rate of its SIMD -- 117 instructions per sample, one every four cycles --
// and a third of them do not belong to the recursion's cross-lane part at all: the gain and df1 (mlsa.rs:54-66: five
// one-tap sections, the same value in every lane), the read of the excitation sample, the collection of the output.
// df1 feeds df2 and nothing comes back, so a PRODUCER wave runs excitation x gain -> df1 one block (bs samples) ahead
// and leaves its outputs in LDS; the CONSUMER wave runs df2 (mlsa.rs:68-94) on them and leaves its outputs in LDS, which
// the producer scales, converts and stores a block later.  Same operations in the same order as k_vocoder: same bits.
// A workgroup = 4 waves = 2 items: waves 0, 1 the consumers, waves 2, 3 the producers -- one wave per SIMD of the CU;
// one barrier per block keeps the two double buffers in step (a block is ~20,000 cycles of work).
// ITEMS = 4 (eight waves, launches of up to four items per CU): waves w and w + 4 of a workgroup share a SIMD, so the
// consumer of an item and its producer do -- the producer's 40 instructions per sample fit into the bubbles of the
// consumer's dependent chain (78 instructions in ~470 cycles).
template <int TPL, int ITEMS>
__global__ __launch_bounds__(128 * ITEMS) void k_vocoder_pair(BatchDev bd, VocDev vd, const VocWork *__restrict__ work,
                                                              uint32_t n_items)
{
    __shared__ double xbuf[ITEMS][2][64]; // [item of the workgroup][block parity][sample]: df1 outputs
    __shared__ double obuf[ITEMS][2][64]; // df2 outputs (before the volume)
    __shared__ uint32_t nblocks_sh[ITEMS];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = wv % ITEMS;
    const bool producer = wv >= ITEMS;
    const uint32_t item = blockIdx.x * (uint32_t)ITEMS + (uint32_t)slot;
    const int fp = vd.fperiod, bs = vd.bs, nblk = vd.nblk, nmcp = vd.nmcp;
    VocWork wk{};
    uint32_t t_begin = 0, t_end = 0, t_out = 0;
    int b = 0;
    if (item < n_items) {
        wk = work[item];
        b = (int)wk.utt;
        t_begin = wk.t_start;
        t_out = wk.t_out;
        t_end = min(wk.t_end, bd.utt[b].T);
        if (t_begin >= t_end)
            t_begin = t_end = 0;
    }
    const uint32_t NB = (t_end - t_begin) * (uint32_t)nblk; // blocks of this item
    if (!producer && lane == 0)
        nblocks_sh[slot] = NB;
    __syncthreads();
    uint32_t NBmax = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; j++)
        NBmax = max(NBmax, nblocks_sh[j]);
    const uint64_t base = NB ? bd.utt[b].frame_off : 0;
    const double a = vd.alpha, iaa = 1.0 - a * a, vol = vd.volume;
    double *const xb = &xbuf[slot][0][0];
    double *const ob = &obuf[slot][0][0];

    if (producer) {
        // ================= producer: gain, df1, PCM stores =================
        double e11[6], e12[6];
#pragma unroll
        for (int i = 0; i < 6; i++)
            e11[i] = e12[i] = 0.0;
        if (NB && wk.load_state) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                e11[i] = wk.load_state[64 * TPL + 64 + i];
                e12[i] = wk.load_state[64 * TPL + 70 + i];
            }
        }
        auto save_state = [&](double *sp) {
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    sp[64 * TPL + 64 + i] = e11[i];
                    sp[64 * TPL + 70 + i] = e12[i];
                }
            }
        };
        uint32_t t = t_begin;
        int q = 0;
        double c1 = 0.0, c1inc = 0.0, gain = 0.0, gq = 0.0;
        for (uint32_t st = 0; st < NBmax + 2; st++) {
            // (1) the block the consumer finished in the previous step: scale, convert, store
            if (st >= 2 && st - 2 < NB) {
                const uint32_t gb = st - 2;
                const uint32_t ts = t_begin + gb / (uint32_t)nblk;
                const int qs = (int)(gb % (uint32_t)nblk), i0 = qs * bs;
                const int blen = min(bs, fp - i0);
                if (ts >= t_out && lane < blen) {
                    const double v = ob[64 * (gb & 1u) + lane] * vol;
                    const uint64_t o = base * (uint64_t)fp + (uint64_t)ts * (uint64_t)fp + (uint64_t)(i0 + lane);
                    if (vd.pcm16)
                        vd.pcm16[o] = (int16_t)pcm_i16(v);
                    else
                        vd.pcm[o] = v;
                }
            }
            // (2) df1 of block st
            if (st < NB) {
                if (q == 0) {
                    if (t == t_out && t_out > t_begin && wk.save_warm)
                        save_state(wk.save_warm);
                    if (t == t_out + vd.ckpt_frames && wk.save_ckpt)
                        save_state(wk.save_ckpt);
                    if (vd.ckpt2_frames && t == t_out + vd.ckpt2_frames && wk.save_ckpt2)
                        save_state(wk.save_ckpt2);
                    // frame setup (vocoder/mod.rs:116-131), as k_vocoder
                    const double *bcur = vd.bcoef + (base + t) * (uint64_t)nmcp;
                    const double *bprev =
                        (t > 0) ? bcur - nmcp : (vd.bfirst ? vd.bfirst + (uint64_t)b * (uint64_t)nmcp : bcur);
                    c1 = bprev[1];
                    c1inc = (bcur[1] - c1) / (double)fp;
                    gain = exp(bprev[0]);
                    gq = exp((bcur[0] - bprev[0]) / (double)fp);
                }
                const int i0 = q * bs;
                const int blen = min(bs, fp - i0);
                const double xin = lane < blen ? exc_block_ptr(vd, base, t, exc_code(vd, base, t), q)[i0 + lane] : 0.0;
                double xo = 0.0;
                for (int i = 0; i < blen; i++) {
                    double x = readlane_f64(xin, i) * gain;
                    gain *= gq;
                    // ---- V6 df1 (mlsa.rs:54-66), uniform across lanes ----
                    double out = 0.0;
#pragma unroll
                    for (int ii = 5; ii >= 1; ii--) {
                        e11[ii] = fma(iaa, e12[ii - 1], a * e11[ii]);
                        e12[ii] = e11[ii] * c1;
                        const double v = e12[ii] * kPPade[ii];
                        x += (ii & 1) ? v : -v;
                        out += v;
                    }
                    e12[0] = x;
                    x += out;
                    c1 += c1inc;
                    xo = (lane == i) ? x : xo;
                }
                if (lane < blen)
                    xb[64 * (st & 1u) + lane] = xo;
                if (++q == nblk) {
                    q = 0;
                    t++;
                }
            }
            __syncthreads();
        }
        if (NB && wk.save_end)
            save_state(wk.save_end);
        return;
    }

    // ================= consumer: df2 =================
    const int M = nmcp - 1; // live taps 1..M
    const int s = lane / kGroups, g = lane % kGroups;
    const bool active = lane < kPade * kGroups;
    const bool head = active && g == 0;
    const double kappa = ipow(-a, TPL);
    ScanCoef kc, ks; // weighted (carry) and unit (sum) coefficients
    {
        const int r16 = lane & 15;
        auto ok = [&](int sh) { return active && g >= sh && r16 >= sh; };
        kc.c1 = ok(1) ? ipow(kappa, 1) : 0.0;
        kc.c2 = ok(2) ? ipow(kappa, 2) : 0.0;
        kc.c4 = ok(4) ? ipow(kappa, 4) : 0.0;
        kc.c8 = ok(8) ? ipow(kappa, 8) : 0.0;
        const int rowstart = lane & ~15, segstart = s * kGroups;
        const bool straddle = active && rowstart > 0 && segstart < rowstart;
        kc.cb = straddle ? ipow(kappa, lane - rowstart + 1) : 0.0;
        ks.c1 = kc.c1 != 0.0 ? 1.0 : 0.0;
        ks.c2 = kc.c2 != 0.0 ? 1.0 : 0.0;
        ks.c4 = kc.c4 != 0.0 ? 1.0 : 0.0;
        ks.c8 = kc.c8 != 0.0 ? 1.0 : 0.0;
        ks.cb = straddle ? 1.0 : 0.0;
    }
    const double nh = (active && !head) ? 1.0 : 0.0;  // takes carry from lane-1
    const double hmask = (head && s > 0) ? 1.0 : 0.0; // stage input from previous stage
    const double l0mask = (lane == 0) ? 1.0 : 0.0;    // stage-1 input = d22[0]
    const double Pl = active ? kPPade[s + 1] : 0.0;
    int tapj[TPL];
    bool dotv[TPL];
#pragma unroll
    for (int k = 0; k < TPL; k++) {
        tapj[k] = g * TPL + k + 1;
        dotv[k] = active && tapj[k] >= 2 && tapj[k] <= M;
    }
    double d[TPL], cd[TPL], cdinc[TPL];
#pragma unroll
    for (int k = 0; k < TPL; k++)
        d[k] = cd[k] = cdinc[k] = 0.0;
    double ulane = 0.0;
    if (NB && wk.load_state) {
#pragma unroll
        for (int k = 0; k < TPL; k++)
            d[k] = wk.load_state[64 * k + lane];
        ulane = wk.load_state[64 * TPL + lane];
    }
    auto save_state = [&](double *sp) {
#pragma unroll
        for (int k = 0; k < TPL; k++)
            sp[64 * k + lane] = d[k];
        sp[64 * TPL + lane] = ulane;
    };
    uint32_t t = t_begin;
    int q = 0;
    for (uint32_t st = 0; st < NBmax + 2; st++) {
        if (st >= 1 && st - 1 < NB) {
            const uint32_t gb = st - 1;
            if (q == 0) {
                if (t == t_out && t_out > t_begin && wk.save_warm)
                    save_state(wk.save_warm);
                if (t == t_out + vd.ckpt_frames && wk.save_ckpt)
                    save_state(wk.save_ckpt);
                if (vd.ckpt2_frames && t == t_out + vd.ckpt2_frames && wk.save_ckpt2)
                    save_state(wk.save_ckpt2);
                const double *bcur = vd.bcoef + (base + t) * (uint64_t)nmcp;
                const double *bprev = (t > 0) ? bcur - nmcp : (vd.bfirst ? vd.bfirst + (uint64_t)b * (uint64_t)nmcp : bcur);
#pragma unroll
                for (int k = 0; k < TPL; k++) {
                    const double c0v = dotv[k] ? bprev[tapj[k]] : 0.0;
                    const double c1v = dotv[k] ? bcur[tapj[k]] : 0.0;
                    cd[k] = c0v;
                    cdinc[k] = (c1v - c0v) / (double)fp;
                }
            }
            const int blen = min(bs, fp - q * bs);
            const double *xs = xb + 64 * (gb & 1u);
            double *os = ob + 64 * (gb & 1u);
            double xn = xs[0]; // the df1 output of the next sample is requested a sample ahead of its use
            for (int i = 0; i < blen; i++) {
                double x = xn;
                xn = xs[i + 1 < blen ? i + 1 : i];
                // ---- V7 df2: five fir() calls at once (as k_vocoder) ----
                double loc = d[0];
#pragma unroll
                for (int k = 1; k < TPL; k++)
                    loc = fma(-a, loc, d[k]);
                double I = fma(kappa, ulane, loc); // ulane != 0 only on head lanes
                I = seg_scan(I, kc);
                double r = fma(nh, dpp_f64<DPP_WAVE_SHR1>(I), ulane); // rem entering tap 0 of lane
#pragma unroll
                for (int k = 0; k < TPL; k++) {
                    const double rn = fma(-a, r, d[k]);
                    d[k] = fma(a, rn, r);
                    r = rn;
                }
                double yl = cd[0] * d[0];
#pragma unroll
                for (int k = 1; k < TPL; k++)
                    yl = fma(cd[k], d[k], yl);
                const double Y = seg_scan(yl, ks); // stage sums at lanes 12*s+11
                const double Yp = Y * Pl;
                const double v1 = readlane_f64(Yp, 11), v2 = readlane_f64(Yp, 23), v3 = readlane_f64(Yp, 35),
                             v4 = readlane_f64(Yp, 47), v5 = readlane_f64(Yp, 59);
                // Pade combine in the reference's order (mlsa.rs:71-78)
                x += v5;
                x -= v4;
                x += v3;
                x -= v2;
                x += v1;
                double out = v5;
                out += v4;
                out += v3;
                out += v2;
                out += v1;
                ulane = fma(hmask, dpp_f64<DPP_WAVE_SHR1>(Y), l0mask * x);
                x += out;
#pragma unroll
                for (int k = 0; k < TPL; k++)
                    cd[k] += cdinc[k];
                if (lane == 0)
                    os[i] = x;
            }
            if (++q == nblk) {
                q = 0;
                t++;
            }
        }
        __syncthreads();
    }
    if (NB && wk.save_end)
        save_state(wk.save_end);
}

// --------------------------------------------------------------------------
// Lane-triple throughput kernel: ONE CHUNK PER THREE ADJACENT LANES.
// With time-chunking there are tens of thousands of independent recursions per batch, so the
// cross-lane machinery of k_vocoder (DPP scans, readlane combines: ~128 VALU instructions per
// sample, most of them data movement) can be dropped.  The five Pade stages of df2 exchange data
// only BETWEEN samples (stage i reads d22[i-1] of the previous sample, mlsa.rs:71-77), so a chunk
// is split over three adjacent lanes, stages {1,2} {3,4} {5,-}, each stage with its (nmcp-1)-tap
// state in architectural VGPRs (68 state doubles per lane), taps outermost so the two slots give
// 2-way ILP; everything but the per-sample exchange is lane-local.  df1's five stages (one tap each) are
// spread the same way.  The interpolated coefficients c(n) = c0 + i*cinc live in LDS as [tap][chunk] pairs
// (broadcast ds_read_b128).  ~13.5 VALU instructions per chunk-sample instead of ~128.
// 21 triples over lanes 0..62 of the wave (lane 63 idle), so 21 chunks per wave; the exchange uses
// the whole-wave DPP shifts (a triple may straddle a 16-lane DPP row): one wave_shr:1 (stage
// outputs to the next lane) and, per Pade partial sum, two wave_shl:1 that fold positions
// 2 -> 1 -> 0 of the triple.  State dumps use k_vocoder's layout so that verification and re-do are
// shared (slots nobody writes stay zero: the state buffers are zeroed at allocation).
// (Earlier forms -- a chunk per lane PAIR with 102 state doubles per lane, which spilled; five triples
// per 16-lane DPP row; single-instruction asm statements; the excitation load behind a per-lane test --
// are recorded in tools/experiments/README.md.)
constexpr int DPP_WAVE_SHL1 = 0x130;
constexpr int DPP_ROW_SHL1 = 0x101, DPP_ROW_SHL2 = 0x102, DPP_ROW_SHL4 = 0x104;
// sum of five adjacent lanes of a DPP row, valid on the first of them: (s0 + s1) + (s2 + s3) + s4
__device__ __forceinline__ double fold5(double s)
{
    const double t1 = s + dpp_f64<DPP_ROW_SHL1>(s);
    const double t2 = t1 + dpp_f64<DPP_ROW_SHL2>(t1);
    return t2 + dpp_f64<DPP_ROW_SHL4>(s);
}
__host__ __device__ constexpr int lt_chunks(int lpc) { return lpc == 3 ? 21 : 12; } // chunks per wave
constexpr int kLtPf = 4; // coefficient reads in flight ahead of their use (2, 3 or 4 measure the same)
// Waves per workgroup.  EIGHT = a whole CU (two waves of 256 VGPRs per SIMD): waves w and w + 4 of a workgroup land on
// the same SIMD, so the two waves that share a SIMD can see each other's progress in LDS.  The issue arbiter serves
// the OLDER wave of a SIMD first: with independent one-wave workgroups the older wave ran at its solo rate (6.7
// cycles per instruction) and finished after 44.5 ms, the younger one got the gaps (12.6 cycles per instruction) and
// then ran ALONE for 17 ms at the solo rate -- the kernel took 61.5 ms where two waves sharing the pipe to the end
// need 58 (tools/lt_clocks.sh).  The pair now keeps level: every kLtBalance samples a wave posts its sample count
// and raises its priority if it is behind its partner, lowers it if ahead.
// FOUR waves per workgroup = one wave per SIMD, for batches that cannot give every SIMD two waves of chunks that
// are long against their warm-up: a lone wave issues an instruction every 6.7 cycles, one of a pair every 8.9.
constexpr int kLtBalance = 8;
#ifdef JB_LT_STAMPS
// occupancy aid (tools/lt_occupancy.sh; round 6): every wave of the kernel leaves the 100 MHz clock at its start and at
// its end, where it ran (HW_ID: SIMD, CU, SE; XCC_ID) and how many frames it walked -- which SIMD held two waves when
constexpr unsigned kLtStampWaves = 4096;
__device__ unsigned long long g_lt_stamp[kLtStampWaves][4];
#endif
#ifdef JB_LT_CLOCKS
// timing aid (tools/lt_clocks.sh): shader clock (s_memtime) and the constant 100 MHz clock (s_memrealtime) at the
// start and end of two waves of the kernel: the clock the chip holds under this kernel
__device__ unsigned long long g_lt_clk[8];
#endif

// Every order the reference takes (round 6).  NM is the order + 1 the CODE is built for: EXACT instantiations run voices
// of exactly that order (nitech's 35 and 25: the tuned forms, unchanged); the others run any order up to NM with the
// taps above the voice's own held at coefficient zero (the all-pass chain goes on through them and nothing reads it:
// same sums), the voice's order, its coefficient stride and its state layout read from `vd` at run time.
// LPC = lanes per chunk: 3 (two stage slots per lane, 21 chunks per wave: orders up to 35, whose 2 x 34 state doubles
// per lane fill the 256 VGPRs) or 5 (ONE stage per lane, 12 chunks per wave on lanes 0..14 of every 16-lane DPP row:
// up to 60 state doubles per lane, orders 36..61).  Per tap a lane then issues 3 FMAs + 1 interpolation for one
// stage where the triple issues 6 + 1 for two: the same 15 stage-taps per instruction, without the inert sixth slot
// and with five-lane folds (row_shl 1, 2, 4) instead of three-lane ones.
template <int NM, int TPLW_, int kLtWaves, bool EXACT = true, int LPC = 3>
__global__ __launch_bounds__(64 * kLtWaves, 2) void k_vocoder_lt(BatchDev bd, VocDev vd,
                                                       const VocWork *__restrict__ work,
                                                       const uint32_t *__restrict__ order,
                                                       uint32_t n_items)
{
    static_assert(LPC == 3 || LPC == 5, "three lanes x two stage slots, or five lanes x one");
    constexpr int M = NM - 1;            // taps 1..M of the code
    constexpr int NS = LPC == 3 ? 2 : 1; // stage slots per lane
    constexpr int kLtChunks = lt_chunks(LPC);
    const int nm = EXACT ? NM : vd.nmcp; // the voice's order + 1 (<= NM)
    const int mreal = nm - 1;
    const int TPLW = EXACT ? TPLW_ : (mreal + kGroups - 1) / kGroups; // taps per lane of k_vocoder's state layout
    const int lane = threadIdx.x % 64;
    const int wv = threadIdx.x / 64; // wave of the workgroup (uniform)
    // LPC 3 -- pos 0: stages 0,1; 1: stages 2,3; 2: stage 4 + inert slot (the idle lane 63 behaves like pos 2)
    // LPC 5 -- pos = stage; lane 15 of every row idles (behaves like pos 4 of the row's last chunk)
    const bool idle = LPC == 3 ? lane == 63 : (lane & 15) == 15;
    const int pos = LPC == 3 ? (idle ? 2 : lane % 3) : (idle ? 4 : (lane & 15) % 5);
    const int ci = LPC == 3 ? (idle ? 20 : lane / 3) : (lane >> 4) * 3 + (idle ? 2 : (lane & 15) / 5); // chunk slot
    const uint32_t slot = (blockIdx.x * (uint32_t)kLtWaves + (uint32_t)wv) * (uint32_t)kLtChunks + (uint32_t)ci;
    const bool has = !idle && slot < n_items;
    const bool lead = has && pos == 0;
    const uint32_t item = has ? order[slot] : 0u;
    struct {
        uint32_t utt, t_start, t_out, t_end;
        const double *load_state;
    } wk = {0, 0, 0, 0, nullptr};
    if (has) {
        const VocWork &w0 = work[item];
        wk.utt = w0.utt;
        wk.t_start = w0.t_start;
        wk.t_out = w0.t_out;
        wk.t_end = w0.t_end;
        wk.load_state = w0.load_state;
    }
    const uint32_t T = has ? bd.utt[wk.utt].T : 0;
    if (wk.t_end > T)
        wk.t_end = T;
    const uint32_t nfr = wk.t_end > wk.t_start ? wk.t_end - wk.t_start : 0;
    uint32_t maxfr = nfr;
    for (int o = 32; o > 0; o >>= 1)
        maxfr = max(maxfr, (uint32_t)__shfl_xor((int)maxfr, o));
    // (every wave works on LDS of its own and there is no workgroup barrier in this kernel: a wave may leave)
    __shared__ uint32_t prog[8]; // samples done, per wave
    if (kLtWaves == 8 && lane == 0)
        prog[wv] = maxfr == 0 ? 0xffffffffu : 0u;
    if (maxfr == 0)
        return;
#ifdef JB_LT_STAMPS
    const unsigned stamp_w = blockIdx.x * (unsigned)kLtWaves + (unsigned)wv;
    if (lane == 0 && stamp_w < kLtStampWaves) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        g_lt_stamp[stamp_w][0] = wall_clock64();
        g_lt_stamp[stamp_w][2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        g_lt_stamp[stamp_w][3] = maxfr;
    }
#endif
#ifdef JB_LT_CLOCKS
    const int clk_slot = wv != 0 ? -1 : blockIdx.x == 0 ? 0 : blockIdx.x == gridDim.x / 2 ? 4 : -1;
    if (clk_slot >= 0 && lane == 0) {
        g_lt_clk[clk_slot] = clock64();
        g_lt_clk[clk_slot + 1] = wall_clock64();
    }
#endif
    const uint64_t base = has ? bd.utt[wk.utt].frame_off : 0;
    const int fp = vd.fperiod;
    const double a = vd.alpha, na = -a, iaa = 1.0 - a * a, vol = vd.volume;
    const int s0 = NS * pos; // first stage of this lane
    // Pade weights of the two slots: stage s -> PPADE[s+1]; slot 0 (odd i) enters the
    // alternating sum with +, slot 1 (even i) with -; the inert slot has weight 0
    const double w0 = kPPade[s0 + 1];
    const double w1 = (NS == 2 && s0 + 1 < kPade) ? kPPade[s0 + 2] : 0.0;
    // (one stage per lane: stages 0, 2, 4 enter the alternating sum with +, stages 1, 3 with -)
    const double wa = (NS == 1 && (pos & 1)) ? -w0 : w0;

    __shared__ double2 cc_[kLtWaves][NM - 1][kLtChunks]; // row k-1: (c_k at frame start, per-sample increment)
    __shared__ double gqs_[kLtWaves][kLtChunks];         // per-sample gain ratio exp(cinc0) of the current frame
    double2 (*const cc)[kLtChunks] = cc_[wv];
    double *const gqs = gqs_[wv];
    // the tables are the wave's own and LDS serves a wave's operations in order: ordering them takes no barrier, only
    // that the compiler keeps the stores of a frame's set-up between the asm-issued reads of the two frames
#define JB_LT_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

    if (!EXACT && has) // the taps above the voice's own: coefficient zero, once
        for (int k = nm + pos; k < NM; k += LPC)
            cc[k - 1][ci] = make_double2(0.0, 0.0);
    double d[NS][M + 1];
    double u[NS]; // slot inputs (d22[stage])
    double gain = 1.0;
    // df1 state, spread over the triple like df2's: this lane's two stages (d11, pt1 of stages s0+1, s0+2) and the
    // input of its first one (pt1[s0]: the lead lane's is pt1[0], the others' comes from the lane before)
    double e11[NS], e12[NS], ein = 0.0;
#pragma unroll
    for (int q = 0; q < NS; q++) {
        e11[q] = e12[q] = 0.0;
        u[q] = 0.0;
#pragma unroll
        for (int j = 0; j <= M; j++)
            d[q][j] = 0.0;
    }

    // k_vocoder state layout: tap j of stage s at [64*k + 12*s + g], j-1 = g*TPLW + k
    // (run-time layout: the slot of a tap is walked, not divided out, behind an opaque copy of the layout's width --
    // seen as loop invariants the 2 x M slot offsets were hoisted out of the frame loop and cost a hundred registers)
    auto layout_tpl = [&]() {
        int tp = TPLW;
        if (!EXACT)
            asm volatile("" : "+s"(tp));
        return tp;
    };
    if (wk.load_state) {
        const double *sp = wk.load_state;
        const int tpl = layout_tpl();
#pragma unroll
        for (int q = 0; q < NS; q++) {
            const int st = s0 + q;
            if (st < kPade) {
                u[q] = sp[64 * tpl + kGroups * st];
                int kk = 0, gg = 0; // (j - 1) % tpl, (j - 1) / tpl
#pragma unroll
                for (int j = 1; j <= M; j++) {
                    if (EXACT)
                        d[q][j] = sp[64 * ((j - 1) % TPLW_) + kGroups * st + (j - 1) / TPLW_];
                    else if (j <= mreal)
                        d[q][j] = sp[64 * kk + kGroups * st + gg];
                    if (++kk == tpl) {
                        kk = 0;
                        gg++;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NS; q++) {
            const int st = s0 + 1 + q; // df1 stage 1..5 (k_vocoder's dump: d11[i] at +64+i, pt1[i] at +70+i)
            if (st <= kPade) {
                e11[q] = sp[64 * tpl + 64 + st];
                e12[q] = sp[64 * tpl + 70 + st];
            }
        }
        ein = sp[64 * tpl + 70 + s0];
    }
    auto save_state = [&](double *sp) {
        const int tpl = layout_tpl();
#pragma unroll
        for (int q = 0; q < NS; q++) {
            const int st = s0 + q;
            if (st < kPade) {
                sp[64 * tpl + kGroups * st] = u[q];
                int kk = 0, gg = 0;
#pragma unroll
                for (int j = 1; j <= M; j++) {
                    if (EXACT)
                        sp[64 * ((j - 1) % TPLW_) + kGroups * st + (j - 1) / TPLW_] = d[q][j];
                    else if (j <= mreal)
                        sp[64 * kk + kGroups * st + gg] = d[q][j];
                    if (++kk == tpl) {
                        kk = 0;
                        gg++;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NS; q++) {
            const int st = s0 + 1 + q;
            if (st <= kPade) {
                sp[64 * tpl + 64 + st] = e11[q];
                sp[64 * tpl + 70 + st] = e12[q];
            }
        }
        if (pos == 0) {
            sp[64 * tpl + 64] = 0.0; // d11[0] is never written by the recursion
            sp[64 * tpl + 70] = ein; // pt1[0]
        }
    };

    for (uint32_t tl = 0; tl < maxfr; tl++) {
        const bool act = tl < nfr;
        const uint32_t t = wk.t_start + (act ? tl : 0);
        const uint64_t f = base + t;
        const bool emit = act && lead && t >= wk.t_out;
        if (act && t == wk.t_out && wk.t_out > wk.t_start) {
            double *sw_ = work[item].save_warm;
            if (sw_)
                save_state(sw_);
        }
        if (act && t == wk.t_out + vd.ckpt_frames) {
            double *sc_ = work[item].save_ckpt;
            if (sc_)
                save_state(sc_);
        }
        if (act && vd.ckpt2_frames && t == wk.t_out + vd.ckpt2_frames) {
            double *sc_ = work[item].save_ckpt2;
            if (sc_)
                save_state(sc_);
        }
        // frame setup (vocoder/mod.rs:116-125): c = previous target, cinc = (cc - c)/fperiod;
        // the three lanes of a triple fill every third tap
        JB_LT_FENCE();
        if (has) {
            const double *bcur = vd.bcoef + f * (uint64_t)nm;
            const double *bprev = (t > 0) ? bcur - nm : (vd.bfirst ? vd.bfirst + (uint64_t)wk.utt * nm : bcur);
            for (int k = 1 + pos; k < nm; k += LPC) {
                const double c0v = bprev[k], c1v = bcur[k];
                cc[k - 1][ci] = make_double2(c0v, (c1v - c0v) / (double)fp);
            }
            // V5 gain exp(c[0]) (mod.rs:129-131): exp once per frame, gain *= exp(cinc0) per sample
            gain = exp(bprev[0]);
            if (pos == 0)
                gqs[ci] = exp((bcur[0] - bprev[0]) / (double)fp);
        }
        JB_LT_FENCE();
        // the excitation of a frame comes block by block (bs samples): its row of xin, the shared table or the
        // noise stream (exc_block_ptr); a lane without work reads the noise table
        const uint32_t xcode = act ? exc_code(vd, base, t) : 1u;
        const int xbs = vd.bs;
        auto xblock = [&](int q) -> const double * { return act ? exc_block_ptr(vd, base, t, xcode, q) : vd.noise; };
        const double *xq = xblock(0);
        double *op = vd.pcm + (base + t) * (uint64_t)fp;
        double xn = act ? xq[0] : 0.0;
        const double gq = gqs[ci];
        // The next sample's excitation is requested at the top of a sample and used at the top of the next
        // one, a whole sample (~1.5 us) later.  Two things made that load cost 3.7 of the kernel's 66 ms
        // (measured by removing it): it sat behind a per-lane test (`act ? ... : 0`), which splits the
        // sample's straight-line code at the top, and the wait in front of its use is vmcnt(0), which every
        // fourth sample also waited for the PCM stores the sample before had just issued.  So: the load is
        // unconditional (a lane without work reads the noise table: a valid address, finite values it
        // never uses) and an asm statement with its own wait; and the PCM leaves the lead lane in PAIRS,
        // one sample late -- at the top of a sample, before the load is issued, so that what the wait
        // covers is a whole sample old.  (Moving the excitation through an LDS ring filled by LDS-DMA, and
        // a wait that counts the stores, both cost more than they saved: tools/experiments/.)
        double oA = 0.0, oB = 0.0;
        auto put_pair = [&](int at) { // samples at, at+1 of this frame
            if (vd.pcm16)
                *reinterpret_cast<uint32_t *>(vd.pcm16 + (base + t) * (uint64_t)fp + at) = pcm_i16x2(oA, oB);
            else
                *reinterpret_cast<double2 *>(op + at) = make_double2(oA, oB);
        };
        // TWO samples ahead, the sample loop unrolled by two: x of even samples lives in xn, of odd ones in
        // xo.  An even sample waits with vmcnt(1): everything but the youngest operation, the odd sample's
        // request, is then done -- its own request and the older pair store.  An odd sample waits with
        // vmcnt(0): what is younger than its request, the even sample's store and request, is a whole
        // sample old.  (Counting that store as well, vmcnt(2), is wrong: stores are not retired in order
        // with loads -- the hand-off check caught it, every chunk was redone.)
        double xo;
        {
            const int n1 = fp > 1 ? 1 : 0;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(xo) : "v"(xq + n1) : "memory");
        }
        auto sample = [&](const int i, auto ODDC) {
            constexpr bool ODD = decltype(ODDC)::value;
            double x;
            if (ODD) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(xo)::"memory");
                x = xo * gain;
            } else {
                asm volatile("s_waitcnt vmcnt(1)" : "+v"(xn)::"memory");
                x = xn * gain;
            }
            gain *= gq;
            if (!ODD && i > 0 && emit)
                put_pair(i - 2);
            {
                const int nx = i + 2 < fp ? i + 2 : fp - 1; // (the last samples re-read the last: nothing past the row)
                if (ODD)
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(xo) : "v"(xq + nx) : "memory");
                else
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(xn) : "v"(xq + nx) : "memory");
            }
            const double fi = (double)i;
            // ---- V6 df1 (mlsa.rs:54-66), its five stages over the triple like df2's (every stage reads the
            // PREVIOUS sample's output of the stage before it): this lane's second stage first, then its first.
            // The alternating sum is folded to the lead lane here (it is pt1[0], the first stage's next input);
            // the plain sum joins df2's alternating one below, where both are added to the same value.
            double db1;
            {
                const double2 c1p = cc[0][ci];
                const double c1 = fma(fi, c1p.y, c1p.x);
                if constexpr (NS == 2) {
                    e11[1] = fma(iaa, e12[0], a * e11[1]);
                    e11[0] = fma(iaa, ein, a * e11[0]);
                    e12[1] = e11[1] * c1;
                    e12[0] = e11[0] * c1;
                    // w0*pt1[s0+1] -/+ w1*pt1[s0+2], each as one product and one FMA (an instruction less than two
                    // products, a difference and a sum)
                    const double dv1 = w1 * e12[1];
                    const double da = fma(w0, e12[0], -dv1);
                    db1 = fma(w0, e12[0], dv1);
                    x += da + dpp_f64<DPP_WAVE_SHL1>(da + dpp_f64<DPP_WAVE_SHL1>(da)); // pt1[0] (valid on the lead lane)
                } else {
                    e11[0] = fma(iaa, ein, a * e11[0]);
                    e12[0] = e11[0] * c1;
                    db1 = w0 * e12[0];
                    x += fold5(wa * e12[0]); // pt1[0] (valid on the lead lane)
                }
                const double eprev = dpp_f64<DPP_WAVE_SHR1>(e12[NS - 1]);
                ein = pos == 0 ? x : eprev;
            }
            // ---- V7 df2: fir() of this lane's stage slots, taps outermost (mlsa.rs:127-163) ----
            double r[NS], y[NS];
#pragma unroll
            for (int q = 0; q < NS; q++)
                r[q] = u[q];
            static_assert(M >= 2, "the dot products start at tap 2");
            // Coefficient reads run kLtPf taps ahead of their use in rotating registers so that
            // their LDS latency overlaps the arithmetic in between.  hipcc sinks ordinary LDS loads
            // next to their use, so they are issued with inline asm and counted s_waitcnt
            // lgkmcnt(N); each wait is tied to its data register ("+v"), which keeps consumers
            // behind it.  LDS returns in order, so the compiler's own waits stay correct.
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d cq[kLtPf];
            const uint32_t cca = (uint32_t)(uintptr_t)&cc[0][ci];
#define JB_LDS_RD(dst, tapj)                                                                     \
    asm volatile("ds_read_b128 %0, %1 offset:%2"                                                 \
                 : "=v"(dst)                                                                     \
                 : "v"(cca), "n"(((tapj)-1) * (int)(sizeof(double2) * kLtChunks)))
#pragma unroll
            for (int k = 0; k < kLtPf; k++)
                JB_LDS_RD(cq[k], 2 + k); // taps 2 .. 1+PF
            // The coefficients are waited for in PAIRS, one tap block ahead of the first use, and each tap is followed
            // by one interpolation and one new request:
            //     W(2,3) c2 R6 B1 | W(4,5) c3 R7 B2 | c4 R8 B3 | W(6,7) c5 R9 B4 | c6 R10 B5 | ...
            // Per sample that is 18 waits instead of 34 and one s_nop instead of 11.  (The hazard recogniser puts an
            // s_nop between an asm statement and a following instruction that reads a register the statement writes
            // unless a compiler-emitted instruction stands between them -- asm statements count as nothing: so the
            // interpolation of the NEXT tap stands between two tap blocks, and the wait that defines a pair's
            // registers stands one interpolation before their first use.)  Every instruction of a wave that is not
            // a VALU instruction is an issue slot in which the pipe is busy only if the other wave of the SIMD has a
            // VALU instruction ready.
            static_assert(kLtPf == 4 && M % 2 == 0 && M >= 6, "pair schedule: four slots, taps 2..M in pairs + one");
            double cv[M + 2];
            auto wait_pair = [&](const int j, const int younger) { // taps j, j+1
                v2d &qa = cq[(j - 2) % kLtPf], &qb = cq[(j - 1) % kLtPf];
                switch (younger) {
                case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qa), "+v"(qb)); break;
                case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(qa), "+v"(qb)); break;
                default: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(qa), "+v"(qb)); break;
                }
            };
            cv[1] = 0.0;
#pragma unroll
            for (int j = 1; j <= M; j++) {
                if (j == 1)
                    wait_pair(2, 2); // requests 4, 5 are younger
                else if (j % 2 == 0 && j + 2 <= M)
                    wait_pair(j + 2, j + 4 <= M ? 1 : 0); // (tap M + 1 does not exist: its slot is just tied)
                if (j + 1 <= M) {
                    const int sq = (j - 1) % kLtPf; // slot of tap j + 1
                    cv[j + 1] = fma(fi, cq[sq].y, cq[sq].x);
                    if (j + 5 <= M)
                        JB_LDS_RD(cq[sq], j + 5);
                }
                const double cj = cv[j];
                // all-pass section of tap j for both stage slots, then the dot-product terms:
                //   rem' = d - a*rem ; d' = (1-a^2)*rem + a*d = rem + a*rem' ; y += c_j * d'
                // as ONE block of three-address v_fma_f64.  (a) hipcc would select the destructive
                // v_fmac_f64 and then need a v_mov_b64 per tap to undo the register rotation of the
                // loop-carried d[] (69 moves per sample, ~19 % of the VALU work); (b) around
                // single-instruction asm statements its hazard recogniser pads with s_nop (47 per
                // sample), inside one block the two slots are interleaved by hand instead.
                if constexpr (NS == 1) {
                    // ONE stage slot: the block of tap j is  rem' = d_j - a*rem ; y += c_(j-1) * d'_(j-1) ; d'_j = rem + a*rem'
                    // -- the dot-product term of the tap BEFORE stands between the two dependent FMAs of this one
                    // (a lone chain: nothing else in the lane to put there), and the last tap's term follows the loop
                    double rn0;
                    if (j <= 2) {
                        asm volatile("v_fma_f64 %[rn0], %[na], %[r0], %[d0]\n\t"
                            "v_fma_f64 %[d0], %[a], %[rn0], %[r0]"
                            : [rn0] "=&v"(rn0), [d0] "+v"(d[0][j])
                            : [na] "s"(na), [a] "s"(a), [r0] "v"(r[0]));
                    } else if (j == 3) {
                        asm volatile("v_fma_f64 %[rn0], %[na], %[r0], %[d0]\n\t"
                            "v_mul_f64 %[y0], %[c], %[dp]\n\t"
                            "v_fma_f64 %[d0], %[a], %[rn0], %[r0]"
                            : [rn0] "=&v"(rn0), [d0] "+v"(d[0][j]), [y0] "=&v"(y[0])
                            : [na] "s"(na), [a] "s"(a), [r0] "v"(r[0]), [c] "v"(cv[j - 1]), [dp] "v"(d[0][j - 1]));
                    } else {
                        asm volatile("v_fma_f64 %[rn0], %[na], %[r0], %[d0]\n\t"
                            "v_fma_f64 %[y0], %[c], %[dp], %[y0]\n\t"
                            "v_fma_f64 %[d0], %[a], %[rn0], %[r0]"
                            : [rn0] "=&v"(rn0), [d0] "+v"(d[0][j]), [y0] "+v"(y[0])
                            : [na] "s"(na), [a] "s"(a), [r0] "v"(r[0]), [c] "v"(cv[j - 1]), [dp] "v"(d[0][j - 1]));
                    }
                    r[0] = rn0;
                    if (j == M)
                        y[0] = fma(cj, d[0][M], y[0]);
                } else {
                    double rn0, rn1;
                    if (j == 2) {
                        // the first term of the dot products: a product, not an FMA onto a zero that a
                        // v_mov would have to make first
                        asm volatile("v_fma_f64 %[rn0], %[na], %[r0], %[d0]\n\t"
                            "v_fma_f64 %[rn1], %[na], %[r1], %[d1]\n\t"
                            "v_fma_f64 %[d0], %[a], %[rn0], %[r0]\n\t"
                            "v_fma_f64 %[d1], %[a], %[rn1], %[r1]\n\t"
                            "v_mul_f64 %[y0], %[c], %[d0]\n\t"
                            "v_mul_f64 %[y1], %[c], %[d1]"
                            : [rn0] "=&v"(rn0), [rn1] "=&v"(rn1), [d0] "+v"(d[0][j]), [d1] "+v"(d[1][j]),
                              [y0] "=&v"(y[0]), [y1] "=&v"(y[1])
                            : [na] "s"(na), [a] "s"(a), [r0] "v"(r[0]), [r1] "v"(r[1]), [c] "v"(cj));
                    } else if (j > 2) {
                        // (the interpolation c_j = c0 + i*cinc stays outside: inside the block it
                        // doubles the padding at the asm boundaries, measured +1 ms)
                        asm volatile("v_fma_f64 %[rn0], %[na], %[r0], %[d0]\n\t"
                            "v_fma_f64 %[rn1], %[na], %[r1], %[d1]\n\t"
                            "v_fma_f64 %[d0], %[a], %[rn0], %[r0]\n\t"
                            "v_fma_f64 %[d1], %[a], %[rn1], %[r1]\n\t"
                            "v_fma_f64 %[y0], %[c], %[d0], %[y0]\n\t"
                            "v_fma_f64 %[y1], %[c], %[d1], %[y1]"
                            : [rn0] "=&v"(rn0), [rn1] "=&v"(rn1), [d0] "+v"(d[0][j]), [d1] "+v"(d[1][j]),
                              [y0] "+v"(y[0]), [y1] "+v"(y[1])
                            : [na] "s"(na), [a] "s"(a), [r0] "v"(r[0]), [r1] "v"(r[1]), [c] "v"(cj));
                    } else {
                        asm volatile("v_fma_f64 %[rn0], %[na], %[r0], %[d0]\n\t"
                            "v_fma_f64 %[rn1], %[na], %[r1], %[d1]\n\t"
                            "v_fma_f64 %[d0], %[a], %[rn0], %[r0]\n\t"
                            "v_fma_f64 %[d1], %[a], %[rn1], %[r1]"
                            : [rn0] "=&v"(rn0), [rn1] "=&v"(rn1), [d0] "+v"(d[0][j]), [d1] "+v"(d[1][j])
                            : [na] "s"(na), [a] "s"(a), [r0] "v"(r[0]), [r1] "v"(r[1]));
                    }
                    r[0] = rn0;
                    r[1] = rn1;
                }
            }
#undef JB_LDS_RD
            // ---- Pade combine (mlsa.rs:71-78): partial sums per lane, gathered on the lead lane ----
            double ssum, psum;
            if constexpr (NS == 2) {
                const double v1 = w1 * y[1];
                const double sb = fma(w0, y[0], v1), sa = fma(w0, y[0], db1 - v1); // (df1's plain sum rides on df2's alternating one)
                // fold position 2 into 1, then 1 into 0: sum(pos 0) = s0 + (s1 + s2)
                ssum = sa + dpp_f64<DPP_WAVE_SHL1>(sa + dpp_f64<DPP_WAVE_SHL1>(sa));
                psum = sb + dpp_f64<DPP_WAVE_SHL1>(sb + dpp_f64<DPP_WAVE_SHL1>(sb));
            } else {
                ssum = fold5(fma(wa, y[0], db1));
                psum = fold5(w0 * y[0]);
            }
            const double yprev = dpp_f64<DPP_WAVE_SHR1>(y[NS - 1]); // previous lane's last stage
            const double xmid = x + ssum; // d22[0] (valid on the lead lane)
            x = xmid + psum;
            // next-sample slot inputs: stage s+1 <- y of stage s; stage 0 <- xmid
            if constexpr (NS == 2)
                u[1] = y[0];
            u[0] = pos == 0 ? xmid : yprev;
            const double pv = x * vol;
            if (ODD)
                oB = pv;
            else
                oA = pv;
        };
        // (this form needs an even frame period: checked on the host, which otherwise builds chunks for
        // the wave kernel)
        int xnext = xbs, xqn = 1; // first sample and number of the next block
        for (int i2 = 0; i2 < fp; i2 += 2) {
            // the requests of this pair of samples are for samples i2 + 2, i2 + 3: on to the next block's source
            // (blocks of an even number of samples; with an odd one the pulse pass writes whole rows and the
            // pointer of block 0 serves the frame)
            if (i2 + 2 == xnext && xnext < fp) {
                xq = xblock(xqn);
                xqn++;
                xnext += xbs;
            }
            if (kLtWaves == 8 && i2 % kLtBalance == 0) {
                // keep level with the wave that shares this SIMD (w ^ 4): post the sample count, read the partner's
                // (0xffffffff once it has left), and take the higher issue priority if behind.  The LDS queue is
                // drained before the tap loop's counted waits start (they are valid among loads only).
                const uint32_t mine = tl * (uint32_t)fp + (uint32_t)i2;
                uint32_t theirs;
                asm volatile("ds_write_b32 %1, %2\n\t"
                             "ds_read_b32 %0, %3\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=v"(theirs)
                             : "v"((uint32_t)(uintptr_t)&prog[wv]), "v"(mine), "v"((uint32_t)(uintptr_t)&prog[wv ^ 4])
                             : "memory");
                if (mine > (uint32_t)__builtin_amdgcn_readfirstlane((int)theirs))
                    __builtin_amdgcn_s_setprio(0);
                else
                    __builtin_amdgcn_s_setprio(1);
            }
            sample(i2, std::false_type{});
            sample(i2 + 1, std::true_type{});
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xn), "+v"(xo)::"memory"); // the last samples' (unused) requests
        if (emit)
            put_pair(fp - 2);
        if (act && tl + 1 == nfr) {
            double *se_ = work[item].save_end;
            if (se_)
                save_state(se_);
        }
    }
    if (kLtWaves == 8 && lane == 0)
        prog[wv] = 0xffffffffu;
#ifdef JB_LT_STAMPS
    if (lane == 0 && stamp_w < kLtStampWaves)
        g_lt_stamp[stamp_w][1] = wall_clock64();
#endif
#ifdef JB_LT_CLOCKS
    if (clk_slot >= 0 && lane == 0) {
        g_lt_clk[clk_slot + 2] = clock64();
        g_lt_clk[clk_slot + 3] = wall_clock64();
    }
#endif
}

// --------------------------------------------------------------------------
// Certification of time-chunked execution: a chunk that started from zero state
// W frames early must have reached the same filter state as its predecessor's end
// state.  One wave per item; the excitation ring is feed-forward and not compared.
//
// The two dumps of a comparison may come from different kernels (first pass: throughput kernel;
// redo: wave kernel).  They agree on everything that is carried state; the wave kernel also
// leaves values in slots that are not: the u row holds per-lane temporaries except at the head
// lane of each 12-lane stage segment, and the tap slots past the last tap (12 groups x TPL
// slots for ntaps taps) and of lanes 60..63 carry the remainder through.  Those are skipped in
// EVERY comparison (a redo overwrites an end-state dump of the first pass in place, and the
// dump is compared again on the next run of the same batch).
__device__ __forceinline__ bool voc_state_differs(const double *__restrict__ a, const double *__restrict__ r,
                                                  int nfilt, int ntaps, double tol, int lane, double *md_out,
                                                  double *mr_out)
{
    double md = 0.0, mr = 0.0;
    const int urow = nfilt - 76; // 64*TPL: start of the u row (layout: d[TPL][64] | u[64] | e11[6] | e12[6])
    const int tpl = urow / 64;
    for (int k = lane; k < nfilt; k += 64) {
        if (ntaps < 0) {
            // Stage::NonZero dumps (d[stage][64]): every slot is carried state
        } else if (k < urow) {
            const int ln = k & 63, kk = k >> 6;
            if (ln >= kGroups * kPade || (ln % kGroups) * tpl + kk >= ntaps)
                continue;
        } else if (k < urow + 64 && ((k - urow) % kGroups != 0 || (k - urow) >= kGroups * kPade)) {
            continue;
        }
        const double x = a[k], y = r[k];
        md = fmax(md, fabs(x - y));
        mr = fmax(mr, fabs(y));
        if (!(x == x) || !(y == y))
            md = 1e300; // NaN anywhere => redo
    }
    for (int o = 32; o > 0; o >>= 1) {
        md = fmax(md, __shfl_xor(md, o));
        mr = fmax(mr, __shfl_xor(mr, o));
    }
    *md_out = md;
    *mr_out = mr;
    return md > tol * mr && md > 1e-300;
}

__global__ __launch_bounds__(64) void k_voc_verify(const VocWork *__restrict__ work, uint32_t n_items,
                                                    int nfilt /* doubles to compare */, int ntaps, double tol,
                                                    uint8_t *bad, uint32_t *n_bad)
{
    const uint32_t i = blockIdx.x;
    if (i >= n_items)
        return;
    const VocWork wk = work[i];
    const int lane = threadIdx.x;
    if (!wk.save_warm || i == 0) {
        if (lane == 0)
            bad[i] = 0;
        return;
    }
    double md, mr;
    const bool isbad = voc_state_differs(wk.save_warm, work[i - 1].save_end, nfilt, ntaps, tol, lane, &md, &mr);
    if (lane == 0) {
        bad[i] = isbad;
        if (isbad)
            atomicAdd(n_bad, 1u);
#ifdef JB_VERIFY_DEBUG
        if (isbad)
            printf("handoff item %u utt %u t_out %u: max|diff| %.3e max|state| %.3e ratio %.3e\n", i, wk.utt,
                   wk.t_out, md, mr, md / mr);
#endif
    }
}

// same comparison for explicit pairs of states (partial redo: recomputed state vs checkpoint;
// re-certification of a successor against the exact end state of a fully redone chunk)
__global__ __launch_bounds__(64) void k_voc_verify_pairs(const double *const *__restrict__ pairs, uint32_t n_pairs,
                                                          int nfilt, int ntaps, double tol, uint8_t *bad,
                                                          uint32_t *n_bad)
{
    const uint32_t i = blockIdx.x;
    if (i >= n_pairs)
        return;
    const int lane = threadIdx.x;
    double md, mr;
    const bool isbad = voc_state_differs(pairs[2 * i], pairs[2 * i + 1], nfilt, ntaps, tol, lane, &md, &mr);
    if (lane == 0) {
        bad[i] = isbad;
        if (isbad)
            atomicAdd(n_bad, 1u);
#ifdef JB_VERIFY_DEBUG
        printf("state pair %u: max|diff| %.3e max|state| %.3e %s\n", i, md, mr, isbad ? "FAIL" : "ok");
#endif
    }
}

static int tpl_for(int nmcp)
{
    int M = nmcp - 1;
    int tpl = (M + kGroups - 1) / kGroups;
    return tpl < 1 ? 1 : tpl;
}

int vocoder_state_doubles(int nmcp) { return voc_state_doubles(tpl_for(nmcp)); }

hipError_t launch_pitch(const BatchDev &bd, const VocDev &vd, hipStream_t stream)
{
    if (bd.B == 0 || bd.maxT == 0)
        return hipSuccess;
    dim3 grid((bd.maxT + 255) / 256, bd.B), block(256);
    hipLaunchKernelGGL(k_pitch, grid, block, 0, stream, bd, vd);
    return hipGetLastError();
}

hipError_t launch_mc2b(const BatchDev &bd, const VocDev &vd, hipStream_t stream)
{
    if (bd.B == 0 || bd.maxT == 0)
        return hipSuccess;
    dim3 grid((bd.maxT + kMc2bFrames - 1) / kMc2bFrames, bd.B), block(kMc2bFrames);
    hipLaunchKernelGGL(k_mc2b, grid, block, 0, stream, bd, vd);
    return hipGetLastError();
}

// CUs of the calling thread's current device, asked once per process and device (0: unknown)
static int current_device_cus()
{
    static int cus_of[64] = {0};
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return 0;
    if ((cus = cus_of[dev]) == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return 0;
        cus_of[dev] = cus;
    }
    return cus;
}

// the split form (k_exc_* + k_excite_fix) is built for these shapes; everything else takes k_excite
bool excite_is_split(const VocDev &vd)
{
    // (fperiod = 4m with m <= 64: one of m, 2m, 4m is the block size, so a frame has at most four mask words --
    // what k_excite_fix holds per frame; checked all the same)
    return vd.fperiod % kExw == 0 && vd.fperiod <= 256 && vd.fperiod >= kExwHalo && (vd.nlpf == 31 || vd.nlpf == 15) &&
           vd.nblk <= 4 && vd.fperiod % vd.bs == 0;
}

hipError_t launch_excite_noise(const BatchDev &bd, const VocDev &vd, hipStream_t stream)
{
    if (bd.B == 0 || bd.maxT == 0 || !excite_is_split(vd))
        return hipSuccess;
    if (!vd.exc_src || !vd.exc_gen || !vd.exc_gen_count || !vd.exc_tab)
        return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(vd.exc_gen_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess)
        return e;
    const dim3 block(256);
    const dim3 gcls((bd.maxT + 255) / 256, bd.B), gtab((bd.maxT + 3) / 4);
    // The per-frame pass runs on a side stream beside the MCP chain's build and band solve, which are chains
    // of round trips at low occupancy: 32 KB of LDS that the kernel does not use ride with every workgroup and
    // cap the workgroups a CU takes at three (12 waves instead of up to 40) -- with all the wave slots it could
    // get it stretched them by more than it gained (measured in round 2 when EVERY frame went through this
    // pass; it still does for a voice whose LPF taps differ from frame to frame).  Fixed grid: three
    // workgroups per CU of the device, each wave taking list entries in turn.
    constexpr size_t pad = 32 * 1024;
    int cus = current_device_cus();
    if (cus <= 0)
        cus = 256;
    const dim3 ggen((unsigned)(3 * cus));
    if (vd.nlpf == 31) {
        hipLaunchKernelGGL(k_exc_classify<31>, gcls, block, 0, stream, bd, vd);
        if (!vd.exc_no_table)
            hipLaunchKernelGGL(k_exc_table<31>, gtab, block, 0, stream, bd, vd);
        JB_DBG_SKIP_IF(4, hipLaunchKernelGGL(k_exc_general<31>, ggen, block, pad, stream, bd, vd));
    } else {
        hipLaunchKernelGGL(k_exc_classify<15>, gcls, block, 0, stream, bd, vd);
        if (!vd.exc_no_table)
            hipLaunchKernelGGL(k_exc_table<15>, gtab, block, 0, stream, bd, vd);
        hipLaunchKernelGGL(k_exc_general<15>, ggen, block, pad, stream, bd, vd);
    }
    return hipGetLastError();
}

hipError_t launch_excite(const BatchDev &bd, const VocDev &vd, hipStream_t stream)
{
    if (bd.B == 0 || bd.maxT == 0)
        return hipSuccess;
    if (excite_is_split(vd)) {
        dim3 gfix((bd.maxT + 4 * kFixFrames - 1) / (4 * kFixFrames), bd.B), block(256);
        if (vd.nlpf == 31)
            JB_DBG_SKIP_IF(8, hipLaunchKernelGGL(k_excite_fix<31>, gfix, block, 0, stream, bd, vd));
        else
            hipLaunchKernelGGL(k_excite_fix<15>, gfix, block, 0, stream, bd, vd);
        return hipGetLastError();
    }
    const uint64_t maxN = (uint64_t)bd.maxT * (uint64_t)vd.fperiod;
    dim3 grid((unsigned)((maxN + kExcBlock - 1) / kExcBlock), bd.B), block(kExcBlock);
    if (vd.nlpf == 0) {
        if (!vd.uv_before)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_uv_scan, dim3(bd.B), dim3(64), 0, stream, bd, vd);
        hipLaunchKernelGGL(k_excite_nolpf, grid, block, 0, stream, bd, vd);
        return hipGetLastError();
    }
    if (excite_fits_staged(vd))
        hipLaunchKernelGGL(k_excite, grid, block, 0, stream, bd, vd);
    else
        hipLaunchKernelGGL(k_excite_any, grid, block, (size_t)(kExcBlock + vd.nlpf - 1) * sizeof(double), stream, bd, vd);
    return hipGetLastError();
}

// 512 waves take the voiced runs from an atomic work queue, utterances longest first
constexpr unsigned kPulseQueueWaves = 512;
hipError_t launch_pulse(const BatchDev &bd, const VocDev &vd, hipStream_t stream)
{
    if (bd.B == 0 || bd.maxT == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_run_scan, dim3(1), dim3(256), 0, stream, bd, vd);
    JB_DBG_SKIP_IF(64, hipLaunchKernelGGL(k_pulse_queue, dim3(kPulseQueueWaves), dim3(64), 0, stream, bd, vd));
    return hipGetLastError();
}

hipError_t launch_voc_verify(const VocWork *work_dev, uint32_t n_items, int state_doubles, int ntaps, double tol,
                             uint8_t *bad, uint32_t *n_bad, hipStream_t stream)
{
    if (n_items == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_voc_verify, dim3(n_items), dim3(64), 0, stream, work_dev, n_items,
                       ntaps < 0 ? state_doubles : state_doubles - 4, ntaps, tol, bad, n_bad);
    return hipGetLastError();
}

hipError_t launch_voc_verify_pairs(const double *const *pairs_dev, uint32_t n_pairs, int state_doubles, int ntaps,
                                   double tol, uint8_t *bad, uint32_t *n_bad, hipStream_t stream)
{
    if (n_pairs == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_voc_verify_pairs, dim3(n_pairs), dim3(64), 0, stream, pairs_dev, n_pairs,
                       ntaps < 0 ? state_doubles : state_doubles - 4, ntaps, tol, bad, n_bad);
    return hipGetLastError();
}

// The code an order runs on: its own for nitech's two (35, 25: the EXACT instantiations), else the next of
// {25, 31, 35} as lane triples or of {41, 51, 61} with one stage per lane, the taps above the voice's own at coefficient
// zero (an order-39 voice pays for 40 taps, an order-49 voice for 50).  Orders below 6 stay with the wave kernels.
static int lt_code_nm(int nmcp)
{
    for (int c : {25, 31, 35, 41, 51, 61})
        if (nmcp <= c)
            return c;
    return 0;
}
bool vocoder_ls_supported(int nmcp) { return nmcp >= 7 && lt_code_nm(nmcp) != 0; }

int vocoder_ls_chunks_per_wave(int nmcp) { return lt_chunks(lt_code_nm(nmcp) <= 35 ? 3 : 5); }

hipError_t launch_vocoder_ls(const BatchDev &bd, const VocDev &vd, const VocWork *work_dev,
                             const uint32_t *order_dev, uint32_t n_items, int waves_per_simd, hipStream_t stream)
{
    if (n_items == 0)
        return hipSuccess;
    if ((waves_per_simd != 1 && waves_per_simd != 2) || !vocoder_ls_supported(vd.nmcp))
        return hipErrorInvalidValue;
    const int wv = 4 * waves_per_simd;
    const int code = lt_code_nm(vd.nmcp);
    const bool exact = vd.nmcp == 35 || vd.nmcp == 25;
    const uint32_t per_wg = (uint32_t)(vocoder_ls_chunks_per_wave(vd.nmcp) * wv);
    dim3 grid((n_items + per_wg - 1) / per_wg), block(64 * wv);
#define JB_LT_LAUNCH(NMc, TPLc, EX, LPCc)                                                                              \
    do {                                                                                                           \
        if (wv == 8)                                                                                               \
            hipLaunchKernelGGL((k_vocoder_lt<NMc, TPLc, 8, EX, LPCc>), grid, block, 0, stream, bd, vd, work_dev,      \
                               order_dev, n_items);                                                                \
        else                                                                                                       \
            hipLaunchKernelGGL((k_vocoder_lt<NMc, TPLc, 4, EX, LPCc>), grid, block, 0, stream, bd, vd, work_dev,      \
                               order_dev, n_items);                                                                \
    } while (0)
    if (exact && code == 35)
        JB_LT_LAUNCH(35, 3, true, 3);
    else if (exact && code == 25)
        JB_LT_LAUNCH(25, 2, true, 3);
    else if (code == 25)
        JB_LT_LAUNCH(25, 2, false, 3);
    else if (code == 31)
        JB_LT_LAUNCH(31, 3, false, 3);
    else if (code == 35)
        JB_LT_LAUNCH(35, 3, false, 3);
    else if (code == 41)
        JB_LT_LAUNCH(41, 4, false, 5);
    else if (code == 51)
        JB_LT_LAUNCH(51, 5, false, 5);
    else
        JB_LT_LAUNCH(61, 5, false, 5);
#undef JB_LT_LAUNCH
#ifdef JB_LT_STAMPS
    if (const char *path = getenv("JB_LT_STAMPS_FILE")) { // the LAST launch of the process is what the file holds
        static unsigned long long st[kLtStampWaves][4];
        hipStreamSynchronize(stream);
        const unsigned nw = std::min<unsigned>(kLtStampWaves, grid.x * (unsigned)wv);
        if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_lt_stamp), sizeof st) == hipSuccess)
            if (FILE *f = fopen(path, "w")) {
                fprintf(f, "# wave start_10ns end_10ns hw_id xcc frames   (k_vocoder_lt, %u waves, %d per workgroup)\n", nw, wv);
                for (unsigned w = 0; w < nw; w++)
                    fprintf(f, "%u %llu %llu %llu %llu %llu\n", w, st[w][0], st[w][1], st[w][2] & 0xffffffffull,
                            st[w][2] >> 32, st[w][3]);
                fclose(f);
            }
        // per launch: when the last wave of every XCD ended, ms behind the kernel's first wave start
        unsigned long long t0 = ~0ull, last[16] = {0};
        for (unsigned w = 0; w < nw; w++)
            if (st[w][0] && st[w][0] < t0)
                t0 = st[w][0];
        for (unsigned w = 0; w < nw; w++)
            last[(st[w][2] >> 32) & 15] = std::max(last[(st[w][2] >> 32) & 15], st[w][1]);
        fprintf(stderr, "k_vocoder_lt last wave end per XCD, ms:");
        for (int x = 0; x < 8; x++)
            fprintf(stderr, " %.2f", last[x] ? (double)(last[x] - t0) / 1e5 : 0.0);
        fprintf(stderr, "\n");
    }
#endif
#ifdef JB_LT_CLOCKS
    {
        unsigned long long c[8];
        hipStreamSynchronize(stream);
        if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_lt_clk), sizeof c) == hipSuccess)
            for (int k = 0; k < 8; k += 4) {
                const double us = (double)(c[k + 3] - c[k + 1]) / 100.0; // 100 MHz
                fprintf(stderr, "k_vocoder_lt wave %d: %.3f ms, %.0f shader-clock ticks = %.3f GHz\n", k / 4, us / 1e3,
                        (double)(c[k + 2] - c[k]), (double)(c[k + 2] - c[k]) / us / 1e3);
            }
    }
#endif
    return hipGetLastError();
}

hipError_t launch_vocoder(const BatchDev &bd, const VocDev &vd, const VocWork *work_dev, uint32_t n_items,
                          hipStream_t stream)
{
    if (n_items == 0)
        return hipSuccess;
    if (vd.stage > 0) // Stage::NonZero: the MGLSA cascade (jb_mglsa.hip)
        return launch_vocoder_mglsa(bd, vd, work_dev, n_items, stream);
    dim3 grid((n_items + 3) / 4), block(256);
    // two waves per item while that still leaves every wave a SIMD of its own (k_vocoder_pair)
    const int cus = current_device_cus();
    const char *nop = getenv("JB_NO_PAIR_KERNEL"); // (unset: both forms; "1": neither; "8": not the eight-wave form)
    if (cus > 0 && n_items <= 2u * (uint32_t)cus && !(nop && nop[0] == '1')) {
        const dim3 gp((n_items + 1) / 2);
        switch (tpl_for(vd.nmcp)) {
#define JB_PAIR_CASE(T)                                                                                            \
    case T:                                                                                                        \
        hipLaunchKernelGGL((k_vocoder_pair<T, 2>), gp, dim3(256), 0, stream, bd, vd, work_dev, n_items);            \
        break;
            JB_PAIR_CASE(1)
            JB_PAIR_CASE(2)
            JB_PAIR_CASE(3)
            JB_PAIR_CASE(4)
            JB_PAIR_CASE(5)
            JB_PAIR_CASE(6)
#undef JB_PAIR_CASE
        default:
            return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (cus > 0 && n_items <= 4u * (uint32_t)cus && !nop) {
        const dim3 gp((n_items + 3) / 4);
        switch (tpl_for(vd.nmcp)) {
#define JB_PAIR_CASE(T)                                                                                            \
    case T:                                                                                                        \
        hipLaunchKernelGGL((k_vocoder_pair<T, 4>), gp, dim3(512), 0, stream, bd, vd, work_dev, n_items);            \
        break;
            JB_PAIR_CASE(1)
            JB_PAIR_CASE(2)
            JB_PAIR_CASE(3)
            JB_PAIR_CASE(4)
            JB_PAIR_CASE(5)
            JB_PAIR_CASE(6)
#undef JB_PAIR_CASE
        default:
            return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (tpl_for(vd.nmcp)) {
    case 1:
        hipLaunchKernelGGL(k_vocoder<1>, grid, block, 0, stream, bd, vd, work_dev, n_items);
        break;
    case 2:
        hipLaunchKernelGGL(k_vocoder<2>, grid, block, 0, stream, bd, vd, work_dev, n_items);
        break;
    case 3:
        hipLaunchKernelGGL(k_vocoder<3>, grid, block, 0, stream, bd, vd, work_dev, n_items);
        break;
    case 4:
        hipLaunchKernelGGL(k_vocoder<4>, grid, block, 0, stream, bd, vd, work_dev, n_items);
        break;
    case 5:
        hipLaunchKernelGGL(k_vocoder<5>, grid, block, 0, stream, bd, vd, work_dev, n_items);
        break;
    case 6: // (orders 61..63: round 6)
        hipLaunchKernelGGL(k_vocoder<6>, grid, block, 0, stream, bd, vd, work_dev, n_items);
        break;
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace jb
