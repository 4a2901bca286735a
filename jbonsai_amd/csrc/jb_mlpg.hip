// jb_mlpg.hip -- parameter generation (MLPG + GV) kernels for gfx950.
//
// Restates, per (utterance, vector dimension), /root/reference/src/mlpg_adjust:
//   k_prep_*          A1/A2  Mask::create + boundary_distances      (mask.rs:20-82)
//   k_mlpg_ivar       A3     MeanVari::with_ivar per state          (model/mean_vari.rs:21-31)
//   k_mlpg_build*     A3/A4  parameter expansion + calc_wuw_and_wum (mod.rs:56-84, mlpg.rs:25-70)
//                            -- elementwise in time: thread per (frame, dim)
//   k_mlpg_fb_lds     A5/A6  band LDL^T + substitutions             (mlpg.rs:79-115)
//                            -- serial in time: one solver wave per utterance (lane = dim) fed
//                            through LDS by mover waves
//   k_mlpg_gv_tp      A8     GV ascent, time-parallel (MCP)         (mlpg.rs:145-292)
//   k_mlpg_gv_vt      A8     GV ascent, lanes over time with the sums in serial order (LF0)
//   k_mlpg_solve3/solve A5-A9 fused serial-order sweeps / generic band width (bit-exact A/B paths)
//   k_mlpg_static     A3-A9  one static window, no GV (LPF)
//   k_mlpg_scatter*, k_mc2b_mt A9 Mask::fill with NODATA (+ mc2b for MCP)  (mask.rs:34-49)
// Workspace layout: [dim][frame] per utterance for band width 3 (`StreamDev::mt`), [frame][dim]
// for the generic solver.  All arithmetic is f64 with FP contraction off, in the reference's
// order of operations, so every path is bit-exact against the oracle except the three GV
// reductions of k_mlpg_gv_tp (fixed-shape tree sums, ~1e-15 relative; JB_BATCH_SERIAL_GV keeps the
// serial order).  LF0 needs the reference order: pulse positions are chaotic w.r.t. the rounding
// of lf0.
//
// Compiled with -ffp-contract=off.
#include "jb_device.h"

#include <cstdlib>
#include <type_traits>

namespace jb {

// --------------------------------------------------------------------------
// A1/A2 in two steps.  The MSD flag is constant within a state, so run boundaries,
// compaction offsets and boundary distances are decided per STATE, then expanded per FRAME in parallel.
// One wave per utterance, 64 states per step.  The walk of mask.rs:20-82 is serial as written -- start
// frame, compaction offset, start and end of the voiced run a state lies in, where a state of duration 0
// is invisible to its neighbours -- but every one of its values is a prefix sum or "the value at the nearest
// lane with a property": wave scans, ballots and one bpermute per value, ~25 us for a 128 s utterance (7.5 k
// states) where the state-by-state walk on readlane scalars took 1.4-2 ms at the head of the LF0 chain.
__device__ __forceinline__ uint32_t ps_scan_incl(uint32_t v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)v, o);
        if (lane >= o)
            v += u;
    }
    return v;
}

__global__ __launch_bounds__(64) void k_prep_states(BatchDev bd, StreamDev sd, int si)
{
    if ((int)blockIdx.x >= bd.B)
        return;
    const int b = (int)bd.order[blockIdx.x]; // longest utterance first
    const int lane = threadIdx.x;
    const UttDev *up = bd.utt + b;
    const StreamStatesDev st = up->st[si];
    const uint32_t S = up->S;
    const uint32_t *dur = up->dur;
    const uint64_t sb = up->state_off;
    const unsigned long long below = (1ull << lane) - 1ull;           // lanes < lane
    const unsigned long long upto = below | (1ull << lane);           // lanes <= lane
    uint32_t t = 0, k = 0, gl = 0, run_start = 0, nrun = 0;
    bool prev_v = false;
    // forward: state start frame, compaction offset, start of the voiced run
    for (uint32_t s0 = 0; s0 < S; s0 += 64) {
        const uint32_t s = s0 + (uint32_t)lane;
        const bool ok = s < S;
        const uint32_t d = ok ? dur[s] : 0u;
        // msd.unwrap_or(f64::MAX) > threshold  (model/mod.rs:113, mask.rs:24)
        const double msd = (ok && st.msd) ? st.msd[s] : 1.7976931348623157e308;
        const bool v = ok && msd > st.msd_threshold;
        const bool g = v && st.gv_switch && st.gv_switch[s];
        const uint32_t dv = v ? d : 0u;
        const uint32_t it = ps_scan_incl(d, lane), ik = ps_scan_incl(dv, lane);
        uint32_t gs = g ? d : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            gs += (uint32_t)__shfl_xor((int)gs, o);
        const uint32_t my_start = t + it - d, my_vpre = k + ik - dv;
        // voicing of the nearest earlier state that has frames (a state without frames leaves prev_v alone)
        const unsigned long long dm = __ballot(d > 0), vm = __ballot(v);
        const unsigned long long lo = dm & below;
        const bool pv = lo ? ((vm >> (63 - __clzll((long long)lo))) & 1ull) != 0 : prev_v;
        const bool isstart = v && !pv;
        const unsigned long long sm = __ballot(isstart);
        // compact list of run starts for the pulse scheduler (a zero-length voiced state may repeat an
        // entry; k_pulse_queue tolerates duplicates)
        if (isstart)
            sd.run_list[sb + nrun + (uint32_t)__popcll(sm & below)] = my_start;
        // start of the run the state lies in: the nearest run start at or before it
        const unsigned long long sl = sm & upto;
        const int src = sl ? 63 - __clzll((long long)sl) : 0;
        const uint32_t from = (uint32_t)__shfl((int)my_start, src);
        const uint32_t my_rstart = sl ? from : run_start;
        if (ok) {
            sd.s_start[sb + s] = my_start;
            sd.s_vpre[sb + s] = my_vpre;
            sd.s_rstart[sb + s] = my_rstart;
            sd.s_voiced[sb + s] = v;
        }
        t += (uint32_t)__shfl((int)it, 63);
        k += (uint32_t)__shfl((int)ik, 63);
        gl += gs;
        nrun += (uint32_t)__popcll(sm);
        run_start = (uint32_t)__shfl((int)my_rstart, 63);
        if (dm)
            prev_v = ((vm >> (63 - __clzll((long long)dm))) & 1ull) != 0;
    }
    // backward: last frame of the voiced run (mask.rs:66-79) -- the nearest run end at or after the state
    uint32_t run_end = 0;
    bool next_v = false;
    const uint32_t nch = (S + 63) / 64;
    for (uint32_t c = nch; c-- > 0;) {
        const uint32_t s0 = c * 64;
        const uint32_t s = s0 + (uint32_t)lane;
        const bool ok = s < S;
        const uint32_t d = ok ? dur[s] : 0u;
        const bool v = ok && sd.s_voiced[sb + s];
        const uint32_t my_start = ok ? sd.s_start[sb + s] : 0u;
        const unsigned long long dm = __ballot(d > 0), vm = __ballot(v);
        const unsigned long long hi = dm & ~upto; // lanes > lane
        const bool nv = hi ? ((vm >> __builtin_ctzll(hi)) & 1ull) != 0 : next_v;
        const bool isend = v && !nv;
        const unsigned long long em = __ballot(isend);
        const uint32_t tend = my_start + d - 1u; // (wraps for a state without frames at frame 0, as the walk's does)
        const unsigned long long eh = em & ~below; // lanes >= lane
        const int src = eh ? __builtin_ctzll(eh) : 0;
        const uint32_t from = (uint32_t)__shfl((int)tend, src);
        const uint32_t my_rend = eh ? from : run_end;
        if (ok)
            sd.s_rend[sb + s] = my_rend;
        run_end = (uint32_t)__shfl((int)my_rend, 0);
        if (dm)
            next_v = ((vm >> __builtin_ctzll(dm)) & 1ull) != 0;
    }
    if (lane == 0) {
        sd.Tv[b] = k;
        sd.gvlen[b] = gl;
        sd.nruns[b] = nrun;
    }
}

// Non-MSD streams (msd = f64::MAX, model/mod.rs:113): every state is voiced, so the walk
// degenerates to an exclusive prefix sum of the durations (wave-level scan, 64 states per step)
// -- 20 us instead of 2 ms at the head of the MCP chain.
__global__ __launch_bounds__(64) void k_prep_states_dense(BatchDev bd, StreamDev sd, int si)
{
    const int b = blockIdx.x;
    if (b >= bd.B)
        return;
    const int lane = threadIdx.x;
    const UttDev *up = bd.utt + b;
    const uint32_t S = up->S, T = up->T;
    const uint32_t *dur = up->dur;
    const uint8_t *gsw = up->st[si].gv_switch;
    const uint64_t sb = up->state_off;
    uint32_t t = 0, gl = 0;
    for (uint32_t s0 = 0; s0 < S; s0 += 64) {
        const uint32_t s = s0 + (uint32_t)lane;
        const bool ok = s < S;
        const uint32_t d = ok ? dur[s] : 0u;
        uint32_t incl = d, g = (ok && gsw && gsw[s]) ? d : 0u;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
            if (lane >= o)
                incl += v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            g += (uint32_t)__shfl_xor((int)g, o);
        if (ok) {
            const uint32_t st = t + incl - d;
            sd.s_start[sb + s] = st;
            sd.s_vpre[sb + s] = st;
            sd.s_rstart[sb + s] = 0;
            sd.s_rend[sb + s] = T ? T - 1 : 0;
            sd.s_voiced[sb + s] = 1;
        }
        t += (uint32_t)__shfl((int)incl, 63);
        gl += g;
    }
    if (lane == 0) {
        sd.Tv[b] = t;
        sd.gvlen[b] = gl;
        sd.nruns[b] = t ? 1u : 0u;
        if (S)
            sd.run_list[sb] = 0;
    }
}

// thread per state: expand to frames
__global__ void k_prep_frames(BatchDev bd, StreamDev sd, int si)
{
    const int b = blockIdx.y;
    const UttDev *up = bd.utt + b;
    struct { uint32_t S, T; uint64_t frame_off, state_off; const uint32_t *dur; } u = {
        up->S, up->T, up->frame_off, up->state_off, up->dur};
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= u.S)
        return;
    const uint8_t *gv_switch = up->st[si].gv_switch;
    const uint64_t sb = u.state_off, base = u.frame_off;
    const uint32_t t0 = sd.s_start[sb + s], d = u.dur[s];
    const bool v = sd.s_voiced[sb + s];
    const uint32_t k0 = sd.s_vpre[sb + s], rs = sd.s_rstart[sb + s], re = sd.s_rend[sb + s];
    const uint8_t sw = (gv_switch && v) ? gv_switch[s] : 0;
    for (uint32_t i = 0; i < d; i++) {
        const uint32_t t = t0 + i;
        sd.fstate[base + t] = s;
        sd.voiced[base + t] = v;
        if (v) {
            const uint32_t dl = t - rs, dr = re - t;
            sd.fl[base + t] = dl > 255 ? 255 : (uint8_t)dl;
            sd.fr[base + t] = dr > 255 ? 255 : (uint8_t)dr;
            sd.vidx[base + k0 + i] = t;
            sd.vsw[base + k0 + i] = sw;
        } else {
            sd.fl[base + t] = 0;
            sd.fr[base + t] = 0;
        }
    }
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
// A3/A4 for one (compacted frame k, dim m): the reference's order of additions.
template <int BW>
__device__ __forceinline__ void build_elem(const StreamDev &sd, const StreamStatesDev &st, uint64_t base,
                                           uint32_t Tv, uint32_t k, int m, double (&wuw)[BW], double &wum)
{
    const int L = sd.L, W = sd.W;
#pragma unroll
    for (int j = 0; j < BW; j++)
        wuw[j] = 0.0;
    wum = 0.0;
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
}

// A3/A4: thread per (compacted frame k, dim m) of utterance blockIdx.y; [frame][dim] workspace.
template <int BW>
__global__ void k_mlpg_build(BatchDev bd, StreamDev sd, int si)
{
    const int b = blockIdx.y;
    const UttDev *up = bd.utt + b;
    const uint32_t Tv = sd.Tv[b];
    const int L = sd.L;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (uint64_t)Tv * L)
        return;
    const uint32_t k = (uint32_t)(tid / L);
    const int m = (int)(tid % L);
    const StreamStatesDev st = up->st[si];
    const uint64_t base = up->frame_off;
    double wuw[BW], wum;
    build_elem<BW>(sd, st, base, Tv, k, m, wuw, wum);
    const uint64_t o = (base + k) * (uint64_t)L + (uint64_t)m;
#pragma unroll
    for (int j = 0; j < BW; j++)
        sd.A[j][o] = wuw[j];
    sd.bvec[o] = wum;
}

// MeanVari::with_ivar per (state, window, dim), once per batch: a frame's inverse variance is
// its state's, and each is used by up to three neighbouring frames of every window, so the
// table replaces ~6 f64 divisions per (frame, dim) in the build by loads.
// kIvarPer elements per thread, block-strided (coalesced): with one element per thread the kernel is a stream of
// 790 k tiny workgroups whose DISPATCH, not whose traffic, sets its time once other queues have work too
// (0.5 ms alone, 2-3 ms at the head of the MCP chain beside the LF0 chain's kernels).
constexpr int kIvarPer = 8;
__global__ __launch_bounds__(256) void k_mlpg_ivar(BatchDev bd, StreamDev sd, int si)
{
    const int b = blockIdx.y;
    const UttDev *up = bd.utt + b;
    const uint32_t WL = (uint32_t)(sd.W * sd.L);
    const uint64_t n = (uint64_t)up->S * WL;
    const uint64_t e0 = (uint64_t)blockIdx.x * (256u * kIvarPer) + threadIdx.x;
    if (e0 >= n || !up->ivar_owner[si])
        return;
    const double *var = up->st[si].var;
    double *out = sd.ivar + up->ivar_state_off[si] * (uint64_t)WL;
    double v[kIvarPer];
#pragma unroll
    for (int k = 0; k < kIvarPer; k++) {
        const uint64_t e = e0 + (uint64_t)k * 256u;
        v[k] = var[e < n ? e : e0];
    }
#pragma unroll
    for (int k = 0; k < kIvarPer; k++) {
        const uint64_t e = e0 + (uint64_t)k * 256u;
        if (e < n)
            out[e] = with_ivar(v[k]);
    }
}

// A3/A4 for the [dim][frame] workspace: a block computes kBuildTF frames x L dims with the
// state tables read dim-fastest (coalesced), turns the tile in LDS and writes rows of
// kBuildTF contiguous frames per dim.  What depends on the frame alone (state index, MSD
// boundary distances of the frame and its +-1 neighbours) is looked up once per block.
// Same arithmetic, in the same order, as build_elem.
// 16 frames per block: 128-byte pieces per row reach the same HBM rate as longer ones (tools/microbench/piece_bw.hip)
// and twice as many workgroups fit a CU: 16 frames 3.6 ms, 32 4.5, 64 4.6, 8 4.6
constexpr int kBuildTF = 16;
#define JB_MAX_WIN_BUILD 3 // windows served by the sliding-window build (static, delta, acceleration)
constexpr int kMtMaxDim = 60; // (BW+1) * L * (kBuildTF+1) * 8 B <= 64 KiB of LDS for BW = 3
// Same kernel with the table loads of neighbouring frames SHARED.  k_mlpg_build_mt loads, for every
// (frame, dim), mean and 1/var of up to three source frames of every window -- 18 loads per element,
// 34 GB through the texture addressers for config 2, which is what its 4.8 ms are (10 % VALU busy,
// 61 % of the wave cycles parked on memory).  Here a thread owns kBuildRun CONSECUTIVE frames of one
// dim and slides a three-frame window over the table: 6 loads per element (2 for a one-tap window).
// The arithmetic per element is the reference's loop nest, unchanged and in the same order
// (test_fused_mlpg_equals_generic_bitwise holds it to the un-fused kernels bit for bit).
constexpr int kBuildRun = 4;
typedef const double __attribute__((address_space(1))) *Gp;
template <int BW>
__global__ void k_mlpg_build_mt2(BatchDev bd, StreamDev sd, int si)
{
    extern __shared__ double tile[]; // [BW+1][L][kBuildTF+1]
    constexpr int HW = BW / 2;       // frames a window can reach to either side
    static_assert(BW == 3 && HW == 1, "written for three-tap windows");
    __shared__ uint32_t f_state[kBuildTF + 2 * HW];
    __shared__ uint8_t f_l[kBuildTF + 2 * HW], f_r[kBuildTF + 2 * HW];
    const int b = blockIdx.y;
    const UttDev *up = bd.utt + b;
    const uint32_t Tv = sd.Tv[b];
    const uint32_t k0 = blockIdx.x * (uint32_t)kBuildTF;
    if (k0 >= Tv)
        return;
    const int L = sd.L, W = sd.W;
    const StreamStatesDev st = up->st[si];
    const uint64_t base = up->frame_off;
    if (threadIdx.x < kBuildTF + 2 * HW) {
        const long k = (long)k0 - HW + (long)threadIdx.x;
        uint32_t s = 0;
        uint8_t dl = 0, dr = 0;
        if (k >= 0 && k < (long)Tv) {
            // (a stream without MSD keeps every frame: one memory round trip less before the table loads)
            const uint32_t f = sd.is_msd ? sd.vidx[base + k] : (uint32_t)k;
            s = sd.fstate[base + f];
            dl = sd.fl[base + f];
            dr = sd.fr[base + f];
        }
        f_state[threadIdx.x] = s;
        f_l[threadIdx.x] = dl;
        f_r[threadIdx.x] = dr;
    }
    __syncthreads();
    const double *ivt = sd.ivar + up->ivar_state_off[si] * (uint64_t)(W * L);
    const double *mnt = st.mean;
    const int pitch = kBuildTF + 1, plane = L * pitch;
    const int WL = W * L;
    constexpr int G = kBuildTF / kBuildRun; // runs per block
    const int tid = threadIdx.x;
    // interior block: frames k0-1 .. k0+TF+1 all exist (sources k-1..k+1, band columns k..k+2)
    const bool interior = k0 >= 1 && (uint64_t)k0 + kBuildTF + 2 <= (uint64_t)Tv;
    if (tid < G * L) {
        const int g = tid / L, m = tid - g * L;
        const int kl0 = g * kBuildRun;
        // ALL table values of the run -- (mean, masked 1/var) of frames kl0-1 .. kl0+kBuildRun of every
        // window -- are requested before the first is used; loaded frame by frame as the window slides,
        // every frame of the run was its own memory round trip.
        double mvr[JB_MAX_WIN_BUILD][kBuildRun + 2], ivr[JB_MAX_WIN_BUILD][kBuildRun + 2];
        const Gp mng = (Gp)mnt, ivg = (Gp)ivt; // global, not generic: the tables come from hipMalloc
        auto fetch = [&](int w, int fi, double &mv, double &iv) {
            const int ww = sd.win_width[w];
            const int lw = ww / 2, rw = ww - lw - 1;
            const uint32_t pi = f_state[fi] * (uint32_t)WL + (uint32_t)(L * w + m);
            mv = mng[pi];
            double ivar = ivg[pi];
            // dynamic windows touching an MSD boundary get ivar = 0 (mod.rs:69-80)
            if (w != 0 && ((int)f_l[fi] < lw || (int)f_r[fi] < rw))
                ivar = 0.0;
            iv = ivar;
        };
#pragma unroll
        for (int w = 0; w < JB_MAX_WIN_BUILD; w++) {
#pragma unroll
            for (int q = 0; q < kBuildRun + 2; q++) {
                mvr[w][q] = ivr[w][q] = 0.0;
                // a one-tap window never looks at its neighbours
                if (w < W && (sd.win_width[w] > 1 || (q >= 1 && q <= kBuildRun)))
                    fetch(w, kl0 + HW - 1 + q, mvr[w][q], ivr[w][q]);
            }
        }
        // One window's share of one frame (mlpg.rs:25-70).  R (frame of the run), the window and its left
        // width are compile-time constants, so that every operand is a named register: with a run-time
        // frame index the compiler kept the run as a loop and picked the operands out of the arrays with
        // chains of v_cndmask (171 of the ~450 VALU instructions per frame).
        // INTERIOR: every source frame and band column exists -- the edge tests are constants and what is
        // left branches on the window shape alone, width and zero coefficients, the same for all lanes.
        // The reference SKIPS a zero coefficient (mlpg.rs:38,47); the general form multiplies by it and
        // adds exact zeros instead, which changes no bit (the accumulators start at +0 and are never -0).
        auto window = [&](auto RC, auto WC, auto LWC, auto INC, const uint32_t k, double (&wuw)[BW], double &wum) {
            constexpr int R = decltype(RC)::value, Wn = decltype(WC)::value, LW = decltype(LWC)::value;
            constexpr bool INTERIOR = decltype(INC)::value;
            const int ww = sd.win_width[Wn];
            const double *coef = sd.win_coef + sd.win_off[Wn];
            double cf[3];
#pragma unroll
            for (int i = 0; i < 3; i++)
                cf[i] = i < ww ? coef[i] : 0.0;
#pragma unroll
            for (int index = 2; index >= 0; index--) {
                // source frame k - d, d = index - LW: slot R + 1 - d of the run's values
                const int d = index - LW;
                const int sl = R + 1 - d;
                const int slc = sl < 0 ? 0 : (sl > kBuildRun + 1 ? kBuildRun + 1 : sl);
                const double ivs = ivr[Wn][slc], mvs = mvr[Wn][slc];
                if (INTERIOR) {
                    if (index >= ww || cf[index] == 0.0)
                        continue;
                    const double wu = cf[index] * ivs;
                    wum += wu * mvs;
#pragma unroll
                    for (int inner = 2; inner >= 0; inner--) {
                        if (inner < index || inner >= ww || cf[inner] == 0.0)
                            continue;
                        wuw[inner - index] += wu * cf[inner];
                    }
                } else {
                    const long idx = (long)k - (long)d;
                    const bool ok = index < ww && idx >= 0 && idx < (long)Tv;
                    const double wu = ok ? cf[index] * ivs : 0.0;
                    wum += wu * mvs;
                    bool live = true; // the reference leaves the inner loop at the first tap past the end
#pragma unroll
                    for (int inner = 2; inner >= 0; inner--) {
                        if (inner < index)
                            continue;
                        const int j = inner - index;
                        // (a zero coefficient is skipped BEFORE the end test in the reference)
                        if (cf[inner] != 0.0 && inner < ww && (uint64_t)k + (uint64_t)j >= Tv)
                            live = false;
                        if (live)
                            wuw[j] += wu * cf[inner];
                    }
                }
            }
        };
        auto frame = [&](auto RC, auto INC) {
            constexpr int R = decltype(RC)::value;
            const int kl = kl0 + R;
            const uint32_t k = k0 + (uint32_t)kl;
            if (!decltype(INC)::value && k >= Tv)
                return;
            double wuw[BW], wum = 0.0;
#pragma unroll
            for (int j = 0; j < BW; j++)
                wuw[j] = 0.0;
            auto one = [&](auto WC) {
                constexpr int Wn = decltype(WC)::value;
                if (Wn >= W)
                    return;
                if (sd.win_width[Wn] / 2 == 0)
                    window(RC, WC, std::integral_constant<int, 0>{}, INC, k, wuw, wum);
                else
                    window(RC, WC, std::integral_constant<int, 1>{}, INC, k, wuw, wum);
            };
            one(std::integral_constant<int, 0>{});
            one(std::integral_constant<int, 1>{});
            one(std::integral_constant<int, 2>{});
#pragma unroll
            for (int j = 0; j < BW; j++)
                tile[j * plane + m * pitch + kl] = wuw[j];
            tile[BW * plane + m * pitch + kl] = wum;
        };
        auto run = [&](auto INC) {
            static_assert(kBuildRun == 2 || kBuildRun == 4 || kBuildRun == 8, "frames of a run, written out");
            frame(std::integral_constant<int, 0>{}, INC);
            frame(std::integral_constant<int, 1>{}, INC);
            if constexpr (kBuildRun > 2) {
                frame(std::integral_constant<int, 2>{}, INC);
                frame(std::integral_constant<int, 3>{}, INC);
            }
            if constexpr (kBuildRun > 4) {
                frame(std::integral_constant<int, 4>{}, INC);
                frame(std::integral_constant<int, 5>{}, INC);
                frame(std::integral_constant<int, 6>{}, INC);
                frame(std::integral_constant<int, 7>{}, INC);
            }
        };
        if (interior)
            run(std::true_type{});
        else
            run(std::false_type{});
    }
    __syncthreads();
    const uint64_t row0 = mt_row0(up, L), Tu = up->mt_rs;
    for (int e = threadIdx.x; e < kBuildTF * L; e += blockDim.x) {
        const int m = e / kBuildTF, kl = e % kBuildTF;
        const uint32_t k = k0 + (uint32_t)kl;
        if (k < Tv) {
            const uint64_t o = row0 + (uint64_t)m * Tu + k;
#pragma unroll
            for (int j = 0; j < BW; j++)
                sd.A[j][o] = tile[j * plane + m * pitch + kl];
            sd.bvec[o] = tile[BW * plane + m * pitch + kl];
        }
    }
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
    const UttDev *up = bd.utt + b;
    const StreamStatesDev st = up->st[si];
    struct { uint32_t S, T; uint64_t frame_off, state_off; const uint32_t *dur; } u = {
        up->S, up->T, up->frame_off, up->state_off, up->dur};
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

// --------------------------------------------------------------------------
// Single static window [c], no GV (the LPF stream): the band system is diagonal,
// D = c*ivar*c, g = c*ivar*mean, par = g/D (mlpg.rs:25-115 with width 1), so the
// whole MlpgAdjust::create is elementwise: thread per (frame, dim), no workspace.
// The same with the canonical-row bookkeeping done per ROW (vector lengths up to 32: the LPF stream): 32 lanes per
// frame, so that one ballot says whether the whole row equals the batch's first row -- and then the row need not be
// stored at all (canon_skip_rows: its readers take row 0, VocDev::lpf_sparse): 1.6 GB per step of config 2, where
// every voiced frame's row is the same.  canon[] is written for every frame (no preset needed).
__global__ __launch_bounds__(256) void k_mlpg_static_rows(BatchDev bd, StreamDev sd, int si)
{
    const int b = blockIdx.y;
    const UttDev *up = bd.utt + b;
    struct { uint32_t T; uint64_t frame_off; } u = {up->T, up->frame_off};
    const int L = sd.L;
    const uint32_t t = blockIdx.x * 8u + (threadIdx.x >> 5);
    if (t >= u.T)
        return; // (a whole row of lanes: the ballot below is per row; an empty utterance has no frame to read)
    const int m = threadIdx.x & 31;
    const bool act = m < L;
    const uint64_t base = u.frame_off;
    const uint32_t tc = t;
    const int mc = m < L ? m : 0;
    const StreamStatesDev st = up->st[si];
    const uint8_t vo = sd.voiced[base + tc];
    const uint32_t s = sd.fstate[base + tc];
    const uint64_t pi = (uint64_t)s * (uint64_t)L + (uint64_t)mc;
    const double var = st.var[pi], mean = st.mean[pi];
    const UttDev *ur = bd.utt + sd.canon_ref_utt;
    const uint64_t pr = (uint64_t)sd.fstate[ur->frame_off] * (uint64_t)L + (uint64_t)mc;
    const StreamStatesDev sr = ur->st[si];
    const bool eq = __double_as_longlong(sr.var[pr]) == __double_as_longlong(var) &&
                    __double_as_longlong(sr.mean[pr]) == __double_as_longlong(mean) && vo == sd.voiced[ur->frame_off];
    const unsigned long long bal = __ballot(eq || !act);
    const bool row_eq = (uint32_t)(bal >> (threadIdx.x & 32)) == 0xFFFFFFFFu;
    const double c = sd.win_coef[0];
    const double wu = c * with_ivar(var);
    const double wum = 0.0 + wu * mean;
    const double d0 = 0.0 + wu * c;
    const double v = vo ? wum / d0 : kNoData;
    if (act && !(sd.canon_skip_rows && row_eq && base + t != 0))
        sd.out[(base + t) * (uint64_t)L + (uint64_t)m] = v;
    if (m == 0)
        sd.canon[base + t] = row_eq ? 1 : 0;
}

__global__ void k_mlpg_static(BatchDev bd, StreamDev sd, int si)
{
    const int b = blockIdx.y;
    const UttDev *up = bd.utt + b;
    struct { uint32_t T; uint64_t frame_off; } u = {up->T, up->frame_off};
    const int L = sd.L;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (uint64_t)u.T * L)
        return;
    const uint32_t t = (uint32_t)(tid / L);
    const int m = (int)(tid % L);
    const uint64_t base = u.frame_off;
    // Two round trips behind the descriptor instead of four: the voiced flag and the state index of the
    // frame together, then both table values, all unconditional (every frame has a state; the flag only
    // selects the result).  With the loads nested in the tests the kernel took 1.0 ms alone and 3.5 ms at
    // the head of the step, where it shares the CUs' wave slots with three other chains.
    const StreamStatesDev st = up->st[si];
    const uint8_t vo = sd.voiced[base + t];
    const uint32_t s = sd.fstate[base + t];
    const uint64_t pi = (uint64_t)s * (uint64_t)L + (uint64_t)m;
    const double var = st.var[pi], mean = st.mean[pi];
    const double c = sd.win_coef[0];
    const double wu = c * with_ivar(var);
    const double wum = 0.0 + wu * mean;
    const double d0 = 0.0 + wu * c;
    const double v = vo ? wum / d0 : kNoData;
    sd.out[(base + t) * (uint64_t)L + (uint64_t)m] = v;
    if (sd.canon) {
        // same inputs as the batch's first frame, bit for bit => the same row (what k_exc_classify asks)
        const UttDev *ur = bd.utt + sd.canon_ref_utt;
        const uint64_t pr = (uint64_t)sd.fstate[ur->frame_off] * (uint64_t)L + (uint64_t)m;
        const StreamStatesDev sr = ur->st[si];
        if (__double_as_longlong(sr.var[pr]) != __double_as_longlong(var) ||
            __double_as_longlong(sr.mean[pr]) != __double_as_longlong(mean) || vo != sd.voiced[ur->frame_off])
            sd.canon[base + t] = 0;
    }
}

// --------------------------------------------------------------------------
// A5-A9 for band width 3 (static + delta + accel windows): same arithmetic and
// summation order as k_mlpg_solve<3>, restructured for the memory system:
//   * every t-loop streams its arrays in chunks of U frames, the next chunk's
//     loads are issued before the current chunk's dependent arithmetic
//     (a lone wave per CU has nothing else to hide HBM latency with);
//   * passes are fused wherever the reference's order of additions allows
//     (mean of the next GV iteration accumulates while par is updated, variance +
//     gradient + objective in one sweep): 15 sweeps instead of 31.
constexpr int MU = 8;

template <bool NONMSD, bool DOGV, bool MT = false>
__global__ __launch_bounds__(64) void k_mlpg_solve3(BatchDev bd, StreamDev sd, int si)
{
    const int b = blockIdx.y;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int L = sd.L;
    if (m >= L)
        return;
    const UttDev *up = bd.utt + b;
    const StreamStatesDev st = up->st[si];
    struct { uint32_t S, T; uint64_t frame_off, state_off; const uint32_t *dur; } u = {
        up->S, up->T, up->frame_off, up->state_off, up->dur};
    const uint32_t n = sd.Tv[b];
    const uint64_t base = u.frame_off;
    // element k of this lane's dim: [frame][dim] rows (stride L) or its own [dim][frame] row
    // (MT is a template parameter so that the stride-1 form can use wide per-lane loads)
    const uint64_t o0 = MT ? mt_row0(up, L) + (uint64_t)m * (uint64_t)up->mt_rs : base * (uint64_t)L + (uint64_t)m;
    const uint64_t Ls = MT ? 1ull : (uint64_t)L;
    const bool fuse_out = NONMSD && !MT; // `out` is [frame][dim]: only then is par's slot also out's
#define IX(k) (o0 + (uint64_t)(k) * Ls)
#define LD(dst, arr, tb)                                                                         \
    _Pragma("unroll") for (int u_ = 0; u_ < MU; u_++)                                            \
    {                                                                                            \
        const uint32_t t_ = (tb) + (uint32_t)u_;                                                 \
        dst[u_] = (arr)[IX(t_ < n ? t_ : n - 1)];                                                \
    }
#define LDS8(dst, arr, tb) /* u8 stream (gv switch), same chunking */                            \
    _Pragma("unroll") for (int u_ = 0; u_ < MU; u_++)                                            \
    {                                                                                            \
        const uint32_t t_ = (tb) + (uint32_t)u_;                                                 \
        dst[u_] = (arr)[t_ < n ? t_ : n - 1];                                                    \
    }
#define LDR(dst, arr, tb) /* descending: element n-1-(tb+u) */                                   \
    _Pragma("unroll") for (int u_ = 0; u_ < MU; u_++)                                            \
    {                                                                                            \
        const uint32_t t_ = (tb) + (uint32_t)u_;                                                 \
        dst[u_] = (arr)[IX(t_ < n ? n - 1 - t_ : 0)];                                            \
    }
#define CP(dst, src) _Pragma("unroll") for (int u_ = 0; u_ < MU; u_++) dst[u_] = src[u_];

    const bool gv_on = sd.use_gv && st.gv_mean && sd.gvlen[b] > 0;
    const uint8_t *sw = sd.vsw + base;
    double *const A0 = sd.A[0], *const A1 = sd.A[1], *const A2 = sd.A[2], *const Bv = sd.bvec;
    double *const F0 = sd.F[0], *const F1 = sd.F[1], *const F2 = sd.F[2], *const Gv = sd.g;
    double *const Pv = sd.par, *const Ov = sd.out;

    if (n > 0) {
        // ---- pass F: ldl_factorization + forward substitution (mlpg.rs:79-105) ----
        {
            double a0[MU], a1[MU], a2[MU], bb[MU], a0n[MU], a1n[MU], a2n[MU], bbn[MU];
            LD(a0, A0, 0) LD(a1, A1, 0) LD(a2, A2, 0) LD(bb, Bv, 0)
            double p1_0 = 0, p1_1 = 0, p1_2 = 0, p2_0 = 0, p2_2 = 0, g1 = 0, g2 = 0;
            for (uint32_t tb = 0; tb < n; tb += MU) {
                LD(a0n, A0, tb + MU) LD(a1n, A1, tb + MU) LD(a2n, A2, tb + MU) LD(bbn, Bv, tb + MU)
#pragma unroll
                for (int uu = 0; uu < MU; uu++) {
                    const uint32_t t = tb + (uint32_t)uu;
                    if (t < n) {
                        double r0 = a0[uu], r1 = a1[uu], r2 = a2[uu], g = bb[uu];
                        if (t >= 1)
                            r0 -= p1_1 * p1_1 * p1_0;
                        if (t >= 2)
                            r0 -= p2_2 * p2_2 * p2_0;
                        if (t >= 1)
                            r1 -= p1_1 * p1_2 * p1_0;
                        r1 /= r0;
                        r2 /= r0;
                        if (t >= 1)
                            g -= p1_1 * g1;
                        if (t >= 2)
                            g -= p2_2 * g2;
                        F0[IX(t)] = r0;
                        F1[IX(t)] = r1;
                        F2[IX(t)] = r2;
                        Gv[IX(t)] = g;
                        p2_0 = p1_0;
                        p2_2 = p1_2;
                        g2 = g1;
                        p1_0 = r0;
                        p1_1 = r1;
                        p1_2 = r2;
                        g1 = g;
                    }
                }
                CP(a0, a0n) CP(a1, a1n) CP(a2, a2n) CP(bb, bbn)
            }
        }
        // ---- pass B: backward substitution (mlpg.rs:106-113), t descending ----
        {
            double f0[MU], f1[MU], f2[MU], gg[MU], f0n[MU], f1n[MU], f2n[MU], ggn[MU];
            LDR(f0, F0, 0) LDR(f1, F1, 0) LDR(f2, F2, 0) LDR(gg, Gv, 0)
            double q1 = 0, q2 = 0;
            for (uint32_t tb = 0; tb < n; tb += MU) {
                LDR(f0n, F0, tb + MU) LDR(f1n, F1, tb + MU) LDR(f2n, F2, tb + MU) LDR(ggn, Gv, tb + MU)
#pragma unroll
                for (int uu = 0; uu < MU; uu++) {
                    const uint32_t r = tb + (uint32_t)uu;
                    if (r < n) {
                        const uint32_t t = n - 1 - r;
                        double p = gg[uu] / f0[uu];
                        if (t + 1 < n)
                            p -= f1[uu] * q1;
                        if (t + 2 < n)
                            p -= f2[uu] * q2;
                        if (fuse_out && !gv_on)
                            Ov[IX(t)] = p; // scatter fused: every frame is voiced
                        else
                            Pv[IX(t)] = p;
                        q2 = q1;
                        q1 = p;
                    }
                }
                CP(f0, f0n) CP(f1, f1n) CP(f2, f2n) CP(gg, ggn)
            }
        }
        // ---- GV (mlpg.rs:145-292) ----
        if (DOGV && gv_on) {
            const double gv_mean = st.gv_mean[m] * st.gv_weight; // mlpg.rs:135-137
            const double gv_vari = st.gv_var[m];
            const double glen = (double)sd.gvlen[b];
            double mean, vari;
            double pc[MU], pn[MU];
            uint8_t wc[MU], wn[MU];
            // conv_gv (mlpg.rs:195-203): mean, variance, rescale; the rescale sweep
            // also accumulates the sum for iteration 1's mean (same order of additions)
            double ssum = 0.0;
            LD(pc, Pv, 0) LDS8(wc, sw, 0)
            for (uint32_t tb = 0; tb < n; tb += MU) {
                LD(pn, Pv, tb + MU) LDS8(wn, sw, tb + MU)
#pragma unroll
                for (int uu = 0; uu < MU; uu++) {
                    const uint32_t t = tb + (uint32_t)uu;
                    if (t < n && wc[uu])
                        ssum += pc[uu];
                }
                CP(pc, pn) CP(wc, wn)
            }
            mean = ssum / glen;
            double vsum = 0.0;
            LD(pc, Pv, 0) LDS8(wc, sw, 0)
            for (uint32_t tb = 0; tb < n; tb += MU) {
                LD(pn, Pv, tb + MU) LDS8(wn, sw, tb + MU)
#pragma unroll
                for (int uu = 0; uu < MU; uu++) {
                    const uint32_t t = tb + (uint32_t)uu;
                    if (t < n && wc[uu])
                        vsum += (pc[uu] - mean) * (pc[uu] - mean);
                }
                CP(pc, pn) CP(wc, wn)
            }
            vari = vsum / glen;
            {
                const double ratio = sqrt(gv_mean / vari);
                ssum = 0.0;
                LD(pc, Pv, 0) LDS8(wc, sw, 0)
                for (uint32_t tb = 0; tb < n; tb += MU) {
                    LD(pn, Pv, tb + MU) LDS8(wn, sw, tb + MU)
#pragma unroll
                    for (int uu = 0; uu < MU; uu++) {
                        const uint32_t t = tb + (uint32_t)uu;
                        if (t < n && wc[uu]) {
                            const double p = ratio * (pc[uu] - mean) + mean;
                            Pv[IX(t)] = p;
                            ssum += p;
                        }
                    }
                    CP(pc, pn) CP(wc, wn)
                }
            }
            double step = 0.1, prev = 0.0; // STEPINIT
            const double length = (double)n;
            const double wgt = 1.0 / (double)((uint64_t)sd.W * (uint64_t)n);
            const double ll = (double)((uint64_t)n * (uint64_t)n);
            const double lm1 = (double)(n - 1);
            for (int it = 1; it <= 5; it++) { // GV_MAX_ITERATION
                mean = ssum / glen; // calc_gv, first half (sum accumulated by the previous sweep)
                // ---- sweep V: variance + calc_hmmobj_derivative (mlpg.rs:173-229) ----
                double hmmobj = 0.0;
                vsum = 0.0;
                {
                    double a0[MU], a1[MU], a2[MU], bb[MU], a0n[MU], a1n[MU], a2n[MU], bbn[MU];
                    // par stream runs 2 frames ahead: pc[u] = par[t+2]
                    LD(a0, A0, 0) LD(a1, A1, 0) LD(a2, A2, 0) LD(bb, Bv, 0) LD(pc, Pv, 2) LDS8(wc, sw, 0)
                    double pm2 = 0, pm1 = 0, p0 = Pv[IX(0)], pp1 = n > 1 ? Pv[IX(1)] : 0.0;
                    double a1m1 = 0, a2m1 = 0, a2m2 = 0;
                    for (uint32_t tb = 0; tb < n; tb += MU) {
                        LD(a0n, A0, tb + MU) LD(a1n, A1, tb + MU) LD(a2n, A2, tb + MU)
                        LD(bbn, Bv, tb + MU) LD(pn, Pv, tb + MU + 2) LDS8(wn, sw, tb + MU)
#pragma unroll
                        for (int uu = 0; uu < MU; uu++) {
                            const uint32_t t = tb + (uint32_t)uu;
                            if (t < n) {
                                const double pp2 = pc[uu];
                                if (wc[uu])
                                    vsum += (p0 - mean) * (p0 - mean);
                                double g = a0[uu] * p0;
                                if (t + 1 < n)
                                    g += a1[uu] * pp1;
                                if (t >= 1)
                                    g += a1m1 * pm1;
                                if (t + 2 < n)
                                    g += a2[uu] * pp2;
                                if (t >= 2)
                                    g += a2m2 * pm2;
                                Gv[IX(t)] = g;
                                hmmobj += 1.0 * wgt * p0 * (bb[uu] - 0.5 * g);
                                pm2 = pm1;
                                pm1 = p0;
                                p0 = pp1;
                                pp1 = pp2;
                                a2m2 = a2m1;
                                a2m1 = a2[uu];
                                a1m1 = a1[uu];
                            }
                        }
                        CP(a0, a0n) CP(a1, a1n) CP(a2, a2n) CP(bb, bbn) CP(pc, pn) CP(wc, wn)
                    }
                }
                vari = vsum / glen;
                const double gvobj = -0.5 * 1.0 * vari * gv_vari * (vari - 2.0 * gv_mean);
                const double obj = -(hmmobj + gvobj);
                if (it > 1) {
                    if (obj > prev)
                        step *= 0.5; // STEPDEC
                    else if (obj < prev)
                        step *= 1.2; // STEPINC
                }
                // ---- sweep N: next_step (mlpg.rs:230-258) + sum for the next mean ----
                const double dv = -2.0 * gv_vari * (vari - gv_mean) / length;
                ssum = 0.0;
                {
                    double a0[MU], bb[MU], gg[MU], a0n[MU], bbn[MU], ggn[MU];
                    LD(a0, A0, 0) LD(bb, Bv, 0) LD(gg, Gv, 0) LD(pc, Pv, 0) LDS8(wc, sw, 0)
                    for (uint32_t tb = 0; tb < n; tb += MU) {
                        LD(a0n, A0, tb + MU) LD(bbn, Bv, tb + MU) LD(ggn, Gv, tb + MU) LD(pn, Pv, tb + MU)
                        LDS8(wn, sw, tb + MU)
#pragma unroll
                        for (int uu = 0; uu < MU; uu++) {
                            const uint32_t t = tb + (uint32_t)uu;
                            if (t < n) {
                                const double p = pc[uu];
                                const double h = -1.0 * wgt * a0[uu] -
                                                 1.0 * 2.0 / ll *
                                                     (lm1 * gv_vari * (vari - gv_mean) +
                                                      2.0 * gv_vari * (p - mean) * (p - mean));
                                const bool on = wc[uu] != 0;
                                double next_g;
                                if (on)
                                    next_g = 1.0 / h *
                                             (1.0 * wgt * (-gg[uu] + bb[uu]) + 1.0 * dv * (p - mean));
                                else
                                    next_g = 1.0 / h * (1.0 * wgt * (-gg[uu] + bb[uu]));
                                const double pnew = p + step * next_g;
                                if (fuse_out && it == 5)
                                    Ov[IX(t)] = pnew; // scatter fused into the last sweep
                                else
                                    Pv[IX(t)] = pnew;
                                if (on)
                                    ssum += pnew;
                            }
                        }
                        CP(a0, a0n) CP(bb, bbn) CP(gg, ggn) CP(pc, pn) CP(wc, wn)
                    }
                }
                prev = obj;
            }
        }
    }
    // A9 scatter for MSD streams: k_mlpg_scatter (time-parallel)
#undef IX
#undef LD
#undef LDR
#undef LDS8
#undef CP
}

// --------------------------------------------------------------------------
// A5/A6 for the [dim][frame] workspace (band LDL^T + forward substitution, then backward substitution,
// mlpg.rs:79-115), LDS-STAGED.  With one lane per (utterance, dim) streaming its own row, a wave touches 35
// cache lines per memory instruction and the texture addresser sets the pace (an uncoalesced wave access
// costs ~150 cycles whatever its width: 10.3 ms; tools/experiments/README.md).
// Here one block owns an utterance: wave 0 is the SOLVER (lane = dim, the same recurrences in the
// same order, operands and results in LDS as [frame][dim] tiles, so its accesses are
// conflict-free), waves 1-3 are MOVERS that stream the [dim][frame] rows between HBM and LDS with
// coalesced accesses (16 consecutive frames of a row = 128 B per 16 lanes).  Chunks of kFlCT
// frames, one __syncthreads per chunk: while the solver works on chunk p, the movers fetch chunk
// p+2 into registers, drain the results of chunk p-1 from LDS to HBM, then park chunk p+2 in its
// LDS slot.  Three input slots, two output slots.  The solver is then bound by its dependency
// chain (two f64 divisions per frame in the factorisation) and the kernel as a whole by HBM.
constexpr int kFlCT = 16;  // frames per chunk
constexpr int kFlIn = 3;   // input slots (chunk p being solved, p+1 parked, p+2 arriving)
constexpr int kFlOut = 3;  // output slots (chunk p being written, p-1 waiting, p-2 draining)
constexpr int kFlNT = 256; // threads per block: the solver wave and three mover waves (a wave per SIMD: with a
                           // fifth wave two share a SIMD's registers, 256 each, and the solver's spill to scratch)
constexpr int kFlMovers = kFlNT - 64;
constexpr int kFlGroups = kFlMovers / 64; // mover groups (one wave each)

// IEEE-754 double division as the compiler expands it (v_rcp_f64, two Newton steps, quotient,
// residual correction) minus v_div_scale / v_div_fixup, with the refined reciprocal shared between
// quotients by the same denominator.  The solver wave's instruction count is the kernel's time and
// the divisions were 45 % of it (13 instructions each).  Bit-identical to `a / d` wherever the
// scaling steps are the identity, i.e. for finite non-zero d and quotients away from the ends of the
// exponent range -- D_t is a sum of inverse variances (|ivar| <= 1e38, mean_vari.rs:21-31); the
// tracks' hashes are unchanged (tests/tools/ab_bits.py) and the oracle comparisons stay bitwise.
__device__ __forceinline__ double fb_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double fb_div(double a, double d, double r)
{
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-d, q, a), r, q);
}

// One pass (NIN input arrays, NOUT output arrays).  lo(c) = first frame of chunk c (may be
// negative / the chunk may run past n: those frames are skipped); the solver visits the frames of
// a chunk ascending (forward) or descending (backward).  LMAX bounds the vector length, so that
// the movers' per-thread element lists are as short as the stream allows (1 for LF0, 12 for MCP).
template <int LMAX, int NIN, int NOUT, bool BACKWARD, class Step>
__device__ __forceinline__ void fl_pass(double *lds, const double *const (&in)[NIN], double *const (&out)[NOUT],
                                        uint32_t n, int L, uint64_t rs, Step step)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const bool solver = tid < 64;
    const int mt = tid - 64;
    // the tile pitch of the MCP instantiation is a constant, not the stream's L: LDS addresses are then base +
    // immediate offset (with a run-time pitch the solver spent 4 of its 42 instructions per frame forming them)
    // (the pitch must stay odd: the movers walk a tile frame-fastest, and with 36 doubles between frames
    // their LDS accesses collide eight ways -- the kernel took 11.9 ms instead of 5.6)
    const int LP = LMAX == 36 ? 37 : L;
    const int nch = (int)((n + kFlCT - 1) / kFlCT);
    double *inb = lds;                                // [kFlIn][NIN][kFlCT][LP]
    double *outb = lds + kFlIn * 4 * kFlCT * LP;      // [kFlOut][NOUT][kFlCT][LP]
    auto lo = [&](int c) { return BACKWARD ? (long)n - (long)(c + 1) * kFlCT : (long)c * kFlCT; };
    // MOVERS: kFlGroups waves; wave g owns the chunks c = g (mod kFlGroups) from HBM to HBM -- it loads chunk
    // c during phase c-2-kFlGroups, parks it in LDS during phase c-2 (and loads its next chunk), drains its
    // results during phase c+2 and idles otherwise.  A thread so holds ONE set of prefetch registers, filled
    // kFlGroups phases before it is read, and a CU keeps kFlGroups chunks (54 KB for MCP) in flight: with two
    // (loads two phases ahead) the kernel moved 36 KB per ~3 us of loaded-HBM latency and CU, 2.9 TB/s.
    // (The first form had every mover do all three jobs every phase with two register sets alternating;
    // whether the compiler kept the sets apart, or funnelled both through the same registers and waited for
    // every load where it was issued, changed with unrelated edits to this function: 5.6 or 11.5 ms.)
    // Element list of a thread, the same for every array: j = mtl + GT*kk -> (dim m, frame-in-chunk tt).
    constexpr int GT = 64, NG = kFlGroups;
    constexpr int KPA = (LMAX * kFlCT + GT - 1) / GT; // 9 for MCP, 1 for LF0
    const int grp = mt >> 6, mtl = mt & 63;
    uint64_t goff[KPA]; // m*rs + tt: the element in a [dim][frame] array
    uint32_t loff[KPA]; // tt*LP + m: the element in one array's tile
    uint32_t ett[KPA];
    bool val[KPA];
#pragma unroll
    for (int kk = 0; kk < KPA; kk++) {
        const int j = mtl + GT * kk;
        const int tt = j % kFlCT, m = j / kFlCT;
        val[kk] = !solver && m < L;
        const int mc = m < L ? m : 0; // (an element past the tile: a valid address, never parked or stored)
        goff[kk] = (uint64_t)mc * rs + (uint64_t)tt;
        loff[kk] = (uint32_t)(tt * LP + mc);
        ett[kk] = (uint32_t)tt;
    }
    double regs[NIN][KPA];
#pragma unroll
    for (int a2 = 0; a2 < NIN; a2++)
#pragma unroll
        for (int kk = 0; kk < KPA; kk++)
            regs[a2][kk] = 0.0;
    // Phase p: the solver works on chunk p.  The phase barrier is a raw s_barrier behind lgkmcnt(0) only:
    // __syncthreads() would also wait for the movers' stores and loads in flight (vmcnt(0)).
    auto mover_phase = [&](const int p) {
        if ((((p + 2) % NG) + NG) % NG == grp) {
            const int cw = p + 2; // loaded NG phases ago
            if (cw >= 0 && cw < nch) {
                double *ib = inb + (cw % kFlIn) * NIN * kFlCT * LP;
#pragma unroll
                for (int a2 = 0; a2 < NIN; a2++)
#pragma unroll
                    for (int kk = 0; kk < KPA; kk++)
                        if (val[kk])
                            ib[a2 * kFlCT * LP + loff[kk]] = regs[a2][kk];
            }
            const int cl = p + 2 + NG;
            if (cl >= 0 && cl < nch) {
                const long l0 = lo(cl);
                if (l0 >= 0 && l0 + kFlCT <= (long)n) { // interior chunk: no per-frame checks, and
                                                        // UNCONDITIONAL loads (no branch per load)
#pragma unroll
                    for (int a2 = 0; a2 < NIN; a2++)
#pragma unroll
                        for (int kk = 0; kk < KPA; kk++)
                            regs[a2][kk] = in[a2][goff[kk] + (uint64_t)l0];
                } else {
#pragma unroll
                    for (int a2 = 0; a2 < NIN; a2++)
#pragma unroll
                        for (int kk = 0; kk < KPA; kk++) {
                            const long t = l0 + (long)ett[kk];
                            regs[a2][kk] = (val[kk] && t >= 0 && t < (long)n) ? in[a2][(long)goff[kk] + l0] : 0.0;
                        }
                }
            }
        } else if ((((p - 2) % NG) + NG) % NG == grp) {
            const int cs = p - 2; // solved two phases ago (with three groups p-1 is the parking group's class)
            if (cs >= 0 && cs < nch) {
                const long l0 = lo(cs);
                const double *ob = outb + (cs % kFlOut) * NOUT * kFlCT * LP;
                if (l0 >= 0 && l0 + kFlCT <= (long)n) {
#pragma unroll
                    for (int a2 = 0; a2 < NOUT; a2++)
#pragma unroll
                        for (int kk = 0; kk < KPA; kk++)
                            if (val[kk])
                                out[a2][goff[kk] + (uint64_t)l0] = ob[a2 * kFlCT * LP + loff[kk]];
                } else {
#pragma unroll
                    for (int a2 = 0; a2 < NOUT; a2++)
#pragma unroll
                        for (int kk = 0; kk < KPA; kk++) {
                            const long t = l0 + (long)ett[kk];
                            if (val[kk] && t >= 0 && t < (long)n)
                                out[a2][(long)goff[kk] + l0] = ob[a2 * kFlCT * LP + loff[kk]];
                        }
                }
            }
        }
    };
    auto solver_phase = [&](const int p) {
        if (p >= 0 && p < nch && lane < L) {
            const long l0 = lo(p);
            const double *ib = inb + (p % kFlIn) * NIN * kFlCT * LP;
            double *ob = outb + (p % kFlOut) * NOUT * kFlCT * LP;
            // A half chunk's operands come out of LDS before its recurrence starts: reads interleaved with the
            // result writes would each wait out a full LDS round trip inside the dependency chain.  HALF a
            // chunk, not the whole: with 16 x 4 operands in flight the kernel needed 256 VGPRs plus ~150
            // AccVGPRs of spill space per wave (now ~80), which is what a SIMD must have free to take a wave of
            // this workgroup beside the waves of the other chains.
            constexpr int kHalf = kFlCT / 2;
            const bool interior = l0 >= 2 && l0 + kFlCT + 2 <= (long)n;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                double ivs[kHalf][NIN];
#pragma unroll
                for (int uu = 0; uu < kHalf; uu++) {
                    const int u = h * kHalf + uu;
                    const int tt = BACKWARD ? kFlCT - 1 - u : u;
#pragma unroll
                    for (int a = 0; a < NIN; a++)
                        ivs[uu][a] = ib[(a * kFlCT + tt) * LP + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (interior) {
                    // interior chunk: every frame exists and has both neighbours on either side, so
                    // the recurrences run without their edge guards (same operations, same order)
#pragma unroll
                    for (int uu = 0; uu < kHalf; uu++) {
                        const int u = h * kHalf + uu;
                        const int tt = BACKWARD ? kFlCT - 1 - u : u;
                        double ov[NOUT];
                        step(std::true_type{}, (uint32_t)(l0 + tt), ivs[uu], ov);
#pragma unroll
                        for (int a = 0; a < NOUT; a++)
                            ob[(a * kFlCT + tt) * LP + lane] = ov[a];
                    }
                } else {
#pragma unroll
                    for (int uu = 0; uu < kHalf; uu++) {
                        const int u = h * kHalf + uu;
                        const int tt = BACKWARD ? kFlCT - 1 - u : u;
                        const long t = l0 + tt;
                        if (t >= 0 && t < (long)n) {
                            double ov[NOUT];
                            step(std::false_type{}, (uint32_t)t, ivs[uu], ov);
#pragma unroll
                            for (int a = 0; a < NOUT; a++)
                                ob[(a * kFlCT + tt) * LP + lane] = ov[a];
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto phase_end = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    static_assert(kFlGroups == 3 && kFlOut == 3, "roles: park c-2 / drain c+2 fall on different groups for 3");
    for (int p = -2 - kFlGroups; p <= nch + 1; p++) {
        if (solver)
            solver_phase(p);
        else
            mover_phase(p);
        phase_end();
    }
}

template <int LMAX>
__global__ __launch_bounds__(kFlNT, 1) void k_mlpg_fb_lds(BatchDev bd, StreamDev sd, int si)
{
    extern __shared__ double lds[];
    const int b = (int)bd.order[blockIdx.x]; // longest utterance first: the serial sweeps of a ragged
                                             // batch end together instead of with a late long one
    const UttDev *up = bd.utt + b;
    const uint32_t n = sd.Tv[b];
    if (n == 0)
        return;
    const int L = sd.L;
    const uint64_t row0 = mt_row0(up, L);
    const uint64_t rs = up->mt_rs; // row stride of the [dim][frame] workspace
    // The solver's dependent chain is the critical path of the kernel (and of the step): it gets
    // the highest issue priority on its SIMD, the movers the next one, so that throughput kernels
    // sharing the CU (the excitation pass runs concurrently) fill the gaps instead of taking turns.
    if (threadIdx.x < 64)
        __builtin_amdgcn_s_setprio(3);
    else
        __builtin_amdgcn_s_setprio(2);
    // ---- pass F: ldl_factorization + forward substitution (mlpg.rs:79-105) ----
    {
        const double *const in[4] = {sd.A[0] + row0, sd.A[1] + row0, sd.A[2] + row0, sd.bvec + row0};
        // what the backward pass needs of a frame: L1, L2 and g/D.  The quotient is formed HERE, off the
        // recurrence's dependency chain and with the reciprocal of D_t that the factors already use (the
        // same two operations on the same operands as mlpg.rs:108's division, so the same bits); the
        // backward chain loses its division and D_t itself never goes to memory (-3.6 GB per step).
        double *const out[3] = {sd.F[1] + row0, sd.F[2] + row0, sd.g + row0};
        double p1_0 = 0, p1_1 = 0, p1_2 = 0, p2_0 = 0, p2_2 = 0, g1 = 0, g2 = 0;
        fl_pass<LMAX, 4, 3, false>(lds, in, out, n, L, rs,
                                   [&](auto interior, uint32_t t, const double (&iv)[4], double (&ov)[3]) {
            constexpr bool IN = decltype(interior)::value;
            double r0 = iv[0], r1 = iv[1], r2 = iv[2], g = iv[3];
            if (IN || t >= 1)
                r0 -= p1_1 * p1_1 * p1_0;
            if (IN || t >= 2)
                r0 -= p2_2 * p2_2 * p2_0;
            if (IN || t >= 1)
                r1 -= p1_1 * p1_2 * p1_0;
            // two quotients by the same D_t: one refined reciprocal serves both (fb_div below)
            const double rr = fb_rcp(r0);
            r1 = fb_div(r1, r0, rr);
            r2 = fb_div(r2, r0, rr);
            if (IN || t >= 1)
                g -= p1_1 * g1;
            if (IN || t >= 2)
                g -= p2_2 * g2;
            ov[0] = r1;
            ov[1] = r2;
            ov[2] = fb_div(g, r0, rr);
            p2_0 = p1_0;
            p2_2 = p1_2;
            g2 = g1;
            p1_0 = r0;
            p1_1 = r1;
            p1_2 = r2;
            g1 = g;
        });
    }
    // the factors were stored by the mover waves; the backward pass reads them back through the
    // same waves in the opposite order
    __threadfence();
    __syncthreads();
    // ---- pass B: backward substitution (mlpg.rs:106-113), t descending ----
    {
        const double *const in[3] = {sd.F[1] + row0, sd.F[2] + row0, sd.g + row0};
        double *const out[1] = {sd.par + row0};
        double q1 = 0, q2 = 0;
        fl_pass<LMAX, 3, 1, true>(lds, in, out, n, L, rs,
                                  [&](auto interior, uint32_t t, const double (&iv)[3], double (&ov)[1]) {
            constexpr bool IN = decltype(interior)::value;
            double p = iv[2]; // g_t / D_t, from the forward pass
            if (IN || t + 1 < n)
                p -= iv[0] * q1;
            if (IN || t + 2 < n)
                p -= iv[1] * q2;
            ov[0] = p;
            q2 = q1;
            q1 = p;
        });
    }
}

// --------------------------------------------------------------------------
// A5/A6 for an MSD stream with one dimension (LF0): ONE LANE PER VOICED RUN.
// The compacted system (mlpg.rs:79-115 on the frames that pass the mask, mod.rs:81) is block
// diagonal: a dynamic window that touches an MSD boundary has its inverse variance zeroed
// (mod.rs:69-80), so the band entries that would couple the last frame of a voiced run to the first
// of the next are exact zeros -- A1[t_e], A2[t_e - 1], A2[t_e] -- and with them L1[t_e], L2[t_e - 1],
// L2[t_e] (0 - L*0*D = 0, 0 / D = 0).  The serial recurrences therefore subtract exact zeros at every
// run start, and a run solved on its own (first frame treated like t = 0, last like t = n - 1) gives
// the same bits.  k_mlpg_fb_lds walks the 15 k voiced frames of an utterance on ONE lane (4.2 ms for
// 128 s); here the ~400 runs of an utterance are walked side by side, the longest run (a few
// hundred frames) setting the time.  Same operations in the same order as k_mlpg_fb_lds's passes.
// Memory side (round 4): a lane that streams its own run touches its own cache line with every access, 64 lines
// per wave instruction -- the ~1,400 waves of config 2 kept the texture addresser of every CU busy and the
// MCP chain's inverse-variance pass beside them took 2.0 ms instead of 0.6.  So the wave moves chunks of
// kFrCh frames of all its 64 runs COOPERATIVELY: eight lanes fetch the eight frames of one run (64 B), eight
// runs per instruction (8 lines instead of 64), through an LDS image [run][frame] (row pitch 9: conflict-free
// both ways); the recurrences read and write the image.  The next chunk's values are requested before the
// current chunk is worked on.
constexpr int kFrCh = 8;            // frames per chunk
constexpr int kFrPitch = kFrCh + 1; // LDS row pitch in doubles
__global__ __launch_bounds__(64) void k_mlpg_fb_runs(BatchDev bd, StreamDev sd, int si)
{
    const int b = (int)bd.order[blockIdx.y];
    const UttDev *up = bd.utt + b;
    const int lane = threadIdx.x;
    const uint32_t r = blockIdx.x * 64u + (uint32_t)lane;
    const uint32_t nr = sd.nruns[b];
    if (blockIdx.x * 64u >= nr)
        return;
    const uint64_t base = up->frame_off, sb = up->state_off;
    const uint32_t T = up->T;
    // this lane's run: compacted offset k0 and length (0: no run on this lane)
    uint32_t k0 = 0, len = 0;
    if (r < nr) {
        const uint32_t t0 = sd.run_list[sb + r];
        // entries of the run list: first frame of a voiced run; a zero-length voiced state can repeat an
        // entry or leave one that continues the previous run (k_prep_states)
        const bool ok = t0 < T && sd.voiced[base + t0] && !(t0 > 0 && sd.voiced[base + t0 - 1]) &&
                        !(r > 0 && sd.run_list[sb + r - 1] == t0);
        if (ok) {
            const uint32_t s0 = sd.fstate[base + t0];
            k0 = sd.s_vpre[sb + s0] + (t0 - sd.s_start[sb + s0]); // compacted index of the run's first frame
            len = sd.s_rend[sb + s0] - t0 + 1;
        }
    }
    __shared__ uint32_t rk0[64], rlen[64];
    __shared__ double img[4][64][kFrPitch]; // four arrays x [run][frame]
    rk0[lane] = k0;
    rlen[lane] = len;
    uint32_t maxlen = len;
    for (int o = 32; o > 0; o >>= 1)
        maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, o));
    if (maxlen == 0)
        return;
    const uint32_t nch = (maxlen + kFrCh - 1) / kFrCh;
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    wsync();
    const int sub = lane & 7, grp = lane >> 3; // cooperative side: frame within the chunk, run within the octet
    const double *const in4[4] = {sd.A[0] + base, sd.A[1] + base, sd.A[2] + base, sd.bvec + base};
    double *const f4[4] = {sd.F[0] + base, sd.F[1] + base, sd.F[2] + base, sd.g + base};
    // fetch chunk c of four arrays for all 64 runs into registers: v[a][j] = arr_a[k0(8j+grp) + 8c + sub]
    auto fetch = [&](const double *const (&arr)[4], uint32_t c, double (&v)[4][8]) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int rr = 8 * j + grp;
            const uint32_t t = c * kFrCh + (uint32_t)sub;
            const bool in = t < rlen[rr];
            const uint64_t o = (uint64_t)rk0[rr] + (in ? t : 0u);
#pragma unroll
            for (int a = 0; a < 4; a++)
                v[a][j] = arr[a][o]; // (a lane past its run's end re-reads the run's first frame: valid, unused)
        }
    };
    auto park = [&](const double (&v)[4][8]) {
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int a = 0; a < 4; a++)
                img[a][8 * j + grp][sub] = v[a][j];
    };
    // the image's rows back to memory: arr_a[k0(run) + 8c + sub] for the frames inside the run
    auto drain = [&](double *const *arr, int na, uint32_t c) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int rr = 8 * j + grp;
            const uint32_t t = c * kFrCh + (uint32_t)sub;
            if (t < rlen[rr]) {
                const uint64_t o = (uint64_t)rk0[rr] + t;
                for (int a = 0; a < na; a++)
                    arr[a][o] = img[a][rr][sub];
            }
        }
    };
    // ---- ldl_factorization + forward substitution (mlpg.rs:79-105) ----
    {
        double p1_0 = 0, p1_1 = 0, p1_2 = 0, p2_0 = 0, p2_2 = 0, g1 = 0, g2 = 0;
        double v[4][8];
        fetch(in4, 0, v);
        for (uint32_t c = 0; c < nch; c++) {
            wsync(); // the previous chunk's drain has read the image
            park(v);
            if (c + 1 < nch)
                fetch(in4, c + 1, v);
            wsync();
            const uint32_t tb = c * kFrCh;
#pragma unroll
            for (int i = 0; i < kFrCh; i++) {
                const uint32_t t = tb + (uint32_t)i;
                if (t < len) {
                    double r0 = img[0][lane][i], r1 = img[1][lane][i], r2 = img[2][lane][i], g = img[3][lane][i];
                    if (t >= 1)
                        r0 -= p1_1 * p1_1 * p1_0;
                    if (t >= 2)
                        r0 -= p2_2 * p2_2 * p2_0;
                    if (t >= 1)
                        r1 -= p1_1 * p1_2 * p1_0;
                    const double rr = fb_rcp(r0);
                    r1 = fb_div(r1, r0, rr);
                    r2 = fb_div(r2, r0, rr);
                    if (t >= 1)
                        g -= p1_1 * g1;
                    if (t >= 2)
                        g -= p2_2 * g2;
                    img[0][lane][i] = r0;
                    img[1][lane][i] = r1;
                    img[2][lane][i] = r2;
                    img[3][lane][i] = g;
                    p2_0 = p1_0;
                    p2_2 = p1_2;
                    g2 = g1;
                    p1_0 = r0;
                    p1_1 = r1;
                    p1_2 = r2;
                    g1 = g;
                }
            }
            wsync();
            drain(f4, 4, c);
        }
    }
    __threadfence_block(); // the backward pass reads back what the wave has just stored
    wsync();
    // ---- backward substitution (mlpg.rs:106-113), chunks and frames descending ----
    {
        double q1 = 0, q2 = 0;
        double v[4][8];
        double *const p1[1] = {sd.par + base};
        const double *const fr4[4] = {sd.F[0] + base, sd.F[1] + base, sd.F[2] + base, sd.g + base};
        fetch(fr4, nch - 1, v);
        for (uint32_t c = nch; c-- > 0;) {
            wsync();
            park(v);
            if (c > 0)
                fetch(fr4, c - 1, v);
            wsync();
            const uint32_t tb = c * kFrCh;
#pragma unroll
            for (int i = kFrCh - 1; i >= 0; i--) {
                const uint32_t t = tb + (uint32_t)i;
                if (t < len) {
                    const double d = img[0][lane][i];
                    double p = fb_div(img[3][lane][i], d, fb_rcp(d));
                    if (t + 1 < len)
                        p -= img[1][lane][i] * q1;
                    if (t + 2 < len)
                        p -= img[2][lane][i] * q2;
                    img[0][lane][i] = p; // (row 0 of the image now carries par)
                    q2 = q1;
                    q1 = p;
                }
            }
            wsync();
            drain(p1, 1, c);
        }
    }
}

// --------------------------------------------------------------------------
// A8 GV ascent with LANES OVER TIME (mlpg.rs:145-292).  The 13 sweeps of conv_gv and the
// five parmgen iterations are elementwise in t plus reductions; only the ORDER of the
// additions is serial.  One wave per (utterance, dim): 64 frames per vector instruction for
// everything elementwise (loads, gradient, Newton step, stores), and the reductions run as
// an in-order chain acc += v[lane 0]; acc += v[lane 1]; ... with v_readlane -- the
// reference's order of additions exactly (a switched-off frame contributes +0.0, which
// leaves the sum bit-identical), so the result stays bit-exact while LF0 uses 64 lanes
// instead of 1 and MCP runs 35x more waves than the lane-per-dim solver.
__device__ __forceinline__ double rl64(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double serial_add64(double acc, double v)
{
#pragma unroll
    for (int u = 0; u < 64; u++)
        acc += rl64(v, u);
    return acc;
}

// Loads of kVtNB blocks of 64 frames are issued together before their serial sums run, so a
// lone wave pays one memory latency per 512 frames instead of one per 64.
#ifndef JB_VT_NB
#define JB_VT_NB 8
#endif
constexpr int kVtNB = JB_VT_NB; // 4: 3.08 ms, 8: 2.85, 16: 2.90 (LF0 of config 2, alone)

// In-order sum of one value per lane, acc += v[lane 0]; acc += v[lane 1]; ... -- through LDS:
// the wave parks the values and every lane reads them back in order with broadcast
// ds_read_b128 (two values per read), so the chain is one dependent v_add_f64 per frame
// (~2.7 ns) instead of two v_readlane + add with their SGPR hazards (~30 ns).
__device__ __forceinline__ double serial_add_lds(double acc, const double *buf)
{
    // reads run one group of 8 (16 summands) ahead of the adds; the scheduling barriers keep
    // hipcc from sinking them next to their use (it otherwise keeps ~2 reads in flight and the
    // chain waits out most of every LDS round trip)
    const double2 *b2 = reinterpret_cast<const double2 *>(buf);
    double2 x[2][8];
#pragma unroll
    for (int u = 0; u < 8; u++)
        x[0][u] = b2[u];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        if (g < 3) {
#pragma unroll
            for (int u = 0; u < 8; u++)
                x[(g + 1) & 1][u] = b2[8 * (g + 1) + u];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc += x[g & 1][u].x;
            acc += x[g & 1][u].y;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// Two independent in-order sums side by side (variance and HMM objective of an ascent iteration): each is
// a chain of dependent v_add_f64, and two chains fill the slots that one leaves empty.
__device__ __forceinline__ void serial_add2_lds(double &acc_a, const double *buf_a, double &acc_b, const double *buf_b)
{
    const double2 *a2 = reinterpret_cast<const double2 *>(buf_a), *b2 = reinterpret_cast<const double2 *>(buf_b);
    double2 x[2][4], y[2][4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        x[0][u] = a2[u];
        y[0][u] = b2[u];
    }
#pragma unroll
    for (int g = 0; g < 8; g++) {
        if (g < 7) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                x[(g + 1) & 1][u] = a2[4 * (g + 1) + u];
                y[(g + 1) & 1][u] = b2[4 * (g + 1) + u];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            acc_a += x[g & 1][u].x;
            acc_b += y[g & 1][u].x;
            acc_a += x[g & 1][u].y;
            acc_b += y[g & 1][u].y;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool NONMSD>
__global__ __launch_bounds__(64) void k_mlpg_gv_vt(BatchDev bd, StreamDev sd, int si)
{
    const int m = blockIdx.x, b = (int)bd.order[blockIdx.y], lane = threadIdx.x; // longest first
    const UttDev *up = bd.utt + b;
    const StreamStatesDev st = up->st[si];
    const uint32_t n = sd.Tv[b];
    const uint32_t gvl = sd.gvlen[b];
    if (n == 0 || !(sd.use_gv && st.gv_mean && gvl > 0))
        return;
    const int L = sd.L;
    const uint64_t base = up->frame_off;
    const uint64_t o0 = base * (uint64_t)L + (uint64_t)m, Ls = (uint64_t)L;
#define IX(k) (o0 + (uint64_t)(k) * Ls)
    __shared__ __attribute__((aligned(16))) double sbuf[2][kVtNB][64]; // parked summands (one wave per block)
    const uint8_t *sw = sd.vsw + base;
    const double *A0 = sd.A[0], *A1 = sd.A[1], *A2 = sd.A[2], *Bv = sd.bvec;
    double *Gv = sd.g, *Pv = sd.par, *Ov = sd.out;
    const double gv_mean = st.gv_mean[m] * st.gv_weight; // mlpg.rs:135-137
    const double gv_vari = st.gv_var[m];
    const double glen = (double)gvl;
    constexpr uint32_t STEP = 64u * kVtNB;

    // conv_gv (mlpg.rs:195-203)
    double ssum = 0.0;
    for (uint32_t tb = 0; tb < n; tb += STEP) {
        double v[kVtNB];
#pragma unroll
        for (int q = 0; q < kVtNB; q++) {
            const uint32_t t = tb + 64u * q + (uint32_t)lane;
            v[q] = (t < n && sw[t]) ? Pv[IX(t)] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kVtNB; q++)
            sbuf[0][q][lane] = v[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kVtNB; q++)
            if (tb + 64u * q < n)
                ssum = serial_add_lds(ssum, sbuf[0][q]);
    }
    double mean = ssum / glen;
    double vsum = 0.0;
    for (uint32_t tb = 0; tb < n; tb += STEP) {
        double v[kVtNB];
#pragma unroll
        for (int q = 0; q < kVtNB; q++) {
            const uint32_t t = tb + 64u * q + (uint32_t)lane;
            const bool on = t < n && sw[t];
            const double p = on ? Pv[IX(t)] : mean;
            v[q] = on ? (p - mean) * (p - mean) : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kVtNB; q++)
            sbuf[0][q][lane] = v[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kVtNB; q++)
            if (tb + 64u * q < n)
                vsum = serial_add_lds(vsum, sbuf[0][q]);
    }
    double vari = vsum / glen;
    {
        const double ratio = sqrt(gv_mean / vari);
        ssum = 0.0;
        for (uint32_t tb = 0; tb < n; tb += STEP) {
            double v[kVtNB];
#pragma unroll
            for (int q = 0; q < kVtNB; q++) {
                const uint32_t t = tb + 64u * q + (uint32_t)lane;
                const bool on = t < n && sw[t];
                double pn = 0.0;
                if (on) {
                    pn = ratio * (Pv[IX(t)] - mean) + mean;
                    Pv[IX(t)] = pn;
                }
                v[q] = pn;
            }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kVtNB; q++)
            sbuf[0][q][lane] = v[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kVtNB; q++)
            if (tb + 64u * q < n)
                ssum = serial_add_lds(ssum, sbuf[0][q]);
        }
    }
    double step = 0.1, prev = 0.0; // STEPINIT
    const double length = (double)n;
    const double wgt = 1.0 / (double)((uint64_t)sd.W * (uint64_t)n);
    const double ll = (double)((uint64_t)n * (uint64_t)n);
    const double lm1 = (double)(n - 1);
    for (int it = 1; it <= 5; it++) { // GV_MAX_ITERATION
        mean = ssum / glen;
        // ---- sweep V: variance + calc_hmmobj_derivative (mlpg.rs:173-229) ----
        double hmmobj = 0.0;
        vsum = 0.0;
        __syncthreads(); // par written by other lanes in the previous sweep is visible (one wave)
        for (uint32_t tb = 0; tb < n; tb += STEP) {
            double vs[kVtNB], hv[kVtNB];
            // every load of the iteration is unconditional (clamped index) and issued before the
            // arithmetic: loads inside the per-lane edge conditions would each be waited for in turn
            double l_p0[kVtNB], l_pp1[kVtNB], l_pm1[kVtNB], l_pp2[kVtNB], l_pm2[kVtNB], l_a0[kVtNB],
                l_a1[kVtNB], l_a1m[kVtNB], l_a2[kVtNB], l_a2m[kVtNB], l_b[kVtNB];
            uint8_t l_sw[kVtNB];
#pragma unroll
            for (int q = 0; q < kVtNB; q++) {
                const uint32_t t = tb + 64u * q + (uint32_t)lane;
                const uint32_t tc = t < n ? t : n - 1;
                const uint32_t t1 = tc + 1 < n ? tc + 1 : n - 1, t2 = tc + 2 < n ? tc + 2 : n - 1;
                const uint32_t m1 = tc >= 1 ? tc - 1 : 0, m2 = tc >= 2 ? tc - 2 : 0;
                l_p0[q] = Pv[IX(tc)];
                l_pp1[q] = Pv[IX(t1)];
                l_pm1[q] = Pv[IX(m1)];
                l_pp2[q] = Pv[IX(t2)];
                l_pm2[q] = Pv[IX(m2)];
                l_a0[q] = A0[IX(tc)];
                l_a1[q] = A1[IX(tc)];
                l_a1m[q] = A1[IX(m1)];
                l_a2[q] = A2[IX(tc)];
                l_a2m[q] = A2[IX(m2)];
                l_b[q] = Bv[IX(tc)];
                l_sw[q] = sw[tc];
            }
#pragma unroll
            for (int q = 0; q < kVtNB; q++) {
                const uint32_t t = tb + 64u * q + (uint32_t)lane;
                vs[q] = hv[q] = 0.0;
                if (t < n) {
                    const double p0 = l_p0[q];
                    if (l_sw[q])
                        vs[q] = (p0 - mean) * (p0 - mean);
                    double g = l_a0[q] * p0;
                    if (t + 1 < n)
                        g += l_a1[q] * l_pp1[q];
                    if (t >= 1)
                        g += l_a1m[q] * l_pm1[q];
                    if (t + 2 < n)
                        g += l_a2[q] * l_pp2[q];
                    if (t >= 2)
                        g += l_a2m[q] * l_pm2[q];
                    Gv[IX(t)] = g;
                    hv[q] = 1.0 * wgt * p0 * (l_b[q] - 0.5 * g);
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < kVtNB; q++) {
                sbuf[0][q][lane] = vs[q];
                sbuf[1][q][lane] = hv[q];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < kVtNB; q++)
                if (tb + 64u * q < n) {
                    // hmmobj has no switch: frames beyond n contribute nothing (exact: + 0.0)
                    serial_add2_lds(vsum, sbuf[0][q], hmmobj, sbuf[1][q]);
                }
        }
        vari = vsum / glen;
        const double gvobj = -0.5 * 1.0 * vari * gv_vari * (vari - 2.0 * gv_mean);
        const double obj = -(hmmobj + gvobj);
        if (it > 1) {
            if (obj > prev)
                step *= 0.5; // STEPDEC
            else if (obj < prev)
                step *= 1.2; // STEPINC
        }
        // ---- sweep N: next_step (mlpg.rs:230-258) + sum for the next mean ----
        const double dv = -2.0 * gv_vari * (vari - gv_mean) / length;
        ssum = 0.0;
        __syncthreads();
        for (uint32_t tb = 0; tb < n; tb += STEP) {
            double sn[kVtNB];
            double l_p[kVtNB], l_a0[kVtNB], l_g[kVtNB], l_b[kVtNB];
            uint8_t l_sw[kVtNB];
#pragma unroll
            for (int q = 0; q < kVtNB; q++) {
                const uint32_t t = tb + 64u * q + (uint32_t)lane;
                const uint32_t tc = t < n ? t : n - 1;
                l_p[q] = Pv[IX(tc)];
                l_a0[q] = A0[IX(tc)];
                l_g[q] = Gv[IX(tc)];
                l_b[q] = Bv[IX(tc)];
                l_sw[q] = sw[tc];
            }
#pragma unroll
            for (int q = 0; q < kVtNB; q++) {
                const uint32_t t = tb + 64u * q + (uint32_t)lane;
                sn[q] = 0.0;
                if (t < n) {
                    const double p = l_p[q];
                    const double h = -1.0 * wgt * l_a0[q] -
                                     1.0 * 2.0 / ll *
                                         (lm1 * gv_vari * (vari - gv_mean) +
                                          2.0 * gv_vari * (p - mean) * (p - mean));
                    const bool on = l_sw[q] != 0;
                    double next_g;
                    if (on)
                        next_g = 1.0 / h * (1.0 * wgt * (-l_g[q] + l_b[q]) + 1.0 * dv * (p - mean));
                    else
                        next_g = 1.0 / h * (1.0 * wgt * (-l_g[q] + l_b[q]));
                    const double pnew = p + step * next_g;
                    if (NONMSD && it == 5)
                        Ov[IX(t)] = pnew; // scatter fused into the last sweep
                    else
                        Pv[IX(t)] = pnew;
                    if (on)
                        sn[q] = pnew;
                }
            }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kVtNB; q++)
            sbuf[0][q][lane] = sn[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kVtNB; q++)
            if (tb + 64u * q < n)
                ssum = serial_add_lds(ssum, sbuf[0][q]);
        }
        prev = obj;
    }
#undef IX
}

// --------------------------------------------------------------------------
// A8 GV ascent, TIME-PARALLEL (mlpg.rs:145-292) for the [dim][frame] workspace.
// conv_gv and each of the five parmgen iterations are elementwise in t apart from three
// reductions per (utterance, dim): sum and centred sum of squares of the switched-on
// frames, and the HMM objective.  Each launch applies one transform (rescale, or one
// ascent step) to a tile of kGvTT frames (+2 halo frames recomputed on each side), writes
// the tile to the other ping-pong buffer, evaluates the band product W'U^-1W * par_new on
// the tile and leaves its three partial sums in gv_part; the next launch adds the partials
// of all tiles in tile order.  The shape of every sum is fixed by (T, kGvTT) alone, so the
// result is deterministic, but it is NOT the reference's serial order: tracks agree with
// the serial-order kernel to ~1e-14 relative instead of bitwise (JB_BATCH_SERIAL_GV keeps
// the bit-exact path).  The variance uses sums shifted by the previous iteration's mean
// (the ascent moves the mean by ~1e-3 of a standard deviation), which loses nothing
// against the reference's two-pass form.
// One pass over par, A0..A2, bvec per iteration instead of the serial kernel's two sweeps
// at one lane per dim: ~70 GB of traffic over >100k blocks instead of 13 x 25k dependent
// steps on 256 waves.
constexpr int kGvTT = 2048;                      // frames per tile
constexpr int kGvNT = 512;                       // threads per block
constexpr int kGvKX = (kGvTT + 4 + kGvNT - 1) / kGvNT; // per-thread frames incl. halo

struct GvScal {
    double mean, vari, step, obj;
};

// MODE 0: statistics of par (shift K = par[0]);  MODE 1: conv_gv rescale;  MODE 2: ascent
// iteration `it` (1..5).  pass_in/pass_out index gv_part; stats == 0 skips the output sums.
// IN: interior tile -- every frame of the tile, its two halo frames on either side and their
// band neighbours exist, so all edge tests are compile-time true (12 of 13 tiles of a 25.5 k-frame
// row): the kernel is instruction-issue bound (157 VALU + 111 SALU instructions per element-wave
// with the guards), not traffic bound.
template <int MODE, bool IN>
__device__ __forceinline__ void gv_tp_body(const BatchDev &bd, const StreamDev &sd, int si, int it,
                                           const double *__restrict__ src, double *__restrict__ dst,
                                           int stats, const UttDev *up, uint32_t n, uint32_t gvl, int t0)
{
    const int tile = blockIdx.x, m = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const StreamStatesDev st = up->st[si];
    if (!st.gv_mean)
        return;
    const int L = sd.L, B = bd.B;
    const uint32_t NT = sd.gv_ntile;
    const uint64_t row = mt_row0(up, L) + (uint64_t)m * (uint64_t)up->mt_rs;
    const uint8_t *sw = sd.vsw + up->frame_off;
    const double *A0 = sd.A[0] + row, *A1 = sd.A[1] + row, *A2 = sd.A[2] + row, *Bv = sd.bvec + row;
    const double *P = src + row;
    double *Q = dst ? dst + row : nullptr;

    // ---- every global load of the block is issued before anything waits on one ----
    // window arrays (frame t0 - 4 + i): par_old, A1, A2; own frames (t0 - 2 + i): A0, bvec, switch
    constexpr int KW = (kGvTT + 8 + kGvNT - 1) / kGvNT;
    // (the off-diagonals A1[t-1], A2[t-2] a frame needs besides its own come straight from global
    // memory -- the neighbouring lanes load the same lines -- instead of through LDS windows:
    // half the LDS footprint, twice the blocks per CU)
    double pw[KW], a0r[kGvKX], a1r[kGvKX], a2r[kGvKX], a1m[kGvKX], a2m[kGvKX], br[kGvKX];
    bool onr[kGvKX];
    double p_own[kGvTT / kGvNT];
    bool on_own[kGvTT / kGvNT];
    if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < kGvTT / kGvNT; k++) {
            const uint32_t t = (uint32_t)t0 + (uint32_t)(tid + kGvNT * k);
            const bool ok = IN || t < n;
            p_own[k] = ok ? P[t] : 0.0;
            on_own[k] = ok && sw[t] != 0;
        }
    } else {
#pragma unroll
        for (int k = 0; k < KW; k++) {
            const int i = tid + kGvNT * k;
            const int t = t0 - 4 + i;
            const bool ok = i < kGvTT + 8 && (IN || (t >= 0 && (uint32_t)t < n));
            pw[k] = ok ? P[t] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < kGvKX; k++) {
            const int i = tid + kGvNT * k;
            const int t = t0 - 2 + i;
            const bool ok = i < kGvTT + 4 && (IN || (t >= 0 && (uint32_t)t < n));
            a0r[k] = ok ? A0[t] : 0.0;
            a1r[k] = ok ? A1[t] : 0.0;
            a2r[k] = ok ? A2[t] : 0.0;
            a1m[k] = (ok && (IN || t >= 1)) ? A1[t - 1] : 0.0;
            a2m[k] = (ok && (IN || t >= 2)) ? A2[t - 2] : 0.0;
            br[k] = ok ? Bv[t] : 0.0;
            onr[k] = ok && sw[t] != 0;
        }
    }

    const double gv_mean = st.gv_mean[m] * st.gv_weight; // mlpg.rs:135-137
    const double gv_vari = st.gv_var[m];
    const double glen = (double)gvl;
    const uint32_t ntile_b = (n + kGvTT - 1) / kGvTT;
    const uint64_t bl = (uint64_t)b * L + m;
    auto part = [&](int pass) { return sd.gv_part + (((uint64_t)pass * B * L + bl) * NT) * 4; };
    auto scal = [&](int k) { return sd.gv_scal + ((uint64_t)k * B * L + bl) * 4; };

    // ---- scalars of this launch from the previous launch's partial sums (tile order) ----
    double K = 0.0, mean = 0.0, vari = 0.0, ratio = 1.0, step = 0.0;
    if (MODE == 0) {
        K = P[0];
    } else {
        const double *pp = part(MODE == 1 ? 0 : it);
        double S1 = 0.0, S2 = 0.0, H = 0.0;
        for (uint32_t j = 0; j < ntile_b; j++) {
            S1 += pp[4 * j + 0];
            S2 += pp[4 * j + 1];
            H += pp[4 * j + 2];
        }
        const double Kprev = (MODE == 1) ? P[0] : scal(it - 1)[0];
        mean = Kprev + S1 / glen;
        vari = (S2 - S1 * S1 / glen) / glen;
        K = mean; // shift of this launch's output statistics
        if (MODE == 1) {
            ratio = sqrt(gv_mean / vari);
            if (tile == 0 && tid == 0) {
                double *sc = scal(0);
                sc[0] = mean;
                sc[1] = 0.1; // STEPINIT
                sc[2] = 0.0;
            }
        } else {
            const double gvobj = -0.5 * 1.0 * vari * gv_vari * (vari - 2.0 * gv_mean);
            const double obj = -(H + gvobj);
            step = 0.1;
            if (it > 1) {
                const double prev = scal(it - 1)[2];
                step = scal(it - 1)[1];
                if (obj > prev)
                    step *= 0.5; // STEPDEC
                else if (obj < prev)
                    step *= 1.2; // STEPINC
            }
            if (tile == 0 && tid == 0) {
                double *sc = scal(it);
                sc[0] = mean;
                sc[1] = step;
                sc[2] = obj;
            }
        }
    }
    const double wgt = 1.0 / (double)((uint64_t)sd.W * (uint64_t)n);

    double s1 = 0.0, s2 = 0.0, hh = 0.0;
    if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < kGvTT / kGvNT; k++) {
            if (on_own[k]) {
                const double dlt = p_own[k] - K;
                s1 += dlt;
                s2 += dlt * dlt;
            }
        }
    } else {
        // LDS windows: po = par_old[t0-4 .. t0+TT+4), pn = par_new[t0-2 .. t0+TT+2)
        __shared__ double po[kGvTT + 8], pn[kGvTT + 4];
#pragma unroll
        for (int k = 0; k < KW; k++) {
            const int i = tid + kGvNT * k;
            if (i < kGvTT + 8)
                po[i] = pw[k];
        }
        __syncthreads();
        const double length = (double)n;
        const double ll = (double)((uint64_t)n * (uint64_t)n);
        const double lm1 = (double)(n - 1);
        const double dv = -2.0 * gv_vari * (vari - gv_mean) / length;
        // ---- transform: par_new on the tile and two halo frames each side ----
#pragma unroll
        for (int k = 0; k < kGvKX; k++) {
            const int i = tid + kGvNT * k; // index into pn; frame t = t0 - 2 + i
            const int t = t0 - 2 + i;
            if (i < kGvTT + 4 && (IN || (t >= 0 && (uint32_t)t < n))) {
                const double p = po[i + 2];
                const bool on = onr[k];
                const double a0 = a0r[k], bb = br[k];
                double pnew;
                if (MODE == 1) {
                    pnew = on ? ratio * (p - mean) + mean : p; // conv_gv (mlpg.rs:195-203)
                } else {
                    // calc_hmmobj_derivative (mlpg.rs:205-229), the reference's order of additions
                    double g = a0 * p;
                    if (IN || (uint32_t)t + 1 < n)
                        g += a1r[k] * po[i + 3];
                    if (IN || t >= 1)
                        g += a1m[k] * po[i + 1];
                    if (IN || (uint32_t)t + 2 < n)
                        g += a2r[k] * po[i + 4];
                    if (IN || t >= 2)
                        g += a2m[k] * po[i];
                    // next_step (mlpg.rs:230-258)
                    const double h = -1.0 * wgt * a0 -
                                     1.0 * 2.0 / ll *
                                         (lm1 * gv_vari * (vari - gv_mean) +
                                          2.0 * gv_vari * (p - mean) * (p - mean));
                    double next_g;
                    if (on)
                        next_g = 1.0 / h * (1.0 * wgt * (-g + bb) + 1.0 * dv * (p - mean));
                    else
                        next_g = 1.0 / h * (1.0 * wgt * (-g + bb));
                    pnew = p + step * next_g;
                }
                pn[i] = pnew;
            }
        }
        __syncthreads();
        // ---- store the tile; statistics of par_new for the next launch ----
#pragma unroll
        for (int k = 0; k < kGvKX; k++) {
            const int i = tid + kGvNT * k;
            const int t = t0 - 2 + i;
            if (i >= 2 && i < kGvTT + 2 && (IN || (uint32_t)t < n)) {
                const double p0 = pn[i];
                Q[t] = p0;
                if (stats) {
                    if (onr[k]) {
                        const double dlt = p0 - K;
                        s1 += dlt;
                        s2 += dlt * dlt;
                    }
                    double g = a0r[k] * p0;
                    if (IN || (uint32_t)t + 1 < n)
                        g += a1r[k] * pn[i + 1];
                    if (IN || t >= 1)
                        g += a1m[k] * pn[i - 1];
                    if (IN || (uint32_t)t + 2 < n)
                        g += a2r[k] * pn[i + 2];
                    if (IN || t >= 2)
                        g += a2m[k] * pn[i - 2];
                    hh += 1.0 * wgt * p0 * (br[k] - 0.5 * g);
                }
            }
        }
    }
    if (!stats)
        return;
    // ---- block sums: xor butterfly inside each wave, then the four waves in order ----
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
        hh += __shfl_xor(hh, o);
    }
    __shared__ double red[kGvNT / 64][3];
    if ((tid & 63) == 0) {
        red[tid >> 6][0] = s1;
        red[tid >> 6][1] = s2;
        red[tid >> 6][2] = hh;
    }
    __syncthreads();
    if (tid == 0) {
        double r0 = red[0][0], r1 = red[0][1], r2 = red[0][2];
#pragma unroll
        for (int w = 1; w < kGvNT / 64; w++) {
            r0 += red[w][0];
            r1 += red[w][1];
            r2 += red[w][2];
        }
        double *po_ = part(MODE == 0 ? 0 : (MODE == 1 ? 1 : it + 1)) + 4 * (uint64_t)tile;
        po_[0] = r0;
        po_[1] = r1;
        po_[2] = r2;
    }
}

template <int MODE>
__global__ __launch_bounds__(kGvNT) void k_mlpg_gv_tp(BatchDev bd, StreamDev sd, int si, int it,
                                                       const double *__restrict__ src,
                                                       double *__restrict__ dst, int stats)
{
    const int b = blockIdx.z;
    const UttDev *up = bd.utt + b;
    const uint32_t n = sd.Tv[b], gvl = sd.gvlen[b];
    const int t0 = blockIdx.x * kGvTT;
    if (n == 0 || gvl == 0 || (uint32_t)t0 >= n)
        return;
    if (t0 >= 4 && (uint32_t)t0 + kGvTT + 4 <= n)
        gv_tp_body<MODE, true>(bd, sd, si, it, src, dst, stats, up, n, gvl, t0);
    else
        gv_tp_body<MODE, false>(bd, sd, si, it, src, dst, stats, up, n, gvl, t0);
}

// A9 for the [dim][frame] workspace: `out` is [frame][dim], so this is a tiled transpose
// (plus the MSD compaction and NODATA fill of mask.rs:34-49).  Block = 64 frames x L dims.
__global__ __launch_bounds__(256) void k_mlpg_scatter_mt(BatchDev bd, StreamDev sd)
{
    extern __shared__ double tile[]; // [L][65]
    const int b = blockIdx.y;
    const UttDev *up = bd.utt + b;
    const uint32_t T = up->T;
    const uint32_t t0 = blockIdx.x * 64u;
    if (t0 >= T)
        return;
    const uint64_t base = up->frame_off, sb = up->state_off;
    const int L = sd.L;
    for (int e = threadIdx.x; e < 64 * L; e += blockDim.x) {
        const int m = e >> 6, tl = e & 63;
        const uint32_t t = t0 + (uint32_t)tl;
        double v = kNoData;
        if (t < T && sd.voiced[base + t]) {
            uint32_t k = t;
            if (sd.is_msd) {
                const uint32_t s_ = sd.fstate[base + t];
                k = sd.s_vpre[sb + s_] + (t - sd.s_start[sb + s_]);
            }
            v = sd.par[mt_row0(up, L) + (uint64_t)m * up->mt_rs + k];
        }
        tile[m * 65 + tl] = v;
    }
    __syncthreads();
    const uint32_t nt = T - t0 < 64u ? T - t0 : 64u;
    for (int e = threadIdx.x; e < (int)nt * L; e += blockDim.x) {
        const int tl = e / L, m = e % L;
        sd.out[(base + t0 + (uint32_t)tl) * (uint64_t)L + (uint64_t)m] = tile[m * 65 + tl];
    }
}

// A9 Mask::fill with NODATA (mask.rs:34-49, mod.rs:89-91) for MSD streams:
// thread per (frame, dim); compacted index = s_vpre[state] + offset within state.
__global__ void k_mlpg_scatter(BatchDev bd, StreamDev sd)
{
    const int b = blockIdx.y;
    const UttDev *up = bd.utt + b;
    const uint32_t T = up->T;
    const uint64_t base = up->frame_off, sb = up->state_off;
    const int L = sd.L;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (uint64_t)T * L)
        return;
    const uint32_t t = (uint32_t)(tid / L);
    const int m = (int)(tid % L);
    double v = kNoData;
    if (sd.voiced[base + t]) {
        const uint32_t s_ = sd.fstate[base + t];
        const uint32_t k = sd.s_vpre[sb + s_] + (t - sd.s_start[sb + s_]);
        v = sd.par[(base + k) * (uint64_t)L + (uint64_t)m];
    }
    sd.out[(base + t) * (uint64_t)L + (uint64_t)m] = v;
}

hipError_t launch_prep(const BatchDev &bd, const StreamDev &sd, int si, hipStream_t stream)
{
    if (bd.B == 0)
        return hipSuccess;
    {
        dim3 grid(bd.B), block(64);
        if (!sd.is_msd)
            hipLaunchKernelGGL(k_prep_states_dense, grid, block, 0, stream, bd, sd, si);
        else
            hipLaunchKernelGGL(k_prep_states, grid, block, 0, stream, bd, sd, si);
    }
    if (bd.maxS > 0) {
        dim3 grid((bd.maxS + 127) / 128, bd.B), block(128);
        hipLaunchKernelGGL(k_prep_frames, grid, block, 0, stream, bd, sd, si);
    }
    return hipGetLastError();
}

// F/B sweeps of the [dim][frame] workspace: LDS-staged block per utterance (L <= kMtMaxDim = 60 dims:
// 2688 B of LDS per dim fit the CU's 160 KB)
template <int LMAX>
static void launch_fb_lds(const BatchDev &bd, const StreamDev &sd, int si, size_t lds, hipStream_t stream)
{
    // once per instantiation; a function-local static's initialisation is thread-safe
    static const hipError_t attr = hipFuncSetAttribute((const void *)k_mlpg_fb_lds<LMAX>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)attr;
    hipLaunchKernelGGL(k_mlpg_fb_lds<LMAX>, dim3(bd.B), dim3(kFlNT), lds, stream, bd, sd, si);
}

static void launch_fb(const BatchDev &bd, const StreamDev &sd, int si, hipStream_t stream)
{
    // per dim of the tile pitch: input slots of four arrays, output slots of at most three (L1, L2, g/D)
    constexpr size_t lds1 = sizeof(double) * (size_t)(kFlIn * 4 + kFlOut * 3) * kFlCT;
    static_assert(lds1 * kMtMaxDim <= 160 * 1024, "the widest [dim][frame] stream must fit the LDS of a CU");
    if (sd.L == 1)
        launch_fb_lds<1>(bd, sd, si, lds1, stream);
    else if (sd.L <= 36)
        launch_fb_lds<36>(bd, sd, si, lds1 * 37, stream); // tile pitch (fl_pass)
    else
        launch_fb_lds<64>(bd, sd, si, lds1 * (size_t)sd.L, stream); // up to 160 KB: the whole LDS of a CU
}

template <int BW>
static hipError_t launch_mlpg_bw(const BatchDev &bd, const StreamDev &sd, int si, hipStream_t stream,
                                 hipEvent_t after_build, jb_enqueue_hook &between, void *between_ctx,
                                 hipEvent_t after_ivar)
{
    const uint64_t work = (uint64_t)bd.maxT * (uint64_t)sd.L;
    if (work == 0 || bd.B == 0)
        return hipSuccess;
    if constexpr (BW == 3) if (sd.mt) {
        // [dim][frame] workspace: build, lane-per-dim factor/substitution sweeps, GV, transpose
        {
            const uint64_t ne = (uint64_t)bd.maxS * (uint64_t)(sd.W * sd.L);
            constexpr uint64_t per = 256u * kIvarPer;
            dim3 grid((unsigned)((ne + per - 1) / per), bd.B), block(256);
            hipLaunchKernelGGL(k_mlpg_ivar, grid, block, 0, stream, bd, sd, si);
            if (after_ivar)
                (void)hipEventRecord(after_ivar, stream);
        }
        {
            // sliding-window loads for up to three windows (StreamDev::mt requires W <= 3, L <= 60: 256 threads)
            dim3 grid((bd.maxT + kBuildTF - 1) / kBuildTF, bd.B);
            const size_t lds = sizeof(double) * (size_t)(BW + 1) * sd.L * (kBuildTF + 1);
            const int nthr = ((kBuildTF / kBuildRun) * sd.L + 63) / 64 * 64;
            JB_DBG_SKIP_IF(si == 0 ? 1 : 0, hipLaunchKernelGGL(k_mlpg_build_mt2<BW>, grid, dim3((unsigned)nthr), lds, stream, bd, sd, si));
        }
        if (after_build)
            (void)hipEventRecord(after_build, stream);
        dim3 grid((sd.L + 63) / 64, bd.B), block(64);
        const bool tp = sd.use_gv && !sd.serial_gv && sd.gv_part;
        if (sd.is_msd) {
            if (tp)
                launch_fb(bd, sd, si, stream);
            else
                hipLaunchKernelGGL((k_mlpg_solve3<false, true, true>), grid, block, 0, stream, bd, sd, si);
        } else {
            if (tp)
                JB_DBG_SKIP_IF(si == 0 ? 2 : 0, launch_fb(bd, sd, si, stream));
            else
                hipLaunchKernelGGL((k_mlpg_solve3<true, true, true>), grid, block, 0, stream, bd, sd, si);
        }
        if (between) {
            hipError_t he = between(between_ctx, stream);
            if (he != hipSuccess)
                return he;
            between = nullptr;
        }
        if (tp && sd.gv_gang_ctl) {
            // resident form: one persistent launch, band matrix in registers (jb_gv_gang.hip)
            hipError_t he = hipSuccess;
            JB_DBG_SKIP_IF(16, he = launch_gv_gang(bd, sd, si, stream));
            if (he != hipSuccess)
                return he;
        } else if (tp) {
            // par -> g -> par -> ... : conv_gv + five iterations = six writes, result back in par
            dim3 gg(sd.gv_ntile, sd.L, bd.B), gb(kGvNT);
            hipLaunchKernelGGL(k_mlpg_gv_tp<0>, gg, gb, 0, stream, bd, sd, si, 0, sd.par, (double *)nullptr, 1);
            hipLaunchKernelGGL(k_mlpg_gv_tp<1>, gg, gb, 0, stream, bd, sd, si, 0, sd.par, sd.g, 1);
            for (int it = 1; it <= 5; it++) { // GV_MAX_ITERATION
                const double *src = (it & 1) ? sd.g : sd.par;
                double *dst = (it & 1) ? sd.par : sd.g;
                hipLaunchKernelGGL(k_mlpg_gv_tp<2>, gg, gb, 0, stream, bd, sd, si, it, src, dst, it < 5 ? 1 : 0);
            }
        }
        if (!sd.defer_out) {
            dim3 g2((bd.maxT + 63) / 64, bd.B), b2(256);
            hipLaunchKernelGGL(k_mlpg_scatter_mt, g2, b2, sizeof(double) * (size_t)sd.L * 65, stream, bd, sd);
        }
        return hipGetLastError();
    }
    {
        dim3 grid((unsigned)((work + 255) / 256), bd.B), block(256);
        hipLaunchKernelGGL(k_mlpg_build<BW>, grid, block, 0, stream, bd, sd, si);
    }
    if (after_build)
        (void)hipEventRecord(after_build, stream);
    {
        dim3 grid((sd.L + 63) / 64, bd.B), block(64);
        if (BW == 3 && !sd.generic_solver) {
            // F/B recurrences lane-per-dim; the GV ascent with lanes over time (one wave per dim)
            const bool gv_vt = sd.use_gv && sd.L <= 2; // frames contiguous per lane only for tiny L
            dim3 gvgrid(sd.L, bd.B);
            if (sd.is_msd) {
                if (gv_vt) {
                    // L == 1: [frame][1] is [1][frame]; one lane per voiced run (the compacted system is block
                    // diagonal: same bits as a sweep over the whole utterance)
                    if (sd.L == 1 && bd.maxS > 0) {
                        dim3 rg((bd.maxS + 63) / 64, bd.B);
                        JB_DBG_SKIP_IF(128, hipLaunchKernelGGL(k_mlpg_fb_runs, rg, dim3(64), 0, stream, bd, sd, si));
                    } else if (sd.L == 1)
                        launch_fb(bd, sd, si, stream);
                    else
                        hipLaunchKernelGGL((k_mlpg_solve3<false, false>), grid, block, 0, stream, bd, sd, si);
                    JB_DBG_SKIP_IF(32, hipLaunchKernelGGL(k_mlpg_gv_vt<false>, gvgrid, block, 0, stream, bd, sd, si));
                } else {
                    hipLaunchKernelGGL((k_mlpg_solve3<false, true>), grid, block, 0, stream, bd, sd, si);
                }
                dim3 g2((unsigned)((work + 255) / 256), bd.B), b2(256);
                hipLaunchKernelGGL(k_mlpg_scatter, g2, b2, 0, stream, bd, sd);
            } else if (gv_vt) {
                hipLaunchKernelGGL((k_mlpg_solve3<true, false>), grid, block, 0, stream, bd, sd, si);
                hipLaunchKernelGGL(k_mlpg_gv_vt<true>, gvgrid, block, 0, stream, bd, sd, si);
            } else {
                hipLaunchKernelGGL((k_mlpg_solve3<true, true>), grid, block, 0, stream, bd, sd, si);
            }
        } else {
            hipLaunchKernelGGL(k_mlpg_solve<BW>, grid, block, 0, stream, bd, sd, si);
        }
    }
    return hipGetLastError();
}

// A9 + V2 (mask.rs:34-49, cepstrum.rs:139-149) for the non-MSD MCP stream on the [dim][frame]
// workspace: a block turns 64 frames x L dims of `par` in LDS, optionally writes the
// [frame][dim] track, runs b[m] = c[m] - alpha*b[m+1] down each frame's column and writes
// bcoef -- one pass over par instead of transpose (read+write) followed by mc2b (read+write).
__global__ __launch_bounds__(256) void k_mc2b_mt(BatchDev bd, StreamDev sd, VocDev vd, int write_out)
{
    extern __shared__ double tile[]; // [L][65]
    const int b = blockIdx.y;
    const UttDev *up = bd.utt + b;
    const uint32_t T = up->T;
    const uint32_t t0 = blockIdx.x * 64u;
    if (t0 >= T)
        return;
    const uint64_t base = up->frame_off;
    const int L = sd.L;
    const uint32_t nt = T - t0 < 64u ? T - t0 : 64u;
    for (int e = threadIdx.x; e < 64 * L; e += blockDim.x) {
        const int m = e >> 6, tl = e & 63;
        tile[m * 65 + tl] = (uint32_t)tl < nt ? sd.par[mt_row0(up, L) + (uint64_t)m * up->mt_rs + t0 + (uint32_t)tl] : 0.0;
    }
    __syncthreads();
    const uint64_t o0 = (base + t0) * (uint64_t)L;
    if (write_out)
        for (int e = threadIdx.x; e < (int)nt * L; e += blockDim.x)
            sd.out[o0 + (uint64_t)e] = tile[(e % L) * 65 + e / L];
    __syncthreads();
    if (threadIdx.x < nt && vd.alpha != 0.0) {
        const int tl = threadIdx.x;
        double prev = tile[(L - 1) * 65 + tl];
        for (int i = L - 2; i >= 0; i--) {
            prev = tile[i * 65 + tl] - vd.alpha * prev;
            tile[i * 65 + tl] = prev;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < (int)nt * L; e += blockDim.x)
        vd.bcoef[o0 + (uint64_t)e] = tile[(e % L) * 65 + e / L];
}

hipError_t launch_mc2b_mt(const BatchDev &bd, const StreamDev &sd, const VocDev &vd, bool write_out,
                          hipStream_t stream)
{
    if (bd.B == 0 || bd.maxT == 0)
        return hipSuccess;
    dim3 grid((bd.maxT + 63) / 64, bd.B), block(256);
    hipLaunchKernelGGL(k_mc2b_mt, grid, block, sizeof(double) * (size_t)sd.L * 65, stream, bd, sd, vd,
                       write_out ? 1 : 0);
    return hipGetLastError();
}

// --------------------------------------------------------------------------
// per-state Gaussians from pdf row indices -- gather from the device-resident
// tables of each voice and blend with the interpolation weights, in the reference's order
// (VoiceSet weighted sum, voice_set.rs:80-95: first * w0, then += w_i * param_i).  Thread per
// (state, element); a state's row is WL contiguous floats, so reads and writes are coalesced.
constexpr double kHalfToneMinLf0 = 2.995732273553991;  // ln 20     (stream_parameter.rs:8-9)
constexpr double kHalfToneMaxLf0 = 9.903487552536127;  // ln 20000
__global__ __launch_bounds__(256) void k_gather_blend(const GatherJob *__restrict__ jobs)
{
    const GatherJob &j = jobs[blockIdx.y];
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)j.S * j.WL)
        return;
    const uint32_t s = (uint32_t)(e / j.WL), k = (uint32_t)(e % j.WL);
    const float *p0 = j.tab[0] + (uint64_t)j.row[0][s] * j.row_len;
    double m = (double)p0[k] * j.w[0], v = (double)p0[j.WL + k] * j.w[0];
    double q = (j.has_msd && k == 0) ? (double)p0[2 * j.WL] * j.w[0] : 0.0;
    for (uint32_t vi = 1; vi < j.nv; vi++) {
        const float *p = j.tab[vi] + (uint64_t)j.row[vi][s] * j.row_len;
        m += j.w[vi] * (double)p[k];
        v += j.w[vi] * (double)p[j.WL + k];
        if (j.has_msd && k == 0)
            q += j.w[vi] * (double)p[2 * j.WL];
    }
    if (j.lf0_offset != 0.0 && k == 0) {
        const double x = m + j.lf0_offset;
        m = fmin(fmax(x, kHalfToneMinLf0), kHalfToneMaxLf0);
    }
    j.mean[e] = m;
    j.var[e] = v;
    if (j.has_msd && k == 0 && j.msd)
        j.msd[s] = q;
}

hipError_t launch_gather(const GatherJob *jobs_dev, uint32_t n_jobs, uint64_t max_elems, hipStream_t stream)
{
    if (n_jobs == 0 || max_elems == 0)
        return hipSuccess;
    dim3 grid((unsigned)((max_elems + 255) / 256), n_jobs), block(256);
    hipLaunchKernelGGL(k_gather_blend, grid, block, 0, stream, jobs_dev);
    return hipGetLastError();
}

int mlpg_mt_max_dim() { return kMtMaxDim; }
int mlpg_gv_tile_frames() { return kGvTT; }

static hipError_t launch_mlpg_inner(const BatchDev &bd, const StreamDev &sd, int si, hipStream_t stream,
                                    hipEvent_t after_build, jb_enqueue_hook &between, void *between_ctx,
                                    hipEvent_t after_ivar);

hipError_t launch_mlpg(const BatchDev &bd, const StreamDev &sd, int si, hipStream_t stream, hipEvent_t after_build,
                       jb_enqueue_hook between, void *between_ctx, hipEvent_t after_ivar)
{
    hipError_t e = launch_mlpg_inner(bd, sd, si, stream, after_build, between, between_ctx, after_ivar);
    if (e == hipSuccess && between) // a path without a distinct GV phase: run the hook at the end
        e = between(between_ctx, stream);
    return e;
}

static hipError_t launch_mlpg_inner(const BatchDev &bd, const StreamDev &sd, int si, hipStream_t stream,
                                    hipEvent_t after_build, jb_enqueue_hook &between, void *between_ctx,
                                    hipEvent_t after_ivar)
{
    if (after_ivar && !(sd.BW == 3 && sd.mt)) // no inverse-variance pass on this path: "done" from the start
        (void)hipEventRecord(after_ivar, stream);
    if (sd.BW == 1 && sd.W == 1 && !sd.use_gv && !sd.generic_solver) {
        const uint64_t work = (uint64_t)bd.maxT * (uint64_t)sd.L;
        if (work == 0 || bd.B == 0)
            return hipSuccess;
        dim3 grid((unsigned)((work + 255) / 256), bd.B), block(256);
        if (sd.canon && sd.canon_n && sd.L <= 32) {
            dim3 grid_rows((bd.maxT + 7) / 8, bd.B);
            JB_DBG_SKIP_IF(256, hipLaunchKernelGGL(k_mlpg_static_rows, grid_rows, block, 0, stream, bd, sd, si));
            if (after_build)
                (void)hipEventRecord(after_build, stream);
            return hipGetLastError();
        }
        if (sd.canon && sd.canon_n) {
            hipError_t e = hipMemsetAsync(sd.canon, 1, sd.canon_n, stream);
            if (e != hipSuccess)
                return e;
        }
        JB_DBG_SKIP_IF(256, hipLaunchKernelGGL(k_mlpg_static, grid, block, 0, stream, bd, sd, si));
        if (after_build)
            (void)hipEventRecord(after_build, stream);
        return hipGetLastError();
    }
    switch (sd.BW) {
    case 1:
        return launch_mlpg_bw<1>(bd, sd, si, stream, after_build, between, between_ctx, after_ivar);
    case 3:
        return launch_mlpg_bw<3>(bd, sd, si, stream, after_build, between, between_ctx, after_ivar);
    case 5:
        return launch_mlpg_bw<5>(bd, sd, si, stream, after_build, between, between_ctx, after_ivar);
    case 7: // (round 6: the reference takes any window width, window.rs:19-56; HTS voices use 3, a few 5)
        return launch_mlpg_bw<7>(bd, sd, si, stream, after_build, between, between_ctx, after_ivar);
    case 9:
        return launch_mlpg_bw<9>(bd, sd, si, stream, after_build, between, between_ctx, after_ivar);
    default:
        return hipErrorInvalidValue;
    }
}

} // namespace jb
