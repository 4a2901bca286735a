// jb_gv_gang.hip -- A8 (GV: conv_gv + five ascent iterations, mlpg.rs:145-292) for the
// [dim][frame] workspace as ONE persistent launch with the band matrix resident in registers.
//
// The time-parallel sweeps of k_mlpg_gv_tp re-stream A0..A2, b and par from HBM for every one of
// their seven launches (66 GB for BASELINE config 2) because a launch boundary is their only way to
// add up the three sums a (utterance, dim) row needs per iteration.  Here a row is handed to a GANG
// of NT workgroups (NT = tiles of the longest row, 7 for 25.5 k frames), each keeping its 3,904
// frames of A0, A1, A2, b and par in registers (8 frames per thread) through all six phases:
//     traffic  = read A0..A2, b, par once + write par once     (11 GB instead of 66)
//     per row  = 6 exchanges of {S1, S2, H} between the NT workgroups of the gang through
//                global memory (agent-scope `sc1` stores / loads and one arrival counter)
// The only cross-tile coupling of the algorithm is those sums (mlpg.rs:173-258); the band product
// couples neighbouring frames, which is handled without any exchange: every WAVE owns 488 frames
// and carries 12 halo frames on either side, which it updates redundantly (an iteration invalidates
// two more halo frames per side; five iterations need ten).  Redundant frames see the same operands
// and operations as their owner's copy, so they are bit-identical to it.
//
// Arithmetic: the reference's expressions in the reference's order (FP contraction off), as in
// k_mlpg_gv_tp; the sums have a fixed shape (8 frames per lane ascending, DPP tree over the
// wave, waves in order, tiles in order), so the result is deterministic and independent of which
// gang took the row, but -- like k_mlpg_gv_tp's -- not the reference's serial order (~1e-15
// relative; JB_BATCH_SERIAL_GV keeps the bit-exact kernel).
//
// Liveness (no launch of this kernel can hang, whatever else runs on the device):
//   * workgroups take TICKETS when they start; ticket q joins gang q / NT as tile q % NT, so a gang
//     is made of workgroups that are already running -- except the last, incomplete one;
//   * a gang starts only once all NT members have arrived.  Members of a gang that is not complete
//     leave as soon as the row queue is empty (they poison the gang's counter by compare-and-swap
//     so that late arrivals leave too); complete gangs drain the queue in finite time, so they do;
//   * rows come from a queue (one atomic per row by the gang's first tile, handed to the others in
//     the record of an exchange the row needs anyway): an incomplete gang owns no work;
//   * once formed, all members of a gang are resident and every exchange completes;
//   * every spin is bounded all the same: on overrun the kernel raises `err` and leaves.  That can
//     happen without any fault when SEVERAL of these launches share a device (more than two batches in
//     flight, or one device listed several times in a *_multi entry): each may hold CUs with incomplete
//     gangs in formation while none owns all its members.  The host then runs that batch's GV as the
//     multi-launch sweeps instead (Batch::sync: the step is enqueued again without the resident kernel,
//     for this batch from then on); it reports JB_ERR_DEVICE only if that fails too.
#include "jb_device.h"

#include <cstddef>
#include <cstdlib>

namespace jb {

#ifndef JB_GG_XCHG
#define JB_GG_XCHG 3 // 2: records polled directly (NTg pollers per workgroup); 3: counter poll, then the records
#endif
#ifndef JB_GG_SC
#define JB_GG_SC "sc1" // agent scope, what __hip_atomic_load/store(relaxed, agent) lower to
#endif
#ifndef JB_GG_SLEEP
#define JB_GG_SLEEP 2
#endif
#ifndef JB_GG_WPS
#define JB_GG_WPS 2 // waves per SIMD the register budget is set for: 4 = two workgroups per CU (128 VGPRs: the
                    // kernel spills and is slower, 12.2 ms against 9.1), 2 = one, which also leaves room for
                    // the kernels of the other chains beside it
#endif
#ifndef JB_GG_FENCE
#define JB_GG_FENCE 0 // 1: frames of a lane one after the other (the scheduler otherwise interleaves all eight
                      // and needs ~90 more VGPRs for their temporaries)
#endif
#if JB_GG_FENCE
#define JB_GG_FRAME_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define JB_GG_FRAME_FENCE (void)0
#endif
#ifndef JB_GG_A0_LDS
#define JB_GG_A0_LDS 0 // 1: A0 stays in a second LDS image per wave as well (16 VGPRs less)
#endif
// JB_GG_PROFILE (jb_device.h; 0 in the product) = 1: thread 0 of every workgroup adds its shader-clock ticks per section to ctl->prof
#if JB_GG_PROFILE
#define GG_T(k)                                                                                    \
    do {                                                                                           \
        const long long now_ = clock64();                                                          \
        prof_[k] += now_ - tprev_;                                                                 \
        tprev_ = now_;                                                                             \
    } while (0)
#else
#define GG_T(k) (void)0
#endif
#ifndef JB_GG_KEEPG
#define JB_GG_KEEPG (JB_GG_WPS <= 2) // band product kept in 16 VGPRs across the exchange instead of evaluated twice
#endif
#ifndef JB_GG_FPT
#define JB_GG_FPT 8
#endif
constexpr int kGgFPT = JB_GG_FPT;                    // frames per thread (contiguous, even)
constexpr int kGgWin = 64 * kGgFPT;                  // frames per wave window: 512
constexpr int kGgHalo = 12;                          // >= 2 frames per ascent iteration x 5, rounded to 4
constexpr int kGgOwn = kGgWin - 2 * kGgHalo;         // frames a wave owns: 488
#ifndef JB_GG_WAVES
#define JB_GG_WAVES 8
#endif
constexpr int kGgWaves = JB_GG_WAVES;
constexpr int kGgNT = 64 * kGgWaves;                 // threads per workgroup
constexpr int kGgBlockOwn = kGgWaves * kGgOwn;       // frames a workgroup owns: 3,904
constexpr uint32_t kGgPoison = 1u << 24;             // added to a gang counter that will never complete
constexpr uint32_t kGgSpinLimit = 1u << 23;          // polls (each >= 0.2 us) before the kernel gives up

int gv_gang_block_frames() { return kGgBlockOwn; }
int gv_gang_max_tiles() { return kGvGangMaxTiles; }
size_t gv_gang_ctl_bytes(int n_gangs) { return sizeof(GvGangCtl) + sizeof(GvGang) * (size_t)n_gangs; }

template <int CTRL> __device__ __forceinline__ double gg_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// sum over the 64 lanes with DPP (row_shr 1/2/4/8, row_bcast 15, row_bcast 31): the total is in lane 63.
// A fixed shape, like the xor butterfly it replaces, without its twelve LDS round trips per value.
__device__ __forceinline__ double gg_wave_sum63(double v)
{
    v += gg_dpp<0x111>(v);
    v += gg_dpp<0x112>(v);
    v += gg_dpp<0x114>(v);
    v += gg_dpp<0x118>(v);
    v += gg_dpp<0x142>(v);
    v += gg_dpp<0x143>(v);
    return v;
}
constexpr int GG_WAVE_SHL1 = 0x130, GG_WAVE_SHR1 = 0x138; // lane i <- lane i+1 / lane i-1 (0 at the wave's end)

// agent-scope relaxed accesses: `sc1` loads / stores (L2 of the issuing XCD bypassed for the
// line's coherence; MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility")
__device__ __forceinline__ uint32_t gg_ld32(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double gg_ld64(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gg_st64(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 16-byte granule accesses with the sc1 bit (agent-coherent: write-through past / read around the
// XCD's L2 state of the line), one instruction each
__device__ __forceinline__ void gg_st128(unsigned long long *p, unsigned long long a, unsigned long long b)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    v.x = (unsigned int)a;
    v.y = (unsigned int)(a >> 32);
    v.z = (unsigned int)b;
    v.w = (unsigned int)(b >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off " JB_GG_SC : : "v"(p), "v"(v) : "memory");
}
// the four granules of one record: four loads in flight, one wait
__device__ __forceinline__ void gg_ld_rec(const unsigned long long *p, unsigned long long (&d)[4],
                                          unsigned long long (&c)[4])
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v0, v1, v2, v3;
    asm volatile("global_load_dwordx4 %0, %4, off " JB_GG_SC "\n\t"
                 "global_load_dwordx4 %1, %4, off offset:16 " JB_GG_SC "\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32 " JB_GG_SC "\n\t"
                 "global_load_dwordx4 %3, %4, off offset:48 " JB_GG_SC "\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                 : "v"(p)
                 : "memory");
    const u32x4 v[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        d[k] = (unsigned long long)v[k].x | ((unsigned long long)v[k].y << 32);
        c[k] = (unsigned long long)v[k].z | ((unsigned long long)v[k].w << 32);
    }
}

// wave-uniform values read through LDS or a lane-0 result: tell the compiler (SGPRs, scalar loads)
__device__ __forceinline__ double gg_uni(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// 1/d as the compiler's IEEE division expands it (v_rcp_f64, two Newton steps, quotient, residual
// correction) minus v_div_scale / v_div_fixup: d is a sum of inverse-variance terms and a GV term,
// far from the ends of the exponent range (same reasoning as fb_rcp / fb_div in jb_mlpg.hip).
__device__ __forceinline__ double gg_recip(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}

// LDS image of a wave window: 512 doubles, lane i's eight frames in row i (64 B) with the four 16-byte
// granules of a row XOR-swizzled by (i >> 2) & 3, so that both the coalesced side (lane = frame,
// ds_write_b64) and the thread-contiguous side (8 frames per lane, ds_read_b128: lanes i, i+4, i+8, i+12
// of a 16-lane pass would otherwise share their four banks) spread over all banks without padding.
static_assert(kGgFPT == 8, "the image layout is written for eight frames per lane");
__device__ __forceinline__ int gg_gran(int row, int j) { return 4 * row + (j ^ ((row >> 2) & 3)); } // double2 index
__device__ __forceinline__ int gg_idx(int w) { return 2 * gg_gran(w >> 3, (w >> 1) & 3) + (w & 1); }
constexpr int kGgXs = kGgWin; // doubles per wave image

__device__ __forceinline__ void gg_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// coalesced load of the wave window [ws, ws + 512) of a row, lane = frame (zero outside [0, n)).
// The loads are unconditional at clamped addresses and selected afterwards: a load nested in the
// bounds test becomes a branch of its own and the 40 loads of a row would be waited for one by one.
template <bool INNER>
__device__ __forceinline__ void gg_fetch(double (&r)[kGgFPT], const double *__restrict__ arr, int ws, int n, int lane)
{
    if (INNER) {
        // the whole window lies inside the row: one address, eight immediate offsets
        const double *q = arr + (ws + lane);
#pragma unroll
        for (int j = 0; j < kGgFPT; j++)
            r[j] = q[64 * j];
        return;
    }
#pragma unroll
    for (int j = 0; j < kGgFPT; j++) {
        const int t = ws + 64 * j + lane;
        const int tc = t < 0 ? 0 : (t >= n ? n - 1 : t);
        const double v = arr[tc];
        r[j] = (t == tc) ? v : 0.0;
    }
}
// lane = frame  ->  8 contiguous frames per lane, in place through the wave's LDS image
__device__ __forceinline__ void gg_turn(double (&r)[kGgFPT], double *xs, int lane)
{
    gg_wave_sync(); // the previous array's reads of the image are done
#pragma unroll
    for (int j = 0; j < kGgFPT; j++)
        xs[gg_idx(64 * j + lane)] = r[j];
    gg_wave_sync();
    const double2 *q = reinterpret_cast<const double2 *>(xs);
#pragma unroll
    for (int j = 0; j < kGgFPT / 2; j++) {
        const double2 d2 = q[gg_gran(lane, j)];
        r[2 * j] = d2.x;
        r[2 * j + 1] = d2.y;
    }
}

__global__ __launch_bounds__(kGgNT, JB_GG_WPS) void k_mlpg_gv_gang(BatchDev bd, StreamDev sd, int si, GvGangCtl *ctl,
                                                            GvGang *gangs, int NTg, int n_gangs)
{
    __shared__ double xs_all[kGgWaves][kGgXs];
#if JB_GG_A0_LDS
    __shared__ double ys_all[kGgWaves][kGgXs];
#endif
    __shared__ double red[2][kGgWaves][4]; // two slots: a wave may write the next phase's sums while wave 0 adds up
    __shared__ double recs[kGvGangMaxTiles][4];
    __shared__ int sh_i[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double *xs = xs_all[wv];
    const uint32_t total_rows = sd.gv_nbins * (uint32_t)sd.L; // (bin, dim) groups of rows

    // ---- ticket -> (gang, tile) ----
    if (tid == 0)
        sh_i[0] = (int)atomicAdd(&ctl->tickets, 1u);
    __syncthreads();
    const int ticket = sh_i[0];
    const int gang = ticket / NTg, tile = ticket % NTg;
    if (gang >= n_gangs)
        return;
    GvGang *G = gangs + gang;

    // ---- formation: all NTg members have started, or the gang is given up ----
    if (tid == 0) {
        atomicAdd(&G->cnt, 1u);
        int state = -1;
        for (uint32_t spins = 0; state < 0; spins++) {
            const uint32_t c = gg_ld32(&G->cnt);
            if (c >= kGgPoison)
                state = 0;
            else if (c >= (uint32_t)NTg)
                state = 1;
            else if (gg_ld32(&ctl->next_row) >= total_rows) {
                // nothing left to do and the gang is not complete: leave, and make late arrivals leave
                if (atomicCAS(&G->cnt, c, c + kGgPoison) == c)
                    state = 0;
            } else if (spins > kGgSpinLimit) {
                atomicExch(&ctl->err, 1u);
                atomicAdd(&G->cnt, kGgPoison);
                state = 0;
            } else {
                __builtin_amdgcn_s_sleep(8);
            }
        }
        sh_i[1] = state;
    }
    __syncthreads();
    if (!sh_i[1])
        return;

    // ---- exchange of one 4-double record per tile between the members of the gang ----
    // A record is four self-validating 16-byte granules {value, bits(value) ^ T(exchange, k)}: the
    // producer stores them with sc1 stores and is done; a consumer lane loads the four granules of
    // one tile with sc1 loads and accepts them when all four checks hold, else polls again.  A stale
    // granule (previous use of the slot) and a torn one (halves of two different stores) both fail the
    // check, so the hand-off relies on no ordering between stores, no flag and no atomic: one store
    // propagation plus one load round trip per exchange instead of store, wait, counter add, counter
    // poll, record loads.  Records alternate between two slots: a member can run at most one exchange
    // ahead of the slowest reader of the previous one (it needs that reader's record to go on).
#if JB_GG_PROFILE
    long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // 0 row setup+load, 1 statistics, 2 block sum, 3 exchange, 4 step, 5 store
    long long tprev_ = clock64();
#endif
    uint32_t kbar = 0;
    bool dead = false;
    // from_red: the first three values are the workgroup sums of block_sum (added up here, waves in order)
    auto exchange = [&](bool from_red, double v0, double v1, double v2, double v3) {
#if JB_GG_PROFILE
        if (tid == 0 && tile < 8 && kbar < 96)
            G->stamp[0][tile][kbar] = __builtin_amdgcn_s_memrealtime();
#endif
        __syncthreads(); // every thread is done with the records of the previous exchange; red[] is written
#if JB_GG_PROFILE
        const long long tx0_ = clock64();
#endif
#if JB_GG_XCHG == 0
        // timing experiment only (wrong sums): no traffic between the workgroups, every tile stands for all
        if (wv == 0) {
            if (lane == 0 && from_red) {
                v0 = red[kbar & 1][0][0];
                v1 = red[kbar & 1][0][1];
                v2 = red[kbar & 1][0][2];
                for (int w = 1; w < kGgWaves; w++) {
                    v0 += red[kbar & 1][w][0];
                    v1 += red[kbar & 1][w][1];
                    v2 += red[kbar & 1][w][2];
                }
            }
            v0 = gg_uni(v0); v1 = gg_uni(v1); v2 = gg_uni(v2); v3 = gg_uni(v3);
            if (lane < NTg) {
                recs[lane][0] = v0; recs[lane][1] = v1; recs[lane][2] = v2; recs[lane][3] = v3;
            }
            if (lane == 0)
                sh_i[0] = 1;
        }
        if (false) {
            typedef unsigned long long u64;
            unsigned long long *slot = &G->rec[kbar & 1][0][0];
#else
        if (wv == 0) {
            typedef unsigned long long u64;
            unsigned long long *slot = &G->rec[kbar & 1][0][0];
#endif
            const u64 tagbase = 0x9E3779B97F4A7C15ull * (u64)(kbar + 1u);
            if (lane == 0) {
                if (from_red) {
                    v0 = red[kbar & 1][0][0];
                    v1 = red[kbar & 1][0][1];
                    v2 = red[kbar & 1][0][2];
#pragma unroll
                    for (int w = 1; w < kGgWaves; w++) {
                        v0 += red[kbar & 1][w][0];
                        v1 += red[kbar & 1][w][1];
                        v2 += red[kbar & 1][w][2];
                    }
                }
                const double v[4] = {v0, v1, v2, v3};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const u64 bits = (u64)__double_as_longlong(v[k]);
                    gg_st128(slot + 8 * tile + 2 * k, bits, bits ^ (tagbase + (u64)k));
                }
            }
            bool ok = true;
#if JB_GG_XCHG == 3
            // arrival by counter (one dword polled per workgroup instead of NTg records: pollers are
            // uncached loads that every memory channel sees), data by granules: the counter add needs no
            // wait for the stores because a granule that has not landed yet fails its check and is polled again
            if (lane == 0)
                atomicAdd(&G->cnt, 1u);
            {
                const uint32_t target = (uint32_t)NTg * (kbar + 2u); // formation + (kbar + 1) exchanges
                uint32_t spins = 0;
                while (gg_ld32(&G->cnt) < target) {
                    if (++spins > kGgSpinLimit) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(JB_GG_SLEEP);
                }
            }
#endif
            if (lane < NTg && ok) {
                const unsigned long long *src = slot + 8 * lane;
                uint32_t spins = 0;
                for (;;) {
                    u64 d[4], c[4];
                    gg_ld_rec(src, d, c);
                    bool good = true;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        good = good && ((d[k] ^ c[k]) == tagbase + (u64)k);
                    if (good) {
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            recs[lane][k] = __longlong_as_double((long long)d[k]);
                        break;
                    }
                    if (++spins > kGgSpinLimit) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(JB_GG_SLEEP);
                }
            }
            const bool all_ok = __all(ok);
            if (!all_ok && lane == 0)
                atomicExch(&ctl->err, 1u);
            if (lane == 0)
                sh_i[0] = all_ok ? 1 : 0;
        }
#if JB_GG_PROFILE
        prof_[6] += clock64() - tx0_; // the hand-off itself (wave 0), without the wait for the workgroup's other waves
#endif
        __syncthreads();
#if JB_GG_PROFILE
        if (tid == 0 && tile < 8 && kbar < 96)
            G->stamp[1][tile][kbar] = __builtin_amdgcn_s_memrealtime();
#endif
        dead = sh_i[0] == 0;
        kbar++;
    };
    // the three sums of an exchange: the tiles of THIS workgroup's row, in order (its idle tiles contributed zeros)
    int sum_t0 = 0, sum_t1 = 0;
    auto gather = [&](double &S1, double &S2, double &H) {
        S1 = S2 = H = 0.0;
        for (int j = sum_t0; j < sum_t1; j++) {
            S1 += recs[j][0];
            S2 += recs[j][1];
            H += recs[j][2];
        }
        S1 = gg_uni(S1);
        S2 = gg_uni(S2);
        H = gg_uni(H);
    };
    // workgroup sums: DPP sum inside each wave, the per-wave sums through LDS; wave 0 adds them
    // in wave order at the head of the exchange (one barrier for both)
    auto block_sum = [&](double &a, double &b2, double &c) {
        a = gg_wave_sum63(a);
        b2 = gg_wave_sum63(b2);
        c = gg_wave_sum63(c);
        if (lane == 63) {
            red[kbar & 1][wv][0] = a;
            red[kbar & 1][wv][1] = b2;
            red[kbar & 1][wv][2] = c;
        }
    };

    // ---- first row of the gang ----
    double nxt = 0.0;
    if (tile == 0 && tid == 0)
        nxt = (double)atomicAdd(&ctl->next_row, 1u);
    exchange(false, 0.0, 0.0, 0.0, nxt);
    if (dead)
        return;
    uint32_t row = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)recs[0][3]);
#if JB_GG_XCHG == 0
    row = (uint32_t)gang; // (timing experiment: no queue, gang g takes rows g, g + n_gangs, ...)
#endif

    // Everything a row needs from the descriptor tables (a chain of four dependent scalar loads: launch
    // order -> utterance -> stream states -> GV pdf) is fetched one row ahead, while the current row's
    // phases run, so that a row starts with its vector loads.
    struct RowInfo {
        uint32_t n, gvl;
        int m;
        bool live;   // this workgroup's row takes part in GV (the reference returns early otherwise)
        bool active; // some row of the group may: the group runs its six exchanges
        int k, t0, nt; // tile of the row this workgroup is, the row's tiles in the gang
        uint64_t rowoff, frame_off;
        double gv_mean, gv_vari;
    };
    auto row_info = [&](uint32_t r) -> RowInfo {
        RowInfo ri{};
        if (r >= total_rows)
            return ri;
        const GvBinEntry en = sd.gv_bins[(uint64_t)(r / (uint32_t)sd.L) * (uint64_t)NTg + (uint64_t)tile];
        ri.active = (en.flags & 1u) != 0;
        ri.k = en.k;
        ri.t0 = en.t0;
        ri.nt = en.nt;
        if (en.b == 0xffffffffu)
            return ri; // an idle tile of this bin
        const int b = (int)en.b;
        ri.m = (int)(r % (uint32_t)sd.L);
        const UttDev *up = bd.utt + b;
        ri.n = sd.Tv[b];
        ri.gvl = sd.gvlen[b];
        const StreamStatesDev st = up->st[si];
        ri.live = st.gv_mean != nullptr && ri.n > 0 && ri.gvl > 0; // the same answer in every tile of the row
        ri.frame_off = up->frame_off;
        ri.rowoff = mt_row0(up, sd.L) + (uint64_t)ri.m * (uint64_t)up->mt_rs;
        if (ri.live) {
            ri.gv_mean = st.gv_mean[ri.m] * st.gv_weight; // mlpg.rs:135-137
            ri.gv_vari = st.gv_var[ri.m];
        }
        return ri;
    };
    RowInfo cur = row_info(row), ahead{};

#if JB_GG_XCHG == 0
    uint32_t row_prev_ = row;
#endif
    while (row < total_rows) {
#if JB_GG_XCHG == 0
        row_prev_ = row;
#endif
        const uint32_t n = cur.n, gvl = cur.gvl;
        // the next row's number rides on this row's first exchange
        nxt = 0.0;
        if (tile == 0 && tid == 0)
            nxt = (double)atomicAdd(&ctl->next_row, 1u);
        if (!cur.active) { // (the same answer in every member: from the bin table)
            exchange(false, 0.0, 0.0, 0.0, nxt);
            if (dead)
                return;
            row = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)recs[0][3]);
#if JB_GG_XCHG == 0
            row = row_prev_ + (uint32_t)n_gangs;
            row_prev_ = row;
#endif
            cur = row_info(row);
            continue;
        }
        sum_t0 = cur.t0;
        sum_t1 = cur.t0 + cur.nt;
        const uint64_t rowoff = cur.live ? cur.rowoff : 0; // (a workgroup without a live row touches no memory)
        const double *A0 = sd.A[0] + rowoff, *A1 = sd.A[1] + rowoff, *A2 = sd.A[2] + rowoff, *Bv = sd.bvec + rowoff;
        double *P = sd.par + rowoff;
        const uint8_t *sw = sd.vsw + cur.frame_off;
        // (frame indices are 32-bit: a row has < 2^31 frames.  The opaque copy of `lane` keeps the
        // compiler from hoisting every per-lane index and bounds test of the row out of the row loop
        // and then spilling them: they cost a few integer instructions per row)
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));
        const int n_i = (int)n;
        const int own_lo = cur.k * kGgBlockOwn + wv * kGgOwn; // first frame this wave owns
        const bool busy = cur.live && own_lo < n_i;           // else: an idle tile / wave of a short row
        const int ws = own_lo - kGgHalo;                     // first frame of the wave window
        const int t0 = ws + kGgFPT * lane_v;                 // this lane's first frame

        // A0..A2 and par live in registers; b (used twice per iteration, never changed) stays in the
        // wave's LDS image, where the load left it: 16 VGPRs less, which is what keeps the kernel from
        // spilling at four waves per SIMD
        double a0[kGgFPT], a1[kGgFPT], a2[kGgFPT], p[kGgFPT];
        const double2 *bq2 = reinterpret_cast<const double2 *>(xs);
#define GG_B(f) (((f)&1) ? bq2[gg_gran(lane_v, (f) >> 1)].y : bq2[gg_gran(lane_v, (f) >> 1)].x)
#if JB_GG_A0_LDS
        double *ys = ys_all[wv];
        const double2 *aq2 = reinterpret_cast<const double2 *>(ys);
#define GG_A0(f) (((f)&1) ? aq2[gg_gran(lane_v, (f) >> 1)].y : aq2[gg_gran(lane_v, (f) >> 1)].x)
#else
#define GG_A0(f) a0[f]
#endif
        double a1m = 0.0, a2m = 0.0, a2mm = 0.0; // A1[t0-1], A2[t0-1], A2[t0-2]
        uint32_t onbits = 0, ownbits = 0;
        if (busy) {
            double btmp[kGgFPT];
            if (ws >= 0 && ws + kGgWin <= n_i) {
                // interior window (all but the first and last wave of a row): all 40 loads of the row
                // are in flight before the first is waited for
                gg_fetch<true>(a0, A0, ws, n_i, lane_v);
                gg_fetch<true>(a1, A1, ws, n_i, lane_v);
                gg_fetch<true>(a2, A2, ws, n_i, lane_v);
                gg_fetch<true>(p, P, ws, n_i, lane_v);
                gg_fetch<true>(btmp, Bv, ws, n_i, lane_v);
#if JB_GG_A0_LDS
                gg_turn(a0, ys, lane_v);
#else
                gg_turn(a0, xs, lane_v);
#endif
                gg_turn(a1, xs, lane_v);
                gg_turn(a2, xs, lane_v);
                gg_turn(p, xs, lane_v);
            } else {
                // window over an end of the row (the first wave of a row's first tile, the last waves of its last):
                // clamped addresses, and -- round 5 -- all 40 loads in flight before the first turn, like the interior
                // windows.  Fetched and turned array by array (a turn's fences keep the next array's loads behind it)
                // this window took FIVE memory round trips where the others take one: the row's first tile reached
                // its first exchange 3-9 us behind the rest of the gang, which waited for it (stamps of
                // -DJB_GG_PROFILE builds, profiles/r05_gv_gang_arrivals.txt)
                gg_fetch<false>(a0, A0, ws, n_i, lane_v);
                gg_fetch<false>(a1, A1, ws, n_i, lane_v);
                gg_fetch<false>(a2, A2, ws, n_i, lane_v);
                gg_fetch<false>(p, P, ws, n_i, lane_v);
                gg_fetch<false>(btmp, Bv, ws, n_i, lane_v);
#if JB_GG_A0_LDS
                gg_turn(a0, ys, lane_v);
#else
                gg_turn(a0, xs, lane_v);
#endif
                gg_turn(a1, xs, lane_v);
                gg_turn(a2, xs, lane_v);
                gg_turn(p, xs, lane_v);
            }
            gg_turn(btmp, xs, lane_v); // leaves b in the image; its registers are dropped here
            a1m = gg_dpp<GG_WAVE_SHR1>(a1[kGgFPT - 1]);
            a2m = gg_dpp<GG_WAVE_SHR1>(a2[kGgFPT - 1]);
            a2mm = gg_dpp<GG_WAVE_SHR1>(a2[kGgFPT - 2]);
            const int own_hi = own_lo + kGgOwn < n_i ? own_lo + kGgOwn : n_i;
            // the eight GV switches of the lane's frames: unconditional loads at clamped positions, all in
            // flight together (behind the bounds test each was a basic block of its own and waited for
            // alone -- eight memory round trips per row and wave)
            uint8_t swb[kGgFPT];
#pragma unroll
            for (int f = 0; f < kGgFPT; f++) {
                const int t = t0 + f;
                swb[f] = sw[t < 0 ? 0 : (t >= n_i ? n_i - 1 : t)];
            }
#pragma unroll
            for (int f = 0; f < kGgFPT; f++) {
                const int t = t0 + f;
                const bool in = t >= 0 && t < n_i;
                if (in && swb[f] != 0)
                    onbits |= 1u << f;
                if (t >= own_lo && t < own_hi)
                    ownbits |= 1u << f;
            }
        } else {
#pragma unroll
            for (int f = 0; f < kGgFPT; f++)
                a0[f] = a1[f] = a2[f] = p[f] = 0.0;
        }
        GG_T(0);
        const double gv_mean = cur.gv_mean, gv_vari = cur.gv_vari;
        const double glen = (double)gvl;
        const double wgt = 1.0 / (double)((uint64_t)sd.W * (uint64_t)n);
        const double length = (double)n;
        const double ll = (double)((uint64_t)n * (uint64_t)n);
        const double lm1 = (double)(n - 1);

        // ---- six phases, one exchange each.  Phase 0: statistics of the solved track (shift K =
        // par[0]), then conv_gv (mlpg.rs:195-203).  Phases 1..5: the ascent iterations
        // (GV_MAX_ITERATION, mlpg.rs:260-292): statistics and HMM objective of the current track, then
        // the step.  One loop body for all six, so that the exchange is expanded once.
        double K = cur.live ? gg_uni(P[0]) : 0.0;
        double step = 0.1, prev = 0.0; // STEPINIT
        uint32_t next_row = total_rows;
#pragma unroll 1
        for (int it = 0; it <= 5; it++) {
            // band neighbours across lanes (the track changes every phase)
            const double pl2 = gg_dpp<GG_WAVE_SHR1>(p[kGgFPT - 2]), pl1 = gg_dpp<GG_WAVE_SHR1>(p[kGgFPT - 1]);
            const double pr1 = gg_dpp<GG_WAVE_SHL1>(p[0]), pr2 = gg_dpp<GG_WAVE_SHL1>(p[1]);
            double s1 = 0.0, s2 = 0.0, hh = 0.0;
#if JB_GG_KEEPG
            double g[kGgFPT];
#endif
            if (busy) { // an idle wave of a short row only takes part in the exchanges
                double o2 = pl2, o1 = pl1;
#pragma unroll
                for (int f = 0; f < kGgFPT; f++) {
                    const double pf = p[f];
                    const bool own = ownbits >> f & 1u;
                    if (own && (onbits >> f & 1u)) {
                        const double dlt = pf - K;
                        s1 += dlt;
                        s2 += dlt * dlt;
                    }
                    if (it > 0 && (own || JB_GG_KEEPG)) {
                        // calc_hmmobj_derivative (mlpg.rs:205-229), the reference's order of additions;
                        // operands outside [0, n) are zeros, which leaves the sum bit-identical
                        const double pp1 = f + 1 < kGgFPT ? p[f + 1 < kGgFPT ? f + 1 : 0] : pr1;
                        const double pp2 =
                            f + 2 < kGgFPT ? p[f + 2 < kGgFPT ? f + 2 : 0] : (f + 2 == kGgFPT ? pr1 : pr2);
                        const double am1 = f >= 1 ? a1[f >= 1 ? f - 1 : 0] : a1m;
                        const double am2 = f >= 2 ? a2[f >= 2 ? f - 2 : 0] : (f == 1 ? a2m : a2mm);
                        double gg = GG_A0(f) * pf;
                        gg += a1[f] * pp1;
                        gg += am1 * o1;
                        gg += a2[f] * pp2;
                        gg += am2 * o2;
#if JB_GG_KEEPG
                        g[f] = gg;
#endif
                        if (own)
                            hh += 1.0 * wgt * pf * (GG_B(f) - 0.5 * gg);
                    }
                    o2 = o1;
                    o1 = pf;
                    JB_GG_FRAME_FENCE;
                }
            }
            GG_T(1);
            block_sum(s1, s2, hh);
            GG_T(2);
            exchange(true, s1, s2, hh, nxt);
            GG_T(3);
            if (dead)
                return;
            if (it == 0) {
                next_row = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)recs[0][3]);
#if JB_GG_XCHG == 0
                next_row = row + (uint32_t)n_gangs;
#endif
                ahead = row_info(next_row);
            }
            double S1, S2, H;
            gather(S1, S2, H);
            const double mean = K + S1 / glen;
            const double vari = (S2 - S1 * S1 / glen) / glen;
            if (it == 0) {
                const double ratio = sqrt(gv_mean / vari);
#pragma unroll
                for (int f = 0; f < kGgFPT; f++)
                    if (onbits >> f & 1u)
                        p[f] = ratio * (p[f] - mean) + mean;
            } else {
                const double gvobj = -0.5 * 1.0 * vari * gv_vari * (vari - 2.0 * gv_mean);
                const double obj = -(H + gvobj);
                if (it > 1) {
                    if (obj > prev)
                        step *= 0.5; // STEPDEC
                    else if (obj < prev)
                        step *= 1.2; // STEPINC
                }
                prev = obj;
                // next_step (mlpg.rs:230-258); the band product again (cheaper than keeping it in 16
                // VGPRs across the exchange), in place and ascending: the two old values to the left of
                // frame f are carried along (o2, o1), the ones to its right have not been touched yet
                const double dv = -2.0 * gv_vari * (vari - gv_mean) / length;
                if (busy) {
#if !JB_GG_KEEPG
                    double o2 = pl2, o1 = pl1;
#endif
#pragma unroll
                    for (int f = 0; f < kGgFPT; f++) {
                        const double pf = p[f];
#if JB_GG_KEEPG
                        const double gg = g[f];
#else
                        const double pp1 = f + 1 < kGgFPT ? p[f + 1 < kGgFPT ? f + 1 : 0] : pr1;
                        const double pp2 =
                            f + 2 < kGgFPT ? p[f + 2 < kGgFPT ? f + 2 : 0] : (f + 2 == kGgFPT ? pr1 : pr2);
                        const double am1 = f >= 1 ? a1[f >= 1 ? f - 1 : 0] : a1m;
                        const double am2 = f >= 2 ? a2[f >= 2 ? f - 2 : 0] : (f == 1 ? a2m : a2mm);
                        double gg = GG_A0(f) * pf;
                        gg += a1[f] * pp1;
                        gg += am1 * o1;
                        gg += a2[f] * pp2;
                        gg += am2 * o2;
#endif
                        const double bf = GG_B(f);
                        const double h = -1.0 * wgt * GG_A0(f) -
                                         1.0 * 2.0 / ll *
                                             (lm1 * gv_vari * (vari - gv_mean) +
                                              2.0 * gv_vari * (pf - mean) * (pf - mean));
                        const double rh = gg_recip(h);
                        double next_g;
                        if (onbits >> f & 1u)
                            next_g = rh * (1.0 * wgt * (-gg + bf) + 1.0 * dv * (pf - mean));
                        else
                            next_g = rh * (1.0 * wgt * (-gg + bf));
#if !JB_GG_KEEPG
                        o2 = o1;
                        o1 = pf;
#endif
                        p[f] = pf + step * next_g;
                        JB_GG_FRAME_FENCE;
                    }
                }
            }
            K = mean;
            GG_T(4);
        }

        // ---- store the frames this wave owns: back through the LDS image, coalesced ----
        if (busy) {
            gg_wave_sync();
            double2 *q = reinterpret_cast<double2 *>(xs);
#pragma unroll
            for (int j = 0; j < kGgFPT / 2; j++)
                q[gg_gran(lane_v, j)] = make_double2(p[2 * j], p[2 * j + 1]);
            gg_wave_sync();
            const int own_hi = own_lo + kGgOwn < n_i ? own_lo + kGgOwn : n_i;
#pragma unroll
            for (int j = 0; j < kGgFPT; j++) {
                const int t = ws + 64 * j + lane_v;
                if (t >= own_lo && t < own_hi)
                    P[t] = xs[gg_idx(64 * j + lane_v)];
            }
        }
        row = next_row;
        cur = ahead;
        GG_T(5);
    }
#if JB_GG_PROFILE
    if (tid == 0)
        for (int k = 0; k < 8; k++)
            atomicAdd(&ctl->prof[k], (unsigned long long)prof_[k]);
#endif
}

// Workgroups that fit the device at once for this kernel (what the register budget of its launch
// bounds is for; the occupancy query can only lower it).
static int gv_gang_capacity(int device)
{
    static int cap[16] = {0};
    if (device < 0 || device >= 16)
        return 0;
    if (cap[device])
        return cap[device];
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0)
        return 0;
    constexpr int want = JB_GG_WPS * 256 / kGgNT; // workgroups per CU the launch bounds are for
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_mlpg_gv_gang, kGgNT, 0) != hipSuccess)
        return 0;
    // (a process that holds TWO HIP runtimes -- this library's loaded before torch's own -- was seen to answer 0 here for
    // every kernel, with hipSuccess, and every batch then ran the multi-launch sweeps without a word: round 6.  The
    // launch bounds are what guarantees `want` workgroups per CU; the query is only allowed to lower that.)
    if (per_cu <= 0 || per_cu > want)
        per_cu = want;
    cap[device] = per_cu * cus;
    return cap[device];
}

int gv_gang_plan(int device, uint32_t maxT, uint32_t n_rows, int *tiles_per_gang, int *n_gangs)
{
    const int cap = gv_gang_capacity(device);
    const uint32_t nt = (maxT + kGgBlockOwn - 1) / kGgBlockOwn;
    if (cap <= 0 || nt == 0 || nt > (uint32_t)kGvGangMaxTiles || (int)nt > cap || n_rows == 0)
        return 0;
    int g = cap / (int)nt;
    if ((uint32_t)g > n_rows)
        g = (int)n_rows;
    *tiles_per_gang = (int)nt;
    *n_gangs = g;
    return 1;
}

void gv_gang_bins(const uint32_t *T, const uint8_t *has_gv, const uint32_t *order, size_t n, int tiles_per_gang,
                  std::vector<GvBinEntry> &out)
{
    out.clear();
    std::vector<int> used; // tiles taken per bin
    const int cap = tiles_per_gang;
    auto new_bin = [&]() {
        used.push_back(0);
        out.resize(out.size() + (size_t)cap, GvBinEntry{0xffffffffu, 0, 0, 0, 0});
        return used.size() - 1;
    };
    for (size_t oi = 0; oi < n; oi++) {
        const uint32_t b = order ? order[oi] : (uint32_t)oi;
        int nt = (int)((T[b] + (uint32_t)kGgBlockOwn - 1) / (uint32_t)kGgBlockOwn);
        if (nt < 1)
            nt = 1; // (an empty utterance: one idle tile, so that its rows are still "visited")
        if (nt > cap)
            nt = cap;
        size_t bin = used.size();
        for (size_t j = 0; j < used.size(); j++)
            if (used[j] + nt <= cap) {
                bin = j;
                break;
            }
        if (bin == used.size())
            bin = new_bin();
        const int t0 = used[bin];
        for (int k = 0; k < nt; k++) {
            GvBinEntry &en = out[bin * (size_t)cap + (size_t)(t0 + k)];
            en.b = b;
            en.k = (uint8_t)k;
            en.t0 = (uint8_t)t0;
            en.nt = (uint8_t)nt;
        }
        used[bin] += nt;
        if (has_gv[b] && T[b] > 0)
            for (int j = 0; j < cap; j++)
                out[bin * (size_t)cap + (size_t)j].flags |= 1u;
    }
    // (flags were set on the entries that existed when an utterance joined: once more over whole bins)
    for (size_t bin = 0; bin < used.size(); bin++) {
        uint8_t f = 0;
        for (int j = 0; j < cap; j++)
            f |= out[bin * (size_t)cap + (size_t)j].flags;
        for (int j = 0; j < cap; j++)
            out[bin * (size_t)cap + (size_t)j].flags = f;
    }
}

hipError_t launch_gv_gang(const BatchDev &bd, const StreamDev &sd, int si, hipStream_t stream)
{
    if (!sd.gv_gang_ctl || sd.gv_gang_n <= 0 || bd.B == 0 || !sd.gv_bins || sd.gv_nbins == 0)
        return hipErrorInvalidValue;
    GvGangCtl *ctl = (GvGangCtl *)sd.gv_gang_ctl;
    GvGang *gangs = (GvGang *)(ctl + 1);
    // tickets, row queue and gang counters start from zero on every launch; `err` is STICKY: with
    // run(); run(); sync() a timeout of the first run must still be there when the host looks
    // (Batch::sync clears it when it has acted on it)
    static_assert(offsetof(GvGangCtl, err) == 8 && sizeof(((GvGangCtl *)nullptr)->err) == 4, "layout of GvGangCtl");
    hipError_t e = hipMemsetAsync(ctl, 0, offsetof(GvGangCtl, err), stream);
    if (e == hipSuccess)
        e = hipMemsetAsync((uint8_t *)ctl + offsetof(GvGangCtl, err) + 4, 0,
                           gv_gang_ctl_bytes(sd.gv_gang_n) - offsetof(GvGangCtl, err) - 4, stream);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(k_mlpg_gv_gang, dim3((unsigned)(sd.gv_gang_n * sd.gv_gang_tiles)), dim3(kGgNT), 0, stream, bd, sd, si,
                       ctl, gangs, sd.gv_gang_tiles, sd.gv_gang_n);
    return hipGetLastError();
}

} // namespace jb
