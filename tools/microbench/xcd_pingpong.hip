// xcd_pingpong.hip -- how long does a word take from one workgroup to another, by the way it is written and polled
// and by whether the two workgroups sit on the same XCD?  (Question behind the resident GV kernel's exchange:
// DESIGN section 4.)  Two one-wave workgroups on different CUs bounce a sequence number kRounds times; the round
// trip is timed with s_memrealtime (100 MHz).  Every poll loop is bounded: a protocol that never sees the other
// side's store reports "no" instead of hanging the card.
//
//   hipcc --offload-arch=gfx950 -O2 -o xcd_pingpong xcd_pingpong.hip && ./xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

constexpr int kRounds = 2000;
constexpr uint32_t kSpin = 1u << 16;

struct Ctl {
    uint32_t ticket[8];    // per XCD: workgroups of that XCD that have registered
    uint32_t role_xcd[2];  // XCD of side 0 / side 1
    uint32_t taken[2];     // side claimed
    uint32_t fail;
    uint32_t pad0[19];
    uint64_t ticks;        // side 0: ticks of kRounds round trips
    uint32_t cu[2];        // HW_ID of both sides
    uint32_t pad1[12];
    alignas(256) uint32_t flag0[64]; // written by side 0
    alignas(256) uint32_t flag1[64]; // written by side 1
};

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x;
}
__device__ __forceinline__ uint32_t hw_id()
{
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(x));
    return x;
}

// MODE 0: sc1 store, sc1 vector load (agent scope: the product's protocol)
// MODE 1: plain store, scalar load after s_dcache_inv (through the XCD's L2)
// MODE 2: plain store, scalar load with glc
// MODE 3: plain store, sc1 vector load
// MODE 4: plain store, sc0 vector load
// MODE 5: plain store, sc0 sc1 vector load (system scope)
// MODE 6: sc1 store, scalar load after s_dcache_inv
// MODE 7: plain store, vector load after buffer_inv sc1
// MODE 8: sc0 sc1 store, sc0 sc1 vector load
template <int MODE> __device__ __forceinline__ void put(uint32_t *p, uint32_t v)
{
    if (MODE == 0 || MODE == 6)
        asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : : "v"(p), "v"(v) : "memory");
    else if (MODE == 8)
        asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : : "v"(p), "v"(v) : "memory");
    else
        asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : : "v"(p), "v"(v) : "memory");
}
template <int MODE> __device__ __forceinline__ uint32_t get(const uint32_t *p)
{
    uint32_t r;
    if (MODE == 0 || MODE == 3)
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    else if (MODE == 4)
        asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    else if (MODE == 5 || MODE == 8)
        asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    else if (MODE == 7)
        asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    else if (MODE == 2) {
        uint32_t s;
        asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(s) : "s"(p) : "memory");
        r = s;
    } else {
        uint32_t s;
        asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s) : "s"(p) : "memory");
        r = s;
    }
    return r;
}

template <int MODE> __global__ void __launch_bounds__(64) k_pingpong(Ctl *c, int same_xcd)
{
    extern __shared__ double hold[]; // sized by the host so that a CU takes ONE workgroup
    if (threadIdx.x == 0)
        hold[0] = 0.0;
    const uint32_t x = xcc_id();
    int side = -1;
    if (threadIdx.x == 0) {
        // side 0: the first workgroup to come; side 1: the first workgroup of the same / of another XCD after it
        if (atomicCAS(&c->taken[0], 0u, 1u) == 0u) {
            __hip_atomic_store(&c->role_xcd[0], x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            side = 0;
        } else {
            uint32_t x0 = 0;
            for (uint32_t s = 0; s < kSpin && !x0; s++)
                x0 = __hip_atomic_load(&c->role_xcd[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x0 && ((x0 - 1 == x) == (same_xcd != 0)) && atomicCAS(&c->taken[1], 0u, 1u) == 0u) {
                __hip_atomic_store(&c->role_xcd[1], x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                side = 1;
            }
        }
    }
    side = __builtin_amdgcn_readfirstlane(side);
    if (side < 0 || threadIdx.x != 0)
        return;
    c->cu[side] = hw_id();
    uint32_t *mine = side == 0 ? c->flag0 : c->flag1;
    const uint32_t *theirs = side == 0 ? c->flag1 : c->flag0;
    // side 0 waits until side 1 exists at all (it may be dispatched late); bounded
    bool ok = true;
    if (side == 0) {
        uint32_t there = 0;
        for (uint32_t s = 0; s < (kSpin << 4) && !there; s++)
            there = __hip_atomic_load(&c->role_xcd[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = there != 0;
    }
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= kRounds && ok; i++) {
        if (side == 0)
            put<MODE>(mine, (uint32_t)i);
        uint32_t s = 0;
        while (get<MODE>(theirs) != (uint32_t)i && ++s < kSpin)
            ;
        if (s >= kSpin)
            ok = false;
        if (side == 1)
            put<MODE>(mine, (uint32_t)i);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    if (!ok)
        atomicAdd(&c->fail, 1u);
    if (side == 0)
        c->ticks = t1 - t0;
}

#define CK(x)                                                                                                          \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                    \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

template <int MODE> static int run(Ctl *d, const char *what)
{
    for (int same = 1; same >= 0; same--) {
        double best = 1e30;
        int fails = 0;
        Ctl h;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipMemset(d, 0, sizeof(Ctl)));
            CK(hipFuncSetAttribute((const void *)k_pingpong<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            hipLaunchKernelGGL(k_pingpong<MODE>, dim3(256), dim3(64), 96 * 1024, 0, d, same);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&h, d, sizeof(Ctl), hipMemcpyDeviceToHost));
            if (h.fail || !h.taken[1] || !h.ticks) {
                fails++;
                continue;
            }
            const double us = (double)h.ticks / 100.0 / kRounds;
            if (us < best)
                best = us;
        }
        if (best < 1e29)
            printf("%-52s %-9s round trip %6.2f us  one way %5.2f us  (xcd %u -> %u, hw_id %08x / %08x)%s\n", what,
                   same ? "same XCD" : "other XCD", best, best / 2, h.role_xcd[0] - 1, h.role_xcd[1] - 1, h.cu[0], h.cu[1],
                   fails ? "  [some repeats failed]" : "");
        else
            printf("%-52s %-9s no: the poll never saw the store (or no partner formed)\n", what, same ? "same XCD" : "other XCD");
    }
    return 0;
}

int main()
{
    Ctl *d;
    CK(hipMalloc(&d, sizeof(Ctl)));
    run<0>(d, "sc1 store, sc1 vector load (product)");
    run<1>(d, "plain store, s_dcache_inv + s_load");
    run<2>(d, "plain store, s_load glc");
    run<3>(d, "plain store, sc1 vector load");
    run<4>(d, "plain store, sc0 vector load");
    run<5>(d, "plain store, sc0 sc1 vector load");
    run<6>(d, "sc1 store, s_dcache_inv + s_load");
    run<7>(d, "plain store, buffer_inv sc1 + plain vector load");
    run<8>(d, "sc0 sc1 store, sc0 sc1 vector load");
    CK(hipFree(d));
    return 0;
}
