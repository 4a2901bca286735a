// jb_batch.cpp -- state-level C ABI (include/jbonsai_amd.h, part 1): uploads a
// batch of state-level utterances to HBM, runs the HIP hot path, reads PCM back.
//
// Stands in for   MlpgAdjust::create x3  (src/mlpg_adjust/mod.rs:51-95)
//               + SpeechGenerator::new/generate_all (src/speech.rs:25-96)
// as called by Engine::generator (src/engine.rs:333-365), batched over utterances.
// There is NO CPU fallback: without a HIP device every entry returns JB_ERR_DEVICE.
#include "jb_host.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <set>
#include <tuple>
#include <numeric>
#include <thread>

namespace jb {

thread_local std::string g_err;

void set_error(const std::string &s) { g_err = s; }

int hip_fail(hipError_t e, const char *what)
{
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return JB_ERR_DEVICE;
}

// ---- shared Gaussian noise stream ------------------------------------------
// Random::nrandom (src/vocoder/excitation.rs:177-237): LCG + polar Box-Muller,
// seed next=1, identical for every utterance => one table per device, grown on
// demand.  Only bits 16..30 of the LCG state are used, so u32 state suffices.
static void noise_fill(std::vector<double> &out, size_t n, uint32_t &next, size_t &have)
{
    // n and have are multiples of 64: whole fill_queue() refills only
    out.resize(n);
    auto rnd = [&]() {
        next = next * 1103515245u + 12345u;
        return (double)((next >> 16) & 32767u) / 32767.0;
    };
    double sv[32];
    while (have < n) {
        double *q = out.data() + have;
        int k = 0;
        while (k < 32) {
            double r1 = 2.0 * rnd() - 1.0, r2 = 2.0 * rnd() - 1.0;
            double s = r1 * r1 + r2 * r2;
            if (0.0 < s && s < 1.0) {
                q[2 * k] = r1;
                q[2 * k + 1] = r2;
                sv[k] = s;
                k++;
            }
        }
        for (k = 0; k < 32; k++) {
            double m = std::sqrt(-2.0 * std::log(sv[k]) / sv[k]);
            q[2 * k] *= m;
            q[2 * k + 1] *= m;
        }
        have += 64;
    }
}

struct NoiseCache {
    std::mutex mu;
    std::vector<double> host; // always a multiple of 64 long
    uint32_t lcg = 1;
    size_t have = 0;
    std::map<int, std::shared_ptr<NoiseDev>> dev; // device -> current table
};
// never destroyed: its tables would be freed after the HIP runtime has shut down
static NoiseCache &g_noise = *new NoiseCache();

NoiseDev::~NoiseDev()
{
    if (ptr) {
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != device)
            hipSetDevice(device);
        hipFree(ptr);
        if (cur >= 0 && cur != device)
            hipSetDevice(cur);
    }
}

// The table of a device grows geometrically (a creeping maximum length does not reallocate every
// time); a superseded table lives on while a Batch still holds it and is freed with the last one.
int noise_table(int device, size_t need, std::shared_ptr<NoiseDev> *out)
{
    std::lock_guard<std::mutex> lk(g_noise.mu);
    need = (need + 63) / 64 * 64;
    if (need == 0)
        need = 64;
    auto it = g_noise.dev.find(device);
    if (it != g_noise.dev.end() && it->second->len >= need) {
        *out = it->second;
        return JB_OK;
    }
    if (it != g_noise.dev.end())
        need = std::max(need, 2 * it->second->len);
    if (g_noise.have < need) {
        // restart from the stored LCG state at a refill boundary
        std::vector<double> &h = g_noise.host;
        size_t have = g_noise.have;
        noise_fill(h, need, g_noise.lcg, have);
        g_noise.have = have;
    }
    double *d = nullptr;
    hipError_t e = hipMalloc(&d, need * sizeof(double));
    if (e != hipSuccess)
        return hip_fail(e, "hipMalloc(noise)");
    e = hipMemcpy(d, g_noise.host.data(), need * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hipFree(d);
        return hip_fail(e, "hipMemcpy(noise)");
    }
    std::shared_ptr<NoiseDev> nd(new NoiseDev());
    nd->device = device;
    nd->ptr = d;
    nd->len = need;
    g_noise.dev[device] = nd; // the older, shorter table stays alive in the batches that hold it
    *out = nd;
    return JB_OK;
}

// ---- batch -----------------------------------------------------------------
// --------------------------------------------------------------------------
// Device memory of finished batches is kept per device and handed to the next batch instead of
// going back to the driver: a batch of config 2 takes ~60 allocations and tens of GB, and
// hipMalloc/hipFree of that cost 25-250 ms per jb_synthesize_batch call (more than the 48 ms of
// GPU work for 64 x 157 s).  Blocks are rounded to 2 MB (4 KB below that) and reused when the
// cached block is at most a quarter larger than the request.  JB_DEVICE_POOL_MB caps what is kept
// (default 65536, 0 = off); jb_release_cached_memory() empties the pool; an out-of-memory
// hipMalloc empties it and retries once.  Reused memory is NOT zero: dalloc(zero) clears.
namespace {
struct DevPool {
    std::multimap<size_t, void *> free_;
    size_t cached = 0;
};
std::mutex g_pool_mu;
std::map<int, DevPool> g_pool;

std::atomic<size_t> &pool_cap_ref()
{
    static std::atomic<size_t> cap{[] {
        const char *ev = getenv("JB_DEVICE_POOL_MB");
        return (size_t)(ev ? std::max(0L, atol(ev)) : 65536L) << 20;
    }()};
    return cap;
}
size_t pool_cap() { return pool_cap_ref().load(std::memory_order_relaxed); }

size_t pool_round(size_t bytes)
{
    const size_t g = bytes >= (2u << 20) ? (2u << 20) : 4096u;
    return (bytes + g - 1) / g * g;
}

void pool_trim_locked(DevPool &dp, size_t keep)
{
    // largest blocks first: they are the ones a differently shaped batch is least likely to fit
    while (dp.cached > keep && !dp.free_.empty()) {
        auto it = std::prev(dp.free_.end());
        hipFree(it->second);
        dp.cached -= it->first;
        dp.free_.erase(it);
    }
}

hipError_t pool_alloc(int device, size_t bytes, void **out, size_t *got)
{
    const size_t need = pool_round(bytes);
    *got = need;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        DevPool &dp = g_pool[device];
        auto it = dp.free_.lower_bound(need);
        if (it != dp.free_.end() && it->first <= need + need / 4) {
            *out = it->second;
            *got = it->first;
            dp.cached -= it->first;
            dp.free_.erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, need);
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(g_pool_mu);
        pool_trim_locked(g_pool[device], 0);
        e = hipMalloc(out, need);
    }
    return e;
}

void pool_free(int device, void *p, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (pool_cap() == 0) {
        hipFree(p);
        return;
    }
    DevPool &dp = g_pool[device];
    dp.free_.emplace(bytes, p);
    dp.cached += bytes;
    if (dp.cached > pool_cap())
        pool_trim_locked(dp, pool_cap());
}
} // namespace

// The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that
// share one run one after the other.  A batch has three streams; two batches in flight, or a batch and a read-back,
// need more than four queues to overlap at all: with 16, two config-2 batches in flight take 82.3 instead of 86.0 ms
// per step and the config-3 job 865 instead of 880 ms per pass (one batch at a time: no difference).  So the
// library asks for 16 when it is loaded, unless the variable is already set; it has to happen before the
// process's first HIP call, which is why this is a load-time constructor and not something jb_engine_load does
// (a host that initialises HIP before loading the library sets GPU_MAX_HW_QUEUES itself: INTEGRATION.md).
// Opt-out: JB_LEAVE_HIP_ENV=1 -- the library then changes nothing in the process's environment (a host that
// manages HIP's settings itself, or that loads the library into a process whose other threads may be reading the
// environment: glibc's setenv is not safe against a concurrent getenv).
__attribute__((constructor)) static void jb_ask_for_hw_queues()
{
    const char *leave = getenv("JB_LEAVE_HIP_ENV");
    if (leave && atoi(leave) != 0)
        return;
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
}

// Streams are kept the same way: hipStreamCreateWithFlags / hipStreamDestroy take 2.4 / 1.9 ms each on
// this stack, three of each per batch = 13 of the 24 ms of a one-sentence jb_synthesize call.
namespace {
std::map<int, std::vector<hipStream_t>> g_streams; // guarded by g_pool_mu
constexpr size_t kStreamPoolMax = 48;

hipError_t stream_acquire(int device, hipStream_t *st)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto &v = g_streams[device];
        if (!v.empty()) {
            *st = v.back();
            v.pop_back();
            return hipSuccess;
        }
    }
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

void stream_release(int device, hipStream_t st)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto &v = g_streams[device];
        if (v.size() < kStreamPoolMax) {
            v.push_back(st);
            return;
        }
    }
    hipStreamDestroy(st);
}

} // namespace

static void release_pinned_chunks();
void release_cached_memory()
{
    release_pinned_chunks(); // the pinned staging chunks kept between batches (up to 16 x 8 MB of locked host memory)
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (auto &kv : g_streams) {
        hipSetDevice(kv.first);
        for (hipStream_t st : kv.second)
            hipStreamDestroy(st);
        kv.second.clear();
    }
    for (auto &kv : g_pool) {
        hipSetDevice(kv.first);
        pool_trim_locked(kv.second, 0);
    }
    if (cur >= 0)
        hipSetDevice(cur);
}

void set_cached_memory_limit(size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    pool_cap_ref().store(bytes, std::memory_order_relaxed);
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (auto &kv : g_pool)
        if (kv.second.cached > bytes) {
            hipSetDevice(kv.first);
            pool_trim_locked(kv.second, bytes);
        }
    if (cur >= 0)
        hipSetDevice(cur);
}

Batch::~Batch()
{
    if (device >= 0)
        hipSetDevice(device);
    // everything this batch enqueued has to be finished before another batch may get the memory
    // or the streams: its own streams (and the legacy stream its memsets ran on), not the whole device --
    // another batch of the process may be in the middle of a step
    for (hipStream_t st : {stream, stream_lf0, stream_lpf, stream_voc})
        if (st)
            hipStreamSynchronize(st);
    if (!allocs.empty())
        hipStreamSynchronize(nullptr);
    for (auto &a : allocs)
        pool_free(device, a.first, a.second);
    if (ev0)
        hipEventDestroy(ev0);
    if (ev1)
        hipEventDestroy(ev1);
    if (ev2)
        hipEventDestroy(ev2);
    if (ev3)
        hipEventDestroy(ev3);
    for (hipEvent_t ev : {ev_fork, ev_lf0, ev_lpf, ev_prep, ev_build, ev_mcpbuild, ev_ivar, ev_fb})
        if (ev)
            hipEventDestroy(ev);
    for (hipEvent_t ev : {ev_mlpg_done, ev_voc_done})
        if (ev)
            hipEventDestroy(ev);
    for (hipStream_t st : {stream_lf0, stream_lpf})
        if (st && st != stream)
            stream_release(device, st);
    if (stream)
        stream_release(device, stream);
}

template <class T> int Batch::dalloc(T **p, size_t n, bool zero)
{
    *p = nullptr;
    if (n == 0)
        n = 1;
    void *v = nullptr;
    size_t got = 0;
    hipError_t e = pool_alloc(device, n * sizeof(T), &v, &got);
    if (e != hipSuccess) {
        char msg[128];
        snprintf(msg, sizeof msg, "hipMalloc(%zu bytes)", n * sizeof(T));
        return hip_fail(e, msg);
    }
    allocs.emplace_back(v, got);
    bytes_alloc += n * sizeof(T);
    if (zero)
        zero_list.emplace_back(v, std::min(got, (n * sizeof(T) + 15) / 16 * 16));
    *p = (T *)v;
    return JB_OK;
}

// up to kZeroSegs blocks of memory cleared by one launch: segment blockIdx.y, 16 bytes per thread and trip.
// (24: an ordinary batch has 25-30 such blocks, so that the second launch of the loop below is what every test runs)
constexpr int kZeroSegs = 24;
struct ZeroSegs {
    uint4 *p[kZeroSegs];
    unsigned long long n16[kZeroSegs]; // 16-byte units
};
__global__ __launch_bounds__(256) void k_zero_segments(ZeroSegs z)
{
    uint4 *p = z.p[blockIdx.y];
    const unsigned long long n = z.n16[blockIdx.y];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 256u)
        p[i] = make_uint4(0u, 0u, 0u, 0u);
}
int Batch::flush_zero()
{
    if (zero_list.empty())
        return JB_OK;
    for (size_t i0 = 0; i0 < zero_list.size(); i0 += kZeroSegs) {
        ZeroSegs z{};
        const size_t cnt = std::min(zero_list.size() - i0, (size_t)kZeroSegs);
        unsigned long long biggest = 0;
        for (size_t i = 0; i < cnt; i++) {
            z.p[i] = (uint4 *)zero_list[i0 + i].first; // (pool blocks are aligned far beyond 16 bytes)
            z.n16[i] = (zero_list[i0 + i].second + 15) / 16;
            biggest = std::max(biggest, z.n16[i]);
        }
        const unsigned gx = (unsigned)std::min<unsigned long long>((biggest + 255) / 256, 2048ull);
        hipLaunchKernelGGL(k_zero_segments, dim3(std::max(gx, 1u), (unsigned)cnt), dim3(256), 0, nullptr, z);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return hip_fail(e, "k_zero_segments");
    }
    zero_list.clear();
    // (the legacy stream does not order itself with the batch's non-blocking streams: whoever uses a block right
    // away -- the redo rounds -- needs it cleared NOW; creation waits for the legacy stream at its end anyway)
    hipError_t e = hipStreamSynchronize(nullptr);
    return e == hipSuccess ? JB_OK : hip_fail(e, "k_zero_segments");
}

// pinned staging chunks of the upload arenas (all of one size), reused between batches
static std::mutex g_pin_mu;
static std::vector<uint8_t *> g_pin_free;
static constexpr size_t kPinChunk = 8u << 20, kPinKeep = 16;
static void release_pinned_chunks()
{
    std::vector<uint8_t *> drop;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        drop.swap(g_pin_free);
    }
    for (uint8_t *q : drop)
        (void)hipHostFree(q);
}
bool Batch::PinnedChunk::acquire(size_t bytes)
{
    reset();
    if (bytes != kPinChunk)
        return false;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        if (!g_pin_free.empty()) {
            p = g_pin_free.back();
            g_pin_free.pop_back();
            pageable = false;
            return true;
        }
    }
    void *v = nullptr;
    if (hipHostMalloc(&v, bytes, hipHostMallocPortable) != hipSuccess) {
        // no pinned memory to be had (a locked-memory limit, a host short of it): a pageable buffer does -- hipMemcpy
        // takes either, the copy is just slower (what every batch did until round 4)
        (void)hipGetLastError();
        p = new (std::nothrow) uint8_t[bytes];
        pageable = p != nullptr;
        return p != nullptr;
    }
    p = (uint8_t *)v;
    pageable = false;
    return true;
}
void Batch::PinnedChunk::reset()
{
    if (!p)
        return;
    if (pageable) {
        delete[] p;
        p = nullptr;
        pageable = false;
        return;
    }
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        if (g_pin_free.size() < kPinKeep) {
            g_pin_free.push_back(p);
            p = nullptr;
            return;
        }
    }
    (void)hipHostFree(p);
    p = nullptr;
}

// upload with de-duplication on (host pointer, byte size)
int Batch::upload(const void *host, size_t bytes, const void **dev)
{
    *dev = nullptr;
    if (!host || bytes == 0)
        return JB_OK;
    auto key = std::make_pair(host, bytes);
    auto it = uploaded.find(key);
    if (it != uploaded.end()) {
        *dev = it->second;
        return JB_OK;
    }
    // Small arrays (a batch of 64 long utterances brings ~570 of them, 45 us per synchronous
    // hipMemcpy = 26 ms) are packed into 8 MB arena chunks that go up in one copy each
    // (flush_uploads); large ones keep a block and a copy of their own.
    constexpr size_t kChunk = 8u << 20;
    uint8_t *d;
    int rc;
    if (bytes > kChunk / 4) {
        if ((rc = dalloc(&d, bytes, false)))
            return rc;
        hipError_t e = hipMemcpy(d, host, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess)
            return hip_fail(e, "hipMemcpy(H2D)");
    } else {
        const size_t need = (bytes + 255) / 256 * 256;
        if (up_chunks.empty() || !up_chunks.back().host.get() || up_chunks.back().used + need > kChunk) {
            UploadChunk c;
            if ((rc = dalloc(&c.dev, kChunk, false)))
                return rc;
            if (!c.host.acquire(kChunk)) {
                set_error("hipHostMalloc(upload arena)");
                return JB_ERR_DEVICE;
            }
            up_chunks.push_back(std::move(c));
        }
        UploadChunk &c = up_chunks.back();
        memcpy(c.host.get() + c.used, host, bytes);
        d = c.dev + c.used;
        c.used += need;
    }
    uploaded[key] = d;
    bytes_input += bytes;
    *dev = d;
    return JB_OK;
}

template <class T> int Batch::stage(const T *host, size_t n, T **dev)
{
    constexpr size_t kChunk = 8u << 20;
    *dev = nullptr;
    const size_t bytes = n * sizeof(T);
    if (bytes > kChunk / 4 || bytes == 0) { // (large, or nothing: a block and a copy of its own)
        int rc = dalloc(dev, n, false);
        if (rc)
            return rc;
        if (bytes) {
            hipError_t e = hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess)
                return hip_fail(e, "hipMemcpy(H2D)");
        }
        return JB_OK;
    }
    const size_t need = (bytes + 255) / 256 * 256;
    // (a chunk whose host buffer create() has already given back takes no more bytes: ADVICE r5)
    if (up_chunks.empty() || !up_chunks.back().host.get() || up_chunks.back().used + need > kChunk) {
        UploadChunk c;
        int rc = dalloc(&c.dev, kChunk, false);
        if (rc)
            return rc;
        if (!c.host.acquire(kChunk)) {
            set_error("host memory for the upload arena");
            return JB_ERR_DEVICE;
        }
        up_chunks.push_back(std::move(c));
    }
    UploadChunk &c = up_chunks.back();
    memcpy(c.host.get() + c.used, host, bytes);
    *dev = (T *)(c.dev + c.used);
    c.used += need;
    bytes_input += bytes;
    return JB_OK;
}

int Batch::flush_uploads()
{
    for (UploadChunk &c : up_chunks) {
        if (c.used > c.sent) {
            if (!c.host.get()) {
                set_error("upload arena: bytes queued in a chunk without a host buffer");
                return JB_ERR_DEVICE;
            }
            hipError_t e = hipMemcpy(c.dev + c.sent, c.host.get() + c.sent, c.used - c.sent, hipMemcpyHostToDevice);
            if (e != hipSuccess)
                return hip_fail(e, "hipMemcpy(H2D arena)");
            c.sent = c.used;
        }
    }
    return JB_OK;
}

PdfSet::~PdfSet()
{
    if (device >= 0)
        hipSetDevice(device);
    for (const float *p : tab)
        if (p)
            hipFree((void *)p);
}

// gather + blend of the per-state Gaussians on the device.  Utterances whose index
// arrays alias (same host pointers, same weights) share one result, like aliased uploads do.
int Batch::gather_states(const jb_voice_desc *voice, const IndexSrc &idx, size_t n,
                         std::vector<StreamStatesDev> &out)
{
    const PdfSet &ps = *idx.set;
    if (ps.device != device) {
        set_error("pdf set lives on another device");
        return JB_ERR_INVALID;
    }
    if (ps.ns < voice->nstream || ps.nv == 0 || ps.nv > (uint32_t)kMaxVoices) {
        set_error("pdf set does not match the voice description");
        return JB_ERR_INVALID;
    }
    const uint32_t ns = voice->nstream;
    out.assign(n * ns, StreamStatesDev{});
    const bool gtrace = getenv("JB_CREATE_TRACE") != nullptr;
    const auto tg0 = std::chrono::steady_clock::now();
    auto gmark = [&](const char *what) {
        if (gtrace)
            fprintf(stderr, "    gather: %-24s at %.3f ms\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tg0).count());
    };
    std::vector<GatherJob> jobs;
    std::vector<size_t> share(n * ns, (size_t)-1); // job whose arrays (utterance, stream) uses
    std::map<std::string, size_t> seen;
    uint64_t max_elems = 0;
    int rc;
    double acc_check = 0, acc_key = 0, acc_up = 0;
    auto tick = [&]() { return gtrace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point(); };
    auto tock = [&](std::chrono::steady_clock::time_point t0, double &acc) {
        if (gtrace)
            acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    for (size_t i = 0; i < n; i++) {
        const jb_index_utt &u = idx.utts[i];
        for (uint32_t si = 0; si < ns; si++) {
            const jb_stream_desc &sdsc = voice->stream[si];
            const jb_index_stream &is = u.stream[si];
            const uint32_t WL = sdsc.vector_length * sdsc.num_windows;
            const uint32_t want_len = 2 * WL + (sdsc.is_msd ? 1u : 0u);
            if (u.num_states == 0)
                continue;
            if (!is.weight) {
                set_error("interpolation weights missing");
                return JB_ERR_INVALID;
            }
            GatherJob j{};
            j.nv = ps.nv;
            j.S = u.num_states;
            j.WL = WL;
            j.row_len = want_len;
            j.has_msd = sdsc.is_msd ? 1 : 0;
            j.lf0_offset = (si == 1) ? u.lf0_offset : 0.0;
            std::string key((const char *)&u.num_states, sizeof u.num_states);
            key.append((const char *)&si, sizeof si);
            key.append((const char *)&j.lf0_offset, sizeof j.lf0_offset);
            const auto tk0 = tick();
            for (uint32_t v = 0; v < ps.nv; v++) {
                const size_t t = (size_t)v * ps.ns + si;
                if (!is.row[v] || ps.row_len[t] != want_len) {
                    set_error("pdf row indices missing or table row length mismatch");
                    return JB_ERR_INVALID;
                }
                uint32_t rmax = 0; // (a plain maximum: the loop vectorises)
                for (uint32_t s = 0; s < u.num_states; s++)
                    rmax = std::max(rmax, is.row[v][s]);
                if (u.num_states && rmax >= ps.n_rows[t]) {
                    set_error("pdf row index out of range");
                    return JB_ERR_INVALID;
                }
                j.tab[v] = ps.tab[t];
                j.w[v] = is.weight[v];
                key.append((const char *)&is.row[v], sizeof(void *));
                key.append((const char *)&is.weight[v], sizeof(double));
            }
            tock(tk0, acc_check);
            const auto tk1 = tick();
            auto it = seen.find(key);
            if (it != seen.end()) {
                share[i * ns + si] = it->second;
                continue;
            }
            share[i * ns + si] = jobs.size();
            tock(tk1, acc_key);
            const auto tk2 = tick();
            for (uint32_t v = 0; v < ps.nv; v++) {
                const void *dp;
                if ((rc = upload(is.row[v], sizeof(uint32_t) * u.num_states, &dp)))
                    return rc;
                j.row[v] = (const uint32_t *)dp;
            }
            tock(tk2, acc_up);
            // (the arrays of all jobs come out of ONE allocation, carved below: three pool allocations per
            // (utterance, stream) were most of the 9 ms this function took for 512 utterances)
            max_elems = std::max<uint64_t>(max_elems, (uint64_t)u.num_states * WL);
            seen[key] = jobs.size();
            jobs.push_back(j);
        }
    }
    gmark("jobs listed, rows staged");
    if (gtrace)
        fprintf(stderr, "    gather: of that checks + key %.3f, map %.3f, staging %.3f ms\n", acc_check, acc_key, acc_up);
    if (jobs.empty())
        return JB_OK;
    {
        auto up32 = [](size_t x) { return (x + 31) / 32 * 32; }; // 256-byte pieces
        size_t total = 0;
        for (const GatherJob &j : jobs)
            total += 2 * up32((size_t)j.S * j.WL) + (j.has_msd ? up32(j.S) : 0);
        double *slab;
        if ((rc = dalloc(&slab, total, false)))
            return rc;
        size_t off = 0;
        for (size_t q = 0; q < jobs.size(); q++) {
            GatherJob &j = jobs[q];
            j.mean = slab + off;
            off += up32((size_t)j.S * j.WL);
            j.var = slab + off;
            off += up32((size_t)j.S * j.WL);
            if (j.has_msd) {
                j.msd = slab + off;
                off += up32(j.S);
            }
        }
        // every (utterance, stream) gets the arrays of the job it shares
        for (size_t i = 0; i < n; i++)
            for (uint32_t si = 0; si < ns; si++) {
                const size_t q = share[i * ns + si];
                if (q == (size_t)-1)
                    continue;
                out[i * ns + si].mean = jobs[q].mean;
                out[i * ns + si].var = jobs[q].var;
                out[i * ns + si].msd = jobs[q].msd;
            }
    }
    gmark("outputs carved");
    GatherJob *jd;
    if ((rc = stage(jobs.data(), jobs.size(), &jd)))
        return rc;
    if ((rc = flush_uploads())) // the index rows the gather reads, and its job list
        return rc;
    gmark("rows uploaded");
    hipError_t e;
    // grid.y is limited to 65535 jobs per launch
    for (size_t j0 = 0; j0 < jobs.size(); j0 += 65535) {
        const uint32_t nj = (uint32_t)std::min<size_t>(65535, jobs.size() - j0);
        if ((e = launch_gather(jd + j0, nj, max_elems, stream)) != hipSuccess)
            return hip_fail(e, "k_gather_blend");
    }
    gmark("kernels launched");
    if ((e = hipStreamSynchronize(stream)) != hipSuccess)
        return hip_fail(e, "k_gather_blend");
    gmark("done");
    return JB_OK;
}

static int check_voice(const jb_voice_desc *v, bool need_windows = true, bool vocoder_level = false)
{
    if (!v)
        return JB_ERR_INVALID;
    if (v->stage > (uint32_t)mglsa_max_stage()) {
        // (the delay lines of a chunk's stages share one CU's LDS: 64 taps x 8 B x stage)
        set_error("Stage::NonZero: stages above " + std::to_string(mglsa_max_stage()) + " are not supported");
        return JB_ERR_UNSUPPORTED;
    }
    if (!(v->beta >= 0.0)) {
        set_error("beta must be >= 0");
        return JB_ERR_INVALID;
    }
    if (v->nstream != 3) {
        // Engine::generator indexes stream_metadata(2) unconditionally (src/engine.rs:305)
        set_error("exactly 3 streams (MCP, LF0, LPF) are required");
        return JB_ERR_UNSUPPORTED;
    }
    if (v->fperiod == 0 || v->sampling_frequency == 0) {
        set_error("fperiod and sampling_frequency must be positive");
        return JB_ERR_INVALID;
    }
    const jb_stream_desc &m = v->stream[0], &l = v->stream[1], &p = v->stream[2];
    if (l.vector_length != 1) {
        set_error("The size of lf0 static vector must be 1."); // src/speech.rs:35-37
        return JB_ERR_INVALID;
    }
    // SpeechGenerator::new panics on an even LPF length, 0 included (speech.rs:38-40): the ring-buffer-less
    // branch of Excitation::get (excitation.rs:87-100) can be reached through Vocoder::synthesize alone,
    // never through Engine / SpeechGenerator, which is the boundary this library mirrors
    // (Vocoder::new takes any nlpf: an even count has its noise tap at (nlpf - 1) / 2 like an odd one, excitation.rs:137-140)
    if (p.vector_length % 2 == 0 && !vocoder_level) {
        set_error("The number of low-pass filter coefficient must be odd numbers."); // speech.rs:38-40
        return JB_ERR_INVALID;
    }
    if (p.vector_length > (uint32_t)excite_max_nlpf()) {
        // (k_excite_any keeps a block's 256 + nlpf - 1 source samples in LDS)
        set_error("nlpf > " + std::to_string(excite_max_nlpf()) + " is not supported");
        return JB_ERR_UNSUPPORTED;
    }
    if (m.vector_length < 2 || m.vector_length - 1 > (uint32_t)(kGroups * kMaxTPL) || m.vector_length > (uint32_t)kMaxNmcp) {
        set_error("nmcp must be in [2, 64]");
        return JB_ERR_UNSUPPORTED;
    }
    if (v->stage != 0 && m.vector_length < 3) {
        set_error("Stage::NonZero needs at least a gain and two line spectral frequencies");
        return JB_ERR_INVALID;
    }
    for (uint32_t i = 0; need_windows && i < v->nstream; i++) {
        const jb_stream_desc &s = v->stream[i];
        if (s.num_windows == 0 || s.num_windows > JB_MAX_WINDOW || !s.win_coef) {
            set_error("bad window description");
            return JB_ERR_INVALID;
        }
        uint32_t tot = 0;
        for (uint32_t w = 0; w < s.num_windows; w++) {
            if (s.win_width[w] == 0 || s.win_width[w] > (uint32_t)kMaxBand) {
                set_error("window widths must be in 1..9");
                return JB_ERR_UNSUPPORTED;
            }
            tot += s.win_width[w];
        }
        if (tot > (uint32_t)kMaxCoef)
            return JB_ERR_UNSUPPORTED;
    }
    return JB_OK;
}

int Batch::create(const jb_voice_desc *voice, const jb_state_utt *utts, size_t n, const jb_batch_opts *opts,
                  Batch **out, const IndexSrc *idx, const TrackSrc *trk)
{
    *out = nullptr;
    int rc = check_voice(voice, trk == nullptr, trk && trk->vocoder_level);
    if (rc)
        return rc;
    // JB_CREATE_TRACE=1: where the creation of a batch spends its time (stderr)
    const bool ctrace = getenv("JB_CREATE_TRACE") != nullptr;
    const auto tc0 = std::chrono::steady_clock::now();
    auto cmark = [&](const char *what) {
        if (ctrace)
            fprintf(stderr, "  create: %-28s at %.3f ms\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count());
    };
    // Parameter tracks as the source (SpeechGenerator::new, src/speech.rs:25-50).  The kernels behind the
    // frame prologue want the LF0 stream's voiced flags and voiced runs, which the state walk of that
    // stream produces: give it the track's runs of voiced / unvoiced frames (a frame is voiced where
    // lf0 != NODATA, vocoder/mod.rs:73-77) as pseudo-states (msd 1 / 0, threshold 0.5).
    std::vector<jb_state_utt> pseudo;
    std::vector<std::vector<uint32_t>> pseudo_dur;
    std::vector<std::vector<double>> pseudo_msd;
    if (trk) {
        if (n && !trk->utts)
            return JB_ERR_INVALID;
        pseudo.resize(n);
        pseudo_dur.resize(n);
        pseudo_msd.resize(n);
        for (size_t i = 0; i < n; i++) {
            const jb_track_utt &t = trk->utts[i];
            if (t.n_spectrum != t.n_lf0 || t.n_spectrum != t.n_lpf) {
                set_error("The length of spectrum, lf0, and lpf must be the same."); // speech.rs:32-34
                return JB_ERR_INVALID;
            }
            if (t.n_lf0 && t.lf0_width != 1) {
                set_error("The size of lf0 static vector must be 1."); // speech.rs:35-37
                return JB_ERR_INVALID;
            }
            if (t.n_lpf && t.lpf_width % 2 == 0 && !trk->vocoder_level) {
                set_error("The number of low-pass filter coefficient must be odd numbers."); // speech.rs:38-40
                return JB_ERR_INVALID;
            }
            if (t.n_lf0 && (t.spectrum_width != voice->stream[0].vector_length ||
                            t.lpf_width != voice->stream[2].vector_length)) {
                set_error("track widths differ from the vocoder's nmcp / nlpf (Vocoder::new, vocoder/mod.rs:45-55)");
                return JB_ERR_INVALID;
            }
            if (t.n_lf0 && (!t.spectrum || !t.lf0 || !t.lpf))
                return JB_ERR_INVALID;
            if (t.n_lf0 > 0xffffffffull / voice->fperiod) {
                set_error("utterance too long");
                return JB_ERR_INVALID;
            }
            std::vector<uint32_t> &d = pseudo_dur[i];
            std::vector<double> &m = pseudo_msd[i];
            for (size_t f = 0; f < t.n_lf0; f++) {
                const double v = t.lf0[f] != kNoData ? 1.0 : 0.0;
                if (m.empty() || m.back() != v) {
                    m.push_back(v);
                    d.push_back(0);
                }
                d.back()++;
            }
            jb_state_utt &u = pseudo[i];
            memset(&u, 0, sizeof u);
            u.num_states = (uint32_t)d.size();
            u.durations = d.data();
            u.stream[1].msd = m.data();
            for (int si = 0; si < JB_MAX_STREAM; si++) {
                u.stream[si].msd_threshold = 0.5;
                u.stream[si].gv_weight = 1.0;
            }
        }
        utts = pseudo.data();
    }
    if (n && !utts)
        return JB_ERR_INVALID;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        set_error("no HIP device available (this library has no CPU path)");
        return JB_ERR_DEVICE;
    }
    std::unique_ptr<Batch> b(new Batch());
    int dev = opts ? opts->device : -1;
    if (dev < 0) {
        e = hipGetDevice(&dev);
        if (e != hipSuccess)
            return hip_fail(e, "hipGetDevice");
    }
    if (dev >= ndev) {
        set_error("device ordinal out of range");
        return JB_ERR_INVALID;
    }
    e = hipSetDevice(dev);
    if (e != hipSuccess)
        return hip_fail(e, "hipSetDevice");
    b->device = dev;
    b->flags = opts ? opts->flags : 0;
    if (b->flags & JB_BATCH_MLPG_ONLY)
        b->flags |= JB_BATCH_KEEP_TRACKS; // the [frame][dim] tracks are the result
    if (trk && (b->flags & JB_BATCH_MLPG_ONLY)) {
        set_error("JB_BATCH_MLPG_ONLY needs state-level input");
        return JB_ERR_INVALID;
    }
    b->from_tracks = trk != nullptr;
    b->voice = *voice;
    // (stream priorities were tried for the critical path and made every latency-bound kernel
    // 2-4x slower on this stack; ordering is done with events instead)
    if (opts && opts->reserved0) {
        // (rounds 1-3 had a CU partition here, hipExtStreamCreateWithCUMask streams for parameter generation and
        // vocoder: it lost at every split and its streams could not be destroyed reliably; removed in round 4)
        set_error("jb_batch_opts.reserved0 must be 0 (the CU partition of earlier versions was removed)");
        return JB_ERR_UNSUPPORTED;
    }
    e = stream_acquire(b->device, &b->stream);
    if (e != hipSuccess)
        return hip_fail(e, "hipStreamCreate");
    // JB_ONE_STREAM=1 (profiling aid): the three parameter-generation chains run back to back
    // on the main stream, so that per-kernel durations are free of overlap effects
    if (getenv("JB_ONE_STREAM") && atoi(getenv("JB_ONE_STREAM")) != 0) {
        b->stream_lf0 = b->stream_lpf = b->stream;
    } else if ((e = stream_acquire(b->device, &b->stream_lf0)) != hipSuccess ||
               (e = stream_acquire(b->device, &b->stream_lpf)) != hipSuccess)
        return hip_fail(e, "hipStreamCreate");
    b->stream_voc = b->stream;
    hipEventCreateWithFlags(&b->ev_mlpg_done, hipEventDisableTiming);
    hipEventCreateWithFlags(&b->ev_voc_done, hipEventDisableTiming);
    hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming);
    hipEventCreateWithFlags(&b->ev_lf0, hipEventDisableTiming);
    hipEventCreateWithFlags(&b->ev_lpf, hipEventDisableTiming);
    hipEventCreateWithFlags(&b->ev_prep, hipEventDisableTiming);
    hipEventCreateWithFlags(&b->ev_build, hipEventDisableTiming);
    hipEventCreateWithFlags(&b->ev_mcpbuild, hipEventDisableTiming);
    hipEventCreateWithFlags(&b->ev_ivar, hipEventDisableTiming);
    hipEventCreateWithFlags(&b->ev_fb, hipEventDisableTiming);
    hipEventCreate(&b->ev0);
    hipEventCreate(&b->ev1);
    hipEventCreate(&b->ev2);
    hipEventCreate(&b->ev3);

    const int B = (int)n;
    b->B = B;
    std::vector<StreamStatesDev> gathered; // indexed source: per-state Gaussians produced on the device
    cmark("streams and events made");
    if (idx && (rc = b->gather_states(voice, *idx, n, gathered)))
        return rc;
    cmark("states gathered on the device");
    b->T.resize(n);
    b->frame_off.resize(n + 1);
    std::vector<UttDev> hu(n);
    uint64_t sumT = 0, sumS = 0, sum_mt = 0; // sum_mt: frames of the [dim][frame] workspace, rows padded to 16
    uint32_t maxT = 0, maxS = 0;
    for (size_t i = 0; i < n; i++) {
        const jb_state_utt &u = utts[i];
        if (u.num_states && !u.durations)
            return JB_ERR_INVALID;
        uint64_t T = 0;
        for (uint32_t s = 0; s < u.num_states; s++)
            T += u.durations[s];
        if (T > 0xffffffffull / voice->fperiod) {
            set_error("utterance too long");
            return JB_ERR_INVALID;
        }
        b->T[i] = (uint32_t)T;
        b->frame_off[i] = sumT;
        hu[i].S = u.num_states;
        hu[i].T = (uint32_t)T;
        hu[i].frame_off = sumT;
        hu[i].state_off = sumS;
        hu[i].mt_rs = (uint32_t)((T + 15) / 16 * 16);
        hu[i].mt_off = sum_mt;
        sum_mt += hu[i].mt_rs;
        sumT += T;
        sumS += u.num_states;
        maxT = std::max(maxT, (uint32_t)T);
        maxS = std::max(maxS, u.num_states);
        const void *dp;
        if ((rc = b->upload(u.durations, sizeof(uint32_t) * u.num_states, &dp)))
            return rc;
        hu[i].dur = (const uint32_t *)dp;
        for (uint32_t si = 0; si < voice->nstream; si++) {
            const jb_stream_states &hs = u.stream[si];
            const jb_stream_desc &sd = voice->stream[si];
            const size_t WL = (size_t)sd.vector_length * sd.num_windows;
            StreamStatesDev &ds = hu[i].st[si];
            if (idx) {
                const StreamStatesDev &g = gathered[i * voice->nstream + si];
                ds.mean = g.mean;
                ds.var = g.var;
                ds.msd = g.msd;
            } else if (trk) {
                ds.mean = ds.var = nullptr; // no MLPG: the tracks are given
                if ((rc = b->upload(hs.msd, sizeof(double) * u.num_states, &dp)))
                    return rc;
                ds.msd = (const double *)dp;
            } else {
                if (u.num_states && (!hs.mean || !hs.var)) {
                    set_error("stream mean/var missing");
                    return JB_ERR_INVALID;
                }
                if ((rc = b->upload(hs.mean, sizeof(double) * WL * u.num_states, &dp)))
                    return rc;
                ds.mean = (const double *)dp;
                if ((rc = b->upload(hs.var, sizeof(double) * WL * u.num_states, &dp)))
                    return rc;
                ds.var = (const double *)dp;
                if ((rc = b->upload(hs.msd, sizeof(double) * u.num_states, &dp)))
                    return rc;
                ds.msd = (const double *)dp;
            }
            ds.gv_mean = ds.gv_var = nullptr;
            ds.gv_switch = nullptr;
            if (sd.use_gv && hs.gv_mean && hs.gv_var && hs.gv_switch) {
                if ((rc = b->upload(hs.gv_mean, sizeof(double) * sd.vector_length, &dp)))
                    return rc;
                ds.gv_mean = (const double *)dp;
                if ((rc = b->upload(hs.gv_var, sizeof(double) * sd.vector_length, &dp)))
                    return rc;
                ds.gv_var = (const double *)dp;
                if ((rc = b->upload(hs.gv_switch, u.num_states, &dp)))
                    return rc;
                ds.gv_switch = (const uint8_t *)dp;
            }
            ds.gv_weight = hs.gv_weight;
            ds.msd_threshold = hs.msd_threshold;
        }
    }
    b->frame_off[n] = sumT;
    b->sumT = sumT;
    b->maxT = maxT;
    // distinct utterances (copies share their uploaded / gathered arrays): what the vocoder's warm-up length goes by
    {
        std::set<std::tuple<const void *, uint32_t, const void *, const void *>> seen_utts;
        b->first_of_kind.assign(n, 0);
        for (size_t i = 0; i < n; i++)
            b->first_of_kind[i] = seen_utts.emplace((const void *)hu[i].dur, hu[i].S, (const void *)hu[i].st[0].mean,
                                                    (const void *)(voice->nstream > 1 ? hu[i].st[1].mean : nullptr)).second;
    }
    // one inverse-variance table per distinct variance array (256 copies of an utterance: one table of 2.5 MB that
    // stays in L2 instead of 256 of them, 640 MB, behind the build's gathers)
    for (uint32_t si = 0; si < voice->nstream; si++) {
        std::map<std::pair<const void *, uint32_t>, size_t> first;
        for (size_t i = 0; i < n; i++) {
            auto it = first.emplace(std::make_pair((const void *)hu[i].st[si].var, hu[i].S), i).first;
            const bool shared = hu[i].st[si].var != nullptr && it->second != i;
            hu[i].ivar_state_off[si] = shared ? hu[it->second].state_off : hu[i].state_off;
            hu[i].ivar_owner[si] = shared ? 0 : 1;
        }
    }

    cmark("descriptors, uploads staged");
    if ((rc = b->flush_uploads()))
        return rc;
    cmark("uploads flushed");
    UttDev *dutt;
    if ((rc = b->stage(hu.data(), n, &dutt)))
        return rc;
    // launch order: longest utterance first (LPT within the GPU)
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t c) { return b->T[a] > b->T[c]; });
    uint32_t *dord;
    if ((rc = b->stage(order.data(), n, &dord)))
        return rc;
    b->bd.B = B;
    b->bd.utt = dutt;
    b->bd.order = dord;
    b->bd.maxT = maxT;
    b->bd.maxS = maxS;

    // ---- per-stream scratch + MLPG workspace ----
    for (uint32_t si = 0; si < voice->nstream; si++) {
        const jb_stream_desc &hs = voice->stream[si];
        StreamDev &sd = b->sd[si];
        memset(&sd, 0, sizeof sd);
        sd.L = (int)hs.vector_length;
        sd.W = (int)hs.num_windows;
        sd.is_msd = trk ? (si == 1) : (int)hs.is_msd;
        sd.use_gv = trk ? 0 : (int)hs.use_gv;
        if (trk)
            sd.W = 1; // the window description is not read without MLPG (and may be absent)
        int off = 0, maxw = trk ? 1 : 0;
        for (int w = 0; !trk && w < sd.W; w++) {
            sd.win_width[w] = (int)hs.win_width[w];
            sd.win_off[w] = off;
            for (int k = 0; k < sd.win_width[w]; k++)
                sd.win_coef[off + k] = hs.win_coef[off + k];
            off += sd.win_width[w];
            maxw = std::max(maxw, sd.win_width[w]);
        }
        sd.BW = (maxw / 2) * 2 + 1; // Windows::max_width()*2+1 (window.rs:19-21, mlpg.rs:27)
        sd.generic_solver = (b->flags & JB_BATCH_GENERIC_MLPG) ? 1 : 0;
        sd.serial_gv = (b->flags & JB_BATCH_SERIAL_GV) ? 1 : 0;
        // [dim][frame] workspace with the fused kernels: band width 3, up to three windows (the sliding-window
        // build), 3..60 dims; everything else takes the generic reference-shaped kernels
        sd.mt = (sd.BW == 3 && sd.W <= 3 && !sd.generic_solver && sd.L > 2 && sd.L <= mlpg_mt_max_dim()) ? 1 : 0;
        // MCP, non-MSD, [dim][frame]: its transpose is fused with mc2b (enqueue_paramgen)
        // (Stage::NonZero reads the [frame][dim] track itself: k_stage_coef)
        sd.defer_out = (si == 0 && sd.mt && !sd.is_msd && voice->stage == 0 && !trk) ? 1 : 0;
        const size_t nf = (size_t)sumT, nfl = std::max((size_t)sumT, (size_t)sum_mt) * (size_t)sd.L, nst = (size_t)sumS;
        if ((rc = b->dalloc(&sd.s_start, nst, false)) || (rc = b->dalloc(&sd.s_vpre, nst, false)) ||
            (rc = b->dalloc(&sd.s_rstart, nst, false)) || (rc = b->dalloc(&sd.s_rend, nst, false)) ||
            (rc = b->dalloc(&sd.s_voiced, nst, false)) || (rc = b->dalloc(&sd.run_list, nst, false)) ||
            (rc = b->dalloc(&sd.nruns, n, true)))
            return rc;
        if ((rc = b->dalloc(&sd.fstate, nf, false)) || (rc = b->dalloc(&sd.voiced, nf, false)) ||
            (rc = b->dalloc(&sd.fl, nf, false)) || (rc = b->dalloc(&sd.fr, nf, false)) ||
            (rc = b->dalloc(&sd.vidx, nf, false)) || (rc = b->dalloc(&sd.vsw, nf, false)) ||
            (rc = b->dalloc(&sd.Tv, n, true)) || (rc = b->dalloc(&sd.gvlen, n, true)))
            return rc;
        const bool is_static = sd.BW == 1 && sd.W == 1 && !sd.use_gv && !sd.generic_solver;
        if (!is_static && !trk) {
            for (int j = 0; j < sd.BW; j++)
                if ((rc = b->dalloc(&sd.A[j], nfl, false)) || (rc = b->dalloc(&sd.F[j], nfl, false)))
                    return rc;
            if ((rc = b->dalloc(&sd.bvec, nfl, false)) || (rc = b->dalloc(&sd.g, nfl, false)) ||
                (rc = b->dalloc(&sd.par, nfl, false)))
                return rc;
            if (sd.mt && (rc = b->dalloc(&sd.ivar, nst * (size_t)sd.W * (size_t)sd.L, false)))
                return rc;
            if (sd.mt && sd.use_gv && !sd.serial_gv) {
                sd.gv_ntile = (maxT + (uint32_t)mlpg_gv_tile_frames() - 1) / (uint32_t)mlpg_gv_tile_frames();
                const size_t nbl = n * (size_t)sd.L;
                if ((rc = b->dalloc(&sd.gv_part, 7 * nbl * (size_t)sd.gv_ntile * 4, false)) ||
                    (rc = b->dalloc(&sd.gv_scal, 6 * nbl * 4, false)))
                    return rc;
                // resident GV (one persistent launch, jb_gv_gang.hip) unless the CUs are partitioned (its
                // grid is sized for the whole device) or a row has more tiles than a gang can hold
                int tiles = 0, gangs = 0;
                if (maxT > 0 &&
                    gv_gang_plan(dev, maxT, (uint32_t)nbl, &tiles, &gangs)) {
                    uint8_t *ctl;
                    if ((rc = b->dalloc(&ctl, gv_gang_ctl_bytes(gangs), true)))
                        return rc;
                    // bins of utterances whose rows share a pass of a gang (the frames of an MSD stream's rows are
                    // counted on the device: its utterances keep a pass each, tiles by their upper bound)
                    std::vector<uint8_t> has_gv(n);
                    std::vector<uint32_t> Tb(n);
                    for (size_t i = 0; i < n; i++) {
                        has_gv[i] = hu[i].st[si].gv_mean != nullptr;
                        Tb[i] = sd.is_msd ? maxT : b->T[i];
                    }
                    std::vector<GvBinEntry> bins;
                    gv_gang_bins(Tb.data(), has_gv.data(), order.data(), n, tiles, bins);
                    GvBinEntry *dbins;
                    if ((rc = b->stage(bins.data(), bins.size(), &dbins)))
                        return rc;
                    sd.gv_bins = dbins;
                    sd.gv_nbins = (uint32_t)(bins.size() / (size_t)tiles);
                    if ((uint32_t)gangs > sd.gv_nbins * (uint32_t)sd.L)
                        gangs = (int)(sd.gv_nbins * (uint32_t)sd.L);
                    sd.gv_gang_ctl = ctl;
                    sd.gv_gang_n = gangs;
                    sd.gv_gang_tiles = tiles;
                }
            }
        }
        if ((rc = b->dalloc(&sd.out, nfl, false)))
            return rc;
        if (trk) // the tracks themselves: [frame][dim] per utterance at frame_off * L
            for (size_t i = 0; i < n; i++) {
                const jb_track_utt &t = trk->utts[i];
                const double *src = si == 0 ? t.spectrum : si == 1 ? t.lf0 : t.lpf;
                const size_t ne = (size_t)b->T[i] * (size_t)sd.L;
                if (ne && (e = hipMemcpy(sd.out + (size_t)b->frame_off[i] * (size_t)sd.L, src, ne * sizeof(double),
                                         hipMemcpyHostToDevice)) != hipSuccess)
                    return hip_fail(e, "hipMemcpy(H2D tracks)");
            }
    }

    cmark("stream workspaces allocated");
    // ---- vocoder ----
    VocDev &vd = b->vd;
    memset(&vd, 0, sizeof vd);
    vd.fs = (int)voice->sampling_frequency;
    vd.fperiod = (int)voice->fperiod;
    vd.nmcp = (int)voice->stream[0].vector_length;
    vd.nlpf = (int)voice->stream[2].vector_length;
    // A frame's samples in blocks of bs <= 64 (one pulse-mask word and one wave pass of lane = sample per block).
    // The largest divisor of the frame period that is <= 64 where there is a useful one (every BASELINE shape: 240 ->
    // 4 x 60; the split excitation kernels and the lane-triple vocoder are built on equal blocks); otherwise -- a
    // prime frame period, 75 = 3 x 25 under a 31-tap filter -- blocks of ceil(fperiod / nblk) samples with a shorter
    // LAST block (block q = samples [q bs, min(fperiod, (q + 1) bs)): sample i is bit i % bs of word i / bs either way).
    int bs = std::min(64, vd.fperiod);
    while (vd.fperiod % bs)
        bs--;
    if ((bs < vd.nlpf - 1 || bs < 16) && bs < std::min(64, vd.fperiod)) {
        const int nblk = (vd.fperiod + 63) / 64;
        bs = (vd.fperiod + nblk - 1) / nblk;
    }
    vd.bs = bs;
    vd.nblk = (vd.fperiod + bs - 1) / bs;
    vd.alpha = voice->alpha;
    vd.volume = voice->volume;
    // postfilter_mcp acts only for beta > 0 and more than two coefficients (cepstrum.rs:24)
    vd.beta = (voice->beta > 0.0 && vd.nmcp > 2 && voice->stage == 0) ? voice->beta : 0.0;
    vd.stage = (int)voice->stage;
    vd.use_log_gain = voice->use_log_gain ? 1 : 0;
    vd.beta_stage = voice->stage ? voice->beta : 0.0; // postfilter_lsp (lsp.rs:113-139)
    vd.voiced = b->sd[1].voiced;
    vd.run_list = b->sd[1].run_list;
    vd.nruns = b->sd[1].nruns;
    if ((rc = b->dalloc(&vd.run_base, n + 1, true)) || (rc = b->dalloc(&vd.run_counter, 1, true)))
        return rc;
    vd.mcp = b->sd[0].out;
    vd.lf0 = b->sd[1].out;
    vd.lpf = b->sd[2].out;
    const size_t nf = (size_t)sumT;
    const bool mlpg_only = (b->flags & JB_BATCH_MLPG_ONLY) != 0; // no excitation, no PCM: no slabs for them
    b->total_samples = nf * (size_t)vd.fperiod;
    if ((rc = b->dalloc(&vd.bcoef, nf * (size_t)vd.nmcp, false)) ||
        (rc = b->dalloc(&vd.pitch, nf, false)) || (rc = b->dalloc(&vd.cur_start, nf, false)) ||
        (rc = b->dalloc(&vd.pinc, nf, false)) ||
        (rc = b->dalloc(&vd.pmask, nf * (size_t)vd.nblk, false)) ||
        (rc = b->dalloc(&vd.xin, mlpg_only ? 1 : b->total_samples, false)))
        return rc;
    if (vd.stage > 0 && (rc = b->dalloc(&vd.bfirst, (size_t)std::max<size_t>(n, 1) * (size_t)vd.nmcp, true)))
        return rc;
    if (vd.beta > 0.0) {
        if ((rc = b->dalloc(&vd.bfirst, (size_t)std::max<size_t>(n, 1) * (size_t)vd.nmcp, true)) ||
            (rc = b->dalloc(&vd.pf_table, (size_t)vd.nmcp * 576, false)) || (rc = b->dalloc(&vd.pf_rcp, 576, false)))
            return rc;
    }
    if (b->flags & JB_BATCH_PCM_I16)
        rc = b->dalloc(&vd.pcm16, mlpg_only ? 1 : b->total_samples, false);
    else
        rc = b->dalloc(&vd.pcm, mlpg_only ? 1 : b->total_samples, false);
    if (rc)
        return rc;
    if ((b->flags & JB_BATCH_KEEP_TRACKS) && !mlpg_only)
        if ((rc = b->dalloc(&vd.exc, b->total_samples, false)))
            return rc;
    // frames whose excitation is the noise stream itself are not stored (not with the debug tap, which
    // wants every sample, nor with the one-pass kernel for other tap counts / frame periods)
    vd.skip_unvoiced = (!vd.exc && excite_is_split(vd)) ? 1 : 0;
    if (vd.nlpf == 0 && !mlpg_only && (rc = b->dalloc(&vd.uv_before, std::max<size_t>(nf, 1), false)))
        return rc; // ring-buffer-less excitation: unvoiced frames before each frame = its place in the noise stream
    if (excite_is_split(vd) && !mlpg_only) {
        // shared pulse-free excitation (jb_device.h): the table, the per-frame source codes, the work list of the
        // per-frame pass.  With the debug tap (every sample wanted as computed per utterance) and on request
        // (JB_BATCH_NO_EXC_TABLE: A/B tests) every frame goes through the per-frame pass.
        vd.exc_no_table = (vd.exc || (b->flags & JB_BATCH_NO_EXC_TABLE)) ? 1 : 0;
        {
            // which frames carry the canonical taps: tracked by the LPF MLPG itself where that is the one-window
            // kernel (a byte per frame instead of a second pass over the track); else k_exc_classify compares rows
            StreamDev &sl = b->sd[2];
            const bool is_static = sl.BW == 1 && sl.W == 1 && !sl.use_gv && !sl.generic_solver;
            if (!trk && is_static && !vd.exc_no_table && nf) {
                if ((rc = b->dalloc(&sl.canon, nf, false)))
                    return rc;
                sl.canon_n = nf;
                sl.canon_ref_utt = 0;
                for (size_t i = 0; i < n; i++)
                    if (b->T[i] > 0) {
                        sl.canon_ref_utt = (uint32_t)i; // owns the batch's first frame (frame_off 0)
                        break;
                    }
                vd.lpf_canon = sl.canon;
                // the canonical rows themselves need not be stored unless the track is a result
                if (sl.L <= 32 && !(b->flags & JB_BATCH_KEEP_TRACKS)) {
                    sl.canon_skip_rows = 1;
                    vd.lpf_sparse = 1;
                }
            }
        }
        if ((rc = b->dalloc(&vd.exc_tab, std::max<size_t>((size_t)maxT * (size_t)vd.fperiod, 1), false)) ||
            (rc = b->dalloc(&vd.exc_src, std::max<size_t>(nf, 1), true)) ||
            (rc = b->dalloc(&vd.exc_gen, 2 * std::max<size_t>(nf, 1), false)) ||
            (rc = b->dalloc(&vd.exc_gen_count, 1, true)))
            return rc;
    }
    vd.state_stride = vd.stage > 0 ? mglsa_state_doubles(vd.stage) : vocoder_state_doubles(vd.nmcp);
    if ((rc = b->dalloc(&vd.state, (size_t)vd.state_stride * n, true)))
        return rc;
    if ((rc = noise_table(dev, (size_t)maxT * (size_t)vd.fperiod, &b->noise)))
        return rc;
    vd.noise = b->noise->ptr;
    vd.noise_len = b->noise->len;
    if (vd.beta > 0.0 && (e = launch_pf_table(vd, b->stream)) != hipSuccess)
        return hip_fail(e, "k_pf_table");
    cmark("vocoder buffers allocated");
    if ((rc = b->build_work(opts)))
        return rc;
    if ((rc = b->flush_zero())) // (legacy stream, like the uploads: in order with them)
        return rc;
    if ((rc = b->flush_uploads()))
        return rc;
    for (auto &c : b->up_chunks)
        c.host.reset(); // staging copies are not needed any more
    // uploads and memsets ran on the legacy stream, the gather and the constant tables on the batch's
    // own: wait for those two, not for the device (another batch may be running a step)
    cmark("work list built");
    e = hipStreamSynchronize(nullptr);
    if (e == hipSuccess)
        e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess)
        return hip_fail(e, "upload");
    cmark("synchronised");
    *out = b.release();
    return JB_OK;
}

// Work list of the vocoder: serial = one item per utterance; chunked = items of
// chunk_frames output frames that start warmup_frames early from zero state.
int Batch::build_work(const jb_batch_opts *opts)
{
    const bool serial = (flags & JB_BATCH_SERIAL) != 0;
    // 18 frames.  Every frame of warm-up is 0.6 % of the vocoder kernel (0.35 ms on config 2); a failing
    // hand-off costs a redo round (~2.5 ms whatever their number).  Same box, config 2 (copies of one utterance
    // / 64 distinct utterances), ms per step: 20 frames 92.2 / 94.1, 19: 91.2 / 93.3, 18: 90.5 / 93.6,
    // 17: 93.0 / 92.6 (at 17 a hand-off of config 2's own utterance fails in all 256 copies; with distinct
    // utterances a few hundred fail at every length and settle at their checkpoint).  24 -> 20 earlier in the
    // round: vocoder 67.8 -> 66.3 ms, 136 -> 220 failing hand-offs with distinct utterances.
    // Round 4, the other side of that trade: wherever a redo round is certain anyway -- any batch with more than a
    // few hundred DISTINCT hand-off positions has failing ones at every length up to ~40 frames -- a shorter warm-up
    // is cheaper as long as the failing chunks still get a SIMD each in the redo launch (<= 1024): same box, ms per
    // step at 18 / 16 / 14 / 12 frames: 512 mixed lengths 84.5 / 83.8 / 83.2 / 84.7, 1024 x 6,386 distinct 83.5 /
    // 82.6 / 82.3 / 83.5, 64 distinct x 4 copies 82.4 / 81.3 / 80.5 / 82.2, 64 x 2,000 10.9 / 9.6 / 9.2 (at 14:
    // 1.7 % of the hand-offs fail, 720-810 chunks; at 12: 1,400-1,470, two to a SIMD).  Copies of ONE utterance
    // keep 18: their 167 positions fail for all copies or for none, and over the FAMILY of such batches 18 is the
    // cheaper length -- round 5, BASELINE config 2 for the utterances of seeds 0..3, ms per step at 18 / 14 frames, same
    // box: 76.6 / 78.4, 79.5 / 80.5 (three positions fail at 18, seven at 14), 79.3 / 78.3, 76.2 / 78.1; mean 77.9
    // against 78.8 (profiles/r05_seed_sweep_before.txt; two of the four have no failing position at 18, all have at
    // 14).  Decided below, once the chunk length is known: 14 frames from 1000 distinct hand-off positions (the chance
    // that none of them fails at 18 frames is then under 0.1 %), else 18 -- small requests keep the geometry they had.
    const bool warmup_given = opts && opts->warmup_frames;
    warmup_frames = warmup_given ? opts->warmup_frames : 18;
    verify_tol = (opts && opts->verify_tol > 0.0) ? opts->verify_tol : kDefaultVerifyTol;
    uint32_t ch = opts ? opts->chunk_frames : 0;
    // lane-triple throughput kernel: worth it once the batch holds enough frames to give every SIMD two
    // waves of 21 chunks that are long against the warm-up (measured crossover against the wave kernel at
    // one item per SIMD: between 6 and 8 utterances of 25.5 k frames -- 22.4 vs 26.7 ms at 6, 28.5 vs 26.9
    // at 8)
    constexpr uint64_t lp_min = 100000;
    // (the lane-triple kernel walks the samples of a frame two at a time)
    lp_mode = !serial && !(flags & JB_BATCH_WAVE_KERNEL) && vd.stage == 0 && vocoder_ls_supported(vd.nmcp) &&
              (vd.fperiod & 1) == 0 &&
              (sumT >= lp_min || (flags & JB_BATCH_LANE_KERNEL));
    if (serial) {
        ch = 0;
    } else if (ch == 0 && lp_mode) {
        // two waves on every SIMD: 8 XCDs x 32 CUs x 4 SIMDs x 2 -- or ONE, while the batch is too small to give
        // every SIMD two waves of chunks that are long against their warm-up.  The launch takes as long as one
        // chunk-with-warm-up at the rate a wave gets: a lone wave issues an instruction every 6-6.7 cycles, one of a
        // pair every 8.9-9.7 (tools/lt_clocks.sh); compare the two at the chunk length each would get (floor below).
        constexpr uint64_t cfloor = 16;
        const uint64_t slots1 = 64ull * 16 * (uint64_t)vocoder_ls_chunks_per_wave(vd.nmcp);
        auto launch_cost = [&](uint64_t slots, double us_per_sample) {
            return (double)(std::max<uint64_t>((sumT + slots - 1) / slots, cfloor) + warmup_frames) * us_per_sample;
        };
        // (per sample and wave, measured: 0.90 us alone on a SIMD, 1.39-1.46 us beside a second wave -- 64 x 11,000
        // frames 18.1 ms per step with one wave per SIMD and 33-frame chunks, 19.8 with two and 17-frame chunks)
        // (the warm-up the chunks will get -- 14 frames from 1000 distinct hand-off positions, decided for good
        // below -- enters the comparison: estimated here from the chunk length two waves per SIMD would give)
        if (!warmup_given) {
            const uint64_t c2 = std::max<uint64_t>((sumT + 2 * slots1 - 1) / (2 * slots1), cfloor);
            uint64_t positions = 0;
            for (int i = 0; i < B; i++)
                if ((size_t)i >= first_of_kind.size() || first_of_kind[(size_t)i])
                    positions += (T[(size_t)i] + c2 - 1) / c2;
            if (positions >= 1000)
                warmup_frames = 14;
        }
        lt_waves_per_simd = launch_cost(slots1, 0.90) < launch_cost(2 * slots1, 1.42) ? 1 : 2;
        const uint64_t target = slots1 * (uint64_t)lt_waves_per_simd;
        uint64_t c = (sumT + target - 1) / target;
        // while the batch cannot fill the chip the time of the launch is that of ONE chunk (chunk +
        // warm-up frames): chunks down to 16 frames.  Shorter chunks mean more hand-off positions and
        // more of them failing the check, but a failed 16-frame chunk is also redone in a third of the
        // time of a 48-frame one; with DISTINCT utterances (bench.py --distinct 32) 16 beats the earlier
        // floor of twice the warm-up on every shape tried -- 32 x 25,546 frames 59.0 -> 43.0 ms per
        // step, 64 x 4,600 32.7 -> 23.0, 256 x 2,000 33.4 -> 23.5, 1024 x 500 22.8 -> 21.0 -- and 12 or
        // 8 gain nothing more.  (The earlier floor had been tuned on copies of one utterance, whose
        // hand-offs all pass.)
        constexpr uint64_t cmin = cfloor;
        ch = (uint32_t)std::max<uint64_t>(c, cmin);
        // (no rounding of the chunk length: 153 frames instead of 156 on config 2 is 1.7 % fewer frames per
        // chunk-with-warm-up and still fits the chip -- 42,752 items for 43,008 slots)
        // every utterance rounds its chunk count up: with ragged lengths the items can exceed the two
        // waves per SIMD the target stands for, and the waves over the limit run as a tail after the
        // others -- lengthen the chunks until the items fit
        auto items_at = [&](uint32_t cf) {
            uint64_t it = 0;
            for (int i = 0; i < B; i++)
                it += (T[(size_t)i] + cf - 1) / cf;
            return it;
        };
        for (int guard = 0; guard < 256 && c >= cmin && items_at(ch) > target; guard++)
            ch += 1;
        // (still more items than one wave per SIMD holds: the second wave takes them rather than a tail launch)
        if (lt_waves_per_simd == 1 && items_at(ch) > slots1)
            lt_waves_per_simd = 2;
    } else if (ch == 0) {
        // auto (wave kernel): one item per SIMD, two once the batch is large.  The launch takes as long
        // as ONE item (warm-up + chunk frames at 0.25 us per sample; 0.47 with two items on a SIMD), so a
        // small batch wants short chunks -- down to 16 frames (one 1.4 s sentence: 24.7 -> 17.5 ms per
        // call; below 16 the extra hand-off positions and their occasional redo round cost more than
        // they save) -- but never more items than SIMDs: the kernel's four-wave workgroups are what puts
        // exactly one on each.  (64 x 2000 frames: 16.0 ms with 1344 items of 96 frames, 11.6 with 1000 of 128.)
        // Round 5: a request of ONE or a few sentences fills a fraction of the SIMDs whatever its chunk length, and its
        // time is that of one item = (chunk + 18 warm-up frames) x 240 samples x 0.25 us: shorter chunks down to 6-8
        // frames, while the hand-off positions stay few enough for a redo round to be rare (same box, chunk 16 / 8 /
        // 6 / 4 frames at 18 of warm-up, ms per run: the reference's three benchmark sentences -- 277, 420, 742 frames
        // -- 2.38 / 1.88 / 1.75 / 1.62, 2.47 / 1.97 / 1.85 / 1.73, 3.66 / 2.68 / 2.47 / 2.54; 8 x 400 frames 2.59 /
        // 2.68 / 2.43 / 2.20; one utterance of 2,000 frames 3.86 / 2.89 / 3.12 / 3.12: tools/small_geometry_sweep.py,
        // profiles/r05_small_geometry_sweep.txt).  A shorter warm-up does not pay there: at 10 frames and below the
        // failing hand-offs cost a redo round more often than the frames saved.
        const uint64_t target = sumT >= 400000 ? 2048 : 1024;
        const uint64_t floor_w = sumT < 1024 ? 6 : sumT < 8192 ? 8 : 16;
        ch = (uint32_t)std::max<uint64_t>((sumT + target - 1) / target, floor_w);
        if (ch >= 16)
            ch = (ch + 7) / 8 * 8;
    }
    chunk_frames = ch;
    if (!warmup_given && ch != 0) {
        warmup_frames = 18;
        uint64_t positions = 0;
        for (int i = 0; i < B; i++)
            if ((size_t)i >= first_of_kind.size() || first_of_kind[(size_t)i])
                positions += (T[(size_t)i] + ch - 1) / ch;
        if (positions >= 1000)
            warmup_frames = 14;
    }
    if (lp_mode && ch != 0) {
        // (a chunk length given by the caller: one wave per SIMD if the items fit)
        uint64_t it = 0;
        for (int i = 0; i < B; i++)
            it += (T[(size_t)i] + ch - 1) / ch;
        if (opts && opts->chunk_frames)
            lt_waves_per_simd = it <= 64ull * 16 * (uint64_t)vocoder_ls_chunks_per_wave(vd.nmcp) ? 1 : 2;
    }
    // the checkpoint a failed chunk is first recomputed to (finish_verify): 48 frames into chunks of 96 and more, 24 into
    // chunks of 36 and more, 16 into chunks of 24 and more (a single 128 s utterance, 799 chunks of 32 frames: all six
    // failing hand-offs settle there and the redo is one round of 16 frames, 10.2 -> 9.2 ms per call; 8 frames into
    // 16-frame chunks settle three in four but the rest still take their rounds: same time, not done)
    vd.ckpt_frames = ch >= 2 * kVocCkptFrames ? kVocCkptFrames
                     : ch >= kVocCkptFramesShort + 12 ? kVocCkptFramesShort
                     : ch >= kVocCkptFramesTiny + 8 ? kVocCkptFramesTiny : 0;
    vd.ckpt2_frames = (vd.ckpt_frames == kVocCkptFrames && ch >= kVocCkpt2Frames + 48) ? kVocCkpt2Frames : 0;
#ifdef JB_DBG_GATES
    if (const char *c1 = getenv("JB_DBG_CKPT1")) // measurement aid: another first checkpoint for long chunks
        if (vd.ckpt2_frames && atoi(c1) >= 16 && atoi(c1) + 12 <= (int)vd.ckpt2_frames)
            vd.ckpt_frames = (uint32_t)atoi(c1);
#endif
    work.clear();
    const int stride = vd.state_stride;
    for (int i = 0; i < B; i++) {
        const uint32_t Ti = T[(size_t)i];
        if (Ti == 0)
            continue;
        if (ch == 0 || Ti <= ch + warmup_frames) {
            work.push_back(VocWork{(uint32_t)i, 0, 0, Ti, nullptr, nullptr, nullptr});
            continue;
        }
        for (uint32_t t0 = 0; t0 < Ti; t0 += ch) {
            VocWork w{};
            w.utt = (uint32_t)i;
            w.t_out = t0;
            w.t_start = t0 > warmup_frames ? t0 - warmup_frames : 0;
            w.t_end = std::min(Ti, t0 + ch);
            work.push_back(w);
        }
    }
    // longest utterances first for the serial case; chunk items are uniform
    if (ch == 0)
        std::stable_sort(work.begin(), work.end(), [&](const VocWork &a, const VocWork &c) {
            return (a.t_end - a.t_start) > (c.t_end - c.t_start);
        });
    n_items = (uint32_t)work.size();
    int rc;
    if ((rc = dalloc(&bad_dev, n_items, true)) ||
        (rc = dalloc(&nbad_dev, 1, true)))
        return rc;
    if (ch != 0) {
        // zeroed: slots of the state layout that a kernel does not write must compare equal
        if ((rc = dalloc(&end_state, (size_t)n_items * stride, true)) ||
            (rc = dalloc(&warm_state, (size_t)n_items * stride, true)) ||
            (rc = dalloc(&ckpt_state, (size_t)n_items * stride, true)) ||
            (vd.ckpt2_frames && (rc = dalloc(&ckpt2_state, (size_t)n_items * stride, true))))
            return rc;
        state_stride = stride;
        for (uint32_t k = 0; k < n_items; k++) {
            VocWork &w = work[k];
            const bool first = w.t_out == 0;
            const bool single = first && w.t_end == T[w.utt];
            if (single)
                continue;
            w.save_end = end_state + (size_t)k * stride;
            w.save_warm = first ? nullptr : warm_state + (size_t)k * stride;
            // checkpoint for the partial redo, wherever the chunk goes on for at least 12 frames behind it.  (A
            // redo round lasts as long as its longest item: when only chunks of twice the checkpoint had one, the
            // short last chunk of an utterance -- up to 95 frames recomputed to their end -- made the round of a
            // batch of distinct utterances 5.7 ms instead of the 2.9 ms of 48 frames.)
            const uint32_t need = vd.ckpt_frames + (vd.ckpt_frames < kVocCkptFramesShort ? 8u : 12u);
            w.save_ckpt = (!first && vd.ckpt_frames && w.t_end - w.t_out >= need) ? ckpt_state + (size_t)k * stride
                                                                                   : nullptr;
            w.save_ckpt2 = (w.save_ckpt && vd.ckpt2_frames && w.t_end - w.t_out >= vd.ckpt2_frames + 12u)
                               ? ckpt2_state + (size_t)k * stride : nullptr;
        }
    }
    if ((rc = stage(work.data(), n_items, &work_dev))) // (with the arena's next flush: create() ends with one)
        return rc;
    if (lp_mode) {
        // launch permutation: equal-length chunks share a wave (lanes run in lock step)
        std::vector<uint32_t> ord(n_items);
        std::iota(ord.begin(), ord.end(), 0u);
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) {
            return (work[x].t_end - work[x].t_start) > (work[y].t_end - work[y].t_start);
        });
        if ((rc = stage(ord.data(), n_items, &order_dev)))
            return rc;
    }
    return JB_OK;
}

// Streaming generator (utterance 0 only): one work item per frame, state carried in vd.state
int Batch::build_generator_work()
{
    if (gen_work_dev || B == 0 || T[0] == 0)
        return JB_OK;
    std::vector<VocWork> gw(T[0]);
    for (uint32_t t = 0; t < T[0]; t++) {
        gw[t] = VocWork{0, t, t, t + 1, t > 0 ? vd.state : nullptr, nullptr, vd.state};
    }
    int rc = dalloc(&gen_work_dev, gw.size(), false);
    if (rc)
        return rc;
    // the serially served head goes to a buffer of its own: the whole-utterance run writes the same frames of
    // vd.pcm meanwhile (and, from the throughput kernel, not bit for bit the same values)
    if (vd.pcm && (rc = dalloc(&gen_pcm, (size_t)vd.fperiod * 64, false)))
        return rc;
    hipError_t e = hipMemcpy(gen_work_dev, gw.data(), sizeof(VocWork) * gw.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess)
        return hip_fail(e, "hipMemcpy(generator work)");
    return JB_OK;
}

int Batch::enqueue_vocoder()
{
    hipError_t e;
    if (lp_mode)
        e = launch_vocoder_ls(bd, vd, work_dev, order_dev, n_items, lt_waves_per_simd, stream_voc);
    else
        e = launch_vocoder(bd, vd, work_dev, n_items, stream_voc);
    if (e != hipSuccess)
        return hip_fail(e, "k_vocoder");
    return JB_OK;
}

// The three MlpgAdjust::create calls (src/engine.rs:333-357) are independent: MCP on the
// main stream, LF0 -> pitch -> pulse schedule and LPF on side streams, forked after
// whatever the main stream was doing (a previous run may still read the tracks) and
// joined before the vocoder.
// Hook of the MCP chain, called between its band solve and its GV: enqueues the LPF chain (width-1 static
// window: one bandwidth-bound kernel) and, behind it on the same side stream, the pulse-free excitation
// pass: both start with the step, beside the MCP build and band solve.  (Until the band solve shed a third
// of its traffic, the excitation pass beside it stretched both by more than its own time and ran on the
// main stream between band solve and GV: 98.8 ms per step that way, 97.8 this way.  With JB_ONE_STREAM the
// side streams are the main stream and the order is simply sequential.)
#ifdef JB_DBG_GATES
static int dbg_sched() { const char *s = getenv("JB_DBG_SCHED"); return s ? atoi(s) : 0; }
#endif
static hipError_t excite_noise_hook(void *ctx, hipStream_t stream)
{
    Batch *b = (Batch *)ctx;
    hipError_t e;
    (void)stream;
    hipEventRecord(b->ev_fb, stream); // the MCP band solve is enqueued up to here (the hook runs between it and the GV)
    // the LPF chain starts behind the MCP chain's inverse-variance pass: both stream at HBM rate, and side by
    // side (with the LF0 band solve) the pass on the critical chain took 3.1 ms instead of 0.5 (-0.45 ms per step).
    // (Behind the MCP BUILD instead -- JB_DBG_SCHED bit 2 of the -DJB_DBG_GATES library: -0.5 ms per step on one
    // box, +0.45 on another for config 2, -0.7 for 1024 x 6,386 frames: not kept.)
    hipStreamWaitEvent(b->stream_lpf, b->ev_ivar, 0);
#ifdef JB_DBG_GATES
    if (dbg_sched() & 2)
        hipStreamWaitEvent(b->stream_lpf, b->ev_mcpbuild, 0);
    if (dbg_sched() & 1) { // the LF0 chain's MLPG, pitch and pulse walk behind the MCP build
        hipStreamWaitEvent(b->stream_lf0, (dbg_sched() & 4) ? b->ev_fb : b->ev_mcpbuild, 0);
        if ((e = launch_mlpg(b->bd, b->sd[1], 1, b->stream_lf0, nullptr)) != hipSuccess ||
            (e = launch_pitch(b->bd, b->vd, b->stream_lf0)) != hipSuccess ||
            (e = launch_pulse(b->bd, b->vd, b->stream_lf0)) != hipSuccess)
            return e;
    }
#endif
    if (b->voice.nstream > 2) {
        if ((e = launch_prep(b->bd, b->sd[2], 2, b->stream_lpf)) != hipSuccess)
            return e;
        if ((e = launch_mlpg(b->bd, b->sd[2], 2, b->stream_lpf, nullptr)) != hipSuccess)
            return e;
    }
    hipEventRecord(b->ev_lpf, b->stream_lpf);
    hipStreamWaitEvent(b->stream_lpf, b->ev_prep, 0); // voiced flags (LF0 state walk)
    e = launch_excite_noise(b->bd, b->vd, b->stream_lpf);
    hipEventRecord(b->ev_build, b->stream_lpf); // "pulse-free excitation done"
    return e;
}

// MlpgAdjust::create x3 and nothing behind it (JB_BATCH_MLPG_ONLY; jb_mlpg_batch): the three chains as
// in enqueue_paramgen without pitch, pulses and excitation; the MCP track is turned to [frame][dim].
int Batch::enqueue_mlpg_only()
{
    hipError_t e;
    hipEventRecord(ev_fork, stream);
    hipStreamWaitEvent(stream_lf0, ev_fork, 0);
    hipStreamWaitEvent(stream_lpf, ev_fork, 0);
    if ((e = launch_prep(bd, sd[1], 1, stream_lf0)) != hipSuccess ||
        (e = launch_mlpg(bd, sd[1], 1, stream_lf0, nullptr)) != hipSuccess)
        return hip_fail(e, "k_mlpg(lf0)");
    hipEventRecord(ev_lf0, stream_lf0);
    if (voice.nstream > 2) {
        if ((e = launch_prep(bd, sd[2], 2, stream_lpf)) != hipSuccess ||
            (e = launch_mlpg(bd, sd[2], 2, stream_lpf, nullptr)) != hipSuccess)
            return hip_fail(e, "k_mlpg(lpf)");
    }
    hipEventRecord(ev_lpf, stream_lpf);
    if ((e = launch_prep(bd, sd[0], 0, stream)) != hipSuccess ||
        (e = launch_mlpg(bd, sd[0], 0, stream, ev_mcpbuild)) != hipSuccess)
        return hip_fail(e, "k_mlpg(mcp)");
    if (sd[0].defer_out && (e = launch_mc2b_mt(bd, sd[0], vd, true, stream)) != hipSuccess)
        return hip_fail(e, "k_mc2b_mt");
    hipStreamWaitEvent(stream, ev_lf0, 0);
    hipStreamWaitEvent(stream, ev_lpf, 0);
    return JB_OK;
}

// SpeechGenerator::new + generate_all on GIVEN parameter tracks (jb_vocode_tracks_batch): the LF0 chain
// starts at the state walk over the track's voiced / unvoiced runs (voiced flags, run list), then pitch
// and pulse schedule; the MCP chain is mc2b (or the Stage::NonZero conversion) straight from the track.
int Batch::enqueue_from_tracks()
{
    hipError_t e;
    hipEventRecord(ev_fork, stream);
    hipStreamWaitEvent(stream_lf0, ev_fork, 0);
    hipStreamWaitEvent(stream_lpf, ev_fork, 0);
    if ((e = launch_prep(bd, sd[1], 1, stream_lf0)) != hipSuccess)
        return hip_fail(e, "k_prep(lf0)");
    hipEventRecord(ev_prep, stream_lf0);
    if ((e = launch_pitch(bd, vd, stream_lf0)) != hipSuccess)
        return hip_fail(e, "k_pitch");
    if ((e = launch_pulse(bd, vd, stream_lf0)) != hipSuccess)
        return hip_fail(e, "k_pulse");
    hipEventRecord(ev_lpf, stream_lpf); // the LPF track is resident
    hipStreamWaitEvent(stream_lpf, ev_prep, 0);
    if ((e = launch_excite_noise(bd, vd, stream_lpf)) != hipSuccess)
        return hip_fail(e, "k_excite(noise)");
    hipEventRecord(ev_build, stream_lpf);
    e = vd.stage > 0 ? launch_stage_coef(bd, vd, stream) : launch_mc2b(bd, vd, stream);
    if (e != hipSuccess)
        return hip_fail(e, "k_mc2b");
    if (vd.beta > 0.0 && (e = launch_postfilter(bd, vd, (uint64_t)sumT, stream)) != hipSuccess)
        return hip_fail(e, "k_postfilter");
    hipStreamWaitEvent(stream_lf0, ev_lpf, 0);
    hipStreamWaitEvent(stream_lf0, ev_build, 0);
    if ((e = launch_excite(bd, vd, stream_lf0)) != hipSuccess)
        return hip_fail(e, "k_excite");
    hipEventRecord(ev_lf0, stream_lf0);
    hipStreamWaitEvent(stream, ev_lf0, 0);
    return JB_OK;
}

int Batch::enqueue_paramgen()
{
    if (from_tracks)
        return enqueue_from_tracks();
    if (flags & JB_BATCH_MLPG_ONLY)
        return enqueue_mlpg_only();
    hipError_t e;
    hipEventRecord(ev_fork, stream);
    hipStreamWaitEvent(stream_lf0, ev_fork, 0);
    hipStreamWaitEvent(stream_lpf, ev_fork, 0);
    // LF0 chain: state walk (voiced flags), MLPG, pitch, pulse schedule
    if ((e = launch_prep(bd, sd[1], 1, stream_lf0)) != hipSuccess)
        return hip_fail(e, "k_prep(lf0)");
    hipEventRecord(ev_prep, stream_lf0); // voiced flags of the LF0 stream
#ifdef JB_DBG_GATES
    if (!(dbg_sched() & 1))
#endif
    {
    if ((e = launch_mlpg(bd, sd[1], 1, stream_lf0, nullptr)) != hipSuccess)
        return hip_fail(e, "k_mlpg(lf0)");
    if ((e = launch_pitch(bd, vd, stream_lf0)) != hipSuccess)
        return hip_fail(e, "k_pitch");
    if ((e = launch_pulse(bd, vd, stream_lf0)) != hipSuccess)
        return hip_fail(e, "k_pulse");
    }
    // MCP chain (the critical path); between its band solve and its GV sweeps the hook enqueues
    // the LPF chain (side stream) and the pulse-free excitation pass (main stream)
    if ((e = launch_prep(bd, sd[0], 0, stream)) != hipSuccess)
        return hip_fail(e, "k_prep(mcp)");
    if ((e = launch_mlpg(bd, sd[0], 0, stream, ev_mcpbuild, excite_noise_hook, this, ev_ivar)) != hipSuccess)
        return hip_fail(e, "k_mlpg(mcp) / k_excite(noise)");
    if (vd.stage > 0)
        e = launch_stage_coef(bd, vd, stream); // Stage::NonZero: LSP track -> MGLSA coefficients
    else if (sd[0].defer_out)
        e = launch_mc2b_mt(bd, sd[0], vd, (flags & JB_BATCH_KEEP_TRACKS) != 0, stream);
    else
        e = launch_mc2b(bd, vd, stream);
    if (e != hipSuccess)
        return hip_fail(e, "k_mc2b");
    if (vd.beta > 0.0 && (e = launch_postfilter(bd, vd, (uint64_t)sumT, stream)) != hipSuccess)
        return hip_fail(e, "k_postfilter");
    // pulses (LF0): the samples after each pulse (split form) or the whole excitation
    hipStreamWaitEvent(stream_lf0, ev_lpf, 0);
    hipStreamWaitEvent(stream_lf0, ev_build, 0);
#ifdef JB_DBG_GATES
    if (dbg_sched() & 64)
        hipStreamWaitEvent(stream_lf0, ev_fb, 0);
#endif
    if ((e = launch_excite(bd, vd, stream_lf0)) != hipSuccess)
        return hip_fail(e, "k_excite");
    hipEventRecord(ev_lf0, stream_lf0);
    hipStreamWaitEvent(stream, ev_lf0, 0);
    return JB_OK;
}

int Batch::run(bool timed)
{
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess)
        return hip_fail(e, "hipSetDevice");
    // a previous run's vocoder may still read what this run's parameter generation rewrites
    hipStreamWaitEvent(stream, ev_voc_done, 0);
    last_run_timed = timed;
    for (int si = 0; si < kMaxStream; si++)
        if (sd[si].gv_gang_ctl && !from_tracks)
            gang_check_pending = true;
    const bool inject_timeout = (flags & JB_BATCH_TEST_GANG_TIMEOUT) && gang_check_pending && gang_fallbacks == 0;
    if (timed)
        hipEventRecord(ev0, stream);
    int rc = enqueue_paramgen();
    if (rc)
        return rc;
    if (inject_timeout) // test aid: as if the resident GV kernel had given up in formation
        for (int si = 0; si < kMaxStream; si++)
            if (sd[si].gv_gang_ctl)
                hipMemsetAsync(&((GvGangCtl *)sd[si].gv_gang_ctl)->err, 1, 1, stream);
    hipEventRecord(ev_mlpg_done, stream);
    hipStreamWaitEvent(stream_voc, ev_mlpg_done, 0);
    if (timed)
        hipEventRecord(ev1, stream_voc);
    if (flags & JB_BATCH_MLPG_ONLY) { // MlpgAdjust::create only: the tracks are the result
        if (timed) {
            hipEventRecord(ev2, stream_voc);
            hipEventRecord(ev3, stream_voc);
        }
        hipEventRecord(ev_voc_done, stream_voc);
        return JB_OK;
    }
    if ((rc = enqueue_vocoder()))
        return rc;
    if (timed)
        hipEventRecord(ev2, stream_voc);
    if (chunk_frames != 0 && n_items > 0) {
        hipMemsetAsync(nbad_dev, 0, sizeof(uint32_t), stream_voc);
        if ((e = launch_voc_verify(work_dev, n_items, vd.state_stride, vd.stage > 0 ? -1 : vd.nmcp - 1, verify_tol, bad_dev,
                                   nbad_dev, stream_voc)) != hipSuccess)
            return hip_fail(e, "k_voc_verify");
        verify_pending = true;
    }
    if (timed)
        hipEventRecord(ev3, stream_voc);
    hipEventRecord(ev_voc_done, stream_voc);
    return JB_OK;
}

// After the stream has drained: act on the chunk hand-off check.  Chunks whose
// warmed-up state disagrees with the predecessor's end state are recomputed from that
// end state (exact continuation), in increasing order so that each re-do starts from
// a final state.
int Batch::finish_verify()
{
    if (!verify_pending)
        return JB_OK;
    verify_pending = false;
    const auto tv0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv0).count(); };
    uint32_t nbad = 0;
    hipError_t e = hipMemcpy(&nbad, nbad_dev, sizeof nbad, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return hip_fail(e, "hipMemcpy(n_bad)");
    n_redo = nbad;
    if (nbad == 0)
        return JB_OK;
    std::vector<uint8_t> bad(n_items);
    if ((e = hipMemcpy(bad.data(), bad_dev, n_items, hipMemcpyDeviceToHost)) != hipSuccess)
        return hip_fail(e, "hipMemcpy(bad)");
    // rounds: every failing chunk whose predecessor is final (not itself pending) is
    // recomputed in the same launch; runs of consecutive failures take one round each.
    // A chunk with a checkpoint is first recomputed only up to it (vd.ckpt_frames frames): if the
    // recomputed state meets the checkpoint the original chunk left there, the rest of that chunk
    // was computed from a trajectory that had already converged and stands; otherwise the
    // recomputation continues from there to the end of the chunk.
    n_redo_partial = n_redo_full = n_recert_failed = 0;
    const size_t stride = state_stride;
    int rc;
    if (!redo_dev && (rc = dalloc(&redo_dev, n_items, false)))
        return rc;
    if (!tmp_state && ((rc = dalloc(&tmp_state, (size_t)n_items * stride, true)) ||
                       (rc = dalloc(&pairs_dev, 2 * (size_t)n_items, false)) || (rc = flush_zero())))
        return rc;
    auto run_round = [&](const std::vector<VocWork> &round) -> int {
        if (round.empty())
            return JB_OK;
        hipError_t he;
        const double t_a = since();

        if ((he = hipMemcpy(redo_dev, round.data(), sizeof(VocWork) * round.size(), hipMemcpyHostToDevice)) !=
            hipSuccess)
            return hip_fail(he, "hipMemcpy(redo)");
        const double t_b = since();
        if ((he = launch_vocoder(bd, vd, redo_dev, (uint32_t)round.size(), stream_voc)) != hipSuccess)
            return hip_fail(he, "k_vocoder(redo)");
        const double t_c = since();
        if ((he = hipStreamSynchronize(stream_voc)) != hipSuccess)
            return hip_fail(he, "redo sync");
        if (getenv("JB_REDO_TRACE"))
            fprintf(stderr, "  {copy %.3f launch %.3f wait %.3f} ", t_b - t_a, t_c - t_b, since() - t_c);
        return JB_OK;
    };
    std::vector<uint8_t> pending(bad);
    pending[0] = 0;
    const bool redo_trace = getenv("JB_REDO_TRACE") != nullptr; // debugging aid: one line per redo round on stderr
    if (redo_trace)
        fprintf(stderr, "redo: lists and scratch ready after %.3f ms\n", since());
    for (;;) {
        // This round: every failing chunk whose predecessor is final, AND -- speculatively -- a failing chunk
        // behind a failing chunk: it starts from the end state its predecessor left in the FIRST pass.  That state
        // stands if the predecessor settles at its checkpoint (295 of 297 do; stage A of a chunk with a checkpoint
        // does not touch the dump), or, for a chunk without a checkpoint, if the end state of its recomputation
        // -- written to a scratch dump, compared after the launch, then copied over -- meets it to the hand-off
        // tolerance (the trajectory that started wrong has had warm-up + chunk frames to converge: every one of
        // 156 did in a batch of 16 x 25,546 frames).  Where it does not stand, the successor's recomputation
        // started from a state that is being replaced: it stays pending for the next round.  (Before, a run of
        // consecutive failures cost one round per chunk: 3 ms each with checkpoints, 1.2 ms each for 20-frame
        // chunks, two or three rounds per step in small batches.)
        std::vector<uint32_t> ids;
        std::vector<uint8_t> in_round(n_items, 0);
        for (uint32_t k = 1; k < n_items; k++) {
            if (!pending[k])
                continue;
            if (!pending[k - 1] || (in_round[k - 1] && work[k - 1].save_end && work[k - 1].utt == work[k].utt)) {
                ids.push_back(k);
                in_round[k] = 1;
            }
        }
        if (ids.empty())
            break;
        // stage A: up to the checkpoint (or the whole chunk where there is none)
        std::vector<VocWork> round;
        std::vector<uint32_t> part; // positions in ids whose recomputed state is compared: with a checkpoint, or
                                    // (spec_end) recomputed to the end with a successor in this round
        std::vector<uint8_t> spec_end(n_items, 0);
        for (size_t j = 0; j < ids.size(); j++) {
            const uint32_t k = ids[j];
            VocWork w = work[k];
            w.t_start = w.t_out;
            w.load_state = work[k - 1].save_end;
            w.save_warm = nullptr;
            if (w.save_ckpt) {
                w.t_end = w.t_out + vd.ckpt_frames;
                w.save_end = tmp_state + (size_t)k * stride;
                part.push_back((uint32_t)j);
            } else if (k + 1 < n_items && in_round[k + 1] && work[k + 1].utt == w.utt && w.save_end) {
                spec_end[k] = 1;
                w.save_end = tmp_state + (size_t)k * stride;
                part.push_back((uint32_t)j);
            }
            w.save_ckpt = w.save_ckpt2 = nullptr;
            round.push_back(w);
        }
        const double tr0 = since();
        if ((rc = run_round(round)))
            return rc;
        if (redo_trace)
            fprintf(stderr, "  [stage A: built by %.3f, ran until %.3f ms] ", tr0, since());
        // does the recomputed state meet the checkpoint?
        std::vector<uint8_t> unsettled(n_items, 0);
        if (!part.empty()) {
            std::vector<const double *> pairs(2 * part.size());
            for (size_t q = 0; q < part.size(); q++) {
                const uint32_t k = ids[part[q]];
                pairs[2 * q] = tmp_state + (size_t)k * stride;
                pairs[2 * q + 1] = spec_end[k] ? work[k].save_end : work[k].save_ckpt;
            }
            hipMemcpy(pairs_dev, pairs.data(), sizeof(double *) * pairs.size(), hipMemcpyHostToDevice);
            hipMemsetAsync(nbad_dev, 0, sizeof(uint32_t), stream_voc);
            if ((e = launch_voc_verify_pairs(pairs_dev, (uint32_t)part.size(), vd.state_stride, vd.stage > 0 ? -1 : vd.nmcp - 1, verify_tol, bad_dev,
                                             nbad_dev, stream_voc)) != hipSuccess)
                return hip_fail(e, "k_voc_verify_pairs");
            std::vector<uint8_t> bad2(part.size());
            if ((e = hipMemcpyAsync(bad2.data(), bad_dev, part.size(), hipMemcpyDeviceToHost, stream_voc)) !=
                    hipSuccess ||
                (e = hipStreamSynchronize(stream_voc)) != hipSuccess)
                return hip_fail(e, "hipMemcpy(bad2)");
            for (size_t q = 0; q < part.size(); q++)
                unsettled[ids[part[q]]] = bad2[q];
        }
        // second checkpoint: a chunk that had not converged at the first one goes on to it and is compared again
        std::vector<uint8_t> at2(n_items, 0);
        if (vd.ckpt2_frames) {
            std::vector<VocWork> mid;
            std::vector<uint32_t> mid_ids;
            for (uint32_t k : ids)
                if (work[k].save_ckpt && work[k].save_ckpt2 && unsettled[k]) {
                    if (!tmp2_state && ((rc = dalloc(&tmp2_state, (size_t)n_items * stride, true)) || (rc = flush_zero())))
                        return rc;
                    VocWork w = work[k];
                    w.t_start = w.t_out = work[k].t_out + vd.ckpt_frames;
                    w.t_end = work[k].t_out + vd.ckpt2_frames;
                    w.load_state = tmp_state + (size_t)k * stride;
                    w.save_end = tmp2_state + (size_t)k * stride;
                    w.save_warm = w.save_ckpt = w.save_ckpt2 = nullptr;
                    mid.push_back(w);
                    mid_ids.push_back(k);
                }
            if (!mid.empty()) {
                if ((rc = run_round(mid)))
                    return rc;
                std::vector<const double *> pairs(2 * mid_ids.size());
                for (size_t q = 0; q < mid_ids.size(); q++) {
                    pairs[2 * q] = tmp2_state + (size_t)mid_ids[q] * stride;
                    pairs[2 * q + 1] = work[mid_ids[q]].save_ckpt2;
                }
                hipMemcpy(pairs_dev, pairs.data(), sizeof(double *) * pairs.size(), hipMemcpyHostToDevice);
                hipMemsetAsync(nbad_dev, 0, sizeof(uint32_t), stream_voc);
                if ((e = launch_voc_verify_pairs(pairs_dev, (uint32_t)mid_ids.size(), vd.state_stride, vd.stage > 0 ? -1 : vd.nmcp - 1,
                                                 verify_tol, bad_dev, nbad_dev, stream_voc)) != hipSuccess)
                    return hip_fail(e, "k_voc_verify_pairs(second checkpoint)");
                std::vector<uint8_t> badm(mid_ids.size());
                if ((e = hipMemcpyAsync(badm.data(), bad_dev, mid_ids.size(), hipMemcpyDeviceToHost, stream_voc)) != hipSuccess ||
                    (e = hipStreamSynchronize(stream_voc)) != hipSuccess)
                    return hip_fail(e, "hipMemcpy(bad, second checkpoint)");
                for (size_t q = 0; q < mid_ids.size(); q++) {
                    at2[mid_ids[q]] = 1;
                    unsettled[mid_ids[q]] = badm[q];
                }
                if (redo_trace)
                    fprintf(stderr, "  second checkpoint: %zu chunks, %zu still not converged\n", mid_ids.size(),
                            (size_t)std::count_if(badm.begin(), badm.end(), [](uint8_t x) { return x != 0; }));
            }
        }
        // in chunk order: what this round's recomputations are worth
        std::vector<VocWork> rest;      // stage B: the rest of valid chunks that had not converged at their checkpoint
        std::vector<uint32_t> full_ids; // chunks recomputed to their end in this round
        std::vector<uint8_t> final_now(n_items, 0);
        for (uint32_t k : ids) {
            // valid: started from a final state -- the predecessor was final before the round, or it was in the
            // round, valid itself, and its first-pass end state stands (settled at its checkpoint / met by the end
            // state of its recomputation)
            const bool pred_in = in_round[k - 1] && pending[k - 1];
            const bool valid = !pred_in || (final_now[k - 1] && (work[k - 1].save_ckpt || spec_end[k - 1]) && !unsettled[k - 1]);
            if (!valid)
                continue; // stays pending: next round, from the state its predecessor is getting now
            final_now[k] = 1;
            if (!work[k].save_ckpt) {
                n_redo_full++;
                full_ids.push_back(k);
            } else if (!unsettled[k]) {
                n_redo_partial++;
            } else {
                n_redo_full++;
                full_ids.push_back(k);
                VocWork w = work[k];
                w.t_start = w.t_out = work[k].t_out + (at2[k] ? vd.ckpt2_frames : vd.ckpt_frames);
                w.load_state = (at2[k] ? tmp2_state : tmp_state) + (size_t)k * stride;
                w.save_warm = nullptr;
                w.save_ckpt = w.save_ckpt2 = nullptr;
                rest.push_back(w);
            }
        }
        // the exact end state of a VALID chunk recomputed to its end goes where the first pass left its own (an
        // invalid one started from a state that is being replaced: its dump keeps the first pass's state, the better
        // start for its successor's next attempt)
        for (uint32_t k : ids)
            if (final_now[k] && spec_end[k] &&
                (e = hipMemcpyAsync(work[k].save_end, tmp_state + (size_t)k * stride, sizeof(double) * stride,
                                    hipMemcpyDeviceToDevice, stream_voc)) != hipSuccess)
                return hip_fail(e, "hipMemcpy(end state)");
        if ((rc = run_round(rest)))
            return rc;
        for (uint32_t k : ids)
            if (final_now[k])
                pending[k] = 0;
        if (redo_trace)
            fprintf(stderr, "  (%.3f ms since the check was read) ", since());
        if (redo_trace)
            fprintf(stderr, "redo round: %zu items (%zu to a checkpoint), %zu continue past it, %u settled so far, %u to the end so far\n",
                    ids.size(), part.size(), rest.size(), n_redo_partial, n_redo_full);
        // Re-certification.  Chunk k+1 was checked against the end state chunk k left in the first
        // pass -- the end of a trajectory now known to have started wrong.  Where chunk k has been
        // recomputed to its end, that dump now holds the exact state: compare it with the warm state of
        // chunk k+1 again and put k+1 on the list if it fails.  (A chunk settled at its checkpoint kept
        // its first-pass end state, whose trajectory was certified at the checkpoint: nothing to re-check.)
        std::vector<uint32_t> succ;
        for (uint32_t k : full_ids)
            // (a successor that was recomputed in this round started from this chunk's end state: nothing to re-check
            // if that was valid, and it is still pending if not)
            if (k + 1 < n_items && work[k + 1].utt == work[k].utt && work[k + 1].save_warm && !pending[k + 1] &&
                !in_round[k + 1] && work[k].save_end)
                succ.push_back(k + 1);
        if (!succ.empty()) {
            std::vector<const double *> pairs(2 * succ.size());
            for (size_t q = 0; q < succ.size(); q++) {
                pairs[2 * q] = work[succ[q]].save_warm;
                pairs[2 * q + 1] = work[succ[q] - 1].save_end;
            }
            hipMemcpy(pairs_dev, pairs.data(), sizeof(double *) * pairs.size(), hipMemcpyHostToDevice);
            hipMemsetAsync(nbad_dev, 0, sizeof(uint32_t), stream_voc);
            if ((e = launch_voc_verify_pairs(pairs_dev, (uint32_t)succ.size(), vd.state_stride, vd.stage > 0 ? -1 : vd.nmcp - 1, verify_tol,
                                             bad_dev, nbad_dev, stream_voc)) != hipSuccess)
                return hip_fail(e, "k_voc_verify_pairs(successors)");
            std::vector<uint8_t> bad3(succ.size());
            if ((e = hipMemcpyAsync(bad3.data(), bad_dev, succ.size(), hipMemcpyDeviceToHost, stream_voc)) !=
                    hipSuccess ||
                (e = hipStreamSynchronize(stream_voc)) != hipSuccess)
                return hip_fail(e, "hipMemcpy(bad3)");
            if (redo_trace)
                fprintf(stderr, "  re-certified %zu successors: %zu fail\n", succ.size(),
                        (size_t)std::count_if(bad3.begin(), bad3.end(), [](uint8_t x) { return x != 0; }));
            for (size_t q = 0; q < succ.size(); q++)
                if (bad3[q]) {
                    pending[succ[q]] = 1;
                    n_redo++;
                    n_recert_failed++;
                }
        }
    }
    return JB_OK;
}

int Batch::sync()
{
    hipError_t e = hipSetDevice(device); // the redo rounds of finish_verify launch kernels
    if (e == hipSuccess)
        e = hipStreamSynchronize(stream_voc);
    if (e == hipSuccess && stream_voc != stream)
        e = hipStreamSynchronize(stream);
    if (e != hipSuccess)
        return hip_fail(e, "stream sync");
    // The persistent GV kernel bounds its spins; an overrun is reported, never waited out.  The flag is read
    // only when such a kernel has been enqueued since the last look (a blocking 4-byte copy per GV stream:
    // a loop of per-utterance reads must not pay it every time).
    bool gang_timed_out = false;
    for (int si = 0; gang_check_pending && si < kMaxStream; si++)
        if (sd[si].gv_gang_ctl) {
            uint32_t err = 0;
            uint32_t *derr = &((GvGangCtl *)sd[si].gv_gang_ctl)->err;
            if ((e = hipMemcpy(&err, derr, sizeof err, hipMemcpyDeviceToHost)) != hipSuccess)
                return hip_fail(e, "hipMemcpy(gv gang err)");
            if (err) {
                // Formation timed out: with several resident launches on one device each can hold CUs with
                // incomplete gangs while none owns all its members.  This batch's GV runs as the multi-launch
                // sweeps from now on (their workspace is allocated either way) and the step is done again.
                (void)hipMemset(derr, 0, sizeof err);
                sd[si].gv_gang_ctl = nullptr;
                gang_timed_out = true;
                gang_fallbacks++;
            } else if (getenv("JB_GG_PROFILE_PRINT")) { // library built with -DJB_GG_PROFILE
                GvGangCtl c;
                hipMemcpy(&c, sd[si].gv_gang_ctl, sizeof c, hipMemcpyDeviceToHost);
                const double nb = (double)sd[si].gv_gang_n * sd[si].gv_gang_tiles;
                fprintf(stderr,
                        "k_mlpg_gv_gang ticks per workgroup: load %.0f stats %.0f blocksum %.0f exchange %.0f (hand-off alone "
                        "%.0f) step %.0f store %.0f\n",
                        c.prof[0] / nb, c.prof[1] / nb, c.prof[2] / nb, c.prof[3] / nb, c.prof[6] / nb, c.prof[4] / nb,
                        c.prof[5] / nb);
#if JB_GG_PROFILE
                // who a gang waits for: per exchange of gangs 0, 9, 18 the arrival of each tile (its sums ready) behind the
                // gang's first arrival, and how long after the LAST arrival the last tile had the exchange over (100 MHz clock)
                const int nt = std::min(sd[si].gv_gang_tiles, 8);
                for (int gi : {0, 9, 18}) {
                    if (gi >= sd[si].gv_gang_n)
                        continue;
                    GvGang gg;
                    hipMemcpy(&gg, (const uint8_t *)sd[si].gv_gang_ctl + sizeof(GvGangCtl) + sizeof(GvGang) * (size_t)gi, sizeof gg,
                              hipMemcpyDeviceToHost);
                    fprintf(stderr, "gang %d: exchange | arrival of tiles 0..%d behind the first, us | last arrival -> all done, us | since previous exchange, us\n", gi, nt - 1);
                    unsigned long long prev_first = 0;
                    for (int e = 0; e < 96; e++) {
                        unsigned long long first = ~0ull, last = 0, done = 0;
                        for (int t = 0; t < nt; t++) {
                            first = std::min(first, gg.stamp[0][t][e]);
                            last = std::max(last, gg.stamp[0][t][e]);
                            done = std::max(done, gg.stamp[1][t][e]);
                        }
                        if (first == 0)
                            break;
                        fprintf(stderr, "  %2d |", e);
                        for (int t = 0; t < nt; t++)
                            fprintf(stderr, " %5.2f", (double)(gg.stamp[0][t][e] - first) / 100.0);
                        fprintf(stderr, " | %5.2f | %6.2f\n", (double)(done - last) / 100.0,
                                prev_first ? (double)(first - prev_first) / 100.0 : 0.0);
                        prev_first = first;
                    }
                }
#endif
            }
        }
    gang_check_pending = false;
    if (gang_timed_out) {
        verify_pending = false; // what that run certified was computed from a GV that did not finish
        int rc = run(last_run_timed);
        return rc ? rc : sync();
    }
    return finish_verify();
}

int Batch::gang_timeout_seen(bool *seen)
{
    *seen = false;
    for (int si = 0; gang_check_pending && si < kMaxStream; si++)
        if (sd[si].gv_gang_ctl) {
            uint32_t err = 0;
            hipError_t e = hipMemcpy(&err, &((GvGangCtl *)sd[si].gv_gang_ctl)->err, sizeof err, hipMemcpyDeviceToHost);
            if (e != hipSuccess)
                return hip_fail(e, "hipMemcpy(gv gang err)");
            if (err)
                *seen = true;
        }
    return JB_OK;
}

// Every read entry waits for the batch's own streams first (they are non-blocking streams: a plain
// hipMemcpy does not order behind them) and lets the hand-off certification finish, so that what
// is read is the certified result.  Cheap when nothing is pending.
int Batch::read(const void *dev, void *dst, size_t bytes, bool do_sync)
{
    int rc = do_sync ? sync() : JB_OK;
    if (rc)
        return rc;
    hipError_t e = hipMemcpy(dst, dev, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return hip_fail(e, "hipMemcpy(D2H)");
    return JB_OK;
}

// --------------------------------------------------------------------------
// D2H of the whole PCM slab into one host buffer per utterance (Engine::synthesize's Vec<f64>,
// src/engine.rs:294).  A pageable hipMemcpy per utterance runs at ~13 GB/s here and takes its page
// faults on the copying thread; instead the slab streams through a ring of pinned slots
// (hipMemcpyAsync at link rate) while worker threads copy finished slots into the callers'
// buffers, first-touching them in parallel.  One ring per device, kept for the process.
namespace {
constexpr size_t kStageSlot = 4u << 20;
constexpr int kStageSlots = 32, kStageWorkers = 8;
// Measured on 3.9 GB (64 x 157 s of f64): the link alone 54 GB/s; with the scatter into fresh 4 KB
// pages 13 GB/s whatever the worker count (page faults); into MADV_HUGEPAGE buffers
// (jb_synthesize_batch) 45 GB/s with 4 to 24 workers.
struct StageRing {
    void *slot[kStageSlots] = {};
    hipEvent_t ev[kStageSlots] = {};
    hipStream_t stream = nullptr;
    std::mutex mu; // one reader at a time per device (the link is shared anyway)
};
std::mutex g_stage_mu;
std::map<int, std::unique_ptr<StageRing>> g_stage;

int stage_ring(int device, StageRing **out)
{
    std::lock_guard<std::mutex> lk(g_stage_mu);
    auto &r = g_stage[device];
    if (!r) {
        std::unique_ptr<StageRing> n(new StageRing());
        hipError_t e = hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking);
        for (int i = 0; i < kStageSlots && e == hipSuccess; i++) {
            e = hipHostMalloc(&n->slot[i], kStageSlot, hipHostMallocDefault);
            if (e == hipSuccess)
                e = hipEventCreateWithFlags(&n->ev[i], hipEventDisableTiming);
        }
        if (e != hipSuccess)
            return hip_fail(e, "pinned staging ring");
        r = std::move(n);
    }
    *out = r.get();
    return JB_OK;
}
} // namespace

int Batch::read_pcm_split(void *const *dst, size_t elem)
{
    const char *slab = elem == 2 ? (const char *)vd.pcm16 : (const char *)vd.pcm;
    if (flags & JB_BATCH_MLPG_ONLY) {
        set_error("a JB_BATCH_MLPG_ONLY batch has no PCM");
        return JB_ERR_INVALID;
    }
    if (!slab) {
        set_error(elem == 2 ? "16-bit PCM needs JB_BATCH_PCM_I16" : "f64 PCM was replaced by the 16-bit sink");
        return JB_ERR_INVALID;
    }
    const size_t total = total_samples * elem;
    if (total == 0)
        return JB_OK;
    int rc = sync();
    if (rc)
        return rc;
    hipError_t e = hipSuccess;
    StageRing *ring = nullptr;
    if ((rc = stage_ring(device, &ring)))
        return rc;
    std::lock_guard<std::mutex> lk(ring->mu);
    // byte offset of every utterance in the slab (utterances are contiguous, in batch order)
    std::vector<size_t> uoff((size_t)B + 1);
    for (int u = 0; u <= B; u++)
        uoff[u] = (size_t)frame_off[u] * voice.fperiod * elem;
    const size_t nchunks = (total + kStageSlot - 1) / kStageSlot;
    if (nchunks <= 4 && nchunks <= (size_t)kStageSlots) {
        // a small request (one sentence: 0.5 MB; a 21 s text: 8 MB): its few copies issued at once and scattered on
        // the calling thread as they land -- no worker threads to start and join (0.1-0.2 ms of such a call)
        for (size_t c = 0; c < nchunks && e == hipSuccess; c++) {
            const size_t lo = c * kStageSlot, n = std::min(total, lo + kStageSlot) - lo;
            e = hipMemcpyAsync(ring->slot[c], slab + lo, n, hipMemcpyDeviceToHost, ring->stream);
            if (e == hipSuccess)
                e = hipEventRecord(ring->ev[c], ring->stream);
        }
        for (size_t c = 0; c < nchunks && e == hipSuccess; c++) {
            e = hipEventSynchronize(ring->ev[c]);
            const size_t lo = c * kStageSlot, hi = std::min(total, lo + kStageSlot);
            size_t u = (size_t)(std::upper_bound(uoff.begin(), uoff.end(), lo) - uoff.begin()) - 1;
            for (; e == hipSuccess && u < (size_t)B && uoff[u] < hi; u++) {
                const size_t a = std::max(lo, uoff[u]), b2 = std::min(hi, uoff[u + 1]);
                if (b2 > a && dst[u])
                    memcpy((char *)dst[u] + (a - uoff[u]), (const char *)ring->slot[c] + (a - lo), b2 - a);
            }
        }
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(ring->stream); // (nothing of this call stays in flight into the ring)
            return hip_fail(e, "staged D2H");
        }
        return JB_OK;
    }
    enum { FREE = 0, ISSUED = 1 };
    std::atomic<int> state[kStageSlots];
    for (auto &st : state)
        st.store(FREE);
    std::atomic<int> failed{0};
    auto worker = [&](int k, int stride) {
        hipSetDevice(device);
        for (size_t c = (size_t)k; c < nchunks; c += (size_t)stride) {
            const int sl = (int)(c % kStageSlots);
            while (state[sl].load(std::memory_order_acquire) != ISSUED) {
                if (failed.load())
                    return;
                std::this_thread::yield();
            }
            if (hipEventSynchronize(ring->ev[sl]) != hipSuccess)
                failed.store(1);
            const size_t lo = c * kStageSlot, hi = std::min(total, lo + kStageSlot);
            // utterances overlapping [lo, hi)
            size_t u = (size_t)(std::upper_bound(uoff.begin(), uoff.end(), lo) - uoff.begin()) - 1;
            for (; u < (size_t)B && uoff[u] < hi; u++) {
                const size_t a = std::max(lo, uoff[u]), b2 = std::min(hi, uoff[u + 1]);
                if (b2 > a && dst[u])
                    memcpy((char *)dst[u] + (a - uoff[u]), (const char *)ring->slot[sl] + (a - lo), b2 - a);
            }
            state[sl].store(FREE, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    int nw = kStageWorkers;
    if (const char *ev = getenv("JB_STAGE_WORKERS"))
        nw = std::max(1, std::min(atoi(ev), (int)kStageSlots));
    nw = (int)std::min<size_t>((size_t)nw, nchunks);
    for (int k = 0; k < nw; k++)
        pool.emplace_back(worker, k, nw);
    for (size_t c = 0; c < nchunks && !failed.load(); c++) {
        const int sl = (int)(c % kStageSlots);
        while (state[sl].load(std::memory_order_acquire) != FREE)
            std::this_thread::yield();
        const size_t lo = c * kStageSlot, n = std::min(total, lo + kStageSlot) - lo;
        e = hipMemcpyAsync(ring->slot[sl], slab + lo, n, hipMemcpyDeviceToHost, ring->stream);
        if (e == hipSuccess)
            e = hipEventRecord(ring->ev[sl], ring->stream);
        if (e != hipSuccess) {
            failed.store(1);
            break;
        }
        state[sl].store(ISSUED, std::memory_order_release);
    }
    for (auto &t : pool)
        t.join();
    if (failed.load())
        return hip_fail(e != hipSuccess ? e : hipErrorUnknown, "staged D2H");
    return JB_OK;
}

} // namespace jb

using jb::Batch;
using jb::IndexSrc;
using jb::PdfSet;

extern "C" {

const char *jb_last_error(void) { return jb::g_err.c_str(); }

double jb_default_verify_tol(void) { return jb::kDefaultVerifyTol; }

const char *jb_version(void) { return "jbonsai_amd 0.1.0 (gfx950; reference jbonsai 0.4.2)"; }

int jb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int jb_device_arch(int dev, char *buf, size_t cap)
{
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess)
        return jb::hip_fail(e, "hipGetDeviceProperties");
    snprintf(buf, cap, "%s", p.gcnArchName);
    return JB_OK;
}

int jb_device_pci_bus_id(int dev, char *buf, size_t cap)
{
    if (!buf || cap < 13)
        return JB_ERR_BUFFER;
    hipError_t e = hipDeviceGetPCIBusId(buf, (int)cap, dev);
    if (e != hipSuccess)
        return jb::hip_fail(e, "hipDeviceGetPCIBusId");
    return JB_OK;
}

int jb_batch_create(const jb_voice_desc *voice, const jb_state_utt *utts, size_t n_utts,
                    const jb_batch_opts *opts, jb_batch **out)
{
    if (!out)
        return JB_ERR_INVALID;
    Batch *b = nullptr;
    int rc = Batch::create(voice, utts, n_utts, opts, &b);
    *out = (jb_batch *)b;
    return rc;
}

int jb_batch_create_from_tracks(const jb_voice_desc *voice, const jb_track_utt *utts, size_t n_utts,
                                const jb_batch_opts *opts, jb_batch **out)
{
    if (!out)
        return JB_ERR_INVALID;
    Batch *b = nullptr;
    jb::TrackSrc src{utts};
    int rc = Batch::create(voice, nullptr, n_utts, opts, &b, nullptr, &src);
    *out = (jb_batch *)b;
    return rc;
}

int jb_pdf_set_create(const jb_pdf_table *tables, uint32_t n_voices, uint32_t nstream, int32_t device,
                      jb_pdf_set **out)
{
    if (!out)
        return JB_ERR_INVALID;
    *out = nullptr;
    if (!tables || n_voices == 0 || n_voices > JB_MAX_VOICES || nstream == 0 || nstream > JB_MAX_STREAM) {
        jb::set_error("bad pdf table set");
        return JB_ERR_INVALID;
    }
    int dev = device;
    hipError_t e;
    if (dev < 0 && (e = hipGetDevice(&dev)) != hipSuccess)
        return jb::hip_fail(e, "hipGetDevice");
    if ((e = hipSetDevice(dev)) != hipSuccess)
        return jb::hip_fail(e, "hipSetDevice");
    std::unique_ptr<PdfSet> ps(new PdfSet());
    ps->device = dev;
    ps->nv = n_voices;
    ps->ns = nstream;
    const size_t nt = (size_t)n_voices * nstream;
    ps->tab.assign(nt, nullptr);
    ps->n_rows.assign(nt, 0);
    ps->row_len.assign(nt, 0);
    for (size_t t = 0; t < nt; t++) {
        const jb_pdf_table &pt = tables[t];
        if (!pt.rows || pt.n_rows == 0 || pt.row_len == 0) {
            jb::set_error("empty pdf table");
            return JB_ERR_INVALID;
        }
        const size_t bytes = (size_t)pt.n_rows * pt.row_len * sizeof(float);
        float *d = nullptr;
        if ((e = hipMalloc((void **)&d, bytes)) != hipSuccess)
            return jb::hip_fail(e, "hipMalloc(pdf table)");
        ps->tab[t] = d;
        if ((e = hipMemcpy(d, pt.rows, bytes, hipMemcpyHostToDevice)) != hipSuccess)
            return jb::hip_fail(e, "hipMemcpy(pdf table)");
        ps->n_rows[t] = pt.n_rows;
        ps->row_len[t] = pt.row_len;
    }
    *out = (jb_pdf_set *)ps.release();
    return JB_OK;
}

void jb_pdf_set_free(jb_pdf_set *s) { delete (PdfSet *)s; }

int jb_batch_create_indexed(const jb_voice_desc *voice, const jb_pdf_set *set, const jb_index_utt *utts,
                            size_t n_utts, const jb_batch_opts *opts, jb_batch **out)
{
    if (!out)
        return JB_ERR_INVALID;
    *out = nullptr;
    if (!voice || !set || (n_utts && !utts))
        return JB_ERR_INVALID;
    std::vector<jb_state_utt> su(n_utts);
    for (size_t i = 0; i < n_utts; i++) {
        su[i] = jb_state_utt{};
        su[i].num_states = utts[i].num_states;
        su[i].durations = utts[i].durations;
        for (uint32_t si = 0; si < voice->nstream && si < JB_MAX_STREAM; si++) {
            const jb_index_stream &is = utts[i].stream[si];
            jb_stream_states &o = su[i].stream[si];
            o.gv_mean = is.gv_mean;
            o.gv_var = is.gv_var;
            o.gv_switch = is.gv_switch;
            o.gv_weight = is.gv_weight;
            o.msd_threshold = is.msd_threshold;
        }
    }
    IndexSrc idx{(const PdfSet *)set, utts};
    Batch *b = nullptr;
    int rc = Batch::create(voice, su.data(), n_utts, opts, &b, &idx);
    *out = (jb_batch *)b;
    return rc;
}

int jb_batch_run(jb_batch *b) { return b ? ((Batch *)b)->run(true) : JB_ERR_INVALID; }
int jb_batch_sync(jb_batch *b) { return b ? ((Batch *)b)->sync() : JB_ERR_INVALID; }

int jb_batch_run_timed(jb_batch *hb, float *total_ms, float *vocoder_ms)
{
    if (!hb)
        return JB_ERR_INVALID;
    Batch *b = (Batch *)hb;
    int rc = b->run(true);
    if (rc)
        return rc;
    if ((rc = b->sync()))
        return rc;
    return jb_batch_last_timing(hb, total_ms, vocoder_ms);
}

int jb_batch_last_timing(jb_batch *hb, float *total_ms, float *vocoder_ms)
{
    if (!hb)
        return JB_ERR_INVALID;
    Batch *b = (Batch *)hb;
    float t = 0, v = 0;
    if (hipEventElapsedTime(&t, b->ev0, b->ev3) != hipSuccess ||
        hipEventElapsedTime(&v, b->ev1, b->ev2) != hipSuccess) {
        jb::set_error("no completed timed run");
        return JB_ERR_INVALID;
    }
    if (total_ms)
        *total_ms = t;
    if (vocoder_ms)
        *vocoder_ms = v;
    return JB_OK;
}

size_t jb_batch_size(const jb_batch *b) { return b ? (size_t)((const Batch *)b)->B : 0; }
size_t jb_batch_num_frames(const jb_batch *hb, size_t i)
{
    const Batch *b = (const Batch *)hb;
    return (b && i < (size_t)b->B) ? b->T[i] : 0;
}
size_t jb_batch_num_samples(const jb_batch *hb, size_t i)
{
    const Batch *b = (const Batch *)hb;
    return (b && i < (size_t)b->B) ? (size_t)b->T[i] * b->voice.fperiod : 0;
}
size_t jb_batch_total_samples(const jb_batch *b) { return b ? ((const Batch *)b)->total_samples : 0; }
size_t jb_batch_pcm_offset(const jb_batch *hb, size_t i)
{
    const Batch *b = (const Batch *)hb;
    return (b && i <= (size_t)b->B) ? (size_t)b->frame_off[i] * b->voice.fperiod : 0;
}
void *jb_batch_device_pcm(jb_batch *hb, size_t *n)
{
    Batch *b = (Batch *)hb;
    if (!b)
        return nullptr;
    if (n)
        *n = b->total_samples;
    if ((b->flags & JB_BATCH_MLPG_ONLY) || b->sync()) // the slab is handed out finished and certified
        return nullptr;
    return b->vd.pcm ? (void *)b->vd.pcm : (void *)b->vd.pcm16; // i16 slab for JB_BATCH_PCM_I16 batches
}

int jb_batch_read_pcm(jb_batch *hb, size_t i, double *dst, size_t cap)
{
    Batch *b = (Batch *)hb;
    if (!b || i >= (size_t)b->B)
        return JB_ERR_INVALID;
    if (b->flags & JB_BATCH_MLPG_ONLY) {
        jb::set_error("a JB_BATCH_MLPG_ONLY batch has no PCM");
        return JB_ERR_INVALID;
    }
    size_t ns = (size_t)b->T[i] * b->voice.fperiod;
    if (cap < ns) {
        jb::set_error("pcm buffer too small");
        return JB_ERR_BUFFER;
    }
    if (ns == 0)
        return JB_OK;
    if (!dst)
        return JB_ERR_INVALID;
    if (!b->vd.pcm) {
        jb::set_error("batch was created with JB_BATCH_PCM_I16: use jb_batch_read_pcm_i16");
        return JB_ERR_INVALID;
    }
    return b->read(b->vd.pcm + (size_t)b->frame_off[i] * b->voice.fperiod, dst, ns * sizeof(double));
}

int jb_batch_read_pcm_i16(jb_batch *hb, size_t i, int16_t *dst, size_t cap)
{
    Batch *b = (Batch *)hb;
    if (!b || i >= (size_t)b->B)
        return JB_ERR_INVALID;
    if (b->flags & JB_BATCH_MLPG_ONLY) {
        jb::set_error("a JB_BATCH_MLPG_ONLY batch has no PCM");
        return JB_ERR_INVALID;
    }
    if (!b->vd.pcm16) {
        jb::set_error("batch was created without JB_BATCH_PCM_I16");
        return JB_ERR_INVALID;
    }
    size_t ns = (size_t)b->T[i] * b->voice.fperiod;
    if (cap < ns) {
        jb::set_error("pcm buffer too small");
        return JB_ERR_BUFFER;
    }
    if (ns == 0)
        return JB_OK;
    if (!dst)
        return JB_ERR_INVALID;
    return b->read(b->vd.pcm16 + (size_t)b->frame_off[i] * b->voice.fperiod, dst, ns * sizeof(int16_t));
}

int jb_batch_read_pcm_all(jb_batch *hb, double *const *dst)
{
    Batch *b = (Batch *)hb;
    if (!b || (b->B && !dst))
        return JB_ERR_INVALID;
    return b->read_pcm_split((void *const *)dst, sizeof(double));
}

int jb_batch_read_pcm_i16_all(jb_batch *hb, int16_t *const *dst)
{
    Batch *b = (Batch *)hb;
    if (!b || (b->B && !dst))
        return JB_ERR_INVALID;
    return b->read_pcm_split((void *const *)dst, sizeof(int16_t));
}

int jb_batch_read_track(jb_batch *hb, size_t i, uint32_t si, double *dst, size_t cap)
{
    Batch *b = (Batch *)hb;
    if (!b || i >= (size_t)b->B || si >= b->voice.nstream)
        return JB_ERR_INVALID;
    if (!(b->flags & JB_BATCH_KEEP_TRACKS)) {
        jb::set_error("parameter tracks need JB_BATCH_KEEP_TRACKS");
        return JB_ERR_INVALID;
    }
    size_t L = (size_t)b->sd[si].L, ne = (size_t)b->T[i] * L;
    if (cap < ne)
        return JB_ERR_BUFFER;
    if (ne == 0)
        return JB_OK;
    return b->read(b->sd[si].out + (size_t)b->frame_off[i] * L, dst, ne * sizeof(double));
}

int jb_release_cached_memory(void)
{
    jb::release_cached_memory();
    return JB_OK;
}

int jb_set_cached_memory_limit(size_t megabytes)
{
    jb::set_cached_memory_limit(megabytes << 20);
    return JB_OK;
}

int jb_batch_read_coefficients(jb_batch *hb, size_t i, double *dst, size_t cap)
{
    Batch *b = (Batch *)hb;
    if (!b || i >= (size_t)b->B)
        return JB_ERR_INVALID;
    size_t L = (size_t)b->vd.nmcp, ne = (size_t)b->T[i] * L;
    if (cap < ne)
        return JB_ERR_BUFFER;
    if (ne == 0)
        return JB_OK;
    return b->read(b->vd.bcoef + (size_t)b->frame_off[i] * L, dst, ne * sizeof(double));
}

int jb_batch_read_first_coefficients(jb_batch *hb, size_t i, double *dst, size_t cap)
{
    Batch *b = (Batch *)hb;
    if (!b || i >= (size_t)b->B)
        return JB_ERR_INVALID;
    const size_t L = (size_t)b->vd.nmcp;
    if (cap < L)
        return JB_ERR_BUFFER;
    if (b->T[i] == 0)
        return JB_OK;
    if (b->vd.bfirst)
        return b->read(b->vd.bfirst + i * L, dst, L * sizeof(double));
    return b->read(b->vd.bcoef + (size_t)b->frame_off[i] * L, dst, L * sizeof(double));
}

int jb_batch_read_excitation(jb_batch *hb, size_t i, double *dst, size_t cap)
{
    Batch *b = (Batch *)hb;
    if (!b || i >= (size_t)b->B)
        return JB_ERR_INVALID;
    if (!b->vd.exc) {
        jb::set_error("excitation tap needs JB_BATCH_KEEP_TRACKS");
        return JB_ERR_INVALID;
    }
    size_t ns = (size_t)b->T[i] * b->voice.fperiod;
    if (cap < ns)
        return JB_ERR_BUFFER;
    if (ns == 0)
        return JB_OK;
    return b->read(b->vd.exc + (size_t)b->frame_off[i] * b->voice.fperiod, dst, ns * sizeof(double));
}

int jb_batch_info(const jb_batch *hb, uint32_t *chunk_frames, uint32_t *warmup_frames,
                  uint32_t *n_items, uint32_t *n_redo)
{
    const Batch *b = (const Batch *)hb;
    if (!b)
        return JB_ERR_INVALID;
    if (chunk_frames)
        *chunk_frames = b->chunk_frames;
    if (warmup_frames)
        *warmup_frames = b->warmup_frames;
    if (n_items)
        *n_items = b->n_items;
    if (n_redo)
        *n_redo = b->n_redo;
    return JB_OK;
}

int jb_batch_kernel_info(const jb_batch *hb, uint32_t *lane_triple, uint32_t *waves_per_simd)
{
    const Batch *b = (const Batch *)hb;
    if (!b)
        return JB_ERR_INVALID;
    if (lane_triple)
        *lane_triple = b->lp_mode ? 1u : 0u;
    if (waves_per_simd)
        *waves_per_simd = b->lp_mode ? (uint32_t)b->lt_waves_per_simd : 0u;
    return JB_OK;
}

int jb_batch_redo_stats(const jb_batch *hb, uint32_t *n_partial, uint32_t *n_full)
{
    const Batch *b = (const Batch *)hb;
    if (!b)
        return JB_ERR_INVALID;
    if (n_partial)
        *n_partial = b->n_redo_partial;
    if (n_full)
        *n_full = b->n_redo_full;
    return JB_OK;
}

uint32_t jb_batch_gang_fallbacks(const jb_batch *b) { return b ? ((const Batch *)b)->gang_fallbacks : 0; }

void jb_batch_free(jb_batch *b) { delete (Batch *)b; }

int jb_paramgen_vocode_batch(const jb_voice_desc *voice, const jb_state_utt *utts, size_t n,
                             const jb_batch_opts *opts, double *const *pcm, size_t *n_samples)
{
    jb_batch *hb = nullptr;
    int rc = jb_batch_create(voice, utts, n, opts, &hb);
    if (rc)
        return rc;
    std::unique_ptr<Batch> guard((Batch *)hb);
    if (n_samples)
        for (size_t i = 0; i < n; i++)
            n_samples[i] = jb_batch_num_samples(hb, i);
    if (!pcm)
        return JB_OK;
    if ((rc = guard->run(false)) || (rc = guard->sync()))
        return rc;
    for (size_t i = 0; i < n; i++) {
        size_t ns = jb_batch_num_samples(hb, i);
        if (ns && (rc = jb_batch_read_pcm(hb, i, pcm[i], ns)))
            return rc;
    }
    return JB_OK;
}

int jb_vocode_tracks_batch(const jb_voice_desc *voice, const jb_track_utt *utts, size_t n, const jb_batch_opts *opts,
                           double *const *pcm, size_t *n_samples)
{
    jb_batch *hb = nullptr;
    int rc = jb_batch_create_from_tracks(voice, utts, n, opts, &hb);
    if (rc)
        return rc;
    std::unique_ptr<Batch> guard((Batch *)hb);
    if (n_samples)
        for (size_t i = 0; i < n; i++)
            n_samples[i] = jb_batch_num_samples(hb, i);
    if (!pcm)
        return JB_OK;
    if ((rc = guard->run(false)) || (rc = guard->sync()))
        return rc;
    for (size_t i = 0; i < n; i++) {
        size_t ns = jb_batch_num_samples(hb, i);
        if (ns && (rc = jb_batch_read_pcm(hb, i, pcm[i], ns)))
            return rc;
    }
    return JB_OK;
}

int jb_vocoder_synthesize_batch(const jb_voice_desc *voice, const jb_track_utt *utts, size_t n,
                                const jb_batch_opts *opts, double *const *pcm, size_t *n_samples)
{
    Batch *b = nullptr;
    jb::TrackSrc src{utts};
    src.vocoder_level = true;
    int rc = Batch::create(voice, nullptr, n, opts, &b, nullptr, &src);
    if (rc)
        return rc;
    std::unique_ptr<Batch> guard(b);
    jb_batch *hb = (jb_batch *)b;
    if (n_samples)
        for (size_t i = 0; i < n; i++)
            n_samples[i] = jb_batch_num_samples(hb, i);
    if (!pcm)
        return JB_OK;
    if ((rc = guard->run(false)) || (rc = guard->sync()))
        return rc;
    for (size_t i = 0; i < n; i++) {
        size_t ns = jb_batch_num_samples(hb, i);
        if (ns && (rc = jb_batch_read_pcm(hb, i, pcm[i], ns)))
            return rc;
    }
    return JB_OK;
}

int jb_mlpg_batch(const jb_voice_desc *voice, const jb_state_utt *utts, size_t n, const jb_batch_opts *opts,
                  double *const *tracks, size_t *n_frames)
{
    jb_batch_opts o{};
    if (opts)
        o = *opts;
    else
        o.device = -1;
    o.flags |= JB_BATCH_MLPG_ONLY;
    jb_batch *hb = nullptr;
    int rc = jb_batch_create(voice, utts, n, &o, &hb);
    if (rc)
        return rc;
    std::unique_ptr<Batch> guard((Batch *)hb);
    if (n_frames)
        for (size_t i = 0; i < n; i++)
            n_frames[i] = jb_batch_num_frames(hb, i);
    if (!tracks)
        return JB_OK;
    if ((rc = guard->run(false)) || (rc = guard->sync()))
        return rc;
    const uint32_t ns = voice->nstream;
    for (size_t i = 0; i < n; i++)
        for (uint32_t si = 0; si < ns; si++) {
            double *dst = tracks[i * ns + si];
            const size_t ne = jb_batch_num_frames(hb, i) * (size_t)voice->stream[si].vector_length;
            if (dst && ne && (rc = jb_batch_read_track(hb, i, si, dst, ne)))
                return rc;
        }
    return JB_OK;
}

} // extern "C"
