// jb_multi.cpp -- multi-GPU entries of the C ABI (SURVEY 8b "device_ids[] / n_devices", 8e).
//
// Utterances are independent (they share only the read-only voice tables and the noise table,
// which every device keeps its own copy of), so a batch shards over the GPUs of a node with no
// data-path collective: a static LPT (longest-processing-time-first) partition by length, one host
// thread per device, each running the single-device path on its share.  The results land in the
// caller's per-utterance buffers in the original order.  The reference has no counterpart (it is
// single-utterance, single-thread); a Rust `Engine::synthesize_batch` over a device list would sit
// on these entries.
#include "jb_host.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <dlfcn.h>
#include <numeric>
#include <queue>
#include <rccl/rccl.h> // types only: the library is bound at run time (below)
#include <thread>

namespace jb {

// Items heaviest first (ties: lower index first), each onto the currently lightest bin (ties:
// lower bin).  Deterministic; jbonsai_amd/shard.py states the same rule for the one-process-per-GPU
// driver (bench.py) and tests/test_shard_dist.py holds the two to the same answer.
void lpt_partition(const uint64_t *weights, size_t n, size_t n_parts, uint32_t *part_of)
{
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weights[a] > weights[b]; });
    typedef std::pair<uint64_t, uint32_t> Bin; // (load, bin): smallest load first, then smallest bin
    std::priority_queue<Bin, std::vector<Bin>, std::greater<Bin>> heap;
    for (uint32_t p = 0; p < (uint32_t)n_parts; p++)
        heap.push(Bin(0, p));
    for (size_t i : order) {
        Bin b = heap.top();
        heap.pop();
        part_of[i] = b.second;
        heap.push(Bin(b.first + weights[i], b.second));
    }
}

namespace {
// Runs fn(part) on one host thread per part (in the calling thread when there is one part) and
// returns the first failure with its message.
template <class F> int for_each_part(size_t n_parts, F fn)
{
    std::vector<int> rcs(n_parts, JB_OK);
    std::vector<std::string> errs(n_parts);
    auto body = [&](size_t p) {
        rcs[p] = fn(p);
        if (rcs[p])
            errs[p] = g_err; // the worker's thread-local message
    };
    if (n_parts == 1) {
        body(0);
    } else {
        std::vector<std::thread> pool;
        for (size_t p = 0; p < n_parts; p++)
            pool.emplace_back(body, p);
        for (auto &t : pool)
            t.join();
    }
    for (size_t p = 0; p < n_parts; p++)
        if (rcs[p]) {
            set_error(errs[p]);
            return rcs[p];
        }
    return JB_OK;
}

int check_devices(const int32_t *devices, size_t n_devices)
{
    if (!devices || n_devices == 0) {
        set_error("device list is empty");
        return JB_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device available (this library has no CPU path)");
        return JB_ERR_DEVICE;
    }
    for (size_t i = 0; i < n_devices; i++)
        if (devices[i] < 0 || devices[i] >= ndev) {
            set_error("device ordinal out of range");
            return JB_ERR_INVALID;
        }
    return JB_OK;
}
} // namespace

static int synthesize_multi(const jb_engine *e, const char *const *lines, const size_t *line_off, size_t n_utts,
                            const int32_t *devices, size_t n_devices, size_t elem, void **pcm, size_t *n_samples)
{
    if (!e || !pcm || !n_samples || (n_utts && (!line_off || !lines)))
        return JB_ERR_INVALID;
    int rc = check_devices(devices, n_devices);
    if (rc)
        return rc;
    for (size_t u = 0; u < n_utts; u++) {
        pcm[u] = nullptr;
        n_samples[u] = 0;
    }
    // the frame count of an utterance is known only after its front half has run; the number of
    // labels is the proxy the partition balances (frames per label vary far less than labels per
    // utterance)
    std::vector<uint64_t> w(n_utts);
    for (size_t u = 0; u < n_utts; u++)
        w[u] = line_off[u + 1] - line_off[u];
    std::vector<uint32_t> part(n_utts);
    lpt_partition(w.data(), n_utts, n_devices, part.data());
    struct Share {
        std::vector<size_t> utts;
        std::vector<const char *> lines;
        std::vector<size_t> off;
        std::vector<void *> pcm;
        std::vector<size_t> ns;
    };
    std::vector<Share> sh(n_devices);
    for (size_t u = 0; u < n_utts; u++)
        sh[part[u]].utts.push_back(u);
    for (Share &s : sh) {
        s.off.push_back(0);
        for (size_t u : s.utts) {
            for (size_t l = line_off[u]; l < line_off[u + 1]; l++)
                s.lines.push_back(lines[l]);
            s.off.push_back(s.lines.size());
        }
        s.pcm.assign(s.utts.size(), nullptr);
        s.ns.assign(s.utts.size(), 0);
    }
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned per_dev = std::max(1u, std::min(16u, (hw ? hw : 1u) / (unsigned)n_devices));
    rc = for_each_part(n_devices, [&](size_t p) -> int {
        Share &s = sh[p];
        if (s.utts.empty())
            return JB_OK;
        return synthesize_batch_impl(e, s.lines.data(), s.off.data(), s.utts.size(), devices[p], elem, s.pcm.data(),
                                     s.ns.data(), n_devices > 1 ? per_dev : 0);
    });
    if (rc) {
        for (Share &s : sh)
            for (void *p : s.pcm)
                free(p);
        return rc;
    }
    for (Share &s : sh)
        for (size_t i = 0; i < s.utts.size(); i++) {
            pcm[s.utts[i]] = s.pcm[i];
            n_samples[s.utts[i]] = s.ns[i];
        }
    return JB_OK;
}

} // namespace jb

extern "C" {

int jb_lpt_partition(const uint64_t *weights, size_t n, size_t n_parts, uint32_t *part_of)
{
    if (n_parts == 0 || n_parts > 0xffffffffu || (n && (!weights || !part_of))) {
        jb::set_error("jb_lpt_partition: n_parts must be positive");
        return JB_ERR_INVALID;
    }
    jb::lpt_partition(weights, n, n_parts, part_of);
    return JB_OK;
}

int jb_paramgen_vocode_batch_multi(const jb_voice_desc *voice, const jb_state_utt *utts, size_t n_utts,
                                   const jb_batch_opts *opts, const int32_t *devices, size_t n_devices,
                                   double *const *pcm, size_t *n_samples)
{
    if (!voice || (n_utts && !utts))
        return JB_ERR_INVALID;
    int rc = jb::check_devices(devices, n_devices);
    if (rc)
        return rc;
    // weights: frames per utterance (the cost of every kernel of the path is linear in them)
    std::vector<uint64_t> w(n_utts, 0);
    for (size_t u = 0; u < n_utts; u++) {
        if (utts[u].num_states && !utts[u].durations)
            return JB_ERR_INVALID;
        for (uint32_t s = 0; s < utts[u].num_states; s++)
            w[u] += utts[u].durations[s];
    }
    std::vector<uint32_t> part(n_utts);
    jb::lpt_partition(w.data(), n_utts, n_devices, part.data());
    struct Share {
        std::vector<size_t> idx;
        std::vector<jb_state_utt> utts;
        std::vector<double *> pcm;
        std::vector<size_t> ns;
    };
    std::vector<Share> sh(n_devices);
    for (size_t u = 0; u < n_utts; u++) {
        Share &s = sh[part[u]];
        s.idx.push_back(u);
        s.utts.push_back(utts[u]);
        s.pcm.push_back(pcm ? pcm[u] : nullptr);
    }
    for (Share &s : sh)
        s.ns.assign(s.idx.size(), 0);
    rc = jb::for_each_part(n_devices, [&](size_t p) -> int {
        Share &s = sh[p];
        if (s.idx.empty())
            return JB_OK;
        jb_batch_opts o{};
        if (opts)
            o = *opts;
        o.device = devices[p];
        return jb_paramgen_vocode_batch(voice, s.utts.data(), s.idx.size(), &o, pcm ? s.pcm.data() : nullptr,
                                        s.ns.data());
    });
    if (rc)
        return rc;
    if (n_samples)
        for (Share &s : sh)
            for (size_t i = 0; i < s.idx.size(); i++)
                n_samples[s.idx[i]] = s.ns[i];
    return JB_OK;
}

int jb_synthesize_batch_multi(const jb_engine *e, const char *const *label_lines, const size_t *line_off,
                              size_t n_utts, const int32_t *devices, size_t n_devices, double **pcm,
                              size_t *n_samples)
{
    return jb::synthesize_multi(e, label_lines, line_off, n_utts, devices, n_devices, sizeof(double), (void **)pcm,
                                n_samples);
}

int jb_synthesize_batch_i16_multi(const jb_engine *e, const char *const *label_lines, const size_t *line_off,
                                  size_t n_utts, const int32_t *devices, size_t n_devices, int16_t **pcm,
                                  size_t *n_samples)
{
    return jb::synthesize_multi(e, label_lines, line_off, n_utts, devices, n_devices, sizeof(int16_t), (void **)pcm,
                                n_samples);
}


// ---- PCM gather over RCCL (north_star: "RCCL over xGMI only to gather output PCM") --------
// Utterance-sharded synthesis needs no data-path collective; the one optional exchange is the sink that
// wants every rank's PCM on one GPU.  The slabs differ in length, so the gather is point to point: grouped
// ncclSend / ncclRecv, one message per peer -- over xGMI every peer has its own link into the root.
// RCCL is bound with dlopen at first use ("librccl.so.1": the copy the process already holds, e.g. the one a
// PyTorch host loaded, else ROCm's), so the library carries no link-time dependency on it and a
// single-rank communicator never loads it.
} // extern "C"

namespace jb {
namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr; // optional: jb_comm_size asks the communicator itself
};

const Rccl *rccl()
{
    static const Rccl r = [] {
        Rccl x;
        // JB_RCCL_LIBRARY = full path of the library to bind instead (a host whose RCCL lives elsewhere; the
        // test double of tests/fake_rccl, which lets N ranks share one device).  Bound RTLD_LOCAL: its symbols
        // are only ever reached through this table.
        if (const char *ov = getenv("JB_RCCL_LIBRARY")) {
            if (*ov)
                x.h = dlopen(ov, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
                x.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (x.h)
                    break;
            }
        }
        if (!x.h)
            return x;
        auto sym = [&](const char *n) { return dlsym(x.h, n); };
        x.GetUniqueId = (decltype(x.GetUniqueId))sym("ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))sym("ncclCommInitRank");
        x.CommDestroy = (decltype(x.CommDestroy))sym("ncclCommDestroy");
        x.GroupStart = (decltype(x.GroupStart))sym("ncclGroupStart");
        x.GroupEnd = (decltype(x.GroupEnd))sym("ncclGroupEnd");
        x.Send = (decltype(x.Send))sym("ncclSend");
        x.Recv = (decltype(x.Recv))sym("ncclRecv");
        x.AllGather = (decltype(x.AllGather))sym("ncclAllGather");
        x.GetErrorString = (decltype(x.GetErrorString))sym("ncclGetErrorString");
        x.CommCount = (decltype(x.CommCount))sym("ncclCommCount");
        if (!x.GetUniqueId || !x.CommInitRank || !x.CommDestroy || !x.GroupStart || !x.GroupEnd || !x.Send || !x.Recv ||
            !x.AllGather || !x.GetErrorString)
            x.h = nullptr;
        return x;
    }();
    return r.h ? &r : nullptr;
}

int rccl_fail(const Rccl *r, ncclResult_t e, const char *what)
{
    set_error(std::string(what) + ": " + (r && r->GetErrorString ? r->GetErrorString(e) : "RCCL error"));
    return JB_ERR_DEVICE;
}
} // namespace

struct Comm {
    int n_ranks = 1, rank = 0, device = -1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    uint64_t *counts_dev = nullptr; // [n_ranks] sample counts (ncclAllGather)
    ~Comm()
    {
        if (device >= 0)
            hipSetDevice(device);
        if (comm && rccl())
            rccl()->CommDestroy(comm);
        if (counts_dev)
            hipFree(counts_dev);
        if (stream)
            hipStreamDestroy(stream);
    }
};

struct Gathered {
    int device = -1, n_ranks = 0;
    size_t elem = 8;
    std::vector<void *> slab;     // [n_ranks] device pointers; the root's own entry aliases its batch's slab
    std::vector<size_t> samples;  // [n_ranks]
    std::vector<uint8_t> owned;   // [n_ranks] hipMalloc'ed here
    ~Gathered()
    {
        if (device >= 0)
            hipSetDevice(device);
        for (size_t i = 0; i < slab.size(); i++)
            if (owned[i] && slab[i])
                hipFree(slab[i]);
    }
};
} // namespace jb

extern "C" {

int jb_comm_unique_id(uint8_t *id, size_t cap)
{
    if (!id || cap < JB_COMM_ID_BYTES)
        return JB_ERR_BUFFER;
    static_assert(JB_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "jb_comm id = ncclUniqueId");
    const jb::Rccl *r = jb::rccl();
    if (!r) {
        jb::set_error("RCCL (librccl.so.1) cannot be loaded");
        return JB_ERR_DEVICE;
    }
    ncclUniqueId u;
    ncclResult_t e = r->GetUniqueId(&u);
    if (e != ncclSuccess)
        return jb::rccl_fail(r, e, "ncclGetUniqueId");
    memcpy(id, u.internal, JB_COMM_ID_BYTES);
    return JB_OK;
}

int jb_comm_init(const uint8_t *id, int n_ranks, int rank, int32_t device, jb_comm **out)
{
    if (!out || n_ranks < 1 || rank < 0 || rank >= n_ranks || (n_ranks > 1 && !id))
        return JB_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        jb::set_error("no HIP device available (this library has no CPU path)");
        return JB_ERR_DEVICE;
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess)
        return JB_ERR_DEVICE;
    if (device >= ndev) {
        jb::set_error("device ordinal out of range");
        return JB_ERR_INVALID;
    }
    std::unique_ptr<jb::Comm> c(new jb::Comm());
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->device = device;
    hipError_t he = hipSetDevice(device);
    if (he == hipSuccess)
        he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (he == hipSuccess)
        he = hipMalloc((void **)&c->counts_dev, sizeof(uint64_t) * (size_t)n_ranks);
    if (he != hipSuccess)
        return jb::hip_fail(he, "jb_comm_init");
    if (n_ranks > 1) { // a communicator of one never touches RCCL
        const jb::Rccl *r = jb::rccl();
        if (!r) {
            jb::set_error("RCCL (librccl.so.1) cannot be loaded");
            return JB_ERR_DEVICE;
        }
        ncclUniqueId u;
        memcpy(u.internal, id, JB_COMM_ID_BYTES);
        ncclResult_t e = r->CommInitRank(&c->comm, n_ranks, u, rank);
        if (e != ncclSuccess)
            return jb::rccl_fail(r, e, "ncclCommInitRank");
    }
    *out = (jb_comm *)c.release();
    return JB_OK;
}

void jb_comm_free(jb_comm *c) { delete (jb::Comm *)c; }
int jb_comm_rank(const jb_comm *c) { return c ? ((const jb::Comm *)c)->rank : -1; }
// the size RCCL itself reports for the communicator (ncclCommCount), not the number the caller passed in: what a
// multi-GPU bench line quotes as proof that its ranks really formed ONE communicator of N (a one-rank communicator
// never loads RCCL: its size is 1 by construction)
int jb_comm_size(const jb_comm *c)
{
    if (!c)
        return 0;
    const jb::Comm *cm = (const jb::Comm *)c;
    if (cm->comm) {
        const jb::Rccl *r = jb::rccl();
        int n = 0;
        if (r && r->CommCount && r->CommCount(cm->comm, &n) == ncclSuccess)
            return n;
    }
    return cm->n_ranks;
}

// Failure is COLLECTIVE: a rank that cannot contribute (no batch, a batch on another device, a run that failed)
// still joins the exchange of the counts with a sentinel, and a root that cannot allocate its receive slabs
// says so in a second one-word exchange -- every rank then leaves with an error instead of one rank returning
// early and its peers blocking in RCCL for good.  (What cannot be made collective after the fact: an RCCL call
// that itself fails on one rank.)
int jb_gather_pcm(jb_comm *hc, jb_batch *hb, int root, jb_gathered **out, float *ms)
{
    jb::Comm *c = (jb::Comm *)hc;
    jb::Batch *b = (jb::Batch *)hb;
    if (out)
        *out = nullptr;
    if (!c || root < 0 || root >= c->n_ranks) // the same answer on every rank of a correct program: before any exchange
        return JB_ERR_INVALID;
    constexpr uint64_t kFail = ~0ull, kI16 = 1ull << 62;
    int local_rc = JB_OK;
    std::string local_err;
    size_t ns = 0;
    void *slab = nullptr;
    if (!b || !out) {
        local_rc = JB_ERR_INVALID;
        local_err = "jb_gather_pcm: no batch / no output handle on this rank";
    } else if (b->device != c->device) {
        local_rc = JB_ERR_INVALID;
        local_err = "the batch lives on another device than the communicator";
    } else {
        slab = jb_batch_device_pcm(hb, &ns); // waits for the run and its certification
        if (!slab && ns) {
            local_rc = JB_ERR_DEVICE;
            local_err = std::string("this rank's batch has no PCM to contribute: ") + jb_last_error();
        }
    }
    const size_t elem = (b && (b->flags & JB_BATCH_PCM_I16)) ? 2 : 8;
    hipError_t he = hipSetDevice(c->device);
    if (he != hipSuccess && local_rc == JB_OK) {
        local_rc = JB_ERR_DEVICE;
        local_err = std::string("hipSetDevice: ") + hipGetErrorString(he);
    }
    if (c->n_ranks == 1) {
        if (local_rc) {
            jb::set_error(local_err);
            return local_rc;
        }
        std::unique_ptr<jb::Gathered> g(new jb::Gathered());
        g->device = c->device;
        g->n_ranks = 1;
        g->elem = elem;
        g->slab.assign(1, slab);
        g->owned.assign(1, 0);
        g->samples.assign(1, ns);
        if (ms)
            *ms = 0.0f;
        *out = (jb_gathered *)g.release();
        return JB_OK;
    }
    const jb::Rccl *r = jb::rccl();
    if (!r) { // cannot happen behind a successful jb_comm_init of more than one rank
        jb::set_error("RCCL (librccl.so.1) cannot be loaded");
        return JB_ERR_DEVICE;
    }
    const auto t0 = std::chrono::steady_clock::now();
    // one word per rank: its sample count (bit 62: 16-bit samples), or the sentinel
    auto exchange = [&](uint64_t mine, std::vector<uint64_t> &all, const char *what) -> int {
        all.assign((size_t)c->n_ranks, 0);
        hipError_t e1 = hipMemcpyAsync(c->counts_dev + c->rank, &mine, sizeof mine, hipMemcpyHostToDevice, c->stream);
        if (e1 != hipSuccess)
            return jb::hip_fail(e1, what);
        ncclResult_t e = r->AllGather(c->counts_dev + c->rank, c->counts_dev, 1, ncclUint64, c->comm, c->stream);
        if (e != ncclSuccess)
            return jb::rccl_fail(r, e, what);
        if ((e1 = hipMemcpyAsync(all.data(), c->counts_dev, sizeof(uint64_t) * all.size(), hipMemcpyDeviceToHost,
                                 c->stream)) != hipSuccess ||
            (e1 = hipStreamSynchronize(c->stream)) != hipSuccess)
            return jb::hip_fail(e1, what);
        return JB_OK;
    };
    std::vector<uint64_t> words, status;
    int rc = exchange(local_rc ? kFail : ((uint64_t)ns | (elem == 2 ? kI16 : 0)), words, "ncclAllGather(counts)");
    if (rc)
        return rc;
    for (int p = 0; p < c->n_ranks; p++)
        if (words[(size_t)p] == kFail) {
            if (local_rc) {
                jb::set_error(local_err + " (the gather was abandoned on every rank)");
                return local_rc;
            }
            jb::set_error("rank " + std::to_string(p) + " could not contribute its PCM: the gather was abandoned on every rank");
            return JB_ERR_DEVICE;
        }
    // f64 against 16-bit is decided from the exchanged words ALONE, the same way on every rank (a rank with an
    // empty batch has no say: its own sample size must not hide a mix between its peers, nor size the root's
    // receive slabs): mixed = two ranks with samples to send differ; the gather's sample size = the senders'.
    std::vector<uint64_t> counts((size_t)c->n_ranks);
    bool mixed = false, any = false, any_i16 = false;
    for (int p = 0; p < c->n_ranks; p++) {
        counts[(size_t)p] = words[(size_t)p] & ~kI16;
        if (!counts[(size_t)p])
            continue;
        const bool i16 = (words[(size_t)p] & kI16) != 0;
        if (any && i16 != any_i16)
            mixed = true; // f64 and 16-bit slabs in one gather
        any = true;
        any_i16 = any_i16 || i16;
    }
    const size_t gelem = any ? (any_i16 ? 2 : 8) : elem; // (nobody sends anything: this rank's own, for the record)
    // the root's receive slabs, then one more word from everybody: can the transfer start?
    std::unique_ptr<jb::Gathered> g;
    uint64_t my_status = mixed ? 2 : 0;
    if (c->rank == root) {
        g.reset(new jb::Gathered());
        g->device = c->device;
        g->n_ranks = c->n_ranks;
        g->elem = gelem;
        g->slab.assign((size_t)c->n_ranks, nullptr);
        g->owned.assign((size_t)c->n_ranks, 0);
        g->samples.assign(counts.begin(), counts.end());
        for (int p = 0; p < c->n_ranks && !my_status; p++) {
            if (p == root) {
                g->slab[(size_t)p] = slab; // no copy: valid while the batch lives
            } else if (counts[(size_t)p]) {
                if ((he = hipMalloc(&g->slab[(size_t)p], counts[(size_t)p] * gelem)) != hipSuccess) {
                    (void)hipGetLastError();
                    my_status = 1;
                } else {
                    g->owned[(size_t)p] = 1;
                }
            }
        }
    }
    if ((rc = exchange(my_status, status, "ncclAllGather(status)")))
        return rc;
    for (int p = 0; p < c->n_ranks; p++)
        if (status[(size_t)p]) {
            jb::set_error(status[(size_t)p] == 2
                              ? "f64 and 16-bit PCM slabs in one gather: the gather was abandoned on every rank"
                              : "rank " + std::to_string(p) +
                                    " (the root) could not allocate its receive slabs: the gather was abandoned on every rank");
            return status[(size_t)p] == 2 ? JB_ERR_INVALID : JB_ERR_DEVICE;
        }
    {
        ncclResult_t e = r->GroupStart();
        if (e != ncclSuccess)
            return jb::rccl_fail(r, e, "ncclGroupStart");
        if (c->rank == root) {
            for (int p = 0; p < c->n_ranks && e == ncclSuccess; p++)
                if (p != root && counts[(size_t)p])
                    e = r->Recv(g->slab[(size_t)p], counts[(size_t)p] * gelem, ncclUint8, p, c->comm, c->stream);
        } else if (ns) {
            e = r->Send(slab, ns * gelem, ncclUint8, root, c->comm, c->stream);
        }
        ncclResult_t e2 = r->GroupEnd(); // always: a group left open would take the next call with it
        if (e != ncclSuccess || e2 != ncclSuccess)
            return jb::rccl_fail(r, e != ncclSuccess ? e : e2, "ncclSend/ncclRecv");
        if ((he = hipStreamSynchronize(c->stream)) != hipSuccess)
            return jb::hip_fail(he, "gather sync");
    }
    if (ms)
        *ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = (jb_gathered *)g.release();
    return JB_OK;
}

size_t jb_gathered_samples(const jb_gathered *g, int rank)
{
    const jb::Gathered *x = (const jb::Gathered *)g;
    return (x && rank >= 0 && rank < x->n_ranks) ? x->samples[(size_t)rank] : 0;
}

size_t jb_gathered_sample_bytes(const jb_gathered *g) { return g ? ((const jb::Gathered *)g)->elem : 0; }

void *jb_gathered_device(const jb_gathered *g, int rank)
{
    const jb::Gathered *x = (const jb::Gathered *)g;
    return (x && rank >= 0 && rank < x->n_ranks) ? x->slab[(size_t)rank] : nullptr;
}

int jb_gathered_read(const jb_gathered *g, int rank, void *dst, size_t cap_bytes)
{
    const jb::Gathered *x = (const jb::Gathered *)g;
    if (!x || rank < 0 || rank >= x->n_ranks)
        return JB_ERR_INVALID;
    const size_t nb = x->samples[(size_t)rank] * x->elem;
    if (cap_bytes < nb)
        return JB_ERR_BUFFER;
    if (nb == 0)
        return JB_OK;
    if (!dst)
        return JB_ERR_INVALID;
    hipError_t he = hipSetDevice(x->device);
    if (he == hipSuccess)
        he = hipMemcpy(dst, x->slab[(size_t)rank], nb, hipMemcpyDeviceToHost);
    return he == hipSuccess ? JB_OK : jb::hip_fail(he, "hipMemcpy(gathered)");
}

void jb_gathered_free(jb_gathered *g) { delete (jb::Gathered *)g; }

} // extern "C"
