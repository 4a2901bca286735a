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
#include <numeric>
#include <queue>
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

} // extern "C"
