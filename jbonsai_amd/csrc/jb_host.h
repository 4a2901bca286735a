// jb_host.h -- host-side internals shared by the C-ABI translation units.
#pragma once
#include "../../include/jbonsai_amd.h"
#include "jb_device.h"

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace jb {

extern thread_local std::string g_err;
void set_error(const std::string &s);
int hip_fail(hipError_t e, const char *what);
int noise_table(int device, size_t need, const double **ptr, size_t *len);

struct Batch {
    int device = -1;
    uint32_t flags = 0;
    int B = 0;
    jb_voice_desc voice{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    std::vector<uint32_t> T;
    std::vector<uint64_t> frame_off;
    uint64_t sumT = 0;
    uint32_t maxT = 0;
    size_t total_samples = 0;
    size_t bytes_alloc = 0, bytes_input = 0;
    BatchDev bd{};
    StreamDev sd[kMaxStream]{};
    VocDev vd{};
    std::vector<void *> allocs;
    std::map<std::pair<const void *, size_t>, const void *> uploaded;

    ~Batch();
    template <class T> int dalloc(T **p, size_t n, bool zero);
    int upload(const void *host, size_t bytes, const void **dev);
    static int create(const jb_voice_desc *voice, const jb_state_utt *utts, size_t n,
                      const jb_batch_opts *opts, Batch **out);
    int enqueue_paramgen();
    int run(bool timed);
    int sync();
    int read(const void *dev, void *dst, size_t bytes);
};

} // namespace jb
