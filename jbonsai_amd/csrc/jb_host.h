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
// Device copy of the shared Gaussian noise stream; a Batch keeps the table it was created with alive
struct NoiseDev {
    int device = -1;
    double *ptr = nullptr;
    size_t len = 0;
    ~NoiseDev();
};
int noise_table(int device, size_t need, std::shared_ptr<NoiseDev> *out);
void release_cached_memory(); // empties the per-device pools of finished batches' memory
void set_cached_memory_limit(size_t bytes); // cap of a device's pool (JB_DEVICE_POOL_MB at start)
// jb_synthesize_batch[_i16] on one device (jb_engine.cpp); host_threads = 0: default front-half thread count
int synthesize_batch_impl(const jb_engine *e, const char *const *lines, const size_t *line_off, size_t n_utts,
                          int32_t device, size_t elem, void **pcm, size_t *n_samples, unsigned host_threads = 0);
// static LPT partition (jb_multi.cpp): part_of[i] = bin of item i
void lpt_partition(const uint64_t *weights, size_t n, size_t n_parts, uint32_t *part_of);

// Device-resident pdf tables of a voice set (jb_pdf_set) and an indexed batch source
struct PdfSet {
    int device = -1;
    uint32_t nv = 0, ns = 0;
    std::vector<const float *> tab;        // [nv * ns] device
    std::vector<uint32_t> n_rows, row_len; // [nv * ns]
    ~PdfSet();
};
struct IndexSrc {
    const PdfSet *set;
    const jb_index_utt *utts;
};
// Parameter tracks as the batch's source (SpeechGenerator::new, src/speech.rs:25-50): no MLPG
struct TrackSrc {
    const jb_track_utt *utts;
    // Vocoder::new + Vocoder::synthesize per frame (vocoder/mod.rs:45-72) instead of SpeechGenerator::new:
    // no check of the LPF length, so nlpf == 0 (the ring-buffer-less branch of Excitation::get,
    // excitation.rs:87-100) is reachable, as it is in the reference through this seam alone
    bool vocoder_level = false;
};

struct Batch {
    int device = -1;
    uint32_t flags = 0;
    int B = 0;
    jb_voice_desc voice{};
    hipStream_t stream = nullptr;            // main: MCP chain (and the vocoder unless CUs are partitioned)
    hipStream_t stream_voc = nullptr;        // vocoder + hand-off check (== stream)
    hipEvent_t ev_mlpg_done = nullptr, ev_voc_done = nullptr;
    hipStream_t stream_lf0 = nullptr, stream_lpf = nullptr; // concurrent parameter-generation chains
    hipEvent_t ev_fork = nullptr, ev_lf0 = nullptr, ev_lpf = nullptr, ev_prep = nullptr, ev_build = nullptr, ev_mcpbuild = nullptr, ev_ivar = nullptr, ev_fb = nullptr;
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
    std::shared_ptr<NoiseDev> noise;         // vd.noise points into it
    // vocoder work items
    std::vector<VocWork> work;       // host copy
    VocWork *work_dev = nullptr;
    uint32_t n_items = 0, chunk_frames = 0, warmup_frames = 0, n_redo = 0;
    double verify_tol = 1e-9;
    double *end_state = nullptr, *warm_state = nullptr;
    double *ckpt_state = nullptr, *tmp_state = nullptr; // partial redo: checkpoints / recomputed states
    double *ckpt2_state = nullptr, *tmp2_state = nullptr; // the same for the second checkpoint of long chunks
    const double **pairs_dev = nullptr;
    uint32_t n_redo_partial = 0, n_redo_full = 0;   // of the last run: settled at the checkpoint / redone to the end
    uint32_t n_recert_failed = 0;                   // successors of fully redone chunks that failed re-certification
    size_t state_stride = 0;
    uint8_t *bad_dev = nullptr;
    uint32_t *nbad_dev = nullptr;
    bool verify_pending = false;
    std::vector<uint8_t> first_of_kind; // [B] 1: no earlier utterance of the batch was made from the same arrays
    bool lp_mode = false;            // lane-triple throughput kernel
    int lt_waves_per_simd = 2;       // its waves per SIMD: 2 (eight-wave workgroups) or 1 (build_work)
    uint32_t *order_dev = nullptr;   // its launch permutation
    VocWork *redo_dev = nullptr;
    VocWork *gen_work_dev = nullptr; // one item per frame of utterance 0 (streaming generator)
    std::vector<std::pair<void *, size_t>> allocs; // device blocks (pointer, pooled size)
    std::map<std::pair<const void *, size_t>, const void *> uploaded;
    // PINNED staging chunks of the upload arena, kept in a process-wide list between batches (a chunk allocated
    // with new[] for every batch made the arena's H2D copy a pageable one of fresh pages: 14 ms of a 28 ms
    // creation when sub-batches of a job are created one after the other)
    struct PinnedChunk { // staging buffer of an upload arena: pinned if the host grants it, else pageable
        uint8_t *p = nullptr;
        bool pageable = false;
        PinnedChunk() = default;
        PinnedChunk(const PinnedChunk &) = delete;
        PinnedChunk &operator=(const PinnedChunk &) = delete;
        PinnedChunk(PinnedChunk &&o) noexcept : p(o.p), pageable(o.pageable) { o.p = nullptr; }
        PinnedChunk &operator=(PinnedChunk &&o) noexcept
        {
            reset();
            p = o.p;
            pageable = o.pageable;
            o.p = nullptr;
            return *this;
        }
        ~PinnedChunk() { reset(); }
        bool acquire(size_t bytes);
        void reset();
        uint8_t *get() const { return p; }
    };
    struct UploadChunk { // arena for small input arrays: one H2D copy per chunk
        uint8_t *dev = nullptr;
        PinnedChunk host;
        size_t used = 0, sent = 0;
    };
    std::vector<UploadChunk> up_chunks;
    int flush_uploads();

    ~Batch();
    template <class T> int dalloc(T **p, size_t n, bool zero);
    // blocks that must start out zeroed: cleared by ONE launch when the batch has been put together (flush_zero),
    // not by a fill of its own each -- two dozen 5 us launches were 0.13 ms of the 2.4 ms of a one-sentence request
    std::vector<std::pair<void *, size_t>> zero_list;
    int flush_zero();
    int upload(const void *host, size_t bytes, const void **dev);
    // the same arena without the de-duplication: for descriptor arrays put together in temporaries of create() (a
    // synchronous hipMemcpy of its own each -- nine of them -- was 0.2 ms of a one-sentence request); the bytes reach
    // the device with the next flush_uploads()
    template <class T> int stage(const T *host, size_t n, T **dev);
    bool from_tracks = false;        // created from parameter tracks: run() starts at the frame prologue
    bool gang_check_pending = false; // a resident GV kernel has been enqueued since its error flag was last read
    // the resident GV kernel of the pending run gave up in formation (flag read, not cleared: sync() acts on it);
    // the caller has waited for ev_mlpg_done
    int gang_timeout_seen(bool *seen);
    double *gen_pcm = nullptr;       // PCM of the streaming generator's serially served frames (its own buffer)
    bool last_run_timed = false;
    uint32_t gang_fallbacks = 0;     // times the resident GV kernel timed out in formation and the sweeps took over
    static int create(const jb_voice_desc *voice, const jb_state_utt *utts, size_t n,
                      const jb_batch_opts *opts, Batch **out, const IndexSrc *idx = nullptr,
                      const TrackSrc *trk = nullptr);
    int enqueue_mlpg_only();
    int enqueue_from_tracks();
    int gather_states(const jb_voice_desc *voice, const IndexSrc &idx, size_t n,
                      std::vector<StreamStatesDev> &out); // [n * nstream]: mean/var/msd filled
    int build_work(const jb_batch_opts *opts);
    int build_generator_work();
    int enqueue_paramgen();
    int enqueue_vocoder();
    int finish_verify();
    int run(bool timed);
    int sync();
    int read(const void *dev, void *dst, size_t bytes, bool do_sync = true); // do_sync: wait for the batch's streams + certification first
    // whole PCM slab -> one host buffer per utterance (dst[u] may be null for empty utterances);
    // elem = 8 (f64) or 2 (JB_BATCH_PCM_I16)
    int read_pcm_split(void *const *dst, size_t elem);
};

} // namespace jb
