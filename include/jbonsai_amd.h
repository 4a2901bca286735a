/*
 * jbonsai_amd.h -- C ABI of libjbonsai_amd.so: an MI355X (gfx950) drop-in for the
 * parameter-generation + MLSA-vocoder hot path of the `jbonsai` crate (v0.4.2).
 *
 * The reference exposes a Rust library API, not an FFI; each entry point below
 * names the reference item it stands in for (file:line under /root/reference).
 * A Rust shim re-exporting `Engine` / `SpeechGenerator` over these symbols is in
 * INTEGRATION.md.  All functions return 0 (JB_OK) or a negative jb_status; no
 * exception or unwinding crosses this boundary; jb_last_error() gives the text
 * for the calling thread.  No torch / HIP types appear in any signature.
 *
 * Two levels:
 *   (1) state level  -- `jb_batch_*`, `jb_paramgen_vocode_batch`: the exact image
 *       of `ModelStream` + `durations` (src/model/model_stream.rs:6-15,
 *       src/engine.rs:321-357), batched over independent utterances.  This is
 *       the seam the HIP kernels sit behind: MlpgAdjust::create x3
 *       (src/mlpg_adjust/mod.rs:51-95) -> SpeechGenerator::generate_all
 *       (src/speech.rs:87-96) -> Vocoder::synthesize (src/vocoder/mod.rs:72-141).
 *   (2) engine level -- `jb_engine_*`, `jb_synthesize*`, `jb_generator_*`: mirrors
 *       Engine::{load, load_from_bytes, synthesize, generator} (src/engine.rs:257-366)
 *       and SpeechGenerator::{fperiod, synthesized_frames, generate_step}
 *       (src/speech.rs:53-82); label parsing / tree search / durations run on
 *       the host, everything from the state level down runs on the GPU.
 */
#ifndef JBONSAI_AMD_H
#define JBONSAI_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JB_MAX_STREAM 3
#define JB_MAX_WINDOW 8
#define JB_NODATA (-1e10) /* src/constants.rs:13 */

typedef enum jb_status {
    JB_OK = 0,
    JB_ERR_INVALID = -1,     /* bad argument / shape (reference: panics in src/speech.rs:32-40) */
    JB_ERR_UNSUPPORTED = -2, /* shapes no kernel is built for: nmcp > 64, window widths above 9, stages above 256, nlpf > 2047, other
                                than three streams (frame periods, stages and low-pass orders are otherwise free: rounds 1-4 refused
                                stages above 8, nlpf > 63 and frame periods without a divisor <= 64 that is >= nlpf - 1) */
    JB_ERR_DEVICE = -3,      /* HIP error or no gfx950 device: the product never falls back to CPU */
    JB_ERR_MODEL = -4,       /* ModelError (src/model/mod.rs:31-46) */
    JB_ERR_LABEL = -5,       /* LabelError (src/label.rs:8-23) */
    JB_ERR_PARSE_OPTION = -6,/* EngineError::ParseOptionError (src/engine.rs:21-23) */
    JB_ERR_WEIGHT = -7,      /* WeightError (src/model/interporation_weight.rs:7-14) */
    JB_ERR_BUFFER = -8       /* output buffer too small (reference: panic, src/speech.rs:69-71) */
} jb_status;

/* ------------------------------------------------------------------------ */
/* (1) state level                                                          */
/* ------------------------------------------------------------------------ */

/* Per-stream static description: StreamModelMetadata + Windows
 * (src/model/voice/mod.rs, src/model/voice/window.rs:4-22). */
typedef struct jb_stream_desc {
    uint32_t vector_length;              /* L */
    uint32_t num_windows;                /* W */
    uint32_t is_msd;
    uint32_t use_gv;
    uint32_t win_width[JB_MAX_WINDOW];   /* odd widths */
    const double *win_coef;              /* concatenated coefficients, sum(win_width) */
} jb_stream_desc;

/* Vocoder::new arguments (src/vocoder/mod.rs:45-55) + stream layout. */
typedef struct jb_voice_desc {
    uint32_t sampling_frequency;  /* rate */
    uint32_t fperiod;
    uint32_t nstream;             /* 3: MCP, LF0, LPF (src/engine.rs:303-313 needs stream 2) */
    uint32_t stage;               /* 0: MLSA (mel-cepstra); 1..256: Stage::NonZero, gamma = -1/stage, spectrum = [gain, LSP...],
                                     MGLSA filter (vocoder/mod.rs:90-107,142-176; parity unpinned: no reference test reaches it) */
    uint32_t use_log_gain;        /* ignored when stage==0 */
    double alpha, beta, volume;   /* beta > 0: post-filter per frame (stage 0: cepstrum.rs:23-37; stage > 0: lsp.rs:113-139) */
    jb_stream_desc stream[JB_MAX_STREAM];
} jb_voice_desc;

/* State-level parameters of one stream of one utterance: StreamParameter +
 * GvParameter (src/model/stream_parameter.rs:11, src/model/mod.rs:49). */
typedef struct jb_stream_states {
    const double *mean;        /* [S][W*L]; element L*w+m (src/mlpg_adjust/mod.rs:62) */
    const double *var;         /* [S][W*L] */
    const double *msd;         /* [S]; NULL => f64::MAX (non-MSD, src/model/mod.rs:113) */
    const double *gv_mean;     /* [L] or NULL (no GV) */
    const double *gv_var;      /* [L] */
    const uint8_t *gv_switch;  /* [S] */
    double gv_weight;          /* Condition::gv_weight[i]   (src/engine.rs:93) */
    double msd_threshold;      /* Condition::msd_threshold[i] (src/engine.rs:92) */
} jb_stream_states;

typedef struct jb_state_utt {
    uint32_t num_states;        /* S = labels * nstate */
    const uint32_t *durations;  /* [S] frames per state (src/duration.rs) */
    jb_stream_states stream[JB_MAX_STREAM];
} jb_state_utt;

typedef struct jb_batch_opts {
    int32_t device;         /* HIP device ordinal; -1 = current */
    uint32_t flags;         /* JB_BATCH_* */
    uint32_t chunk_frames;  /* vocoder time-chunk length in frames; 0 = auto */
    uint32_t warmup_frames; /* frames each chunk starts early from zero state; 0 = default (18; 14 for batches with 1000 and more distinct hand-off positions) */
    double verify_tol;      /* chunk hand-off check: max|state diff| <= tol*max|state|; 0 = default (1e-9) */
    uint32_t reserved0;     /* must be 0 (rounds 1-3: mlpg_cus_per_xcd, a CU partition that lost at every split; removed) */
    uint32_t reserved;
} jb_batch_opts;

#define JB_BATCH_KEEP_TRACKS 1u  /* keep MLPG parameter tracks readable (tests) */
#define JB_BATCH_GENERIC_MLPG 2u /* un-fused, reference-shaped MLPG kernels (A/B parity tests) */
#define JB_BATCH_SERIAL 4u       /* one wave per utterance, no time-chunking (reference-shaped recursion): an utterance's
                                    audio is then bitwise independent of the rest of the batch */
#define JB_BATCH_WAVE_KERNEL 8u  /* always the wave-per-chunk vocoder kernel (A/B tests) */
#define JB_BATCH_LANE_KERNEL 16u /* always the lane-triple throughput kernel, whatever the batch size (A/B tests) */
#define JB_BATCH_PCM_I16 64u     /* fused 16-bit sink: the vocoder writes clamped i16 PCM (value.min(32767).max(-32768)
                                    as i16, examples/is-bonsai/main.rs:44-48) instead of f64: 2 B/sample leave
                                    the GPU instead of 8.  jb_batch_read_pcm / device_pcm then fail; use the
                                    _i16 forms */
#define JB_BATCH_SERIAL_GV 32u   /* GV sums in the reference's serial order: parameter tracks bit-exact
                                    against src/mlpg_adjust/mlpg.rs:145-292, slower.  The default runs the
                                    GV sweeps time-parallel with fixed-shape tree reductions (deterministic;
                                    tracks agree to ~1e-14 relative) */

#define JB_BATCH_MLPG_ONLY 128u  /* MlpgAdjust::create only (src/mlpg_adjust/mod.rs:31-51): the run ends with the
                                    three parameter tracks (implies KEEP_TRACKS); no excitation, no PCM, and
                                    no device memory for them.  What jb_mlpg_batch sets */

#define JB_BATCH_NO_EXC_TABLE 512u /* A/B tests: the pulse-free excitation of EVERY frame is computed per utterance
                                    (what a voice whose LPF taps differ from frame to frame gets anyway) instead of
                                    read from the table all utterances share where their taps are the batch's
                                    canonical ones; same bits either way */

#define JB_BATCH_TEST_GANG_TIMEOUT 256u /* test aid: the first run behaves as if the resident GV kernel had timed
                                    out in formation (possible without a fault when several such launches share a
                                    device), which makes jb_batch_sync redo the step with the multi-launch GV
                                    sweeps; jb_batch_gang_fallbacks counts these.  (JB_GENERATOR_TEST_GANG_TIMEOUT=1
                                    in the environment does the same to the batch behind jb_generator_new.) */

/* Time-chunked vocoder (default).  The MLSA recursion is time-serial per utterance
 * (src/vocoder/mlsa.rs), but it forgets its initial state within ~16 frames (measured:
 * <=2e-11 relative after 16, rounding floor after 24; tools/warmup_study.py).  Each
 * utterance is cut into chunks that start `warmup_frames` early from zero state; a
 * device-side check then requires every chunk's warmed-up filter state to match its
 * predecessor's end state within verify_tol, and any chunk that fails is recomputed
 * serially from that end state, so the result is certified against the serial one. */

/* ---- indexed state level --------------------------------------------------
 * The per-state Gaussians of an utterance are rows of the voice's pdf tables, selected by the
 * decision trees (src/model/voice/model.rs:51-82) and, with several voices, blended with the
 * interpolation weights (VoiceSet weighted sum, src/model/voice_set.rs:80-95).  Here the tables
 * live on the device and an utterance is given by its row indices: the gather and the blend
 * (first*w0, then += w_i*param_i in voice order, as the reference) run on the GPU, and 12 bytes
 * per state and voice cross PCIe instead of 2.2 kB per state. */
#define JB_MAX_VOICES 8
typedef struct jb_pdf_table {
    const float *rows;  /* [n_rows][row_len] f32 as stored in the voice: means | variances | (msd weight) */
    uint32_t n_rows;    /* all trees of the stream concatenated */
    uint32_t row_len;   /* 2*L*W (+1 for MSD streams) */
} jb_pdf_table;
typedef struct jb_pdf_set jb_pdf_set;
/* tables[v * nstream + s] = stream s of voice v.  The rows are copied to `device` (-1 = current). */
int jb_pdf_set_create(const jb_pdf_table *tables, uint32_t n_voices, uint32_t nstream, int32_t device,
                      jb_pdf_set **out);
void jb_pdf_set_free(jb_pdf_set *set);

typedef struct jb_index_stream {
    const uint32_t *row[JB_MAX_VOICES]; /* [S] row of the state's pdf in voice v's table */
    const double *weight;               /* [n_voices] interpolation weights (Condition, engine.rs:211-243) */
    const double *gv_mean, *gv_var;     /* as jb_stream_states (already blended: one pdf per utterance) */
    const uint8_t *gv_switch;
    double gv_weight, msd_threshold;
} jb_index_stream;
typedef struct jb_index_utt {
    uint32_t num_states;
    const uint32_t *durations;
    jb_index_stream stream[JB_MAX_STREAM];
    double lf0_offset; /* additional_half_tone * ln2/12 added to the static LF0 mean and clamped to
                          [ln 20, ln 20000] (stream_parameter.rs:29-37); 0 = none */
} jb_index_utt;

typedef struct jb_batch jb_batch;

/* Upload a batch of utterances to HBM and allocate outputs/workspace.
 * Utterances may alias each other's input arrays. */
int jb_batch_create(const jb_voice_desc *voice, const jb_state_utt *utts, size_t n_utts,
                    const jb_batch_opts *opts, jb_batch **out);
/* Same from row indices into device-resident pdf tables (gather + blend on the GPU). */
int jb_batch_create_indexed(const jb_voice_desc *voice, const jb_pdf_set *set, const jb_index_utt *utts,
                            size_t n_utts, const jb_batch_opts *opts, jb_batch **out);
/* Enqueue the whole hot path (MLPG+GV x3 -> frame prologue -> pulse schedule ->
 * excitation + MLSA) on the batch's HIP stream.  Inputs are already resident. */
int jb_batch_run(jb_batch *b);
int jb_batch_sync(jb_batch *b);
/* run + sync, returning device time of the launch sequence in ms (HIP events on
 * the batch's own stream); vocoder_ms = the MLSA kernel alone. */
int jb_batch_run_timed(jb_batch *b, float *total_ms, float *vocoder_ms);
/* Device times of the last completed run (call after jb_batch_sync): whole launch
 * sequence and the vocoder kernel alone, from HIP events on the batch's stream. */
int jb_batch_last_timing(jb_batch *b, float *total_ms, float *vocoder_ms);
size_t jb_batch_size(const jb_batch *b);
size_t jb_batch_num_frames(const jb_batch *b, size_t utt);
size_t jb_batch_num_samples(const jb_batch *b, size_t utt);
size_t jb_batch_total_samples(const jb_batch *b);
/* The read entries below (and jb_batch_device_pcm) wait for the batch's pending run and for the
 * hand-off certification + redo first, as jb_batch_sync does: what they return is the finished result.
 * Copy utterance `utt`'s PCM (f64, un-clipped, as Vec<f64> of src/engine.rs:294). */
int jb_batch_read_pcm(jb_batch *b, size_t utt, double *dst, size_t cap);
/* Same for a batch created with JB_BATCH_PCM_I16 (16-bit PCM as written to WAV by the reference's
 * examples, examples/is-bonsai/main.rs:37-49). */
int jb_batch_read_pcm_i16(jb_batch *b, size_t utt, int16_t *dst, size_t cap);
/* Whole batch at once: dst[u] must hold jb_batch_num_samples(b, u) samples (may be NULL for empty
 * utterances).  The slab streams through a ring of pinned slots at link rate while worker threads
 * scatter finished slots into the caller's buffers (what jb_synthesize_batch uses): several times the
 * rate of one pageable copy per utterance. */
int jb_batch_read_pcm_all(jb_batch *b, double *const *dst);
int jb_batch_read_pcm_i16_all(jb_batch *b, int16_t *const *dst);
/* Parameter track of stream s ([T][L], NODATA in unvoiced frames); needs KEEP_TRACKS. */
int jb_batch_read_track(jb_batch *b, size_t utt, uint32_t stream, double *dst, size_t cap);
/* Device memory and HIP streams of freed batches are kept per device for the next batch
 * (hipMalloc/hipFree of a config-2 batch cost more than its GPU work, stream creation more than a
 * one-sentence synthesis); this hands them back to the driver.  The memory cap is
 * JB_DEVICE_POOL_MB (environment, default 65536; 0 disables the memory pool). */
int jb_release_cached_memory(void);
/* Changes the cap of each device's memory pool (megabytes; 0 disables the pool) and trims what is cached
 * beyond it.  A caller that keeps several large batches alive in turn (bench.py's config-3 job) raises it. */
int jb_set_cached_memory_limit(size_t megabytes);
/* Parity tap: the MLSA filter coefficients the vocoder interpolates between, [T][nmcp] =
 * mc2b(postfilter_mcp(spectrum)) per frame (src/vocoder/mod.rs:116-118). */
int jb_batch_read_coefficients(jb_batch *b, size_t utt, double *dst, size_t cap);
/* The coefficients frame 0 STARTS from, [nmcp]: equal to frame 0's row above unless a post-filter or
 * Stage::NonZero is on (then: of the un-filtered spectrum, vocoder/mod.rs:80-106). */
int jb_batch_read_first_coefficients(jb_batch *b, size_t utt, double *dst, size_t cap);
/* Debug/parity taps: excitation before gain [N]; needs KEEP_TRACKS. */
int jb_batch_read_excitation(jb_batch *b, size_t utt, double *dst, size_t cap);
/* Device pointer + sample count of the batch's contiguous PCM slab (for an RCCL gather
 * by the caller); utterance i starts at sample jb_batch_pcm_offset(b,i).  f64 samples, or
 * i16 for a JB_BATCH_PCM_I16 batch. */
void *jb_batch_device_pcm(jb_batch *b, size_t *n_samples);
size_t jb_batch_pcm_offset(const jb_batch *b, size_t utt);
/* Execution facts of the last run: chunk length / warm-up actually used, number of
 * vocoder work items, and how many chunks failed the hand-off check and were redone. */
int jb_batch_info(const jb_batch *b, uint32_t *chunk_frames, uint32_t *warmup_frames,
                  uint32_t *n_items, uint32_t *n_redo);
/* Of the chunks that failed the hand-off check in the last run: how many were settled by
 * recomputing only up to their checkpoint (48 frames) and how many had to be recomputed to the end. */
int jb_batch_redo_stats(const jb_batch *b, uint32_t *n_partial, uint32_t *n_full);
/* Which vocoder kernel the last run's work list was built for: *lane_triple = 1 the throughput kernel
 * (k_vocoder_lt: one time-chunk per lane triple), 0 the wave kernel (k_vocoder: one chunk per wave);
 * *waves_per_simd = 1 or 2 (four- / eight-wave workgroups of the throughput kernel; 0 for the wave kernel). */
int jb_batch_kernel_info(const jb_batch *b, uint32_t *lane_triple, uint32_t *waves_per_simd);
/* Times the resident GV kernel of this batch gave up in formation and the step was redone with the
 * multi-launch sweeps (0 in normal operation; see jb_gv_gang.hip "Liveness"). */
uint32_t jb_batch_gang_fallbacks(const jb_batch *b);
void jb_batch_free(jb_batch *b);

/* One-shot convenience: create + run + read + free.  pcm[i] must hold
 * n_samples[i] doubles; call with pcm==NULL to get n_samples only. */
int jb_paramgen_vocode_batch(const jb_voice_desc *voice, const jb_state_utt *utts, size_t n_utts,
                             const jb_batch_opts *opts, double *const *pcm, size_t *n_samples);

/* ---- the two inner seams of the reference -----------------------------------------
 * MlpgAdjust::new(gv_weight, msd_threshold, model_stream).create(&durations) -> Vec<Vec<f64>>
 * (src/mlpg_adjust/mod.rs:31-51; called once per stream, src/engine.rs:333-357): for every utterance the
 * tracks of all streams.  tracks[u * nstream + s] receives T_u x L_s doubles ([frame][dim], JB_NODATA in
 * unvoiced frames of an MSD stream) or is NULL (that track is not wanted); n_frames[u] = T_u.  Call with
 * tracks == NULL for the frame counts only. */
int jb_mlpg_batch(const jb_voice_desc *voice, const jb_state_utt *utts, size_t n_utts, const jb_batch_opts *opts,
                  double *const *tracks, size_t *n_frames);

/* SpeechGenerator::new(fperiod, vocoder, spectrum, lf0, lpf) (src/speech.rs:25-50) for one utterance: the
 * three parameter tracks as Vec<Vec<f64>> flattened row-major, with their outer and inner lengths so that
 * the reference's three panics can be mirrored as JB_ERR_INVALID with the same messages:
 * outer lengths differ; lf0 inner length != 1; lpf inner length even. */
typedef struct jb_track_utt {
    size_t n_spectrum, n_lf0, n_lpf;                /* outer lengths (frames) */
    uint32_t spectrum_width, lf0_width, lpf_width;  /* inner lengths: nmcp, 1, nlpf (odd) */
    uint32_t reserved;
    const double *spectrum; /* [n_spectrum][spectrum_width]: mel-cepstra (stage 0) or [gain, LSP...] */
    const double *lf0;      /* [n_lf0][1]; JB_NODATA = unvoiced frame (vocoder/mod.rs:73-77) */
    const double *lpf;      /* [n_lpf][lpf_width] */
} jb_track_utt;
/* A batch whose source is parameter tracks: jb_batch_run starts at the frame prologue
 * (Vocoder::synthesize per frame, src/vocoder/mod.rs:72-178).  Of `voice` the Vocoder::new arguments are
 * read (sampling_frequency, fperiod, stage, use_log_gain, alpha, beta, volume, and the vector lengths of
 * streams 0 and 2 = nmcp, nlpf); the window descriptions are not. */
int jb_batch_create_from_tracks(const jb_voice_desc *voice, const jb_track_utt *utts, size_t n_utts,
                                const jb_batch_opts *opts, jb_batch **out);
/* SpeechGenerator::new + generate_all (src/speech.rs:25-50,87-96) for a batch: create + run + read + free.
 * pcm[i] must hold n_samples[i] = n_lf0 * fperiod doubles; pcm == NULL: n_samples only. */
int jb_vocode_tracks_batch(const jb_voice_desc *voice, const jb_track_utt *utts, size_t n_utts,
                           const jb_batch_opts *opts, double *const *pcm, size_t *n_samples);
/* Vocoder::new(nmcp, nlpf, stage, use_log_gain, rate, alpha, beta, volume, fperiod) followed by
 * Vocoder::synthesize(lf0, spectrum, lpf, rawdata) frame after frame over the given tracks
 * (src/vocoder/mod.rs:45-72,72-178): the Vocoder seam itself, WITHOUT SpeechGenerator::new's checks of the LPF
 * length.  This is the one way to the ring-buffer-less branch of Excitation::get (nlpf == 0,
 * src/vocoder/excitation.rs:87-100: bare pulses on voiced samples, the noise stream drawn on unvoiced samples
 * only, no delay): voice->stream[2].vector_length == 0, lpf_width == 0, lpf may be NULL.  An even non-zero
 * count is JB_ERR_UNSUPPORTED (the reference's ring buffer takes it; the kernels here do not).  Outer-length
 * and lf0-width mismatches stay JB_ERR_INVALID.  Same buffers as jb_vocode_tracks_batch. */
int jb_vocoder_synthesize_batch(const jb_voice_desc *voice, const jb_track_utt *utts, size_t n_utts,
                                const jb_batch_opts *opts, double *const *pcm, size_t *n_samples);

/* ---- multi-GPU (SURVEY 8b "device_ids[] / n_devices", 8e) ----------------------------------
 * Utterances are independent, so a batch shards over the GPUs of a node with no data-path
 * collective: static LPT partition by length, one host thread per device, results in the caller's
 * order.  A device may be listed more than once (two shares run side by side on it). */
/* part_of[i] = bin of item i: items heaviest first (ties: lower index), each onto the currently
 * lightest bin (ties: lower bin).  The rule jbonsai_amd/shard.py states for one-process-per-GPU drivers. */
int jb_lpt_partition(const uint64_t *weights, size_t n, size_t n_parts, uint32_t *part_of);
/* jb_paramgen_vocode_batch over a device list (weights = frames per utterance; opts->device is ignored). */
int jb_paramgen_vocode_batch_multi(const jb_voice_desc *voice, const jb_state_utt *utts, size_t n_utts,
                                   const jb_batch_opts *opts, const int32_t *devices, size_t n_devices,
                                   double *const *pcm, size_t *n_samples);

/* ---- PCM gather over RCCL (north_star: "RCCL over xGMI only to gather output PCM") --------
 * One process (or thread) per GPU synthesises its share; nothing is exchanged on the data path.  The one
 * optional exchange is the sink that wants every rank's PCM slab on ONE GPU: variable-length slabs, so
 * point to point -- grouped ncclSend / ncclRecv, one message per peer (over xGMI every peer has its own link
 * into the root).  The reference has no counterpart (single process, single utterance).
 * RCCL is bound at run time (dlopen "librccl.so.1": the copy the process already holds, else ROCm's); a
 * communicator of one rank never loads it.  The unique id travels between the ranks by the caller's own
 * means (a file, an environment variable, the launcher's store): it is control plane. */
#define JB_COMM_ID_BYTES 128 /* ncclUniqueId */
typedef struct jb_comm jb_comm;
typedef struct jb_gathered jb_gathered;
/* rank 0: a fresh id for jb_comm_init on every rank (ncclGetUniqueId). */
int jb_comm_unique_id(uint8_t *id, size_t cap);
/* ncclCommInitRank on `device` (-1 = current); collective over the n_ranks ranks.  n_ranks == 1: id may be NULL. */
int jb_comm_init(const uint8_t *id, int n_ranks, int rank, int32_t device, jb_comm **out);
int jb_comm_rank(const jb_comm *c);
int jb_comm_size(const jb_comm *c);
void jb_comm_free(jb_comm *c);
/* Collective: every rank passes its (finished or running: the call waits) batch.  On `root`, *out holds
 * one device slab per rank (f64 samples, or i16 for JB_BATCH_PCM_I16 batches; utterance i of rank r's batch
 * at its jb_batch_pcm_offset); the root's own entry aliases its batch's slab (no copy: valid while that
 * batch lives).  On the other ranks *out is NULL.  *ms (may be NULL) = wall time of the exchange.
 * Failure is collective: a rank whose local work failed still calls this (b = NULL is allowed for it) and
 * EVERY rank then returns an error instead of blocking; so does every rank when the root cannot allocate its
 * receive slabs or when f64 and 16-bit slabs are mixed.  RCCL is bound at first use with dlopen
 * ("librccl.so.1"; JB_RCCL_LIBRARY = full path of another library to bind, e.g. the test double of
 * tests/fake_rccl that lets several ranks share one device). */
int jb_gather_pcm(jb_comm *c, jb_batch *b, int root, jb_gathered **out, float *ms);
size_t jb_gathered_samples(const jb_gathered *g, int rank);
/* Bytes per sample of the gathered slabs: 8 (f64) or 2 (the 16-bit sink) -- what the SENDING ranks' batches
 * were made with (a root whose own batch is empty takes it from them). */
size_t jb_gathered_sample_bytes(const jb_gathered *g);
void *jb_gathered_device(const jb_gathered *g, int rank);
/* Device-to-host copy of rank r's slab (cap in BYTES). */
int jb_gathered_read(const jb_gathered *g, int rank, void *dst, size_t cap_bytes);
void jb_gathered_free(jb_gathered *g);

/* ------------------------------------------------------------------------ */
/* (2) engine level                                                         */
/* ------------------------------------------------------------------------ */
typedef struct jb_engine jb_engine;
typedef struct jb_generator jb_generator;

/* Engine::load (src/engine.rs:257) / load_from_bytes (:263). */
int jb_engine_load(const char *const *paths, size_t n, jb_engine **out);
int jb_engine_load_from_bytes(const uint8_t *const *bufs, const size_t *lens, size_t n,
                              jb_engine **out);
/* Engine::new(VoiceSet, Condition) (src/engine.rs:289-291): an engine over the voices of `voices_of`
 * (shared, as the reference's Arc<Voice>) with a copy of the Condition of `condition_of`; the same engine
 * for both = Engine::clone.  JB_ERR_WEIGHT if the condition was made for another number of voices. */
int jb_engine_new(const jb_engine *voices_of, const jb_engine *condition_of, jb_engine **out);
void jb_engine_free(jb_engine *e);

/* Condition accessors (src/engine.rs:127-243); setters clamp like the reference. */
int jb_engine_set_sampling_frequency(jb_engine *e, size_t v);
size_t jb_engine_get_sampling_frequency(const jb_engine *e);
int jb_engine_set_fperiod(jb_engine *e, size_t v);
size_t jb_engine_get_fperiod(const jb_engine *e);
int jb_engine_set_volume(jb_engine *e, double db);
double jb_engine_get_volume(const jb_engine *e);
int jb_engine_set_msd_threshold(jb_engine *e, size_t stream, double v);
double jb_engine_get_msd_threshold(const jb_engine *e, size_t stream);
int jb_engine_set_gv_weight(jb_engine *e, size_t stream, double v);
double jb_engine_get_gv_weight(const jb_engine *e, size_t stream);
int jb_engine_set_phoneme_alignment_flag(jb_engine *e, int flag);
int jb_engine_get_phoneme_alignment_flag(const jb_engine *e);
/* New (no reference counterpart; the reference has no batches).  By default an utterance's audio depends,
 * to about 1e-10 relative, on what else is in its batch: the time-chunked vocoder picks its chunk length from
 * the batch's total length and hand-offs are certified to 1e-9 of the filter state, not to the last bit
 * (README.md:124 of the reference advertises bitwise-identical audio between builds).  With this flag every
 * batch of the engine runs with JB_BATCH_SERIAL | JB_BATCH_SERIAL_GV -- one wave per utterance, the
 * reference-shaped recursion from zero state, and the GV sums in the reference's serial order (parameter tracks
 * bit-exact against the CPU path) -- and the same labels give the SAME BITS alone, in any batch, through the
 * generator and on any device count; throughput drops to one SIMD per utterance. */
int jb_engine_set_batch_invariant(jb_engine *e, int flag);
int jb_engine_get_batch_invariant(const jb_engine *e);
int jb_engine_set_speed(jb_engine *e, double v);
double jb_engine_get_speed(const jb_engine *e);
int jb_engine_set_alpha(jb_engine *e, double v);
double jb_engine_get_alpha(const jb_engine *e);
int jb_engine_set_beta(jb_engine *e, double v);
double jb_engine_get_beta(const jb_engine *e);
int jb_engine_set_additional_half_tone(jb_engine *e, double v);
double jb_engine_get_additional_half_tone(const jb_engine *e);
size_t jb_engine_num_voices(const jb_engine *e);
size_t jb_engine_num_streams(const jb_engine *e);
size_t jb_engine_num_states(const jb_engine *e);
/* InterporationWeight setters (src/model/interporation_weight.rs:48-126);
 * which: 0 duration, 1 parameter[stream], 2 gv[stream]. */
int jb_engine_set_interpolation_weight(jb_engine *e, int which, size_t stream, const double *w,
                                       size_t n);
/* InterporationWeight::{get_duration, get_parameter, get_gv} (interporation_weight.rs:115-125): *n = number
 * of voices; w (may be NULL) receives the weights, JB_ERR_BUFFER if cap is too small. */
int jb_engine_get_interpolation_weight(const jb_engine *e, int which, size_t stream, double *w, size_t cap,
                                       size_t *n);

/* Engine::synthesize (src/engine.rs:294): label lines ("label" or "start end label").
 * *pcm is library-owned; release with jb_pcm_free.  Zero labels => n_samples 0. */
int jb_synthesize(const jb_engine *e, const char *const *label_lines, size_t n_lines,
                  double **pcm, size_t *n_samples);
void jb_pcm_free(double *pcm);
/* 16-bit mono RIFF/WAVE writer for the i16 sink -- what the reference's examples do with hound
 * (examples/is-bonsai/main.rs:37-49: 1 channel, 16 bits, SampleFormat::Int).  JB_ERR_MODEL (Io) if the
 * file cannot be written. */
int jb_write_wav_i16(const char *path, const int16_t *pcm, size_t n_samples, uint32_t sampling_frequency);
/* Same from f64 samples, converting like jb's i16 sink: value.min(32767).max(-32768) as i16. */
int jb_write_wav_f64(const char *path, const double *pcm, size_t n_samples, uint32_t sampling_frequency);

/* Batched synthesize: utterance u has lines [line_off[u], line_off[u+1]).  New
 * entry (the reference is single-utterance); a Rust `Engine::synthesize_batch`
 * would sit on it.  pcm[u] library-owned (jb_pcm_free each). */
int jb_synthesize_batch(const jb_engine *e, const char *const *label_lines,
                        const size_t *line_off, size_t n_utts, int32_t device, double **pcm,
                        size_t *n_samples);
/* Same with the 16-bit sink fused into the vocoder (what the reference's callers do with the
 * result: clamp to i16 and write WAV, examples/is-bonsai/main.rs:37-49): a quarter of the PCIe
 * traffic.  pcm[u] library-owned (jb_pcm_i16_free each). */
int jb_synthesize_batch_i16(const jb_engine *e, const char *const *label_lines,
                            const size_t *line_off, size_t n_utts, int32_t device,
                            int16_t **pcm, size_t *n_samples);
void jb_pcm_i16_free(int16_t *pcm);
/* The same two over a device list: the utterances are split by LPT on their label counts (the frame
 * counts are known only after the front half), one host thread per device runs jb_synthesize_batch's
 * path on its share (front half on that thread's workers, GPU work on that device). */
int jb_synthesize_batch_multi(const jb_engine *e, const char *const *label_lines, const size_t *line_off,
                              size_t n_utts, const int32_t *devices, size_t n_devices, double **pcm,
                              size_t *n_samples);
int jb_synthesize_batch_i16_multi(const jb_engine *e, const char *const *label_lines, const size_t *line_off,
                                  size_t n_utts, const int32_t *devices, size_t n_devices, int16_t **pcm,
                                  size_t *n_samples);

/* Host front half only (tree search + durations): fills a state-level utterance
 * owned by the returned handle; used by tests and by jb_synthesize itself. */
typedef struct jb_states jb_states;
int jb_engine_states(const jb_engine *e, const char *const *label_lines, size_t n_lines,
                     jb_states **out);
const jb_state_utt *jb_states_utt(const jb_states *s);
/* Models::duration() (src/model/mod.rs:80-92): the (mean, variance) pairs the durations were estimated from, blended
 * over the voices with the duration weights -- [num_states][2] doubles owned by the handle (NULL for no states).
 * What the reference's `multiple_models` test pins for two voices (src/model/mod.rs:395-428). */
const double *jb_states_duration_params(const jb_states *s);
const jb_voice_desc *jb_engine_voice_desc(const jb_engine *e);
void jb_states_free(jb_states *s);

/* Model introspection (pub fields of Voice/Model, src/model/voice/model.rs:12-82).
 * kind: 0 duration model, 1+s stream model s, 4+s GV model of stream s. */
int jb_engine_model_shape(const jb_engine *e, size_t voice, int kind, size_t *ntree,
                          size_t *pdf_len);
/* pdf table of one tree: *table -> npdf*pdf_len f32 (means, variances, [msd]); engine-owned. */
int jb_engine_pdf_table(const jb_engine *e, size_t voice, int kind, size_t tree,
                        const float **table, size_t *npdf);
/* Model::get_index (src/model/voice/model.rs:51-68): tree_state = matched tree's state
 * (or -1 when none has that state index), pdf_index 1-based. */
int jb_engine_tree_index(const jb_engine *e, size_t voice, int kind, int state_index,
                         const char *label, int *tree_state, int *pdf_index);

/* Engine::generator (src/engine.rs:301) + SpeechGenerator (src/speech.rs:25-96). */
int jb_generator_new(const jb_engine *e, const char *const *label_lines, size_t n_lines,
                     jb_generator **out);
/* SpeechGenerator::new(fperiod, vocoder, spectrum, lf0, lpf) on tracks the caller holds (src/speech.rs:25-50),
 * for jb_generator_step = generate_step (:65-82).  Of `voice` the Vocoder::new arguments are read, as in
 * jb_batch_create_from_tracks; the three panics of SpeechGenerator::new come back as JB_ERR_INVALID with the
 * reference's messages.  opts may be NULL (current device, defaults); the 16-bit sink flag is ignored
 * (generate_step hands out f64 samples). */
int jb_generator_new_from_tracks(const jb_voice_desc *voice, const jb_track_utt *utt, const jb_batch_opts *opts,
                                 jb_generator **out);
size_t jb_generator_fperiod(const jb_generator *g);
size_t jb_generator_synthesized_frames(const jb_generator *g);
size_t jb_generator_total_frames(const jb_generator *g);
/* generate_step: writes fperiod samples to buf, returns fperiod, 0 when exhausted,
 * or a negative jb_status (JB_ERR_BUFFER where the reference panics). */
long jb_generator_step(jb_generator *g, double *buf, size_t buf_len);
/* Up to max_frames generate_step calls in one: writes n * fperiod samples to buf, n = min(max_frames,
 * frames left, buf_len / fperiod), and returns that sample count (0 when exhausted, JB_ERR_BUFFER if buf
 * cannot hold one frame).  One device-to-host copy for the n frames.
 * How the generator works: the whole utterance is enqueued on the device when the generator is made (the
 * path of jb_synthesize: nothing a SpeechGenerator holds can change between steps) and the call returns
 * without waiting; steps hand out the finished PCM.  While the utterance is still in flight the first 8
 * single-frame steps are served by the serial recursion with persistent state on a side stream, so the
 * first frame does not wait for the last. */
long jb_generator_step_n(jb_generator *g, double *buf, size_t buf_len, size_t max_frames);
void jb_generator_free(jb_generator *g);

/* ------------------------------------------------------------------------ */
const char *jb_last_error(void);
int jb_device_count(void);
/* "gfx950" etc. of device `dev` into buf. */
int jb_device_arch(int dev, char *buf, size_t cap);
/* hipDeviceGetPCIBusId of device `dev` ("0000:05:00.0") into buf: which CARD a rank really runs on -- two ranks that
 * report the same id share one GPU, whatever WORLD_SIZE says (bench.py --gpus N puts it into its line). */
int jb_device_pci_bus_id(int dev, char *buf, size_t cap);
const char *jb_version(void);
/* The tolerance of the chunk hand-off check a batch runs with when jb_batch_opts.verify_tol is 0 (1e-9 of the largest
 * state value): the ONE number the PCM gates of the tests and sweeps are derived from (tests/helpers.py). */
double jb_default_verify_tol(void);

#ifdef __cplusplus
}
#endif

/* Layout of every struct that crosses the boundary, as a binding in another language must declare it
 * (LP64, natural alignment: what `#[repr(C)]` and ctypes.Structure give).  Checked here at compile time
 * and against the ctypes mirror in tests/test_abi.py; INTEGRATION.md section 1 carries the same table. */
#if defined(__cplusplus) || (defined(__STDC_VERSION__) && __STDC_VERSION__ >= 201112L)
#ifdef __cplusplus
#define JB_LAYOUT_ASSERT(c, m) static_assert(c, m)
#else
#define JB_LAYOUT_ASSERT(c, m) _Static_assert(c, m)
#endif
JB_LAYOUT_ASSERT(sizeof(jb_stream_desc) == 56 && offsetof(jb_stream_desc, win_width) == 16 &&
                     offsetof(jb_stream_desc, win_coef) == 48, "jb_stream_desc");
JB_LAYOUT_ASSERT(sizeof(jb_voice_desc) == 216 && offsetof(jb_voice_desc, alpha) == 24 &&
                     offsetof(jb_voice_desc, stream) == 48, "jb_voice_desc");
JB_LAYOUT_ASSERT(sizeof(jb_stream_states) == 64 && offsetof(jb_stream_states, gv_weight) == 48, "jb_stream_states");
JB_LAYOUT_ASSERT(sizeof(jb_state_utt) == 208 && offsetof(jb_state_utt, durations) == 8 &&
                     offsetof(jb_state_utt, stream) == 16, "jb_state_utt");
JB_LAYOUT_ASSERT(sizeof(jb_batch_opts) == 32 && offsetof(jb_batch_opts, verify_tol) == 16 &&
                     offsetof(jb_batch_opts, reserved0) == 24, "jb_batch_opts");
JB_LAYOUT_ASSERT(sizeof(jb_pdf_table) == 16 && offsetof(jb_pdf_table, n_rows) == 8, "jb_pdf_table");
JB_LAYOUT_ASSERT(sizeof(jb_index_stream) == 112 && offsetof(jb_index_stream, weight) == 64 &&
                     offsetof(jb_index_stream, gv_weight) == 96, "jb_index_stream");
JB_LAYOUT_ASSERT(sizeof(jb_index_utt) == 360 && offsetof(jb_index_utt, stream) == 16 &&
                     offsetof(jb_index_utt, lf0_offset) == 352, "jb_index_utt");
JB_LAYOUT_ASSERT(sizeof(jb_track_utt) == 64 && offsetof(jb_track_utt, spectrum_width) == 24 &&
                     offsetof(jb_track_utt, spectrum) == 40, "jb_track_utt");
#undef JB_LAYOUT_ASSERT
#endif
#endif /* JBONSAI_AMD_H */
