// jb_device.h -- device-side data model and kernel launchers (internal).
//
// HBM layout of one batch (B utterances, independent):
//   * per-utterance descriptor table  UttDev[B]  (pointers into de-duplicated
//     state arrays; utterances that alias the same host arrays share one copy)
//   * frame-indexed arrays are concatenated over utterances, utterance b owning
//     frames [frame_off[b], frame_off[b]+T_b); frame-major, vector dim fastest,
//     so that "lane = vector dim" (MLPG) and "lane = sample" (PCM) accesses are
//     coalesced.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace jb {

constexpr int kMaxStream = 3;
constexpr int kMaxWin = 8;
constexpr int kMaxCoef = 32;
constexpr int kMaxBand = 9;  // band width 2 * max_width + 1 of the generic MLPG kernels: windows of up to nine taps
constexpr int kPade = 5;          // MelLogSpectrumApproximation<6>: 5 stages (src/vocoder/mlsa.rs:31)
constexpr int kGroups = 12;       // lane groups per Pade stage: 5*12 = 60 lanes of a wave64
constexpr int kMaxTPL = 6;        // taps per lane of the wave kernels => nmcp-1 <= 72
constexpr int kMaxNmcp = 64;      // ... and the 64-lane kernels (mc2b tile, post-filter, MGLSA): nmcp <= 64
constexpr double kNoData = -1e10; // src/constants.rs:13

struct StreamStatesDev {
    const double *mean;   // [S][W*L]
    const double *var;    // [S][W*L]
    const double *msd;    // [S] or nullptr
    const double *gv_mean;// [L] or nullptr
    const double *gv_var; // [L]
    const uint8_t *gv_switch; // [S]
    double gv_weight;
    double msd_threshold;
};

struct UttDev {
    uint32_t S;
    uint32_t T;
    uint64_t frame_off;       // first frame in concatenated frame arrays
    uint64_t state_off;       // first state in concatenated state scratch
    const uint32_t *dur;      // [S]
    StreamStatesDev st[kMaxStream];
    // the inverse-variance table of a stream (StreamDev::ivar) is a function of the variance array alone: utterances
    // that share one (copies of an utterance: uploads are de-duplicated) share the table of the first of them
    uint64_t ivar_state_off[kMaxStream]; // first row of this utterance's table
    uint8_t ivar_owner[kMaxStream];      // 1: this utterance computes it (k_mlpg_ivar)
    // [dim][frame] workspace (StreamDev::mt): row m of this utterance starts at element mt_off * L + m * mt_rs of
    // A / bvec / F / g / par.  mt_rs = T rounded up to 16 frames and mt_off a multiple of 16: every row starts on a
    // 128-byte line, so that the 16-frame pieces the build, the band solve's movers and the GV kernel move are whole
    // lines (with rows at frame_off * L + m * T a piece straddled two: 12-15 % more bytes written than stored)
    uint64_t mt_off;
    uint32_t mt_rs;
};
__device__ __forceinline__ uint64_t mt_row0(const UttDev *up, int L) { return up->mt_off * (uint64_t)L; }

struct StreamDev {
    int L, W, is_msd, use_gv;
    int BW;                   // band width = 2*max_width+1 (src/mlpg_adjust/mlpg.rs:27)
    int win_width[kMaxWin];
    int win_off[kMaxWin];
    double win_coef[kMaxCoef];
    int generic_solver;       // 1: force the un-fused reference-shaped solver (A/B tests)
    // Workspace layout of A/bvec/F/g/par.  mt == 0: [frame][dim] (dim fastest).  mt == 1:
    // [dim][frame] per utterance (row m of utterance b starts at frame_off*L + m*T, frames
    // contiguous), which is what the time-parallel GV sweeps and the per-lane streams of the
    // serial substitutions want.  `out` is always [frame][dim].
    int mt;
    int serial_gv;            // 1: GV reductions in the reference's serial order (bit-exact tracks)
    int defer_out;            // 1: launch_mlpg leaves `par` ([dim][frame]) and launch_mc2b_mt produces out/bcoef
    // time-parallel GV (k_mlpg_gv_tp): per-tile partial sums and per-iteration scalars
    double *ivar;             // [sumS][W*L] MeanVari::with_ivar of the state variances (k_mlpg_ivar), mt only
    double *gv_part;          // [7 passes][B][L][gv_ntile][4]
    double *gv_scal;          // [6][B][L][4] = {mean, step, obj, -}
    uint32_t gv_ntile;
    // resident GV (k_mlpg_gv_gang): control block + gang records (device), gangs and tiles per gang of
    // this batch; null = the multi-launch sweeps (k_mlpg_gv_tp)
    void *gv_gang_ctl;
    int gv_gang_n, gv_gang_tiles;
    // rows of several SHORT utterances share one pass of a gang (round 4): a bin = utterances whose tiles fit the
    // gang's, entry [bin][tile] = which utterance and which of its tiles that workgroup works on; the queue hands
    // out (bin, dim) groups.  (A ragged batch -- BASELINE config 3 -- took 16.4 ms where equal lengths take 8.)
    const struct GvBinEntry *gv_bins; // [gv_nbins][gv_gang_tiles]
    uint32_t gv_nbins;
    // ---- per-state scratch written by k_prep_states (concatenated states) ----
    uint32_t *s_start;  // [sumS] first frame of state
    uint32_t *s_vpre;   // [sumS] voiced frames before state (compaction offset)
    uint32_t *s_rstart; // [sumS] first frame of the voiced run containing the state
    uint32_t *s_rend;   // [sumS] last frame of that run
    uint8_t *s_voiced;  // [sumS]
    uint32_t *run_list; // [sumS] first frame of each voiced run, compact per utterance
    uint32_t *nruns;    // [B]
    // ---- per-frame scratch written by k_prep_frames (concatenated frames) ----
    uint32_t *fstate;   // [sumT] state index of frame
    uint8_t *voiced;    // [sumT]
    uint8_t *fl, *fr;   // [sumT] boundary distances clipped to 255 (mask.rs:51-82)
    uint32_t *vidx;     // [sumT] compacted index -> frame (voiced frames only)
    uint8_t *vsw;       // [sumT] gv switch per compacted frame
    uint32_t *Tv;       // [B] number of voiced frames
    uint32_t *gvlen;    // [B] number of switched-on voiced frames
    // ---- MLPG workspace, each [sumT][L] ----
    double *A[kMaxBand]; // un-factored W'U^-1W band (needed again by GV)
    double *bvec;       // W'U^-1 mu
    double *F[kMaxBand]; // LDL^T factors
    double *g;          // forward-substitution result / GV gradient
    double *par;        // compacted solution
    double *out;        // [sumT][L] scattered parameter track (NODATA in unvoiced frames)
    // LPF stream, static path only (k_mlpg_static): canon[f] = 1 while every (mean, var) the frame's row was made
    // from equals, bit for bit, that of the batch's first frame (utterance canon_ref_utt's first state) -- then so
    // does the row.  Preset to 1 per run; a thread that sees a difference stores 0.  Null: not tracked.
    uint8_t *canon;     // [canon_n = sumT]
    uint64_t canon_n;
    uint32_t canon_ref_utt;
    // 1: k_mlpg_static does not STORE a canonical row (but the batch's first): its readers take row 0 instead
    // (VocDev::lpf_sparse).  Not with JB_BATCH_KEEP_TRACKS: then the track is a result.
    int canon_skip_rows;
};

// One workgroup's share of a bin (above): utterance b (0xffffffff: none), tile k of its nt tiles, which stand at
// tiles t0 .. t0 + nt - 1 of the gang; flags bit 0: some utterance of the bin has a GV pdf and frames.
struct GvBinEntry {
    uint32_t b;
    uint8_t k, t0, nt, flags;
};
// Control block of one launch of k_mlpg_gv_gang (zeroed before the launch) and per-gang records.
// measurement aid of the resident GV kernel (tools/gg_prof.sh builds EVERY translation unit with -DJB_GG_PROFILE=1):
// defined here, once, and tested with #if everywhere, so that the layout of GvGang cannot differ between units
#ifndef JB_GG_PROFILE
#define JB_GG_PROFILE 0
#endif
constexpr int kGvGangMaxTiles = 64; // workgroups per gang (one poller lane each): rows of up to 64 tiles
struct GvGangCtl {
    uint32_t tickets;  // next workgroup to start -> (gang, tile)
    uint32_t next_row; // row queue: (utterance, dim) rows in launch order
    uint32_t err;      // a bounded spin ran out
    uint32_t pad[13];
    unsigned long long prof[8]; // -DJB_GG_PROFILE: shader-clock ticks of thread 0 of every workgroup, by section
};
struct GvGang {
    uint32_t cnt; // arrivals: formation, then one per tile and exchange
    uint32_t pad[31];
    // {S1, S2, H, next row} per tile as four 16-byte granules {value, value ^ tag}, two alternating slots
    unsigned long long rec[2][kGvGangMaxTiles][8];
#if JB_GG_PROFILE
    // measurement aid (-DJB_GG_PROFILE builds of every translation unit): the 100 MHz real-time clock of thread 0 of
    // each of the gang's first 8 tiles at its first 96 exchanges -- [0]: its sums are ready (arrival), [1]: the
    // exchange is over
    unsigned long long stamp[2][8][96];
#endif
};
static_assert(sizeof(GvGang) == 128 + sizeof(unsigned long long) * 2 * kGvGangMaxTiles * 8 +
                                    (JB_GG_PROFILE ? sizeof(unsigned long long) * 2 * 8 * 96 : 0),
              "GvGang layout (host offsets in jb_batch.cpp and the kernel must agree)");

struct VocDev {
    int fs, fperiod, nmcp, nlpf, bs, nblk; // bs = samples per block (divides fperiod, <=64)
    double alpha, volume;
    const double *mcp;    // [sumT][nmcp] (MLPG out of stream 0)
    const double *lf0;    // [sumT]
    const double *lpf;    // [sumT][nlpf]
    double *bcoef;        // [sumT][nmcp]  mc2b(mcp)  (src/vocoder/cepstrum.rs:139-149)
    double beta;          // post-filter coefficient (cepstrum.rs:23-37); 0 = off
    // Stage::NonZero (GAMMA != 0, stage.rs:24-39; jb_mglsa.hip): stage > 0 selects it; its post-filter
    // coefficient (postfilter_lsp, lsp.rs:113-139) is beta_stage, `beta` above stays 0 then
    int stage, use_log_gain;
    double beta_stage;
    double *bfirst;       // [B][nmcp] un-filtered bcoef of each utterance's first frame, or nullptr (beta == 0)
    double *pf_table;     // [nmcp][576] freqt(575, -alpha) as a linear operator (k_pf_table), or nullptr
    double *pf_rcp;       // [576] 1/n
    double *pitch;        // [sumT]  period in samples, 0 = unvoiced
    double *cur_start;    // [sumT]  pitch_of_curr_point at frame start
    double *pinc;         // [sumT]  pitch_inc_per_point
    unsigned long long *pmask; // [sumT][nblk] pulse bit per sample of each block
    const uint8_t *voiced;     // [sumT] MSD voiced flag of the LF0 stream (k_prep_frames)
    const uint32_t *run_list;  // voiced runs of the LF0 stream (k_prep_states)
    const uint32_t *nruns;
    uint32_t *run_base;        // [B+1] exclusive prefix of nruns (k_run_scan) for the pulse work queue
    uint32_t *run_counter;     // [1] next run to hand out
    const double *noise;  // [noise_len] shared Gaussian stream
    uint64_t noise_len;
    double *xin;          // [sumT*fperiod] excitation after the LPF mix (k_excite*), gain not yet applied
    // 1: frames that are unvoiced and follow an unvoiced frame (never the first of an utterance) have no
    // entry in xin: their excitation IS the shared noise stream, x[n] = noise[n - (nlpf-1)/2]
    // (excitation.rs:43-45,83-86), and the vocoder kernels read it there (exc_frame_ptr)
    int skip_unvoiced;
    // Shared pulse-free excitation (split form only; null otherwise).  Before its pulses are added, the
    // excitation of a voiced frame behind a voiced frame depends on nothing but the frame's position, the
    // shared noise stream and the LPF taps of the two frames; where both rows of taps equal the canonical
    // taps (those of the batch's first frame: HTS voices carry ONE low-pass filter, the nitech voice too),
    // it is the same for every utterance of the batch.  exc_tab[n] holds that excitation for every sample
    // position (k_exc_table, once per run, the arithmetic of the per-frame pass); the per-frame pass
    // (k_exc_general) then only computes the frames of exc_gen (frames at a voiced/unvoiced boundary, frames
    // whose taps differ, first frames), and the pulse pass writes a row of xin only for frames a pulse reaches.
    // exc_src[f]: where the vocoder finds frame f's excitation -- bits 0..1: 0 its row of xin, 1 the noise stream
    // (unvoiced behind unvoiced), 2 exc_tab; bits 4..7 (with 2): the blocks of bs samples that a pulse reaches and
    // that therefore have a piece of a row of their own in xin (the pulse pass writes only those: round 4).
    double *exc_tab;          // [maxT * fperiod]
    const uint8_t *lpf_canon; // [sumT] StreamDev::canon of the LPF stream, or nullptr: k_exc_classify compares the rows
    int lpf_sparse;           // 1: rows of vd.lpf exist only for frames with lpf_canon == 0 and for frame 0 of the
                              // batch, which every canonical frame's row equals bit for bit (lpf_row below)
    uint8_t *exc_src;         // [sumT]
    uint32_t *exc_gen;        // [sumT][2] = (utterance, frame | vcur << 30 | vprev << 31)
    uint32_t *exc_gen_count;  // [1]
    int exc_no_table;         // 1: every frame goes through the per-frame pass (JB_BATCH_NO_EXC_TABLE, debug tap)
    // nlpf == 0 (Excitation::get without a ring buffer, excitation.rs:87-100: noise is drawn on unvoiced samples
    // only): unvoiced frames of the utterance before each frame, i.e. its place in the noise stream (k_uv_scan)
    uint32_t *uv_before;      // [sumT], or nullptr
    double *pcm;          // [sumT*fperiod] f64 PCM, or nullptr when the i16 sink is selected
    int16_t *pcm16;       // [sumT*fperiod] clamped i16 PCM (JB_BATCH_PCM_I16), or nullptr
    double *exc;          // optional [sumT*fperiod] excitation before gain, or nullptr
    double *state;        // optional per-utterance filter state (streaming), or nullptr
    int state_stride;     // doubles per utterance
    uint32_t ckpt_frames; // checkpoint position inside a chunk (frames past t_out); 0 = no checkpoints
    uint32_t ckpt2_frames; // second checkpoint (long chunks only), 0 = none
};

// The row of LPF taps of frame f (index in the concatenated arrays), NL taps per row
__device__ __forceinline__ const double *lpf_row(const VocDev &vd, uint64_t f, int NL)
{
    return vd.lpf + ((vd.lpf_sparse && vd.lpf_canon[f]) ? 0ull : f) * (uint64_t)NL;
}

// Where the vocoder finds the excitation of block q (bs samples) of frame t of an utterance (base = its first
// frame in the concatenated arrays): the stored row of xin, or -- for a frame the excitation kernels skipped -- the
// shared noise stream at the frame's first sample minus the ring buffer's delay, or the shared pulse-free table.
// exc_code: the frame's exc_src byte (or its equivalent where there is none); exc_block_ptr: a pointer p with
// p[i] = sample i OF THE FRAME, valid for the samples of block q.
__device__ __forceinline__ uint32_t exc_code(const VocDev &vd, uint64_t base, uint32_t t)
{
    if (vd.exc_src)
        return vd.exc_src[base + t];
    return (vd.skip_unvoiced && t >= 1 && !vd.voiced[base + t] && !vd.voiced[base + t - 1]) ? 1u : 0u;
}
__device__ __forceinline__ const double *exc_block_ptr(const VocDev &vd, uint64_t base, uint32_t t, uint32_t code, int q)
{
    const uint32_t src = code & 3u;
    if (src == 1)
        return vd.noise + ((uint64_t)t * (uint64_t)vd.fperiod - (uint64_t)((vd.nlpf - 1) / 2));
    if (src == 2 && !((code >> (4 + q)) & 1u))
        return vd.exc_tab + (uint64_t)t * (uint64_t)vd.fperiod;
    return vd.xin + (base + t) * (uint64_t)vd.fperiod;
}

// One unit of vocoder work: output frames [t_out, t_end) of utterance `utt`, with the
// recursion started at t_start <= t_out (frames [t_start, t_out) are warm-up: computed,
// not stored).  load_state != nullptr resumes from a saved filter state instead of zeros.
struct VocWork {
    uint32_t utt, t_start, t_out, t_end;
    const double *load_state; // exact continuation (streaming / re-do), or nullptr
    double *save_warm;        // state on entering t_out (after warm-up), or nullptr
    double *save_end;         // state after t_end, or nullptr
    double *save_ckpt;        // state on entering frame t_out + VocDev::ckpt_frames, or nullptr (partial redo)
    double *save_ckpt2;       // state on entering frame t_out + VocDev::ckpt2_frames, or nullptr
};
// A failing chunk is first recomputed only up to VocDev::ckpt_frames frames past its start; if the
// recomputed state meets the checkpoint the original chunk left there, the rest of the chunk stands.
// 48 frames into chunks of 96 frames and more, 24 into chunks of 36 to 95, 16 into chunks of 24 to 35, none below
// (Batch::build_work).  A redo round lasts as long as the frames to the checkpoint (0.06 ms per frame, one wave per
// chunk) and ALL of a round's chunks wait for the one that goes furthest.  A hand-off that failed behind 18 frames
// of warm-up settles at the checkpoint if 18 + 48 frames from zero state are enough.  Tried in round 4: 32 frames
// (-1 ms per round) -- but about 1 % of the failing hand-offs have not converged there, and a batch of 512 or 1024
// distinct utterances (BASELINE configs 3 to 5: ~300 failing hand-offs) then nearly always has one and pays the
// second stage: 2.0 + 3.0 ms instead of 2.9 (tools/ckpt_sweep.sh: 1024 x 6,386 frames 86.6 / 86.3 / 83.9 / 84.6 ms
// per step with the first checkpoint at 32 / 40 / 48 / 56).
// Chunks of 144 frames and more leave a SECOND checkpoint 96 frames in: the rare chunk that has not converged at
// the first one is recomputed 48 frames further and compared again, instead of to its end (105 frames = 6.4 ms).
// chunk hand-off check: max|state diff| <= tol * max|state| (jb_batch_opts.verify_tol = 0; jb_default_verify_tol())
constexpr double kDefaultVerifyTol = 1e-9;
constexpr uint32_t kVocCkptFrames = 48, kVocCkptFramesShort = 24, kVocCkptFramesTiny = 16, kVocCkpt2Frames = 96;

// Timing experiments only (library built with -DJB_DBG_GATES, never the product): JB_DBG_SKIP is a bit mask of
// launches to leave out once a launcher has been called JB_DBG_SKIP_AFTER times (default 2: bench.py's warm-up
// steps run in full, so the workspaces the skipped kernels would have written hold the right values) --
// the ceiling of what removing a kernel can be worth (tools/gate.sh).  1 MCP build, 2 MCP band solve,
// 4 pulse-free excitation pass, 8 pulse repair pass, 16 resident GV, 32 LF0 GV, 64 pulse walk.
#ifdef JB_DBG_GATES
#include <cstdlib>
inline bool dbg_skip(int bit, int &calls)
{
    const char *m = getenv("JB_DBG_SKIP"), *a = getenv("JB_DBG_SKIP_AFTER");
    const int after = a ? atoi(a) : 2;
    return m && (atoi(m) & bit) && calls++ >= after;
}
#define JB_DBG_SKIP_IF(bit, stmt_else)                                                                         \
    do {                                                                                                        \
        static int calls_ = 0;                                                                                  \
        if (!dbg_skip(bit, calls_)) {                                                                           \
            stmt_else;                                                                                          \
        }                                                                                                       \
    } while (0)
#else
#define JB_DBG_SKIP_IF(bit, stmt_else)                                                                         \
    do {                                                                                                        \
        stmt_else;                                                                                              \
    } while (0)
#endif

struct BatchDev {
    int B;
    const UttDev *utt;        // device
    const uint32_t *order;    // [B] launch order (longest first)
    uint32_t maxT;
    uint32_t maxS;
};


// One gather + blend job: S states of one stream of one utterance.
// mean[s][k] = sum_v w[v] * tab[v][row[v][s]][k], var likewise from the second half of the row,
// msd from the trailing element (voice_set.rs:80-95: first*w0, then += w_i*p_i in voice order).
constexpr int kMaxVoices = 8;
struct GatherJob {
    const uint32_t *row[kMaxVoices]; // device, [S]
    const float *tab[kMaxVoices];    // device, [n_rows][row_len]
    double w[kMaxVoices];
    uint32_t nv, S, WL, row_len;
    int has_msd;
    double lf0_offset;               // != 0: static LF0 mean += offset, clamped (stream_parameter.rs:29-37)
    double *mean, *var, *msd;        // device outputs [S][WL], [S][WL], [S] (msd may be null)
};
hipError_t launch_gather(const GatherJob *jobs_dev, uint32_t n_jobs, uint64_t max_elems, hipStream_t stream);

// launchers (all asynchronous on `stream`)
hipError_t launch_prep(const BatchDev &bd, const StreamDev &sd, int stream_index, hipStream_t stream);
// after_build (optional) is recorded once A/bvec are built, before the serial sweeps start;
// between (optional) is called after the band solve has been enqueued and before the GV sweeps
// are: the caller may enqueue other work on `stream` there.
typedef hipError_t (*jb_enqueue_hook)(void *ctx, hipStream_t stream);
// after_ivar (optional) is recorded behind the inverse-variance pass of the [dim][frame] path (at once on the others)
hipError_t launch_mlpg(const BatchDev &bd, const StreamDev &sd, int stream_index, hipStream_t stream,
                       hipEvent_t after_build, jb_enqueue_hook between = nullptr, void *between_ctx = nullptr,
                       hipEvent_t after_ivar = nullptr);
// resident GV: plan (0 = not applicable: row too long for a gang, no device capacity), bytes of the
// control block for n gangs, launch (memset of the control block + the persistent kernel)
int gv_gang_plan(int device, uint32_t maxT, uint32_t n_rows, int *tiles_per_gang, int *n_gangs);
// bins of utterances (first fit in launch order, longest first) for gangs of tiles_per_gang tiles; out: [bins][tiles]
void gv_gang_bins(const uint32_t *T, const uint8_t *has_gv, const uint32_t *order, size_t n, int tiles_per_gang,
                  std::vector<GvBinEntry> &out);
size_t gv_gang_ctl_bytes(int n_gangs);
hipError_t launch_gv_gang(const BatchDev &bd, const StreamDev &sd, int si, hipStream_t stream);
int mlpg_mt_max_dim();      // largest vector length served by the [dim][frame] fast path
int mlpg_gv_tile_frames();  // frames per block of the time-parallel GV sweeps
hipError_t launch_pitch(const BatchDev &bd, const VocDev &vd, hipStream_t stream);
hipError_t launch_mc2b(const BatchDev &bd, const VocDev &vd, hipStream_t stream);
// MCP stream on the [dim][frame] workspace: transpose + mc2b in one pass over `par`
// (write_out: also materialise the [frame][dim] parameter track for jb_batch_read_track)
hipError_t launch_mc2b_mt(const BatchDev &bd, const StreamDev &sd, const VocDev &vd, bool write_out,
                          hipStream_t stream);
hipError_t launch_pulse(const BatchDev &bd, const VocDev &vd, hipStream_t stream);
// Mixed excitation.  When excite_is_split(vd): launch_excite_noise needs only the LF0 stream's
// voiced flags (k_prep) and the LPF track and computes every sample as if there were no pulses;
// launch_excite (after k_pulse) then recomputes the <= nlpf samples after each pulse exactly.
// Otherwise launch_excite does everything in one pass and launch_excite_noise is a no-op.
bool excite_is_split(const VocDev &vd);
hipError_t launch_excite_noise(const BatchDev &bd, const VocDev &vd, hipStream_t stream);
hipError_t launch_excite(const BatchDev &bd, const VocDev &vd, hipStream_t stream);
// X1 post-filter (cepstrum.rs:23-37): constant tables once, then bcoef in place for every frame
hipError_t launch_pf_table(const VocDev &vd, hipStream_t stream);
hipError_t launch_postfilter(const BatchDev &bd, const VocDev &vd, uint64_t nframes, hipStream_t stream);
hipError_t launch_vocoder(const BatchDev &bd, const VocDev &vd, const VocWork *work_dev, uint32_t n_items,
                          hipStream_t stream);
// lane-serial throughput kernel (one chunk per lane); order_dev = launch permutation of items
bool vocoder_ls_supported(int nmcp);
int vocoder_ls_chunks_per_wave(int nmcp); // 21 (lane triples: orders up to 34) or 12 (one stage per lane)
// waves_per_simd: 2 = eight-wave workgroups (a whole CU), 1 = four-wave workgroups
hipError_t launch_vocoder_ls(const BatchDev &bd, const VocDev &vd, const VocWork *work_dev,
                             const uint32_t *order_dev, uint32_t n_items, int waves_per_simd, hipStream_t stream);
// compares save_warm of item i with save_end of item i-1 (same utterance): bad[i]=1 and
// ++*n_bad when max|diff| > tol * max|state|
// (carried-state slots only: ntaps = nmcp - 1 live taps, see voc_state_differs)
hipError_t launch_voc_verify(const VocWork *work_dev, uint32_t n_items, int state_doubles, int ntaps, double tol,
                             uint8_t *bad, uint32_t *n_bad, hipStream_t stream);
int vocoder_state_doubles(int nmcp);
// Stage::NonZero: per-frame coefficients (bcoef, bfirst) from the [frame][dim] LSP track; the filter
// (dispatched by launch_vocoder when vd.stage > 0); doubles of its state dump
hipError_t launch_stage_coef(const BatchDev &bd, const VocDev &vd, hipStream_t stream);
hipError_t launch_vocoder_mglsa(const BatchDev &bd, const VocDev &vd, const VocWork *work_dev, uint32_t n_items,
                                hipStream_t stream);
int mglsa_state_doubles(int stage);
int excite_max_nlpf(); // low-pass taps the excitation kernels take (jb_vocoder.hip)
int mglsa_max_stage(); // Stage::NonZero: 1..8 with the delay lines in registers, up to this many with them in LDS
// bad[j] = 1 (and ++*n_bad) when states pairs[2j] and pairs[2j+1] differ by more than tol * max|state|
hipError_t launch_voc_verify_pairs(const double *const *pairs_dev, uint32_t n_pairs, int state_doubles, int ntaps,
                                   double tol, uint8_t *bad, uint32_t *n_bad, hipStream_t stream);

} // namespace jb
