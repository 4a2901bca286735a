/*
 * oracle/jbo.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C `double` restatement of jbonsai's CPU path, in the reference's own
 * operation order, used ONLY as the checker in tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.  Nothing under jbonsai_amd/ may link, load
 * or call anything declared here.
 *
 * The reference (Rust, crate jbonsai v0.4.2) cannot be compiled in this image
 * (no cargo/rustc); parity is pinned by the reference's own golden samples
 * (/root/reference/src/lib.rs:39-160), duration vectors (src/duration.rs:144-179),
 * state pdfs (src/model/mod.rs:191-392) and mask tests (src/mlpg_adjust/mask.rs:89-159),
 * all of which tests/test_oracle_golden.py checks.
 *
 * Every function cites the reference file:line it restates.
 */
#ifndef JBO_H
#define JBO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JBO_MAX_STREAM 3
#define JBO_MAX_VOICES 8
#define JBO_NODATA (-1e10) /* src/constants.rs:13 */

typedef struct jbo_voice jbo_voice;

/* Synthesis knobs; mirror of Condition (src/engine.rs:31-81). */
typedef struct {
    double speed;                 /* 1.0 */
    double volume;                /* linear, 1.0 */
    double beta;                  /* post-filter coefficient, 0 = off (engine.rs:75,215-218) */
    double additional_half_tone;  /* 0.0 */
    double msd_threshold[JBO_MAX_STREAM]; /* 0.5 */
    double gv_weight[JBO_MAX_STREAM];     /* 1.0 */
    int phoneme_alignment;        /* 0 */
} jbo_cond;

/* Flat, state-level description of one utterance for one stream: the exact
 * image of ModelStream (src/model/model_stream.rs:6-15). */
typedef struct {
    uint32_t vector_length;  /* L */
    uint32_t num_windows;    /* W */
    uint32_t is_msd;
    uint32_t use_gv;
    const uint32_t *win_width; /* [W] */
    const double *win_coef;    /* concatenated, sum(win_width) */
    const double *mean;      /* [S][W*L] */
    const double *var;       /* [S][W*L] */
    const double *msd;       /* [S] (DBL_MAX for non-MSD streams) */
    const double *gv_mean;   /* [L] or NULL */
    const double *gv_var;    /* [L] or NULL */
    const uint8_t *gv_switch;/* [S] or NULL */
    double gv_weight;
    double msd_threshold;
} jbo_stream;

void jbo_cond_default(jbo_cond *c);

/* ---- voice (cold path) ------------------------------------------------ */
jbo_voice *jbo_voice_load(const char *path);
jbo_voice *jbo_voice_load_bytes(const uint8_t *bytes, size_t n);
void jbo_voice_free(jbo_voice *v);
int jbo_voice_sampling_frequency(const jbo_voice *v);
int jbo_voice_fperiod(const jbo_voice *v);
int jbo_voice_nstate(const jbo_voice *v);
int jbo_voice_nstream(const jbo_voice *v);
double jbo_voice_alpha(const jbo_voice *v);
int jbo_voice_stage(const jbo_voice *v);
int jbo_voice_vector_length(const jbo_voice *v, int stream);
int jbo_voice_num_windows(const jbo_voice *v, int stream);
int jbo_voice_is_msd(const jbo_voice *v, int stream);
int jbo_voice_use_gv(const jbo_voice *v, int stream);
/* window w of stream: returns width, writes coefs (cap >= width) */
int jbo_voice_window(const jbo_voice *v, int stream, int w, double *coef, int cap);
/* model kinds: 0 duration, 1+s stream s, 4+s gv of stream s */
int jbo_voice_ntree(const jbo_voice *v, int kind);
int jbo_voice_npdf(const jbo_voice *v, int kind, int tree);
int jbo_voice_pdf_len(const jbo_voice *v, int kind);
const float *jbo_voice_pdf(const jbo_voice *v, int kind, int tree, int pdf_index_1based);
/* tree search: Model::get_index (src/model/voice/model.rs:51-68).
 * Returns 0 and fills tree_state (or -1 when no tree has that state) and
 * 1-based pdf index. */
int jbo_voice_get_index(const jbo_voice *v, int kind, int state_index, const char *label,
                        int *tree_state, int *pdf_index);
int jbo_gv_off(const jbo_voice *v, const char *label);

/* ---- front half: labels -> state-level inputs ------------------------- */
/* Duration pdfs (Models::duration, src/model/mod.rs:80-92): out[2*S] mean,var. */
int jbo_duration_params(const jbo_voice *v, const char *const *labels, int n, double *mean_var);
/* DurationEstimator::create / create_with_alignment (src/duration.rs:28-65).
 * times: [n][2] frames (label time * fs/(fperiod*1e7), src/label.rs:44) or NULL. */
int jbo_durations(const jbo_voice *v, const char *const *labels, int n, double speed,
                  const double *times, uint32_t *dur_out);
/* Models::stream (src/model/mod.rs:98-118): mean/var [S][W*L], msd [S]. */
int jbo_stream_params(const jbo_voice *v, int stream, const char *const *labels, int n,
                      double *mean, double *var, double *msd);
/* Models::gv (src/model/mod.rs:119-146): gv_mean/var [L], switch [S]. */
int jbo_gv_params(const jbo_voice *v, int stream, const char *const *labels, int n,
                  double *gv_mean, double *gv_var, uint8_t *gv_switch);
/* label time alignment parse (src/label.rs:35-79,83-113): lines may be
 * "start end label"; writes label pointers (into lines) and times[n][2]. */
int jbo_parse_label_lines(int fs, int fperiod, const char *const *lines, int n,
                          const char **label_out, double *times_out);

/* ---- hot path --------------------------------------------------------- */
/* Mask::create + boundary_distances (src/mlpg_adjust/mask.rs:20-82). */
size_t jbo_mask(const jbo_stream *st, uint32_t S, const uint32_t *dur, uint8_t *mask /*[T]*/);
void jbo_boundary_distances(const uint8_t *mask, size_t T, size_t *left, size_t *right);
/* MlpgAdjust::create (src/mlpg_adjust/mod.rs:51-95): par [T][L]. */
int jbo_mlpg(const jbo_stream *st, uint32_t S, const uint32_t *dur, double *par /*[T][L]*/);

/* Vocoder + SpeechGenerator::generate_all (src/vocoder/mod.rs:72-141, src/speech.rs:87-96).
 * Stage::Zero only.  lf0[T], mcp[T][nmcp], lpf[T][nlpf]; pcm[T*fperiod].
 * Optional dumps (may be NULL): exc[T*fperiod] = excitation *before* gain,
 * pulse_pos: per-sample pulse amplitude (0 when none) [T*fperiod]. */
int jbo_vocoder(int fs, int fperiod, double alpha, double volume, int nmcp, int nlpf,
                size_t T, const double *lf0, const double *mcp, const double *lpf,
                double *pcm, double *exc, double *pulse);
/* Same with the mel-cepstral post-filter (beta > 0): postfilter_mcp per frame
 * (src/vocoder/cepstrum.rs:23-37).  No reference test exercises beta > 0: parity unpinned. */
int jbo_vocoder_beta(int fs, int fperiod, double alpha, double beta, double volume, int nmcp,
                     int nlpf, size_t T, const double *lf0, const double *mcp, const double *lpf,
                     double *pcm, double *exc, double *pulse);
/* X1 pieces: freqt (cepstrum.rs:153-173), c2ir (:175-186), b2en (coefficients.rs:75-78),
 * postfilter_mcp in place (cepstrum.rs:23-37). */
void jbo_freqt(const double *c1, size_t n1, double *out /*[m2+1]*/, size_t m2, double alpha);
void jbo_c2ir(const double *c, size_t nc, double *ir, size_t len);
double jbo_b2en(const double *b, size_t n, double alpha);
void jbo_postfilter_mcp(double *mc, size_t n, double alpha, double beta);
/* X2: Stage::NonZero (GAMMA != 0: LSP spectra, MGLSA filter).  PARITY UNPINNED (no reference test or
 * voice reaches it): src/vocoder/{lsp.rs, generalized.rs, cepstrum.rs:69-103, mglsa.rs, mod.rs:90-107,142-176}. */
void jbo_lsp2lpc(const double *lsp, size_t m, double *out /*[m+1]*/);
void jbo_gnorm(double *c, size_t n, double gamma);
void jbo_ignorm(double *c, size_t n, double gamma);
void jbo_gc2gc(const double *c1, size_t n1, double g1, double *c2 /*[m2+1]*/, size_t m2, double g2);
void jbo_mgc2mgc(const double *c1, size_t n1, double a1, double g1, double *out /*[m2+1]*/, size_t m2, double a2,
                 double g2);
void jbo_lsp2mgc(const double *lsp, size_t n, double alpha, int use_log_gain, size_t stage, double gamma,
                 double *mgc /*[n]*/);
void jbo_postfilter_lsp(double *lsp, size_t n, double alpha, int use_log_gain, size_t stage, double gamma, double beta);
void jbo_check_lsp_stability(double *lsp, size_t n);
void jbo_stage_coefficients(const double *spectrum, size_t n, double alpha, double beta, int use_log_gain,
                            size_t stage, int filtered, double *cc /*[n]*/);
void jbo_mglsa_df(double *d /*[stage][n]*/, size_t stage, size_t n, double *x, double alpha, const double *c);
int jbo_vocoder_stage(int fs, int fperiod, double alpha, double beta, double volume, int stage, int use_log_gain,
                      int nmcp, int nlpf, size_t T, const double *lf0, const double *mcp, const double *lpf,
                      double *pcm, double *exc);
/* same loop on GIVEN coefficients coef[T][nmcp] / cfirst[nmcp] (the conversion is ill-conditioned) */
int jbo_vocoder_stage_coef(int fs, int fperiod, double alpha, double beta, double volume, int stage, int use_log_gain,
                           int nmcp, int nlpf, size_t T, const double *lf0, const double *mcp, const double *lpf,
                           const double *coef, const double *cfirst, double *pcm, double *exc);
/* Random::nrandom stream (src/vocoder/excitation.rs:177-237), seed next=1. */
void jbo_noise(double *out, size_t n);

/* Engine::synthesize (src/engine.rs:294-366).  Returns malloc'd pcm. */
int jbo_synthesize(const jbo_voice *v, const jbo_cond *c, const char *const *lines, int n,
                   double **pcm, size_t *n_samples);
/* Same but also returns the three parameter tracks (malloc'd; may pass NULL). */
int jbo_synthesize_ex(const jbo_voice *v, const jbo_cond *c, const char *const *lines, int n,
                      double **pcm, size_t *n_samples, uint32_t **dur, uint32_t *S,
                      double **mcp, double **lf0, double **lpf, size_t *T);
/* ---- several voices (VoiceSet::weighted, src/model/voice_set.rs:80-95) ----------------------
 * InterporationWeight (src/model/interporation_weight.rs:48-126): one weight vector [nv] for the
 * durations and one per stream for the parameters and for the GV pdfs.  Metadata, windows and options
 * are the first voice's.  The single-voice entries above are these with nv = 1, weight 1.0. */
typedef struct {
    const double *duration;
    const double *parameter[JBO_MAX_STREAM];
    const double *gv[JBO_MAX_STREAM];
} jbo_weights;
int jbo_duration_params_multi(const jbo_voice *const *vs, int nv, const double *w, const char *const *labels,
                              int n, double *mean_var);
int jbo_durations_multi(const jbo_voice *const *vs, int nv, const double *w, const char *const *labels, int n,
                        double speed, const double *times, uint32_t *dur_out);
int jbo_stream_params_multi(const jbo_voice *const *vs, int nv, const double *w, int stream,
                            const char *const *labels, int n, double *mean, double *var, double *msd);
int jbo_gv_params_multi(const jbo_voice *const *vs, int nv, const double *w, int stream,
                        const char *const *labels, int n, double *gv_mean, double *gv_var, uint8_t *gv_switch);
int jbo_synthesize_multi_ex(const jbo_voice *const *vs, int nv, const jbo_weights *w, const jbo_cond *c,
                            const char *const *lines, int n, double **pcm, size_t *n_samples, uint32_t **dur,
                            uint32_t *S, double **mcp, double **lf0, double **lpf, size_t *T);
/* state-level hot path: MLPG x3 + vocoder.  Returns malloc'd pcm. */
int jbo_paramgen_vocode(int fs, int fperiod, double alpha, double volume,
                        const jbo_stream st[3], int nstream, uint32_t S, const uint32_t *dur,
                        double **pcm, size_t *n_samples);
int jbo_paramgen_vocode_beta(int fs, int fperiod, double alpha, double beta, double volume,
                             const jbo_stream st[3], int nstream, uint32_t S, const uint32_t *dur,
                             double **pcm, size_t *n_samples);
void jbo_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
