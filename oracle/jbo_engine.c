/*
 * oracle/jbo_engine.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Front half + orchestration, restating:
 *   src/model/mod.rs:80-156          (Models::duration/stream/gv/model_stream)
 *   src/duration.rs:28-131           (DurationEstimator)
 *   src/label.rs:35-113              (time-aligned label lines)
 *   src/engine.rs:84-125,294-366     (Condition::load_model, Engine::synthesize/generator)
 *   src/model/stream_parameter.rs:29-37 (additional half tone)
 *   src/model/voice_set.rs:80-95     (VoiceSet::weighted: first*w0, then += w_i*param_i in voice order)
 *   src/model/voice/model.rs:111-129 (ModelParameter::{mul, mul_add_assign})
 * The single-voice entries are the multi-voice ones with one voice and weight 1.0 (`weighted`
 * multiplies by 1.0, which is exact), so the reference's goldens pin the shared code.  The blend of
 * two DIFFERENT voices itself is parity-unpinned: its only golden (`bonsai_multi`, src/lib.rs:77-91)
 * needs the tohoku-f01 files, which are not in the reference tree.
 */
#include "jbo_internal.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void jbo_cond_default(jbo_cond *c)
{
    c->speed = 1.0;
    c->volume = 1.0;
    c->beta = 0.0;
    c->additional_half_tone = 0.0;
    c->phoneme_alignment = 0;
    for (int i = 0; i < JBO_MAX_STREAM; i++) {
        c->msd_threshold[i] = 0.5; /* src/engine.rs:92 */
        c->gv_weight[i] = 1.0;     /* src/engine.rs:93 */
    }
}

void jbo_free(void *p) { free(p); }

/* VoiceSet::weighted (src/model/voice_set.rs:80-95) over one pdf row of every voice:
 * result = first.mul(w0) (model.rs:121-129), then result.mul_add_assign(w_i, param_i) (:111-119)
 * in voice order; plain multiply then add (no FMA: the file is built with -ffp-contract=off). */
static void weighted_rows(const float *const *rows, const double *w, int nv, int len, double *out)
{
    for (int k = 0; k < len; k++)
        out[k] = (double)rows[0][k] * w[0];
    for (int v = 1; v < nv; v++)
        for (int k = 0; k < len; k++)
            out[k] += w[v] * (double)rows[v][k];
}

static const double ONE_WEIGHT[1] = {1.0};

/* Models::duration (src/model/mod.rs:80-92): state index 2, nstate MeanVari per label */
int jbo_duration_params_multi(const jbo_voice *const *vs, int nv, const double *w, const char *const *labels,
                              int n, double *mean_var)
{
    int ns = vs[0]->nstate;
    const float *rows[JBO_MAX_VOICES];
    double buf[64];
    if (nv < 1 || nv > JBO_MAX_VOICES || 2 * ns > 64)
        return -1;
    for (int i = 0; i < n; i++) {
        for (int v = 0; v < nv; v++) {
            int tp, pi;
            if (jbo_model_get_index(&vs[v]->dur, 2, labels[i], &tp, &pi) || tp < 0)
                return -1;
            rows[v] = vs[v]->dur.pdf[tp] + (size_t)(pi - 1) * (size_t)vs[v]->dur.pdf_len;
        }
        weighted_rows(rows, w, nv, 2 * ns, buf);
        for (int s = 0; s < ns; s++) {
            /* ModelParameter::from_linear (voice/model.rs:99-109) */
            mean_var[2 * (i * ns + s)] = buf[s];
            mean_var[2 * (i * ns + s) + 1] = buf[s + ns];
        }
    }
    return 0;
}

int jbo_duration_params(const jbo_voice *v, const char *const *labels, int n, double *mean_var)
{
    return jbo_duration_params_multi(&v, 1, ONE_WEIGHT, labels, n, mean_var);
}

/* Rust f64::round = half away from zero = C round() */
static void estimate_duration(const double *mv, size_t n, double rho, uint32_t *d)
{
    for (size_t i = 0; i < n; i++) {
        double r = round(mv[2 * i] + rho * mv[2 * i + 1]);
        if (!(r > 1.0))
            r = 1.0;
        d[i] = (uint32_t)r;
    }
}

/* estimate_duration_with_frame_length (src/duration.rs:75-131) */
static void estimate_with_frame_length(const double *mv, size_t n, double frame_length, uint32_t *d)
{
    double tl = round(frame_length);
    if (!(tl > 1.0))
        tl = 1.0;
    size_t target = (size_t)tl;
    if (target <= n) {
        for (size_t i = 0; i < n; i++)
            d[i] = 1;
        return;
    }
    double mean = 0.0, vari = 0.0;
    for (size_t i = 0; i < n; i++) {
        mean = mean + mv[2 * i];
        vari = vari + mv[2 * i + 1];
    }
    double rho = ((double)target - mean) / vari;
    estimate_duration(mv, n, rho, d);
    if (n == 0)
        return;
    size_t sum = 0;
    for (size_t i = 0; i < n; i++)
        sum += d[i];
    while (target != sum) {
        size_t best = (size_t)-1;
        double bc = 0.0;
        if (target > sum) {
            for (size_t i = 0; i < n; i++) {
                double c = fabs(rho - ((double)(d[i] + 1) - mv[2 * i]) / mv[2 * i + 1]);
                /* min_by returns the first of equal minima; total_cmp order */
                if (best == (size_t)-1 || c < bc) {
                    best = i;
                    bc = c;
                }
            }
            d[best] += 1;
            sum += 1;
        } else {
            for (size_t i = 0; i < n; i++) {
                if (d[i] <= 1)
                    continue;
                double c = fabs(rho - ((double)(d[i] - 1) - mv[2 * i]) / mv[2 * i + 1]);
                if (best == (size_t)-1 || c < bc) {
                    best = i;
                    bc = c;
                }
            }
            d[best] -= 1;
            sum -= 1;
        }
    }
}

int jbo_durations_multi(const jbo_voice *const *vs, int nv, const double *w, const char *const *labels, int n,
                        double speed, const double *times, uint32_t *dur)
{
    size_t ns = (size_t)vs[0]->nstate, S = (size_t)n * ns;
    if (n == 0)
        return 0;
    double *mv = (double *)malloc(sizeof(double) * 2 * S);
    if (jbo_duration_params_multi(vs, nv, w, labels, n, mv)) {
        free(mv);
        return -1;
    }
    if (!times) {
        /* create (duration.rs:28-38) */
        estimate_duration(mv, S, 0.0, dur);
        if (speed != 1.0) {
            size_t length = 0;
            for (size_t i = 0; i < S; i++)
                length += dur[i];
            estimate_with_frame_length(mv, S, (double)length / speed, dur);
        }
    } else {
        /* create_with_alignment (duration.rs:41-65) */
        size_t frame_count = 0, next_state = 0, state = 0, nd = 0;
        for (int i = 0; i < n; i++) {
            double end_frame = times[2 * i + 1];
            if (end_frame >= 0.0) {
                size_t cnt = state + ns - next_state;
                estimate_with_frame_length(mv + 2 * next_state, cnt,
                                           end_frame - (double)frame_count, dur + nd);
                for (size_t k = 0; k < cnt; k++)
                    frame_count += dur[nd + k];
                nd += cnt;
                next_state = state + ns;
            }
            state += ns;
        }
        /* states after the last aligned label get no duration in the reference
         * (duration vector is shorter); signal by zero */
        for (size_t k = nd; k < S; k++)
            dur[k] = 0;
    }
    free(mv);
    return 0;
}

int jbo_durations(const jbo_voice *v, const char *const *labels, int n, double speed,
                  const double *times, uint32_t *dur)
{
    return jbo_durations_multi(&v, 1, ONE_WEIGHT, labels, n, speed, times, dur);
}

/* Models::stream (src/model/mod.rs:98-118) */
int jbo_stream_params_multi(const jbo_voice *const *vs, int nv, const double *w, int si,
                            const char *const *labels, int n, double *mean, double *var, double *msd)
{
    const jbo_vstream *st0 = &vs[0]->st[si];
    int ns = vs[0]->nstate, WL = st0->L * st0->W, plen = 2 * WL + (st0->is_msd ? 1 : 0);
    const float *rows[JBO_MAX_VOICES];
    if (nv < 1 || nv > JBO_MAX_VOICES)
        return -1;
    double *buf = (double *)malloc(sizeof(double) * (size_t)plen);
    for (int i = 0; i < n; i++)
        for (int s = 0; s < ns; s++) {
            for (int v = 0; v < nv; v++) {
                const jbo_vstream *st = &vs[v]->st[si];
                int tp, pi;
                if (jbo_model_get_index(&st->model, 2 + s, labels[i], &tp, &pi) || tp < 0) {
                    free(buf);
                    return -1;
                }
                rows[v] = st->model.pdf[tp] + (size_t)(pi - 1) * (size_t)st->model.pdf_len;
            }
            /* mean, variance and (MSD streams) the msd weight blend alike (model.rs:111-129) */
            weighted_rows(rows, w, nv, plen, buf);
            size_t row = (size_t)(i * ns + s);
            for (int k = 0; k < WL; k++) {
                mean[row * (size_t)WL + (size_t)k] = buf[k];
                var[row * (size_t)WL + (size_t)k] = buf[k + WL];
            }
            /* msd.unwrap_or(f64::MAX) (mod.rs:113) */
            msd[row] = st0->is_msd ? buf[2 * WL] : DBL_MAX;
        }
    free(buf);
    return 0;
}

int jbo_stream_params(const jbo_voice *v, int si, const char *const *labels, int n, double *mean,
                      double *var, double *msd)
{
    return jbo_stream_params_multi(&v, 1, ONE_WEIGHT, si, labels, n, mean, var, msd);
}

/* Models::gv (src/model/mod.rs:119-146) */
int jbo_gv_params_multi(const jbo_voice *const *vs, int nv, const double *w, int si,
                        const char *const *labels, int n, double *gv_mean, double *gv_var, uint8_t *gv_switch)
{
    const jbo_voice *v0 = vs[0];
    const jbo_vstream *st0 = &v0->st[si];
    if (!st0->use_gv || n == 0)
        return 1;
    const float *rows[JBO_MAX_VOICES];
    if (nv < 1 || nv > JBO_MAX_VOICES)
        return -1;
    for (int v = 0; v < nv; v++) {
        const jbo_vstream *st = &vs[v]->st[si];
        int tp, pi;
        if (jbo_model_get_index(&st->gv, 2, labels[0], &tp, &pi) || tp < 0)
            return -1;
        rows[v] = st->gv.pdf[tp] + (size_t)(pi - 1) * (size_t)st->gv.pdf_len;
    }
    double *buf = (double *)malloc(sizeof(double) * 2 * (size_t)st0->L);
    weighted_rows(rows, w, nv, 2 * st0->L, buf);
    for (int k = 0; k < st0->L; k++) {
        gv_mean[k] = buf[k];
        gv_var[k] = buf[k + st0->L];
    }
    free(buf);
    /* gv_off_context of the first voice's global metadata (mod.rs:136-143, voice_set.rs:64-66) */
    for (int i = 0; i < n; i++) {
        uint8_t sw = !jbo_gv_off(v0, labels[i]);
        for (int s = 0; s < v0->nstate; s++)
            gv_switch[i * v0->nstate + s] = sw;
    }
    return 0;
}

int jbo_gv_params(const jbo_voice *v, int si, const char *const *labels, int n, double *gv_mean,
                  double *gv_var, uint8_t *gv_switch)
{
    return jbo_gv_params_multi(&v, 1, ONE_WEIGHT, si, labels, n, gv_mean, gv_var, gv_switch);
}

/* Labels::load_from_strings + Labels::new (src/label.rs:35-113) */
int jbo_parse_label_lines(int fs, int fperiod, const char *const *lines, int n,
                          const char **label_out, double *times)
{
    double rate = (double)fs / ((double)fperiod * 1e+7);
    int m = 0;
    for (int i = 0; i < n; i++) {
        const char *line = lines[i];
        const char *sp1 = strchr(line, ' ');
        if (sp1) {
            const char *sp2 = strchr(sp1 + 1, ' ');
            if (!sp2)
                return -1;
            double start = strtod(line, NULL), end = strtod(sp1 + 1, NULL);
            times[2 * m] = start * rate;
            times[2 * m + 1] = end * rate;
            label_out[m] = sp2 + 1;
            m++;
        } else if (line[0] == 0) {
            continue;
        } else {
            times[2 * m] = -1.0;
            times[2 * m + 1] = -1.0;
            label_out[m] = line;
            m++;
        }
    }
    for (int i = 0; i < m; i++) {
        if (i + 1 < m) {
            if (times[2 * i + 1] < 0.0 && times[2 * (i + 1)] >= 0.0)
                times[2 * i + 1] = times[2 * (i + 1)];
            else if (times[2 * i + 1] >= 0.0 && times[2 * (i + 1)] < 0.0)
                times[2 * (i + 1)] = times[2 * i + 1];
        }
        if (times[2 * i] < 0.0)
            times[2 * i] = -1.0;
        if (times[2 * i + 1] < 0.0)
            times[2 * i + 1] = -1.0;
    }
    return m;
}

int jbo_paramgen_vocode(int fs, int fperiod, double alpha, double volume, const jbo_stream st[3],
                        int nstream, uint32_t S, const uint32_t *dur, double **pcm_out,
                        size_t *n_samples)
{
    return jbo_paramgen_vocode_beta(fs, fperiod, alpha, 0.0, volume, st, nstream, S, dur, pcm_out,
                                    n_samples);
}

int jbo_paramgen_vocode_beta(int fs, int fperiod, double alpha, double beta, double volume,
                             const jbo_stream st[3], int nstream, uint32_t S, const uint32_t *dur,
                             double **pcm_out, size_t *n_samples)
{
    size_t T = 0;
    for (uint32_t s = 0; s < S; s++)
        T += dur[s];
    *n_samples = T * (size_t)fperiod;
    *pcm_out = NULL;
    if (T == 0)
        return 0;
    size_t Lm = st[0].vector_length, Ll = nstream > 2 ? st[2].vector_length : 0;
    double *mcp = (double *)malloc(sizeof(double) * T * Lm);
    double *lf0 = (double *)malloc(sizeof(double) * T);
    double *lpf = Ll ? (double *)malloc(sizeof(double) * T * Ll) : NULL;
    jbo_mlpg(&st[0], S, dur, mcp);
    jbo_mlpg(&st[1], S, dur, lf0);
    if (Ll)
        jbo_mlpg(&st[2], S, dur, lpf);
    double *pcm = (double *)malloc(sizeof(double) * T * (size_t)fperiod);
    int r = jbo_vocoder_beta(fs, fperiod, alpha, beta, volume, (int)Lm, (int)Ll, T, lf0, mcp, lpf, pcm,
                             NULL, NULL);
    free(mcp);
    free(lf0);
    free(lpf);
    if (r) {
        free(pcm);
        return r;
    }
    *pcm_out = pcm;
    return 0;
}

/* Engine::generator + generate_all (src/engine.rs:301-366) */
int jbo_synthesize_multi_ex(const jbo_voice *const *voices, int nv, const jbo_weights *w, const jbo_cond *c,
                            const char *const *lines, int n, double **pcm_out, size_t *n_samples,
                            uint32_t **dur_out, uint32_t *S_out, double **mcp_out, double **lf0_out,
                            double **lpf_out, size_t *T_out)
{
    /* metadata, windows, options: the first voice's (voice_set.rs:64-77; engine.rs:84-125) */
    const jbo_voice *v = voices[0];
    *pcm_out = NULL;
    *n_samples = 0;
    if (dur_out)
        *dur_out = NULL;
    if (S_out)
        *S_out = 0;
    if (mcp_out)
        *mcp_out = NULL;
    if (lf0_out)
        *lf0_out = NULL;
    if (lpf_out)
        *lpf_out = NULL;
    if (T_out)
        *T_out = 0;
    if (v->stage != 0)
        return -2; /* Stage::NonZero not restated */
    const char **labels = (const char **)malloc(sizeof(char *) * (size_t)(n ? n : 1));
    double *times = (double *)malloc(sizeof(double) * 2 * (size_t)(n ? n : 1));
    int m = jbo_parse_label_lines(v->fs, v->fperiod, lines, n, labels, times);
    if (m < 0) {
        free(labels);
        free(times);
        return -1;
    }
    if (m == 0) {
        free(labels);
        free(times);
        return 0;
    }
    uint32_t S = (uint32_t)(m * v->nstate);
    uint32_t *dur = (uint32_t *)calloc(S, sizeof(uint32_t));
    int rc = jbo_durations_multi(voices, nv, w->duration, labels, m, c->speed, c->phoneme_alignment ? times : NULL, dur);
    jbo_stream st[3];
    double *mean[3] = {0}, *var[3] = {0}, *msd[3] = {0}, *gm[3] = {0}, *gv[3] = {0};
    uint8_t *gs[3] = {0};
    memset(st, 0, sizeof st);
    for (int i = 0; i < v->nstream && rc == 0; i++) {
        const jbo_vstream *vs = &v->st[i];
        size_t WL = (size_t)(vs->L * vs->W);
        mean[i] = (double *)malloc(sizeof(double) * S * WL);
        var[i] = (double *)malloc(sizeof(double) * S * WL);
        msd[i] = (double *)malloc(sizeof(double) * S);
        rc = jbo_stream_params_multi(voices, nv, w->parameter[i], i, labels, m, mean[i], var[i], msd[i]);
        st[i].vector_length = (uint32_t)vs->L;
        st[i].num_windows = (uint32_t)vs->W;
        st[i].is_msd = (uint32_t)vs->is_msd;
        st[i].use_gv = (uint32_t)vs->use_gv;
        st[i].win_width = vs->win_width;
        st[i].win_coef = vs->win_coef;
        st[i].mean = mean[i];
        st[i].var = var[i];
        st[i].msd = msd[i];
        st[i].gv_weight = c->gv_weight[i];
        st[i].msd_threshold = c->msd_threshold[i];
        if (vs->use_gv && rc == 0) {
            gm[i] = (double *)malloc(sizeof(double) * (size_t)vs->L);
            gv[i] = (double *)malloc(sizeof(double) * (size_t)vs->L);
            gs[i] = (uint8_t *)malloc(S);
            if (jbo_gv_params_multi(voices, nv, w->gv[i], i, labels, m, gm[i], gv[i], gs[i]) == 0) {
                st[i].gv_mean = gm[i];
                st[i].gv_var = gv[i];
                st[i].gv_switch = gs[i];
            } else {
                rc = -1;
            }
        }
    }
    /* apply_additional_half_tone (stream_parameter.rs:29-37) on LF0 static mean */
    if (rc == 0 && c->additional_half_tone != 0.0 && v->nstream > 1) {
        const double HALF_TONE = 0.05776226504666211, MAX_LF0 = 9.903487552536127,
                     MIN_LF0 = 2.995732273553991;
        size_t WL = (size_t)(v->st[1].L * v->st[1].W);
        for (uint32_t s = 0; s < S; s++) {
            double x = mean[1][s * WL] + c->additional_half_tone * HALF_TONE;
            x = x < MIN_LF0 ? MIN_LF0 : (x > MAX_LF0 ? MAX_LF0 : x);
            mean[1][s * WL] = x;
        }
    }
    if (rc == 0) {
        size_t T = 0;
        for (uint32_t s = 0; s < S; s++)
            T += dur[s];
        size_t Lm = (size_t)v->st[0].L, Ll = v->nstream > 2 ? (size_t)v->st[2].L : 0;
        double *mcp = (double *)malloc(sizeof(double) * (T ? T : 1) * Lm);
        double *lf0 = (double *)malloc(sizeof(double) * (T ? T : 1));
        double *lpf = (double *)malloc(sizeof(double) * (T ? T : 1) * (Ll ? Ll : 1));
        jbo_mlpg(&st[0], S, dur, mcp);
        jbo_mlpg(&st[1], S, dur, lf0);
        if (Ll)
            jbo_mlpg(&st[2], S, dur, lpf);
        double *pcm = (double *)malloc(sizeof(double) * (T ? T : 1) * (size_t)v->fperiod);
        rc = jbo_vocoder_beta(v->fs, v->fperiod, v->alpha, c->beta, c->volume, (int)Lm, (int)Ll, T, lf0,
                              mcp, Ll ? lpf : NULL, pcm, NULL, NULL);
        *pcm_out = pcm;
        *n_samples = T * (size_t)v->fperiod;
        if (T_out)
            *T_out = T;
        if (mcp_out)
            *mcp_out = mcp;
        else
            free(mcp);
        if (lf0_out)
            *lf0_out = lf0;
        else
            free(lf0);
        if (lpf_out)
            *lpf_out = lpf;
        else
            free(lpf);
    }
    if (dur_out && rc == 0) {
        *dur_out = dur;
        if (S_out)
            *S_out = S;
    } else {
        free(dur);
    }
    for (int i = 0; i < 3; i++) {
        free(mean[i]);
        free(var[i]);
        free(msd[i]);
        free(gm[i]);
        free(gv[i]);
        free(gs[i]);
    }
    free(labels);
    free(times);
    return rc;
}

int jbo_synthesize_ex(const jbo_voice *v, const jbo_cond *c, const char *const *lines, int n,
                      double **pcm_out, size_t *n_samples, uint32_t **dur_out, uint32_t *S_out,
                      double **mcp_out, double **lf0_out, double **lpf_out, size_t *T_out)
{
    jbo_weights w;
    w.duration = ONE_WEIGHT;
    for (int i = 0; i < JBO_MAX_STREAM; i++)
        w.parameter[i] = w.gv[i] = ONE_WEIGHT;
    return jbo_synthesize_multi_ex(&v, 1, &w, c, lines, n, pcm_out, n_samples, dur_out, S_out, mcp_out, lf0_out,
                                   lpf_out, T_out);
}

int jbo_synthesize(const jbo_voice *v, const jbo_cond *c, const char *const *lines, int n,
                   double **pcm, size_t *n_samples)
{
    return jbo_synthesize_ex(v, c, lines, n, pcm, n_samples, NULL, NULL, NULL, NULL, NULL, NULL);
}
