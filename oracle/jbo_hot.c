/*
 * oracle/jbo_hot.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * `double` restatement, in the reference's operation order (non-FMA x86-64
 * expansion of `mul_add!`, src/vocoder/mlsa.rs:117-125), of the hot path:
 *   A1-A9  src/mlpg_adjust/{mod.rs,mask.rs,mlpg.rs}, src/model/mean_vari.rs:21-37,
 *          src/model/voice/window.rs:19-76
 *   V1-V9  src/vocoder/{mod.rs,mlsa.rs,excitation.rs,cepstrum.rs:139-149}, src/speech.rs
 *   X1     post-filter, beta > 0: src/vocoder/cepstrum.rs:23-37,153-186,
 *          src/vocoder/coefficients.rs:65-78 -- PARITY UNPINNED for this part: the reference's
 *          tests never set beta > 0, so nothing but analytic identities holds it
 *          (tests/test_oracle_golden.py::test_postfilter_pieces)
 * Compile with -ffp-contract=off so that no FMA is formed.
 */
#include "jbo_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================= */
/* A1: Mask::create (src/mlpg_adjust/mask.rs:20-28)                        */
size_t jbo_mask(const jbo_stream *st, uint32_t S, const uint32_t *dur, uint8_t *mask)
{
    size_t t = 0;
    for (uint32_t s = 0; s < S; s++) {
        uint8_t m = st->msd[s] > st->msd_threshold;
        for (uint32_t d = 0; d < dur[s]; d++)
            mask[t++] = m;
    }
    return t;
}

/* A2: Mask::boundary_distances (src/mlpg_adjust/mask.rs:51-82) */
void jbo_boundary_distances(const uint8_t *mask, size_t T, size_t *left, size_t *right)
{
    if (T == 0)
        return;
    for (size_t i = 0; i < T; i++)
        left[i] = right[i] = 0;
    size_t l = 0;
    for (size_t f = 0; f < T; f++) {
        if (mask[f])
            left[f] = f - l;
        else
            l = f + 1;
    }
    size_t r = T - 1;
    for (size_t f = T; f-- > 0;) {
        if (mask[f]) {
            right[f] = r - f;
        } else {
            if (f == 0)
                break;
            r = f - 1;
        }
    }
}

/* MeanVari::with_ivar (src/model/mean_vari.rs:21-31) */
static double with_ivar(double vari)
{
    if (fabs(vari) > 1e19)
        return 0.0;
    if (fabs(vari) < 1e-19)
        return 1e38;
    return 1.0 / vari;
}

typedef struct {
    size_t length; /* T' */
    int width;     /* band width = max_width*2+1 */
    int win_size;
    double *wuw;   /* [T'][width] */
    double *wum;   /* [T'] */
} mtx_t;

/* A4: MlpgMatrix::calc_wuw_and_wum (src/mlpg_adjust/mlpg.rs:25-70).
 * pm/pi: [W][T'] mean / ivar. */
static void calc_wuw_and_wum(mtx_t *m, const jbo_stream *st, const double *pm, const double *pi)
{
    size_t len = m->length;
    int width = m->width;
    for (size_t t = 0; t < len; t++) {
        double *row = m->wuw + t * (size_t)width;
        for (int j = 0; j < width; j++)
            row[j] = 0.0;
        m->wum[t] = 0.0;
        size_t coff = 0;
        for (uint32_t i = 0; i < st->num_windows; i++) {
            int ww = (int)st->win_width[i];
            const double *coef = st->win_coef + coff;
            coff += (size_t)ww;
            /* iter_rev(0): index ww-1 .. 0; position = index - ww/2 */
            for (int index = ww - 1; index >= 0; index--) {
                double c = coef[index];
                if (c == 0.0)
                    continue;
                long idx = (long)t - ((long)index - (long)(ww / 2));
                if (idx < 0 || idx >= (long)len)
                    continue;
                double wu = c * pi[(size_t)i * len + (size_t)idx];
                m->wum[t] += wu * pm[(size_t)i * len + (size_t)idx];
                /* iter_rev(index): inner index ww-1 .. index */
                for (int inner = ww - 1; inner >= index; inner--) {
                    double c2 = coef[inner];
                    if (c2 == 0.0)
                        continue;
                    int j = inner - index;
                    if (t + (size_t)j >= len)
                        break;
                    row[j] += wu * c2;
                }
            }
        }
    }
}

/* A5: ldl_factorization (src/mlpg_adjust/mlpg.rs:79-92) */
static void ldl(mtx_t *m)
{
    size_t T = m->length;
    int w = m->width;
    double *A = m->wuw;
#define W_(t, i) A[(size_t)(t) * (size_t)w + (size_t)(i)]
    for (size_t t = 0; t < T; t++) {
        int lim = (int)((size_t)w < t + 1 ? (size_t)w : t + 1);
        for (int i = 1; i < lim; i++)
            W_(t, 0) -= W_(t - (size_t)i, i) * W_(t - (size_t)i, i) * W_(t - (size_t)i, 0);
        for (int i = 1; i < w; i++) {
            int lim2 = (int)((size_t)(w - i) < t + 1 ? (size_t)(w - i) : t + 1);
            for (int j = 1; j < lim2; j++)
                W_(t, i) -= W_(t - (size_t)j, j) * W_(t - (size_t)j, i + j) * W_(t - (size_t)j, 0);
            W_(t, i) /= W_(t, 0);
        }
    }
}

/* A6: substitutions (src/mlpg_adjust/mlpg.rs:95-115) */
static void substitutions(const mtx_t *m, double *g, double *par)
{
    size_t T = m->length;
    int w = m->width;
    const double *A = m->wuw;
    for (size_t t = 0; t < T; t++) {
        g[t] = m->wum[t];
        int lim = (int)((size_t)w < t + 1 ? (size_t)w : t + 1);
        for (int i = 1; i < lim; i++)
            g[t] -= W_(t - (size_t)i, i) * g[t - (size_t)i];
    }
    for (size_t t = T; t-- > 0;) {
        par[t] = g[t] / W_(t, 0);
        int lim = (int)((size_t)w < T - t ? (size_t)w : T - t);
        for (int i = 1; i < lim; i++)
            par[t] -= W_(t, i) * par[t + (size_t)i];
    }
}

/* A8: MlpgGlobalVariance (src/mlpg_adjust/mlpg.rs:145-292) */
typedef struct {
    double *par;
    const uint8_t *sw;
    size_t gv_length;
    const mtx_t *mtx; /* un-factored */
} gv_t;

static void calc_gv(const gv_t *g, double *mean, double *vari)
{
    size_t T = g->mtx->length;
    double s = 0.0;
    for (size_t t = 0; t < T; t++)
        if (g->sw[t])
            s += g->par[t];
    double mu = s / (double)g->gv_length;
    double v = 0.0;
    for (size_t t = 0; t < T; t++)
        if (g->sw[t])
            v += (g->par[t] - mu) * (g->par[t] - mu);
    *mean = mu;
    *vari = v / (double)g->gv_length;
}

static void conv_gv(gv_t *g, double gv_mean)
{
    double mean, vari;
    calc_gv(g, &mean, &vari);
    double ratio = sqrt(gv_mean / vari);
    size_t T = g->mtx->length;
    for (size_t t = 0; t < T; t++)
        if (g->sw[t])
            g->par[t] = ratio * (g->par[t] - mean) + mean;
}

static double calc_hmmobj_derivative(const gv_t *gg, double *g)
{
    const mtx_t *m = gg->mtx;
    size_t T = m->length;
    int w = m->width;
    const double *A = m->wuw;
    const double *par = gg->par;
    for (size_t t = 0; t < T; t++) {
        g[t] = W_(t, 0) * par[t];
        for (int i = 1; i < w; i++) {
            if (t + (size_t)i < T)
                g[t] += W_(t, i) * par[t + (size_t)i];
            if (t + 1 > (size_t)i)
                g[t] += W_(t - (size_t)i, i) * par[t - (size_t)i];
        }
    }
    double wgt = 1.0 / (double)((size_t)m->win_size * T);
    double hmmobj = 0.0;
    for (size_t t = 0; t < T; t++)
        hmmobj += 1.0 * wgt * par[t] * (m->wum[t] - 0.5 * g[t]);
    return hmmobj;
}

static void next_step(gv_t *gg, const double *g, double step, double mean, double vari,
                      double gv_mean, double gv_vari)
{
    const mtx_t *m = gg->mtx;
    size_t length = m->length;
    int w_ = m->width;
    const double *A = m->wuw;
    (void)w_;
    int w = m->width;
    double wgt = 1.0 / (double)((size_t)m->win_size * length);
    double dv = -2.0 * gv_vari * (vari - gv_mean) / (double)length;
    double *par = gg->par;
    for (size_t t = 0; t < length; t++) {
        double h = -1.0 * wgt * W_(t, 0) -
                   1.0 * 2.0 / (double)(length * length) *
                       ((double)(length - 1) * gv_vari * (vari - gv_mean) +
                        2.0 * gv_vari * (par[t] - mean) * (par[t] - mean));
        double next_g;
        if (gg->sw[t])
            next_g = 1.0 / h * (1.0 * wgt * (-g[t] + m->wum[t]) + 1.0 * dv * (par[t] - mean));
        else
            next_g = 1.0 / h * (1.0 * wgt * (-g[t] + m->wum[t]));
        par[t] += step * next_g;
    }
}
#undef W_

static void parmgen(gv_t *g, double gv_mean, double gv_vari)
{
    const int GV_MAX_ITERATION = 5;
    const double STEPINIT = 0.1, STEPDEC = 0.5, STEPINC = 1.2;
    if (g->gv_length == 0)
        return;
    size_t T = g->mtx->length;
    double *gr = (double *)malloc(sizeof(double) * (T ? T : 1));
    double step = STEPINIT, prev = 0.0;
    conv_gv(g, gv_mean);
    for (int i = 1; i <= GV_MAX_ITERATION; i++) {
        double mean, vari;
        calc_gv(g, &mean, &vari);
        double gvobj = -0.5 * 1.0 * vari * gv_vari * (vari - 2.0 * gv_mean);
        double hmmobj = calc_hmmobj_derivative(g, gr);
        double obj = -(hmmobj + gvobj);
        if (i > 1) {
            if (obj > prev)
                step *= STEPDEC;
            else if (obj < prev)
                step *= STEPINC;
        }
        next_step(g, gr, step, mean, vari, gv_mean, gv_vari);
        prev = obj;
    }
    free(gr);
}

/* A3..A9: MlpgAdjust::create (src/mlpg_adjust/mod.rs:51-95) */
int jbo_mlpg(const jbo_stream *st, uint32_t S, const uint32_t *dur, double *pars)
{
    size_t T = 0;
    for (uint32_t s = 0; s < S; s++)
        T += dur[s];
    if (T == 0)
        return 0;
    uint32_t L = st->vector_length, W = st->num_windows;
    uint8_t *mask = (uint8_t *)malloc(T);
    jbo_mask(st, S, dur, mask);
    size_t *left = (size_t *)malloc(sizeof(size_t) * T), *right = (size_t *)malloc(sizeof(size_t) * T);
    jbo_boundary_distances(mask, T, left, right);
    size_t Tv = 0;
    for (size_t t = 0; t < T; t++)
        Tv += mask[t];
    /* Windows::max_width (src/model/voice/window.rs:19-21) */
    uint32_t maxw = 0;
    for (uint32_t w = 0; w < W; w++)
        if (st->win_width[w] > maxw)
            maxw = st->win_width[w];
    int width = (int)(maxw / 2) * 2 + 1;

    /* gv_switch expanded by duration and compacted by mask (mlpg.rs:128-133) */
    uint8_t *sw = (uint8_t *)malloc(Tv ? Tv : 1);
    int has_gv = st->use_gv && st->gv_mean && st->gv_switch;
    size_t gv_len = 0;
    if (has_gv) {
        size_t t = 0, k = 0;
        for (uint32_t s = 0; s < S; s++)
            for (uint32_t d = 0; d < dur[s]; d++, t++)
                if (mask[t]) {
                    sw[k] = st->gv_switch[s];
                    gv_len += sw[k];
                    k++;
                }
    }

    size_t n1 = Tv ? Tv : 1;
    double *pm = (double *)malloc(sizeof(double) * W * n1);
    double *pi = (double *)malloc(sizeof(double) * W * n1);
    mtx_t m0, m1;
    m0.length = m1.length = Tv;
    m0.width = m1.width = width;
    m0.win_size = m1.win_size = (int)W;
    m0.wuw = (double *)malloc(sizeof(double) * n1 * (size_t)width);
    m0.wum = (double *)malloc(sizeof(double) * n1);
    m1.wuw = (double *)malloc(sizeof(double) * n1 * (size_t)width);
    m1.wum = (double *)malloc(sizeof(double) * n1);
    double *g = (double *)malloc(sizeof(double) * n1);
    double *par = (double *)malloc(sizeof(double) * n1);

    for (uint32_t vi = 0; vi < L; vi++) {
        for (uint32_t w = 0; w < W; w++) {
            uint32_t mi = L * w + vi;
            size_t lw = st->win_width[w] / 2;
            size_t rw = st->win_width[w] - lw - 1;
            size_t t = 0, k = 0;
            for (uint32_t s = 0; s < S; s++) {
                double mean = st->mean[(size_t)s * W * L + mi];
                double ivar = with_ivar(st->var[(size_t)s * W * L + mi]);
                for (uint32_t d = 0; d < dur[s]; d++, t++) {
                    double iv = ivar;
                    if ((left[t] < lw || right[t] < rw) && w != 0)
                        iv = 0.0;
                    if (mask[t]) {
                        pm[(size_t)w * Tv + k] = mean;
                        pi[(size_t)w * Tv + k] = iv;
                        k++;
                    }
                }
            }
        }
        if (Tv > 0) {
            calc_wuw_and_wum(&m0, st, pm, pi);
            /* MlpgMatrix::par (mlpg.rs:118-142) */
            memcpy(m1.wuw, m0.wuw, sizeof(double) * Tv * (size_t)width);
            memcpy(m1.wum, m0.wum, sizeof(double) * Tv);
            ldl(&m1);
            substitutions(&m1, g, par);
            if (has_gv) {
                gv_t gv = {par, sw, gv_len, &m0};
                parmgen(&gv, st->gv_mean[vi] * st->gv_weight, st->gv_var[vi]);
            }
        }
        /* Mask::fill with NODATA (mask.rs:34-49, mod.rs:89-91) */
        size_t k = 0;
        for (size_t t = 0; t < T; t++)
            pars[t * L + vi] = mask[t] ? par[k++] : JBO_NODATA;
    }
    free(mask);
    free(left);
    free(right);
    free(sw);
    free(pm);
    free(pi);
    free(m0.wuw);
    free(m0.wum);
    free(m1.wuw);
    free(m1.wum);
    free(g);
    free(par);
    return 0;
}

/* ======================================================================= */
/* V4: Random (src/vocoder/excitation.rs:177-237)                          */
typedef struct {
    double queue[64];
    double s[32];
    size_t used;
    uint64_t next;
} rnd_t;

static double rnd(uint64_t *next)
{
    *next = *next * 1103515245ull + 12345ull;
    uint64_t r = (*next / 65536ull) % 32768ull;
    return (double)r / 32767.0;
}

static void fill_queue(rnd_t *r)
{
    int i = 0;
    while (i < 32) {
        double r1 = 2.0 * rnd(&r->next) - 1.0;
        double r2 = 2.0 * rnd(&r->next) - 1.0;
        double s = r1 * r1 + r2 * r2;
        if (0.0 < s && s < 1.0) {
            r->queue[2 * i] = r1;
            r->queue[2 * i + 1] = r2;
            r->s[i] = s;
            i++;
        }
    }
    for (i = 0; i < 32; i++) {
        double m = sqrt(-2.0 * log(r->s[i]) / r->s[i]);
        r->queue[2 * i] *= m;
        r->queue[2 * i + 1] *= m;
    }
}

static double nrandom(rnd_t *r)
{
    if (r->used >= 64) {
        fill_queue(r);
        r->used = 0;
    }
    return r->queue[r->used++];
}

void jbo_noise(double *out, size_t n)
{
    rnd_t r;
    r.used = 64;
    r.next = 1;
    for (size_t i = 0; i < n; i++)
        out[i] = nrandom(&r);
}

/* V3: Excitation + RingBuffer (src/vocoder/excitation.rs:1-152) */
typedef struct {
    double pitch_of_curr_point, pitch_counter, pitch_inc_per_point;
    double *ring;
    size_t nring, index;
    rnd_t random;
} exc_t;

static void exc_start(exc_t *e, double pitch, size_t fperiod)
{
    if (e->pitch_of_curr_point != 0.0 && pitch != 0.0) {
        e->pitch_inc_per_point = (pitch - e->pitch_of_curr_point) / (double)fperiod;
    } else {
        e->pitch_inc_per_point = 0.0;
        e->pitch_of_curr_point = pitch;
        e->pitch_counter = pitch;
    }
}

static double exc_get(exc_t *e, const double *lpf, double *pulse_out)
{
    *pulse_out = 0.0;
    if (e->nring > 0) {
        double noise = nrandom(&e->random);
        size_t anti = (e->index + (e->nring - 1) / 2) % e->nring;
        if (e->pitch_of_curr_point == 0.0) {
            e->ring[anti] += noise;
        } else {
            e->pitch_counter += 1.0;
            double pulse;
            if (e->pitch_counter >= e->pitch_of_curr_point) {
                e->pitch_counter -= e->pitch_of_curr_point;
                pulse = sqrt(e->pitch_of_curr_point);
            } else {
                pulse = 0.0;
            }
            *pulse_out = pulse;
            /* voiced_frame (excitation.rs:48-64) */
            e->ring[anti] += noise;
            double c = pulse - noise;
            size_t nright = e->nring - e->index;
            for (size_t i = 0; i < nright; i++)
                e->ring[e->index + i] += c * lpf[i];
            for (size_t i = 0; i < e->index; i++)
                e->ring[i] += c * lpf[nright + i];
            e->pitch_of_curr_point += e->pitch_inc_per_point;
        }
        double x = e->ring[e->index];
        e->ring[e->index] = 0.0;
        e->index++;
        if (e->index >= e->nring)
            e->index = 0;
        return x;
    } else if (e->pitch_of_curr_point == 0.0) {
        return nrandom(&e->random);
    } else {
        e->pitch_counter += 1.0;
        double x;
        if (e->pitch_counter >= e->pitch_of_curr_point) {
            e->pitch_counter -= e->pitch_of_curr_point;
            x = sqrt(e->pitch_of_curr_point);
        } else {
            x = 0.0;
        }
        *pulse_out = x;
        e->pitch_of_curr_point += e->pitch_inc_per_point;
        return x;
    }
}

/* V7: fir (src/vocoder/mlsa.rs:127-163), non-FMA expansion. d has n taps. */
static double fir(double *d, size_t n, double x, double alpha, const double *coefficients)
{
    double a = alpha;
    double aa = a * a;
    double aaaa = aa * aa;
    double iaa = 1.0 - aa;
    double rem = -a * x + d[1];
    d[0] = a * x;
    d[1] = iaa * x + a * d[1];
    double y0 = 0.0, y1 = 0.0;
    const double *c = coefficients + 2;
    double *dd = d + 2;
    size_t m = n - 2, nch = m / 4;
    for (size_t k = 0; k < nch; k++, c += 4, dd += 4) {
        double o0 = dd[0], o1 = dd[1], o2 = dd[2], o3 = dd[3];
        double n0 = iaa * rem + a * o0;
        double n1 = iaa * (-a * rem + o0) + a * o1;
        double n2 = iaa * (aa * rem + (-a * o0 + o1)) + a * o2;
        double n3 = iaa * (aa * (-a * rem + o0) + (-a * o1 + o2)) + a * o3;
        double nr = aaaa * rem + (aa * (-a * o0 + o1) + (-a * o2 + o3));
        dd[0] = n0;
        dd[1] = n1;
        dd[2] = n2;
        dd[3] = n3;
        rem = nr;
        y0 += c[0] * dd[0] + c[2] * dd[2];
        y1 += c[1] * dd[1] + c[3] * dd[3];
    }
    for (size_t k = nch * 4; k < m; k++, c++, dd++) {
        double o = *dd;
        *dd = iaa * rem + a * o;
        rem = -a * rem + o;
        y0 += *c * *dd;
    }
    return y0 + y1;
}

/* PPADE for N=6 (src/vocoder/mlsa.rs:31) */
static const double PPADE[6] = {1.00000000000, 0.49993910000, 0.11070980000,
                                0.01369984000, 0.00095648530, 0.00003041721};

typedef struct {
    double d11[6], d12[6], d22[6];
    double *d21[6];
} mlsa_t;

/* V6: df1 (mlsa.rs:54-66) */
static void df1(mlsa_t *f, double *x, double alpha, const double *c)
{
    double aa = 1.0 - alpha * alpha;
    double out = 0.0;
    for (int i = 5; i >= 1; i--) {
        f->d11[i] = aa * f->d12[i - 1] + alpha * f->d11[i];
        f->d12[i] = f->d11[i] * c[1];
        double v = f->d12[i] * PPADE[i];
        *x += (i & 1) ? v : -v;
        out += v;
    }
    f->d12[0] = *x;
    *x += out;
}

/* V7: df2 (mlsa.rs:69-79) */
static void df2(mlsa_t *f, double *x, double alpha, const double *c, size_t nmcp)
{
    double out = 0.0;
    for (int i = 5; i >= 1; i--) {
        f->d22[i] = fir(f->d21[i - 1], nmcp, f->d22[i - 1], alpha, c);
        double v = f->d22[i] * PPADE[i];
        *x += (i & 1) ? v : -v;
        out += v;
    }
    f->d22[0] = *x;
    *x += out;
}

/* mc2b (src/vocoder/cepstrum.rs:139-149) */
static void mc2b(const double *mc, double *b, size_t n, double alpha)
{
    for (size_t i = 0; i < n; i++)
        b[i] = mc[i];
    if (alpha != 0.0) {
        size_t last = n - 1;
        b[last] = mc[last];
        for (size_t i = last; i-- > 0;)
            b[i] = mc[i] - alpha * b[i + 1];
    }
}

/* b2mc (src/vocoder/coefficients.rs:65-73) */
static void b2mc(const double *b, double *mc, size_t n, double alpha)
{
    size_t last = n - 1;
    mc[last] = b[last];
    for (size_t i = last; i-- > 0;)
        mc[i] = b[i] + alpha * b[i + 1];
}

/* X1: freqt (src/vocoder/cepstrum.rs:153-173).  The reference feeds the input coefficients in
 * ASCENDING index order (self[i], i = 0..len), where hts_engine's HTS_freqt walks c1[m1]..c1[0];
 * this restatement follows the reference as written.  out[m2+1], scratch f[m2+1]. */
void jbo_freqt(const double *c1, size_t n1, double *out, size_t m2, double alpha)
{
    const double aa = 1.0 - alpha * alpha;
    const size_t n2 = m2 + 1;
    double *f = (double *)calloc(n2, sizeof(double));
    for (size_t j = 0; j < n2; j++)
        out[j] = 0.0;
    for (size_t i = 0; i < n1; i++) {
        f[0] = out[0];
        out[0] = c1[i] + alpha * out[0];
        if (1 <= m2) {
            f[1] = out[1];
            out[1] = aa * f[0] + alpha * out[1];
        }
        for (size_t j = 2; j < n2; j++) {
            f[j] = out[j];
            out[j] = f[j - 1] + alpha * (out[j] - out[j - 1]);
        }
    }
    free(f);
}

/* X1: c2ir (src/vocoder/cepstrum.rs:175-186): impulse response of exp(C(z)), len samples. */
void jbo_c2ir(const double *c, size_t nc, double *ir, size_t len)
{
    ir[0] = exp(c[0]);
    for (size_t n = 1; n < len; n++) {
        double d = 0.0;
        size_t kend = nc < n + 1 ? nc : n + 1;
        for (size_t k = 1; k < kend; k++)
            d += (double)k * c[k] * ir[n - k];
        ir[n] = d / (double)n;
    }
}

#define JBO_IRLENG 576 /* coefficients.rs:76 */

/* X1: b2en (src/vocoder/coefficients.rs:75-78) */
double jbo_b2en(const double *b, size_t n, double alpha)
{
    double *mc = (double *)malloc(sizeof(double) * n);
    double g[JBO_IRLENG], ir[JBO_IRLENG];
    b2mc(b, mc, n, alpha);
    jbo_freqt(mc, n, g, JBO_IRLENG - 1, -alpha);
    jbo_c2ir(g, JBO_IRLENG, ir, JBO_IRLENG);
    double e = 0.0;
    for (size_t i = 0; i < JBO_IRLENG; i++)
        e += ir[i] * ir[i];
    free(mc);
    return e;
}

/* X1: MelCepstrum::postfilter_mcp (src/vocoder/cepstrum.rs:23-37), in place on mc[n]. */
void jbo_postfilter_mcp(double *mc, size_t n, double alpha, double beta)
{
    if (!(beta > 0.0 && n > 2))
        return;
    double *b = (double *)malloc(sizeof(double) * n);
    mc2b(mc, b, n, alpha);
    const double e1 = jbo_b2en(b, n, alpha);
    b[1] -= beta * alpha * b[2];
    for (size_t k = 2; k < n; k++)
        b[k] *= 1.0 + beta;
    const double e2 = jbo_b2en(b, n, alpha);
    b[0] += log(e1 / e2) / 2.0;
    b2mc(b, mc, n, alpha);
    free(b);
}

/* ---------------------------------------------------------------------------------------------
 * X2: Stage::NonZero (voices with GAMMA != 0: the spectrum stream holds [gain, LSP...]), restated
 * from src/vocoder/{lsp.rs, generalized.rs, cepstrum.rs:69-103, mglsa.rs, mod.rs:90-107,142-176}.
 * PARITY UNPINNED: no reference test reaches this branch and no voice with GAMMA != 0 exists here;
 * the restatement follows the reference as written (including lsp2lpc taking ALL len() entries of
 * the buffer -- the gain slot too -- as line spectral frequencies, lsp.rs:27-43, where hts_engine
 * passes lsp + 1) and is held by the identities of tests/test_oracle_stage.py. */

/* LineSpectralPairs::lsp2lpc (lsp.rs:27-94): out[m + 1], m = number of entries of lsp */
void jbo_lsp2lpc(const double *lsp, size_t m, double *out)
{
    const size_t mh1 = (m % 2 == 0) ? m / 2 : (m + 1) / 2, mh2 = (m % 2 == 0) ? m / 2 : (m - 1) / 2;
    double *p = (double *)calloc(mh1 + 1, sizeof(double)), *q = (double *)calloc(mh2 + 1, sizeof(double));
    double *a0 = (double *)calloc(mh1 + 1, sizeof(double)), *a1 = (double *)calloc(mh1 + 1, sizeof(double));
    double *a2 = (double *)calloc(mh1 + 1, sizeof(double)), *b0 = (double *)calloc(mh2 + 1, sizeof(double));
    double *b1 = (double *)calloc(mh2 + 1, sizeof(double)), *b2 = (double *)calloc(mh2 + 1, sizeof(double));
    for (size_t i = 0, k = 0; k < m; i++, k += 2)
        p[i] = -2.0 * cos(lsp[k]);
    for (size_t i = 0, k = 1; k < m; i++, k += 2)
        q[i] = -2.0 * cos(lsp[k]);
    double xff = 0.0, xf = 0.0;
    for (size_t i = 0; i <= m; i++)
        out[i] = 0.0;
    for (size_t k = 0; k <= m; k++) {
        const double xx = k == 0 ? 1.0 : 0.0;
        if (m % 2 == 1) {
            a0[0] = xx;
            b0[0] = xx - xff;
            xff = xf;
            xf = xx;
        } else {
            a0[0] = xx + xf;
            b0[0] = xx - xf;
            xf = xx;
        }
        for (size_t i = 0; i < mh1; i++) {
            a0[i + 1] = a0[i] + p[i] * a1[i] + a2[i];
            a2[i] = a1[i];
            a1[i] = a0[i];
        }
        for (size_t i = 0; i < mh2; i++) {
            b0[i + 1] = b0[i] + q[i] * b1[i] + b2[i];
            b2[i] = b1[i];
            b1[i] = b0[i];
        }
        if (k > 0)
            out[k - 1] = -0.5 * (a0[mh1] + b0[mh2]);
    }
    for (size_t i = m; i-- > 0;)
        out[i + 1] = -out[i];
    out[0] = 1.0;
    free(p); free(q); free(a0); free(a1); free(a2); free(b0); free(b1); free(b2);
}

/* Generalized::gnorm / ignorm (generalized.rs:6-38), in place */
void jbo_gnorm(double *c, size_t n, double gamma)
{
    if (gamma != 0.0) {
        const double k = 1.0 + gamma * c[0];
        c[0] = pow(k, 1.0 / gamma);
        for (size_t i = 1; i < n; i++)
            c[i] = c[i] / k;
    } else {
        c[0] = exp(c[0]);
    }
}
void jbo_ignorm(double *c, size_t n, double gamma)
{
    if (gamma != 0.0) {
        const double k = pow(c[0], gamma);
        c[0] = (k - 1.0) / gamma;
        for (size_t i = 1; i < n; i++)
            c[i] = c[i] * k;
    } else {
        c[0] = log(c[0]);
    }
}

/* MelGeneralizedCepstrum::gc2gc (cepstrum.rs:69-92): c1[n1] with gamma g1 -> c2[m2 + 1] with gamma g2 */
void jbo_gc2gc(const double *c1, size_t n1, double g1, double *c2, size_t m2, double g2)
{
    c2[0] = c1[0];
    for (size_t i = 1; i <= m2; i++) {
        double ss1 = 0.0, ss2 = 0.0;
        const size_t kend = n1 < i ? n1 : i;
        for (size_t k = 1; k < kend; k++) {
            const size_t mk = i - k;
            const double cc = c1[k] * c2[mk];
            ss1 += (double)mk * cc;
            ss2 += (double)k * cc;
        }
        if (i < n1)
            c2[i] = c1[i] + (g2 * ss2 - g1 * ss1) / (double)i;
        else
            c2[i] = (g2 * ss2 - g1 * ss1) / (double)i;
    }
}

/* MelGeneralizedCepstrum::mgc2mgc (cepstrum.rs:94-102): c1[n1] (alpha a1, gamma g1) -> out[m2 + 1] */
void jbo_mgc2mgc(const double *c1, size_t n1, double a1, double g1, double *out, size_t m2, double a2, double g2)
{
    double *t;
    size_t nt;
    if (a1 == a2) {
        nt = n1;
        t = (double *)malloc(sizeof(double) * nt);
        memcpy(t, c1, sizeof(double) * nt);
    } else {
        const double a = (a2 - a1) / (1.0 - a1 * a2); /* cepstrum.rs:98: 1.0 - self.alpha * alpha */
        nt = m2 + 1;
        t = (double *)malloc(sizeof(double) * nt);
        jbo_freqt(c1, n1, t, m2, a);
    }
    jbo_gnorm(t, nt, g1);
    jbo_gc2gc(t, nt, g1, out, m2, g2);
    jbo_ignorm(out, m2 + 1, g2);
    free(t);
}

/* LineSpectralPairs::lsp2mgc (lsp.rs:96-107): lsp[n] -> mgc[n] */
void jbo_lsp2mgc(const double *lsp, size_t n, double alpha, int use_log_gain, size_t stage, double gamma, double *mgc)
{
    double *lpc = (double *)malloc(sizeof(double) * (n + 1));
    jbo_lsp2lpc(lsp, n, lpc);
    lpc[0] = use_log_gain ? exp(lsp[0]) : lsp[0];
    jbo_ignorm(lpc, n + 1, gamma);
    for (size_t i = 1; i < n + 1; i++)
        lpc[i] *= -(double)stage;
    jbo_mgc2mgc(lpc, n + 1, alpha, gamma, mgc, n - 1, alpha, gamma);
    free(lpc);
}

static double lsp2en(const double *lsp, size_t n, double alpha, int use_log_gain, size_t stage, double gamma)
{
    double *m = (double *)malloc(sizeof(double) * n);
    jbo_lsp2mgc(lsp, n, alpha, use_log_gain, stage, gamma, m);
    double e = 0.0;
    for (size_t i = 0; i < n; i++)
        e += m[i] * m[i];
    free(m);
    return e;
}

/* LineSpectralPairs::postfilter_lsp (lsp.rs:113-139), in place */
void jbo_postfilter_lsp(double *lsp, size_t n, double alpha, int use_log_gain, size_t stage, double gamma, double beta)
{
    if (!(beta > 0.0 && n > 2))
        return;
    double *buf = (double *)calloc(n, sizeof(double));
    const double en1 = lsp2en(lsp, n, alpha, use_log_gain, stage, gamma);
    for (size_t i = 0; i < n; i++) {
        if (i > 1 && i < n - 1) {
            const double d1 = beta * (lsp[i + 1] - lsp[i]);
            const double d2 = beta * (lsp[i] - lsp[i - 1]);
            buf[i] = lsp[i - 1] + d2 + (d2 * d2 * ((lsp[i + 1] - lsp[i - 1]) - (d1 + d2))) / ((d2 * d2) + (d1 * d1));
        } else {
            buf[i] = lsp[i];
        }
    }
    memcpy(lsp, buf, sizeof(double) * n);
    free(buf);
    const double en2 = lsp2en(lsp, n, alpha, use_log_gain, stage, gamma);
    if (en1 != en2) {
        if (use_log_gain)
            lsp[0] += 0.5 * log(en1 / en2);
        else
            lsp[0] *= sqrt(en1 / en2);
    }
}

/* LineSpectralPairs::check_lsp_stability (lsp.rs:141-165), in place */
void jbo_check_lsp_stability(double *lsp, size_t n)
{
    const double PI = 3.14159265358979323846;
    const double min = 0.25 * PI / (double)n;
    const size_t last = n - 1;
    for (int it = 0; it < 4; it++) {
        int find = 0;
        for (size_t j = 1; j < last; j++) {
            const double tmp = lsp[j + 1] - lsp[j];
            if (tmp < min) {
                lsp[j] -= 0.5 * (min - tmp);
                lsp[j + 1] += 0.5 * (min - tmp);
                find = 1;
            }
        }
        if (lsp[1] < min) {
            lsp[1] = min;
            find = 1;
        }
        if (lsp[last] > PI - min) {
            lsp[last] = PI - min;
            find = 1;
        }
        if (!find)
            break;
    }
}

/* the coefficients of one frame (mod.rs:92-106 first frame: filtered = 0; mod.rs:148-156: filtered = 1) */
void jbo_stage_coefficients(const double *spectrum, size_t n, double alpha, double beta, int use_log_gain,
                            size_t stage, int filtered, double *cc)
{
    const double gamma = -1.0 / (double)stage; /* stage.rs:31 */
    double *lsp = (double *)malloc(sizeof(double) * n), *mgc = (double *)malloc(sizeof(double) * n);
    memcpy(lsp, spectrum, sizeof(double) * n);
    if (filtered) {
        jbo_postfilter_lsp(lsp, n, alpha, use_log_gain, stage, gamma, beta);
        jbo_check_lsp_stability(lsp, n);
    }
    jbo_lsp2mgc(lsp, n, alpha, use_log_gain, stage, gamma, mgc);
    mc2b(mgc, cc, n, alpha);
    jbo_gnorm(cc, n, gamma);
    for (size_t i = 1; i < n; i++)
        cc[i] *= gamma;
    free(lsp);
    free(mgc);
}

/* MelGeneralizedLogSpectrumApproximation::df / dff (mglsa.rs:15-41): d[stage][n], one sample */
void jbo_mglsa_df(double *d, size_t stage, size_t n, double *x, double alpha, const double *c)
{
    const double aa = 1.0 - alpha * alpha;
    for (size_t s = 0; s < stage; s++) {
        double *ds = d + s * n;
        double y = ds[0] * c[1];
        for (size_t i = 1; i < n - 1; i++) {
            ds[i] += alpha * (ds[i + 1] - ds[i - 1]);
            y += ds[i] * c[i + 1];
        }
        *x -= y;
        for (size_t i = n - 1; i >= 1; i--)
            ds[i] = ds[i - 1];
        ds[0] = alpha * ds[0] + aa * *x;
    }
}

/* Vocoder::synthesize, Stage::NonZero branch (mod.rs:90-107,142-176), looped over the frames.
 * coef != NULL: the per-frame coefficients cc are TAKEN from coef[T][nmcp] (and the first frame's start
 * from cfirst[nmcp]) instead of computed from the spectrum -- the LSP -> coefficient conversion is
 * ill-conditioned (tests/test_oracle_stage.py), so the filter loop is checked on given coefficients. */
int jbo_vocoder_stage_coef(int fs, int fperiod_i, double alpha, double beta, double volume, int stage_i,
                           int use_log_gain, int nmcp_i, int nlpf_i, size_t T, const double *lf0, const double *mcp,
                           const double *lpf, const double *coef, const double *cfirst, double *pcm, double *excd);
int jbo_vocoder_stage(int fs, int fperiod_i, double alpha, double beta, double volume, int stage_i, int use_log_gain,
                      int nmcp_i, int nlpf_i, size_t T, const double *lf0, const double *mcp, const double *lpf,
                      double *pcm, double *excd)
{
    return jbo_vocoder_stage_coef(fs, fperiod_i, alpha, beta, volume, stage_i, use_log_gain, nmcp_i, nlpf_i, T, lf0,
                                  mcp, lpf, NULL, NULL, pcm, excd);
}
int jbo_vocoder_stage_coef(int fs, int fperiod_i, double alpha, double beta, double volume, int stage_i,
                           int use_log_gain, int nmcp_i, int nlpf_i, size_t T, const double *lf0, const double *mcp,
                           const double *lpf, const double *coef, const double *cfirst, double *pcm, double *excd)
{
    const double MAX_LF0 = 9.903487552536127, MIN_LF0 = 2.995732273553991; /* constants.rs:4-6 */
    const size_t fperiod = (size_t)fperiod_i, nmcp = (size_t)nmcp_i, nlpf = (size_t)nlpf_i, stage = (size_t)stage_i;
    if (nmcp < 3 || stage == 0)
        return -1;
    double *d = (double *)calloc(stage * nmcp, sizeof(double));
    exc_t e;
    memset(&e, 0, sizeof e);
    e.nring = nlpf;
    e.ring = (double *)calloc(nlpf ? nlpf : 1, sizeof(double));
    e.random.used = 64;
    e.random.next = 1;
    double *c = (double *)calloc(nmcp, sizeof(double)), *cc = (double *)calloc(nmcp, sizeof(double));
    double *cinc = (double *)calloc(nmcp, sizeof(double));
    int is_first = 1;
    for (size_t t = 0; t < T; t++) {
        const double l = lf0[t];
        double p;
        if (l == JBO_NODATA) {
            p = 0.0;
        } else {
            const double cl = l < MIN_LF0 ? MIN_LF0 : (l > MAX_LF0 ? MAX_LF0 : l);
            p = (double)fs / exp(cl);
        }
        const double *spec = mcp + t * nmcp;
        if (is_first) {
            is_first = 0;
            if (coef)
                memcpy(c, cfirst, sizeof(double) * nmcp);
            else
                jbo_stage_coefficients(spec, nmcp, alpha, beta, use_log_gain, stage, 0, c);
        }
        if (coef)
            memcpy(cc, coef + t * nmcp, sizeof(double) * nmcp);
        else
            jbo_stage_coefficients(spec, nmcp, alpha, beta, use_log_gain, stage, 1, cc);
        for (size_t k = 0; k < nmcp; k++)
            cinc[k] = (cc[k] - c[k]) / (double)fperiod;
        exc_start(&e, p, fperiod);
        const double *lp = lpf ? lpf + t * nlpf : NULL;
        double *raw = pcm + t * fperiod;
        for (size_t i = 0; i < fperiod; i++) {
            double pu;
            double x = exc_get(&e, lp, &pu);
            if (excd)
                excd[t * fperiod + i] = x;
            x *= c[0];
            jbo_mglsa_df(d, stage, nmcp, &x, alpha, c);
            for (size_t k = 0; k < nmcp; k++)
                c[k] += cinc[k];
            raw[i] = x * volume;
        }
        e.pitch_of_curr_point = p; /* Excitation::end (excitation.rs:102-104) */
        memcpy(c, cc, sizeof(double) * nmcp);
    }
    free(d);
    free(e.ring);
    free(c);
    free(cc);
    free(cinc);
    return 0;
}

/* V2,V5,V8,V9: Vocoder::synthesize Stage::Zero (src/vocoder/mod.rs:72-141) looped
 * as SpeechGenerator::generate_all does (src/speech.rs:87-96). */
int jbo_vocoder(int fs, int fperiod_i, double alpha, double volume, int nmcp_i, int nlpf_i,
                size_t T, const double *lf0, const double *mcp, const double *lpf, double *pcm,
                double *excd, double *pulsed)
{
    return jbo_vocoder_beta(fs, fperiod_i, alpha, 0.0, volume, nmcp_i, nlpf_i, T, lf0, mcp, lpf, pcm,
                            excd, pulsed);
}

/* Same with the post-filter coefficient beta (Vocoder::new's beta, src/vocoder/mod.rs:45-70).
 * The first frame starts from the UN-filtered mc2b(spectrum) (mod.rs:80-89) and interpolates to
 * the filtered one (mod.rs:116-118). */
int jbo_vocoder_beta(int fs, int fperiod_i, double alpha, double beta, double volume, int nmcp_i,
                     int nlpf_i, size_t T, const double *lf0, const double *mcp, const double *lpf,
                     double *pcm, double *excd, double *pulsed)
{
    const double MAX_LF0 = 9.903487552536127, MIN_LF0 = 2.995732273553991; /* constants.rs:4-6 */
    size_t fperiod = (size_t)fperiod_i, nmcp = (size_t)nmcp_i, nlpf = (size_t)nlpf_i;
    if (nmcp < 2)
        return -1;
    mlsa_t f;
    memset(&f, 0, sizeof f);
    for (int i = 0; i < 6; i++)
        f.d21[i] = (double *)calloc(nmcp, sizeof(double));
    exc_t e;
    memset(&e, 0, sizeof e);
    e.nring = nlpf;
    e.ring = (double *)calloc(nlpf ? nlpf : 1, sizeof(double));
    e.random.used = 64;
    e.random.next = 1;
    double *c = (double *)calloc(nmcp, sizeof(double));
    double *cc = (double *)calloc(nmcp, sizeof(double));
    double *cinc = (double *)calloc(nmcp, sizeof(double));
    double *pf = (double *)calloc(nmcp, sizeof(double));
    int is_first = 1;
    for (size_t t = 0; t < T; t++) {
        double l = lf0[t];
        double p;
        if (l == JBO_NODATA) {
            p = 0.0;
        } else {
            double cl = l < MIN_LF0 ? MIN_LF0 : (l > MAX_LF0 ? MAX_LF0 : l);
            p = (double)fs / exp(cl);
        }
        const double *spec = mcp + t * nmcp;
        if (is_first) {
            is_first = 0;
            mc2b(spec, c, nmcp, alpha);
        }
        if (beta > 0.0 && nmcp > 2) {
            memcpy(pf, spec, sizeof(double) * nmcp);
            jbo_postfilter_mcp(pf, nmcp, alpha, beta);
            mc2b(pf, cc, nmcp, alpha);
        } else {
            mc2b(spec, cc, nmcp, alpha);
        }
        for (size_t k = 0; k < nmcp; k++)
            cinc[k] = (cc[k] - c[k]) / (double)fperiod;
        exc_start(&e, p, fperiod);
        const double *lp = lpf ? lpf + t * nlpf : NULL;
        double *raw = pcm + t * fperiod;
        for (size_t i = 0; i < fperiod; i++) {
            double pu;
            double x = exc_get(&e, lp, &pu);
            if (excd)
                excd[t * fperiod + i] = x;
            if (pulsed)
                pulsed[t * fperiod + i] = pu;
            if (x != 0.0)
                x *= exp(c[0]);
            df1(&f, &x, alpha, c);
            df2(&f, &x, alpha, c, nmcp);
            for (size_t k = 0; k < nmcp; k++)
                c[k] += cinc[k];
            raw[i] = x * volume;
        }
        e.pitch_of_curr_point = p; /* Excitation::end (excitation.rs:102-104) */
        memcpy(c, cc, sizeof(double) * nmcp);
    }
    for (int i = 0; i < 6; i++)
        free(f.d21[i]);
    free(e.ring);
    free(c);
    free(cc);
    free(cinc);
    free(pf);
    return 0;
}
