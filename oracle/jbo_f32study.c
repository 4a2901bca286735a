/*
 * oracle/jbo_f32study.c -- CPU ORACLE side study (test infrastructure, NOT the product).
 *
 * Same vocoder as jbo_vocoder (jbo_hot.c) but with the MLSA filter state, the
 * interpolated coefficients and the filter arithmetic in `float`, excitation and
 * gain kept in double.  Used to quantify what an f32-state HIP kernel can reach
 * against the f64 reference (precision contract).
 */
#include "jbo_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static const float PP[6] = {1.00000000000f, 0.49993910000f, 0.11070980000f,
                            0.01369984000f, 0.00095648530f, 0.00003041721f};

static float fir32(float *d, size_t n, float x, float a, const float *c)
{
    float aa = a * a, iaa = 1.0f - aa;
    float rem = x;
    float y = 0.0f;
    /* uniform all-pass chain, tap 1 has no dot contribution */
    for (size_t j = 1; j < n; j++) {
        float o = d[j];
        d[j] = iaa * rem + a * o;
        rem = o - a * rem;
        if (j >= 2)
            y += c[j] * d[j];
    }
    return y;
}

int jbo_vocoder_f32state(int fs, int fperiod_i, double alpha, double volume, int nmcp_i, int nlpf_i,
                         size_t T, const double *lf0, const double *mcp, const double *lpf,
                         const double *exc_in /* excitation before gain [T*fperiod], from the f64 run */,
                         double *pcm)
{
    (void)fs; (void)lf0; (void)lpf; (void)nlpf_i;
    size_t fperiod = (size_t)fperiod_i, nmcp = (size_t)nmcp_i;
    float a = (float)alpha, iaa = 1.0f - a * a;
    float d11[6] = {0}, d12[6] = {0}, d22[6] = {0};
    float *d21[6];
    for (int i = 0; i < 6; i++)
        d21[i] = (float *)calloc(nmcp, sizeof(float));
    double *cprev = (double *)calloc(nmcp, sizeof(double));
    double *cc = (double *)calloc(nmcp, sizeof(double));
    float *c = (float *)calloc(nmcp, sizeof(float));
    float *c0f = (float *)calloc(nmcp, sizeof(float));
    float *cinc = (float *)calloc(nmcp, sizeof(float));
    for (size_t t = 0; t < T; t++) {
        const double *spec = mcp + t * nmcp;
        /* mc2b in double (frame prologue stays f64) */
        cc[nmcp - 1] = spec[nmcp - 1];
        for (size_t i = nmcp - 1; i-- > 0;)
            cc[i] = spec[i] - alpha * cc[i + 1];
        if (t == 0)
            memcpy(cprev, cc, sizeof(double) * nmcp);
        for (size_t k = 0; k < nmcp; k++) {
            c0f[k] = (float)cprev[k];
            cinc[k] = (float)((cc[k] - cprev[k]) / (double)fperiod);
        }
        double c0d = cprev[0], c0inc = (cc[0] - cprev[0]) / (double)fperiod;
        for (size_t i = 0; i < fperiod; i++) {
            double xe = exc_in[t * fperiod + i];
            if (xe != 0.0)
                xe *= exp(c0d + (double)i * c0inc);
            float x = (float)xe;
            for (size_t k = 1; k < nmcp; k++)
                c[k] = fmaf((float)i, cinc[k], c0f[k]);
            /* df1 */
            float out = 0.0f;
            for (int ii = 5; ii >= 1; ii--) {
                d11[ii] = iaa * d12[ii - 1] + a * d11[ii];
                d12[ii] = d11[ii] * c[1];
                float v = d12[ii] * PP[ii];
                x += (ii & 1) ? v : -v;
                out += v;
            }
            d12[0] = x;
            x += out;
            /* df2 */
            out = 0.0f;
            for (int ii = 5; ii >= 1; ii--) {
                d22[ii] = fir32(d21[ii - 1], nmcp, d22[ii - 1], a, c);
                float v = d22[ii] * PP[ii];
                x += (ii & 1) ? v : -v;
                out += v;
            }
            d22[0] = x;
            x += out;
            pcm[t * fperiod + i] = (double)x * volume;
        }
        memcpy(cprev, cc, sizeof(double) * nmcp);
    }
    for (int i = 0; i < 6; i++)
        free(d21[i]);
    free(cprev); free(cc); free(c); free(c0f); free(cinc);
    return 0;
}

/* f64 filter (same arithmetic as jbo_vocoder) started from ZERO state at frame
 * t_start, fed the given pre-gain excitation: quantifies how fast the MLSA filter
 * forgets its initial state (warm-up length study for time-chunked execution). */
int jbo_vocoder_from_exc(int fperiod_i, double alpha, int nmcp_i, size_t T, size_t t_start,
                         const double *mcp, const double *exc_in, double *pcm)
{
    size_t fperiod = (size_t)fperiod_i, nmcp = (size_t)nmcp_i;
    double a = alpha, iaa = 1.0 - a * a;
    double d11[6] = {0}, d12[6] = {0}, d22[6] = {0};
    double *d21[6];
    static const double P6[6] = {1.0, 0.4999391, 0.1107098, 0.01369984, 0.0009564853, 0.00003041721};
    for (int i = 0; i < 6; i++)
        d21[i] = (double *)calloc(nmcp, sizeof(double));
    double *c = (double *)calloc(nmcp, sizeof(double));
    double *cc = (double *)calloc(nmcp, sizeof(double));
    double *cinc = (double *)calloc(nmcp, sizeof(double));
    for (size_t t = t_start; t < T; t++) {
        const double *spec = mcp + t * nmcp;
        cc[nmcp - 1] = spec[nmcp - 1];
        for (size_t i = nmcp - 1; i-- > 0;)
            cc[i] = spec[i] - alpha * cc[i + 1];
        if (t == 0) {
            memcpy(c, cc, sizeof(double) * nmcp);
        } else {
            const double *sp = mcp + (t - 1) * nmcp;
            c[nmcp - 1] = sp[nmcp - 1];
            for (size_t i = nmcp - 1; i-- > 0;)
                c[i] = sp[i] - alpha * c[i + 1];
        }
        for (size_t k = 0; k < nmcp; k++)
            cinc[k] = (cc[k] - c[k]) / (double)fperiod;
        for (size_t i = 0; i < fperiod; i++) {
            double x = exc_in[t * fperiod + i];
            if (x != 0.0)
                x *= exp(c[0]);
            double out = 0.0;
            for (int ii = 5; ii >= 1; ii--) {
                d11[ii] = iaa * d12[ii - 1] + a * d11[ii];
                d12[ii] = d11[ii] * c[1];
                double v = d12[ii] * P6[ii];
                x += (ii & 1) ? v : -v;
                out += v;
            }
            d12[0] = x;
            x += out;
            out = 0.0;
            for (int ii = 5; ii >= 1; ii--) {
                double *d = d21[ii - 1];
                double rem = d22[ii - 1], y = 0.0;
                for (size_t j = 1; j < nmcp; j++) {
                    double o = d[j];
                    d[j] = iaa * rem + a * o;
                    rem = o - a * rem;
                    if (j >= 2)
                        y += c[j] * d[j];
                }
                d22[ii] = y;
                double v = y * P6[ii];
                x += (ii & 1) ? v : -v;
                out += v;
            }
            d22[0] = x;
            x += out;
            for (size_t k = 0; k < nmcp; k++)
                c[k] += cinc[k];
            pcm[t * fperiod + i] = x;
        }
    }
    for (int i = 0; i < 6; i++)
        free(d21[i]);
    free(c); free(cc); free(cinc);
    return 0;
}
