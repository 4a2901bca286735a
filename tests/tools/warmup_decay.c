/* Study helper (CPU, test infrastructure): how fast does the MLSA filter forget a wrong start state, per
 * start frame?  The difference between the exact run and a run started from ZERO state at frame t0 obeys the
 * filter's homogeneous recursion (zero input) from the exact state at t0 (the recursion of
 * oracle/jbo_f32study.c:jbo_vocoder_from_exc, src/vocoder/mlsa.rs:54-163).  out[s][w-1] = max|diff| /
 * max|exact state| over the carried state at frame t0 + w, the quantity the product's hand-off check bounds
 * by verify_tol.  Built ad hoc by tests/tools/warmup_decay.py. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double d11[6], d12[6], d22[6], d21[5][64]; } St;

static void mc2b(const double *spec, double *b, size_t n, double alpha)
{
    b[n - 1] = spec[n - 1];
    for (size_t i = n - 1; i-- > 0;)
        b[i] = spec[i] - alpha * b[i + 1];
}

static void frame(St *s, const double *mcp, size_t t, size_t nmcp, size_t fperiod, double a, const double *exc)
{
    static const double P6[6] = {1.0, 0.4999391, 0.1107098, 0.01369984, 0.0009564853, 0.00003041721};
    double c[64], cc[64], cinc[64];
    const double iaa = 1.0 - a * a;
    mc2b(mcp + t * nmcp, cc, nmcp, a);
    if (t == 0)
        memcpy(c, cc, sizeof(double) * nmcp);
    else
        mc2b(mcp + (t - 1) * nmcp, c, nmcp, a);
    for (size_t k = 0; k < nmcp; k++)
        cinc[k] = (cc[k] - c[k]) / (double)fperiod;
    for (size_t i = 0; i < fperiod; i++) {
        double x = exc ? exc[t * fperiod + i] : 0.0;
        if (x != 0.0)
            x *= exp(c[0]);
        double out = 0.0;
        for (int ii = 5; ii >= 1; ii--) {
            s->d11[ii] = iaa * s->d12[ii - 1] + a * s->d11[ii];
            s->d12[ii] = s->d11[ii] * c[1];
            double v = s->d12[ii] * P6[ii];
            x += (ii & 1) ? v : -v;
            out += v;
        }
        s->d12[0] = x;
        x += out;
        out = 0.0;
        for (int ii = 5; ii >= 1; ii--) {
            double *d = s->d21[ii - 1];
            double rem = s->d22[ii - 1], y = 0.0;
            for (size_t j = 1; j < nmcp; j++) {
                double o = d[j];
                d[j] = iaa * rem + a * o;
                rem = o - a * rem;
                if (j >= 2)
                    y += c[j] * d[j];
            }
            s->d22[ii] = y;
            double v = y * P6[ii];
            x += (ii & 1) ? v : -v;
            out += v;
        }
        s->d22[0] = x;
        x += out;
        for (size_t k = 0; k < nmcp; k++)
            c[k] += cinc[k];
    }
}

static double st_max(const St *s, size_t nmcp)
{
    double m = 0.0;
    for (int i = 1; i <= 5; i++) m = fmax(m, fabs(s->d11[i]));
    for (int i = 0; i <= 4; i++) { m = fmax(m, fabs(s->d12[i])); m = fmax(m, fabs(s->d22[i])); }
    for (int q = 0; q < 5; q++)
        for (size_t j = 1; j < nmcp; j++) m = fmax(m, fabs(s->d21[q][j]));
    return m;
}

/* smax[t] = max|exact state| on entering frame t (t = 0..T); out[(t0/stride)][w-1], w = 1..wmax */
int warmup_decay(int fperiod, double alpha, int nmcp, size_t T, const double *mcp, const double *exc, int wmax,
                 int stride, double *smax, double *out)
{
    St *snap = (St *)calloc(T + 1, sizeof(St));
    if (!snap) return 1;
    St s;
    memset(&s, 0, sizeof s);
    for (size_t t = 0; t < T; t++) {
        snap[t] = s;
        smax[t] = st_max(&s, (size_t)nmcp);
        frame(&s, mcp, t, (size_t)nmcp, (size_t)fperiod, alpha, exc);
    }
    snap[T] = s;
    smax[T] = st_max(&s, (size_t)nmcp);
    long ns = (long)((T + stride - 1) / stride);
#pragma omp parallel for schedule(dynamic, 16)
    for (long k = 0; k < ns; k++) {
        size_t t0 = (size_t)k * stride;
        St e = snap[t0];
        for (int w = 1; w <= wmax; w++) {
            double r = NAN;
            if (t0 + w <= T) {
                frame(&e, mcp, t0 + w - 1, (size_t)nmcp, (size_t)fperiod, alpha, NULL);
                double m = smax[t0 + w];
                r = m > 0.0 ? st_max(&e, (size_t)nmcp) / m : (st_max(&e, (size_t)nmcp) > 0.0 ? INFINITY : 0.0);
            }
            out[k * wmax + (w - 1)] = r;
        }
    }
    free(snap);
    return 0;
}
