/*
 * cqpsk_ref.c -- CPU oracle (plain C, float64) for row A12: the P25 Phase-2 CQPSK receive
 * chain of wavecapsdr/dsp/p25/cqpsk.py (RRC matched filter :287-290, CostasLoop :84-196,
 * differential decode :308-350) with the Mueller-Muller timing recovery of
 * dsp/p25/symbol_timing.py:214-380, and the standalone GardnerTED (symbol_timing.py:60-211).
 * TEST INFRASTRUCTURE ONLY (see oracle/ref_np.py header); pinned by tests/golden/cqpsk.npz.
 *
 * Everything in the reference is Python float / complex128 arithmetic, restated here operation
 * by operation in double (gcc -ffp-contract=off).  Transcendentals (cexp, angle) come from
 * libm; numpy uses SVML kernels for them on AVX-512 hosts, so agreement with the reference is to
 * ~1 ulp of float64 through feedback loops -- symbols are compared at 1e-9, dibits exactly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PI 3.141592653589793

typedef struct { double re, im; } cd;

typedef struct cqpsk_ref {
    double sps;
    int nt;
    double *taps;         /* RRC taps (float32 values widened) */
    cd *state;            /* carried lfilter state, len nt-1 */
    double *zi0;          /* lfilter_zi template */
    /* Costas */
    double c_kp, c_ki, c_maxf, c_phase, c_freq;
    /* Mueller-Muller */
    double t_kp, t_ki, t_maxdev, t_phase, t_integ;
    cd buf[4];
    int buf_idx;
    cd prev_sym, prev_dec;
    double prev_phase;
} cqpsk_ref;

void cqpsk_ref_reset(cqpsk_ref *s) {
    for (int i = 0; i < s->nt - 1; ++i) { s->state[i].re = s->zi0[i]; s->state[i].im = 0.0; }
    s->c_phase = s->c_freq = 0.0;
    s->t_phase = s->t_integ = 0.0;
    memset(s->buf, 0, sizeof(s->buf));
    s->buf_idx = 0;
    s->prev_sym.re = s->prev_sym.im = s->prev_dec.re = s->prev_dec.im = 0.0;
    s->prev_phase = 0.0;
}

cqpsk_ref *cqpsk_ref_create(double sps, const float *taps, int nt, const double *zi0, double c_kp, double c_ki,
                            double c_maxf, double t_kp, double t_ki) {
    cqpsk_ref *s = (cqpsk_ref *)calloc(1, sizeof(*s));
    s->sps = sps; s->nt = nt;
    s->taps = (double *)malloc(sizeof(double) * nt);
    for (int i = 0; i < nt; ++i) s->taps[i] = (double)taps[i];
    s->state = (cd *)calloc(nt, sizeof(cd));
    s->zi0 = (double *)malloc(sizeof(double) * nt);
    memcpy(s->zi0, zi0, sizeof(double) * (nt - 1));
    s->c_kp = c_kp; s->c_ki = c_ki; s->c_maxf = c_maxf;
    s->t_kp = t_kp; s->t_ki = t_ki; s->t_maxdev = sps / 4;
    cqpsk_ref_reset(s);
    return s;
}

void cqpsk_ref_destroy(cqpsk_ref *s) {
    if (!s) return;
    free(s->taps); free(s->state); free(s->zi0); free(s);
}

static double interp1(double v0, double v1, double v2, double v3, double mu) {
    /* symbol_timing.py:298-303 */
    double c0 = v1;
    double c1 = (v2 - v0) / 2;
    double c2 = v0 - 5 * v1 / 2 + 2 * v2 - v3 / 2;
    double c3 = (v3 - v0) / 2 + 3 * (v1 - v2) / 2;
    return c0 + mu * (c1 + mu * (c2 + mu * c3));
}

/* returns number of dibits; symbols (optional) gets the complex symbols */
int cqpsk_ref_demodulate(cqpsk_ref *s, const float *iq, int n, uint8_t *dibits, double *symbols, int cap) {
    if (n <= 0) return 0;
    const int L = s->nt, H = L - 1;
    /* matched filter: full = convolve(taps, x); full[:H] += state * x[0]; out = full[:n]; state = full[n:] */
    cd *full = (cd *)calloc((size_t)n + H, sizeof(cd));
    for (int t = 0; t < n + H; ++t) {
        double ar = 0.0, ai = 0.0;
        int k0 = t - (n - 1) > 0 ? t - (n - 1) : 0;
        int k1 = t < H ? t : H;
        for (int k = k0; k <= k1; ++k) {
            ar += s->taps[k] * (double)iq[2 * (t - k)];
            ai += s->taps[k] * (double)iq[2 * (t - k) + 1];
        }
        full[t].re = ar; full[t].im = ai;
    }
    {
        double xr = (double)iq[0], xi = (double)iq[1];
        for (int t = 0; t < H; ++t) {   /* zi = state * iq[0] (complex product) */
            double zr = s->state[t].re * xr - s->state[t].im * xi;
            double zi = s->state[t].re * xi + s->state[t].im * xr;
            full[t].re += zr; full[t].im += zi;
        }
    }
    for (int t = 0; t < H; ++t) s->state[t] = full[n + t];
    int count = 0;
    const double q = PI / 4;
    static const double CR[4] = {1, -1, -1, 1}, CI[4] = {1, 1, -1, -1};
    const double inv = 1.4142135623730951;
    for (int t = 0; t < n; ++t) {
        /* ---- Costas loop (cqpsk.py:126-177) */
        double c = cos(s->c_phase), sn = sin(s->c_phase);
        /* sample * exp(-1j*phase) = (a + bi)(c - i sn) */
        double cr = full[t].re * c - full[t].im * (-sn);
        double ci = full[t].re * (-sn) + full[t].im * c;
        double ph = atan2(ci, cr);
        double ideal = rint(ph / q) * q;
        double err = ph - ideal;
        while (err > PI) err -= 2 * PI;
        while (err < -PI) err += 2 * PI;
        s->c_freq += s->c_ki * err;
        if (s->c_freq < -s->c_maxf) s->c_freq = -s->c_maxf;
        if (s->c_freq > s->c_maxf) s->c_freq = s->c_maxf;
        double adj = s->c_kp * err + s->c_freq;
        s->c_phase += adj;
        while (s->c_phase > PI) s->c_phase -= 2 * PI;
        while (s->c_phase < -PI) s->c_phase += 2 * PI;
        /* ---- Mueller-Muller (symbol_timing.py:339-372) */
        s->buf_idx = (s->buf_idx + 1) % 4;
        s->buf[s->buf_idx].re = cr; s->buf[s->buf_idx].im = ci;
        s->t_phase += 1.0;
        if (s->t_phase >= s->sps) {
            s->t_phase -= s->sps;
            double mu = s->t_phase / s->sps;
            int ix = s->buf_idx;
            cd v0 = s->buf[(ix + 1) % 4], v1 = s->buf[(ix + 2) % 4], v2 = s->buf[(ix + 3) % 4], v3 = s->buf[ix % 4];
            double sr = interp1(v0.re, v1.re, v2.re, v3.re, mu);
            double si = interp1(v0.im, v1.im, v2.im, v3.im, mu);
            int best = 0;
            double bd = 0.0;
            for (int k = 0; k < 4; ++k) {
                double d = hypot(CR[k] / inv - sr, CI[k] / inv - si);
                if (k == 0 || d < bd) { bd = d; best = k; }
            }
            double dr = CR[best] / inv, di = CI[best] / inv;
            /* Re{conj(prev_dec)*cur - conj(dec)*prev_sym} */
            double e1 = s->prev_dec.re * sr - (-s->prev_dec.im) * si;
            double e2 = dr * s->prev_sym.re - (-di) * s->prev_sym.im;
            double e = e1 - e2;
            s->t_integ += s->t_ki * e;
            if (s->t_integ < -s->t_maxdev) s->t_integ = -s->t_maxdev;
            if (s->t_integ > s->t_maxdev) s->t_integ = s->t_maxdev;
            double tadj = s->t_kp * e + s->t_integ;
            s->t_phase += tadj;
            s->prev_sym.re = sr; s->prev_sym.im = si;
            s->prev_dec.re = dr; s->prev_dec.im = di;
            /* ---- differential decode (cqpsk.py:321-348) */
            double p = atan2(si, sr);
            double dp = p - s->prev_phase;
            while (dp > PI) dp -= 2 * PI;
            while (dp < -PI) dp += 2 * PI;
            long idx = (long)rint((dp + PI) / q);
            idx = ((idx % 8) + 8) % 8;
            if (count < cap) {
                dibits[count] = (uint8_t)(idx >> 1);
                if (symbols) { symbols[2 * count] = sr; symbols[2 * count + 1] = si; }
                count++;
            }
            s->prev_phase = p;
        }
    }
    free(full);
    return count;
}

/* ---- GardnerTED.process_block (symbol_timing.py:157-211) ------------------------------------ */
typedef struct gardner_ref {
    double sps, kp, ki, maxdev, phase, integ, buf[4], prev_symbol, prev_mid;
    int buf_idx;
} gardner_ref;

gardner_ref *gardner_ref_create(double sps, double kp, double ki) {
    gardner_ref *g = (gardner_ref *)calloc(1, sizeof(*g));
    g->sps = sps; g->kp = kp; g->ki = ki; g->maxdev = sps / 4;
    return g;
}
void gardner_ref_destroy(gardner_ref *g) { free(g); }
void gardner_ref_reset(gardner_ref *g) {
    double sps = g->sps, kp = g->kp, ki = g->ki;
    memset(g, 0, sizeof(*g));
    g->sps = sps; g->kp = kp; g->ki = ki; g->maxdev = sps / 4;
}

static double g_interp(const gardner_ref *g, double mu) {
    int ix = g->buf_idx;
    return interp1(g->buf[(ix + 1) % 4], g->buf[(ix + 2) % 4], g->buf[(ix + 3) % 4], g->buf[ix % 4], mu);
}

int gardner_ref_process(gardner_ref *g, const float *x, int n, double *symbols, double *errors, int cap) {
    int count = 0;
    for (int t = 0; t < n; ++t) {
        g->buf_idx = (g->buf_idx + 1) % 4;
        g->buf[g->buf_idx] = (double)x[t];
        g->phase += 1.0;
        if (g->phase >= g->sps) {
            g->phase -= g->sps;
            double mu = g->phase / g->sps;
            double cur = g_interp(g, mu);
            double mid_phase = g->phase + g->sps / 2;
            double mid;
            if (mid_phase >= 1.0) mid = g_interp(g, mid_phase - (double)(long)mid_phase);
            else mid = g->prev_mid;
            double e = mid * (g->prev_symbol - cur);
            g->integ += g->ki * e;
            if (g->integ < -g->maxdev) g->integ = -g->maxdev;
            if (g->integ > g->maxdev) g->integ = g->maxdev;
            double adj = g->kp * e + g->integ;
            g->phase += adj;
            if (count < cap) { symbols[count] = cur; errors[count] = e; count++; }
            g->prev_symbol = cur;
            g->prev_mid = mid;
        }
    }
    return count;
}
