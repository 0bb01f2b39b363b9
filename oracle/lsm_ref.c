/*
 * lsm_ref.c -- CPU oracle (plain C) for the P25 Phase-1 LSM/CQPSK demodulator, row A12.
 *
 * TEST INFRASTRUCTURE ONLY: built into oracle/_build/liblsm_ref.so and loaded by tests/ only; the
 * product never links it.
 *
 * Restates wavecapsdr/decoders/p25.py:190-669 (CQPSKDemodulator.demodulate :413-482 and
 * _cqpsk_timing_recovery :484-669) under "numpy 2.2 (NEP 50 scalar promotion)", one instance per
 * object.  Precision follows the Python / numpy scalar types the reference ends up with:
 *   - AGC (:436-455): |x| float32, mean float32, gain float32 (a weak Python float until the first update,
 *     which gives the same float32 values);
 *   - NCO (:460-465): only when |freq_offset| > 1e-7; exp(-1j*(phase_acc + freq*n)) in complex128, which
 *     promotes the block to complex128, so the low-pass then runs in float64;
 *   - low-pass (:468-471): np.convolve(..., 'same') PER CALL (zero padding at both block edges, no carried
 *     state), float32 while the NCO is idle, float64 after; result cast to complex64; skipped when the
 *     block is shorter than the 63 taps;
 *   - timing loop: symbol clock and symbol time are Python floats (float64) until the first Gardner update
 *     adds a float32 to them -- from then on both are float32 (:603-608); when round(sps)+4 >= 32 the
 *     Gardner block never runs and they stay float64;
 *   - MMSE interpolation (:325-355): complex64 accumulation in tap order; offsets < 0 are skipped, so the
 *     "current" symbol uses only taps 3..7; float32 tap * complex64 sample is a complex64 multiply by
 *     (t + 0j);
 *   - differential phase (:531-537): complex64 / float32 is numpy's Smith division by (m + 0j); np.angle on
 *     complex64 is float32 arctan2; the very first symbol multiplies by np.conj(0j) = complex128(0, -0),
 *     so its phase is a float64 atan2 of signed zeros (0, -0, pi or -pi depending on the signs of curr);
 *   - frequency loop (:584-585): float32 products added to a float64 accumulator; comparisons of a float32
 *     phase with pi/2 etc. happen in float32.
 * Parity pin: tests/test_lsm_oracle.py checks dibits bit-exact and the per-symbol phases / end-of-call
 * state against goldens captured from the reference (tests/golden/lsm.npz).
 *
 * flavour 0: libm atan2f / hypotf / sincos / atan2, float32-accumulated low-pass (the reference calls SVML
 *            and BLAS kernels whose rounding differs between hosts, so neither flavour can be "the" bits);
 * flavour 1: the portable single-rounding functions of wh_portable_math.h shared with the HIP kernel,
 *            float64-accumulated low-pass, strided-tree mean.  The tests require both flavours to give the
 *            reference's dibits; HIP == flavour 1 bit for bit.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../wavecap-sdr_amd/csrc/wh_portable_math.h"

#define NT 32  /* MMSE_NTAPS: history depth (p25.py:221) */
#define NLPF 63

typedef struct lsm_ref {
    double sps;
    int half_sps, full_sps, gardner, flavour;
    float lpf[NLPF];
    float mmse[129 * 8];
    /* state */
    float agc_gain;
    double freq_offset, phase_acc;
    int f32mode;
    double clock64, symtime64;
    float clock32, symtime32, omega32;
    int first;
    float prev_re, prev_im;
    float hist[NT][2]; /* last 32 filtered samples, oldest first */
} lsm_ref;

lsm_ref *lsm_ref_create(double sps, const float *lpf, const float *mmse, int flavour) {
    lsm_ref *h = (lsm_ref *)calloc(1, sizeof(lsm_ref));
    h->sps = sps;
    h->half_sps = (int)rint(sps / 2.0); /* int(round(sps / 2)): Python round = half to even */
    h->full_sps = (int)rint(sps);
    h->gardner = h->full_sps + 4 < NT;
    h->flavour = flavour;
    memcpy(h->lpf, lpf, sizeof(h->lpf));
    memcpy(h->mmse, mmse, sizeof(h->mmse));
    h->agc_gain = 1.0f;
    h->symtime64 = 1.0 / sps;
    h->first = 1;
    return h;
}

void lsm_ref_destroy(lsm_ref *h) { free(h); }

void lsm_ref_get_state(const lsm_ref *h, double *out) {
    out[0] = h->agc_gain;
    out[1] = h->freq_offset;
    out[2] = h->phase_acc;
    out[3] = h->f32mode ? (double)h->clock32 : h->clock64;
    out[4] = h->prev_re;
    out[5] = h->prev_im;
    out[6] = h->f32mode;
}

/* numpy's float32 pairwise summation (the add.reduce inner loop) */
static float np_pairwise_f32(const float *a, size_t n) {
    if (n < 8) {
        float res = 0.f;
        for (size_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        size_t i;
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        size_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_f32(a, n2) + np_pairwise_f32(a + n2, n - n2);
    }
}

static float mean_mag(const lsm_ref *h, const float *iq, int n) {
    if (n <= 0) return 0.0f; /* (callers never pass an empty block: demodulate() returns before the AGC) */
    if (h->flavour == 0) {
        float *m = (float *)malloc(sizeof(float) * (size_t)n);
        for (int i = 0; i < n; ++i) m[i] = hypotf(iq[2 * i], iq[2 * i + 1]);
        float s = m[0]; /* add.reduce: first element, then the pairwise sum of the rest */
        if (n > 1) s = s + np_pairwise_f32(m + 1, (size_t)n - 1);
        free(m);
        return s / (float)n;
    }
    /* portable: 256 strided float64 partial sums, then a halving tree */
    double part[256];
    for (int j = 0; j < 256; ++j) part[j] = 0.0;
    for (int i = 0; i < n; ++i) part[i & 255] += (double)whm_hypotf(iq[2 * i], iq[2 * i + 1]);
    for (int s = 128; s >= 1; s >>= 1)
        for (int j = 0; j < s; ++j) part[j] += part[j + s];
    return (float)(part[0] / (double)n);
}

static float angle_f32(const lsm_ref *h, float im, float re) {
    if (re == 0.0f && im == 0.0f) { /* atan2 of signed zeros */
        if (signbit(re)) return signbit(im) ? -3.14159274101257324f : 3.14159274101257324f;
        return im;
    }
    if (im == 0.0f && re < 0.0f) return signbit(im) ? -3.14159274101257324f : 3.14159274101257324f;
    return h->flavour == 0 ? atan2f(im, re) : whm_atan2f(im, re);
}

typedef struct { float re, im; } cf;

static cf interp(const lsm_ref *h, const float (*yb)[2], int pos, int so, int imu) {
    cf acc = {0.0f, 0.0f};
    for (int tap = 0; tap < 8; ++tap) {
        int off = so + tap - 3;
        if (off < 0 || off >= NT) continue;
        float t = h->mmse[imu * 8 + tap];
        float a = yb[pos - off][0], b = yb[pos - off][1];
        float pr = t * a - 0.0f * b; /* (t + 0j) * (a + bj) in complex64 */
        float pi = t * b + 0.0f * a;
        acc.re = acc.re + pr;
        acc.im = acc.im + pi;
    }
    return acc;
}

static cf cdiv_real(cf z, float m) { /* numpy complex64 / (m + 0j), Smith's algorithm */
    float rat = 0.0f / m;
    float scl = 1.0f / (m + 0.0f * rat);
    cf o;
    o.re = (z.re + z.im * rat) * scl;
    o.im = (z.im - z.re * rat) * scl;
    return o;
}

static float hyp(const lsm_ref *h, cf z) { return h->flavour == 0 ? hypotf(z.re, z.im) : whm_hypotf(z.re, z.im); }

/* iq: interleaved complex64 [n]; dibits / phases: capacity cap; returns the symbol count (or -1). */
int lsm_ref_demodulate(lsm_ref *h, const float *iq, int n, uint8_t *dibits, float *phases, float *filtered, int cap) {
    if (n <= 0) return 0;
    const float PI_F = (float)M_PI, HALF_PI_F = (float)(M_PI / 2), NHALF_PI_F = (float)(-(M_PI / 2));
    /* AGC :436-455 */
    float mean = mean_mag(h, iq, n);
    if (mean > 1e-8f) {
        float target = 1.0f / mean;
        float g = h->agc_gain * 0.995f + target * 0.005f;
        if (g < 0.01f) g = 0.01f;
        if (g > 500.0f) g = 500.0f;
        h->agc_gain = g;
    }
    const float g = h->agc_gain;
    float(*yb)[2] = (float(*)[2])malloc(sizeof(float) * 2 * (size_t)(n + NT));
    memcpy(yb, h->hist, sizeof(h->hist));
    float(*y)[2] = yb + NT;
    const int nco = fabs(h->freq_offset) > 1e-7;
    const int filt = n >= NLPF;
    if (!nco) {
        float *xs = (float *)malloc(sizeof(float) * 2 * (size_t)n);
        for (int i = 0; i < n; ++i) { /* complex64 * (g + 0j) */
            float a = iq[2 * i], b = iq[2 * i + 1];
            xs[2 * i] = a * g - b * 0.0f;
            xs[2 * i + 1] = a * 0.0f + b * g;
        }
        for (int i = 0; i < n; ++i) {
            if (!filt) { y[i][0] = xs[2 * i]; y[i][1] = xs[2 * i + 1]; continue; }
            if (h->flavour == 0) {
                float ar = 0.f, ai = 0.f;
                for (int j = 0; j < NLPF; ++j) { /* window order, reversed taps (np.correlate form) */
                    int s = i - 31 + j;
                    if (s < 0 || s >= n) continue;
                    ar += xs[2 * s] * h->lpf[NLPF - 1 - j];
                    ai += xs[2 * s + 1] * h->lpf[NLPF - 1 - j];
                }
                y[i][0] = ar; y[i][1] = ai;
            } else {
                double ar = 0.0, ai = 0.0;
                for (int k = 0; k < NLPF; ++k) {
                    int s = i + 31 - k;
                    if (s < 0 || s >= n) continue;
                    ar = ar + (double)h->lpf[k] * (double)xs[2 * s];
                    ai = ai + (double)h->lpf[k] * (double)xs[2 * s + 1];
                }
                y[i][0] = (float)ar; y[i][1] = (float)ai;
            }
        }
        free(xs);
    } else {
        double *xd = (double *)malloc(sizeof(double) * 2 * (size_t)n);
        for (int i = 0; i < n; ++i) {
            float a = iq[2 * i], b = iq[2 * i + 1];
            double xr = (double)(a * g - b * 0.0f), xi = (double)(a * 0.0f + b * g);
            double th = h->phase_acc + h->freq_offset * (double)i;
            double sn, cs;
            if (h->flavour == 0) { sn = sin(th); cs = cos(th); } else whm_sincos_f64(th, &sn, &cs);
            double nr = cs, ni = -sn; /* exp(-1j * th) */
            xd[2 * i] = xr * nr - xi * ni;
            xd[2 * i + 1] = xr * ni + xi * nr;
        }
        h->phase_acc += h->freq_offset * (double)n;
        if (h->flavour == 0) {
            h->phase_acc = atan2(sin(h->phase_acc), cos(h->phase_acc));
        } else {
            h->phase_acc = h->phase_acc - 6.283185307179586 * rint(h->phase_acc * 0.15915494309189535);
        }
        for (int i = 0; i < n; ++i) {
            if (!filt) { y[i][0] = (float)xd[2 * i]; y[i][1] = (float)xd[2 * i + 1]; continue; }
            double ar = 0.0, ai = 0.0;
            for (int k = 0; k < NLPF; ++k) {
                int s = i + 31 - k;
                if (s < 0 || s >= n) continue;
                ar = ar + (double)h->lpf[k] * xd[2 * s];
                ai = ai + (double)h->lpf[k] * xd[2 * s + 1];
            }
            y[i][0] = (float)ar; y[i][1] = (float)ai;
        }
        free(xd);
    }
    if (filtered) memcpy(filtered, y, sizeof(float) * 2 * (size_t)n);

    /* timing recovery :484-669 */
    int count = 0;
    for (int i = 0; i < n; ++i) {
        int fire;
        if (!h->f32mode) { h->clock64 += h->symtime64; fire = h->clock64 >= 1.0; }
        else { h->clock32 = h->clock32 + h->symtime32; fire = h->clock32 >= 1.0f; }
        if (!fire) continue;
        int imu;
        if (!h->f32mode) {
            h->clock64 -= 1.0;
            double mu = h->clock64 / h->symtime64;
            if (mu < 0.0) mu = 0.0;
            if (mu > 1.0 - 1e-6) mu = 1.0 - 1e-6;
            imu = (int)rint(mu * 128.0);
        } else {
            h->clock32 = h->clock32 - 1.0f;
            float mu = h->clock32 / h->symtime32;
            if (mu < 0.0f) mu = 0.0f;
            if (mu > (float)(1.0 - 1e-6)) mu = (float)(1.0 - 1e-6);
            imu = (int)rintf(mu * 128.0f);
        }
        if (imu > 128) imu = 128;
        const int pos = NT + i;
        cf curr = interp(h, (const float(*)[2])yb, pos, 0, imu);
        float curr_mag = hyp(h, curr);
        int dibit;
        if (h->first) {
            /* diff = complex128(curr) * complex128(0, -0); phase = float64 atan2 of signed zeros */
            double a = curr.re, b = curr.im;
            double re = a * 0.0 - b * (-0.0), im = a * (-0.0) + b * 0.0;
            double phase = signbit(re) ? (signbit(im) ? -M_PI : M_PI) : im;
            double expected;
            if (phase >= M_PI / 2) { dibit = 1; expected = 3 * M_PI / 4; }
            else if (phase >= 0) { dibit = 0; expected = M_PI / 4; }
            else if (phase >= -(M_PI / 2)) { dibit = 2; expected = -(M_PI / 4); }
            else { dibit = 3; expected = -(3 * M_PI / 4); }
            double pe = phase - expected;
            if (pe > M_PI) pe -= 2 * M_PI; else if (pe < -M_PI) pe += 2 * M_PI;
            h->freq_offset += 0.0005 * pe * (double)curr_mag;
            if (phases && count < cap) phases[count] = (float)phase;
        } else {
            cf prev = {h->prev_re, h->prev_im};
            float prev_mag = hyp(h, prev);
            cf d;
            if (curr_mag > 1e-6f && prev_mag > 1e-6f) {
                cf a = cdiv_real(curr, curr_mag), b = cdiv_real(prev, prev_mag);
                b.im = -b.im;
                d.re = a.re * b.re - a.im * b.im;
                d.im = a.re * b.im + a.im * b.re;
            } else {
                float bi = -prev.im;
                d.re = curr.re * prev.re - curr.im * bi;
                d.im = curr.re * bi + curr.im * prev.re;
            }
            float phase = angle_f32(h, d.im, d.re);
            float expected;
            if (phase >= HALF_PI_F) { dibit = 1; expected = (float)(3 * M_PI / 4); }
            else if (phase >= 0.0f) { dibit = 0; expected = (float)(M_PI / 4); }
            else if (phase >= NHALF_PI_F) { dibit = 2; expected = (float)(-(M_PI / 4)); }
            else { dibit = 3; expected = (float)(-(3 * M_PI / 4)); }
            float pe = phase - expected;
            if (pe > PI_F) pe = pe - (float)(2 * M_PI); else if (pe < -PI_F) pe = pe + (float)(2 * M_PI);
            float f = 0.0005f * pe;
            f = f * curr_mag;
            h->freq_offset += (double)f;
            if (phases && count < cap) phases[count] = phase;
        }
        if (h->freq_offset < -0.02) h->freq_offset = -0.02;
        if (h->freq_offset > 0.02) h->freq_offset = 0.02;
        if (count < cap) dibits[count] = (uint8_t)dibit;
        ++count;
        if (h->gardner) {
            cf mid = interp(h, (const float(*)[2])yb, pos, h->half_sps, imu);
            cf ps = interp(h, (const float(*)[2])yb, pos, h->full_sps, imu);
            float dr = curr.re - ps.re, di = curr.im - ps.im;
            float ted = dr * mid.re - di * (-mid.im);
            float step = 0.015f * ted;
            if (!h->f32mode) {
                h->clock32 = (float)h->clock64 + step;
                h->omega32 = (float)h->sps + 0.0f * ted;
                h->f32mode = 1;
            } else {
                h->clock32 = h->clock32 + step;
                h->omega32 = h->omega32 + 0.0f * ted;
            }
            h->symtime32 = 1.0f / h->omega32;
        }
        if (!h->f32mode) {
            while (h->clock64 >= 1.0) h->clock64 -= 1.0;
            while (h->clock64 < 0.0) h->clock64 += 1.0;
        } else {
            while (h->clock32 >= 1.0f) h->clock32 = h->clock32 - 1.0f;
            while (h->clock32 < 0.0f) h->clock32 = h->clock32 + 1.0f;
        }
        h->prev_re = curr.re;
        h->prev_im = curr.im;
        h->first = 0;
    }
    memcpy(h->hist, yb + n, sizeof(h->hist));
    free(yb);
    return count;
}
