/*
 * c4fm_ref.c -- CPU oracle (plain C) for the P25 C4FM demodulator, rows A9-A11.
 *
 * TEST INFRASTRUCTURE ONLY: built into oracle/_build/libc4fm_ref.so and loaded by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product never links it.
 *
 * Restates wavecapsdr/dsp/p25/c4fm.py (C4FMDemodulator.demodulate, :2528-2807) under the
 * oracle semantics "numpy 2.2.6 + scipy 1.15.3, numba absent" (SURVEY.md F10), sequentially,
 * one demodulator instance per object.  Precision follows the Python scalar types exactly:
 *   - baseband LPF and RRC: scipy.signal.lfilter(b_f32, 1.0, x, zi) takes its len(a)==1 branch
 *     (np.convolve) and, because `1.0` is a float64 0-d array, computes in FLOAT64; the RRC
 *     consumes the float64 LPF output; the result is cast to float32 (c4fm.py:2575-2593);
 *   - differential demodulator: float32, one rounding per operation, 8-tap interpolation
 *     accumulated in index order (c4fm.py:373-393, 401-409);
 *   - symbol clock `sample_point`: float64; its Python *type* (float vs np.float64) decides
 *     whether the linear interpolation / equalisation run in float32 ("phase A") or float64
 *     ("phase B") (c4fm.py:704-774, 2681-2688);
 *   - sync correlators, optimiser scores and corrections: float32 with weak Python-float
 *     operands rounded to float32; message re-slice: float64 (c4fm.py:416-644, 795-869).
 * Parity pin: tests/test_c4fm_oracle.py checks dibits bit-exact and soft symbols against the
 * goldens captured from the reference (tests/golden/c4fm.npz).
 *
 * atan2: the reference calls np.arctan2 on float32 scalars, which on AVX-512 hosts is an
 * SVML kernel and on others libm -- i.e. not reproducible across machines.  mode 0 uses
 * libm atan2f, mode 1 the portable polynomial shared with the HIP kernel
 * (wavecap-sdr_amd/csrc/wh_portable_math.h); the tests require both to give the reference's
 * dibits on the goldens.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (no FMA contraction, SSE2 float semantics).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../wavecap-sdr_amd/csrc/wh_portable_math.h"

#define BUF_LEN 65536
#define HALF_BUF 32768
#define TSDU_MESSAGE_DIBITS 340 /* c4fm.py:792 */

typedef struct c4fm_ref {
    double sps;
    int nl, nr;
    float *lpf, *rrc;          /* float32 taps */
    float taps[129][8];        /* MMSE interpolator table (c4fm.py:907-2202) */
    int atan_mode;
    /* FIR histories (equivalent to the carried zi of lfilter) */
    double *hx_i, *hx_q;       /* last nl-1 inputs */
    double *hy_i, *hy_q;       /* last nr-1 LPF outputs (float64) */
    /* differential demodulator (c4fm.py:275-395) */
    int interp_offset, overlap, fm_row;
    float *zh_i, *zh_q;        /* last `overlap` RRC outputs (float32) */
    /* equaliser (c4fm.py:212-272) */
    double pll, gain;
    int eq_initialized;
    /* sync detectors (c4fm.py:2268-2328) */
    float sync_sym[24];
    float det_buf[48], lag_buf[48];
    int det_ptr, lag_ptr;
    /* sync / timing state */
    int fine_sync, symbols_since_sync, sync_count;
    double max_fine_adj, lagging_offset;
    double sample_point;
    int sp_np64;               /* Python type of sample_point: 0 = float, 1 = np.float64 */
    float *buffer;             /* 65536 phase samples */
    int buffer_pointer;
    /* scratch */
    int cap_n;
    double *y_i, *y_q;
    float *z_i, *z_q, *phases;
    int32_t *sym_idx;
} c4fm_ref;

static float f_atan2(const c4fm_ref *s, float y, float x) { return s->atan_mode ? whm_atan2f(y, x) : atan2f(y, x); }

static void sync_symbols_init(float *sym) {
    const uint64_t pattern = 0x5575F5FF77FFULL; /* c4fm.py:2279 */
    for (int i = 0; i < 24; ++i) {
        int d = (int)((pattern >> ((23 - i) * 2)) & 3);
        sym[i] = d == 1 ? 3.0f : -3.0f;
    }
}

void c4fm_ref_reset(c4fm_ref *s) { /* c4fm.py:2505-2521 */
    memset(s->hx_i, 0, sizeof(double) * (s->nl - 1));
    memset(s->hx_q, 0, sizeof(double) * (s->nl - 1));
    memset(s->hy_i, 0, sizeof(double) * (s->nr - 1));
    memset(s->hy_q, 0, sizeof(double) * (s->nr - 1));
    memset(s->zh_i, 0, sizeof(float) * s->overlap);
    memset(s->zh_q, 0, sizeof(float) * s->overlap);
    s->pll = 0.0;
    s->gain = 1.219;
    s->eq_initialized = 0;
    memset(s->det_buf, 0, sizeof(s->det_buf));
    memset(s->lag_buf, 0, sizeof(s->lag_buf));
    s->det_ptr = s->lag_ptr = 0;
    s->sample_point = s->sps;
    s->sp_np64 = 0;
    memset(s->buffer, 0, sizeof(float) * BUF_LEN);
    s->buffer_pointer = 0;
    s->sync_count = 0;
    s->fine_sync = 0;
    s->symbols_since_sync = 0;
}

c4fm_ref *c4fm_ref_create(double sps, const float *lpf, int nl, const float *rrc, int nr, const float *interp_taps,
                          int atan_mode) {
    c4fm_ref *s = (c4fm_ref *)calloc(1, sizeof(c4fm_ref));
    s->sps = sps;
    s->nl = nl;
    s->nr = nr;
    s->lpf = (float *)malloc(sizeof(float) * nl);
    s->rrc = (float *)malloc(sizeof(float) * nr);
    memcpy(s->lpf, lpf, sizeof(float) * nl);
    memcpy(s->rrc, rrc, sizeof(float) * nr);
    memcpy(s->taps, interp_taps, sizeof(s->taps));
    s->atan_mode = atan_mode;
    s->hx_i = (double *)calloc(nl, sizeof(double));
    s->hx_q = (double *)calloc(nl, sizeof(double));
    s->hy_i = (double *)calloc(nr, sizeof(double));
    s->hy_q = (double *)calloc(nr, sizeof(double));
    /* c4fm.py:299-313 */
    double mu = fmod(sps, 1.0);
    int fl = (int)floor(sps);
    s->interp_offset = fl - 4 > 0 ? fl - 4 : 0;
    s->overlap = fl + 4;
    {   /* _Interpolator.filter row selection, c4fm.py:2226-2230 */
        double mu_inv = 1.0 - mu;
        int row = (int)(mu_inv * 128.0 + 0.5);
        s->fm_row = row < 0 ? 0 : (row > 128 ? 128 : row);
    }
    s->zh_i = (float *)calloc(s->overlap, sizeof(float));
    s->zh_q = (float *)calloc(s->overlap, sizeof(float));
    sync_symbols_init(s->sync_sym);
    s->max_fine_adj = sps * 0.2;
    s->lagging_offset = sps / 2.0;
    s->buffer = (float *)calloc(BUF_LEN, sizeof(float));
    c4fm_ref_reset(s);
    return s;
}

void c4fm_ref_destroy(c4fm_ref *s) {
    if (!s) return;
    free(s->lpf); free(s->rrc); free(s->hx_i); free(s->hx_q); free(s->hy_i); free(s->hy_q);
    free(s->zh_i); free(s->zh_q); free(s->buffer);
    free(s->y_i); free(s->y_q); free(s->z_i); free(s->z_q); free(s->phases); free(s->sym_idx);
    free(s);
}

void c4fm_ref_get_state(const c4fm_ref *s, double *out) {
    out[0] = s->sync_count;
    out[1] = s->fine_sync;
    out[2] = s->pll;
    out[3] = s->gain;
    out[4] = s->sample_point;
    out[5] = s->sp_np64;
    out[6] = s->buffer_pointer;
}

static void ensure_scratch(c4fm_ref *s, int n) {
    if (n <= s->cap_n) return;
    free(s->y_i); free(s->y_q); free(s->z_i); free(s->z_q); free(s->phases); free(s->sym_idx);
    s->cap_n = n;
    s->y_i = (double *)malloc(sizeof(double) * (n + s->nr));
    s->y_q = (double *)malloc(sizeof(double) * (n + s->nr));
    s->z_i = (float *)malloc(sizeof(float) * (n + s->overlap));
    s->z_q = (float *)malloc(sizeof(float) * (n + s->overlap));
    s->phases = (float *)malloc(sizeof(float) * n);
    s->sym_idx = (int32_t *)malloc(sizeof(int32_t) * (n / 4 + 16));
}

/* float64 FIR over history + new block: out[t] = sum_k taps[k] * in[t-k] */
static void fir_f64(const float *taps, int nt, const double *hist /* nt-1 */, const double *x, int n, double *out) {
    for (int t = 0; t < n; ++t) {
        double acc = 0.0;
        for (int k = 0; k < nt; ++k) {
            int idx = t - k;
            double v = idx >= 0 ? x[idx] : hist[nt - 1 + idx];
            acc += (double)taps[k] * v;
        }
        out[t] = acc;
    }
}

static void update_hist(double *hist, int hl, const double *x, int n) {
    if (hl <= 0) return;
    if (n >= hl) memcpy(hist, x + n - hl, sizeof(double) * hl);
    else {
        memmove(hist, hist + n, sizeof(double) * (hl - n));
        memcpy(hist + hl - n, x, sizeof(double) * n);
    }
}

/* 8-tap interpolation, float32 sequential (c4fm.py:401-409) */
static float interp8(const float *p, const float *taps) {
    float r = p[0] * taps[0];  /* 0.0 + x == x */
    for (int i = 1; i < 8; ++i) r = r + p[i] * taps[i];
    return r;
}

/* dibit decision on a float32 soft value with weak float64 boundaries == float32 compare */
static int slice_f32(float v) {
    const float B = (float)1.5707963267948966;
    if (v >= B) return 1;
    if (v >= 0.0f) return 0;
    if (v >= -B) return 2;
    return 3;
}
static int slice_f64(double v) {
    const double B = 1.5707963267948966;
    if (v >= B) return 1;
    if (v >= 0.0) return 0;
    if (v >= -B) return 2;
    return 3;
}

/* _SoftSyncDetector.process (c4fm.py:2306-2328): push then correlate, float32 sequential */
static float detector_process(const float *sync, float *buf, int *ptr, float v) {
    buf[*ptr] = v;
    buf[*ptr + 24] = v;
    *ptr = (*ptr + 1) % 24;
    float score = sync[0] * buf[*ptr];
    for (int i = 1; i < 24; ++i) score = score + sync[i] * buf[*ptr + i];
    return score;
}

/* _timing_score_jit (c4fm.py:416-463).  Returns float32 score (0 if no term was valid). */
static float timing_score(const c4fm_ref *s, double offset, float pll32, float gain32) {
    const int max_offset = BUF_LEN - 8;
    float score = 0.0f;
    int first = 1;
    double ptr = offset - (23.0 * s->sps);
    for (int i = 0; i < 24; ++i) {
        int buf_idx = (int)ptr;
        int io = buf_idx - 3;
        if (io >= 0 && io <= max_offset) {
            double mu = ptr - (double)buf_idx;
            double mu_inv = 1.0 - mu;
            int row = (int)(mu_inv * 128.0 + 0.5);
            if (row < 0) row = 0;
            if (row > 128) row = 128;
            float v = interp8(s->buffer + io, s->taps[row]);
            float soft = (v + pll32) * gain32;
            float term = soft * s->sync_sym[i];
            if (first) { score = term; first = 0; } else score = score + term;
        }
        ptr += s->sps;
    }
    return score;
}

/* _timing_correction_jit (c4fm.py:466-540). pll_corr may be a float64 when clipped. */
static void timing_correction(const c4fm_ref *s, double offset, float pll32, float gain32, double *pll_corr,
                              int *pll_is_f64, float *gain_corr) {
    const int max_offset = BUF_LEN - 8;
    float bp = 0.0f, bm = 0.0f, ga = 0.0f;
    int pc = 0, mc = 0, any = 0;
    double ptr = offset - (23.0 * s->sps);
    for (int i = 0; i < 24; ++i) {
        int buf_idx = (int)ptr;
        int io = buf_idx - 3;
        if (io >= 0 && io <= max_offset) {
            double mu = ptr - (double)buf_idx;
            double mu_inv = 1.0 - mu;
            int row = (int)(mu_inv * 128.0 + 0.5);
            if (row < 0) row = 0;
            if (row > 128) row = 128;
            float v = interp8(s->buffer + io, s->taps[row]);
            float soft = (v + pll32) * gain32;
            float ideal = s->sync_sym[i];
            if (ideal > 0.0f) { bp = bp + (soft - ideal); pc++; }
            else { bm = bm + (soft - ideal); mc++; }
            ga = ga + (fabsf(ideal) - fabsf(soft));
            any = 1;
        }
        ptr += s->sps;
    }
    if (pc > 0) bp = bp / (float)(-pc);
    if (mc > 0) bm = bm / (float)(-mc);
    float pc32 = (bp + bm) / 2.0f;
    const float HP32 = (float)1.5707963267948966;
    *pll_is_f64 = 0;
    *pll_corr = (double)pc32;
    if (pc32 < -HP32) { *pll_corr = -1.5707963267948966; *pll_is_f64 = 1; }
    else if (pc32 > HP32) { *pll_corr = 1.5707963267948966; *pll_is_f64 = 1; }
    *gain_corr = any ? ga / (float)(24.0 * 2.356194490192345) : 0.0f;
}

/* _timing_optimize_jit (c4fm.py:543-644) */
static void timing_optimize(const c4fm_ref *s, double offset, int fine, double *adj_out, float *score_out,
                            double *pll_corr, int *pll_is_f64, float *gain_corr) {
    const double sps = s->sps;
    const float pll32 = (float)s->pll, gain32 = (float)s->gain;
    double step, step_min = sps / 200.0, max_adj;
    if (fine) { step = sps / 16.0; max_adj = sps; }
    else { step = sps / 8.0; max_adj = sps / 2.0; }
    double adj = 0.0;
    float sc = timing_score(s, offset, pll32, gain32);
    float sl = timing_score(s, offset - step, pll32, gain32);
    float sr = timing_score(s, offset + step, pll32, gain32);
    while (step > step_min && fabs(adj) <= max_adj) {
        if (sl > sr && sl > sc) {
            adj -= step;
            sr = sc;
            sc = sl;
            sl = timing_score(s, offset + adj - step, pll32, gain32);
        } else if (sr > sl && sr > sc) {
            adj += step;
            sl = sc;
            sc = sr;
            sr = timing_score(s, offset + adj + step, pll32, gain32);
        } else {
            step *= 0.5;
            if (step > step_min) {
                sl = timing_score(s, offset + adj - step, pll32, gain32);
                sr = timing_score(s, offset + adj + step, pll32, gain32);
            }
        }
    }
    timing_correction(s, offset + adj, pll32, gain32, pll_corr, pll_is_f64, gain_corr);
    *adj_out = adj;
    *score_out = sc;
}

/* _Equalizer.apply_correction (c4fm.py:260-272) */
static void apply_correction(c4fm_ref *s, double pll_adj, int pll_is_f64, float gain_adj) {
    const double MAXPLL = 3.141592653589793 / 3.0;
    if (!pll_is_f64) {
        float p32 = (float)s->pll, a32 = (float)pll_adj, r;
        if (s->eq_initialized) r = p32 + a32 * (float)0.15;
        else r = p32 + a32;
        float lo = (float)(-MAXPLL), hi = (float)MAXPLL;   /* np.clip on a float32 scalar: weak bounds */
        if (r < lo) r = lo;
        if (r > hi) r = hi;
        s->pll = (double)r;
    } else {
        double r = s->eq_initialized ? s->pll + pll_adj * 0.15 : s->pll + pll_adj;
        if (r < -MAXPLL) r = -MAXPLL;
        if (r > MAXPLL) r = MAXPLL;
        s->pll = r;
    }
    {
        float g32 = (float)s->gain, r;
        if (s->eq_initialized) r = g32 + gain_adj * (float)0.15;
        else r = g32 + gain_adj;
        if (r < 1.0f) r = 1.0f;
        if (r > 1.25f) r = 1.25f;
        s->gain = (double)r;
    }
    s->eq_initialized = 1;
}

int c4fm_ref_demodulate(c4fm_ref *s, const float *iq, int n, uint8_t *dibits, float *soft, int cap) {
    if (n <= 0) return 0;
    ensure_scratch(s, n);
    const double sps = s->sps;

    /* ---- A9: baseband LPF + RRC in float64, cast to float32 ---------------------------- */
    {
        double *xi = (double *)malloc(sizeof(double) * n), *xq = (double *)malloc(sizeof(double) * n);
        for (int t = 0; t < n; ++t) { xi[t] = (double)iq[2 * t]; xq[t] = (double)iq[2 * t + 1]; }
        fir_f64(s->lpf, s->nl, s->hx_i, xi, n, s->y_i);
        fir_f64(s->lpf, s->nl, s->hx_q, xq, n, s->y_q);
        update_hist(s->hx_i, s->nl - 1, xi, n);
        update_hist(s->hx_q, s->nl - 1, xq, n);
        double *zi = xi, *zq = xq;  /* reuse */
        fir_f64(s->rrc, s->nr, s->hy_i, s->y_i, n, zi);
        fir_f64(s->rrc, s->nr, s->hy_q, s->y_q, n, zq);
        update_hist(s->hy_i, s->nr - 1, s->y_i, n);
        update_hist(s->hy_q, s->nr - 1, s->y_q, n);
        /* demod buffer = [last `overlap` samples] + new (c4fm.py:339-363) */
        memcpy(s->z_i, s->zh_i, sizeof(float) * s->overlap);
        memcpy(s->z_q, s->zh_q, sizeof(float) * s->overlap);
        for (int t = 0; t < n; ++t) { s->z_i[s->overlap + t] = (float)zi[t]; s->z_q[s->overlap + t] = (float)zq[t]; }
        free(xi); free(xq);
    }
    /* ---- A10: symbol-spaced differential demodulation (c4fm.py:365-395) ------------------ */
    {
        const float *taps = s->taps[s->fm_row];
        const int blen = n + s->overlap;
        for (int x = 0; x < n; ++x) {
            float i_prev = s->z_i[x];
            float q_prev_conj = -s->z_q[x];
            int off = s->interp_offset + x;
            float i_cur, q_cur;
            if (off >= 0 && off + 8 <= blen) {
                i_cur = interp8(s->z_i + off, taps);
                q_cur = interp8(s->z_q + off, taps);
            } else {
                int idx = off + 4 < blen - 1 ? off + 4 : blen - 1;
                i_cur = s->z_i[idx];
                q_cur = s->z_q[idx];
            }
            float diff_i = (i_prev * i_cur) - (q_prev_conj * q_cur);
            float diff_q = (i_prev * q_cur) + (i_cur * q_prev_conj);
            s->phases[x] = f_atan2(s, diff_q, diff_i);
        }
        /* carry the last `overlap` samples of the buffer */
        memmove(s->zh_i, s->z_i + n, sizeof(float) * s->overlap);
        memmove(s->zh_q, s->z_q + n, sizeof(float) * s->overlap);
    }
    /* ---- A11a: fixed-rate symbol recovery (c4fm.py:649-783) ------------------------------ */
    int count = 0;
    {
        const float pll32 = (float)s->pll, gain32 = (float)s->gain;
        double sp = s->sample_point;
        int bp = s->buffer_pointer;
        float *buf = s->buffer;
        for (int x = 0; x < n; ++x) {
            bp += 1;
            sp -= 1.0;
            if (bp >= BUF_LEN - 1) {
                memmove(buf, buf + HALF_BUF, sizeof(float) * HALF_BUF);
                memset(buf + HALF_BUF, 0, sizeof(float) * HALF_BUF);
                bp -= HALF_BUF;
                for (int j = 0; j < count; ++j) {
                    s->sym_idx[j] -= HALF_BUF;
                    if (s->sym_idx[j] < 0) s->sym_idx[j] = -1;
                }
            }
            buf[bp] = s->phases[x];
            if (sp < 1.0) {
                int idx = bp;
                double mu = 1.0 - sp;
                if (idx - 1 >= 0 && idx < BUF_LEN && count < cap) {
                    float x1 = buf[idx - 1], x2 = buf[idx];
                    int dib;
                    float sn;
                    if (mu < 0.0 || mu > 1.0 || !s->sp_np64) {
                        /* float32 path: clamp cases always; interpolation when mu is a Python float */
                        float v;
                        if (mu < 0.0) v = x1;
                        else if (mu > 1.0) v = x2;
                        else v = x1 + (x2 - x1) * (float)mu;
                        float sr = (v + pll32) * gain32;
                        dib = slice_f32(sr);
                        sn = sr * (float)1.2732395447351628;
                    } else {
                        /* mu is np.float64: (x2-x1) in float32, the rest in float64 */
                        double v = (double)x1 + (double)(x2 - x1) * mu;
                        double sr = (v + s->pll) * s->gain;
                        dib = slice_f64(sr);
                        sn = (float)(sr * 1.2732395447351628);
                    }
                    dibits[count] = (uint8_t)dib;
                    soft[count] = sn;
                    s->sym_idx[count] = idx;
                    count++;
                }
                sp += sps;
            }
        }
        s->buffer_pointer = bp;
        s->sample_point = sp;
    }
    /* ---- A11b: sync detection, timing optimisation, equaliser, message re-slice ---------- */
    {
        float *buf = s->buffer;
        for (int k = 0; k < count; ++k) {
            s->symbols_since_sync += 1;
            float score_pri = detector_process(s->sync_sym, s->det_buf, &s->det_ptr, soft[k]);
            float score;
            double additional = 0.0;
            if (s->fine_sync || s->sym_idx[k] < 0) {
                score = score_pri;
            } else {
                float score_lag = 0.0f;
                int ilo = (int)s->lagging_offset;
                int lag_pos = s->sym_idx[k] - ilo;
                if (lag_pos >= 4) {
                    double lag_mu = 1.0 - (s->lagging_offset - (double)ilo);
                    int lag_off = lag_pos - 4;
                    if (lag_off >= 0 && lag_pos < BUF_LEN) {
                        /* _Equalizer.get_equalized_symbol (c4fm.py:234-258), float32 */
                        float v;
                        if (lag_off + 1 < BUF_LEN) {
                            float x1 = buf[lag_off], x2 = buf[lag_off + 1];
                            if (lag_mu < 0.0) v = x1;
                            else if (lag_mu > 1.0) v = x2;
                            else v = x1 + ((x2 - x1) * (float)lag_mu);
                        } else {
                            v = buf[lag_off];
                        }
                        float sl = (v + (float)s->pll) * (float)s->gain;
                        float sln = sl * (float)(4.0 / 3.141592653589793);
                        score_lag = detector_process(s->sync_sym, s->lag_buf, &s->lag_ptr, sln);
                    }
                }
                if (score_lag > score_pri && score_lag >= 100.0f) {
                    score = score_lag;
                    additional = -s->lagging_offset;
                } else {
                    score = score_pri;
                }
            }
            if (score >= 100.0f) {
                if (s->sym_idx[k] < 0) continue;
                double adj, pll_corr;
                float opt_score, gain_corr;
                int pll_is_f64;
                double offset = ((double)s->sym_idx[k] + 0.5) + additional;
                timing_optimize(s, offset, s->fine_sync, &adj, &opt_score, &pll_corr, &pll_is_f64, &gain_corr);
                if (opt_score >= 100.0f) {
                    if (s->fine_sync) {
                        if (adj < -s->max_fine_adj) adj = -s->max_fine_adj;
                        if (adj > s->max_fine_adj) adj = s->max_fine_adj;
                        s->sp_np64 = 1;   /* np.clip(...) returns np.float64 (c4fm.py:2682-2688) */
                    }
                    s->sample_point += adj + additional;
                    apply_correction(s, pll_corr, pll_is_f64, gain_corr);
                    s->sync_count += 1;
                    s->fine_sync = 1;
                    s->symbols_since_sync = 0;
                    /* message re-slice (c4fm.py:2703-2746, 795-869) */
                    double sync_start = (((double)s->sym_idx[k] - 23.0 * sps) + adj) + additional;
                    int remaining = count - (k + 1);
                    int nres = remaining < TSDU_MESSAGE_DIBITS ? remaining : TSDU_MESSAGE_DIBITS;
                    double msg_start = sync_start + 24.0 * sps;
                    for (int i = 0; i < nres; ++i) {
                        double pos = msg_start + (double)i * sps;
                        int idx = (int)pos;
                        double mu = pos - (double)idx;
                        int dib;
                        float sn;
                        int f32path = 0;
                        float v32 = 0.0f;
                        double v64 = 0.0;
                        if (idx >= 0 && idx + 1 < BUF_LEN) {
                            float x1 = buf[idx], x2 = buf[idx + 1];
                            if (mu < 0.0) { v32 = x1; f32path = 1; }
                            else if (mu > 1.0) { v32 = x2; f32path = 1; }
                            else v64 = (double)x1 + (double)(x2 - x1) * mu;
                        } else {
                            int c = idx < BUF_LEN - 1 ? idx : BUF_LEN - 1;
                            if (c < 0) c = 0;
                            v32 = buf[c];
                            f32path = 1;
                        }
                        if (f32path) {
                            float sr = (v32 + (float)s->pll) * (float)s->gain;
                            sn = sr * (float)1.2732395447351628;
                            dib = slice_f32(sr);
                        } else {
                            double sr = (v64 + s->pll) * s->gain;
                            sn = (float)(sr * 1.2732395447351628);
                            dib = slice_f64(sr);
                        }
                        dibits[k + 1 + i] = (uint8_t)dib;
                        soft[k + 1 + i] = sn;
                    }
                }
            }
            if (s->symbols_since_sync > 3600) {
                s->fine_sync = 0;
                s->symbols_since_sync = 0;
            }
        }
    }
    return count;
}
