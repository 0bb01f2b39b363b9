// Row A12 for gfx950: P25 Phase-2 CQPSK receive chain (reference dsp/p25/cqpsk.py:199-350: RRC
// matched filter, Costas loop :84-196, differential decode :308-350) with Mueller-Muller timing
// recovery (dsp/p25/symbol_timing.py:214-380), and the standalone Gardner TED
// (symbol_timing.py:60-211).  The reference computes all of it in Python float / complex128.
//   k_cq_rrc   matched filter, complex128, one thread per output of the full convolution
//              (the reference re-seeds lfilter with zi = state * iq[0] on every call -- kept);
//   k_cq_seq   ONE LANE PER CHANNEL: the Costas PLL feeds back every sample, so the loop is
//              sequential in time; a bank of channels fills the wavefronts.  Float64 with
//              explicit *_rn ops in the order of oracle/cqpsk_ref.c (no FMA contraction);
//   k_gardner  Gardner TED bank, one lane per channel, float64, bit-exact vs the oracle.
#include "wh_common.h"
#include <cmath>
#include <memory>
#include <vector>

using namespace wh;

#define DM(a, b) __dmul_rn((a), (b))
#define DA(a, b) __dadd_rn((a), (b))
#define DS(a, b) __dsub_rn((a), (b))
#define DD(a, b) __ddiv_rn((a), (b))

namespace {

constexpr double PI_D = 3.141592653589793;

struct CqState {
    double c_phase, c_freq, t_phase, t_integ, prev_phase;
    double2 buf[4];
    double2 prev_sym, prev_dec;
    int buf_idx, pad;
};

struct CqArgs {
    const float2 *iq;
    size_t iq_stride;
    int n, n_max, C, L;
    const double *taps;
    const double2 *state_in;   // [C][L-1]
    double2 *state_out;        // [C][L-1]
    double2 *full;             // [C][n_max + L-1]
    CqState *st;
    double sps, c_kp, c_ki, c_maxf, t_kp, t_ki, t_maxdev;
    uint8_t *dibits;
    double2 *symbols;          // optional [C][cap]
    size_t cap;
    int *counts;
};

__global__ __launch_bounds__(256) void k_cq_rrc(CqArgs a) {
    extern __shared__ __attribute__((aligned(16))) double tp[];
    const int c = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    const int H = a.L - 1;
    for (int i = threadIdx.x; i < a.L; i += 256) tp[i] = a.taps[i];
    __syncthreads();
    if (t >= a.n + H) return;
    const float2 *x = a.iq + (size_t)c * a.iq_stride;
    int k0 = t - (a.n - 1) > 0 ? t - (a.n - 1) : 0;
    int k1 = t < H ? t : H;
    double ar = 0.0, ai = 0.0;
    for (int k = k0; k <= k1; ++k) {
        float2 v = x[t - k];
        ar = fma(tp[k], (double)v.x, ar);
        ai = fma(tp[k], (double)v.y, ai);
    }
    if (t < H) {  // zi = state * iq[0]
        double2 s = a.state_in[(size_t)c * H + t];
        double xr = (double)x[0].x, xi = (double)x[0].y;
        ar += s.x * xr - s.y * xi;
        ai += s.x * xi + s.y * xr;
    }
    if (t < a.n) a.full[(size_t)c * (a.n_max + H) + t] = make_double2(ar, ai);
    if (t >= a.n) a.state_out[(size_t)c * H + (t - a.n)] = make_double2(ar, ai);
    // (when n < H the entries t in [n, H) are both "zi-added" and part of the new state, as in scipy)
}

__device__ __forceinline__ double interp1(double v0, double v1, double v2, double v3, double mu) {
    double c0 = v1;
    double c1 = DD(DS(v2, v0), 2.0);
    double c2 = DS(DA(DS(v0, DD(DM(5.0, v1), 2.0)), DM(2.0, v2)), DD(v3, 2.0));
    double c3 = DA(DD(DS(v3, v0), 2.0), DD(DM(3.0, DS(v1, v2)), 2.0));
    return DA(c0, DM(mu, DA(c1, DM(mu, DA(c2, DM(mu, c3))))));
}

__global__ __launch_bounds__(64) void k_cq_seq(CqArgs a) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= a.C) return;
    CqState s = a.st[c];
    const double2 *full = a.full + (size_t)c * (a.n_max + a.L - 1);
    uint8_t *dib = a.dibits + (size_t)c * a.cap;
    double2 *sym = a.symbols ? a.symbols + (size_t)c * a.cap : nullptr;
    const double q = PI_D / 4;
    const double inv = 1.4142135623730951;
    // the 4-entry circular buffer lives in registers: keep it rotated so that b3 is the newest
    double2 b0 = s.buf[(s.buf_idx + 1) & 3], b1 = s.buf[(s.buf_idx + 2) & 3], b2 = s.buf[(s.buf_idx + 3) & 3],
            b3 = s.buf[s.buf_idx & 3];
    int count = 0;
    for (int t = 0; t < a.n; ++t) {
        double2 x = full[t];
        double sn, cs;
        sincos(s.c_phase, &sn, &cs);
        double cr = DS(DM(x.x, cs), DM(x.y, -sn));
        double ci = DA(DM(x.x, -sn), DM(x.y, cs));
        double ph = atan2(ci, cr);
        double ideal = DM(rint(DD(ph, q)), q);
        double err = DS(ph, ideal);
        while (err > PI_D) err = DS(err, 2 * PI_D);
        while (err < -PI_D) err = DA(err, 2 * PI_D);
        s.c_freq = DA(s.c_freq, DM(a.c_ki, err));
        if (s.c_freq < -a.c_maxf) s.c_freq = -a.c_maxf;
        if (s.c_freq > a.c_maxf) s.c_freq = a.c_maxf;
        s.c_phase = DA(s.c_phase, DA(DM(a.c_kp, err), s.c_freq));
        while (s.c_phase > PI_D) s.c_phase = DS(s.c_phase, 2 * PI_D);
        while (s.c_phase < -PI_D) s.c_phase = DA(s.c_phase, 2 * PI_D);
        b0 = b1; b1 = b2; b2 = b3; b3 = make_double2(cr, ci);
        s.t_phase = DA(s.t_phase, 1.0);
        if (s.t_phase >= a.sps) {
            s.t_phase = DS(s.t_phase, a.sps);
            double mu = DD(s.t_phase, a.sps);
            double sr = interp1(b0.x, b1.x, b2.x, b3.x, mu);
            double si = interp1(b0.y, b1.y, b2.y, b3.y, mu);
            int best = 0;
            double bd = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double kr = (k == 0 || k == 3) ? 1.0 : -1.0, ki = (k < 2) ? 1.0 : -1.0;
                double d = hypot(DS(DD(kr, inv), sr), DS(DD(ki, inv), si));
                if (k == 0 || d < bd) { bd = d; best = k; }
            }
            double dr = DD((best == 0 || best == 3) ? 1.0 : -1.0, inv), di = DD(best < 2 ? 1.0 : -1.0, inv);
            double e1 = DS(DM(s.prev_dec.x, sr), DM(-s.prev_dec.y, si));
            double e2 = DS(DM(dr, s.prev_sym.x), DM(-di, s.prev_sym.y));
            double e = DS(e1, e2);
            s.t_integ = DA(s.t_integ, DM(a.t_ki, e));
            if (s.t_integ < -a.t_maxdev) s.t_integ = -a.t_maxdev;
            if (s.t_integ > a.t_maxdev) s.t_integ = a.t_maxdev;
            s.t_phase = DA(s.t_phase, DA(DM(a.t_kp, e), s.t_integ));
            s.prev_sym = make_double2(sr, si);
            s.prev_dec = make_double2(dr, di);
            double p = atan2(si, sr);
            double dp = DS(p, s.prev_phase);
            while (dp > PI_D) dp = DS(dp, 2 * PI_D);
            while (dp < -PI_D) dp = DA(dp, 2 * PI_D);
            long long idx = (long long)rint(DD(DA(dp, PI_D), q));
            idx = ((idx % 8) + 8) % 8;
            if ((size_t)count < a.cap) {
                dib[count] = (uint8_t)(idx >> 1);
                if (sym) sym[count] = make_double2(sr, si);
                count++;
            }
            s.prev_phase = p;
        }
    }
    s.buf[0] = b0; s.buf[1] = b1; s.buf[2] = b2; s.buf[3] = b3;
    s.buf_idx = 3;
    a.st[c] = s;
    a.counts[c] = count;
}

struct GState {
    double phase, integ, prev_symbol, prev_mid, b0, b1, b2, b3;
};

__global__ __launch_bounds__(64) void k_gardner(const float *x, size_t stride, int n, int C, GState *st, double sps,
                                                double kp, double ki, double *symbols, double *errors, size_t cap,
                                                int *counts) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    GState g = st[c];
    const float *xc = x + (size_t)c * stride;
    double *so = symbols + (size_t)c * cap, *eo = errors + (size_t)c * cap;
    const double maxdev = sps / 4;
    int count = 0;
    for (int t = 0; t < n; ++t) {
        g.b0 = g.b1; g.b1 = g.b2; g.b2 = g.b3; g.b3 = (double)xc[t];
        g.phase = DA(g.phase, 1.0);
        if (g.phase >= sps) {
            g.phase = DS(g.phase, sps);
            double mu = DD(g.phase, sps);
            double cur = interp1(g.b0, g.b1, g.b2, g.b3, mu);
            double mid_phase = DA(g.phase, DD(sps, 2.0));
            double mid = g.prev_mid;
            if (mid_phase >= 1.0) mid = interp1(g.b0, g.b1, g.b2, g.b3, DS(mid_phase, (double)(long long)mid_phase));
            double e = DM(mid, DS(g.prev_symbol, cur));
            g.integ = DA(g.integ, DM(ki, e));
            if (g.integ < -maxdev) g.integ = -maxdev;
            if (g.integ > maxdev) g.integ = maxdev;
            g.phase = DA(g.phase, DA(DM(kp, e), g.integ));
            if ((size_t)count < cap) { so[count] = cur; eo[count] = e; count++; }
            g.prev_symbol = cur;
            g.prev_mid = mid;
        }
    }
    st[c] = g;
    counts[c] = count;
}

}  // namespace

struct wh_cqpsk_bank {
    int C, n_max, L;
    double sps, c_kp, c_ki, c_maxf, t_kp, t_ki;
    std::vector<double> zi0;
    double *d_taps = nullptr;
    double2 *d_state[2] = {nullptr, nullptr}, *d_full = nullptr;
    CqState *d_st = nullptr;
    int cur = 0;
};

static int cq_reset(wh_cqpsk_bank *b, hipStream_t st) {
    const int H = b->L - 1;
    std::vector<double2> z((size_t)b->C * H);
    for (int c = 0; c < b->C; ++c)
        for (int i = 0; i < H; ++i) z[(size_t)c * H + i] = make_double2(b->zi0[i], 0.0);
    WH_HIP(hipMemcpyAsync(b->d_state[0], z.data(), z.size() * sizeof(double2), hipMemcpyHostToDevice, st));
    WH_HIP(hipMemsetAsync(b->d_st, 0, (size_t)b->C * sizeof(CqState), st));
    WH_HIP(hipStreamSynchronize(st));
    b->cur = 0;
    return WH_OK;
}

extern "C" int wh_cqpsk_bank_create(wh_cqpsk_bank **out, int C, double sps, const float *h_rrc, int ntaps,
                                    const double *h_zi, double c_kp, double c_ki, double c_maxf, double t_kp,
                                    double t_ki, int n_max) {
    if (!out || !h_rrc || !h_zi || C < 1 || ntaps < 2 || ntaps > 4096 || !(sps > 1.0) || n_max < 1)
        return set_err(WH_E_ARG, "wh_cqpsk_bank_create: bad arguments");
    wh_cqpsk_bank *b = new wh_cqpsk_bank();
    std::unique_ptr<wh_cqpsk_bank, void (*)(wh_cqpsk_bank *)> guard(b, wh_cqpsk_bank_destroy);  // frees partial state on early return
    b->C = C; b->n_max = n_max; b->L = ntaps; b->sps = sps;
    b->c_kp = c_kp; b->c_ki = c_ki; b->c_maxf = c_maxf; b->t_kp = t_kp; b->t_ki = t_ki;
    b->zi0.assign(h_zi, h_zi + ntaps - 1);
    std::vector<double> taps(ntaps);
    for (int i = 0; i < ntaps; ++i) taps[i] = (double)h_rrc[i];
    const int H = ntaps - 1;
    WH_HIP(hipMalloc(&b->d_taps, ntaps * sizeof(double)));
    WH_HIP(hipMemcpy(b->d_taps, taps.data(), ntaps * sizeof(double), hipMemcpyHostToDevice));
    for (int i = 0; i < 2; ++i) WH_HIP(hipMalloc(&b->d_state[i], (size_t)C * H * sizeof(double2)));
    WH_HIP(hipMalloc(&b->d_full, (size_t)C * (n_max + H) * sizeof(double2)));
    WH_HIP(hipMalloc(&b->d_st, (size_t)C * sizeof(CqState)));
    int rc = cq_reset(b, nullptr);
    if (rc != WH_OK) return rc;
    *out = guard.release();
    return WH_OK;
}

extern "C" void wh_cqpsk_bank_destroy(wh_cqpsk_bank *b) {
    if (!b) return;
    (void)hipFree(b->d_taps); (void)hipFree(b->d_state[0]); (void)hipFree(b->d_state[1]); (void)hipFree(b->d_full);
    (void)hipFree(b->d_st);
    delete b;
}

extern "C" int wh_cqpsk_bank_reset(wh_cqpsk_bank *b, void *stream) {
    if (!b) return set_err(WH_E_ARG, "wh_cqpsk_bank_reset: null handle");
    return cq_reset(b, as_stream(stream));
}

extern "C" int wh_cqpsk_bank_run(wh_cqpsk_bank *b, const float *d_iq, size_t n, size_t iq_stride, uint8_t *d_dibits,
                                 double *d_symbols, size_t cap, int32_t *d_counts, void *stream) {
    if (!b || !d_counts) return set_err(WH_E_ARG, "wh_cqpsk_bank_run: null handle/counts");
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        WH_HIP(hipMemsetAsync(d_counts, 0, (size_t)b->C * sizeof(int32_t), st));
        return WH_OK;
    }
    if (!d_iq || !d_dibits) return set_err(WH_E_ARG, "wh_cqpsk_bank_run: null buffer");
    if (n > (size_t)b->n_max || iq_stride < n) return set_err(WH_E_ARG, "wh_cqpsk_bank_run: n too large / bad stride");
    if (cap < (size_t)(n / (size_t)(b->sps * 0.5)) + 2) return set_err(WH_E_ARG, "wh_cqpsk_bank_run: cap too small");
    CqArgs a;
    a.iq = reinterpret_cast<const float2 *>(d_iq);
    a.iq_stride = iq_stride;
    a.n = (int)n; a.n_max = b->n_max; a.C = b->C; a.L = b->L;
    a.taps = b->d_taps;
    a.state_in = b->d_state[b->cur];
    a.state_out = b->d_state[b->cur ^ 1];
    a.full = b->d_full;
    a.st = b->d_st;
    a.sps = b->sps; a.c_kp = b->c_kp; a.c_ki = b->c_ki; a.c_maxf = b->c_maxf; a.t_kp = b->t_kp; a.t_ki = b->t_ki;
    a.t_maxdev = b->sps / 4;
    a.dibits = d_dibits;
    a.symbols = reinterpret_cast<double2 *>(d_symbols);
    a.cap = cap;
    a.counts = d_counts;
    const int H = b->L - 1;
    hipLaunchKernelGGL(k_cq_rrc, dim3((unsigned)((n + H + 255) / 256), b->C), dim3(256), b->L * sizeof(double), st, a);
    WH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cq_seq, dim3((b->C + 63) / 64), dim3(64), 0, st, a);
    WH_LAUNCH_CHECK();
    b->cur ^= 1;
    return WH_OK;
}

struct wh_gardner_bank {
    int C;
    double sps, kp, ki;
    GState *d_st = nullptr;
};

extern "C" int wh_gardner_bank_create(wh_gardner_bank **out, int C, double sps, double kp, double ki) {
    if (!out || C < 1 || !(sps > 1.0)) return set_err(WH_E_ARG, "wh_gardner_bank_create: bad arguments");
    wh_gardner_bank *g = new wh_gardner_bank();
    std::unique_ptr<wh_gardner_bank, void (*)(wh_gardner_bank *)> guard(g, wh_gardner_bank_destroy);  // frees partial state on early return
    g->C = C; g->sps = sps; g->kp = kp; g->ki = ki;
    WH_HIP(hipMalloc(&g->d_st, (size_t)C * sizeof(GState)));
    WH_HIP(hipMemset(g->d_st, 0, (size_t)C * sizeof(GState)));
    *out = guard.release();
    return WH_OK;
}

extern "C" void wh_gardner_bank_destroy(wh_gardner_bank *g) {
    if (!g) return;
    (void)hipFree(g->d_st);
    delete g;
}

extern "C" int wh_gardner_bank_reset(wh_gardner_bank *g, void *stream) {
    if (!g) return set_err(WH_E_ARG, "wh_gardner_bank_reset: null handle");
    WH_HIP(hipMemsetAsync(g->d_st, 0, (size_t)g->C * sizeof(GState), as_stream(stream)));
    return WH_OK;
}

extern "C" int wh_gardner_bank_run(wh_gardner_bank *g, const float *d_x, size_t n, size_t stride, double *d_symbols,
                                   double *d_errors, size_t cap, int32_t *d_counts, void *stream) {
    if (!g || !d_counts) return set_err(WH_E_ARG, "wh_gardner_bank_run: null handle/counts");
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        WH_HIP(hipMemsetAsync(d_counts, 0, (size_t)g->C * sizeof(int32_t), st));
        return WH_OK;
    }
    if (!d_x || !d_symbols || !d_errors || stride < n || n > 0x7fffffff)
        return set_err(WH_E_ARG, "wh_gardner_bank_run: bad buffers");
    hipLaunchKernelGGL(k_gardner, dim3((g->C + 63) / 64), dim3(64), 0, st, d_x, stride, (int)n, g->C, g->d_st, g->sps,
                       g->kp, g->ki, d_symbols, d_errors, cap, d_counts);
    WH_LAUNCH_CHECK();
    return WH_OK;
}
