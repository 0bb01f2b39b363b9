// Trunking front-end for gfx950:
//   * DDC: phase-continuous float64-phase NCO (reference trunking/system.py:1434-1466) fused with
//     the first stage of the two-stage decimating FIR (system.py:1392-1406, 1753-1779;
//     dsp/filters.py:558-646 fir_decimate), computing ONLY the kept outputs.  The reference
//     filters at full rate with lfilter (float64 / complex128), rounds to complex64 and slices
//     [::D] per chunk -- so the decimation phase restarts at every call and the filter state is
//     the input history; both are reproduced.  First call: zi = lfilter_zi(taps) * x[0], i.e. a
//     history filled with the first (mixed) sample.
//   * scanner measurement (trunking/cc_scanner.py:165-264): for a list of candidate offsets, the
//     stateless float32-phase NCO (capture.py:166-193) -> 65-tap low-pass -> [::D] -> mean and
//     peak |.|^2 (float64, the reference keeps complex128 here), all offsets in one launch.
#include "wh_common.h"
#include "wh_portable_math.h"
#include <cmath>
#include <memory>
#include <vector>

using namespace wh;

namespace {

struct DdcArgs {
    const float2 *x;      // stage input (raw IQ for stage 1, decimated1 for stage 2)
    const float2 *hist;   // [L-1] previous (mixed) inputs, oldest first
    float2 *out;          // [n_out]
    const double *taps;   // [L]
    int L, D, n, n_out, opw;
    int mix;              // 1: apply the NCO to x
    int first;            // 1: history := first (mixed) sample (lfilter_zi * x[0])
    double c1;            // (-2.0*np.pi) * offset_hz
    double fs;
    long long idx0;       // sample index of x[0]
};

__device__ __forceinline__ float2 ddc_mix(float2 v, const DdcArgs &a, long long i) {
    if (!a.mix) return v;
    // phase = -2.0*np.pi*offset_hz*n/sample_rate, left to right in float64 (system.py:1456-1457)
    double nn = (double)(a.idx0 + i);
    double ph = __ddiv_rn(__dmul_rn(a.c1, nn), a.fs);
    double s, c;
    sincos(ph, &s, &c);
    float sr = (float)c, si = (float)s;   // np.exp(1j*phase).astype(complex64)
    return make_float2(__fsub_rn(__fmul_rn(v.x, sr), __fmul_rn(v.y, si)),
                       __fadd_rn(__fmul_rn(v.x, si), __fmul_rn(v.y, sr)));
}

__global__ __launch_bounds__(256) void ddc_stage_kernel(DdcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    double *tp = reinterpret_cast<double *>(sm_raw);
    float2 *tile = reinterpret_cast<float2 *>(sm_raw + (size_t)a.L * sizeof(double));
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * a.opw;
    int mc = a.n_out - m0;
    if (mc > a.opw) mc = a.opw;
    const int H = a.L - 1;
    const long long p0 = (long long)m0 * a.D - H;      // stream position of tile[0]
    const int span = (mc - 1) * a.D + a.L;
    for (int i = tid; i < a.L; i += 256) tp[i] = a.taps[i];
    for (int i = tid; i < span; i += 256) {
        long long p = p0 + i;
        float2 v;
        if (p >= 0) v = ddc_mix(a.x[p], a, p);
        else if (a.first) v = ddc_mix(a.x[0], a, 0);
        else v = a.hist[p + H];
        tile[i] = v;
    }
    __syncthreads();
    // 256/opw lanes share one output (opw is a power of two <= 256)
    const int S = 256 / a.opw;
    const int o = tid / S, sub = tid - o * S;
    double ar = 0.0, ai = 0.0;
    if (o < mc) {
        const float2 *w = tile + o * a.D + H;
        for (int k = sub; k < a.L; k += S) {
            float2 v = w[-k];
            double t = tp[k];
            ar = fma(t, (double)v.x, ar);
            ai = fma(t, (double)v.y, ai);
        }
    }
    for (int w = 1; w < S; w <<= 1) {
        ar += __shfl_xor(ar, w);
        ai += __shfl_xor(ai, w);
    }
    if (o < mc && sub == 0) a.out[m0 + o] = make_float2((float)ar, (float)ai);
}

// new_hist[i] = (mixed) stream sample n - (L-1) + i
__global__ __launch_bounds__(256) void ddc_carry_kernel(DdcArgs a, float2 *hist_new) {
    const int H = a.L - 1;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H) return;
    long long p = (long long)a.n - H + i;
    float2 v;
    if (p >= 0) v = ddc_mix(a.x[p], a, p);
    else if (a.first) v = ddc_mix(a.x[0], a, 0);
    else v = a.hist[p + H];
    hist_new[i] = v;
}


// ---- bank form of the front-end: K voice recorders / control monitors on ONE wideband buffer -------------
// (trunking/system.py:453-656 VoiceRecorder.process_iq runs the same NCO + two lfilter/[::D] stages per recorder,
// each with its own offset, sample index and filter state; inactive recorders are skipped and keep their state.)
constexpr int DDC_BANK_MAX = 64;

struct DdcChan {
    double c1;            // (-2.0*np.pi) * offset_hz
    long long idx0;       // sample index of x[0]
    int mix, first, active, cur;
};

struct DdcBankArgs {
    const float2 *x;      // stage 1: the shared input (x_stride 0); stage 2: [K][x_stride]
    size_t x_stride;
    float2 *hist[2];      // [K][L-1] each, double-buffered per channel (DdcChan.cur selects the valid one)
    float2 *out;          // [K][out_stride]
    size_t out_stride;
    const double *taps;
    int L, D, n, n_out, opw, stage2;
    double fs;
    DdcChan ch[DDC_BANK_MAX];
};

__device__ __forceinline__ float2 ddcb_fetch(const DdcBankArgs &a, const DdcChan &c, const float2 *x, const float2 *hist,
                                             long long p) {
    DdcArgs m;            // reuse ddc_mix's arithmetic
    m.mix = a.stage2 ? 0 : c.mix; m.c1 = c.c1; m.fs = a.fs; m.idx0 = c.idx0;
    if (p >= 0) return ddc_mix(x[p], m, p);
    if (c.first) return ddc_mix(x[0], m, 0);
    return hist[p + (a.L - 1)];
}

__global__ __launch_bounds__(256) void ddcb_stage_kernel(DdcBankArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    const DdcChan &c = a.ch[blockIdx.y];
    if (!c.active) return;
    double *tp = reinterpret_cast<double *>(sm_raw);
    float2 *tile = reinterpret_cast<float2 *>(sm_raw + (size_t)a.L * sizeof(double));
    const float2 *x = a.x + (size_t)blockIdx.y * a.x_stride;
    const float2 *hist = a.hist[c.cur] + (size_t)blockIdx.y * (a.L - 1);
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * a.opw;
    int mc = a.n_out - m0;
    if (mc > a.opw) mc = a.opw;
    const int H = a.L - 1;
    const long long p0 = (long long)m0 * a.D - H;
    const int span = (mc - 1) * a.D + a.L;
    for (int i = tid; i < a.L; i += 256) tp[i] = a.taps[i];
    for (int i = tid; i < span; i += 256) tile[i] = ddcb_fetch(a, c, x, hist, p0 + i);
    __syncthreads();
    const int S = 256 / a.opw;
    const int o = tid / S, sub = tid - o * S;
    double ar = 0.0, ai = 0.0;
    if (o < mc) {
        const float2 *w = tile + o * a.D + H;
        for (int k = sub; k < a.L; k += S) {
            float2 v = w[-k];
            double t = tp[k];
            ar = fma(t, (double)v.x, ar);
            ai = fma(t, (double)v.y, ai);
        }
    }
    for (int w = 1; w < S; w <<= 1) {
        ar += __shfl_xor(ar, w);
        ai += __shfl_xor(ai, w);
    }
    if (o < mc && sub == 0) a.out[(size_t)blockIdx.y * a.out_stride + m0 + o] = make_float2((float)ar, (float)ai);
}

__global__ __launch_bounds__(256) void ddcb_carry_kernel(DdcBankArgs a) {
    const DdcChan &c = a.ch[blockIdx.y];
    if (!c.active) return;
    const int H = a.L - 1;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H) return;
    const float2 *x = a.x + (size_t)blockIdx.y * a.x_stride;
    const float2 *hist = a.hist[c.cur] + (size_t)blockIdx.y * H;
    a.hist[c.cur ^ 1][(size_t)blockIdx.y * H + i] = ddcb_fetch(a, c, x, hist, (long long)a.n - H + i);
}

// ---- scanner measurement ----------------------------------------------------------------------
struct ScanArgs {
    const float2 *x;
    const double *taps;    // [L] float64
    const float *nco_c;    // [n_off] f32(-2 pi off/fs), 0 => no mix
    double *part;          // [n_off][blocks][2] partial {sum p, max p}
    int L, D, n, n_out, blocks;
};

__global__ __launch_bounds__(256) void scan_measure_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    double *tp = reinterpret_cast<double *>(sm_raw);
    __shared__ double rs[4], rm[4];
    const int tid = threadIdx.x, off = blockIdx.y;
    for (int i = tid; i < a.L; i += 256) tp[i] = a.taps[i];
    __syncthreads();
    const float c = a.nco_c[off];
    const bool do_mix = c != 0.0f;
    double s = 0.0, mx = 0.0;
    for (int m = blockIdx.x * 256 + tid; m < a.n_out; m += a.blocks * 256) {
        const long long p = (long long)m * a.D;
        double ar = 0.0, ai = 0.0;
        for (int k = 0; k < a.L; ++k) {
            long long q = p - k;
            if (q < 0) break;            // zero initial state (lfilter without zi)
            float2 v = a.x[q];
            if (do_mix) {
                float ph = __fmul_rn(c, (float)q);
                float sn, cs;
                whm_sincos_phase(ph, &sn, &cs);
                v = make_float2(v.x * cs - v.y * sn, v.x * sn + v.y * cs);
            }
            double t = tp[k];
            ar = fma(t, (double)v.x, ar);
            ai = fma(t, (double)v.y, ai);
        }
        double pw = ar * ar + ai * ai;
        s += pw;
        mx = pw > mx ? pw : mx;
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        mx = fmax(mx, __shfl_xor(mx, o));
    }
    if ((tid & 63) == 0) {
        rs[tid >> 6] = s;
        rm[tid >> 6] = mx;
    }
    __syncthreads();
    if (tid == 0) {
        double *o = a.part + ((size_t)off * a.blocks + blockIdx.x) * 2;
        o[0] = rs[0] + rs[1] + rs[2] + rs[3];
        o[1] = fmax(fmax(rm[0], rm[1]), fmax(rm[2], rm[3]));
    }
}

// ---- sync-pattern check of cc_scanner.py:266-353 ---------------------------------------------
// symbol_samples[k] = angle(y[j+1] * conj(y[j])), j = 5 + 10 k, y = the decimated complex128 stream of
// scan_measure (only the two y values per symbol that are needed get evaluated).
__global__ __launch_bounds__(256) void scan_symbols_kernel(ScanArgs a, double *sym /*[n_off][n_sym]*/, int n_sym) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    double *tp = reinterpret_cast<double *>(sm_raw);
    const int tid = threadIdx.x, off = blockIdx.y;
    for (int i = tid; i < a.L; i += 256) tp[i] = a.taps[i];
    __syncthreads();
    const int k = blockIdx.x * 256 + tid;
    if (k >= n_sym) return;
    const float c = a.nco_c[off];
    const bool do_mix = c != 0.0f;
    double yr[2], yi[2];
    for (int e = 0; e < 2; ++e) {
        const long long p = (long long)(5 + 10 * k + e) * a.D;
        double ar = 0.0, ai = 0.0;
        for (int t = 0; t < a.L; ++t) {
            long long q = p - t;
            if (q < 0) break;
            float2 v = a.x[q];
            if (do_mix) {
                float ph = __fmul_rn(c, (float)q);
                float sn, cs;
                whm_sincos_phase(ph, &sn, &cs);
                v = make_float2(v.x * cs - v.y * sn, v.x * sn + v.y * cs);
            }
            ar = fma(tp[t], (double)v.x, ar);
            ai = fma(tp[t], (double)v.y, ai);
        }
        yr[e] = ar; yi[e] = ai;
    }
    // y[j+1] * conj(y[j])
    double re = yr[1] * yr[0] + yi[1] * yi[0];
    double im = yi[1] * yr[0] - yr[1] * yi[0];
    sym[(size_t)off * n_sym + k] = atan2(im, re);
}

// best normalised correlation against the +-0.2356 sync waveform over all window positions
__global__ __launch_bounds__(256) void scan_sync_kernel(const double *sym, int n_sym, int search_len, double *best_out) {
    const int off = blockIdx.x, tid = threadIdx.x;
    const double *s = sym + (size_t)off * n_sym;
    const unsigned long long PAT = 0x5575F5FF77FFULL;   // dibit 1 -> +dev, dibit 3 -> -dev
    const double DEV = 0.2356;
    const double norm_sync = sqrt(24.0 * DEV * DEV);
    double best = 0.0;
    int best_i = 0x7fffffff;
    for (int i = tid; i < search_len; i += 256) {
        double dot = 0.0, nn = 0.0;
        for (int j = 0; j < 24; ++j) {
            double w = s[i + j];
            double sw = ((PAT >> ((23 - j) * 2)) & 3ULL) == 1ULL ? DEV : -DEV;
            dot += w * sw;
            nn += w * w;
        }
        double corr = dot / (sqrt(nn + 1e-10) * norm_sync);
        if (fabs(corr) > fabs(best)) { best = corr; best_i = i; }   // i increases per thread: first max kept
    }
    __shared__ double sb[256];
    __shared__ int si[256];
    sb[tid] = best; si[tid] = best_i;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            double o = sb[tid + st];
            int oi = si[tid + st];
            if (fabs(o) > fabs(sb[tid]) || (fabs(o) == fabs(sb[tid]) && oi < si[tid])) { sb[tid] = o; si[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) best_out[off] = sb[0];
}

__global__ void scan_final_kernel(const double *part, int blocks, int n_off, int n_out, double *out) {
    int off = blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= n_off) return;
    double s = 0.0, mx = 0.0;
    for (int b = 0; b < blocks; ++b) {
        s += part[((size_t)off * blocks + b) * 2];
        mx = fmax(mx, part[((size_t)off * blocks + b) * 2 + 1]);
    }
    out[off * 2] = s / (double)n_out;   // np.mean(np.abs(decimated) ** 2)
    out[off * 2 + 1] = mx;              // np.max(...)
}

int pick_opw(int L, int D) {
    int opw = 256;
    while (opw > 4 && (size_t)((opw - 1) * D + L) * sizeof(float2) + (size_t)L * sizeof(double) > 60 * 1024) opw >>= 1;
    return opw;   // >= 4 so that the 256/opw lanes sharing an output stay inside one wavefront
}

}  // namespace

struct wh_ddc {
    int fs, L1, D1, L2, D2, max_n;
    double *d_t1 = nullptr, *d_t2 = nullptr;
    float2 *d_h1[2] = {nullptr, nullptr}, *d_h2[2] = {nullptr, nullptr};
    float2 *d_mid = nullptr;
    int cur = 0;
    bool first = true;
    long long sample_idx = 0;
    double last_offset = 0.0;
};

extern "C" int wh_ddc_create(wh_ddc **out, int sample_rate, const double *h_taps1, int n1, int d1,
                             const double *h_taps2, int n2, int d2, int max_samples_per_call) {
    if (!out || !h_taps1 || n1 < 1 || d1 < 1 || sample_rate < 1 || max_samples_per_call < 1 || n1 > 4096 ||
        (d2 > 1 && (!h_taps2 || n2 < 1 || n2 > 4096)))
        return set_err(WH_E_ARG, "wh_ddc_create: bad arguments");
    wh_ddc *d = new wh_ddc();
    std::unique_ptr<wh_ddc, void (*)(wh_ddc *)> guard(d, wh_ddc_destroy);  // frees partial state on early return
    d->fs = sample_rate; d->L1 = n1; d->D1 = d1; d->L2 = d2 > 1 ? n2 : 0; d->D2 = d2 > 1 ? d2 : 1;
    d->max_n = max_samples_per_call;
    WH_HIP(hipMalloc(&d->d_t1, n1 * sizeof(double)));
    WH_HIP(hipMemcpy(d->d_t1, h_taps1, n1 * sizeof(double), hipMemcpyHostToDevice));
    for (int i = 0; i < 2; ++i) WH_HIP(hipMalloc(&d->d_h1[i], (size_t)(n1 > 1 ? n1 - 1 : 1) * sizeof(float2)));
    if (d->L2) {
        WH_HIP(hipMalloc(&d->d_t2, n2 * sizeof(double)));
        WH_HIP(hipMemcpy(d->d_t2, h_taps2, n2 * sizeof(double), hipMemcpyHostToDevice));
        for (int i = 0; i < 2; ++i) WH_HIP(hipMalloc(&d->d_h2[i], (size_t)(n2 > 1 ? n2 - 1 : 1) * sizeof(float2)));
        WH_HIP(hipMalloc(&d->d_mid, ((size_t)max_samples_per_call / d1 + 2) * sizeof(float2)));
    }
    *out = guard.release();
    return WH_OK;
}

extern "C" void wh_ddc_destroy(wh_ddc *d) {
    if (!d) return;
    (void)hipFree(d->d_t1); (void)hipFree(d->d_t2); (void)hipFree(d->d_mid);
    for (int i = 0; i < 2; ++i) { (void)hipFree(d->d_h1[i]); (void)hipFree(d->d_h2[i]); }
    delete d;
}

extern "C" int wh_ddc_reset(wh_ddc *d) {   /* overflow recovery, system.py:1574-1588 */
    if (!d) return set_err(WH_E_ARG, "wh_ddc_reset: null handle");
    d->first = true;
    d->sample_idx = 0;
    d->last_offset = 0.0;
    return WH_OK;
}

extern "C" size_t wh_ddc_out_len(const wh_ddc *d, size_t n) {
    if (!d || n == 0) return 0;
    size_t n1 = (n + d->D1 - 1) / d->D1;
    return (n1 + d->D2 - 1) / d->D2;
}

static int ddc_launch(const DdcArgs &a0, float2 *hist_new, hipStream_t st) {
    DdcArgs a = a0;
    a.opw = pick_opw(a.L, a.D);
    size_t smem = (size_t)a.L * sizeof(double) + (size_t)((a.opw - 1) * a.D + a.L) * sizeof(float2);
    if (smem > 64 * 1024) return set_err(WH_E_ARG, "ddc: decimation %d x %d taps does not fit the LDS tile", a.D, a.L);
    unsigned blocks = (unsigned)((a.n_out + a.opw - 1) / a.opw);
    hipLaunchKernelGGL(ddc_stage_kernel, dim3(blocks), dim3(256), smem, st, a);
    WH_LAUNCH_CHECK();
    if (a.L > 1) {
        hipLaunchKernelGGL(ddc_carry_kernel, dim3((a.L - 1 + 255) / 256), dim3(256), 0, st, a, hist_new);
        WH_LAUNCH_CHECK();
    }
    return WH_OK;
}

extern "C" int wh_ddc_run(wh_ddc *d, const float *d_iq, size_t n, double offset_hz, float *d_out, void *stream) {
    if (!d) return set_err(WH_E_ARG, "wh_ddc_run: null handle");
    if (n == 0) return WH_OK;
    if (!d_iq || !d_out) return set_err(WH_E_ARG, "wh_ddc_run: null buffer");
    if (n > (size_t)d->max_n) return set_err(WH_E_ARG, "wh_ddc_run: n exceeds max_samples_per_call");
    hipStream_t st = as_stream(stream);
    const bool mix = offset_hz != 0.0;
    if (mix && offset_hz != d->last_offset) {   // system.py:1450-1452
        d->sample_idx = 0;
        d->last_offset = offset_hz;
    }
    const int n1 = (int)((n + d->D1 - 1) / d->D1);
    DdcArgs a;
    a.x = reinterpret_cast<const float2 *>(d_iq);
    a.hist = d->d_h1[d->cur];
    a.out = d->L2 ? d->d_mid : reinterpret_cast<float2 *>(d_out);
    a.taps = d->d_t1; a.L = d->L1; a.D = d->D1; a.n = (int)n; a.n_out = n1; a.opw = 0;
    a.mix = mix ? 1 : 0; a.first = d->first ? 1 : 0;
    a.c1 = (-2.0 * M_PI) * offset_hz;
    a.fs = (double)d->fs;
    a.idx0 = d->sample_idx;
    int rc = ddc_launch(a, d->d_h1[d->cur ^ 1], st);
    if (rc != WH_OK) return rc;
    if (d->L2) {
        DdcArgs b;
        b.x = d->d_mid;
        b.hist = d->d_h2[d->cur];
        b.out = reinterpret_cast<float2 *>(d_out);
        b.taps = d->d_t2; b.L = d->L2; b.D = d->D2; b.n = n1; b.n_out = (n1 + d->D2 - 1) / d->D2; b.opw = 0;
        b.mix = 0; b.first = d->first ? 1 : 0; b.c1 = 0.0; b.fs = 1.0; b.idx0 = 0;
        rc = ddc_launch(b, d->d_h2[d->cur ^ 1], st);
        if (rc != WH_OK) return rc;
    }
    d->cur ^= 1;
    d->first = false;
    if (mix) {   // system.py:1461-1466
        d->sample_idx += (long long)n;
        if (d->sample_idx >= d->fs) d->sample_idx %= d->fs;
    }
    return WH_OK;
}

// ---- wh_ddc_bank ---------------------------------------------------------------------------------
struct wh_ddc_bank {
    int K, fs, L1, D1, L2, D2, max_n;
    double *d_t1 = nullptr, *d_t2 = nullptr;
    float2 *d_h1[2] = {nullptr, nullptr}, *d_h2[2] = {nullptr, nullptr};
    float2 *d_mid = nullptr;
    size_t mid_stride = 0;
    struct Chan { bool first = true; long long sample_idx = 0; double last_offset = 0.0; int cur = 0; };
    std::vector<Chan> ch;
};

extern "C" int wh_ddc_bank_create(wh_ddc_bank **out, int n_channels, int sample_rate, const double *h_taps1, int n1, int d1,
                                  const double *h_taps2, int n2, int d2, int max_samples_per_call) {
    if (!out || !h_taps1 || n_channels < 1 || n_channels > DDC_BANK_MAX || n1 < 2 || d1 < 1 || sample_rate < 1 ||
        max_samples_per_call < 1 || n1 > 4096 || (d2 > 1 && (!h_taps2 || n2 < 2 || n2 > 4096)))
        return set_err(WH_E_ARG, "wh_ddc_bank_create: bad arguments (1..%d channels)", DDC_BANK_MAX);
    wh_ddc_bank *d = new wh_ddc_bank();
    std::unique_ptr<wh_ddc_bank, void (*)(wh_ddc_bank *)> guard(d, wh_ddc_bank_destroy);  // frees partial state on early return
    d->K = n_channels; d->fs = sample_rate; d->L1 = n1; d->D1 = d1; d->L2 = d2 > 1 ? n2 : 0; d->D2 = d2 > 1 ? d2 : 1;
    d->max_n = max_samples_per_call;
    d->ch.resize(n_channels);
    WH_HIP(hipMalloc(&d->d_t1, n1 * sizeof(double)));
    WH_HIP(hipMemcpy(d->d_t1, h_taps1, n1 * sizeof(double), hipMemcpyHostToDevice));
    for (int i = 0; i < 2; ++i) WH_HIP(hipMalloc(&d->d_h1[i], (size_t)n_channels * (n1 - 1) * sizeof(float2)));
    if (d->L2) {
        WH_HIP(hipMalloc(&d->d_t2, n2 * sizeof(double)));
        WH_HIP(hipMemcpy(d->d_t2, h_taps2, n2 * sizeof(double), hipMemcpyHostToDevice));
        for (int i = 0; i < 2; ++i) WH_HIP(hipMalloc(&d->d_h2[i], (size_t)n_channels * (n2 - 1) * sizeof(float2)));
        d->mid_stride = (size_t)max_samples_per_call / d1 + 2;
        WH_HIP(hipMalloc(&d->d_mid, (size_t)n_channels * d->mid_stride * sizeof(float2)));
    }
    *out = guard.release();
    return WH_OK;
}

extern "C" void wh_ddc_bank_destroy(wh_ddc_bank *d) {
    if (!d) return;
    (void)hipFree(d->d_t1); (void)hipFree(d->d_t2); (void)hipFree(d->d_mid);
    for (int i = 0; i < 2; ++i) { (void)hipFree(d->d_h1[i]); (void)hipFree(d->d_h2[i]); }
    delete d;
}

extern "C" int wh_ddc_bank_reset(wh_ddc_bank *d, int channel) {
    if (!d || channel < -1 || channel >= d->K) return set_err(WH_E_ARG, "wh_ddc_bank_reset: bad arguments");
    for (int k = 0; k < d->K; ++k)
        if (channel < 0 || channel == k) {
            int cur = d->ch[k].cur;
            d->ch[k] = wh_ddc_bank::Chan();
            d->ch[k].cur = cur;
        }
    return WH_OK;
}

extern "C" size_t wh_ddc_bank_out_len(const wh_ddc_bank *d, size_t n) {
    if (!d || n == 0) return 0;
    size_t n1 = (n + d->D1 - 1) / d->D1;
    return (n1 + d->D2 - 1) / d->D2;
}

static int ddcb_launch(DdcBankArgs &a, int K, hipStream_t st) {
    a.opw = pick_opw(a.L, a.D);
    size_t smem = (size_t)a.L * sizeof(double) + (size_t)((a.opw - 1) * a.D + a.L) * sizeof(float2);
    if (smem > 64 * 1024) return set_err(WH_E_ARG, "ddc bank: decimation %d x %d taps does not fit the LDS tile", a.D, a.L);
    hipLaunchKernelGGL(ddcb_stage_kernel, dim3((unsigned)((a.n_out + a.opw - 1) / a.opw), K), dim3(256), smem, st, a);
    WH_LAUNCH_CHECK();
    hipLaunchKernelGGL(ddcb_carry_kernel, dim3((a.L - 1 + 255) / 256, K), dim3(256), 0, st, a);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_ddc_bank_run(wh_ddc_bank *d, const float *d_iq, size_t n, const double *h_offsets_hz,
                               const unsigned char *h_active, float *d_out, size_t out_stride, void *stream) {
    if (!d) return set_err(WH_E_ARG, "wh_ddc_bank_run: null handle");
    if (n == 0) return WH_OK;
    if (!d_iq || !d_out || !h_offsets_hz) return set_err(WH_E_ARG, "wh_ddc_bank_run: null buffer");
    if (n > (size_t)d->max_n) return set_err(WH_E_ARG, "wh_ddc_bank_run: n exceeds max_samples_per_call");
    if (out_stride < wh_ddc_bank_out_len(d, n)) return set_err(WH_E_ARG, "wh_ddc_bank_run: out_stride too small");
    hipStream_t st = as_stream(stream);
    const int n1 = (int)((n + d->D1 - 1) / d->D1);
    DdcBankArgs a;
    a.x = reinterpret_cast<const float2 *>(d_iq);
    a.x_stride = 0;
    a.hist[0] = d->d_h1[0]; a.hist[1] = d->d_h1[1];
    a.out = d->L2 ? d->d_mid : reinterpret_cast<float2 *>(d_out);
    a.out_stride = d->L2 ? d->mid_stride : out_stride;
    a.taps = d->d_t1; a.L = d->L1; a.D = d->D1; a.n = (int)n; a.n_out = n1; a.stage2 = 0;
    a.fs = (double)d->fs;
    for (int k = 0; k < DDC_BANK_MAX; ++k) a.ch[k] = DdcChan{0.0, 0, 0, 0, 0, 0};
    for (int k = 0; k < d->K; ++k) {
        wh_ddc_bank::Chan &c = d->ch[k];
        DdcChan &g = a.ch[k];
        g.active = (!h_active || h_active[k]) ? 1 : 0;
        if (!g.active) continue;
        const double off = h_offsets_hz[k];
        g.mix = off != 0.0;
        if (g.mix && off != c.last_offset) {   // system.py:604-606 / 1450-1452
            c.sample_idx = 0;
            c.last_offset = off;
        }
        g.c1 = (-2.0 * M_PI) * off;
        g.idx0 = c.sample_idx;
        g.first = c.first ? 1 : 0;
        g.cur = c.cur;
    }
    int rc = ddcb_launch(a, d->K, st);
    if (rc != WH_OK) return rc;
    if (d->L2) {
        DdcBankArgs b = a;
        b.x = d->d_mid; b.x_stride = d->mid_stride;
        b.hist[0] = d->d_h2[0]; b.hist[1] = d->d_h2[1];
        b.out = reinterpret_cast<float2 *>(d_out); b.out_stride = out_stride;
        b.taps = d->d_t2; b.L = d->L2; b.D = d->D2; b.n = n1; b.n_out = (n1 + d->D2 - 1) / d->D2; b.stage2 = 1;
        rc = ddcb_launch(b, d->K, st);
        if (rc != WH_OK) return rc;
    }
    for (int k = 0; k < d->K; ++k) {
        if (!a.ch[k].active) continue;
        wh_ddc_bank::Chan &c = d->ch[k];
        c.cur ^= 1;
        c.first = false;
        if (a.ch[k].mix) {   // system.py:609-611 / 1461-1466
            c.sample_idx += (long long)n;
            if (c.sample_idx >= d->fs) c.sample_idx %= d->fs;
        }
    }
    return WH_OK;
}

extern "C" int wh_scan_measure(const float *d_iq, size_t n, int sample_rate, const int *h_offsets_hz, int n_off,
                               const double *h_taps, int ntaps, int decim, double *h_out /* [n_off][2] */,
                               double *h_sync_corr /* [n_off] or NULL */, void *stream) {
    if (!d_iq || !h_offsets_hz || !h_taps || !h_out || n == 0 || n_off < 1 || n_off > 65535 || ntaps < 1 ||
        ntaps > 4096 || decim < 1 || n > (size_t)1 << 24)
        return set_err(WH_E_ARG, "wh_scan_measure: bad arguments");
    hipStream_t st = as_stream(stream);
    const int n_out = (int)((n + decim - 1) / decim);
    int blocks = (n_out + 255) / 256;
    if (blocks > 64) blocks = 64;
    std::vector<float> nco(n_off);
    for (int i = 0; i < n_off; ++i)
        nco[i] = h_offsets_hz[i] == 0 ? 0.0f : (float)(-2.0 * M_PI * ((double)h_offsets_hz[i] / (double)sample_rate));
    double *d_taps = nullptr, *d_part = nullptr, *d_res = nullptr;
    float *d_nco = nullptr;
    WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_taps), ntaps * sizeof(double), st));
    WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_nco), n_off * sizeof(float), st));
    WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_part), (size_t)n_off * blocks * 2 * sizeof(double), st));
    WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_res), (size_t)n_off * 2 * sizeof(double), st));
    WH_HIP(hipMemcpyAsync(d_taps, h_taps, ntaps * sizeof(double), hipMemcpyHostToDevice, st));
    WH_HIP(hipMemcpyAsync(d_nco, nco.data(), n_off * sizeof(float), hipMemcpyHostToDevice, st));
    ScanArgs a;
    a.x = reinterpret_cast<const float2 *>(d_iq);
    a.taps = d_taps; a.nco_c = d_nco; a.part = d_part;
    a.L = ntaps; a.D = decim; a.n = (int)n; a.n_out = n_out; a.blocks = blocks;
    hipLaunchKernelGGL(scan_measure_kernel, dim3(blocks, n_off), dim3(256), ntaps * sizeof(double), st, a);
    WH_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_final_kernel, dim3((n_off + 63) / 64), dim3(64), 0, st, d_part, blocks, n_off, n_out, d_res);
    WH_LAUNCH_CHECK();
    WH_HIP(hipMemcpyAsync(h_out, d_res, (size_t)n_off * 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    double *d_sym = nullptr, *d_best = nullptr;
    if (h_sync_corr) {
        // cc_scanner.py:289-325: 10 samples/symbol, fm = angle(y[1:] conj y[:-1]), symbols at 5::10
        for (int i = 0; i < n_off; ++i) h_sync_corr[i] = 0.0;
        const int fm_len = n_out - 1;
        const int symbols_count = fm_len / 10;
        const int have = fm_len > 5 ? (fm_len - 5 + 9) / 10 : 0;
        const int n_sym = have < symbols_count ? have : symbols_count;
        const int search_len = n_sym - 24;           // min(len(symbol_samples) - 24, symbols_count - 24)
        if (n_out >= 10 * 24 + 10 && symbols_count >= 24 && search_len > 0) {
            WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_sym), (size_t)n_off * n_sym * sizeof(double), st));
            WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_best), (size_t)n_off * sizeof(double), st));
            hipLaunchKernelGGL(scan_symbols_kernel, dim3((n_sym + 255) / 256, n_off), dim3(256), ntaps * sizeof(double), st,
                               a, d_sym, n_sym);
            WH_LAUNCH_CHECK();
            hipLaunchKernelGGL(scan_sync_kernel, dim3(n_off), dim3(256), 0, st, d_sym, n_sym, search_len, d_best);
            WH_LAUNCH_CHECK();
            WH_HIP(hipMemcpyAsync(h_sync_corr, d_best, (size_t)n_off * sizeof(double), hipMemcpyDeviceToHost, st));
        }
    }
    WH_HIP(hipStreamSynchronize(st));   // results are host scalars (the scanner picks a channel with them)
    if (d_sym) WH_HIP(hipFreeAsync(d_sym, st));
    if (d_best) WH_HIP(hipFreeAsync(d_best, st));
    WH_HIP(hipFreeAsync(d_taps, st));
    WH_HIP(hipFreeAsync(d_nco, st));
    WH_HIP(hipFreeAsync(d_part, st));
    WH_HIP(hipFreeAsync(d_res, st));
    return WH_OK;
}
