// Spectrum frames (reference dsp/fft/scipy_backend.py:38-79): Hann window -> FFT -> fftshift ->
// 20 log10(|X| + 1e-10), fused in one kernel, one workgroup per frame (batched over frames).
// FFT: frame sizes 256 .. 4096 go through the compile-time-shaped passes of pfb_mid.hip (spectrum_mid_kernel: in-place
// DIF FFT in padded LDS images, several frames per workgroup, next frames prefetched); other power-of-two sizes take the
// LDS Stockham kernel below (radix-4 passes), anything else the direct DFT.
#include "pfb_internal.h"
#include "wh_common.h"
#include <cmath>
#include <memory>
#include <vector>

using namespace wh;

namespace {

__global__ __launch_bounds__(256) void spectrum_kernel(const float2 *iq, size_t frame_stride, float *out,
                                                       const float *window, const float2 *tw, int N, int log2N) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    const float2 *x = iq + (size_t)blockIdx.x * frame_stride;
    float *o = out + (size_t)blockIdx.x * N;
    for (int n = threadIdx.x; n < N; n += 256) {
        float2 v = x[n];
        float w = window[n];
        sm[n] = make_float2(v.x * w, v.y * w);
    }
    __syncthreads();
    const int shift = N - N / 2;  // fftshift: shifted[i] = X[(i + ceil(N/2)) mod N]
    if (log2N > 0) {
        float2 *src = sm, *dst = sm + N;
        int n = N, s = 1;
        // autosort Stockham: radix-4 passes (half the LDS round trips and barriers of radix-2), one radix-2 pass at the
        // end when log2(N) is odd
        for (int st = 0; st + 1 < log2N; st += 2) {
            const int m = n >> 2;
            for (int i = threadIdx.x; i < (N >> 2); i += 256) {
                const int pp = i / s, q = i - pp * s;
                const float2 a = src[q + s * pp];
                const float2 b = src[q + s * (pp + m)];
                const float2 c = src[q + s * (pp + 2 * m)];
                const float2 d = src[q + s * (pp + 3 * m)];
                const float2 w1 = tw[(size_t)pp * s];
                const float2 w2 = tw[(size_t)2 * pp * s];
                const float2 w3 = tw[(size_t)3 * pp * s];
                const float2 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = csub(b, d);
                const float2 jbmd = make_float2(-bmd.y, bmd.x);               // j (b - d)
                dst[q + s * (4 * pp)] = cadd(apc, bpd);
                dst[q + s * (4 * pp + 1)] = cmul(csub(amc, jbmd), w1);          // a - j b - c + j d
                dst[q + s * (4 * pp + 2)] = cmul(csub(apc, bpd), w2);
                dst[q + s * (4 * pp + 3)] = cmul(cadd(amc, jbmd), w3);          // a + j b - c - j d
            }
            __syncthreads();
            float2 *tmp = src; src = dst; dst = tmp;
            n = m; s <<= 2;
        }
        if (log2N & 1) {   // n == 2 here
            const int half = N >> 1;
            for (int i = threadIdx.x; i < half; i += 256) {
                float2 c0 = src[i];
                float2 c1 = src[i + half];
                dst[i] = cadd(c0, c1);
                dst[i + half] = csub(c0, c1);
            }
            __syncthreads();
            float2 *tmp = src; src = dst; dst = tmp;
        }
        for (int i = threadIdx.x; i < N; i += 256) {
            int k = i + shift;
            if (k >= N) k -= N;
            float2 X = src[k];
            float mag = sqrtf(X.x * X.x + X.y * X.y) + 1e-10f;
            o[i] = 20.0f * log10f(mag);
        }
    } else {
        for (int i = threadIdx.x; i < N; i += 256) {
            int k = i + shift;
            if (k >= N) k -= N;
            float re = 0.f, im = 0.f;
            int idx = 0;
            for (int n = 0; n < N; ++n) {
                float2 y = sm[n];
                float2 w = tw[idx];
                re += y.x * w.x - y.y * w.y;
                im += y.x * w.y + y.y * w.x;
                idx += k;
                if (idx >= N) idx -= N;
            }
            float mag = sqrtf(re * re + im * im) + 1e-10f;
            o[i] = 20.0f * log10f(mag);
        }
    }
}

// rocFFT engine helpers: the FFT itself is rocFFT (through the caller's binding); these two kernels are
// the fused prologue (window) and epilogue (|X|, fftshift, 20 log10) around it.
__global__ __launch_bounds__(256) void spectrum_window_kernel(const float2 *iq, size_t frame_stride, float2 *out,
                                                              const float *window, int N, size_t total) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    size_t f = i / N;
    int n = (int)(i - f * N);
    float2 v = iq[f * frame_stride + n];
    float w = window[n];
    out[i] = make_float2(v.x * w, v.y * w);
}

__global__ __launch_bounds__(256) void spectrum_post_kernel(const float2 *X, float *out, int N, size_t total) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    size_t f = i / N;
    int b = (int)(i - f * N);
    int k = b + (N - N / 2);
    if (k >= N) k -= N;
    float2 v = X[f * N + k];
    out[i] = 20.0f * log10f(sqrtf(v.x * v.x + v.y * v.y) + 1e-10f);
}

}  // namespace

struct wh_spectrum {
    int N, log2N;
    float *d_window = nullptr;
    float2 *d_tw = nullptr;
    float *d_sink = nullptr;   // [N] row for the shaped kernel's stores of frames past the end of a call
    size_t smem = 0;
    bool shaped = false;       // a shaped instance exists for N
    int path = 0;              // wh_spectrum_tune: 0 auto, 1 Stockham / DFT kernel, 2 shaped kernel
    int cu_count = 256;
};

extern "C" int wh_spectrum_create(wh_spectrum **out, int N) {
    if (!out || N < 2 || N > 16384) return set_err(WH_E_ARG, "wh_spectrum_create: fft_size must be in [2, 16384]");
    wh_spectrum *s = new wh_spectrum();
    std::unique_ptr<wh_spectrum, void (*)(wh_spectrum *)> guard(s, wh_spectrum_destroy);  // frees partial state on early return
    s->N = N;
    int l2 = 0;
    while ((1 << l2) < N) ++l2;
    s->log2N = ((1 << l2) == N && N <= 8192) ? l2 : 0;
    std::vector<float> w(N);
    for (int n = 0; n < N; ++n)  // np.hanning(N): 0.5 - 0.5 cos(2 pi n/(N-1)), float64 then cast
        w[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)(N - 1)));
    std::vector<float2> tw(N);
    for (int m = 0; m < N; ++m) {
        double ang = -2.0 * M_PI * (double)m / (double)N;
        tw[m] = make_float2((float)cos(ang), (float)sin(ang));
    }
    WH_HIP(hipMalloc(&s->d_window, (size_t)N * sizeof(float)));
    WH_HIP(hipMalloc(&s->d_tw, (size_t)N * sizeof(float2)));
    WH_HIP(hipMemcpy(s->d_window, w.data(), (size_t)N * sizeof(float), hipMemcpyHostToDevice));
    WH_HIP(hipMemcpy(s->d_tw, tw.data(), (size_t)N * sizeof(float2), hipMemcpyHostToDevice));
    WH_HIP(hipMalloc(&s->d_sink, (size_t)N * sizeof(float)));
    s->shaped = spectrum_mid_supported(N);
    {
        int dev = 0;
        hipDeviceProp_t prop;
        WH_HIP(hipGetDevice(&dev));
        WH_HIP(hipGetDeviceProperties(&prop, dev));
        s->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    s->smem = (size_t)N * sizeof(float2) * (s->log2N ? 2 : 1);
    if (s->smem > 64 * 1024)
        WH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(spectrum_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->smem));
    *out = guard.release();
    return WH_OK;
}

extern "C" void wh_spectrum_destroy(wh_spectrum *s) {
    if (!s) return;
    (void)hipFree(s->d_window);
    (void)hipFree(s->d_tw);
    (void)hipFree(s->d_sink);
    delete s;
}

extern "C" int wh_spectrum_run(wh_spectrum *s, const float *d_iq, size_t n_frames, size_t frame_stride,
                               float *d_power_db, void *stream) {
    if (!s) return set_err(WH_E_ARG, "wh_spectrum_run: null handle");
    if (n_frames == 0) return WH_OK;
    if (!d_iq || !d_power_db) return set_err(WH_E_ARG, "wh_spectrum_run: null buffer");
    if (n_frames > 0x7fffffff) return set_err(WH_E_ARG, "wh_spectrum_run: too many frames");
    if (s->shaped && s->path != 1) {
        SpectrumMidCall c;
        c.x = reinterpret_cast<const float2 *>(d_iq); c.frame_stride = frame_stride; c.n_frames = (long long)n_frames;
        c.out = d_power_db; c.sink = s->d_sink; c.sink_elems = (size_t)s->N; c.window = s->d_window; c.tw = s->d_tw; c.cu_count = s->cu_count;
        return spectrum_mid_launch(s->N, c, as_stream(stream));
    }
    hipLaunchKernelGGL(spectrum_kernel, dim3((unsigned)n_frames), dim3(256), s->smem, as_stream(stream),
                       reinterpret_cast<const float2 *>(d_iq), frame_stride, d_power_db, s->d_window, s->d_tw, s->N,
                       s->log2N);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_spectrum_tune(wh_spectrum *s, int key, int value) {
    if (!s) return set_err(WH_E_ARG, "wh_spectrum_tune: null handle");
    if (key != 1 || value < 0 || value > 2) return set_err(WH_E_ARG, "wh_spectrum_tune: key 1 (kernel), value 0 auto / 1 stockham / 2 shaped");
    if (value == 2 && !s->shaped) return set_err(WH_E_ARG, "wh_spectrum_tune: no shaped kernel for fft_size=%d", s->N);
    s->path = value;
    return WH_OK;
}

extern "C" int wh_spectrum_window(wh_spectrum *s, const float *d_iq, size_t n_frames, size_t frame_stride,
                                  float *d_windowed, void *stream) {
    if (!s || !d_iq || !d_windowed) return set_err(WH_E_ARG, "wh_spectrum_window: null");
    if (n_frames == 0) return WH_OK;
    size_t total = n_frames * (size_t)s->N;
    hipLaunchKernelGGL(spectrum_window_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2 *>(d_iq), frame_stride, reinterpret_cast<float2 *>(d_windowed),
                       s->d_window, s->N, total);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_spectrum_post(wh_spectrum *s, const float *d_fft, size_t n_frames, float *d_power_db, void *stream) {
    if (!s || !d_fft || !d_power_db) return set_err(WH_E_ARG, "wh_spectrum_post: null");
    if (n_frames == 0) return WH_OK;
    size_t total = n_frames * (size_t)s->N;
    hipLaunchKernelGGL(spectrum_post_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2 *>(d_fft), d_power_db, s->N, total);
    WH_LAUNCH_CHECK();
    return WH_OK;
}
