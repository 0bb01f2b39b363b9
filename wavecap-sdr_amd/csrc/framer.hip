// Small "next" rows:
//   N2  P25P1SoftSyncDetector.process_batch (reference decoders/p25_framer.py:124-231): sliding 24-tap
//       correlation of soft symbols with the frame-sync pattern, one score per symbol, state = the last
//       24 symbols; banked over channels.
//   N4  pack_f32 (reference capture.py:134-144): clip to [-1, 1] before the float32 wire format.
#include "wh_common.h"
#include <memory>

using namespace wh;

namespace {

__device__ __forceinline__ float sync_sym(int i) {
    return ((0x5575F5FF77FFULL >> ((23 - i) * 2)) & 3ULL) == 1ULL ? 3.0f : -3.0f;
}

__global__ __launch_bounds__(256) void sync_corr_kernel(const float *soft, size_t stride, int n, const float *hist_in,
                                                        float *scores) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *s = soft + (size_t)c * stride;
    const float *h = hist_in + (size_t)c * 24;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 24; ++j) {
        int k = i - 23 + j;                    // symbol index relative to this call
        float v = k >= 0 ? s[k] : h[24 + k];   // history: the 24 symbols before the call, oldest first
        acc = fmaf(sync_sym(j), v, acc);
    }
    scores[(size_t)c * stride + i] = acc;
}

__global__ void sync_hist_kernel(const float *soft, size_t stride, int n, const float *hist_in, float *hist_out, int C) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * 24) return;
    int c = idx / 24, j = idx % 24;
    int k = n - 24 + j;
    hist_out[idx] = k >= 0 ? soft[(size_t)c * stride + k] : hist_in[c * 24 + 24 + k];
}

__global__ void clip_kernel(const float *in, float *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        float v = in[i];
        out[i] = v != v ? v : fminf(fmaxf(v, -1.0f), 1.0f);   // np.clip keeps NaN
    }
}

// validate_audio_samples (validation.py:41-52) + the signal-power metric (capture.py:436-437) in one pass.
__global__ __launch_bounds__(256) void audio_stats_kernel(const float *x, size_t n, float *out) {
    double p = 0.0;
    float mx = 0.f;
    int bad = 0;
    for (size_t i = threadIdx.x; i < n; i += 256) {
        float v = x[i];
        p += (double)v * (double)v;
        mx = fmaxf(mx, fabsf(v));
        if (!(fabsf(v) <= 3.4028235e38f)) bad = 1;
    }
    __shared__ double sp[256];
    __shared__ float sm[256];
    __shared__ int sb[256];
    sp[threadIdx.x] = p; sm[threadIdx.x] = mx; sb[threadIdx.x] = bad;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            sp[threadIdx.x] += sp[threadIdx.x + w];
            sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + w]);
            sb[threadIdx.x] |= sb[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (float)(sp[0] / (double)n);
        out[1] = sm[0];
        out[2] = sb[0] ? 0.f : 1.f;
    }
}

// ChannelClassifier.update (channel_classifier.py:100-125): per-bin running {sum, sum_sq, count, min, max} over
// spectrum frames, float64, frames added in order -- the same additions in the same order as the reference's
// Python-float BinStats, so the accumulators are bit-identical.
__global__ __launch_bounds__(256) void binstats_kernel(const float *power_db, size_t n_frames, int N, double *stats) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= N) return;
    double *s = stats + (size_t)b * 5;
    double sum = s[0], sq = s[1], cnt = s[2], mn = s[3], mx = s[4];
    for (size_t f = 0; f < n_frames; ++f) {
        const double v = (double)power_db[f * N + b];
        sum = __dadd_rn(sum, v);
        sq = __dadd_rn(sq, __dmul_rn(v, v));
        cnt += 1.0;
        if (v < mn) mn = v;
        if (v > mx) mx = v;
    }
    s[0] = sum; s[1] = sq; s[2] = cnt; s[3] = mn; s[4] = mx;
}

// ---- N2: NID front half (decoders/p25_framer.py:475-617, dsp/fec/bch.py) -------------------------------------
// positions where the soft sync score exceeds the threshold (p25_framer.py:493), unordered (host sorts)
__global__ void sync_positions_kernel(const float *scores, int n, float threshold, int32_t *out, int cap, int32_t *count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (scores[i] > threshold) {
        int k = atomicAdd(count, 1);
        if (k < cap) out[k] = i;
    }
}

// NID word of the 33 dibits starting at `start` (p25_framer.py:587-597): status dibit at index 11 dropped, 32 dibits ->
// 64 bits MSB first, the first 63 form the BCH word (bit 62 = first bit).  Starts whose 33 dibits are not all inside
// [0, n) give ~0 (never a codeword neighbour).
__global__ void nid_extract_kernel(const uint8_t *dibits, int n, const int32_t *starts, int n_starts, uint64_t *words) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_starts) return;
    int s0 = starts[k];
    if (s0 < 0 || s0 + 33 > n) { words[k] = ~0ULL; return; }
    uint64_t w = 0;
    for (int j = 0; j < 33; ++j) {
        if (j == 11) continue;
        w = (w << 2) | (uint64_t)(dibits[s0 + j] & 3);
    }
    words[k] = w >> 1;   // drop the 64th bit
}

// status-symbol stripping (decoders/p25.py:2796-2862 _strip_status_symbols): a counter starts at c0, is incremented per
// dibit, and the dibit on which it reaches 36 is a status symbol (dropped, counter back to 0) -- i.e. the inputs
// s0 + 36 k with s0 = 35 - c0 are dropped (none when c0 > 35: the counter never meets 36).  Closed form per OUTPUT j:
// input j below s0, j + 1 + (j - s0) / 35 from there on.  One thread per output, rows batched over blockIdx.y.
__global__ void strip_status_kernel(const uint8_t *in, size_t in_stride, long long s0, long long n_out, uint8_t *out,
                                    size_t out_stride) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    const long long i = (s0 < 0 || j < s0) ? j : j + 1 + (j - s0) / 35;
    out[(size_t)blockIdx.y * out_stride + j] = in[(size_t)blockIdx.y * in_stride + i];
}

// BCH(63,16,23) bounded-distance decoding by exhaustive nearest-codeword search: d_min = 23, so at most one of the
// 65536 codewords lies within 11 bit errors of a word -- exactly the words the reference's Berlekamp-Massey / Chien /
// re-check chain (bch.py:575-638) corrects, with the same (data, error count).  One workgroup per word; pass 2
// (bch.py:556-571) retries with the tracked NAC written over the first 12 bits.
__device__ __forceinline__ unsigned bch_search(const uint64_t *cw, uint64_t w, unsigned *red) {
    unsigned best = 0xffffffffu;   // (distance << 16) | data
    for (int i = threadIdx.x; i < 65536; i += 256) {
        unsigned d = (unsigned)__popcll(cw[i] ^ w);
        unsigned key = (d << 16) | (unsigned)i;
        best = key < best ? key : best;
    }
    for (int o = 32; o > 0; o >>= 1) {
        unsigned other = __shfl_xor(best, o);
        best = other < best ? other : best;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    unsigned b = red[0];
    for (int k = 1; k < 4; ++k) b = red[k] < b ? red[k] : b;
    return b;
}

__global__ __launch_bounds__(256) void bch_decode_kernel(const uint64_t *cw, const uint64_t *words, const int32_t *tracked,
                                                         int32_t *data, int32_t *errors) {
    __shared__ unsigned red[4];
    const int k = blockIdx.x;
    uint64_t w = words[k] & 0x7fffffffffffffffULL;
    unsigned b = bch_search(cw, w, red);
    int dat = 0, err = -1;
    if ((b >> 16) <= 11) { dat = (int)(b & 0xffff); err = (int)(b >> 16); }
    else {
        const int tn = tracked ? tracked[k] : 0;
        const int cur = (int)((w >> 51) & 0xfff);
        if (tn > 0 && cur != tn) {
            uint64_t w2 = (w & ((1ULL << 51) - 1)) | ((uint64_t)(tn & 0xfff) << 51);
            b = bch_search(cw, w2, red);
            if ((b >> 16) <= 11) { dat = (int)(b & 0xffff); err = (int)(b >> 16); }
        }
    }
    if (threadIdx.x == 0) { data[k] = dat; errors[k] = err; }
}

}  // namespace

struct wh_bch {
    uint64_t *d_cw = nullptr;
};

extern "C" int wh_bch_create(wh_bch **out, const uint64_t *h_codewords) {
    if (!out || !h_codewords) return set_err(WH_E_ARG, "wh_bch_create: null");
    wh_bch *b = new wh_bch();
    std::unique_ptr<wh_bch, void (*)(wh_bch *)> guard(b, wh_bch_destroy);  // frees partial state on early return
    WH_HIP(hipMalloc(&b->d_cw, 65536 * sizeof(uint64_t)));
    WH_HIP(hipMemcpy(b->d_cw, h_codewords, 65536 * sizeof(uint64_t), hipMemcpyHostToDevice));
    *out = guard.release();
    return WH_OK;
}

extern "C" void wh_bch_destroy(wh_bch *b) {
    if (!b) return;
    (void)hipFree(b->d_cw);
    delete b;
}

extern "C" int wh_bch_decode(wh_bch *b, const uint64_t *d_words, size_t n, const int32_t *d_tracked_nac, int32_t *d_data,
                             int32_t *d_errors, void *stream) {
    if (!b) return set_err(WH_E_ARG, "wh_bch_decode: null handle");
    if (n == 0) return WH_OK;
    if (!d_words || !d_data || !d_errors || n > 0x7fffffff) return set_err(WH_E_ARG, "wh_bch_decode: bad arguments");
    hipLaunchKernelGGL(bch_decode_kernel, dim3((unsigned)n), dim3(256), 0, as_stream(stream), b->d_cw, d_words,
                       d_tracked_nac, d_data, d_errors);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_sync_positions(const float *d_scores, size_t n, float threshold, int32_t *d_positions, size_t cap,
                                 int32_t *d_count, void *stream) {
    if (!d_scores || !d_positions || !d_count || n > 0x7fffffff || cap > 0x7fffffff)
        return set_err(WH_E_ARG, "wh_sync_positions: bad arguments");
    hipStream_t st = as_stream(stream);
    WH_HIP(hipMemsetAsync(d_count, 0, sizeof(int32_t), st));
    if (n == 0) return WH_OK;
    hipLaunchKernelGGL(sync_positions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scores, (int)n,
                       threshold, d_positions, (int)cap, d_count);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_nid_extract(const uint8_t *d_dibits, size_t n, const int32_t *d_starts, size_t n_starts,
                              uint64_t *d_words, void *stream) {
    if (n_starts == 0) return WH_OK;
    if (!d_dibits || !d_starts || !d_words || n > 0x7fffffff || n_starts > 0x7fffffff)
        return set_err(WH_E_ARG, "wh_nid_extract: bad arguments");
    hipLaunchKernelGGL(nid_extract_kernel, dim3((unsigned)((n_starts + 255) / 256)), dim3(256), 0, as_stream(stream),
                       d_dibits, (int)n, d_starts, (int)n_starts, d_words);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_strip_status(const uint8_t *d_dibits, size_t n, size_t in_stride, int n_rows, int initial_counter,
                               uint8_t *d_out, size_t out_stride, size_t *n_out, void *stream) {
    if (!n_out || n_rows < 0 || n > 0x7fffffff) return set_err(WH_E_ARG, "wh_strip_status: bad arguments");
    // dropped inputs: s0 + 36 k < n (s0 = 35 - initial_counter; none when the counter starts above 35)
    const long long s0 = initial_counter <= 35 ? 35LL - initial_counter : -1;
    const long long dropped = (s0 >= 0 && (long long)n > s0) ? ((long long)n - 1 - s0) / 36 + 1 : 0;
    const long long kept = (long long)n - dropped;
    *n_out = (size_t)kept;
    if (kept == 0 || n_rows == 0) return WH_OK;
    if (!d_dibits || !d_out || in_stride < n || out_stride < (size_t)kept)
        return set_err(WH_E_ARG, "wh_strip_status: null buffer or stride shorter than the row");
    hipLaunchKernelGGL(strip_status_kernel, dim3((unsigned)((kept + 255) / 256), (unsigned)n_rows), dim3(256), 0,
                       as_stream(stream), d_dibits, in_stride, s0, kept, d_out, out_stride);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_binstats_update(const float *d_power_db, size_t n_frames, int n_bins, double *d_stats, void *stream) {
    if (n_frames == 0) return WH_OK;
    if (!d_power_db || !d_stats || n_bins < 1) return set_err(WH_E_ARG, "wh_binstats_update: bad arguments");
    hipLaunchKernelGGL(binstats_kernel, dim3((n_bins + 255) / 256), dim3(256), 0, as_stream(stream), d_power_db, n_frames,
                       n_bins, d_stats);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_audio_stats(const float *d_x, size_t n, float *h_out, void *stream) {
    if (!d_x || !h_out || n == 0) return set_err(WH_E_ARG, "wh_audio_stats: bad arguments");
    hipStream_t st = as_stream(stream);
    float *d_out = nullptr;
    WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_out), 3 * sizeof(float), st));
    hipLaunchKernelGGL(audio_stats_kernel, dim3(1), dim3(256), 0, st, d_x, n, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_out, d_out, 3 * sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFreeAsync(d_out, st);
    if (e != hipSuccess) return set_err(WH_E_HIP, hipGetErrorString(e));
    return WH_OK;
}

extern "C" int wh_sync_correlate(const float *d_soft, size_t n, size_t stride, int n_channels, const float *d_hist_in,
                                 float *d_hist_out, float *d_scores, void *stream) {
    if (n == 0) return WH_OK;
    if (!d_soft || !d_hist_in || !d_hist_out || !d_scores || n_channels < 1 || n_channels > 65535 || stride < n ||
        n > 0x7fffffff || d_hist_in == d_hist_out)
        return set_err(WH_E_ARG, "wh_sync_correlate: bad arguments");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(sync_corr_kernel, dim3((unsigned)((n + 255) / 256), n_channels), dim3(256), 0, st, d_soft, stride,
                       (int)n, d_hist_in, d_scores);
    WH_LAUNCH_CHECK();
    hipLaunchKernelGGL(sync_hist_kernel, dim3((n_channels * 24 + 255) / 256), dim3(256), 0, st, d_soft, stride, (int)n,
                       d_hist_in, d_hist_out, n_channels);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_clip_f32(const float *d_in, float *d_out, size_t n, void *stream) {
    if (n == 0) return WH_OK;
    if (!d_in || !d_out) return set_err(WH_E_ARG, "wh_clip_f32: null buffer");
    size_t g = (n + 255) / 256;
    hipLaunchKernelGGL(clip_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, as_stream(stream), d_in, d_out, n);
    WH_LAUNCH_CHECK();
    return WH_OK;
}
