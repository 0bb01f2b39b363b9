// Internal (not exported) interface between pfb.hip (handle, C ABI) and pfb_mid.hip (the compile-time-shaped
// filterbank kernels for the mid-size / large channel counts).
#pragma once
#include <hip/hip_runtime.h>

namespace wh {

struct PfbMidCall {
    const void *x;           // input samples (complex64, or interleaved int16 IQ when fmt == 1)
    int fmt;
    size_t n;                // samples in this call
    const float2 *hist;      // [M][T] carried history (column j = block_{-1-j}); read
    float2 *new_hist;        // [M][T] history after this call; written
    float2 *out;             // [H][M]
    float2 *sink;            // [sink_elems >= M] scratch row nobody reads (stores of hops past the end of the call)
    size_t sink_elems;       // checked at launch: a sink smaller than a row would put those stores outside any allocation
    const float *arms;       // float32 [M][T]
    const float2 *tw;        // exp(-2 pi i m / M), m in [0, M)
    long long H;             // hops of this call (> 0)
    int cu_count;
    int hops_per_run;        // 0 = choose; tuning override otherwise
    int stats_only;          // bit 0: statistics-only mode (no channel outputs are stored)
    double *stats_ws;        // statistics-only: workspace [workgroups][4][M] (sized from the planned grid)
    double *stats_out;       // statistics-only: float64 [M][5] = {sum p, sum p^2, hops, min p, max p}
    int stats_accumulate;    // merge into stats_out instead of overwriting
};

// ---- the activity statistic of a channel (A13; the repo's own definition, consumed by the scanner / classifier as dB
// levels): p = float32(float32(re^2) + float32(im^2)) of every output; {sum p, sum p^2} are taken in float32 over short
// blocks of hops (at most 16 of a lane's consecutive visits) and the block sums added in float64; min / max are exact
// float32.  Every producer (the statistics-only kernels and wh_pfb_channel_stats over a stored output) computes the
// same p bit for bit, so count, min and max agree exactly; the sums agree to the float32 block rounding (<= 2e-6
// relative, typically 2e-7): the grouping into blocks is the kernel's business and not part of the definition.
// (Round 2 took p, p^2 and the four reductions in float64: 16 issue slots per output against ~5 here.)
struct StAcc {
    double s, s2;          // float64 sums of the folded blocks
    float fs, fs2, mn, mx; // open block (float32), exact extremes
};
__device__ __forceinline__ void stacc_init(StAcc &a) { a.s = 0.0; a.s2 = 0.0; a.fs = 0.f; a.fs2 = 0.f; a.mn = INFINITY; a.mx = 0.f; }
__device__ __forceinline__ void stacc_fold(StAcc &a) {
    a.s += (double)a.fs; a.s2 += (double)a.fs2;
    a.fs = 0.f; a.fs2 = 0.f;
}
__device__ __forceinline__ float stat_power(float re, float im) {
    return __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));   // no contraction into an FMA: every producer gets the same bits
}
__device__ __forceinline__ void stacc_add(StAcc &a, float pw) {
    a.fs = __fadd_rn(a.fs, pw);
    a.fs2 = fmaf(pw, pw, a.fs2);
    a.mn = fminf(a.mn, pw);
    a.mx = fmaxf(a.mx, pw);
}
// workgroup rows [rows][4][M] = {sum p, sum p^2, min, max} (float64) -> d_stats [M][5] = {sum p, sum p^2, hops, min, max};
// row `skip` (or -1) holds nothing
int pfb_stats_rows_reduce(const double *ws, int rows, int skip, int M, double hops, double *stats, int accumulate,
                          hipStream_t st);

// true when a compiled instance exists for (M, T)
bool pfb_mid_supported(int M, int T);
// one launch: head hops (carried history), the runs, and the history update.  Returns a WH_* status.  With grid_out
// nothing is launched and the planned grid size is returned (the statistics workspace has one row per workgroup).
int pfb_mid_launch(int M, int T, const PfbMidCall &c, hipStream_t st, long long *grid_out = nullptr);
// spectrum frames (window -> FFT -> fftshift -> dB) through the shaped passes
struct SpectrumMidCall {
    const float2 *x;         // samples; frame f starts at x + f * frame_stride
    size_t frame_stride;
    long long n_frames;      // > 0
    float *out;              // [n_frames][N] dB
    float *sink;             // [sink_elems >= N] scratch row nobody reads
    size_t sink_elems;
    const float *window;     // [N]
    const float2 *tw;        // exp(-2 pi i m / N)
    int cu_count;
};
bool spectrum_mid_supported(int N);
int spectrum_mid_launch(int N, const SpectrumMidCall &c, hipStream_t st);
// kernel name fragment for profiles / bench reporting
const char *pfb_mid_kernel_name();

}  // namespace wh
