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
