/*
 * wavehip.h -- C ABI of libwavehip.so, the MI355X (gfx950) implementation of
 * WaveCap-SDR's per-channel DSP hot path.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HBM) owned by the caller; every
 *     pointer named h_* is a host pointer read during the call only;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only
 *     enqueue work, they never synchronise the device;
 *   - return value: 0 = ok, <0 = WH_E_* ; wh_last_error() gives a thread-local text;
 *   - no global mutable state: handles may be used from different host threads as
 *     long as one handle is used by one thread at a time (the reference calls the
 *     channel operator from a 3-thread pool, capture.py:1906-1925);
 *   - complex samples are interleaved float32 (re, im) == numpy complex64.
 *
 * Every entry point cites the reference interface it replaces (paths relative to
 * the reference checkout, backend/wavecapsdr/...).
 */
#ifndef WAVEHIP_H
#define WAVEHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WH_OK 0
#define WH_E_ARG (-1)      /* bad argument / unsupported size */
#define WH_E_HIP (-2)      /* HIP runtime error (text in wh_last_error) */
#define WH_E_NOMEM (-3)

int wh_abi_version(void);            /* bumps when a signature changes */
const char *wh_last_error(void);     /* thread-local, never NULL */
int wh_device_info(int *cu_count, int *lds_bytes, char *name, size_t name_len);

/* ---- A1: int16 IQ wire conventions ------------------------------------------------
 * unpack: cli.py:447-452, harness.py:274   (int16 -> f32 / 32768.0, I,Q interleaved)
 * pack  : capture.py:102-116 pack_iq16     (clip(-1,1) * 32767.0 -> int16, truncation)
 * pcm16 : capture.py:119-131 pack_pcm16                                              */
int wh_unpack_i16_cf32(const int16_t *d_in, float *d_out, size_t n_complex, void *stream);
int wh_pack_cf32_i16(const float *d_in, int16_t *d_out, size_t n_complex, void *stream);
int wh_pack_f32_pcm16(const float *d_in, int16_t *d_out, size_t n, void *stream);

/* pack_f32, capture.py:134-144: clip to [-1, 1] (the bytes are the float32 array itself) */
int wh_clip_f32(const float *d_in, float *d_out, size_t n, void *stream);

/* validate_audio_samples (validation.py:41-52) + signal power (capture.py:436-437) of a float32 device
 * buffer: h_out[3] = {mean(x^2), max |x|, 1 if all finite else 0}.  Synchronous.                      */
int wh_audio_stats(const float *d_x, size_t n, float *h_out, void *stream);

/* noise_blanker, dsp/filters.py:267-343 (real float32): samples whose magnitude exceeds median|x| * threshold_factor
 * (= float32(10^(threshold_db/20))) and blanking_width neighbours on each side are zeroed; d_out != d_in.        */
int wh_noise_blanker(const float *d_x, float *d_out, size_t n, float threshold_factor, int blanking_width, void *stream);

/* ---- A2: stateless NCO mix, capture.py:166-193 freq_shift -------------------------
 * phase[n] = f32(-2 pi off/fs) * f32(n) (float32 product), restarted every call.     */
int wh_nco_mix(const float *d_iq, float *d_out, size_t n, int offset_hz, int sample_rate,
               void *stream);

/* ---- A3: FM discriminator, dsp/fm.py:65-97 quadrature_demod ----------------------- */
int wh_fm_discriminate(const float *d_iq, float *d_out, size_t n, int sample_rate, void *stream);

/* ---- A6: rational resampler, dsp/fm.py:184-221 resample_poly ----------------------
 * h_taps: float64[ntaps] = firwin(...)*up exactly as scipy.signal.resample_poly builds
 * them; d0: alignment so that y[m] = sum_j h[j]*xup[m*down + d0 - j].
 * d_x: float32[batch][n_in]  ->  d_y: float32[batch][n_out].                         */
typedef struct wh_resampler wh_resampler;
int wh_resampler_create(wh_resampler **out, const double *h_taps, int ntaps, int up, int down, int d0);
int wh_resampler_run(wh_resampler *r, const float *d_x, size_t n_in, size_t batch, float *d_y,
                     size_t n_out, void *stream);
void wh_resampler_destroy(wh_resampler *r);

/* ---- Channel operator, batched: capture.py:298-439 _process_channel_dsp_stateless
 * for the analog modes nbfm / wbfm / am / ssb over K channels of one capture (replaces the
 * per-channel ThreadPoolExecutor fan-out of capture.py:2489-2597).
 *   input : one chunk-major IQ buffer shared by all channels, int16 interleaved
 *           (format 1, A1 unpack rule fused) or complex64 (format 0);
 *   output: d_audio float32[n_chunks][K][n_out], d_metrics float32[n_chunks][K][4] =
 *           {rssi_db, signal_power_db, max_abs_audio, finite_flag} (capture.py:331-334,
 *           432-437; validation.py:41-52 is applied by the host shim from the last two).
 * Chain: NCO mix (capture.py:166-193) -> demod front -> IIR stages -> [AGC] -> resample_poly ->
 * post.  The host designs every filter exactly as the reference does and passes coefficients:
 *   demod 0 FM discriminator (dsp/fm.py:65-97), 1 AM envelope (dsp/am.py:103),
 *         2 SSB product detector with float64-phase BFO at bfo_hz (dsp/am.py:23-42, 200-210),
 *         3 / 4 / 5 synchronous AM dsb / usb / lsb: carrier-recovery PLL, fresh per chunk, float64
 *         (dsp/sam.py:73-122, 219-232) with loop coefficients pll_alpha / pll_beta (dsp/sam.py:55-66);
 *   stages: lfilter(b, a) sections in application order, float32 (de-emphasis, dsp/fm.py:101-126)
 *         or float64 (Butterworth / notch, dsp/filters.py:86-264, dsp/fm.py:129-181), zero state
 *         per chunk, output of each rounded to float32;
 *   agc : dsp/agc.py:169-242 (float32 attack/release one-poles, gain cap, tanh soft clip);
 *   post 0: rms_normalize(0.18) + FM soft clip x0.95 (dsp/fm.py:26-62) -- FM modes
 *        1: AM/SSB: nothing after AGC, agc.soft_clip when agc == 0 (dsp/am.py:135-139);
 *   ntaps == 0: sample_rate == audio_rate, no resampling (dsp/fm.py:198-199), n_out == chunk_len. */
typedef struct wh_iir_stage {
    int is_f64;             /* 1: float64 recurrence, 0: float32 */
    int n;                  /* max(len(b), len(a)) <= 11, shorter one zero padded */
    double b[11], a[11];
} wh_iir_stage;
typedef struct wh_chanbank wh_chanbank;
typedef struct wh_chanbank_cfg {
    int sample_rate;
    int chunk_len;          /* N samples per chunk (capture.py:3035) */
    int n_channels;
    const int *h_offsets_hz;/* round(offset_hz) per channel; 0 = no mix (capture.py:328) */
    int input_format;       /* 0 = complex64, 1 = int16 IQ */
    int demod;              /* 0 FM, 1 AM, 2 SSB, 3/4/5 SAM dsb/usb/lsb */
    double bfo_hz;          /* demod 2: +bfo (USB) / -bfo (LSB) */
    int n_stages;           /* <= 8 */
    const wh_iir_stage *h_stages;
    int agc;
    float agc_target, agc_max_gain;                 /* float32(10^(target_db/20)), float32(10^(60/20)) */
    float agc_att_b0, agc_att_a1, agc_rel_b0, agc_rel_a1; /* float32 one-pole coefficient arrays */
    int post;               /* 0 FM, 1 AM/SSB */
    const double *h_taps;   /* resample_poly taps (see wh_resampler_create); NULL when ntaps == 0 */
    int ntaps, up, down, d0;
    int n_out;              /* ceil(N*up/down), or N when ntaps == 0 */
    double pll_alpha, pll_beta; /* demod 3..5: 2*damping*omega_n/fs, omega_n^2/fs^2 */
    int noise_reduction;    /* post 0 only: spectral_noise_reduction (dsp/filters.py:346-460) between the IIR stages and
                               rms_normalize; chunks shorter than 1024 pass through; otherwise the row shrinks to
                               ((N-1024)/512)*512 + 1024 samples and n_out must be computed from that length */
    float nr_reduction_linear;      /* float32(10^(noise_reduction_db/20)) */
    const float *h_nr_window;       /* float32[1024] scipy.signal.windows.hann(1024, sym=False) */
    int iir_warmup;                 /* > 0: every IIR stage's impulse response has decayed below 1e-10 after this many
                                       samples (derived by the host from the pole radii); lets chains without AGC run
                                       time-parallel (64 segments per row, each warmed up over iir_warmup samples);
                                       0 = strictly sequential recurrences */
    int iir_scan;                   /* 1: the chain may run as an exact linear-recurrence scan (64, 256 or 512 segments per row; per
                                       stage: zero-state pass, start states from s' = M s + e with M = transition^segment,
                                       output pass; the AGC envelopes likewise).  The host sets it only when every M is
                                       well conditioned: the reference's order-5 ba-form high-/band-passes are NOT (their
                                       DF2T states cancel over ~1e6), those chains stay sequential */
    int iir_warmup_form;            /* time-parallel warm-up form: 0 = the segments' start states are the dot product of the
                                       iir_warmup samples before the segment with the chain's impulse-response states
                                       (produced at creation by the recurrences themselves); 1 = by running the recurrences
                                       over those samples (the same truncation, 16x the arithmetic; kept for comparison) */
    const float *h_squelch_db;      /* optional float32[n_channels] (NaN = none): audio of a chunk whose rssi_db is below
                                       the channel's threshold is zeroed (capture.py:2918-2921); metrics unaffected */
} wh_chanbank_cfg;
int wh_chanbank_create(wh_chanbank **out, const wh_chanbank_cfg *cfg);
int wh_chanbank_run(wh_chanbank *b, const void *d_in, size_t n_chunks, float *d_audio, float *d_metrics,
                    void *stream);
/* same, and the audio additionally leaves in its wire format (N4, capture.py:119-144), written by the finalize kernel:
 * wire_format 1: d_wire int16 [n_chunks][K][n_out] by the pack_pcm16 rule (clip to [-1, 1], x 32767, truncate);
 * wire_format 2: d_wire float32 clipped to [-1, 1] (pack_f32); 0: none (d_wire ignored).  Squelched rows are zeros in
 * both buffers.  Only d_wire (and the 16-byte metrics rows) need to cross PCIe. */
int wh_chanbank_run_wire(wh_chanbank *b, const void *d_in, size_t n_chunks, float *d_audio, float *d_metrics,
                         int wire_format, void *d_wire, void *stream);
/* Retune / re-squelch a live bank (API PATCH of a channel's offset_hz or squelch_db -> capture.py:442-501 cfg update;
 * the reference's stateless operator reads cfg.offset_hz on every chunk, capture.py:326-329): one small host-to-device
 * copy enqueued on `stream` -- launches queued before it keep the old values -- no allocation, no synchronisation.
 * h_offsets_hz: round(offset_hz) per channel, 0 = no mix; h_squelch_db: NaN = none (bank created with h_squelch_db). */
int wh_chanbank_set_offsets(wh_chanbank *b, const int *h_offsets_hz, int n_channels, void *stream);
int wh_chanbank_set_squelch(wh_chanbank *b, const float *h_squelch_db, int n_channels, void *stream);
size_t wh_chanbank_workspace_bytes(const wh_chanbank *b, size_t n_chunks);
void wh_chanbank_destroy(wh_chanbank *b);

/* ---- N1: Channel.update_signal_metrics, capture.py:749-798, for K channels of one chunk -----------
 * h_out float32 [K][3] = {rssi_db, |base| of rank n//10, |base| of rank n - n//10 - 1}: the RSSI and the
 * two exact order statistics the reference takes with np.partition (noise floor / signal peak).
 * Synchronous (returns host scalars).                                                         */
int wh_channel_signal_metrics(const void *d_in, int input_format, size_t n, int sample_rate,
                              const int *h_offsets_hz, int n_channels, float *h_out, void *stream);

/* ---- A7: polyphase channelizer, dsp/channelizer.py:28-158 PolyphaseChannelizer ----
 * h_arms: float64[M][T] exactly as _design_filter builds them (channelizer.py:69-89).
 * run(): processes (n-M)/(M/2)+1 hops, writes complex64 d_out[hops][M] (row h == the
 * reference's results[h]), updates the carried arm history (9 columns, col j = block
 * h-1-j) and optionally accumulates per-channel activity statistics.                 */
typedef struct wh_pfb wh_pfb;
int wh_pfb_create(wh_pfb **out, int channel_count, int taps_per_channel, const double *h_arms);
size_t wh_pfb_hops(const wh_pfb *p, size_t n_samples);
int wh_pfb_run(wh_pfb *p, const float *d_iq, size_t n_samples, float *d_out, void *stream);
/* same with interleaved int16 IQ input (A1 unpack rule int16/32768 fused into the loads: 4 B read
 * per sample instead of 8) */
int wh_pfb_run_i16(wh_pfb *p, const int16_t *d_iq16, size_t n_samples, float *d_out, void *stream);
int wh_pfb_reset(wh_pfb *p, void *stream);                       /* channelizer.py:139-142 */
int wh_pfb_get_history(wh_pfb *p, float *h_hist /* c64[M][T] */, void *stream);
int wh_pfb_set_history(wh_pfb *p, const float *h_hist, void *stream);
/* Explicit tuning / test switches (no reference counterpart; nothing in the library reads environment variables for
 * these).  PATH: 0 = automatic, 1 = one-workgroup-per-hop kernel only, 2 = run kernel (one wave per run of hops),
 * 3 = kernel shaped at compile time for this channel count; PREFETCH (M = 1024): 0 = automatic, 1 = registers,
 * 3 = LDS DMA with counted waits, 5 / 7 = the same two at three workgroups per CU; HOPS_PER_RUN: 0 = automatic;
 * RUN_MAP (M = 1024): which run of hops a workgroup walks -- 0 = automatic, -1 = run index = workgroup index, -2 = one
 * contiguous range of runs per XCD, C > 0 = chunks of C consecutive runs per XCD, chunks round the XCDs; ALT_DIR
 * (M = 1024, default 1): odd runs are walked downwards, so that neighbouring runs meet at their common border at the same
 * time (halo loads become L2 hits).  Outputs do not depend on any of these. */
#define WH_PFB_TUNE_PATH 1
#define WH_PFB_TUNE_PREFETCH 2
#define WH_PFB_TUNE_HOPS_PER_RUN 3
#define WH_PFB_TUNE_RUN_MAP 5
#define WH_PFB_TUNE_ALT_DIR 6
int wh_pfb_tune(wh_pfb *p, int key, int value);
/* measurement aid (bench.py roofline): when enabled, run() brackets the main filterbank kernel
 * with HIP events on the caller's stream; kernel_ms() waits for and returns the duration of
 * the most recent one. */
int wh_pfb_profile(wh_pfb *p, int enable);
int wh_pfb_kernel_ms(wh_pfb *p, float *ms);
/* the same for the launch `back` launches before the most recent one (0 = most recent; the handle keeps the event pairs of
 * its last 64 profiled launches): a caller times every launch of a run without synchronising between them */
int wh_pfb_kernel_ms_back(wh_pfb *p, int back, float *ms);
/* extract_channel, channelizer.py:144-158: d_col[h] = d_out[h][idx] */
int wh_pfb_extract_channel(const float *d_out, size_t hops, int channel_count, int idx, float *d_col,
                           void *stream);
/* A13 activity statistics over a filterbank output block (BinStats fields of
 * channel_classifier.py:17-48 per channel): d_stats float64[M][5] =
 * {sum p, sum p^2, count, min p, max p}, p = |y|^2; accumulate != 0 merges into d_stats. */
int wh_pfb_channel_stats(wh_pfb *p, const float *d_out, size_t hops, double *d_stats, int accumulate,
                         void *stream);

/* Statistics-only pass (the scanner's / classifier's real workload: scanner.py:203-208, channel_classifier.py:100-125 need
 * per-channel power statistics, not samples): the filterbank runs as in wh_pfb_run / wh_pfb_run_i16 (input_format 0 / 1),
 * carries its history, but the last FFT pass reduces |y|^2 into {sum, sum of squares, min, max} in registers instead of
 * storing the 16 bytes per input sample of channel outputs; d_stats float64 [M][5] as wh_pfb_channel_stats over ALL hops
 * of the call (accumulate != 0 merges).  Sums are added in a different order than wh_pfb_channel_stats (agreement 1e-12
 * relative, not bitwise).  Channel counts with a shaped kernel only (64 ... 2048 of the table; else WH_E_ARG). */
int wh_pfb_run_stats(wh_pfb *p, const void *d_iq, int input_format, size_t n_samples, double *d_stats, int accumulate,
                     void *stream);

/* Cross-stream merge of the activity statistics (cc_scanner.py:355-400 / channel_classifier.py:17-48
 * across device streams): d_gathered float64 [n_ranks][n_channels][5] -- the blocks wh_pfb_channel_stats wrote on every
 * GPU, brought together by ONE all-gather (RCCL ncclAllGather on these caller-owned buffers, or torch.distributed) --
 * -> d_out [n_channels][5] = {sum, sum, sum, min, max} in fixed rank order (identical bits on every rank).  The library
 * itself makes no collective call: the communicator belongs to the host program. */
int wh_stats_merge(const double *d_gathered, int n_ranks, int n_channels, double *d_out, void *stream);

/* Diagnostics (no reference counterpart): the filterbank's HBM traffic shape with no arithmetic -- reads n complex64
 * from d_in, writes them twice (2n complex64) to d_out; n even.  bench.py times it beside the filterbank as the
 * in-process yardstick for a 1 : 2 read : write stream. */
int wh_diag_stream_1r2w(const float *d_in, float *d_out, size_t n, void *stream);
/* the same with FOUR copies written (4n complex64 to d_out): the 1 : 4 read : write stream of the int16-input filterbank */
int wh_diag_stream_1r4w(const float *d_in, float *d_out, size_t n, void *stream);

/* ---- A8: spectrum, dsp/fft/scipy_backend.py:38-79 ScipyFFTBackend.execute ---------
 * d_iq: complex64, frame f starts at d_iq + f*frame_stride (complex units); uses the
 * first fft_size samples; d_power_db float32[n_frames][fft_size], fft-shifted,
 * 20*log10(|X| + 1e-10) with a symmetric Hann window (dsp/fft/base.py:54-59).        */
typedef struct wh_spectrum wh_spectrum;
int wh_spectrum_create(wh_spectrum **out, int fft_size);
int wh_spectrum_run(wh_spectrum *s, const float *d_iq, size_t n_frames, size_t frame_stride,
                    float *d_power_db, void *stream);
/* Explicit kernel selection for tests / measurements: key 1, value 0 auto (default), 1 the Stockham / direct-DFT kernel,
 * 2 the shaped kernel (fft_size 256 .. 4096; WH_E_ARG when the size has none). */
int wh_spectrum_tune(wh_spectrum *s, int key, int value);
/* rocFFT engine: window prologue -> (rocFFT C2C forward of length fft_size, batched, run by the caller's
 * binding) -> |X| / fftshift / 20 log10 epilogue.  d_windowed, d_fft: complex64 [n_frames][fft_size]. */
int wh_spectrum_window(wh_spectrum *s, const float *d_iq, size_t n_frames, size_t frame_stride,
                       float *d_windowed, void *stream);
int wh_spectrum_post(wh_spectrum *s, const float *d_fft, size_t n_frames, float *d_power_db, void *stream);
void wh_spectrum_destroy(wh_spectrum *s);

void wh_pfb_destroy(wh_pfb *p);

/* ---- A4/A5: trunking front-end: phase-continuous float64-phase NCO
 * (trunking/system.py:1434-1466) + two-stage decimating FIR (system.py:1392-1406, 1753-1779;
 * dsp/filters.py:558-646 fir_decimate), kept outputs only.  h_taps*: float64 firwin designs;
 * decim2 <= 1 -> single stage.  run(): d_iq complex64[n] -> d_out complex64[wh_ddc_out_len(n)];
 * offset_hz == 0 skips the mix; a changed offset restarts the phase (system.py:1450-1452);
 * the filter state (input history; first call = lfilter_zi * x[0]) and the sample index
 * (wrapped at one second) are carried; reset() == the overflow recovery of system.py:1574-1588. */
typedef struct wh_ddc wh_ddc;
int wh_ddc_create(wh_ddc **out, int sample_rate, const double *h_taps1, int ntaps1, int decim1,
                  const double *h_taps2, int ntaps2, int decim2, int max_samples_per_call);
size_t wh_ddc_out_len(const wh_ddc *d, size_t n);
int wh_ddc_run(wh_ddc *d, const float *d_iq, size_t n, double offset_hz, float *d_out, void *stream);
int wh_ddc_reset(wh_ddc *d);
void wh_ddc_destroy(wh_ddc *d);

/* Bank form (N3, trunking/system.py:453-656 VoiceRecorder.process_iq + the control monitor): n_channels <= 64
 * independent front-ends on ONE wideband buffer, each with its own offset, phase index and filter state.
 * run(): h_offsets_hz float64[n_channels]; h_active (optional) uint8[n_channels], 0 = recorder idle: skipped,
 * state kept; d_out complex64 [n_channels][out_stride], out_stride >= wh_ddc_bank_out_len(n).
 * reset(channel): one channel (a recorder reassigned to a new call), or all with channel = -1.          */
typedef struct wh_ddc_bank wh_ddc_bank;
int wh_ddc_bank_create(wh_ddc_bank **out, int n_channels, int sample_rate, const double *h_taps1, int ntaps1,
                       int decim1, const double *h_taps2, int ntaps2, int decim2, int max_samples_per_call);
size_t wh_ddc_bank_out_len(const wh_ddc_bank *d, size_t n);
int wh_ddc_bank_run(wh_ddc_bank *d, const float *d_iq, size_t n, const double *h_offsets_hz,
                    const unsigned char *h_active, float *d_out, size_t out_stride, void *stream);
int wh_ddc_bank_reset(wh_ddc_bank *d, int channel);
void wh_ddc_bank_destroy(wh_ddc_bank *d);

/* ---- A13: control-channel scanner measurement, trunking/cc_scanner.py:165-264 --------------
 * For each candidate offset (already round()ed, 0 = no mix): capture.freq_shift ->
 * lfilter(h_taps float64, zero state) -> [::decim] -> h_out[i] = {mean |y|^2, max |y|^2}
 * (float64).  Synchronous: the scanner consumes the numbers on the host.
 * h_sync_corr (optional, [n_offsets]): best normalised soft correlation of the decimated stream's
 * FM-demodulated symbol samples with the P25 frame sync, cc_scanner.py:266-353 (_detect_sync_pattern
 * returns |corr| > 0.6; the SNR gate and threshold are applied by the host shim).              */
int wh_scan_measure(const float *d_iq, size_t n, int sample_rate, const int *h_offsets_hz, int n_offsets,
                    const double *h_taps, int ntaps, int decim, double *h_out, double *h_sync_corr,
                    void *stream);

/* ---- A9-A11: P25 C4FM demodulator bank, dsp/p25/c4fm.py:2379-2807 ------------------
 * One independent C4FMDemodulator per channel (ctor c4fm.py:2412-2503).  Filters are
 * designed by the host exactly as the reference does (remez / RRC) and passed in.
 * run(): every channel consumes n samples (complex64, d_iq[ch][n], row stride
 * `iq_stride` complex) and appends its dibits / soft symbols to d_dibits[ch][cap] /
 * d_soft[ch][cap]; d_counts[ch] = symbols produced by this call.  State (filter zi,
 * 65536-sample phase buffer, sample_point, equaliser, sync rings) is carried on the
 * device between calls; reset() == C4FMDemodulator.reset() (c4fm.py:2505-2521).
 * Call size: ONE run() == ONE demodulate(iq) of the reference, whatever n is (the reference's block processing is
 * not invariant to where a stream is cut: the whole call's symbols are extracted with the call-start equaliser, the
 * sync search runs on the 65 536-sample phase buffer as it stands after the call, symbols that were shifted out of
 * it carry index -1, c4fm.py:704-728, 2621-2770 -- production calls are 72 000-75 000 samples,
 * trunking/system.py:1548-1549).  n <= the bank's max_samples_per_call <= 2^24; reserve() grows that bound (it
 * allocates and synchronises -- not for the hot path), out_cap >= n / 4 + 2.                                   */
typedef struct wh_c4fm_bank wh_c4fm_bank;
int wh_c4fm_bank_create(wh_c4fm_bank **out, int n_channels, double samples_per_symbol,
                        const float *h_lpf, int n_lpf, const float *h_rrc, int n_rrc,
                        const float *h_interp_taps /* float32[129][8] */, int max_samples_per_call);
int wh_c4fm_bank_run(wh_c4fm_bank *b, const float *d_iq, size_t n, size_t iq_stride,
                     uint8_t *d_dibits, float *d_soft, size_t out_cap, int32_t *d_counts, void *stream);
int wh_c4fm_bank_reserve(wh_c4fm_bank *b, size_t max_samples_per_call, void *stream);
int wh_c4fm_bank_reset(wh_c4fm_bank *b, void *stream);
void wh_c4fm_bank_destroy(wh_c4fm_bank *b);

/* ---- N2: P25P1SoftSyncDetector.process_batch, decoders/p25_framer.py:124-231 -----------------
 * d_soft float32 [C][stride] (n symbols per channel) -> d_scores float32 [C][stride]: correlation of the
 * 24 most recent symbols with the frame sync 0x5575F5FF77FF (+-3), one score per symbol; d_hist_in /
 * d_hist_out float32 [C][24] carry the last 24 symbols (oldest first) between calls (distinct buffers). */
int wh_sync_correlate(const float *d_soft, size_t n, size_t stride, int n_channels, const float *d_hist_in,
                      float *d_hist_out, float *d_scores, void *stream);

/* ---- N2: NID front half, decoders/p25_framer.py:475-617 + dsp/fec/bch.py ----------------------------
 * wh_sync_positions: indices i with d_scores[i] > threshold (p25_framer.py:493), UNORDERED, *d_count may exceed cap.
 * wh_nid_extract: for each start, the 33 dibits d_dibits[start .. start+32] minus the status dibit at index 11 ->
 *   64 bits MSB first -> the 63-bit BCH word in bits 62..0 of d_words (p25_framer.py:587-601); ~0 if out of range.
 * wh_bch_*: BCH(63,16,23) bounded-distance decode (bch.py:533-638) of n words: h_codewords uint64[65536] = the
 *   systematic codewords (data in bits 62..47); d_tracked_nac (optional, per word, 0 = none) enables the second
 *   pass that overwrites the NAC field (bch.py:556-571); d_data[k] = 16-bit NAC|DUID, d_errors[k] = corrected bit
 *   count or -1.                                                                                           */
int wh_sync_positions(const float *d_scores, size_t n, float threshold, int32_t *d_positions, size_t cap,
                      int32_t *d_count, void *stream);
int wh_nid_extract(const uint8_t *d_dibits, size_t n, const int32_t *d_starts, size_t n_starts, uint64_t *d_words,
                   void *stream);
/* Status-symbol stripping, decoders/p25.py:2796-2862 P25Decoder._strip_status_symbols: a counter starting at
 * initial_counter (21 for a TSDU at frame position 57) is incremented per dibit; the dibit on which it reaches 36 is a
 * status symbol and is dropped (counter back to 0).  d_dibits uint8 [n_rows][in_stride] (n dibits per row) -> d_out uint8
 * [n_rows][out_stride]; *n_out (host) = dibits kept per row (the same for every row; also returned when n_rows == 0). */
int wh_strip_status(const uint8_t *d_dibits, size_t n, size_t in_stride, int n_rows, int initial_counter,
                    uint8_t *d_out, size_t out_stride, size_t *n_out, void *stream);
typedef struct wh_bch wh_bch;
int wh_bch_create(wh_bch **out, const uint64_t *h_codewords);
int wh_bch_decode(wh_bch *b, const uint64_t *d_words, size_t n, const int32_t *d_tracked_nac, int32_t *d_data,
                  int32_t *d_errors, void *stream);
void wh_bch_destroy(wh_bch *b);

/* ---- A12: P25 Phase-2 CQPSK bank, dsp/p25/cqpsk.py:199-350 + dsp/p25/symbol_timing.py -------
 * RRC matched filter (h_rrc float32 = design_rrc_filter_phase2, h_zi = lfilter_zi(rrc, 1.0)),
 * Costas loop (c_kp, c_ki, c_maxf; cqpsk.py:94-119), Mueller-Muller timing (t_kp, t_ki;
 * symbol_timing.py:238-270), pi/4-DQPSK differential decode.  All float64 like the reference.
 * run(): d_iq complex64 [C][iq_stride] -> d_dibits uint8 [C][cap]; d_symbols (optional, may be
 * NULL) complex128 [C][cap]; d_counts int32 [C].  State carried on the device.
 * ONE run() == ONE demodulate(iq) of the reference for any n <= max_samples_per_call (the reference re-seeds the
 * matched filter with zi = state * iq[0] on every call, cqpsk.py:283-285, so a stream cut differently gives different
 * dibits); reserve() grows max_samples_per_call (allocates and synchronises).  samples_per_symbol >= 2;
 * cap >= n / (samples_per_symbol / 2) + 2 (the timing loop's period never drops below sps / 2).              */
typedef struct wh_cqpsk_bank wh_cqpsk_bank;
int wh_cqpsk_bank_create(wh_cqpsk_bank **out, int n_channels, double samples_per_symbol, const float *h_rrc,
                         int ntaps, const double *h_zi, double c_kp, double c_ki, double c_maxf, double t_kp,
                         double t_ki, int max_samples_per_call);
int wh_cqpsk_bank_run(wh_cqpsk_bank *b, const float *d_iq, size_t n, size_t iq_stride, uint8_t *d_dibits,
                      double *d_symbols, size_t cap, int32_t *d_counts, void *stream);
int wh_cqpsk_bank_reserve(wh_cqpsk_bank *b, size_t max_samples_per_call, void *stream);
int wh_cqpsk_bank_reset(wh_cqpsk_bank *b, void *stream);
void wh_cqpsk_bank_destroy(wh_cqpsk_bank *b);

/* ---- N4: ChannelClassifier.update, channel_classifier.py:100-125 ---------------------------------
 * d_power_db float32 [n_frames][n_bins] spectrum frames (wh_spectrum_run output) are folded, in frame order, into
 * d_stats float64 [n_bins][5] = {sum, sum_sq, count, min, max}; initialise a row to {0, 0, 0, +inf, -inf}.  */
int wh_binstats_update(const float *d_power_db, size_t n_frames, int n_bins, double *d_stats, void *stream);

/* ---- A12 (LSM): P25 Phase-1 CQPSK / linear simulcast demodulator, decoders/p25.py:190-669 -------------
 * Per call and channel: block AGC (:436-455), NCO from the tracked frequency offset (:460-465), 63-tap
 * 'same'-mode low-pass (:468-471; per call, zero-padded edges, float64 once the NCO runs), then the
 * symbol-clock loop with 8-tap MMSE interpolation, pi/4-DQPSK slicer, frequency loop and Gardner timing
 * error (:484-669), in the reference's NumPy-2 scalar types.  h_lpf float32[63] (:370-383), h_mmse
 * float32[129*8] (:289-323).  run(): d_iq complex64 [C][iq_stride], n <= max_samples_per_call samples per
 * channel = ONE reference demodulate() call -> d_dibits uint8 [C][cap] (cap >= n), d_phases (optional, may
 * be NULL) float32 [C][cap] differential phase per symbol, d_counts int32 [C].  get_state(): h_out[7] =
 * agc_gain, freq_offset, phase_acc, symbol_clock, prev_symbol re / im, clock_is_float32.            */
typedef struct wh_lsm_bank wh_lsm_bank;
int wh_lsm_bank_create(wh_lsm_bank **out, int n_channels, double samples_per_symbol, const float *h_lpf,
                       const float *h_mmse, int max_samples_per_call);
int wh_lsm_bank_run(wh_lsm_bank *b, const float *d_iq, size_t n, size_t iq_stride, uint8_t *d_dibits, float *d_phases,
                    size_t cap, int32_t *d_counts, void *stream);
/* grow the per-call work buffer so that calls of up to n_max samples are accepted (decoders/p25.py:413 takes any length);
 * carried state is untouched; synchronises the stream; no-op when n_max is not larger than what the bank already takes */
int wh_lsm_bank_reserve(wh_lsm_bank *b, int n_max, void *stream);
int wh_lsm_bank_reset(wh_lsm_bank *b, void *stream);
int wh_lsm_bank_get_state(wh_lsm_bank *b, int channel, double *h_out, void *stream);
void wh_lsm_bank_destroy(wh_lsm_bank *b);

/* Gardner timing error detector bank, dsp/p25/symbol_timing.py:60-211 (GardnerTED.process_block):
 * d_x float32 [C][stride] -> d_symbols / d_errors float64 [C][cap], d_counts int32 [C].       */
/* CostasLoop, dsp/p25/cqpsk.py:84-196, as a bank (standalone drop-in; the same loop runs fused inside wh_cqpsk_bank_run):
 * per sample corrected = x exp(-j phase); decision-directed pi/4 phase detector; PI loop filter with the frequency clipped
 * to +-max_freq; phase wrapped.  d_x, d_out: complex128 [n_channels][stride] (n used); h_freq (optional, synchronous):
 * the loop's frequency estimates after the call (the frequency_offset property).  kp / ki from loop_bw and damping as
 * cqpsk.py:107-110 computes them. */
typedef struct wh_costas_bank wh_costas_bank;
int wh_costas_bank_create(wh_costas_bank **out, int n_channels, double kp, double ki, double max_freq);
int wh_costas_bank_run(wh_costas_bank *b, const double *d_x, size_t n, size_t stride, double *d_out, double *h_freq,
                       void *stream);
int wh_costas_bank_reset(wh_costas_bank *b, void *stream);
void wh_costas_bank_destroy(wh_costas_bank *b);

/* MuellerMullerTED, dsp/p25/symbol_timing.py:214-380, as a bank: complex128 samples in; per symbol the cubic-interpolated
 * sample, the nearest point of (+-1 +-j)/sqrt 2 and the timing error Re{conj(d[n-1]) x[n] - conj(d[n]) x[n-1]};
 * d_symbols / d_decisions complex128 [n_channels][cap], d_errors float64 [n_channels][cap], d_counts int32. */
typedef struct wh_mm_bank wh_mm_bank;
int wh_mm_bank_create(wh_mm_bank **out, int n_channels, double samples_per_symbol, double kp, double ki);
int wh_mm_bank_run(wh_mm_bank *b, const double *d_x, size_t n, size_t stride, double *d_symbols, double *d_decisions,
                   double *d_errors, size_t cap, int32_t *d_counts, void *stream);
int wh_mm_bank_reset(wh_mm_bank *b, void *stream);
void wh_mm_bank_destroy(wh_mm_bank *b);

typedef struct wh_gardner_bank wh_gardner_bank;
int wh_gardner_bank_create(wh_gardner_bank **out, int n_channels, double samples_per_symbol, double kp, double ki);
int wh_gardner_bank_run(wh_gardner_bank *g, const float *d_x, size_t n, size_t stride, double *d_symbols,
                        double *d_errors, size_t cap, int32_t *d_counts, void *stream);
int wh_gardner_bank_reset(wh_gardner_bank *g, void *stream);
void wh_gardner_bank_destroy(wh_gardner_bank *g);

#ifdef __cplusplus
}
#endif
#endif /* WAVEHIP_H */
