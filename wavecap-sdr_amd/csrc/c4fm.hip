// P25 Phase-1 C4FM demodulator bank for gfx950 (reference dsp/p25/c4fm.py:2379-2807).
//
// One independent demodulator per channel, state carried on the device between calls.
// Per call (n samples per channel) the work is split by its parallel shape:
//   k_lpf / k_rrc   baseband LPF and RRC FIRs -- float64 like the reference's lfilter call
//                   (len(a)==1 -> np.convolve in float64, c4fm.py:2575-2589); one thread per
//                   (channel, sample), taps + tile staged in LDS, history prefix in the workspace;
//   k_fm            symbol-spaced differential demodulator (c4fm.py:365-395): 2 x 8-tap float32
//                   interpolation + portable atan2f, one thread per sample;
//   k_seq           ONE WAVEFRONT PER CHANNEL for the feedback part (c4fm.py:649-783, 2621-2770):
//                   fixed-rate symbol clock scan, parallel symbol extraction, then the sync loop in
//                   blocks of 64 symbols -- every lane evaluates one symbol's 24-tap correlations,
//                   a ballot finds the first trigger, and the rare sync event (hill-climb timing
//                   optimiser, equaliser update, message re-slice) runs wave-cooperatively with
//                   shuffles: lanes 0..23 interpolate the 24 sync symbols, the float32 sums are
//                   then taken in the reference's order.
// Every parity-critical float op uses the *_rn intrinsics (no FMA contraction) in the order of
// the scalar restatement oracle/c4fm_ref.c, so dibits are bit-exact against it.
#include "wh_common.h"
#include "wh_portable_math.h"
#include <cmath>
#include <memory>
#include <vector>

using namespace wh;

namespace {

constexpr int BUF_LEN = 65536;
constexpr int HALF_BUF = 32768;
constexpr int MSG_DIBITS = 340;  // c4fm.py:792

struct ChanScalars {
    double sample_point, pll, gain;
    int sp_np64, eq_init, fine_sync, symbols_since_sync, sync_count, buffer_pointer;
};
struct ChanState {
    ChanScalars s;
    float det_hist[24];  // last 24 soft symbols pushed to the primary detector (oldest first)
    float lag_hist[24];  // same for the lagging detector
};

struct C4Args {
    const float2 *iq;    // [C][iq_stride]
    size_t iq_stride;
    int n, n_max, C;
    double sps, lagging_offset, max_fine_adj;
    int nl, nr, overlap, interp_offset, fm_row, ns_max, ns_call;
    const float *lpf, *rrc, *taps;  // taps [129][8]
    float2 *xh;          // [C][nl-1]      input history
    double2 *y;          // [C][nr-1 + n_max]  LPF output with history prefix
    float2 *z;           // [C][overlap + n_max] RRC output (float32) with history prefix
    float *phases;       // [C][n_max]
    float *buffer;       // [C][65536]
    ChanState *st;       // [C]
    int *sym_x;          // [C][ns_max] sample index of each symbol
    double *sym_sp;      // [C][ns_max] sample_point at the symbol
    int *sym_idx;        // [C][ns_max] buffer index in the FINAL buffer of the call (-1: shifted out)
    float *ss;           // [C][2 (24 + ns_max)] detector streams of calls too long for LDS (GSTREAM form)
    uint8_t *dibits;     // [C][out_cap]
    float *soft;         // [C][out_cap]
    size_t out_cap;
    int *counts;         // [C]
};

// ---- FIR stages (float64) -------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lpf(C4Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    float *tp = reinterpret_cast<float *>(sm_raw);                       // nl taps
    float2 *tile = reinterpret_cast<float2 *>(sm_raw + ((a.nl * 4 + 15) & ~15));  // 256 + nl-1 inputs
    const int c = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
    const int H = a.nl - 1;
    for (int i = tid; i < a.nl; i += 256) tp[i] = a.lpf[i];
    const float2 *x = a.iq + (size_t)c * a.iq_stride;
    const float2 *xh = a.xh + (size_t)c * H;
    for (int i = tid; i < 256 + H; i += 256) {
        int t = t0 - H + i;  // stream offset relative to this call
        float2 v = make_float2(0.f, 0.f);
        if (t < 0) { if (t + H >= 0) v = xh[t + H]; }
        else if (t < a.n) v = x[t];
        tile[i] = v;
    }
    __syncthreads();
    const int t = t0 + tid;
    if (t < a.n) {
        double ai = 0.0, aq = 0.0;
        for (int k = 0; k < a.nl; ++k) {
            float2 v = tile[tid + H - k];
            double w = (double)tp[k];
            ai = fma(w, (double)v.x, ai);
            aq = fma(w, (double)v.y, aq);
        }
        a.y[(size_t)c * (a.nr - 1 + a.n_max) + (a.nr - 1) + t] = make_double2(ai, aq);
    }
}

__global__ __launch_bounds__(256) void k_rrc(C4Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    float *tp = reinterpret_cast<float *>(sm_raw);
    double2 *tile = reinterpret_cast<double2 *>(sm_raw + ((a.nr * 4 + 15) & ~15));
    const int c = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
    const int H = a.nr - 1;
    for (int i = tid; i < a.nr; i += 256) tp[i] = a.rrc[i];
    const double2 *y = a.y + (size_t)c * (H + a.n_max);  // y[H + t] is sample t; history at [0, H)
    for (int i = tid; i < 256 + H; i += 256) {
        int t = t0 + i;  // index into y (history-prefixed)
        tile[i] = (t < H + a.n) ? y[t] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    const int t = t0 + tid;
    if (t < a.n) {
        double ai = 0.0, aq = 0.0;
        for (int k = 0; k < a.nr; ++k) {
            double2 v = tile[tid + H - k];
            double w = (double)tp[k];
            ai = fma(w, v.x, ai);
            aq = fma(w, v.y, aq);
        }
        a.z[(size_t)c * (a.overlap + a.n_max) + a.overlap + t] = make_float2((float)ai, (float)aq);
    }
}

__device__ __forceinline__ float interp8(const float *p, const float *tp) {
    float r = __fmul_rn(p[0], tp[0]);
#pragma unroll
    for (int i = 1; i < 8; ++i) r = __fadd_rn(r, __fmul_rn(p[i], tp[i]));
    return r;
}

// ---- differential demodulator (c4fm.py:365-395) ------------------------------------------------
__global__ __launch_bounds__(256) void k_fm(C4Args a) {
    const int c = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= a.n) return;
    const float2 *z = a.z + (size_t)c * (a.overlap + a.n_max);
    const float *tp = a.taps + a.fm_row * 8;
    float2 prev = z[x];
    float i_prev = prev.x, q_prev_conj = -prev.y;
    const int off = a.interp_offset + x;   // off + 8 <= n + overlap always holds for floor(sps) >= 4
    float wi[8], wq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float2 v = z[off + i];
        wi[i] = v.x;
        wq[i] = v.y;
    }
    float i_cur = interp8(wi, tp), q_cur = interp8(wq, tp);
    float diff_i = __fsub_rn(__fmul_rn(i_prev, i_cur), __fmul_rn(q_prev_conj, q_cur));
    float diff_q = __fadd_rn(__fmul_rn(i_prev, q_cur), __fmul_rn(i_cur, q_prev_conj));
    a.phases[(size_t)c * a.n_max + x] = whm_atan2f(diff_q, diff_i);
}

// ---- carry histories to the front of the workspace (after everything else used them) --------
__global__ __launch_bounds__(256) void k_carry(C4Args a) {
    const int c = blockIdx.x, tid = threadIdx.x;
    const int n = a.n;
    {   // input history: last nl-1 samples of [xh | x]
        const int H = a.nl - 1;
        float2 *xh = a.xh + (size_t)c * H;
        const float2 *x = a.iq + (size_t)c * a.iq_stride;
        for (int b = 0; b < H; b += 256) {
            int i = b + tid;
            float2 v = make_float2(0.f, 0.f);
            if (i < H) { int t = n - H + i; v = t >= 0 ? x[t] : xh[t + H]; }
            __syncthreads();
            if (i < H) xh[i] = v;
            __syncthreads();
        }
    }
    {   // LPF output history
        const int H = a.nr - 1;
        double2 *y = a.y + (size_t)c * (H + a.n_max);
        for (int b = 0; b < H; b += 256) {
            int i = b + tid;
            double2 v = make_double2(0.0, 0.0);
            if (i < H) v = y[n + i];
            __syncthreads();
            if (i < H) y[i] = v;
            __syncthreads();
        }
    }
    {   // RRC output history
        const int H = a.overlap;
        float2 *z = a.z + (size_t)c * (H + a.n_max);
        for (int b = 0; b < H; b += 256) {
            int i = b + tid;
            float2 v = make_float2(0.f, 0.f);
            if (i < H) v = z[n + i];
            __syncthreads();
            if (i < H) z[i] = v;
            __syncthreads();
        }
    }
}

// ---- sequential part: one wave per channel ---------------------------------------------------
__device__ __forceinline__ int slice32(float v) {
    const float B = (float)1.5707963267948966;
    return v >= B ? 1 : (v >= 0.0f ? 0 : (v >= -B ? 2 : 3));
}
__device__ __forceinline__ int slice64(double v) {
    const double B = 1.5707963267948966;
    return v >= B ? 1 : (v >= 0.0 ? 0 : (v >= -B ? 2 : 3));
}
__device__ __forceinline__ float sync_sym(int i) {
    return ((0x5575F5FF77FFULL >> ((23 - i) * 2)) & 3ULL) == 1ULL ? 3.0f : -3.0f;
}

struct ScoreCtx {
    const float *buf;
    const float *taps;
    double sps;
    float pll32, gain32;
    int lane;
};

// lanes 0..23 evaluate symbol i = lane of _timing_score_jit / _timing_correction_jit:
// returns soft (equalised interpolated sample) and validity.
__device__ __forceinline__ void sync_symbol_soft(const ScoreCtx &c, double offset, float &soft, int &valid) {
    double ptr = __dsub_rn(offset, __dmul_rn(23.0, c.sps));
    for (int t = 0; t < 23; ++t)
        if (t < c.lane) ptr = __dadd_rn(ptr, c.sps);   // ptr += sps, lane times (sequential rounding)
    int buf_idx = (int)ptr;
    int io = buf_idx - 3;
    valid = (c.lane < 24) && io >= 0 && io <= BUF_LEN - 8;
    soft = 0.f;
    if (valid) {
        double mu = __dsub_rn(ptr, (double)buf_idx);
        double mu_inv = __dsub_rn(1.0, mu);
        int row = (int)__dadd_rn(__dmul_rn(mu_inv, 128.0), 0.5);
        row = row < 0 ? 0 : (row > 128 ? 128 : row);
        float v = interp8(c.buf + io, c.taps + row * 8);
        soft = __fmul_rn(__fadd_rn(v, c.pll32), c.gain32);
    }
}

__device__ float timing_score(const ScoreCtx &c, double offset) {
    float soft;
    int valid;
    sync_symbol_soft(c, offset, soft, valid);
    float term = __fmul_rn(soft, sync_sym(c.lane < 24 ? c.lane : 0));
    float score = 0.f;
    bool first = true;
    for (int i = 0; i < 24; ++i) {
        float t = __shfl(term, i);
        int v = __shfl(valid, i);
        if (v) {
            score = first ? t : __fadd_rn(score, t);
            first = false;
        }
    }
    return score;
}

// GSTREAM: the two detector streams live in the workspace instead of LDS (calls of more than ~8 000 symbols)
template <bool GSTREAM>
__global__ __launch_bounds__(64) void k_seq(C4Args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int c = blockIdx.x, lane = threadIdx.x;
    float *SS, *LL;                         // [24 + ns] primary / lagging detector streams (history first)
    if constexpr (GSTREAM) {
        SS = a.ss + (size_t)c * 2 * (24 + a.ns_max);
        LL = SS + 24 + a.ns_max;
    } else {
        SS = sm;
        LL = sm + 24 + a.ns_call;
    }
    ChanScalars st = a.st[c].s;
    float *buf = a.buffer + (size_t)c * BUF_LEN;
    const float *ph = a.phases + (size_t)c * a.n_max;
    int *sym_x = a.sym_x + (size_t)c * a.ns_max;
    double *sym_sp = a.sym_sp + (size_t)c * a.ns_max;
    int *sym_idx = a.sym_idx + (size_t)c * a.ns_max;
    uint8_t *dib = a.dibits + (size_t)c * a.out_cap;
    float *soft = a.soft + (size_t)c * a.out_cap;
    const int n = a.n;
    const double sps = a.sps;

    // ---- symbol clock scan (sequential, uniform across lanes; lane 0 records) --------------
    // reference loop: per sample sample_point -= 1.0 (exact); symbol when < 1.0; then += sps.
    int count = 0;
    {
        double sp = st.sample_point;
        long long x = -1;
        const int cap = (int)(a.out_cap < (size_t)a.ns_call ? a.out_cap : (size_t)a.ns_call);
        while (true) {
            double d = floor(sp - 1.0) + 1.0;      // decrements until sp - d < 1.0 (at least one)
            if (!(d >= 1.0)) d = 1.0;
            x += (long long)d;
            if (x >= n) {
                // clock state at the end of the call: (n-1 - x_prev) decrements applied
                sp = sp - (double)((long long)n - 1 - (x - (long long)d));
                break;
            }
            sp = sp - d;
            if (count < cap) {
                if (lane == 0) {
                    sym_x[count] = (int)x;
                    sym_sp[count] = sp;
                }
                count++;
            }
            sp = __dadd_rn(sp, sps);
        }
        st.sample_point = sp;
    }
    // ---- phase buffer management (c4fm.py:705-728) in closed form --------------------------------
    // The reference appends sample by sample and, whenever the write pointer reaches the end, moves the upper half
    // down and zeroes it; the sync search below runs on the buffer as it stands AFTER the whole call.  With ptr0 the
    // pointer at call start the first shift happens at sample x_first = 65534 - ptr0, the next ones every 32768
    // samples: S shifts in all, sample x ends at position base + x with base = ptr0 + 1 - 32768 S (negative:
    // shifted out), everything beyond the final pointer is zero, and what earlier calls left moves down by 32768 S.
    const int ptr0 = st.buffer_pointer;
    const long long x_first = (long long)(BUF_LEN - 2) - ptr0;
    const int S = x_first < n ? 1 + (int)(((long long)n - 1 - x_first) / HALF_BUF) : 0;
    const long long base = (long long)ptr0 + 1 - (long long)S * HALF_BUF;
    const float prev0 = buf[ptr0];   // the sample before this call's first: x1 of a symbol at x = 0
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (S == 1) {
        for (int i = lane; i < HALF_BUF; i += 64) {
            buf[i] = buf[i + HALF_BUF];
            buf[i + HALF_BUF] = 0.f;
        }
    } else if (S >= 2) {             // nothing of the earlier calls is left (base <= 0)
        for (int i = lane; i < BUF_LEN; i += 64) buf[i] = 0.f;
    }
    __threadfence_block();
    for (long long x = (base < 0 ? -base : 0) + lane; x < n; x += 64) buf[base + x] = ph[x];
    st.buffer_pointer = (int)(base + n - 1);
    __threadfence_block();
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    // ---- symbol extraction, parallel over symbols (c4fm.py:731-771) ---------------------------
    // x1 / x2 are the previous and the current phase sample at the time of the symbol, i.e. ph[x - 1], ph[x];
    // the recorded index is the symbol's place in the final buffer, -1 once a shift moved it out (c4fm.py:722-728)
    {
        const float pll32 = (float)st.pll, gain32 = (float)st.gain;
        for (int k = lane; k < count; k += 64) {
            const int x = sym_x[k];
            const long long fidx = base + x;
            double spk = sym_sp[k];
            double mu = __dsub_rn(1.0, spk);
            float x1 = x > 0 ? ph[x - 1] : prev0, x2 = ph[x];
            int d;
            float sn;
            if (mu < 0.0 || mu > 1.0 || !st.sp_np64) {
                float v = mu < 0.0 ? x1 : (mu > 1.0 ? x2 : __fadd_rn(x1, __fmul_rn(__fsub_rn(x2, x1), (float)mu)));
                float sr = __fmul_rn(__fadd_rn(v, pll32), gain32);
                d = slice32(sr);
                sn = __fmul_rn(sr, (float)1.2732395447351628);
            } else {
                double v = __dadd_rn((double)x1, __dmul_rn((double)__fsub_rn(x2, x1), mu));
                double sr = __dmul_rn(__dadd_rn(v, st.pll), st.gain);
                d = slice64(sr);
                sn = (float)__dmul_rn(sr, 1.2732395447351628);
            }
            dib[k] = (uint8_t)d;
            soft[k] = sn;
            sym_idx[k] = fidx < 0 ? -1 : (int)fidx;
            SS[24 + k] = sn;
        }
        if (lane < 24) {
            SS[lane] = a.st[c].det_hist[lane];
            LL[lane] = a.st[c].lag_hist[lane];
        }
    }
    if constexpr (GSTREAM) __threadfence_block();
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    // ---- sync loop in blocks of 64 symbols (c4fm.py:2621-2770) -------------------------------
    int n_lag = 0;  // lag pushes committed so far (LL[24 + n_lag) is the stream)
    const int ilo = (int)a.lagging_offset;
    const float lag_mu32 = (float)(1.0 - (a.lagging_offset - (double)ilo));
    const double lag_mu = 1.0 - (a.lagging_offset - (double)ilo);
    int kb = 0;
    while (kb < count) {
        const int k = kb + lane;
        const bool in = k < count;
        // primary score for this lane's symbol
        float P = 0.f;
        if (in) {
            const float *w = SS + k + 1;  // SS index of symbol k-23
            P = __fmul_rn(sync_sym(0), w[0]);
#pragma unroll
            for (int i = 1; i < 24; ++i) P = __fadd_rn(P, __fmul_rn(sync_sym(i), w[i]));
        }
        const bool coarse = !st.fine_sync;
        float L = 0.f;
        bool pushes = false;
        if (coarse) {
            float sln = 0.f;
            if (in) {
                int idx = sym_idx[k];
                int lag_pos = idx - ilo;
                if (idx >= 0 && lag_pos >= 4 && lag_pos < BUF_LEN) {
                    int lo = lag_pos - 4;
                    float v;
                    if (lo + 1 < BUF_LEN) {
                        float x1 = buf[lo], x2 = buf[lo + 1];
                        v = lag_mu < 0.0 ? x1 : (lag_mu > 1.0 ? x2 : __fadd_rn(x1, __fmul_rn(__fsub_rn(x2, x1), lag_mu32)));
                    } else {
                        v = buf[lo];
                    }
                    float sl = __fmul_rn(__fadd_rn(v, (float)st.pll), (float)st.gain);
                    sln = __fmul_rn(sl, (float)(4.0 / 3.141592653589793));
                    pushes = true;
                }
            }
            unsigned long long pm = __ballot(pushes);
            int my_pos = n_lag + __popcll(pm & ((1ULL << lane) - 1ULL));  // stream position of my push
            if (pushes) LL[24 + my_pos] = sln;
            if constexpr (GSTREAM) __threadfence_block();
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            if (pushes) {
                const float *w = LL + my_pos + 1;
                L = __fmul_rn(sync_sym(0), w[0]);
#pragma unroll
                for (int i = 1; i < 24; ++i) L = __fadd_rn(L, __fmul_rn(sync_sym(i), w[i]));
            }
        }
        const bool use_lag = coarse && L > P && L >= 100.0f;
        const float score = use_lag ? L : P;
        // a detection on a symbol whose samples have left the buffer is skipped with `continue` (c4fm.py:2663-2665):
        // no optimiser run and no fine-sync-loss check for that symbol either
        const bool hit = in && score >= 100.0f;
        const bool gone = in && sym_idx[k] < 0;
        const bool trig = hit && !gone;
        const bool lose = in && !(hit && gone) && (st.symbols_since_sync + lane + 1 > 3600);
        unsigned long long em = __ballot(trig || lose);
        if (em == 0ULL) {  // no event in this block: commit and advance
            int nin = count - kb < 64 ? count - kb : 64;
            st.symbols_since_sync += nin;
            if (coarse) n_lag += __popcll(__ballot(pushes));
            kb += 64;
            continue;
        }
        const int j = __ffsll((long long)em) - 1;  // first event lane
        const int ke = kb + j;
        st.symbols_since_sync += j + 1;
        if (coarse) n_lag += __popcll(__ballot(pushes) & ((2ULL << j) - 1ULL));
        const bool e_trig = __shfl((int)trig, j) != 0;
        if (e_trig) {
            const bool e_lag = __shfl((int)use_lag, j) != 0;
            const double additional = e_lag ? -a.lagging_offset : 0.0;
            const int e_idx = sym_idx[ke];
            // ---- _timing_optimize_jit (c4fm.py:543-644) ---------------------------------
            ScoreCtx sc{buf, a.taps, sps, (float)st.pll, (float)st.gain, lane};
            const double offset = __dadd_rn(__dadd_rn((double)e_idx, 0.5), additional);
            double step, step_min = sps / 200.0, max_adj;
            if (st.fine_sync) { step = sps / 16.0; max_adj = sps; }
            else { step = sps / 8.0; max_adj = sps / 2.0; }
            double adj = 0.0;
            float s_c = timing_score(sc, offset);
            float s_l = timing_score(sc, __dsub_rn(offset, step));
            float s_r = timing_score(sc, __dadd_rn(offset, step));
            while (step > step_min && fabs(adj) <= max_adj) {
                if (s_l > s_r && s_l > s_c) {
                    adj = __dsub_rn(adj, step);
                    s_r = s_c;
                    s_c = s_l;
                    s_l = timing_score(sc, __dsub_rn(__dadd_rn(offset, adj), step));
                } else if (s_r > s_l && s_r > s_c) {
                    adj = __dadd_rn(adj, step);
                    s_l = s_c;
                    s_c = s_r;
                    s_r = timing_score(sc, __dadd_rn(__dadd_rn(offset, adj), step));
                } else {
                    step = __dmul_rn(step, 0.5);
                    if (step > step_min) {
                        s_l = timing_score(sc, __dsub_rn(__dadd_rn(offset, adj), step));
                        s_r = timing_score(sc, __dadd_rn(__dadd_rn(offset, adj), step));
                    }
                }
            }
            // ---- _timing_correction_jit (c4fm.py:466-540) -------------------------------
            double pll_corr;
            int pll_is_f64 = 0;
            float gain_corr;
            {
                float sf;
                int valid;
                sync_symbol_soft(sc, __dadd_rn(offset, adj), sf, valid);
                float bp = 0.f, bm = 0.f, ga = 0.f;
                int pc = 0, mc = 0, any = 0;
                for (int i = 0; i < 24; ++i) {
                    float s_i = __shfl(sf, i);
                    int v = __shfl(valid, i);
                    if (v) {
                        float ideal = sync_sym(i);
                        if (ideal > 0.f) { bp = __fadd_rn(bp, __fsub_rn(s_i, ideal)); pc++; }
                        else { bm = __fadd_rn(bm, __fsub_rn(s_i, ideal)); mc++; }
                        ga = __fadd_rn(ga, __fsub_rn(fabsf(ideal), fabsf(s_i)));
                        any = 1;
                    }
                }
                if (pc > 0) bp = __fdiv_rn(bp, (float)(-pc));
                if (mc > 0) bm = __fdiv_rn(bm, (float)(-mc));
                float pc32 = __fdiv_rn(__fadd_rn(bp, bm), 2.0f);
                const float HP32 = (float)1.5707963267948966;
                pll_corr = (double)pc32;
                if (pc32 < -HP32) { pll_corr = -1.5707963267948966; pll_is_f64 = 1; }
                else if (pc32 > HP32) { pll_corr = 1.5707963267948966; pll_is_f64 = 1; }
                gain_corr = any ? __fdiv_rn(ga, (float)(24.0 * 2.356194490192345)) : 0.f;
            }
            if (s_c >= 100.0f) {
                if (st.fine_sync) {
                    if (adj < -a.max_fine_adj) adj = -a.max_fine_adj;
                    if (adj > a.max_fine_adj) adj = a.max_fine_adj;
                    st.sp_np64 = 1;
                }
                st.sample_point = __dadd_rn(st.sample_point, __dadd_rn(adj, additional));
                // _Equalizer.apply_correction (c4fm.py:260-272)
                const double MAXPLL = 3.141592653589793 / 3.0;
                if (!pll_is_f64) {
                    float p32 = (float)st.pll, a32 = (float)pll_corr;
                    float r = st.eq_init ? __fadd_rn(p32, __fmul_rn(a32, (float)0.15)) : __fadd_rn(p32, a32);
                    float lo = (float)(-MAXPLL), hi = (float)MAXPLL;
                    r = r < lo ? lo : r;
                    r = r > hi ? hi : r;
                    st.pll = (double)r;
                } else {
                    double r = st.eq_init ? __dadd_rn(st.pll, __dmul_rn(pll_corr, 0.15)) : __dadd_rn(st.pll, pll_corr);
                    r = r < -MAXPLL ? -MAXPLL : r;
                    r = r > MAXPLL ? MAXPLL : r;
                    st.pll = r;
                }
                {
                    float g32 = (float)st.gain;
                    float r = st.eq_init ? __fadd_rn(g32, __fmul_rn(gain_corr, (float)0.15)) : __fadd_rn(g32, gain_corr);
                    r = r < 1.0f ? 1.0f : r;
                    r = r > 1.25f ? 1.25f : r;
                    st.gain = (double)r;
                }
                st.eq_init = 1;
                st.sync_count += 1;
                st.fine_sync = 1;
                st.symbols_since_sync = 0;
                // ---- message re-slice (c4fm.py:2703-2746, 795-869), parallel over dibits ----
                const double sync_start = __dadd_rn(__dadd_rn(__dsub_rn((double)e_idx, __dmul_rn(23.0, sps)), adj), additional);
                const double msg_start = __dadd_rn(sync_start, __dmul_rn(24.0, sps));
                int remaining = count - (ke + 1);
                int nres = remaining < MSG_DIBITS ? remaining : MSG_DIBITS;
                const float pll32 = (float)st.pll, gain32 = (float)st.gain;
                for (int i = lane; i < nres; i += 64) {
                    double pos = __dadd_rn(msg_start, __dmul_rn((double)i, sps));
                    int idx = (int)pos;
                    double mu = __dsub_rn(pos, (double)idx);
                    bool f32path = false;
                    float v32 = 0.f;
                    double v64 = 0.0;
                    if (idx >= 0 && idx + 1 < BUF_LEN) {
                        float x1 = buf[idx], x2 = buf[idx + 1];
                        if (mu < 0.0) { v32 = x1; f32path = true; }
                        else if (mu > 1.0) { v32 = x2; f32path = true; }
                        else v64 = __dadd_rn((double)x1, __dmul_rn((double)__fsub_rn(x2, x1), mu));
                    } else {
                        int cc = idx < BUF_LEN - 1 ? idx : BUF_LEN - 1;
                        cc = cc < 0 ? 0 : cc;
                        v32 = buf[cc];
                        f32path = true;
                    }
                    int d;
                    float sn;
                    if (f32path) {
                        float sr = __fmul_rn(__fadd_rn(v32, pll32), gain32);
                        sn = __fmul_rn(sr, (float)1.2732395447351628);
                        d = slice32(sr);
                    } else {
                        double sr = __dmul_rn(__dadd_rn(v64, st.pll), st.gain);
                        sn = (float)__dmul_rn(sr, 1.2732395447351628);
                        d = slice64(sr);
                    }
                    dib[ke + 1 + i] = (uint8_t)d;
                    soft[ke + 1 + i] = sn;
                    SS[24 + ke + 1 + i] = sn;
                }
                if constexpr (GSTREAM) __threadfence_block();
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (st.symbols_since_sync > 3600) {
            st.fine_sync = 0;
            st.symbols_since_sync = 0;
        }
        kb = ke + 1;
    }
    // ---- write back state --------------------------------------------------------------------
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    float dh = 0.f, lh = 0.f;
    if (lane < 24) {
        dh = SS[count + lane];   // last 24 entries of the stream [24 + count)
        lh = LL[n_lag + lane];
    }
    if (lane < 24) {
        a.st[c].det_hist[lane] = dh;
        a.st[c].lag_hist[lane] = lh;
    }
    if (lane == 0) {
        a.st[c].s = st;
        a.counts[c] = count;
    }
}

__global__ void k_reset(ChanState *st, int C, double sps) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    ChanScalars s;
    s.sample_point = sps;
    s.pll = 0.0;
    s.gain = 1.219;   // EQUALIZER_INITIAL_GAIN, c4fm.py:66
    s.sp_np64 = s.eq_init = s.fine_sync = s.symbols_since_sync = s.sync_count = s.buffer_pointer = 0;
    st[c].s = s;
    for (int i = 0; i < 24; ++i) st[c].det_hist[i] = st[c].lag_hist[i] = 0.f;
}

}  // namespace

constexpr int C4_N_CAP = 1 << 24;          // samples per call and channel the workspace may be sized for
constexpr size_t C4_LDS_STREAMS = 64 * 1024;  // detector streams beyond this many bytes go to the workspace

struct wh_c4fm_bank {
    int C, n_max = 0, nl, nr, overlap, interp_offset, fm_row, ns_max = 0;
    double sps;
    float *d_lpf = nullptr, *d_rrc = nullptr, *d_taps = nullptr;
    float2 *d_xh = nullptr;
    double2 *d_y = nullptr;
    float2 *d_z = nullptr;
    float *d_phases = nullptr, *d_buffer = nullptr, *d_ss = nullptr;
    ChanState *d_st = nullptr;
    int *d_sym_x = nullptr, *d_sym_idx = nullptr;
    double *d_sym_sp = nullptr;
};

static int c4_ns(double sps, int n) { return (int)((double)n / sps) + 16; }
static size_t c4_stream_bytes(int ns) { return (size_t)2 * (24 + ns) * sizeof(float); }

// (re)size the per-call workspaces for calls of up to n_max samples; the FIR histories at the head of every channel's
// y / z row are carried over
static int c4fm_resize(wh_c4fm_bank *b, int n_max, hipStream_t st) {
    const int C = b->C, Hy = b->nr - 1, Hz = b->overlap;
    const int ns_max = c4_ns(b->sps, n_max);
    double2 *y = nullptr;
    float2 *z = nullptr;
    float *ph = nullptr, *ss = nullptr;
    int *sx = nullptr, *si = nullptr;
    double *sp = nullptr;
    auto drop = [&]() {
        (void)hipFree(y); (void)hipFree(z); (void)hipFree(ph); (void)hipFree(ss); (void)hipFree(sx); (void)hipFree(si);
        (void)hipFree(sp);
    };
#define C4_TRY(x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) { drop(); return set_err(WH_E_HIP, "c4fm workspace: %s", hipGetErrorString(e_)); }       \
    } while (0)
    C4_TRY(hipMalloc(&y, (size_t)C * (Hy + n_max) * sizeof(double2)));
    C4_TRY(hipMalloc(&z, (size_t)C * (Hz + n_max) * sizeof(float2)));
    C4_TRY(hipMalloc(&ph, (size_t)C * n_max * sizeof(float)));
    C4_TRY(hipMalloc(&sx, (size_t)C * ns_max * sizeof(int)));
    C4_TRY(hipMalloc(&si, (size_t)C * ns_max * sizeof(int)));
    C4_TRY(hipMalloc(&sp, (size_t)C * ns_max * sizeof(double)));
    if (c4_stream_bytes(ns_max) > C4_LDS_STREAMS) C4_TRY(hipMalloc(&ss, (size_t)C * c4_stream_bytes(ns_max)));
    if (b->d_y) {
        C4_TRY(hipMemcpy2DAsync(y, (size_t)(Hy + n_max) * sizeof(double2), b->d_y,
                                (size_t)(Hy + b->n_max) * sizeof(double2), (size_t)Hy * sizeof(double2), C,
                                hipMemcpyDeviceToDevice, st));
        C4_TRY(hipMemcpy2DAsync(z, (size_t)(Hz + n_max) * sizeof(float2), b->d_z,
                                (size_t)(Hz + b->n_max) * sizeof(float2), (size_t)Hz * sizeof(float2), C,
                                hipMemcpyDeviceToDevice, st));
    } else {
        C4_TRY(hipMemsetAsync(y, 0, (size_t)C * (Hy + n_max) * sizeof(double2), st));
        C4_TRY(hipMemsetAsync(z, 0, (size_t)C * (Hz + n_max) * sizeof(float2), st));
    }
    C4_TRY(hipStreamSynchronize(st));   // the copies above
    C4_TRY(hipDeviceSynchronize());     // earlier calls (on whichever stream) may still use the old workspaces
#undef C4_TRY
    (void)hipFree(b->d_y); (void)hipFree(b->d_z); (void)hipFree(b->d_phases); (void)hipFree(b->d_ss);
    (void)hipFree(b->d_sym_x); (void)hipFree(b->d_sym_idx); (void)hipFree(b->d_sym_sp);
    b->d_y = y; b->d_z = z; b->d_phases = ph; b->d_ss = ss; b->d_sym_x = sx; b->d_sym_idx = si; b->d_sym_sp = sp;
    b->n_max = n_max; b->ns_max = ns_max;
    return WH_OK;
}

static int c4fm_zero_state(wh_c4fm_bank *b, hipStream_t st) {
    WH_HIP(hipMemsetAsync(b->d_xh, 0, (size_t)b->C * (b->nl - 1) * sizeof(float2), st));
    WH_HIP(hipMemsetAsync(b->d_y, 0, (size_t)b->C * (b->nr - 1 + b->n_max) * sizeof(double2), st));
    WH_HIP(hipMemsetAsync(b->d_z, 0, (size_t)b->C * (b->overlap + b->n_max) * sizeof(float2), st));
    WH_HIP(hipMemsetAsync(b->d_buffer, 0, (size_t)b->C * BUF_LEN * sizeof(float), st));
    hipLaunchKernelGGL(k_reset, dim3((b->C + 63) / 64), dim3(64), 0, st, b->d_st, b->C, b->sps);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_c4fm_bank_create(wh_c4fm_bank **out, int C, double sps, const float *h_lpf, int nl,
                                   const float *h_rrc, int nr, const float *h_taps, int n_max) {
    if (!out || !h_lpf || !h_rrc || !h_taps) return set_err(WH_E_ARG, "wh_c4fm_bank_create: null");
    if (C < 1 || nl < 2 || nr < 2 || nl > 1024 || nr > 2048 || !(sps >= 4.0) || sps > 512.0 || n_max < 1 ||
        n_max > C4_N_CAP)
        return set_err(WH_E_ARG, "wh_c4fm_bank_create: need sps >= 4, 1 <= max_samples_per_call <= 2^24");
    wh_c4fm_bank *b = new wh_c4fm_bank();
    std::unique_ptr<wh_c4fm_bank, void (*)(wh_c4fm_bank *)> guard(b, wh_c4fm_bank_destroy);  // frees partial state on early return
    b->C = C; b->nl = nl; b->nr = nr; b->sps = sps;
    int fl = (int)floor(sps);
    b->interp_offset = fl - 4 > 0 ? fl - 4 : 0;
    b->overlap = fl + 4;
    double mu = fmod(sps, 1.0);
    int row = (int)((1.0 - mu) * 128.0 + 0.5);
    b->fm_row = row < 0 ? 0 : (row > 128 ? 128 : row);
    WH_HIP(hipMalloc(&b->d_lpf, nl * sizeof(float)));
    WH_HIP(hipMalloc(&b->d_rrc, nr * sizeof(float)));
    WH_HIP(hipMalloc(&b->d_taps, 129 * 8 * sizeof(float)));
    WH_HIP(hipMemcpy(b->d_lpf, h_lpf, nl * sizeof(float), hipMemcpyHostToDevice));
    WH_HIP(hipMemcpy(b->d_rrc, h_rrc, nr * sizeof(float), hipMemcpyHostToDevice));
    WH_HIP(hipMemcpy(b->d_taps, h_taps, 129 * 8 * sizeof(float), hipMemcpyHostToDevice));
    WH_HIP(hipMalloc(&b->d_xh, (size_t)C * (nl - 1) * sizeof(float2)));
    WH_HIP(hipMalloc(&b->d_buffer, (size_t)C * BUF_LEN * sizeof(float)));
    WH_HIP(hipMalloc(&b->d_st, (size_t)C * sizeof(ChanState)));
    int rc = c4fm_resize(b, n_max, nullptr);
    if (rc != WH_OK) return rc;
    rc = c4fm_zero_state(b, nullptr);
    if (rc != WH_OK) return rc;
    WH_HIP(hipDeviceSynchronize());
    *out = guard.release();
    return WH_OK;
}

// Grow the bank's per-call workspaces so that calls of up to n_max samples per channel fit (never shrinks); all
// demodulator state is kept.  Allocates and synchronises the stream: call it outside the hot path.
extern "C" int wh_c4fm_bank_reserve(wh_c4fm_bank *b, size_t n_max, void *stream) {
    if (!b) return set_err(WH_E_ARG, "wh_c4fm_bank_reserve: null handle");
    if (n_max > (size_t)C4_N_CAP) return set_err(WH_E_ARG, "wh_c4fm_bank_reserve: more than 2^24 samples per call");
    if (n_max <= (size_t)b->n_max) return WH_OK;
    return c4fm_resize(b, (int)n_max, as_stream(stream));
}

extern "C" void wh_c4fm_bank_destroy(wh_c4fm_bank *b) {
    if (!b) return;
    (void)hipFree(b->d_lpf); (void)hipFree(b->d_rrc); (void)hipFree(b->d_taps); (void)hipFree(b->d_xh);
    (void)hipFree(b->d_y); (void)hipFree(b->d_z); (void)hipFree(b->d_phases); (void)hipFree(b->d_buffer);
    (void)hipFree(b->d_ss);
    (void)hipFree(b->d_st); (void)hipFree(b->d_sym_x); (void)hipFree(b->d_sym_idx); (void)hipFree(b->d_sym_sp);
    delete b;
}

extern "C" int wh_c4fm_bank_reset(wh_c4fm_bank *b, void *stream) {
    if (!b) return set_err(WH_E_ARG, "wh_c4fm_bank_reset: null handle");
    return c4fm_zero_state(b, as_stream(stream));
}

extern "C" int wh_c4fm_bank_run(wh_c4fm_bank *b, const float *d_iq, size_t n, size_t iq_stride, uint8_t *d_dibits,
                                float *d_soft, size_t out_cap, int32_t *d_counts, void *stream) {
    if (!b) return set_err(WH_E_ARG, "wh_c4fm_bank_run: null handle");
    if (!d_counts) return set_err(WH_E_ARG, "wh_c4fm_bank_run: null counts");
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        WH_HIP(hipMemsetAsync(d_counts, 0, (size_t)b->C * sizeof(int32_t), st));
        return WH_OK;
    }
    if (!d_iq || !d_dibits || !d_soft) return set_err(WH_E_ARG, "wh_c4fm_bank_run: null buffer");
    if (n > (size_t)b->n_max)
        return set_err(WH_E_ARG, "wh_c4fm_bank_run: n exceeds max_samples_per_call (grow with wh_c4fm_bank_reserve)");
    if (iq_stride < n) return set_err(WH_E_ARG, "wh_c4fm_bank_run: iq_stride < n");
    if (out_cap < n / 4 + 2) return set_err(WH_E_ARG, "wh_c4fm_bank_run: out_cap must be >= n/4 + 2");
    C4Args a;
    a.iq = reinterpret_cast<const float2 *>(d_iq);
    a.iq_stride = iq_stride;
    a.n = (int)n; a.n_max = b->n_max; a.C = b->C;
    a.sps = b->sps; a.lagging_offset = b->sps / 2.0; a.max_fine_adj = b->sps * 0.2;
    a.nl = b->nl; a.nr = b->nr; a.overlap = b->overlap; a.interp_offset = b->interp_offset; a.fm_row = b->fm_row;
    a.ns_max = b->ns_max;
    a.ns_call = c4_ns(b->sps, (int)n) < b->ns_max ? c4_ns(b->sps, (int)n) : b->ns_max;
    a.ss = b->d_ss;
    a.lpf = b->d_lpf; a.rrc = b->d_rrc; a.taps = b->d_taps;
    a.xh = b->d_xh; a.y = b->d_y; a.z = b->d_z; a.phases = b->d_phases; a.buffer = b->d_buffer; a.st = b->d_st;
    a.sym_x = b->d_sym_x; a.sym_sp = b->d_sym_sp; a.sym_idx = b->d_sym_idx;
    a.dibits = d_dibits; a.soft = d_soft; a.out_cap = out_cap; a.counts = d_counts;
    const unsigned tiles = (unsigned)((n + 255) / 256);
    size_t sm_l = ((size_t)(b->nl * 4 + 15) & ~(size_t)15) + (size_t)(256 + b->nl - 1) * sizeof(float2);
    size_t sm_r = ((size_t)(b->nr * 4 + 15) & ~(size_t)15) + (size_t)(256 + b->nr - 1) * sizeof(double2);
    hipLaunchKernelGGL(k_lpf, dim3(tiles, b->C), dim3(256), sm_l, st, a);
    WH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rrc, dim3(tiles, b->C), dim3(256), sm_r, st, a);
    WH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fm, dim3(tiles, b->C), dim3(256), 0, st, a);
    WH_LAUNCH_CHECK();
    // symbol clock, sync search, equaliser: one wave per channel; the detector streams of a call sit in LDS when they
    // fit 64 KB (calls up to ~8 000 symbols: 81 000 samples at 48 kS/s), in the workspace otherwise
    const size_t sm_s = c4_stream_bytes(a.ns_call);
    if (sm_s <= C4_LDS_STREAMS) {
        hipLaunchKernelGGL(k_seq<false>, dim3(b->C), dim3(64), sm_s, st, a);
    } else {
        if (!b->d_ss) return set_err(WH_E_ARG, "wh_c4fm_bank_run: stream workspace missing");
        hipLaunchKernelGGL(k_seq<true>, dim3(b->C), dim3(64), 0, st, a);
    }
    WH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_carry, dim3(b->C), dim3(256), 0, st, a);
    WH_LAUNCH_CHECK();
    return WH_OK;
}
