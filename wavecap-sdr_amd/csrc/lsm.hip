// Row A12 (LSM) for gfx950: the P25 Phase-1 CQPSK / linear-simulcast demodulator of the reference's
// decoders/p25.py:190-669 -- block AGC (:436-455), block NCO from the tracked frequency offset (:460-465),
// 63-tap 'same'-mode low-pass per call (:468-471), then the per-sample symbol clock with 8-tap MMSE
// interpolation, differential pi/4-DQPSK slicing, magnitude-weighted frequency loop and Gardner timing
// error (:484-669).  Scalar types follow the reference's NumPy-2 promotion exactly (see oracle/lsm_ref.c,
// whose "portable" flavour this file reproduces bit for bit: explicit *_rn operations in the same order,
// the shared wh_portable_math.h for atan2f / hypotf / sincos).
//   k_lsm_agc    one workgroup per channel: mean |x| (256 strided float64 partials + halving tree), AGC
//                gain update, and the per-call NCO scalars;
//   k_lsm_front  thread per output sample: scale, NCO rotation (complex128) staged once per tile in LDS,
//                63-tap FIR accumulated in float64, cast to complex64 behind a 32-sample carried history;
//   k_lsm_seq    ONE WAVE PER CHANNEL for the feedback part (uniform scalar loop; samples in a register window
//                handed out by v_readlane, interpolator taps in LDS); one LANE per channel beyond 2048 channels;
//   k_lsm_carry  saves the last 32 filtered samples for the next call.
#include "wh_common.h"
#include "wh_portable_math.h"
#include <cmath>
#include <memory>
#include <vector>

using namespace wh;

namespace {

constexpr int NT = 32;    // MMSE_NTAPS: history depth (p25.py:221)
constexpr int NLPF = 63;  // baseband filter taps (p25.py:370)
constexpr int TILE = 256;

struct LsmState {
    double freq_offset, phase_acc, clock64;
    float agc_gain, clock32, symtime32, omega32, prev_re, prev_im;
    int f32mode, first;
};

struct LsmCall {  // per channel, per call
    double phase0, freq;
    float gain;
    int nco;
};

struct LsmArgs {
    const float2 *iq;
    size_t iq_stride;
    int n, n_max, C;
    double sps, symtime64;
    int half_sps, full_sps, gardner;
    const float *lpf;   // [63]
    const float *mmse;  // [129][8]
    LsmState *st;
    LsmCall *call;
    float2 *hist;       // [C][32]
    float2 *filt;       // [C][32 + n_max]
    uint8_t *dibits;
    float *phases;      // optional
    size_t cap;
    int32_t *counts;
};

__global__ __launch_bounds__(256) void k_lsm_agc(LsmArgs a) {
    __shared__ double part[256];
    const int c = blockIdx.x, t = threadIdx.x;
    const float2 *x = a.iq + (size_t)c * a.iq_stride;
    double s = 0.0;
    for (int i = t; i < a.n; i += 256) {
        float2 v = x[i];
        s = __dadd_rn(s, (double)whm_hypotf(v.x, v.y));
    }
    part[t] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (t < w) part[t] = __dadd_rn(part[t], part[t + w]);
        __syncthreads();
    }
    if (t == 0) {
        LsmState &S = a.st[c];
        float mean = (float)__ddiv_rn(part[0], (double)a.n);
        float g = S.agc_gain;
        if (mean > 1e-8f) {
            float target = __fdiv_rn(1.0f, mean);
            g = __fadd_rn(__fmul_rn(g, 0.995f), __fmul_rn(target, 0.005f));
            if (g < 0.01f) g = 0.01f;
            if (g > 500.0f) g = 500.0f;
            S.agc_gain = g;
        }
        LsmCall k;
        k.gain = g;
        k.nco = fabs(S.freq_offset) > 1e-7;
        k.phase0 = S.phase_acc;
        k.freq = S.freq_offset;
        a.call[c] = k;
        if (k.nco) {
            double p = __dadd_rn(S.phase_acc, __dmul_rn(S.freq_offset, (double)a.n));
            S.phase_acc = __dsub_rn(p, __dmul_rn(6.283185307179586, rint(__dmul_rn(p, 0.15915494309189535))));
        }
    }
}

__global__ __launch_bounds__(TILE) void k_lsm_front(LsmArgs a) {
    __shared__ double2 xs[TILE + NLPF - 1];
    __shared__ float tp[NLPF];
    const int c = blockIdx.y, t = threadIdx.x, base = blockIdx.x * TILE;
    const float2 *x = a.iq + (size_t)c * a.iq_stride;
    float2 *y = a.filt + (size_t)c * (NT + a.n_max);
    const LsmCall k = a.call[c];
    if (blockIdx.x == 0 && t < NT) y[t] = a.hist[(size_t)c * NT + t];
    if (t < NLPF) tp[t] = a.lpf[t];
    for (int j = t; j < TILE + NLPF - 1; j += TILE) {
        int s = base - 31 + j;
        double2 v = make_double2(0.0, 0.0);
        if (s >= 0 && s < a.n) {
            float2 q = x[s];
            // complex64 * (g + 0j)
            float xr = __fsub_rn(__fmul_rn(q.x, k.gain), __fmul_rn(q.y, 0.0f));
            float xi = __fadd_rn(__fmul_rn(q.x, 0.0f), __fmul_rn(q.y, k.gain));
            if (k.nco) {
                double th = __dadd_rn(k.phase0, __dmul_rn(k.freq, (double)s));
                double sn, cs;
                whm_sincos_f64(th, &sn, &cs);
                double nr = cs, ni = -sn;  // exp(-1j * th)
                v.x = __dsub_rn(__dmul_rn((double)xr, nr), __dmul_rn((double)xi, ni));
                v.y = __dadd_rn(__dmul_rn((double)xr, ni), __dmul_rn((double)xi, nr));
            } else {
                v.x = (double)xr;
                v.y = (double)xi;
            }
        }
        xs[j] = v;
    }
    __syncthreads();
    const int i = base + t;
    if (i >= a.n) return;
    float2 o;
    if (a.n >= NLPF) {
        double ar = 0.0, ai = 0.0;
        for (int kk = 0; kk < NLPF; ++kk) {  // sample i + 31 - kk sits at xs[t + 62 - kk]; out-of-block samples are +0
            double2 v = xs[t + 62 - kk];
            double h = (double)tp[kk];
            ar = __dadd_rn(ar, __dmul_rn(h, v.x));
            ai = __dadd_rn(ai, __dmul_rn(h, v.y));
        }
        o = make_float2((float)ar, (float)ai);
    } else {
        double2 v = xs[t + 31];
        o = make_float2((float)v.x, (float)v.y);
    }
    y[NT + i] = o;
}

// 8-tap MMSE interpolator at look-back `so` from the newest sample; offsets count back in time, offsets < 0 are skipped
// (p25.py:350); products and sums in the reference's order.  One LANE per channel form: samples from memory.
__device__ __forceinline__ float2 lsm_interp_lane(const float *mmse, const float2 *y, int inew, int so, int imu) {
    float ar = 0.0f, ai = 0.0f;
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
        int off = so + tap - 3;
        if (off < 0 || off >= NT) continue;
        float t = mmse[imu * 8 + tap];
        float2 v = y[inew - off];
        float pr = __fsub_rn(__fmul_rn(t, v.x), __fmul_rn(0.0f, v.y));  // (t + 0j) * (a + bj)
        float pi = __fadd_rn(__fmul_rn(t, v.y), __fmul_rn(0.0f, v.x));
        ar = __fadd_rn(ar, pr);
        ai = __fadd_rn(ai, pi);
    }
    return make_float2(ar, ai);
}

// One WAVE per channel form: the three interpolators of a symbol (now, half a symbol back, a symbol back) side by side.
// The filtered samples sit in a 128-entry circular LDS window (entry idx & 127; the look-back is < 64); lane 16 j + tap
// forms product `tap` of interpolator j, and the eight products of an interpolator are summed IN THE REFERENCE'S ORDER by
// a DPP chain inside its row of 16 lanes: acc <- row_shr:1(acc) + p, eight times -- lane t then holds
// ((0 + p_0) + p_1) ... + p_t (the shift feeds lane 0 of a row +0.0f, the reference's initial sum).  A skipped tap
// contributes -0.0f, the identity of float addition (x + -0.0f == x for every x, +-0 included), so the rounding sequence
// is exactly the reference's with the tap left out.  24 products in one instruction instead of 24 in sequence.
__device__ __forceinline__ void lsm_interp3_wave(const float *mmse_s, const float2 *win, int inew, int so1, int so2, int imu,
                                                 int lane, float2 &curr, float2 &mid, float2 &ps) {
    const int j = lane >> 4, tap = lane & 15;
    const int so = j == 0 ? 0 : (j == 1 ? so1 : so2);
    const int off = so + tap - 3;
    const bool on = j < 3 && tap < 8 && off >= 0 && off < NT;
    float pr = -0.0f, pi = -0.0f;
    if (on) {
        const float t = mmse_s[imu * 8 + tap];
        const float2 v = win[(inew - off) & 127];
        pr = __fsub_rn(__fmul_rn(t, v.x), __fmul_rn(0.0f, v.y));  // (t + 0j) * (a + bj)
        pi = __fadd_rn(__fmul_rn(t, v.y), __fmul_rn(0.0f, v.x));
    }
    float ar = 0.0f, ai = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // row_shr:1 with bound_ctrl: lane t reads lane t - 1 of its row, lane 0 reads 0
        const float sr = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ar), 0x111, 0xf, 0xf, true));
        const float si = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ai), 0x111, 0xf, 0xf, true));
        ar = __fadd_rn(sr, pr);
        ai = __fadd_rn(si, pi);
    }
    curr = make_float2(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ar), 7)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ai), 7)));
    mid = make_float2(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ar), 23)),
                      __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ai), 23)));
    ps = make_float2(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ar), 39)),
                     __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ai), 39)));
}

__device__ __forceinline__ float2 lsm_cdiv_real(float2 z, float m) {  // numpy complex64 / (m + 0j)
    float rat = __fdiv_rn(0.0f, m);
    float scl = __fdiv_rn(1.0f, __fadd_rn(m, __fmul_rn(0.0f, rat)));
    return make_float2(__fmul_rn(__fadd_rn(z.x, __fmul_rn(z.y, rat)), scl),
                       __fmul_rn(__fsub_rn(z.y, __fmul_rn(z.x, rat)), scl));
}

__device__ __forceinline__ float lsm_angle(float im, float re) {
    const float PI_F = 3.14159274101257324f;
    if (re == 0.0f && im == 0.0f) {
        if (__builtin_signbit(re)) return __builtin_signbit(im) ? -PI_F : PI_F;
        return im;
    }
    if (im == 0.0f && re < 0.0f) return __builtin_signbit(im) ? -PI_F : PI_F;
    return whm_atan2f(im, re);
}

// ONE WAVE PER CHANNEL, every lane running the same scalar loop (round 2; one lane per channel before): the loop waits on
// its own samples -- 24 dependent loads per symbol whose addresses follow the symbol clock (3.2 us per symbol, all of it
// memory latency) -- so the samples now sit in an LDS window, the interpolator taps in LDS too, and the 24 products of a
// symbol's three interpolators are formed by 24 lanes at once (lsm_interp3_wave).  The arithmetic is untouched (same
// operations, same order of every sum): outputs and carried state stay bit-identical to oracle/lsm_ref.c.
//
// WAVE = false is the one-LANE-per-channel form (64 channels per wave, samples from memory): 1.7x slower per call, but a
// wave carries 64 channels, so beyond ~2000 channels -- where the wave form has filled every SIMD -- it has the higher
// aggregate rate.  The launch picks by channel count; both forms run the same arithmetic.
template <bool WAVE>
__global__ __launch_bounds__(64) void k_lsm_seq(LsmArgs a) {
    __shared__ float mmse_s[WAVE ? 129 * 8 : 1];
    const int lane = threadIdx.x;
    const int c = WAVE ? (int)blockIdx.x : (int)(blockIdx.x * 64 + threadIdx.x);
    const bool writer = WAVE ? lane == 0 : true;
    const float *mmse = a.mmse;
    if constexpr (WAVE) {
        for (int j = lane; j < 129 * 8; j += 64) mmse_s[j] = a.mmse[j];
        __syncthreads();
        mmse = mmse_s;
    } else {
        if (c >= a.C) return;
    }
    LsmState S = a.st[c];
    const float2 *y = a.filt + (size_t)c * (NT + a.n_max) + NT;
    // circular window of the filtered samples: entry idx & 127 holds sample idx, idx in [base - 64, base + 64) (the carried
    // history covers -32 .. -1; older ones are never read)
    __shared__ float2 win[WAVE ? 128 : 1];
    int base = 0;
    if constexpr (WAVE) {
        win[64 + lane] = lane >= 64 - NT ? y[lane - 64] : make_float2(0.f, 0.f);   // (lane - 64) & 127
        win[lane] = lane < a.n ? y[lane] : make_float2(0.f, 0.f);
        __builtin_amdgcn_wave_barrier();
    }
    uint8_t *dib = a.dibits + (size_t)c * a.cap;
    float *pho = a.phases ? a.phases + (size_t)c * a.cap : nullptr;
    const double PI_D = 3.141592653589793;
    const float PI_F = (float)PI_D, HALF_PI_F = (float)(PI_D / 2), NHALF_PI_F = (float)(-(PI_D / 2));
    const float MU_MAX_F = (float)(1.0 - 1e-6);
    const float BIG = 16777216.0f;  // beyond 2^24 "clock -= 1" no longer changes the value: never spin there
    int count = 0;
    int i = 0;
    while (true) {
        bool fire = false;
        if (!S.f32mode) {
            while (i < a.n) {
                S.clock64 = __dadd_rn(S.clock64, a.symtime64);
                ++i;
                if (S.clock64 >= 1.0) { fire = true; break; }
            }
        } else {
            while (i < a.n) {
                S.clock32 = __fadd_rn(S.clock32, S.symtime32);
                ++i;
                if (S.clock32 >= 1.0f) { fire = true; break; }
            }
        }
        if (!fire) break;
        const int inew = (WAVE ? __builtin_amdgcn_readfirstlane(i) : i) - 1;   // newest sample of this symbol
        if constexpr (WAVE) {
            while (inew >= base + 64) {                                // slide the window
                base += 64;
                __builtin_amdgcn_wave_barrier();
                win[(base + lane) & 127] = base + lane < a.n ? y[base + lane] : make_float2(0.f, 0.f);
                __builtin_amdgcn_wave_barrier();
            }
        }
        int imu;
        if (!S.f32mode) {
            S.clock64 = __dsub_rn(S.clock64, 1.0);
            double mu = __ddiv_rn(S.clock64, a.symtime64);
            if (mu < 0.0) mu = 0.0;
            if (mu > 1.0 - 1e-6) mu = 1.0 - 1e-6;
            imu = (int)rint(__dmul_rn(mu, 128.0));
        } else {
            S.clock32 = __fsub_rn(S.clock32, 1.0f);
            float mu = __fdiv_rn(S.clock32, S.symtime32);
            if (mu < 0.0f) mu = 0.0f;
            if (mu > MU_MAX_F) mu = MU_MAX_F;
            imu = (int)rintf(__fmul_rn(mu, 128.0f));
        }
        if (imu > 128) imu = 128;
        if (imu < 0) imu = 0;  // NaN clock
        float2 curr, mid, ps;
        if constexpr (WAVE) lsm_interp3_wave(mmse, win, inew, a.half_sps, a.full_sps, imu, lane, curr, mid, ps);
        else curr = lsm_interp_lane(mmse, y, inew, 0, imu);
        float curr_mag = whm_hypotf(curr.x, curr.y);
        int dibit;
        float phase_out;
        if (S.first) {
            // complex128(curr) * complex128(0, -0): float64 atan2 of signed zeros
            double ca = (double)curr.x, cb = (double)curr.y;
            double re = __dsub_rn(__dmul_rn(ca, 0.0), __dmul_rn(cb, -0.0));
            double im = __dadd_rn(__dmul_rn(ca, -0.0), __dmul_rn(cb, 0.0));
            double phase = __builtin_signbit(re) ? (__builtin_signbit(im) ? -PI_D : PI_D) : im;
            double expected;
            if (phase >= PI_D / 2) { dibit = 1; expected = 3 * PI_D / 4; }
            else if (phase >= 0) { dibit = 0; expected = PI_D / 4; }
            else if (phase >= -(PI_D / 2)) { dibit = 2; expected = -(PI_D / 4); }
            else { dibit = 3; expected = -(3 * PI_D / 4); }
            double pe = __dsub_rn(phase, expected);
            if (pe > PI_D) pe = __dsub_rn(pe, 2 * PI_D); else if (pe < -PI_D) pe = __dadd_rn(pe, 2 * PI_D);
            S.freq_offset = __dadd_rn(S.freq_offset, __dmul_rn(__dmul_rn(0.0005, pe), (double)curr_mag));
            phase_out = (float)phase;
        } else {
            float2 prev = make_float2(S.prev_re, S.prev_im);
            float prev_mag = whm_hypotf(prev.x, prev.y);
            float dr, di;
            if (curr_mag > 1e-6f && prev_mag > 1e-6f) {
                float2 p = lsm_cdiv_real(curr, curr_mag), q = lsm_cdiv_real(prev, prev_mag);
                q.y = -q.y;
                dr = __fsub_rn(__fmul_rn(p.x, q.x), __fmul_rn(p.y, q.y));
                di = __fadd_rn(__fmul_rn(p.x, q.y), __fmul_rn(p.y, q.x));
            } else {
                float bi = -prev.y;
                dr = __fsub_rn(__fmul_rn(curr.x, prev.x), __fmul_rn(curr.y, bi));
                di = __fadd_rn(__fmul_rn(curr.x, bi), __fmul_rn(curr.y, prev.x));
            }
            float phase = lsm_angle(di, dr);
            float expected;
            if (phase >= HALF_PI_F) { dibit = 1; expected = (float)(3 * PI_D / 4); }
            else if (phase >= 0.0f) { dibit = 0; expected = (float)(PI_D / 4); }
            else if (phase >= NHALF_PI_F) { dibit = 2; expected = (float)(-(PI_D / 4)); }
            else { dibit = 3; expected = (float)(-(3 * PI_D / 4)); }
            float pe = __fsub_rn(phase, expected);
            if (pe > PI_F) pe = __fsub_rn(pe, (float)(2 * PI_D)); else if (pe < -PI_F) pe = __fadd_rn(pe, (float)(2 * PI_D));
            float f = __fmul_rn(__fmul_rn(0.0005f, pe), curr_mag);
            S.freq_offset = __dadd_rn(S.freq_offset, (double)f);
            phase_out = phase;
        }
        if (S.freq_offset < -0.02) S.freq_offset = -0.02;
        if (S.freq_offset > 0.02) S.freq_offset = 0.02;
        if ((size_t)count < a.cap && writer) {
            dib[count] = (uint8_t)dibit;
            if (pho) pho[count] = phase_out;
        }
        ++count;
        if (a.gardner) {
            if constexpr (!WAVE) {
                mid = lsm_interp_lane(mmse, y, inew, a.half_sps, imu);
                ps = lsm_interp_lane(mmse, y, inew, a.full_sps, imu);
            }
            float er = __fsub_rn(curr.x, ps.x), ei = __fsub_rn(curr.y, ps.y);
            float ted = __fsub_rn(__fmul_rn(er, mid.x), __fmul_rn(ei, -mid.y));
            float step = __fmul_rn(0.015f, ted);
            if (!S.f32mode) {
                S.clock32 = __fadd_rn((float)S.clock64, step);
                S.omega32 = __fadd_rn((float)a.sps, __fmul_rn(0.0f, ted));
                S.f32mode = 1;
            } else {
                S.clock32 = __fadd_rn(S.clock32, step);
                S.omega32 = __fadd_rn(S.omega32, __fmul_rn(0.0f, ted));
            }
            S.symtime32 = __fdiv_rn(1.0f, S.omega32);
        }
        if (!S.f32mode) {
            while (S.clock64 >= 1.0 && S.clock64 < 9007199254740992.0) S.clock64 = __dsub_rn(S.clock64, 1.0);
            while (S.clock64 < 0.0 && S.clock64 > -9007199254740992.0) S.clock64 = __dadd_rn(S.clock64, 1.0);
        } else {
            while (S.clock32 >= 1.0f && S.clock32 < BIG) S.clock32 = __fsub_rn(S.clock32, 1.0f);
            while (S.clock32 < 0.0f && S.clock32 > -BIG) S.clock32 = __fadd_rn(S.clock32, 1.0f);
        }
        S.prev_re = curr.x;
        S.prev_im = curr.y;
        S.first = 0;
    }
    if (writer) {
        a.st[c] = S;
        a.counts[c] = count;
    }
}

__global__ void k_lsm_carry(LsmArgs a) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.C * NT) return;
    int c = idx / NT, j = idx % NT;
    a.hist[idx] = a.filt[(size_t)c * (NT + a.n_max) + a.n + j];
}

}  // namespace

// measured crossover of the two k_lsm_seq forms (tools/symbol_bank_scaling.py: 1024 ch 12.5 ms, 4096 ch 24.5 vs 15.6 ms)
static constexpr int LSM_WAVE_FORM_MAX_CHANNELS = 2048;

struct wh_lsm_bank {
    int C, n_max;
    double sps;
    float *d_lpf = nullptr, *d_mmse = nullptr;
    LsmState *d_st = nullptr;
    LsmCall *d_call = nullptr;
    float2 *d_hist = nullptr, *d_filt = nullptr;
};

static int lsm_reset(wh_lsm_bank *b, hipStream_t st) {
    std::vector<LsmState> z((size_t)b->C);
    for (auto &s : z) {
        s = LsmState{};
        s.agc_gain = 1.0f;
        s.first = 1;
    }
    WH_HIP(hipMemcpyAsync(b->d_st, z.data(), z.size() * sizeof(LsmState), hipMemcpyHostToDevice, st));
    WH_HIP(hipMemsetAsync(b->d_hist, 0, (size_t)b->C * NT * sizeof(float2), st));
    WH_HIP(hipStreamSynchronize(st));
    return WH_OK;
}

extern "C" int wh_lsm_bank_create(wh_lsm_bank **out, int C, double sps, const float *h_lpf, const float *h_mmse,
                                  int n_max) {
    if (!out || !h_lpf || !h_mmse || C < 1 || !(sps >= 2.0) || !(sps < 1e6) || n_max < 1)
        return set_err(WH_E_ARG, "wh_lsm_bank_create: bad arguments");
    wh_lsm_bank *b = new wh_lsm_bank();
    std::unique_ptr<wh_lsm_bank, void (*)(wh_lsm_bank *)> guard(b, wh_lsm_bank_destroy);  // frees partial state on early return
    b->C = C; b->n_max = n_max; b->sps = sps;
    WH_HIP(hipMalloc(&b->d_lpf, NLPF * sizeof(float)));
    WH_HIP(hipMalloc(&b->d_mmse, 129 * 8 * sizeof(float)));
    WH_HIP(hipMemcpy(b->d_lpf, h_lpf, NLPF * sizeof(float), hipMemcpyHostToDevice));
    WH_HIP(hipMemcpy(b->d_mmse, h_mmse, 129 * 8 * sizeof(float), hipMemcpyHostToDevice));
    WH_HIP(hipMalloc(&b->d_st, (size_t)C * sizeof(LsmState)));
    WH_HIP(hipMalloc(&b->d_call, (size_t)C * sizeof(LsmCall)));
    WH_HIP(hipMalloc(&b->d_hist, (size_t)C * NT * sizeof(float2)));
    WH_HIP(hipMalloc(&b->d_filt, (size_t)C * (NT + (size_t)n_max) * sizeof(float2)));
    int rc = lsm_reset(b, nullptr);
    if (rc != WH_OK) return rc;
    *out = guard.release();
    return WH_OK;
}

extern "C" void wh_lsm_bank_destroy(wh_lsm_bank *b) {
    if (!b) return;
    (void)hipFree(b->d_lpf); (void)hipFree(b->d_mmse); (void)hipFree(b->d_st); (void)hipFree(b->d_call);
    (void)hipFree(b->d_hist); (void)hipFree(b->d_filt);
    delete b;
}

// The reference's demodulate(iq) takes a call of any length (decoders/p25.py:413); only the per-call work buffer of the
// low-passed samples depends on the call size -- the carried state (AGC, loops, filter history) does not -- so a longer call
// than the bank was created for grows that buffer and nothing else.  Synchronises the stream (a steady caller never grows).
extern "C" int wh_lsm_bank_reserve(wh_lsm_bank *b, int n_max, void *stream) {
    if (!b || n_max < 1) return set_err(WH_E_ARG, "wh_lsm_bank_reserve: bad arguments");
    if (n_max <= b->n_max) return WH_OK;
    WH_HIP(hipStreamSynchronize(as_stream(stream)));
    float2 *nf = nullptr;
    WH_HIP(hipMalloc(&nf, (size_t)b->C * (NT + (size_t)n_max) * sizeof(float2)));
    (void)hipFree(b->d_filt);
    b->d_filt = nf;
    b->n_max = n_max;
    return WH_OK;
}

extern "C" int wh_lsm_bank_reset(wh_lsm_bank *b, void *stream) {
    if (!b) return set_err(WH_E_ARG, "wh_lsm_bank_reset: null handle");
    return lsm_reset(b, as_stream(stream));
}

extern "C" int wh_lsm_bank_get_state(wh_lsm_bank *b, int channel, double *h_out, void *stream) {
    if (!b || !h_out || channel < 0 || channel >= b->C) return set_err(WH_E_ARG, "wh_lsm_bank_get_state: bad arguments");
    LsmState s;
    WH_HIP(hipStreamSynchronize(as_stream(stream)));
    WH_HIP(hipMemcpy(&s, b->d_st + channel, sizeof(s), hipMemcpyDeviceToHost));
    h_out[0] = s.agc_gain;
    h_out[1] = s.freq_offset;
    h_out[2] = s.phase_acc;
    h_out[3] = s.f32mode ? (double)s.clock32 : s.clock64;
    h_out[4] = s.prev_re;
    h_out[5] = s.prev_im;
    h_out[6] = s.f32mode;
    return WH_OK;
}

extern "C" int wh_lsm_bank_run(wh_lsm_bank *b, const float *d_iq, size_t n, size_t iq_stride, uint8_t *d_dibits,
                               float *d_phases, size_t cap, int32_t *d_counts, void *stream) {
    if (!b || !d_counts) return set_err(WH_E_ARG, "wh_lsm_bank_run: null handle/counts");
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        WH_HIP(hipMemsetAsync(d_counts, 0, (size_t)b->C * sizeof(int32_t), st));
        return WH_OK;
    }
    if (!d_iq || !d_dibits) return set_err(WH_E_ARG, "wh_lsm_bank_run: null buffer");
    if (n > (size_t)b->n_max || iq_stride < n) return set_err(WH_E_ARG, "wh_lsm_bank_run: n too large / bad stride");
    if (cap < n) return set_err(WH_E_ARG, "wh_lsm_bank_run: cap must be >= n (at most one symbol per sample)");
    if (b->C > 65535) return set_err(WH_E_ARG, "wh_lsm_bank_run: too many channels");
    LsmArgs a;
    a.iq = reinterpret_cast<const float2 *>(d_iq);
    a.iq_stride = iq_stride;
    a.n = (int)n; a.n_max = b->n_max; a.C = b->C;
    a.sps = b->sps;
    a.symtime64 = 1.0 / b->sps;
    a.half_sps = (int)std::nearbyint(b->sps / 2.0);  // int(round(x)): half to even
    a.full_sps = (int)std::nearbyint(b->sps);
    a.gardner = a.full_sps + 4 < NT;
    a.lpf = b->d_lpf; a.mmse = b->d_mmse;
    a.st = b->d_st; a.call = b->d_call; a.hist = b->d_hist; a.filt = b->d_filt;
    a.dibits = d_dibits; a.phases = d_phases; a.cap = cap; a.counts = d_counts;
    hipLaunchKernelGGL(k_lsm_agc, dim3(b->C), dim3(256), 0, st, a);
    WH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_lsm_front, dim3((unsigned)((n + TILE - 1) / TILE), b->C), dim3(TILE), 0, st, a);
    WH_LAUNCH_CHECK();
    if (b->C <= LSM_WAVE_FORM_MAX_CHANNELS)
        hipLaunchKernelGGL(k_lsm_seq<true>, dim3(b->C), dim3(64), 0, st, a);
    else
        hipLaunchKernelGGL(k_lsm_seq<false>, dim3((b->C + 63) / 64), dim3(64), 0, st, a);
    WH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_lsm_carry, dim3((b->C * NT + 255) / 256), dim3(256), 0, st, a);
    WH_LAUNCH_CHECK();
    return WH_OK;
}
