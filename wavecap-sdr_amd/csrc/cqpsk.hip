// Row A12 for gfx950: P25 Phase-2 CQPSK receive chain (reference dsp/p25/cqpsk.py:199-350: RRC
// matched filter, Costas loop :84-196, differential decode :308-350) with Mueller-Muller timing
// recovery (dsp/p25/symbol_timing.py:214-380), and the standalone Gardner TED
// (symbol_timing.py:60-211).  The reference computes all of it in Python float / complex128.
//   k_cq_rrc   matched filter, complex128, one thread per output of the full convolution
//              (the reference re-seeds lfilter with zi = state * iq[0] on every call -- kept);
//   k_cq_seq   ONE WAVE PER CHANNEL: the Costas PLL feeds back every sample, so the loop is
//              sequential in time and its speed is the latency of one dependent float64 chain;
//              channels spread over the SIMDs (see the kernel).  Float64, the loop equations with
//              explicit *_rn ops in the order of oracle/cqpsk_ref.c; sincos / atan2 by short
//              polynomial chains (1e-15), distances compared squared;
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
    double2 *symws;            // [C][symcap]: the symbols when the caller wants none (the decode kernel reads them)
    size_t symcap;
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
    // (x / 2.0 written as x * 0.5: the same double bit for bit -- halving is exact -- without the IEEE division sequence;
    // four of them per call made a symbol cost ~1 600 cycles)
    double c0 = v1;
    double c1 = DM(DS(v2, v0), 0.5);
    double c2 = DS(DA(DS(v0, DM(DM(5.0, v1), 0.5)), DM(2.0, v2)), DM(v3, 0.5));
    double c3 = DA(DM(DS(v3, v0), 0.5), DM(DM(3.0, DS(v1, v2)), 0.5));
    return DA(c0, DM(mu, DA(c1, DM(mu, DA(c2, DM(mu, c3))))));
}

// ---- float64 elementary functions for the feedback loop: accuracy ~1e-15 (the parity bar on the symbols is 1e-9), built
// for a SHORT DEPENDENT CHAIN -- the loop below is one wave walking one channel sample by sample, so its time is the
// latency of this chain, not instruction throughput (library sincos / atan2 are 3-5x deeper, with slow paths and scratch).

// sin and cos of x in [-pi, pi] (the Costas phase is wrapped every sample): quadrant reduction by pi/2 in two parts, then
// the fdlibm kernels on [-pi/4, pi/4], Horner in fused multiply-adds; the two chains are independent (ILP 2)
__device__ __forceinline__ void cq_sincos(double x, double &sn, double &cs) {
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double sp = 1.58969099521155010221e-10;
    sp = fma(sp, z, -2.50507602534068634195e-08);
    sp = fma(sp, z, 2.75573137070700676789e-06);
    sp = fma(sp, z, -1.98412698298579493134e-04);
    sp = fma(sp, z, 8.33333333332248946124e-03);
    sp = fma(sp, z, -1.66666666666666324348e-01);
    const double s = fma(r * z, sp, r);
    double cp = -1.13596475577881948265e-11;
    cp = fma(cp, z, 2.08757232129817482790e-09);
    cp = fma(cp, z, -2.75573143513906633035e-07);
    cp = fma(cp, z, 2.48015872894767294178e-05);
    cp = fma(cp, z, -1.38888888888741095749e-03);
    cp = fma(cp, z, 4.16666666666666019037e-02);
    const double c = fma(z * z, cp, fma(-0.5, z, 1.0));
    const int q = (int)k & 3;
    sn = q == 0 ? s : q == 1 ? c : q == 2 ? -s : -c;
    cs = q == 0 ? c : q == 1 ? -s : q == 2 ? -c : s;
}

// atan(t) / t = g(t^2) on t in [0, 1]: degree-17 interpolant (tools/gen_atan_coeffs.py, max error 1.1e-15), Estrin scheme
__device__ __forceinline__ double cq_atan_g(double z) {
    const double z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    const double p0 = fma(-3.33333333332591741e-01, z, 9.99999999999999001e-01);
    const double p1 = fma(-1.42857139184933563e-01, z, 1.99999999916576798e-01);
    const double p2 = fma(-9.09078887746185699e-02, z, 1.11111026064682761e-01);
    const double p3 = fma(-6.65914387633817773e-02, z, 7.69117547592351042e-02);
    const double p4 = fma(-5.12804947274475356e-02, z, 5.84566961719584707e-02);
    const double p5 = fma(-3.48821635173907907e-02, z, 4.37765634493981481e-02);
    const double p6 = fma(-1.44677131781058912e-02, z, 2.45987495873041459e-02);
    const double p7 = fma(-2.20104195614385977e-03, z, 6.64853643130743337e-03);
    const double p8 = fma(-4.58125276638879266e-05, z, 4.61862979263086531e-04);
    const double q0 = fma(p1, z2, p0), q1 = fma(p3, z2, p2), q2 = fma(p5, z2, p4), q3 = fma(p7, z2, p6);
    const double r0 = fma(q1, z4, q0), r1 = fma(q3, z4, q2);
    return fma(fma(p8, z8, r1), z8, r0);     // r0 + z^8 (r1 + z^8 p8)
}

__device__ __forceinline__ double cq_atan2(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    double r = __builtin_amdgcn_rcp(mx);             // ~2^-26 relative; two Newton steps -> full precision
    r = fma(fma(-mx, r, 1.0), r, r);
    r = fma(fma(-mx, r, 1.0), r, r);
    const double t = mx > 0.0 ? mn * r : 0.0;        // atan2(0, 0) = 0
    double a = t * cq_atan_g(t * t);
    a = ay > ax ? 1.5707963267948966 - a : a;
    a = x < 0.0 ? PI_D - a : a;
    return copysign(a, y);
}

// ONE WAVE PER CHANNEL.  The Costas loop feeds back every sample, but only through a PHASE: the sample is rotated by
// exp(-j phi_n) and the detector takes the angle of the result, which is arg(x_n) - phi_n (mod 2 pi) -- so the feedback
// path itself is the short recurrence
//     d = theta_n - phi;  e = d - rint(d / q) q;  f = clamp(f + ki e);  phi = wrap(phi + kp e + f)        (q = pi / 4)
// with theta_n = arg(x_n) known in advance, and the two transcendental evaluations per sample (sincos of phi for the
// rotation, atan2 for the angle) leave the sequential path (round 2 walked ~120 dependent float64 instructions per sample,
// 22 ms per 64 channels x 1 s).  Per block of 64 samples: lane l computes theta_l (one atan2 per lane, parallel), the wave
// runs the recurrence over the block as uniform scalar code and hands phi_n to lane n, the lanes rotate their samples
// (sincos in parallel, the same products as before), then the timing loop (Mueller-Muller, one decision per symbol) walks
// the rotated samples.  Against the reference's complex128 arithmetic the detector angle differs by the rounding of
// theta - phi instead of that of the rotated product (both ~2e-16; the loop is contractive): symbols agree to ~1e-13,
// dibits exactly (tests: 1e-9 / equal).
struct CqCostas {
    double phase, freq;
};

// the recurrence over the m samples of a block: theta (lane-resident) in, phi_n (the phase sample n is rotated by: the
// loop's phase BEFORE its update) out in lane n
__device__ __forceinline__ double cq_costas_block(CqCostas &c, double theta, int m, int lane, double kp, double ki, double maxf) {
    const double q = PI_D / 4, inv_q = 1.0 / (PI_D / 4);
    double myphi = 0.0;
    for (int j = 0; j < m; ++j) {
        const double th = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(theta), j),
                                           __builtin_amdgcn_readlane(__double2loint(theta), j));
        myphi = lane == j ? c.phase : myphi;
        const double d = DS(th, c.phase);
        // (an exactly-zero sample arrives as NaN: its rotated value is (+-0, +-0), whose angle is 0 or +-pi in the reference
        // -- a multiple of pi / 4 either way, so the detector error is 0 and the loop coasts)
        const double err = th != th ? 0.0 : DS(d, DM(rint(DM(d, inv_q)), q));
        c.freq = fmin(fmax(DA(c.freq, DM(ki, err)), -maxf), maxf);
        double ph = DA(c.phase, DA(DM(kp, err), c.freq));
        // the reference's `while phase > pi: phase -= 2 pi` loops as selects: |phase| <= pi before the step and the step is
        // at most max_freq + kp pi / 8 < 2 pi, so each loop runs at most once (no branch on a vector compare per sample)
        ph = ph > PI_D ? DS(ph, 2 * PI_D) : ph;
        ph = ph < -PI_D ? DA(ph, 2 * PI_D) : ph;
        c.phase = ph;
    }
    return myphi;
}

// Two waves per channel, a pipeline over blocks of 64 samples: wave 0 runs the carrier loop of block i (angles, the phase
// recurrence, the rotation) and leaves the rotated samples in LDS, wave 1 runs the timing loop over block i - 1; one
// workgroup barrier per block.  (One wave doing both took 4.1 + 6.5 ms per 64 channels x 1 s; the timing loop is the
// longer stage.)
__global__ __launch_bounds__(128) void k_cq_seq(CqArgs a) {
    __shared__ double2 rot[2][64];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    CqState *const sp = a.st + c;
    // (field by field: a whole-struct copy of the state, with its buf[4] array, lands in scratch)
    struct { double t_phase, t_integ; double2 prev_sym, prev_dec; } s;
    CqCostas cl{sp->c_phase, sp->c_freq};
    s.t_phase = sp->t_phase; s.t_integ = sp->t_integ;
    s.prev_sym = sp->prev_sym; s.prev_dec = sp->prev_dec;
    const double2 sb0 = sp->buf[0], sb1 = sp->buf[1], sb2 = sp->buf[2], sb3 = sp->buf[3];
    const double2 *full = a.full + (size_t)c * (a.n_max + a.L - 1);
    double2 *sym = a.symbols ? a.symbols + (size_t)c * a.cap : a.symws + (size_t)c * a.symcap;
    const size_t sym_lim = a.symbols ? a.cap : (a.cap < a.symcap ? a.cap : a.symcap);   // the buffer the symbols go to
    const double inv_sps = 1.0 / a.sps;
    const double R2 = 0.70710678118654746;   // 1 / 1.4142135623730951 as the reference computes it (kr / inv)
    // the 4-entry circular buffer of the timing loop, oldest first
    const int bi = sp->buf_idx & 3;
    const double2 h0 = bi == 0 ? sb1 : bi == 1 ? sb2 : bi == 2 ? sb3 : sb0;
    const double2 h1 = bi == 0 ? sb2 : bi == 1 ? sb3 : bi == 2 ? sb0 : sb1;
    const double2 h2 = bi == 0 ? sb3 : bi == 1 ? sb0 : bi == 2 ? sb1 : sb2;
    const double2 h3 = bi == 0 ? sb0 : bi == 1 ? sb1 : bi == 2 ? sb2 : sb3;
    // rotated samples (wave 1): lane l of (crl, cil) holds sample l of the block at hand, lane 63 - i of (pcr, pci) the
    // sample i before it (so the four samples a symbol interpolates are always one readlane away)
    double pcr = lane == 60 ? h0.x : lane == 61 ? h1.x : lane == 62 ? h2.x : lane == 63 ? h3.x : 0.0;
    double pci = lane == 60 ? h0.y : lane == 61 ? h1.y : lane == 62 ? h2.y : lane == 63 ? h3.y : 0.0;
#define CQ_RL(v, i) __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), (i)), __builtin_amdgcn_readlane(__double2loint(v), (i)))
    int count = 0;
    const int nb = (a.n + 63) / 64;
    for (int blk = 0; blk <= nb; ++blk) {
        if (wave == 0 && blk < nb) {
            // carrier loop of block blk: angles in parallel, the recurrence in sequence, the rotation in parallel
            const int t0 = blk * 64;
            const double2 xl = t0 + lane < a.n ? full[t0 + lane] : make_double2(0.0, 0.0);
            const int m = a.n - t0 < 64 ? a.n - t0 : 64;
            const double theta = (xl.x == 0.0 && xl.y == 0.0) ? __longlong_as_double(0x7ff8000000000000LL) : cq_atan2(xl.y, xl.x);
            const double phi = cq_costas_block(cl, theta, m, lane, a.c_kp, a.c_ki, a.c_maxf);
            double sn, cs;
            cq_sincos(phi, sn, cs);
            rot[blk & 1][lane] = make_double2(DS(DM(xl.x, cs), DM(xl.y, -sn)),     // sample * exp(-j phi), the reference's
                                              DA(DM(xl.x, -sn), DM(xl.y, cs)));    // products
        }
        if (wave == 1 && blk >= 1) {
            const int t0 = (blk - 1) * 64;
            const int m = a.n - t0 < 64 ? a.n - t0 : 64;
            const double2 rv = rot[(blk - 1) & 1][lane];
            const double crl = rv.x, cil = rv.y;
            // timing recovery: the reference adds 1.0 to its phase per sample and fires when it reaches sps; here the wave
            // steps from symbol to symbol -- k = ceil(sps - phase) samples ahead (checked against the sequential
            // additions, which are kept: they round) -- instead of testing every sample
            int j = -1;                                // last sample consumed (block relative)
            while (true) {
                double need = ceil(DS(a.sps, s.t_phase));
                int k = __builtin_amdgcn_readfirstlane((int)fmin(fmax(need, 1.0), 1.0e6));
                if (j + k > m - 1) {                   // the next symbol lies beyond this block: consume the rest
                    for (int i = j; i < m - 1; ++i) s.t_phase = DA(s.t_phase, 1.0);
                    break;
                }
                double tp = s.t_phase, tprev = tp;
                for (int i = 0; i < k; ++i) { tprev = tp; tp = DA(tp, 1.0); }
                if (k > 1 && tprev >= a.sps) { tp = tprev; k -= 1; }          // (rounding of the sequential sums: at most one off)
                else if (!(tp >= a.sps)) {
                    if (j + k + 1 > m - 1) {           // the correction step crosses the block end
                        s.t_phase = tp;
                        j += k;
                        continue;
                    }
                    tp = DA(tp, 1.0); k += 1;
                }
                j += k;
                s.t_phase = tp;
                double2 b0, b1, b2, b3;
                if (__builtin_expect(j >= 3, 1)) {     // all four samples in this block (all but its first symbol)
                    b0 = make_double2(CQ_RL(crl, j - 3), CQ_RL(cil, j - 3));
                    b1 = make_double2(CQ_RL(crl, j - 2), CQ_RL(cil, j - 2));
                    b2 = make_double2(CQ_RL(crl, j - 1), CQ_RL(cil, j - 1));
                    b3 = make_double2(CQ_RL(crl, j), CQ_RL(cil, j));
                } else {
#define CQ_FETCH(dst, r)                                                     \
                    if ((r) >= 0) dst = make_double2(CQ_RL(crl, (r)), CQ_RL(cil, (r)));  \
                    else dst = make_double2(CQ_RL(pcr, 64 + (r)), CQ_RL(pci, 64 + (r)));
                    CQ_FETCH(b0, j - 3) CQ_FETCH(b1, j - 2) CQ_FETCH(b2, j - 1) CQ_FETCH(b3, j)
#undef CQ_FETCH
                }
                s.t_phase = DS(s.t_phase, a.sps);
                double mu = DM(s.t_phase, inv_sps);
                double sr = interp1(b0.x, b1.x, b2.x, b3.x, mu);
                double si = interp1(b0.y, b1.y, b2.y, b3.y, mu);
                // nearest constellation point (+-1 +-i) / sqrt 2, first minimum in the reference's order k = 0..3; the
                // squared distances order like the reference's hypot()s
                int best = 0;
                double bd = 0.0;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double kr = (kk == 0 || kk == 3) ? R2 : -R2, ki = (kk < 2) ? R2 : -R2;
                    const double dx = DS(kr, sr), dy = DS(ki, si);
                    const double d = fma(dx, dx, dy * dy);
                    if (kk == 0 || d < bd) { bd = d; best = kk; }
                }
                double dr = (best == 0 || best == 3) ? R2 : -R2, di = best < 2 ? R2 : -R2;
                double e1 = DS(DM(s.prev_dec.x, sr), DM(-s.prev_dec.y, si));
                double e2 = DS(DM(dr, s.prev_sym.x), DM(-di, s.prev_sym.y));
                double e = DS(e1, e2);
                s.t_integ = DA(s.t_integ, DM(a.t_ki, e));
                s.t_integ = fmin(fmax(s.t_integ, -a.t_maxdev), a.t_maxdev);
                s.t_phase = DA(s.t_phase, DA(DM(a.t_kp, e), s.t_integ));
                s.prev_sym = make_double2(sr, si);
                s.prev_dec = make_double2(dr, di);
                // (the differential decode of the symbol -- an atan2 and a rounding, no feedback -- is k_cq_decode's)
                if ((size_t)count < sym_lim) {
                    if (lane == 0) sym[count] = make_double2(sr, si);
                    count++;
                }
            }
            // the block becomes history: lane 63 - i <- the sample i before the next block
            {
                const int sh = 64 - m;                 // 0 for a full block
                const double ncr = __shfl(crl, lane - sh), nci = __shfl(cil, lane - sh);
                const double ocr = __shfl(pcr, (lane + m) & 63), oci = __shfl(pci, (lane + m) & 63);
                pcr = lane >= sh ? ncr : ocr;
                pci = lane >= sh ? nci : oci;
            }
        }
        __syncthreads();
    }
    if (wave == 0 && lane == 0) { sp->c_phase = cl.phase; sp->c_freq = cl.freq; }
    if (wave == 1) {
        if (lane == 0) {
            sp->t_phase = s.t_phase; sp->t_integ = s.t_integ;
            sp->prev_sym = s.prev_sym; sp->prev_dec = s.prev_dec;
            sp->buf_idx = 3;
            a.counts[c] = count;
        }
        // the last four rotated samples (oldest first) are the carried buffer
        if (lane >= 60) sp->buf[lane - 60] = make_double2(pcr, pci);
    }
#undef CQ_RL
}

// pi/4-DQPSK differential decode (cqpsk.py:308-350) of the symbols k_cq_seq produced, one thread per symbol: the phase
// of a symbol and of its predecessor (the carried prev_phase for the first), the difference wrapped and rounded to a
// multiple of pi / 4.  Nothing feeds back, so it is no part of the sequential kernel.
__global__ __launch_bounds__(256) void k_cq_decode(CqArgs a) {
    const int c = blockIdx.x, tid = threadIdx.x;
    CqState *const sp = a.st + c;
    const double2 *sym = a.symbols ? a.symbols + (size_t)c * a.cap : a.symws + (size_t)c * a.symcap;
    uint8_t *dib = a.dibits + (size_t)c * a.cap;
    const int count = a.counts[c];
    const double carried = sp->prev_phase;
    const double inv_q = 1.0 / (PI_D / 4);
    __syncthreads();                               // every thread has the carried phase before it is replaced
    for (int i = tid; i < count; i += 256) {
        const double2 v = sym[i];
        const double p = cq_atan2(v.y, v.x);
        double pp = carried;
        if (i > 0) { const double2 u = sym[i - 1]; pp = cq_atan2(u.y, u.x); }
        double dp = DS(p, pp);
        dp = dp > PI_D ? DS(dp, 2 * PI_D) : dp;
        dp = dp < -PI_D ? DA(dp, 2 * PI_D) : dp;
        long long idx = (long long)rint(DM(DA(dp, PI_D), inv_q));
        idx &= 7;                                  // == ((idx % 8) + 8) % 8 in two's complement
        dib[i] = (uint8_t)(idx >> 1);
        if (i == count - 1) sp->prev_phase = p;
    }
}

// ---- standalone drop-ins of the two loops of the chain (dsp/p25/cqpsk.py:84-196 CostasLoop, symbol_timing.py:214-380
// MuellerMullerTED): the same per-sample recurrences as k_cq_seq, one wave per channel, complex128 in and out.
struct CostasState { double phase, freq; };

__global__ __launch_bounds__(64) void k_costas(const double2 *x, size_t stride, int n, CostasState *st, double kp,
                                               double ki, double maxf, double2 *out, size_t out_stride) {
    const int c = blockIdx.x, lane = threadIdx.x;
    CqCostas cl{st[c].phase, st[c].freq};
    const double2 *xc = x + (size_t)c * stride;
    double2 *oc = out + (size_t)c * out_stride;
    for (int t0 = 0; t0 < n; t0 += 64) {
        const double2 xl = t0 + lane < n ? xc[t0 + lane] : make_double2(0.0, 0.0);
        const int m = n - t0 < 64 ? n - t0 : 64;
        const double th = (xl.x == 0.0 && xl.y == 0.0) ? __longlong_as_double(0x7ff8000000000000LL) : cq_atan2(xl.y, xl.x);
        const double phi = cq_costas_block(cl, th, m, lane, kp, ki, maxf);   // (see k_cq_seq)
        double sn, cs;
        cq_sincos(phi, sn, cs);
        if (t0 + lane < n)                                     // sample * exp(-j phi): one coalesced store per 64
            oc[t0 + lane] = make_double2(DS(DM(xl.x, cs), DM(xl.y, -sn)), DA(DM(xl.x, -sn), DM(xl.y, cs)));
    }
    if (lane == 0) { st[c].phase = cl.phase; st[c].freq = cl.freq; }
}

struct MmState {
    double phase, integ;
    double2 b0, b1, b2, b3;      // b3 = newest
    double2 prev_sym, prev_dec;
};

__global__ __launch_bounds__(64) void k_mm(const double2 *x, size_t stride, int n, MmState *st, double sps, double kp,
                                           double ki, double2 *symbols, double2 *decisions, double *errors, size_t cap,
                                           int *counts) {
    const int c = blockIdx.x, lane = threadIdx.x;
    MmState *sp = st + c;
    double phase = sp->phase, integ = sp->integ;
    double2 b0 = sp->b0, b1 = sp->b1, b2 = sp->b2, b3 = sp->b3, prev_sym = sp->prev_sym, prev_dec = sp->prev_dec;
    const double2 *xc = x + (size_t)c * stride;
    const double inv_sps = 1.0 / sps, maxdev = sps / 4, R2 = 0.70710678118654746;
    int count = 0;
    for (int t0 = 0; t0 < n; t0 += 64) {
        const double2 xl = t0 + lane < n ? xc[t0 + lane] : make_double2(0.0, 0.0);
        const int m = n - t0 < 64 ? n - t0 : 64;
        for (int j = 0; j < m; ++j) {
            const int xlo = __builtin_amdgcn_readlane(__double2loint(xl.x), j), xhi = __builtin_amdgcn_readlane(__double2hiint(xl.x), j);
            const int ylo = __builtin_amdgcn_readlane(__double2loint(xl.y), j), yhi = __builtin_amdgcn_readlane(__double2hiint(xl.y), j);
            b0 = b1; b1 = b2; b2 = b3; b3 = make_double2(__hiloint2double(xhi, xlo), __hiloint2double(yhi, ylo));
            phase = DA(phase, 1.0);
            if (phase >= sps) {
                phase = DS(phase, sps);
                const double mu = DM(phase, inv_sps);
                const double sr = interp1(b0.x, b1.x, b2.x, b3.x, mu), si = interp1(b0.y, b1.y, b2.y, b3.y, mu);
                int best = 0;
                double bd = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double kr = (k == 0 || k == 3) ? R2 : -R2, kq = (k < 2) ? R2 : -R2;
                    const double dx = DS(kr, sr), dy = DS(kq, si);
                    const double d = fma(dx, dx, dy * dy);
                    if (k == 0 || d < bd) { bd = d; best = k; }
                }
                const double dr = (best == 0 || best == 3) ? R2 : -R2, di = best < 2 ? R2 : -R2;
                const double e = DS(DS(DM(prev_dec.x, sr), DM(-prev_dec.y, si)), DS(DM(dr, prev_sym.x), DM(-di, prev_sym.y)));
                integ = fmin(fmax(DA(integ, DM(ki, e)), -maxdev), maxdev);
                phase = DA(phase, DA(DM(kp, e), integ));
                if ((size_t)count < cap) {
                    if (lane == 0) {
                        symbols[(size_t)c * cap + count] = make_double2(sr, si);
                        decisions[(size_t)c * cap + count] = make_double2(dr, di);
                        errors[(size_t)c * cap + count] = e;
                    }
                    count++;
                }
                prev_sym = make_double2(sr, si);
                prev_dec = make_double2(dr, di);
            }
        }
    }
    if (lane == 0) {
        sp->phase = phase; sp->integ = integ;
        sp->b0 = b0; sp->b1 = b1; sp->b2 = b2; sp->b3 = b3; sp->prev_sym = prev_sym; sp->prev_dec = prev_dec;
        counts[c] = count;
    }
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

// symbols a call of n samples can produce: the Mueller-Muller period is clamped to [sps / 2, ...] (integrator within
// +-sps / 4, symbol_timing.py:330-345), bound computed in floating point (sps / 2 need not be an integer)
static size_t mm_symbol_bound(size_t n, double sps) { return (size_t)((double)n / (sps * 0.5)) + 2; }

struct wh_cqpsk_bank {
    int C, n_max, L;
    double sps, c_kp, c_ki, c_maxf, t_kp, t_ki;
    std::vector<double> zi0;
    double *d_taps = nullptr;
    double2 *d_state[2] = {nullptr, nullptr}, *d_full = nullptr, *d_symws = nullptr;
    size_t symcap = 0;
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
    if (!out || !h_rrc || !h_zi || C < 1 || ntaps < 2 || ntaps > 4096 || !(sps >= 2.0) || n_max < 1)
        return set_err(WH_E_ARG, "wh_cqpsk_bank_create: bad arguments (samples_per_symbol >= 2)");
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
    b->symcap = mm_symbol_bound((size_t)n_max, sps);
    WH_HIP(hipMalloc(&b->d_symws, (size_t)C * b->symcap * sizeof(double2)));
    WH_HIP(hipMalloc(&b->d_st, (size_t)C * sizeof(CqState)));
    int rc = cq_reset(b, nullptr);
    if (rc != WH_OK) return rc;
    *out = guard.release();
    return WH_OK;
}

extern "C" void wh_cqpsk_bank_destroy(wh_cqpsk_bank *b) {
    if (!b) return;
    (void)hipFree(b->d_taps); (void)hipFree(b->d_state[0]); (void)hipFree(b->d_state[1]); (void)hipFree(b->d_full);
    (void)hipFree(b->d_symws);
    (void)hipFree(b->d_st);
    delete b;
}

// symbols a call of n samples can produce: the Mueller-Muller period is clamped to [sps / 2, ...] (integrator within
// +-sps / 4, symbol_timing.py:330-345), bound computed in floating point (sps / 2 need not be an integer)

// Grow the matched-filter workspace for calls of up to n_max samples per channel (never shrinks; no state lives in it).
// Allocates and synchronises the stream: not for the hot path.
extern "C" int wh_cqpsk_bank_reserve(wh_cqpsk_bank *b, size_t n_max, void *stream) {
    if (!b) return set_err(WH_E_ARG, "wh_cqpsk_bank_reserve: null handle");
    if (n_max > (size_t)1 << 26) return set_err(WH_E_ARG, "wh_cqpsk_bank_reserve: more than 2^26 samples per call");
    if (n_max <= (size_t)b->n_max) return WH_OK;
    double2 *full = nullptr, *symws = nullptr;
    const size_t symcap = mm_symbol_bound(n_max, b->sps);
    WH_HIP(hipMalloc(&full, (size_t)b->C * (n_max + b->L - 1) * sizeof(double2)));
    hipError_t e = hipMalloc(&symws, (size_t)b->C * symcap * sizeof(double2));
    (void)stream;
    if (e == hipSuccess) e = hipDeviceSynchronize();   // earlier calls (on whichever stream) may still read the old workspaces
    if (e != hipSuccess) { (void)hipFree(full); (void)hipFree(symws); return set_err(WH_E_HIP, "wh_cqpsk_bank_reserve: %s", hipGetErrorString(e)); }
    (void)hipFree(b->d_full);
    (void)hipFree(b->d_symws);
    b->d_full = full;
    b->d_symws = symws;
    b->symcap = symcap;
    b->n_max = (int)n_max;
    return WH_OK;
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
    if (cap < mm_symbol_bound(n, b->sps)) return set_err(WH_E_ARG, "wh_cqpsk_bank_run: cap too small");
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
    a.symws = b->d_symws; a.symcap = b->symcap;
    a.cap = cap;
    a.counts = d_counts;
    const int H = b->L - 1;
    hipLaunchKernelGGL(k_cq_rrc, dim3((unsigned)((n + H + 255) / 256), b->C), dim3(256), b->L * sizeof(double), st, a);
    WH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cq_seq, dim3((unsigned)b->C), dim3(128), 0, st, a);   // two waves per channel (carrier | timing)
    WH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cq_decode, dim3((unsigned)b->C), dim3(256), 0, st, a);
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

// ---- CostasLoop / MuellerMullerTED banks ------------------------------------------------------------------------------
struct wh_costas_bank {
    int C;
    double kp, ki, maxf;
    CostasState *d_st = nullptr;
};

extern "C" int wh_costas_bank_create(wh_costas_bank **out, int C, double kp, double ki, double max_freq) {
    if (!out || C < 1 || !(max_freq >= 0.0)) return set_err(WH_E_ARG, "wh_costas_bank_create: bad arguments");
    wh_costas_bank *b = new wh_costas_bank();
    std::unique_ptr<wh_costas_bank, void (*)(wh_costas_bank *)> guard(b, wh_costas_bank_destroy);
    b->C = C; b->kp = kp; b->ki = ki; b->maxf = max_freq;
    WH_HIP(hipMalloc(&b->d_st, (size_t)C * sizeof(CostasState)));
    WH_HIP(hipMemset(b->d_st, 0, (size_t)C * sizeof(CostasState)));
    *out = guard.release();
    return WH_OK;
}

extern "C" void wh_costas_bank_destroy(wh_costas_bank *b) {
    if (!b) return;
    (void)hipFree(b->d_st);
    delete b;
}

extern "C" int wh_costas_bank_reset(wh_costas_bank *b, void *stream) {
    if (!b) return set_err(WH_E_ARG, "wh_costas_bank_reset: null handle");
    WH_HIP(hipMemsetAsync(b->d_st, 0, (size_t)b->C * sizeof(CostasState), as_stream(stream)));
    return WH_OK;
}

extern "C" int wh_costas_bank_run(wh_costas_bank *b, const double *d_x, size_t n, size_t stride, double *d_out,
                                  double *h_freq, void *stream) {
    if (!b) return set_err(WH_E_ARG, "wh_costas_bank_run: null handle");
    hipStream_t st = as_stream(stream);
    if (n > 0) {
        if (!d_x || !d_out || stride < n || n > 0x7fffffff) return set_err(WH_E_ARG, "wh_costas_bank_run: bad buffers");
        hipLaunchKernelGGL(k_costas, dim3((unsigned)b->C), dim3(64), 0, st, reinterpret_cast<const double2 *>(d_x), stride,
                           (int)n, b->d_st, b->kp, b->ki, b->maxf, reinterpret_cast<double2 *>(d_out), stride);
        WH_LAUNCH_CHECK();
    }
    if (h_freq) {   // frequency_offset property: synchronous read-back of the loop integrators
        std::vector<CostasState> h((size_t)b->C);
        WH_HIP(hipMemcpyAsync(h.data(), b->d_st, h.size() * sizeof(CostasState), hipMemcpyDeviceToHost, st));
        WH_HIP(hipStreamSynchronize(st));
        for (int c = 0; c < b->C; ++c) h_freq[c] = h[c].freq;
    }
    return WH_OK;
}

struct wh_mm_bank {
    int C;
    double sps, kp, ki;
    MmState *d_st = nullptr;
};

extern "C" int wh_mm_bank_create(wh_mm_bank **out, int C, double sps, double kp, double ki) {
    if (!out || C < 1 || !(sps >= 2.0)) return set_err(WH_E_ARG, "wh_mm_bank_create: bad arguments (samples_per_symbol >= 2)");
    wh_mm_bank *b = new wh_mm_bank();
    std::unique_ptr<wh_mm_bank, void (*)(wh_mm_bank *)> guard(b, wh_mm_bank_destroy);
    b->C = C; b->sps = sps; b->kp = kp; b->ki = ki;
    WH_HIP(hipMalloc(&b->d_st, (size_t)C * sizeof(MmState)));
    WH_HIP(hipMemset(b->d_st, 0, (size_t)C * sizeof(MmState)));
    *out = guard.release();
    return WH_OK;
}

extern "C" void wh_mm_bank_destroy(wh_mm_bank *b) {
    if (!b) return;
    (void)hipFree(b->d_st);
    delete b;
}

extern "C" int wh_mm_bank_reset(wh_mm_bank *b, void *stream) {
    if (!b) return set_err(WH_E_ARG, "wh_mm_bank_reset: null handle");
    WH_HIP(hipMemsetAsync(b->d_st, 0, (size_t)b->C * sizeof(MmState), as_stream(stream)));
    return WH_OK;
}

extern "C" int wh_mm_bank_run(wh_mm_bank *b, const double *d_x, size_t n, size_t stride, double *d_symbols,
                              double *d_decisions, double *d_errors, size_t cap, int32_t *d_counts, void *stream) {
    if (!b || !d_counts) return set_err(WH_E_ARG, "wh_mm_bank_run: null handle/counts");
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        WH_HIP(hipMemsetAsync(d_counts, 0, (size_t)b->C * sizeof(int32_t), st));
        return WH_OK;
    }
    if (!d_x || !d_symbols || !d_decisions || !d_errors || stride < n || n > 0x7fffffff)
        return set_err(WH_E_ARG, "wh_mm_bank_run: bad buffers");
    if (cap < mm_symbol_bound(n, b->sps)) return set_err(WH_E_ARG, "wh_mm_bank_run: cap too small");
    hipLaunchKernelGGL(k_mm, dim3((unsigned)b->C), dim3(64), 0, st, reinterpret_cast<const double2 *>(d_x), stride, (int)n,
                       b->d_st, b->sps, b->kp, b->ki, reinterpret_cast<double2 *>(d_symbols),
                       reinterpret_cast<double2 *>(d_decisions), d_errors, cap, d_counts);
    WH_LAUNCH_CHECK();
    return WH_OK;
}
