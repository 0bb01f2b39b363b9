// Analog per-channel chain for gfx950: int16 unpack/pack (A1), stateless NCO (A2), FM
// discriminator (A3), rational resampler (A6) and the batched channel operator ("fmbank",
// reference capture.py:298-439 for modes nbfm / wbfm).
//
// fmbank fused kernel (up == 1, nbfm): one workgroup = (chunk, channel, tile of TO outputs).
//   phase 1  NCO mix + discriminator for the tile's span of input samples, straight from the
//            shared int16 / complex64 chunk (coalesced), two samples per lane in packed arithmetic, the
//            neighbour sample x[n-1] of a lane's first sample taken from the previous lane with a wavefront
//            shuffle (each wave advances 127 samples per step, loads one step ahead), fm kept in LDS;
//   phase 2  the decimating FIR (resample_poly taps, float64 accumulation like scipy) computed
//            only at the kept outputs, from LDS;
//   the RMS normalisation commutes with the FIR, so its scale (which needs the whole chunk) and
//   the tanh soft clip are applied by a tiny finalize kernel that also produces the metrics.
// HBM traffic: the input chunk (4 or 8 B per sample, shared by all channels through L2) and
// n_out floats per channel -- no intermediate fm array.
#include "wh_common.h"
#include "wh_portable_math.h"
#include <cmath>
#include <type_traits>
#include <memory>
#include <vector>

using namespace wh;

namespace {

__device__ __forceinline__ float2 load_iq(const void *in, int fmt, size_t idx) {
    if (fmt == 1) {
        short2 v = reinterpret_cast<const short2 *>(in)[idx];
        // A1 unpack rule: int16 -> float32 / 32768.0 (exact)
        return make_float2((float)v.x * (1.0f / 32768.0f), (float)v.y * (1.0f / 32768.0f));
    }
    return reinterpret_cast<const float2 *>(in)[idx];
}

// base[n] = x[n] * exp(i * f32(c) * f32(n))   (capture.py:166-193; c already negative)
__device__ __forceinline__ float2 mix(float2 x, float c, int n, bool do_mix) {
    if (!do_mix) return x;
    float ph = __fmul_rn(c, (float)n);
    float s, co;
    whm_sincos_phase(ph, &s, &co);
    return make_float2(x.x * co - x.y * s, x.x * s + x.y * co);
}

// Same mix for the fused FM bank (tolerance 1e-5, not bit-exact): the float32 phase product is the reference's,
// its reduction to revolutions uses a two-float product (exact to ~1e-10 rev), then the hardware sin / cos,
// which take revolutions.  ~13 issue slots instead of ~40.
__device__ __forceinline__ float2 mix_fast(float2 x, float c, int n) {
    const float C_HI = 0.15915494f;                 // float32(1 / 2 pi)
    const float C_LO = 6.4206382e-09f;              // 1 / 2 pi - C_HI
    float ph = __fmul_rn(c, (float)n);
    float t_hi = __fmul_rn(ph, C_HI);
    float t_lo = fmaf(ph, C_LO, fmaf(ph, C_HI, -t_hi));
    float fr = __builtin_amdgcn_fractf(t_hi) + t_lo;
    float s = __builtin_amdgcn_sinf(fr), co = __builtin_amdgcn_cosf(fr);
    return make_float2(fmaf(x.x, co, -(x.y * s)), fmaf(x.x, s, x.y * co));
}

// atan2f for the fused FM bank (tolerance 1e-5, no bit-exactness requirement): one v_rcp quotient in [0, 1] and a
// single-range odd polynomial (degree 15, near-minimax fit of atan(t)/t in t^2, max error 1.7e-7 rad in float32),
// then the octant fix-ups -- ~25 issue slots instead of ~38 for the two-range cephes form.
__device__ __forceinline__ float fast_atan2f(float y, float x) {
    const float PI_F = 3.14159265358979323846f, PIO2_F = 1.57079632679489661923f;
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(mx + 1e-37f);   // (0, 0) -> 0
    const float z = t * t;
    float p = -0.004668773151934147f;
    p = fmaf(p, z, 0.02416618913412094f);
    p = fmaf(p, z, -0.0593671016395092f);
    p = fmaf(p, z, 0.09906096756458282f);
    p = fmaf(p, z, -0.14016585052013397f);
    p = fmaf(p, z, 0.19969235360622406f);
    p = fmaf(p, z, -0.33331960439682007f);
    p = fmaf(p, z, 0.9999998807907104f);
    float r = p * t;
    r = ay > ax ? PIO2_F - r : r;
    r = x < 0.0f ? PI_F - r : r;
    return __builtin_copysignf(r, y);
}

__global__ void unpack_kernel(const short2 *in, float2 *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        short2 v = in[i];
        out[i] = make_float2((float)v.x * (1.0f / 32768.0f), (float)v.y * (1.0f / 32768.0f));
    }
}

__device__ __forceinline__ short pack1(float f) {
    // np.clip(-1, 1) then * 32767.0 then C cast (truncate toward zero); NaN -> clip keeps NaN,
    // numpy's x86 cast of NaN (cvttss2si -> 0x80000000, low 16 bits) gives 0: mirror it.
    if (f != f) return (short)0;
    f = fminf(fmaxf(f, -1.0f), 1.0f);
    return (short)(int)(__fmul_rn(f, 32767.0f));
}

__global__ void pack_iq_kernel(const float2 *in, short2 *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float2 v = in[i];
        short2 o;
        o.x = pack1(v.x);
        o.y = pack1(v.y);
        out[i] = o;
    }
}

__global__ void pack_pcm_kernel(const float *in, short *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = pack1(in[i]);
}

__global__ void nco_kernel(const float2 *in, float2 *out, size_t n, float c) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = mix(in[i], c, (int)i, true);
}

__global__ void disc_kernel(const float2 *in, float *out, size_t n, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float v = 0.f;
        if (i > 0) {
            float2 a = in[i], b = in[i - 1];
            float re = a.x * b.x + a.y * b.y;
            float im = a.y * b.x - a.x * b.y;
            v = whm_atan2f(im, re) * scale;
        }
        out[i] = v;
    }
}

// ---- generic rational resampler: one wave per output ---------------------------------------
__global__ __launch_bounds__(256) void resample_kernel(const float *x, size_t n_in, size_t x_stride, float *y, size_t n_out,
                                                       const double *h, int ntaps, int up, int down, int d0) {
    const int lane = threadIdx.x & 63;
    const size_t m = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t b = blockIdx.y;
    if (m >= n_out) return;
    const float *xb = x + b * x_stride;
    long long base = (long long)m * down + d0;
    int j0 = (int)(base % up);
    long long q0 = (base - j0) / up;
    int cnt = (ntaps - 1 - j0) / up + 1;  // taps j0 + up*i < ntaps
    if (j0 > ntaps - 1) cnt = 0;
    double acc = 0.0;
    for (int i = lane; i < cnt; i += 64) {
        long long q = q0 - i;
        if (q >= 0 && q < (long long)n_in) acc = fma((double)xb[q], h[j0 + up * i], acc);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) y[b * n_out + m] = (float)acc;
}

// ---- channel bank -----------------------------------------------------------------------------
constexpr int MAX_STAGES = 8;
constexpr int MAX_ORD = 11;   // coefficients per stage (butter(5) band-pass = order 10)

struct StageDev {
    int is_f64, n;
    double b[MAX_ORD], a[MAX_ORD];
};

struct FmArgs {
    const void *in;     // [n_chunks][N] int16x2 or float2
    float *audio;       // [n_chunks][K][n_out]   (unscaled FIR output until finalize)
    double *acc;        // [n_chunks][K][2] = {sum |base|^2, sum v^2}
    float *fm_out;      // unfused path: [n_chunks][K][N] demodulated rows
    const float *rows_src;  // fmbank_fused_kernel as the unfused path's decimating FIR: the window comes from these rows
                            // ([n_chunks][K][N], already demodulated and filtered) instead of the discriminator
    int skip_fm_sum;    // unfused fronts: leave acc[..][1] at zero (the rows kernel that follows owns it)
    const float *nco_c; // [K] f32(-2 pi off/fs); 0 => no mix
    const double *taps; // [ntaps]
    int fmt, N, K, n_out, ntaps, down, d0, TO, R;   // R: output tiles per workgroup (fused kernel)
    const float *araw;      // fused kernel, fast front: [n_chunks][N] raw discriminator angles (fm_rawdisc_kernel); NaN where the
                            // sample pair's product is exactly zero; nullptr = the per-channel front of round 2
    const float2 *tphase;   // fir_phase: float2 [21][32] = the tap pairs (trev[q down + 2 rp], trev[.. + 1]) by block and phase pair
    int fir_phase;      // fused kernel: != 0 -> the decimating FIR runs in its polyphase register form (fm_fir_phases), TO = 128
    float scale;        // fs / (2 pi 75000)
    int demod;          // 0 FM discriminator, 1 AM envelope, 2 SSB product detector, 3/4/5 SAM dsb/usb/lsb
    double bfo_c;       // 2 pi bfo_hz (SSB), sample_rate in fs_d
    double fs_d;
    double pll_alpha, pll_beta;   // SAM carrier-recovery PLL loop filter (dsp/sam.py:55-66)
};

constexpr int FM_MAX_SPAN = 8192;    // floats of fm kept in LDS (32 KiB -> 4 workgroups per CU)
constexpr int FM_MAX_TAPS = 2048;    // float64 taps kept in LDS (16 KiB)

// One workgroup walks a RUN of a.R consecutive output tiles of one (chunk, channel) row.  The discriminator
// output lives in LDS as a window of ntaps + (TO-1)*down samples; after a tile's FIR the window slides by
// TO*down: its last ntaps-down samples are moved to the front and only the TO*down new ones are computed, so
// every input sample is mixed / discriminated once per run instead of once per tile that overlaps it.
typedef float fm_v2f __attribute__((ext_vector_type(2)));

// packed pair of fast_atan2f: same operations per component (the multiplies / FMAs go out as v_pk_* instructions)
__device__ __forceinline__ fm_v2f fast_atan2f_x2(fm_v2f y, fm_v2f x) {
    const float PI_F = 3.14159265358979323846f, PIO2_F = 1.57079632679489661923f;
    const fm_v2f ax = {fabsf(x.x), fabsf(x.y)}, ay = {fabsf(y.x), fabsf(y.y)};
    const fm_v2f mx = {fmaxf(ax.x, ay.x), fmaxf(ax.y, ay.y)}, mn = {fminf(ax.x, ay.x), fminf(ax.y, ay.y)};
    const fm_v2f d = mx + fm_v2f{1e-37f, 1e-37f};                 // (0, 0) -> 0
    const fm_v2f t = mn * fm_v2f{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const fm_v2f z = t * t;
#define WH_BC(v) (fm_v2f{(v), (v)})
    fm_v2f p = WH_BC(-0.004668773151934147f);
    p = __builtin_elementwise_fma(p, z, WH_BC(0.02416618913412094f));
    p = __builtin_elementwise_fma(p, z, WH_BC(-0.0593671016395092f));
    p = __builtin_elementwise_fma(p, z, WH_BC(0.09906096756458282f));
    p = __builtin_elementwise_fma(p, z, WH_BC(-0.14016585052013397f));
    p = __builtin_elementwise_fma(p, z, WH_BC(0.19969235360622406f));
    p = __builtin_elementwise_fma(p, z, WH_BC(-0.33331960439682007f));
    p = __builtin_elementwise_fma(p, z, WH_BC(0.9999998807907104f));
#undef WH_BC
    fm_v2f r = p * t;
    r.x = ay.x > ax.x ? PIO2_F - r.x : r.x;
    r.y = ay.y > ax.y ? PIO2_F - r.y : r.y;
    r.x = x.x < 0.0f ? PI_F - r.x : r.x;
    r.y = x.y < 0.0f ? PI_F - r.y : r.y;
    return fm_v2f{__builtin_copysignf(r.x, y.x), __builtin_copysignf(r.y, y.y)};
}

// A sample whose conjugate product with its predecessor is EXACTLY zero (a zero sample: muted input, overflow recovery
// returning zero buffers): the reference's discriminator value then hangs on the SIGNS of zeros -- numpy mixes with
// (xr c - xi s, xr s + xi c), multiplies by the conjugate as (ar br - ai bi', ar bi' + ai br) with bi' = -bi, single
// roundings, and np.angle is arctan2, which maps (+-0, -0) to +-pi: a muted stretch comes out as 0 or +-pi x scale
// depending on the quadrant of the mixer phase.  Reproduced here operation for operation (the magnitudes are zero; only
// the IEEE sign rules act).  Rare and per sample, so the value never depends on the pass that computes it.
template <int FMT, bool MIX>
__device__ __forceinline__ float fm_single_zero(const FmArgs &a, size_t in_base, int n, float c) {
    const float PI_F = 3.14159265358979323846f;
    const float2 xp = load_iq(a.in, FMT, in_base + n - 1), xc = load_iq(a.in, FMT, in_base + n);
    float2 bp = xp, bc = xc;
    if (MIX) {
        const fm_v2f C_HI = {0.15915494f, 0.15915494f}, C_LO = {6.4206382e-09f, 6.4206382e-09f};
        const fm_v2f ph = fm_v2f{c, c} * fm_v2f{(float)(n - 1), (float)n};
        const fm_v2f t_hi = ph * C_HI;
        const fm_v2f t_lo = __builtin_elementwise_fma(ph, C_LO, __builtin_elementwise_fma(ph, C_HI, -t_hi));
        const fm_v2f fr = fm_v2f{__builtin_amdgcn_fractf(t_hi.x), __builtin_amdgcn_fractf(t_hi.y)} + t_lo;
        const float sp = __builtin_amdgcn_sinf(fr.x), cp = __builtin_amdgcn_cosf(fr.x);
        const float sc = __builtin_amdgcn_sinf(fr.y), cc = __builtin_amdgcn_cosf(fr.y);
        bp = make_float2(__fsub_rn(__fmul_rn(xp.x, cp), __fmul_rn(xp.y, sp)), __fadd_rn(__fmul_rn(xp.x, sp), __fmul_rn(xp.y, cp)));
        bc = make_float2(__fsub_rn(__fmul_rn(xc.x, cc), __fmul_rn(xc.y, sc)), __fadd_rn(__fmul_rn(xc.x, sc), __fmul_rn(xc.y, cc)));
    }
    const float nbi = -bp.y;
    const float re = __fsub_rn(__fmul_rn(bc.x, bp.x), __fmul_rn(bc.y, nbi));
    const float im = __fadd_rn(__fmul_rn(bc.x, nbi), __fmul_rn(bc.y, bp.x));
    if (re == 0.0f && im == 0.0f) return __builtin_signbitf(re) ? __builtin_copysignf(PI_F, im) : __builtin_copysignf(0.0f, im);
    return fast_atan2f_x2(fm_v2f{im, im}, fm_v2f{re, re}).x;
}

// Phase 1 of the fused kernel, TWO samples per lane: lane l of a wave takes the window indices s0 + 2l - 1 and s0 + 2l (a
// wave advances 127 samples per step; lane 0's first sample only supplies x[n-1]).  The previous sample of the second one
// is the lane's own first, that of the first one the neighbour's second (one shuffle per two samples), both samples of
// int16 input come in with one 8-byte load, and the NCO / mix / discriminator / atan2 arithmetic of the pair goes out as
// packed instructions.  Per sample the operations are those of the one-sample form (same results for a sample whichever
// slot computes it): 87 -> about 65 issue slots per sample and channel.
// Power sums (rms_normalize needs sum v^2 of the whole row, the metrics sum |base|^2): float32 PER PASS AND LANE -- a
// lane meets at most ~30 samples in a pass -- folded into the float64 run sums once per pass.  A pass's step grid is
// anchored at the pass's first window index, and which pass computes a sample does not depend on how the row is cut
// into runs (see the ownership rule in the kernel), so the sums stay independent of the batch size bit for bit.
// BASE: sum |x|^2 of the RAW samples -- |x e^(j theta)| = |x|, the reference's rssi_db is the power of the mixed but
// unfiltered chunk (capture.py:330-334), i.e. the same number for every channel of a capture up to the rounding of
// |e^(j theta)| (1e-7) -- so only channel 0's workgroups take it and the finalize kernel shares it.
template <bool CHECK, int FMT, bool MIX>
__device__ __forceinline__ void fm_phase1(const FmArgs &a, float *fm_s, int i_lo, int i_hi, int n_lo, int N, float c,
                                          size_t in_base, int own_lo, int own_hi, int lane, int wave,
                                          double &p_base, double &p_fm, const bool base) {
    float sb_a = 0.f, sb_b = 0.f, sf_a = 0.f, sf_b = 0.f;   // the pass's float32 sums: a-slots and b-slots apart
    // interior passes (no bounds checks) load one step ahead: the pair of the NEXT step is in flight while this one is
    // mixed and discriminated (the last step re-reads its own pair)
    typedef typename std::conditional<FMT == 1, short4, float4>::type raw_t;   // two IQ pairs, 4-byte aligned
    const char *in_b = reinterpret_cast<const char *>(a.in) + (in_base + (size_t)n_lo) * (FMT == 1 ? 4 : 8);
    raw_t q_next;
    if (!CHECK && i_lo + wave * 127 < i_hi)
        __builtin_memcpy(&q_next, in_b + (long long)(i_lo + wave * 127 + 2 * lane - 1) * (FMT == 1 ? 4 : 8), sizeof(q_next));
    for (int s0 = i_lo + wave * 127; s0 < i_hi; s0 += 4 * 127) {
        const int ia = s0 + 2 * lane - 1, ib = ia + 1;   // window indices
        const int na = n_lo + ia, nb = na + 1;           // chunk sample indices
        const bool va = !CHECK || ((na >= 0) && (na < N)), vb = !CHECK || ((nb >= 0) && (nb < N));
        fm_v2f bx = {0.f, 0.f}, by = {0.f, 0.f};         // (re of a, re of b), (im of a, im of b)
        if (!CHECK) {
            const raw_t q = q_next;
            const int sn = s0 + 4 * 127 < i_hi ? s0 + 4 * 127 : s0;
            __builtin_memcpy(&q_next, in_b + (long long)(sn + 2 * lane - 1) * (FMT == 1 ? 4 : 8), sizeof(q_next));
            if (FMT == 1) {
                bx = fm_v2f{(float)q.x, (float)q.z} * fm_v2f{1.0f / 32768.0f, 1.0f / 32768.0f};   // A1 unpack rule (exact)
                by = fm_v2f{(float)q.y, (float)q.w} * fm_v2f{1.0f / 32768.0f, 1.0f / 32768.0f};
            } else {
                bx = fm_v2f{(float)q.x, (float)q.z};
                by = fm_v2f{(float)q.y, (float)q.w};
            }
        } else {
            if (va) { const float2 v = load_iq(a.in, FMT, in_base + na); bx.x = v.x; by.x = v.y; }
            if (vb) { const float2 v = load_iq(a.in, FMT, in_base + nb); bx.y = v.x; by.y = v.y; }
        }
        fm_v2f pw = {0.f, 0.f};
        if (base) pw = __builtin_elementwise_fma(bx, bx, by * by);   // raw power (uniform branch: channel 0 only)
        if (MIX) {   // mix_fast on the pair
            const fm_v2f C_HI = {0.15915494f, 0.15915494f}, C_LO = {6.4206382e-09f, 6.4206382e-09f};
            const fm_v2f ph = fm_v2f{c, c} * fm_v2f{(float)na, (float)nb};
            const fm_v2f t_hi = ph * C_HI;
            const fm_v2f t_lo = __builtin_elementwise_fma(ph, C_LO, __builtin_elementwise_fma(ph, C_HI, -t_hi));
            const fm_v2f fr = fm_v2f{__builtin_amdgcn_fractf(t_hi.x), __builtin_amdgcn_fractf(t_hi.y)} + t_lo;
            const fm_v2f sn = {__builtin_amdgcn_sinf(fr.x), __builtin_amdgcn_sinf(fr.y)};
            const fm_v2f co = {__builtin_amdgcn_cosf(fr.x), __builtin_amdgcn_cosf(fr.y)};
            const fm_v2f mr = __builtin_elementwise_fma(bx, co, -(by * sn));
            const fm_v2f mi = __builtin_elementwise_fma(bx, sn, by * co);
            // (an invalid sample is (0, 0) before and after the rotation)
            bx = mr;
            by = mi;
        }
        // previous samples: of a the neighbour lane's b, of b the lane's own a
        const fm_v2f px = {__shfl_up(bx.y, 1), bx.x}, py = {__shfl_up(by.y, 1), by.x};
        const fm_v2f re = __builtin_elementwise_fma(bx, px, by * py);
        const fm_v2f im = __builtin_elementwise_fma(by, px, -(bx * py));
        fm_v2f v = fast_atan2f_x2(im, re) * fm_v2f{a.scale, a.scale};
        // exactly-zero products: the reference's signed-zero result (fm_single_zero), per sample
        if (lane > 0 && re.x == 0.f && im.x == 0.f && va && na >= 1) v.x = fm_single_zero<FMT, MIX>(a, in_base, na, c) * a.scale;
        if (re.y == 0.f && im.y == 0.f && vb && nb >= 1) v.y = fm_single_zero<FMT, MIX>(a, in_base, nb, c) * a.scale;
        if (CHECK) {
            if (!(va && na >= 1)) v.x = 0.f;
            if (!(vb && nb >= 1)) v.y = 0.f;
        }
        if (lane > 0 && ia < i_hi) {
            fm_s[ia] = v.x;
            if (!CHECK || (va && na >= own_lo && na < own_hi)) {
                sb_a += pw.x;
                sf_a = fmaf(v.x, v.x, sf_a);
            }
        }
        if (ib < i_hi) {
            fm_s[ib] = v.y;
            if (!CHECK || (vb && nb >= own_lo && nb < own_hi)) {
                sb_b += pw.y;
                sf_b = fmaf(v.y, v.y, sf_b);
            }
        }
    }
    p_base += (double)(sb_a + sb_b);
    p_fm += (double)(sf_a + sf_b);
}

// ---- the FM front without per-channel transcendentals ---------------------------------------------------------------
// The discriminator of a MIXED channel is the discriminator of the RAW stream plus the mixer's phase step:
//     angle(b[n] conj b[n-1]),  b[n] = x[n] e^{j phi_n}   =   wrap( angle(x[n] conj x[n-1]) + (phi_n - phi_{n-1}) )
// exactly, and the reference's mixer phase is phi_n = float32(c) * float32(n) (capture.py:173-176), whose consecutive
// values differ by an exactly representable float32 (Sterbenz).  So the raw angles a[n] are taken ONCE per chunk
// (fm_rawdisc_kernel: the discriminator arithmetic of fm_phase1 on the unmixed samples) and a channel's front is
//     v = scale * wrap(a[n] + (fl(c n) - fl(c (n - 1))))
// -- two multiplies, a subtraction, an addition and a wrap instead of sincos, mixing, the conjugate product and an
// arctangent per sample AND channel (~90 -> ~25 issue slots per pair).  Against the reference this moves the rounding of
// the mixed products into the rounding of a[n] (both ~1e-7 rad; audio of the NBFM goldens: 1.1e-7 .. 2.4e-7 of peak by
// this route in numpy, 0.6e-7 .. 1.5e-7 by the reference's own order).  Where a sample pair's product is EXACTLY zero
// (a zero sample: the reference's result then hangs on signed zeros of the mixed values) a[n] is NaN and that sample
// alone is evaluated by fm_single_zero.
template <int FMT>
__global__ __launch_bounds__(256) void fm_rawdisc_kernel(FmArgs a, float *araw) {
    const int chunk = blockIdx.y, tid = threadIdx.x;
    const int N = a.N;
    const size_t in_base = (size_t)chunk * N;
    float pw = 0.f;
    for (int n0 = blockIdx.x * 1024; n0 < N; n0 += gridDim.x * 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = n0 + u * 256 + tid;
            if (n >= N) continue;
            const float2 x = load_iq(a.in, FMT, in_base + n);
            float v = 0.f;                                     // dsp/fm.py:94: out[0] = 0
            if (n > 0) {
                const float2 p = load_iq(a.in, FMT, in_base + n - 1);
                const float re = fmaf(x.x, p.x, x.y * p.y), im = fmaf(x.y, p.x, -(x.x * p.y));   // fm_phase1's products
                const fm_v2f r = fast_atan2f_x2(fm_v2f{im, im}, fm_v2f{re, re});
                v = (re == 0.0f && im == 0.0f) ? __int_as_float(0x7fc00000) : r.x;
            }
            araw[in_base + n] = v;
            pw += fmaf(x.x, x.x, x.y * x.y);
        }
    }
    // sum |x|^2 of the chunk -> channel 0's slot (shared by the chunk's channels, see fmbank_finalize_kernel)
    __shared__ double red[4];
    double d = (double)pw;
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if ((tid & 63) == 0) red[tid >> 6] = d;
    __syncthreads();
    if (tid == 0) atomicAdd(a.acc + (size_t)chunk * a.K * 2, red[0] + red[1] + red[2] + red[3]);
}

// the fast front over window indices [i_lo, i_hi): lane l takes the pair i_lo + 128 (wave + 4 k) + 2 l, + 1; the float32
// sums of v^2 as in fm_phase1.  CHECK: passes at the chunk's edges (samples outside [0, N) are zeros, sample 0 has no
// predecessor: v = 0, dsp/fm.py:94) and ownership masks for the sums -- the SAME formula as the interior passes, so that a
// sample's value never depends on which pass computes it.
template <bool CHECK, int FMT, bool MIX>
__device__ __forceinline__ void fm_phase1_fast(const FmArgs &a, float *fm_s, int i_lo, int i_hi, int n_lo, int N, float c,
                                               size_t in_base, int own_lo, int own_hi, int lane, int wave, double &p_fm,
                                               const bool sums) {
    const float *ar = a.araw + in_base;
    const float INV2PI = 0.15915494309189535f, TWOPI = 6.283185307179586f;
    float sf_a = 0.f, sf_b = 0.f;
    for (int s0 = i_lo + wave * 128; s0 < i_hi; s0 += 4 * 128) {
        const int ia = s0 + 2 * lane, ib = ia + 1;
        const int na = n_lo + ia, nb = na + 1;
        const bool oka = ia < i_hi, okb = ib < i_hi;
        const bool va = !CHECK || (na >= 1 && na < N), vb = !CHECK || (nb >= 1 && nb < N);
        int la = oka ? na : n_lo + i_hi - 1, lb = okb ? nb : n_lo + i_hi - 1;
        if (CHECK) {
            la = la < 0 ? 0 : (la > N - 1 ? N - 1 : la);
            lb = lb < 0 ? 0 : (lb > N - 1 ? N - 1 : lb);
        }
        const float aa = ar[la], ab = ar[lb];
        fm_v2f s = {aa, ab};
        if (MIX) {
            // phi_n - phi_{n-1} with phi_n = fl(c n): single roundings, no contraction (an FMA here would be the exact c)
            const float naf = (float)na;
            const float pm = __fmul_rn(c, naf - 1.0f), pa = __fmul_rn(c, naf), pb = __fmul_rn(c, naf + 1.0f);
            s = s + fm_v2f{__fsub_rn(pa, pm), __fsub_rn(pb, pa)};
        }
        const fm_v2f t = s * fm_v2f{INV2PI, INV2PI};
        const fm_v2f r = {__builtin_rintf(t.x), __builtin_rintf(t.y)};
        s = __builtin_elementwise_fma(r, fm_v2f{-TWOPI, -TWOPI}, s);          // into [-pi, pi]
        fm_v2f v = s * fm_v2f{a.scale, a.scale};
        // a sample whose raw product is exactly zero (NaN marker): the reference's signed-zero result for that sample alone
        if (oka && va && aa != aa) v.x = fm_single_zero<FMT, MIX>(a, in_base, na, c) * a.scale;
        if (okb && vb && ab != ab) v.y = fm_single_zero<FMT, MIX>(a, in_base, nb, c) * a.scale;
        if (CHECK) {
            if (!va) v.x = 0.f;
            if (!vb) v.y = 0.f;
        }
        if (oka) {
            fm_s[ia] = v.x;
            if (!CHECK || (na >= own_lo && na < own_hi)) sf_a = fmaf(v.x, v.x, sf_a);
        }
        if (okb) {
            fm_s[ib] = v.y;
            if (!CHECK || (nb >= own_lo && nb < own_hi)) sf_b = fmaf(v.y, v.y, sf_b);
        }
    }
    if (sums) p_fm += (double)(sf_a + sf_b);
}

// ---- phase 2 in polyphase register form ---------------------------------------------------------------------------
// y[o] = sum_i w[o D + i] trev[i] (D = down) read the window from LDS once per MAC pair and tap pair: 5 bytes per MAC,
// 77 GB of LDS reads per 32 x 200 x 120 000 launch = 1.1 ms at the chip's 69 TB/s of LDS bandwidth -- the launch was
// bound by that, not by its arithmetic (cutting 16 % of the VALU work of phase 1 did not move it).  Here the sum is
// split by tap PHASE: i = q D + 2 rp + c (q < 21 blocks, rp < D / 2 phase pairs, c < 2), lane (e, rp) of a wave owns
// phase pair rp for the outputs o = o0 + 2 j + e of the wave's quarter of the tile, keeps its 21 tap pairs in registers
// for the whole kernel and the 21 window pairs w[(o + q) D + 2 rp .. + 1] in a register ring that advances by two
// blocks per output: 2 eight-byte LDS reads per 21 packed FMAs (0.7 B per MAC incl. the ring fill).  The 25 phase
// sums of an output are added across the half-wave by five DPP adds (row_shr 1 / 2 / 4 / 8, row_bcast 15), in a fixed
// order, float32 (the 42-term lane sums too; error against scipy's float64 ~3e-7 of the output scale, as before).
constexpr int FIR_Q = 21;          // resample_poly designs 20 max(up, down) + 1 taps: 21 blocks of `down` for up == 1

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    // v[lane] + v[source lane of the DPP pattern] (0 where the pattern has no source / the row is masked)
    const int sh = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(sh);
}

template <int NS>
__device__ __forceinline__ void fm_fir_phases(const float *fm_s, const fm_v2f (&T)[FIR_Q], int D, int PP, int lane, int wave,
                                              int mcnt, float *out) {
    const int e = lane >> 5, rp = lane & 31;
    const int rpc = rp < PP ? rp : PP - 1;               // idle lanes (zero taps) read valid words
    const int o0 = wave * 2 * NS + e;
    const float *wb = fm_s + o0 * D + 2 * rpc;
    fm_v2f x[FIR_Q];
#pragma unroll
    for (int q = 0; q < FIR_Q; ++q) x[q] = *reinterpret_cast<const fm_v2f *>(wb + q * D);
    float outv = 0.f;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        fm_v2f a0 = {0.f, 0.f}, a1 = a0, a2 = a0;
#pragma unroll
        for (int q = 0; q < FIR_Q; q += 3) {
            a0 = __builtin_elementwise_fma(x[(q + 2 * j) % FIR_Q], T[q], a0);
            a1 = __builtin_elementwise_fma(x[(q + 1 + 2 * j) % FIR_Q], T[q + 1], a1);
            a2 = __builtin_elementwise_fma(x[(q + 2 + 2 * j) % FIR_Q], T[q + 2], a2);
        }
        if (j + 1 < NS) {   // blocks o + 21, o + 22 replace blocks o, o + 1
            x[(2 * j) % FIR_Q] = *reinterpret_cast<const fm_v2f *>(wb + (2 * j + FIR_Q) * D);
            x[(2 * j + 1) % FIR_Q] = *reinterpret_cast<const fm_v2f *>(wb + (2 * j + FIR_Q + 1) * D);
        }
        const fm_v2f a = (a0 + a1) + a2;
        float sum = a.x + a.y;
        sum = dpp_add<0x111, 0xf>(sum);      // row_shr:1
        sum = dpp_add<0x112, 0xf>(sum);      // row_shr:2
        sum = dpp_add<0x114, 0xf>(sum);      // row_shr:4
        sum = dpp_add<0x118, 0xf>(sum);      // row_shr:8  -> lane 15 of every row: the row's total
        sum = dpp_add<0x142, 0xa>(sum);      // row_bcast:15 into rows 1 and 3 -> lanes 31 / 63: the half-waves' totals
        const float y_even = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sum), 31));
        const float y_odd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sum), 63));
        outv = lane == j ? y_even : (lane == 32 + j ? y_odd : outv);   // lane (e, j) keeps the output of step j
    }
    const int o = wave * 2 * NS + 2 * rp + e;            // lane (e, l) holds the output of step l
    if (rp < NS && o < mcnt) out[o] = outv;
}

// ROWS: the unfused chains' decimating FIR (the window is copied from a.rows_src instead of computed by phase 1)
template <bool ROWS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 5))) void fmbank_fused_kernel(FmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *taps_s = reinterpret_cast<float *>(smem_raw);                       // taps rounded to float32, reversed
    float *fm_s = reinterpret_cast<float *>(smem_raw + (size_t)((a.ntaps + 1) & ~1) * sizeof(float));
    __shared__ double red[8];

    const int run = blockIdx.x, k = blockIdx.y, chunk = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N;
    const int n_tiles = (a.n_out + a.TO - 1) / a.TO;
    const int t_lo = run * a.R;
    int t_hi = t_lo + a.R;
    if (t_hi > n_tiles) t_hi = n_tiles;
    const int W = a.ntaps + (a.TO - 1) * a.down;     // window length
    const int adv = a.TO * a.down;                   // window advance per tile
    const int keep = W - adv;                        // samples carried over (may be <= 0: then nothing is reused)
    // Which run sums a sample: the one whose tile computes it in its STANDARD pass.  Tile T > 0 of a run adds the window
    // indices [keep, W) = samples [T adv + dlt, (T + 1) adv + dlt), dlt = d0 + 1 - down; the first tile of a run has to
    // produce the whole window, but its part [0, keep) belongs to earlier tiles -- so it is made by a pass of its own
    // that sums nothing, and every sample is summed by the same pass with the same step grid wherever the row is cut.
    const int dlt = keep > 0 ? a.d0 + 1 - a.down : 0;
    // (always the case for resample_poly's designs: ntaps = 20 down + 1, d0 = 10 down -> dlt = 9 down + 1, keep = 19 down + 1;
    // a design outside that keeps the round-2 rule: owned = [t_lo adv, t_hi adv), sums then depend on the cut at 1e-8)
    const bool split_first = keep > 0 && dlt >= 0 && dlt <= keep;
    const int own_lo = t_lo == 0 ? 0 : (split_first ? t_lo * adv + dlt : t_lo * adv);
    const int own_hi = t_hi >= n_tiles ? N : (split_first ? t_hi * adv + dlt : t_hi * adv);   // exclusive

    // taps reversed (trev[i] = h[ntaps-1-i]) so that y[o] = sum_i fm_s[o*down + i] * trev[i] walks both arrays
    // upwards; one zero pad makes the count even for the paired reads
    const int ntp = (a.ntaps + 1) & ~1;
    const bool phases = a.fir_phase != 0;
    fm_v2f Tq[FIR_Q];                // polyphase form: the lane's 21 tap pairs (phase pair rp = lane & 31), kept for the run
    if (phases) {
        fm_s = reinterpret_cast<float *>(smem_raw);       // no tap table in LDS
#pragma unroll
        for (int q = 0; q < FIR_Q; ++q) {
            const float2 t = a.tphase[q * 32 + (lane & 31)];
            Tq[q] = fm_v2f{t.x, t.y};
        }
        if (tid < 64) fm_s[W + tid] = 0.0f;   // the last block of the last output reaches 49 words past the window (zero taps)
    } else {
#pragma unroll
        for (int q = 0; q < FIR_Q; ++q) Tq[q] = fm_v2f{0.f, 0.f};
        for (int j = tid; j < ntp; j += 256) taps_s[j] = j < a.ntaps ? (float)a.taps[a.ntaps - 1 - j] : 0.0f;
        if (tid == 0) fm_s[W] = 0.0f;   // partner of the zero tap pad
    }

    const float c = a.nco_c[k];
    const bool do_mix = c != 0.0f;
    const size_t in_base = (size_t)chunk * N;
    double p_base = 0.0, p_fm = 0.0;

    for (int tile = t_lo; tile < t_hi; ++tile) {
        const int m0 = tile * a.TO;
        int mcnt = a.n_out - m0;
        if (mcnt > a.TO) mcnt = a.TO;
        const int n_lo = m0 * a.down + a.d0 - (a.ntaps - 1);   // chunk index of window slot 0
        // phase 1: the whole window for the first tile of the run, the new part afterwards
        const bool head = !ROWS && split_first && tile == t_lo && tile > 0;   // first tile of a later run: [0, keep) first
        const int i_lo = ((tile == t_lo && !head) || keep <= 0) ? 0 : keep;
        // interior: every sample this pass touches (incl. lane 0's predecessor sample and the idle lanes of its last
        // 127-sample step) exists, every sample it stores is owned and has a predecessor
        const bool interior = n_lo + i_lo >= (own_lo > 1 ? own_lo : 1) && n_lo + W <= own_hi && n_lo + W + 128 <= N;
        const bool fast = a.araw != nullptr;           // raw angles taken once per chunk (fm_rawdisc_kernel, with sum |x|^2)
        const bool base = k == 0 && !fast;
#define WH_P1X(CHK, F, M, LO, HI, OL, OH, PB, PF) \
    fm_phase1<CHK, F, M>(a, fm_s, LO, HI, n_lo, N, c, in_base, OL, OH, lane, wave, PB, PF, base)
#define WH_P1(CHK, F, M) WH_P1X(CHK, F, M, i_lo, W, own_lo, own_hi, p_base, p_fm)
#define WH_P1F(CHK, F, M, LO, HI, OL, OH, PF, SUMS) \
    fm_phase1_fast<CHK, F, M>(a, fm_s, LO, HI, n_lo, N, c, in_base, OL, OH, lane, wave, PF, SUMS)
        if (head) {
            // the carried part of the window, owned by earlier runs: computed, not summed (empty ownership range; its sums
            // go to scratch variables)
            double nb_ = 0.0, nf_ = 0.0;
            const bool in_a = n_lo >= 1 && n_lo + keep + 128 <= N;
#define WH_P1A(CHK, F, M) WH_P1X(CHK, F, M, 0, keep, 0, 0, nb_, nf_)
            if (fast) {
                if (in_a) {
                    if (a.fmt == 1) { if (do_mix) WH_P1F(false, 1, true, 0, keep, 0, 0, nf_, false); else WH_P1F(false, 1, false, 0, keep, 0, 0, nf_, false); }
                    else            { if (do_mix) WH_P1F(false, 0, true, 0, keep, 0, 0, nf_, false); else WH_P1F(false, 0, false, 0, keep, 0, 0, nf_, false); }
                } else {
                    if (a.fmt == 1) { if (do_mix) WH_P1F(true, 1, true, 0, keep, 0, 0, nf_, false); else WH_P1F(true, 1, false, 0, keep, 0, 0, nf_, false); }
                    else            { if (do_mix) WH_P1F(true, 0, true, 0, keep, 0, 0, nf_, false); else WH_P1F(true, 0, false, 0, keep, 0, 0, nf_, false); }
                }
            } else if (in_a) {
                if (a.fmt == 1) { if (do_mix) WH_P1A(false, 1, true); else WH_P1A(false, 1, false); }
                else            { if (do_mix) WH_P1A(false, 0, true); else WH_P1A(false, 0, false); }
            } else {
                if (a.fmt == 1) { if (do_mix) WH_P1A(true, 1, true); else WH_P1A(true, 1, false); }
                else            { if (do_mix) WH_P1A(true, 0, true); else WH_P1A(true, 0, false); }
            }
#undef WH_P1A
        }
        if (ROWS) {
            // unfused chains (IIR stages, AGC, AM / SSB fronts): the rows are in memory already; this kernel is their
            // decimating FIR (the generic one-wave-per-output resampler took 13 of 18 ms for 32 default-config NBFM
            // channels x 200 chunks)
            const float *rw = a.rows_src + ((size_t)chunk * a.K + k) * N;
            for (int i = i_lo + tid; i < W; i += 256) {
                const int n = n_lo + i;
                fm_s[i] = (n >= 0 && n < N) ? rw[n] : 0.0f;
            }
        } else if (fast) {
            if (interior) {
                if (a.fmt == 1) { if (do_mix) WH_P1F(false, 1, true, i_lo, W, own_lo, own_hi, p_fm, true); else WH_P1F(false, 1, false, i_lo, W, own_lo, own_hi, p_fm, true); }
                else            { if (do_mix) WH_P1F(false, 0, true, i_lo, W, own_lo, own_hi, p_fm, true); else WH_P1F(false, 0, false, i_lo, W, own_lo, own_hi, p_fm, true); }
            } else {
                if (a.fmt == 1) { if (do_mix) WH_P1F(true, 1, true, i_lo, W, own_lo, own_hi, p_fm, true); else WH_P1F(true, 1, false, i_lo, W, own_lo, own_hi, p_fm, true); }
                else            { if (do_mix) WH_P1F(true, 0, true, i_lo, W, own_lo, own_hi, p_fm, true); else WH_P1F(true, 0, false, i_lo, W, own_lo, own_hi, p_fm, true); }
            }
        } else if (interior) {
            if (a.fmt == 1) { if (do_mix) WH_P1(false, 1, true); else WH_P1(false, 1, false); }
            else            { if (do_mix) WH_P1(false, 0, true); else WH_P1(false, 0, false); }
        } else {
            if (a.fmt == 1) { if (do_mix) WH_P1(true, 1, true); else WH_P1(true, 1, false); }
            else            { if (do_mix) WH_P1(true, 0, true); else WH_P1(true, 0, false); }
        }
#undef WH_P1
#undef WH_P1X
#undef WH_P1F
        __syncthreads();  // window (and, first time, taps) visible

        // phase 2: y[m0+o] = sum_i fm_s[o*down + i] * trev[i].
        if (phases) {
            fm_fir_phases<16>(fm_s, Tq, a.down, a.down >> 1, lane, wave, mcnt,
                              a.audio + ((size_t)chunk * a.K + k) * a.n_out + m0);
        } else if (a.TO > 96 && a.TO <= 128 && (a.TO & 3) == 0 && (a.down & 1) == 0) {
            // 8 lanes share 4 outputs: each lane takes a contiguous 1/8 of the (paired) taps for all 4, so a tap pair
            // is read once per 8 MACs; packed FMAs; float32 partial sums over 32 pairs, added in float64 (error ~3e-7
            // of the output scale; scipy accumulates in float64).
            // lane -> (output group og, tap slice sub): a half-wave holds 8 groups x 4 slices and the slice length is
            // = 1 (mod 32) pairs, so its 32 eight-byte reads fall on float offsets 8*og + 2*sub (mod 64): all 64 LDS
            // banks exactly once
            typedef float v2f __attribute__((ext_vector_type(2)));
            const int og = (wave << 3) | ((lane >> 2) & 7), sub = (lane & 3) | ((lane >> 5) << 2);
            const int npairs = ntp >> 1;
            int per = (npairs + 7) >> 3;
            per += (33 - (per & 31)) & 31;   // round up to 1 (mod 32)
            int p_lo = sub * per;
            if (p_lo > npairs) p_lo = npairs;
            int p_hi = p_lo + per;
            if (p_hi > npairs) p_hi = npairs;
            const v2f *tp = reinterpret_cast<const v2f *>(taps_s);
            const v2f *f0 = reinterpret_cast<const v2f *>(fm_s + (4 * og) * a.down);
            const int dstep = a.down >> 1;   // output stride in pairs
            double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
            if (4 * og >= a.TO) p_hi = p_lo;   // tiles of 100..124 outputs: the last groups have nothing to do
            for (int pb = p_lo; pb < p_hi; pb += 32) {
                int pe = pb + 32 < p_hi ? pb + 32 : p_hi;
                v2f s0 = {0.f, 0.f}, s1 = {0.f, 0.f}, s2 = {0.f, 0.f}, s3 = {0.f, 0.f};
#pragma unroll 4
                for (int pp = pb; pp < pe; ++pp) {
                    v2f t = tp[pp];
                    s0 = __builtin_elementwise_fma(f0[pp], t, s0);
                    s1 = __builtin_elementwise_fma(f0[pp + dstep], t, s1);
                    s2 = __builtin_elementwise_fma(f0[pp + 2 * dstep], t, s2);
                    s3 = __builtin_elementwise_fma(f0[pp + 3 * dstep], t, s3);
                }
                acc0 += (double)(s0.x + s0.y);
                acc1 += (double)(s1.x + s1.y);
                acc2 += (double)(s2.x + s2.y);
                acc3 += (double)(s3.x + s3.y);
            }
            for (int w = 1; w <= 32; w = (w == 2 ? 32 : w << 1)) {   // slice bits live in lane bits 0, 1 and 5
                acc0 += __shfl_xor(acc0, w);
                acc1 += __shfl_xor(acc1, w);
                acc2 += __shfl_xor(acc2, w);
                acc3 += __shfl_xor(acc3, w);
            }
            if (sub < 4) {
                const int o = 4 * og + sub;
                const double v = sub == 0 ? acc0 : (sub == 1 ? acc1 : (sub == 2 ? acc2 : acc3));
                if (o < mcnt) a.audio[((size_t)chunk * a.K + k) * a.n_out + m0 + o] = (float)v;
            }
        } else {
            // generic: S = 256/TO lanes share one output
            const int S = 256 / a.TO;
            const int o = tid / S, sub = tid - o * S;
            double accv = 0.0;
            if (o < mcnt) {
                const float *f = fm_s + o * a.down;
                for (int j0 = sub; j0 < a.ntaps; j0 += 32 * S) {
                    float part = 0.f;
                    int j1 = j0 + 32 * S < a.ntaps ? j0 + 32 * S : a.ntaps;
                    for (int j = j0; j < j1; j += S) part = fmaf(f[j], taps_s[j], part);
                    accv += (double)part;
                }
            }
            for (int w = 1; w < S; w <<= 1) accv += __shfl_xor(accv, w);
            if (o < mcnt && sub == 0) a.audio[((size_t)chunk * a.K + k) * a.n_out + m0 + o] = (float)accv;
        }
        if (tile + 1 < t_hi && keep > 0) {   // slide the window
            __syncthreads();
            float carry[4];                  // keep <= ntaps <= FM_MAX_TAPS: at most 8 per thread, 4 at a time
            for (int j0 = 0; j0 < keep; j0 += 4 * 256) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int j = j0 + u * 256 + tid;
                    carry[u] = j < keep ? fm_s[adv + j] : 0.f;
                }
                __syncthreads();             // source and destination ranges can overlap when keep > adv
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int j = j0 + u * 256 + tid;
                    if (j < keep) fm_s[j] = carry[u];
                }
                __syncthreads();
            }
        }
    }
    if (ROWS) return;   // (the rows' power sums were taken where the rows were made)
    // block-reduce the two power sums (float64) and publish with one atomic pair
    double db = p_base, df = p_fm;
    for (int o = 32; o > 0; o >>= 1) {
        db += __shfl_xor(db, o);
        df += __shfl_xor(df, o);
    }
    if (lane == 0) {
        red[wave] = db;
        red[4 + wave] = df;
    }
    __syncthreads();
    if (tid == 0) {
        double *acc = a.acc + ((size_t)chunk * a.K + k) * 2;
        if (k == 0 && a.araw == nullptr) atomicAdd(acc, red[0] + red[1] + red[2] + red[3]);   // shared by the chunk's channels (finalize); the fast front's pre-pass owns it
        atomicAdd(acc + 1, red[4] + red[5] + red[6] + red[7]);
    }
}

// unfused first stage: NCO mix + demodulator front (FM discriminator / AM envelope / SSB product
// detector) to HBM rows + power sums
__global__ __launch_bounds__(256) void chan_front_kernel(FmArgs a) {
    const int k = blockIdx.y, chunk = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N;
    const float c = a.nco_c[k];
    const bool do_mix = c != 0.0f;
    const size_t in_base = (size_t)chunk * N;
    float *fm = a.fm_out + ((size_t)chunk * a.K + k) * N;
    __shared__ double red[8];
    float p_base = 0.f, p_fm = 0.f;
    const int per_block = 4 * 63 * 16;
    const int b_lo = blockIdx.x * per_block;
    int b_hi = b_lo + per_block;
    if (b_hi > N) b_hi = N;
    if (a.demod == 0) {
        // FM rows: the fused kernel's phase 1 (two samples per lane, packed arithmetic, hardware sin / cos, loads one step
        // ahead) writing the row to memory instead of the LDS window -- window index == chunk index, this block owns
        // [b_lo, b_hi)
        double pb = 0.0, pf = 0.0;
        const bool interior = b_lo >= 2 && b_hi + 128 <= N;
#define WH_F1(CHK, F, M) fm_phase1<CHK, F, M>(a, fm, b_lo, b_hi, 0, N, c, in_base, b_lo, b_hi, lane, wave, pb, pf, true)
        if (interior) {
            if (a.fmt == 1) { if (do_mix) WH_F1(false, 1, true); else WH_F1(false, 1, false); }
            else            { if (do_mix) WH_F1(false, 0, true); else WH_F1(false, 0, false); }
        } else {
            if (a.fmt == 1) { if (do_mix) WH_F1(true, 1, true); else WH_F1(true, 1, false); }
            else            { if (do_mix) WH_F1(true, 0, true); else WH_F1(true, 0, false); }
        }
#undef WH_F1
        for (int o = 32; o > 0; o >>= 1) {
            pb += __shfl_xor(pb, o);
            pf += __shfl_xor(pf, o);
        }
        if (lane == 0) {
            red[wave] = pb;
            red[4 + wave] = pf;
        }
        __syncthreads();
        if (tid == 0) {
            double *acc = a.acc + ((size_t)chunk * a.K + k) * 2;
            atomicAdd(acc, red[0] + red[1] + red[2] + red[3]);
            if (!a.skip_fm_sum) atomicAdd(acc + 1, red[4] + red[5] + red[6] + red[7]);
        }
        return;
    }
    for (int s0 = b_lo + wave * 63; s0 < b_hi; s0 += 4 * 63) {
        int n = s0 + lane - 1;
        bool valid = (n >= 0) && (n < N);
        float2 bse = make_float2(0.f, 0.f);
        if (valid) bse = mix(load_iq(a.in, a.fmt, in_base + n), c, n, do_mix);
        float2 prv;
        prv.x = __shfl_up(bse.x, 1);
        prv.y = __shfl_up(bse.y, 1);
        if (lane > 0 && valid && n < b_hi) {
            float v = 0.f;
            if (a.demod == 0) {          // dsp/fm.py:65-97
                if (n >= 1) {
                    float re = bse.x * prv.x + bse.y * prv.y;
                    float im = bse.y * prv.x - bse.x * prv.y;
                    v = whm_atan2f(im, re) * a.scale;
                }
            } else if (a.demod == 1) {   // dsp/am.py:103 np.abs(iq)
                v = hypotf(bse.x, bse.y);
            } else {                     // dsp/am.py:23-42, 204-210: BFO with float64 phase, + sign; real part
                double t = (double)n / a.fs_d;
                double th = a.bfo_c * t;
                double sn, cs;
                sincos(th, &sn, &cs);
                v = __fsub_rn(__fmul_rn(bse.x, (float)cs), __fmul_rn(bse.y, (float)sn));
            }
            fm[n] = v;
            p_base += bse.x * bse.x + bse.y * bse.y;
            p_fm += v * v;
        }
    }
    double db = (double)p_base, df = (double)p_fm;
    for (int o = 32; o > 0; o >>= 1) {
        db += __shfl_xor(db, o);
        df += __shfl_xor(df, o);
    }
    if (lane == 0) {
        red[wave] = db;
        red[4 + wave] = df;
    }
    __syncthreads();
    if (tid == 0) {
        double *acc = a.acc + ((size_t)chunk * a.K + k) * 2;
        atomicAdd(acc, red[0] + red[1] + red[2] + red[3]);
        if (!a.skip_fm_sum) atomicAdd(acc + 1, red[4] + red[5] + red[6] + red[7]);
    }
}

// Synchronous AM front (dsp/sam.py:73-122 CarrierRecoveryPLL.process + :219-232 sideband selection): a
// type-2 PLL that feeds back every sample, fresh per chunk (sam_demod_simple passes no pll_state), all in
// float64 like the reference's Python floats / complex128; outputs rounded to float32.  One lane per
// (chunk, channel) row; lanes of a wave read the same input sample (broadcast).
__global__ __launch_bounds__(64) void sam_front_kernel(FmArgs a, int n_rows) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rows) return;
    const int chunk = r / a.K, k = r - chunk * a.K;
    const int N = a.N;
    const float c = a.nco_c[k];
    const bool do_mix = c != 0.0f;
    const size_t in_base = (size_t)chunk * N;
    float *row = a.fm_out + (size_t)r * N;
    const double PI_D = 3.141592653589793;
    double phase = 0.0, integ = 0.0;
    double p_base = 0.0;
    for (int n = 0; n < N; ++n) {
        float2 x = mix(load_iq(a.in, a.fmt, in_base + n), c, n, do_mix);
        p_base += (double)(x.x * x.x + x.y * x.y);
        double sn, cs;
        sincos(phase, &sn, &cs);
        const double lr = cs, li = -sn;                       // exp(-1j * phase)
        const double xr = (double)x.x, xi = (double)x.y;
        const double mr = __dsub_rn(__dmul_rn(xr, lr), __dmul_rn(xi, li));
        const double mi = __dadd_rn(__dmul_rn(xr, li), __dmul_rn(xi, lr));
        const float ci = (float)mr, cq = (float)mi;
        row[n] = a.demod == 3 ? ci : (a.demod == 4 ? __fadd_rn(ci, cq) : __fsub_rn(ci, cq));
        const double pe = atan2(mi, __dadd_rn(fabs(mr), 1e-10));
        integ = __dadd_rn(integ, __dmul_rn(a.pll_beta, pe));
        const double fc = __dadd_rn(__dmul_rn(a.pll_alpha, pe), integ);
        phase = __dadd_rn(phase, fc);
        if (phase > PI_D) phase = __dsub_rn(phase, 2 * PI_D);
        else if (phase < -PI_D) phase = __dadd_rn(phase, 2 * PI_D);
    }
    a.acc[(size_t)r * 2] = p_base;
}

// Sequential IIR stages on one (chunk, channel) row, zero initial state per chunk (stateless
// operator): scipy.signal.lfilter's direct-form-II-transposed recurrence in the stage's dtype
// (float32 stages: de-emphasis dsp/fm.py:101-126, AGC one-poles; float64 stages: Butterworth /
// notch, dsp/filters.py:86-264, dsp/fm.py:129-181), each output rounded to float32 like the
// reference's .astype(np.float32); optional AGC (dsp/agc.py:169-242).  One lane per row.
struct AgcDev {
    int on;
    float target, max_gain, att_b0, att_a1, rel_b0, rel_a1;
};

struct StageArr {
    StageDev st[MAX_STAGES];   // by value: the coefficients are wave-uniform and arrive through scalar loads
};

// One lfilter stage (direct form II transposed, scipy's recurrence) with NC coefficients over the cnt samples of one
// row held in the LDS tile: coefficients and state in registers, every operation rounded on its own (no FMA), in the
// stage's dtype; the next sample is fetched from LDS while the current one walks the recurrence.
constexpr int ROWS_CH = 64;

// FUSED: the state updates as fused multiply-adds (half the float64 issue slots of the loop).  Only the time-parallel form
// uses it: it serves chains whose recurrences are well conditioned (their impulse responses die out inside the chunk), where
// one rounding more or less per update is far below the float32 output, and it is not bit-comparable with the sequential
// recurrence anyway (its start states are truncated sums).  The sequential form -- the only one the reference's
// ill-conditioned ba-form high-passes can take -- keeps scipy's operation order exactly.
template <int NC, bool F64, bool UNI = false, bool FUSED = false>
__device__ __forceinline__ void iir_run(float (*tile)[65], int cnt, int lane, double (&z)[MAX_ORD - 1], const StageDev &S) {
    // S lives in LDS: the coefficients arrive in VGPRs (read from the scalar kernarg copy the compiler keeps them in
    // SGPRs, runs out of those and round-trips them through v_readlane inside the sample loop)
    double b[NC], a[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { b[k] = S.b[k]; a[k] = S.a[k]; }
    auto step = [&](float x) -> float {
        if (F64 && FUSED) {
            const double xd = (double)x;
            const double y = fma(b[0], xd, z[0]);
#pragma unroll
            for (int k = 0; k < NC - 2; ++k) z[k] = fma(-y, a[k + 1], fma(xd, b[k + 1], z[k + 1]));
            if (NC >= 2) z[NC - 2] = fma(-y, a[NC - 1], xd * b[NC - 1]);
            return (float)y;
        } else if (F64) {
            const double xd = (double)x;
            const double y = __dadd_rn(z[0], __dmul_rn(b[0], xd));
#pragma unroll
            for (int k = 0; k < NC - 2; ++k) z[k] = __dsub_rn(__dadd_rn(z[k + 1], __dmul_rn(xd, b[k + 1])), __dmul_rn(y, a[k + 1]));
            if (NC >= 2) z[NC - 2] = __dsub_rn(__dmul_rn(xd, b[NC - 1]), __dmul_rn(y, a[NC - 1]));
            return (float)y;
        } else if (FUSED) {
            const float y = fmaf((float)b[0], x, (float)z[0]);
#pragma unroll
            for (int k = 0; k < NC - 2; ++k) z[k] = (double)fmaf(-y, (float)a[k + 1], fmaf(x, (float)b[k + 1], (float)z[k + 1]));
            if (NC >= 2) z[NC - 2] = (double)fmaf(-y, (float)a[NC - 1], x * (float)b[NC - 1]);
            return y;
        } else {
            const float y = __fadd_rn((float)z[0], __fmul_rn((float)b[0], x));
#pragma unroll
            for (int k = 0; k < NC - 2; ++k)
                z[k] = (double)__fsub_rn(__fadd_rn((float)z[k + 1], __fmul_rn(x, (float)b[k + 1])), __fmul_rn(y, (float)a[k + 1]));
            if (NC >= 2) z[NC - 2] = (double)__fsub_rn(__fmul_rn(x, (float)b[NC - 1]), __fmul_rn(y, (float)a[NC - 1]));
            return y;
        }
    };
    if (UNI) {
        // cnt is the same in every lane (the caller pads short rows with zeros where that is exact): a scalar trip
        // count and a branch-free prefetch -- no exec masking or divergent branches in the sample loop
        cnt = __builtin_amdgcn_readfirstlane(cnt);
        float xn = tile[0][lane];
        for (int j = 0; j < cnt; ++j) {
            const float x = xn;
            const int jn = j + 1 < cnt ? j + 1 : j;   // scalar
            xn = tile[jn][lane];
            tile[j][lane] = step(x);
        }
        return;
    }
    float xn = tile[0][lane];
    for (int j = 0; j < cnt; ++j) {
        const float x = xn;
        if (j + 1 < cnt) xn = tile[j + 1][lane];
        tile[j][lane] = step(x);
    }
}

template <bool F64, bool UNI = false, bool FUSED = false>
__device__ __forceinline__ void iir_stage(float (*tile)[65], int cnt, int lane, double (&z)[MAX_ORD - 1], const StageDev &S) {
    switch (__builtin_amdgcn_readfirstlane(S.n)) {   // wave-uniform, once per (tile, stage)
        case 1: iir_run<1, F64, UNI, FUSED>(tile, cnt, lane, z, S); break;
        case 2: iir_run<2, F64, UNI, FUSED>(tile, cnt, lane, z, S); break;
        case 3: iir_run<3, F64, UNI, FUSED>(tile, cnt, lane, z, S); break;
        case 4: iir_run<4, F64, UNI, FUSED>(tile, cnt, lane, z, S); break;
        case 5: iir_run<5, F64, UNI, FUSED>(tile, cnt, lane, z, S); break;
        case 6: iir_run<6, F64, UNI, FUSED>(tile, cnt, lane, z, S); break;
        case 7: iir_run<7, F64, UNI, FUSED>(tile, cnt, lane, z, S); break;
        case 8: iir_run<8, F64, UNI, FUSED>(tile, cnt, lane, z, S); break;
        case 9: iir_run<9, F64, UNI, FUSED>(tile, cnt, lane, z, S); break;
        case 10: iir_run<10, F64, UNI, FUSED>(tile, cnt, lane, z, S); break;
        default: iir_run<11, F64, UNI, FUSED>(tile, cnt, lane, z, S); break;
    }
}

// ---- start states of the time-parallel form by the chain's IMPULSE RESPONSE --------------------------------------------
// The warm-up pass of the time-parallel form runs the recurrences over the `warm` samples before a segment only to get the
// states at the segment's start.  The chain is linear, so those states are a dot product: with G[d] = the states of all
// stages d samples after a unit impulse entered the cascade (zero state before), the state after sample st - 1 is
// sum_d x[st - 1 - d] G[d], d < warm (the same truncation as the warm-up: the response has decayed below 1e-10 by then).
// m FMAs per sample and stage instead of the recurrence's ~50 dependent operations.  G is produced at bank creation by the
// recurrences themselves (iir_impulse_kernel: same operations, same float32 roundings between stages).
__device__ __forceinline__ float iir_step_rt(const StageDev &S, float x, double *z) {
    const int n = S.n;
    if (S.is_f64) {
        const double xd = (double)x;
        const double y = __dadd_rn(z[0], __dmul_rn(S.b[0], xd));
        for (int k = 0; k < n - 2; ++k) z[k] = __dsub_rn(__dadd_rn(z[k + 1], __dmul_rn(xd, S.b[k + 1])), __dmul_rn(y, S.a[k + 1]));
        if (n >= 2) z[n - 2] = __dsub_rn(__dmul_rn(xd, S.b[n - 1]), __dmul_rn(y, S.a[n - 1]));
        return (float)y;
    }
    const float y = __fadd_rn((float)z[0], __fmul_rn((float)S.b[0], x));
    for (int k = 0; k < n - 2; ++k)
        z[k] = (double)__fsub_rn(__fadd_rn((float)z[k + 1], __fmul_rn(x, (float)S.b[k + 1])), __fmul_rn(y, (float)S.a[k + 1]));
    if (n >= 2) z[n - 2] = (double)__fsub_rn(__fmul_rn(x, (float)S.b[n - 1]), __fmul_rn(y, (float)S.a[n - 1]));
    return y;
}

// G[d][e], d < warm, e = compact state index (stage 0's states, then stage 1's, ...; row length EP >= the state count,
// zero padded): one thread walks the cascade over a unit impulse (bank creation, once)
__global__ void iir_impulse_kernel(StageArr sa, int n_stages, int warm, int EP, double *G) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double z[MAX_STAGES][MAX_ORD - 1];
    for (int s = 0; s < MAX_STAGES; ++s)
        for (int k = 0; k < MAX_ORD - 1; ++k) z[s][k] = 0.0;
    for (int d = 0; d < warm; ++d) {
        float v = d == 0 ? 1.0f : 0.0f;
        for (int s = 0; s < n_stages; ++s) v = iir_step_rt(sa.st[s], v, z[s]);
        int e = 0;
        for (int s = 0; s < n_stages; ++s)
            for (int k = 0; k < sa.st[s].n - 1; ++k) G[(size_t)d * EP + e++] = z[s][k];
        for (; e < EP; ++e) G[(size_t)d * EP + e] = 0.0;
    }
}

// Start states of every segment of every row: S0[row][segment][e] = sum_{d < warm} x[st - 1 - d] G[d][e], st = segment
// start -- a decimating FIR with the impulse-response states as taps.  Workgroup = 16 segments of one row: wave w takes
// the segments 4 w .. 4 w + 3 of the group, lane l the delays l, l + 64, ... (a G row is loaded once for the four
// segments; a wave's lanes read 64 consecutive samples and 64 consecutive G rows), then the lanes are summed by shuffles.
// The whole chip works on what the warm-up pass did with eight waves per row.
constexpr int START_SEGS = 16, START_SPT = 4;
template <int EP>
__global__ __launch_bounds__(256) void iir_start_kernel(const float *rows, int N, int seg, int warm, int nseg,
                                                        const double *G, double *S0) {
    const int r = blockIdx.x, t = threadIdx.x;
    const int slot = t >> 6, tl = t & 63;
    const int p0 = blockIdx.y * START_SEGS + START_SPT * slot;
    const float *x = rows + (size_t)r * N;
    double acc[START_SPT][EP];
#pragma unroll
    for (int q = 0; q < START_SPT; ++q)
#pragma unroll
        for (int e = 0; e < EP; ++e) acc[q][e] = 0.0;
    const long long st0 = (long long)p0 * seg, stl = st0 + (long long)(START_SPT - 1) * seg;
    if (st0 < N) {
        const int d_end = stl < warm ? (int)stl : warm;   // delays that still meet a sample of the last segment's past
        // Four delays per round, written out as "all loads, then all FMAs": every load is unconditional (clamped index,
        // the value zeroed afterwards) so that a round's 28 loads are in flight together -- with a conditional load, or
        // the loop left to the compiler's unrolling, every delay paid its own memory round trips (1.9 us per delay).
        typedef double d2 __attribute__((ext_vector_type(2)));
        // interior groups (every delay of every segment meets a sample of the row, whole rounds): no index clamps, no
        // selects -- the edge form below spends more instructions on those than on the FMAs
        const bool interior = st0 - warm >= 0 && stl <= N && d_end == warm;
        const int d_full = interior ? (warm / 256) * 256 : 0;   // delays covered by whole unchecked rounds
        const float *xe = x + st0 - 1;
        for (int d = tl; d < d_full; d += 256) {
            float xv[4][START_SPT];
            d2 gv[4][EP / 2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int dd = d + 64 * u;
#pragma unroll
                for (int q = 0; q < START_SPT; ++q) xv[u][q] = xe[q * seg - dd];
                const d2 *g2 = reinterpret_cast<const d2 *>(G + (size_t)dd * EP);
#pragma unroll
                for (int e = 0; e < EP / 2; ++e) gv[u][e] = g2[e];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < START_SPT; ++q) {
                    const double xd = (double)xv[u][q];
#pragma unroll
                    for (int e = 0; e < EP / 2; ++e) {
                        acc[q][2 * e] = fma(xd, gv[u][e].x, acc[q][2 * e]);
                        acc[q][2 * e + 1] = fma(xd, gv[u][e].y, acc[q][2 * e + 1]);
                    }
                }
        }
        for (int d = d_full + tl; d < d_end; d += 256) {
            float xv[4][START_SPT];
            d2 gv[4][EP / 2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int dd = d + 64 * u;
                const int dc = dd < d_end ? dd : tl;          // (a delay past the end re-reads a valid row; its x is zeroed)
#pragma unroll
                for (int q = 0; q < START_SPT; ++q) {
                    const long long n = st0 + (long long)q * seg - 1 - dc;
                    const long long nc = n < 0 ? 0 : (n >= N ? N - 1 : n);
                    xv[u][q] = x[nc];
                }
                const d2 *g2 = reinterpret_cast<const d2 *>(G + (size_t)dc * EP);   // rows are 16-byte aligned (EP even)
#pragma unroll
                for (int e = 0; e < EP / 2; ++e) gv[u][e] = g2[e];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int dd = d + 64 * u;
#pragma unroll
                for (int q = 0; q < START_SPT; ++q) {
                    const long long n = st0 + (long long)q * seg - 1 - dd;
                    // (a segment starting beyond the row keeps zero states)
                    const double xd = (dd < d_end && n >= 0 && n < N) ? (double)xv[u][q] : 0.0;
#pragma unroll
                    for (int e = 0; e < EP / 2; ++e) {
                        acc[q][2 * e] = fma(xd, gv[u][e].x, acc[q][2 * e]);
                        acc[q][2 * e + 1] = fma(xd, gv[u][e].y, acc[q][2 * e + 1]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < START_SPT; ++q) {
        const int p = p0 + q;
#pragma unroll
        for (int e = 0; e < EP; ++e) {
            double v = acc[q][e];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (tl == 0 && p < nseg) S0[((size_t)r * nseg + p) * EP + e] = (long long)p * seg < N ? v : 0.0;
        }
    }
}

// One lane per "virtual row"; samples move through an LDS staging tile of ROWS_CH samples x 64 virtual rows: the wave
// loads it with coalesced 256-byte reads (all 64 lanes on one virtual row at a time), each lane then runs its row
// through stage after stage (cascading per tile is the same arithmetic as cascading per sample), then the AGC, and the
// tile is written back coalesced.
//   seg == 0: a wave owns 64 (chunk, channel) rows, one lane each, zero initial state (the stateless operator).
//   seg  > 0: TIME-PARALLEL mode for chains whose impulse responses have died out after `warm` samples (the host
//             derives `warm` from the pole radii, target 1e-10): a wave owns ONE row cut into 64 segments of `seg`
//             samples; pass A runs every lane over the `warm` samples before its segment (from zero state, nothing
//             written: the true state differs by < 1e-10 relative), pass B runs it over its segment in place.  64x less
//             sequential depth for (seg + warm) / seg times the arithmetic.  Not available with the AGC (its release
//             time constant spans the chunk).
//             With few rows the launch is pure latency and most CUs idle, so the row may be cut finer: blockDim.x / 64
//             waves per row, 64 segments each (the host picks the wave count from the bank's channel count only, so a
//             bank's numbers do not depend on how many chunks a call carries).
constexpr int ROWS_MAXW = 8;
template <int NS>
__global__ __launch_bounds__(NS <= 3 ? 64 * ROWS_MAXW : 64) void chan_rows_kernel(float *rows, double *acc, int n_rows, int N, StageArr sa,
                                                                  AgcDev agc, int seg, int warm, const double *S0, int ep) {
    extern __shared__ __attribute__((aligned(16))) float tile_raw[];   // per wave [ROWS_CH][65]: [sample][row], padded:
    __shared__ StageDev st_s[NS > 0 ? NS : 1];                         // column walks and row walks are conflict-free
    __shared__ double red_s[ROWS_MAXW];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float (*tile)[65] = reinterpret_cast<float (*)[65]>(tile_raw + (size_t)wave * ROWS_CH * 65);
    for (int i = threadIdx.x; i < NS * (int)(sizeof(StageDev) / 4); i += blockDim.x)
        reinterpret_cast<int *>(st_s)[i] = reinterpret_cast<const int *>(sa.st)[i];
    __syncthreads();
    const int r0 = seg ? blockIdx.x : blockIdx.x * 64;
    // time-parallel mode: gridDim.y workgroups share a row (segments [blockIdx.y, +1) * 64 * waves)
    const int wpg = (int)(blockDim.x >> 6), ngrp = seg ? (int)gridDim.y : 1;
    const int s0 = ((seg ? (int)blockIdx.y : 0) * wpg + wave) * 64;   // first segment of this wave
    int nr = seg ? (N + seg - 1) / seg - s0 : n_rows - r0;     // live lanes (virtual rows)
    nr = nr < 0 ? 0 : (nr > 64 ? 64 : nr);
    double z[NS > 0 ? NS : 1][MAX_ORD - 1];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int k = 0; k < MAX_ORD - 1; ++k) z[s][k] = 0.0;
    float za = 0.f, zr = 0.f;
    const float NORM = (float)(1.0 / 0.90514825364486640);
    double ss = 0.0;
    // S0 != nullptr: the segments' start states were computed by iir_start_kernel (impulse-response dot product); the
    // warm-up pass (pass 0) is skipped
    if (seg && S0 != nullptr && NS > 0) {
        const int nseg = ngrp * wpg * 64;
        const double *src = S0 + ((size_t)r0 * nseg + s0 + lane) * ep;   // compact: stage 0's states, then stage 1's, ...
        int off = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int m = __builtin_amdgcn_readfirstlane(st_s[s].n) - 1;
            const bool f64 = st_s[s].is_f64 != 0;
#pragma unroll
            for (int k = 0; k < MAX_ORD - 1; ++k)
                if (k < m) {
                    const double v = src[off + k];
                    z[s][k] = f64 ? v : (double)(float)v;   // float32 stages keep float32 states
                }
            off += m;
        }
    }
    for (int pass = (seg && S0 == nullptr) ? 0 : 1; pass < 2; ++pass) {
        // virtual row l: offset vo(l) into the buffer, length vn(l)
        // Every lane walks the same number of samples per tile (a scalar trip count in iir_run): the warm-up pass is
        // RIGHT-aligned -- virtual sample i of segment l is chunk sample st - warm + i, zero where that is before the
        // row (zero input on zero state leaves the state exactly zero) -- and a short last segment is followed by
        // zeros whose outputs are never written.
        auto vo = [&](int l) -> long long {
            if (!seg) return (long long)(r0 + l) * N;
            const int st = (s0 + l) * seg;
            return (long long)r0 * N + (pass == 0 ? st - warm : st);
        };
        auto vlo = [&](int l) -> int {   // first virtual index that exists
            if (!seg || pass == 1) return 0;
            const int st = (s0 + l) * seg;
            return st < warm ? warm - st : 0;
        };
        auto vn = [&](int l) -> int {    // one past the last virtual index that exists
            if (!seg) return N;
            if (pass == 0) return warm;
            const int left = N - (s0 + l) * seg;
            return left < seg ? (left > 0 ? left : 0) : seg;
        };
        const int span = seg ? (pass == 0 ? warm : seg) : N;
        const int my_n = lane < nr ? vn(lane) : 0;
        for (int i0 = 0; i0 < span; i0 += ROWS_CH) {
            constexpr int LG = NS <= 3 ? 32 : 16;   // loads in flight before the first LDS write: the whole tile when
            for (int l0 = 0; l0 < nr; l0 += LG) {   // the stage states leave room (one exposed memory latency per tile)
                float v[LG];
#pragma unroll
                for (int u = 0; u < LG; ++u) {
                    const int l = l0 + u;
                    const int i = i0 + lane;
                    v[u] = (l < nr && i < vn(l) && i >= vlo(l)) ? rows[vo(l) + i] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < LG; ++u) tile[lane][l0 + u] = v[u];   // columns nr.. of the last group: zeros
            }
            __syncthreads();
            int cnt = my_n - i0;                       // this lane's live samples in the tile (metrics, AGC)
            if (cnt > ROWS_CH) cnt = ROWS_CH;
            const int cntu = span - i0 < ROWS_CH ? span - i0 : ROWS_CH;   // wave-uniform walk
            {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    // time-parallel form of the short chains: fused updates (see iir_run; longer chains keep one code
                    // path -- both would not unroll, and z[s] must stay in registers)
                    if (NS <= 3 && seg) {
                        if (sa.st[s].is_f64) iir_stage<true, true, true>(tile, cntu, lane, z[s], st_s[s]);
                        else iir_stage<false, true, true>(tile, cntu, lane, z[s], st_s[s]);
                    } else {
                        if (sa.st[s].is_f64) iir_stage<true, true>(tile, cntu, lane, z[s], st_s[s]);
                        else iir_stage<false, true>(tile, cntu, lane, z[s], st_s[s]);
                    }
                }
                if (agc.on) {
                    for (int j = 0; j < cnt; ++j) {
                        float x = tile[j][lane];
                        float ax = fabsf(x);
                        float ya = __fadd_rn(za, __fmul_rn(agc.att_b0, ax));
                        za = __fsub_rn(__fmul_rn(ax, 0.0f), __fmul_rn(ya, agc.att_a1));
                        float yr = __fadd_rn(zr, __fmul_rn(agc.rel_b0, ya));
                        zr = __fsub_rn(__fmul_rn(ya, 0.0f), __fmul_rn(yr, agc.rel_a1));
                        float env = fmaxf(ya, yr);
                        float g = __fdiv_rn(agc.target, fmaxf(env, (float)1e-6));
                        g = fminf(g, agc.max_gain);
                        float y = __fmul_rn(x, g);
                        x = tanhf(y * 1.5f) * NORM;
                        tile[j][lane] = x;
                        ss += (double)x * (double)x;
                    }
                } else if (pass == 1) {
                    for (int j = 0; j < cnt; ++j) {
                        float x = tile[j][lane];
                        ss += (double)x * (double)x;
                    }
                }
            }
            __syncthreads();
            if (pass == 1)
                for (int l = 0; l < nr; ++l) {
                    const int c = vn(l) - i0;
                    if (lane < c) rows[vo(l) + i0 + lane] = tile[lane][l];
                }
            __syncthreads();
        }
    }
    if (seg) {
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        if (lane == 0) red_s[wave] = ss;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red_s[w];
            // (two workgroups per row: the slot was left at zero for them -- FmArgs::skip_fm_sum --, and a + b == b + a)
            if (ngrp > 1) atomicAdd(&acc[(size_t)r0 * 2 + 1], t);
            else acc[(size_t)r0 * 2 + 1] = t;
        }
    } else if (lane < nr) {
        acc[(size_t)(r0 + lane) * 2 + 1] = ss;
    }
}

// ---- exact time-parallel form of the rows (linear-recurrence scan) ---------------------------------------------
// For chains the warm-up trick cannot serve (poles too close to the unit circle for the chunk, or the AGC, whose
// release constant spans the chunk): a wave owns one row cut into 64 segments and every stage runs in three steps --
//   1. each lane runs the stage over its segment from ZERO state and keeps only the end state e_p;
//   2. the true start states follow from s_{p+1} = M s_p + e_p, M = (state transition)^segment, 63 small mat-vecs;
//   3. each lane re-runs the stage over its segment from s_p and writes the outputs --
// the same DF2T operations as the sequential kernel in step 3; only the start states carry the rounding of the
// scan instead of that of the sequential walk.  M is produced at bank creation by the recurrence itself on the unit
// state vectors (iir_pow_kernel).  The AGC's two one-pole envelopes (float32, linear in |x|) take the same treatment:
// attack envelope scan, release envelope scan, then the gain / tanh pass.
struct ScanPow {
    double m[MAX_STAGES][MAX_ORD - 1][MAX_ORD - 1];   // per stage: M[i][j]
    double att, rel;                                  // (-a1)^segment of the two AGC one-poles
};

__global__ void iir_pow_kernel(StageArr sa, int n_stages, AgcDev agc, int seg, ScanPow *out) {
    // thread (s, j): column j of stage s: state after `seg` zero-input steps from the unit vector e_j
    const int s = blockIdx.x, j = threadIdx.x;
    if (s < n_stages) {
        const StageDev &S = sa.st[s];
        const int m = S.n - 1;
        if (j < m) {
            double z[MAX_ORD - 1];
            for (int k = 0; k < MAX_ORD - 1; ++k) z[k] = k == j ? 1.0 : 0.0;
            for (int t = 0; t < seg; ++t) {
                if (S.is_f64) {
                    const double y = z[0];
                    for (int k = 0; k < m - 1; ++k) z[k] = __dsub_rn(z[k + 1], __dmul_rn(y, S.a[k + 1]));
                    z[m - 1] = -__dmul_rn(y, S.a[m]);
                } else {
                    const float y = (float)z[0];
                    for (int k = 0; k < m - 1; ++k) z[k] = (double)__fsub_rn((float)z[k + 1], __fmul_rn(y, (float)S.a[k + 1]));
                    z[m - 1] = (double)(-__fmul_rn(y, (float)S.a[m]));
                }
            }
            for (int i = 0; i < MAX_ORD - 1; ++i) out->m[s][i][j] = i < m ? z[i] : 0.0;
        }
    } else if (j == 0) {
        float za = 1.0f, zr = 1.0f;   // z' = -a1 * y, y = z (zero input)
        for (int t = 0; t < seg; ++t) {
            za = -__fmul_rn(za, agc.att_a1);
            zr = -__fmul_rn(zr, agc.rel_a1);
        }
        out->att = (double)za;
        out->rel = (double)zr;
    }
}

// Start states of a wave's 64 segments from their zero-state end states e_p (in z): s_0 = start, s_{p+1} = M s_p + e_p.
// On return z holds the lane's start state and `start` the state after the wave's last segment (the next wave's start
// when this wave started from `start`; with start = 0 the wave's own zero-state end state).
template <int NC>
__device__ __forceinline__ void scan_states(double (&z)[MAX_ORD - 1], const double (*M)[MAX_ORD - 1], int lane, bool f64,
                                            double (&start)[MAX_ORD - 1]) {
    constexpr int m = NC - 1;
    double cur[m > 0 ? m : 1], mine[m > 0 ? m : 1];
#pragma unroll
    for (int k = 0; k < m; ++k) { cur[k] = start[k]; mine[k] = start[k]; }
    for (int p = 0; p < 64; ++p) {
        double nxt[m > 0 ? m : 1];
#pragma unroll
        for (int i = 0; i < m; ++i) {
            double acc = __shfl(z[i], p);
#pragma unroll
            for (int j = 0; j < m; ++j) acc = fma(M[i][j], cur[j], acc);
            nxt[i] = f64 ? acc : (double)(float)acc;
        }
#pragma unroll
        for (int i = 0; i < m; ++i) {
            cur[i] = nxt[i];
            if (lane == p + 1) mine[i] = nxt[i];
        }
    }
#pragma unroll
    for (int k = 0; k < m; ++k) { z[k] = mine[k]; start[k] = cur[k]; }
}

__device__ __forceinline__ void scan_stage_states(double (&z)[MAX_ORD - 1], const double (*M)[MAX_ORD - 1], int n, int lane,
                                                  bool f64, double (&start)[MAX_ORD - 1]) {
    switch (n) {
        case 2: scan_states<2>(z, M, lane, f64, start); break;
        case 3: scan_states<3>(z, M, lane, f64, start); break;
        case 4: scan_states<4>(z, M, lane, f64, start); break;
        case 5: scan_states<5>(z, M, lane, f64, start); break;
        case 6: scan_states<6>(z, M, lane, f64, start); break;
        case 7: scan_states<7>(z, M, lane, f64, start); break;
        case 8: scan_states<8>(z, M, lane, f64, start); break;
        case 9: scan_states<9>(z, M, lane, f64, start); break;
        case 10: scan_states<10>(z, M, lane, f64, start); break;
        case 11: scan_states<11>(z, M, lane, f64, start); break;
        default: break;   // n == 1: no state
    }
}

// scalar affine scan for a one-pole: s_0 = start, s_{p+1} = pw * s_p + e_p (float64, rounded to float32 each step);
// returns the lane's start value, `start` becomes the value after the wave's last segment
__device__ __forceinline__ float scan_scalar(float e, double pw, int lane, double &start) {
    double cur = start;
    float mine = (float)start;
    for (int p = 0; p < 64; ++p) {
        cur = (double)(float)fma(pw, cur, (double)__shfl(e, p));
        if (lane == p + 1) mine = (float)cur;
    }
    start = cur;
    return mine;
}

// A row is cut into 64 W segments, W = blockDim.x / 64 waves (the host picks W from the bank's channel count and chunk
// length only).  With W > 1 a scan runs twice: from zero (the wave's own end state E_w goes to LDS), then -- after the
// wave's true start W_w = fold of E_0 .. E_{w-1} through the 64-segment transition (pw[1]) -- from W_w.
__global__ __launch_bounds__(64 * ROWS_MAXW) void chan_rows_scan_kernel(float *rows, double *acc, int N, StageArr sa,
                                                                       int n_stages, AgcDev agc, int seg, const ScanPow *pw) {
    extern __shared__ __attribute__((aligned(16))) float tile_raw[];   // per wave [ROWS_CH][65]
    __shared__ StageDev st_s[MAX_STAGES];
    __shared__ double M_s[2][MAX_ORD - 1][MAX_ORD - 1];   // segment transition, 64-segment transition
    __shared__ double exch[ROWS_MAXW][MAX_ORD - 1];
    __shared__ double red_s[ROWS_MAXW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    float (*tile)[65] = reinterpret_cast<float (*)[65]>(tile_raw + (size_t)wave * ROWS_CH * 65);
    for (int i = threadIdx.x; i < n_stages * (int)(sizeof(StageDev) / 4); i += blockDim.x)
        reinterpret_cast<int *>(st_s)[i] = reinterpret_cast<const int *>(sa.st)[i];
    __syncthreads();
    float *row = rows + (size_t)blockIdx.x * N;
    const int s0 = wave * 64;                       // first segment of this wave
    int nr = (N + seg - 1) / seg - s0;              // live segments of this wave
    nr = nr < 0 ? 0 : (nr > 64 ? 64 : nr);
    auto vn = [&](int l) -> int {
        const int left = N - (s0 + l) * seg;
        return left < seg ? (left > 0 ? left : 0) : seg;
    };
    const int my_n = lane < nr ? vn(lane) : 0;
    // stream the segment through the tile; body(cnt) works on tile[0..cnt)[lane]
    auto stream = [&](bool write, auto body) {
        for (int i0 = 0; i0 < seg; i0 += ROWS_CH) {
            for (int l0 = 0; l0 < nr; l0 += 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int l = l0 + u;
                    v[u] = (l < nr && lane < vn(l) - i0) ? row[(size_t)(s0 + l) * seg + i0 + lane] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (l0 + u < nr) tile[lane][l0 + u] = v[u];
            }
            __syncthreads();
            int cnt = my_n - i0;
            if (cnt > ROWS_CH) cnt = ROWS_CH;
            if (cnt > 0) body(cnt);
            __syncthreads();
            if (write)
                for (int l = 0; l < nr; ++l)
                    if (lane < vn(l) - i0) row[(size_t)(s0 + l) * seg + i0 + lane] = tile[lane][l];
            __syncthreads();
        }
    };
    double ss = 0.0;
    for (int s = 0; s < n_stages; ++s) {
        const bool f64 = st_s[s].is_f64 != 0;
        const int n = __builtin_amdgcn_readfirstlane(st_s[s].n);
        for (int i = threadIdx.x; i < (MAX_ORD - 1) * (MAX_ORD - 1); i += blockDim.x) {
            (&M_s[0][0][0])[i] = (&pw[0].m[s][0][0])[i];
            (&M_s[1][0][0])[i] = (&pw[1].m[s][0][0])[i];
        }
        double z[MAX_ORD - 1];
#pragma unroll
        for (int k = 0; k < MAX_ORD - 1; ++k) z[k] = 0.0;
        auto run = [&](int cnt) {
            if (f64) iir_stage<true>(tile, cnt, lane, z, st_s[s]);
            else iir_stage<false>(tile, cnt, lane, z, st_s[s]);
        };
        stream(false, run);                       // 1. zero-state end states
        __syncthreads();
        double start[MAX_ORD - 1];
#pragma unroll
        for (int k = 0; k < MAX_ORD - 1; ++k) start[k] = 0.0;
        if (W > 1) {                              // 2a. the wave's start state
            double e[MAX_ORD - 1];
#pragma unroll
            for (int k = 0; k < MAX_ORD - 1; ++k) e[k] = z[k];
            scan_stage_states(e, M_s[0], n, lane, f64, start);
            if (lane == 0)
#pragma unroll
                for (int k = 0; k < MAX_ORD - 1; ++k) exch[wave][k] = start[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < MAX_ORD - 1; ++k) start[k] = 0.0;
            for (int w = 0; w < wave; ++w) {
                double nxt[MAX_ORD - 1];
#pragma unroll
                for (int i = 0; i < MAX_ORD - 1; ++i) {
                    double a2 = exch[w][i];
#pragma unroll
                    for (int j = 0; j < MAX_ORD - 1; ++j) a2 = fma(M_s[1][i][j], start[j], a2);
                    nxt[i] = f64 ? a2 : (double)(float)a2;
                }
#pragma unroll
                for (int i = 0; i < MAX_ORD - 1; ++i) start[i] = nxt[i];
            }
        }
        scan_stage_states(z, M_s[0], n, lane, f64, start);   // 2. true start states
        const bool last = (s == n_stages - 1) && !agc.on;
        stream(true, [&](int cnt) {               // 3. the outputs
            run(cnt);
            if (last)
                for (int j = 0; j < cnt; ++j) {
                    float x = tile[j][lane];
                    ss += (double)x * (double)x;
                }
        });
        __syncthreads();
    }
    if (agc.on) {
        const float NORM = (float)(1.0 / 0.90514825364486640);
        float za = 0.f, zr = 0.f;
        // scalar two-level scan: e = the lane's zero-state end value, pws / pww = one-pole transition over a segment / 64
        auto scan2 = [&](float e, double pws, double pww) -> float {
            double start = 0.0;
            if (W > 1) {
                (void)scan_scalar(e, pws, lane, start);
                if (lane == 0) exch[wave][0] = start;
                __syncthreads();
                start = 0.0;
                for (int w = 0; w < wave; ++w) start = (double)(float)fma(pww, start, exch[w][0]);
                __syncthreads();
            }
            return scan_scalar(e, pws, lane, start);
        };
        // attack envelope: zero-state end value, scan
        stream(false, [&](int cnt) {
            for (int j = 0; j < cnt; ++j) {
                float ya = __fadd_rn(za, __fmul_rn(agc.att_b0, fabsf(tile[j][lane])));
                za = __fsub_rn(0.0f, __fmul_rn(ya, agc.att_a1));
            }
        });
        const float za0 = scan2(za, pw[0].att, pw[1].att);
        // release envelope fed by the true attack envelope: zero-state end value, scan
        za = za0;
        stream(false, [&](int cnt) {
            for (int j = 0; j < cnt; ++j) {
                float ax = fabsf(tile[j][lane]);
                float ya = __fadd_rn(za, __fmul_rn(agc.att_b0, ax));
                za = __fsub_rn(__fmul_rn(ax, 0.0f), __fmul_rn(ya, agc.att_a1));
                float yr = __fadd_rn(zr, __fmul_rn(agc.rel_b0, ya));
                zr = __fsub_rn(__fmul_rn(ya, 0.0f), __fmul_rn(yr, agc.rel_a1));
            }
        });
        const float zr0 = scan2(zr, pw[0].rel, pw[1].rel);
        za = za0;
        zr = zr0;
        stream(true, [&](int cnt) {
            for (int j = 0; j < cnt; ++j) {
                float x = tile[j][lane];
                float ax = fabsf(x);
                float ya = __fadd_rn(za, __fmul_rn(agc.att_b0, ax));
                za = __fsub_rn(__fmul_rn(ax, 0.0f), __fmul_rn(ya, agc.att_a1));
                float yr = __fadd_rn(zr, __fmul_rn(agc.rel_b0, ya));
                zr = __fsub_rn(__fmul_rn(ya, 0.0f), __fmul_rn(yr, agc.rel_a1));
                float env = fmaxf(ya, yr);
                float g = __fdiv_rn(agc.target, fmaxf(env, (float)1e-6));
                g = fminf(g, agc.max_gain);
                float y = __fmul_rn(x, g);
                x = tanhf(y * 1.5f) * NORM;
                tile[j][lane] = x;
                ss += (double)x * (double)x;
            }
        });
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) red_s[wave] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < W; ++w) t += red_s[w];
        acc[(size_t)blockIdx.x * 2 + 1] = t;
    }
}

// finalize.  post 0 (FM): scale by 0.18/rms (dsp/fm.py:42-62) + soft clip x0.95 (fm.py:26-39);
// post 1 (AM/SSB with AGC): audio as is;  post 2 (AM/SSB without AGC): agc.soft_clip (agc.py:58-70)
// wire != 0: the audio additionally leaves in its wire format (N4): 1 = int16 PCM by the pack_pcm16 rule (capture.py:119-131:
// clip to [-1, 1], x 32767, truncate), 2 = float32 clipped to [-1, 1] (pack_f32, capture.py:134-144), written to wire_out
// [rows][n_out] -- what the caller sends on needs no second pass and only that buffer has to cross PCIe.
__global__ __launch_bounds__(256) void fmbank_finalize_kernel(float *audio, const double *acc, float *metrics,
                                                              int N, int n_fm, int n_out, int post,
                                                              const float *squelch_db, int K, int wire, void *wire_out,
                                                              int shared_base) {
    const size_t row = blockIdx.x;
    // shared_base: sum |base|^2 sits in channel 0's slot of the chunk (fused FM kernel: the power of the mixed chunk is the
    // power of the raw chunk, the same for every channel)
    const double base_pw = acc[(shared_base ? row - row % K : row) * 2];
    float *au = audio + row * n_out;
    // squelch (capture.py:2918-2921): rssi below the channel's threshold -> zeros (metrics keep the unsquelched power)
    bool squelched = false;
    if (squelch_db) {
        const float sq = squelch_db[row % K];
        const float rssi = (float)(10.0 * log10(base_pw / (double)N + 1e-10));
        squelched = sq == sq && rssi < sq;   // NaN = no squelch configured
    }
    short *w16 = wire == 1 ? reinterpret_cast<short *>(wire_out) + row * n_out : nullptr;
    float *w32 = wire == 2 ? reinterpret_cast<float *>(wire_out) + row * n_out : nullptr;
    const double *ac = acc + row * 2;
    float s = 1.0f;
    if (post == 0) {
        const float mean_fm = (float)(ac[1] / (double)n_fm);   // n_fm < N after spectral noise reduction
        const float rms = sqrtf(mean_fm);
        if ((double)rms > 1e-4) s = (float)(0.18 / (double)rms);
    }
    const float NORM = (float)(1.0 / 0.90514825364486640);  // 1/tanh(1.5)
    float p = 0.f, mx = 0.f;
    int bad = 0;
    for (int i = threadIdx.x; i < n_out; i += 256) {
        float v = au[i];
        if (post == 0) v = tanhf((v * s) * 1.5f) * NORM * 0.95f;
        else if (post == 2) v = tanhf(v * 1.5f) * NORM;
        p += v * v;
        mx = fmaxf(mx, fabsf(v));
        if (!(fabsf(v) <= 3.0e38f)) bad = 1;
        if (squelched) v = 0.0f;
        au[i] = v;
        if (w16) w16[i] = pack1(v);
        if (w32) w32[i] = v != v ? v : fminf(fmaxf(v, -1.0f), 1.0f);   // np.clip keeps NaN
    }
    __shared__ float rp[4], rm[4];
    __shared__ int rb[4];
    for (int o = 32; o > 0; o >>= 1) {
        p += __shfl_xor(p, o);
        mx = fmaxf(mx, __shfl_xor(mx, o));
        bad |= __shfl_xor(bad, o);
    }
    if ((threadIdx.x & 63) == 0) {
        rp[threadIdx.x >> 6] = p;
        rm[threadIdx.x >> 6] = mx;
        rb[threadIdx.x >> 6] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        p = rp[0] + rp[1] + rp[2] + rp[3];
        mx = fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3]));
        bad = rb[0] | rb[1] | rb[2] | rb[3];
        float *m = metrics + row * 4;
        m[0] = (float)(10.0 * log10(base_pw / (double)N + 1e-10));
        m[1] = (float)(10.0 * log10((double)(p / (float)n_out) + 1e-10));
        m[2] = mx;
        m[3] = bad ? 0.f : 1.f;
    }
}

// exact k-th smallest of a row of non-negative floats (np.partition semantics, capture.py:781-788):
// 3-pass radix select on the IEEE bit patterns (monotonic for x >= 0): 11 + 11 + 10 bits, LDS histograms.
__global__ __launch_bounds__(256) void select_kth_kernel(const float *rows, int N, int k_lo, int k_hi, float *out) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned s_prefix, s_rank;
    const int row = blockIdx.x, which = blockIdx.y, tid = threadIdx.x;
    const unsigned *r = reinterpret_cast<const unsigned *>(rows + (size_t)row * N);
    unsigned prefix = 0, rank = (unsigned)(which == 0 ? k_lo : k_hi);
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int sh = shifts[pass], nb = 1 << widths[pass];
        for (int i = tid; i < 2048; i += 256) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < N; i += 256) {
            unsigned b = r[i] & 0x7fffffffu;    // |x| (inputs are magnitudes; -0 -> +0)
            bool match = pass == 0 || (b >> (sh + widths[pass])) == prefix;
            if (match) atomicAdd(&hist[(b >> sh) & (nb - 1)], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned acc = 0;
            int bin = 0;
            for (; bin < nb; ++bin) {
                if (acc + hist[bin] > rank) break;
                acc += hist[bin];
            }
            s_prefix = (prefix << widths[pass]) | (unsigned)bin;
            s_rank = rank - acc;
        }
        __syncthreads();
        prefix = s_prefix;
        rank = s_rank;
        __syncthreads();
    }
    if (tid == 0) out[row * 2 + which] = __uint_as_float(prefix);
}

__global__ void metrics_rssi_kernel(const double *acc, int rows, int N, float *out) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) out[r] = (float)(10.0 * log10(acc[(size_t)r * 2] / (double)N + 1e-10));
}

inline unsigned grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g > 4096 ? 4096 : (g ? g : 1));
}

}  // namespace

extern "C" int wh_unpack_i16_cf32(const int16_t *d_in, float *d_out, size_t n, void *stream) {
    if (n == 0) return WH_OK;
    if (!d_in || !d_out) return set_err(WH_E_ARG, "wh_unpack_i16_cf32: null buffer");
    hipLaunchKernelGGL(unpack_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const short2 *>(d_in), reinterpret_cast<float2 *>(d_out), n);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_pack_cf32_i16(const float *d_in, int16_t *d_out, size_t n, void *stream) {
    if (n == 0) return WH_OK;
    if (!d_in || !d_out) return set_err(WH_E_ARG, "wh_pack_cf32_i16: null buffer");
    hipLaunchKernelGGL(pack_iq_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2 *>(d_in), reinterpret_cast<short2 *>(d_out), n);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_pack_f32_pcm16(const float *d_in, int16_t *d_out, size_t n, void *stream) {
    if (n == 0) return WH_OK;
    if (!d_in || !d_out) return set_err(WH_E_ARG, "wh_pack_f32_pcm16: null buffer");
    hipLaunchKernelGGL(pack_pcm_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), d_in,
                       reinterpret_cast<short *>(d_out), n);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

static float nco_const(int offset_hz, int sample_rate) {
    // f32(-2 pi (off/fs)) : the Python complex scalar is weak -> cast to complex64 (capture.py:173-176)
    return (float)(-2.0 * M_PI * ((double)offset_hz / (double)sample_rate));
}

extern "C" int wh_nco_mix(const float *d_iq, float *d_out, size_t n, int offset_hz, int sample_rate, void *stream) {
    if (n == 0) return WH_OK;
    if (!d_iq || !d_out || sample_rate <= 0) return set_err(WH_E_ARG, "wh_nco_mix: bad args");
    if (n > (size_t)1 << 24) return set_err(WH_E_ARG, "wh_nco_mix: n > 2^24 (float32 ramp is not exact beyond)");
    if (offset_hz == 0) {
        if (d_out != d_iq)
            WH_HIP(hipMemcpyAsync(d_out, d_iq, n * sizeof(float2), hipMemcpyDeviceToDevice, as_stream(stream)));
        return WH_OK;
    }
    hipLaunchKernelGGL(nco_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2 *>(d_iq), reinterpret_cast<float2 *>(d_out), n,
                       nco_const(offset_hz, sample_rate));
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_fm_discriminate(const float *d_iq, float *d_out, size_t n, int sample_rate, void *stream) {
    if (n == 0) return WH_OK;
    if (!d_iq || !d_out) return set_err(WH_E_ARG, "wh_fm_discriminate: null buffer");
    float scale = (float)((double)sample_rate / (2.0 * M_PI * 75000.0));
    hipLaunchKernelGGL(disc_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2 *>(d_iq), d_out, n, scale);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

struct wh_resampler {
    double *d_taps = nullptr;
    int ntaps, up, down, d0;
};

extern "C" int wh_resampler_create(wh_resampler **out, const double *h_taps, int ntaps, int up, int down, int d0) {
    if (!out || !h_taps || ntaps < 1 || up < 1 || down < 1) return set_err(WH_E_ARG, "wh_resampler_create: bad args");
    wh_resampler *r = new wh_resampler();
    std::unique_ptr<wh_resampler, void (*)(wh_resampler *)> guard(r, wh_resampler_destroy);  // frees partial state on early return
    r->ntaps = ntaps; r->up = up; r->down = down; r->d0 = d0;
    WH_HIP(hipMalloc(&r->d_taps, (size_t)ntaps * sizeof(double)));
    WH_HIP(hipMemcpy(r->d_taps, h_taps, (size_t)ntaps * sizeof(double), hipMemcpyHostToDevice));
    *out = guard.release();
    return WH_OK;
}

extern "C" void wh_resampler_destroy(wh_resampler *r) {
    if (!r) return;
    (void)hipFree(r->d_taps);
    delete r;
}

static int launch_resample(const float *d_x, size_t n_in, size_t x_stride, size_t batch, float *d_y, size_t n_out,
                           const double *taps, int ntaps, int up, int down, int d0, hipStream_t st) {
    if (batch == 0 || n_out == 0) return WH_OK;
    for (size_t b0 = 0; b0 < batch; b0 += 65535) {
        size_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
        hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n_out + 3) / 4), (unsigned)nb), dim3(256), 0, st,
                           d_x + b0 * x_stride, n_in, x_stride, d_y + b0 * n_out, n_out, taps, ntaps, up, down, d0);
        WH_LAUNCH_CHECK();
    }
    return WH_OK;
}

extern "C" int wh_resampler_run(wh_resampler *r, const float *d_x, size_t n_in, size_t batch, float *d_y,
                                size_t n_out, void *stream) {
    if (!r || !d_x || !d_y) return set_err(WH_E_ARG, "wh_resampler_run: null");
    return launch_resample(d_x, n_in, n_in, batch, d_y, n_out, r->d_taps, r->ntaps, r->up, r->down, r->d0, as_stream(stream));
}


// ---- spectral noise reduction (dsp/filters.py:346-460), default-off stage of the FM chains ------------
// STFT 1024 / hop 512 with the float32 periodic Hann window, per-bin noise floor = 10th percentile of the
// magnitudes over the frames of the chunk, Wiener-like gain max(1 - (floor*10^(dB/20)/|X|)^2, 0.1), inverse
// FFT, windowed overlap-add divided by the summed squared window.  The output is (n_frames-1)*512+1024
// samples long -- SHORTER than the input when the chunk is not a whole number of hops (the reference
// returns output[:len(x)] of a buffer that is only padded_length long).
constexpr int NR_FFT = 1024, NR_HOP = 512, NR_BINS = 513;

// LDS autosort Stockham FFT, 1024 points = five radix-4 passes, 256 threads (one butterfly each per pass);
// sign = -1 forward, +1 inverse (unscaled).  Returns the buffer holding the result.
__device__ __forceinline__ float2 *nr_fft1024(float2 *src, float2 *dst, const float2 *tw, float sign) {
    int n = NR_FFT, s = 1;
    for (int st = 0; st < 5; ++st) {
        const int m = n >> 2;
        const int i = threadIdx.x;
        const int pp = i / s, q = i - pp * s;
        const float2 a = src[q + s * pp];
        const float2 b = src[q + s * (pp + m)];
        const float2 c = src[q + s * (pp + 2 * m)];
        const float2 d = src[q + s * (pp + 3 * m)];
        float2 w1 = tw[(size_t)pp * s], w2 = tw[(size_t)2 * pp * s], w3 = tw[(size_t)3 * pp * s];
        w1.y *= -sign; w2.y *= -sign; w3.y *= -sign;   // table holds exp(-2 pi i k / N)
        const float2 apc = make_float2(a.x + c.x, a.y + c.y), amc = make_float2(a.x - c.x, a.y - c.y);
        const float2 bpd = make_float2(b.x + d.x, b.y + d.y), bmd = make_float2(b.x - d.x, b.y - d.y);
        const float2 r = make_float2(-sign * bmd.y, sign * bmd.x);   // forward (sign -1): -j (b - d); inverse: +j (b - d)
        const float2 y1 = make_float2(amc.x + r.x, amc.y + r.y), y3 = make_float2(amc.x - r.x, amc.y - r.y);
        const float2 y2 = make_float2(apc.x - bpd.x, apc.y - bpd.y);
        dst[q + s * (4 * pp)] = make_float2(apc.x + bpd.x, apc.y + bpd.y);
        dst[q + s * (4 * pp + 1)] = make_float2(y1.x * w1.x - y1.y * w1.y, y1.x * w1.y + y1.y * w1.x);
        dst[q + s * (4 * pp + 2)] = make_float2(y2.x * w2.x - y2.y * w2.y, y2.x * w2.y + y2.y * w2.x);
        dst[q + s * (4 * pp + 3)] = make_float2(y3.x * w3.x - y3.y * w3.y, y3.x * w3.y + y3.y * w3.x);
        __syncthreads();
        float2 *t = src; src = dst; dst = t;
        n = m; s <<= 2;
    }
    return src;
}

__global__ __launch_bounds__(256) void nr_stft_kernel(const float *rows, int N, int F, const float *window,
                                                      const float2 *tw, float2 *stft, float *magT) {
    __shared__ float2 buf[2][NR_FFT];
    const int f = blockIdx.x, r = blockIdx.y;
    const float *x = rows + (size_t)r * N + (size_t)f * NR_HOP;
    for (int t = threadIdx.x; t < NR_FFT; t += 256) buf[0][t] = make_float2(x[t] * window[t], 0.0f);
    __syncthreads();
    const float2 *X = nr_fft1024(buf[0], buf[1], tw, -1.0f);
    for (int b = threadIdx.x; b < NR_BINS; b += 256) {
        float2 v = X[b];
        stft[((size_t)r * F + f) * NR_BINS + b] = v;
        magT[((size_t)r * NR_BINS + b) * F + f] = hypotf(v.x, v.y);
    }
}

__global__ __launch_bounds__(256) void nr_istft_kernel(float *rows, int N, int F, const float *window, const float2 *tw,
                                                       const float2 *stft, const float *sel, int k_lo, float gamma,
                                                       float lin) {
    __shared__ float2 buf[2][NR_FFT];
    const int f = blockIdx.x, r = blockIdx.y;
    for (int b = threadIdx.x; b < NR_BINS; b += 256) {
        float2 v = stft[((size_t)r * F + f) * NR_BINS + b];
        const float *q = sel + ((size_t)r * NR_BINS + b) * 2;
        float lo = q[0], hi = q[1];
        float d = hi - lo;   // np.percentile 'linear': a + (b-a) t, evaluated from b when t >= 0.5
        float nf = gamma >= 0.5f ? hi - d * (1.0f - gamma) : lo + d * gamma;
        float mag = hypotf(v.x, v.y);
        float ratio = (nf * lin) / fmaxf(mag, 1e-10f);
        float g = fmaxf(fmaxf(0.0f, 1.0f - ratio * ratio), 0.1f);
        float2 y = make_float2(v.x * g, v.y * g);   // = |X| g exp(j arg X)
        if (b == 0 || b == NR_FFT / 2) y.y = 0.0f;    // c2r ignores the imaginary part of DC / Nyquist
        buf[0][b] = y;
        if (b > 0 && b < NR_FFT / 2) buf[0][NR_FFT - b] = make_float2(y.x, -y.y);
    }
    __syncthreads();
    const float2 *xt = nr_fft1024(buf[0], buf[1], tw, 1.0f);
    float *o = rows + (size_t)r * N + (size_t)f * NR_HOP;
    for (int t = threadIdx.x; t < NR_FFT; t += 256) {
        float v = xt[t].x * (1.0f / NR_FFT);
        atomicAdd(o + t, v * window[t]);   // <= 2 contributions per sample: float add is commutative, so deterministic
    }
}

__global__ __launch_bounds__(256) void nr_norm_kernel(float *rows, int N, int F, const float *window, double *acc) {
    const int r = blockIdx.x;
    float *o = rows + (size_t)r * N;
    const int L = (F - 1) * NR_HOP + NR_FFT;
    double ss = 0.0;
    for (int t = threadIdx.x; t < L; t += 256) {
        int f1 = t / NR_HOP, i1 = t - f1 * NR_HOP;
        float ws = 0.0f;
        if (f1 >= 1) { float w = window[i1 + NR_HOP]; ws += w * w; }
        if (f1 < F) { float w = window[i1]; ws += w * w; }
        float v = o[t] / fmaxf(ws, 1e-10f);
        o[t] = v;
        ss += (double)v * (double)v;
    }
    __shared__ double red[256];
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[(size_t)r * 2 + 1] = red[0];
}

// tiles per workgroup of the fused FM kernel: as long as the launch keeps >= ~4096 workgroups (16 per CU) longer
// runs only save work (less window overlap recomputed); a single live chunk of a few channels stays at one tile per
// workgroup.  Tiles are spread evenly over the runs.
static int run_tiles(int tiles, size_t rows) {
    size_t total = (size_t)tiles * rows;
    int R = (int)(total / 4096 < 1 ? 1 : (total / 4096 > (size_t)tiles ? (size_t)tiles : total / 4096));
    if (R > tiles) R = tiles;
    if (R < 1) R = 1;
    const int runs = (tiles + R - 1) / R;
    return (tiles + runs - 1) / runs;
}

// two order statistics of many SHORT rows (the per-bin noise floor over the frames of a chunk: hundreds of values per
// row, hundreds of thousands of rows): one wave per row, values in LDS, every lane ranks its candidates by counting
// (#smaller + #equal-with-lower-index gives each value a unique rank), the lanes holding ranks k_lo / k_hi write them.
constexpr int RANK_MAX = 2048;

__global__ __launch_bounds__(256) void rank_select_kernel(const float *rows, int n_rows, int F, int k_lo, int k_hi, float *out) {
    extern __shared__ float rv[];   // [4][F]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= n_rows) return;
    float *v = rv + (size_t)wave * F;
    const float *src = rows + (size_t)row * F;
    for (int i = lane; i < F; i += 64) v[i] = src[i];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int i = lane; i < F; i += 64) {
        const float x = v[i];
        int rank = 0;
        for (int m = 0; m < F; ++m) {
            const float y = v[m];
            rank += (y < x || (y == x && m < i)) ? 1 : 0;
        }
        if (rank == k_lo) out[(size_t)row * 2] = x;
        if (rank == k_hi) out[(size_t)row * 2 + 1] = x;
    }
}

// Shape of the time-parallel rows: rw waves per workgroup (64 segments each) and rg workgroups per row -- a property of the
// bank (channel count, chunk length, chain), never of the call, so a bank's numbers do not depend on how many chunks a call
// carries.  Chains of more than 3 stages keep one wave per row (their state alone is 20 VGPRs per stage).  Banks of one or
// two channels whose start states come from the impulse response (no warm-up walk per segment) cut the row into 1024
// segments, two workgroups of 8 waves: such launches are pure latency on an idle chip.
static void rows_shape(const wh_chanbank_cfg &c, bool fir_start, int *rw_out, int *rg_out) {
    int rw = c.n_stages > 3 ? 1 : c.n_channels <= 16 ? ROWS_MAXW : c.n_channels <= 64 ? 4 : 1;
    while (rw > 1 && c.chunk_len < 64 * rw * 16) rw >>= 1;
    int rg = (fir_start && rw == ROWS_MAXW && c.n_channels <= 2 && c.chunk_len >= 64 * rw * 2 * 16) ? 2 : 1;
    *rw_out = rw;
    *rg_out = rg;
}

struct wh_chanbank {
    wh_chanbank_cfg cfg;
    float *d_nco = nullptr;
    float *d_squelch = nullptr;
    std::vector<float> h_stage[2];   // host side of wh_chanbank_set_offsets / _set_squelch (alternating, kept alive)
    int h_stage_cur = 0;
    double *d_taps = nullptr;
    StageArr stages{};
    ScanPow *d_pow = nullptr;   // scan form: per-stage state-transition powers for the bank's segment length
    int scan_seg = 0, scan_w = 1;   // scan form: segment length, waves per row (64 segments each)
    double *d_G = nullptr;          // warm-up form: impulse-response states [iir_warmup][start_ep]
    double *d_S0 = nullptr;         // ... and the segments' start states of a call [rows][segments][start_ep]
    int start_ep = 0;               // row length of d_G / d_S0: the chain's state count rounded up to even
    double *d_acc = nullptr;
    float *d_fm = nullptr;
    float *d_araw = nullptr;   // fused FM path: raw discriminator angles of the call's chunks [n_chunks][N]
    size_t cap_chunks = 0;
    bool fused = false;
    bool rows_fir = false;   // unfused chain whose resampler is the fused kernel's decimating FIR
    int fir_phase = 0;       // the FIR's polyphase register form (see fm_fir_phases)
    float2 *d_tphase = nullptr;
    int TO = 128;
    size_t smem = 0;
    int post = 0;
    // spectral noise reduction workspace
    bool nr = false;
    int nr_frames = 0, nr_len = 0;
    float *d_nr_win = nullptr, *d_nr_mag = nullptr, *d_nr_sel = nullptr;
    float2 *d_nr_tw = nullptr, *d_nr_stft = nullptr;
};

// Retune without rebuilding the bank (API PATCH of offset_hz -> capture.py:442-501): the mixer constant is one float32
// per channel in a device array, nothing else in the bank depends on the offsets.  The copy is enqueued on `stream`, so
// launches already queued there keep the old offsets and later ones see the new; no allocation, no synchronisation.
extern "C" int wh_chanbank_set_offsets(wh_chanbank *b, const int *h_offsets_hz, int n_channels, void *stream) {
    if (!b || !h_offsets_hz) return set_err(WH_E_ARG, "wh_chanbank_set_offsets: null");
    if (n_channels != b->cfg.n_channels) return set_err(WH_E_ARG, "wh_chanbank_set_offsets: the bank has %d channels", b->cfg.n_channels);
    std::vector<float> &h = b->h_stage[b->h_stage_cur ^= 1];
    h.resize((size_t)n_channels);
    for (int k = 0; k < n_channels; ++k) h[k] = h_offsets_hz[k] == 0 ? 0.0f : nco_const(h_offsets_hz[k], b->cfg.sample_rate);
    WH_HIP(hipMemcpyAsync(b->d_nco, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, as_stream(stream)));
    return WH_OK;
}

// New squelch thresholds (float32 dB per channel, NaN = none) for a bank created with h_squelch_db.
extern "C" int wh_chanbank_set_squelch(wh_chanbank *b, const float *h_squelch_db, int n_channels, void *stream) {
    if (!b || !h_squelch_db) return set_err(WH_E_ARG, "wh_chanbank_set_squelch: null");
    if (n_channels != b->cfg.n_channels || !b->d_squelch)
        return set_err(WH_E_ARG, "wh_chanbank_set_squelch: the bank has %d channels and %s squelch array", b->cfg.n_channels,
                       b->d_squelch ? "a" : "no");
    std::vector<float> &h = b->h_stage[b->h_stage_cur ^= 1];
    h.assign(h_squelch_db, h_squelch_db + n_channels);
    WH_HIP(hipMemcpyAsync(b->d_squelch, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, as_stream(stream)));
    return WH_OK;
}

extern "C" int wh_chanbank_create(wh_chanbank **out, const wh_chanbank_cfg *c) {
    if (!out || !c || !c->h_offsets_hz) return set_err(WH_E_ARG, "wh_chanbank_create: null");
    const bool resample = c->ntaps > 0;
    if (c->n_channels < 1 || c->n_channels > 65535 || c->chunk_len < 2 || c->chunk_len > (1 << 24) ||
        (resample && (!c->h_taps || c->up < 1 || c->down < 1)) || c->n_out < 1 || c->demod < 0 || c->demod > 5 ||
        (c->input_format != 0 && c->input_format != 1) || c->n_stages < 0 || c->n_stages > MAX_STAGES ||
        (c->n_stages > 0 && !c->h_stages) || c->post < 0 || c->post > 1)
        return set_err(WH_E_ARG, "wh_chanbank_create: bad configuration");
    for (int i = 0; i < c->n_stages; ++i)
        if (c->h_stages[i].n < 1 || c->h_stages[i].n > MAX_ORD || c->h_stages[i].a[0] == 0.0)
            return set_err(WH_E_ARG, "wh_chanbank_create: bad IIR stage %d", i);
    wh_chanbank *b = new wh_chanbank();
    std::unique_ptr<wh_chanbank, void (*)(wh_chanbank *)> guard(b, wh_chanbank_destroy);  // frees partial state on early return
    b->cfg = *c;
    b->cfg.h_offsets_hz = nullptr;
    b->cfg.h_taps = nullptr;
    b->cfg.h_stages = nullptr;
    b->post = c->post == 0 ? 0 : (c->agc ? 1 : 2);
    if (c->noise_reduction) {
        if (!c->h_nr_window || c->post != 0 || !(c->nr_reduction_linear > 0.0f))
            return set_err(WH_E_ARG, "wh_chanbank_create: noise reduction needs the FM chain, a window and a positive factor");
        if (c->chunk_len >= NR_FFT) {   // shorter chunks pass through unchanged (dsp/filters.py:388-389)
            b->nr = true;
            b->nr_frames = (c->chunk_len - NR_FFT) / NR_HOP + 1;
            b->nr_len = (b->nr_frames - 1) * NR_HOP + NR_FFT;
            std::vector<float2> tw(NR_FFT);
            for (int m = 0; m < NR_FFT; ++m) {
                double ang = -2.0 * M_PI * (double)m / (double)NR_FFT;
                tw[m] = make_float2((float)cos(ang), (float)sin(ang));
            }
            WH_HIP(hipMalloc(&b->d_nr_win, NR_FFT * sizeof(float)));
            WH_HIP(hipMalloc(&b->d_nr_tw, NR_FFT * sizeof(float2)));
            WH_HIP(hipMemcpy(b->d_nr_win, c->h_nr_window, NR_FFT * sizeof(float), hipMemcpyHostToDevice));
            WH_HIP(hipMemcpy(b->d_nr_tw, tw.data(), NR_FFT * sizeof(float2), hipMemcpyHostToDevice));
        }
    }
    b->cfg.h_nr_window = nullptr;
    {
        const int n_eff = b->nr ? b->nr_len : c->chunk_len;
        const long long n_up = (long long)n_eff * (resample ? c->up : 1);
        const long long want = resample ? n_up / c->down + (n_up % c->down ? 1 : 0) : n_eff;
        if (want != c->n_out) return set_err(WH_E_ARG, "wh_chanbank_create: n_out must be %lld for this configuration", want);
    }
    std::vector<float> nco(c->n_channels);
    for (int k = 0; k < c->n_channels; ++k) nco[k] = c->h_offsets_hz[k] == 0 ? 0.0f : nco_const(c->h_offsets_hz[k], c->sample_rate);
    WH_HIP(hipMalloc(&b->d_nco, nco.size() * sizeof(float)));
    WH_HIP(hipMemcpy(b->d_nco, nco.data(), nco.size() * sizeof(float), hipMemcpyHostToDevice));
    if (c->h_squelch_db) {
        WH_HIP(hipMalloc(&b->d_squelch, (size_t)c->n_channels * sizeof(float)));
        WH_HIP(hipMemcpy(b->d_squelch, c->h_squelch_db, (size_t)c->n_channels * sizeof(float), hipMemcpyHostToDevice));
    }
    b->cfg.h_squelch_db = nullptr;
    if (resample) {
        WH_HIP(hipMalloc(&b->d_taps, (size_t)c->ntaps * sizeof(double)));
        WH_HIP(hipMemcpy(b->d_taps, c->h_taps, (size_t)c->ntaps * sizeof(double), hipMemcpyHostToDevice));
    }
    if (c->n_stages > 0) {
        // normalise by a[0] in the stage's dtype, as lfilter does on its dtype-cast copies
        std::vector<StageDev> sd(c->n_stages);
        for (int i = 0; i < c->n_stages; ++i) {
            const wh_iir_stage &h = c->h_stages[i];
            sd[i].is_f64 = h.is_f64;
            sd[i].n = h.n;
            for (int k = 0; k < MAX_ORD; ++k) {
                if (k >= h.n) { sd[i].b[k] = 0.0; sd[i].a[k] = 0.0; continue; }
                if (h.is_f64) {
                    sd[i].b[k] = h.b[k] / h.a[0];
                    sd[i].a[k] = h.a[k] / h.a[0];
                } else {
                    sd[i].b[k] = (double)((float)h.b[k] / (float)h.a[0]);
                    sd[i].a[k] = (double)((float)h.a[k] / (float)h.a[0]);
                }
            }
        }
        for (int i = 0; i < c->n_stages; ++i) b->stages.st[i] = sd[i];
    }
    // exact scan form of the rows (see chan_rows_scan_kernel): for long chunks that the warm-up form cannot take, and
    // wherever its sequential depth is the shorter one -- the warm-up form walks segment + warm-up samples per lane, the
    // scan form twice the segment plus the state scans (about 40 sample-equivalents per stage and scan).  Waves per row:
    // a property of the bank (channel count, chunk length), never of the call.
    if (c->iir_scan && (c->n_stages > 0 || c->agc) && c->chunk_len >= 4096) {
        int w = c->n_channels <= 16 ? ROWS_MAXW : c->n_channels <= 64 ? 4 : 1;
        while (w > 1 && c->chunk_len < 64 * w * 16) w >>= 1;
        const int sg = (c->chunk_len + 64 * w - 1) / (64 * w);
        long long depth_warm = -1;
        if (!c->agc && c->iir_warmup > 0 && c->n_stages > 0 && (c->chunk_len + 63) / 64 + c->iir_warmup <= c->chunk_len / 2) {
            int rw = c->n_stages > 3 ? 1 : w;   // (the warm-up kernel's own wave count, see wh_chanbank_run)
            depth_warm = (c->chunk_len + 64 * rw - 1) / (64 * rw) + c->iir_warmup;
        }
        const long long depth_scan = 2LL * sg + 40LL * (c->n_stages + (c->agc ? 2 : 0)) * (w > 1 ? 2 : 1);
        if (depth_warm < 0 || depth_scan < depth_warm) {
            b->scan_seg = sg;
            b->scan_w = w;
            AgcDev g;
            g.on = c->agc; g.target = c->agc_target; g.max_gain = c->agc_max_gain;
            g.att_b0 = c->agc_att_b0; g.att_a1 = c->agc_att_a1; g.rel_b0 = c->agc_rel_b0; g.rel_a1 = c->agc_rel_a1;
            // transition powers over one segment and over a wave's 64 segments
            WH_HIP(hipMalloc(&b->d_pow, 2 * sizeof(ScanPow)));
            WH_HIP(hipMemset(b->d_pow, 0, 2 * sizeof(ScanPow)));
            hipLaunchKernelGGL(iir_pow_kernel, dim3(c->n_stages + 1), dim3(MAX_ORD - 1), 0, nullptr, b->stages, c->n_stages, g,
                               sg, b->d_pow);
            WH_LAUNCH_CHECK();
            if (w > 1) {
                hipLaunchKernelGGL(iir_pow_kernel, dim3(c->n_stages + 1), dim3(MAX_ORD - 1), 0, nullptr, b->stages, c->n_stages,
                                   g, 64 * sg, b->d_pow + 1);
                WH_LAUNCH_CHECK();
            }
            WH_HIP(hipDeviceSynchronize());
            const size_t tile_bytes = (size_t)w * ROWS_CH * 65 * sizeof(float);
            if (tile_bytes > 64 * 1024)
                WH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chan_rows_scan_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_bytes));
        }
    }
    // warm-up form with the start states from the chain's impulse response (see iir_start_kernel)
    if (!b->scan_seg && c->iir_warmup_form == 0 && !c->agc && c->iir_warmup > 0 && c->n_stages > 0 &&
        (c->chunk_len + 63) / 64 + c->iir_warmup <= c->chunk_len / 2) {
        int E = 0;
        for (int i = 0; i < c->n_stages; ++i) E += b->stages.st[i].n - 1;
        const int EP = (E + 1) & ~1;
        if (E >= 1 && EP <= 16 && c->n_stages <= 3) {   // (longer state vectors keep the recurrence warm-up)
            b->start_ep = EP;
            WH_HIP(hipMalloc(&b->d_G, (size_t)c->iir_warmup * EP * sizeof(double)));
            hipLaunchKernelGGL(iir_impulse_kernel, dim3(1), dim3(1), 0, nullptr, b->stages, c->n_stages, c->iir_warmup, EP,
                               b->d_G);
            WH_LAUNCH_CHECK();
            WH_HIP(hipDeviceSynchronize());
        }
    }
    // fused path: plain FM (no IIR stage, no AGC), pure decimation, everything fits in LDS.  Every other chain with a pure
    // decimation that fits uses the same kernel as the decimating FIR of its rows (rows_fir).
    b->fused = false;
    const bool plain_fm = c->demod == 0 && c->n_stages == 0 && !c->agc && c->post == 0;
    if (!b->nr && resample && c->up == 1 && c->ntaps <= FM_MAX_TAPS) {
        int TO = 256;
        while (TO > 8 && c->ntaps + (TO - 1) * c->down > FM_MAX_SPAN) TO >>= 1;
        // the kernel's 86 VGPRs allow 5 workgroups per CU, LDS only 4 at 128 outputs per tile with the reference's
        // 1001 taps at 50:1 (33.4 KB): a few outputs fewer per tile fit 5, and the launch is occupancy-sensitive (2 / 3 /
        // 4 workgroups per CU: 4.07 / 2.96 / 2.48 ms).  Measured: 124 outputs (32.6 KB) still runs 4 per CU, 120
        // (31.8 KB) runs 5: 2.48 -> 2.35 ms; 116 / 112 give the gain back to emptier tiles.
        if (TO == 128 && (c->down & 1) == 0) {
            auto bytes = [&](int to) {
                return (size_t)((c->ntaps + 1) & ~1) * sizeof(float) + (size_t)(c->ntaps + 1 + (to - 1) * c->down) * sizeof(float);
            };
            int to = 128;
            while (to > 100 && bytes(to) > 32000) to -= 4;
            if (bytes(to) <= 32000) TO = to;
        }
        // polyphase register form of the FIR (fm_fir_phases): 21 tap blocks of `down`, 17..32 phase pairs, tiles of 128
        // outputs, no tap table in LDS (29.7 KB at 1001 taps / 50:1: five workgroups per CU)
        b->fir_phase = 0;
        if ((c->down & 1) == 0 && c->down >= 34 && c->down <= 64 && c->ntaps <= FIR_Q * c->down &&
            c->ntaps + 127 * c->down + 64 <= FM_MAX_SPAN) {
            b->fir_phase = 1;
            TO = 128;
            // tap pairs by block q and phase pair rp: (trev[q down + 2 rp], trev[q down + 2 rp + 1]), trev[i] = h[ntaps - 1 - i]
            // rounded to float32; zero beyond the filter and for the idle lanes rp >= down / 2
            std::vector<float2> tp((size_t)FIR_Q * 32, make_float2(0.f, 0.f));
            for (int q = 0; q < FIR_Q; ++q)
                for (int rp = 0; rp < c->down / 2; ++rp) {
                    const int i = q * c->down + 2 * rp;
                    tp[(size_t)q * 32 + rp] = make_float2(i < c->ntaps ? (float)c->h_taps[c->ntaps - 1 - i] : 0.f,
                                                          i + 1 < c->ntaps ? (float)c->h_taps[c->ntaps - 2 - i] : 0.f);
                }
            WH_HIP(hipMalloc(&b->d_tphase, tp.size() * sizeof(float2)));
            WH_HIP(hipMemcpy(b->d_tphase, tp.data(), tp.size() * sizeof(float2), hipMemcpyHostToDevice));
        }
        if (c->ntaps + (TO - 1) * c->down <= FM_MAX_SPAN) {
            b->fused = plain_fm;
            b->rows_fir = !plain_fm;
            b->TO = TO;
            b->smem = b->fir_phase ? (size_t)(c->ntaps + (TO - 1) * c->down + 64) * sizeof(float)
                                   : (size_t)((c->ntaps + 1) & ~1) * sizeof(float) + (size_t)(c->ntaps + 1 + (TO - 1) * c->down) * sizeof(float);
            WH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fmbank_fused_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->smem));
            WH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fmbank_fused_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->smem));
        }
    }
    *out = guard.release();
    return WH_OK;
}

extern "C" void wh_chanbank_destroy(wh_chanbank *b) {
    if (!b) return;
    (void)hipFree(b->d_nco);
    (void)hipFree(b->d_tphase);
    (void)hipFree(b->d_squelch);
    (void)hipFree(b->d_pow);
    (void)hipFree(b->d_G);
    (void)hipFree(b->d_S0);
    (void)hipFree(b->d_taps);
    (void)hipFree(b->d_acc);
    (void)hipFree(b->d_fm);
    (void)hipFree(b->d_araw);
    (void)hipFree(b->d_nr_win); (void)hipFree(b->d_nr_tw); (void)hipFree(b->d_nr_mag); (void)hipFree(b->d_nr_sel);
    (void)hipFree(b->d_nr_stft);
    delete b;
}

extern "C" size_t wh_chanbank_workspace_bytes(const wh_chanbank *b, size_t n_chunks) {
    if (!b) return 0;
    size_t rows = n_chunks * (size_t)b->cfg.n_channels;
    size_t bytes = rows * 2 * sizeof(double);
    if (!b->fused) bytes += rows * (size_t)b->cfg.chunk_len * sizeof(float);
    if (b->nr) bytes += rows * (size_t)b->nr_frames * NR_BINS * (sizeof(float2) + sizeof(float)) + rows * NR_BINS * 2 * sizeof(float);
    return bytes;
}

extern "C" int wh_chanbank_run(wh_chanbank *b, const void *d_in, size_t n_chunks, float *d_audio, float *d_metrics,
                               void *stream) {
    return wh_chanbank_run_wire(b, d_in, n_chunks, d_audio, d_metrics, 0, nullptr, stream);
}

extern "C" int wh_chanbank_run_wire(wh_chanbank *b, const void *d_in, size_t n_chunks, float *d_audio, float *d_metrics,
                                    int wire_format, void *d_wire, void *stream) {
    if (!b) return set_err(WH_E_ARG, "wh_chanbank_run: null handle");
    if (n_chunks == 0) return WH_OK;
    if (!d_in || !d_audio || !d_metrics) return set_err(WH_E_ARG, "wh_chanbank_run: null buffer");
    if (wire_format < 0 || wire_format > 2 || (wire_format != 0 && !d_wire))
        return set_err(WH_E_ARG, "wh_chanbank_run_wire: wire_format 0 (none) / 1 (int16 PCM) / 2 (clipped float32) with a buffer");
    if (n_chunks > 65535) return set_err(WH_E_ARG, "wh_chanbank_run: n_chunks > 65535 per call");
    hipStream_t st = as_stream(stream);
    const wh_chanbank_cfg &c = b->cfg;
    const size_t rows = n_chunks * (size_t)c.n_channels;
    const bool resample = c.ntaps > 0;
    if (n_chunks > b->cap_chunks) {  // grow the workspace (synchronises; steady state never does)
        WH_HIP(hipStreamSynchronize(st));
        (void)hipFree(b->d_acc);
        (void)hipFree(b->d_fm);
        (void)hipFree(b->d_araw);
        b->d_acc = nullptr;
        b->d_fm = nullptr;
        b->d_araw = nullptr;
        WH_HIP(hipMalloc(&b->d_acc, rows * 2 * sizeof(double)));
        if (b->fused) WH_HIP(hipMalloc(&b->d_araw, n_chunks * (size_t)c.chunk_len * sizeof(float)));
        if (!b->fused && (resample || b->nr)) WH_HIP(hipMalloc(&b->d_fm, rows * (size_t)c.chunk_len * sizeof(float)));
        if (b->d_G) {
            (void)hipFree(b->d_S0);
            b->d_S0 = nullptr;
            // (segments per row: the warm-up form's wave count is a property of the bank, see below)
            int rw, rg;
            rows_shape(c, b->d_G != nullptr, &rw, &rg);
            WH_HIP(hipMalloc(&b->d_S0, rows * (size_t)(64 * rw * rg) * b->start_ep * sizeof(double)));
        }
        if (b->nr) {
            (void)hipFree(b->d_nr_mag); (void)hipFree(b->d_nr_sel); (void)hipFree(b->d_nr_stft);
            b->d_nr_mag = b->d_nr_sel = nullptr;
            b->d_nr_stft = nullptr;
            WH_HIP(hipMalloc(&b->d_nr_stft, rows * (size_t)b->nr_frames * NR_BINS * sizeof(float2)));
            WH_HIP(hipMalloc(&b->d_nr_mag, rows * (size_t)b->nr_frames * NR_BINS * sizeof(float)));
            WH_HIP(hipMalloc(&b->d_nr_sel, rows * (size_t)NR_BINS * 2 * sizeof(float)));
        }
        b->cap_chunks = n_chunks;
    }
    WH_HIP(hipMemsetAsync(b->d_acc, 0, rows * 2 * sizeof(double), st));
    FmArgs a;
    a.in = d_in;
    a.audio = d_audio;
    a.acc = b->d_acc;
    a.fm_out = (resample || b->nr) ? b->d_fm : d_audio;   // no resampling, no length change: rows are the audio
    a.rows_src = nullptr;
    a.skip_fm_sum = (!b->fused && (c.n_stages > 0 || c.agc)) ? 1 : 0;
    a.nco_c = b->d_nco;
    a.taps = b->d_taps;
    a.fmt = c.input_format;
    a.N = c.chunk_len;
    a.K = c.n_channels;
    a.n_out = c.n_out;
    a.ntaps = c.ntaps;
    a.down = c.down;
    a.d0 = c.d0;
    a.TO = b->TO;
    a.fir_phase = b->fir_phase;
    a.tphase = b->d_tphase;
    a.araw = nullptr;
    a.R = 1;
    a.scale = (float)((double)c.sample_rate / (2.0 * M_PI * 75000.0));
    a.demod = c.demod;
    a.bfo_c = 2.0 * M_PI * c.bfo_hz;   // Python: (2j*np.pi) * offset_hz
    a.fs_d = (double)c.sample_rate;
    a.pll_alpha = c.pll_alpha;
    a.pll_beta = c.pll_beta;
    if (b->fused) {
        // raw discriminator angles + sum |x|^2, once per chunk (shared by the channels)
        {
            const unsigned gx = (unsigned)((c.chunk_len + 4095) / 4096);
            if (c.input_format == 1) hipLaunchKernelGGL(fm_rawdisc_kernel<1>, dim3(gx, (unsigned)n_chunks), dim3(256), 0, st, a, b->d_araw);
            else hipLaunchKernelGGL(fm_rawdisc_kernel<0>, dim3(gx, (unsigned)n_chunks), dim3(256), 0, st, a, b->d_araw);
            WH_LAUNCH_CHECK();
            a.araw = b->d_araw;
        }
        const int tiles = (c.n_out + b->TO - 1) / b->TO;
        const int R = run_tiles(tiles, rows);
        a.R = R;
        hipLaunchKernelGGL(fmbank_fused_kernel<false>, dim3((tiles + R - 1) / R, c.n_channels, (unsigned)n_chunks), dim3(256),
                           b->smem, st, a);
        WH_LAUNCH_CHECK();
    } else {
        int per_block = 4 * 63 * 16;
        int blocks = (c.chunk_len + per_block - 1) / per_block;
        if (c.demod >= 3)
            hipLaunchKernelGGL(sam_front_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st, a, (int)rows);
        else
            hipLaunchKernelGGL(chan_front_kernel, dim3(blocks, c.n_channels, (unsigned)n_chunks), dim3(256), 0, st, a);
        WH_LAUNCH_CHECK();
        if (c.n_stages > 0 || c.agc) {
            AgcDev g;
            g.on = c.agc;
            g.target = c.agc_target;
            g.max_gain = c.agc_max_gain;
            g.att_b0 = c.agc_att_b0; g.att_a1 = c.agc_att_a1;
            g.rel_b0 = c.agc_rel_b0; g.rel_a1 = c.agc_rel_a1;
            // time-parallel mode when it at least halves the sequential depth (see chan_rows_kernel)
            int seg = 0, rw = 1, rg = 1;
            if (!c.agc && c.iir_warmup > 0 && c.n_stages > 0) {
                const int sg = (c.chunk_len + 63) / 64;
                if (sg + c.iir_warmup <= c.chunk_len / 2) {
                    rows_shape(c, b->d_G != nullptr, &rw, &rg);
                    seg = (c.chunk_len + 64 * rw * rg - 1) / (64 * rw * rg);
                }
            }
            const double *S0 = nullptr;
            if (seg && b->d_G) {   // start states of all segments, chip-wide (iir_start_kernel)
                const dim3 sg2((unsigned)rows, (unsigned)((64 * rw * rg + START_SEGS - 1) / START_SEGS));
#define WH_START(EP_)                                                                                                     \
    hipLaunchKernelGGL(iir_start_kernel<EP_>, sg2, dim3(256), 0, st, a.fm_out, c.chunk_len, seg, c.iir_warmup, 64 * rw * rg, \
                       b->d_G, b->d_S0)
                switch (b->start_ep) {
                    case 2: WH_START(2); break;
                    case 4: WH_START(4); break;
                    case 6: WH_START(6); break;
                    case 8: WH_START(8); break;
                    case 10: WH_START(10); break;
                    case 12: WH_START(12); break;
                    case 14: WH_START(14); break;
                    default: WH_START(16); break;
                }
#undef WH_START
                WH_LAUNCH_CHECK();
                S0 = b->d_S0;
            }
            if (b->scan_seg) {
                hipLaunchKernelGGL(chan_rows_scan_kernel, dim3((unsigned)rows), dim3(64 * b->scan_w),
                                   (size_t)b->scan_w * ROWS_CH * 65 * sizeof(float), st, a.fm_out, b->d_acc, c.chunk_len,
                                   b->stages, c.n_stages, g, b->scan_seg, b->d_pow);
                WH_LAUNCH_CHECK();
            } else {
            const dim3 rgrid(seg ? (unsigned)rows : (unsigned)((rows + 63) / 64), (unsigned)(seg ? rg : 1));
            const size_t tile_bytes = (size_t)rw * ROWS_CH * 65 * sizeof(float);
#define WH_ROWS(NS_)                                                                                                      \
    do {                                                                                                                  \
        if (tile_bytes > 64 * 1024)                                                                                       \
            WH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chan_rows_kernel<NS_>),                             \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_bytes));                     \
        hipLaunchKernelGGL(chan_rows_kernel<NS_>, rgrid, dim3(64 * rw), tile_bytes, st, a.fm_out, b->d_acc, (int)rows,    \
                           c.chunk_len, b->stages, g, seg, c.iir_warmup, S0, b->start_ep);                                \
    } while (0)
            switch (c.n_stages) {
                case 0: WH_ROWS(0); break;
                case 1: WH_ROWS(1); break;
                case 2: WH_ROWS(2); break;
                case 3: WH_ROWS(3); break;
                case 4: WH_ROWS(4); break;
                case 5: WH_ROWS(5); break;
                case 6: WH_ROWS(6); break;
                case 7: WH_ROWS(7); break;
                default: WH_ROWS(8); break;
            }
#undef WH_ROWS
            WH_LAUNCH_CHECK();
            }
        }
        if (b->nr) {
            const int F = b->nr_frames;
            if (rows > 65535) return set_err(WH_E_ARG, "wh_chanbank_run: noise reduction supports <= 65535 rows per call");
            hipLaunchKernelGGL(nr_stft_kernel, dim3(F, (unsigned)rows), dim3(256), 0, st, b->d_fm, c.chunk_len, F,
                               b->d_nr_win, b->d_nr_tw, b->d_nr_stft, b->d_nr_mag);
            WH_LAUNCH_CHECK();
            const double virt = (double)(F - 1) * 0.1;   // np.percentile(..., 10), method 'linear'
            int k_lo = (int)floor(virt);
            int k_hi = k_lo + 1 < F ? k_lo + 1 : F - 1;
            if (F <= RANK_MAX)
                hipLaunchKernelGGL(rank_select_kernel, dim3((unsigned)((rows * NR_BINS + 3) / 4)), dim3(256),
                                   (size_t)4 * F * sizeof(float), st, b->d_nr_mag, (int)(rows * NR_BINS), F, k_lo, k_hi, b->d_nr_sel);
            else
                hipLaunchKernelGGL(select_kth_kernel, dim3((unsigned)(rows * NR_BINS), 2), dim3(256), 0, st, b->d_nr_mag, F, k_lo,
                                   k_hi, b->d_nr_sel);
            WH_LAUNCH_CHECK();
            WH_HIP(hipMemsetAsync(b->d_fm, 0, rows * (size_t)c.chunk_len * sizeof(float), st));
            hipLaunchKernelGGL(nr_istft_kernel, dim3(F, (unsigned)rows), dim3(256), 0, st, b->d_fm, c.chunk_len, F,
                               b->d_nr_win, b->d_nr_tw, b->d_nr_stft, b->d_nr_sel, k_lo, (float)(virt - (double)k_lo),
                               c.nr_reduction_linear);
            WH_LAUNCH_CHECK();
            hipLaunchKernelGGL(nr_norm_kernel, dim3((unsigned)rows), dim3(256), 0, st, b->d_fm, c.chunk_len, F, b->d_nr_win,
                               b->d_acc);
            WH_LAUNCH_CHECK();
        }
        const size_t n_fm = b->nr ? (size_t)b->nr_len : (size_t)c.chunk_len;
        if (resample && b->rows_fir) {
            const int tiles = (c.n_out + b->TO - 1) / b->TO;
            const int R = run_tiles(tiles, rows);
            a.R = R;
            a.rows_src = b->d_fm;
            hipLaunchKernelGGL(fmbank_fused_kernel<true>, dim3((tiles + R - 1) / R, c.n_channels, (unsigned)n_chunks), dim3(256),
                               b->smem, st, a);
            WH_LAUNCH_CHECK();
        } else if (resample) {
            int rc = launch_resample(b->d_fm, n_fm, (size_t)c.chunk_len, rows, d_audio, (size_t)c.n_out, b->d_taps, c.ntaps,
                                     c.up, c.down, c.d0, st);
            if (rc != WH_OK) return rc;
        } else if (b->nr) {   // compact the shortened rows into the audio buffer
            WH_HIP(hipMemcpy2DAsync(d_audio, (size_t)c.n_out * sizeof(float), b->d_fm, (size_t)c.chunk_len * sizeof(float),
                                    (size_t)c.n_out * sizeof(float), rows, hipMemcpyDeviceToDevice, st));
        }
    }
    hipLaunchKernelGGL(fmbank_finalize_kernel, dim3((unsigned)rows), dim3(256), 0, st, d_audio, b->d_acc, d_metrics,
                       c.chunk_len, b->nr ? b->nr_len : c.chunk_len, c.n_out, b->post, b->d_squelch, c.n_channels, wire_format,
                       d_wire, b->fused ? 1 : 0);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

/* Channel.update_signal_metrics (capture.py:749-798) for K channels of one chunk: h_out[k] =
 * {rssi_db, |x| at rank n//10, |x| at rank n - n//10 - 1} (the two order statistics the reference takes
 * with np.partition; the host shim turns them into snr_db). */
extern "C" int wh_channel_signal_metrics(const void *d_in, int input_format, size_t n, int sample_rate,
                                         const int *h_offsets_hz, int K, float *h_out, void *stream) {
    if (!d_in || !h_offsets_hz || !h_out || n < 2 || n > ((size_t)1 << 24) || K < 1 || K > 65535 ||
        (input_format != 0 && input_format != 1))
        return set_err(WH_E_ARG, "wh_channel_signal_metrics: bad arguments");
    hipStream_t st = as_stream(stream);
    std::vector<float> nco(K);
    for (int k = 0; k < K; ++k) nco[k] = h_offsets_hz[k] == 0 ? 0.0f : nco_const(h_offsets_hz[k], sample_rate);
    float *d_nco = nullptr, *d_rows = nullptr, *d_sel = nullptr, *d_rssi = nullptr;
    double *d_acc = nullptr;
    WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_nco), K * sizeof(float), st));
    WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_rows), (size_t)K * n * sizeof(float), st));
    WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_sel), (size_t)K * 2 * sizeof(float), st));
    WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_rssi), (size_t)K * sizeof(float), st));
    WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_acc), (size_t)K * 2 * sizeof(double), st));
    WH_HIP(hipMemcpyAsync(d_nco, nco.data(), K * sizeof(float), hipMemcpyHostToDevice, st));
    WH_HIP(hipMemsetAsync(d_acc, 0, (size_t)K * 2 * sizeof(double), st));
    FmArgs a;
    a.in = d_in; a.audio = nullptr; a.acc = d_acc; a.fm_out = d_rows; a.rows_src = nullptr; a.skip_fm_sum = 0; a.nco_c = d_nco; a.taps = nullptr;
    a.fmt = input_format; a.N = (int)n; a.K = K; a.n_out = 0; a.ntaps = 0; a.down = 1; a.d0 = 0; a.TO = 1; a.R = 1; a.fir_phase = 0; a.tphase = nullptr; a.araw = nullptr;
    a.scale = 0.f; a.demod = 1; a.bfo_c = 0.0; a.fs_d = (double)sample_rate; a.pll_alpha = a.pll_beta = 0.0;
    const int per_block = 4 * 63 * 16;
    hipLaunchKernelGGL(chan_front_kernel, dim3((unsigned)((n + per_block - 1) / per_block), K, 1), dim3(256), 0, st, a);
    WH_LAUNCH_CHECK();
    const int k_lo = (int)(n / 10), k_hi = (int)(n - n / 10 - 1);
    hipLaunchKernelGGL(select_kth_kernel, dim3(K, 2), dim3(256), 0, st, d_rows, (int)n, k_lo, k_hi, d_sel);
    WH_LAUNCH_CHECK();
    hipLaunchKernelGGL(metrics_rssi_kernel, dim3((K + 63) / 64), dim3(64), 0, st, d_acc, K, (int)n, d_rssi);
    WH_LAUNCH_CHECK();
    std::vector<float> sel((size_t)K * 2), rssi(K);
    WH_HIP(hipMemcpyAsync(sel.data(), d_sel, sel.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    WH_HIP(hipMemcpyAsync(rssi.data(), d_rssi, rssi.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    WH_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < K; ++k) {
        h_out[k * 3] = rssi[k];
        h_out[k * 3 + 1] = sel[k * 2];
        h_out[k * 3 + 2] = sel[k * 2 + 1];
    }
    WH_HIP(hipFreeAsync(d_nco, st));
    WH_HIP(hipFreeAsync(d_rows, st));
    WH_HIP(hipFreeAsync(d_sel, st));
    WH_HIP(hipFreeAsync(d_rssi, st));
    WH_HIP(hipFreeAsync(d_acc, st));
    return WH_OK;
}

// ---- noise blanker (reference dsp/filters.py:267-343), real float32 rows ----------------------------------------
// median |x| (np.median: mean of the two middle order statistics for even n, in float32) -> threshold = median *
// float32(10^(dB/20)) -> samples above it and `width` neighbours on each side are zeroed.  The reference's
// dispatcher never forwards enable_noise_blanker (capture.py:340-414); this is the standalone function.
namespace {
__global__ __launch_bounds__(256) void nb_abs_kernel(const float *x, float *mag, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) mag[i] = fabsf(x[i]);
}
__global__ __launch_bounds__(256) void nb_apply_kernel(const float *x, const float *sel, float factor, int n, int width,
                                                       int odd, float *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float med = odd ? sel[1] : __fmul_rn(__fadd_rn(sel[0], sel[1]), 0.5f);
    float v = x[i];
    if (med >= 1e-10f) {
        const float thr = __fmul_rn(med, factor);
        bool hit = false;
        for (int d = -width; d <= width; ++d) {
            const int j = i + d;
            if (j >= 0 && j < n && fabsf(x[j]) > thr) hit = true;
        }
        if (hit) v = 0.0f;
    }
    out[i] = v;
}
}  // namespace

extern "C" int wh_noise_blanker(const float *d_x, float *d_out, size_t n, float threshold_factor, int blanking_width,
                                void *stream) {
    if (n == 0) return WH_OK;
    if (!d_x || !d_out || d_x == d_out || n > ((size_t)1 << 24) || blanking_width < 0 || blanking_width > 4096)
        return set_err(WH_E_ARG, "wh_noise_blanker: bad arguments (out of place, n <= 2^24)");
    hipStream_t st = as_stream(stream);
    float *d_mag = nullptr, *d_sel = nullptr;
    WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_mag), n * sizeof(float), st));
    WH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_sel), 2 * sizeof(float), st));
    hipLaunchKernelGGL(nb_abs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_x, d_mag, n);
    const int k_hi = (int)(n / 2), k_lo = (n & 1) ? k_hi : k_hi - 1;
    hipLaunchKernelGGL(select_kth_kernel, dim3(1, 2), dim3(256), 0, st, d_mag, (int)n, k_lo, k_hi, d_sel);
    hipLaunchKernelGGL(nb_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_x, d_sel, threshold_factor,
                       (int)n, blanking_width, (int)(n & 1), d_out);
    hipError_t e = hipGetLastError();
    (void)hipFreeAsync(d_mag, st);
    (void)hipFreeAsync(d_sel, st);
    if (e != hipSuccess) return set_err(WH_E_HIP, "wh_noise_blanker: %s", hipGetErrorString(e));
    return WH_OK;
}
