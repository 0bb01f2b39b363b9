// Polyphase filterbank (reference dsp/channelizer.py:28-158) for channel counts other than 1024: kernels shaped at
// compile time per channel count M (4 | M, M/4 = 2^a 3^b 5^c), 9 taps per arm.  M = 320 is the shape of the reference's
// own benchmark (benchmark_dsp.py:112-141, PolyphaseChannelizer(8e6) -> 8e6 / 25e3).
//
// One workgroup walks R runs of consecutive hops side by side, GH hops of each run per iteration ("group"):
//   * thread (r, u), u < M/4, owns the quad of columns u, u + M/4, u + M/2, u + 3M/4 of run r.  As in the M = 1024
//     kernel, block_g[k + M/2] == block_{g+1}[k], so two sliding register windows (columns u and u + M/4 of the
//     half-blocks) feed all four columns: every input sample is loaded once per run, the 36 taps stay in registers,
//     the arm MAC is 36 packed FMAs per hop and thread.  R is chosen so that R * M/4 fills whole wavefronts (M = 320:
//     4 runs x 80 quads = 5 waves, every lane busy; one wave per run left 37 % of the lanes idle).
//   * FFT_M, decimation in frequency, IN PLACE: the radix-4 stage over the quad happens in registers; the hop's image
//     (M complex in LDS) then holds four M/4-point sub-transforms, done by radix-4/2/3/5 passes whose butterflies read
//     and write the same LDS words -- no ping-pong image, no read-all-before-write, one image per hop.  Results end
//     digit-reversed in the image, which costs nothing: the last pass's butterfly with digit-reversed index kb owns the
//     outputs kb + (M/r) j, so lane <-> kb makes every store instruction write consecutive channels.
//   * a wave runs the passes of its own hops (NWF waves x HPW hop images: synchronisation between passes is the
//     in-order LDS queue of one wave, no s_barrier), and the workgroup meets twice per group (images written / images
//     free).  Channel counts whose hop does not fit a wave's share use all threads per pass with workgroup barriers.
//   * all index arithmetic (butterfly -> image word, twiddle slot, digit reversal) folds into constants per M.
// The same passes serve the spectrum frames of spectrum.hip (spectrum_mid_kernel below: a frame is an image, the last
// pass writes dB at the fftshift-ed bin).
// The first T-1 hops of a call read the carried history, where the half-block identity does not hold (after reset()
// or an assigned arm_history): they are computed column by column by extra workgroups of the same launch, through the
// same passes (so a hop's bits do not depend on where a stream is cut into calls); one more workgroup writes the
// history for the next call.  One launch per call.
#include "pfb_internal.h"
#include "wh_common.h"

#include <cstdint>
#include <cstdlib>

using namespace wh;

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

struct MidPlan {
    int np;        // LDS passes of the M/4-point sub-transform
    int r[12];     // radix of pass p
    int L[12];     // sub-length entering pass p (m = L / r: stride between a butterfly's words)
    int two[12];   // offset of pass p's twiddle table [k-1][u'] (k = 1..r-1, u' < m) in the LDS twiddle array
    int twn;       // twiddle entries in total (the last pass has none)
    bool ok;
};

// r8: radix-8 passes for the power-of-two part (2^k -> as many 8s as leave no lone radix-2 pass: k = 3a -> a eights,
// 3a + 2 -> a eights and a 4, 3a + 1 -> a - 1 eights and two 4s): a third fewer trips through LDS than radix 4.
constexpr MidPlan mid_plan(int Q, bool r8 = false) {
    MidPlan p{};
    int rem = Q, L = Q, n = 0;
    if (r8) {
        int k = 0;
        for (int q = Q; q % 2 == 0; q /= 2) ++k;
        int n8 = k % 3 == 1 ? k / 3 - 1 : k / 3;
        for (; n8 > 0; --n8) {
            p.r[n] = 8;
            p.L[n] = L;
            L /= 8;
            rem /= 8;
            ++n;
        }
    }
    const int rad[4] = {4, 2, 3, 5};
    for (int ri = 0; ri < 4; ++ri)
        while (rem > 1 && rem % rad[ri] == 0 && n < 12) {
            p.r[n] = rad[ri];
            p.L[n] = L;
            L /= rad[ri];
            rem /= rad[ri];
            ++n;
        }
    p.np = n;
    p.ok = rem == 1 && n >= 1;
    int off = 0;
    for (int i = 0; i < n; ++i) {
        p.two[i] = off;
        if (i < n - 1) off += (p.r[i] - 1) * (p.L[i] / p.r[i]);
    }
    p.twn = off;
    return p;
}

template <int FMT>
__device__ __forceinline__ v2f ld_iq(const void *p, long long i) {
    if (FMT == 1) {   // A1 unpack rule (cli.py:447-452): int16 / 32768
        const short2 v = reinterpret_cast<const short2 *>(p)[i];
        return v2f{(float)v.x * (1.0f / 32768.0f), (float)v.y * (1.0f / 32768.0f)};
    }
    return reinterpret_cast<const v2f *>(p)[i];
}

// complex arithmetic on packed pairs: every line below is one v_pk_* instruction (swaps, broadcasts and sign patterns
// fold into op_sel / constant operands); the float2 helpers of wh_common.h compile to 1.5-2x as many instructions
#define WH_SW(a) __builtin_shufflevector(a, a, 1, 0)
#define WH_XX(a) __builtin_shufflevector(a, a, 0, 0)
#define WH_YY(a) __builtin_shufflevector(a, a, 1, 1)
#define WH_FMA(a, b, c) __builtin_elementwise_fma(a, b, c)
__device__ __forceinline__ v2f bc(float c) { return v2f{c, c}; }
// a * w with w = (wx, wy)
__device__ __forceinline__ v2f cmul3(v2f a, v2f w) {
    const v2f t = WH_XX(a) * w;
    const v2f s = WH_YY(a) * v2f{-1.f, 1.f};   // (-ay, ay)
    return WH_FMA(s, WH_SW(w), t);
}
// a * w with the table entry w4 = (wx, wy, -wy, wx)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2f cmul2(v2f a, v4f w4) {
    const v2f t = WH_XX(a) * __builtin_shufflevector(w4, w4, 0, 1);
    return WH_FMA(WH_YY(a), __builtin_shufflevector(w4, w4, 2, 3), t);
}
// x + (-i) d and x - (-i) d:  (-i) d = (d.y, -d.x)
#define WH_PMI(x, d) WH_FMA(WH_SW(d), (v2f{1.f, -1.f}), x)
#define WH_MMI(x, d) WH_FMA(WH_SW(d), (v2f{-1.f, 1.f}), x)

// forward butterflies, natural order in and out
__device__ __forceinline__ void bfly(v2f (&v)[2]) {
    const v2f a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
}
__device__ __forceinline__ void bfly(v2f (&v)[3]) {
    const float S = 0.86602540378443864676f;   // exp(-2 pi i / 3) = -1/2 - i S
    const v2f t1 = v[1] + v[2], t2 = v[1] - v[2];
    const v2f u = WH_FMA(t1, bc(-0.5f), v[0]);
    v[0] = v[0] + t1;
    v[1] = WH_FMA(WH_SW(t2), (v2f{S, -S}), u);    // u + (-i S) t2
    v[2] = WH_FMA(WH_SW(t2), (v2f{-S, S}), u);
}
__device__ __forceinline__ void bfly(v2f (&v)[4]) {
    const v2f s02 = v[0] + v[2], d02 = v[0] - v[2], s13 = v[1] + v[3], d13 = v[1] - v[3];
    v[0] = s02 + s13;
    v[2] = s02 - s13;
    v[1] = WH_PMI(d02, d13);
    v[3] = WH_MMI(d02, d13);
}
__device__ __forceinline__ void bfly(v2f (&v)[8]) {
    // even / odd halves through the radix-4 butterfly, then X[k] = E[k] + W8^k O[k], X[k + 4] = E[k] - W8^k O[k] with
    // W8 = (1 - i) / sqrt 2, W8^2 = -i, W8^3 = -(1 + i) / sqrt 2
    const float C = 0.70710678118654752440f;
    v2f e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    bfly(e);
    bfly(o);
    const v2f t1 = WH_PMI(o[1], o[1]);    // o1 - i o1 = sqrt 2 W8 o1
    const v2f t3 = WH_MMI(o[3], o[3]);    // o3 + i o3 = -sqrt 2 W8^3 o3
    v[0] = e[0] + o[0];
    v[4] = e[0] - o[0];
    v[1] = WH_FMA(t1, bc(C), e[1]);
    v[5] = WH_FMA(t1, bc(-C), e[1]);
    v[2] = WH_PMI(e[2], o[2]);
    v[6] = WH_MMI(e[2], o[2]);
    v[3] = WH_FMA(t3, bc(-C), e[3]);
    v[7] = WH_FMA(t3, bc(C), e[3]);
}
__device__ __forceinline__ void bfly(v2f (&v)[5]) {
    const float C1 = 0.30901699437494742410f, S1 = 0.95105651629515357212f;    // cos, sin 2 pi / 5
    const float C2 = -0.80901699437494742410f, S2 = 0.58778525229247312917f;   // cos, sin 4 pi / 5
    const v2f v0 = v[0];
    const v2f a1 = v[1] + v[4], b1 = v[1] - v[4], a2 = v[2] + v[3], b2 = v[2] - v[3];
    const v2f r1 = WH_FMA(a2, bc(C2), WH_FMA(a1, bc(C1), v0));
    const v2f r2 = WH_FMA(a2, bc(C1), WH_FMA(a1, bc(C2), v0));
    const v2f t1 = WH_FMA(b2, bc(S2), b1 * bc(S1));
    const v2f t2 = WH_FMA(b2, bc(-S1), b1 * bc(S2));
    v[0] = v0 + a1 + a2;
    v[1] = WH_PMI(r1, t1);
    v[4] = WH_MMI(r1, t1);
    v[2] = WH_PMI(r2, t2);
    v[3] = WH_MMI(r2, t2);
}

// (re, im) * tap + acc as one packed FMA, the tap being one half of a register PAIR of taps (the shuffle folds into the
// instruction's op_sel bits): the 36 taps of a quad take 18 register pairs.  From separate scalars the compiler keeps a
// duplicated {t, t} pair per tap (72 registers).
__device__ __forceinline__ v2f pk_fma_bc(int hi, v2f a, v2f tpair, v2f c) {   // hi: constant after unrolling
    return WH_FMA(a, hi ? WH_YY(tpair) : WH_XX(tpair), c);
}

struct MidArgs {
    const void *x;
    const v2f *hist;
    v2f *new_hist;
    v2f *out;
    v2f *sink;             // M complex nobody reads: the stores of hops past the end of the call go here, so that every
                           // wave issues the same number of stores per group and the compiler can wait for the next
                           // group's prefetched samples with a COUNTED vmcnt (its stores stay in flight)
    const float *arms;
    const float2 *tw;
    long long H;           // hops of the call
    long long n;           // samples of the call
    long long max_block;   // last half-block index fully inside the input
    int hpr;               // hops per run (multiple of GH)
    int n_head;            // workgroups [0, n_head) do the head hops, workgroup n_head the history, the rest the runs
    double *stats_ws;      // statistics-only mode: [gridDim.x][4][M] per-workgroup {sum p, sum p^2, min p, max p}
#ifdef WH_DIAG
    int ablate;            // diagnostics build: bit 0 = output stores to the sink, bit 1 = no prefetch loads, bit 2 = no passes
    long long *stamps;     // diagnostics build: [wave][group][4] cycle stamps of workgroup stamp_wg (nullptr = off)
    int stamp_wg;
#endif
};
#ifdef WH_DIAG
#define WH_STAMP(k)                                                                                         \
    if (a.stamps && bid == a.n_head + 1 + a.stamp_wg && (tid & 63) == 0 && g < 64)                          \
        a.stamps[((tid >> 6) * 64 + g) * 4 + (k)] = (long long)__builtin_readcyclecounter();
#else
#define WH_STAMP(k)
#endif

// lanes of a hop's share of a wave -> butterflies of a pass: l_sub = bl UL + ul owns, in round (cb, cu), the butterfly of
// sub-block bl + BL cb at offset ul + UL cu.  With that split every address of a round is the lane's base plus a
// compile-time constant (an instruction offset), instead of a division chain per butterfly.
struct PassMap { bool ok; int UL, BL, CU, CB; };
constexpr PassMap pass_map(int LS, int nblk, int m) {
    for (int ul = (m < LS ? m : LS); ul >= 1; --ul)
        if (m % ul == 0 && LS % ul == 0 && nblk % (LS / ul) == 0) return PassMap{true, ul, LS / ul, m / ul, nblk / (LS / ul)};
    return PassMap{false, 1, 1, 1, 1};
}

// Shape of one instance: M channels, T taps per arm, R runs per workgroup, GH hops per group, NWF waves running the
// passes (0 = all threads, workgroup barriers between passes), WPE waves per SIMD the registers are held to; image
// layout: word p of a hop image sits at p + (p / PB) PADN (PB = 0: unpadded) and images are IMGS = ph(M) + IMGX words
// apart -- chosen per M so that the pass reads / writes spread over the LDS banks; TREG of the quad's 4 T taps stay in
// registers, the rest is re-read from LDS every group; LSP lanes per hop image in the workgroup-wide passes (0 = all).
template <int M_, int T_, int R_, int GH_, int NWF_, int WPE_, int PB_, int PADN_, int IMGX_, int TREG_, int LSP_, int R8_ = 0>
struct MidCfg {
    static constexpr int M = M_, T = T_, R = R_, GH = GH_, NWF = NWF_, WPE = WPE_, PB = PB_, PADN = PADN_, TREG = TREG_;
    static constexpr int Q = M / 4, HB = M / 2, NT = R * Q, NW = (NT + 63) / 64, NIMG = R * GH;
    static constexpr int NTL = NW * 64;               // launched threads (threads >= NT own no quad; they join the passes)
    static constexpr bool WAVE_MODE = NWF > 0;
    static constexpr int HPW = WAVE_MODE ? NIMG / NWF : NIMG;
    // workgroup-wide passes: TH threads cooperate on all NIMG images, then every wave does the last pass of HPL images
    // LSP_ lanes per hop image (0: as many as the quad threads give); fewer when the pass shapes need a rounder number
    static constexpr int TH = WAVE_MODE ? 64 : (LSP_ > 0 ? LSP_ : NT / NIMG) * NIMG;
    static constexpr int HPL = HPW;                   // images of a wave's last pass in wave mode
    static constexpr MidPlan P = mid_plan(Q, R8_ != 0);   // R8_: radix-8 passes for the power-of-two part
    static constexpr int ph(int p) { return PB_ ? p + (p / PB_) * PADN_ : p; }
    static constexpr int IMGS = ph(M_) + IMGX_;
    static constexpr int NL4 = (4 * T - TREG + 3) / 4;   // float4 of LDS taps per quad
    // the images a wave transforms are written by that wave alone: no workgroup barrier at all
    static constexpr bool SELF = WAVE_MODE && NWF == NW && HPW % GH == 0 && (HPW / GH) * Q == 64;
    // lanes per hop in the affine forms (0: none)
    static constexpr int LS = WAVE_MODE ? (64 % HPW == 0 ? 64 / HPW : 0) : TH / NIMG;

    static_assert(P.ok, "M/4 must factor into 2, 3, 5");
    static_assert(NT <= 1024, "workgroup too large");
    static_assert(!WAVE_MODE || (NIMG % NWF == 0 && NWF <= NW), "wave mode: whole hop images per wave");
    static_assert(PB == 0 || Q % PB == 0, "pad blocks must tile a quarter image");
    static_assert(TREG % 4 == 0 && TREG <= 4 * T, "register taps: whole float4 groups");

    static constexpr bool pass_affine(int pi) {
        if (LS == 0) return false;
        const int r = P.r[pi], L = P.L[pi], m = L / r;
        const PassMap pm = pass_map(LS, M / L, m);
        if (!pm.ok) return false;
        for (int ls = 0; ls < LS; ++ls) {
            const int bl = ls / pm.UL, ul = ls % pm.UL;
            for (int cb = 0; cb < pm.CB; ++cb)
                for (int cu = 0; cu < pm.CU; ++cu)
                    for (int j = 0; j < r; ++j)
                        if (ph((bl + pm.BL * cb) * L + ul + pm.UL * cu + j * m) !=
                            ph(bl * L + ul) + ph(pm.BL * cb * L + pm.UL * cu) + ph(j * m))
                            return false;
        }
        return true;
    }
    // digit reversal: output index kb = d1 + 4 (d2 + r_0 (d3 + ...)) of the last pass sits at d1 Q + d2 m_0 + d3 m_1 + ...
    static constexpr int pos_of(int kb) {
        int rem = kb >> 2, pos = (kb & 3) * Q;
        for (int p = 0; p < P.np - 1; ++p) {
            const int rp = P.r[p], mp = P.L[p] / P.r[p];
            pos += (rem % rp) * mp;
            rem /= rp;
        }
        return pos;
    }
    static constexpr int RL = P.r[P.np - 1];       // last radix
    static constexpr int BPL = M / RL;             // butterflies of the last pass per image
    static constexpr int LSL = BPL < 64 ? BPL : 64;   // wave mode: lanes per hop in the last pass
    static constexpr bool lsl_affine(int lsl) {
        for (int kl = 0; kl < lsl; ++kl)
            for (int c = 0; c < BPL / lsl; ++c)
                for (int j = 0; j < RL; ++j)
                    if (ph(pos_of(kl + lsl * c) + j) != ph(pos_of(kl)) + ph(pos_of(lsl * c)) + j) return false;
        return true;
    }
    static constexpr bool last_affine() {
        if (64 % LSL != 0 || BPL % LSL != 0 || HPL % (64 / LSL) != 0) return false;
        return lsl_affine(LSL);
    }
    // workgroup mode: the last pass is cut into units (64 / LSU hops) x (LSU consecutive outputs), NU / NW units per wave
    static constexpr int pick_lsu() {
        for (int c = 64; c >= 4; c >>= 1)
            if (BPL % c == 0 && NIMG % (64 / c) == 0 && ((NIMG / (64 / c)) * (BPL / c)) % NW == 0 && lsl_affine(c)) return c;
        return 0;
    }
    static constexpr int LSU = WAVE_MODE ? 0 : pick_lsu();
    // statistics-only mode (activity scan): the last pass keeps {sum p, sum p^2, min p, max p} of its outputs in registers
    // instead of storing them.  Units are dealt chunk-major, so that a lane meets few distinct channels: NCD chunks x RL.
    static constexpr int S_HR = LSU > 0 ? 64 / LSU : 1, S_KC = LSU > 0 ? BPL / LSU : 1, S_NHG = NIMG / S_HR;
    static constexpr int S_UPW = (S_NHG * S_KC) / NW;
    static constexpr int NCD = S_UPW % S_NHG == 0 ? S_UPW / S_NHG : 1;
    static constexpr int NACC = NCD * RL;
    // the accumulators cost 8 registers per channel: the statistics variant is held to fewer waves per SIMD (as few as
    // still fit one workgroup on the CU's 4 SIMDs)
#ifndef WH_WPE_STATS_MIN
#define WH_WPE_STATS_MIN 2
#endif
    static constexpr int WPE_STATS = (NW + 3) / 4 > WH_WPE_STATS_MIN ? (NW + 3) / 4 : WH_WPE_STATS_MIN;
    static constexpr bool STATS_OK = LSU > 0 && (S_UPW % S_NHG == 0 || S_NHG % S_UPW == 0) && NACC <= 48 &&
                                     (long long)NIMG * IMGS >= 4LL * M;   // the [M][4] staging fits the image memory
};

// ---- passes, generic form: butterflies flattened over TH cooperating threads (any shape; index arithmetic per butterfly)
template <class C, int PI, int TH, int NI, bool WG>
__device__ __forceinline__ void mid_passes_generic(v2f *im, const v4f *twp, int lt) {
    constexpr MidPlan P = C::P;
    if constexpr (PI < P.np - 1) {
        constexpr int r = P.r[PI], L = P.L[PI], m = L / r;
        constexpr int BPI = C::M / r;           // butterflies per image
        constexpr int NB = NI * BPI;
        constexpr int ROUNDS = (NB + TH - 1) / TH;
        constexpr int TWO = P.two[PI];
#pragma unroll
        for (int c = 0; c < ROUNDS; ++c) {
            const int b = lt + TH * c;
            if ((c + 1) * TH <= NB || b < NB) {
                const int s = b / BPI, bb = b - s * BPI;
                const int blk = bb / m, up = bb - blk * m;
                v2f *p = im + s * C::IMGS;
                const int a0 = blk * L + up;
                const v4f *tq = twp + TWO + up;
                v2f v[r];
#pragma unroll
                for (int j = 0; j < r; ++j) v[j] = p[C::ph(a0 + j * m)];
                bfly(v);
                p[C::ph(a0)] = v[0];
#pragma unroll
                for (int k = 1; k < r; ++k) p[C::ph(a0 + k * m)] = cmul2(v[k], tq[(k - 1) * m]);
            }
        }
        if (WG) __syncthreads();
        else __builtin_amdgcn_wave_barrier();
        mid_passes_generic<C, PI + 1, TH, NI, WG>(im, twp, lt);
    }
}

template <class C, int TH, int NI>
__device__ __forceinline__ void mid_last_generic(const v2f *im, v2f *out, v2f *sink, int lt, int s0, long long hop0,
                                                 int stride_r, long long limit) {
    constexpr int r = C::RL, BPI = C::BPL;
    constexpr int NB = NI * BPI;
    constexpr int ROUNDS = (NB + TH - 1) / TH;
#pragma unroll
    for (int c = 0; c < ROUNDS; ++c) {
        const int b = lt + TH * c;
        if ((c + 1) * TH <= NB || b < NB) {
            const int s = b / BPI, kb = b - s * BPI;
            const int pos = C::pos_of(kb);
            const v2f *q = im + s * C::IMGS;
            v2f v[r];
#pragma unroll
            for (int j = 0; j < r; ++j) v[j] = q[C::ph(pos + j)];
            bfly(v);
            const int sg = s0 + s;
            const long long hop = hop0 + (long long)(sg / C::GH) * stride_r + (sg % C::GH);
            v2f *o = (hop < limit ? out + (size_t)hop * C::M : sink) + kb;
#pragma unroll
            for (int k = 0; k < r; ++k) o[k * BPI] = v[k];
        }
    }
}

// ---- passes with affine addressing: TH threads (a wave, or the workgroup's first TH threads) on NI images, LS = TH / NI
// lanes per hop
template <class C, int PI, int TH, int NI, bool WG>
__device__ __forceinline__ void mid_passes_affine(v2f *im, const v4f *twp, int lt) {
    constexpr MidPlan P = C::P;
    if constexpr (PI < P.np - 1) {
        if constexpr (C::pass_affine(PI)) {
            constexpr int r = P.r[PI], L = P.L[PI], m = L / r, LS = C::LS;
            static_assert(LS * NI == TH, "lanes per hop");
            constexpr PassMap pm = pass_map(LS, C::M / L, m);
            if (!WG || lt < TH) {
                const int hopl = lt / LS, ls = lt - hopl * LS;
                const int bl = ls / pm.UL, ul = ls - bl * pm.UL;
                v2f *base = im + hopl * C::IMGS + C::ph(bl * L + ul);
                const v4f *tq = twp + P.two[PI] + ul;
#pragma unroll
                for (int cb = 0; cb < pm.CB; ++cb)
#pragma unroll
                    for (int cu = 0; cu < pm.CU; ++cu) {
                        const int off = C::ph(pm.BL * cb * L + pm.UL * cu);
                        v2f v[r];
#pragma unroll
                        for (int j = 0; j < r; ++j) v[j] = base[off + C::ph(j * m)];
                        bfly(v);
                        base[off] = v[0];
#pragma unroll
                        for (int k = 1; k < r; ++k) base[off + C::ph(k * m)] = cmul2(v[k], tq[(k - 1) * m + pm.UL * cu]);
                    }
            }
            if (WG) __syncthreads();
            else __builtin_amdgcn_wave_barrier();
            mid_passes_affine<C, PI + 1, TH, NI, WG>(im, twp, lt);
        } else {
            // this pass (and, for simplicity, the later ones) in the flattened form
            mid_passes_generic<C, PI, TH, NI, WG>(im, twp, lt);
        }
    }
}

// last pass of a wave's HPL images (lane <-> output index: every store instruction writes consecutive channels)
template <class C>
__device__ __forceinline__ void mid_last_wave(const v2f *imw, v2f *out, v2f *sink, int lane, int s0, long long hop0,
                                              int stride_r, long long limit) {
    if constexpr (C::last_affine()) {
        constexpr int r = C::RL, BPL = C::BPL, LSL = C::LSL, HR = 64 / LSL, KC = BPL / LSL;
        const int hs = HR == 1 ? 0 : lane / LSL, kl = lane - hs * LSL;   // HR == 1: a round is one hop, wave-uniform
        const v2f *base = imw + hs * C::IMGS + C::ph(C::pos_of(kl));
#pragma unroll
        for (int sg = 0; sg < C::HPL / HR; ++sg) {
            const int s = s0 + sg * HR + hs;                       // image index in the workgroup (wave-uniform if HR == 1)
            const long long hop = hop0 + (long long)(s / C::GH) * stride_r + (s % C::GH);
            v2f *orow = (hop < limit ? out + (size_t)hop * C::M : sink) + kl;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int off = sg * HR * C::IMGS + C::ph(C::pos_of(LSL * c));
                v2f v[r];
#pragma unroll
                for (int j = 0; j < r; ++j) v[j] = base[off + j];
                bfly(v);
#pragma unroll
                for (int k = 0; k < r; ++k) orow[LSL * c + k * BPL] = v[k];
            }
        }
    } else {
        mid_last_generic<C, 64, C::HPL>(imw, out, sink, lane, s0, hop0, stride_r, limit);
    }
}

// last pass in workgroup mode: unit ui = (hop group, chunk of LSU outputs); a wave's lanes = (hop within the group, output)
template <class C>
__device__ __forceinline__ void mid_last_units(const v2f *img, v2f *out, v2f *sink, int lane, int wave, long long hop0,
                                               int stride_r, long long limit) {
    constexpr int r = C::RL, BPL = C::BPL, LSU = C::LSU, HR = 64 / LSU, KC = BPL / LSU;
    constexpr int NU = (C::NIMG / HR) * KC, UPW = NU / C::NW;
    const int hs = HR == 1 ? 0 : lane / LSU, kl = lane - hs * LSU;
    const v2f *base = img + hs * C::IMGS + C::ph(C::pos_of(kl));
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int ui = wave * UPW + i;               // wave-uniform
        const int hg = ui / KC, c = ui - hg * KC;
        const int s = hg * HR + hs;                  // image index in the workgroup
        const long long hop = hop0 + (long long)(s / C::GH) * stride_r + (s % C::GH);
        v2f *orow = (hop < limit ? out + (size_t)hop * C::M : sink) + kl + LSU * c;
        const v2f *q = base + hg * HR * C::IMGS + C::ph(C::pos_of(LSU * c));
        v2f v[r];
#pragma unroll
        for (int j = 0; j < r; ++j) v[j] = q[j];
        bfly(v);
#pragma unroll
        for (int k = 0; k < r; ++k) orow[k * BPL] = v[k];
    }
}

// last pass in statistics-only mode: same butterflies, outputs reduced into the lane's accumulators instead of stored.
// Units chunk-major: ui = c S_NHG + hg.
template <class C>
__device__ __forceinline__ void mid_last_stats(const v2f *img, int lane, int wave, long long hop0, int stride_r,
                                               long long limit, StAcc (&acc)[C::NACC]) {
    constexpr int r = C::RL, LSU = C::LSU, HR = C::S_HR, NHG = C::S_NHG, UPW = C::S_UPW;
    const int hs = HR == 1 ? 0 : lane / LSU, kl = lane - hs * LSU;
    const v2f *base = img + hs * C::IMGS + C::ph(C::pos_of(kl));
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int ui = wave * UPW + i;               // wave-uniform
        const int c = ui / NHG, hg = ui - c * NHG;
        const int ci = UPW % NHG == 0 ? i / NHG : 0; // which of the lane's chunks (compile time)
        const int s = hg * HR + hs;
        const long long hop = hop0 + (long long)(s / C::GH) * stride_r + (s % C::GH);
        const v2f *q = base + hg * HR * C::IMGS + C::ph(C::pos_of(LSU * c));
        v2f v[r];
#pragma unroll
        for (int j = 0; j < r; ++j) v[j] = q[j];
        bfly(v);
        if (hop < limit) {
#pragma unroll
            for (int k = 0; k < r; ++k) stacc_add(acc[ci * r + k], stat_power(v[k].x, v[k].y));
        }
    }
}

// the passes of the hop images + their stores.  Wave mode: wave w < NWF owns images [w HPW, (w+1) HPW), no workgroup
// barrier inside.  Workgroup mode: every thread takes part in every pass (barriers between them), then wave w does the
// last pass of images [w HPL, (w+1) HPL).  Every wave of the workgroup must call this in workgroup mode.
template <class C, bool STATS = false>
__device__ __forceinline__ void mid_transform(v2f *img, const v4f *twp, v2f *out, v2f *sink, int tid, int wave,
                                              long long hop_g, int stride_r, long long limit,
                                              StAcc *acc = nullptr) {
    // opaque copy of the thread index: the image / twiddle / output offsets of the passes are loop invariant and would
    // otherwise be hoisted out of the group loop and parked in registers for the whole run
    int lt = C::WAVE_MODE ? (tid & 63) : tid;
    if (C::WPE >= 4 || !C::WAVE_MODE) asm volatile("" : "+v"(lt));   // (with registers to spare the hoisted offsets are cheaper)
    if constexpr (C::WAVE_MODE) {
        if (wave < C::NWF) {
            v2f *im = img + wave * C::HPW * C::IMGS;
            mid_passes_affine<C, 0, 64, C::HPW, false>(im, twp, lt);
            mid_last_wave<C>(im, out, sink, lt, wave * C::HPW, hop_g, stride_r, limit);
        }
    } else {
        mid_passes_affine<C, 0, C::TH, C::NIMG, true>(img, twp, lt);
        if constexpr (STATS) mid_last_stats<C>(img, lt & 63, wave, hop_g, stride_r, limit,
                                               *reinterpret_cast<StAcc (*)[C::NACC]>(acc));
        else if constexpr (C::LSU > 0) mid_last_units<C>(img, out, sink, lt & 63, wave, hop_g, stride_r, limit);
        else mid_last_generic<C, C::NTL, C::NIMG>(img, out, sink, lt, 0, hop_g, stride_r, limit);
    }
}

// one group of a thread's run: arm MAC + radix-4 stage of GH hops into the images, window slide, next prefetch
template <class C, int FMT>
__device__ __forceinline__ void mid_mac_group(const MidArgs &a, v2f (&wA)[C::T + C::GH], v2f (&wB)[C::T + C::GH],
                                              const v2f (&tpr_in)[C::TREG > 0 ? C::TREG / 2 : 1], const v4f *tapl, v2f *imr,
                                              v2f tw1, v2f tw2, v2f tw3, int u, long long &h, bool more, bool quad) {
    constexpr int T = C::T, GH = C::GH, Q = C::Q, HB = C::HB, IMGS = C::IMGS, TREG = C::TREG, NL4 = C::NL4;
    // the register taps stay PAIRS: without the opaque touch the compiler hoists a broadcast {t, t} copy of every tap
    // out of the group loop (two registers per tap instead of one)
    v2f tpr[TREG > 0 ? TREG / 2 : 1];
#pragma unroll
    for (int e = 0; e < TREG / 2; ++e) {
        tpr[e] = tpr_in[e];
        asm volatile("" : "+v"(tpr[e]));
    }
    {
        // the quad's LDS taps, re-read every group (opaque index: hoisted out of the loop they would be held in
        // registers through the transform phase; here they are live during the MAC only)
        int uo = u;
        asm volatile("" : "+v"(uo));
        v2f tpl[NL4 > 0 ? 2 * NL4 : 1];
#pragma unroll
        for (int k = 0; k < NL4; ++k) {
            const v4f t4 = tapl[uo * NL4 + k];
            tpl[2 * k] = __builtin_shufflevector(t4, t4, 0, 1);
            tpl[2 * k + 1] = __builtin_shufflevector(t4, t4, 2, 3);
        }
#define WH_TAPFMA(q, j, w, acc)                                                                         \
(((q) * T + (j)) < TREG ? pk_fma_bc(((q) * T + (j)) & 1, w, tpr[(((q) * T + (j)) < TREG ? ((q) * T + (j)) : 0) >> 1], acc) \
                        : pk_fma_bc(((q) * T + (j) - TREG) & 1, w, tpl[(((q) * T + (j)) >= TREG ? ((q) * T + (j) - TREG) : 0) >> 1], acc))
#pragma unroll
        for (int i = 0; i < GH; ++i) {
            // hop h+i: column u uses c_{h+i-j} = w[i + T-1 - j]; column u + HB uses w[i + T - j]
            v2f y[4] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}};
#pragma unroll
            for (int j = 0; j < T; ++j) {
                y[0] = WH_TAPFMA(0, j, wA[i + T - 1 - j], y[0]);
                y[1] = WH_TAPFMA(1, j, wB[i + T - 1 - j], y[1]);
                y[2] = WH_TAPFMA(2, j, wA[i + T - j], y[2]);
                y[3] = WH_TAPFMA(3, j, wB[i + T - j], y[3]);
            }
            bfly(y);
            v2f *L = imr + i * IMGS;
            L[0] = y[0];
            L[C::ph(Q)] = cmul3(y[1], tw1);
            L[C::ph(2 * Q)] = cmul3(y[2], tw2);
            L[C::ph(3 * Q)] = cmul3(y[3], tw3);
        }
#undef WH_TAPFMA
    }
    // slide the windows, then load the next group's blocks into the freed tail slots
#pragma unroll
    for (int i = 0; i < T; ++i) {
        wA[i] = wA[i + GH];
        wB[i] = wB[i + GH];
    }
    h += GH;
    if (more) {
#pragma unroll
        for (int i = 0; i < GH; ++i) {
            long long gb = h + 1 + i;
            if (gb > a.max_block) gb = a.max_block;
            wA[T + i] = ld_iq<FMT>(a.x, gb * HB + u);
            wB[T + i] = ld_iq<FMT>(a.x, gb * HB + u + Q);
        }
    }
}

// the workgroup's accumulators -> [M][4] float64 in LDS (the image memory, free by now) -> its row of the workspace
template <class C>
__device__ __forceinline__ void mid_stats_flush(v2f *img, StAcc (&acc)[C::NACC], double *ws_row, int tid, int wave) {
    constexpr int M = C::M, r = C::RL, LSU = C::LSU, NHG = C::S_NHG, UPW = C::S_UPW;
    double *stg = reinterpret_cast<double *>(img);                       // [4][M]: sum, sum^2, min, max
    unsigned long long *stgu = reinterpret_cast<unsigned long long *>(img);
    __syncthreads();
    for (int i = tid; i < M; i += C::NTL) {
        stg[i] = 0.0;
        stg[M + i] = 0.0;
        stgu[2 * M + i] = 0x7ff0000000000000ULL;   // +inf
        stgu[3 * M + i] = 0ULL;
    }
    __syncthreads();
    const int lane = tid & 63;
    const int hs = C::S_HR == 1 ? 0 : lane / LSU, kl = lane - hs * LSU;
    (void)hs;
#pragma unroll
    for (int ci = 0; ci < C::NCD; ++ci) {
        const int c = (wave * UPW) / NHG + ci;       // the lane's chunk(s), as dealt in mid_last_stats
#pragma unroll
        for (int k = 0; k < r; ++k) {
            const int ch = kl + LSU * c + k * C::BPL;
            StAcc &a4 = acc[ci * r + k];
            stacc_fold(a4);
            atomicAdd(&stg[ch], a4.s);
            atomicAdd(&stg[M + ch], a4.s2);
            atomicMin(&stgu[2 * M + ch], (unsigned long long)__double_as_longlong((double)a4.mn));   // p >= 0: bit patterns order like values
            atomicMax(&stgu[3 * M + ch], (unsigned long long)__double_as_longlong((double)a4.mx));
        }
    }
    __syncthreads();
    for (int i = tid; i < 4 * M; i += C::NTL) ws_row[i] = stg[i];
}

template <class C, int FMT, bool STATS = false>
__global__ __launch_bounds__(C::NTL) __attribute__((amdgpu_waves_per_eu(STATS ? C::WPE_STATS : C::WPE,
                                                                          STATS ? C::WPE_STATS : C::WPE)))
void pfb_mid_kernel(MidArgs a) {
    constexpr int M = C::M, T = C::T, R = C::R, GH = C::GH, Q = C::Q, HB = C::HB, NT = C::NTL, NIMG = C::NIMG;
    constexpr int IMGS = C::IMGS, TREG = C::TREG, NL4 = C::NL4;
    constexpr MidPlan P = C::P;

    __shared__ __attribute__((aligned(16))) v2f img[NIMG * IMGS];
    // per-pass twiddle tables, entries (wx, wy, -wy, wx): a complex product is two packed instructions
    __shared__ v4f twp[P.twn > 0 ? P.twn : 1];
    // LDS taps of quad u: the taps e = q T + j >= TREG, NL4 float4 per quad (M = 320, TREG = 8: 7 float4 = 112 bytes per
    // row, and the ds_read_b128 of 16 consecutive lanes cover all 64 banks once)
    __shared__ v4f tapl[NL4 > 0 ? Q * NL4 : 1];

    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    StAcc acc[STATS ? C::NACC : 1];
    if (STATS) {
#pragma unroll
        for (int i = 0; i < C::NACC; ++i) stacc_init(acc[i]);
    }
    // a lane visits a channel S_NHG times per group: its open float32 block is folded after at most 16 visits
    constexpr int FOLD_G = C::S_NHG >= 16 ? 1 : 16 / (C::S_NHG > 0 ? C::S_NHG : 1);

    if (bid == a.n_head) {   // history for the next call: new_hist[k][j] = block_{H-1-j}[k]
        for (int idx = tid; idx < M * T; idx += NT) {
            const int k = idx / T, j = idx - k * T;
            const long long g = a.H - 1 - j;
            v2f v;
            if (g >= 0) v = ld_iq<FMT>(a.x, g * HB + k);
            else {
                const int col = (int)(-g - 1);
                v = col < T ? a.hist[(size_t)k * T + col] : v2f{0.f, 0.f};
            }
            a.new_hist[idx] = v;
        }
        return;
    }

    // per-pass twiddle tables: twp[two_p + (k-1) m_p + u'] = W_{L_p}^(u' k) = W_M^(u' k M / L_p)
#pragma unroll
    for (int p = 0; p < P.np - 1; ++p) {
        const int rp = P.r[p], Lp = P.L[p], mp = Lp / rp;
        for (int e = tid; e < (rp - 1) * mp; e += NT) {
            const int k = e / mp + 1, up = e - (k - 1) * mp;
            const float2 w = a.tw[(up * k * (M / Lp)) % M];
            twp[P.two[p] + e] = v4f{w.x, w.y, -w.y, w.x};
        }
    }

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool quad = tid < C::NT;                       // this thread owns a quad of a run
    const int r = quad ? tid / Q : R - 1, u = quad ? tid - r * Q : Q - 1;   // (the others mirror the last quad, nothing written)

    if (bid < a.n_head) {
        // head hops: every column from the carried history / the stream, j ascending like the packed MAC below
        const long long hop_base = (long long)bid * NIMG;
        const long long limit = a.H < T - 1 ? a.H : T - 1;
        constexpr int HJ = NIMG * Q;
        for (int j0 = tid; j0 < HJ; j0 += NT) {
            const int s = j0 / Q, uu = j0 - s * Q;
            const long long hop = hop_base + s;
            if (hop >= limit) continue;
            v2f z[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = uu + q * Q;
                v2f acc = v2f{0.f, 0.f};
#pragma nounroll
                for (int j = 0; j < T; ++j) {
                    const long long gb = hop - j;
                    v2f c;
                    if (gb >= 0) {
                        long long idx = gb * HB + k;
                        if (idx >= a.n) idx = a.n - 1;
                        c = ld_iq<FMT>(a.x, idx);
                    } else {
                        c = a.hist[(size_t)k * T + (int)(-gb - 1)];
                    }
                    acc = WH_FMA(c, bc(a.arms[(size_t)k * T + j]), acc);
                }
                z[q] = acc;
            }
            bfly(z);
            const float2 w1 = a.tw[uu], w2 = a.tw[2 * uu], w3 = a.tw[3 * uu];
            v2f *L = img + s * IMGS + C::ph(uu);
            L[0] = z[0];
            L[C::ph(Q)] = cmul3(z[1], v2f{w1.x, w1.y});
            L[C::ph(2 * Q)] = cmul3(z[2], v2f{w2.x, w2.y});
            L[C::ph(3 * Q)] = cmul3(z[3], v2f{w3.x, w3.y});
        }
        __syncthreads();   // images and twiddle tables
        mid_transform<C, STATS>(img, twp, a.out, a.sink, tid, wave, hop_base, GH, limit, acc);
        if constexpr (STATS) mid_stats_flush<C>(img, acc, a.stats_ws + (size_t)bid * 4 * M, tid, wave);
        return;
    }

    const long long hop_base = (T - 1) + (long long)(bid - a.n_head - 1) * R * a.hpr;
    const int ngroups = a.hpr / GH;

    // stage-1 twiddles W_M^(u k1)
    v2f tw1, tw2, tw3;
    {
        const float2 w1 = a.tw[u], w2 = a.tw[2 * u], w3 = a.tw[3 * u];
        tw1 = v2f{w1.x, w1.y}; tw2 = v2f{w2.x, w2.y}; tw3 = v2f{w3.x, w3.y};
    }
    // taps e = q T + j: the first TREG in registers (pairs), the rest in LDS
    v2f tpr[TREG > 0 ? TREG / 2 : 1];
#pragma unroll
    for (int e = 0; e < TREG; e += 2) {
        const int q0 = e / T, j0 = e - q0 * T, q1 = (e + 1) / T, j1 = (e + 1) - q1 * T;
        tpr[e / 2] = v2f{a.arms[(size_t)(u + q0 * Q) * T + j0], a.arms[(size_t)(u + q1 * Q) * T + j1]};
        asm volatile("" : "+v"(tpr[e / 2]));   // a PAIR from here on: seen through, each tap gets a {t, t} register pair
    }
    if constexpr (NL4 > 0) {
        for (int e = tid; e < Q * NL4 * 4; e += NT) {
            const int uu = e / (NL4 * 4), f = TREG + e - uu * (NL4 * 4);   // f = q T + j
            const int q = f / T, j = f - q * T;
            reinterpret_cast<float *>(tapl)[e] = f < 4 * T ? a.arms[(size_t)(uu + q * Q) * T + j] : 0.f;
        }
    }
    // wA[i] = x[(h - (T-1) + i) HB + u], wB[i] = ... + Q; slots T.. hold the group's new blocks
    v2f wA[T + GH], wB[T + GH];
    long long h = hop_base + (long long)r * a.hpr;   // this thread's run: next hop
#pragma unroll
    for (int i = 0; i < T + GH; ++i) {
        long long g = h - (T - 1) + i;
        if (g > a.max_block) g = a.max_block;
        wA[i] = ld_iq<FMT>(a.x, g * HB + u);
        wB[i] = ld_iq<FMT>(a.x, g * HB + u + Q);
    }
    v2f *const imr = img + r * GH * IMGS + C::ph(u);   // this thread's word of its run's first image
    __syncthreads();   // twiddle and tap tables

#ifdef WH_DIAG
#define WH_MORE(g) ((g) + 1 < ngroups && !(a.ablate & 2))
#else
#define WH_MORE(g) ((g) + 1 < ngroups)
#endif
#ifdef WH_DIAG
    const long long limit = (a.ablate & 1) ? 0 : a.H;   // bit 0: every output store goes to the sink row
    const bool passes = !(a.ablate & 4);
#else
    const long long limit = a.H;
    constexpr bool passes = true;
#endif
    for (int g = 0; g < ngroups; ++g) {
        const long long hop_g = hop_base + (long long)g * GH;
        if (hop_g >= a.H) break;   // uniform: this and every later group of the workgroup is past the end
        WH_STAMP(0)
        mid_mac_group<C, FMT>(a, wA, wB, tpr, tapl, imr, tw1, tw2, tw3, u, h, WH_MORE(g), quad);
        WH_STAMP(1)
        if (C::SELF) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
        WH_STAMP(2)
        if (passes) mid_transform<C, STATS>(img, twp, a.out, a.sink, tid, wave, hop_g, a.hpr, limit, acc);
        if constexpr (STATS) {
            if ((g + 1) % FOLD_G == 0) {
#pragma unroll
                for (int i = 0; i < C::NACC; ++i) stacc_fold(acc[i]);
            }
        }
        WH_STAMP(3)
        if (C::SELF) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
    }
    if constexpr (STATS) mid_stats_flush<C>(img, acc, a.stats_ws + (size_t)bid * 4 * M, tid, wave);
}

// statistics-only mode, second step: the workgroups' rows [rows][4][M] -> d_stats [M][5] = {sum p, sum p^2, hops, min, max}
// (row `skip` belongs to the history workgroup and holds nothing).  One block of (64 channels) x (16 row slices).
__global__ __launch_bounds__(1024) void mid_stats_reduce_kernel(const double *ws, int rows, int skip, int M, double hops,
                                                               double *stats, int accumulate) {
    const int c = blockIdx.x * 64 + threadIdx.x, y = threadIdx.y;
    double s = 0, s2 = 0, mn = INFINITY, mx = -INFINITY;
    if (c < M)
        for (int r = y; r < rows; r += 16) {
            if (r == skip) continue;
            const double *p = ws + (size_t)r * 4 * M + c;
            s += p[0]; s2 += p[M];
            mn = fmin(mn, p[2 * M]); mx = fmax(mx, p[3 * M]);
        }
    __shared__ double red[4][16][64];
    red[0][y][threadIdx.x] = s; red[1][y][threadIdx.x] = s2; red[2][y][threadIdx.x] = mn; red[3][y][threadIdx.x] = mx;
    __syncthreads();
    if (y == 0 && c < M) {
        for (int k = 1; k < 16; ++k) {
            s += red[0][k][threadIdx.x]; s2 += red[1][k][threadIdx.x];
            mn = fmin(mn, red[2][k][threadIdx.x]); mx = fmax(mx, red[3][k][threadIdx.x]);
        }
        double *o = stats + (size_t)c * 5;
        // (a row with count 0 holds no observation: accumulate == overwrite, as in pfb_stats_final_kernel)
        if (accumulate && o[2] > 0.0) { o[0] += s; o[1] += s2; o[2] += hops; o[3] = fmin(o[3], mn); o[4] = fmax(o[4], mx); }
        else { o[0] = s; o[1] = s2; o[2] = hops; o[3] = mn; o[4] = mx; }
    }
}

#ifdef WH_DIAG
long long *g_diag_stamps = nullptr;
int g_diag_stamp_wg = 0;
#endif

// ---- host side -------------------------------------------------------------------------------------------------

// (M, R, GH, NWF, WPE, PB, PADN, IMGX, TREG, LSP): see MidCfg
#ifndef WH_MID_320
#define WH_MID_320 3, 4, 0, 3, 20, 1, 0, 36, 0, 0
#endif
#define WH_MID_X(X, ...) X(__VA_ARGS__)
// Shapes found by tools/pfb_mid_configs.py (lane utilisation, LDS footprint, simulated bank conflicts).  Last column R8:
// radix-8 passes for the power-of-two part (tools/pfb_mid_configs.py --r8) where fewer LDS passes measured faster
// (2^26 samples: M = 2048 552 -> 446 us, 4096 697 -> 553, 800 482 -> 454, 640 431 -> 399, 480 438 -> 409, 384 417 -> 406, 512 367 -> 356; no gain for
// M <= 256, which run at the memory system's pace already; 768 / 1280 have no affine radix-8 shape at these
// workgroup sizes; 4096: 697 -> 553 us with 20 of its taps in registers, 3 spilled dwords in the complex64 form):
//      64: plan [4, 4] waves 4 lanes/hop 4 util 1.00 pass-util 1.00 last-pass lanes/hop 16 LDS 35008 B conflicts rd x1.50 wr x1.00
//      80: plan [4, 5] waves 4 lanes/hop 5 util 0.94 pass-util 0.94 last-pass lanes/hop 16 LDS 32880 B conflicts rd x2.90 wr x1.00
//      96: plan [4, 2, 3] waves 4 lanes/hop 6 util 0.94 pass-util 0.94 last-pass lanes/hop 32 LDS 32976 B conflicts rd x1.66 wr x1.31
//     128: plan [4, 4, 2] waves 4 lanes/hop 8 util 1.00 pass-util 1.00 last-pass lanes/hop 64 LDS 34784 B conflicts rd x1.67 wr x1.67
//     160: plan [4, 2, 5] waves 4 lanes/hop 10 util 0.94 pass-util 0.94 last-pass lanes/hop 32 LDS 32240 B conflicts rd x1.96 wr x1.50
//     192: plan [4, 4, 3] waves 4 lanes/hop 12 util 0.94 pass-util 0.94 last-pass lanes/hop 64 LDS 33360 B conflicts rd x2.04 wr x1.31
//     240: plan [4, 3, 5] waves 4 lanes/hop 10 util 0.94 pass-util 0.62 last-pass lanes/hop 16 LDS 32496 B conflicts rd x1.88 wr x1.58
//     256: plan [4, 4, 4] waves 4 lanes/hop 16 util 1.00 pass-util 1.00 last-pass lanes/hop 64 LDS 37824 B conflicts rd x2.00 wr x1.33
//     384: plan [8, 4, 3] waves 8 lanes/hop 24 util 0.94 pass-util 0.94 last-pass lanes/hop 64 LDS 64848 B conflicts rd x1.96 wr x1.62
//     400: plan [4, 5, 5] waves 5 lanes/hop 20 util 0.94 pass-util 0.75 last-pass lanes/hop 16 LDS 40304 B conflicts rd x1.62 wr x1.35
//     480: plan [8, 3, 5] waves 4 lanes/hop 20 util 0.94 pass-util 0.62 last-pass lanes/hop 32 LDS 45104 B conflicts rd x1.59 wr x1.58
//     512: plan [8, 4, 4] waves 4 lanes/hop 32 util 1.00 pass-util 1.00 last-pass lanes/hop 64 LDS 38848 B conflicts rd x1.67 wr x1.33
//     640: plan [8, 4, 5] waves 8 lanes/hop 40 util 0.94 pass-util 0.94 last-pass lanes/hop 64 LDS 67376 B conflicts rd x1.57 wr x1.44
//     768: plan [4, 4, 4, 3] waves 3 (R = 1, GH = 3; measured +11 % over the 12-wave shape R = 4) lanes/hop 64 util 1.00 LDS 22224 B conflicts rd x2.25 wr x1.50
//     800: plan [8, 5, 5] waves 4 (R = 1, GH = 8: 482 -> 454 us against the 10-wave shape R = 3) lanes/hop 20 util 0.78 pass-util 0.62 last-pass lanes/hop 32 LDS 54320 B conflicts rd x1.59 wr x1.54
//     960: plan [4, 4, 3, 5] waves 4 lanes/hop 40 util 0.94 pass-util 0.62 last-pass lanes/hop 64 LDS 46896 B conflicts rd x1.83 wr x1.69
//    1280: plan [4, 4, 4, 5] waves 5 lanes/hop 64 util 1.00 pass-util 1.00 last-pass lanes/hop 64 LDS 57520 B conflicts rd x2.25 wr x1.50
//          (held to 168 VGPRs since round 3: at 177 a CU took ONE five-wave workgroup; two of them: 497 -> 486 us per 2^26 samples)
//    2048: plan [8, 8, 8] waves 8 lanes/hop 128 util 1.00 pass-util 1.00 last-pass lanes/hop 64 LDS 74624 B conflicts rd x2.00 wr x1.33
//    4096: plan [8, 8, 4, 4] waves 16 lanes/hop 512 util 1.00 pass-util 1.00 last-pass lanes/hop 64 LDS 147456 B (20 register taps: the
//          1024-thread workgroup has 128 registers per lane; one workgroup per CU) conflicts rd x3.75 wr x1.75
#define WH_MID_CONFIGS(X) \
    WH_MID_X(X, 320, WH_MID_320) \
    X(1024, 1, 4, 0, 3, 32, 2, 0, 36, 0, 1) \
    X(64, 16, 4, 0, 3, 16, 1, 0, 36, 0, 0) \
    X(80, 12, 4, 0, 3, 20, 1, 1, 36, 0, 0) \
    X(96, 10, 4, 0, 3, 24, 1, 2, 36, 0, 0) \
    X(128, 8, 4, 0, 3, 32, 1, 2, 36, 0, 0) \
    X(160, 6, 4, 0, 3, 40, 1, 1, 36, 0, 0) \
    X(192, 5, 4, 0, 3, 48, 2, 4, 36, 0, 0) \
    X(240, 4, 4, 0, 3, 60, 1, 3, 36, 10, 0) \
    X(256, 4, 4, 0, 3, 16, 2, 0, 36, 0, 0) \
    X(384, 5, 4, 0, 2, 96, 2, 4, 36, 0, 1) \
    X(400, 3, 4, 0, 2, 100, 1, 0, 36, 20, 0) \
    X(480, 2, 4, 0, 3, 5, 2, 4, 36, 20, 1) \
    X(512, 2, 4, 0, 3, 16, 2, 0, 36, 0, 1) \
    X(640, 3, 4, 0, 2, 20, 1, 4, 36, 0, 1) \
    X(768, 1, 3, 0, 3, 48, 2, 0, 36, 0, 0) \
    X(800, 1, 8, 0, 2, 0, 0, 0, 36, 20, 1) \
    X(960, 1, 4, 0, 3, 5, 2, 4, 36, 40, 0) \
    X(1280, 1, 5, 0, 3, 80, 2, 0, 36, 0, 0) \
    X(2048, 1, 4, 0, 2, 64, 1, 0, 36, 0, 1) \
    X(4096, 1, 2, 0, 2, 1024, 1, 0, 20, 0, 1)

template <class C>
int mid_launch_t(const PfbMidCall &c, hipStream_t st, long long *grid_out) {
    constexpr int M = C::M, T = C::T, R = C::R, GH = C::GH, NT = C::NTL, NIMG = C::NIMG;
    static int wg_per_cu[2] = {0, 0};   // resident workgroups per CU of the two format instances
    const int f = c.fmt == 1 ? 1 : 0;
    if (c.stats_only && !C::STATS_OK) return set_err(WH_E_ARG, "wh_pfb_run_stats: no statistics-only form for M=%d", C::M);
    auto kern = f ? pfb_mid_kernel<C, 1> : pfb_mid_kernel<C, 0>;
    if (wg_per_cu[f] == 0) {
        // by registers: the kernel is compiled for WPE waves per SIMD (4 SIMDs per CU); by LDS: 160 KiB per CU
        hipFuncAttributes fa;
        WH_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)));
        const int by_waves = 4 * C::WPE / (NT / 64);
        const int by_lds = (int)((160 * 1024) / (((size_t)fa.sharedSizeBytes + 511) / 512 * 512));
        const int nb = by_waves < by_lds ? by_waves : by_lds;
        wg_per_cu[f] = nb > 0 ? nb : 1;
    }
    MidArgs a;
    a.x = c.x; a.arms = c.arms; a.tw = c.tw;
    a.hist = reinterpret_cast<const v2f *>(c.hist); a.new_hist = reinterpret_cast<v2f *>(c.new_hist);
    a.out = reinterpret_cast<v2f *>(c.out); a.sink = reinterpret_cast<v2f *>(c.sink);
    a.H = c.H; a.n = (long long)c.n;
    a.max_block = (long long)(c.n / (size_t)(M / 2)) - 1;
    const long long nh = c.H - (T - 1);   // hops of the runs
    int hpr = GH;
    long long n_main = 0;
    if (nh > 0) {
        // Run length: the launch takes rounds(hpr) x (hpr + fixed cost of a run, about 12 hops: window fill, tables), with
        // rounds = workgroups / resident slots rounded UP -- a nearly empty last round costs as much as a full one, so
        // small calls get every workgroup resident at once.  16..128 hops (halo 9 / hpr of the reads).
        const long long slots = (long long)c.cu_count * wg_per_cu[f];
        long long v = 16, best = -1;
        for (long long cand = (16 + GH - 1) / GH * GH; cand <= 128; cand += GH) {
            const long long wgs = ((nh + cand - 1) / cand + R - 1) / R;
            const long long cost = ((wgs + slots - 1) / slots) * (cand + 12);
            if (best < 0 || cost <= best) { best = cost; v = cand; }
        }
        if (c.hops_per_run > 0) v = (c.hops_per_run + GH - 1) / GH * GH;
        hpr = (int)v;
        const long long runs = (nh + hpr - 1) / hpr;
        n_main = (runs + R - 1) / R;
    }
    a.hpr = hpr;
#ifdef WH_DIAG
    a.ablate = c.stats_only >> 8;   // diagnostics build: wh_pfb_tune(key 4) rides in the upper bits of stats_only
    a.stamps = g_diag_stamps;
    a.stamp_wg = g_diag_stamp_wg;
#endif
    const long long head_hops = c.H < T - 1 ? c.H : T - 1;
    a.n_head = (int)((head_hops + NIMG - 1) / NIMG);
    const long long grid = a.n_head + 1 + n_main;
    if (grid > 0x7fffffffLL) return set_err(WH_E_ARG, "wh_pfb_run: input too long for one launch");
    if (grid_out) {           // plan only: the caller sizes the statistics workspace from it
        *grid_out = grid;
        return WH_OK;
    }
    a.stats_ws = c.stats_ws;
    if (c.stats_only & 1) {
        if constexpr (C::STATS_OK) {
            if (!c.stats_ws || !c.stats_out) return set_err(WH_E_ARG, "wh_pfb_run_stats: null statistics buffer");
            auto skern = f ? pfb_mid_kernel<C, 1, true> : pfb_mid_kernel<C, 0, true>;
            hipLaunchKernelGGL(skern, dim3((unsigned)grid), dim3(NT), 0, st, a);
            WH_LAUNCH_CHECK();
            hipLaunchKernelGGL(mid_stats_reduce_kernel, dim3((M + 63) / 64), dim3(64, 16), 0, st, c.stats_ws, (int)grid,
                               a.n_head, M, (double)c.H, c.stats_out, c.stats_accumulate);
            WH_LAUNCH_CHECK();
            return WH_OK;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), 0, st, a);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

// ---- spectrum frames through the same passes (reference dsp/fft/scipy_backend.py:38-79: window -> FFT -> fftshift ->
// 20 log10(|X| + 1e-10)).  A frame is an image: thread (r, u) windows the quad u, u + N/4, u + N/2, u + 3N/4 of GH frames,
// does the radix-4 stage in registers and writes the images; the last pass turns its outputs into dB and stores them at
// the fftshift-ed bin.  12 bytes of HBM traffic per sample (8 in, 4 out); the next round's samples are in flight during
// the passes.
struct SpecArgs {
    const v2f *x;
    size_t frame_stride;    // samples between frame starts
    float *out;             // [n_frames][N]
    float *sink;            // [N] nobody reads (frames past the end of the call)
    const float *window;    // [N]
    const float2 *tw;       // exp(-2 pi i m / N)
    long long n_frames;
};

template <class C>
__device__ __forceinline__ void spec_load(const SpecArgs &a, v2f (&cur)[C::GH][4], long long f0, int u) {
#pragma unroll
    for (int i = 0; i < C::GH; ++i) {
        long long f = f0 + i;
        if (f >= a.n_frames) f = a.n_frames - 1;
        const v2f *px = a.x + (size_t)f * a.frame_stride + u;
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[i][q] = px[q * C::Q];
    }
}

// last pass, spectrum form: bin f = kb + k BPL of the transform lands at (f + N/2) mod N = kb + ((k + RL/2) mod RL) BPL
template <class C>
__device__ __forceinline__ void mid_last_units_db(const v2f *img, float *out, float *sink, int lane, int wave,
                                                  long long f0, long long limit) {
    constexpr int r = C::RL, BPL = C::BPL, LSU = C::LSU, HR = 64 / LSU, KC = BPL / LSU;
    constexpr int NU = (C::NIMG / HR) * KC, UPW = NU / C::NW;
    static_assert(r % 2 == 0, "fftshift by a whole number of last-pass strides");
    const int hs = HR == 1 ? 0 : lane / LSU, kl = lane - hs * LSU;
    const v2f *base = img + hs * C::IMGS + C::ph(C::pos_of(kl));
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int ui = wave * UPW + i;               // wave-uniform
        const int hg = ui / KC, c = ui - hg * KC;
        const long long frame = f0 + hg * HR + hs;
        float *orow = (frame < limit ? out + (size_t)frame * C::M : sink) + kl + LSU * c;
        const v2f *q = base + hg * HR * C::IMGS + C::ph(C::pos_of(LSU * c));
        v2f v[r];
#pragma unroll
        for (int j = 0; j < r; ++j) v[j] = q[j];
        bfly(v);
#pragma unroll
        for (int k = 0; k < r; ++k) {
            // |X| + 1e-10 >= 1e-10 is a normal float: v_sqrt_f32 / v_log_f32 (1 ulp each; 20 log10 m = 20 log10(2) log2 m)
            const float mag = __builtin_amdgcn_sqrtf(v[k].x * v[k].x + v[k].y * v[k].y) + 1e-10f;
            orow[((k + r / 2) % r) * BPL] = 6.02059991327962390427f * __builtin_amdgcn_logf(mag);
        }
    }
}

template <class C>
__global__ __launch_bounds__(C::NTL) void spectrum_mid_kernel(SpecArgs a) {
    constexpr int M = C::M, GH = C::GH, Q = C::Q, NT = C::NTL, NIMG = C::NIMG, IMGS = C::IMGS;
    constexpr MidPlan P = C::P;
    static_assert(!C::WAVE_MODE && C::LSU > 0 && C::NT == C::NTL, "spectrum: workgroup-mode shapes with whole waves");
    __shared__ __attribute__((aligned(16))) v2f img[NIMG * IMGS];
    __shared__ v4f twp[P.twn > 0 ? P.twn : 1];
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < P.np - 1; ++p) {
        const int rp = P.r[p], Lp = P.L[p], mp = Lp / rp;
        for (int e = tid; e < (rp - 1) * mp; e += NT) {
            const int k = e / mp + 1, up = e - (k - 1) * mp;
            const float2 w = a.tw[(up * k * (M / Lp)) % M];
            twp[P.two[p] + e] = v4f{w.x, w.y, -w.y, w.x};
        }
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = tid / Q, u = tid - r * Q;
    v2f tw1, tw2, tw3;
    {
        const float2 w1 = a.tw[u], w2 = a.tw[2 * u], w3 = a.tw[3 * u];
        tw1 = v2f{w1.x, w1.y}; tw2 = v2f{w2.x, w2.y}; tw3 = v2f{w3.x, w3.y};
    }
    float win[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) win[q] = a.window[u + q * Q];
    const long long ngroups = (a.n_frames + NIMG - 1) / NIMG;
    long long fg = blockIdx.x;
    v2f cur[GH][4];
    spec_load<C>(a, cur, fg * NIMG + r * GH, u);
    v2f *const imr = img + r * GH * IMGS + C::ph(u);
    __syncthreads();   // twiddle tables
    for (; fg < ngroups; fg += gridDim.x) {
#pragma unroll
        for (int i = 0; i < GH; ++i) {
            v2f y[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) y[q] = cur[i][q] * bc(win[q]);
            bfly(y);
            v2f *L = imr + i * IMGS;
            L[0] = y[0];
            L[C::ph(Q)] = cmul3(y[1], tw1);
            L[C::ph(2 * Q)] = cmul3(y[2], tw2);
            L[C::ph(3 * Q)] = cmul3(y[3], tw3);
        }
        const long long nxt = fg + gridDim.x;
        if (nxt < ngroups) spec_load<C>(a, cur, nxt * NIMG + r * GH, u);
        __syncthreads();
        int lt = tid;
        asm volatile("" : "+v"(lt));
        mid_passes_affine<C, 0, C::TH, C::NIMG, true>(img, twp, lt);
        mid_last_units_db<C>(img, a.out, a.sink, lt & 63, wave, fg * NIMG, a.n_frames);
        __syncthreads();
    }
}

template <class C>
int spectrum_launch_t(const wh::SpectrumMidCall &c, hipStream_t st) {
    static int wg_per_cu = 0;
    auto kern = spectrum_mid_kernel<C>;
    if (wg_per_cu == 0) {
        int nb = 0;
        WH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kern), C::NTL, 0));
        wg_per_cu = nb > 0 ? nb : 1;
    }
    SpecArgs a;
    a.x = reinterpret_cast<const v2f *>(c.x); a.frame_stride = c.frame_stride; a.out = c.out; a.sink = c.sink;
    a.window = c.window; a.tw = c.tw; a.n_frames = c.n_frames;
    const long long ngroups = (c.n_frames + C::NIMG - 1) / C::NIMG;
    const long long slots = (long long)c.cu_count * wg_per_cu;
    // whole rounds: every workgroup walks the same number of frame groups (a partial last round costs a full one)
    const long long rounds = (ngroups + slots - 1) / slots;
    const long long grid = (ngroups + rounds - 1) / rounds;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NTL), 0, st, a);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

// ---- N = 4096 spectrum frames as 16 x 16 x 16 (the shaped instance took four trips through LDS -- radix 4 in registers,
// then 8, 8, 4, 4 -- in one 1024-thread workgroup per CU: 3.3 TB/s).  Here a 256-thread workgroup owns a frame: thread
// t = 16 n2 + n3 loads x[t + 256 n1], n1 < 16 (every load instruction of a wave reads 512 consecutive bytes), windows them and
// does the radix-16 stage over n1 in registers; two more radix-16 stages follow through ONE padded LDS image
// (n = 256 n1 + 16 n2 + n3, k = k1 + 16 k2 + 256 k3):
//   image 1 [k1][t]       at k1 * 272 + t             (stage-1 results times W4096^(t k1))
//   image 2 [k2][k1][n3]  at k2 * 272 + k1 * 17 + n3  (stage-2 results times W256^(n3 k2))
// both conflict-free for their 8-byte accesses (272 = 16 mod 32 complex: the two k1 / k2 of a 32-lane group fall on the two
// halves of the banks; 17 k1 mod 32 is distinct over k1 < 16 and leaves exactly the residues that + 16 fills).  Thread r of
// stage 3 holds X[r + 256 k3]: every store instruction of a wave writes 64 consecutive bins.  The next frame's 16 samples
// per thread are loaded right after the current ones are consumed and stay in flight across stages 2 and 3 (registers:
// 30 stage-1 twiddles + 16 window values + 32 samples + 32 prefetched + the butterflies' temporaries = two workgroups per
// CU; without the prefetch the kernel fits three per CU and measured 219 us per 2^26 samples against 175 with it -- memory
// parallelism, not occupancy, is what a 12-byte-per-sample kernel needs; 246 us before).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void spectrum4096_kernel(SpecArgs a) {
    constexpr int N = 4096, B1 = 272;
    __shared__ __attribute__((aligned(16))) float2 img[16 * B1];
    __shared__ float2 tw256[256];   // [k2][n3] = W256^(n3 k2)
    const int t = threadIdx.x;
    const float2 *twt = reinterpret_cast<const float2 *>(a.tw);
    float2 tw1[15];
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) tw1[k1 - 1] = twt[(t * k1) & (N - 1)];
    float win[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) win[n1] = a.window[t + 256 * n1];
    tw256[t] = twt[(16 * (t & 15) * (t >> 4)) & (N - 1)];
    __syncthreads();
    float2 nx[16];                                  // the next frame's samples, in flight across stages 2 and 3
    {
        const float2 *px = reinterpret_cast<const float2 *>(a.x) + (size_t)blockIdx.x * a.frame_stride + t;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) nx[n1] = px[256 * n1];
    }
    for (long long f = blockIdx.x; f < a.n_frames; f += gridDim.x) {
        float2 v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = make_float2(nx[n1].x * win[n1], nx[n1].y * win[n1]);
        {
            long long fn = f + gridDim.x;           // (the last round re-reads its own frame: nobody uses it)
            if (fn >= a.n_frames) fn = f;
            const float2 *px = reinterpret_cast<const float2 *>(a.x) + (size_t)fn * a.frame_stride + t;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) nx[n1] = px[256 * n1];
        }
        fft16(v);                                   // over n1: v[k1]
        img[t] = v[0];
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) img[k1 * B1 + t] = cmul(v[k1], tw1[k1 - 1]);
        __syncthreads();
        {
            const int k1 = t >> 4, n3 = t & 15;
            const float2 *src = img + k1 * B1 + n3;
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) v[n2] = src[16 * n2];
            fft16(v);                               // over n2: v[k2]
            __syncthreads();                        // image 1 has been read by everybody: image 2 overwrites it
            float2 *dst = img + k1 * 17 + n3;
            dst[0] = v[0];
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2) dst[k2 * B1] = cmul(v[k2], tw256[k2 * 16 + n3]);
        }
        __syncthreads();
        {
            const float2 *row = img + (t >> 4) * B1 + (t & 15) * 17;   // k2 = t >> 4, k1 = t & 15
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = row[j];
        }
        __syncthreads();                            // image 2 has been read: the next frame's image 1 may land
        fft16(v);                                   // over n3: v[k3] = X[t + 256 k3]
        float *o = a.out + (size_t)f * N;
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) {
            // |X| + 1e-10 >= 1e-10 is a normal float: v_sqrt_f32 / v_log_f32 (1 ulp each; 20 log10 m = 20 log10(2) log2 m)
            const float mag = __builtin_amdgcn_sqrtf(v[k3].x * v[k3].x + v[k3].y * v[k3].y) + 1e-10f;
            o[(t + 256 * k3 + N / 2) & (N - 1)] = 6.02059991327962390427f * __builtin_amdgcn_logf(mag);
        }
    }
}

int spectrum4096_launch(const wh::SpectrumMidCall &c, hipStream_t st) {
    static int wg_per_cu = 0;
    if (wg_per_cu == 0) {
        int nb = 0;
        WH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(spectrum4096_kernel), 256, 0));
        wg_per_cu = nb > 0 ? nb : 1;
    }
    SpecArgs a;
    a.x = reinterpret_cast<const v2f *>(c.x); a.frame_stride = c.frame_stride; a.out = c.out; a.sink = c.sink;
    a.window = c.window; a.tw = c.tw; a.n_frames = c.n_frames;
    // whole rounds: every workgroup walks the same number of frames (a partial last round costs a full one)
    const long long slots = (long long)c.cu_count * wg_per_cu;
    const long long rounds = (c.n_frames + slots - 1) / slots;
    const long long grid = (c.n_frames + rounds - 1) / rounds;
    hipLaunchKernelGGL(spectrum4096_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

}  // namespace

#ifdef WH_DIAG
extern "C" void wh_diag_mid_stamps(long long *d_buf, int wg) { g_diag_stamps = d_buf; g_diag_stamp_wg = wg; }
#endif

namespace wh {

// Other arm lengths (taps_per_channel != 9; every reference call site uses the default 9, the constructor takes any):
// the two headline channel counts at T = 5, 7, 13, 17.  The arm windows are T + GH - 1 registers per quad element, so
// the long arms run at 2 waves per SIMD and keep only part of their taps in registers.
//        M     T  R GH NWF WPE PB PADN IMGX TREG LSP
#define WH_MID_CONFIGS_T(X) \
    X(320, 5, 3, 4, 0, 3, 20, 1, 0, 20, 0) \
    X(320, 7, 3, 4, 0, 3, 20, 1, 0, 28, 0) \
    X(320, 13, 3, 4, 0, 2, 20, 1, 0, 36, 0) \
    X(320, 17, 3, 4, 0, 2, 20, 1, 0, 20, 0) \
    X(1024, 5, 1, 4, 0, 3, 16, 2, 0, 20, 0) \
    X(1024, 7, 1, 4, 0, 3, 16, 2, 0, 28, 0) \
    X(1024, 13, 1, 4, 0, 2, 16, 2, 0, 36, 0) \
    X(1024, 17, 1, 4, 0, 2, 16, 2, 0, 36, 0)

bool pfb_mid_supported(int M, int T) {
#define X(M_, T_, R_, GH_, NWF_, WPE_, PB_, PADN_, IMGX_, TREG_, LSP_) if (M == M_ && T == T_) return true;
    WH_MID_CONFIGS_T(X)
#undef X
    if (T != 9) return false;
#define X(M_, R_, GH_, NWF_, WPE_, PB_, PADN_, IMGX_, TREG_, LSP_, R8_) if (M == M_) return true;
    WH_MID_CONFIGS(X)
#undef X
    return false;
}

int pfb_mid_launch(int M, int T, const PfbMidCall &c, hipStream_t st, long long *grid_out) {
    // every store of a hop past the end of the call -- and, in a DIAG build's ablation mode, EVERY store -- lands in the
    // sink row at [0, M): refuse a launch whose sink is missing or shorter than a row (the two memory-access faults of
    // round 2 came from a work-in-progress build of exactly that path)
    if (!grid_out && (!c.sink || c.sink_elems < (size_t)M))
        return set_err(WH_E_ARG, "pfb_mid_launch: sink row missing or shorter than M=%d", M);
    if (!grid_out && !(c.stats_only & 1) && !c.out) return set_err(WH_E_ARG, "pfb_mid_launch: null output");
    if (T == 9) {
#define X(M_, R_, GH_, NWF_, WPE_, PB_, PADN_, IMGX_, TREG_, LSP_, R8_) \
    if (M == M_) return mid_launch_t<MidCfg<M_, 9, R_, GH_, NWF_, WPE_, PB_, PADN_, IMGX_, TREG_, LSP_, R8_>>(c, st, grid_out);
        WH_MID_CONFIGS(X)
#undef X
    }
#define X(M_, T_, R_, GH_, NWF_, WPE_, PB_, PADN_, IMGX_, TREG_, LSP_) \
    if (M == M_ && T == T_) return mid_launch_t<MidCfg<M_, T_, R_, GH_, NWF_, WPE_, PB_, PADN_, IMGX_, TREG_, LSP_>>(c, st, grid_out);
    WH_MID_CONFIGS_T(X)
#undef X
    return set_err(WH_E_ARG, "pfb_mid_launch: no instance for M=%d T=%d", M, T);
}

// spectrum frame sizes with a shaped kernel (layouts as the filterbank's at the same size; T = 1: no arms)
//        N    R GH NWF WPE PB PADN IMGX TREG LSP R8
#define WH_SPEC_CONFIGS(X) \
    X(256, 4, 4, 0, 3, 8, 1, 0, 0, 0, 1) \
    X(512, 2, 4, 0, 3, 16, 2, 0, 0, 0, 1) \
    X(1024, 1, 4, 0, 3, 32, 2, 0, 0, 0, 1) \
    X(2048, 1, 4, 0, 2, 64, 1, 0, 0, 0, 1)

bool spectrum_mid_supported(int N) {
    if (N == 4096) return true;   // spectrum4096_kernel
#define X(N_, R_, GH_, NWF_, WPE_, PB_, PADN_, IMGX_, TREG_, LSP_, R8_) if (N == N_) return true;
    WH_SPEC_CONFIGS(X)
#undef X
    return false;
}

int spectrum_mid_launch(int N, const SpectrumMidCall &c, hipStream_t st) {
    if (!c.sink || c.sink_elems < (size_t)N || !c.out)
        return set_err(WH_E_ARG, "spectrum_mid_launch: null output, or sink row missing / shorter than N=%d", N);
    if (N == 4096) return spectrum4096_launch(c, st);
#define X(N_, R_, GH_, NWF_, WPE_, PB_, PADN_, IMGX_, TREG_, LSP_, R8_) \
    if (N == N_) return spectrum_launch_t<MidCfg<N_, 1, R_, GH_, NWF_, WPE_, PB_, PADN_, IMGX_, TREG_, LSP_, R8_>>(c, st);
    WH_SPEC_CONFIGS(X)
#undef X
    return set_err(WH_E_ARG, "spectrum_mid_launch: no instance for N=%d", N);
}

int pfb_stats_rows_reduce(const double *ws, int rows, int skip, int M, double hops, double *stats, int accumulate,
                          hipStream_t st) {
    hipLaunchKernelGGL(mid_stats_reduce_kernel, dim3((M + 63) / 64), dim3(64, 16), 0, st, ws, rows, skip, M, hops, stats,
                       accumulate);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

const char *pfb_mid_kernel_name() { return "pfb_mid_kernel"; }

}  // namespace wh
