// 2x-oversampled polyphase filterbank (reference dsp/channelizer.py:28-158) for gfx950.
//
// Math (reference quirks kept):  block_g[k] = x[g*M/2 + k], k in [0,M);
//   y_h[k] = sum_{j<T} block_{h-j}[k] * arms[k][j]      (block_g for g<0 = carried history col -g-1)
//   out[h][c] = sum_k y_h[k] * exp(-2 pi i k c / M)     (forward, unnormalised)
//
// Fast path (M = 1024, T = 9): one fused kernel, HBM traffic = read x once (+halo) and write
// out once = 24 B per input sample.  A 256-thread workgroup walks a run of consecutive hops (default since round 3: three
// workgroups per CU, runs of 16-48 hops dealt so that the chip sweeps one band of addresses, odd runs walked downwards so
// that the halo is an L2 hit, the next group's samples prefetched by the LDS DMA -- see pfb1024_body).
//   * arm MAC: thread t owns columns k0 = t and t+256 of the half-block; because block_g[k0+512]
//     == block_{g+1}[k0], one sliding register window c_g = x[g*512+k0] serves both halves, so
//     every input sample is loaded once per workgroup (coalesced 8 B/lane) and the 36 tap
//     values live in registers -- no LDS in this phase.
//   * FFT-1024 = 4 x 16 x 16: the radix-4 stage needs y[t], y[t+256], y[t+512], y[t+768], which
//     is exactly what thread t holds -> done in registers; then two radix-16 stages with one wave
//     per hop (4 hops in flight per workgroup), exchanging through padded LDS images
//     (conflict-free ds_write_b64 / ds_read_b64 / ds_read_b128).
// Run path (4 | M, 64 <= M <= 512, T = 9; M = 320 is the reference's benchmark_dsp.py shape): pfb_run_kernel, one
// wave per run of hops -- the same register-window MAC, first radix-4 pass in registers, remaining Stockham passes
// in place in the wave's LDS image, last pass straight to HBM (see the comment above the kernel).
// Generic path (any even M): one workgroup per hop, MAC into LDS, then mixed-radix (4/2/3/5) Stockham passes in LDS,
// or the direct DFT for a count with another prime factor.  Also used for the first T-1 hops of every call (they
// read the carried history) and the tail hops of the fast path.
#include "pfb_internal.h"
#include "wh_common.h"
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <vector>

using namespace wh;

namespace {

constexpr int FM = 1024;       // fast-path channel count
constexpr int FHOP = 512;
constexpr int FT = 9;
constexpr int GH = 4;          // hops per group (= waves per workgroup)
constexpr int LDS1_K1 = 272;   // stage-1 image: idx = k1*272 + n2*16 + n3   (complex units)
// stage-2 image: idx = (k2*4 + k1)*LDS2_ROW + n3, LDS2_ROW = 18 (17 in the three-workgroups-per-CU form); a hop image is
// LDS_HOP = max(4*272, 64*LDS2_ROW) complex (both set in pfb1024_body)

// input accessor: complex64 (FMT 0) or interleaved int16 IQ (FMT 1, A1 unpack rule int16/32768)
template <int FMT>
__device__ __forceinline__ float2 ld_iq(const void *p, long long i) {
    if (FMT == 1) {
        short2 v = reinterpret_cast<const short2 *>(p)[i];
        return make_float2((float)v.x * (1.0f / 32768.0f), (float)v.y * (1.0f / 32768.0f));
    }
    return reinterpret_cast<const float2 *>(p)[i];
}
// the tuned kernel's loader: int16 samples come back UNSCALED (int16 -> float32 only) -- its taps carry the 1 / 32768 of
// the A1 unpack rule instead.  Scaling by a power of two is exact on either factor and an FMA rounds once, so
// fma(x / 32768, tap, acc) and fma(x, tap / 32768, acc) are the same bits: 16 multiplies per group and thread saved
template <int FMT>
__device__ __forceinline__ float2 ld_iq_raw(const void *p, long long i) {
    if (FMT == 1) {
        short2 v = reinterpret_cast<const short2 *>(p)[i];
        return make_float2((float)v.x, (float)v.y);
    }
    return reinterpret_cast<const float2 *>(p)[i];
}
__device__ __forceinline__ float2 ld_iq_rt(const void *p, int fmt, long long i) {
    return fmt == 1 ? ld_iq<1>(p, i) : ld_iq<0>(p, i);
}

struct PfbFastArgs {
    const void *x;          // input samples (complex64 or int16 pairs)
    float2 *out;            // [hops][1024]
    const float *arms;      // float32 [1024][9]
    const float2 *tw1024;   // exp(-2 pi i m/1024)
    long long first_hop;    // first hop handled by the fast kernel (>= 8)
    long long n_groups;     // total groups of GH hops
    int groups_per_wg;
    int n_wg;
    int alt_dir;            // odd runs walk downwards (three-workgroups form; see pfb1024_body)
    int map_chunk;          // run -> workgroup mapping: -2 = one contiguous range of runs per XCD, -1 = run = workgroup, C > 0 = chunks of C
    long long max_block;    // last half-block (512 samples) fully inside the input: prefetches past it are clamped
    double *stats_ws;       // STATS: one row [4][1024] float64 per workgroup (sum p, sum p^2, min, max)
#ifdef WH_DIAG
    int ablate;             // diagnostics build only (WH_PFB_ABLATE): 1 = suppress stores
#endif
};

// GLDS: the next group's samples are prefetched into LDS by the DMA path (global_load_lds, inline asm: no register
// results, so nothing the register allocator could copy early) and waited for with a COUNTED s_waitcnt: completion is
// reported in issue order (tools/ubench/vmcnt_order.hip), so vmcnt(8) -- the eight output stores issued after the
// prefetch -- retires the prefetch and leaves the stores in flight.  With ordinary loads hipcc's wait for the
// prefetched registers is vmcnt(0): every group drained its predecessor's stores before its arm MAC could finish.
// STATS: statistics-only mode (A13, the scanner's pass): stage 3 turns its 16 outputs per lane into float32 powers, the four
// hops of a group meet in LDS (each wave's finished image is free by then) and thread t folds them into the accumulators
// of its four channels t + 256 q; nothing is stored but one [4][1024] row per workgroup at the end.
// W3: the form held to THREE workgroups per CU (3 waves per SIMD, <= 168 VGPRs, <= 53 KB of LDS): the stage-2 image rows are
// 17 complex apart instead of 18 (conflict-free for 8-byte reads: 34 l mod 64 is distinct over 32 lanes; stage 3 then reads
// 16 x 8 bytes instead of 8 x 16), which makes a hop image 1088 complex = the stage-1 image's size; the complex64 DMA
// target is a single buffer, its copy issued AFTER the group's first barrier (every wave has consumed the previous copy by
// then: its LDS reads feed the arm MAC that precedes the barrier).
// DIR: the direction a workgroup walks its run of hops: +1 upwards (groups g0, g0 + 1, ...), -1 DOWNWARDS (g1 - 1, g1 - 2, ...:
// the window slides the other way, the new blocks enter at its low end).  Odd runs walk downwards when the launch asks
// for it (PfbFastArgs::alt_dir): run r - 1 (up) and run r (down) then both reach their common border at the END of their
// walks, run r (down) and run r + 1 (up) both start at theirs -- the 8 halo blocks either neighbour needs from the other's
// range are loaded by both at about the same time, so the second load is an L2 hit (same XCD: chunked mapping).  Walking
// every run upwards the two loads are a whole run apart in time (all resident workgroups are in phase), further than the
// 4 MB L2 reaches: the halo was fetched through the fabric a second time (FETCH_SIZE 1.45 x the input at 20-hop runs).
// Packed complex arithmetic of the tuned kernel: every line is ONE v_pk_* instruction (swaps, broadcasts and sign patterns fold
// into op_sel / constant operands).  The float2 helpers of wh_common.h cost ~130 register moves per group of four hops on top
// of their arithmetic (pairs re-assembled for every +-i rotation and complex product); at three waves per SIMD the kernel's
// compute-side floor is VALU issue, so they count.
typedef float p2f __attribute__((ext_vector_type(2)));
#define P_SW(a) __builtin_shufflevector(a, a, 1, 0)
#define P_XX(a) __builtin_shufflevector(a, a, 0, 0)
#define P_YY(a) __builtin_shufflevector(a, a, 1, 1)
#define P_FMA(a, b, c) __builtin_elementwise_fma(a, b, c)
#define P_PMI(x, d) P_FMA(P_SW(d), (p2f{1.f, -1.f}), x)    // x + (-i) d
#define P_MMI(x, d) P_FMA(P_SW(d), (p2f{-1.f, 1.f}), x)    // x - (-i) d
__device__ __forceinline__ p2f pcmul(p2f a, p2f w) {        // a * w, three instructions
    const p2f t = P_XX(a) * w;
    const p2f s = P_YY(a) * p2f{-1.f, 1.f};
    return P_FMA(s, P_SW(w), t);
}
__device__ __forceinline__ p2f pcmulc(p2f a, float wx, float wy) {   // a * (wx + i wy), compile-time constant: two instructions
    return P_FMA(P_YY(a), (p2f{-wy, wx}), (P_XX(a) * p2f{wx, wy}));
}
__device__ __forceinline__ void pfft4(p2f &a0, p2f &a1, p2f &a2, p2f &a3) {   // forward radix 4, natural order (same bits as fft4)
    const p2f s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13;
    a2 = s02 - s13;
    a1 = P_PMI(d02, d13);
    a3 = P_MMI(d02, d13);
}
__device__ __forceinline__ void pfft16(p2f (&v)[16]) {      // forward 16-point transform in place, natural order in and out
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f, R2 = 0.70710678118654752440f;
#pragma unroll
    for (int na = 0; na < 4; ++na) pfft4(v[na], v[na + 4], v[na + 8], v[na + 12]);
    // v[na + 4 kb] = Y[na][kb] times W16^(na kb)
    v[5] = pcmulc(v[5], C1, -S1);                       // W^1
    v[9] = P_PMI(v[9], v[9]) * p2f{R2, R2};             // W^2 = R2 (1 - i)
    v[13] = pcmulc(v[13], S1, -C1);                     // W^3
    v[6] = P_PMI(v[6], v[6]) * p2f{R2, R2};             // W^2
    v[10] = P_SW(v[10]) * p2f{1.f, -1.f};               // W^4 = -i
    v[14] = P_MMI(v[14], v[14]) * p2f{-R2, -R2};        // W^6 = -R2 (1 + i)
    v[7] = pcmulc(v[7], S1, -C1);                       // W^3
    v[11] = P_MMI(v[11], v[11]) * p2f{-R2, -R2};        // W^6
    v[15] = pcmulc(v[15], -C1, S1);                     // W^9
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) pfft4(v[4 * kb], v[4 * kb + 1], v[4 * kb + 2], v[4 * kb + 3]);
    // v[ka + 4 kb] = X[4 ka + kb]: transpose to natural order (register renaming)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) {
            const p2f tmp = v[a + 4 * b];
            v[a + 4 * b] = v[b + 4 * a];
            v[b + 4 * a] = tmp;
        }
}

template <int FMT, bool GLDS, bool STATS, bool W3>
struct Pfb1024Lds {
    static constexpr int LDS_HOP = W3 ? 1088 : 1152;
    static constexpr int PRE_BYTES = GLDS ? GH * FHOP * (FMT == 1 ? 4 : 8) : 0;   // GH half-blocks of samples
    static constexpr bool PRE_SINGLE = W3 && FMT == 0;
    static constexpr int WORDS = GH * LDS_HOP + 256 + (PRE_SINGLE ? 1 : 2) * PRE_BYTES / 8;   // float2 units
};

template <int FMT, bool GLDS, bool STATS, bool W3, int DIR>
__device__ __forceinline__ void pfb1024_body(const PfbFastArgs &a, float2 *lds, long long g0, long long g1) {
    constexpr bool ST8 = W3;   // 8-byte stores without the lane exchange (see stage 3)
    constexpr int LDS2_ROW = W3 ? 17 : 18;
    constexpr int LDS_HOP = Pfb1024Lds<FMT, GLDS, STATS, W3>::LDS_HOP;
    constexpr int PRE_BYTES = Pfb1024Lds<FMT, GLDS, STATS, W3>::PRE_BYTES;
    constexpr bool PRE_SINGLE = Pfb1024Lds<FMT, GLDS, STATS, W3>::PRE_SINGLE;
    // the DMA target is double-buffered: the copy for group g+2 is issued by whichever wave finishes group g+1's arm MAC
    // first, and lands while slower waves may still be reading group g+1's samples -- it must not share their buffer.
    // With two buffers a buffer is rewritten only after a workgroup barrier that follows its last reads.
    float2 *tw256 = lds + GH * LDS_HOP;
    unsigned char *pre0 = reinterpret_cast<unsigned char *>(lds + GH * LDS_HOP + 256);
    constexpr int NEW0 = DIR > 0 ? 9 : 0;   // window slots the next group's new blocks enter

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;

    // Set-up: EVERY load of the prologue is issued before the first wait (a run is as short as 12 hops, so a workgroup's
    // set-up latency is a visible share of its life: twiddles, window and taps used to be three round trips one after another --
    // the table's barrier stood between the tap loads and the window fill).
    // stage-2 twiddles W256^(n3*k2) are read from a 2 KB LDS table laid out [k2][n3], so a wave's
    // read for one k2 is 16 consecutive entries (conflict-free, broadcast over k1, immediate offsets).
    // (Keeping the 15 per-lane values in registers pushes the kernel to 256 VGPRs: 18 % slower.)
    const float2 tw256_t = a.tw1024[(4 * (t & 15) * (t >> 4)) & 1023];
    // stage-1 twiddles W1024^(t*k1), k1 = 1..3
    float2 tw1 = a.tw1024[t], tw2 = a.tw1024[(2 * t) & 1023], tw3 = a.tw1024[(3 * t) & 1023];
    long long h = a.first_hop + (DIR > 0 ? g0 : g1 - 1) * GH;  // first hop of the first group of the walk
    // windows: wA[i] = x[(h-8+i)*512 + t], wB[i] = x[(h-8+i)*512 + t + 256], i = 0..8 carried,
    // i = 9..12 filled per group
    // window slots 0..8 carry c_{h-8..h}; slots 9..12 double as the prefetch buffer of the
    // group's new samples (loaded one group ahead, in flight across the FFT phase)
    float2 wA[9 + GH], wB[9 + GH];
    const long long xp = (h - 8) * FHOP + t;
#pragma unroll
    for (int i = 0; i < 9 + GH; ++i) {
        wA[i] = ld_iq_raw<FMT>(a.x, xp + i * FHOP);
        wB[i] = ld_iq_raw<FMT>(a.x, xp + i * FHOP + 256);
    }
    // taps: arms[k][j] for k = t, t+256, t+512, t+768, kept as 18 register PAIRS (tap e = q * 9 + j is half e & 1 of pair
    // e / 2): the arm MAC below is packed -- (re, im) * tap as one v_pk_fma_f32 with the tap broadcast from its half of the
    // pair through op_sel -- half the instructions of the scalar form
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f tpr[(4 * FT) / 2];
#pragma unroll
    for (int e = 0; e < 4 * FT; e += 2) {
        const int q0 = e / FT, j0 = e - q0 * FT, q1 = (e + 1) / FT, j1 = (e + 1) - q1 * FT;
        constexpr float TS = FMT == 1 ? 1.0f / 32768.0f : 1.0f;   // int16 input: see ld_iq_raw
        tpr[e / 2] = v2f{a.arms[(t + 256 * q0) * FT + j0] * TS, a.arms[(t + 256 * q1) * FT + j1] * TS};
    }
    tw256[t] = tw256_t;
    __syncthreads();
    const p2f tw1p = {tw1.x, tw1.y}, tw2p = {tw2.x, tw2.y}, tw3p = {tw3.x, tw3.y};
    const p2f tw1r = {-tw1.y, tw1.x}, tw2r = {-tw2.y, tw2.x}, tw3r = {-tw3.y, tw3.x};

    StAcc sacc[STATS ? 4 : 1];
    if (STATS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) stacc_init(sacc[q]);
    }

    // GH half-blocks from block `blk` on, contiguous in memory, copied linearly by the LDS DMA: 16 bytes per lane, lane-linear
    // in LDS
    constexpr int NJ = GLDS ? PRE_BYTES / 16 / 256 : 0;
    auto dma_copy = [&](unsigned char *dst, long long blk) {
#ifdef WH_DIAG
        if (a.ablate & 2) blk = 9 + (blk & 15);   // diagnostics: every copy reads the same few (cache-resident) blocks
#endif
        const unsigned char *src = reinterpret_cast<const unsigned char *>(a.x) + (size_t)blk * FHOP * (FMT == 1 ? 4 : 8);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned char *gp = src + (size_t)(j * 256 + t) * 16;
            unsigned base = (unsigned)(uintptr_t)dst + (unsigned)(j * 256 + wave * 64) * 16;
            base = __builtin_amdgcn_readfirstlane(base);
            unsigned save;
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %1\n\t"
                         "global_load_lds_dwordx4 %2, off\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(save) : "s"(base), "v"(gp) : "memory");
        }
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the window fill is complete before the loop, so that no wait for it is
                                          // placed inside the loop (the static wait would drain the stores in every iteration)
    const long long ng = g1 - g0;
    for (long long it = 0; it < ng; ++it, h += DIR * GH) {
        const bool more = it + 1 < ng;
        unsigned char *pre = pre0 + ((!PRE_SINGLE && (it & 1)) ? PRE_BYTES : 0);   // buffer of the copy issued in this iteration
        // ---- arm MAC + radix-4 stage for GH hops --------------------------------------------
        // (the pairs stay PAIRS: seen through, the compiler keeps a broadcast {t, t} copy of every tap -- 72 registers)
#pragma unroll
        for (int e = 0; e < (4 * FT) / 2; ++e) asm volatile("" : "+v"(tpr[e]));
#define WH_TAP(q, j) ((((q) * FT + (j)) & 1) ? __builtin_shufflevector(tpr[((q) * FT + (j)) >> 1], tpr[((q) * FT + (j)) >> 1], 1, 1) \
                                             : __builtin_shufflevector(tpr[((q) * FT + (j)) >> 1], tpr[((q) * FT + (j)) >> 1], 0, 0))
#define WH_V2(f) (v2f{(f).x, (f).y})
#pragma unroll
        for (int i = 0; i < GH; ++i) {
            // hop h+i: y[k0] uses c_{h+i-j} = w[i+8-j]; y[k0+512] uses c_{h+i-j+1} = w[i+9-j]
            v2f p0 = {0.f, 0.f}, p1 = p0, p2 = p0, p3 = p0;
#pragma unroll
            for (int j = 0; j < FT; ++j) {
                p0 = __builtin_elementwise_fma(WH_V2(wA[i + 8 - j]), WH_TAP(0, j), p0);   // k = t
                p1 = __builtin_elementwise_fma(WH_V2(wB[i + 8 - j]), WH_TAP(1, j), p1);   // k = t+256
                p2 = __builtin_elementwise_fma(WH_V2(wA[i + 9 - j]), WH_TAP(2, j), p2);   // k = t+512
                p3 = __builtin_elementwise_fma(WH_V2(wB[i + 9 - j]), WH_TAP(3, j), p3);   // k = t+768
            }
            pfft4(p0, p1, p2, p3);  // A[k1], n1 = k/256
            p2f *L = reinterpret_cast<p2f *>(lds + i * LDS_HOP);
            L[t] = p0;
            // a * w as two instructions: the rotated copy (-wy, wx) of each stage-1 twiddle is kept beside it (6 registers)
            L[LDS1_K1 + t] = P_FMA(P_YY(p1), tw1r, (P_XX(p1) * tw1p));
            L[2 * LDS1_K1 + t] = P_FMA(P_YY(p2), tw2r, (P_XX(p2) * tw2p));
            L[3 * LDS1_K1 + t] = P_FMA(P_YY(p3), tw3r, (P_XX(p3) * tw3p));
        }
        // slide the windows, then issue the next group's loads into the freed slots (the tail walking up, the head walking down)
        if (DIR > 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                wA[i] = wA[i + GH];
                wB[i] = wB[i + GH];
            }
        } else {
#pragma unroll
            for (int i = 8; i >= 0; --i) {
                wA[i + GH] = wA[i];
                wB[i + GH] = wB[i];
            }
        }
        const long long nblk = DIR > 0 ? h + GH + 1 : h - GH - 8;   // first of the next group's GH new half-blocks
        // (a copy TWO groups ahead -- a whole iteration of lead, and the counted wait below then leaves the previous
        // iteration's stores in flight as well -- measured no faster on int16 input, 1.104 / 1.113 against 1.095 / 1.113 ms at
        // two run lengths in one process: neither the copy's latency nor the store acknowledgements are what the loop waits
        // for; complex64 input has no room for the second buffer at three workgroups per CU)
        auto issue_dma = [&]() {
            if (more) dma_copy(pre, nblk);
        };
        if (GLDS) {
            if (!PRE_SINGLE) issue_dma();
        } else {
            // register prefetch, UNCONDITIONAL (the last group of a run loads clamped blocks nobody uses): with the loads
            // under `if (g + 1 < g1)` the two paths into the next iteration carry different numbers of operations in
            // flight, and hipcc's merged wait for the prefetched registers ends in vmcnt(0) -- every group then drained its
            // predecessor's output stores before its arm MAC; with one path the waits are counted (the 8 stores stay in
            // flight)
#pragma unroll
            for (int i = 0; i < GH; ++i) {
                long long blk = nblk + i;
                blk = blk > a.max_block ? a.max_block : (DIR < 0 && blk < 0 ? 0 : blk);
                wA[NEW0 + i] = ld_iq_raw<FMT>(a.x, blk * FHOP + t);
                wB[NEW0 + i] = ld_iq_raw<FMT>(a.x, blk * FHOP + t + 256);
            }
        }
        __syncthreads();
        if (GLDS && PRE_SINGLE) issue_dma();
        // ---- stage 2 (radix-16 over n2) : wave = hop, lane = (k1, n3) ------------------------
        {
            p2f *L = reinterpret_cast<p2f *>(lds + wave * LDS_HOP);
            const int k1 = lane >> 4, n3 = lane & 15;
            p2f v[16];
            {   // (forcing 16 ds_read_b64 through inline asm instead of hipcc's 8 ds_read2_b64 measured no gain)
                const p2f *src = L + k1 * LDS1_K1 + n3;
#pragma unroll
                for (int n2 = 0; n2 < 16; ++n2) v[n2] = src[n2 * 16];
            }
            pfft16(v);
            // twiddle W256^(n3*k2) and store to image 2 (same wave only: no barrier needed,
            // every lane's reads above were issued before these writes)
            __builtin_amdgcn_wave_barrier();
            p2f *dst = L + k1 * LDS2_ROW + n3;
            const p2f *twp = reinterpret_cast<const p2f *>(tw256);
            dst[0] = v[0];
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2) dst[k2 * 4 * LDS2_ROW] = pcmul(v[k2], twp[k2 * 16 + n3]);
        }
        // ---- stage 3 (radix-16 over n3) : lane = k1 + 4*k2, row = lane ------------------------
        __builtin_amdgcn_wave_barrier();
        {
            float2 *L = lds + wave * LDS_HOP;
            p2f v[16];
            // 8 x ds_read_b128 by inline asm: written as float4 loads, hipcc scalarises them into
            // 16 ds_read2_b32 (real/imag de-interleaved), which at the 144-byte row stride is a 4-way
            // bank conflict (measured: SQ_LDS_BANK_CONFLICT > SQ_ACTIVE_INST_LDS).  b128 reads of the
            // padded rows are conflict-free.  The asm loads are waited for explicitly (lgkmcnt).
            if (W3) {
                // rows 17 complex apart: 8-byte reads, conflict-free (34 l mod 64 distinct over 32 lanes)
                const p2f *src = reinterpret_cast<const p2f *>(L) + lane * LDS2_ROW;
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = src[j];
            } else {
                typedef float f4 __attribute__((ext_vector_type(4)));
                const unsigned addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(L + lane * LDS2_ROW));
                f4 r0, r1, r2, r3, r4, r5, r6, r7;
                asm volatile("ds_read_b128 %0, %8\n\t"
                             "ds_read_b128 %1, %8 offset:16\n\t"
                             "ds_read_b128 %2, %8 offset:32\n\t"
                             "ds_read_b128 %3, %8 offset:48\n\t"
                             "ds_read_b128 %4, %8 offset:64\n\t"
                             "ds_read_b128 %5, %8 offset:80\n\t"
                             "ds_read_b128 %6, %8 offset:96\n\t"
                             "ds_read_b128 %7, %8 offset:112\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                             : "v"(addr)
                             : "memory");
                __builtin_amdgcn_sched_barrier(0);
                v[0] = p2f{r0.x, r0.y};   v[1] = p2f{r0.z, r0.w};
                v[2] = p2f{r1.x, r1.y};   v[3] = p2f{r1.z, r1.w};
                v[4] = p2f{r2.x, r2.y};   v[5] = p2f{r2.z, r2.w};
                v[6] = p2f{r3.x, r3.y};   v[7] = p2f{r3.z, r3.w};
                v[8] = p2f{r4.x, r4.y};   v[9] = p2f{r4.z, r4.w};
                v[10] = p2f{r5.x, r5.y};  v[11] = p2f{r5.z, r5.w};
                v[12] = p2f{r6.x, r6.y};  v[13] = p2f{r6.z, r6.w};
                v[14] = p2f{r7.x, r7.y};  v[15] = p2f{r7.z, r7.w};
            }
            pfft16(v);
            float2 vf[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) vf[j] = make_float2(v[j].x, v[j].y);
            if (STATS) {
                // lane l holds X[l + 64 j] of hop h + wave: their powers go to the wave's own (finished) image, [channel]
                float *pwv = reinterpret_cast<float *>(L);
#pragma unroll
                for (int j = 0; j < 16; ++j) pwv[lane + 64 * j] = stat_power(vf[j].x, vf[j].y);
            } else if (ST8) {
                // 8-byte stores, no lane exchange: 16 store instructions of 512 contiguous bytes per wave.  At three
                // workgroups per CU the exchange (48 selects + 16 ds_bpermute per group; as 32 v_cndmask_b32_dpp it measured
                // the same) no longer pays for the wider stores: 1.2075 against 1.2175 ms (exchange) / 1.2213 (DPP form)
                float2 *o = a.out + (h + wave) * FM + lane;
#pragma unroll
                for (int j = 0; j < 16; ++j)
#ifdef WH_DIAG
                    if (!(a.ablate & 1) || vf[j].x == 1.2345e30f)
#endif
                    o[64 * j] = vf[j];
            } else {
                // 16-byte stores (+2.3 % over 8-byte ones): lanes 2m / 2m+1 swap half of their outputs so that
                // the even lane owns (X[2m + 64 j], X[2m+1 + 64 j]) for j < 8 and the odd lane the pair for j >= 8
                const bool even = (lane & 1) == 0;
                float4 *o4 = reinterpret_cast<float4 *>(a.out + (h + wave) * FM + (lane & ~1) + (even ? 0 : 512));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float2 snd = even ? vf[j + 8] : vf[j];
                    float2 rcv;
                    rcv.x = __shfl_xor(snd.x, 1);
                    rcv.y = __shfl_xor(snd.y, 1);
                    float2 lo = even ? vf[j] : rcv;
                    float2 hi = even ? rcv : vf[j + 8];
#ifdef WH_DIAG
                    if (!(a.ablate & 1) || lo.x == 1.2345e30f)
#endif
                    o4[32 * j] = make_float4(lo.x, lo.y, hi.x, hi.y);
                }
            }
        }
        if (STATS) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int w = 0; w < GH; ++w)      // the group's hops in order
                    stacc_add(sacc[q], reinterpret_cast<const float *>(lds + w * LDS_HOP)[t + 256 * q]);
            if (((it + 1) & 3) == 0) {            // open float32 blocks are folded after 16 hops
#pragma unroll
                for (int q = 0; q < 4; ++q) stacc_fold(sacc[q]);
            }
        }
        if (GLDS) {
            // the prefetch is older than this group's 8 stores: retire it, keep them in flight.  vmcnt counts loads,
            // stores and LDS-DMA together, in issue order (MI355X_MICROARCH.md, "s_waitcnt vmcnt(N) waits until all
            // but the wave's N youngest vector-memory operations are done"); tools/ubench/vmcnt_order.hip checks it
#ifdef WH_DIAG
            if (a.ablate) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else
#endif
            if (STATS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no stores behind the prefetch in this mode
            else if (ST8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        __syncthreads();
        if (GLDS && more) {
#pragma unroll
            for (int i = 0; i < GH; ++i) {
                if (FMT == 1) {
                    const short2 *p16 = reinterpret_cast<const short2 *>(pre);
                    const short2 va = p16[i * FHOP + t], vb = p16[i * FHOP + t + 256];
                    wA[NEW0 + i] = make_float2((float)va.x, (float)va.y);   // (unscaled: the taps carry the 1 / 32768)
                    wB[NEW0 + i] = make_float2((float)vb.x, (float)vb.y);
                } else {
                    const float2 *p32 = reinterpret_cast<const float2 *>(pre);
                    wA[NEW0 + i] = p32[i * FHOP + t];
                    wB[NEW0 + i] = p32[i * FHOP + t + 256];
                }
            }
        }
    }
    if (STATS) {
        double *row = a.stats_ws + (size_t)blockIdx.x * 4 * FM;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            stacc_fold(sacc[q]);
            const int c = t + 256 * q;
            row[c] = sacc[q].s;
            row[FM + c] = sacc[q].s2;
            row[2 * FM + c] = (double)sacc[q].mn;
            row[3 * FM + c] = (double)sacc[q].mx;
        }
    }
}

template <int FMT, bool GLDS = false, bool STATS = false, bool W3 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W3 ? 3 : 2))) void pfb1024_kernel(PfbFastArgs a) {
    __shared__ __attribute__((aligned(16))) float2 lds[Pfb1024Lds<FMT, GLDS, STATS, W3>::WORDS];
    // XCD-aware run mapping: blocks b and b+8 share an XCD (round-robin dispatch); give each
    // XCD a contiguous range of runs so that the 9-block halo of neighbouring runs hits its L2.
    // map_chunk C > 0: chunks of C consecutive runs stay on one XCD, the chunks go round the XCDs -- the halo of C - 1 of C
    // runs is an L2 hit AND the resident workgroups of the whole chip sweep ONE band of addresses (short runs: the band is
    // what the HBM channels see); the grid is padded to a multiple of 8 C, the workgroups past the last run leave at once.
    int b = blockIdx.x;
    int nwg = a.n_wg;
    int per = nwg >> 3;
    int run;
    if (a.map_chunk > 0) {
        const int i = b >> 3, ch = i / a.map_chunk;
        run = (ch * 8 + (b & 7)) * a.map_chunk + (i - ch * a.map_chunk);
    } else if (a.map_chunk == -2) {
        run = (nwg & 7) == 0 ? (b & 7) * per + (b >> 3) : b;
    } else {
        run = b;
    }

    long long g0 = (long long)run * a.groups_per_wg;
    long long g1 = g0 + a.groups_per_wg;
    if (g1 > a.n_groups) g1 = a.n_groups;
    if (g0 >= g1) {
        if (STATS) {   // (never the case with the launcher's grid; an empty row all the same)
            double *row = a.stats_ws + (size_t)blockIdx.x * 4 * FM;
            for (int c = threadIdx.x; c < FM; c += 256) { row[c] = 0.0; row[FM + c] = 0.0; row[2 * FM + c] = INFINITY; row[3 * FM + c] = 0.0; }
        }
        return;
    }

    if (W3 && !STATS && a.alt_dir && (run & 1)) pfb1024_body<FMT, GLDS, STATS, W3, -1>(a, lds, g0, g1);
    else pfb1024_body<FMT, GLDS, STATS, W3, 1>(a, lds, g0, g1);
}

// ------------------------------------------------------------------------------------------
// generic path: one workgroup per hop
// ------------------------------------------------------------------------------------------
struct PfbGenArgs {
    const void *x;
    int fmt;
    const float2 *hist;  // [M][T] carried history (column j = block_{-1-j})
    float2 *out;
    const float *arms;   // float32 [M][T]
    const float2 *tw;    // exp(-2 pi i m / M), m in [0, M)
    int M, T, log2M;     // log2M = 0 when M is not a power of two
    int n_radix;         // > 0: M = product of radix[0..n_radix), each in {2, 3, 4, 5}: mixed-radix Stockham
    int radix[16];
    long long hop0;      // first hop of range A (blocks [0, n_hops))
    long long n_hops;
    long long hop0b;     // first hop of range B (blocks [n_hops, n_hops + n_hops_b)); lets the head
    long long n_hops_b;  // (carried history) and the ragged tail of the fast path share one launch
    long long row_b;     // output row of range B's first hop (statistics-only mode: a compact scratch block); -1 = its hop index
};

__device__ __forceinline__ float2 gen_block(const PfbGenArgs &a, long long g, int k) {
    if (g >= 0) return ld_iq_rt(a.x, a.fmt, g * (a.M / 2) + k);
    int col = (int)(-g - 1);
    if (col >= a.T) return make_float2(0.f, 0.f);
    return a.hist[(size_t)k * a.T + col];
}

__global__ __launch_bounds__(256) void pfb_generic_kernel(PfbGenArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];  // M complex (+M for pow2 ping-pong)
    const long long h = (long long)blockIdx.x < a.n_hops ? a.hop0 + blockIdx.x
                                                          : a.hop0b + ((long long)blockIdx.x - a.n_hops);
    const int M = a.M, T = a.T;
    for (int k = threadIdx.x; k < M; k += blockDim.x) {
        float re = 0.f, im = 0.f;
        for (int j = 0; j < T; ++j) {
            float2 c = gen_block(a, h - j, k);
            float w = a.arms[(size_t)k * T + j];
            re = fmaf(c.x, w, re);
            im = fmaf(c.y, w, im);
        }
        sm[k] = make_float2(re, im);
    }
    __syncthreads();
    const bool in_b = (long long)blockIdx.x >= a.n_hops;
    float2 *o = a.out + (size_t)((in_b && a.row_b >= 0) ? a.row_b + (h - a.hop0b) : h) * M;
    if (a.n_radix > 0) {
        // Stockham autosort, mixed radix (DIF), ping-pong between sm[0:M) and sm[M:2M): a pass of radix r with
        // sub-length n and stride s (n*s = M), m = n/r: for p < m, q < s
        //   y[q + s*(r*p + k)] = (sum_j x[q + s*(p + j*m)] * W_r^(j*k)) * W_n^(p*k),   W_n^(p*k) = tw[(p*k*s) mod M]
        float2 *src = sm, *dst = sm + M;
        int n = M, s = 1;
        for (int st = 0; st < a.n_radix; ++st) {
            const int r = a.radix[st];
            const int m = n / r;
            for (int i = threadIdx.x; i < M / r; i += blockDim.x) {
                const int pp = i / s, q = i - pp * s;
                const float2 *x = src + q + s * pp;
                float2 *y = dst + q + s * r * pp;
                const int sm_ = s * m;
                const int tws = pp * s;   // k * tws < M for k < r: no wrap
                if (r == 2) {
                    float2 c0 = x[0], c1 = x[sm_];
                    y[0] = cadd(c0, c1);
                    y[s] = cmul(csub(c0, c1), a.tw[tws]);
                } else if (r == 4) {
                    float2 v0 = x[0], v1 = x[sm_], v2 = x[2 * sm_], v3 = x[3 * sm_];
                    fft4(v0, v1, v2, v3);
                    y[0] = v0;
                    y[s] = cmul(v1, a.tw[tws]);
                    y[2 * s] = cmul(v2, a.tw[2 * tws]);
                    y[3 * s] = cmul(v3, a.tw[3 * tws]);
                } else if (r == 3) {
                    const float C = -0.5f, S = -0.86602540378443864676f;   // exp(-2 pi i / 3)
                    float2 v0 = x[0], v1 = x[sm_], v2 = x[2 * sm_];
                    float2 t1 = cadd(v1, v2), t2 = csub(v1, v2);
                    float2 u = make_float2(fmaf(C, t1.x, v0.x), fmaf(C, t1.y, v0.y));
                    float2 w = make_float2(-S * t2.y, S * t2.x);            // i * S * t2
                    y[0] = cadd(v0, t1);
                    y[s] = cmul(cadd(u, w), a.tw[tws]);
                    y[2 * s] = cmul(csub(u, w), a.tw[2 * tws]);
                } else {   // r == 5
                    const float C1 = 0.30901699437494742410f, S1 = 0.95105651629515357212f;    // cos, sin 2 pi / 5
                    const float C2 = -0.80901699437494742410f, S2 = 0.58778525229247312917f;   // cos, sin 4 pi / 5
                    float2 v0 = x[0], v1 = x[sm_], v2 = x[2 * sm_], v3 = x[3 * sm_], v4 = x[4 * sm_];
                    float2 a1 = cadd(v1, v4), b1 = csub(v1, v4), a2 = cadd(v2, v3), b2 = csub(v2, v3);
                    float2 r1 = make_float2(v0.x + C1 * a1.x + C2 * a2.x, v0.y + C1 * a1.y + C2 * a2.y);
                    float2 r2 = make_float2(v0.x + C2 * a1.x + C1 * a2.x, v0.y + C2 * a1.y + C1 * a2.y);
                    // -i (S1 b1 + S2 b2), -i (S2 b1 - S1 b2)   (forward transform)
                    float2 i1 = make_float2(S1 * b1.y + S2 * b2.y, -(S1 * b1.x + S2 * b2.x));
                    float2 i2 = make_float2(S2 * b1.y - S1 * b2.y, -(S2 * b1.x - S1 * b2.x));
                    y[0] = make_float2(v0.x + a1.x + a2.x, v0.y + a1.y + a2.y);
                    y[s] = cmul(cadd(r1, i1), a.tw[tws]);
                    y[2 * s] = cmul(cadd(r2, i2), a.tw[2 * tws]);
                    y[3 * s] = cmul(csub(r2, i2), a.tw[3 * tws]);
                    y[4 * s] = cmul(csub(r1, i1), a.tw[4 * tws]);
                }
            }
            __syncthreads();
            float2 *tmp = src; src = dst; dst = tmp;
            n = m; s *= r;
        }
        for (int k = threadIdx.x; k < M; k += blockDim.x) o[k] = src[k];
    } else if (a.log2M > 0) {
        // Stockham autosort radix-2 (DIF), ping-pong between sm[0:M) and sm[M:2M):
        // stage with sub-length n and stride s (n*s = M): for p < n/2, q < s
        //   y[q + s*2p] = a + b ; y[q + s*(2p+1)] = (a - b) * exp(-2 pi i p/n)
        float2 *src = sm, *dst = sm + M;
        const int half = M >> 1;
        int n = M, s = 1;
        for (int st = 0; st < a.log2M; ++st) {
            const int m = n >> 1;
            for (int i = threadIdx.x; i < half; i += blockDim.x) {
                int pp = i / s, q = i - pp * s;
                float2 c0 = src[q + s * pp];
                float2 c1 = src[q + s * (pp + m)];
                float2 w = a.tw[(size_t)pp * s];
                dst[q + s * 2 * pp] = cadd(c0, c1);
                dst[q + s * (2 * pp + 1)] = cmul(csub(c0, c1), w);
            }
            __syncthreads();
            float2 *tmp = src; src = dst; dst = tmp;
            n = m; s <<= 1;
        }
        for (int k = threadIdx.x; k < M; k += blockDim.x) o[k] = src[k];
    } else {
        for (int c = threadIdx.x; c < M; c += blockDim.x) {
            float re = 0.f, im = 0.f;
            int idx = 0;
            for (int k = 0; k < M; ++k) {
                float2 y = sm[k];
                float2 w = a.tw[idx];
                re += y.x * w.x - y.y * w.y;
                im += y.x * w.y + y.y * w.x;
                idx += c;
                if (idx >= M) idx -= M;
            }
            o[c] = make_float2(re, im);
        }
    }
}

// ------------------------------------------------------------------------------------------
// run path: any M with 4 | M, 64 <= M <= 512, T = 9 (the reference's benchmark_dsp.py shape is M = 320)
// ------------------------------------------------------------------------------------------
// One WAVE walks a run of consecutive hops (4 independent waves per workgroup; no workgroup barrier in the loop).
// Lane l owns the "quads" u = l + 64 c (c < CPL, u < M/4), i.e. the columns u, u + M/4, u + M/2, u + 3M/4:
//   * arm MAC as in the M = 1024 kernel: two sliding register windows per quad (columns u and u + M/4 of the
//     half-block serve all four columns because block_g[k + M/2] == block_{g+1}[k]); every input sample is
//     loaded once per wave; the 36 taps of a quad are staged in LDS ([tap][quad] so a wave's read is contiguous);
//   * the first Stockham pass (radix 4, stride 1) needs exactly the four values the lane holds -> in registers;
//   * the remaining passes run in place in the wave's own LDS image: every lane reads the inputs of all its
//     butterflies before any lane writes (one instruction stream, in-order LDS), so no ping-pong image and no
//     s_barrier; the last pass stores straight to HBM (stride M/r: whole 512-byte rows per store).
// Same pass plan and arithmetic as pfb_generic_kernel, which still does the first T-1 hops (carried history).
constexpr int RT = 9;
struct PfbRunArgs {
    const void *x;
    float2 *out;
    const float *arms;     // float32 [M][9]
    const float2 *tw;      // exp(-2 pi i m / M)
    int M, n_radix;        // radix[0] == 4 (register pass)
    int radix[12];
    int stride[12];        // s of pass st (product of the earlier radices)
    float inv_stride[12];
    long long first_hop, end_hop;
    long long max_block;   // last half-block index fully inside the input
    int hops_per_wave;
};

template <int CPL>
__device__ __forceinline__ void run_fft_passes(float2 *img, const float2 *twl, const PfbRunArgs &a, float2 *o,
                                               int lane) {
    const int M = a.M;
    for (int st = 1; st < a.n_radix; ++st) {
        const int r = a.radix[st], s = a.stride[st];
        const float inv_s = a.inv_stride[st];
        const int cnt = M / r;               // butterflies of this pass; input j of butterfly i is img[i + j*cnt]
        const bool last = st == a.n_radix - 1;
        float2 *d = last ? o : img;
        // ps = s * floor(i / s): output base = i + (r-1)*ps, twiddle W_M^(k*ps)
#define WH_PS(i) ((int)(((float)(i) + 0.5f) * inv_s) * s)
        if (r == 4) {
            float2 v[CPL][4];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int i = lane + 64 * c, ii = i < cnt ? i : 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[c][j] = img[ii + j * cnt];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int i = lane + 64 * c;
                if (i < cnt) {
                    fft4(v[c][0], v[c][1], v[c][2], v[c][3]);
                    if (last) {
                        d[i] = v[c][0]; d[i + s] = v[c][1]; d[i + 2 * s] = v[c][2]; d[i + 3 * s] = v[c][3];
                    } else {
                        const int ps = WH_PS(i);
                        float2 *y = d + i + 3 * ps;
                        y[0] = v[c][0];
                        y[s] = cmul(v[c][1], twl[ps]);
                        y[2 * s] = cmul(v[c][2], twl[2 * ps]);
                        y[3 * s] = cmul(v[c][3], twl[3 * ps]);
                    }
                }
            }
        } else if (r == 2) {
            constexpr int R = 2 * CPL;
            float2 v[R][2];
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const int i = lane + 64 * c, ii = i < cnt ? i : 0;
                v[c][0] = img[ii];
                v[c][1] = img[ii + cnt];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const int i = lane + 64 * c;
                if (i < cnt) {
                    const int ps = last ? 0 : WH_PS(i);
                    float2 *y = d + i + ps;
                    y[0] = cadd(v[c][0], v[c][1]);
                    y[s] = cmul(csub(v[c][0], v[c][1]), twl[ps]);
                }
            }
        } else if (r == 3) {
            constexpr int R = (4 * CPL + 2) / 3;
            const float C = -0.5f, S = -0.86602540378443864676f;   // exp(-2 pi i / 3)
            float2 v[R][3];
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const int i = lane + 64 * c, ii = i < cnt ? i : 0;
#pragma unroll
                for (int j = 0; j < 3; ++j) v[c][j] = img[ii + j * cnt];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const int i = lane + 64 * c;
                if (i < cnt) {
                    const int ps = last ? 0 : WH_PS(i);
                    float2 v0 = v[c][0], v1 = v[c][1], v2 = v[c][2];
                    float2 t1 = cadd(v1, v2), t2 = csub(v1, v2);
                    float2 u = make_float2(fmaf(C, t1.x, v0.x), fmaf(C, t1.y, v0.y));
                    float2 w = make_float2(-S * t2.y, S * t2.x);            // i * S * t2
                    float2 *y = d + i + 2 * ps;
                    y[0] = cadd(v0, t1);
                    y[s] = cmul(cadd(u, w), twl[ps]);
                    y[2 * s] = cmul(csub(u, w), twl[2 * ps]);
                }
            }
        } else {   // r == 5
            constexpr int R = (4 * CPL + 4) / 5;
            const float C1 = 0.30901699437494742410f, S1 = 0.95105651629515357212f;    // cos, sin 2 pi / 5
            const float C2 = -0.80901699437494742410f, S2 = 0.58778525229247312917f;   // cos, sin 4 pi / 5
            float2 v[R][5];
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const int i = lane + 64 * c, ii = i < cnt ? i : 0;
#pragma unroll
                for (int j = 0; j < 5; ++j) v[c][j] = img[ii + j * cnt];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const int i = lane + 64 * c;
                if (i < cnt) {
                    const int ps = last ? 0 : WH_PS(i);
                    float2 v0 = v[c][0], v1 = v[c][1], v2 = v[c][2], v3 = v[c][3], v4 = v[c][4];
                    float2 a1 = cadd(v1, v4), b1 = csub(v1, v4), a2 = cadd(v2, v3), b2 = csub(v2, v3);
                    float2 r1 = make_float2(v0.x + C1 * a1.x + C2 * a2.x, v0.y + C1 * a1.y + C2 * a2.y);
                    float2 r2 = make_float2(v0.x + C2 * a1.x + C1 * a2.x, v0.y + C2 * a1.y + C1 * a2.y);
                    float2 i1 = make_float2(S1 * b1.y + S2 * b2.y, -(S1 * b1.x + S2 * b2.x));
                    float2 i2 = make_float2(S2 * b1.y - S1 * b2.y, -(S2 * b1.x - S1 * b2.x));
                    float2 *y = d + i + 4 * ps;
                    y[0] = make_float2(v0.x + a1.x + a2.x, v0.y + a1.y + a2.y);
                    y[s] = cmul(cadd(r1, i1), twl[ps]);
                    y[2 * s] = cmul(cadd(r2, i2), twl[2 * ps]);
                    y[3 * s] = cmul(csub(r2, i2), twl[3 * ps]);
                    y[4 * s] = cmul(csub(r1, i1), twl[4 * ps]);
                }
            }
        }
#undef WH_PS
        __builtin_amdgcn_wave_barrier();
    }
}

template <int FMT, int CPL, int GH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CPL == 1 ? 4 : 3, CPL == 1 ? 4 : 3))) void pfb_run_kernel(PfbRunArgs a) {
    // three separate LDS objects, so that the compiler knows a twiddle / tap read cannot alias an image write (as
    // slices of one array every twiddle read of a pass waited for the preceding image write: three exposed LDS round
    // trips per butterfly)
    __shared__ float2 twl[512];                                 // [M]
    __shared__ float tapl[36 * CPL * 64];                       // [36][CPL*64]
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    const int M = a.M, Q = M >> 2, HB = M >> 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2 *img = sm + wave * GH * M;                           // GH images of M per wave

    for (int i = threadIdx.x; i < M; i += 256) twl[i] = a.tw[i];
    for (int i = threadIdx.x; i < 36 * CPL * 64; i += 256) {
        const int e = i / (CPL * 64), u = i - e * (CPL * 64);   // e = qq*9 + j
        const int qq = e / RT, j = e - qq * RT;
        tapl[i] = u < Q ? a.arms[(size_t)(u + qq * Q) * RT + j] : 0.f;
    }
    __syncthreads();

    const long long wid = (long long)blockIdx.x * 4 + wave;
    long long h = a.first_hop + wid * a.hops_per_wave;
    long long h1 = h + a.hops_per_wave;
    if (h1 > a.end_hop) h1 = a.end_hop;
    if (h >= h1) return;

    int ue[CPL];            // clamped quad index (idle lanes of a partial round recompute quad Q-1, stores masked)
    float2 wA[CPL][RT + GH], wB[CPL][RT + GH];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int u = lane + 64 * c;
        ue[c] = u < Q ? u : Q - 1;
#pragma unroll
        for (int i = 0; i < RT + GH; ++i) {
            long long g = h - 8 + i;
            if (g > a.max_block) g = a.max_block;
            wA[c][i] = ld_iq<FMT>(a.x, g * HB + ue[c]);
            wB[c][i] = ld_iq<FMT>(a.x, g * HB + ue[c] + Q);
            __builtin_amdgcn_sched_barrier(0);   // address, load, next: not 2 x 11 x CPL 64-bit addresses up front
        }
    }

    for (; h < h1; h += GH) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            // tap-outer order: 4 taps live at a time (all 36 would cost a wave of occupancy); the accumulation order
            // per output (j ascending) is unchanged
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f y[GH][4];
#pragma unroll
            for (int i = 0; i < GH; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) y[i][q] = v2f{0.f, 0.f};
            const float *tq = tapl + 64 * c + lane;
            // (re, im) x tap as one packed FMA with the tap broadcast: written with vector types so that the
            // compiler does not pair unrelated scalars (which costs a second, swapped copy of all 36 taps)
#define WH_V2(f) (v2f{(f).x, (f).y})
#pragma unroll
            for (int j = 0; j < RT; ++j) {
                // taps in three batches (the LDS reads may not cross the asm), or a quad's 36 taps plus
                // their even-register copies overflow the occupancy budget (4 waves per SIMD with one quad per lane, 3 with two)
                if (j == 3 || j == 6) asm volatile("" ::: "memory");
                const float t0 = tq[j * (CPL * 64)], t1 = tq[(RT + j) * (CPL * 64)];
                const float t2 = tq[(2 * RT + j) * (CPL * 64)], t3 = tq[(3 * RT + j) * (CPL * 64)];
#pragma unroll
                for (int i = 0; i < GH; ++i) {
                    y[i][0] = __builtin_elementwise_fma(WH_V2(wA[c][i + 8 - j]), v2f{t0, t0}, y[i][0]);
                    y[i][1] = __builtin_elementwise_fma(WH_V2(wB[c][i + 8 - j]), v2f{t1, t1}, y[i][1]);
                    y[i][2] = __builtin_elementwise_fma(WH_V2(wA[c][i + 9 - j]), v2f{t2, t2}, y[i][2]);
                    y[i][3] = __builtin_elementwise_fma(WH_V2(wB[c][i + 9 - j]), v2f{t3, t3}, y[i][3]);
                }
            }
#undef WH_V2
#pragma unroll
            for (int i = 0; i < GH; ++i) {
                float2 y0 = make_float2(y[i][0].x, y[i][0].y), y1 = make_float2(y[i][1].x, y[i][1].y);
                float2 y2 = make_float2(y[i][2].x, y[i][2].y), y3 = make_float2(y[i][3].x, y[i][3].y);
                fft4(y0, y1, y2, y3);
                y1 = cmul(y1, twl[ue[c]]);        // W_M^(u k) from the LDS table (6 registers per quad otherwise)
                y2 = cmul(y2, twl[2 * ue[c]]);
                y3 = cmul(y3, twl[3 * ue[c]]);
                if (lane + 64 * c < Q) {
                    float4 *dst = reinterpret_cast<float4 *>(img + i * M + 4 * ue[c]);
                    dst[0] = make_float4(y0.x, y0.y, y1.x, y1.y);
                    dst[1] = make_float4(y2.x, y2.y, y3.x, y3.y);
                }
            }
            // slide, then prefetch the next group's blocks into the freed tail slots
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                wA[c][i] = wA[c][i + GH];
                wB[c][i] = wB[c][i + GH];
            }
#pragma unroll
            for (int i = 0; i < GH; ++i) {
                long long g = h + GH + 1 + i;
                if (g > a.max_block) g = a.max_block;
                wA[c][RT + i] = ld_iq<FMT>(a.x, g * HB + ue[c]);
                wB[c][RT + i] = ld_iq<FMT>(a.x, g * HB + ue[c] + Q);
            }
            __builtin_amdgcn_sched_barrier(0);   // one quad at a time: interleaving them doubles the live registers
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < GH; ++i) {
            // opaque copy of the lane index: keeps the per-pass LDS addresses from being hoisted out of the hop loop
            // (loop-invariant, so the compiler would otherwise park ~100 of them in registers for the whole run)
            int lane_v = lane;
            asm volatile("" : "+v"(lane_v));
            if (h + i < h1) run_fft_passes<CPL>(img + i * M, twl, a, a.out + (size_t)(h + i) * M, lane_v);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// new_hist[k][j] = block_{H-1-j}[k]
__global__ void pfb_hist_kernel(const void *x, int fmt, const float2 *old_hist, float2 *new_hist, int M, int T,
                                long long H) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * T) return;
    int k = idx / T, j = idx % T;
    long long g = H - 1 - j;
    float2 v;
    if (g >= 0) v = ld_iq_rt(x, fmt, g * (M / 2) + k);
    else {
        int col = (int)(-g - 1);
        v = col < T ? old_hist[(size_t)k * T + col] : make_float2(0.f, 0.f);
    }
    new_hist[idx] = v;
}

__global__ void extract_channel_kernel(const float2 *out, size_t hops, int M, int idx, float2 *col) {
    size_t h = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h < hops) col[h] = out[h * M + idx];
}

// per-channel activity statistics: grid.x = ceil(M/64), grid.y = hop slices; block (64, 4)
__global__ void pfb_stats_kernel(const float2 *out, size_t hops, int M, double *part /*[slices][M][4]*/,
                                 int slices) {
    int c = blockIdx.x * 64 + threadIdx.x;
    int sl = blockIdx.y;
    size_t per = (hops + slices - 1) / slices;
    size_t h0 = (size_t)sl * per, h1 = h0 + per < hops ? h0 + per : hops;
    // the statistic's definition (see StAcc in pfb_mid.hip): p = float32(float32(re^2) + float32(im^2)); sum p and
    // sum p^2 in float32 over blocks of <= 16 of the thread's hops, the block sums added in float64; min / max exact
    double s = 0, s2 = 0, mn = INFINITY, mx = -INFINITY;
    if (c < M) {
        float fs = 0.f, fs2 = 0.f, fmn = INFINITY, fmx = -INFINITY;
        int open = 0;
        for (size_t h = h0 + threadIdx.y; h < h1; h += blockDim.y) {
            const float2 v = out[h * M + c];
            const float p = __fadd_rn(__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y));
            fs = __fadd_rn(fs, p);
            fs2 = fmaf(p, p, fs2);
            fmn = fminf(fmn, p);
            fmx = fmaxf(fmx, p);
            if (++open == 16) { s += (double)fs; s2 += (double)fs2; fs = 0.f; fs2 = 0.f; open = 0; }
        }
        s += (double)fs; s2 += (double)fs2;
        mn = (double)fmn; mx = (double)fmx;
    }
    __shared__ double red[4][4][64];
    red[0][threadIdx.y][threadIdx.x] = s;
    red[1][threadIdx.y][threadIdx.x] = s2;
    red[2][threadIdx.y][threadIdx.x] = mn;
    red[3][threadIdx.y][threadIdx.x] = mx;
    __syncthreads();
    if (threadIdx.y == 0 && c < M) {
        for (int y = 1; y < 4; ++y) {
            s += red[0][y][threadIdx.x];
            s2 += red[1][y][threadIdx.x];
            mn = fmin(mn, red[2][y][threadIdx.x]);
            mx = fmax(mx, red[3][y][threadIdx.x]);
        }
        double *p = part + ((size_t)sl * M + c) * 4;
        p[0] = s; p[1] = s2; p[2] = mn; p[3] = mx;
    }
}

// one wave per channel, one lane per slice (slices <= 64): the partials of a channel arrive in one round of loads
// instead of a 64-step walk of dependent 32 KB strides (21 us -> a few us per scan window at M = 1024)
__global__ __launch_bounds__(256) void pfb_stats_final_kernel(const double *part, int slices, int M, size_t hops,
                                                             double *stats, int accumulate) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), sl = threadIdx.x & 63;
    if (c >= M) return;
    double s = 0, s2 = 0, mn = INFINITY, mx = -INFINITY;
    if (sl < slices) {
        const double *p = part + ((size_t)sl * M + c) * 4;
        s = p[0]; s2 = p[1]; mn = p[2]; mx = p[3];
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        s2 += __shfl_xor(s2, o);
        mn = fmin(mn, __shfl_xor(mn, o));
        mx = fmax(mx, __shfl_xor(mx, o));
    }
    if (sl != 0) return;
    double *o = stats + (size_t)c * 5;
    // accumulate into a row that holds nothing yet (count 0 -- e.g. a zero-initialised buffer) == overwrite: its min / max
    // fields are not observations.  An empty row reads {0, 0, 0, +inf, -inf} (the convention of wh_binstats_update too).
    if (accumulate && o[2] > 0.0) {
        o[0] += s; o[1] += s2; o[2] += (double)hops;
        o[3] = fmin(o[3], mn); o[4] = fmax(o[4], mx);
    } else {
        o[0] = s; o[1] = s2; o[2] = (double)hops; o[3] = mn; o[4] = mx;
    }
}

// merged[c] = {sum_r s0, sum_r s1, sum_r s2, min_r s3, max_r s4} over the gathered rows [R][M][5], ranks in order
__global__ void stats_merge_kernel(const double *g, int R, int M, double *out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    double s0 = 0, s1 = 0, s2 = 0, mn = INFINITY, mx = -INFINITY;   // no observation anywhere: {0, 0, 0, +inf, -inf}
    for (int r = 0; r < R; ++r) {
        const double *p = g + ((size_t)r * M + c) * 5;
        s0 += p[0]; s1 += p[1]; s2 += p[2];
        if (p[2] > 0.0) { mn = fmin(mn, p[3]); mx = fmax(mx, p[4]); }   // a rank without hops contributes no min / max
    }
    double *o = out + (size_t)c * 5;
    o[0] = s0; o[1] = s1; o[2] = s2; o[3] = mn; o[4] = mx;
}

}  // namespace

struct wh_pfb {
    int M, T, log2M;
    float *d_arms = nullptr;    // [M][T] float32
    float2 *d_tw = nullptr;     // [M]
    float2 *d_hist[2] = {nullptr, nullptr};
    int cur = 0;
    double *d_part = nullptr;   // stats partials
    float2 *d_sink = nullptr;   // [M] write-only scratch row of the shaped kernels
    float2 *d_edge = nullptr;   // M = 1024 statistics-only mode: outputs of the <= 8 head and <= 3 tail hops [11][1024]
    double *d_stats_ws = nullptr;   // statistics-only mode: one [4][M] row per workgroup of the launch
    size_t stats_ws_rows = 0;
    int cu_count = 256;
    int gpw_override = 0;       // wh_pfb_tune(WH_PFB_TUNE_HOPS_PER_RUN)
    int map_chunk = 0;          // wh_pfb_tune(WH_PFB_TUNE_RUN_MAP), M = 1024: see PfbFastArgs::map_chunk
    int alt_dir = 1;            // wh_pfb_tune(WH_PFB_TUNE_ALT_DIR), M = 1024: odd runs walk downwards
    int ablate = 0;             // diagnostics build only (WH_PFB_ABLATE)
    int variant = 0;            // wh_pfb_tune(WH_PFB_TUNE_PREFETCH): 1 = register prefetch, 3 = LDS-DMA prefetch, for both formats
    int path = 0;               // wh_pfb_tune(WH_PFB_TUNE_PATH): 0 auto, 1 per-hop kernel only, 2 run kernel, 3 shaped kernel
    bool run_ok = false;        // M, T fit pfb_run_kernel
    bool mid_ok = false;        // a compile-time-shaped instance exists (pfb_mid.hip)
    bool prof = false;          // bracket the fused kernel with events (bench roofline)
    // profiling events: a ring of the last EVR launches' (begin, end) pairs, so that a caller can time EVERY launch of a
    // run and read the durations afterwards -- without a host synchronisation between launches (wh_pfb_kernel_ms_back)
    static constexpr int EVR = 64;
    hipEvent_t ev0[EVR] = {}, ev1[EVR] = {};
    int ev_cur = 0, ev_last = -1, ev_n = 0;
    // side stream of the fast path: the head / tail hops (per-hop kernel) and the history update are independent of the fused
    // kernel and run beside it (fork / join by events on the caller's stream: still enqueue-only, and capturable)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int ev_begin() { ev_cur = (ev_last + 1) % EVR; return ev_cur; }
    void ev_end() { ev_last = ev_cur; if (ev_n < EVR) ++ev_n; }
};

extern "C" int wh_pfb_create(wh_pfb **out, int M, int T, const double *h_arms) {
    if (!out || !h_arms || M < 2 || (M & 1) || T < 1 || T > 64) return set_err(WH_E_ARG, "wh_pfb_create: bad M/T");
    if ((size_t)M * 16 > 160 * 1024) return set_err(WH_E_ARG, "wh_pfb_create: M=%d exceeds the LDS-resident limit", M);
    wh_pfb *p = new wh_pfb();
    std::unique_ptr<wh_pfb, void (*)(wh_pfb *)> guard(p, wh_pfb_destroy);  // frees partial state on early return
    p->M = M; p->T = T;
#ifdef WH_DIAG
    if (const char *e = getenv("WH_PFB_ABLATE")) p->ablate = atoi(e);
#endif
    int l2 = 0;
    while ((1 << l2) < M) ++l2;
    p->log2M = ((1 << l2) == M) ? l2 : 0;
    {
        int rem = M;
        while (rem % 2 == 0) rem /= 2;
        while (rem % 3 == 0) rem /= 3;
        while (rem % 5 == 0) rem /= 5;
        p->run_ok = T == RT && M % 4 == 0 && M >= 64 && M <= 512 && M != FM && rem == 1;
    }
    p->mid_ok = pfb_mid_supported(M, T);
    std::vector<float> arms((size_t)M * T);
    for (size_t i = 0; i < arms.size(); ++i) arms[i] = (float)h_arms[i];
    std::vector<float2> tw(M);
    for (int m = 0; m < M; ++m) {
        double ang = -2.0 * M_PI * (double)m / (double)M;
        tw[m] = make_float2((float)cos(ang), (float)sin(ang));
    }
    hipDeviceProp_t prop;
    int dev = 0;
    WH_HIP(hipGetDevice(&dev));
    WH_HIP(hipGetDeviceProperties(&prop, dev));
    p->cu_count = prop.multiProcessorCount;
    if (M == FM && T == FT) {
        if (hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess) p->side = nullptr;
        if (p->side && (hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) != hipSuccess)) {
            (void)hipStreamDestroy(p->side);
            p->side = nullptr;
        }
    }
    WH_HIP(hipMalloc(&p->d_arms, arms.size() * sizeof(float)));
    WH_HIP(hipMalloc(&p->d_tw, tw.size() * sizeof(float2)));
    WH_HIP(hipMalloc(&p->d_hist[0], (size_t)M * T * sizeof(float2)));
    WH_HIP(hipMalloc(&p->d_hist[1], (size_t)M * T * sizeof(float2)));
    WH_HIP(hipMalloc(&p->d_part, (size_t)64 * M * 4 * sizeof(double)));
    WH_HIP(hipMalloc(&p->d_sink, (size_t)M * sizeof(float2)));
    if (M == FM && T == FT) WH_HIP(hipMalloc(&p->d_edge, (size_t)11 * FM * sizeof(float2)));
    WH_HIP(hipMemcpy(p->d_arms, arms.data(), arms.size() * sizeof(float), hipMemcpyHostToDevice));
    WH_HIP(hipMemcpy(p->d_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    WH_HIP(hipMemset(p->d_hist[0], 0, (size_t)M * T * sizeof(float2)));
    WH_HIP(hipMemset(p->d_hist[1], 0, (size_t)M * T * sizeof(float2)));
    *out = guard.release();
    return WH_OK;
}

extern "C" void wh_pfb_destroy(wh_pfb *p) {
    if (!p) return;
    (void)hipFree(p->d_arms);
    (void)hipFree(p->d_tw);
    (void)hipFree(p->d_hist[0]);
    (void)hipFree(p->d_hist[1]);
    (void)hipFree(p->d_part);
    (void)hipFree(p->d_sink);
    (void)hipFree(p->d_edge);
    (void)hipFree(p->d_stats_ws);
    for (int i = 0; i < wh_pfb::EVR; ++i) {
        if (p->ev0[i]) (void)hipEventDestroy(p->ev0[i]);
        if (p->ev1[i]) (void)hipEventDestroy(p->ev1[i]);
    }
    if (p->side) {
        (void)hipStreamSynchronize(p->side);
        (void)hipStreamDestroy(p->side);
        (void)hipEventDestroy(p->ev_fork);
        (void)hipEventDestroy(p->ev_join);
    }
    delete p;
}

extern "C" int wh_pfb_tune(wh_pfb *p, int key, int value) {
    if (!p) return set_err(WH_E_ARG, "wh_pfb_tune: null handle");
    switch (key) {
    case WH_PFB_TUNE_PATH:
        if (value < 0 || value > 3) return set_err(WH_E_ARG, "wh_pfb_tune: path %d", value);
        if (value == 2 && !p->run_ok) return set_err(WH_E_ARG, "wh_pfb_tune: the run kernel does not take M=%d T=%d", p->M, p->T);
        if (value == 3 && !p->mid_ok) return set_err(WH_E_ARG, "wh_pfb_tune: no shaped kernel for M=%d T=%d", p->M, p->T);
        p->path = value;
        return WH_OK;
    case WH_PFB_TUNE_PREFETCH:
        if (value != 0 && value != 1 && value != 3 && value != 5 && value != 7 ) return set_err(WH_E_ARG, "wh_pfb_tune: prefetch %d", value);
        p->variant = value;
        return WH_OK;
#ifdef WH_DIAG
    case 4:   // diagnostics build only: ablation bits of the shaped kernel
        p->ablate = value;
        return WH_OK;
#endif
    case WH_PFB_TUNE_HOPS_PER_RUN:
        if (value < 0 || value > 4096) return set_err(WH_E_ARG, "wh_pfb_tune: hops per run %d", value);
        p->gpw_override = value;
        return WH_OK;
    case WH_PFB_TUNE_RUN_MAP:
        if (value < -2 || value > 4096) return set_err(WH_E_ARG, "wh_pfb_tune: run map %d", value);
        p->map_chunk = value;
        return WH_OK;
    case WH_PFB_TUNE_ALT_DIR:
        if (value != 0 && value != 1) return set_err(WH_E_ARG, "wh_pfb_tune: alt dir %d", value);
        p->alt_dir = value;
        return WH_OK;
    }
    return set_err(WH_E_ARG, "wh_pfb_tune: unknown key %d", key);
}

extern "C" int wh_pfb_profile(wh_pfb *p, int enable) {
    if (!p) return set_err(WH_E_ARG, "wh_pfb_profile: null handle");
    if (enable && !p->ev0[0]) {
        for (int i = 0; i < wh_pfb::EVR; ++i) {
            WH_HIP(hipEventCreate(&p->ev0[i]));
            WH_HIP(hipEventCreate(&p->ev1[i]));
        }
    }
    p->prof = enable != 0;
    p->ev_n = 0;
    p->ev_last = -1;
    return WH_OK;
}

extern "C" int wh_pfb_kernel_ms_back(wh_pfb *p, int back, float *ms) {
    if (!p || !ms) return set_err(WH_E_ARG, "wh_pfb_kernel_ms: null");
    if (back < 0 || back >= p->ev_n) return set_err(WH_E_ARG, "wh_pfb_kernel_ms: no profiled launch %d back (have %d)", back, p->ev_n);
    const int i = ((p->ev_last - back) % wh_pfb::EVR + wh_pfb::EVR) % wh_pfb::EVR;
    WH_HIP(hipEventSynchronize(p->ev1[i]));
    WH_HIP(hipEventElapsedTime(ms, p->ev0[i], p->ev1[i]));
    return WH_OK;
}

extern "C" int wh_pfb_kernel_ms(wh_pfb *p, float *ms) { return wh_pfb_kernel_ms_back(p, 0, ms); }

extern "C" size_t wh_pfb_hops(const wh_pfb *p, size_t n) {
    if (!p || n < (size_t)p->M) return 0;
    return (n - p->M) / (p->M / 2) + 1;
}

static int launch_generic(wh_pfb *p, const void *d_iq, int fmt, float *d_out, long long hop0, long long n_hops,
                          hipStream_t st, long long hop0b = 0, long long n_hops_b = 0, long long row_b = -1) {
    if (n_hops_b < 0) n_hops_b = 0;
    if (n_hops <= 0 && n_hops_b <= 0) return WH_OK;
    if (n_hops < 0) n_hops = 0;
    PfbGenArgs a;
    a.x = d_iq;
    a.fmt = fmt;
    a.hist = p->d_hist[p->cur];
    a.out = reinterpret_cast<float2 *>(d_out);
    a.arms = p->d_arms;
    a.tw = p->d_tw;
    a.M = p->M; a.T = p->T; a.log2M = p->log2M;
    a.hop0 = hop0; a.n_hops = n_hops;
    a.hop0b = hop0b; a.n_hops_b = n_hops_b; a.row_b = row_b;
    // factor M into radices 4, 2, 3, 5 (mixed-radix Stockham); anything else falls back to the direct DFT
    a.n_radix = 0;
    {
        int rem = p->M, k = 0;
        int rad[16];
        while (rem % 4 == 0 && k < 16) { rad[k++] = 4; rem /= 4; }
        while (rem % 2 == 0 && k < 16) { rad[k++] = 2; rem /= 2; }
        while (rem % 3 == 0 && k < 16) { rad[k++] = 3; rem /= 3; }
        while (rem % 5 == 0 && k < 16) { rad[k++] = 5; rem /= 5; }
        if (rem == 1 && k > 0) {
            a.n_radix = k;
            for (int i = 0; i < 16; ++i) a.radix[i] = i < k ? rad[i] : 1;
        } else {
            for (int i = 0; i < 16; ++i) a.radix[i] = 1;
        }
    }
    size_t smem = (size_t)p->M * sizeof(float2) * ((p->log2M || a.n_radix) ? 2 : 1);
    if (smem > 64 * 1024) {
        WH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pfb_generic_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    // block size: small channel counts are latency-bound per workgroup (a few hundred points, barriers between
    // passes), so fewer threads per hop and more hops in flight per CU win
    int bt = p->M <= 512 ? 64 : 256;
    if (n_hops + n_hops_b <= 2LL * p->cu_count) bt = 256;   // a handful of hops (head of a run-kernel call): latency only
    // grid.x is limited to 2^31-1; chunk very long tails (never happens for the fast path)
    const long long MAXG = 1 << 30;
    if (n_hops_b > 0) {   // two short ranges in one launch
        hipLaunchKernelGGL(pfb_generic_kernel, dim3((unsigned)(n_hops + n_hops_b)), dim3(bt), smem, st, a);
        WH_LAUNCH_CHECK();
        return WH_OK;
    }
    for (long long off = 0; off < n_hops; off += MAXG) {
        long long cnt = n_hops - off < MAXG ? n_hops - off : MAXG;
        a.hop0 = hop0 + off;
        a.n_hops = cnt;
        hipLaunchKernelGGL(pfb_generic_kernel, dim3((unsigned)cnt), dim3(bt), smem, st, a);
        WH_LAUNCH_CHECK();
    }
    return WH_OK;
}

static int pfb_run_fmt(wh_pfb *p, const void *d_iq, int fmt, size_t n, float *d_out, void *stream);

extern "C" int wh_pfb_run(wh_pfb *p, const float *d_iq, size_t n, float *d_out, void *stream) {
    return pfb_run_fmt(p, d_iq, 0, n, d_out, stream);
}

extern "C" int wh_pfb_run_i16(wh_pfb *p, const int16_t *d_iq16, size_t n, float *d_out, void *stream) {
    return pfb_run_fmt(p, d_iq16, 1, n, d_out, stream);
}

static int pfb_run_fmt(wh_pfb *p, const void *d_iq, int fmt, size_t n, float *d_out, void *stream) {
    if (!p) return set_err(WH_E_ARG, "wh_pfb_run: null handle");
    hipStream_t st = as_stream(stream);
    long long H = (long long)wh_pfb_hops(p, n);
    if (H == 0) return WH_OK;
    if (!d_iq || !d_out) return set_err(WH_E_ARG, "wh_pfb_run: null buffer");
    int rc;
    const bool fast = (p->M == FM && p->T == FT) && p->path != 1 && !(p->path == 3 && p->mid_ok);
    if (!fast && p->mid_ok && (p->path == 0 || p->path == 3)) {
        // one launch: head hops, runs and the history update (pfb_mid.hip)
        PfbMidCall c;
        c.x = d_iq; c.fmt = fmt; c.n = n;
        c.hist = p->d_hist[p->cur]; c.new_hist = p->d_hist[p->cur ^ 1];
        c.out = reinterpret_cast<float2 *>(d_out);
        c.sink = p->d_sink; c.sink_elems = (size_t)p->M;
        c.arms = p->d_arms; c.tw = p->d_tw;
        c.H = H; c.cu_count = p->cu_count; c.hops_per_run = p->gpw_override; c.stats_only = 0;
        c.stats_ws = nullptr; c.stats_out = nullptr; c.stats_accumulate = 0;
#ifdef WH_DIAG
        c.stats_only = p->ablate << 8;
#endif
        if (p->prof) WH_HIP(hipEventRecord(p->ev0[p->ev_begin()], st));
        if ((rc = pfb_mid_launch(p->M, p->T, c, st)) != WH_OK) return rc;
        if (p->prof) {
            WH_HIP(hipEventRecord(p->ev1[p->ev_cur], st));
            p->ev_end();
        }
        p->cur ^= 1;
        return WH_OK;
    }
    const bool run = !fast && p->run_ok && p->path != 1;
    long long head = (fast || run) ? (H < 8 ? H : 8) : H;
    // long calls of the fast path: the per-hop kernel (head / tail hops: 11 workgroups, ~16 us of dependent passes) and the
    // history update go to the handle's side stream and run BESIDE the fused kernel instead of before / after it
    const bool fork = fast && p->side && (H - 8) / GH >= 8LL * p->cu_count;
    hipStream_t sg = st;
    if (fork) {
        WH_HIP(hipEventRecord(p->ev_fork, st));
        WH_HIP(hipStreamWaitEvent(p->side, p->ev_fork, 0));
        sg = p->side;
    }
    {
        // head hops (need the carried history) + the <= 3 ragged tail hops of the fast path, one launch
        long long tail0 = 0, tailn = 0;
        if (fast && H > 8) {
            tail0 = 8 + ((H - 8) / GH) * GH;
            tailn = H - tail0;
        }
        if ((rc = launch_generic(p, d_iq, fmt, d_out, 0, head, sg, tail0, tailn)) != WH_OK) return rc;
    }
    const int hist_nxt = p->cur ^ 1;
    if (fork) {
        hipLaunchKernelGGL(pfb_hist_kernel, dim3((p->M * p->T + 255) / 256), dim3(256), 0, sg, d_iq, fmt, p->d_hist[p->cur],
                           p->d_hist[hist_nxt], p->M, p->T, H);
        WH_LAUNCH_CHECK();
        WH_HIP(hipEventRecord(p->ev_join, sg));
    }
    if (run && H > 8) {
        PfbRunArgs a;
        a.x = d_iq;
        a.out = reinterpret_cast<float2 *>(d_out);
        a.arms = p->d_arms;
        a.tw = p->d_tw;
        a.M = p->M;
        int k = 0, rem = p->M / 4;
        a.radix[k++] = 4;
        while (rem % 4 == 0) { a.radix[k++] = 4; rem /= 4; }
        while (rem % 2 == 0) { a.radix[k++] = 2; rem /= 2; }
        while (rem % 3 == 0) { a.radix[k++] = 3; rem /= 3; }
        while (rem % 5 == 0) { a.radix[k++] = 5; rem /= 5; }
        a.n_radix = k;
        for (int i = 0, s = 1; i < 12; ++i) {
            if (i >= k) a.radix[i] = 1;
            a.stride[i] = s;
            a.inv_stride[i] = 1.0f / (float)s;
            if (i < k) s *= a.radix[i];
        }
        a.first_hop = 8;
        a.end_hop = H;
        a.max_block = (long long)(n / (size_t)(p->M / 2)) - 1;
        const int cpl = p->M <= 256 ? 1 : 2;
        const int rgh = cpl == 1 ? 2 : 1;   // hops per loop iteration (the two-quad form has no registers for a second one)
        // runs of up to 64 hops per wave (halo 12 %), but at least ~24 waves per CU when the input allows
        long long nh = H - 8;
        long long hpw = (nh + (long long)p->cu_count * 24 - 1) / ((long long)p->cu_count * 24);
        if (hpw < 8) hpw = 8;
        if (hpw > 64) hpw = 64;
        if (p->gpw_override > 0) hpw = p->gpw_override;
        hpw = (hpw + rgh - 1) / rgh * rgh;
        a.hops_per_wave = (int)hpw;
        const long long waves = (nh + hpw - 1) / hpw;
        const unsigned nwg = (unsigned)((waves + 3) / 4);
        const size_t smem = (size_t)4 * rgh * p->M * sizeof(float2);   // images; twiddles and taps are static LDS
        if (p->prof) WH_HIP(hipEventRecord(p->ev0[p->ev_begin()], st));
        if (cpl == 1) {
            if (fmt == 1) hipLaunchKernelGGL((pfb_run_kernel<1, 1, 2>), dim3(nwg), dim3(256), smem, st, a);
            else hipLaunchKernelGGL((pfb_run_kernel<0, 1, 2>), dim3(nwg), dim3(256), smem, st, a);
        } else {
            if (fmt == 1) hipLaunchKernelGGL((pfb_run_kernel<1, 2, 1>), dim3(nwg), dim3(256), smem, st, a);
            else hipLaunchKernelGGL((pfb_run_kernel<0, 2, 1>), dim3(nwg), dim3(256), smem, st, a);
        }
        WH_LAUNCH_CHECK();
        if (p->prof) {
            WH_HIP(hipEventRecord(p->ev1[p->ev_cur], st));
            p->ev_end();
        }
    }
    if (fast && H > 8) {
        long long n_groups = (H - 8) / GH;
        if (n_groups > 0) {
            PfbFastArgs a;
            a.x = d_iq;
            a.out = reinterpret_cast<float2 *>(d_out);
            a.arms = p->d_arms;
            a.tw1024 = p->d_tw;
            a.first_hop = 8;
            a.n_groups = n_groups;
            // Run length.  Three-workgroups form (default): runs of 20 hops (32 for int16 input; 31-round sweeps: complex64
            // 1.186 / 1.195 / 1.211 / 1.223 / 1.281 ms at 16 / 20 / 24 / 32 / 48 hops, int16 1.021 / 1.020 / 1.010 / 1.037 / 1.067 at
            // 20 / 28 / 32 / 40 / 48), neighbouring runs walked in
            // opposite directions -- the resident workgroups of the chip then sweep one narrow band of addresses, which the
            // HBM channels serve better than 768 scattered streams, and the 9-block halo of a run is an L2 hit (FETCH_SIZE
            // 1.08 x the input at 32 hops; 1.28 x when every run walks upwards).  Two-workgroups forms: runs of
            // 256 hops (halo 3 %).  Either way at least two rounds of resident workgroups when the input allows.
            const bool w3 = p->variant == 0 || p->variant == 5 || p->variant == 7;
            int gpw = w3 ? (fmt == 1 ? 8 : 5) : 64;
            long long nwg = (n_groups + gpw - 1) / gpw;
            while (gpw > 2 && nwg < (long long)p->cu_count * (w3 ? 6 : 8)) {
                gpw = w3 ? gpw - 1 : gpw >> 1;
                nwg = (n_groups + gpw - 1) / gpw;
            }
            if (p->gpw_override > 0) {
                gpw = p->gpw_override;
                nwg = (n_groups + gpw - 1) / gpw;
            }
            a.groups_per_wg = gpw;
            a.n_wg = (int)nwg;
            // automatic mapping: short runs in chunks of 16 per XCD (one band of addresses for the whole chip), long runs as
            // one contiguous range per XCD (what rounds 1-2 measured best at 128-256 hops)
            a.map_chunk = p->map_chunk != 0 ? p->map_chunk : (gpw <= 16 ? 16 : -2);
            a.alt_dir = p->alt_dir;
            if (a.map_chunk > 0) nwg = (nwg + 8LL * a.map_chunk - 1) / (8LL * a.map_chunk) * (8LL * a.map_chunk);
            a.max_block = (long long)(n / (size_t)FHOP) - 1;
#ifdef WH_DIAG
            a.ablate = p->ablate;
#endif
            if (p->prof) WH_HIP(hipEventRecord(p->ev0[p->ev_begin()], st));
            // kernel form (wh_pfb_tune PREFETCH): 0 / 5 = three workgroups per CU with the LDS-DMA prefetch (default, both input
            // formats), 7 = three per CU with register prefetch, 1 / 3 = the two-workgroup forms of rounds 1-2 (register / DMA)
            const bool dma = p->variant == 3;
            if (p->variant == 0 || p->variant == 5 || p->variant == 7) {
                if (p->variant != 7) {
                    if (fmt == 1) hipLaunchKernelGGL((pfb1024_kernel<1, true, false, true>), dim3((unsigned)nwg), dim3(256), 0, st, a);
                    else hipLaunchKernelGGL((pfb1024_kernel<0, true, false, true>), dim3((unsigned)nwg), dim3(256), 0, st, a);
                } else {
                    if (fmt == 1) hipLaunchKernelGGL((pfb1024_kernel<1, false, false, true>), dim3((unsigned)nwg), dim3(256), 0, st, a);
                    else hipLaunchKernelGGL((pfb1024_kernel<0, false, false, true>), dim3((unsigned)nwg), dim3(256), 0, st, a);
                }
            } else if (dma) {
                if (fmt == 1) hipLaunchKernelGGL((pfb1024_kernel<1, true>), dim3((unsigned)nwg), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((pfb1024_kernel<0, true>), dim3((unsigned)nwg), dim3(256), 0, st, a);
            } else if (fmt == 1)
                hipLaunchKernelGGL(pfb1024_kernel<1>, dim3((unsigned)nwg), dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL(pfb1024_kernel<0>, dim3((unsigned)nwg), dim3(256), 0, st, a);
            WH_LAUNCH_CHECK();
            if (p->prof) {
                WH_HIP(hipEventRecord(p->ev1[p->ev_cur], st));
                p->ev_end();
            }
        }
    }
    // carry the history
    int nxt = hist_nxt;
    if (fork) {
        WH_HIP(hipStreamWaitEvent(st, p->ev_join, 0));   // join: everything of this call is ordered before what follows on st
    } else {
        int tot = p->M * p->T;
        hipLaunchKernelGGL(pfb_hist_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, d_iq, fmt, p->d_hist[p->cur],
                           p->d_hist[nxt], p->M, p->T, H);
        WH_LAUNCH_CHECK();
    }
    p->cur = nxt;
    return WH_OK;
}

// diagnostics: the filterbank's HBM traffic shape (8 B read, 16 B written per sample) with no arithmetic at all --
// the in-process yardstick for what the memory system gives a 1 : 2 read : write stream
template <int MODE>
__global__ __launch_bounds__(256) void stream_1r2w_kernel(const float4 *in, float4 *out, size_t n4) {
    // MODE 0: two output streams (out[i], out[n4 + i]); 1: interleaved rows of 1 KiB (row r of the input 4 KiB block
    // goes to rows 2r, 2r+1); 2: mode 0 with non-temporal stores
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i0 = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i0 < n4; i0 += stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + u * 256 < n4 ? in[i0 + u * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * 256 < n4) {
                const size_t i = i0 + u * 256;
                if (MODE == 1) {
                    const size_t o = 2 * (i - threadIdx.x) + threadIdx.x;
                    out[o] = v[u];
                    out[o + 256] = v[u];
                } else if (MODE == 2) {
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    const f4 q = {v[u].x, v[u].y, v[u].z, v[u].w};
                    __builtin_nontemporal_store(q, reinterpret_cast<f4 *>(out + i));
                    __builtin_nontemporal_store(q, reinterpret_cast<f4 *>(out + n4 + i));
                } else {
                    out[i] = v[u];
                    out[n4 + i] = v[u];
                }
            }
    }
}

extern "C" int wh_diag_stream_1r2w(const float *d_in, float *d_out, size_t n, void *stream) {
    if (!d_in || !d_out || (n & 1)) return set_err(WH_E_ARG, "wh_diag_stream_1r2w: null buffer or odd n");
    if (n == 0) return WH_OK;
    const size_t n4 = n / 2;
    size_t blocks = (n4 + 1023) / 1024;
    size_t cap = (size_t)1 << 24;   // one pass per workgroup measured best (5.4 TB/s; 4.9 with a 4096-block grid-stride walk)
    int mode = 0;
#ifdef WH_DIAG
    if (const char *e = getenv("WH_DIAG_BLOCKS")) cap = (size_t)atol(e);
    if (const char *e = getenv("WH_DIAG_MODE")) mode = atoi(e);
#endif
    if (blocks > cap) blocks = cap;
    const float4 *in4 = reinterpret_cast<const float4 *>(d_in);
    float4 *out4 = reinterpret_cast<float4 *>(d_out);
    if (mode == 1) hipLaunchKernelGGL(stream_1r2w_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), in4, out4, n4);
    else if (mode == 2) hipLaunchKernelGGL(stream_1r2w_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), in4, out4, n4);
    else hipLaunchKernelGGL(stream_1r2w_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), in4, out4, n4);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

// the int16 filterbank's traffic shape: n complex64-sized reads of 4 bytes each would be n float2 halves -- here n float
// pairs (8 bytes) are read and written FOUR times (1 : 4 read : write, like 4 B in / 16 B out per sample)
__global__ __launch_bounds__(256) void stream_1r4w_kernel(const float4 *in, float4 *out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i0 = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i0 < n4; i0 += stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + u * 256 < n4 ? in[i0 + u * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * 256 < n4) {
                const size_t i = i0 + u * 256;
                out[i] = v[u];
                out[n4 + i] = v[u];
                out[2 * n4 + i] = v[u];
                out[3 * n4 + i] = v[u];
            }
    }
}

extern "C" int wh_diag_stream_1r4w(const float *d_in, float *d_out, size_t n, void *stream) {
    if (!d_in || !d_out || (n & 1)) return set_err(WH_E_ARG, "wh_diag_stream_1r4w: null buffer or odd n");
    if (n == 0) return WH_OK;
    const size_t n4 = n / 2;
    const size_t blocks = (n4 + 1023) / 1024;   // one pass per workgroup, as wh_diag_stream_1r2w
    hipLaunchKernelGGL(stream_1r4w_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(d_in), reinterpret_cast<float4 *>(d_out), n4);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_pfb_run_stats(wh_pfb *p, const void *d_iq, int input_format, size_t n, double *d_stats, int accumulate,
                                void *stream) {
    if (!p) return set_err(WH_E_ARG, "wh_pfb_run_stats: null handle");
    if (input_format != 0 && input_format != 1) return set_err(WH_E_ARG, "wh_pfb_run_stats: input_format 0 (complex64) / 1 (int16)");
    if (!p->mid_ok) return set_err(WH_E_ARG, "wh_pfb_run_stats: no shaped kernel for M=%d T=%d (run + wh_pfb_channel_stats instead)", p->M, p->T);
    hipStream_t st = as_stream(stream);
    const long long H = (long long)wh_pfb_hops(p, n);
    if (H == 0) return WH_OK;
    if (!d_iq || !d_stats) return set_err(WH_E_ARG, "wh_pfb_run_stats: null buffer");
    if (p->M == FM && p->T == FT && p->path != 3 && p->d_edge) {
        // 1024 channels: the tuned kernel in its statistics-only form (radix-16 stages; 1.13 ms without its stores against
        // the shaped kernel's 1.36 at 2 workgroups per CU), head / tail hops through the per-hop kernel into a scratch block
        int rc;
        const long long n_groups = H > 8 ? (H - 8) / GH : 0;
        const long long head = H < 8 ? H : 8;
        const long long tail0 = 8 + n_groups * GH, tailn = H > 8 ? H - tail0 : 0;
        int acc = accumulate;
        if (n_groups > 0) {
            PfbFastArgs a;
            a.x = d_iq; a.out = nullptr; a.arms = p->d_arms; a.tw1024 = p->d_tw;
            a.first_hop = 8; a.n_groups = n_groups;
            int gpw = 64;
            long long nwg = (n_groups + gpw - 1) / gpw;
            while (gpw > 2 && nwg < (long long)p->cu_count * 8) { gpw >>= 1; nwg = (n_groups + gpw - 1) / gpw; }
            // nothing is stored, so nothing argues for short runs: one resident round (two workgroups per CU) of equal runs
            // when the call is long enough -- a quarter of the workspace rows for the reduction to read (81 -> ~20 us)
            if (nwg > 2LL * p->cu_count) {
                nwg = 2LL * p->cu_count;
                gpw = (int)((n_groups + nwg - 1) / nwg);
                nwg = (n_groups + gpw - 1) / gpw;
            }
            if (p->gpw_override > 0) { gpw = p->gpw_override; nwg = (n_groups + gpw - 1) / gpw; }
            a.groups_per_wg = gpw; a.n_wg = (int)nwg; a.map_chunk = -2; a.alt_dir = 0;
            a.max_block = (long long)(n / (size_t)FHOP) - 1;
#ifdef WH_DIAG
            a.ablate = 0;
#endif
            if ((size_t)nwg > p->stats_ws_rows) {   // rows of [4][M]: grow (synchronises; a steady call size never does)
                WH_HIP(hipStreamSynchronize(st));
                (void)hipFree(p->d_stats_ws);
                p->d_stats_ws = nullptr;
                p->stats_ws_rows = 0;
                WH_HIP(hipMalloc(&p->d_stats_ws, (size_t)nwg * 4 * p->M * sizeof(double)));
                p->stats_ws_rows = (size_t)nwg;
            }
            a.stats_ws = p->d_stats_ws;
            if (p->prof) WH_HIP(hipEventRecord(p->ev0[p->ev_begin()], st));
            if (input_format == 1) hipLaunchKernelGGL((pfb1024_kernel<1, true, true>), dim3((unsigned)nwg), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((pfb1024_kernel<0, false, true>), dim3((unsigned)nwg), dim3(256), 0, st, a);
            WH_LAUNCH_CHECK();
            if (p->prof) {
                WH_HIP(hipEventRecord(p->ev1[p->ev_cur], st));
                p->ev_end();
            }
            if ((rc = pfb_stats_rows_reduce(p->d_stats_ws, (int)nwg, -1, p->M, (double)(n_groups * GH), d_stats, acc, st)) != WH_OK)
                return rc;
            acc = 1;
        }
        if ((rc = launch_generic(p, d_iq, input_format, reinterpret_cast<float *>(p->d_edge), 0, head, st, tail0, tailn, head)) != WH_OK)
            return rc;
        if ((rc = wh_pfb_channel_stats(p, reinterpret_cast<const float *>(p->d_edge), (size_t)(head + tailn), d_stats, acc, stream)) != WH_OK)
            return rc;
        int nxt = p->cur ^ 1;
        int tot = p->M * p->T;
        hipLaunchKernelGGL(pfb_hist_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, d_iq, input_format, p->d_hist[p->cur],
                           p->d_hist[nxt], p->M, p->T, H);
        WH_LAUNCH_CHECK();
        p->cur = nxt;
        return WH_OK;
    }
    PfbMidCall c;
    c.x = d_iq; c.fmt = input_format; c.n = n;
    c.hist = p->d_hist[p->cur]; c.new_hist = p->d_hist[p->cur ^ 1];
    c.out = nullptr; c.sink = p->d_sink; c.sink_elems = (size_t)p->M;
    c.arms = p->d_arms; c.tw = p->d_tw;
    c.H = H; c.cu_count = p->cu_count; c.hops_per_run = p->gpw_override;
    c.stats_only = 1; c.stats_ws = nullptr; c.stats_out = d_stats; c.stats_accumulate = accumulate;
    long long grid = 0;
    int rc = pfb_mid_launch(p->M, p->T, c, st, &grid);
    if (rc != WH_OK) return rc;
    if ((size_t)grid > p->stats_ws_rows) {      // grow the workspace (synchronises; a steady call size never does)
        WH_HIP(hipStreamSynchronize(st));
        (void)hipFree(p->d_stats_ws);
        p->d_stats_ws = nullptr;
        p->stats_ws_rows = 0;
        WH_HIP(hipMalloc(&p->d_stats_ws, (size_t)grid * 4 * p->M * sizeof(double)));
        p->stats_ws_rows = (size_t)grid;
    }
    c.stats_ws = p->d_stats_ws;
    if (p->prof) WH_HIP(hipEventRecord(p->ev0[p->ev_begin()], st));
    if ((rc = pfb_mid_launch(p->M, p->T, c, st)) != WH_OK) return rc;
    if (p->prof) {
        WH_HIP(hipEventRecord(p->ev1[p->ev_cur], st));
        p->ev_end();
    }
    p->cur ^= 1;
    return WH_OK;
}

extern "C" int wh_stats_merge(const double *d_gathered, int n_ranks, int n_channels, double *d_out, void *stream) {
    if (!d_gathered || !d_out || n_ranks < 1 || n_channels < 1) return set_err(WH_E_ARG, "wh_stats_merge: bad arguments");
    hipLaunchKernelGGL(stats_merge_kernel, dim3((n_channels + 255) / 256), dim3(256), 0, as_stream(stream), d_gathered,
                       n_ranks, n_channels, d_out);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_pfb_reset(wh_pfb *p, void *stream) {
    if (!p) return set_err(WH_E_ARG, "wh_pfb_reset: null handle");
    WH_HIP(hipMemsetAsync(p->d_hist[p->cur], 0, (size_t)p->M * p->T * sizeof(float2), as_stream(stream)));
    return WH_OK;
}

extern "C" int wh_pfb_get_history(wh_pfb *p, float *h_hist, void *stream) {
    if (!p || !h_hist) return set_err(WH_E_ARG, "wh_pfb_get_history: null");
    WH_HIP(hipMemcpyAsync(h_hist, p->d_hist[p->cur], (size_t)p->M * p->T * sizeof(float2), hipMemcpyDeviceToHost,
                          as_stream(stream)));
    WH_HIP(hipStreamSynchronize(as_stream(stream)));
    return WH_OK;
}

extern "C" int wh_pfb_set_history(wh_pfb *p, const float *h_hist, void *stream) {
    if (!p || !h_hist) return set_err(WH_E_ARG, "wh_pfb_set_history: null");
    WH_HIP(hipMemcpyAsync(p->d_hist[p->cur], h_hist, (size_t)p->M * p->T * sizeof(float2), hipMemcpyHostToDevice,
                          as_stream(stream)));
    WH_HIP(hipStreamSynchronize(as_stream(stream)));
    return WH_OK;
}

extern "C" int wh_pfb_extract_channel(const float *d_out, size_t hops, int M, int idx, float *d_col, void *stream) {
    if (hops == 0) return WH_OK;
    if (!d_out || !d_col || idx < 0 || idx >= M) return set_err(WH_E_ARG, "wh_pfb_extract_channel: bad args");
    hipLaunchKernelGGL(extract_channel_kernel, dim3((unsigned)((hops + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2 *>(d_out), hops, M, idx, reinterpret_cast<float2 *>(d_col));
    WH_LAUNCH_CHECK();
    return WH_OK;
}

extern "C" int wh_pfb_channel_stats(wh_pfb *p, const float *d_out, size_t hops, double *d_stats, int accumulate,
                                    void *stream) {
    if (!p || !d_out || !d_stats) return set_err(WH_E_ARG, "wh_pfb_channel_stats: bad args");
    if (hops == 0) return WH_OK;
    const int M = p->M;
    hipStream_t st = as_stream(stream);
    int slices = (int)((hops + 15) / 16);   // >= 16 hops per slice, up to 64 slices x M/64 column blocks
    if (slices > 64) slices = 64;
    if (slices < 1) slices = 1;
    double *part = p->d_part;   // [64][M][4] workspace owned by the handle (no allocation on the hot path)
    hipLaunchKernelGGL(pfb_stats_kernel, dim3((M + 63) / 64, slices), dim3(64, 4), 0, st,
                       reinterpret_cast<const float2 *>(d_out), hops, M, part, slices);
    WH_LAUNCH_CHECK();
    hipLaunchKernelGGL(pfb_stats_final_kernel, dim3((M + 3) / 4), dim3(256), 0, st, part, slices, M, hops,
                       d_stats, accumulate);
    WH_LAUNCH_CHECK();
    return WH_OK;
}
