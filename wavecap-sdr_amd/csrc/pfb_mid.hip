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

constexpr MidPlan mid_plan(int Q) {
    MidPlan p{};
    int rem = Q, L = Q, n = 0;
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
__device__ __forceinline__ float2 ld_iq(const void *p, long long i) {
    if (FMT == 1) {   // A1 unpack rule (cli.py:447-452): int16 / 32768
        short2 v = reinterpret_cast<const short2 *>(p)[i];
        return make_float2((float)v.x * (1.0f / 32768.0f), (float)v.y * (1.0f / 32768.0f));
    }
    return reinterpret_cast<const float2 *>(p)[i];
}

// forward butterflies, natural order in and out
__device__ __forceinline__ void bfly(float2 (&v)[2]) {
    float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
__device__ __forceinline__ void bfly(float2 (&v)[3]) {
    const float C = -0.5f, S = -0.86602540378443864676f;   // exp(-2 pi i / 3)
    float2 t1 = cadd(v[1], v[2]), t2 = csub(v[1], v[2]);
    float2 u = make_float2(fmaf(C, t1.x, v[0].x), fmaf(C, t1.y, v[0].y));
    float2 w = make_float2(-S * t2.y, S * t2.x);            // i * S * t2
    v[0] = cadd(v[0], t1);
    v[1] = cadd(u, w);
    v[2] = csub(u, w);
}
__device__ __forceinline__ void bfly(float2 (&v)[4]) { fft4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void bfly(float2 (&v)[5]) {
    const float C1 = 0.30901699437494742410f, S1 = 0.95105651629515357212f;    // cos, sin 2 pi / 5
    const float C2 = -0.80901699437494742410f, S2 = 0.58778525229247312917f;   // cos, sin 4 pi / 5
    float2 v0 = v[0];
    float2 a1 = cadd(v[1], v[4]), b1 = csub(v[1], v[4]), a2 = cadd(v[2], v[3]), b2 = csub(v[2], v[3]);
    float2 r1 = make_float2(fmaf(C2, a2.x, fmaf(C1, a1.x, v0.x)), fmaf(C2, a2.y, fmaf(C1, a1.y, v0.y)));
    float2 r2 = make_float2(fmaf(C1, a2.x, fmaf(C2, a1.x, v0.x)), fmaf(C1, a2.y, fmaf(C2, a1.y, v0.y)));
    // -i (S1 b1 + S2 b2), -i (S2 b1 - S1 b2)   (forward transform)
    float2 i1 = make_float2(fmaf(S2, b2.y, S1 * b1.y), -fmaf(S2, b2.x, S1 * b1.x));
    float2 i2 = make_float2(fmaf(-S1, b2.y, S2 * b1.y), -fmaf(-S1, b2.x, S2 * b1.x));
    v[0] = make_float2(v0.x + a1.x + a2.x, v0.y + a1.y + a2.y);
    v[1] = cadd(r1, i1);
    v[2] = cadd(r2, i2);
    v[3] = csub(r2, i2);
    v[4] = csub(r1, i1);
}

// packed FMA with one float of a tap PAIR broadcast to both halves (op_sel): (re, im) * tap + acc as one instruction,
// with the 36 taps of a quad in 18 register pairs.  Written as the compiler would for a {t, t} operand, it keeps a
// duplicated copy of every tap (72 registers).
__device__ __forceinline__ v2f pk_fma_bc(int hi, v2f a, v2f tpair, v2f c) {   // hi: constant after unrolling
    v2f d;
    if (hi) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(tpair), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(tpair), "v"(c));
    return d;
}

struct MidArgs {
    const void *x;
    const float2 *hist;
    float2 *new_hist;
    float2 *out;
    const float *arms;
    const float2 *tw;
    long long H;           // hops of the call
    long long n;           // samples of the call
    long long max_block;   // last half-block index fully inside the input
    int hpr;               // hops per run (multiple of GH)
    int n_head;            // workgroups [0, n_head) do the head hops, workgroup n_head the history, the rest the runs
};

// passes PI .. np-2 of the sub-transforms of NI hop images, in place; TH threads cooperate (lt = index among them)
template <int M, int PI, int TH, int NI, bool WG>
__device__ __forceinline__ void mid_passes(float2 *im, const float2 *twp, int lt) {
    constexpr MidPlan P = mid_plan(M / 4);
    if constexpr (PI < P.np - 1) {
        constexpr int r = P.r[PI], L = P.L[PI], m = L / r;
        constexpr int BPI = M / r;              // butterflies per image
        constexpr int NB = NI * BPI;
        constexpr int ROUNDS = (NB + TH - 1) / TH;
        constexpr int TWO = P.two[PI];
#pragma unroll
        for (int c = 0; c < ROUNDS; ++c) {
            const int b = lt + TH * c;
            if ((c + 1) * TH <= NB || b < NB) {
                const int s = b / BPI, bb = b - s * BPI;
                const int blk = bb / m, up = bb - blk * m;
                float2 *p = im + s * M + blk * L + up;
                const float2 *tq = twp + TWO + up;
                float2 v[r];
#pragma unroll
                for (int j = 0; j < r; ++j) v[j] = p[j * m];
                bfly(v);
                p[0] = v[0];
#pragma unroll
                for (int k = 1; k < r; ++k) p[k * m] = cmul(v[k], tq[(k - 1) * m]);
            }
        }
        if (WG) __syncthreads();
        else __builtin_amdgcn_wave_barrier();
        mid_passes<M, PI + 1, TH, NI, WG>(im, twp, lt);
    }
}

// last pass: image words -> channel outputs in HBM.  Image s of the NI belongs to hop hop0 + (s0 + s) / GH * stride_r +
// (s0 + s) % GH (s0 = index of im's first image in the workgroup), stored when < limit.
template <int M, int GH, int TH, int NI>
__device__ __forceinline__ void mid_last(const float2 *im, float2 *out, int lt, int s0, long long hop0, int stride_r,
                                         long long limit) {
    constexpr MidPlan P = mid_plan(M / 4);
    constexpr int Q = M / 4;
    constexpr int r = P.r[P.np - 1];
    constexpr int BPI = M / r;
    constexpr int NB = NI * BPI;
    constexpr int ROUNDS = (NB + TH - 1) / TH;
#pragma unroll
    for (int c = 0; c < ROUNDS; ++c) {
        const int b = lt + TH * c;
        if ((c + 1) * TH <= NB || b < NB) {
            const int s = b / BPI, kb = b - s * BPI;
            // digit reversal: kb = d1 + 4 (d2 + r_0 (d3 + ...)) sits at d1 Q + d2 m_0 + d3 m_1 + ...
            int rem = kb >> 2, pos = (kb & 3) * Q;
#pragma unroll
            for (int p = 0; p < P.np - 1; ++p) {
                const int rp = P.r[p], mp = P.L[p] / P.r[p];
                const int nx = rem / rp;
                pos += (rem - nx * rp) * mp;
                rem = nx;
            }
            const float2 *q = im + s * M + pos;
            float2 v[r];
#pragma unroll
            for (int j = 0; j < r; ++j) v[j] = q[j];
            bfly(v);
            const int sg = s0 + s;
            const long long hop = hop0 + (long long)(sg / GH) * stride_r + (sg % GH);
            if (hop < limit) {
                float2 *o = out + (size_t)hop * M + kb;
#pragma unroll
                for (int k = 0; k < r; ++k) o[k * BPI] = v[k];
            }
        }
    }
}


// the passes of the workgroup's hop images + their stores (wave mode: wave w owns images [w HPW, (w+1) HPW))
template <int M, int GH, int NT, int NIMG, int NWF>
__device__ __forceinline__ void mid_transform(float2 *img, const float2 *twp, float2 *out, int tid, long long hop_g,
                                              int stride_r, long long limit) {
    constexpr bool WAVE_MODE = NWF > 0;
    constexpr int HPW = WAVE_MODE ? NIMG / NWF : NIMG;
    // opaque copy of the thread index: the image / twiddle / output offsets of the passes are loop invariant and would
    // otherwise be hoisted out of the group loop and parked in registers for the whole run
    int lt = WAVE_MODE ? (tid & 63) : tid;
    asm volatile("" : "+v"(lt));
    if (WAVE_MODE) {
        const int wave = tid >> 6;
        if (wave < NWF) {
            float2 *im = img + wave * HPW * M;
            mid_passes<M, 0, 64, HPW, false>(im, twp, lt);
            mid_last<M, GH, 64, HPW>(im, out, lt, wave * HPW, hop_g, stride_r, limit);
        }
    } else {
        mid_passes<M, 0, NT, NIMG, true>(img, twp, lt);
        mid_last<M, GH, NT, NIMG>(img, out, lt, 0, hop_g, stride_r, limit);
    }
}

template <int M, int T, int R, int GH, int NWF, int WPE, int FMT>
__global__ __launch_bounds__(R * (M / 4)) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void pfb_mid_kernel(MidArgs a) {
    constexpr int Q = M / 4, HB = M / 2, NT = R * Q, NW = (NT + 63) / 64, NIMG = R * GH;
    constexpr MidPlan P = mid_plan(Q);
    static_assert(P.ok, "M/4 must factor into 2, 3, 5");
    static_assert(NT <= 1024, "workgroup too large");
    constexpr bool WAVE_MODE = NWF > 0;
    static_assert(!WAVE_MODE || (NIMG % NWF == 0 && NWF <= NW), "wave mode: whole hop images per wave");
    constexpr int HPW = WAVE_MODE ? NIMG / NWF : NIMG;
    // the images a wave transforms are written by that wave alone: no workgroup barrier at all
    constexpr bool SELF = WAVE_MODE && NWF == NW && HPW % GH == 0 && (HPW / GH) * Q == 64;

    __shared__ __attribute__((aligned(16))) float2 img[NIMG * M];
    __shared__ float2 twp[P.twn > 0 ? P.twn : 1];
    // taps of quad u: 36 floats (e = q T + j) padded to TP4 float4; the 144-byte row stride makes the ds_read_b128 of 16
    // consecutive lanes cover all 64 banks once
    constexpr int TP4 = (4 * T + 3) / 4;
    __shared__ float4 tapl[Q * TP4];

    const int tid = threadIdx.x;
    const int bid = blockIdx.x;

    if (bid == a.n_head) {   // history for the next call: new_hist[k][j] = block_{H-1-j}[k]
        for (int idx = tid; idx < M * T; idx += NT) {
            const int k = idx / T, j = idx - k * T;
            const long long g = a.H - 1 - j;
            float2 v;
            if (g >= 0) v = ld_iq<FMT>(a.x, g * HB + k);
            else {
                const int col = (int)(-g - 1);
                v = col < T ? a.hist[(size_t)k * T + col] : make_float2(0.f, 0.f);
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
            twp[P.two[p] + e] = a.tw[(up * k * (M / Lp)) % M];
        }
    }

    const int r = tid / Q, u = tid - r * Q;

    if (bid < a.n_head) {
        // head hops: every column from the carried history / the stream, j ascending like the packed MAC below
        const long long hop_base = (long long)bid * NIMG;
        const long long limit = a.H < T - 1 ? a.H : T - 1;
        constexpr int HJ = NIMG * Q;
        for (int j0 = tid; j0 < HJ; j0 += NT) {
            const int s = j0 / Q, uu = j0 - s * Q;
            const long long hop = hop_base + s;
            if (hop >= limit) continue;
            float2 z[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = uu + q * Q;
                float re = 0.f, im = 0.f;
#pragma nounroll
                for (int j = 0; j < T; ++j) {
                    const long long gb = hop - j;
                    float2 c;
                    if (gb >= 0) {
                        long long idx = gb * HB + k;
                        if (idx >= a.n) idx = a.n - 1;
                        c = ld_iq<FMT>(a.x, idx);
                    } else {
                        c = a.hist[(size_t)k * T + (int)(-gb - 1)];
                    }
                    const float w = a.arms[(size_t)k * T + j];
                    re = fmaf(c.x, w, re);
                    im = fmaf(c.y, w, im);
                }
                z[q] = make_float2(re, im);
            }
            fft4(z[0], z[1], z[2], z[3]);
            float2 *L = img + s * M + uu;
            L[0] = z[0];
            L[Q] = cmul(z[1], a.tw[uu]);
            L[2 * Q] = cmul(z[2], a.tw[2 * uu]);
            L[3 * Q] = cmul(z[3], a.tw[3 * uu]);
        }
        __syncthreads();   // images and twiddle tables
        mid_transform<M, GH, NT, NIMG, NWF>(img, twp, a.out, tid, hop_base, GH, limit);
        return;
    }

    const long long hop_base = (T - 1) + (long long)(bid - a.n_head - 1) * R * a.hpr;
    const int ngroups = a.hpr / GH;

    // stage-1 twiddles W_M^(u k1)
    const float2 tw1 = a.tw[u], tw2 = a.tw[2 * u], tw3 = a.tw[3 * u];
    for (int e = tid; e < Q * TP4 * 4; e += NT) {
        const int uu = e / (TP4 * 4), f = e - uu * (TP4 * 4);   // f = q T + j
        const int q = f / T, j = f - q * T;
        reinterpret_cast<float *>(tapl)[e] = f < 4 * T ? a.arms[(size_t)(uu + q * Q) * T + j] : 0.f;
    }
    // wA[i] = x[(h - (T-1) + i) HB + u], wB[i] = ... + Q; slots T.. hold the group's new blocks
    v2f wA[T + GH], wB[T + GH];
    long long h = hop_base + (long long)r * a.hpr;   // this thread's run: next hop
#pragma unroll
    for (int i = 0; i < T + GH; ++i) {
        long long g = h - (T - 1) + i;
        if (g > a.max_block) g = a.max_block;
        const float2 va = ld_iq<FMT>(a.x, g * HB + u), vb = ld_iq<FMT>(a.x, g * HB + u + Q);
        wA[i] = v2f{va.x, va.y};
        wB[i] = v2f{vb.x, vb.y};
    }
    __syncthreads();   // twiddle and tap tables

    for (int g = 0; g < ngroups; ++g) {
        const long long hop_g = hop_base + (long long)g * GH;
        if (hop_g >= a.H) break;   // uniform: this and every later group of the workgroup is past the end
        {
            // the quad's taps, re-read every group (opaque index: hoisted out of the loop they would hold 36 registers
            // through the transform phase; here they are live during the MAC only)
            int uo = u;
            asm volatile("" : "+v"(uo));
            v2f tp[2 * TP4];
#pragma unroll
            for (int k = 0; k < TP4; ++k) {
                const float4 t4 = tapl[uo * TP4 + k];
                tp[2 * k] = v2f{t4.x, t4.y};
                tp[2 * k + 1] = v2f{t4.z, t4.w};
            }
#define WH_TAPFMA(q, j, w, acc) pk_fma_bc(((q) * T + (j)) & 1, w, tp[((q) * T + (j)) >> 1], acc)
#pragma unroll
            for (int i = 0; i < GH; ++i) {
                // hop h+i: column u uses c_{h+i-j} = w[i + T-1 - j]; column u + HB uses w[i + T - j]
                v2f y0 = v2f{0.f, 0.f}, y1 = y0, y2 = y0, y3 = y0;
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    y0 = WH_TAPFMA(0, j, wA[i + T - 1 - j], y0);
                    y1 = WH_TAPFMA(1, j, wB[i + T - 1 - j], y1);
                    y2 = WH_TAPFMA(2, j, wA[i + T - j], y2);
                    y3 = WH_TAPFMA(3, j, wB[i + T - j], y3);
                }
                float2 z0 = make_float2(y0.x, y0.y), z1 = make_float2(y1.x, y1.y);
                float2 z2 = make_float2(y2.x, y2.y), z3 = make_float2(y3.x, y3.y);
                fft4(z0, z1, z2, z3);
                float2 *L = img + (r * GH + i) * M + u;
                L[0] = z0;
                L[Q] = cmul(z1, tw1);
                L[2 * Q] = cmul(z2, tw2);
                L[3 * Q] = cmul(z3, tw3);
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
        if (g + 1 < ngroups) {
#pragma unroll
            for (int i = 0; i < GH; ++i) {
                long long gb = h + 1 + i;
                if (gb > a.max_block) gb = a.max_block;
                const float2 va = ld_iq<FMT>(a.x, gb * HB + u), vb = ld_iq<FMT>(a.x, gb * HB + u + Q);
                wA[T + i] = v2f{va.x, va.y};
                wB[T + i] = v2f{vb.x, vb.y};
            }
        }
        if (SELF) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
        mid_transform<M, GH, NT, NIMG, NWF>(img, twp, a.out, tid, hop_g, a.hpr, a.H);
        if (SELF) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
    }
}

// ---- host side -------------------------------------------------------------------------------------------------

// (M, R runs per workgroup, GH hops per group, NWF waves running the passes; 0 = all threads with workgroup barriers,
//  WPE waves per SIMD the register allocation is held to)
#define WH_MID_CONFIGS(X) \
    X(320, 4, 4, 4, 4)

template <int M, int T, int R, int GH, int NWF, int WPE>
int mid_launch_t(const PfbMidCall &c, hipStream_t st) {
    constexpr int NT = R * (M / 4), NIMG = R * GH;
    static int wg_per_cu[2] = {0, 0};   // occupancy of the two format instances (same for every device of the node)
    const int f = c.fmt == 1 ? 1 : 0;
    auto kern = f ? pfb_mid_kernel<M, T, R, GH, NWF, WPE, 1> : pfb_mid_kernel<M, T, R, GH, NWF, WPE, 0>;
    if (wg_per_cu[f] == 0) {
        int nb = 0;
        WH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NT, 0));
        wg_per_cu[f] = nb > 0 ? nb : 1;
    }
    MidArgs a;
    a.x = c.x; a.hist = c.hist; a.new_hist = c.new_hist; a.out = c.out; a.arms = c.arms; a.tw = c.tw;
    a.H = c.H; a.n = (long long)c.n;
    a.max_block = (long long)(c.n / (size_t)(M / 2)) - 1;
    const long long nh = c.H - (T - 1);   // hops of the runs
    int hpr = GH;
    long long n_main = 0;
    if (nh > 0) {
        // every workgroup resident at once when the input allows (a second, nearly empty round of workgroups would
        // double the time of a small call), runs of at most 128 hops (halo 9 / 128), at least 16 (halo 9 / 16)
        const long long slots = (long long)c.cu_count * wg_per_cu[f];
        long long v = 0;
        for (long long k = 1; k <= 4096; ++k) {
            v = (nh + R * slots * k - 1) / (R * slots * k);
            v = (v + GH - 1) / GH * GH;
            if (v <= 128) break;
        }
        if (v < 16) v = (16 + GH - 1) / GH * GH;
        if (c.hops_per_run > 0) v = (c.hops_per_run + GH - 1) / GH * GH;
        hpr = (int)v;
        const long long runs = (nh + hpr - 1) / hpr;
        n_main = (runs + R - 1) / R;
    }
    a.hpr = hpr;
    const long long head_hops = c.H < T - 1 ? c.H : T - 1;
    a.n_head = (int)((head_hops + NIMG - 1) / NIMG);
    const long long grid = a.n_head + 1 + n_main;
    if (grid > 0x7fffffffLL) return set_err(WH_E_ARG, "wh_pfb_run: input too long for one launch");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), 0, st, a);
    WH_LAUNCH_CHECK();
    return WH_OK;
}

}  // namespace

namespace wh {

bool pfb_mid_supported(int M, int T) {
    if (T != 9) return false;
#define X(M_, R_, GH_, NWF_, WPE_) if (M == M_) return true;
    WH_MID_CONFIGS(X)
#undef X
    return false;
}

int pfb_mid_launch(int M, int T, const PfbMidCall &c, hipStream_t st) {
    if (T == 9) {
#define X(M_, R_, GH_, NWF_, WPE_) if (M == M_) return mid_launch_t<M_, 9, R_, GH_, NWF_, WPE_>(c, st);
        WH_MID_CONFIGS(X)
#undef X
    }
    return set_err(WH_E_ARG, "pfb_mid_launch: no instance for M=%d T=%d", M, T);
}

const char *pfb_mid_kernel_name() { return "pfb_mid_kernel"; }

}  // namespace wh
