// Shared host/device helpers for libwavehip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "../../include/wavehip.h"

namespace wh {

// thread-local error text (no global mutable state shared across threads)
inline char *err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
inline int set_err(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define WH_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return wh::set_err(WH_E_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,        \
                               hipGetErrorString(_e));                                       \
    } while (0)

#define WH_LAUNCH_CHECK()                                                                    \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess)                                                                \
            return wh::set_err(WH_E_HIP, "%s:%d launch -> %s", __FILE__, __LINE__,           \
                               hipGetErrorString(_e));                                       \
    } while (0)

struct cf {  // complex float32, layout-compatible with numpy complex64 / float2
    float x, y;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i : (x + iy)(-i) = y - ix
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }
// multiply by +i
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }

// forward radix-4 butterfly (W = exp(-2 pi i/4) = -i), in place, natural order
__device__ __forceinline__ void fft4(float2 &a0, float2 &a1, float2 &a2, float2 &a3) {
    float2 s02 = cadd(a0, a2), d02 = csub(a0, a2);
    float2 s13 = cadd(a1, a3), d13 = csub(a1, a3);
    float2 md = mul_mi(d13);
    a0 = cadd(s02, s13);
    a1 = cadd(d02, md);
    a2 = csub(s02, s13);
    a3 = csub(d02, md);
}

// forward 16-point FFT on registers; v[n] natural order in, natural order out.
__device__ __forceinline__ void fft16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128673848f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;  // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;
    // step 1: for each na, radix-4 over nb (elements na, na+4, na+8, na+12) -> Y[na][kb]
#pragma unroll
    for (int na = 0; na < 4; ++na) fft4(v[na], v[na + 4], v[na + 8], v[na + 12]);
    // v[na + 4*kb] = Y[na][kb];  step 2: twiddle W16^(na*kb)
    v[1 + 4] = cmul(v[1 + 4], make_float2(C1, -S1));    // W^1
    v[1 + 8] = cmul(v[1 + 8], make_float2(R2, -R2));    // W^2
    v[1 + 12] = cmul(v[1 + 12], make_float2(S1, -C1));  // W^3
    v[2 + 4] = cmul(v[2 + 4], make_float2(R2, -R2));    // W^2
    v[2 + 8] = mul_mi(v[2 + 8]);                        // W^4
    v[2 + 12] = cmul(v[2 + 12], make_float2(-R2, -R2)); // W^6
    v[3 + 4] = cmul(v[3 + 4], make_float2(S1, -C1));    // W^3
    v[3 + 8] = cmul(v[3 + 8], make_float2(-R2, -R2));   // W^6
    v[3 + 12] = cmul(v[3 + 12], make_float2(-C1, S1));  // W^9
    // step 3: for each kb, radix-4 over na -> X[4*ka + kb] lands in v[ka + 4*kb]
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) fft4(v[4 * kb], v[4 * kb + 1], v[4 * kb + 2], v[4 * kb + 3]);
    // now v[ka + 4*kb] = X[4*ka + kb]: transpose the 4x4 index to natural order
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) {
            float2 t = v[a + 4 * b];
            v[a + 4 * b] = v[b + 4 * a];
            v[b + 4 * a] = t;
        }
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace wh
