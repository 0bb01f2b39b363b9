// Dependent-issue latency and independent-issue rate of VALU ops on gfx950, one wave per CU (diagnostics).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_latency tools/ubench/valu_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(double *out, long long *cyc, int iters, double a, double b) {
    double x0 = threadIdx.x * 1e-3 + 1.0, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    float f0 = (float)x0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    const float fa = (float)a, fb = (float)b;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) { x0 = __dmul_rn(x0, a); }                                    // dependent f64 mul
            if (MODE == 1) { x0 = __dadd_rn(x0, b); }                                    // dependent f64 add
            if (MODE == 2) { x0 = fma(x0, a, b); }                                       // dependent f64 fma
            if (MODE == 3) { x0 = __dmul_rn(x0, a); x1 = __dmul_rn(x1, a); x2 = __dmul_rn(x2, a); x3 = __dmul_rn(x3, a); }  // 4 independent
            if (MODE == 4) { f0 = __fmul_rn(f0, fa); }                                   // dependent f32 mul
            if (MODE == 5) { f0 = __fmul_rn(f0, fa); f1 = __fmul_rn(f1, fa); f2 = __fmul_rn(f2, fa); f3 = __fmul_rn(f3, fa); }
            if (MODE == 6) { x0 = __dadd_rn(__dmul_rn(x0, a), b); }                      // dependent mul -> add pair
            if (MODE == 7) { x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b); }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + f0 + f1 + f2 + f3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 1024 * 256 * 8); hipMalloc(&cyc, 1024 * 8);
    const int iters = 20000;
    const char *names[8] = {"dep f64 mul", "dep f64 add", "dep f64 fma", "4 indep f64 mul", "dep f32 mul", "4 indep f32 mul",
                            "dep f64 mul+add", "4 indep f64 fma"};
    const int ops[8] = {16, 16, 16, 64, 16, 64, 32, 64};
#define RUN(M, BL, TH)                                                                                    \
    {                                                                                                      \
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);                                       \
        k<M><<<BL, TH>>>(out, cyc, 100, 1.0000001, 1e-9); hipDeviceSynchronize();                          \
        hipEventRecord(e0); k<M><<<BL, TH>>>(out, cyc, iters, 1.0000001, 1e-9); hipEventRecord(e1);        \
        hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);                                \
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);                                         \
        printf("%-18s blocks %4d x %3d thr: %8.3f ms  %7.2f ns/op  %7.2f counter ticks/op\n", names[M], BL, TH, ms,   \
               ms * 1e6 / ((double)iters * ops[M]), (double)c / ((double)iters * ops[M]));                 \
    }
    RUN(0, 1, 64) RUN(1, 1, 64) RUN(2, 1, 64) RUN(6, 1, 64) RUN(3, 1, 64) RUN(7, 1, 64) RUN(4, 1, 64) RUN(5, 1, 64)
    RUN(3, 1, 256) RUN(7, 1, 256) RUN(3, 1, 512) RUN(5, 1, 512)
    return 0;
}
