// Chip-wide VALU issue rate on gfx950 under full load (diagnostics): every SIMD runs W waves of independent packed
// float32 FMAs (or plain FMAs / moves); reports wave-instructions per ns per SIMD, i.e. the clock the chip sustains
// divided by the issue cycles per wave64 instruction.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate tools/ubench/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    v2f x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = v2f{threadIdx.x * 1e-3f + i, 1.0f + i};
    const v2f va = {a, a}, vb = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) x[i] = __builtin_elementwise_fma(x[i], va, vb);        // v_pk_fma_f32
                if (MODE == 1) { x[i].x = fmaf(x[i].x, a, b); }                       // v_fma_f32
                if (MODE == 2) x[i] = x[i] + vb;                                      // v_pk_add_f32
                if (MODE == 3) { asm volatile("v_mov_b64 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 7])); }   // v_mov_b64
                if (MODE == 4) { asm volatile("v_mov_b32 %0, %1" : "=v"(x[i].x) : "v"(x[(i + 1) & 7].y)); }
            }
    }
    v2f s = x[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

int main() {
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const char *names[5] = {"v_pk_fma_f32", "v_fma_f32", "v_pk_add_f32", "v_mov_b64", "v_mov_b32"};
#define RUN(M, WPS)                                                                                                 \
    {                                                                                                                \
        const int iters = 20000, blocks = cus * WPS;   /* 4 waves per block = one per SIMD */                        \
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);                                                 \
        k<M><<<blocks, 256>>>(out, 100, 1.0000001f, 1e-9f); hipDeviceSynchronize();                                  \
        hipEventRecord(e0); k<M><<<blocks, 256>>>(out, iters, 1.0000001f, 1e-9f); hipEventRecord(e1);                \
        hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);                                          \
        const double inst = (double)iters * 64 * WPS;   /* wave instructions per SIMD */                             \
        printf("%-14s %d waves/SIMD on %d CUs: %8.3f ms  %6.3f wave-instr/ns/SIMD  (%.2f ns each; at 4 cycles: %.2f GHz)\n", \
               names[M], WPS, cus, ms, inst / (ms * 1e6), ms * 1e6 / inst, 4.0 * inst / (ms * 1e6));                  \
    }
    RUN(0, 1) RUN(0, 2) RUN(0, 4) RUN(1, 4) RUN(2, 4) RUN(3, 4) RUN(4, 4)
    return 0;
}
