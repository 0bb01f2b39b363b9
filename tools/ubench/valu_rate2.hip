// VALU issue rate by operand source on gfx950 (diagnostics): packed FMA with 1 / 2 / 3 VGPR-pair sources, with op_sel
// broadcast, plain FMA with 3 VGPR sources, packed add / mul with 2 VGPR-pair sources.  4 waves per SIMD, all CUs.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate2 tools/ubench/valu_rate2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    v2f x[8], t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = v2f{threadIdx.x * 1e-3f + i, 1.0f + i}; t[i] = v2f{1.0f + 1e-7f * threadIdx.x, 1.0f - 1e-7f * i}; }
    const v2f va = {a, a}, vb = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) x[i] = __builtin_elementwise_fma(x[i], va, vb);                 // 1 VGPR pair + 2 SGPR
                if (MODE == 1) x[i] = __builtin_elementwise_fma(x[i], t[i], vb);               // 2 VGPR pairs
                if (MODE == 2) x[i] = __builtin_elementwise_fma(t[(i + 1) & 7], t[i], x[i]);   // 3 VGPR pairs (MAC shape)
                if (MODE == 3) x[i] = __builtin_elementwise_fma(t[(i + 1) & 7], __builtin_shufflevector(t[i], t[i], 0, 0), x[i]);  // op_sel broadcast
                if (MODE == 4) { x[i].x = fmaf(t[(i + 1) & 7].x, t[i].x, x[i].x); }          // v_fma_f32, 3 VGPRs
                if (MODE == 5) x[i] = x[i] + t[i];                                             // v_pk_add_f32, 2 VGPR pairs
                if (MODE == 6) x[i] = x[i] * t[i];                                             // v_pk_mul_f32
                if (MODE == 7) { x[i].x = x[i].x + t[i].x; }                                   // v_add_f32
            }
        asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
    }
    v2f s = x[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

int main() {
    float *out; (void)hipMalloc(&out, 4096 * 256 * 4);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const char *names[8] = {"pk_fma 1 vgpr-pair", "pk_fma 2 vgpr-pairs", "pk_fma 3 vgpr-pairs", "pk_fma 3 + op_sel bc", "v_fma_f32 3 vgprs",
                            "pk_add 2 pairs", "pk_mul 2 pairs", "v_add_f32"};
#define RUN(M, WPS)                                                                                                 \
    {                                                                                                                \
        const int iters = 20000, blocks = cus * WPS;                                                                 \
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);                                                 \
        k<M><<<blocks, 256>>>(out, 100, 1.0000001f, 1e-9f); (void)hipDeviceSynchronize();                            \
        hipEventRecord(e0); k<M><<<blocks, 256>>>(out, iters, 1.0000001f, 1e-9f); hipEventRecord(e1);                \
        (void)hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);                                    \
        const double inst = (double)iters * 64 * WPS;                                                                \
        printf("%-22s %d waves/SIMD: %8.3f ms  %6.3f wave-instr/ns/SIMD  (%.2f ns each)\n", names[M], WPS, ms,       \
               inst / (ms * 1e6), ms * 1e6 / inst);                                                                  \
    }
    RUN(0, 4) RUN(1, 4) RUN(2, 4) RUN(3, 4) RUN(4, 4) RUN(5, 4) RUN(6, 4) RUN(7, 4) RUN(2, 1) RUN(2, 2)
    return 0;
}
