// Do global stores count in vmcnt on gfx950, and do loads issued BEFORE a batch of stores complete in order (so that
// `s_waitcnt vmcnt(<number of younger stores>)` is enough to use the loaded value)?  Diagnostics.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/vmcnt_order tools/ubench/vmcnt_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// A: store + vmcnt(0) per iteration vs store only: is the store's completion waited for?
template <int WAIT>
__global__ void k_store_wait(float4 *out, long long *cyc, int iters) {
    float4 v = make_float4(threadIdx.x, 1, 2, 3);
    size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        out[idx + (size_t)i * 64 * gridDim.x] = v;
        if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// B: load (asm, untracked by the compiler) -> 8 stores -> s_waitcnt vmcnt(8) -> use the loaded value.
// Counts mismatches against the known content of the source buffer.  N_AFTER = count passed to vmcnt.
template <int N_AFTER>
__global__ void k_order(const float *src, float4 *sink, int *bad, int iters, size_t span) {
    int nbad = 0;
    size_t base = ((size_t)blockIdx.x * 64 + threadIdx.x);
    for (int i = 0; i < iters; ++i) {
        const float *p = src + (base * 97 + (size_t)i * 1048583) % span;   // scattered: misses the caches
        float got;
        asm volatile("global_load_dword %0, %1, off" : "=v"(got) : "v"(p) : "memory");
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v = {(float)i, 1.f, 2.f, 3.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float4 *q = sink + ((base + (size_t)(i * 8 + u) * 64 * gridDim.x) % span) / 1;
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(q), "v"(v) : "memory");
        }
        if (N_AFTER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float expect = (float)(((base * 97 + (size_t)i * 1048583) % span) & 0xFFFFF);
        asm volatile("" : "+v"(got));
        if (got != expect) nbad++;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    atomicAdd(bad, nbad);
}

int main() {
    const size_t span = (size_t)1 << 28;   // 1 GiB of floats / 4 GiB of float4 is too much: use separate sizes
    float *src; float4 *sink; long long *cyc; int *bad;
    hipMalloc(&src, span * 4);
    hipMalloc(&sink, span * 16);
    hipMalloc(&cyc, 4096 * 8); hipMalloc(&bad, 4);
    std::vector<float> h(span);
    for (size_t i = 0; i < span; ++i) h[i] = (float)(i & 0xFFFFF);
    hipMemcpy(src, h.data(), span * 4, hipMemcpyHostToDevice);
    for (int blocks : {1, 1024}) {
        long long c0, c1;
        k_store_wait<0><<<blocks, 64>>>(sink, cyc, 2000); hipDeviceSynchronize(); hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
        k_store_wait<1><<<blocks, 64>>>(sink, cyc, 2000); hipDeviceSynchronize(); hipMemcpy(&c1, cyc, 8, hipMemcpyDeviceToHost);
        printf("blocks %4d: store only %6.1f cycles/iter, store + vmcnt(0) %7.1f cycles/iter\n", blocks, c0 / 2000.0, c1 / 2000.0);
    }
    for (int blocks : {64, 2048}) {
        int z = 0, b8 = 0, b0 = 0;
        hipMemcpy(bad, &z, 4, hipMemcpyHostToDevice);
        k_order<8><<<blocks, 64>>>(src, sink, bad, 4000, span); hipDeviceSynchronize(); hipMemcpy(&b8, bad, 4, hipMemcpyDeviceToHost);
        hipMemcpy(bad, &z, 4, hipMemcpyHostToDevice);
        k_order<0><<<blocks, 64>>>(src, sink, bad, 4000, span); hipDeviceSynchronize(); hipMemcpy(&b0, bad, 4, hipMemcpyDeviceToHost);
        printf("blocks %4d x 4000 iters x 64 lanes: load then 8 stores: wrong values with vmcnt(8): %d, with vmcnt(0): %d\n", blocks, b8, b0);
    }
    return 0;
}
