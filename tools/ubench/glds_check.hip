// global_load_lds_dwordx4 on gfx950 from inline asm: does lane l's 16 bytes land at M0 base + 16 l (lane-linear image)?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/glds_check tools/ubench/glds_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float4 *g, float4 *out) {
    __shared__ __attribute__((aligned(16))) float4 buf[256];
    const float4 *p = g + blockIdx.x * 256 + threadIdx.x;
    unsigned base = (unsigned)(uintptr_t)buf + (threadIdx.x >> 6) * 1024;   // wave-uniform LDS byte address
    base = __builtin_amdgcn_readfirstlane(base);
    unsigned save;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "global_load_lds_dwordx4 %2, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(save) : "s"(base), "v"(p) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x];
}
int main() {
    const int n = 256 * 64;
    std::vector<float> h(n * 4), r(n * 4);
    for (int i = 0; i < n * 4; ++i) h[i] = (float)i;
    float4 *g, *o;
    hipMalloc(&g, n * 16); hipMalloc(&o, n * 16);
    hipMemcpy(g, h.data(), n * 16, hipMemcpyHostToDevice);
    k<<<64, 256>>>(g, o);
    hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n * 4; ++i) bad += r[i] != h[i];
    printf("glds dwordx4 lane-linear copy: %d mismatches of %d\n", bad, n * 4);
    return bad != 0;
}
