// What does the memory system give the FILTERBANK's traffic shape, and does the shape matter?  No arithmetic: a
// 256-thread workgroup walks a run of R hops; per group of 4 hops it reads 4 half-blocks (4 x 4 KiB, 8-byte loads per
// lane like the register-prefetch kernel, or 16-byte) and writes 4 output rows (4 x 8 KiB, 16-byte stores, 512 B or
// 1 KiB contiguous per wave instruction).  Variants: run length (powers of two put every workgroup at the same offset
// inside its 2 MiB region at the same time -- channel camping if the HBM interleave uses those bits), a per-workgroup
// rotation of the walk's start, linear vs XCD-interleaved run -> workgroup mapping.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/stream_shapes tools/ubench/stream_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Args {
    const float2 *in;     // [hops * 512 + 512]
    float4 *out;          // [hops][512] float4 (= 1024 complex64)
    long long hops;
    int R;                // hops per run (multiple of 4)
    int rotate;           // 1: workgroup w starts its walk at group (w * 7) % groups_per_run
    int xcd;              // 1: run = (b % 8) * runs_per_xcd + b / 8
    int wide;             // 1: a wave's store instruction covers 1 KiB contiguous
    int halo;             // 1: every group also re-reads the 9 half-blocks before it (what a workgroup without a carried
                          //    register window would do: those are L2 hits when neighbouring groups run on the same XCD)
};

__global__ __launch_bounds__(256) void walk_kernel(Args a) {
    extern __shared__ float occ_lds[];      // dynamic LDS only limits the resident workgroups per CU
    if (a.hops < 0) occ_lds[threadIdx.x] = 0.f;
    const long long runs = (a.hops + a.R - 1) / a.R;
    long long run = blockIdx.x;
    if (a.xcd) {
        const long long per = (runs + 7) / 8;
        run = (long long)(blockIdx.x % 8) * per + blockIdx.x / 8;
    }
    if (run >= runs) return;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int groups = a.R / 4;
    const int g0 = a.rotate ? (int)((run * 7) % groups) : 0;
    float2 pre[8];
    auto load_group = [&](int g) {
        const long long h = run * a.R + 4LL * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long hh = h + j < a.hops ? h + j : a.hops - 1;
            pre[2 * j] = a.in[hh * 512 + t];
            pre[2 * j + 1] = a.in[hh * 512 + 256 + t];
        }
        if (a.halo) {
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int j = 1; j <= 9; ++j) {
                const long long hh = h - j >= 0 ? h - j : 0;
                const float2 u = a.in[hh * 512 + t], v = a.in[hh * 512 + 256 + t];
                acc.x += u.x + v.x; acc.y += u.y + v.y;
            }
            pre[0].x += acc.x; pre[0].y += acc.y;
        }
    };
    load_group(g0);
    for (int i = 0; i < groups; ++i) {
        const int g = (g0 + i) % groups;
        float2 cur[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) cur[j] = pre[j];
        if (i + 1 < groups) load_group((g0 + i + 1) % groups);
        // wave w stores hop 4 g + w: 8 KiB = 8 store instructions of 1 KiB (16 B per lane)
        const long long h = run * a.R + 4LL * g + wave;
        if (h < a.hops) {
            float4 *row = a.out + h * 512;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float4 v = make_float4(cur[s].x, cur[s].y, cur[(s + 1) & 7].x, cur[(s + 1) & 7].y);
                const int idx = a.wide ? s * 64 + lane : (s >> 1) * 128 + (lane >> 5) * 64 + (s & 1) * 32 + (lane & 31);
                row[idx] = v;
            }
        }
    }
}

int main(int argc, char **argv) {
    const long long n = 1LL << 28;
    const long long hops = n / 512 - 1;
    float2 *in;
    float4 *out;
    CK(hipMalloc(&in, (n + 1024) * sizeof(float2)));
    CK(hipMalloc(&out, (size_t)hops * 512 * sizeof(float4)));
    CK(hipMemset(in, 0, (n + 1024) * sizeof(float2)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(walk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    auto run = [&](int xcd, int occ, int halo, int R) {
        Args a{in, out, hops, R, 0, xcd, 0, halo};
        const long long runs = (hops + R - 1) / R;
        const unsigned grid = (unsigned)(((runs + 7) / 8) * 8);
        const size_t lds = occ >= 8 ? 0 : (size_t)(160 * 1024 / occ) - 1024;   // dynamic LDS only limits residency
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(walk_kernel, dim3(grid), dim3(256), lds, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("mapping=%s resident WGs/CU=%d halo re-reads=%d R=%4d hops  %.4f ms  %.2f TB/s (%.3f of 8)\n",
               xcd ? "xcd-contiguous" : "linear", occ, halo, R, best, 24.0 * (double)n / best * 1e-9, 24.0 * (double)n / best * 1e-9 / 8.0);
        fflush(stdout);
    };
    printf("# 1. run length x mapping, 8 resident workgroups per CU\n");
    for (int xcd = 0; xcd < 2; ++xcd)
        for (int R : {4, 8, 16, 32, 64, 128, 256, 1024}) run(xcd, 8, 0, R);
    printf("# 2. the 9-block halo re-read per group (a workgroup without a carried window), 8 resident per CU\n");
    for (int xcd = 0; xcd < 2; ++xcd)
        for (int R : {4, 16, 64, 256}) run(xcd, 8, 1, R);
    printf("# 3. resident workgroups per CU\n");
    for (int xcd = 0; xcd < 2; ++xcd)
        for (int occ : {1, 2, 4, 8})
            for (int R : {4, 16, 64, 256}) run(xcd, occ, 0, R);
    return 0;
}
