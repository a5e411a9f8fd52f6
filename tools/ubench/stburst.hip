// Micro-benchmark: how long does ONE CU need to get a 256 KiB output tile (new lines, 32 rows x 128 B per wave-pass at
// a 4 KiB row stride, 16-byte stores -- the dense kernel's epilogue pattern) out of its vector-store path?
//   stburst <workgroups> <mode> <gap_cycles>     mode 0 plain, 1 nt, 2 sc1, 3 sc0 sc1
// Each workgroup (4 waves, 160 KiB LDS = one per CU) writes `tiles` tiles; per tile: issue 64 stores per wave, then
// s_waitcnt vmcnt(0) (the drain the next tile's first counted wait implies), then idles `gap` cycles (the main loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__device__ __forceinline__ void st16(char* p, f32x4 v) {
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void stburst(char* out, int ntiles_total, int tiles, int gap, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) char smem[160 * 1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t yrowb = 4096;
    f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    unsigned long long t_issue = 0, t_drain = 0;
    for (int t = 0; t < tiles; ++t) {
        const int tile = (blockIdx.x + t * gridDim.x) % ntiles_total;   // (panel, column tile) as in the dense kernel
        const int mt = tile >> 2, nt = tile & 3;
        char* base = out + (size_t)(mt * 256 + (w >> 1) * 128) * yrowb + (size_t)(nt * 256 + (w & 1) * 128) * 4;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int pass = 0; pass < 16; ++pass) {
            char* line0 = base + (size_t)((pass & 3) * 32) * yrowb + (size_t)((pass >> 2) * 32) * 4;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) st16<MODE>(line0 + (size_t)(qq * 8 + (lane >> 3)) * yrowb + (lane & 7) * 16, v);
            // the epilogue's arithmetic between two passes (~500 cycles)
            __builtin_amdgcn_s_sleep(4);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t_issue += t1 - t0;
        t_drain += t2 - t1;
        for (int g = 0; g < gap; g += 64 * 16) __builtin_amdgcn_s_sleep(16);
    }
    if (lane == 0) {
        cyc[(blockIdx.x * 4 + w) * 2] = t_issue / tiles;
        cyc[(blockIdx.x * 4 + w) * 2 + 1] = t_drain / tiles;
    }
    if (v[0] == -1.f) smem[threadIdx.x] = 1;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const int gap = argc > 3 ? atoi(argv[3]) : 100000;
    const int tiles = 8;
    const int ntiles_total = 1024;                      // 65536 rows x 1024 columns
    const size_t bytes = (size_t)65536 * 4096;
    char* out; CHECK(hipMalloc((void**)&out, bytes));
    unsigned long long* cyc; CHECK(hipMalloc((void**)&cyc, wgs * 8 * sizeof(unsigned long long)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        switch (mode) {
            case 0: hipLaunchKernelGGL(stburst<0>, dim3(wgs), dim3(256), 0, 0, out, ntiles_total, tiles, gap, cyc); break;
            case 1: hipLaunchKernelGGL(stburst<1>, dim3(wgs), dim3(256), 0, 0, out, ntiles_total, tiles, gap, cyc); break;
            case 2: hipLaunchKernelGGL(stburst<2>, dim3(wgs), dim3(256), 0, 0, out, ntiles_total, tiles, gap, cyc); break;
            default: hipLaunchKernelGGL(stburst<3>, dim3(wgs), dim3(256), 0, 0, out, ntiles_total, tiles, gap, cyc); break;
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(wgs * 8);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double si = 0, sd = 0; unsigned long long mx = 0;
    for (int i = 0; i < wgs * 4; ++i) { si += h[2 * i]; sd += h[2 * i + 1]; mx = std::max(mx, h[2 * i] + h[2 * i + 1]); }
    printf("wgs %d mode %d gap %d: per tile (64 KiB per wave): issue %.0f cycles, drain after issue %.0f cycles, worst wave %llu; "
           "kernel %.3f ms for %d tiles per WG -> %.2f TB/s while storing\n", wgs, mode, gap, si / (wgs * 4), sd / (wgs * 4), mx, ms, tiles,
           (double)wgs * 262144 * tiles / (ms * 1e-3) / 1e12);
    return 0;
}
