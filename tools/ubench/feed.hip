// Micro-benchmark: how fast can ONE CU pull L2-resident data into LDS?
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction)
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: global_load_dwordx4 -> VGPR only (sink into an xor so the loads are not dead)
//   mode 3 (round 6): waves 0-3 as mode 0, waves 4-7 as mode 1 at the same time -- are the two paths separate resources?
//   mode 4: waves 0-5 as mode 0, waves 6-7 as mode 1 (the byte split of a 128 x 64 tile: W rows by DMA, X rows through registers)
// grid = #CUs workgroups of 512 threads (8 waves), 128 KiB of LDS so that one workgroup sits on a CU.
// Every workgroup walks a private 1 MiB window of a 256 MiB buffer?  No: to stay L2 resident all workgroups
// of an XCD walk the SAME 2 MiB region (like the weight/X panel re-use of the dense kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(512) void feed(const char* __restrict__ buf, size_t region, int iters, int per_iter, float* sink,
                                             unsigned long long* cycles) {
    __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const char* base = buf + (size_t)(blockIdx.x & 7) * region;  // one region per XCD
    f32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    size_t off = (size_t)w * 1024 * per_iter + (size_t)(blockIdx.x >> 3) * 65536;
    for (int it = 0; it < iters; ++it) {
        char* dst = smem + ((it & 1) * 8 + w) * (1024 * per_iter > 8192 ? 8192 : 1024 * per_iter);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j >= per_iter) break;
            const size_t o = (off + (size_t)j * 1024) % region;
            const bool dma = MODE == 0 || (MODE == 3 && w < 4) || (MODE == 4 && w < 6);
            if (dma) {
                glds16(base + o + lane * 16, dst + j * 1024);
            } else {
                const f32x4 v = *(const f32x4*)(base + o + lane * 16);
                if (MODE == 1 || MODE >= 3) *(f32x4*)(dst + j * 1024 + lane * 16) = v;
                else acc += v;
            }
        }
        off += (size_t)8 * 1024 * per_iter;
        if (MODE == 0 || (MODE == 3 && w < 4) || (MODE == 4 && w < 6)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (MODE != 0) {
        f32x4 s = *(f32x4*)(smem + threadIdx.x * 16);
        acc += s;
    }
    if (acc[0] == 1234.5f) sink[threadIdx.x] = acc[1];
}

int main(int argc, char** argv) {
    int cus = 256;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0)); cus = prop.multiProcessorCount;
    const size_t region = 2u << 20;
    char* buf; CHECK(hipMalloc(&buf, region * 8)); CHECK(hipMemset(buf, 0, region * 8));
    float* sink; CHECK(hipMalloc(&sink, 4096));
    unsigned long long* cyc; CHECK(hipMalloc(&cyc, cus * 8));
    const int iters = 2000;
    for (int per_iter : {4, 8}) for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(feed<0>, dim3(cus), dim3(512), 0, 0, buf, region, iters, per_iter, sink, cyc);
            if (mode == 1) hipLaunchKernelGGL(feed<1>, dim3(cus), dim3(512), 0, 0, buf, region, iters, per_iter, sink, cyc);
            if (mode == 2) hipLaunchKernelGGL(feed<2>, dim3(cus), dim3(512), 0, 0, buf, region, iters, per_iter, sink, cyc);
            if (mode == 3) hipLaunchKernelGGL(feed<3>, dim3(cus), dim3(512), 0, 0, buf, region, iters, per_iter, sink, cyc);
            if (mode == 4) hipLaunchKernelGGL(feed<4>, dim3(cus), dim3(512), 0, 0, buf, region, iters, per_iter, sink, cyc);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        }
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(cus); CHECK(hipMemcpy(h.data(), cyc, cus * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (auto c : h) avg += c; avg /= cus;
        const double bytes_cu = (double)iters * per_iter * 8 * 1024;
        printf("mode %d per_iter %d: %.3f ms, %.1f B/clk/CU (s_memtime), %.2f TB/s chip\n", mode, per_iter, ms, bytes_cu / avg,
               bytes_cu * cus / (ms * 1e-3) / 1e12);
    }
    return 0;
}
