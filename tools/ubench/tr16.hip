// semantics probe of ds_read_b64_tr_b16 (gfx950): every lane reads 4 consecutive u16 at its own address (lane-linear image,
// s[i] = i); prints what each lane receives.  hipcc --offload-arch=gfx950 tr16.hip -o bin/tr16 && bin/tr16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(unsigned short* out, int stride_elems) {
    __shared__ __attribute__((aligned(16))) unsigned short s[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) s[i] = (unsigned short)i;
    __syncthreads();
    auto p = (__attribute__((address_space(3))) fp16x4*)(s + threadIdx.x * stride_elems);
    fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
    const unsigned short* u = (const unsigned short*)&v;
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = u[j];
}
int main() {
    unsigned short* d;
    hipMalloc((void**)&d, 64 * 4 * 2);
    for (int stride : {4, 8}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        unsigned short h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("lane address stride %d elements:\n", stride);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "   ");
    }
    return 0;
}
