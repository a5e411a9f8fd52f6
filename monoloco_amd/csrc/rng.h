// rng.h -- counter-based uniform numbers for dropout masks and Laplace sampling (stateless: a value depends
// only on (seed, row/site, column), so forward and backward regenerate identical masks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mlk {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {  // lowbias32 finaliser
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float u01(uint32_t a, uint32_t b, uint32_t c) {  // uniform in (0,1)
    const uint32_t r = mix32(a * 0x9e3779b9U + mix32(b + 0x85ebca6bU + mix32(c ^ 0xc2b2ae35U)));
    return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

}  // namespace mlk
