// mc_kernels.h -- MC-dropout epistemic uncertainty (reference monoloco/network/net.py:135-161,
// process.py:101-122): n_dropout stochastic forwards with dropout active ONLY at the model's top-level
// sites (after relu(bn1) and after relu(bn3), architectures.py:53,66 -- net.py:141 re-enables
// `model.dropout` alone), per pass 100 samples of Laplace(mu = d, b = |exp(s) d|) drawn with the SAME
// seed every pass (process.py:103), standard deviation over all passes x samples.
// The RNG is a counter-based hash, so parity with the reference's torch stream is statistical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dense_kernel.h"
#include "rng.h"

namespace mlk {

// in-place inverted dropout on a line-format activation (m_pad x n): v = keep ? v/(1-p) : 0, one thread
// per 8 values (a hi chunk and its lo chunk)
__global__ __launch_bounds__(256) void dropout_lines_kernel(char* __restrict__ act, int64_t rows, int n, float p,
                                                           uint32_t seed, uint32_t site) {
    const int gpr = n / 8;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= rows * gpr) return;
    const int64_t row = id / gpr;
    const int g = (int)(id - row * gpr);
    const int b = g >> 2, sub = g & 3;
    char* q = act + row * (int64_t)n * 4 + b * LINE + sub * 16;
    half8 hi = *(const half8*)q;
    half8 lo = *(const half8*)(q + 64);
    const float inv = 1.0f / (1.0f - p);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t col = (uint32_t)(b * 32 + sub * 8 + e);
        const bool keep = u01(seed, (uint32_t)row * 4099u + site, col) >= p;
        const float v = keep ? ((float)hi[e] + (float)lo[e]) * inv : 0.0f;
        _Float16 a, c;
        split_f16(v, a, c);
        hi[e] = a;
        lo[e] = c;
    }
    *(half8*)q = hi;
    *(half8*)(q + 64) = lo;
}

// one pass: per person accumulate sum and sum of squares of  mu + |b| * L_i,  i < n_samples, where L_i are
// standard Laplace draws that depend on (seed, person, i) only -- identical for every pass, like the
// reference's re-seeding.  Inverse CDF: L = -sign(u-1/2) * ln(1 - 2|u-1/2|).
__global__ __launch_bounds__(256) void mc_accumulate_kernel(const float* __restrict__ raw, int out_f, int col_d, int64_t m,
                                                           int n_samples, uint32_t seed, double* __restrict__ sum,
                                                           double* __restrict__ sumsq) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float mu = raw[i * out_f + col_d];
    const float b = fabsf(expf(raw[i * out_f + col_d + 1]) * mu);
    double s = 0.0, s2 = 0.0;
    for (int k = 0; k < n_samples; ++k) {
        const float u = u01(seed, (uint32_t)i, (uint32_t)k) - 0.5f;
        const float l = -copysignf(logf(1.0f - 2.0f * fabsf(u)), u);
        const double x = (double)(mu + b * l);
        s += x;
        s2 += x * x;
    }
    sum[i] += s;
    sumsq[i] += s2;
}

// laplace_sampling (reference process.py:101-122): n_samples draws of Laplace(mu, |b|) per person, laid out
// (n_samples, m) like laplace.sample((n_samples,)).  Same draws as mc_accumulate_kernel for the same seed.
__global__ __launch_bounds__(256) void laplace_sample_kernel(const float* __restrict__ mu_b, int64_t m, int n_samples,
                                                            uint32_t seed, float* __restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= m * (int64_t)n_samples) return;
    const int64_t k = id / m, i = id - k * m;
    const float mu = mu_b[i * 2], b = fabsf(mu_b[i * 2 + 1]);
    const float u = u01(seed, (uint32_t)i, (uint32_t)k) - 0.5f;
    const float l = -copysignf(logf(1.0f - 2.0f * fabsf(u)), u);
    out[id] = mu + b * l;
}

// unbiased standard deviation over n_total = passes * n_samples draws (torch.std default)
__global__ __launch_bounds__(256) void mc_finish_kernel(const double* __restrict__ sum, const double* __restrict__ sumsq,
                                                       int64_t m, double n_total, float* __restrict__ epi) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const double mean = sum[i] / n_total;
    double var = (sumsq[i] - n_total * mean * mean) / (n_total - 1.0);
    if (var < 0.0) var = 0.0;
    epi[i] = (float)sqrt(var);
}

}  // namespace mlk
