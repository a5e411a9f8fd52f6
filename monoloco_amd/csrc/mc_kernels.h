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
// Several stochastic passes are batched along the rows: network row R = pass * m_per + person; pass k of a call
// uses seed + k and the person index, so the masks do not depend on how the passes are grouped into launches.
__global__ __launch_bounds__(256) void dropout_lines_kernel(char* __restrict__ act, int64_t rows, int n, float p,
                                                           uint32_t seed0, uint32_t site, int64_t m_per) {
    const int gpr = n / 8;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= rows * gpr) return;
    const int64_t grow = id / gpr;
    const int64_t pass = grow / m_per;
    const int64_t row = grow - pass * m_per;      // person
    const uint32_t seed = seed0 + (uint32_t)pass;
    const int g = (int)(id - grow * gpr);
    const int b = g >> 2, sub = g & 3;
    char* q = act + grow * (int64_t)n * 4 + b * LINE + sub * 16;
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
// raw holds n_pass batched passes, row = pass * m + person.  One thread per (pass, person) draws the samples of that
// pass and leaves (sum, sum of squares) in part[row]; mc_reduce_kernel adds the passes of a person in pass order
// (deterministic, unlike atomics).
__global__ __launch_bounds__(256) void mc_accumulate_kernel(const float* __restrict__ raw, int out_f, int col_d, int64_t m,
                                                           int n_pass, int n_samples, uint32_t seed,
                                                           double* __restrict__ part) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= m * n_pass) return;
    const int64_t i = row % m;
    const float* r = raw + row * out_f;
    const float mu = r[col_d];
    const float b = fabsf(expf(r[col_d + 1]) * mu);
    double s = 0.0, s2 = 0.0;
    for (int k = 0; k < n_samples; ++k) {
        const float u = u01(seed, (uint32_t)i, (uint32_t)k) - 0.5f;
        const float l = -copysignf(logf(1.0f - 2.0f * fabsf(u)), u);
        const double x = (double)(mu + b * l);
        s += x;
        s2 += x * x;
    }
    part[row * 2] = s;
    part[row * 2 + 1] = s2;
}

__global__ __launch_bounds__(256) void mc_reduce_kernel(const double* __restrict__ part, int64_t m, int n_pass,
                                                       double* __restrict__ sum, double* __restrict__ sumsq) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    double s = sum[i], s2 = sumsq[i];
    for (int ps = 0; ps < n_pass; ++ps) {
        s += part[((int64_t)ps * m + i) * 2];
        s2 += part[((int64_t)ps * m + i) * 2 + 1];
    }
    sum[i] = s;
    sumsq[i] = s2;
}

// copies of the first `bytes` bytes of a buffer behind themselves: buf[c * bytes + o] = buf[o], c = 1 .. copies-1
__global__ __launch_bounds__(256) void replicate_kernel(char* __restrict__ buf, int64_t bytes, int copies) {
    const int64_t chunks = bytes / 16;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= chunks * (copies - 1)) return;
    const int64_t c = id / chunks + 1, o = id - (c - 1) * chunks;
    *(f32x4*)(buf + c * bytes + o * 16) = *(const f32x4*)(buf + o * 16);
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
