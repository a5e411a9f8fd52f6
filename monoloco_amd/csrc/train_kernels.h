// train_kernels.h -- device kernels of the training step (reference monoloco/train/trainer.py:150-161,
// losses.py:46-142, network/architectures.py:48-102 in train mode).  All tensors are fp32, reductions accumulate in
// fp64.  GEMMs: sgemm_kernel (exact-fp32 MFMA, v_mfma_f32_32x32x2_f32 == an fmaf chain) for every shape; for large batches
// the hidden x hidden products run on the inference path's 3-product fp16 MFMA kernel instead (dense_kernel_w4.h) and this
// file supplies what feeds it: device-side weight packing, line / transposed-line writers, and the narrow-layer kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rng.h"

namespace mlt {
// The (m, n) matrices of a large-batch step are read once per pass and not again before ~2 GB of other traffic has gone by: their loads are
// NON-TEMPORAL (round 6; -DML_NT=0 builds the A/B reference: 65536-row step 11.66 -> 11.43 ms with them; non-temporal STORES of the lines
// measured nothing, profiles/r06_ablation.md section 7).  ML_LDSV: the same for any vector type.
#ifndef ML_NT
#define ML_NT 1
#endif
#if ML_NT & 1
#define ML_LDS4(p) __builtin_nontemporal_load((const f32x4*)(p))
#define ML_LDSV(T, p) __builtin_nontemporal_load((const T*)(p))
#else
#define ML_LDS4(p) (*(const f32x4*)(p))
#define ML_LDSV(T, p) (*(const T*)(p))
#endif
#define ML_STU4(p, v) (*(u4*)(p) = (v))

typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// Generic fp32 GEMM on the matrix cores:  C[i][j] = sum_k A(i,k) * B(k,j) (+ bias[j]) (+ C if accumulate)
// with A(i,k) = a[i*sai + k*sak], B(k,j) = b[k*sbk + j*sbj]: covers x.W^T (forward), dz.W (data gradient)
// and dz^T.x (weight gradient) without materialising transposes.  Tile 128x128x16, 4 waves (2x2), each
// 64x64 = 2x2 MFMA 32x32x2 tiles; operands staged in LDS k-major so the fragment reads are contiguous.
// Any M, N, K (zero fill at the edges).
struct GemmParams {
    const float* a;
    const float* b;
    const float* bias;  // [N] or nullptr
    float* c;           // [M][ldc]  (split-K: partial z goes to c + z * M * ldc, bias/accumulate ignored)
    int M, N, K;
    long sai, sak, sbk, sbj;
    int ldc;
    int accumulate;     // C += instead of C =
    int kchunk;         // K range per blockIdx.z (multiple of 16); K if not split
};

constexpr int GBM = 128, GBN = 128, GBK = 16, GLD = 132;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one operand tile (128 "mn" x 16 k) from global memory into registers (8 floats per thread), fast path:
// the tile is fully inside the matrix and the contiguous dimension has 16-byte aligned rows
struct TileRegs {
    f32x4 v[2];
};
// (mn_last: the k-contiguous layout may run over the edge of the matrix in mn -- the row index is clamped, the
// surplus rows are computed and never stored)
__device__ __forceinline__ TileRegs load_tile_fast(const float* base, long s_mn, long s_k, int mn0, int k0, int tid,
                                                   int mn_last) {
    TileRegs r;
    if (s_mn == 1) {  // mn contiguous: thread -> (k = tid/16, mn = 8*(tid%16) .. +7)
        const float* p = base + (long)(k0 + (tid >> 4)) * s_k + mn0 + (tid & 15) * 8;
        r.v[0] = *(const f32x4*)p;
        r.v[1] = *(const f32x4*)(p + 4);
    } else {          // k contiguous: thread -> (mn = tid/2, k = 8*(tid%2) .. +7)
        const int mn = mn0 + (tid >> 1);
        const float* p = base + (long)(mn < mn_last ? mn : mn_last) * s_mn + k0 + (tid & 1) * 8;
        r.v[0] = *(const f32x4*)p;
        r.v[1] = *(const f32x4*)(p + 4);
    }
    return r;
}
__device__ __forceinline__ void store_tile_fast(float* lds, const TileRegs& r, bool mn_contig, int tid) {
    if (mn_contig) {
        float* q = lds + (tid >> 4) * GLD + (tid & 15) * 8;
        *(f32x4*)q = r.v[0];
        *(f32x4*)(q + 4) = r.v[1];
    } else {
        const int mn = tid >> 1, kb = (tid & 1) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lds[(kb + e) * GLD + mn] = r.v[0][e];
            lds[(kb + 4 + e) * GLD + mn] = r.v[1][e];
        }
    }
}

__global__ __launch_bounds__(256) void sgemm_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) float As[GBK * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[GBK * GLD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w >> 1, wj = w & 1;
    const int i0 = blockIdx.y * GBM, j0 = blockIdx.x * GBN;
    const int kbeg = blockIdx.z * p.kchunk;
    const int kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    const bool a_mn = (p.sai == 1), b_mn = (p.sbj == 1);
    // fast path: one unit stride per operand, 16-byte aligned rows; an operand whose CONTIGUOUS dimension is mn
    // must have its tile fully inside the matrix, a k-contiguous one may hang over the edge (rows clamped).
    // Whole 16-wide k tiles go through the fast loads, a ragged k tail through the guarded element loads.
    const bool fast = (a_mn ? (i0 + GBM <= p.M) : true) && (b_mn ? (j0 + GBN <= p.N) : true) && (p.sai == 1 || p.sak == 1) &&
                      (p.sbj == 1 || p.sbk == 1) && (((a_mn ? p.sak : p.sai) & 3) == 0) && (((b_mn ? p.sbk : p.sbj) & 3) == 0) &&
                      ((((size_t)p.a) & 15) == 0) && ((((size_t)p.b) & 15) == 0) && ((kbeg & 3) == 0);
    auto mma = [&]() {
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 2) {
            float av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                av[t] = As[(kk + (lane >> 5)) * GLD + wi * 64 + t * 32 + (lane & 31)];
                bv[t] = Bs[(kk + (lane >> 5)) * GLD + wj * 64 + t * 32 + (lane & 31)];
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ti], bv[tj], acc[ti][tj], 0, 0, 0);
        }
    };
    int k_done = kbeg;
    if (fast) {
        const int kfull = kbeg + (kend - kbeg) / GBK * GBK;
        if (kfull > kbeg) {
            TileRegs ra = load_tile_fast(p.a, p.sai, p.sak, i0, kbeg, tid, p.M - 1);
            TileRegs rb = load_tile_fast(p.b, p.sbj, p.sbk, j0, kbeg, tid, p.N - 1);
            for (int k0 = kbeg; k0 < kfull; k0 += GBK) {
                store_tile_fast(As, ra, a_mn, tid);
                store_tile_fast(Bs, rb, b_mn, tid);
                __syncthreads();
                if (k0 + GBK < kfull) {  // prefetch the next k tile while this one is multiplied
                    ra = load_tile_fast(p.a, p.sai, p.sak, i0, k0 + GBK, tid, p.M - 1);
                    rb = load_tile_fast(p.b, p.sbj, p.sbk, j0, k0 + GBK, tid, p.N - 1);
                }
                mma();
                __syncthreads();
            }
        }
        k_done = kfull;
    }
    for (int k0 = k_done; k0 < kend; k0 += GBK) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int ia, ka, jb, kb;
            const int id = e * 256 + tid;
            if (a_mn) { ia = id & 127; ka = id >> 7; } else { ka = id & 15; ia = id >> 4; }
            if (b_mn) { jb = id & 127; kb = id >> 7; } else { kb = id & 15; jb = id >> 4; }
            const int gi = i0 + ia, gka = k0 + ka, gj = j0 + jb, gkb = k0 + kb;
            As[ka * GLD + ia] = (gi < p.M && gka < kend) ? p.a[(long)gi * p.sai + (long)gka * p.sak] : 0.f;
            Bs[kb * GLD + jb] = (gj < p.N && gkb < kend) ? p.b[(long)gkb * p.sbk + (long)gj * p.sbj] : 0.f;
        }
        __syncthreads();
        mma();
        __syncthreads();
    }
    const bool split = gridDim.z > 1;
    float* cbase = p.c + (split ? (size_t)blockIdx.z * p.M * p.ldc : 0);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int j = j0 + wj * 64 + tj * 32 + (lane & 31);
            const float bj = (!split && p.bias && j < p.N) ? p.bias[j] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wi * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (i < p.M && j < p.N) {
                    float v = acc[ti][tj][r] + bj;
                    float* dst = cbase + (long)i * p.ldc + j;
                    if (!split && p.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
}

// deterministic split-K combine: c[i][j] = (accumulate ? c[i][j] : 0) + bias[j] + sum_z part[z][i][j] for j < N
// (partials and c share the row stride ldc; columns >= N of c are not touched)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, int M, int N, int ldc,
                                                           const float* __restrict__ bias, int accumulate,
                                                           float* __restrict__ c) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)M * N) return;
    const int64_t i = id / N;
    const int j = (int)(id - i * N);
    const int64_t o = i * ldc + j, plane = (int64_t)M * ldc;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += part[(int64_t)z * plane + o];
    if (bias) s += bias[j];
    if (accumulate) s += c[o];
    c[o] = s;
}

// ------------------------------------------------------------------------------------------------
// column statistics over the batch: for each column j of z (m x n): sum and sum of squares in fp64
// (two outputs), one workgroup per 64 columns, 4 row-groups per workgroup.
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ z, const float* __restrict__ w2,
                                                        int64_t m, int n, double* __restrict__ s1,
                                                        double* __restrict__ s2) {
    // s1[j] = sum_i z[i][j]; s2[j] = sum_i z[i][j] * (w2 ? w2[i][j] : z[i][j])
    // thread = (column group of 4 via one 16-byte load, row group): 16 x 16 per workgroup, 64 columns per workgroup
    __shared__ double r1[16][64], r2[16][64];
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int j0 = blockIdx.x * 64 + cg * 4;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    const int64_t step = (int64_t)gridDim.y * 16;
    if ((n & 3) == 0 && j0 + 3 < n) {
        for (int64_t i = (int64_t)blockIdx.y * 16 + rg; i < m; i += step) {
            const f32x4 v = ML_LDS4(z + i * n + j0);
            const f32x4 u = w2 ? ML_LDS4(w2 + i * n + j0) : v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] += (double)v[e];
                b[e] += (double)v[e] * (double)u[e];
            }
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.y * 16 + rg; i < m; i += step)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (j0 + e < n) {
                    const double v = (double)z[i * n + j0 + e];
                    a[e] += v;
                    b[e] += v * (w2 ? (double)w2[i * n + j0 + e] : v);
                }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r1[rg][cg * 4 + e] = a[e];
        r2[rg][cg * 4 + e] = b[e];
    }
    __syncthreads();
    const int cj = threadIdx.x;
    if (cj < 64 && blockIdx.x * 64 + cj < n) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            sa += r1[g][cj];
            sb += r2[g][cj];
        }
        atomicAdd(&s1[blockIdx.x * 64 + cj], sa);
        atomicAdd(&s2[blockIdx.x * 64 + cj], sb);
    }
}

// BatchNorm1d, training mode (reference architectures.py:51,91,96 with nn.BatchNorm1d defaults): batch mean
// and biased variance; running stats updated with momentum 0.1 (unbiased variance); then ReLU and inverted
// dropout.  Stores invstd and mean for the backward pass; y = dropout(relu(g * (z - mean) * invstd + b)).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ s1, const double* __restrict__ s2,
                                                         int64_t m, int n, float eps, float momentum,
                                                         float* __restrict__ mean, float* __restrict__ invstd,
                                                         float* __restrict__ run_mean, float* __restrict__ run_var) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const double mu = s1[j] / (double)m;
    double var = s2[j] / (double)m - mu * mu;
    if (var < 0) var = 0;
    mean[j] = (float)mu;
    invstd[j] = (float)(1.0 / sqrt(var + (double)eps));
    const double unb = m > 1 ? var * (double)m / (double)(m - 1) : var;
    run_mean[j] = (1.f - momentum) * run_mean[j] + momentum * (float)mu;
    run_var[j] = (1.f - momentum) * run_var[j] + momentum * (float)unb;
}

// bn_finalize_kernel on the per-128-row-block partial sums the forward GEMM's epilogue left (dense_kernel_w4<.., -2>, p.colpart:
// part[b][0][j] = sum, part[b][1][j] = sum of squares over the block's rows < m): added in block order (deterministic), then as above.
__global__ __launch_bounds__(256) void bn_finalize_parts_kernel(const double* __restrict__ part, int nblocks, int64_t m, int n, float eps,
                                                               float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                                               float* __restrict__ run_mean, float* __restrict__ run_var) {
    // workgroup = 8 columns x 32 groups (128 workgroups at H = 1024); group g adds the blocks b = g, g + 32, .. in order, then the 32
    // group sums are added in order: fixed order whatever the launch
    __shared__ double r1[32][8], r2[32][8];
    const int c = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int j = blockIdx.x * 8 + c;
    double s1 = 0.0, s2 = 0.0;
    if (j < n) {
#pragma unroll 4
        for (int b = g; b < nblocks; b += 32) {
            s1 += part[(size_t)b * 2 * n + j];
            s2 += part[(size_t)b * 2 * n + n + j];
        }
    }
    r1[g][c] = s1;
    r2[g][c] = s2;
    __syncthreads();
    if (g != 0 || j >= n) return;
    s1 = s2 = 0.0;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        s1 += r1[q][c];
        s2 += r2[q][c];
    }
    const double mu = s1 / (double)m;
    double var = s2 / (double)m - mu * mu;
    if (var < 0) var = 0;
    mean[j] = (float)mu;
    invstd[j] = (float)(1.0 / sqrt(var + (double)eps));
    const double unb = m > 1 ? var * (double)m / (double)(m - 1) : var;
    run_mean[j] = (1.f - momentum) * run_mean[j] + momentum * (float)mu;
    run_var[j] = (1.f - momentum) * run_var[j] + momentum * (float)unb;
}

__global__ __launch_bounds__(256) void bn_relu_drop_kernel(const float* __restrict__ z, int64_t m, int n,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float p_drop, uint32_t seed, uint32_t site,
                                                          const float* __restrict__ residual, float* __restrict__ y) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= m * n) return;
    const int j = (int)(id % n);
    const int64_t i = id / n;
    float v = gamma[j] * ((z[id] - mean[j]) * invstd[j]) + beta[j];
    v = v > 0.f ? v : 0.f;
    if (p_drop > 0.f) v = (mlk::u01(seed, (uint32_t)i * 4099u + site, (uint32_t)j) >= p_drop) ? v / (1.f - p_drop) : 0.f;
    if (residual) v += residual[id];
    y[id] = v;
}

// backward of dropout + ReLU (+ BN affine part): dy = dout * mask/(1-p) * (pre-dropout activation > 0), written
// in place; also needs xhat = (z - mean) * invstd for the BN reductions, computed on the fly by the consumers.
__global__ __launch_bounds__(256) void relu_drop_bwd_kernel(float* __restrict__ dout, const float* __restrict__ z,
                                                           int64_t m, int n, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float p_drop, uint32_t seed,
                                                           uint32_t site, float* __restrict__ xhat) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= m * n) return;
    const int j = (int)(id % n);
    const int64_t i = id / n;
    const float xh = (z[id] - mean[j]) * invstd[j];
    const float pre = gamma[j] * xh + beta[j];
    float g = pre > 0.f ? dout[id] : 0.f;
    if (p_drop > 0.f) g = (mlk::u01(seed, (uint32_t)i * 4099u + site, (uint32_t)j) >= p_drop) ? g / (1.f - p_drop) : 0.f;
    dout[id] = g;
    xhat[id] = xh;
}

// BN backward, elementwise part: dz = gamma*invstd/m * (m*dy - sum(dy) - xhat*sum(dy*xhat)); in place on dy
__global__ __launch_bounds__(256) void bn_bwd_kernel(float* __restrict__ dy, const float* __restrict__ xhat, int64_t m, int n,
                                                    const double* __restrict__ sdy, const double* __restrict__ sdyx,
                                                    const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= m * n) return;
    const int j = (int)(id % n);
    const float a = (float)sdy[j], b = (float)sdyx[j];
    const float g = gamma[j] * invstd[j] / (float)m;
    dy[id] = g * ((float)m * dy[id] - a - xhat[id] * b);
    if (id < n) {  // one thread per column publishes the parameter gradients
        dgamma[j] = (float)sdyx[j];
        dbeta[j] = (float)sdy[j];
    }
}

// ------------------------------------------------------------------------------------------------
// The backward element-wise chain of a block, fused (round 2).  Before: relu_drop_bwd (dy, xhat written), col_stats(dy,
// xhat), bn_bwd (dz written), col_stats(dz) = 7 reads + 3 writes of an (m, n) fp32 matrix.  Now two passes that recompute
// dy and xhat from (dout, z) on the fly: 4 reads + 1 write.  Same per-element arithmetic and the same thread -> (column
// group, row group) mapping and fp64 summation order as col_stats_kernel, so the results are bit-identical.
//   dy   = dout * [gamma * xhat + beta > 0] * dropout_mask / (1 - p),   xhat = (z - mean) * invstd
__device__ __forceinline__ void bwd_elem(float dout, float z, float mean, float invstd, float gamma, float beta, float p_drop,
                                         uint32_t seed, uint32_t site, int64_t i, int j, float& dy, float& xh) {
    xh = (z - mean) * invstd;
    const float pre = gamma * xh + beta;
    float g = pre > 0.f ? dout : 0.f;
    if (p_drop > 0.f) g = (mlk::u01(seed, (uint32_t)i * 4099u + site, (uint32_t)j) >= p_drop) ? g / (1.f - p_drop) : 0.f;
    dy = g;
}

// rows a thread of the two backward passes requests per trip of its row loop (-DBWD_ROWS=<n> for the A/B, tools/lib_ab_train.py)
#ifndef BWD_ROWS
#define BWD_ROWS 2
#endif
// pass 1: s1[j] += sum_i dy[i][j], s2[j] += sum_i dy[i][j] * xhat[i][j]   (nothing written but the sums); n % 4 == 0
// NHG (round 6; 0: dout is the (m, n) gradient): the block sits right under an output head -- its incoming gradient is dy = hs . hw (hs: the
// loss gradient of the head's NHG outputs, row stride ldh; hw: the head's weights (NHG, n)) and is formed HERE, with skinny_out_kernel's very
// fma chain (c = 0 .. NHG - 1 from zero: the same bits), instead of being written by that kernel and read back by both backward passes --
// three crossings of the (m, n) matrix per step.
template <int NHG>
__device__ __forceinline__ f32x4 head_grad(const float* __restrict__ hs, int ldh, const f32x4 (&hw4)[NHG > 0 ? NHG : 1], int64_t i) {
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    const float* sp = hs + i * ldh;
#pragma unroll
    for (int c = 0; c < NHG; ++c) {
        const float sv = sp[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = __builtin_fmaf(sv, hw4[c][e], d[e]);
    }
    return d;
}
template <int NHG = 0>
__global__ __launch_bounds__(256) void bwd_stats_kernel(const float* __restrict__ dout, const float* __restrict__ z, int64_t m,
                                                       int n, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float p_drop, uint32_t seed, uint32_t site, double* __restrict__ s1,
                                                       double* __restrict__ s2, float* __restrict__ colmax = nullptr,
                                                       const float* __restrict__ hs = nullptr, int ldh = 0,
                                                       const float* __restrict__ hw = nullptr) {
    __shared__ double r1[16][64], r2[16][64];
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int j0 = blockIdx.x * 64 + cg * 4;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    // colmax (2 n floats, zeroed; may be null): per column max |dy| and max |xhat| -- the bound bn_bwd_lines_kernel scales dz by
    float mdy[4] = {0.f, 0.f, 0.f, 0.f}, mxh[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t step = (int64_t)gridDim.y * 16;
    if (j0 + 3 < n) {
        const f32x4 mu = *(const f32x4*)(mean + j0), is = *(const f32x4*)(invstd + j0);
        const f32x4 ga = *(const f32x4*)(gamma + j0), be = *(const f32x4*)(beta + j0);
        auto one = [&](const f32x4& d, const f32x4& zz, int64_t i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float dy, xh;
                bwd_elem(d[e], zz[e], mu[e], is[e], ga[e], be[e], p_drop, seed, site, i, j0 + e, dy, xh);
                a[e] += (double)dy;
                b[e] += (double)dy * (double)xh;
                mdy[e] = __builtin_fmaxf(mdy[e], __builtin_fabsf(dy));
                mxh[e] = __builtin_fmaxf(mxh[e], __builtin_fabsf(xh));
            }
        };
        f32x4 hw4[NHG > 0 ? NHG : 1];
#pragma unroll
        for (int c = 0; c < NHG; ++c) hw4[c] = *(const f32x4*)(hw + (int64_t)c * n + j0);
        auto grad_in = [&](int64_t r) { return NHG > 0 ? head_grad<NHG>(hs, ldh, hw4, r) : ML_LDS4(dout + r * n + j0); };
        // BWD_ROWS rows per trip, all their loads requested before the first use (same rows in the same order: the same sums)
        int64_t i = (int64_t)blockIdx.y * 16 + rg;
        for (; i + (BWD_ROWS - 1) * step < m; i += BWD_ROWS * step) {
            f32x4 dd[BWD_ROWS], zv[BWD_ROWS];
#pragma unroll
            for (int u = 0; u < BWD_ROWS; ++u) {
                dd[u] = grad_in(i + u * step);
                zv[u] = ML_LDS4(z + (i + u * step) * n + j0);
            }
#pragma unroll
            for (int u = 0; u < BWD_ROWS; ++u) one(dd[u], zv[u], i + u * step);
        }
        for (; i < m; i += step) one(grad_in(i), *(const f32x4*)(z + i * n + j0), i);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r1[rg][cg * 4 + e] = a[e];
        r2[rg][cg * 4 + e] = b[e];
    }
    __syncthreads();
    const int cj = threadIdx.x;
    if (cj < 64 && blockIdx.x * 64 + cj < n) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            sa += r1[g][cj];
            sb += r2[g][cj];
        }
        atomicAdd(&s1[blockIdx.x * 64 + cj], sa);
        atomicAdd(&s2[blockIdx.x * 64 + cj], sb);
    }
    if (colmax) {   // (uniform) the column maxima: through LDS over the 16 row groups, then ONE atomic per column and workgroup
        // (an atomic per thread and column -- 4 M of them on 2 K addresses at 65536 x 1024 -- doubled the pass: 127 -> 243 us)
        __syncthreads();
        float* f1 = (float*)&r1[0][0];
        float* f2 = (float*)&r2[0][0];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f1[rg * 64 + cg * 4 + e] = mdy[e];
            f2[rg * 64 + cg * 4 + e] = mxh[e];
        }
        __syncthreads();
        if (cj < 64 && blockIdx.x * 64 + cj < n) {
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                v0 = __builtin_fmaxf(v0, f1[g * 64 + cj]);
                v1 = __builtin_fmaxf(v1, f2[g * 64 + cj]);
            }
            // non-negative floats order like their bit patterns; NaN / inf saturate
            v0 = v0 < 3.0e38f ? v0 : 3.0e38f;
            v1 = v1 < 3.0e38f ? v1 : 3.0e38f;
            if (v0 > 0.f) atomicMax((unsigned*)colmax + blockIdx.x * 64 + cj, __builtin_bit_cast(unsigned, v0));
            if (v1 > 0.f) atomicMax((unsigned*)colmax + n + blockIdx.x * 64 + cj, __builtin_bit_cast(unsigned, v1));
        }
    }
}

// pass 2: dz = gamma * invstd / m * (m * dy - sum(dy) - xhat * sum(dy * xhat)) read from din and written to dout (the same
// buffer, or another one: the residual stages keep the incoming gradient for the skip connection, which used to cost a 268 MB
// device copy per stage at 65536 rows), and the column sums of dz (the Linear bias gradient) accumulated into sdz on the way;
// also publishes dgamma / dbeta.
__global__ __launch_bounds__(256) void bn_bwd_fused_kernel(float* dout, const float* din, const float* __restrict__ z, int64_t m, int n,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float p_drop, uint32_t seed, uint32_t site,
                                                          const double* __restrict__ sdy, const double* __restrict__ sdyx,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          double* __restrict__ sdz, float* __restrict__ dzmax) {
    __shared__ double r1[16][64];
    float mx = 0.f;   // max |dz| of this thread (dzmax: the scale of the fast gradient GEMMs' operand, may be null)
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int j0 = blockIdx.x * 64 + cg * 4;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    const int64_t step = (int64_t)gridDim.y * 16;
    if (j0 + 3 < n) {
        const f32x4 mu = *(const f32x4*)(mean + j0), is = *(const f32x4*)(invstd + j0);
        const f32x4 ga = *(const f32x4*)(gamma + j0), be = *(const f32x4*)(beta + j0);
        float sa[4], sb[4], gg[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sa[e] = (float)sdy[j0 + e];
            sb[e] = (float)sdyx[j0 + e];
            gg[e] = ga[e] * is[e] / (float)m;
        }
        for (int64_t i = (int64_t)blockIdx.y * 16 + rg; i < m; i += step) {
            const f32x4 d = ML_LDS4(din + i * n + j0);
            const f32x4 zz = ML_LDS4(z + i * n + j0);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float dy, xh;
                bwd_elem(d[e], zz[e], mu[e], is[e], ga[e], be[e], p_drop, seed, site, i, j0 + e, dy, xh);
                o[e] = gg[e] * ((float)m * dy - sa[e] - xh * sb[e]);
                a[e] += (double)o[e];
            }
            *(f32x4*)(dout + i * n + j0) = o;
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(o[0]), __builtin_fabsf(o[1]))),
                                 __builtin_fmaxf(__builtin_fabsf(o[2]), __builtin_fabsf(o[3])));
        }
        if (blockIdx.y == 0 && rg == 0) {  // one thread per column publishes the parameter gradients
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dgamma[j0 + e] = (float)sdyx[j0 + e];
                dbeta[j0 + e] = (float)sdy[j0 + e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) r1[rg][cg * 4 + e] = a[e];
    __syncthreads();
    const int cj = threadIdx.x;
    if (cj < 64 && blockIdx.x * 64 + cj < n) {
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += r1[g][cj];
        atomicAdd(&sdz[blockIdx.x * 64 + cj], s);
    }
    if (dzmax) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((threadIdx.x & 63) == 0) {
            if (!(mx < 3.0e38f)) mx = 3.0e38f;
            atomicMax((unsigned*)dzmax, __builtin_bit_cast(unsigned, mx));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fast forward GEMMs of a training step (round 2): z = y . W^T + b on the inference path's 3-product fp16 MFMA kernel
// (fp32-class accuracy at ~3x the rate of the exact-fp32 MFMA).  Its operands are "k32 hi|lo lines"; the weights change
// every step, so they are scaled / split / packed on the device:
//   wmax_kernel:   max|W|;  e = floor(log2(2^14 / max|W|)) as the host packer, descale = 2^-e
//   wpack_kernel:  lines of W * 2^e (fp16 hi | lo per k32 block), bias * 2^e; <true>: of W^T (data gradient)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split_h(float v, _Float16& hi, _Float16& lo) {
    const float c = __builtin_fminf(__builtin_fmaxf(v, -65504.0f), 65504.0f);
    hi = (_Float16)c;
    lo = (_Float16)(c - (float)hi);
}

// Scale words of one Linear (8 floats, zeroed at the start of a step):
//   [0] 2^e_w  [1] 2^-e_w  [2] max|W|  [3] max|dz|  [4] 2^-(e_w + e_dz) (descale of dx = dz . W)  [5] 2^-e_dz (of dW = dz^T . x)
// wmax_kernel folds a max of absolute values into a word (non-negative floats order like their bit patterns); the pack /
// line kernels derive the exponent from it and one thread publishes the powers of two the GEMM epilogues read.
__global__ __launch_bounds__(256) void wmax_kernel(const float* __restrict__ w, int64_t numel, float* __restrict__ maxword) {
    __shared__ float red[256];
    float mx = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < numel; i += (int64_t)gridDim.x * 1024) {
        const f32x4 v = *(const f32x4*)(w + i);
        mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1]))),
                             __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (numel & 3)) mx = __builtin_fmaxf(mx, __builtin_fabsf(w[numel - 1 - threadIdx.x]));
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = __builtin_fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float r = red[0];
        if (!(r < 3.0e38f)) r = 3.0e38f;   // inf / NaN weights: the step is lost anyway, keep the exponent finite
        atomicMax((unsigned*)maxword, __builtin_bit_cast(unsigned, r));
    }
}

__device__ __forceinline__ int wscale_exp(float m) {
    int e = 0;
    if (m > 0.f && m < 3.0e38f) e = (int)floorf(log2f(16384.0f / m));   // max|W| * 2^e in [2^13, 2^15): far inside fp16
    return e > 40 ? 40 : (e < -40 ? -40 : e);
}

// one thread per (row, group of 8 k) of W (n x k, k % 8 == 0), kpad = k rounded up to 64 (zero filled).
// TRANS: the image of W^T (k rows of n values) for the data-gradient GEMM; its scale is the forward image's (same
// weights, same step) and there is no bias.
template <bool TRANS>
__device__ __forceinline__ void wpack_body(const float* __restrict__ w, const float* __restrict__ bias, int n, int k, int kpad,
                                           float* __restrict__ scale2, char* __restrict__ lines, float* __restrict__ bias_scaled,
                                           int64_t id) {
    const int e2 = wscale_exp(scale2[2]);
    const float sc = ldexpf(1.0f, e2);
    if (id == 0 && !TRANS) {
        scale2[0] = sc;
        scale2[1] = ldexpf(1.0f, -e2);
    }
    h8 hi, lo;
    int row, g, width;
    if (TRANS) {   // consecutive threads = consecutive rows of W^T (columns of W): coalesced reads
        if (id >= (int64_t)k * (n / 8)) return;
        g = (int)(id / k);
        row = (int)(id - (int64_t)g * k);
        width = n;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 a, b;
            split_h(w[(int64_t)(g * 8 + e) * k + row] * sc, a, b);
            hi[e] = a;
            lo[e] = b;
        }
    } else {
        const int gpr = kpad / 8;
        if (id >= (int64_t)n * gpr) return;
        row = (int)(id / gpr);
        g = (int)(id - (int64_t)row * gpr);
        width = kpad;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = g * 8 + e;
            _Float16 a, b;
            split_h(kk < k ? w[(int64_t)row * k + kk] * sc : 0.f, a, b);
            hi[e] = a;
            lo[e] = b;
        }
        if (g == 0) bias_scaled[row] = bias[row] * sc;
    }
    char* dst = lines + (int64_t)row * width * 4 + (g >> 2) * 128 + (g & 3) * 16;
    *(h8*)dst = hi;
    *(h8*)(dst + 64) = lo;
}

template <bool TRANS>
__global__ __launch_bounds__(256) void wpack_kernel(const float* __restrict__ w, const float* __restrict__ bias, int n, int k, int kpad,
                                                    float* __restrict__ scale2, char* __restrict__ lines,
                                                    float* __restrict__ bias_scaled) {
    wpack_body<TRANS>(w, bias, n, k, kpad, scale2, lines, bias_scaled, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// All H x H Linears of a step at once (round 4: 8 x (wmax + wpack + wpack<true>) = 24 launches of 6-14 us each became 2): the
// weights only change in the optimizer, so every image of a step -- W for the forward GEMM, W^T for the data-gradient GEMM -- can be
// packed before the forward starts.  blockIdx.y = the Linear; wpack_multi_kernel's blockIdx.z: 0 = W, 1 = W^T.
struct WLayer {
    const float* w;       // [H][H]
    const float* bias;    // [H]
    float* sc;            // the Linear's 8 scale words
    char* lines;          // image of W   (forward)
    char* linesT;         // image of W^T (data gradient)
    float* bias_scaled;   // bias * 2^e
};
__global__ __launch_bounds__(256) void wmax_multi_kernel(const WLayer* __restrict__ L, int64_t numel) {
    __shared__ float red[256];
    const float* w = L[blockIdx.y].w;
    float mx = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < numel; i += (int64_t)gridDim.x * 1024) {
        const f32x4 v = *(const f32x4*)(w + i);
        mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1]))),
                             __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
    }
    red[threadIdx.x] = mx;   // (numel = H x H with H % 256 == 0: no tail)
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = __builtin_fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float r = red[0];
        if (!(r < 3.0e38f)) r = 3.0e38f;
        atomicMax((unsigned*)(L[blockIdx.y].sc + 2), __builtin_bit_cast(unsigned, r));
    }
}
__global__ __launch_bounds__(256) void wpack_multi_kernel(const WLayer* __restrict__ L, int H) {
    const WLayer l = L[blockIdx.y];
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.z == 0) wpack_body<false>(l.w, l.bias, H, H, H, l.sc, l.lines, l.bias_scaled, id);
    else wpack_body<true>(l.w, nullptr, H, H, H, l.sc, l.linesT, nullptr, id);
}

// fp32 (m, n) -> lines (rows >= m of the padded buffer are left alone: a row of the GEMM only depends on its own input row)
__global__ __launch_bounds__(256) void act_lines_kernel(const float* __restrict__ y, int64_t m, int n, char* __restrict__ lines) {
    const int gpr = n / 8;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= m * gpr) return;
    const int64_t row = id / gpr;
    const int g = (int)(id - row * gpr);
    const f32x4 a = *(const f32x4*)(y + row * n + g * 8), b = *(const f32x4*)(y + row * n + g * 8 + 4);
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        _Float16 p, q;
        split_h(e < 4 ? a[e] : b[e - 4], p, q);
        hi[e] = p;
        lo[e] = q;
    }
    char* dst = lines + row * (int64_t)n * 4 + (g >> 2) * 128 + (g & 3) * 16;
    *(h8*)dst = hi;
    *(h8*)(dst + 64) = lo;
}

// a gradient dz (m, n) fp32 -> SCALED lines (the operand of dx = dz . W and, reduction-major, of dW = dz^T . x): scale 2^e from
// its measured max (sc[3]; gradients carry a 1/m factor and would sit in fp16's subnormal range otherwise); thread 0 publishes the
// two descale words (sc[4] for the data gradient: 2^-e x the weights' descale, sc[5] = 2^-e for the weight gradient).
__global__ __launch_bounds__(256) void grad_lines_kernel(const float* __restrict__ dz, int64_t m, int n, float* __restrict__ sc,
                                                        char* __restrict__ lines) {
    const int e2 = wscale_exp(sc[3]);
    const float scale = ldexpf(1.0f, e2);
    const int gpr = n / 8;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id == 0) {
        sc[4] = ldexpf(1.0f, -e2) * sc[1];
        sc[5] = ldexpf(1.0f, -e2);
    }
    if (id >= m * gpr) return;
    const int64_t row = id / gpr;
    const int g = (int)(id - row * gpr);
    const f32x4 a = *(const f32x4*)(dz + row * n + g * 8), b = *(const f32x4*)(dz + row * n + g * 8 + 4);
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        _Float16 p, q;
        split_h((e < 4 ? a[e] : b[e - 4]) * scale, p, q);
        hi[e] = p;
        lo[e] = q;
    }
    char* dst = lines + row * (int64_t)n * 4 + (g >> 2) * 128 + (g & 3) * 16;
    *(h8*)dst = hi;
    *(h8*)(dst + 64) = lo;
}

// bn_bwd_fused_kernel whose dz leaves as SCALED LINES only (the operand of both gradient GEMMs of the Linear below; no fp32 copy:
// nothing else on the large-batch route reads it): the per-layer dz -> lines pass (grad_lines_kernel, 537 MB of traffic at
// 65536 x 1024) disappears.  The scale 2^e must be known before the first element is written, so it comes from a BOUND instead
// of the measured maximum:  |dz_ij| <= |g_j| (m max_i |dy_ij| + |sum dy_j| + max_i |xhat_ij| |sum dy xhat_j|),  g = gamma invstd / m,
// with the column maxima bwd_stats_kernel collected (colmax).  The first term dominates (the sums are ~sqrt(m) terms), so the
// bound sits within a few per cent of the true maximum: the scaled values stay in [0, 2^15), fp16's top octaves.  Every workgroup
// evaluates the same bound over all n columns (L2-resident vectors): same scale everywhere, no extra launch; workgroup (0, 0)
// publishes the descales (sc[4] = 2^-e x the weights' descale for dx, sc[5] = 2^-e for dW).  Same per-element arithmetic as
// bn_bwd_fused_kernel (the fp32 dz is formed, then scaled by a power of two and split).
template <int NHG = 0>
__global__ __launch_bounds__(256) void bn_bwd_lines_kernel(const float* din, const float* __restrict__ z, int64_t m, int n,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float p_drop, uint32_t seed, uint32_t site,
                                                          const double* __restrict__ sdy, const double* __restrict__ sdyx,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          double* __restrict__ sdz, const float* __restrict__ colmax,
                                                          float* __restrict__ sc, char* __restrict__ lines,
                                                          const float* __restrict__ hs = nullptr, int ldh = 0,
                                                          const float* __restrict__ hw = nullptr) {
    __shared__ double r1[16][64];
    __shared__ float wmax[4];
    float bound = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float g = __builtin_fabsf(gamma[j] * invstd[j] / (float)m);
        const float b = g * ((float)m * colmax[j] + __builtin_fabsf((float)sdy[j]) + colmax[n + j] * __builtin_fabsf((float)sdyx[j]));
        bound = __builtin_fmaxf(bound, b < 3.0e38f ? b : 3.0e38f);   // (NaN -> 3e38: scale 2^-113.., finite garbage instead of NaN lines)
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bound = __builtin_fmaxf(bound, __shfl_xor(bound, o, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = bound;
    __syncthreads();
    bound = __builtin_fmaxf(__builtin_fmaxf(wmax[0], wmax[1]), __builtin_fmaxf(wmax[2], wmax[3]));
    const int e2 = wscale_exp(bound);
    const float scale = ldexpf(1.0f, e2);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        sc[3] = bound;
        sc[4] = ldexpf(1.0f, -e2) * sc[1];
        sc[5] = ldexpf(1.0f, -e2);
    }
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int j0 = blockIdx.x * 64 + cg * 4;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    const int64_t step = (int64_t)gridDim.y * 16;
    if (j0 + 3 < n) {
        const f32x4 mu = *(const f32x4*)(mean + j0), is = *(const f32x4*)(invstd + j0);
        const f32x4 ga = *(const f32x4*)(gamma + j0), be = *(const f32x4*)(beta + j0);
        float sa[4], sb[4], gg[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sa[e] = (float)sdy[j0 + e];
            sb[e] = (float)sdyx[j0 + e];
            gg[e] = ga[e] * is[e] / (float)m;
        }
        const bool odd = cg & 1;
        const int g8 = j0 >> 3;   // 8-column group of the row: the even lane stores its hi halves, the odd lane its lo halves
        auto one = [&](const f32x4& d, const f32x4& zz, int64_t i) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float dy, xh;
                bwd_elem(d[e], zz[e], mu[e], is[e], ga[e], be[e], p_drop, seed, site, i, j0 + e, dy, xh);
                o[e] = gg[e] * ((float)m * dy - sa[e] - xh * sb[e]);
                a[e] += (double)o[e];
            }
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            unsigned hi2[2], lo2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                _Float16 a0, b0, a1, b1;
                split_h(o[2 * e] * scale, a0, b0);
                split_h(o[2 * e + 1] * scale, a1, b1);
                hi2[e] = __builtin_bit_cast(unsigned, h2{a0, a1});
                lo2[e] = __builtin_bit_cast(unsigned, h2{b0, b1});
            }
            const unsigned s0 = __shfl_xor(odd ? hi2[0] : lo2[0], 1, 64), s1 = __shfl_xor(odd ? hi2[1] : lo2[1], 1, 64);
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const u4 out = odd ? u4{s0, s1, lo2[0], lo2[1]} : u4{hi2[0], hi2[1], s0, s1};
            ML_STU4(lines + i * (int64_t)n * 4 + (g8 >> 2) * 128 + (g8 & 3) * 16 + (odd ? 64 : 0), out);
        };
        f32x4 hw4[NHG > 0 ? NHG : 1];
#pragma unroll
        for (int c = 0; c < NHG; ++c) hw4[c] = *(const f32x4*)(hw + (int64_t)c * n + j0);
        auto grad_in = [&](int64_t r) { return NHG > 0 ? head_grad<NHG>(hs, ldh, hw4, r) : ML_LDS4(din + r * n + j0); };
        // BWD_ROWS rows per trip, all their loads requested before the first use (bwd_stats_kernel's form: same rows, same order, same sums)
        int64_t i = (int64_t)blockIdx.y * 16 + rg;
        for (; i + (BWD_ROWS - 1) * step < m; i += BWD_ROWS * step) {
            f32x4 dd[BWD_ROWS], zv[BWD_ROWS];
#pragma unroll
            for (int u = 0; u < BWD_ROWS; ++u) {
                dd[u] = grad_in(i + u * step);
                zv[u] = ML_LDS4(z + (i + u * step) * n + j0);
            }
#pragma unroll
            for (int u = 0; u < BWD_ROWS; ++u) one(dd[u], zv[u], i + u * step);
        }
        for (; i < m; i += step) one(grad_in(i), *(const f32x4*)(z + i * n + j0), i);
        if (blockIdx.y == 0 && rg == 0) {  // one thread per column publishes the parameter gradients
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dgamma[j0 + e] = (float)sdyx[j0 + e];
                dbeta[j0 + e] = (float)sdy[j0 + e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) r1[rg][cg * 4 + e] = a[e];
    __syncthreads();
    const int cj = threadIdx.x;
    if (cj < 64 && blockIdx.x * 64 + cj < n) {
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += r1[g][cj];
        atomicAdd(&sdz[blockIdx.x * 64 + cj], s);
    }
}

// bn_relu_drop_kernel that ALSO writes its output as lines (the next layer's GEMM operand); identical per-element
// arithmetic.  4 columns per lane (every fp32 access is a fully coalesced 16-byte one); an even / odd lane pair covers one
// 8-column group of a line: the even lane stores the group's 8 hi halves, the odd lane its 8 lo halves.  n % 8 == 0.
__global__ __launch_bounds__(256) void bn_relu_drop_lines_kernel(const float* __restrict__ z, int64_t m, int n,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float p_drop, uint32_t seed, uint32_t site,
                                                                const float* __restrict__ residual, float* __restrict__ y,
                                                                char* __restrict__ lines, const char* __restrict__ res_lines = nullptr) {
    const int qpr = n / 4;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = id < m * qpr;   // (m * n / 4 is even: a lane pair is live or dead together)
    const int64_t i = live ? id / qpr : 0;
    const int q = live ? (int)(id - i * qpr) : 0;
    const int j0 = q * 4;
    const int64_t base = i * n + j0;
    f32x4 v = ML_LDS4(z + base);
    const f32x4 mu = *(const f32x4*)(mean + j0), is = *(const f32x4*)(invstd + j0);
    const f32x4 ga = *(const f32x4*)(gamma + j0), be = *(const f32x4*)(beta + j0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = ga[e] * ((v[e] - mu[e]) * is[e]) + be[e];
        t = t > 0.f ? t : 0.f;
        if (p_drop > 0.f) t = (mlk::u01(seed, (uint32_t)i * 4099u + site, (uint32_t)(j0 + e)) >= p_drop) ? t / (1.f - p_drop) : 0.f;
        v[e] = t;
    }
    if (residual) {
        const f32x4 r = *(const f32x4*)(residual + base);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
    } else if (res_lines) {
        // the residual stream a_s exists as lines only on the large-batch route (round 4: no layer reads it as fp32): the value of
        // a line entry is hi + lo, exact in fp32 (22 significant bits; the stream is re-rounded to that once per stage)
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const char* rp = res_lines + i * (int64_t)n * 4 + (q >> 3) * 128 + ((q >> 1) & 3) * 16 + (q & 1) * 8;
        const h4 rh = ML_LDSV(h4, rp), rl = ML_LDSV(h4, rp + 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)rh[e] + (float)rl[e];
    }
    if (live && y) *(f32x4*)(y + base) = v;   // (y null: only the lines are needed -- a stage's inner activation on the large-batch route)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    unsigned hi2[2], lo2[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        _Float16 a0, b0, a1, b1;
        split_h(v[2 * e], a0, b0);
        split_h(v[2 * e + 1], a1, b1);
        hi2[e] = __builtin_bit_cast(unsigned, h2{a0, a1});
        lo2[e] = __builtin_bit_cast(unsigned, h2{b0, b1});
    }
    const bool odd = q & 1;
    const unsigned s0 = __shfl_xor(odd ? hi2[0] : lo2[0], 1, 64), s1 = __shfl_xor(odd ? hi2[1] : lo2[1], 1, 64);
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 out = odd ? u4{s0, s1, lo2[0], lo2[1]} : u4{hi2[0], hi2[1], s0, s1};
    const int g = q >> 1;   // 8-column group
    if (live && lines) ML_STU4(lines + i * (int64_t)n * 4 + (g >> 2) * 128 + (g & 3) * 16 + (odd ? 64 : 0), out);
}

// fp32 (m, n) -> TRANSPOSED lines [n][m_pad] (row j = column j of src over the batch, k32 blocks of 32 consecutive rows:
// the operands of the weight-gradient GEMM dW = dz^T . x, whose reduction runs over the batch), rows m..m_pad zero.
// PLAIN: src is a gradient dz: it is scaled by 2^e_dz (sc[3] = max|dz| -> max * 2^e in [2^13, 2^15); gradients carry a
// 1/m factor and would sit in fp16's subnormal range otherwise), ALSO written as ordinary lines (the operand of
// dx = dz . W), and workgroup (0, 0) publishes the two descale words.  One workgroup per 64 x 64 tile.
template <bool PLAIN>
__global__ __launch_bounds__(256) void tlines_kernel(const float* __restrict__ src, int64_t m, int n, int64_t m_pad, float* __restrict__ sc,
                                                    char* __restrict__ linesT, char* __restrict__ lines) {
    __shared__ float tile[64][68];   // [column][row]
    const int j0 = blockIdx.x * 64;
    const int64_t i0 = (int64_t)blockIdx.y * 64;
    float scale = 1.0f;
    if (PLAIN) {
        const int e = wscale_exp(sc[3]);
        scale = ldexpf(1.0f, e);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
            sc[4] = ldexpf(1.0f, -e) * sc[1];
            sc[5] = ldexpf(1.0f, -e);
        }
    }
    const int c4 = threadIdx.x & 15, r0 = threadIdx.x >> 4;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int r = r0 + 16 * ps;
        const int64_t i = i0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (i < m) v = *(const f32x4*)(src + i * n + j0 + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] *= scale;
            tile[c4 * 4 + e][r] = v[e];
        }
        if (PLAIN) {   // ordinary lines of the same (scaled) values: lane pairs exchange halves (see bn_relu_drop_lines_kernel)
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            unsigned hi2[2], lo2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                _Float16 a0, b0, a1, b1;
                split_h(v[2 * e], a0, b0);
                split_h(v[2 * e + 1], a1, b1);
                hi2[e] = __builtin_bit_cast(unsigned, h2{a0, a1});
                lo2[e] = __builtin_bit_cast(unsigned, h2{b0, b1});
            }
            const bool odd = c4 & 1;
            const unsigned s0 = __shfl_xor(odd ? hi2[0] : lo2[0], 1, 64), s1 = __shfl_xor(odd ? hi2[1] : lo2[1], 1, 64);
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const u4 out = odd ? u4{s0, s1, lo2[0], lo2[1]} : u4{hi2[0], hi2[1], s0, s1};
            const int g = (j0 >> 3) + (c4 >> 1);   // 8-column group of the row
            if (i < m) *(u4*)(lines + i * (int64_t)n * 4 + (g >> 2) * 128 + (g & 3) * 16 + (odd ? 64 : 0)) = out;
        }
    }
    __syncthreads();
    // 64 columns x 8 groups of 8 rows; a lane octet writes the two 128-byte lines of one column (256 contiguous bytes)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int col = (threadIdx.x >> 3) + 32 * ps, grp = threadIdx.x & 7;
        const f32x4 a = *(const f32x4*)&tile[col][grp * 8], b = *(const f32x4*)&tile[col][grp * 8 + 4];
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 p, q;
            split_h(e < 4 ? a[e] : b[e - 4], p, q);
            hi[e] = p;
            lo[e] = q;
        }
        char* dst = linesT + (int64_t)(j0 + col) * m_pad * 4 + (i0 >> 5) * 128 + (grp >> 2) * 128 + (grp & 3) * 16;
        *(h8*)dst = hi;
        *(h8*)(dst + 64) = lo;
    }
}

// ------------------------------------------------------------------------------------------------
// "Skinny" products (one side at most 68 wide: the input layer, the two output heads).  On the 128 x 128 tiles of
// sgemm_kernel they cost 0.2 - 0.9 ms each at 65536 rows (mostly padding); these are bound by the one (m x H) matrix they
// read or write.  NCMAX = 68 = the stereo input width.
constexpr int SK_NC = 68;

// out (m x n) [+]= s (m x nc, row stride lds) . W + bias,  W(c, j) = w[c * wsc + j * wsj]      (n % 64 == 0)
//   input layer forward (W = w1^T), data gradient of the heads (dy3 = dout . w_fin, dy2 += daux . w_aux).
// One workgroup per 64 columns x (64 rows per pass); thread = 4 columns x 4 rows, operands through LDS.
__global__ __launch_bounds__(256) void skinny_out_kernel(const float* __restrict__ s, int lds, int nc, const float* __restrict__ w,
                                                        int64_t wsc, int64_t wsj, const float* __restrict__ bias,
                                                        float* __restrict__ out, int64_t m, int n, int accumulate) {
    __shared__ __attribute__((aligned(16))) float wl[SK_NC][64];
    __shared__ __attribute__((aligned(16))) float sl[SK_NC][64];   // [c][row]
    const int jb = blockIdx.x * 64;
    for (int idx = threadIdx.x; idx < nc * 64; idx += 256) {
        const int c = idx >> 6, jj = idx & 63;
        wl[c][jj] = w[c * wsc + (int64_t)(jb + jj) * wsj];
    }
    const int cg = threadIdx.x & 15, rq = threadIdx.x >> 4;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias) b4 = *(const f32x4*)(bias + jb + cg * 4);
    // the skinny rows of the NEXT pass are requested before this pass is multiplied (round 4; a pass used to start with a global
    // load -> LDS -> barrier chain; measured neutral for the input layer's forward at 65536 rows: 158 us = 1.7 TB/s either way)
    constexpr int SPT = (SK_NC * 64 + 255) / 256;
    float sn[SPT];
    const int64_t step = (int64_t)gridDim.y * 64;
    auto request = [&](int64_t i0) {
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int idx = threadIdx.x + 256 * q;
            const int r = idx / nc, c = idx - r * nc;
            sn[q] = (idx < nc * 64 && i0 + r < m) ? s[(i0 + r) * lds + c] : 0.f;
        }
    };
    request((int64_t)blockIdx.y * 64);
    for (int64_t i0 = (int64_t)blockIdx.y * 64; i0 < m; i0 += step) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int idx = threadIdx.x + 256 * q;
            if (idx < nc * 64) {
                const int r = idx / nc, c = idx - r * nc;
                sl[c][r] = sn[q];
            }
        }
        __syncthreads();
        if (i0 + step < m) request(i0 + step);
        // packed fp32 FMAs (v_pk_fma_f32: two of the same fmaf per lane and instruction -- same bits, half the VALU instructions;
        // measured neutral at 65536 rows: the pass is bound by its store / staging latency, not by the VALU)
        f32x2 alo[4] = {b4.xy, b4.xy, b4.xy, b4.xy}, ahi[4] = {b4.zw, b4.zw, b4.zw, b4.zw};
        for (int c = 0; c < nc; ++c) {
            const f32x4 wv = *(const f32x4*)&wl[c][cg * 4];
            const f32x4 sv = *(const f32x4*)&sl[c][rq * 4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x2 sr = {sv[r], sv[r]};
                alo[r] = __builtin_elementwise_fma(sr, wv.xy, alo[r]);
                ahi[r] = __builtin_elementwise_fma(sr, wv.zw, ahi[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t i = i0 + rq * 4 + r;
            if (i < m) {
                float* o = out + i * n + jb + cg * 4;
                f32x4 a4 = {alo[r].x, alo[r].y, ahi[r].x, ahi[r].y};
                if (accumulate) a4 += *(const f32x4*)o;
                *(f32x4*)o = a4;
            }
        }
    }
}

// part[blockIdx.y][c][j] = sum over this workgroup's rows of s[i][c] * x[i][j]   (x: m x n, n % 64 == 0; c < nc <= 12 * NZ)
//   weight gradients of the heads (s = dout) and of the input layer (s = the input, x = dz; result transposed by the
//   reduction).  Thread = 4 columns x a row group, NCT accumulator rows; blockIdx.z selects a group of NCT skinny columns.
template <int NCT>
__global__ __launch_bounds__(256) void skinny_dw_kernel(const float* __restrict__ s, int lds, int nc, const float* __restrict__ x,
                                                       int64_t m, int n, float* __restrict__ part) {
    __shared__ float red[4][16][NCT * 4];
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int j0 = blockIdx.x * 64 + cg * 4;
    const int c0 = blockIdx.z * NCT;
    float acc[NCT][4];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c][e] = 0.f;
    if (NCT >= 8) {
        // Round 4 (the 34-column input layer's weight gradient at 65536 rows ran at 1 TB/s, 263 us for one read of dz): with NCT x 4
        // accumulators a SIMD holds ONE wave, and a wave with one 16-byte load in flight cannot cover the memory latency; the NCT
        // 4-byte loads of the skinny row per 16 bytes of x were load-issue-bound on top.  Now a workgroup walks chunks of 64
        // consecutive rows: the chunk's skinny rows go through LDS (a thread reads its row's values as NCT / 4 broadcast
        // ds_read_b128), its four x rows per thread are requested one chunk AHEAD (4 loads in flight per thread).
        constexpr int SP = (NCT + 3) / 4 * 4;
        constexpr int SPT = (64 * SP + 255) / 256;   // skinny values a thread stages per chunk
        __shared__ __attribute__((aligned(16))) float ss[64][SP];
        const int64_t nchunk = (m + 63) / 64;
        f32x4 vn[4];
        float sn[SPT];
        int sr[SPT], scn[SPT];   // (row, column) of this thread's staged values: chunk-invariant
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int idx = threadIdx.x + 256 * q;
            sr[q] = idx / SP;
            scn[q] = idx - sr[q] * SP;
        }
        auto request = [&](int64_t chunk) {   // the NEXT chunk's x rows and skinny values: in flight while this chunk is multiplied
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t i = chunk * 64 + rg + 16 * q;
                vn[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (chunk < nchunk && i < m) vn[q] = ML_LDS4(x + i * n + j0);
            }
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const int64_t i = chunk * 64 + sr[q];
                sn[q] = 0.f;
                if (chunk < nchunk && sr[q] < 64 && i < m && c0 + scn[q] < nc && scn[q] < NCT) sn[q] = s[i * lds + c0 + scn[q]];
            }
        };
        request(blockIdx.y);
        for (int64_t chunk = blockIdx.y; chunk < nchunk; chunk += gridDim.y) {
#pragma unroll
            for (int q = 0; q < SPT; ++q)
                if (sr[q] < 64) ss[sr[q]][scn[q]] = sn[q];
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = vn[q];
            __syncthreads();
            request(chunk + gridDim.y);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c4 = 0; c4 < SP / 4; ++c4) {
                    const f32x4 sv = *(const f32x4*)&ss[rg + 16 * q][c4 * 4];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
                        if (c4 * 4 + cc < NCT) {   // packed fp32 FMAs (v_pk_fma_f32): the same fmaf per element, two per instruction (neutral)
                            const f32x2 s2 = {sv[cc], sv[cc]};
                            f32x2 lo = {acc[c4 * 4 + cc][0], acc[c4 * 4 + cc][1]}, hi = {acc[c4 * 4 + cc][2], acc[c4 * 4 + cc][3]};
                            lo = __builtin_elementwise_fma(s2, v[q].xy, lo);
                            hi = __builtin_elementwise_fma(s2, v[q].zw, hi);
                            acc[c4 * 4 + cc][0] = lo.x;
                            acc[c4 * 4 + cc][1] = lo.y;
                            acc[c4 * 4 + cc][2] = hi.x;
                            acc[c4 * 4 + cc][3] = hi.y;
                        }
                }
            __syncthreads();
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.y * 16 + rg; i < m; i += (int64_t)gridDim.y * 16) {
            const f32x4 v = *(const f32x4*)(x + i * n + j0);
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const float sv = (c0 + c < nc) ? s[i * lds + c0 + c] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[c][e] = __builtin_fmaf(sv, v[e], acc[c][e]);
            }
        }
    }
    // the 16 row groups: 4 per wave (lane bits 4, 5), 4 waves
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = acc[c][e];
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            acc[c][e] = a;
        }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < 16) {
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wv][cg][c * 4 + e] = acc[c][e];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 16 * NCT * 4; idx += 256) {
        const int g = idx / (NCT * 4), r = idx - g * (NCT * 4);
        const int c = r >> 2, e = r & 3;
        if (c0 + c < nc)
            part[((int64_t)blockIdx.y * nc + c0 + c) * n + blockIdx.x * 64 + g * 4 + e] =
                (red[0][g][r] + red[1][g][r]) + (red[2][g][r] + red[3][g][r]);
    }
}

// dst[c][j] (or dst[j][c] when transpose) = sum over the partial planes, in order
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const float* __restrict__ part, int planes, int nc, int n,
                                                           float* __restrict__ dst, int transpose) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= nc * n) return;
    float a = 0.f;
    for (int z = 0; z < planes; ++z) a += part[(int64_t)z * nc * n + id];
    const int c = id / n, j = id - c * n;
    dst[transpose ? (int64_t)j * nc + c : id] = a;
}

// out[i][c] = x[i] . w[c] + b[c] for c < NC (the output heads, x: m x n, n % 4 == 0, NC * n <= 15360, out row stride ldo): a wave per
// row, the head weights in LDS (NC * n <= 15360 floats), butterfly reduction.
template <int NC>
__global__ __launch_bounds__(256) void skinny_heads_kernel(const float* __restrict__ x, int64_t m, int n, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int ldo) {
    // (dynamic: NC * n floats -- a fixed 60 KiB array held a CU to two workgroups = 8 waves, and the w_fin head at 65536 x 1024 ran at
    //  2.2 TB/s: 119 us; with its own 32 KiB five workgroups fit)
    extern __shared__ __attribute__((aligned(16))) float wl[];
    for (int idx = threadIdx.x; idx < NC * n; idx += 256) wl[idx] = w[idx];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // two rows per wave and pass (round 4): twice the loads in flight, every weight fragment read from LDS once for both
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wv; i < m; i += 2 * stride) {
        const int64_t i2 = i + stride;
        const bool two = i2 < m;
        float acc[NC], acc2[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = acc2[c] = 0.f;
        for (int j = lane * 4; j < n; j += 256) {
            const f32x4 v = ML_LDS4(x + i * n + j);
            f32x4 v2 = {0.f, 0.f, 0.f, 0.f};
            if (two) v2 = ML_LDS4(x + i2 * n + j);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4 ww = *(const f32x4*)&wl[c * n + j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[c] = __builtin_fmaf(v[e], ww[e], acc[c]);
                    acc2[c] = __builtin_fmaf(v2[e], ww[e], acc2[c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float a = acc[c], a2 = acc2[c];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                a += __shfl_xor(a, o, 64);
                a2 += __shfl_xor(a2, o, 64);
            }
            if (lane == c) {
                out[i * ldo + c] = a + bias[c];
                if (two) out[i2 * ldo + c] = a2 + bias[c];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 6, the large-batch route's w2 -> w3 pair as ONE Linear (reference architectures.py:60-66: y = w2(y); aux = w_aux(y); y = w3(y) --
// nothing non-linear between the two): z3 = a (W3 W2)^T + (W3 b2 + b3) and aux = a (W2^T w_aux) + (w_aux . b2 + b_aux), so the forward
// needs one batch-sized GEMM where it ran two, and the backward -- dW3 = (dz3^T a) W2^T + s3 (x) b2, dW2 = W3^T (dz3^T a) + w_aux (x) v,
// da = dz3 (W3 W2) + daux (x) u with v = daux^T a, u = W2^T w_aux, s3 = sum dz3 -- two where it ran four; y2 and its gradient never
// exist.  The H x H products of weights run on the exact-fp32 GEMM of the mid route (xgemm_kernel); these are the vector pieces.

// y[i] = sum_k A[i * lda + k] x[k] + add[i] * (scale ? *scale : 1)      (a wave per row)
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ A, int lda, int rows, int cols, const float* __restrict__ x,
                                                       const float* __restrict__ add, const float* __restrict__ scale,
                                                       float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    float a = 0.f;
    for (int k = lane; k < cols; k += 64) a = __builtin_fmaf(A[(int64_t)i * lda + k], x[k], a);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) y[i] = a + (add ? add[i] * (scale ? *scale : 1.0f) : 0.0f);
}
// y[k] = sum_i A[i * lda + k] x[i] + add[k] * (scale ? *scale : 1)      (a workgroup per 16 columns: 16 row groups x 16 columns, the row
// groups' sums added in order; cols % 16 == 0 -- one thread per column walking all rows took 200 us for a 1024 x 1024 matrix)
__global__ __launch_bounds__(256) void gemv_cols_kernel(const float* __restrict__ A, int lda, int rows, int cols, const float* __restrict__ x,
                                                       const float* __restrict__ add, const float* __restrict__ scale,
                                                       float* __restrict__ y) {
    __shared__ float red[16][16];
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + c;
    float a = 0.f;
#pragma unroll 8
    for (int i = rg; i < rows; i += 16) a = __builtin_fmaf(A[(int64_t)i * lda + k], x[i], a);
    red[rg][c] = a;
    __syncthreads();
    if (threadIdx.x < 16) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += red[r][threadIdx.x];
        y[k] = sum + (add ? add[k] * (scale ? *scale : 1.0f) : 0.0f);
    }
}
// C[i][j] += a[i] * b[j]
__global__ __launch_bounds__(256) void rank1_add_kernel(float* __restrict__ C, int ldc, int rows, int cols, const float* __restrict__ a,
                                                       const float* __restrict__ b) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)rows * cols) return;
    const int i = (int)(id / cols), j = (int)(id - (int64_t)i * cols);
    C[(int64_t)i * ldc + j] = __builtin_fmaf(a[i], b[j], C[(int64_t)i * ldc + j]);
}
// out[i * ldo] = sum_k value(lines[i][k]) u[k] + c[0]   -- skinny_heads_kernel<1> for an activation that exists as lines only (the value of
// a line entry is hi + lo, exact in fp32); a wave per row, two rows per pass, u in LDS.  n % 8 == 0, n <= 4096.
__global__ __launch_bounds__(256) void aux_lines_kernel(const char* __restrict__ lines, int64_t m, int n, const float* __restrict__ u,
                                                       const float* __restrict__ c, float* __restrict__ out, int ldo) {
    __shared__ __attribute__((aligned(16))) float ul[4096];
    for (int idx = threadIdx.x; idx < n; idx += 256) ul[idx] = u[idx];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t stride = (int64_t)gridDim.x * 4;
    auto dot = [&](int64_t i) {
        float a = 0.f;
        for (int g = lane; g < n / 8; g += 64) {   // 8-column group g of the row: 8 hi halves, 64 bytes further its 8 lo halves
            const char* q = lines + i * (int64_t)n * 4 + (g >> 2) * 128 + (g & 3) * 16;
            const h8 hi = ML_LDSV(h8, q), lo = ML_LDSV(h8, q + 64);
#pragma unroll
            for (int e = 0; e < 8; ++e) a = __builtin_fmaf((float)hi[e] + (float)lo[e], ul[g * 8 + e], a);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64);
        return a;
    };
    for (int64_t i = (int64_t)blockIdx.x * 4 + wv; i < m; i += 2 * stride) {
        const int64_t i2 = i + stride;
        const float a = dot(i);
        const float a2 = i2 < m ? dot(i2) : 0.f;
        if (lane == 0) {
            out[i * ldo] = a + c[0];
            if (i2 < m) out[i2 * ldo] = a2 + c[0];
        }
    }
}
// part[blockIdx.y][k] = sum over this workgroup's rows of s[i * lds] * value(lines[i][k])  -- skinny_dw_kernel<1> for an activation that
// exists as lines only (v = daux^T a); thread = one 8-column group x a row group, two rows per trip; skinny_reduce_kernel adds the planes.
__global__ __launch_bounds__(256) void dvec_lines_kernel(const float* __restrict__ s, int lds, const char* __restrict__ lines, int64_t m,
                                                        int n, float* __restrict__ part) {
    __shared__ float red[16][128];
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int g = blockIdx.x * 16 + cg;                  // 8-column group of the row
    const char* base = lines + (g >> 2) * 128 + (g & 3) * 16;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const int64_t step = (int64_t)gridDim.y * 16;
    auto one = [&](const h8& hi, const h8& lo, float sv) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(sv, (float)hi[e] + (float)lo[e], acc[e]);
    };
    int64_t i = (int64_t)blockIdx.y * 16 + rg;
    for (; i + step < m; i += 2 * step) {
        const char* q0 = base + i * (int64_t)n * 4;
        const char* q1 = base + (i + step) * (int64_t)n * 4;
        const h8 h0 = ML_LDSV(h8, q0), l0 = ML_LDSV(h8, q0 + 64), h1 = ML_LDSV(h8, q1), l1 = ML_LDSV(h8, q1 + 64);
        const float s0 = s[i * lds], s1 = s[(i + step) * lds];
        one(h0, l0, s0);
        one(h1, l1, s1);
    }
    if (i < m) {
        const char* q0 = base + i * (int64_t)n * 4;
        one(*(const h8*)q0, *(const h8*)(q0 + 64), s[i * lds]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rg][cg * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 128) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += red[r][threadIdx.x];
        part[(int64_t)blockIdx.y * n + blockIdx.x * 128 + threadIdx.x] = a;
    }
}

__global__ __launch_bounds__(256) void col_sum_to_float_kernel(const double* __restrict__ s, int n, float* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) out[j] = (float)s[j];
}

// all the fp64 column sums of a step that become fp32 gradient vectors (the Linear / head biases), in ONE launch before the optimizer
// (round 4: they were 11 launches of ~5 us each at 65536 rows); blockIdx.y = the item
struct ColSumItems {
    const double* src[16];
    float* dst[16];
    int n[16];
    int count;
};
__global__ __launch_bounds__(256) void col_sum_multi_kernel(ColSumItems it) {
    const int k = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (k < it.count && j < it.n[k]) it.dst[k][j] = (float)it.src[k][j];
}

__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ a, const float* __restrict__ b, int64_t n) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id < n) a[id] += b[id];
}

// ------------------------------------------------------------------------------------------------
// MultiTaskLoss of the reference (losses.py:59-73 with CompositeLoss :80-83): tasks d, x, y, h, w, l, ori
// (+ aux for stereo), all lambdas 1.  out (m, C) raw network rows, lab (m, L) label rows
// [theta, psi, z, d, h, w, l, sin, cos, yaw(, aux)] (process.py:293-301).  Writes dout (m, C) = dLoss/dout and
// accumulates the 8 per-task mean losses in fp64 (index 0..7 = d, x, y, h, w, l, ori, aux).
// tw (8 floats, or null = all 1): per-task weights of AutoTuneMultiTaskLoss (losses.py:17-43), 1 / (2 exp(log_sigma)^2);
// they scale dout only -- the accumulated means stay the plain task losses.  losses: LOSS_NV doubles (see loss_row).
// one row: o = raw network row (C), y = label row; g (C) receives dLoss/dout (null: values only); t[0..8) = the task terms d, x,
// y, h, w, l, ori, aux of this row (to be averaged over the batch); t[8], t[9] = the two VALIDATION-type terms that differ from
// the training ones (losses.py:85-96): |mu - d| (L1 instead of Laplace) and the angle error |atan2(o7, o8) - atan2(y7, y8)| in
// radians (the reference logs them for the training phase as well, trainer.py:163-165).
__device__ __forceinline__ void loss_row(const float* o, const float* y, int C, int64_t m, const float* w8, float* g, double* t) {
    const float invm = 1.f / (float)m;
    // LaplacianLoss on (mu, s) = out[2:4] vs d = lab[3]: |1 - mu/x| exp(-s) + 0.01 + s + 2   (losses.py:112-131)
    const float mu = o[2], s = o[3], x = y[3];
    const float norm = 1.f - mu / x, es = expf(-s);
    t[0] = (double)(fabsf(norm) * es + 0.01f + s + 2.f);
    const float sg = norm > 0.f ? 1.f : (norm < 0.f ? -1.f : 0.f);
    if (g) {
        g[2] = -sg / x * es * invm * w8[0];
        g[3] = (1.f - fabsf(norm) * es) * invm * w8[0];
    }
    // L1 on x (col 0 vs lab 0), y (1 vs 1), h, w, l (4..6 vs 4..6)
    const int oc[5] = {0, 1, 4, 5, 6}, lc[5] = {0, 1, 4, 5, 6};
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const float d = o[oc[q]] - y[lc[q]];
        t[1 + q] = (double)fabsf(d);
        if (g) g[oc[q]] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * invm * w8[1 + q];
    }
    // L1 on ori (cols 7,8 vs lab 7,8): mean over 2m elements
    double so = 0;
#pragma unroll
    for (int q = 7; q < 9; ++q) {
        const float d = o[q] - y[q];
        so += (double)fabsf(d);
        if (g) g[q] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * invm * 0.5f * w8[6];
    }
    t[6] = so * 0.5;
    t[7] = 0.0;
    if (C == 10) {  // BCEWithLogits on the aux logit vs lab[10]
        const float a = o[9], yy = y[10];
        t[7] = (double)(fmaxf(a, 0.f) - a * yy + log1pf(expf(-fabsf(a))));
        if (g) g[9] = (1.f / (1.f + expf(-a)) - yy) * invm * w8[7];
    }   // (mono: column 8 is (sin, cos)[1]; the aux head is the last column only for stereo)
    t[8] = (double)fabsf(mu - x);
    t[9] = (double)fabsf(atan2f(o[7], o[8]) - atan2f(y[7], y[8]));
}

constexpr int LOSS_NV = 10;   // values per row / per step: 8 task terms + the two validation-type ones

__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ out, int C, const float* __restrict__ lab, int L,
                                                  int64_t m, float* __restrict__ dout, double* __restrict__ losses,
                                                  const float* __restrict__ tw) {
    __shared__ double red[LOSS_NV][256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double t[LOSS_NV] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (i < m) {
        float w8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) w8[q] = tw ? tw[q] : 1.f;
        loss_row(out + i * C, lab + i * L, C, m, w8, dout + i * C, t);
    }
#pragma unroll
    for (int q = 0; q < LOSS_NV; ++q) red[q][threadIdx.x] = t[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int q = 0; q < LOSS_NV; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < LOSS_NV) atomicAdd(&losses[threadIdx.x], red[threadIdx.x][0] / (double)m);
}

// Validation statistics of raw outputs against labels, what the reference's Trainer computes with torch on the host
// (trainer.py:163-165, 197-246; losses.py:85-96): per row the LOSS_NV terms of loss_row, then bi = exp(s) d (unnormalize_bi,
// process.py:125-133), [|mu - d| <= bi], |mu - d| and its square (for the unbiased std), and for stereo |[sigmoid(a) >= 0.5] - label|
// (get_accuracy, trainer.py:384-389).  Per-workgroup partial SUMS (not means) to part[blockIdx.x][VAL_NV]; the host adds them in order.
constexpr int VAL_NV = LOSS_NV + 5;
__global__ __launch_bounds__(256) void val_stats_kernel(const float* __restrict__ out, int C, const float* __restrict__ lab, int L,
                                                       int64_t m, double* __restrict__ part) {
    __shared__ double red[VAL_NV][4];
    double t[VAL_NV];
#pragma unroll
    for (int q = 0; q < VAL_NV; ++q) t[q] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
        const float* o = out + i * C;
        const float* y = lab + i * L;
        float w8[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
        double r[LOSS_NV];
        loss_row(o, y, C, m, w8, nullptr, r);
#pragma unroll
        for (int q = 0; q < LOSS_NV; ++q) t[q] += r[q];
        const float err = fabsf(o[2] - y[3]);
        const float bi = expf(o[3]) * o[2];
        t[LOSS_NV + 0] += (double)bi;
        t[LOSS_NV + 1] += (err <= bi) ? 1.0 : 0.0;
        t[LOSS_NV + 2] += (double)err;
        t[LOSS_NV + 3] += (double)err * (double)err;
        if (C == 10) {
            const float sg = 1.f / (1.f + expf(-o[9]));
            t[LOSS_NV + 4] += (double)fabsf((sg >= 0.5f ? 1.f : 0.f) - y[10]);
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < VAL_NV; ++q) {
        double v = t[q];
        for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
        if (lane == 0) red[q][wv] = v;
    }
    __syncthreads();
    if (threadIdx.x < VAL_NV)
        part[(size_t)blockIdx.x * VAL_NV + threadIdx.x] =
            ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
}

// dst[i, :] = src[idx[i], :] (the batch of an epoch's row permutation: what the reference's DataLoader collates on the host)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int width, const int64_t* __restrict__ idx,
                                                         int64_t n, float* __restrict__ dst) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * width) return;
    const int64_t i = e / width;
    const int c = (int)(e - i * width);
    dst[e] = src[idx[i] * width + c];
}

// sum of squares of a flat fp32 buffer (gradient norm), fp64
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ out) {
    __shared__ double red[256];
    double a = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += (double)g[i] * (double)g[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, red[0]);
}

// clip_grad_norm_(params, max_norm) (trainer.py:159; torch: coef = min(1, max_norm / (norm + 1e-6))) fused with
// Adam (torch.optim.Adam defaults: betas 0.9/0.999, eps 1e-8, no weight decay, trainer.py:129) on the flat buffers.
__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m1,
                                                       float* __restrict__ m2, int64_t n, const double* __restrict__ sumsq,
                                                       float max_norm, float lr, float b1, float b2, float eps, float bc1,
                                                       float bc2, int do_adam) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float norm = (float)sqrt(*sumsq);
    float coef = max_norm / (norm + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    const float gi = g[i] * coef;
    g[i] = gi;
    if (!do_adam) return;
    const float a = m1[i] + (gi - m1[i]) * (1.f - b1);  // exp_avg.lerp_(grad, 1 - beta1)
    const float v = b2 * m2[i] + (1.f - b2) * gi * gi;
    m1[i] = a;
    m2[i] = v;
    const float denom = sqrtf(v) / sqrtf(bc2) + eps;
    w[i] -= (lr / bc1) * (a / denom);
}

}  // namespace mlt
