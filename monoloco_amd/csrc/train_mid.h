// train_mid.h -- the training step at the reference's REAL batch sizes (run.py:95 `--bs 512`, hyp_tuning.py:50 64..1024,
// loop body trainer.py:150-161): a few hundred to a few thousand rows, where the 256 x 256-tile kernels of the large-batch
// route keep a handful of CUs busy and the exact-fp32 GEMM needed ~175 launches per step.
//
// Data layout of this route ("mid"): every tensor is plain fp32 in HBM.
//   * activations / gradients that feed a GEMM exist row-major [m][H] AND transposed [H][ldt] (ldt = m rounded up to 32,
//     pad rows zero): the weight-gradient GEMM dW = dz^T . x reduces over the batch, so both of its operands must be
//     contiguous along the batch;
//   * every H x H weight exists as W [n][k] and as W^T [k][n]; the optimizer maintains both (adam_tile_kernel);
//   * per tensor one "max |.|" word (non-negative floats order like their bit patterns: atomicMax on the bits): the
//     power-of-two scale that keeps the fp16 lo halves of the 3-product scheme in range is derived from it by the CONSUMER.
// Kernels:
//   tgemm_kernel<BM>   C = A . B^T (+ bias) (+ res) for row-major fp32 A [M][K], B [N][K]: 3-product fp16 MFMA
//                      (hi.lo + lo.hi + hi.hi, fp32 accumulate) with the operands split into fp16 hi | lo ON THE FLY while
//                      they are staged into LDS (32 x 64 or 64 x 64 tiles: 176 .. 256 workgroups for a 331 .. 512-row batch).
//                      One kernel for forward (A = y, B = W), data gradient (A = dz, B = W^T) and weight gradient
//                      (A = dz^T, B = y^T).
//   fwd_apply_kernel   a workgroup OWNS 16 columns over ALL rows: batch statistics (exact, no atomics), BatchNorm, ReLU,
//                      dropout, residual; writes y row-major and transposed.  The input layer's narrow product runs in it.
//   bwd_apply_kernel   the same ownership backwards: (dy from the output heads |) dropout / ReLU mask, both BatchNorm
//                      reductions, dz row-major + transposed, bias / gamma / beta gradients, max |dz|.
//   adam_tile_kernel   clip + Adam on the H x H matrices in 64 x 64 tiles, writing W, W^T and max |W|;
//   adam_small_kernel  the remaining (narrow) tensors;  sumsq4_kernel  the gradient norm.
// Per-element arithmetic of BatchNorm, dropout, loss, clip and Adam is the exact route's (train_kernels.h).
#pragma once
#include "train_kernels.h"
#include "dense_kernel_pp.h"   // mlk::split2_scaled (v_fma_mix based fp32 -> fp16 hi | lo split)

namespace mlt {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8t __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------
struct TGemmParams {
    const float* a;      // [M][K] row-major, row stride lda (floats; multiple of 4, base 16-byte aligned)
    const float* b;      // [N][K] row-major, row stride ldb
    float* c;            // [M][N] row stride ldc
    const float* res;    // optional [M][N] (stride ldc) added to the result; may alias c
    const float* bias;   // optional [N]
    const float* amax;   // optional device word: max |A| (null: A is used unscaled)
    const float* bmax;   // optional device word: max |B|
    float* ct;           // optional transposed copy CT [N][ldct]; rows M .. ldct of it are written as zeros
    long lda, ldb, ldc, ldct;
    int M, N, K;         // N % 64 == 0, K % 32 == 0, any M >= 1 (rows beyond M are clamped on load, never stored)
    int dbg;             // timing ablations (ml_debug_tgemm only; 0 in the product): 1 no loads in the loop, 2 no conversion / LDS
                         // stores, 4 no fragment reads / MFMAs, 8 no barriers
};

// 8 consecutive fp32 -> one 16-byte chunk of fp16 hi halves and one of lo halves of (v * d), hi clamped to +-65504
__device__ __forceinline__ void conv8(const f32x4& v0, const f32x4& v1, float d, float lim, u32x4& h, u32x4& l) {
    unsigned h0, l0, h1, l1, h2, l2, h3, l3;
    mlk::split2_scaled<false>(v0[0], v0[1], d, lim, h0, l0);
    mlk::split2_scaled<false>(v0[2], v0[3], d, lim, h1, l1);
    mlk::split2_scaled<false>(v1[0], v1[1], d, lim, h2, l2);
    mlk::split2_scaled<false>(v1[2], v1[3], d, lim, h3, l3);
    h = u32x4{h0, h1, h2, h3};
    l = u32x4{l0, l1, l2, l3};
}

// LDS image of one k32 step of an operand tile: one 128-byte "line" per row = 4 chunks of 8 fp16 hi halves, then 4 chunks of
// lo halves; chunk c of row r sits at position c ^ ((r >> 1) & 7) (the layout dense_kernel.h reads conflict-free with
// ds_read_b128: lane (r = lane & 31, h = lane >> 5) takes chunk 2 kk + h (+ 4 for lo) of the k16 half-step kk).
template <int BM>
__global__ __launch_bounds__(256) void tgemm_kernel(TGemmParams p) {
    constexpr int BN = 64;
    constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + BN * 128;
    static_assert(BM == 32 || BM == 64, "tile");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // BM = 64: waves 2 (m) x 2 (n), each one 32 x 32 MFMA tile over the whole K.
    // BM = 32: waves 2 (n) x 2 (k16 half-steps of every k32 step); the two halves meet in LDS at the end.
    const int tn = w & 1;
    const int tm = (BM == 64) ? (w >> 1) : 0;
    const int wk = (BM == 64) ? 0 : (w >> 1);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    const int ea = p.amax ? wscale_exp(*p.amax) : 0;
    const int eb = p.bmax ? wscale_exp(*p.bmax) : 0;
    const float sa = ldexpf(1.0f, ea), sb = ldexpf(1.0f, eb);
    const float descale = ldexpf(1.0f, -(ea + eb));
    const float lima = 65504.0f / sa, limb = 65504.0f / sb;

    // ---- loader.  B tile (64 rows): thread -> (row = tid / 4, 8 consecutive k = (tid % 4) * 8 ..) of the k32 step: a wave
    // reads 16 full 128-byte row segments per instruction pair.  A tile: the same for BM = 64; for BM = 32 (4 KiB per step)
    // thread -> (row = tid / 8, 4 consecutive k): one 16-byte load, two 8-byte LDS stores.  Every thread issues the same
    // loads in the same order every step (no predicated or conditional load: the compiler's vmcnt counting stays exact and
    // the loads of step t+2 really stay in flight across the conversion of step t+1).
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int lrow = tid >> 2, kq = tid & 3;
    const int swl = (lrow >> 1) & 7;
    const int o_hi = lrow * 128 + ((kq ^ swl) * 16), o_lo = lrow * 128 + (((kq + 4) ^ swl) * 16);
    const float* bp = p.b + (size_t)(n0 + lrow) * p.ldb + kq * 8;
    const int arow_l = (BM == 64) ? lrow : (tid >> 3);
    int ar = m0 + arow_l;
    if (ar > p.M - 1) ar = p.M - 1;
    const float* ap = p.a + (size_t)ar * p.lda + ((BM == 64) ? kq * 8 : (tid & 7) * 4);
    // BM = 32: 4 floats = half a chunk: chunk (tid & 7) >> 1, 8-byte half tid & 1
    const int swa = (arow_l >> 1) & 7;
    const int oa_hi = (BM == 64) ? o_hi : arow_l * 128 + (((((tid & 7) >> 1)) ^ swa) * 16) + (tid & 1) * 8;
    const int oa_lo = (BM == 64) ? o_lo : arow_l * 128 + ((((((tid & 7) >> 1)) + 4) ^ swa) * 16) + (tid & 1) * 8;
    struct Raw {
        f32x4 a0, a1, b0, b1;
    };
    Raw R0, R1;
    auto gload = [&](Raw& R, int t) {
        R.a0 = *(const f32x4*)(ap + t * 32);
        if (BM == 64) R.a1 = *(const f32x4*)(ap + t * 32 + 4);
        R.b0 = *(const f32x4*)(bp + t * 32);
        R.b1 = *(const f32x4*)(bp + t * 32 + 4);
    };
    auto cstore = [&](const Raw& R, int stage) {
        char* sbuf = smem + stage * STAGE;
        u32x4 h, l;
        if (BM == 64) {
            conv8(R.a0, R.a1, sa, lima, h, l);
            *(u32x4*)(sbuf + oa_hi) = h;
            *(u32x4*)(sbuf + oa_lo) = l;
        } else {
            unsigned h0, l0, h1, l1;
            mlk::split2_scaled<false>(R.a0[0], R.a0[1], sa, lima, h0, l0);
            mlk::split2_scaled<false>(R.a0[2], R.a0[3], sa, lima, h1, l1);
            *(u32x2*)(sbuf + oa_hi) = u32x2{h0, h1};
            *(u32x2*)(sbuf + oa_lo) = u32x2{l0, l1};
        }
        conv8(R.b0, R.b1, sb, limb, h, l);
        *(u32x4*)(sbuf + A_BYTES + o_hi) = h;
        *(u32x4*)(sbuf + A_BYTES + o_lo) = l;
    };

    // ---- fragments
    const int ml = lane & 31, hh = lane >> 5;
    const int swf = (lane >> 1) & 7;
    const int a_row = (tm * 32 + ml) * 128, b_row = A_BYTES + (tn * 32 + ml) * 128;
    f32x16 acc, accx;   // hi.hi and the two cross products on separate accumulators: no MFMA waits for its predecessor
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = accx[e] = 0.f;
    auto step16 = [&](const char* sbuf, int kk) {
        const int c_hi = ((kk * 2 + hh) ^ swf) * 16, c_lo = ((kk * 2 + hh + 4) ^ swf) * 16;
        const half8t ah = *(const half8t*)(sbuf + a_row + c_hi), al = *(const half8t*)(sbuf + a_row + c_lo);
        const half8t bh = *(const half8t*)(sbuf + b_row + c_hi), bl = *(const half8t*)(sbuf + b_row + c_lo);
        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accx, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx, 0, 0, 0);
    };
    auto compute = [&](int stage) {
        const char* sbuf = smem + stage * STAGE;
        if (BM == 64) {
            step16(sbuf, 0);
            step16(sbuf, 1);
        } else {
            step16(sbuf, wk);
        }
    };

    // two k-steps deep: step t+2 is requested while step t is multiplied and step t+1 sits converted in the other stage.
    // Unconditional body (loads past the end re-read the last step, the surplus conversion lands in the stage nobody reads).
    const int nk = p.K / 32;
    const int last = nk - 1;
    gload(R0, 0);
    gload(R1, last < 1 ? last : 1);
    cstore(R0, 0);
    __syncthreads();
    int t = 0;
    if (p.dbg) {   // timing ablations: the same loop with parts left out (results are garbage)
        for (; t + 1 < nk; t += 2) {
            if (!(p.dbg & 1)) gload(R0, t + 2 < last ? t + 2 : last);
            __builtin_amdgcn_sched_barrier(0);
            if (!(p.dbg & 4)) compute(0);
            if (!(p.dbg & 2)) cstore(R1, 1);
            if (!(p.dbg & 8)) __syncthreads();
            if (!(p.dbg & 1)) gload(R1, t + 3 < last ? t + 3 : last);
            __builtin_amdgcn_sched_barrier(0);
            if (!(p.dbg & 4)) compute(1);
            if (!(p.dbg & 2)) cstore(R0, 0);
            if (!(p.dbg & 8)) __syncthreads();
        }
    }
    for (; t + 1 < nk; t += 2) {
        gload(R0, t + 2 < last ? t + 2 : last);
        __builtin_amdgcn_sched_barrier(0);   // the requests leave FIRST: hipcc otherwise sinks them behind the conversion
        compute(0);
        cstore(R1, 1);       // step t+1; stage 1 was last read one barrier ago
        __syncthreads();
        gload(R1, t + 3 < last ? t + 3 : last);
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        cstore(R0, 0);       // step t+2
        __syncthreads();
    }
    if (t < nk) compute(0);  // odd number of steps: the last one sits in stage 0
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] += accx[e];

    bool fin = true;
    if (BM == 32) {   // the k halves meet: waves 2, 3 hand their partial tile to waves 0, 1
        float* red = (float*)smem;   // [2][16][64]
        if (wk == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(tn * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        fin = (wk == 0);
        if (fin) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += red[(tn * 16 + r) * 64 + lane];
        }
    }
    if (!fin) return;

    // ---- epilogue.  D layout of the 32 x 32 MFMA: lane holds column j = lane & 31 (a B row) and rows
    // i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (A rows): for a fixed r the lanes 0..31 write 128 contiguous bytes of a row of C
    const int j = n0 + tn * 32 + ml;
    const float bj = p.bias ? p.bias[j] : 0.f;
    const int ibase = m0 + tm * 32 + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 vt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = ibase + 8 * g + e;
            float v = 0.f;
            if (i < p.M) {
                v = acc[g * 4 + e] * descale + bj;
                const size_t o = (size_t)i * p.ldc + j;
                if (p.res) v += p.res[o];
                p.c[o] = v;
            }
            vt[e] = v;
        }
        if (p.ct) {
            const int i0 = ibase + 8 * g;
            if (i0 < p.ldct) *(f32x4*)(p.ct + (size_t)j * p.ldct + i0) = vt;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Forward of one block Linear -> BatchNorm1d (train mode) -> ReLU -> Dropout (-> + residual), everything behind the GEMM
// (reference architectures.py:50-52, 90-100).  One workgroup per 16 columns, all rows: thread = (row r = tid / 4 + 64 pass,
// 4 columns) -- 64-byte row segments.  Pass 1: column sums in fp64; pass 2: the activation, written row-major and (through
// a 16 x 64 LDS tile) transposed.  Input-layer mode (x_in != null): z = x_in . w_in^T + b_in is computed here (the narrow
// product of train_kernels.h' skinny_out_kernel, same fma order) and written to z_out.
struct FwdApplyParams {
    const float* z;
    const float* x_in;
    const float* w_in;
    const float* b_in;
    float* z_out;
    int in_dim;
    long m;
    int H;
    long ldt;
    const float* gamma;
    const float* beta;
    float* run_mean;
    float* run_var;
    float* mean_out;
    float* invstd_out;
    float p_drop;
    uint32_t seed, site;
    const float* residual;
    float* y;
    float* yT;           // [H][ldt] or null
    float* zero_words;   // optional: block 0 zeroes n_zero floats (the next step's max |W| words)
    int n_zero;
};

__global__ __launch_bounds__(256) void fwd_apply_kernel(FwdApplyParams p) {
    __shared__ double r1[64][17], r2[64][17];
    __shared__ float stat[2][16];
    __shared__ __attribute__((aligned(16))) float tile[16][68];
    __shared__ float wl[16 * SK_NC];
    const int tid = threadIdx.x, cq = tid & 3, r = tid >> 2;
    const int j0 = blockIdx.x * 16, j = j0 + cq * 4;
    const int H = p.H;
    const bool inl = p.x_in != nullptr;
    if (p.zero_words && blockIdx.x == 0 && tid < p.n_zero) p.zero_words[tid] = 0.f;
    if (inl) {
        for (int idx = tid; idx < 16 * p.in_dim; idx += 256) wl[idx] = p.w_in[(size_t)j0 * p.in_dim + idx];
        __syncthreads();
    }
    const float* zsrc = inl ? p.z_out : p.z;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    for (long i = r; i < p.m; i += 64) {
        f32x4 v;
        if (inl) {
            v = *(const f32x4*)(p.b_in + j);
            const float* xr = p.x_in + i * p.in_dim;
            for (int c = 0; c < p.in_dim; ++c) {
                const float xv = xr[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(xv, wl[(cq * 4 + e) * p.in_dim + c], v[e]);
            }
            *(f32x4*)(p.z_out + i * H + j) = v;
        } else {
            v = *(const f32x4*)(p.z + i * H + j);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] += (double)v[e];
            b[e] += (double)v[e] * (double)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r1[r][cq * 4 + e] = a[e];
        r2[r][cq * 4 + e] = b[e];
    }
    __syncthreads();
    if (tid < 16) {
        double sa = 0.0, sb = 0.0;
        for (int g = 0; g < 64; ++g) {
            sa += r1[g][tid];
            sb += r2[g][tid];
        }
        // bn_finalize_kernel's arithmetic (train_kernels.h)
        const double mu = sa / (double)p.m;
        double var = sb / (double)p.m - mu * mu;
        if (var < 0) var = 0;
        const float mean = (float)mu, inv = (float)(1.0 / sqrt(var + 1e-5));
        const double unb = p.m > 1 ? var * (double)p.m / (double)(p.m - 1) : var;
        const int jj = j0 + tid;
        p.mean_out[jj] = mean;
        p.invstd_out[jj] = inv;
        p.run_mean[jj] = (1.f - 0.1f) * p.run_mean[jj] + 0.1f * (float)mu;
        p.run_var[jj] = (1.f - 0.1f) * p.run_var[jj] + 0.1f * (float)unb;
        stat[0][tid] = mean;
        stat[1][tid] = inv;
    }
    __syncthreads();
    f32x4 mu, is;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        mu[e] = stat[0][cq * 4 + e];
        is[e] = stat[1][cq * 4 + e];
    }
    const f32x4 ga = *(const f32x4*)(p.gamma + j), be = *(const f32x4*)(p.beta + j);
    const long npass = ((p.yT ? p.ldt : p.m) + 63) / 64;
    for (long ps = 0; ps < npass; ++ps) {
        const long i = ps * 64 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (i < p.m) {
            v = *(const f32x4*)(zsrc + i * H + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // bn_relu_drop_kernel's arithmetic
                float t = ga[e] * ((v[e] - mu[e]) * is[e]) + be[e];
                t = t > 0.f ? t : 0.f;
                if (p.p_drop > 0.f)
                    t = (mlk::u01(p.seed, (uint32_t)i * 4099u + p.site, (uint32_t)(j + e)) >= p.p_drop) ? t / (1.f - p.p_drop) : 0.f;
                v[e] = t;
            }
            if (p.residual) {
                const f32x4 rr = *(const f32x4*)(p.residual + i * H + j);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rr[e];
            }
            *(f32x4*)(p.y + i * H + j) = v;
        }
        if (p.yT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[cq * 4 + e][r] = v[e];
            __syncthreads();
            const int c = tid >> 4, rq = tid & 15;
            const long it = ps * 64 + rq * 4;
            if (it < p.ldt) *(f32x4*)(p.yT + (size_t)(j0 + c) * p.ldt + it) = *(const f32x4*)&tile[c][rq * 4];
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the same block, behind the data-gradient GEMM that produced dy (the gradient wrt the block's output):
//   dy   = incoming * [gamma * xhat + beta > 0] * dropout_mask / (1 - p),   xhat = (z - mean) * invstd      (bwd_elem)
//   dz   = gamma * invstd / m * (m * dy - sum(dy) - xhat * sum(dy * xhat))                                  (bn_bwd_fused_kernel)
//   dgamma = sum(dy * xhat), dbeta = sum(dy), dbias (of the Linear) = sum(dz)
// One workgroup per 16 columns, all rows; the incoming gradient is re-derived in both passes:
//   * from dy (a buffer), optionally + aux_d[i] * w_aux[j] (the one-output head's data gradient, skinny_out_kernel's
//     accumulate form), or
//   * from the output heads: sum_c dout[i][c] * w_head[c][j]  (skinny_out_kernel's fma order).
// z == null: a Linear without BatchNorm (w2): dz = dy.  Writes dz row-major and transposed and folds max |dz| into a word.
struct BwdApplyParams {
    const float* dy;
    const float* dout;
    int dld, nc;
    const float* w_head;
    const float* aux_d;
    int aux_ld;
    const float* w_aux;
    const float* z;
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
    float p_drop;
    uint32_t seed, site;
    long m;
    int H;
    long ldt;
    float* dz;
    float* dzT;     // [H][ldt] or null
    float* dgamma;
    float* dbeta;
    float* dbias;
    float* dzmax;   // word or null
};

__device__ __forceinline__ f32x4 bwd_incoming(const BwdApplyParams& p, long i, int j, const float* wh /* LDS [nc][16] */, int cq) {
    f32x4 v;
    if (p.dout) {
        v = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* dr = p.dout + i * p.dld;
        for (int c = 0; c < p.nc; ++c) {
            const float s = dr[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(s, wh[c * 16 + cq * 4 + e], v[e]);
        }
    } else {
        v = *(const f32x4*)(p.dy + i * p.H + j);
    }
    if (p.aux_d) {
        const float s = p.aux_d[i * p.aux_ld];
        const f32x4 wa = *(const f32x4*)(p.w_aux + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(s, wa[e], 0.f) + v[e];
    }
    return v;
}

__global__ __launch_bounds__(256) void bwd_apply_kernel(BwdApplyParams p) {
    __shared__ double r1[64][17], r2[64][17];
    __shared__ float stat[2][16];
    __shared__ __attribute__((aligned(16))) float tile[16][68];
    __shared__ float wh[16 * 16];   // head weights of these 16 columns, [c][16]
    const int tid = threadIdx.x, cq = tid & 3, r = tid >> 2;
    const int j0 = blockIdx.x * 16, j = j0 + cq * 4;
    const int H = p.H;
    const bool bn = p.z != nullptr;
    if (p.dout) {
        for (int idx = tid; idx < p.nc * 16; idx += 256) wh[idx] = p.w_head[(size_t)(idx >> 4) * H + j0 + (idx & 15)];
        __syncthreads();
    }
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = mu, ga = mu, be = mu;
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f}, gg[4] = {0.f, 0.f, 0.f, 0.f};
    if (bn) {
        mu = *(const f32x4*)(p.mean + j);
        is = *(const f32x4*)(p.invstd + j);
        ga = *(const f32x4*)(p.gamma + j);
        be = *(const f32x4*)(p.beta + j);
        double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
        for (long i = r; i < p.m; i += 64) {
            const f32x4 d = bwd_incoming(p, i, j, wh, cq);
            const f32x4 zz = *(const f32x4*)(p.z + i * H + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float dy, xh;
                bwd_elem(d[e], zz[e], mu[e], is[e], ga[e], be[e], p.p_drop, p.seed, p.site, i, j + e, dy, xh);
                a[e] += (double)dy;
                b[e] += (double)dy * (double)xh;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r1[r][cq * 4 + e] = a[e];
            r2[r][cq * 4 + e] = b[e];
        }
        __syncthreads();
        if (tid < 16) {
            double s1 = 0.0, s2 = 0.0;
            for (int g = 0; g < 64; ++g) {
                s1 += r1[g][tid];
                s2 += r2[g][tid];
            }
            stat[0][tid] = (float)s1;
            stat[1][tid] = (float)s2;
            p.dbeta[j0 + tid] = (float)s1;
            p.dgamma[j0 + tid] = (float)s2;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sa[e] = stat[0][cq * 4 + e];
            sb[e] = stat[1][cq * 4 + e];
            gg[e] = ga[e] * is[e] / (float)p.m;
        }
    }
    double a2[4] = {0.0, 0.0, 0.0, 0.0};
    float mx = 0.f;
    const long npass = ((p.dzT ? p.ldt : p.m) + 63) / 64;
    for (long ps = 0; ps < npass; ++ps) {
        const long i = ps * 64 + r;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (i < p.m) {
            const f32x4 d = bwd_incoming(p, i, j, wh, cq);
            if (bn) {
                const f32x4 zz = *(const f32x4*)(p.z + i * H + j);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float dy, xh;
                    bwd_elem(d[e], zz[e], mu[e], is[e], ga[e], be[e], p.p_drop, p.seed, p.site, i, j + e, dy, xh);
                    o[e] = gg[e] * ((float)p.m * dy - sa[e] - xh * sb[e]);
                }
            } else {
                o = d;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a2[e] += (double)o[e];
                mx = __builtin_fmaxf(mx, __builtin_fabsf(o[e]));
            }
            *(f32x4*)(p.dz + i * H + j) = o;
        }
        if (p.dzT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[cq * 4 + e][r] = o[e];
            __syncthreads();
            const int c = tid >> 4, rq = tid & 15;
            const long it = ps * 64 + rq * 4;
            if (it < p.ldt) *(f32x4*)(p.dzT + (size_t)(j0 + c) * p.ldt + it) = *(const f32x4*)&tile[c][rq * 4];
            __syncthreads();
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) r1[r][cq * 4 + e] = a2[e];
    __syncthreads();
    if (tid < 16) {
        double s = 0.0;
        for (int g = 0; g < 64; ++g) s += r1[g][tid];
        p.dbias[j0 + tid] = (float)s;
    }
    if (p.dzmax) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((tid & 63) == 0) {
            if (!(mx < 3.0e38f)) mx = 3.0e38f;
            atomicMax((unsigned*)p.dzmax, __builtin_bit_cast(unsigned, mx));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Optimizer.  clip_adam_kernel's arithmetic (clip_grad_norm_(3) + torch.optim.Adam defaults), on
//   * the H x H matrices in 64 x 64 tiles: W, m1, m2, the clipped gradient, AND W^T (through LDS) and max |W| of the new
//     weights (what the next step's GEMMs scale by);
//   * everything else through a segment table.
constexpr int MID_MAXMAT = 40;
struct AdamMats {
    long off[MID_MAXMAT];   // flat offsets of the H x H weight matrices
    int count;
};
struct AdamSegs {
    long off[MID_MAXMAT];     // flat offset of each segment
    long start[MID_MAXMAT + 1];   // prefix sums of the segment lengths
    int count;
};
struct AdamHyper {
    const double* sumsq;
    float max_norm, lr, b1, b2, eps, bc1, bc2;
    int do_adam;
};

__device__ __forceinline__ float adam_elem(float& w, float g, float& m1, float& m2, float coef, const AdamHyper& hp) {
    const float gi = g * coef;
    if (hp.do_adam) {
        const float a = m1 + (gi - m1) * (1.f - hp.b1);  // exp_avg.lerp_(grad, 1 - beta1)
        const float v = hp.b2 * m2 + (1.f - hp.b2) * gi * gi;
        m1 = a;
        m2 = v;
        const float denom = sqrtf(v) / sqrtf(hp.bc2) + hp.eps;
        w -= (hp.lr / hp.bc1) * (a / denom);
    }
    return gi;
}
__device__ __forceinline__ float clip_coef(const AdamHyper& hp) {
    const float norm = (float)sqrt(*hp.sumsq);
    float coef = hp.max_norm / (norm + 1e-6f);
    return coef > 1.f ? 1.f : coef;
}

__global__ __launch_bounds__(256) void adam_tile_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m1,
                                                       float* __restrict__ m2, AdamMats mats, int H, float* __restrict__ wT,
                                                       float* __restrict__ wmax, AdamHyper hp) {
    __shared__ float tile[64][65];   // [column][row]
    const int tid = threadIdx.x, cq = tid & 15, r = tid >> 4;
    const int TR = H / 64;
    const int tr = blockIdx.x / TR, tc = blockIdx.x - tr * TR;
    const long base = mats.off[blockIdx.y];
    const float coef = clip_coef(hp);
    float mx = 0.f;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int row = ps * 16 + r;
        const long idx = base + (long)(tr * 64 + row) * H + tc * 64 + cq * 4;
        f32x4 wv = *(const f32x4*)(w + idx), gv = *(const f32x4*)(g + idx);
        f32x4 av = {0.f, 0.f, 0.f, 0.f}, vv = av;
        if (hp.do_adam) {
            av = *(const f32x4*)(m1 + idx);
            vv = *(const f32x4*)(m2 + idx);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float we = wv[e], ae = av[e], ve = vv[e];
            gv[e] = adam_elem(we, gv[e], ae, ve, coef, hp);
            wv[e] = we;
            av[e] = ae;
            vv[e] = ve;
            tile[cq * 4 + e][row] = we;
            mx = __builtin_fmaxf(mx, __builtin_fabsf(we));
        }
        *(f32x4*)(g + idx) = gv;
        if (hp.do_adam) {
            *(f32x4*)(w + idx) = wv;
            *(f32x4*)(m1 + idx) = av;
            *(f32x4*)(m2 + idx) = vv;
        }
    }
    if (!hp.do_adam) return;   // (uniform) the weights did not move: W^T and max |W| stay valid
    __syncthreads();
    float* dstT = wT + (size_t)blockIdx.y * H * H;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int c = ps * 16 + r;   // column of W = row of W^T
        const f32x4 o = {tile[c][cq * 4], tile[c][cq * 4 + 1], tile[c][cq * 4 + 2], tile[c][cq * 4 + 3]};
        *(f32x4*)(dstT + (size_t)(tc * 64 + c) * H + tr * 64 + cq * 4) = o;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) {
        if (!(mx < 3.0e38f)) mx = 3.0e38f;
        atomicMax((unsigned*)(wmax + blockIdx.y), __builtin_bit_cast(unsigned, mx));
    }
}

__global__ __launch_bounds__(256) void adam_small_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m1,
                                                        float* __restrict__ m2, AdamSegs segs, AdamHyper hp) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= segs.start[segs.count]) return;
    int s = 0;
    while (s + 1 < segs.count && id >= segs.start[s + 1]) ++s;
    const long i = segs.off[s] + (id - segs.start[s]);
    const float coef = clip_coef(hp);
    float we = w[i], ae = 0.f, ve = 0.f;
    if (hp.do_adam) {
        ae = m1[i];
        ve = m2[i];
    }
    g[i] = adam_elem(we, g[i], ae, ve, coef, hp);
    if (hp.do_adam) {
        w[i] = we;
        m1[i] = ae;
        m2[i] = ve;
    }
}

// W^T and max |W| of every H x H matrix from W (after set_tensor / load_state_dict; the words are zeroed by the caller)
__global__ __launch_bounds__(256) void wt_refresh_kernel(const float* __restrict__ w, AdamMats mats, int H, float* __restrict__ wT,
                                                        float* __restrict__ wmax) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x, cq = tid & 15, r = tid >> 4;
    const int TR = H / 64;
    const int tr = blockIdx.x / TR, tc = blockIdx.x - tr * TR;
    const long base = mats.off[blockIdx.y];
    float mx = 0.f;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int row = ps * 16 + r;
        const f32x4 wv = *(const f32x4*)(w + base + (long)(tr * 64 + row) * H + tc * 64 + cq * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[cq * 4 + e][row] = wv[e];
            mx = __builtin_fmaxf(mx, __builtin_fabsf(wv[e]));
        }
    }
    __syncthreads();
    float* dstT = wT + (size_t)blockIdx.y * H * H;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int c = ps * 16 + r;
        const f32x4 o = {tile[c][cq * 4], tile[c][cq * 4 + 1], tile[c][cq * 4 + 2], tile[c][cq * 4 + 3]};
        *(f32x4*)(dstT + (size_t)(tc * 64 + c) * H + tr * 64 + cq * 4) = o;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) {
        if (!(mx < 3.0e38f)) mx = 3.0e38f;
        atomicMax((unsigned*)(wmax + blockIdx.y), __builtin_bit_cast(unsigned, mx));
    }
}

// sum of squares of a flat fp32 buffer with 16-byte loads (the gradient norm); n % 4 tail by block 0
__global__ __launch_bounds__(256) void sumsq4_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ out) {
    __shared__ double red[256];
    double a = 0;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(g + i * 4);
        a += ((double)v[0] * (double)v[0] + (double)v[1] * (double)v[1]) + ((double)v[2] * (double)v[2] + (double)v[3] * (double)v[3]);
    }
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < (n & 3)) {
        const double v = (double)g[n4 * 4 + threadIdx.x];
        a += v * v;
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, red[0]);
}

}  // namespace mlt
