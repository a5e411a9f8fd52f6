// dense_kernel.h -- the hot kernel: one Linear(+folded BatchNorm)(+ReLU)(+residual) layer of
// LocoModel (reference monoloco/network/architectures.py:50-66, 90-100) on CDNA4 MFMA.
//
//   Y[m][n] = act( descale * sum_k X[m][k] * Ws[n][k] + bias[n] ) (+ R[m][n])
//
// Precision: the reference computes in fp32 and the parity bar is 1e-4 abs on metres, which
// plain fp16/bf16 operands miss by 2-3 orders of magnitude.  Operands are therefore
// carried as an fp16 (hi, lo) pair, v = hi + lo (22 significant bits), and each product is
// three MFMAs, hi*hi + hi*lo + lo*hi, accumulated in fp32.  Weights are pre-scaled by a per-layer
// power of two (Ws = W * 2^e, descale = 2^-e) so that their lo parts stay in fp16's normal range.
//
// Memory format ("k32 hi|lo lines"), used for activations and weights alike: a row of K values is
// K/32 lines of 128 bytes; line b holds fp16 hi[32b .. 32b+31] (64 B) then fp16 lo[...] (64 B).
// One k-step of the GEMM therefore streams exactly one full 128-byte line per row.
//
// Decomposition: workgroup tile 256 (n, weight rows) x 256 (m, persons), 8 waves as 2 (n) x 4 (m),
// wave tile 128 n x 64 m = 4 x 2 MFMA 32x32x16 tiles, computed TRANSPOSED (A operand = weights,
// B operand = activations) so that each lane ends up holding 4 consecutive n for one m -- what the
// line format wants in the epilogue.  K step 32 (one line); LDS holds 2 stages x (256+256) rows x
// 128 B = 128 KiB, filled with global_load_lds (16 B / lane) and XOR-swizzled on the SOURCE side
// (chunk ^= (row>>1)&7) so that the ds_read_b128 fragment reads are bank-conflict free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mlk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256;            // persons per workgroup tile  (MFMA j dimension)
constexpr int BN = 256;            // weight rows per workgroup tile (MFMA i dimension)
constexpr int LINE = 128;          // bytes per row per k-step (32 hi + 32 lo fp16)
constexpr int TILE_BYTES = 256 * LINE;      // 32 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES; // W tile then X tile
constexpr int DENSE_LDS = 2 * STAGE_BYTES;  // 128 KiB
constexpr int DENSE_THREADS = 512;

struct DenseParams {
    const char* x;      // [M_pad][K] line format
    const char* w;      // [N][K]     line format (pre-scaled)
    const float* bias;  // [N]
    const float* bias_scaled;  // [N] bias * 2^e: dense_kernel_pp starts its accumulators there
    const char* res;    // [M_pad][N] line format or nullptr (may alias y: same tile, same threads)
    char* y;            // [M_pad][N] line format
    float descale;      // 2^-e
    const float* descale_ptr;  // if not null: 2^-e lives on the device (weights packed on the device, training path)
    int M_pad, N, K;    // M_pad % 256 == 0, N % 256 == 0, K % 32 == 0
    int ksplit = 1;     // dense_kernel_w4<.., -3> only: work items per output tile, each reducing over K columns
    int relu;
    int debug;          // bring-up/ablation bits (0 in production), see dense_kernel_pp.h
    unsigned long long* trace;  // optional s_memtime trace [grid][8 waves][64], nullptr in production
    // fused head (dense_kernel_pp<.., HEAD>): out[m][o] partial sums instead of the activation tile
    const float* head_w;   // [HEAD][N] fp32
    float* head_part;      // [2*N/256][M_pad][16] fp32 partial dot products (per 128-column slice)
    // dense_kernel_w4<.., -2> (training forward): column sums and sums of squares of the STORED fp32 tile, rows < m_valid, per
    // 128-row block: colpart[(row block) * 2 * N + j] = sum, [.. + N + j] = sum of squares (fp64; BatchNorm batch statistics
    // without a second pass over z); nullptr: none
    double* colpart = nullptr;
    int m_valid = 0;
};

__device__ __forceinline__ void glds16(const char* gsrc, char* lds_dst) {
    // async 16 B/lane global -> LDS; LDS destination = wave-uniform base + lane*16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// split an fp32 value into fp16 hi + fp16 lo (round-to-nearest-even both), hi clamped to the
// finite fp16 range so that an out-of-range activation degrades instead of producing inf/NaN.
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
    float c = __builtin_fminf(__builtin_fmaxf(v, -65504.0f), 65504.0f);
    hi = (_Float16)c;
    lo = (_Float16)(c - (float)hi);  // of the CLAMPED value: beyond the range the pair saturates at +-65504, lo = 0
}

#ifdef ML_BRINGUP  // the first-generation kernel (one tile per workgroup, one barrier per k-step): A/B reference only
template <int NSPLIT>
__global__ __launch_bounds__(DENSE_THREADS, 2) void dense_kernel(DenseParams p) {
    __shared__ __attribute__((aligned(16))) char smem[DENSE_LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 1;   // 2 waves along n (128 weight rows each)
    const int wm = w >> 1;  // 4 waves along m (64 persons each)

    // ---- XCD-aware, bijective workgroup -> tile map: the N/256 column tiles of one row panel
    // run back to back on the same XCD so the X panel is fetched from HBM once and re-used from
    // that XCD's L2 (block b is observed to land on XCD b % 8; speed only, never correctness).
    const int NT = p.N / BN;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    const int mt = tile / NT, nt = tile - mt * NT;
    const int m0 = mt * BM, n0 = nt * BN;

    const size_t rowb = (size_t)p.K * 4;  // bytes per row of X and W
    const char* wtile = p.w + (size_t)n0 * rowb;
    const char* xtile = p.x + (size_t)m0 * rowb;

    // ---- stage loader: 64 wave-instructions of 1 KiB (8 rows x 128 B) per stage, 8 per wave.
    // lane -> (row = base + lane/8, LDS chunk position = lane%8); it fetches source chunk
    // pos ^ ((row>>1)&7), i.e. the swizzle is applied on the global address (LDS dest is linear).
    unsigned goff[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const int r = w * 32 + q4 * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        goff[q4] = (unsigned)(r * (int)rowb + c * 16);
    }
    auto issue = [&](int kt, int stage) {
        char* sb = smem + stage * STAGE_BYTES + (w * 32) * LINE;
        const unsigned ko = (unsigned)kt * LINE;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            glds16(wtile + goff[q4] + ko, sb + q4 * 8 * LINE);
            glds16(xtile + goff[q4] + ko, sb + TILE_BYTES + q4 * 8 * LINE);
        }
    };

    // ---- fragment addressing.  MFMA 32x32x16: lane l supplies row (l&31), k = 8*(l>>5)..+7 of
    // the k16 step for both operands; that is chunk kk*2 + (l>>5) of the hi half (+4 for lo).
    const int sw = (lane >> 1) & 7;  // (row>>1)&7 with row = 32*t + (lane&31)
    const int h = lane >> 5;
    const int wrow = (wn * 128 + (lane & 31)) * LINE;
    const int xrow = TILE_BYTES + (wm * 64 + (lane & 31)) * LINE;
    int coff[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        coff[kk][0] = (((kk * 2 + h)) ^ sw) * 16;
        coff[kk][1] = (((kk * 2 + h + 4)) ^ sw) * 16;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[it][jt][e] = 0.0f;

    const int nk = (p.debug & 2) ? 0 : p.K / 32;
    if (!(p.debug & 4)) issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        // one barrier per k-step: (a) this wave's and every other wave's DMA of stage kt has
        // landed (the compiler drains vmcnt before the barrier), (b) everybody is done reading
        // the other buffer, so the next stage may be streamed into it while we compute.
        __syncthreads();
        if (kt + 1 < nk && !(p.debug & 4)) issue(kt + 1, (kt + 1) & 1);
        const char* sb = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 whi[4], wlo[4], xhi[2], xlo[2];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                whi[it] = *(const half8*)(sb + wrow + it * 32 * LINE + coff[kk][0]);
                if (NSPLIT == 3) wlo[it] = *(const half8*)(sb + wrow + it * 32 * LINE + coff[kk][1]);
            }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                xhi[jt] = *(const half8*)(sb + xrow + jt * 32 * LINE + coff[kk][0]);
                if (NSPLIT == 3) xlo[jt] = *(const half8*)(sb + xrow + jt * 32 * LINE + coff[kk][1]);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    if (NSPLIT == 3) {
                        acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[it], xlo[jt], acc[it][jt], 0, 0, 0);
                        acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[it], xhi[jt], acc[it][jt], 0, 0, 0);
                    }
                    acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[it], xhi[jt], acc[it][jt], 0, 0, 0);
                }
        }
    }
    __syncthreads();  // all waves done with the stage buffers: LDS becomes epilogue scratch

    // ---- epilogue.  C/D layout of the 32x32 MFMA: lane holds column j = lane&31 (person) and rows
    // i = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (weight row): 4 consecutive n per register quad.
    // Each wave transposes its tile through a private 16 KiB LDS region (32 persons x 512 B, 16-B
    // chunks XOR-swizzled by the person index) and writes full 128-B lines with 16-B stores.
    if (p.debug & 1) {  // ablation: keep the accumulators live, store (almost) nothing
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[it][jt][e];
        if (s == 123456.789f) p.y[tid] = 1;
        return;
    }
    char* region = smem + w * 16384;
    const int ml = lane & 31;
    const size_t yrowb = (size_t)p.N * 4;
    const int nbase = n0 + wn * 128;  // first n of this wave
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const int mrow = m0 + wm * 64 + jt * 32;  // first person of this pass
        const size_t rbase = (size_t)(mrow + ml) * yrowb + (size_t)nbase * 4;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = it * 32 + g * 8 + h * 4;  // local n of element 0
                const f32x4 b4 = *(const f32x4*)(p.bias + nbase + nl);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = __builtin_fmaf(acc[it][jt][g * 4 + e], p.descale, b4[e]);
                    if (p.relu) v[e] = __builtin_fmaxf(v[e], 0.0f);
                }
                if (p.res) {
                    const char* rp = p.res + rbase + (size_t)(it * LINE + g * 16 + h * 8);
                    const half4 rh = *(const half4*)rp;
                    const half4 rl = *(const half4*)(rp + 64);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)rh[e] + (float)rl[e];
                }
                half4 oh, ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 a, b;
                    split_f16(v[e], a, b);
                    oh[e] = a;
                    ol[e] = b;
                }
                const int ci = it * 8 + g;  // 16-B chunk of the 512-B row (hi); lo = +4
                *(half4*)(region + ml * 512 + ((ci ^ ml) * 16) + h * 8) = oh;
                *(half4*)(region + ml * 512 + (((ci + 4) ^ ml) * 16) + h * 8) = ol;
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) {
            const int id = qq * 64 + lane;
            const int row = id >> 5, c = id & 31;
            const f32x4 d = *(const f32x4*)(region + row * 512 + ((c ^ row) * 16));
            *(f32x4*)(p.y + (size_t)(mrow + row) * yrowb + (size_t)nbase * 4 + c * 16) = d;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
#endif  // ML_BRINGUP

}  // namespace mlk
