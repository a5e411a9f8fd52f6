// dense_kernel.h -- what every dense kernel shares (operand format, tile constants, DenseParams, the fp16 split; the
// first-generation kernel that used to live here was the round-1 A/B reference and is gone): one Linear(+folded BatchNorm)(+ReLU)(+residual) layer of
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
    int ksplit = 1;     // dense_kernel_w4<.., -3>: work items per output tile, each reducing over K columns; dense_mid_kernel<.., SPLITK>:
                        // k ranges per output tile (the K / 32 steps divide evenly), partial tiles in kpart, tickets in kcount
    float* kpart = nullptr;       // [tiles][ksplit][128 x TM] fp32 partial tiles (dense_mid_kernel<.., SPLITK>)
    unsigned* kcount = nullptr;   // [tiles] arrival counters, zero between launches
    // dense_mid_kernel<.., PREP> (the mono pipeline's input layer): the workgroup pre-processes its own persons -- raw keypoints
    // (prep_m, 3, 17) fp32, rows 0 and 1 of inverse(K), z (10 m), optional box centres out (prep_m, 2); x is not read then
    const float* prep_kps = nullptr;
    float* prep_centre = nullptr;
    float prep_kinv[6] = {0, 0, 0, 0, 0, 0};
    float prep_z = 0.f;
    int prep_m = 0;
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


}  // namespace mlk
