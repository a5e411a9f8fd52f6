// train_mid.h -- the training step at the reference's REAL batch sizes (run.py:95 `--bs 512`, hyp_tuning.py:50 64..1024,
// loop body trainer.py:150-161): a few hundred to a few thousand rows, where the 256 x 256-tile kernels of the large-batch
// route keep a handful of CUs busy and the generic exact-fp32 GEMM needed ~175 launches per step.
//
// The "mid" route: every tensor stays plain row-major fp32 in HBM -- no packed images, no transposed copies, no scales.
//   xgemm_kernel<ALAY, BLAY>  C = A . B (+ bias) (+ res) on the EXACT fp32 matrix instruction (v_mfma_f32_32x32x2_f32 == an
//                      fmaf chain, 64 FLOP / clk / SIMD) with 32 x 64 workgroup tiles (176 .. 512 workgroups for a
//                      331 .. 512-row batch: every SIMD of the chip holds a wave), operands staged through LDS with
//                      full-line loads two k-steps ahead.  Each operand is either k-contiguous ([row][k]) or
//                      reduction-major ([k][row], read from LDS with a stride): forward z = y . W^T (k, k), data gradient
//                      dx = dz . W (k, reduction-major W as it lies), weight gradient dW = dz^T . x (both reduction-major:
//                      the batch is the reduction) run on the SAME row-major tensors.  A first version of this route
//                      (round 3, profiles/r03_*_v1.txt) used the 3-product fp16 scheme with operands split on the fly:
//                      the per-element conversion (3 VALU instructions at ~6 cycles) on every workgroup that re-reads an
//                      operand cost more than the 16 x slower matrix instruction does at these sizes, and it needed W^T
//                      and transposed activation copies (the transposing writes ran at 0.7 TB/s).
//   fwd_apply_kernel<NC>  a workgroup OWNS NC columns over ALL rows: batch statistics (exact, fixed order, no atomics),
//                      BatchNorm, ReLU, dropout, residual, in one launch behind the GEMM.  The input layer's narrow product
//                      runs inside it.  All of a thread's rows are requested before the first is used (one memory latency).
//   bwd_apply_kernel<NC>  the same ownership backwards: (dy from the output heads |) dropout / ReLU mask, both BatchNorm
//                      reductions, dz, bias / gamma / beta gradients.
//   gradnorm_kernel    the gradient norm from the per-workgroup partial sums the weight-gradient GEMMs leave (fixed order:
//                      deterministic) + the narrow tensors.
// Per-element arithmetic of BatchNorm, dropout, loss, clip and Adam is the exact route's (train_kernels.h).
#pragma once
#include <type_traits>

#include "train_kernels.h"

namespace mlt {

// ------------------------------------------------------------------------------------------------
struct XGemmParams {
    const float* a;      // ALAY 0: A(i, k) = a[i * lda + k]   (K % 32 == 0);  ALAY 1: A(i, k) = a[k * lda + i]  (M % 32 == 0, any K)
    const float* b;      // BLAY 0: B(j, k) = b[j * ldb + k];                  BLAY 1: B(j, k) = b[k * ldb + j]
    float* c;            // C(i, j) = sum_k A(i, k) B(j, k), row stride ldc
    const float* res;    // optional [M][N] (stride ldc) added to the result; may alias c
    const float* bias;   // optional [N]
    double* sumsq;       // optional: sum of squares of this workgroup's part of C -> sumsq[row tile * (N / 64) + column tile]
    long lda, ldb, ldc;
    int M, N, K;         // N % 64 == 0; rows beyond M are clamped on load and never stored
};

constexpr int XG_BM = 32, XG_BN = 64;
constexpr int XG_A_BYTES = XG_BM * 128, XG_STAGE = XG_A_BYTES + XG_BN * 128;

// One k32 step in LDS.  k-contiguous operand: a 128-byte row of 32 fp32 per tile row, its eight 16-byte chunks XOR-swizzled
// by (row >> 1) & 7 (conflict-free ds_read_b128 for lane -> (row = lane & 31, chunk pair by lane >> 5), conflict-free
// ds_write_b128 for 8 lanes per row).  Reduction-major operand: [32 k][tile rows] as it comes from memory; a lane reads its
// 8 k values with a stride (32 consecutive lanes = 32 consecutive banks).
// Waves: 2 (n halves of the 64 columns) x 2 (k16 halves of every k32 step); a wave issues 8 MFMAs per k-step: MFMA e
// contracts k = 16 wk + e (lanes 0..31) and k = 16 wk + 8 + e (lanes 32..63).  The two k halves meet in LDS at the end.
// (Round 3 also measured the 3-product fp16 scheme with the operands split in the consumer's registers -- the 8 fp32 values a
// lane holds per operand ARE the A / B register of v_mfma_f32_32x32x16_f16 for its k16 half -- at 20.0 us per 331 x 1024 x 1024
// product against 15.7 us for this kernel: 40 conversion instructions per wave and k-step cost more than the 16 x slower matrix
// instruction; profiles/r03_train_kernel_stats_rows331_f16x3_in_registers.txt.)
// (the kernel body: tile (bx, by) of a problem with nbx column tiles; smem = 2 stages, wsum = 4 doubles of LDS)
// NACC independent accumulators (round 4): accumulator (step parity, step half) takes every NACC-th group of four matrix
// instructions, so a K = 1024 reduction is NACC chains of 256 / NACC accumulations (2 products each) per k half instead of one of
// 256, summed at the end: the rounding error of a product falls with the square root of the chain length.  It matters for the
// FORWARD product: a pre-activation within rounding of zero flips its ReLU mask against the reference's, and every flip moves a
// BatchNorm bias gradient by ~1/m of its size (measured at 512 rows against the reference's fp64 run: gradient rms error 2.5-5.6e-4
// with one chain -- 2.3x the generic GEMM's 64-long split-K chains -- profiles/r04_train_parity.md).
template <int ALAY, int BLAY, int NACC = 1>
__device__ __forceinline__ void xgemm_tile(const XGemmParams& p, int bx, int by, int nbx, char* smem, double* wsum) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = w & 1, wk = w >> 1;
    const int m0 = by * XG_BM, n0 = bx * XG_BN;
    const int nk = (p.K + 31) / 32, last = nk - 1;
    // position i of the k loop -> k-step (positions past the end repeat the final one)
    auto seq = [&](int i) -> int { return i < last ? i : last; };

    // ---- loader: every thread issues 3 16-byte loads per k-step (1 of A, 2 of B), full 128 / 256-byte row segments per
    // 8 / 16 lanes; unconditional, same order every step.  Addresses = a wave-uniform base that moves with the k-step (SGPRs)
    // + one loop-invariant 32-bit lane offset: NO vector register is written for an address inside the loop (hipcc otherwise
    // computes it in the destination registers of the load, which it then believes busy: it waited for all but two of the
    // outstanding requests at the top of every step and the prefetch depth was gone).  Reduction-major operands are read up
    // to the end of the last k32 step: rows K .. ceil32(K) must be readable (the trainer's buffers are); their contribution
    // is zeroed in the fragments.
    unsigned a_off, b_off[2];
    int a_lds, b_lds[2];
    size_t a_step, b_step;   // bytes per k-step of the moving base
    if (ALAY == 0) {
        const int ra = tid >> 3, ca = tid & 7;
        int gr = m0 + ra;
        if (gr > p.M - 1) gr = p.M - 1;
        a_off = (unsigned)(((size_t)gr * p.lda + ca * 4) * 4);
        a_lds = ra * 128 + ((ca ^ ((ra >> 1) & 7)) * 16);
        a_step = 128;
    } else {
        const int ak = tid >> 3;
        a_off = (unsigned)(((size_t)ak * p.lda + m0 + (tid & 7) * 4) * 4);
        a_lds = ak * 128 + (tid & 7) * 16;
        a_step = (size_t)p.lda * 128;
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        if (BLAY == 0) {
            const int rb = (tid >> 3) + 32 * jj, cb = tid & 7;
            b_off[jj] = (unsigned)(((size_t)(n0 + rb) * p.ldb + cb * 4) * 4);
            b_lds[jj] = XG_A_BYTES + rb * 128 + ((cb ^ ((rb >> 1) & 7)) * 16);
        } else {
            const int bk = (tid >> 4) + 16 * jj;
            b_off[jj] = (unsigned)(((size_t)bk * p.ldb + n0 + (tid & 15) * 4) * 4);
            b_lds[jj] = XG_A_BYTES + bk * 256 + (tid & 15) * 16;
        }
    }
    b_step = (BLAY == 0) ? 128 : (size_t)p.ldb * 128;
    struct Raw {
        f32x4 a, b0, b1;
    };
    Raw R0, R1, R2, R3;
    auto gload = [&](Raw& R, int t) {
        const char* ab = (const char*)p.a + (size_t)t * a_step;   // wave-uniform
        const char* bb = (const char*)p.b + (size_t)t * b_step;
        R.a = *(const f32x4*)(ab + a_off);
        R.b0 = *(const f32x4*)(bb + b_off[0]);
        R.b1 = *(const f32x4*)(bb + b_off[1]);
    };
    auto lstore = [&](const Raw& R, int stage) {
        char* sbuf = smem + stage * XG_STAGE;
        *(f32x4*)(sbuf + a_lds) = R.a;
        *(f32x4*)(sbuf + b_lds[0]) = R.b0;
        *(f32x4*)(sbuf + b_lds[1]) = R.b1;
    };

    // ---- fragments: the 8 k values of this lane for both operands, read one k-step AHEAD of their MFMAs into a second
    // register set (the LDS latency and the barrier skew hide behind the 512 cycles the previous step's MFMA chain takes)
    const int ml = lane & 31, hh = lane >> 5;
    const int swf = (lane >> 1) & 7;
    const int kq = 16 * wk + 8 * hh;   // first of this lane's 8 k values within the step
    struct Frag {
        float a[8], b[8];
    };
    auto fragread = [&](Frag& F, int stage) {
        const char* sbuf = smem + stage * XG_STAGE;
        if (ALAY == 0) {
            const char* row = sbuf + ml * 128;
            const f32x4 v0 = *(const f32x4*)(row + (((kq >> 2)) ^ swf) * 16), v1 = *(const f32x4*)(row + (((kq >> 2) + 1) ^ swf) * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                F.a[e] = v0[e];
                F.a[4 + e] = v1[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) F.a[e] = *(const float*)(sbuf + (kq + e) * 128 + ml * 4);
        }
        if (BLAY == 0) {
            const char* row = sbuf + XG_A_BYTES + (tn * 32 + ml) * 128;
            const f32x4 v0 = *(const f32x4*)(row + (((kq >> 2)) ^ swf) * 16), v1 = *(const f32x4*)(row + (((kq >> 2) + 1) ^ swf) * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                F.b[e] = v0[e];
                F.b[4 + e] = v1[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) F.b[e] = *(const float*)(sbuf + XG_A_BYTES + (kq + e) * 256 + (tn * 32 + ml) * 4);
        }
    };
    f32x16 accs[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) accs[a][e] = 0.f;
    const bool ragged = (p.K & 31) != 0;
    // (uniform) the reduction's zero padding: k >= K of the last step -- both operands, whatever lies behind the matrices
    // (NaN included) contributes exactly 0 -- and the steps that only fill the loop up to a multiple of 4
    auto mask = [&](Frag& F, int i) {
        const int t = seq(i);
        if (i >= nk || (ragged && t == last)) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (i >= nk || t * 32 + kq + e >= p.K) {
                    F.a[e] = 0.f;
                    F.b[e] = 0.f;
                }
        }
    };
    auto mma4 = [&](const Frag& F, int e0, auto which) {   // `which`: compile-time accumulator index
        constexpr int A = decltype(which)::value;
#pragma unroll
        for (int e = 0; e < 4; ++e) accs[A] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a[e0 + e], F.b[e0 + e], accs[A], 0, 0, 0);
    };
    // Pipeline.  The operands come from HBM / the last-level cache for the first time (the producer ran on other XCDs): ~0.8 us
    // per request, against 0.25 us of MFMA time per k-step -- with the rows of only one further step in flight the loop ran at
    // 0.52 us per step whatever it computed (round 3, profiles/r03_*).  So FOUR register sets: at the top of step t, stage
    // t & 1 holds step t (its fragments already in registers), the other stage step t+1, and the sets hold the raw rows of
    // steps t+2 .. t+4, in flight.  A step reads the fragments of step t+1, requests step t+5 into the free set, multiplies
    // step t and moves step t+2 from its registers into the stage step t leaves (its fragment reads completed before the last
    // barrier).  Program order inside a step = issue order (an MFMA on the same accumulator blocks the wave's issue for the 64
    // cycles its predecessor runs, so whatever is to overlap with the chain sits BETWEEN its links): reads + requests, 4 MFMAs,
    // the LDS stores, 4 MFMAs, barrier (the last MFMA still runs).  Requests past the end repeat the final step; what they
    // bring is stored and read but never multiplied.
    Frag F0, F1;
    auto step = [&](Frag& Fc, Frag& Fn, Raw& Rl, Raw& Rs, auto scc, int i) {
        constexpr int sc = decltype(scc)::value;
        fragread(Fn, sc ^ 1);             // step i+1
        gload(Rl, seq(i + 5));
        mask(Fc, i);
        __builtin_amdgcn_sched_barrier(0);
        mma4(Fc, 0, std::integral_constant<int, (NACC >= 4 ? 2 * sc : 0)>());
        __builtin_amdgcn_sched_barrier(0);
        lstore(Rs, sc);                   // step i+2
        __builtin_amdgcn_sched_barrier(0);
        mma4(Fc, 4, std::integral_constant<int, (NACC >= 4 ? 2 * sc + 1 : (NACC >= 2 ? 1 : 0))>());
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    const std::integral_constant<int, 0> S0;
    const std::integral_constant<int, 1> S1;
    gload(R0, seq(0));
    gload(R1, seq(1));
    gload(R2, seq(2));
    gload(R3, seq(3));
    lstore(R0, 0);
    gload(R0, seq(4));
    lstore(R1, 1);
    __syncthreads();
    fragread(F0, 0);
    // Stage 0 is overwritten by the FIRST step (its LDS store of step 2) and, unlike every later stage hand-over, no step
    // barrier lies between that store and these reads: without this barrier a wave that was slow to issue them multiplied 8 rows
    // of step 2's A operand as step 0 -- never with one or two workgroups per CU (the store sits >= 4 MFMAs = 256 cycles behind
    // the reads), 1 - 8 wrong tiles per 1024 with four (profiles/r03_xgemm_occupancy.md).
    __syncthreads();
    for (int i = 0; i < nk; i += 4) {   // whole groups of 4 steps (one exit: no register shuffling between the sets)
        step(F0, F1, R1, R2, S0, i);
        step(F1, F0, R2, R3, S1, i + 1);
        step(F0, F1, R3, R0, S0, i + 2);
        step(F1, F0, R0, R1, S1, i + 3);
    }
    __syncthreads();
    f32x16 acc = accs[0];
    if (NACC == 2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = accs[0][e] + accs[NACC - 1][e];
    } else if (NACC >= 4) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = (accs[0][e] + accs[1 % NACC][e]) + (accs[2 % NACC][e] + accs[3 % NACC][e]);
    }

    // ---- the k halves meet: waves 2, 3 hand their partial tile to waves 0, 1
    float* red = (float*)smem;   // [2][16][64]
    if (wk == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(tn * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    double ss = 0.0;
    if (wk == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[(tn * 16 + r) * 64 + lane];
        // D layout of the 32 x 32 MFMA: lane holds column j = lane & 31 (a B row) and rows i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        // (A rows): for a fixed r the lanes 0..31 write 128 contiguous bytes of a row of C
        const int j = n0 + tn * 32 + ml;
        const float bj = p.bias ? p.bias[j] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = m0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (i < p.M) {
                float v = acc[r] + bj;
                const size_t o = (size_t)i * p.ldc + j;
                if (p.res) v += p.res[o];
                p.c[o] = v;
                ss += (double)v * (double)v;
            }
        }
    }
    if (p.sumsq) {   // (uniform) fixed-order sum over the workgroup: lanes by butterfly, then the two finishing waves
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if (lane == 0) wsum[w] = ss;
        __syncthreads();
        if (tid == 0) p.sumsq[(size_t)by * nbx + bx] = wsum[0] + wsum[1];
    }
}

// accumulators per layout pair: the forward product (k-contiguous x k-contiguous) 4, the data gradient 2, the weight gradient
// (its reduction is the batch: 11 .. 16 k-steps at the reference's batch sizes) 1 -- see xgemm_tile
template <int ALAY, int BLAY>
constexpr int xg_nacc() { return (ALAY == 0 && BLAY == 0) ? 4 : ((ALAY == 0 && BLAY == 1) ? 2 : 1); }

template <int ALAY, int BLAY>
__global__ __launch_bounds__(256) void xgemm_kernel(XGemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * XG_STAGE];
    __shared__ double wsum[4];
    xgemm_tile<ALAY, BLAY, xg_nacc<ALAY, BLAY>()>(p, blockIdx.x, blockIdx.y, gridDim.x, smem, wsum);
}

// The two products that read the same dz -- the data gradient dx = dz . W (+ res) (k-contiguous x reduction-major) and the
// weight gradient dW = dz^T . x (both reduction-major) -- in ONE launch: workgroups [0, nd) are the data gradient's tiles
// (dispatched first: 64 k-steps each at hidden 1024 against 11 at 331 rows), the rest the weight gradient's.  One kernel floor
// (~4.5 us) instead of two per Linear, and the chip sees 176 + 512 workgroups at once instead of 176, then 512.
__global__ __launch_bounds__(256) void xgemm_pair_kernel(XGemmParams pd, XGemmParams pw, int nd, int ndx, int nwx) {
    __shared__ __attribute__((aligned(16))) char smem[2 * XG_STAGE];
    __shared__ double wsum[4];
    const int id = blockIdx.x;   // (uniform)
    if (id < nd) xgemm_tile<0, 1, xg_nacc<0, 1>()>(pd, id % ndx, id / ndx, ndx, smem, wsum);
    else xgemm_tile<1, 1, xg_nacc<1, 1>()>(pw, (id - nd) % nwx, (id - nd) / nwx, nwx, smem, wsum);
}

// ------------------------------------------------------------------------------------------------
// Forward of one block Linear -> BatchNorm1d (train mode) -> ReLU -> Dropout (-> + residual), everything behind the GEMM
// (reference architectures.py:50-52, 90-100).  One workgroup per NC columns, ALL rows: thread = (row, 4 columns); a thread
// requests its up to AP_PMAX rows before using the first (larger batches walk in chunks of that and re-read).  Column sums
// in fp64: per thread over its rows, butterfly over the lanes that share the columns, then the four waves in order.
// Input-layer mode (x_in != null): z = x_in . w_in^T + b_in is computed here (train_kernels.h' skinny_out_kernel's fma
// order) and written to z_out.
struct FwdApplyParams {
    const float* z;
    const float* x_in;
    const float* w_in;
    const float* b_in;
    float* z_out;
    int in_dim;
    long m;
    int H;
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
    int eval;            // 1: evaluation mode -- the running statistics instead of the batch's (nothing is updated), no dropout
};

// rows a workgroup of the column-owner kernels holds in registers at once (512: the reference's default batch); per thread
// 512 / (rows per pass) = NC / 2 rows of 4 columns
constexpr int AP_CHUNK = 512;

// sums of v[q][0..4) over the threads of the workgroup that own the same 4 columns (lanes == cq mod TPR), fixed order:
// butterfly inside each wave, then red[q][wave][column]; the caller adds the four waves after a barrier
template <int NC, int NV>
__device__ __forceinline__ void column_reduce(double (&v)[NV][4], double (*red)[4][NC], int tid) {
    constexpr int TPR = NC / 4;
    const int lane = tid & 63, wv = tid >> 6, cq = tid % TPR;
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double s = v[q][e];
#pragma unroll
            for (int o = TPR; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
            v[q][e] = s;
        }
    if (lane < TPR) {
#pragma unroll
        for (int q = 0; q < NV; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[q][wv][cq * 4 + e] = v[q][e];
    }
}

template <int NC>
__global__ __launch_bounds__(256) void fwd_apply_kernel(FwdApplyParams p) {
    constexpr int TPR = NC / 4, RPP = 256 / TPR, AP_PMAX = AP_CHUNK / RPP;
    __shared__ double red[2][4][NC];
    __shared__ float stat[2][NC];
    __shared__ float wl[NC * SK_NC];
    const int tid = threadIdx.x, cq = tid % TPR, r = tid / TPR;
    const int j0 = blockIdx.x * NC, j = j0 + cq * 4;
    const int H = p.H;
    const bool inl = p.x_in != nullptr;
    if (inl) {
        for (int idx = tid; idx < NC * p.in_dim; idx += 256) wl[idx] = p.w_in[(size_t)j0 * p.in_dim + idx];
        __syncthreads();
    }
    const float* zsrc = inl ? p.z_out : p.z;
    const long chunk = (long)RPP * AP_PMAX;
    const bool single = p.m <= chunk;
    // everything that does not depend on the statistics is requested NOW (each global load costs ~1 us here: the producer ran
    // on other XCDs; three of them used to stand one after the other behind the reduction)
    const f32x4 ga = *(const f32x4*)(p.gamma + j), be = *(const f32x4*)(p.beta + j);
    float rmean = 0.f, rvar = 0.f;
    if (tid < NC) {
        rmean = p.run_mean[j0 + tid];
        rvar = p.run_var[j0 + tid];
    }
    f32x4 rr[AP_PMAX];
#pragma unroll
    for (int q = 0; q < AP_PMAX; ++q) {
        const long i = q * RPP + r;
        rr[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (single && p.residual && i < p.m) rr[q] = *(const f32x4*)(p.residual + i * H + j);
    }
    f32x4 zr[AP_PMAX];
    double s[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    for (long base = 0; base < p.m && !(p.eval && !inl && !single); base += chunk) {   // (evaluation of a large batch: pass 2 reads z)
#pragma unroll
        for (int q = 0; q < AP_PMAX; ++q) {
            const long i = base + q * RPP + r;
            zr[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < p.m) {
                if (inl) {
                    f32x4 v = *(const f32x4*)(p.b_in + j);
                    const float* xr = p.x_in + i * p.in_dim;
                    for (int c = 0; c < p.in_dim; ++c) {
                        const float xv = xr[c];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(xv, wl[(cq * 4 + e) * p.in_dim + c], v[e]);
                    }
                    *(f32x4*)(p.z_out + i * H + j) = v;
                    zr[q] = v;
                } else {
                    zr[q] = *(const f32x4*)(p.z + i * H + j);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < AP_PMAX; ++q) {
            if (base + q * RPP + r < p.m) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[0][e] += (double)zr[q][e];
                    s[1][e] += (double)zr[q][e] * (double)zr[q][e];
                }
            }
        }
    }
    if (!p.eval) column_reduce<NC, 2>(s, red, tid);
    __syncthreads();
    if (p.eval) {   // (uniform) nn.BatchNorm1d in eval mode: (z - running_mean) / sqrt(running_var + eps), fp32
        if (tid < NC) {
            stat[0][tid] = rmean;
            stat[1][tid] = 1.0f / sqrtf(rvar + 1e-5f);
        }
    } else if (tid < NC) {
        const double sa = (red[0][0][tid] + red[0][1][tid]) + (red[0][2][tid] + red[0][3][tid]);
        const double sb = (red[1][0][tid] + red[1][1][tid]) + (red[1][2][tid] + red[1][3][tid]);
        // bn_finalize_kernel's arithmetic (train_kernels.h)
        const double mu = sa / (double)p.m;
        double var = sb / (double)p.m - mu * mu;
        if (var < 0) var = 0;
        const float mean = (float)mu, inv = (float)(1.0 / sqrt(var + 1e-5));
        const double unb = p.m > 1 ? var * (double)p.m / (double)(p.m - 1) : var;
        const int jj = j0 + tid;
        p.mean_out[jj] = mean;
        p.invstd_out[jj] = inv;
        p.run_mean[jj] = (1.f - 0.1f) * rmean + 0.1f * (float)mu;
        p.run_var[jj] = (1.f - 0.1f) * rvar + 0.1f * (float)unb;
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
    for (long base = 0; base < p.m; base += chunk) {
        if (!single) {
#pragma unroll
            for (int q = 0; q < AP_PMAX; ++q) {
                const long i = base + q * RPP + r;
                rr[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (i < p.m) {
                    zr[q] = *(const f32x4*)(zsrc + i * H + j);
                    if (p.residual) rr[q] = *(const f32x4*)(p.residual + i * H + j);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < AP_PMAX; ++q) {
            const long i = base + q * RPP + r;
            if (i < p.m) {
                f32x4 v = zr[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {   // bn_relu_drop_kernel's arithmetic
                    float t = ga[e] * ((v[e] - mu[e]) * is[e]) + be[e];
                    t = t > 0.f ? t : 0.f;
                    if (p.p_drop > 0.f && !p.eval)
                        t = (mlk::u01(p.seed, (uint32_t)i * 4099u + p.site, (uint32_t)(j + e)) >= p.p_drop) ? t / (1.f - p.p_drop) : 0.f;
                    v[e] = t;
                }
                if (p.residual) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rr[q][e];
                }
                *(f32x4*)(p.y + i * H + j) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the same block, behind the data-gradient GEMM that produced dy (the gradient wrt the block's output):
//   dy   = incoming * [gamma * xhat + beta > 0] * dropout_mask / (1 - p),   xhat = (z - mean) * invstd      (bwd_elem)
//   dz   = gamma * invstd / m * (m * dy - sum(dy) - xhat * sum(dy * xhat))                                  (bn_bwd_fused_kernel)
//   dgamma = sum(dy * xhat), dbeta = sum(dy), dbias (of the Linear) = sum(dz)
// The incoming gradient comes
//   * from dy (a buffer), optionally + aux_d[i] * w_aux[j] (the one-output head's data gradient, skinny_out_kernel's
//     accumulate form), or
//   * from the output heads: sum_c dout[i][c] * w_head[c][j]  (skinny_out_kernel's fma order).
// z == null: a Linear without BatchNorm (w2): dz = dy.
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
    float* dz;
    float* dgamma;
    float* dbeta;
    float* dbias;
    // narrow weight gradients that ride along (each optional):
    //  * of an output head that reads this block's forward activation ysrc (m x H):  dwh[c][j] = sum_i sc[i * sld + c] * ysrc[i][j],
    //    c < ns <= 9 (w_fin: sc = dout, ysrc = y3; w_aux: sc = dout + C - 1, ysrc = y2)   (train_kernels.h: skinny_dw_kernel)
    //  * the head biases (workgroup 0): hb0[c] = sum_i dout[i][c], c < hb_n0;  hb1[0] = sum_i dout[i][hb_n0]
    const float* sc;
    int sld, ns;
    const float* ysrc;
    float* dwh;
    const float* hb_src;
    int hb_ld, hb_n0;
    float* hb0;
    float* hb1;
};

template <int NC>
__global__ __launch_bounds__(256) void bwd_apply_kernel(BwdApplyParams p) {
    constexpr int TPR = NC / 4, RPP = 256 / TPR, AP_PMAX = AP_CHUNK / RPP;
    __shared__ double red[2][4][NC];
    __shared__ float stat[2][NC];
    __shared__ float wh[16 * NC];   // head weights of these columns, [c][NC]
    __shared__ float redf[9][4][NC];
    const int tid = threadIdx.x, cq = tid % TPR, r = tid / TPR;
    const int j0 = blockIdx.x * NC, j = j0 + cq * 4;
    const int H = p.H;
    const bool bn = p.z != nullptr;
    if (p.dout) {
        for (int idx = tid; idx < p.nc * NC; idx += 256) wh[idx] = p.w_head[(size_t)(idx / NC) * H + j0 + (idx % NC)];
        __syncthreads();
    }
    const long chunk = (long)RPP * AP_PMAX;
    const bool single = p.m <= chunk;
    f32x4 dr[AP_PMAX], zr[AP_PMAX];
    // incoming gradient and pre-activation of up to AP_PMAX rows of this thread: every load is requested before the first use
    auto fetch = [&](long base) {
#pragma unroll
        for (int q = 0; q < AP_PMAX; ++q) {
            const long i = base + q * RPP + r;
            dr[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            zr[q] = dr[q];
            if (i < p.m) {
                if (bn) zr[q] = *(const f32x4*)(p.z + i * H + j);
                if (!p.dout) dr[q] = *(const f32x4*)(p.dy + i * H + j);
            }
        }
        if (p.dout) {
#pragma unroll
            for (int q = 0; q < AP_PMAX; ++q) {
                const long i = base + q * RPP + r;
                if (i < p.m) {
                    const float* drow = p.dout + i * p.dld;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    for (int c = 0; c < p.nc; ++c) {
                        const float sv = drow[c];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(sv, wh[c * NC + cq * 4 + e], v[e]);
                    }
                    dr[q] = v;
                }
            }
        }
        if (p.aux_d) {
            const f32x4 wa = *(const f32x4*)(p.w_aux + j);
#pragma unroll
            for (int q = 0; q < AP_PMAX; ++q) {
                const long i = base + q * RPP + r;
                if (i < p.m) {
                    const float sv = p.aux_d[i * p.aux_ld];
#pragma unroll
                    for (int e = 0; e < 4; ++e) dr[q][e] = __builtin_fmaf(sv, wa[e], 0.f) + dr[q][e];
                }
            }
        }
    };
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = mu, ga = mu, be = mu;
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f}, gg[4] = {0.f, 0.f, 0.f, 0.f};
    if (single) fetch(0);
    f32x4 yv[AP_PMAX];   // rows of the forward activation a riding head-weight gradient needs: requested with everything else
#pragma unroll
    for (int q = 0; q < AP_PMAX; ++q) {
        const long i = q * RPP + r;
        yv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (single && p.dwh && i < p.m) yv[q] = *(const f32x4*)(p.ysrc + i * H + j);
    }
    if (bn) {
        mu = *(const f32x4*)(p.mean + j);
        is = *(const f32x4*)(p.invstd + j);
        ga = *(const f32x4*)(p.gamma + j);
        be = *(const f32x4*)(p.beta + j);
        double s[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        for (long base = 0; base < p.m; base += chunk) {
            if (!single) fetch(base);
#pragma unroll
            for (int q = 0; q < AP_PMAX; ++q) {
                const long i = base + q * RPP + r;
                if (i < p.m) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float dy, xh;
                        bwd_elem(dr[q][e], zr[q][e], mu[e], is[e], ga[e], be[e], p.p_drop, p.seed, p.site, i, j + e, dy, xh);
                        s[0][e] += (double)dy;
                        s[1][e] += (double)dy * (double)xh;
                    }
                }
            }
        }
        column_reduce<NC, 2>(s, red, tid);
        __syncthreads();
        if (tid < NC) {
            const double s1 = (red[0][0][tid] + red[0][1][tid]) + (red[0][2][tid] + red[0][3][tid]);
            const double s2 = (red[1][0][tid] + red[1][1][tid]) + (red[1][2][tid] + red[1][3][tid]);
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
    double s2[1][4] = {{0.0, 0.0, 0.0, 0.0}};
    for (long base = 0; base < p.m; base += chunk) {
        if (!single) fetch(base);
#pragma unroll
        for (int q = 0; q < AP_PMAX; ++q) {
            const long i = base + q * RPP + r;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (i < p.m) {
                o = dr[q];
                if (bn) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float dy, xh;
                        bwd_elem(dr[q][e], zr[q][e], mu[e], is[e], ga[e], be[e], p.p_drop, p.seed, p.site, i, j + e, dy, xh);
                        o[e] = gg[e] * ((float)p.m * dy - sa[e] - xh * sb[e]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) s2[0][e] += (double)o[e];
                *(f32x4*)(p.dz + i * H + j) = o;
            }
        }
    }
    __syncthreads();   // (red is read above by the threads tid < NC)
    column_reduce<NC, 1>(s2, red, tid);
    __syncthreads();
    if (tid < NC) p.dbias[j0 + tid] = (float)((red[0][0][tid] + red[0][1][tid]) + (red[0][2][tid] + red[0][3][tid]));

    // ---- a head's weight gradient from this block's forward activation (fp32 fma per thread over its rows, then the fixed
    // order reduction over the threads that share the columns)
    if (p.dwh) {   // (uniform)
        float acc[9][4];
#pragma unroll
        for (int c = 0; c < 9; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[c][e] = 0.f;
        for (long base = 0; base < p.m; base += chunk) {
            if (!single) {
#pragma unroll
                for (int q = 0; q < AP_PMAX; ++q) {
                    const long i = base + q * RPP + r;
                    yv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (i < p.m) yv[q] = *(const f32x4*)(p.ysrc + i * H + j);
                }
            }
#pragma unroll
            for (int q = 0; q < AP_PMAX; ++q) {
                const long i = base + q * RPP + r;
                if (i < p.m) {
                    const float* srow = p.sc + i * p.sld;
#pragma unroll
                    for (int c = 0; c < 9; ++c) {
                        if (c < p.ns) {
                            const float sv = srow[c];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[c][e] = __builtin_fmaf(sv, yv[q][e], acc[c][e]);
                        }
                    }
                }
            }
        }
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int c = 0; c < 9; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[c][e];
#pragma unroll
                for (int o = TPR; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
                if (lane < TPR) redf[c][wv][cq * 4 + e] = v;
            }
        __syncthreads();
        for (int idx = tid; idx < p.ns * NC; idx += 256) {
            const int c = idx / NC, col = idx - c * NC;
            p.dwh[(size_t)c * H + j0 + col] = (redf[c][0][col] + redf[c][1][col]) + (redf[c][2][col] + redf[c][3][col]);
        }
    }
    // ---- the head biases: column sums of dout (workgroup 0; one thread per row walks it, fp64)
    if (p.hb0 && blockIdx.x == 0) {
        __shared__ double redb[16][4];
        double hs[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) hs[c] = 0.0;
        for (long i = tid; i < p.m; i += 256) {
            const float* srow = p.hb_src + i * p.hb_ld;
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c <= p.hb_n0) hs[c] += (double)srow[c];
        }
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            double v = hs[c];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) redb[c][wv] = v;
        }
        __syncthreads();
        if (tid <= p.hb_n0) {
            const float v = (float)((redb[tid][0] + redb[tid][1]) + (redb[tid][2] + redb[tid][3]));
            if (tid < p.hb_n0) p.hb0[tid] = v;
            else p.hb1[0] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Both output heads + the multi-task loss + its gradient in ONE launch (train_kernels.h: skinny_heads_kernel<1>,
// skinny_heads_kernel<8|9>, loss_kernel, and the memset their atomics needed): a wave per row computes the C dot products
// (w_fin on y3, w_aux on y2; skinny_heads_kernel's fma order and butterfly), lane 0 evaluates loss_row and writes the raw row
// and dLoss/dout; the per-workgroup sums of the LOSS_NV row terms go to part[blockIdx.x][.] / m -- the host adds the few
// workgroups in order (no atomics, no zeroing).  dout == null: values only (evaluation).
struct HeadsLossParams {
    const float* y3;
    const float* y2;
    const float* w_fin;   // [C-1][H]
    const float* b_fin;
    const float* w_aux;   // [H]
    const float* b_aux;
    const float* lab;
    int L;
    long m;
    int H;
    float* out;           // [m][C]
    float* dout;          // [m][C] or null
    const float* tw;      // 8 task weights or null
    double* part;         // [gridDim.x][LOSS_NV]
};
constexpr int HL_MAXW = 15360;   // floats of head weights a workgroup stages in LDS (C * H <= HL_MAXW)

template <int C>
__global__ __launch_bounds__(256) void heads_loss_kernel(HeadsLossParams p) {
    __shared__ __attribute__((aligned(16))) float wl[HL_MAXW];
    __shared__ double wred[4][LOSS_NV];
    const int n = p.H;
    for (int idx = threadIdx.x; idx < (C - 1) * n; idx += 256) wl[idx] = p.w_fin[idx];
    for (int idx = threadIdx.x; idx < n; idx += 256) wl[(C - 1) * n + idx] = p.w_aux[idx];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float w8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) w8[q] = p.tw ? p.tw[q] : 1.f;
    double tsum[LOSS_NV];
#pragma unroll
    for (int q = 0; q < LOSS_NV; ++q) tsum[q] = 0.0;
    for (long i = (long)blockIdx.x * 4 + wv; i < p.m; i += (long)gridDim.x * 4) {
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.f;
        for (int j = lane * 4; j < n; j += 256) {
            const f32x4 v3 = *(const f32x4*)(p.y3 + i * n + j), v2 = *(const f32x4*)(p.y2 + i * n + j);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const f32x4 ww = *(const f32x4*)&wl[c * n + j];
                const f32x4 v = (c == C - 1) ? v2 : v3;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[c] = __builtin_fmaf(v[e], ww[e], acc[c]);
            }
        }
        float o[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float a = acc[c];
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) a += __shfl_xor(a, s, 64);
            o[c] = a + (c == C - 1 ? p.b_aux[0] : p.b_fin[c]);
        }
        if (lane == 0) {
            float g[C];
            double t[LOSS_NV];
            loss_row(o, p.lab + i * p.L, C, p.m, w8, p.dout ? g : nullptr, t);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                p.out[i * C + c] = o[c];
                if (p.dout) p.dout[i * C + c] = g[c];
            }
#pragma unroll
            for (int q = 0; q < LOSS_NV; ++q) tsum[q] += t[q];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < LOSS_NV; ++q) wred[wv][q] = tsum[q];
    }
    __syncthreads();
    if (threadIdx.x < LOSS_NV)
        p.part[(size_t)blockIdx.x * LOSS_NV + threadIdx.x] =
            ((wred[0][threadIdx.x] + wred[1][threadIdx.x]) + (wred[2][threadIdx.x] + wred[3][threadIdx.x])) / (double)p.m;
}

// ------------------------------------------------------------------------------------------------
// Gradient norm^2 for clip_grad_norm_: the weight-gradient GEMMs leave one partial sum of squares per workgroup (nslots
// doubles); the narrow tensors (everything that is not an H x H matrix: `segs`) are summed here.  Fixed order at both
// levels: the same gradients always give the same norm (the exact route's sumsq_kernel adds with atomics).
constexpr int MID_MAXMAT = 40;
struct AdamSegs {
    long off[MID_MAXMAT];         // flat offset of each segment
    long start[MID_MAXMAT + 1];   // prefix sums of the segment lengths
    int count;
};

constexpr int GN_PARTS = 64;
// level 1: GN_PARTS workgroups, each a strided share of the slots and of every narrow segment -> part[blockIdx.x]
__global__ __launch_bounds__(256) void gradnorm_kernel(const float* __restrict__ g, AdamSegs segs, const double* __restrict__ slots, int nslots,
                                                       double* __restrict__ part) {
    __shared__ double red[256];
    const int gt = blockIdx.x * 256 + threadIdx.x, gn = GN_PARTS * 256;
    double a = 0.0;
    for (int i = gt; i < nslots; i += gn) a += slots[i];
    for (int s = 0; s < segs.count; ++s) {
        const float* gs = g + segs.off[s];
        const long n = segs.start[s + 1] - segs.start[s];
        for (long i = gt; i < n; i += gn) a += (double)gs[i] * (double)gs[i];
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// level 2 inside the optimizer: every workgroup adds the GN_PARTS partial sums in the same order.  clip_adam_kernel's arithmetic.
__global__ __launch_bounds__(256) void clip_adam_parts_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m1,
                                                             float* __restrict__ m2, int64_t n, const double* __restrict__ part,
                                                             float max_norm, float lr, float b1, float b2, float eps, float bc1,
                                                             float bc2, int do_adam) {
    __shared__ double tot;
    if (threadIdx.x < 64) {
        double v = part[threadIdx.x];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (threadIdx.x == 0) tot = v;
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float norm = (float)sqrt(tot);
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
