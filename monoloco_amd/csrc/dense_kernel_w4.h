// dense_kernel_w4.h -- third generation of the dense layer: 4 waves per workgroup, ONE wave per SIMD with the whole
// 512-register file (256 accumulators in AGPRs + 192 fragment registers), 128 x 128 wave tiles, a 4-slot LDS ring fed by
// LDS-DMA three slots ahead, and a k-stream that never restarts between the tiles a persistent workgroup walks.
//
// Why (round-2 measurements, profiles/r02_*): in dense_kernel_pp the bf16 comparison mode -- one third of the MFMA work --
// still needs 0.245 ms per layer against 0.355 ms for the 3-product mode: a k-step costs ~3800 cycles whatever the MFMA
// count.  The loop is bound by the L2 -> LDS feed LATENCY, not by MFMA issue: with two 64 KiB stages only one stage is
// ever in flight and its last quarter is requested < 1000 cycles before it is needed.  This kernel
//   * splits a k32 step of the line format into an H slot (the 64-byte hi halves of 256 W rows + 256 X rows = 32 KiB) and
//     an L slot (the lo halves): 4 slots = 128 KiB ring = TWO k-steps; a slot is requested 4000-5000 cycles before its
//     first read and ~96 KiB per CU are in flight at any time (was: 64 KiB sawtooth, average 32);
//   * reads fragments one phase ahead into a second register set, so a wave never waits for LDS in front of an MFMA and
//     the slot a phase has just read is free for the next request at the following barrier;
//   * runs the three products of the split precision as  phase A: hi.hi (32 MFMAs, needs only the H slot)  and
//     phase B: hi.lo + lo.hi (64 MFMAs, needs the L slot and the hi fragments A already holds) -- per k32 step a wave
//     issues 96 MFMAs, 32 ds_read_b128 (12 -> 8 fragment reads per 24 MFMAs), 16 LDS-DMA instructions and 2 barriers;
//   * 128 x 128 wave tiles: half the LDS read traffic per MFMA of the 128 x 64 tiles of dense_kernel_pp;
//   * keeps streaming across tile boundaries: the last phases of a tile already request (and read the first fragments
//     of) the next tile, so there is no per-tile prologue and the epilogue's store burst overlaps the next tile's loads.
// Same math, operand format, XCD-aware tile map and epilogue arithmetic as dense_kernel_pp; the fp32 summation order
// inside a k32 step differs (hh k0, hh k1, [hl, lh] k0, [hl, lh] k1 instead of [hl, lh, hh] per k16).
//
// vmcnt discipline: LDS-DMA and stores retire in issue order on one counter (CDNA4 vmcnt counts stores too); every wait
// is a counted one placed by hand:  phase start = "the slot read in THIS phase has landed" = at most the DMA groups of the
// previous two phases (8 instructions each) still outstanding.  No ordinary vector load exists between the first DMA and
// the drain in front of an epilogue (hipcc would wait vmcnt(0) for it); the bias comes through the scalar cache.
#pragma once
#include "dense_kernel_pp.h"

// timing ablations are COMPILE-TIME (-DML_W4_ABL=<bits>): a run-time debug branch per DMA / fragment read distorts exactly
// what is being measured.  1 no epilogue, 4 no DMA in the loop, 8 no fragment reads in the loop, 32 no barrier / vmcnt
// wait per phase, 64 epilogue without global stores, 128 epilogue stores to one L2-resident tile, 256 epilogue stores tile-contiguous (results are garbage)
#ifndef ML_W4_ABL
#define ML_W4_ABL 0
#endif
#define W4_DBG(bit) (((ML_W4_ABL) & (bit)) != 0)
// swizzle of the per-wave epilogue scratch (32 rows x 128 B per pass): the 16-byte chunk of row r is XOR-ed with
//   0: r & 7        (rounds 2-4) -- the 8-byte MFMA-layout accesses of rows r, r+8, r+16, r+24 meet in the same banks (4-way)
//   1: (r >> 1) & 7 -- rows that share banks (same r & 1: a row is 32 of the 64 banks) spread over all 8 chunks (2-way: the floor
//                      for 8-byte accesses that use one half of every chunk); the 16-byte store-layout side is a permutation either way
#ifndef ML_W4_ESWZ
#define ML_W4_ESWZ 1
#endif

namespace mlk {

constexpr int W4_THREADS = 256;
constexpr int W4_SLOT = 32768;                 // 256 W rows x 64 B, then 256 X rows x 64 B
constexpr int W4_XOFF = 16384;
constexpr int W4_RING = 4 * W4_SLOT;           // H(t), L(t), H(t+1), L(t+1)
constexpr int W4_LDS = W4_RING + 4 * 8192;     // + 2 x 4 KiB epilogue buffers per wave = all 160 KiB

// one 32x32x16 MFMA of the mode: fp16 operands, or bf16 (NSPLIT == 0)
template <int NSPLIT>
__device__ __forceinline__ f32x16 w4_mfma(half8 a, half8 b, f32x16 c) {
    if (NSPLIT == 0)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// fp32 pair -> packed fp16 hi pair + packed fp16 lo pair, hi clamped to the fp16 range (same bits as split2_res' tail)
__device__ __forceinline__ void w4_split2(float v0, float v1, unsigned& h, unsigned& l) {
    const float big = 65504.0f;
    asm("v_med3_f32 %0, %1, -%2, %2" : "=v"(v0) : "v"(v0), "v"(big));
    asm("v_med3_f32 %0, %1, -%2, %2" : "=v"(v1) : "v"(v1), "v"(big));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(v0), "v"(v1));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(v1), "v"(h));
}

// 8 consecutive floats through the scalar cache (s_load_dwordx8): the bias of one (row block, register group) for both
// lane halves.  An ordinary vector load here would make hipcc drain vmcnt(0) and break the epilogue's counted waits.
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x8 w4_sload8(const float* base_uniform, int byte_off_uniform) {
    f32x8 v;
    asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(base_uniform), "s"(byte_off_uniform) : "memory");
    return v;
}

// Epilogue arithmetic of one value PAIR in ONE asm statement each (hipcc pads every asm statement with an s_nop and knows
// nothing about the AGPR reads; 38 s_nop per 32x32 pass with one statement per instruction).  The accumulators start at
// bias * 2^e, so v = acc * 2^-e is exact and needs no bias add; results are bit-identical to dense_kernel_pp's
// split2_scaled / split2_res (same instructions on the same values).
//   plain:    c = med3(acc, lo_lim, lim) [ReLU and the fp16-range clamp in the accumulator's scale];
//             h = packed rn_f16(c * d);  l = packed rn_f16(c * d - h)
__device__ __forceinline__ void w4_pair_plain(float a0, float a1, float d, float lo_lim, float lim, unsigned& h, unsigned& l) {
    float t0, t1;
    asm("v_accvgpr_read_b32 %2, %4\n\t"
        "v_accvgpr_read_b32 %3, %5\n\t"
        "v_med3_f32 %2, %2, %6, %7\n\t"
        "v_med3_f32 %3, %3, %6, %7\n\t"
        "v_fma_mixlo_f16 %0, %2, %8, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %8, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %8, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l), "=&v"(t0), "=&v"(t1)
        : "a"(a0), "a"(a1), "v"(lo_lim), "v"(lim), "v"(d));
}
//   residual: v = relu(acc) * d + (rh + rl), clamped to the fp16 range; h = rn_f16(v); l = rn_f16(v - h)
template <bool RELU>
__device__ __forceinline__ void w4_pair_res(float a0, float a1, float d, unsigned rh, unsigned rl, unsigned& h, unsigned& l) {
    float t0, t1, r0, r1;
    const float big = 65504.0f;
    if (RELU)
        asm("v_accvgpr_read_b32 %2, %6\n\t"
            "v_accvgpr_read_b32 %3, %7\n\t"
            "v_fma_mix_f32 %4, %8, 1.0, %9 op_sel_hi:[1,0,1]\n\t"
            "v_fma_mix_f32 %5, %8, 1.0, %9 op_sel:[1,0,1] op_sel_hi:[1,0,1]\n\t"
            "v_max_f32 %2, 0, %2\n\t"
            "v_max_f32 %3, 0, %3\n\t"
            "v_fma_f32 %2, %2, %10, %4\n\t"
            "v_fma_f32 %3, %3, %10, %5\n\t"
            "v_med3_f32 %2, %2, -%11, %11\n\t"
            "v_med3_f32 %3, %3, -%11, %11\n\t"
            "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
            "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(h), "=&v"(l), "=&v"(t0), "=&v"(t1), "=&v"(r0), "=&v"(r1)
            : "a"(a0), "a"(a1), "v"(rh), "v"(rl), "v"(d), "v"(big));
    else
        asm("v_accvgpr_read_b32 %2, %6\n\t"
            "v_accvgpr_read_b32 %3, %7\n\t"
            "v_fma_mix_f32 %4, %8, 1.0, %9 op_sel_hi:[1,0,1]\n\t"
            "v_fma_mix_f32 %5, %8, 1.0, %9 op_sel:[1,0,1] op_sel_hi:[1,0,1]\n\t"
            "v_fma_f32 %2, %2, %10, %4\n\t"
            "v_fma_f32 %3, %3, %10, %5\n\t"
            "v_med3_f32 %2, %2, -%11, %11\n\t"
            "v_med3_f32 %3, %3, -%11, %11\n\t"
            "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
            "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(h), "=&v"(l), "=&v"(t0), "=&v"(t1), "=&v"(r0), "=&v"(r1)
            : "a"(a0), "a"(a1), "v"(rh), "v"(rl), "v"(d), "v"(big));
}

// One accumulator element, AGPR -> VGPR, exactly where it is consumed.  Left to itself hipcc splits the accumulators'
// live ranges at the loop exit and copies dozens of them to VGPRs up front (spilling the epilogue's own registers).
__device__ __forceinline__ float w4_acc(float a) {
    float v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}

// four of them with ONE wait: 32 consecutive weight rows' worth (4 register groups x both lane halves)
__device__ __forceinline__ void w4_sload32(const float* base_uniform, int byte_off_uniform, f32x8& v0, f32x8& v1, f32x8& v2, f32x8& v3) {
    asm volatile("s_load_dwordx8 %0, %4, %5\n\ts_load_dwordx8 %1, %4, %5 offset:32\n\ts_load_dwordx8 %2, %4, %5 offset:64\n\t"
                 "s_load_dwordx8 %3, %4, %5 offset:96\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3) : "s"(base_uniform), "s"(byte_off_uniform) : "memory");
}

struct W4Frag {          // the fragments of one k32 step of one half (hi or lo): [tile][k16 step]
    half8 w[4][2];       // weights:     rows wn*128 + 32*it + (lane & 31)
    half8 x[4][2];       // activations: rows wm*128 + 32*jt + (lane & 31)
};

// HEAD > 0: the tile is not stored, its HEAD-wide output head is (partial sums);  HEAD == -1: the tile IS stored and, on
// top, the one-output auxiliary head w_aux (reference architectures.py:60) is accumulated from the very values being
// stored (hi + lo, what heads_kernel<1> would re-read from HBM: 268 MB and a launch saved); HEAD == 0: plain layer;
// HEAD == -2: the tile is stored as plain fp32 rows (y = [M_pad][N] floats: the pre-BatchNorm z of the TRAINING forward,
// csrc/train.hip) -- 32 floats of a row are 128 bytes, so addressing and transposition are those of the line format;
// with RES, res = an fp32 [M_pad][N] matrix that is ADDED (may alias y: the data-gradient GEMM accumulating into da).
// HEAD == -3: as -2, and the reduction is split: p.ksplit work items per output tile, each over p.K of the operands'
// p.K * p.ksplit columns, partial s of the output at y + s * M_pad * N floats (the weight-gradient GEMM: 16 output tiles,
// K = the batch; a fixed-order reduction kernel adds the partials).
// TRANS (round 4; HEAD == -3 only): both operands are given REDUCTION-MAJOR -- x = [rows][M_pad] lines, w = [rows][N] lines, the
// reduction runs over the rows (p.K * p.ksplit of them): the weight-gradient GEMM dW = dz^T . x reads dz and x as the [batch][H]
// lines they already exist as (the data-gradient operand and the forward's activations), no transposed copies (tlines_kernel:
// 2 ms of the 65536-row step).  A k32 step = 32 batch rows; per operand and slot 32 x 512 B (the hi or the lo halves of 256
// columns) = the same 16 KiB: the DMA fetches two batch rows per instruction (lane = row * 32 + 16-byte chunk of the 512 B), the
// LDS image is [k][256 columns] fp16, and the MFMA fragments (lane = operand row, 8 consecutive k) come out of it through
// ds_read_b64_tr_b16, gfx950's transposing LDS read (two per 8-k fragment: lane s of a 16-lane group supplies the address of
// columns 4 (s & 3) .. + 3 of k row s >> 2 and receives column s over the 4 k rows; semantics probed by tools/ubench/tr16.hip).
// 64-byte column chunks are XOR-swizzled with k & 3 (on the DMA's source address and on the read address): the 4 k rows of a read
// fall into the 4 different 64-byte windows of the 256-byte bank line.  Same k grouping per MFMA as the transposed-lines path:
// bit-identical results.
// NJ (round 4): MFMA tiles per wave along m.  4 = the 256 (n) x 256 (m) workgroup tile above; 2 = the HALF-SIZE tile, 256 (n) x
// 128 (m), 4 waves x 128 x 64, for 4096 < rows <= ~12000 where the full tiles are fewer than the CUs (8192 rows: 128 tiles on 256
// CUs, 57 us per layer whatever the rows): same ring, same phases, same counted waits with 6 instead of 8 DMA instructions per
// wave and phase (4 of 64 weight rows + 2 of 32 activation rows), 16 + 32 MFMAs per phase pair, 8 epilogue passes.  The X half of
// a slot is half used.
// RESNT (round 6, RES only): the residual tile is read with non-temporal loads -- for a residual matrix larger than the Infinity Cache
// (65536 rows x 1024: 256 MiB), which is read once per launch, long after it was written: 2526 -> 2506 us per 65536-row forward; at 16384
// rows, where the matrix is still cached when its reader comes by, the same loads cost 0.6 % (the host picks by size).
template <int NSPLIT, bool RELU, bool RES, int HEAD, bool TRANS = false, int NJ = 4, bool RESNT = false>
__global__ __launch_bounds__(W4_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void dense_kernel_w4(DenseParams p) {
    __shared__ __attribute__((aligned(16))) char smem[W4_LDS];
    static_assert(!TRANS || HEAD == -3, "reduction-major operands: the split-K weight-gradient variant only");
    static_assert(!RESNT || (RES && NJ == 4 && HEAD >= -1), "non-temporal residual loads: the full-size inference tile with a residual");
    static_assert(NJ == 4 || (NJ == 2 && !TRANS && HEAD >= -1), "half-size tile: inference layers only (plain / residual, fused w_aux, fused w_fin)");
    constexpr int BMT = 64 * NJ;   // rows (m) of a workgroup tile
    constexpr int NQ = 4 + NJ;     // fragment quarters = DMA instructions per wave and slot: 4 of W, NJ of X
    constexpr bool SPLIT = NSPLIT == 3;
    constexpr bool AUX = HEAD == -1;
    constexpr bool F32OUT = HEAD <= -2;
    constexpr bool SPLITK = HEAD == -3;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 1;    // 2 waves along n (128 weight rows each)
    const int wm = w >> 1;   // 2 waves along m (128 persons each)

    const int NT = p.N / BN;
    const int otiles = (p.M_pad / BMT) * NT;                  // output tiles
    const int ntiles = SPLITK ? otiles * p.ksplit : otiles;  // work items
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    // bytes per operand row: a row of the k range (normal) / a batch row of all columns (TRANS: x and w may differ in width)
    const size_t rowb = TRANS ? (size_t)p.N * 4 : (SPLITK ? (size_t)p.K * p.ksplit * 4 : (size_t)p.K * 4);
    const size_t rowbx = TRANS ? (size_t)p.M_pad * 4 : rowb;
    const size_t yrowb = (size_t)p.N * 4;
    const int nk = p.K / 32;   // even (K % 64 == 0, guaranteed by the host)
    const float descale = p.descale_ptr ? *p.descale_ptr : p.descale;   // (before the first LDS-DMA: an ordinary load)

    // ---- LDS-DMA duty: per slot a wave fetches 64 W rows and 64 X rows, 4 instructions of 16 rows x 64 B each.
    // lane -> (row = lane / 4, position = lane % 4); the LDS image is lane-linear, the bank swizzle
    // chunk ^= (row >> 2) & 3 is applied on the SOURCE address (and again on the ds_read address).
    // Source address = wave-uniform 64-bit base (request-stream pointer) + one of four lane offsets (+ 64 for lo halves).
    unsigned goffq[4], goffx[TRANS ? 4 : 1];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (TRANS) {   // instruction q: batch rows 2q, 2q + 1 of this wave's 8; lane = row * 32 + (64-byte chunk * 4 + 16-byte part)
            const int kr = 2 * q + (lane >> 5), ch = (lane >> 2) & 7;
            goffq[q] = (unsigned)(kr * (int)rowb + ((ch ^ (kr & 3)) * 128) + (lane & 3) * 16);
            goffx[q] = (unsigned)(kr * (int)rowbx + ((ch ^ (kr & 3)) * 128) + (lane & 3) * 16);
        } else {
            goffq[q] = (unsigned)(((lane >> 2) + 16 * q) * (int)rowb + (((lane & 3) ^ ((lane >> 4) & 3)) * 16));
        }
    }
    // instruction q (0..3 = W rows, 4..7 = X rows) of this wave's share of ring slot `slot` (odd = lo halves) from the
    // request-stream pointers (rw, rx) = this wave's first row of the k32 step being requested
    // One asm statement per instruction: SGPR base + 32-bit lane offset + immediate (hipcc's own lowering of the builtin
    // spends a 64-bit VALU add per instruction on a full 64-bit lane address); M0 is written in the same statement that
    // uses it (it is compiler-reserved); s_mov + s_nop 3 = the 5 wait states an SALU-written SGPR (the base pointer, wherever
    // hipcc schedules its update) needs before a VMEM instruction reads it; these loads are invisible to hipcc's vmcnt bookkeeping -- every wait for them is
    // one of the counted ones below, and the stream is drained before any code hipcc counts for.
    int dma_base = (int)(size_t)(__attribute__((address_space(3))) char*)smem + (w * 64) * 64;   // LDS byte address of this
                                                                   // wave's first row in a slot's W part (opaque per phase)
    auto issue1 = [&](const char* rw, const char* rx, int slot, int q) {
        if (W4_DBG(4)) return;
        // (the instruction's immediate offset is added to the global AND to the LDS address: the lo halves' + 64 is taken
        //  out of M0 again)
        // (X rows of this wave: 16 NJ of the tile's 64 NJ, i.e. its LDS rows start 64 - 16 NJ rows earlier than the W share's)
        const int ldsa = dma_base + slot * W4_SLOT + (q < 4 ? 0 : W4_XOFF - w * 64 * (64 - 16 * NJ)) + (q & 3) * 1024 -
                         ((slot & 1) ? 64 : 0);
        const unsigned go = (TRANS && q >= 4) ? goffx[TRANS ? (q & 3) : 0] : goffq[q & 3];
        if (slot & 1)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1 offset:64"
                         :: "v"(go), "s"(q < 4 ? rw : rx), "s"(ldsa) : "memory");
        else
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(go), "s"(q < 4 ? rw : rx), "s"(ldsa) : "memory");
    };

    // ---- fragment addressing: MFMA 32x32x16, lane l supplies row (l & 31), k = 8 * (l >> 5) .. + 7 of the k16 step,
    // i.e. 16-byte chunk 2 * kk + (l >> 5) of the row's 64-byte half line
    const int ml = lane & 31, hh = lane >> 5;
    const int swz = (ml >> 2) & 3;
    const int wrow = (wn * 128 + ml) * 64;
    const int xrow = W4_XOFF + (wm * 32 * NJ + ml) * 64;
    const int c0 = ((0 + hh) ^ swz) * 16, c1 = ((2 + hh) ^ swz) * 16;

    // TRANS: lane (16-lane group g, s = lane & 15) of a transposing read addresses k row 8 (g >> 1) + (s >> 2) (+ 16 kk + 4 h
    // per instruction), columns 32 it + 16 (g & 1) + 4 (s & 3) .. + 3 of the wave's 128: 64-byte chunk 4 wn + it, swizzled by k & 3
    // = s >> 2, i.e. (it ^ (s >> 2)) in its low two bits -- one lane offset per it
    int troff[TRANS ? 4 : 1];
    if (TRANS) {
        const int g = lane >> 4, sl = lane & 15;
#pragma unroll
        for (int it = 0; it < 4; ++it)
            troff[it] = (8 * (g >> 1) + (sl >> 2)) * 512 + ((it ^ (sl >> 2)) * 64) + (16 * (g & 1) + 4 * (sl & 3)) * 2;
    }
    typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    struct H8Pair { fp16x4 a, b; };
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds0 = (lds_char*)smem;
    auto tr_read8 = [&](int off) -> half8 {   // k rows +0..3 and +4..7: the 8 k values of this lane's operand row
        H8Pair pr;
        pr.a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(lds0 + off));
        pr.b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(lds0 + off + 4 * 512));
        return __builtin_bit_cast(half8, pr);
    };

    // read quarter `qr` (0..7) of a fragment set from slot `slot`: 2 ds_read_b128 (both k16 steps of one 32-row tile)
    auto read_q = [&](W4Frag& f, int slot, int qr) {
        if (W4_DBG(8)) return;
        const char* sb = smem + slot * W4_SLOT;
        if (TRANS) {
            const int ob = slot * W4_SLOT + (qr < 4 ? wn * 256 : W4_XOFF + wm * 256) + troff[TRANS ? (qr & 3) : 0];
            if (qr < 4) {
                f.w[qr][0] = tr_read8(ob);
                f.w[qr][1] = tr_read8(ob + 16 * 512);
            } else {
                f.x[qr - 4][0] = tr_read8(ob);
                f.x[qr - 4][1] = tr_read8(ob + 16 * 512);
            }
            return;
        }
        if (qr < 4) {
            f.w[qr][0] = *(const half8*)(sb + wrow + qr * 2048 + c0);
            f.w[qr][1] = *(const half8*)(sb + wrow + qr * 2048 + c1);
        } else {
            f.x[qr - 4][0] = *(const half8*)(sb + xrow + (qr - 4) * 2048 + c0);
            f.x[qr - 4][1] = *(const half8*)(sb + xrow + (qr - 4) * 2048 + c1);
        }
    };

    // virtual block id -> tile: the XCD-aware bijective map of dense_kernel.h
    auto tile_of = [&](int vb, int& m0, int& n0, int& ks) {
        const int xcd = vb & 7, idx = vb >> 3;
        int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        ks = 0;
        if (SPLITK) {
            ks = tile / otiles;
            tile -= ks * otiles;
        }
        const int mt = tile / NT, nt = tile - mt * NT;
        m0 = mt * BMT;
        n0 = nt * BN;
    };

    int vb = blockIdx.x;
    if (vb >= ntiles) return;
    int m0, n0, ks0;
    tile_of(vb, m0, n0, ks0);

    char* const scr = smem + W4_RING + w * 8192;   // two 4 KiB epilogue buffers of this wave

    // ---- the request stream: it runs two k32 steps ahead of the compute stream and simply continues into the next
    // tile of this workgroup; behind the last tile it re-requests that tile (valid memory, data never read), so that no
    // request is conditional and every phase has the same vmcnt arithmetic.
    int rq_vb = vb, rq_left = nk;
    const size_t wave_rows = (size_t)(w * 64) * rowb, wave_rows_x = (size_t)(w * 16 * NJ) * rowb;
    // TRANS: first batch row of the work item's k range + this wave's 8 rows of every k32 step; column tile = byte n0 * 4 of a row
    auto rq_w_of = [&](int tn0, int tks) -> const char* {
        return TRANS ? p.w + ((size_t)tks * p.K + w * 8) * rowb + (size_t)tn0 * 4
                     : p.w + (size_t)tn0 * rowb + wave_rows + (size_t)tks * p.K * 4;   // (k offset in bytes: 128 per k32 block)
    };
    auto rq_x_of = [&](int tm0, int tks) -> const char* {
        return TRANS ? p.x + ((size_t)tks * p.K + w * 8) * rowbx + (size_t)tm0 * 4
                     : p.x + (size_t)tm0 * rowb + wave_rows_x + (size_t)tks * p.K * 4;
    };
    const char* rq_w = rq_w_of(n0, ks0);
    const char* rq_x = rq_x_of(m0, ks0);
    auto rq_advance = [&]() {   // after both slots of a k32 step have been requested
        rq_w += TRANS ? 32 * rowb : (size_t)LINE;
        rq_x += TRANS ? 32 * rowbx : (size_t)LINE;
        if (--rq_left == 0) {
            if (rq_vb + (int)gridDim.x < ntiles) rq_vb += (int)gridDim.x;
            int rm0, rn0, rks;
            tile_of(rq_vb, rm0, rn0, rks);
            rq_w = rq_w_of(rn0, rks);
            rq_x = rq_x_of(rm0, rks);
            rq_left = nk;
        }
    };

    // ---- prologue of the stream (once per workgroup): the first two k-steps, then the first hi fragments
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            issue1(rq_w, rq_x, 2 * st, q);
            if (SPLIT) issue1(rq_w, rq_x, 2 * st + 1, q);
        }
        rq_advance();
        // SALU write of an SGPR -> VMEM read of it needs 5 wait states, and hipcc pads nothing for an asm statement
        // (in the loop a whole phase start lies between rq_advance and the next request)
        asm volatile("s_nop 4" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    W4Frag fa, fb, fl;   // hi fragments of even / odd k-steps, lo fragments of the current step
#pragma unroll
    for (int qr = 0; qr < NQ; ++qr) read_q(fa, 0, qr);

    // optional timeline (bring-up builds, -DML_BRINGUP -DML_DENSE_TRACE): per tile 16 slots of this wave <- s_memtime:
    //   0 tile start, 1..8 end of the first 8 phases, 9 end of main loop, 10 stream drained, 11 epilogue issued
#ifdef ML_DENSE_TRACE
    unsigned long long* const trc = p.trace ? p.trace + ((size_t)blockIdx.x * 8 + w) * 64 : nullptr;
    int ttile = 0;
    auto stamp = [&](int slot) {
        if (trc && ttile < 4) {
            const unsigned long long ts = __builtin_amdgcn_s_memtime();
            if (lane == 0) trc[ttile * 16 + slot] = ts;
        }
    };
#else
    auto stamp = [&](int) {};
#endif
    while (true) {
        const int vb_next = vb + (int)gridDim.x;
        const bool more = vb_next < ntiles;   // workgroup-uniform

        // accumulators start at bias * 2^e (pre-scaled on the host, exact): the epilogue is (bias * 2^e + sum) * 2^-e with
        // no bias add.  Register r of MFMA tile (it, *) is weight row nbase + 32 it + 8 (r >> 2) + 4 (lane >> 5) + (r & 3):
        // eight consecutive floats per (it, r >> 2) through the scalar cache, the lane half selects four of them.
        f32x16 acc[4][NJ];
        {
            const float* bsc = p.bias_scaled + n0 + wn * 128;   // wave-uniform
            // (lane half recomputed here and made opaque: a kernel-lifetime copy gets spilled around the main loop)
            int il = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(il));
            const bool upper = il >= 32;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                f32x8 b8[4];
                w4_sload32(bsc, it * 128, b8[0], b8[1], b8[2], b8[3]);   // one scalar-cache round trip per row block
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float b = upper ? b8[g][4 + e] : b8[g][e];
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt) acc[it][jt][g * 4 + e] = b;
                    }
            }
        }

        // The slot a phase reads was requested three phases earlier (split mode; one phase earlier in the single-product
        // modes): in steady state exactly the DMA groups (8 instructions per wave each) of the two phases in between are
        // younger.  The first phases after a drain (prologue / epilogue) read slots that were drained: no vmcnt wait at
        // all there, so the epilogue's stores keep retiring behind the next tile's MFMAs.
        int since_drain = 0;          // phases since the last vmcnt(0)
        stamp(0);
        // REBAL (half-size tile): a 16-MFMA phase A cannot cover six LDS-DMA issues (~100 cycles each), so phase A requests only the
        // four W instructions of H(t+2) and phase B its two X instructions in front of L(t+2)'s six (4 + 8 per k-step): in steady
        // state phase A (reads L(t), requested as the last six of B(t-2)) has A(t-1)'s 4 + B(t-1)'s 8 = 12 younger instructions,
        // phase B (reads H(t+1): A(t-1)'s 4 + the first 2 of B(t-1)) the other 6 of B(t-1) + A(t)'s 4 = 10
        // The same split on the FULL-size tile (A: the 4 W instructions, B: H(t+2)'s 4 X instructions + L(t+2)'s 8; vmcnt 16 / 12) is
        // kept behind -DML_W4_REBAL4: measured 2.560 vs 2.545 ms per 65536-row step, three alternations on one box
        // (profiles/r04_ablation.md) -- its 32-MFMA phase A already covers its issues, and the chip sits at its power cap.
#ifdef ML_W4_REBAL4
        constexpr bool REBAL = SPLIT;
#else
        constexpr bool REBAL = SPLIT && NJ == 2;
#endif
        auto phase_wait = [&](bool phase_b = false) {
#ifdef ML_DENSE_TRACE
            if (since_drain >= 1 && since_drain <= 8) stamp(since_drain);
#endif
            asm volatile("" : "+s"(since_drain));   // (opaque: keeps hipcc from peeling a copy of the loop body)
            asm volatile("" : "+s"(dma_base));      // (opaque: m0 = base + constant per DMA instead of 32 hoisted SGPRs)
            if (W4_DBG(32)) return;
            if (since_drain >= (SPLIT ? 3 : 1)) {
                // (the DMA groups of the two phases in between: 2 x NQ instructions of this wave)
                if (SPLIT && NJ == 4 && REBAL && phase_b) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (SPLIT && NJ == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (REBAL && phase_b && NJ == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else if (SPLIT) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            ++since_drain;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of the slot that is about to be refilled
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };

        // One k32 step.  FH = hi fragments of this step (already in registers), FN = the other hi set (receives the next
        // step's), S = 0 / 2: ring slots of this step's H (S) and L (S + 1); the next step's are S ^ 2.
        auto kstep = [&](W4Frag& FH, W4Frag& FN, const int S) {
            // ---------------- phase A: hi.hi, 32 MFMAs; reads the lo fragments of this step; requests H(t+2) into this
            // step's H slot (its fragments are in FH since the previous phase)
            phase_wait();
            // per block of 4 MFMAs one DMA instruction and two fragment reads, issued BEHIND the block (a read in front
            // of the first block would make hipcc's own lgkmcnt(0) for FH wait for it); blocks 4 and 5 carry two quarters
            // each so that blocks 6 and 7 (8 MFMAs, 256 cycles) cover the last reads' latency before the next phase's wait
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int blk = kk * 4 + it;   // 8 blocks of 4 MFMAs
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt) acc[it][jt] = w4_mfma<NSPLIT>(FH.w[it][kk], FH.x[jt][kk], acc[it][jt]);
                    __builtin_amdgcn_sched_barrier(0);
                    // NJ == 4: quarters 0..3 behind blocks 0..3, two each behind blocks 4 and 5; NJ == 2: quarters 0..5 behind blocks 0..5
#pragma unroll
                    for (int qd = (NJ == 2 ? (blk < 6 ? blk : 6) : (blk < 4 ? blk : (blk < 6 ? 4 + 2 * (blk - 4) : 8)));
                         qd < (NJ == 2 ? (blk < 6 ? blk + 1 : 6) : (blk < 4 ? blk + 1 : (blk < 6 ? 6 + 2 * (blk - 4) : 8))); ++qd) {
                        if (REBAL) {   // the W instructions behind blocks 0, 2, 4, 6; the fragment reads as before
                            if ((blk & 1) == 0 && (NJ == 2 || blk < 4 || qd == 4 + 2 * (blk - 4))) issue1(rq_w, rq_x, S, blk >> 1);
                        } else {
                            issue1(rq_w, rq_x, S, qd);
                        }
                        if (SPLIT) read_q(fl, S + 1, qd);
                        else read_q(FN, S ^ 2, qd);   // single-product modes: the next step's hi fragments are read here
                    }
                    if (REBAL && blk == 6) issue1(rq_w, rq_x, S, 3);   // (blocks 6, 7 carry no fragment quarter)
                }
            if (SPLIT) {
                // ---------------- phase B: hi.lo + lo.hi, 64 MFMAs; reads the NEXT step's hi fragments; requests L(t+2)
                phase_wait(true);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int half = 0; half < 2; ++half)
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int blk = (kk * 2 + half) * 4 + it;   // 16 blocks of 4 MFMAs
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int jt = 0; jt < NJ; ++jt) {
                                if (half == 0) acc[it][jt] = w4_mfma<NSPLIT>(FH.w[it][kk], fl.x[jt][kk], acc[it][jt]);
                                else acc[it][jt] = w4_mfma<NSPLIT>(fl.w[it][kk], FH.x[jt][kk], acc[it][jt]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            // quarters behind blocks 0, 2, .., 10, 11, 12: blocks 13-15 cover the last reads
                            const int qd = (blk <= 10) ? ((blk & 1) == 0 ? blk >> 1 : -1) : (blk <= 12 ? blk - 5 : -1);
                            if (REBAL && NJ == 4) {
                                // H(t+2)'s four X instructions behind blocks 0..3, L(t+2)'s eight behind blocks 4..11
                                if (blk < 4) issue1(rq_w, rq_x, S, 4 + blk);
                                else if (blk < 12) issue1(rq_w, rq_x, S + 1, blk - 4);
                                if (qd >= 0 && qd < NQ) read_q(FN, S ^ 2, qd);
                            } else if (REBAL) {
                                // H(t+2)'s two X instructions first, then L(t+2)'s six: one instruction per block 0, 1, 3, 5, 7, 9, 11, 12
                                if (blk == 0) issue1(rq_w, rq_x, S, 4);
                                else if (blk == 1) issue1(rq_w, rq_x, S, 5);
                                else if (blk >= 3 && blk <= 11 && (blk & 1)) issue1(rq_w, rq_x, S + 1, (blk - 3) >> 1);
                                else if (blk == 12) issue1(rq_w, rq_x, S + 1, 5);
                                if (qd >= 0 && qd < NQ) read_q(FN, S ^ 2, qd);
                            } else if (qd >= 0 && qd < NQ) {
                                issue1(rq_w, rq_x, S + 1, qd);
                                read_q(FN, S ^ 2, qd);
                            }
                        }
            }
            rq_advance();
        };

#pragma clang loop unroll(disable)
        for (int t = 0; t < nk; t += 2) {
            kstep(fa, fb, 0);
            kstep(fb, fa, 2);
        }
        // every request of this tile's loop is now waited for except the last one or two groups; the epilogue below
        // counts its own vector-memory operations, so drain the stream first (the youngest group was requested a whole
        // phase ago)
        stamp(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(10);
        // the last MFMAs' results are read by hand-placed v_accvgpr_read below and no hazard padding is inserted for asm
        // operands: 16-pass MFMA -> AGPR read needs up to 18 wait states
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue of the finished tile, one 32 (n) x 32 (m) MFMA tile per pass, it-major (4 passes share a bias)
        const int nbase = n0 + wn * 128;
        const int mbase = m0 + wm * 32 * NJ;
        int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(elane));   // lane-dependent epilogue addresses are derived here, per tile (not hoisted)
        const int eml = elane & 31, eh = elane >> 5;
        const unsigned st_off = (unsigned)((elane >> 3) * (int)yrowb + ((elane & 7) * 16));
        const int scr_row = eml * LINE + eh * 8;                       // + ((chunk ^ esw) * 16)
        const int esw = ML_W4_ESWZ ? ((eml >> 1) & 7) : (eml & 7);
        // store-layout side: row 8 qq + lane / 8, chunk lane % 8 -> + qq * 1024; ESWZ 1: the row's swizzle is (4 (qq & 1) + lane / 16) & 7,
        // i.e. the offset below with bit 6 flipped for odd qq (RDX)
        const int rd_off = (elane >> 3) * LINE + (((elane & 7) ^ (ML_W4_ESWZ ? (elane >> 4) : ((elane >> 3) & 7))) * 16);
#define RDX(qq) ((ML_W4_ESWZ && ((qq) & 1)) ? 64 : 0)
        // addresses of a pass = one wave-uniform 64-bit tile base + a 32-bit offset (kept as a running, opaque value so
        // that hipcc does not pre-compute 16 + 16 address pairs into SGPRs and spill them)
        const size_t tile_off = (size_t)mbase * yrowb + (size_t)nbase * 4 + (SPLITK ? (size_t)ks0 * p.M_pad * yrowb : 0);
        const unsigned row8 = (unsigned)(8 * (int)yrowb);
        auto pass_off = [&](int pass) {   // pass = it * NJ + jt
            unsigned o = (unsigned)((pass % NJ) * 32) * (unsigned)yrowb + (unsigned)((pass / NJ) * 128);
            asm volatile("" : "+s"(o));
            return o;
        };
        const float lim = 65504.0f / descale;   // the fp16 range in the accumulator's scale (descale is a power of two)

        if (W4_DBG(1)) {   // ablation: keep the accumulators live, store (almost) nothing
            float sdbg = 0.f;
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) sdbg += w4_acc(acc[it][jt][(it * 4 + jt) & 15]);
            if (sdbg == 123456.789f) p.y[tid] = 1;
        } else if (HEAD > 0) {
            // the activation tile is not stored: each wave multiplies its relu'd 128-column slice with the HEAD x 128
            // slice of the head weights (staged in its 8 KiB epilogue area: ordinary loads, the stream is drained)
            float* hw = (float*)scr;
            for (int idx = elane; idx < HEAD * 32; idx += 64) {
                const int o = idx >> 5, c4 = idx & 31;
                *(f32x4*)(hw + o * 128 + c4 * 4) = *(const f32x4*)(p.head_w + (size_t)o * p.N + nbase + c4 * 4);
            }
            __builtin_amdgcn_wave_barrier();
            const int slice = (n0 / BN) * 2 + wn;
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                float part[HEAD > 0 ? HEAD : 1];
#pragma unroll
                for (int o = 0; o < HEAD; ++o) part[o] = 0.0f;
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = w4_acc(acc[it][jt][g * 4 + e]) * descale;
                            if (RELU) v[e] = __builtin_fmaxf(v[e], 0.0f);
                        }
#pragma unroll
                        for (int o = 0; o < HEAD; ++o) {
                            const f32x4 w4 = *(const f32x4*)(hw + o * 128 + it * 32 + g * 8 + eh * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) part[o] = __builtin_fmaf(v[e], w4[e], part[o]);
                        }
                    }
#pragma unroll
                for (int o = 0; o < HEAD; ++o) part[o] += __shfl_xor(part[o], 32, 64);
                if (eh == 0) {
                    float* dst = p.head_part + ((size_t)slice * p.M_pad + (mbase + jt * 32 + eml)) * 16;
#pragma unroll
                    for (int o4 = 0; o4 < (HEAD + 3) / 4; ++o4) {
                        f32x4 q4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) q4[e] = (o4 * 4 + e < HEAD) ? part[o4 * 4 + e] : 0.0f;
                        *(f32x4*)(dst + o4 * 4) = q4;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the staging loads' and these few stores' waits are hipcc's)
        } else {
            // Software-pipelined over the 16 passes (two 4 KiB buffers per wave):
            //   region p:  [residual tile p+3 requested: plain 16-byte loads in store layout, three passes ahead, so that
            //               the in-order vmcnt wait for tile p only needs the stores of pass p-4 to have retired]
            //              [residual tile p: registers -> buffer p&1 -> MFMA layout]   [arithmetic of pass p]
            //              [global stores of pass p-1 (its transposed lines were read in region p-1: latency hidden)]
            //              [packed hi|lo of pass p -> buffer p&1 -> 16-byte lines read back for region p+1]
            // hipcc counts these loads and stores itself (no LDS-DMA is in flight here), in issue order.
            f32x4 rq[RES ? 4 * NJ : 1][4];
            auto load_res = [&](int pass) {
                const char* src = p.res + tile_off;
                const unsigned o = pass_off(pass) + st_off;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    if (RESNT) rq[RES ? pass : 0][qq] = __builtin_nontemporal_load((const f32x4*)(src + (o + qq * row8)));
                    else rq[RES ? pass : 0][qq] = *(const f32x4*)(src + (o + qq * row8));
                }
            };
            constexpr int RDEPTH = 3;   // residual tiles in flight (deeper does not help: 8 = same time; the tile-wide burst is HBM-bound)
            if (RES) {
#pragma unroll
                for (int q = 0; q < RDEPTH; ++q) load_res(q);
            }
            f32x4 d[4];
            auto flush = [&](int pass) {   // the transposed lines of `pass` (in d[]) -> global memory
                char* dst = p.y + tile_off;
                const unsigned o = pass_off(pass) + st_off;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {  // rows 8qq + lane/8, chunk lane%8: full 128-byte lines
                    if (W4_DBG(64)) asm volatile("" :: "v"(d[qq]));
                    else if (W4_DBG(128)) *(f32x4*)(p.y + (size_t)(blockIdx.x * 4 + w) * 4096 + qq * 1024 + elane * 16) = d[qq];
                    else *(f32x4*)(dst + (o + qq * row8)) = d[qq];
                }
            };
            float auxp[4] = {0.f, 0.f, 0.f, 0.f};   // AUX: this lane's part of w_aux . y for its person of row block jt
            double cs[4] = {0.0, 0.0, 0.0, 0.0}, cq[4] = {0.0, 0.0, 0.0, 0.0};   // colpart: column sums / sums of squares of row block `it`
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float auxw[16];
                if (AUX) {   // the 16 head weights of this lane's weight rows of row block `it`
                    f32x8 a8[4];
                    w4_sload32(p.head_w + nbase, it * 128, a8[0], a8[1], a8[2], a8[3]);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) auxw[g * 4 + e] = eh ? a8[g][4 + e] : a8[g][e];
                }
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const int pass = it * NJ + jt;
                    char* const buf = scr + (pass & 1) * 4096;
                    u32x2 rh[4], rl[4];
                    if (RES) {
                        if (pass + RDEPTH < 4 * NJ) load_res(pass + RDEPTH);
                    }
                    if (RES && !F32OUT) {
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) *(f32x4*)(buf + (rd_off ^ RDX(qq)) + qq * 1024) = rq[pass][qq];
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            rh[g] = *(const u32x2*)(buf + scr_row + ((g ^ esw) * 16));
                            rl[g] = *(const u32x2*)(buf + scr_row + (((g + 4) ^ esw) * 16));
                        }
                    }
                    u32x2 oh[4], ol[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            const float a0 = acc[it][jt][g * 4 + 2 * e2], a1 = acc[it][jt][g * 4 + 2 * e2 + 1];
                            unsigned hq, lq;
                            if (F32OUT) {   // plain fp32: hq / lq carry the two floats of the pair
                                float v0 = w4_acc(a0) * descale, v1 = w4_acc(a1) * descale;
                                if (RELU) {
                                    v0 = __builtin_fmaxf(v0, 0.0f);
                                    v1 = __builtin_fmaxf(v1, 0.0f);
                                }
                                hq = __builtin_bit_cast(unsigned, v0);
                                lq = __builtin_bit_cast(unsigned, v1);
                            } else if (NSPLIT == 0) {   // bf16 lines: one bf16 in the hi slot
                                float v0 = w4_acc(a0) * descale, v1 = w4_acc(a1) * descale;
                                if (RELU) {
                                    v0 = __builtin_fmaxf(v0, 0.0f);
                                    v1 = __builtin_fmaxf(v1, 0.0f);
                                }
                                if (RES) {
                                    v0 += __builtin_bit_cast(float, rh[g][e2] << 16);
                                    v1 += __builtin_bit_cast(float, rh[g][e2] & 0xffff0000u);
                                }
                                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hq) : "v"(v0), "v"(v1));
                                lq = 0u;
                            } else if (RES) {
                                w4_pair_res<RELU>(a0, a1, descale, rh[g][e2], rl[g][e2], hq, lq);
                            } else {
                                w4_pair_plain(a0, a1, descale, RELU ? 0.0f : -lim, lim, hq, lq);
                            }
                            oh[g][e2] = hq;
                            ol[g][e2] = lq;
                            if (AUX) {   // the stored value itself, hi + lo (bf16 mode: the bf16 value), times its head weight;
                                         // one statement, so that the two temporaries die at once
                                if (NSPLIT == 0) {
                                    auxp[jt] = __builtin_fmaf(__builtin_bit_cast(float, hq << 16), auxw[g * 4 + 2 * e2], auxp[jt]);
                                    auxp[jt] = __builtin_fmaf(__builtin_bit_cast(float, hq & 0xffff0000u), auxw[g * 4 + 2 * e2 + 1], auxp[jt]);
                                } else {
                                    float y0, y1;
                                    asm("v_fma_mix_f32 %1, %3, 1.0, %4 op_sel_hi:[1,0,1]\n\t"
                                        "v_fma_mix_f32 %2, %3, 1.0, %4 op_sel:[1,0,1] op_sel_hi:[1,0,1]\n\t"
                                        "v_fmac_f32 %0, %1, %5\n\t"
                                        "v_fmac_f32 %0, %2, %6"
                                        : "+v"(auxp[jt]), "=&v"(y0), "=&v"(y1)
                                        : "v"(hq), "v"(lq), "v"(auxw[g * 4 + 2 * e2]), "v"(auxw[g * 4 + 2 * e2 + 1]));
                                }
                            }
                        }
                    }
                    if (pass > 0) flush(pass - 1);
                    // transpose through the buffer (its residual image, if any, is in registers: LDS operations of one wave
                    // execute in order)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (F32OUT) {   // 4 consecutive floats of this person = 16-byte chunk 2g + (lane half) of its 128-byte row
                            u32x4 q4 = {oh[g][0], ol[g][0], oh[g][1], ol[g][1]};
                            *(u32x4*)(buf + eml * LINE + (((2 * g + eh) ^ esw) * 16)) = q4;
                        } else {
                            *(u32x2*)(buf + scr_row + ((g ^ esw) * 16)) = oh[g];
                            *(u32x2*)(buf + scr_row + (((g + 4) ^ esw) * 16)) = ol[g];
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        d[qq] = *(const f32x4*)(buf + (rd_off ^ RDX(qq)) + qq * 1024);
                        if (F32OUT && RES) d[qq] += rq[pass][qq];   // fp32 accumulate: the residual is fp32 in store layout already
                    }
                    if (F32OUT && !RES && !SPLITK && NJ == 4) {
                        // BatchNorm batch statistics of the tile being stored (training forward, p.colpart): in store layout a lane
                        // holds 4 consecutive columns (32 it + 4 (lane & 7) ..) of rows 32 jt + 8 qq + (lane >> 3): fp64 sums over
                        // the 16 rows of the four passes of `it`, then over the 8 lanes that share the columns, one 128-row block per
                        // wave -- no second pass over z (col_stats_kernel: 44 us per layer at 65536 x 1024), no atomics
                        if (p.colpart) {   // (uniform)
                            if (jt == 0) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) cs[e] = cq[e] = 0.0;
                            }
#pragma unroll
                            for (int qq = 0; qq < 4; ++qq) {
                                const bool valid = mbase + jt * 32 + 8 * qq + (elane >> 3) < p.m_valid;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const double v = valid ? (double)d[qq][e] : 0.0;
                                    cs[e] += v;
                                    cq[e] = __builtin_fma(v, v, cq[e]);
                                }
                            }
                            if (jt == 3) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
#pragma unroll
                                    for (int o = 8; o <= 32; o <<= 1) {
                                        cs[e] += __shfl_xor(cs[e], o, 64);
                                        cq[e] += __shfl_xor(cq[e], o, 64);
                                    }
                                }
                                if ((elane >> 3) == 0) {
                                    double* dstp = p.colpart + (size_t)(mbase >> 7) * 2 * p.N + nbase + it * 32 + (elane & 7) * 4;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        dstp[e] = cs[e];
                                        dstp[p.N + e] = cq[e];
                                    }
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);  // one region per pass: interleaving more of them costs registers
                }
            }
            flush(4 * NJ - 1);
            if (AUX) {   // combine the two lane halves (weight rows 4h..4h+3 of every group of 8), one partial per person
                const int slice = (n0 / BN) * 2 + wn;
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const float sum = auxp[jt] + __shfl_xor(auxp[jt], 32, 64);
                    if (eh == 0) p.head_part[(size_t)slice * p.M_pad + (mbase + jt * 32 + eml)] = sum;
                }
            }
        }

#undef RDX
        stamp(11);
#ifdef ML_DENSE_TRACE
        ++ttile;
#endif
        if (!more) break;
        vb = vb_next;
        tile_of(vb, m0, n0, ks0);
        // the next tile's first hi fragments again (the copy read in the last phase is not kept live across the
        // epilogue: 64 registers the epilogue needs; slot 0 is untouched until the next request behind the barrier)
#pragma unroll
        for (int qr = 0; qr < NQ; ++qr) read_q(fa, 0, qr);
    }
}

}  // namespace mlk
