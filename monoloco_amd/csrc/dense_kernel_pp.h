// dense_kernel_pp.h -- second generation of the dense layer: persistent workgroups + ping-pong
// main loop + fully asynchronous epilogue.  Same math, tile shape (256 n x 256 m), memory format,
// LDS stage layout and fragment addressing as dense_kernel.h; what changes is the schedule.
//
// Why (measured on MI355X, profiles/r01_ablation.md): with one workgroup-wide barrier per k-step both
// waves of a SIMD sit in the same phase and the MFMA pipe idles while they wait for LDS; the epilogue's
// HBM stores and residual loads overlap with nothing; every ordinary vector load in the epilogue makes
// hipcc drain vmcnt(0) (stores included) because LDS-DMA is in flight; and an LDS-DMA instruction costs
// its wave ~100 cycles of issue time, so eight of them in one phase make that phase twice as long as
// the 768 MFMA cycles it is supposed to hide behind.
//
//  * persistent: grid = min(tiles, CUs); a workgroup walks tiles b, b+grid, ...  Stores are
//    fire-and-forget, so the 256 KiB a tile writes drain while the next tile computes, and the next
//    tile's first stage is requested BEFORE the epilogue runs.
//  * ping-pong: the 8 waves form two groups (waves 0-3 / 4-7 = the two waves of each SIMD).  A wave
//    alternates an L phase (12 ds_read_b128 of one k16 half-step + 2 LDS-DMA instructions) and a C phase
//    (its 24 MFMAs, with the other 2 DMA instructions of its share issued between them); group 1 runs one
//    phase behind group 0, so on every SIMD one wave computes while the other reads LDS.  Phases are separated by raw
//    s_barrier; all vmcnt waits are counted ones placed by hand (vmcnt retires in order).
//  * epilogue per 32x32 MFMA tile through a private 4 KiB LDS scratch per wave (the 32 KiB the two
//    stage buffers leave free).  No ordinary vector load: bias comes through the scalar path
//    (s_load), residual lines by LDS-DMA into the stage-1 buffer, which is idle during the epilogue.
//  * HEAD > 0 (the layer feeding w_fin, reference architectures.py:64-67): the activation tile is not
//    stored at all; each wave multiplies its relu'd 128-column slice with the HEAD x 128 slice of the
//    head weights (staged in its part of the idle stage-1 buffer), combines the two lane halves and
//    writes HEAD partial sums per person; head_reduce_kernel adds the 2*N/256 slices and the bias.
#pragma once
#include "dense_kernel.h"

// Where the LDS-DMA instructions of a k-step are issued (profiles/r01_ablation.md, "schedule B"):
//   ML_SCHED_B 0: all 4 per operand share in the L phases;  1: 3 in L + 1 between the MFMAs of the next C phase;
//   2 (default): 2 + 2.   ML_HOOK_STYLE: position of the C-phase ones (1 = one per MFMA row block, default).
#ifndef ML_SCHED_B
#define ML_SCHED_B 2
#endif
#ifndef ML_HOOK_STYLE
#define ML_HOOK_STYLE 1
#endif
#ifndef ML_SETPRIO
#define ML_SETPRIO 0   // 1: s_setprio(1) around every C phase; 2: static priority 1 for waves 4-7
#endif
#ifndef ML_L_ORDER
#define ML_L_ORDER 0   // 1: in an L phase the fragment reads are issued before the DMA instructions
#endif

// (the run-time bring-up / ablation bits of rounds 1-2 are gone; the timing ablations that remain are compile-time, ML_W4_ABL)
#define ML_DBG(p, bit) (0)

namespace mlk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PP_LDS = DENSE_LDS + 8 * 4096;  // 160 KiB: 2 stages + 8 x 4 KiB epilogue scratch

// debug bits (DenseParams::debug, env ML_DENSE_DEBUG; 0 in production):
//   1 skip the epilogue (accumulators kept live)     2 skip the main loop      4 no stage DMA
//   64 half the stage DMA (timing only)          16 residual always from the first row panel (timing only)
__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Epilogue arithmetic on value PAIRS with the mixed-precision FMA (v_fma_mix*): 3 VALU instructions per value
// instead of the ~8 hipcc emits for  v = acc * 2^-e; clamp; hi = (f16) v; lo = (f16)(v - (float) hi).
// Same results bit for bit for in-range values: the product with a power of two and v - hi are exact, so
// rounding once inside the fma equals rounding the separately computed fp32 value.
//   h = packed fp16 (rn(c0*d), rn(c1*d)),  l = packed fp16 (rn(c0*d - h.lo), rn(c1*d - h.hi)),  c = med3(acc, lo_lim, lim)
template <bool RELU>
__device__ __forceinline__ void split2_scaled(float a0, float a1, float d, float lim, unsigned& h, unsigned& l) {
    // (asm: the builtin makes hipcc canonicalise the MFMA results first, one more VALU instruction per value)
    const float lo_lim = RELU ? 0.0f : -lim;
    float c0, c1;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(c0) : "v"(a0), "v"(lo_lim), "v"(lim));
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(c1) : "v"(a1), "v"(lo_lim), "v"(lim));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(c0), "v"(d));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(c1), "v"(d));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(c0), "v"(d), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(c1), "v"(d), "v"(h));
}
// with a residual given as packed fp16 (hi pair rh, lo pair rl):  v = relu(acc)*d + (rh + rl), clamped to fp16 range
template <bool RELU>
__device__ __forceinline__ void split2_res(float a0, float a1, float d, unsigned rh, unsigned rl, unsigned& h, unsigned& l) {
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(r0) : "v"(rh), "v"(rl));                  // exact
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r1) : "v"(rh), "v"(rl));
    float c0 = a0, c1 = a1;
    if (RELU) {
        asm("v_max_f32 %0, 0, %1" : "=v"(c0) : "v"(a0));
        asm("v_max_f32 %0, 0, %1" : "=v"(c1) : "v"(a1));
    }
    float v0, v1;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(v0) : "v"(c0), "v"(d), "v"(r0));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(v1) : "v"(c1), "v"(d), "v"(r1));
    const float big = 65504.0f;
    asm("v_med3_f32 %0, %1, -%2, %2" : "=v"(v0) : "v"(v0), "v"(big));
    asm("v_med3_f32 %0, %1, -%2, %2" : "=v"(v1) : "v"(v1), "v"(big));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(v0), "v"(v1));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(v1), "v"(h));
}

// bf16 comparison mode (NSPLIT == 0): one bf16 value in the hi slot of the line, lo slot unused
template <bool RELU>
__device__ __forceinline__ void bf16_2_scaled(float a0, float a1, float d, unsigned& h) {
    float v0 = a0 * d, v1 = a1 * d;
    if (RELU) {
        v0 = __builtin_fmaxf(v0, 0.0f);
        v1 = __builtin_fmaxf(v1, 0.0f);
    }
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(v0), "v"(v1));
}
template <bool RELU>
__device__ __forceinline__ void bf16_2_res(float a0, float a1, float d, unsigned rh, unsigned& h) {
    float c0 = a0, c1 = a1;
    if (RELU) {
        c0 = __builtin_fmaxf(c0, 0.0f);
        c1 = __builtin_fmaxf(c1, 0.0f);
    }
    const float v0 = __builtin_fmaf(c0, d, __builtin_bit_cast(float, rh << 16));
    const float v1 = __builtin_fmaf(c1, d, __builtin_bit_cast(float, rh & 0xffff0000u));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(v0), "v"(v1));
}

// NSPLIT: 3 = fp16 hi+lo operands, three products per term (the product mode); 1 = plain fp16; 0 = plain bf16
template <int NSPLIT, bool RELU, bool RES, int HEAD>
__global__ __launch_bounds__(DENSE_THREADS, 2) void dense_kernel_pp(DenseParams p) {
    __shared__ __attribute__((aligned(16))) char smem[PP_LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = w >> 2;  // waves w and w+4 share a SIMD
    const int wn = w & 1;    // 2 waves along n (128 weight rows each)
    const int wm = w >> 1;   // 4 waves along m (64 persons each)

    const int NT = p.N / BN;
    const int ntiles = (p.M_pad / BM) * NT;
    const int q = ntiles >> 3, r8 = ntiles & 7;
    const size_t rowb = (size_t)p.K * 4;
    const size_t yrowb = (size_t)p.N * 4;
    const int nk = ML_DBG(p, 2) ? 0 : p.K / 32;
    const bool loads = !ML_DBG(p, 4);
    const float lim = 65504.0f / p.descale;  // fp16 range in the accumulator's scale (descale is a power of two)

    // ---- LDS-DMA duty: per stage a wave fetches 32 rows of X and 32 rows of W (4 instructions of 8
    // rows each):   group 0: X rows   0..127 (wave j: 32j..) and W rows 128..255
    //               group 1: X rows 128..255                 and W rows   0..127
    // lane -> (row = base + lane/8, chunk pos = lane%8), source chunk = pos ^ ((row>>1)&7), which only
    // depends on the parity of the instruction index (the LDS destination is lane-linear).
    const int jw = w & 3;
    const int xbase = (grp ? 128 : 0) + jw * 32, wbase = (grp ? 0 : 128) + jw * 32;
    const unsigned goff_e = (unsigned)((lane >> 3) * (int)rowb + (((lane & 7) ^ (lane >> 4)) * 16));
    const unsigned goff_o = (unsigned)((lane >> 3) * (int)rowb + (((lane & 7) ^ (4 + (lane >> 4))) * 16));
    const unsigned row8 = (unsigned)(8 * (int)rowb);
    // quarters [qa, qb) of a wave's 4-instruction share of one operand of stage kt
    auto issue_q = [&](const char* tile, int base, int lds_off, int kt, int qa, int qb) {
        char* sb = smem + (kt & 1) * STAGE_BYTES + lds_off + base * LINE;
        const char* src = tile + (size_t)base * rowb + (unsigned)kt * LINE;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            if (q4 < qa || q4 >= qb) continue;
            if (ML_DBG(p, 64) && q4 >= 2) break;
            glds16(src + ((q4 & 1) ? goff_o : goff_e) + q4 * row8, sb + q4 * 8 * LINE);
        }
    };
    auto issue4 = [&](const char* tile, int base, int lds_off, int kt) { issue_q(tile, base, lds_off, kt, 0, 4); };

    // ---- fragment addressing (as dense_kernel.h)
    const int sw = (lane >> 1) & 7;
    const int h = lane >> 5;
    const int ml = lane & 31;
    const int wrow = (wn * 128 + ml) * LINE;
    const int xrow = TILE_BYTES + (wm * 64 + ml) * LINE;
    int coff[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        coff[kk][0] = (((kk * 2 + h)) ^ sw) * 16;
        coff[kk][1] = (((kk * 2 + h + 4)) ^ sw) * 16;
    }
    char* const scr = smem + DENSE_LDS + w * 4096;          // epilogue scratch of this wave
    char* const resbuf = smem + STAGE_BYTES + w * 8192;      // residual landing zone (stage-1 buffer)

    // virtual block id -> tile: the XCD-aware bijective map of dense_kernel.h
    auto tile_of = [&](int vb, int& m0, int& n0) {
        const int xcd = vb & 7, idx = vb >> 3;
        const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
        const int mt = tile / NT, nt = tile - mt * NT;
        m0 = mt * BM;
        n0 = nt * BN;
    };

    int vb = blockIdx.x;
    if (vb >= ntiles) return;
    if (ML_SETPRIO == 2 && grp == 1) __builtin_amdgcn_s_setprio(1);  // grp is wave-uniform (readfirstlane)
    // optional timeline (bring-up builds only, -DML_DENSE_TRACE): slot ts of this wave <- s_memtime
#ifdef ML_DENSE_TRACE
    unsigned long long* const trc = p.trace ? p.trace + ((size_t)blockIdx.x * 8 + w) * 64 : nullptr;
    int tslot = 0;
    auto stamp = [&]() {
        if (trc && tslot < 64) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) trc[tslot] = t;
            ++tslot;
        }
    };
#else
    auto stamp = [&]() {};
#endif
    stamp();
    int m0, n0;
    tile_of(vb, m0, n0);
    const char* wtile = p.w + (size_t)n0 * rowb;
    const char* xtile = p.x + (size_t)m0 * rowb;
    auto issueX = [&](int kt) { issue4(xtile, xbase, TILE_BYTES, kt); };
    auto issueW = [&](int kt) { issue4(wtile, wbase, 0, kt); };
    auto issueXq = [&](int kt, int qa, int qb) { issue_q(xtile, xbase, TILE_BYTES, kt, qa, qb); };
    auto issueWq = [&](int kt, int qa, int qb) { issue_q(wtile, wbase, 0, kt, qa, qb); };
    // schedule B: an L phase carries only NL of the 4 DMA instructions of an operand share, the others ride
    // between the MFMAs of the following C phase (an L phase with 4 of them is ~40 % longer than a C phase)
    constexpr bool schedB = ML_SCHED_B != 0;
    constexpr int NL = ML_SCHED_B == 2 ? 2 : 3;  // DMA instructions left in an L phase
    if (nk > 0 && loads) {
        issueX(0);
        issueW(0);
    }
    bool first = true;
    int staged_n0 = -1;

    while (true) {
        // accumulators start at bias * 2^e (pre-scaled on the host, exact), so the epilogue needs no bias
        // and no load: (bias*2^e + sum) * 2^-e.  Register r of MFMA tile `it` is weight row
        // n = nbase + 32*it + (r&3) + 8*(r>>2) + 4*(lane>>5): one 16-byte load per (it, r>>2), straight into
        // the accumulator registers (the vmcnt(0) hipcc puts behind them is harmless here: the stage-0
        // DMA they queue behind was requested a whole epilogue ago).
        f32x16 acc[4][2];
        {
            const char* bs = (const char*)(p.bias_scaled + n0 + wn * 128);  // wave-uniform
            int il = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(il));  // lane id recomputed here: nothing lane-dependent stays live for this
            const unsigned hoff = (unsigned)(il >> 5) * 16u;
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *(const f32x4*)(bs + (it * 32 + g * 8) * 4 + hoff);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[it][0][g * 4 + e] = b4[e];
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < 4; ++it) acc[it][1] = acc[it][0];
            __builtin_amdgcn_sched_barrier(0);
        }

        // stage 0 of this tile was requested before the previous tile's epilogue stores (32 per wave);
        // vmcnt retires in order, so "at most 32 outstanding" means that DMA has landed.
        // (HEAD epilogues issue 2*ceil(HEAD/4) partial-sum stores instead)
        // (residual epilogues issue 7 x 4 residual DMA instructions behind it as well; vmcnt goes up to 63)
        constexpr int EPI_VMEM = HEAD > 0 ? 2 * ((HEAD + 3) / 4) : (RES ? 60 : 32);
        if (first) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EPI_VMEM) : "memory");
        }
        first = false;
        stamp();      // [tile*6 + 1] stage 0 landed (this wave)
        pp_barrier();
        stamp();      // [+2] everybody's stage 0 landed

        half8 whi[4], wlo[4], xhi[2], xlo[2];
        auto load_frags = [&](const char* sb, int kk) {
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
        };
        // hook (schedule B): the DMA instructions that ride between the MFMAs of a C phase, one per slot
        // (slot k = after the first 3 MFMAs of row block k):
        //   1 = group 0, first C phase : X(kt) quarters [NL,4) then W(kt) quarters [0,4-NL)
        //   2 = group 1, first C phase : W(kt) quarters [NL,4)
        //   3 = group 1, second C phase: X(kt) quarters [NL,4)
        auto compute = [&](int hook, int kt) {
            if (ML_SETPRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    if (schedB && hook) {
                        constexpr int NC = 4 - NL;                     // instructions per operand moved into C phases
                        const int count = (hook == 1) ? 2 * NC : NC;
                        // which of the `count` instructions sit in front of MFMA block (it, jt):
                        //   style 0: all of them before block (1,0);  style 1: one per row block, before (k+first, 1);
                        //   style 2: two per row block, before (1 + k/2, 0)
                        int k0 = 0, k1 = 0;
                        if (ML_HOOK_STYLE == 0) {
                            if (it == 1 && jt == 0) k1 = count;
                        } else if (ML_HOOK_STYLE == 1) {
                            const int first = (count == 4) ? 0 : 1;
                            if (jt == 1 && it - first >= 0 && it - first < count) { k0 = it - first; k1 = k0 + 1; }
                        } else {
                            if (jt == 0 && it >= 1 && 2 * (it - 1) < count) { k0 = 2 * (it - 1); k1 = (k0 + 2 < count) ? k0 + 2 : count; }
                        }
                        if (k1 > k0) {
                            __builtin_amdgcn_sched_barrier(0);
                            for (int k = k0; k < k1; ++k) {
                                if (hook == 1) {
                                    if (k < NC) issueXq(kt, NL + k, NL + k + 1);
                                    else issueWq(kt, k - NC, k - NC + 1);
                                } else if (hook == 2) {
                                    issueWq(kt, NL + k, NL + k + 1);
                                } else {
                                    issueXq(kt, NL + k, NL + k + 1);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (NSPLIT == 3) {
                        acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[it], xlo[jt], acc[it][jt], 0, 0, 0);
                        acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[it], xhi[jt], acc[it][jt], 0, 0, 0);
                    }
                    if (NSPLIT == 0)
                        acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, whi[it]),
                                                                              __builtin_bit_cast(bf16x8, xhi[jt]), acc[it][jt], 0, 0, 0);
                    else
                        acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[it], xhi[jt], acc[it][jt], 0, 0, 0);
                }
            if (ML_SETPRIO == 1) __builtin_amdgcn_s_setprio(0);
        };

        // Phase p (after the tile-start barrier): group 0 reads half-step h in phase 2h and computes it in
        // 2h+1; group 1 does the same one phase later.  Stage s (half-steps 2s, 2s+1) is read in phases
        // 4s..4s+3 and its buffer previously held stage s-2, last read in phase 4s-5.  Its four quarters
        // are requested in four consecutive phases (schedule A: all 4 instructions of a quarter in the L phase
        // named here; schedule B, the default: NL of them there, the rest between the MFMAs of the same wave's
        // next C phase -- group 0's W share starts one phase earlier so that nothing is issued in phase 4s-1):
        //   X rows 128.. : group 1, end of phase 4s-5 (after its own last reads of stage s-2)   window 5
        //   X rows   0.. : group 0, phase 4s-4                                                  window 4
        //   W rows   0.. : group 1, phase 4s-3                                                  window 3
        //   W rows 128.. : group 0, phase 4s-2                                                  window 2
        // and everything of stage s has landed by the end of phase 4s-1 (counted waits below).
        // One loop body for both groups (two copies would get two register assignments and 128 accumulator
        // copies at the merge); the groups differ only in the wave-uniform DMA issue / wait statements.
        if (grp == 1) {
            if (loads && nk > 1) issueX(1);                    // phase 0 (group 1 has nothing else to do in it)
            pp_barrier();
        }
        for (int t = 0; t < nk; ++t) {
            const char* sb = smem + (t & 1) * STAGE_BYTES;
            const bool pre = loads && t + 1 < nk;
            const bool nxt = loads && t + 2 < nk;
            if constexpr (!schedB) {
                if (pre) {                                         // G0: phase 4t     G1: phase 4t+1
                    if (grp == 0) issueX(t + 1);
                    else issueW(t + 1);
                }
                load_frags(sb, 0);
                pp_barrier();
                compute(0, 0);                                     // G0: phase 4t+1   G1: phase 4t+2
                pp_barrier();
                if (pre && grp == 0) issueW(t + 1);                // G0: phase 4t+2
                load_frags(sb, 1);                                 //                  G1: phase 4t+3
                if (grp == 1) {
                    if (nxt) {
                        // stage t is now completely read (group 0 finished it a phase ago, our own reads are
                        // drained here): its X rows 128.. can already be overwritten with stage t+2
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        issueX(t + 2);
                    }
                    if (pre) {
                        if (nxt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    if (t < 2) stamp();
                }
                pp_barrier();
                compute(0, 0);                                     // G0: phase 4t+3   G1: phase 4t+4
                if (grp == 0) {
                    if (pre) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (t < 2) stamp();                            // [+3], [+4] end of k-step 0 / 1 incl. DMA wait
                }
                pp_barrier();
            } else {
                // schedule B.  Per wave and k-step, in program (= vmcnt) order:
                //   G0: X(t+1) q[0,NL) | C: X(t+1) q[NL,4), W(t+1) q[0,4-NL) | W(t+1) q[4-NL,4) | C: -     then vmcnt(0)
                //   G1: W(t+1) q[0,NL) | C: W(t+1) q[NL,4)  | X(t+2) q[0,NL), vmcnt(NL) | C: X(t+2) q[NL,4)
                // (t = 0: group 1 issued all of X(1) before the loop, so its first vmcnt(3) also covers that)
                if (ML_L_ORDER) load_frags(sb, 0);
                if (pre) {
                    if (grp == 0) issueXq(t + 1, 0, NL);
                    else issueWq(t + 1, 0, NL);
                }
                if (!ML_L_ORDER) load_frags(sb, 0);
                pp_barrier();
                compute(pre ? (grp == 0 ? 1 : 2) : 0, t + 1);
                pp_barrier();
                if (ML_L_ORDER) load_frags(sb, 1);
                if (pre && grp == 0) issueWq(t + 1, 4 - NL, 4);
                if (!ML_L_ORDER) load_frags(sb, 1);
                if (grp == 1) {
                    if (nxt) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        issueXq(t + 2, 0, NL);
                    }
                    if (pre) {
                        if (nxt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    if (t < 2) stamp();
                }
                pp_barrier();
                compute((grp == 1 && nxt) ? 3 : 0, t + 2);
                if (grp == 0) {
                    if (pre) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (t < 2) stamp();
                }
                pp_barrier();
            }
        }
        if (grp == 0) pp_barrier();  // group 1's last compute phase
        // every wave has passed the same number of barriers; nobody reads the stage buffers any more
        stamp();      // [+5] main loop done

        // ---- epilogue of the finished tile, one 32 (n) x 32 (m) MFMA tile per pass
        const int cur_m0 = m0, cur_n0 = n0;
        const int nbase = cur_n0 + wn * 128;
        auto line0_of = [&](int pass) {
            const int jt = pass >> 2, it = pass & 3;
            return (size_t)(cur_m0 + wm * 64 + jt * 32) * yrowb + (size_t)(nbase + it * 32) * 4;
        };
        // DMA of one pass' residual tile (32 rows x 128 B): lane -> (row = id/8, pos = id%8), source chunk
        // pos ^ (row&7) so that the LDS image carries the same swizzle as the output scratch
        // (addresses = wave-uniform 64-bit base + one 32-bit lane offset, made opaque per tile so that hipcc
        // neither hoists 32 per-lane 64-bit addresses out of the tile loop nor spills them)
        // lane id recomputed and made opaque: every lane-dependent epilogue address is derived from it HERE,
        // per tile, instead of being hoisted out of the tile loop and kept live through the main loop
        int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(elane));
        const int eml = elane & 31, eh = elane >> 5;
        const unsigned res_goff = (unsigned)((elane >> 3) * (int)yrowb + (((elane & 7) ^ ((elane >> 3) & 7)) * 16));
        const unsigned st_off = (unsigned)((elane >> 3) * (int)yrowb + ((elane & 7) * 16));
        const int scr_row = eml * LINE + eh * 8;                       // + ((chunk ^ (eml&7)) * 16)
        const int rd_off = (elane >> 3) * LINE + (((elane & 7) ^ ((elane >> 3) & 7)) * 16);  // + qq*1024
        auto fetch_res = [&](int pass) {
            // debug bit 16 (timing only): every tile reads the residual of the first row panel -> L2 resident
            const char* src = p.res + (ML_DBG(p, 16) ? (line0_of(pass) % ((size_t)BM * yrowb)) : line0_of(pass));
            char* dst = resbuf + (pass & 1) * 4096;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) glds16(src + (size_t)(qq * 8) * yrowb + res_goff, dst + qq * 1024);
        };
        const bool has_res = RES && !ML_DBG(p, 1);
        if (has_res) fetch_res(0);  // ahead of the next tile's DMA so that it does not queue behind it
        if (HEAD > 0) {
            // this wave's slice of the head weights, hw[o][128] = head_w[o][nbase .. nbase+127].  Rows 0..7 live
            // in the wave's 4 KiB epilogue scratch (not needed for a transpose in this mode) and are re-staged
            // only when the column tile changes (it does not for the usual grids); a 9th row (stereo) goes to
            // the idle stage-1 buffer every tile.  Ordinary loads: issued BEFORE the next tile's DMA is
            // requested, so the vmcnt(0) hipcc puts behind them has nothing young to wait for.
            if (cur_n0 != staged_n0) {
                for (int idx = elane; idx < (HEAD < 8 ? HEAD : 8) * 32; idx += 64) {
                    const int o = idx >> 5, c4 = idx & 31;
                    *(f32x4*)((float*)scr + o * 128 + c4 * 4) = *(const f32x4*)(p.head_w + (size_t)o * p.N + nbase + c4 * 4);
                }
                staged_n0 = cur_n0;
            }
            if (HEAD > 8) {
                for (int idx = elane; idx < (HEAD - 8) * 32; idx += 64) {
                    const int o = 8 + (idx >> 5), c4 = idx & 31;
                    *(f32x4*)((float*)resbuf + (o - 8) * 128 + c4 * 4) = *(const f32x4*)(p.head_w + (size_t)o * p.N + nbase + c4 * 4);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }

        // ---- next tile: request its first stage now, it lands while the epilogue runs
        vb += gridDim.x;
        const bool more = vb < ntiles;
        const bool pref = more && nk > 0 && loads;
        if (more) {
            tile_of(vb, m0, n0);
            wtile = p.w + (size_t)n0 * rowb;
            xtile = p.x + (size_t)m0 * rowb;
            if (pref) {
                issueX(0);
                issueW(0);
            }
        }

        if (ML_DBG(p, 1)) {
            float s = 0.f;
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) s += acc[it][jt][e];
            if (s == 123456.789f) p.y[tid] = 1;
        } else if (HEAD > 0) {
            const int slice = (cur_n0 / BN) * 2 + wn;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
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
                            v[e] = acc[it][jt][g * 4 + e] * p.descale;
                            if (RELU) v[e] = __builtin_fmaxf(v[e], 0.0f);
                        }
#pragma unroll
                        for (int o = 0; o < HEAD; ++o) {
                            const float* hwo = (o < 8 ? (const float*)scr + o * 128 : (const float*)resbuf + (o - 8) * 128);
                            const f32x4 w4 = *(const f32x4*)(hwo + it * 32 + g * 8 + eh * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) part[o] = __builtin_fmaf(v[e], w4[e], part[o]);
                        }
                    }
#pragma unroll
                for (int o = 0; o < HEAD; ++o) part[o] += __shfl_xor(part[o], 32, 64);
                if (eh == 0) {
                    float* dst = p.head_part + ((size_t)slice * p.M_pad + (cur_m0 + wm * 64 + jt * 32 + eml)) * 16;
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
        } else {
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                const int jt = pass >> 2, it = pass & 3;
                const size_t line0 = line0_of(pass);
                u32x2 rh[4], rl[4];
                if (RES) {
                    const char* rb = resbuf + (pass & 1) * 4096;
                    if (pass + 1 < 8) fetch_res(pass + 1);  // one pass ahead, ahead of this pass's stores
                    // wait for this pass' residual DMA: count the younger vector-memory operations
                    if (pass == 0) {
                        if (pref) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // 8 stage DMA + 4 res DMA
                        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    } else if (pass < 7) {
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // 4 stores + 4 res DMA
                    } else {
                        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // 4 stores
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        rh[g] = *(const u32x2*)(rb + scr_row + ((g ^ (eml & 7)) * 16));
                        rl[g] = *(const u32x2*)(rb + scr_row + (((g + 4) ^ (eml & 7)) * 16));
                    }
                }
                u32x2 oh[4], ol[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const float a0 = acc[it][jt][g * 4 + 2 * e2], a1 = acc[it][jt][g * 4 + 2 * e2 + 1];
                        unsigned hh, ll;
                        if (NSPLIT == 0) {
                            ll = 0u;
                            if (RES) bf16_2_res<RELU>(a0, a1, p.descale, rh[g][e2], hh);
                            else bf16_2_scaled<RELU>(a0, a1, p.descale, hh);
                        } else if (RES) split2_res<RELU>(a0, a1, p.descale, rh[g][e2], rl[g][e2], hh, ll);
                        else split2_scaled<RELU>(a0, a1, p.descale, lim, hh, ll);
                        oh[g][e2] = hh;
                        ol[g][e2] = ll;
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    *(u32x2*)(scr + scr_row + ((g ^ (eml & 7)) * 16)) = oh[g];
                    *(u32x2*)(scr + scr_row + (((g + 4) ^ (eml & 7)) * 16)) = ol[g];
                }
                __builtin_amdgcn_wave_barrier();
                f32x4 d[4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) d[qq] = *(const f32x4*)(scr + rd_off + qq * 1024);
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)  // rows 8qq + lane/8, chunk lane%8
                    *(f32x4*)(p.y + line0 + (size_t)(qq * 8) * yrowb + st_off) = d[qq];
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);  // keep the passes apart: interleaving them costs registers
                // the scratch is rewritten by the next pass only after these reads returned (in-order LDS);
                // the residual buffer of this pass is re-filled by the DMA issued at the top of the next pass
            }
        }
        stamp();      // [+6] epilogue issued
        if (!more) break;
    }
}

}  // namespace mlk
