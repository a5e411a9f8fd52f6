// dense_mid.h -- the dense layer for MID-SIZE batches (a few thousand to ~16k rows: a video batch, a handful of
// camera streams; reference use: Loco.forward over the persons of several frames, monoloco/network/net.py:83-133).
//
// The 256x256-tile persistent kernels (dense_kernel_pp / _w4) have (rows/256) * (N/256) tiles: 64 at 4096 rows, 128 at
// 8192 -- a quarter / half of the 256 CUs, whatever the kernel does inside a tile.  The small-row kernels
// (dense_small.h) have tiles enough but re-read their operands from L2 once per 32x32 tile (1 GB per layer at 4096
// rows).  This kernel sits between the two: workgroup tile 128 (n) x TM (m), TM = 64 or 128, 4 waves as 2 (n) x 2 (m),
// wave tile 64 x TM/2 = 2 x NB MFMA 32x32x16 tiles, same arithmetic as the other dense kernels (k32 hi|lo lines,
// hi*lo + lo*hi + hi*hi in fp32, A operand = weights so that a lane ends up with 4 consecutive n of one person).
//
//   grid    (N/128) * (M_pad/TM) workgroups, numbered so that the 8 column tiles of a row panel land on ONE XCD (its L2
//           then serves the activation panel 8 times); TM = 64 while the 128-row tiles are fewer than the CUs (< 4096 rows at N = 1024)
//   stage   one k32 line per row: 128 W rows + TM X rows = 24 / 32 KiB; 2 stages in LDS, chunk ^= (row>>1)&7 as in
//           dense_kernel.h (conflict-free ds_read_b128 fragment reads)
//   loader  register-staged: every thread moves 4 (W) + TM/32 (X) 16-byte chunks per step, global -> VGPR two steps
//           ahead (two register sets), VGPR -> LDS one step ahead; one workgroup barrier per step.  (LDS-DMA costs its
//           wave ~100 cycles of issue per instruction -- dense_kernel_pp.h -- and the two or three resident workgroups
//           per CU hide the latency a single wave per SIMD cannot.)
//   LDS     reads per step: 4 waves * 2 half-steps * (2 + NB) blocks * 2 KiB; writes 16 + TM/8 KiB: 768 cycles at
//           128 B/clk for TM = 128 (= the 768 MFMA cycles per SIMD), 576 vs 384 for TM = 64 -- the small tile is
//           LDS-bound at 2/3 of the MFMA rate, the price of the tile count
//   epilogue per 32x32 MFMA tile as in dense_kernel_pp: bias rides in the accumulators, * 2^-e, ReLU, + residual, split
//           to fp16 hi|lo, transposed through a private 4 KiB LDS scratch, four 16-byte stores per lane (whole 128-byte
//           lines); the residual comes straight from global memory in the accumulator layout (8-byte loads).
// HEAD (round 5): the heads ride in this kernel's epilogues as in the tile kernels' -- 8 | 9: w_fin's partial sums over this wave's
// 64 columns instead of the activation tile (the layer that feeds w_fin is never stored); -1: w_aux's partial sums beside the
// stored tile (of the stored hi + lo values); head_part holds N/64 slices per head here (the tile kernels: N/128), tail_mono_kernel /
// head_reduce_kernel add them in slice order.  0: no head (heads_pair_kernel or launch_heads behind the last layer).
// SPLITK (round 6): the reduction of ONE output tile is cut into p.ksplit contiguous k ranges, one workgroup each -- a 1024 x 1024 layer
// at 2048 rows is 256 tiles = one workgroup per CU, whose single wave per SIMD pays every wait in full (fragment read -> MFMA -> barrier:
// ~1100 cycles per k-step for 384 cycles of MFMA, 384 of LDS pipe and 384 of vector-memory pipe).  With 2-4 k ranges per tile the chip holds
// 512 workgroups, two or three per CU, and one's waits hide behind another's MFMAs.  The epilogue (ReLU, residual, hi|lo split, fused heads)
// needs the COMPLETE sum: every workgroup stores its fp32 partial tile (thread-linear float4s: coalesced) to p.kpart with agent-scope
// write-through stores (the XCDs' L2s are not coherent with each other; no fence -- see below) and takes a ticket from the tile's counter;
// whoever draws the last ticket reads the others' partials with agent-scope loads, adds them IN SPLIT ORDER (its own from registers at its
// own position: the bits do not depend on who arrives last), runs the unchanged epilogue and re-arms the counter for the next launch.  Nobody ever waits for anybody: no co-residency
// assumption, no deadlock.  Split 0 starts from the bias, the others from zero.
#pragma once
#include "dense_kernel_pp.h"
#include "geom_kernels.h"   // cam_row, NKP / KPS_ROW / NIN: the fused pre-process of the input layer (PREP)

// timing ablations of the LDS-DMA loop, COMPILE-TIME (-DML_MID_ABL=<bits>, results are garbage): 1 no requests inside the loop, 2 no MFMAs,
// 4 no fragment reads, 8 no barrier / vmcnt wait per step, 16 no epilogue stores
#ifndef ML_MID_ABL
#define ML_MID_ABL 0
#endif
#define MID_DBG(bit) (((ML_MID_ABL) & (bit)) != 0)

namespace mlk {

constexpr int MID_TN = 128;
constexpr int MID_THREADS = 256;

template <int TM>
struct MidCfg {
    static constexpr int NB = TM / 64;                   // MFMA column (person) blocks per wave
    static constexpr int XL = TM / 32;                   // 16-byte X chunks a thread moves per step
    static constexpr int W_BYTES = MID_TN * LINE;        // 16 KiB
    static constexpr int STAGE = W_BYTES + TM * LINE;    // W rows then X rows
    static constexpr int LDS = 2 * STAGE;
    // The LDS-DMA loader's ring: three stages (one being read, one in flight, one being requested).  -DMID_RING=<n> builds deeper rings
    // for the A/B: as many stages as the CU's 160 KiB hold (6 x 24 / 5 x 32 KiB, one workgroup per CU) measured SLOWER -- 2048 rows 144.8 vs
    // 133.8 us per forward, 4096 rows 224.6 vs 211.4 (profiles/r06_ablation.md): the loop is not short of requests in flight
#ifndef MID_RING
#define MID_RING 3
#endif
    static constexpr int RING = MID_RING;
    static constexpr int LDS_DMA = RING * STAGE;
    static constexpr int NI = (MID_TN + TM) / 32;        // LDS-DMA instructions (8 rows each) per wave and step
};

// DMA (round 6): the loader is LDS-DMA (global_load_lds_dwordx4 straight into a ring of three stages, the chunk swizzle applied on the
// source address) instead of global -> VGPR -> ds_write.  Why: the register-staged loop is FEED-bound, not LDS- or latency-bound -- a
// 128 x 64 tile pulls (128 + 64) x 4 KiB = 768 KiB through its CU in 20 us = 18 B/clk, exactly what tools/ubench/feed.hip measures for
// `global_load_dwordx4 -> VGPR` with every CU pulling (18.5 B/clk/CU, 11 TB/s chip-wide; profiles/r01_ablation.md), which is why deeper
// prefetch, pipelined fragment reads (r05_ablation.md section 5) and two co-resident workgroups per CU (split-K, r06_ablation.md) all
// changed nothing; the LDS-DMA path delivers 34-36 B/clk/CU in the same micro-benchmark.  One raw s_barrier per step; the stage read in
// step i - 1 is refilled with step i + 2 right behind step i's barrier; every wave waits for its OWN share of a stage with a counted
// vmcnt(NI) (requests retire in order; past the end the requests repeat the last step so that the count never changes).
// PREP (round 6, the input layer of the mono pipeline, K = 64): the workgroup computes its TM persons' network inputs itself --
// preprocess_monoloco (reference process.py:47-67 = pixel_to_camera, utils/camera.py:10-29, of the 17 keypoints at z = 10) with
// prep_kernel's very arithmetic (cam_row: the same bits) -- and writes them as hi|lo lines straight into the two stages' X rows; only the
// weight rows are requested.  prep_kernel and its 8 + 0.5 bytes per person and input of HBM round trip disappear from the mid window
// (5.6 us per forward); the column tile 0 workgroups also leave the box centres (get_keypoints(.., 'center'), camera.py:82-86).
template <int NSPLIT, bool RELU, bool RES, int TM, int HEAD = 0, bool SPLITK = false, bool DMA = false, bool PREP = false>
__global__ __launch_bounds__(MID_THREADS, (DMA && MidCfg<TM>::LDS_DMA > 80 * 1024) ? 1 : 2) void dense_mid_kernel(DenseParams p) {   // (a ring above 80 KiB: one workgroup per CU)
    typedef MidCfg<TM> C;
    constexpr int NB = C::NB, XL = C::XL;
    static_assert(!PREP || (DMA && !SPLITK && !RES && HEAD == 0 && MidCfg<TM>::RING >= 3), "the fused pre-process: the plain input layer on the LDS-DMA loader");
    constexpr bool AUX = HEAD == -1;
    static_assert(HEAD == 0 || HEAD == -1 || ((HEAD == 8 || HEAD == 9) && RELU && !RES), "fused head: w_aux (-1) or w_fin (8 | 9) behind relu, no residual");
    __shared__ __attribute__((aligned(16))) char smem[DMA ? C::LDS_DMA : C::LDS];
    // (the split-K ticket lives in the first word of the stage buffers, free behind the loop: a SECOND LDS object makes hipcc guard
    //  every fragment read that follows an LDS-DMA request with its own vmcnt(0) -- seen in the ISA, tools/check_loops.py)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 1, wm = w >> 1;

    // workgroup -> tile: blockIdx round-robins over the 8 XCDs; XCD x takes the tiles [x * per, (x+1) * per), walked
    // column tile fastest, so that the workgroups resident on an XCD share a few row panels of X
    const int tiles_n = p.N / MID_TN;
    const int tiles = tiles_n * (p.M_pad / TM);
    const int S = SPLITK ? p.ksplit : 1;
    const int units = tiles * S;             // work items: (row panel, k range, column tile), column tile fastest: the 8 workgroups that run
    const int per = (units + 7) >> 3;        // side by side on an XCD share one k range of one X panel
    const int unit = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (unit >= units) return;
    const int nt = unit % tiles_n;
    const int split = SPLITK ? (unit / tiles_n) % S : 0;
    const int mt = unit / (tiles_n * S);
    const int tile = mt * tiles_n + nt;
    const int n0 = nt * MID_TN;
    const int m0 = mt * TM;

    const size_t rowb = (size_t)p.K * 4;
    const int nk = (p.K / 32) / S;           // k32 steps of this work item (the host guarantees divisibility)
    const size_t kbase = (size_t)split * nk * LINE;
    const float descale = p.descale_ptr ? *p.descale_ptr : p.descale;

    // loader: thread -> (row lrow + 32 j, chunk lch) of both operands
    const int lrow = tid >> 3, lch = tid & 7;
    const char* gw = p.w + (size_t)(n0 + lrow) * rowb + lch * 16 + kbase;
    const char* gx = p.x + (size_t)(m0 + lrow) * rowb + lch * 16 + kbase;
    const int lst = lrow * LINE + ((lch ^ ((lrow >> 1) & 7)) * 16);   // + 32 j rows (the swizzle term repeats every 16 rows)

    struct Raw {
        f32x4 w[4];
        f32x4 x[XL];
    };
    auto gload = [&](Raw& r, int k) {
        const int kk = k < nk ? k : nk - 1;   // past the end: a harmless repeat (unconditional loads keep vmcnt countable)
        const size_t o = (size_t)kk * LINE;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.w[j] = *(const f32x4*)(gw + (size_t)(32 * j) * rowb + o);
#pragma unroll
        for (int j = 0; j < XL; ++j) r.x[j] = *(const f32x4*)(gx + (size_t)(32 * j) * rowb + o);
    };
    auto lstore = [&](const Raw& r, int s) {
        char* b = smem + s * C::STAGE + lst;
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(b + j * 32 * LINE) = r.w[j];
#pragma unroll
        for (int j = 0; j < XL; ++j) *(f32x4*)(b + C::W_BYTES + j * 32 * LINE) = r.x[j];
    };

    // fragments: lane (r, q) reads row r of a 32-row block, chunk 2s+q (hi) / 4+2s+q (lo) of the line
    const int r = lane & 31, q = lane >> 5;
    const int sw = (r >> 1) & 7;
    const int fa_row = (wn * 64 + r) * LINE;                       // + a * 32 rows
    const int fb_row = C::W_BYTES + (wm * (TM / 2) + r) * LINE;    // + b * 32 rows
    struct Frag {
        half8 whi[2], wlo[2], xhi[NB], xlo[NB];
    };
    auto fread = [&](Frag& f, int s, int hs) {
        const char* b = smem + s * C::STAGE;
        const int chi = ((2 * hs + q) ^ sw) * 16, clo = ((4 + 2 * hs + q) ^ sw) * 16;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            f.whi[a] = *(const half8*)(b + fa_row + a * 32 * LINE + chi);
            if (NSPLIT == 3) f.wlo[a] = *(const half8*)(b + fa_row + a * 32 * LINE + clo);
        }
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            f.xhi[c] = *(const half8*)(b + fb_row + c * 32 * LINE + chi);
            if (NSPLIT == 3) f.xlo[c] = *(const half8*)(b + fb_row + c * 32 * LINE + clo);
        }
    };

    f32x16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 b4 = *(const f32x4*)(p.bias_scaled + n0 + wn * 64 + a * 32 + 8 * g + 4 * q);
            if (SPLITK && split != 0) b4 = f32x4{0.f, 0.f, 0.f, 0.f};   // the bias enters the sum once, with the first k range
#pragma unroll
            for (int c = 0; c < NB; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[a][c][g * 4 + e] = b4[e];
        }
    auto mma_rows = [&](const Frag& f, int a) {   // the MFMAs of weight-row block a
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if (NSPLIT == 3) {
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.whi[a], f.xlo[c], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wlo[a], f.xhi[c], acc[a][c], 0, 0, 0);
                }
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.whi[a], f.xhi[c], acc[a][c], 0, 0, 0);
            }
    };

    if (DMA) {
        constexpr int NI = C::NI;
        // instruction j of wave w fetches row group gq = w + 4 j of the stage (8 rows x 128 B = 1 KiB, lane-linear in LDS): groups
        // 0..15 are the 128 W rows, the rest the TM X rows; lane -> row 8 gq + lane / 8, position lane % 8, source chunk = position ^ swizzle(row)
        const char* gsrc[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int gq = w + 4 * j;
            const bool is_w = gq < MID_TN / 8;   // (wave-uniform)
            const int rr = 8 * gq + (lane >> 3) - (is_w ? 0 : MID_TN);
            const char* base = (is_w || PREP) ? p.w + (size_t)(n0 + (is_w ? rr : 0)) * rowb : p.x + (size_t)(m0 + rr) * rowb;
            gsrc[j] = base + kbase + (((lane & 7) ^ ((rr >> 1) & 7)) * 16);
        }
        // instructions [ja, jb) of this wave's share of stage `st` <- k-step k
        auto issue = [&](int st, int k, int ja, int jb) {
            const int kk = k < nk ? k : nk - 1;   // past the end: a harmless repeat (the vmcnt arithmetic stays constant)
            char* sb = smem + st * C::STAGE + w * 1024;
#pragma unroll
            for (int j = 0; j < NI; ++j)
                if (j >= ja && j < jb && !(PREP && j >= 4)) glds16(gsrc[j] + (size_t)kk * LINE, sb + j * 4096);   // (PREP: the W rows only)
        };
        // the bias loads above are the only ordinary vector loads in front of the epilogue: complete them HERE (left alone, hipcc
        // sinks their wait to the first MFMA inside the loop and, with LDS-DMA in the same queue, makes it a vmcnt(0) per step --
        // seen in the split-K instantiations' ISA, tools/check_loops.py)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < NB; ++c) asm volatile("" : "+v"(acc[a][c]));
        constexpr int RING = C::RING;
        if (PREP) {
            // K = 64: two k-steps, both stages requested now (W rows), nothing inside the loop; stage 2 is the keypoint scratch
            issue(0, 0, 0, NI);
            issue(1, 1, 0, NI);
            float* const s_in = (float*)(smem + 2 * C::STAGE);                 // TM persons x 51 floats (u[17], v[17], conf[17])
            const int64_t left = (int64_t)p.prep_m - m0;
            const int nvalid = left < 0 ? 0 : (left < TM ? (int)left : TM);
            const float* src = p.prep_kps + (size_t)m0 * KPS_ROW;
            for (int i = tid; i < nvalid * KPS_ROW; i += MID_THREADS) s_in[i] = src[i];
            __syncthreads();
            if (p.prep_centre && nt == 0 && tid < nvalid) {                     // get_keypoints(.., 'center'): (max - min) / 2 + min
                const float* u = s_in + tid * KPS_ROW;
                const float* v = u + NKP;
                float umin = u[0], umax = u[0], vmin = v[0], vmax = v[0];
#pragma unroll
                for (int j = 1; j < NKP; ++j) {
                    umin = __builtin_fminf(umin, u[j]);
                    umax = __builtin_fmaxf(umax, u[j]);
                    vmin = __builtin_fminf(vmin, v[j]);
                    vmax = __builtin_fmaxf(vmax, v[j]);
                }
                p.prep_centre[(size_t)(m0 + tid) * 2 + 0] = __fadd_rn(__fmul_rn(__fsub_rn(umax, umin), 0.5f), umin);
                p.prep_centre[(size_t)(m0 + tid) * 2 + 1] = __fadd_rn(__fmul_rn(__fsub_rn(vmax, vmin), 0.5f), vmin);
            }
            // one 16-byte chunk of a line per thread and pass: person pi, line b (= k-step = stage), chunk sub (< 4: hi, else lo) of
            // k = 32 b + 8 (sub & 3) .. + 7; input k = 2 j + (0: x, 1: y) of joint j (process.py:65: interleaved), zero beyond 34 inputs
            // and beyond the valid rows -- prep_kernel's statement, value for value
            for (int id = tid; id < TM * 16; id += MID_THREADS) {
                const int pi = id >> 4, c = id & 15, b = c >> 3, sub = c & 7;
                const int k0 = b * 32 + (sub & 3) * 8;
                const float* u = s_in + pi * KPS_ROW;
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = k0 + e, j = k >> 1;
                    float val = 0.0f;
                    if (k < NIN && pi < nvalid) val = cam_row(u[j], u[NKP + j], p.prep_kinv + ((k & 1) ? 3 : 0), p.prep_z);
                    _Float16 hi, lo;
                    split_f16(val, hi, lo);
                    o[e] = (sub < 4) ? hi : lo;
                }
                *(half8*)(smem + b * C::STAGE + C::W_BYTES + pi * LINE + ((sub ^ ((pi >> 1) & 7)) * 16)) = o;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // both stages' W rows (the loop's counted waits find nothing outstanding)
        } else {
#pragma unroll
            for (int r0 = 0; r0 < RING; ++r0) issue(r0, r0, 0, NI);
        }
        // One request behind every unit of three MFMAs (one 32 x 32 output block's hi.lo + lo.hi + hi.hi: 96 matrix-pipe cycles against
        // ~110 of issue) -- TM = 128: eight units, eight requests; TM = 64: four units, six requests (2 + 2 + 1 + 1).
        auto mma_unit = [&](const Frag& f, int a, int c) {
            if (MID_DBG(2)) return;
            if (NSPLIT == 3) {
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.whi[a], f.xlo[c], acc[a][c], 0, 0, 0);
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wlo[a], f.xhi[c], acc[a][c], 0, 0, 0);
            }
            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.whi[a], f.xhi[c], acc[a][c], 0, 0, 0);
        };
        auto read_step = [&](Frag& g0, Frag& g1, int st) {
            if (!MID_DBG(4)) {
                fread(g0, st, 0);
                fread(g1, st, 1);
            } else {
                asm volatile("" : "=v"(g0.whi[0]), "=v"(g0.whi[1]), "=v"(g0.wlo[0]), "=v"(g0.wlo[1]));
                asm volatile("" : "=v"(g1.whi[0]), "=v"(g1.whi[1]), "=v"(g1.wlo[0]), "=v"(g1.wlo[1]));
#pragma unroll
                for (int c = 0; c < NB; ++c) asm volatile("" : "=v"(g0.xhi[c]), "=v"(g0.xlo[c]), "=v"(g1.xhi[c]), "=v"(g1.xlo[c]));
            }
        };
        // The fragment reads are pipelined ONE STEP AHEAD in registers (round 6; ablation: un-pipelined they cost 300-400 exposed cycles
        // per step between the barrier and the first MFMA -- 6.4 us of a 26 us layer at 4096 rows, profiles/r06_ablation.md).  Step i:
        //   wait: my share of stage i + 1 has landed (the requests of stages i + 2 .. i + RING - 1 are younger: vmcnt((RING - 2) NI)), my
        //         reads of stage i are complete
        //   barrier: everybody's share of stage i + 1; everybody is done reading stage i, whose buffer is therefore free
        //   read the fragments of stage i + 1 into the other register set; the MFMAs of stage i, with stage i + RING's requests
        //   (into stage i's buffer; past the end: harmless repeats, the count never changes) between them
        Frag fa0, fa1, fb0, fb1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 1) * NI) : "memory");
        pp_barrier();
        read_step(fa0, fa1, 0);
        int st = 0;   // buffer of stage i
        auto step_dma = [&](Frag& c0, Frag& c1, Frag& n0f, Frag& n1f, int i) {
            const int nst = st == RING - 1 ? 0 : st + 1;
            if (!MID_DBG(8)) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * NI) : "memory");
                pp_barrier();
            }
            read_step(n0f, n1f, nst);   // (unconditional: behind the last step it reads a landed repeat nobody uses -- a branch here makes hipcc wait lgkmcnt(0) at the join, in front of the MFMAs)
            int j = 0;
#pragma unroll
            for (int hs = 0; hs < 2; ++hs)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < NB; ++c) {
                        mma_unit(hs ? c1 : c0, a, c);     // (same order as mma_rows: the same bits as the register-staged loop)
                        const int unit = (hs * 2 + a) * NB + c, n_here = NB == 2 ? 1 : (unit < 2 ? 2 : 1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (!MID_DBG(1) && !PREP) issue(st, i + RING, j, j + n_here);
                        __builtin_amdgcn_sched_barrier(0);
                        j += n_here;
                    }
            st = nst;
        };
        for (int i = 0; i < nk; i += 2) {
            step_dma(fa0, fa1, fb0, fb1, i);
            if (i + 1 < nk) step_dma(fb0, fb1, fa0, fa1, i + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the repeats behind the end have landed: the stages become epilogue scratch
        pp_barrier();
    } else {
        Raw R0, R1;
        gload(R0, 0);
        gload(R1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + XL) : "memory");
        lstore(R0, 0);
        gload(R0, 2);
        __syncthreads();

        // step i: LDS stage i&1 holds k-step i; the set named `nxt` holds step i+1 (requested two steps ago), the other set
        // step i+2 (requested one step ago, stays in flight across the wait).  Schedules measured against this one (us per forward
        // at 4096 / 8192 rows, 128-row tiles; this one: 256-259 / 403-406):
        //   * the barrier in the middle of the step's MFMAs, the next step's first fragments requested right behind it: 271 / 422;
        //   * the loader without a branch (the `if (i + 1 < nk)` below makes hipcc's own waitcnt insertion drain ALL requests,
        //     vmcnt(0), before every second LDS store; branch-free it waits for exactly the set it stores, vmcnt(8)): 256 / 428 --
        //     no gain with one workgroup per CU, a loss with two;
        //   * three register sets (requests three steps ahead), loader unconditional, MFMAs of padded steps skipped: 267 / 425;
        //   * round 5, for ONE 128 x 64 tile per CU (<= 2048 rows): three register sets behind two stages (148 vs 150 us per forward at
        //     1024 rows, 160 vs 161 at 2048: -1 %), and three LDS stages + three sets with the next step's first fragments read behind
        //     this step's first MFMAs, branch-free groups of six steps so that hipcc's waits stay counted (vmcnt(12) in front of every
        //     store, checked in the ISA): 150.7 vs 150.0 at 1024 rows, 158 vs 156 at 2048 -- neither the request latency nor the
        //     fragment round trip is what a step waits for; the step is LDS-pipe time (12 reads + 6 writes of 1 KiB per wave and step)
        //     plus a barrier.  Both removed again (profiles/r05_ablation.md section 5).
        // With two workgroups per CU the kernel is bound by the LDS pipe both of them feed through (reads + writes = the 768 MFMA
        // cycles of a step at 128 B/clk; profiles/r03_mid_pmc_rows8192.txt: matrix pipe 39-44 % busy, 16 % of the wave cycles waiting on
        // LDS, 3.5 % bank conflicts), not by request latency.
        auto step = [&](Raw& nxt, int i) {
            const int s = i & 1;
            Frag f0, f1;
            fread(f0, s, 0);
            fread(f1, s, 1);
            mma_rows(f0, 0);
            mma_rows(f0, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + XL) : "memory");
            if (i + 1 < nk) lstore(nxt, s ^ 1);
            gload(nxt, i + 3);
            mma_rows(f1, 0);
            mma_rows(f1, 1);
            __syncthreads();
        };
        for (int i = 0; i < nk; i += 2) {
            step(R1, i);
            if (i + 1 < nk) step(R0, i + 1);
        }
    }

    if (SPLITK) {
        // ---- the partial tile of this k range -> p.kpart[tile][split][quad][thread] (float4 per thread and quad: whole 4 KiB rows).
        // The XCDs' L2s are not coherent with each other, and a release / acquire FENCE at agent scope writes back and invalidates a whole
        // L2 (measured: a forward 4-7x slower with __threadfence() -- every workgroup threw away the weights its neighbours were reading).
        // So the partials never live in a non-coherent cache line at all: stores and loads carry the agent-scope bit (sc1: write-through
        // to / read from the device-coherent level), which is what an agent-scope relaxed atomic access compiles to; a store's vmcnt
        // return means it is performed at that scope.  Order: my stores performed (vmcnt 0) -> workgroup barrier -> the ticket (a relaxed
        // agent-scope atomic) -> barrier -> the last arriver's loads.
        constexpr int QUADS = 2 * NB * 4;
        const f32x4* const all = (const f32x4*)p.kpart + (size_t)tile * S * (QUADS * MID_THREADS) + tid;
        f32x4* const mine = (f32x4*)all + (size_t)split * (QUADS * MID_THREADS);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < NB; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[a][c][g * 4], acc[a][c][g * 4 + 1], acc[a][c][g * 4 + 2], acc[a][c][g * 4 + 3]};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(mine + ((a * NB + c) * 4 + g) * MID_THREADS), "v"(v) : "memory");
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        volatile unsigned* const ticket_s = (volatile unsigned*)smem;
        if (tid == 0) *ticket_s = __hip_atomic_fetch_add(p.kcount + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = *ticket_s;
        __syncthreads();                             // (everybody has read it: the epilogue's scratch may overwrite the word)
        if (ticket != (unsigned)(S - 1)) return;   // (workgroup-uniform) not the last k range to finish: done
        // the partials IN SPLIT ORDER, whoever reduces (the bits do not depend on the arrival order): the ranges before mine summed
        // first, then mine (a + b == b + a exactly), then the ranges behind it
        f32x4 pre[QUADS], tmp[QUADS];
        auto fetch = [&](f32x4* dst, int s2) {
#pragma unroll
            for (int qd = 0; qd < QUADS; ++qd)
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst[qd]) : "v"(all + ((size_t)s2 * QUADS + qd) * MID_THREADS) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int qd = 0; qd < QUADS; ++qd) asm volatile("" : "+v"(dst[qd]));   // (values are defined from here on)
        };
        auto add_to_acc = [&](const f32x4* src) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < NB; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[a][c][g * 4 + e] += src[(a * NB + c) * 4 + g][e];
        };
        if (split > 0) {
            fetch(pre, 0);
            for (int s2 = 1; s2 < split; ++s2) {
                fetch(tmp, s2);
#pragma unroll
                for (int qd = 0; qd < QUADS; ++qd) pre[qd] += tmp[qd];
            }
            add_to_acc(pre);
        }
        for (int s2 = split + 1; s2 < S; ++s2) {
            fetch(tmp, s2);
            add_to_acc(tmp);
        }
        if (tid == 0) __hip_atomic_store(p.kcount + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
    }

    // ---- epilogue: the stage buffers are free behind the last barrier; wave w takes 4 KiB of them as its scratch
    char* scr = smem + w * 4096;
    const size_t yrowb = (size_t)p.N * 4;
    const float lim = 65504.0f / descale;
    const int eml = r, eh = q;
    const int nbase = n0 + wn * 64;            // this wave's 64 weight rows = head slice nbase / 64
    const int mbase = m0 + wm * (TM / 2);
    if (HEAD > 0) {
        // the activation tile is not stored: the wave multiplies its relu'd 64-column slice with the HEAD x 64 slice of the head
        // weights (staged in its scratch) and leaves one partial sum per person and output
        float* hw = (float*)scr;
        for (int idx = lane; idx < HEAD * 16; idx += 64) {
            const int o = idx >> 4, c4 = idx & 15;
            *(f32x4*)(hw + o * 64 + c4 * 4) = *(const f32x4*)(p.head_w + (size_t)o * p.N + nbase + c4 * 4);
        }
        __builtin_amdgcn_wave_barrier();
        const int slice = nbase >> 6;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            float part[HEAD > 0 ? HEAD : 1];
#pragma unroll
            for (int o = 0; o < HEAD; ++o) part[o] = 0.0f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaxf(acc[a][c][g * 4 + e] * descale, 0.0f);
#pragma unroll
                    for (int o = 0; o < HEAD; ++o) {
                        const f32x4 w4 = *(const f32x4*)(hw + o * 64 + a * 32 + g * 8 + eh * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) part[o] = __builtin_fmaf(v[e], w4[e], part[o]);
                    }
                }
#pragma unroll
            for (int o = 0; o < HEAD; ++o) part[o] += __shfl_xor(part[o], 32, 64);
            if (eh == 0) {
                float* dst = p.head_part + ((size_t)slice * p.M_pad + (mbase + c * 32 + eml)) * 16;
#pragma unroll
                for (int o4 = 0; o4 < (HEAD + 3) / 4; ++o4) {
                    f32x4 q4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) q4[e] = (o4 * 4 + e < HEAD) ? part[o4 * 4 + e] : 0.0f;
                    *(f32x4*)(dst + o4 * 4) = q4;
                }
            }
        }
        return;
    }
    const int scr_row = eml * LINE + eh * 8;
    const int rd_off = (lane >> 3) * LINE + (((lane & 7) ^ ((lane >> 3) & 7)) * 16);
    const size_t st_off = (size_t)(lane >> 3) * yrowb + (size_t)((lane & 7) * 16);
    float auxp[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) auxp[c] = 0.0f;
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const size_t line0 = (size_t)(mbase + c * 32) * yrowb + (size_t)(nbase + a * 32) * 4;
            u32x2 rh[4], rl[4];
            if (RES) {
                const char* rb = p.res + line0 + (size_t)eml * yrowb + eh * 8;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    rh[g] = *(const u32x2*)(rb + g * 16);
                    rl[g] = *(const u32x2*)(rb + g * 16 + 64);
                }
            }
            u32x2 oh[4], ol[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 aw = {0.f, 0.f, 0.f, 0.f};
                if (AUX) aw = *(const f32x4*)(p.head_w + nbase + a * 32 + 8 * g + 4 * eh);   // w_aux of this lane's 4 weight rows
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const float a0 = acc[a][c][g * 4 + 2 * e2], a1 = acc[a][c][g * 4 + 2 * e2 + 1];
                    unsigned hh, ll;
                    if (RES) split2_res<RELU>(a0, a1, descale, rh[g][e2], rl[g][e2], hh, ll);
                    else split2_scaled<RELU>(a0, a1, descale, lim, hh, ll);
                    oh[g][e2] = hh;
                    ol[g][e2] = ll;
                    if (AUX) {   // the stored value itself, hi + lo, times its head weight (dense_kernel_w4's statement)
                        float y0, y1;
                        asm("v_fma_mix_f32 %1, %3, 1.0, %4 op_sel_hi:[1,0,1]\n\t"
                            "v_fma_mix_f32 %2, %3, 1.0, %4 op_sel:[1,0,1] op_sel_hi:[1,0,1]\n\t"
                            "v_fmac_f32 %0, %1, %5\n\t"
                            "v_fmac_f32 %0, %2, %6"
                            : "+v"(auxp[c]), "=&v"(y0), "=&v"(y1)
                            : "v"(hh), "v"(ll), "v"(aw[2 * e2]), "v"(aw[2 * e2 + 1]));
                    }
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
            for (int qq = 0; qq < 4; ++qq) {
                if (MID_DBG(16)) asm volatile("" ::"v"(d[qq]));
                else *(f32x4*)(p.y + line0 + (size_t)(qq * 8) * yrowb + st_off) = d[qq];
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    if (AUX) {   // combine the two lane halves (weight rows 4 eh .. + 3 of every group of 8), one partial per person and 64-column slice
        const int slice = nbase >> 6;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const float sum = auxp[c] + __shfl_xor(auxp[c], 32, 64);
            if (eh == 0) p.head_part[(size_t)slice * p.M_pad + (mbase + c * 32 + eml)] = sum;
        }
    }
}

}  // namespace mlk
