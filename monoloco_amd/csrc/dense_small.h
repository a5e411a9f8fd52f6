// dense_small_kernel — the same Linear(+folded BN)+ReLU(+residual) layer as dense_kernel_pp, for SMALL row
// counts (a single image: a handful to a few hundred persons; reference use: Loco.forward per frame,
// monoloco/network/net.py:83-133).  The 256x256-tile persistent kernel keeps only (rows/256)*(N/256)
// CUs busy there and costs ~75 us per layer whatever the row count; here the layer is cut into
// 16 (n) x 16 (m) output tiles so that even 16 persons spread over N/16 = 64 workgroups, every wave
// streams its operands straight from L2 into MFMA registers (no LDS staging: nothing is reused inside a
// 16x16 tile) and the K range is split over the 4 waves of the workgroup.
//
//   grid  = (N/16, ceil(rows/16)),  block = 256 threads = 4 waves, wave w takes the k32 lines w, w+4, ...
//   MFMA  = v_mfma_f32_16x16x32_f16, A = 16 weight rows, B = 16 persons, hi*hi + hi*lo + lo*hi
//   operand fetch: lane (r = lane&15, q = lane>>4) loads 16 B = k 8q..8q+7 of row r: chunk q of the line is the
//                  hi half, chunk 4+q the lo half -- exactly the A/B register layout of the 16x16x32 MFMA
//   reduce: the 4 partial accumulators meet in 4 KiB of LDS; wave 0 finishes: * 2^-e, ReLU, + residual,
//           split to fp16 hi+lo, 8-byte stores into the line format (D: person = lane&15, n = 4q + reg)
// Arithmetic is the same as the big kernel (same operands, fp32 accumulation, same epilogue); only the order
// of the fp32 accumulation differs.
#pragma once
#include "dense_kernel.h"

namespace mlk {

constexpr int SMALL_THREADS = 256;

template <int NSPLIT, bool RELU, bool RES>
__global__ __launch_bounds__(SMALL_THREADS) void dense_small_kernel(DenseParams p) {
    __shared__ __attribute__((aligned(16))) float red[3][4][64];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 16;
    const int m0 = blockIdx.y * 16;
    const int r = lane & 15, q = lane >> 4;
    const size_t rowb = (size_t)p.K * 4;
    const int nl = p.K / 32;

    const char* wp = p.w + (size_t)(n0 + r) * rowb + q * 16;
    const char* xp = p.x + (size_t)(m0 + r) * rowb + q * 16;

    f32x4 acc;
    if (w == 0) {
        // bias * 2^e of weight rows n0 + 4q .. +3 (the D rows of this lane)
        acc = *(const f32x4*)(p.bias_scaled + n0 + 4 * q);
    } else {
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // Lines of this wave: l = w + 4*i.  They are fetched in groups of 4 (16 independent 16-byte loads in
    // flight) into two register sets that alternate; every array index below is a compile-time constant
    // (a run-time buffer index would turn the fragment arrays into select chains).
    struct Frag {
        half8 whi, wlo, xhi, xlo;
    };
    Frag fa[4], fb[4];
    auto fetch4 = [&](Frag(&f)[4], int g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = w + 4 * (4 * g + j);
            if (l < nl) {
                const size_t o = (size_t)l * LINE;
                f[j].whi = *(const half8*)(wp + o);
                f[j].xhi = *(const half8*)(xp + o);
                if (NSPLIT == 3) {
                    f[j].wlo = *(const half8*)(wp + o + 64);
                    f[j].xlo = *(const half8*)(xp + o + 64);
                }
            }
        }
    };
    auto compute4 = [&](Frag(&f)[4], int g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = w + 4 * (4 * g + j);
            if (l < nl) {
                if (NSPLIT == 3) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[j].whi, f[j].xlo, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[j].wlo, f[j].xhi, acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[j].whi, f[j].xhi, acc, 0, 0, 0);
            }
        }
    };
    // the residual of the finishing wave is requested first: its latency hides behind the operand stream instead of
    // standing alone behind the reduction (res may alias y: same thread, read long before the write)
    const size_t yoff = (size_t)(m0 + r) * ((size_t)p.N * 4) + (size_t)(n0 >> 5) * LINE + (size_t)((n0 & 16) + 4 * q) * 2;
    half4 rh, rl;
    if (RES && w == 0) {
        rh = *(const half4*)(p.res + yoff);
        rl = *(const half4*)(p.res + yoff + 64);
    }
    const int nw = nl > w ? (nl - w + 3) / 4 : 0;  // lines of this wave
    const int ng = (nw + 3) / 4;
    if (ng > 0) fetch4(fa, 0);
    for (int g = 0; g < ng; g += 2) {
        if (g + 1 < ng) fetch4(fb, g + 1);
        compute4(fa, g);
        if (g + 2 < ng) fetch4(fa, g + 2);
        if (g + 1 < ng) compute4(fb, g + 1);
    }

    if (w > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[w - 1][e][lane] = acc[e];
    }
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int ww = 0; ww < 3; ++ww)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += red[ww][e][lane];

    // lane holds person m0 + r, outputs n0 + 4q + e (e = 0..3): 4 consecutive fp16 of the hi half and of the
    // lo half of line n0/32 of the output row
    half4 oh, ol;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v = acc[e] * p.descale;
        if (RELU) v = __builtin_fmaxf(v, 0.0f);
        if (RES) v += (float)rh[e] + (float)rl[e];
        _Float16 a, b;
        split_f16(v, a, b);
        oh[e] = a;
        ol[e] = b;
    }
    *(half4*)(p.y + yoff) = oh;
    *(half4*)(p.y + yoff + 64) = ol;
}

// dense_small_multi_kernel (round 5) -- the 16 x 16 tiles of dense_small_kernel, but a workgroup KEEPS its 16 weight rows in
// registers and walks several row tiles with them.  Why: with one tile per workgroup a call of 80 .. 512 rows re-reads the 4 MB
// weight matrix once per row tile (5 .. 32 times) and every tile pays the full operand latency; the layer's time followed the
// tile count in steps (tools/sweep_small_rows.py: 52 us per forward up to 64 rows, 77 us at 80 .. 128, 145 at 256, 250 at 512).
// Here the grid is (N/16, gy) with (N/16) * gy <= the CU count; workgroup (bx, by) takes row tiles by, by + gy, ...: its waves load
// their share of the 16 weight rows ONCE (wave w: lines w, w + 4, ... -- at most 8 lines = 64 registers, K <= 1024) together with
// the activations of their first two row tiles (two register sets, re-requested two tiles ahead), then per row tile 24 MFMAs, the
// 4-wave reduction through LDS
// (double-buffered by tile parity: ONE barrier per tile) and wave 0's epilogue, which overlaps the other waves' next tile.
// Same operands, same per-tile arithmetic and summation order as dense_small_kernel: bit-identical results.
constexpr int SMALL_MULTI_MAX_LINES = 32;   // K <= 1024
template <int NSPLIT, bool RELU, bool RES>
__global__ __launch_bounds__(SMALL_THREADS) void dense_small_multi_kernel(DenseParams p, int n_row_tiles) {
    __shared__ __attribute__((aligned(16))) float red[2][3][4][64];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 16;
    const int r = lane & 15, q = lane >> 4;
    const size_t rowb = (size_t)p.K * 4;
    const int nl = p.K / 32;

    // this wave's weight fragments, for every row tile (lines past the end of K: never used)
    half8 whi[8], wlo[8];
    {
        const char* wp = p.w + (size_t)(n0 + r) * rowb + q * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int l = w + 4 * i;
            if (l < nl) {
                whi[i] = *(const half8*)(wp + (size_t)l * LINE);
                if (NSPLIT == 3) wlo[i] = *(const half8*)(wp + (size_t)l * LINE + 64);
            }
        }
    }
    const f32x4 bias4 = *(const f32x4*)(p.bias_scaled + n0 + 4 * q);   // bias * 2^e of weight rows n0 + 4q .. +3 (this lane's D rows)

    struct XF {
        half8 hi[8], lo[8];
    };
    auto fetch_x = [&](XF& f, int mt) {
        const char* xp = p.x + (size_t)(mt * 16 + r) * rowb + q * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int l = w + 4 * i;
            if (l < nl) {
                f.hi[i] = *(const half8*)(xp + (size_t)l * LINE);
                if (NSPLIT == 3) f.lo[i] = *(const half8*)(xp + (size_t)l * LINE + 64);
            }
        }
    };
    // one row tile: cur holds its activations; once its MFMAs are issued the set is re-requested for the tile after next
    auto tile = [&](XF& cur, int mt, int par) {
        const int m0 = mt * 16;
        const size_t yoff = (size_t)(m0 + r) * ((size_t)p.N * 4) + (size_t)(n0 >> 5) * LINE + (size_t)((n0 & 16) + 4 * q) * 2;
        half4 rh, rl;
        if (RES && w == 0) {
            rh = *(const half4*)(p.res + yoff);
            rl = *(const half4*)(p.res + yoff + 64);
        }
        f32x4 acc = (w == 0) ? bias4 : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (w + 4 * i < nl) {
                if (NSPLIT == 3) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[i], cur.lo[i], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[i], cur.hi[i], acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[i], cur.hi[i], acc, 0, 0, 0);
            }
        }
        const int mt_again = mt + 2 * (int)gridDim.y;
        if (mt_again < n_row_tiles) fetch_x(cur, mt_again);
        if (w > 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[par][w - 1][e][lane] = acc[e];
        }
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int ww = 0; ww < 3; ++ww)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += red[par][ww][e][lane];
            half4 oh, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[e] * p.descale;
                if (RELU) v = __builtin_fmaxf(v, 0.0f);
                if (RES) v += (float)rh[e] + (float)rl[e];
                _Float16 a, b;
                split_f16(v, a, b);
                oh[e] = a;
                ol[e] = b;
            }
            *(half4*)(p.y + yoff) = oh;
            *(half4*)(p.y + yoff + 64) = ol;
        }
    };
    // the activations of the workgroup's first TWO row tiles are requested together with the weights: one memory latency for all
    // three (up to 128 rows a workgroup has at most two tiles: nothing is requested later)
    XF xa, xb;
    int mt = blockIdx.y;
    if (mt >= n_row_tiles) return;
    fetch_x(xa, mt);
    if (mt + (int)gridDim.y < n_row_tiles) fetch_x(xb, mt + (int)gridDim.y);
    for (; mt < n_row_tiles; mt += 2 * (int)gridDim.y) {
        tile(xa, mt, 0);
        if (mt + (int)gridDim.y < n_row_tiles) tile(xb, mt + (int)gridDim.y, 1);
    }
}

// dense_small32_kernel -- the same idea with 32 (n) x 32 (m) output tiles and v_mfma_f32_32x32x16_f16, for a few
// hundred to ~2000 rows: a 16x16 tile re-reads 8 KiB of operands per output value from L2, a 32x32 tile half of
// that, and above ~256 rows there are enough 32x32 tiles ((N/32) * ceil(rows/32) >= 256) to fill the chip.
//   operand fetch per k32 line: lane (r = lane&31, q = lane>>5) loads, for both k16 half-steps s, 16 B = k 8q..8q+7
//   of the half-step: chunk 2s+q (hi) and 4+2s+q (lo) -- the A/B register layout of the 32x32x16 MFMA.
//   D: person = lane&31, n = 8g + 4q + e (g = reg>>2, e = reg&3): 4 x 8-byte stores per half (hi / lo) of the line
template <int NSPLIT, bool RELU, bool RES>
__global__ __launch_bounds__(SMALL_THREADS) void dense_small32_kernel(DenseParams p) {
    __shared__ __attribute__((aligned(16))) float red[3][16][64];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 32;
    const int m0 = blockIdx.y * 32;
    const int r = lane & 31, q = lane >> 5;
    const size_t rowb = (size_t)p.K * 4;
    const int nl = p.K / 32;

    const char* wp = p.w + (size_t)(n0 + r) * rowb + q * 16;
    const char* xp = p.x + (size_t)(m0 + r) * rowb + q * 16;

    f32x16 acc;
    if (w == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *(const f32x4*)(p.bias_scaled + n0 + 8 * g + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[g * 4 + e] = b4[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    }

    struct Frag {
        half8 whi[2], wlo[2], xhi[2], xlo[2];
    };
    Frag fa[2], fb[2];  // groups of 2 lines (16 independent loads in flight), two alternating register sets
    auto fetch2 = [&](Frag(&f)[2], int g) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int l = w + 4 * (2 * g + j);
            if (l < nl) {
                const size_t o = (size_t)l * LINE;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f[j].whi[s] = *(const half8*)(wp + o + s * 32);
                    f[j].xhi[s] = *(const half8*)(xp + o + s * 32);
                    if (NSPLIT == 3) {
                        f[j].wlo[s] = *(const half8*)(wp + o + 64 + s * 32);
                        f[j].xlo[s] = *(const half8*)(xp + o + 64 + s * 32);
                    }
                }
            }
        }
    };
    auto compute2 = [&](Frag(&f)[2], int g) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int l = w + 4 * (2 * g + j);
            if (l < nl) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (NSPLIT == 3) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[j].whi[s], f[j].xlo[s], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[j].wlo[s], f[j].xhi[s], acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[j].whi[s], f[j].xhi[s], acc, 0, 0, 0);
                }
            }
        }
    };
    const size_t ybase = (size_t)(m0 + r) * ((size_t)p.N * 4) + (size_t)(n0 >> 5) * LINE + (size_t)(4 * q) * 2;
    half4 rh[4], rl[4];   // residual of the finishing wave, requested ahead of the operand stream (see dense_small_kernel)
    if (RES && w == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            rh[g] = *(const half4*)(p.res + ybase + (size_t)(8 * g) * 2);
            rl[g] = *(const half4*)(p.res + ybase + (size_t)(8 * g) * 2 + 64);
        }
    }
    const int nw = nl > w ? (nl - w + 3) / 4 : 0;  // lines of this wave
    const int ng = (nw + 1) / 2;
    if (ng > 0) fetch2(fa, 0);
    for (int g = 0; g < ng; g += 2) {
        if (g + 1 < ng) fetch2(fb, g + 1);
        compute2(fa, g);
        if (g + 2 < ng) fetch2(fa, g + 2);
        if (g + 1 < ng) compute2(fb, g + 1);
    }

    if (w > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[w - 1][e][lane] = acc[e];
    }
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int ww = 0; ww < 3; ++ww)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += red[ww][e][lane];

#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const size_t yoff = ybase + (size_t)(8 * g) * 2;
        half4 oh, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[g * 4 + e] * p.descale;
            if (RELU) v = __builtin_fmaxf(v, 0.0f);
            if (RES) v += (float)rh[g][e] + (float)rl[g][e];
            _Float16 a, b;
            split_f16(v, a, b);
            oh[e] = a;
            ol[e] = b;
        }
        *(half4*)(p.y + yoff) = oh;
        *(half4*)(p.y + yoff + 64) = ol;
    }
}

}  // namespace mlk
