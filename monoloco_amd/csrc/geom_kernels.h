// geom_kernels.h -- the HBM-bound kernels either side of the dense layers:
//   prep_kernel      preprocess_monoloco / pixel_to_camera / get_keypoints('center')
//                    (reference monoloco/network/process.py:47-67, utils/camera.py:10-29,82-86)
//   pairs_kernel     preprocess_monstereo all-vs-all rows (process.py:25-44)
//   f32_to_lines / lines_to_f32   fp32 matrix <-> the "k32 hi|lo line" format of dense_kernel.h
//   heads_kernel     the GEMV-shaped heads w_aux (H->1) and w_fin (H->8|9) (architectures.py:60,67)
//   stereo_best      per-left arg-max of the aux logit (process.py:319-327)
//   post_kernel      extract_outputs + back-projection geometry (process.py:231-278,
//                    camera.py:161-177,202-237, net.py:195-215)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dense_kernel.h"

namespace mlk {

struct Kinv {
    float k[9];  // inverse(K), row-major
};

constexpr int NKP = 17;
constexpr int KPS_ROW = 3 * NKP;  // 51 floats per person: u[17], v[17], conf[17]
constexpr int NIN = 2 * NKP;      // 34 network inputs per person

// [u, v, 1] . Kinv^T row `r`, times z.  The reference does this as an fp32 GEMM with K = 3
// starting from a zero accumulator (camera.py:25-26): acc = u*k0; acc = fma(v,k1,acc);
// acc = fma(1,k2,acc).  Written out explicitly so the compiler cannot re-associate it.
__device__ __forceinline__ float cam_row(float u, float v, const float* kr, float z) {
    float acc = __fmul_rn(u, kr[0]);
    acc = __builtin_fmaf(v, kr[1], acc);
    acc = __fadd_rn(acc, kr[2]);
    return __fmul_rn(acc, z);
}

// ------------------------------------------------------------------------------------------
// prep: one workgroup of 256 threads = PB persons (256 for the big batches; 32 below ~8k persons, where 256 per workgroup
// leave most CUs without work: 15 -> 6 us at 4096 persons).  The (PB x 51) fp32 slab is contiguous in HBM, so it is
// loaded fully coalesced into LDS, each thread then normalises its own person out of LDS (row
// stride 51 dwords: odd, conflict free), and every output is written back coalesced:
//   x_f32   (m,34) fp32 reference-format inputs            (optional)
//   centre  (m,2)  box-centre pixel, (max-min)/2+min        (optional)
//   x_lines (m_pad, kpad) line-format network input, zero padded in k and in rows >= m (optional)
template <int PB>
__global__ __launch_bounds__(256) void prep_kernel(const float* __restrict__ kps, int64_t m, Kinv ki,
                                                   float z_met, float* __restrict__ x_f32,
                                                   float* __restrict__ centre, char* __restrict__ x_lines,
                                                   int kpad, int64_t m_pad, int zero_center) {
    __shared__ float s_in[PB * KPS_ROW];
    __shared__ float s_x[PB * NIN];
    const int t = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * PB;
    const int64_t nvalid = (m - p0) < PB ? (m - p0 > 0 ? m - p0 : 0) : PB;
    const float* src = kps + p0 * KPS_ROW;
    const int nfl = (int)nvalid * KPS_ROW;
    for (int i = t; i < nfl; i += 256) s_in[i] = src[i];
    __syncthreads();
    // (a) per person: box centre (and, for zero_center, its normalised image) -- only when somebody needs it
    __shared__ float s_c[PB][2];
    const bool need_centre = centre != nullptr || zero_center;
    if (need_centre && t < nvalid) {
        const float* u = s_in + t * KPS_ROW;
        const float* v = u + NKP;
        float umin = u[0], umax = u[0], vmin = v[0], vmax = v[0];
#pragma unroll
        for (int j = 1; j < NKP; ++j) {
            umin = __builtin_fminf(umin, u[j]);
            umax = __builtin_fmaxf(umax, u[j]);
            vmin = __builtin_fminf(vmin, v[j]);
            vmax = __builtin_fmaxf(vmax, v[j]);
        }
        const float uc = __fadd_rn(__fmul_rn(__fsub_rn(umax, umin), 0.5f), umin);  // camera.py:85
        const float vc = __fadd_rn(__fmul_rn(__fsub_rn(vmax, vmin), 0.5f), vmin);
        // zero_center (legacy MonoLoco, process.py:61-62): subtract the normalised box centre
        s_c[t][0] = zero_center ? cam_row(uc, vc, ki.k + 0, z_met) : 0.0f;
        s_c[t][1] = zero_center ? cam_row(uc, vc, ki.k + 3, z_met) : 0.0f;
        if (centre) {
            centre[(p0 + t) * 2 + 0] = uc;
            centre[(p0 + t) * 2 + 1] = vc;
        }
    }
    if (zero_center) __syncthreads();
    // (b) one (person, joint) per thread and pass: a single image keeps 17 threads busy per person instead of one
    for (int id = t; id < (int)nvalid * NKP; id += 256) {
        const int pi = id / NKP, j = id - pi * NKP;
        const float* u = s_in + pi * KPS_ROW;
        const float xj = cam_row(u[j], u[NKP + j], ki.k + 0, z_met);
        const float yj = cam_row(u[j], u[NKP + j], ki.k + 3, z_met);
        s_x[pi * NIN + 2 * j] = zero_center ? __fsub_rn(xj, s_c[pi][0]) : xj;
        s_x[pi * NIN + 2 * j + 1] = zero_center ? __fsub_rn(yj, s_c[pi][1]) : yj;
    }
    __syncthreads();
    if (x_f32) {
        float* dst = x_f32 + p0 * NIN;
        const int n = (int)nvalid * NIN;
        for (int i = t; i < n; i += 256) dst[i] = s_x[i];
    }
    if (x_lines) {
        // 16-byte chunks: person pi, chunk c of its row; chunk (b, sub): sub<4 hi, sub>=4 lo of
        // k = 32b + 8(sub&3) .. +7
        const int cpr = kpad / 4;  // 16-B chunks per row = kpad*4/16
        const int total = PB * cpr;
        const int64_t rows_here = (m_pad - p0) < PB ? (m_pad - p0) : PB;
        char* dst = x_lines + p0 * (int64_t)kpad * 4;
        for (int id = t; id < total; id += 256) {
            const int pi = id / cpr, c = id - pi * cpr;
            if (pi >= rows_here) break;
            const int b = c >> 3, sub = c & 7;
            const int k0 = b * 32 + (sub & 3) * 8;
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                const float val = (k < NIN && pi < (int)nvalid) ? s_x[pi * NIN + k] : 0.0f;  // zero pad rows and k
                _Float16 hi, lo;
                split_f16(val, hi, lo);
                o[e] = (sub < 4) ? hi : lo;
            }
            *(half8*)(dst + (int64_t)id * 16) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// generic fp32 (m,k) -> line format (m_pad, kpad), zero padded.  One thread per 16-B chunk.
__global__ __launch_bounds__(256) void f32_to_lines_kernel(const float* __restrict__ x, int64_t m, int k,
                                                          char* __restrict__ lines, int kpad, int64_t m_pad) {
    const int cpr = kpad / 4;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= m_pad * cpr) return;
    const int64_t row = id / cpr;
    const int c = (int)(id - row * cpr);
    const int b = c >> 3, sub = c & 7;
    const int k0 = b * 32 + (sub & 3) * 8;
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kk = k0 + e;
        const float val = (row < m && kk < k) ? x[row * k + kk] : 0.0f;
        _Float16 hi, lo;
        split_f16(val, hi, lo);
        o[e] = (sub < 4) ? hi : lo;
    }
    *(half8*)(lines + id * 16) = o;
}

// line format (m_pad, n) -> fp32 (m, n): value = hi + lo.  One thread per 8 values.
__device__ __forceinline__ float bf16_bits_to_f32(_Float16 hslot) {
    return __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, hslot) << 16);
}

__global__ __launch_bounds__(256) void lines_to_f32_kernel(const char* __restrict__ lines, int64_t m, int n,
                                                          float* __restrict__ y, int bf16) {
    const int gpr = n / 8;  // groups of 8 values per row
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= m * gpr) return;
    const int64_t row = id / gpr;
    const int g = (int)(id - row * gpr);
    const int b = g >> 2, sub = g & 3;
    const char* p = lines + row * (int64_t)n * 4 + b * LINE + sub * 16;
    const half8 hi = *(const half8*)p;
    const half8 lo = *(const half8*)(p + 64);
#pragma unroll
    for (int e = 0; e < 8; ++e)
        y[row * n + b * 32 + sub * 8 + e] = bf16 ? bf16_bits_to_f32(hi[e]) : (float)hi[e] + (float)lo[e];
}

// bf16 comparison mode: fp16 hi|lo lines -> one bf16 (round-to-nearest-even of hi + lo) in the hi slot, lo slot
// zeroed, in place.  One thread per (line, 16-byte hi chunk): `pairs` = lines * 4.
__global__ __launch_bounds__(256) void lines_to_bf16_kernel(char* __restrict__ lines, int64_t pairs) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= pairs) return;
    char* p = lines + (id >> 2) * LINE + (id & 3) * 16;
    const half8 hi = *(const half8*)p;
    const half8 lo = *(const half8*)(p + 64);
    half8 o, z;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = (float)hi[e] + (float)lo[e];
        unsigned x = __builtin_bit_cast(unsigned, v);
        x += 0x7fffu + ((x >> 16) & 1u);  // round to nearest even (inputs are finite)
        o[e] = __builtin_bit_cast(_Float16, (unsigned short)(x >> 16));
        z[e] = (_Float16)0.0f;
    }
    *(half8*)p = o;
    *(half8*)(p + 64) = z;
}

// ------------------------------------------------------------------------------------------
// stereo pair rows, written straight into the line format: row (i*mr + j) = [L_i, L_i - R_j]
// (68 values, kpad = 96), optionally also as fp32 (ml*mr, 68).  One thread per 16-B chunk.
__global__ __launch_bounds__(256) void pairs_kernel(const float* __restrict__ xl, int64_t ml,
                                                    const float* __restrict__ xr, int64_t mr,
                                                    float* __restrict__ rows_f32, char* __restrict__ lines,
                                                    int kpad, int64_t rows_pad) {
    const int64_t rows = ml * mr;
    if (lines) {
        const int cpr = kpad / 4;
        const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (id < rows_pad * cpr) {
            const int64_t row = id / cpr;
            const int c = (int)(id - row * cpr);
            const int b = c >> 3, sub = c & 7;
            const int k0 = b * 32 + (sub & 3) * 8;
            const int64_t i = row / mr, j = row - i * mr;
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                float val = 0.0f;
                if (row < rows && k < 2 * NIN) {
                    if (k < NIN) val = xl[i * NIN + k];
                    else val = __fsub_rn(xl[i * NIN + k - NIN], xr[j * NIN + k - NIN]);
                }
                _Float16 hi, lo;
                split_f16(val, hi, lo);
                o[e] = (sub < 4) ? hi : lo;
            }
            *(half8*)(lines + id * 16) = o;
        }
    }
    if (rows_f32) {
        const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (id < rows * 2 * NIN) {
            const int64_t row = id / (2 * NIN);
            const int k = (int)(id - row * 2 * NIN);
            const int64_t i = row / mr, j = row - i * mr;
            rows_f32[id] = (k < NIN) ? xl[i * NIN + k] : __fsub_rn(xl[i * NIN + k - NIN], xr[j * NIN + k - NIN]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// A stereo FRAME's front in one launch (ml_loco_frame_stereo, ml + mr <= STEREO_FRONT_MAX persons): both pre-processes and the
// all-vs-all pairing -- prep_kernel (left, with the box centres), prep_kernel (right) and pairs_kernel in sequence are three
// dependent launches of ~5 us each for a few KiB of work.  Every workgroup loads all keypoints of the frame into LDS, normalises
// them (the same cam_row per joint as prep_kernel: same bits) and writes ITS share of the line-format pair rows [L_i, L_i - R_j];
// workgroup 0 also writes the left persons' box centres.  No workgroup depends on another.
constexpr int STEREO_FRONT_MAX = 96;
__global__ __launch_bounds__(256) void stereo_front_kernel(const float* __restrict__ kps_l, int ml, const float* __restrict__ kps_r,
                                                           int mr, Kinv ki, float z_met, float* __restrict__ centre,
                                                           char* __restrict__ lines, int kpad, int64_t rows_pad) {
    __shared__ float s_in[STEREO_FRONT_MAX * KPS_ROW];
    __shared__ float s_x[STEREO_FRONT_MAX * NIN];
    const int t = threadIdx.x;
    const int np = ml + mr;
    for (int i = t; i < ml * KPS_ROW; i += 256) s_in[i] = kps_l[i];
    for (int i = t; i < mr * KPS_ROW; i += 256) s_in[ml * KPS_ROW + i] = kps_r[i];
    __syncthreads();
    if (blockIdx.x == 0 && centre && t < ml) {   // get_keypoints(.., 'center') of the left persons (camera.py:82-86), as prep_kernel
        const float* u = s_in + t * KPS_ROW;
        const float* v = u + NKP;
        float umin = u[0], umax = u[0], vmin = v[0], vmax = v[0];
#pragma unroll
        for (int j = 1; j < NKP; ++j) {
            umin = __builtin_fminf(umin, u[j]);
            umax = __builtin_fmaxf(umax, u[j]);
            vmin = __builtin_fminf(vmin, v[j]);
            vmax = __builtin_fmaxf(vmax, v[j]);
        }
        centre[t * 2 + 0] = __fadd_rn(__fmul_rn(__fsub_rn(umax, umin), 0.5f), umin);
        centre[t * 2 + 1] = __fadd_rn(__fmul_rn(__fsub_rn(vmax, vmin), 0.5f), vmin);
    }
    for (int id = t; id < np * NKP; id += 256) {
        const int pi = id / NKP, j = id - pi * NKP;
        const float* u = s_in + pi * KPS_ROW;
        s_x[pi * NIN + 2 * j] = cam_row(u[j], u[NKP + j], ki.k + 0, z_met);
        s_x[pi * NIN + 2 * j + 1] = cam_row(u[j], u[NKP + j], ki.k + 3, z_met);
    }
    __syncthreads();
    const int64_t rows = (int64_t)ml * mr;
    const int cpr = kpad / 4;
    const float* xl = s_x;
    const float* xr = s_x + ml * NIN;
    for (int64_t id = (int64_t)blockIdx.x * 256 + t; id < rows_pad * cpr; id += (int64_t)gridDim.x * 256) {   // pairs_kernel's chunk
        const int64_t row = id / cpr;
        const int c = (int)(id - row * cpr);
        const int b = c >> 3, sub = c & 7;
        const int k0 = b * 32 + (sub & 3) * 8;
        const int64_t i = row / mr, j = row - i * mr;
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            float val = 0.0f;
            if (row < rows && k < 2 * NIN) {
                if (k < NIN) val = xl[i * NIN + k];
                else val = __fsub_rn(xl[i * NIN + k - NIN], xr[j * NIN + k - NIN]);
            }
            _Float16 hi, lo;
            split_f16(val, hi, lo);
            o[e] = (sub < 4) ? hi : lo;
        }
        *(half8*)(lines + id * 16) = o;
    }
}

// ------------------------------------------------------------------------------------------
// heads: raw[row][col0 + o] = sum_n act[row][n] * wh[o][n] + bh[o], o < NH, exact fp32 FMAs on
// hi+lo reconstructed activations.  One wave per row (4 rows per pass to amortise the weight
// reads from LDS), lanes stride over (hi chunk, lo chunk) pairs = 8 consecutive n, butterfly
// reduction over the 64 lanes.  HBM-bound: reads H*4 bytes per row once.
template <int NH>
__global__ __launch_bounds__(256) void heads_kernel(const char* __restrict__ act, int H,
                                                    const float* __restrict__ wh, const float* __restrict__ bh,
                                                    float* __restrict__ raw, int raw_stride, int col0,
                                                    int64_t m, int bf16) {
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    float* s_w = (float*)dyn_smem;  // [NH][H]
    for (int i = threadIdx.x; i < NH * H; i += 256) s_w[i] = wh[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int npairs = H / 8;
    const int64_t nquads = (m + 3) / 4;
    for (int64_t qd = (int64_t)blockIdx.x * 4 + wave; qd < nquads; qd += (int64_t)gridDim.x * 4) {
        const int64_t r0 = qd * 4;
        float acc[4][NH];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int o = 0; o < NH; ++o) acc[r][o] = 0.0f;
        for (int pr = lane; pr < npairs; pr += 64) {
            const int b = pr >> 2, sub = pr & 3;
            const int n = b * 32 + sub * 8;
            float xv[4][8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = (r0 + r < m) ? (r0 + r) : (m - 1);
                const char* p = act + row * (int64_t)H * 4 + b * LINE + sub * 16;
                const half8 hi = *(const half8*)p;
                const half8 lo = *(const half8*)(p + 64);
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[r][e] = bf16 ? bf16_bits_to_f32(hi[e]) : (float)hi[e] + (float)lo[e];
            }
#pragma unroll
            for (int o = 0; o < NH; ++o) {
                const f32x4 wa = *(const f32x4*)(s_w + o * H + n);
                const f32x4 wb = *(const f32x4*)(s_w + o * H + n + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = acc[r][o];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = __builtin_fmaf(xv[r][e], wa[e], a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = __builtin_fmaf(xv[r][4 + e], wb[e], a);
                    acc[r][o] = a;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int o = 0; o < NH; ++o) {
                float a = acc[r][o];
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) a += __shfl_xor(a, s, 64);
                acc[r][o] = a;
            }
        if (lane < 4 * NH) {
            const int r = lane / NH, o = lane - r * NH;
            if (r0 + r < m) {
                float val = 0.0f;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int oo = 0; oo < NH; ++oo)
                        if (rr == r && oo == o) val = acc[rr][oo];
                raw[(r0 + r) * raw_stride + col0 + o] = val + bh[o];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// second half of the fused w_fin head (dense_kernel_pp<.., HEAD>): raw[row][col0 + o] = bias[o] +
// sum over the nparts 128-column slices of part[slice][row][o].  One thread per (row, o).
__global__ __launch_bounds__(256) void head_reduce_kernel(const float* __restrict__ part, int nparts, int64_t m_pad,
                                                         int64_t m, int nh, const float* __restrict__ bh,
                                                         float* __restrict__ raw, int raw_stride, int col0) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= m * 16) return;
    const int64_t row = id >> 4;
    const int o = (int)(id & 15);
    if (o >= nh) return;
    float s = bh[o];
    for (int s0 = 0; s0 < nparts; s0 += 8) {   // in slice order, requested eight at a time (see tail_mono_kernel)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[((int64_t)(s0 + u < nparts ? s0 + u : nparts - 1) * m_pad + row) * 16 + o];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < nparts) s += v[u];
    }
    raw[row * raw_stride + col0 + o] = s;
}

// second half of the fused w_aux head (dense_kernel_w4<.., HEAD = -1>): raw[row][col0] = bias + sum over the nparts
// 128-column slices of part[slice][row].  One thread per row.
__global__ __launch_bounds__(256) void aux_reduce_kernel(const float* __restrict__ part, int nparts, int64_t m_pad, int64_t m,
                                                        const float* __restrict__ bh, float* __restrict__ raw, int raw_stride,
                                                        int col0) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= m) return;
    float s = bh[0];
    for (int s0 = 0; s0 < nparts; s0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(s0 + u < nparts ? s0 + u : nparts - 1) * m_pad + row];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < nparts) s += v[u];
    }
    raw[row * raw_stride + col0] = s;
}

// ------------------------------------------------------------------------------------------
// stereo: per left person the first right index whose aux logit (last column) is maximal, and a
// global count of left persons with tied maxima (the reference keeps every tied row,
// process.py:325-326; the host re-does those rare cases).
__global__ __launch_bounds__(256) void stereo_best_kernel(const float* __restrict__ raw_all, int out_f,
                                                          int64_t ml, int64_t mr, int32_t* __restrict__ best,
                                                          int32_t* __restrict__ row_index,
                                                          int32_t* __restrict__ ties) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ml) return;
    const float* p = raw_all + i * mr * out_f + (out_f - 1);
    float bv = p[0];
    int bj = 0, cnt = 1;
    bool has_nan = (bv != bv);
    for (int64_t j = 1; j < mr; ++j) {
        const float v = p[j * out_f];
        has_nan |= (v != v);
        if (v > bv) {
            bv = v;
            bj = (int)j;
            cnt = 1;
        } else if (v == bv) {
            ++cnt;
        }
    }
    best[i] = bj;
    row_index[i] = (int32_t)(i * mr + bj);
    if (cnt > 1 || has_nan) atomicAdd(ties, 1);
}

// filter_outputs' mask as a row list (process.py:319-327): every pair row whose aux logit (last column) is >= the maximum of its
// left person, left persons in order, rows in order.  ONE workgroup walks the left persons 256 at a time: per person the maximum and
// the number of kept rows, an exclusive scan of the counts through LDS, then every thread writes its person's rows behind the
// running base.  A NaN among a person's candidates keeps none of its rows (`val >= nan` is false everywhere).  Exact ties are
// rare: this runs only when stereo_best_kernel has counted one.
__global__ __launch_bounds__(256) void stereo_tied_rows_kernel(const float* __restrict__ raw_all, int out_f, int64_t ml, int64_t mr,
                                                               int32_t* __restrict__ rows, int32_t* __restrict__ count) {
    __shared__ int32_t scan[256];
    __shared__ int32_t base_s;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < ml; i0 += 256) {
        const int64_t i = i0 + threadIdx.x;
        float bv = 0.f;
        int cnt = 0;
        if (i < ml) {
            const float* p = raw_all + i * mr * out_f + (out_f - 1);
            bv = p[0];
            bool has_nan = (bv != bv);
            cnt = 1;
            for (int64_t j = 1; j < mr; ++j) {
                const float v = p[j * out_f];
                has_nan |= (v != v);
                if (v > bv) {
                    bv = v;
                    cnt = 1;
                } else if (v == bv) {
                    ++cnt;
                }
            }
            if (has_nan) cnt = 0;
        }
        scan[threadIdx.x] = cnt;
        __syncthreads();
        for (int s = 1; s < 256; s <<= 1) {   // inclusive Hillis-Steele scan
            const int32_t add = ((int)threadIdx.x >= s) ? scan[threadIdx.x - s] : 0;
            __syncthreads();
            scan[threadIdx.x] += add;
            __syncthreads();
        }
        const int32_t base = base_s;
        if (cnt > 0) {
            int32_t w = base + scan[threadIdx.x] - cnt;
            const float* p = raw_all + i * mr * out_f + (out_f - 1);
            for (int64_t j = 0; j < mr; ++j)
                if (p[j * out_f] >= bv) rows[w++] = (int32_t)(i * mr + j);
        }
        __syncthreads();
        if (threadIdx.x == 255) base_s = base + scan[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base_s;
}

// ------------------------------------------------------------------------------------------
// heads for a single image's worth of rows: ONE launch for all heads (w_fin and w_aux read different activation
// buffers), one workgroup per row, the outputs dealt round-robin to its 4 waves, weights read straight from L2
// (no LDS staging: with a handful of rows nothing amortises it).  Same per-(row, output) arithmetic and reduction
// order as heads_kernel.
struct SmallHeads {
    const char* act[2];   // line-format activations each head reads
    const float* w[2];    // [nh][H]
    const float* b[2];    // [nh]
    int nh[2];            // outputs of each head (nh[1] may be 0)
    int col0[2];          // first raw column of each head
};

// ------------------------------------------------------------------------------------------
// post: one thread per person.  Column meaning of raw: theta, psi, d, s=log(b/d), h, w, l,
// sin, cos [, aux logit].  All arithmetic fp32 in the reference's association order.
// One person of the post-process; r = its out_f raw network outputs (a global row, or registers of the fused tail).
template <typename R>
__device__ __forceinline__ void post_person(const R& r, int out_f, int64_t i, const float* __restrict__ centre, const Kinv& ki,
                                            const float* __restrict__ box_conf, float* __restrict__ out,
                                            float* __restrict__ xyzds) {
    const float theta = r[0], psi = r[1], d = r[2], s = r[3];
    const float bi = __fmul_rn(expf(s), d);                       // process.py:131
    const float x = __fmul_rn(__fmul_rn(d, sinf(psi)), cosf(theta));  // camera.py:232
    const float y = __fmul_rn(d, cosf(psi));                      // camera.py:236
    const float z = sqrtf(__fsub_rn(__fsub_rn(__fmul_rn(d, d), __fmul_rn(x, x)), __fmul_rn(y, y)));  // process.py:265
    const float yaw = atan2f(r[7], r[8]);                         // process.py:272
    float ego = __fadd_rn(yaw, atan2f(x, z));                     // camera.py:203-204
    const float PI_F = 3.14159274101257324f, TWO_PI_F = 6.28318548202514648f;
    if (ego > PI_F) ego = __fsub_rn(ego, TWO_PI_F);
    if (ego < -PI_F) ego = __fadd_rn(ego, TWO_PI_F);
    float aux = r[out_f - 1];
    if (out_f == 10) aux = 1.0f / (1.0f + expf(-aux));            // process.py:277
    float* o = out + i * 16;
    float uc = 0.f, vc = 0.f, px = 0.f, py = 0.f, pz = 0.f, conf = 0.f;
    if (centre) {
        uc = centre[i * 2];
        vc = centre[i * 2 + 1];
        const float xc = cam_row(uc, vc, ki.k + 0, 1.0f);         // net.py:198
        const float yc = cam_row(uc, vc, ki.k + 3, 1.0f);
        const float zc = cam_row(uc, vc, ki.k + 6, 1.0f);
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(1.0f, __fmul_rn(xc, xc)), __fmul_rn(yc, yc)));  // camera.py:177
        px = __fmul_rn(xc, d) / nrm;
        py = __fmul_rn(yc, d) / nrm;
        pz = __fmul_rn(zc, d) / nrm;
        if (box_conf) {
            const float dist = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py)), __fmul_rn(pz, pz)));
            conf = __fmul_rn(0.035f, box_conf[i]) / (bi / dist);  // net.py:214-215
        }
    }
    o[0] = x; o[1] = y; o[2] = z; o[3] = d;
    o[4] = bi; o[5] = yaw; o[6] = ego; o[7] = aux;
    o[8] = r[4]; o[9] = r[5]; o[10] = r[6]; o[11] = conf;
    o[12] = r[7]; o[13] = r[8]; o[14] = uc; o[15] = vc;
    if (xyzds) {
        float* q = xyzds + i * 5;
        q[0] = px; q[1] = py; q[2] = pz; q[3] = d; q[4] = bi;
    }
}

__global__ __launch_bounds__(256) void post_kernel(const float* __restrict__ raw, int out_f,
                                                   const int32_t* __restrict__ row_index, int64_t m,
                                                   const float* __restrict__ centre, Kinv ki,
                                                   const float* __restrict__ box_conf, float* __restrict__ out,
                                                   float* __restrict__ xyzds) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int64_t src = row_index ? (int64_t)row_index[i] : i;
    const float* r = raw + src * out_f;
    post_person(r, out_f, i, centre, ki, box_conf, out, xyzds);
}

// The per-person geometry block of Loco.post_process (net.py:195-215), out[i] = { uv_shoulder(2), uv_head(2), uv_center(2),
// xy_center(3) = pixel_to_camera(uv_center, K, 1), xyz_pred(3) = xyz_from_distance(dd, xy_center) } (12 floats): same arithmetic,
// in the same order, as keypoints_kernel / pix2cam_kernel / xyz_from_distance_kernel (geom_ops.h).
constexpr int POSTGEO_STRIDE = 12;
__device__ __forceinline__ void post_geometry_person(const float* __restrict__ kps, int64_t i, const Kinv& ki, float dd,
                                                     float* __restrict__ out) {
    const float* u = kps + i * KPS_ROW;
    const float* v = u + NKP;
    float* o = out + i * POSTGEO_STRIDE;
    float su = 0.f, sv = 0.f;
    for (int j = 5; j < 7; ++j) {
        su = __fadd_rn(su, u[j]);
        sv = __fadd_rn(sv, v[j]);
    }
    o[0] = su / 2.0f;
    o[1] = sv / 2.0f;
    su = 0.f;
    sv = 0.f;
    for (int j = 0; j < 5; ++j) {
        su = __fadd_rn(su, u[j]);
        sv = __fadd_rn(sv, v[j]);
    }
    o[2] = su / 5.0f;
    o[3] = sv / 5.0f;
    float umin = u[0], umax = u[0], vmin = v[0], vmax = v[0];
    for (int j = 1; j < NKP; ++j) {
        umin = __builtin_fminf(umin, u[j]);
        umax = __builtin_fmaxf(umax, u[j]);
        vmin = __builtin_fminf(vmin, v[j]);
        vmax = __builtin_fmaxf(vmax, v[j]);
    }
    const float uc = __fadd_rn(__fmul_rn(__fsub_rn(umax, umin), 0.5f), umin);
    const float vc = __fadd_rn(__fmul_rn(__fsub_rn(vmax, vmin), 0.5f), vmin);
    o[4] = uc;
    o[5] = vc;
    const float cx = cam_row(uc, vc, ki.k + 0, 1.0f), cy = cam_row(uc, vc, ki.k + 3, 1.0f), cz = cam_row(uc, vc, ki.k + 6, 1.0f);
    o[6] = cx;
    o[7] = cy;
    o[8] = cz;
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(1.0f, __fmul_rn(cx, cx)), __fmul_rn(cy, cy)));
    o[9] = __fmul_rn(cx, dd) / nrm;
    o[10] = __fmul_rn(cy, dd) / nrm;
    o[11] = __fmul_rn(cz, dd) / nrm;
}

// "The frame is complete", told to a host that busy-polls a word of pinned memory instead of sleeping in hipStreamSynchronize
// (ml_loco_frame_mono; one launch + its completion signal cost a host ~20 us on this stack, tools/exp_sync.py): every workgroup
// makes its stores visible system-wide and checks in at `arrive`; the last one to arrive resets the counter and releases `seq`
// into `flag`.  flag == nullptr: nothing of this happens.
struct FrameDone {
    int* arrive = nullptr;   // device counter, 0 between frames
    int* flag = nullptr;     // device address of the pinned host word
    int seq = 0;
};

// post_out != null: the row is post-processed here as well (post_person by thread 0: a single image's forward ends in this
// launch); raw may then be null.  geo_out != null (with post_out): + the post_process geometry of the row (thread 64).
__global__ __launch_bounds__(256) void heads_small_kernel(SmallHeads hp, int H, float* __restrict__ raw, int raw_stride,
                                                          int64_t m, const float* __restrict__ centre, Kinv ki,
                                                          const float* __restrict__ box_conf, float* __restrict__ post_out,
                                                          float* __restrict__ xyzds, const float* __restrict__ geo_kps,
                                                          float* __restrict__ geo_out, FrameDone done = FrameDone()) {
    __shared__ float srow[16];
    const int64_t row = blockIdx.x;
    if (row >= m) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total = hp.nh[0] + hp.nh[1];
    const int npairs = H / 8;
    for (int o = wave; o < total; o += 4) {
        const int hsel = o < hp.nh[0] ? 0 : 1;
        const int oo = o - (hsel ? hp.nh[0] : 0);
        const char* arow = hp.act[hsel] + row * (int64_t)H * 4;
        const float* wrow = hp.w[hsel] + (size_t)oo * H;
        float a = 0.0f;
        for (int pr = lane; pr < npairs; pr += 64) {
            const int bb = pr >> 2, sub = pr & 3;
            const char* q = arow + bb * LINE + sub * 16;
            const half8 hi = *(const half8*)q;
            const half8 lo = *(const half8*)(q + 64);
            const f32x4 wa = *(const f32x4*)(wrow + bb * 32 + sub * 8);
            const f32x4 wb = *(const f32x4*)(wrow + bb * 32 + sub * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) a = __builtin_fmaf((float)hi[e] + (float)lo[e], wa[e], a);
#pragma unroll
            for (int e = 0; e < 4; ++e) a = __builtin_fmaf((float)hi[4 + e] + (float)lo[4 + e], wb[e], a);
        }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) a += __shfl_xor(a, sft, 64);
        if (lane == 0) {
            const float v = a + hp.b[hsel][oo];
            if (raw) raw[row * raw_stride + hp.col0[hsel] + oo] = v;
            srow[hp.col0[hsel] + oo] = v;
        }
    }
    if (post_out) {
        __syncthreads();
        if (threadIdx.x == 0) post_person((const float*)srow, raw_stride, row, centre, ki, box_conf, post_out, xyzds);
        // geo_out != null: the geometry block post_process needs (distance = raw column 2, what post_person calls d) as well:
        // a frame then ends in this launch, and both blocks may lie in pinned host memory (ml_loco_frame_mono)
        if (geo_out && threadIdx.x == 64) post_geometry_person(geo_kps, row, ki, srow[2], geo_out);
    }
    if (done.flag) {   // (uniform; the grid is exactly m workgroups: no early return above)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            if (atomicAdd(done.arrive, 1) == (int)gridDim.x - 1) {
                __hip_atomic_store(done.arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
                __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}


// Both output heads of a LocoModel (w_fin: NH outputs from act_fin; w_aux: one output from act_aux) in ONE launch, for the
// mid-size path, where two heads_kernel launches + post_kernel are ~10 % of a 4096-row forward: heads_kernel's loop and fma
// order per head (same bits as the separate launches), then, when post_out is given, the mono post-process of the row by the
// lane that holds it (post_person) -- raw may then be null.  Dynamic LDS: (NH + 1) * H floats.
// RW = rows per wave and pass (round 4: 2; with 4 the kernel held 256 VGPRs + 76 AGPRs = ONE wave per SIMD, one workgroup per CU,
// and the 512 workgroups of an 8192-row batch ran in two rounds: 46 us for 64 MB; with 2 rows three workgroups share a CU).
template <int NH, int RW = 2>
__global__ __launch_bounds__(256) void heads_pair_kernel(const char* __restrict__ act_fin, const char* __restrict__ act_aux, int H,
                                                         const float* __restrict__ w_fin, const float* __restrict__ b_fin,
                                                         const float* __restrict__ w_aux, const float* __restrict__ b_aux,
                                                         float* __restrict__ raw, int64_t m,
                                                         const float* __restrict__ centre, Kinv ki,
                                                         const float* __restrict__ box_conf, float* __restrict__ post_out,
                                                         float* __restrict__ xyzds) {
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    float* s_w = (float*)dyn_smem;  // [NH][H] then the aux head's [H]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int npairs = H / 8;
    const int64_t nquads = (m + RW - 1) / RW;   // ("quads": groups of RW rows)
    // one 8-column group of RW rows of both sources: 4 RW independent 16-byte loads per lane
    struct Rows {
        half8 hi[RW], lo[RW], hj[RW], lj[RW];
    };
    auto fetch = [&](Rows& x, int64_t r0, int pr) {
        const int b = pr >> 2, sub = pr & 3;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int64_t row = (r0 + r < m) ? (r0 + r) : (m - 1);
            const size_t off = (size_t)row * H * 4 + b * LINE + sub * 16;
            x.hi[r] = *(const half8*)(act_fin + off);
            x.lo[r] = *(const half8*)(act_fin + off + 64);
            x.hj[r] = *(const half8*)(act_aux + off);
            x.lj[r] = *(const half8*)(act_aux + off + 64);
        }
    };
    // (round 3 requested the wave's first rows BEFORE staging the weights, in a second register set: with the 2-row passes of round 4
    // the other resident workgroups cover that latency, and the 32 registers buy occupancy)
    const int64_t qd0 = (int64_t)blockIdx.x * 4 + wave;
    for (int i = threadIdx.x * 4; i < NH * H; i += 1024) *(f32x4*)(s_w + i) = *(const f32x4*)(w_fin + i);
    for (int i = threadIdx.x * 4; i < H; i += 1024) *(f32x4*)(s_w + NH * H + i) = *(const f32x4*)(w_aux + i);
    __syncthreads();
    for (int64_t qd = qd0; qd < nquads; qd += (int64_t)gridDim.x * 4) {
        const int64_t r0 = qd * RW;
        float acc[RW][NH + 1];
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int o = 0; o <= NH; ++o) acc[r][o] = 0.0f;
        for (int pr = lane; pr < npairs; pr += 64) {
            const int b = pr >> 2, sub = pr & 3;
            const int n = b * 32 + sub * 8;
            Rows x;
            fetch(x, r0, pr);
            float xv[RW][8], xa[RW][8];
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xv[r][e] = (float)x.hi[r][e] + (float)x.lo[r][e];
                    xa[r][e] = (float)x.hj[r][e] + (float)x.lj[r][e];
                }
#pragma unroll
            for (int o = 0; o <= NH; ++o) {
                const f32x4 wa = *(const f32x4*)(s_w + o * H + n);
                const f32x4 wb = *(const f32x4*)(s_w + o * H + n + 4);
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    float a = acc[r][o];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = __builtin_fmaf(o < NH ? xv[r][e] : xa[r][e], wa[e], a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = __builtin_fmaf(o < NH ? xv[r][4 + e] : xa[r][4 + e], wb[e], a);
                    acc[r][o] = a;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int o = 0; o <= NH; ++o) {
                float a = acc[r][o];
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) a += __shfl_xor(a, s, 64);
                acc[r][o] = a;
            }
        // every lane holds all sums; lane r finishes row r0 + r
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            if (lane == r && r0 + r < m) {
                float rowv[NH + 1];   // the raw row: w_fin's NH outputs, then the aux logit (the host checks that column order)
#pragma unroll
                for (int o = 0; o < NH; ++o) rowv[o] = acc[r][o] + b_fin[o];
                rowv[NH] = acc[r][NH] + b_aux[0];
                if (raw) {
#pragma unroll
                    for (int c = 0; c <= NH; ++c) raw[(r0 + r) * (NH + 1) + c] = rowv[c];
                }
                if (post_out) post_person(rowv, NH + 1, r0 + r, centre, ki, box_conf, post_out, xyzds);
            }
        }
    }
}

// The tail of the mono tile path in ONE launch instead of three: both fused heads' partial sums are added exactly as
// head_reduce_kernel / aux_reduce_kernel add them (bias first, slices in order), the raw row lives in registers (and is
// written out only on request), then post_person.  OUT_F = NH + 1: w_fin's NH outputs, then the aux logit.
template <int NH>
__global__ __launch_bounds__(256) void tail_mono_kernel(const float* __restrict__ part_fin, const float* __restrict__ part_aux,
                                                       int nparts, int64_t m_pad, int64_t m, const float* __restrict__ b_fin,
                                                       const float* __restrict__ b_aux, float* __restrict__ raw,
                                                       const float* __restrict__ centre, Kinv ki,
                                                       const float* __restrict__ box_conf, float* __restrict__ out,
                                                       float* __restrict__ xyzds) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    float r[NH + 1];
#pragma unroll
    for (int o = 0; o < NH; ++o) r[o] = b_fin[o];
    r[NH] = b_aux[0];
    // the slices are added in order (bias first: the sums head_reduce_kernel / aux_reduce_kernel form), but requested eight at a
    // time: a lane's loads of one batch are independent, so a call pays nparts / 8 memory round trips instead of nparts
    // (2048 rows: 8.9 -> ~3 us; the kernel is eight workgroups there, all latency)
    for (int s0 = 0; s0 < nparts; s0 += 8) {
        f32x4 a[8], b[8];
        float c8[8], ax[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sl = s0 + u < nparts ? s0 + u : nparts - 1;   // (past the end: a harmless repeat, not added)
            const float* pf = part_fin + ((int64_t)sl * m_pad + i) * 16;
            a[u] = *(const f32x4*)pf;
            b[u] = *(const f32x4*)(pf + 4);
            c8[u] = NH == 9 ? pf[8] : 0.0f;
            ax[u] = part_aux[(int64_t)sl * m_pad + i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (s0 + u < nparts) {
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    r[o] += a[u][o];
                    r[4 + o] += b[u][o];
                }
                if (NH == 9) r[8] += c8[u];
                r[NH] += ax[u];
            }
        }
    }
    if (raw) {
#pragma unroll
        for (int o = 0; o <= NH; ++o) raw[i * (NH + 1) + o] = r[o];
    }
    post_person(r, NH + 1, i, centre, ki, box_conf, out, xyzds);
}

// ------------------------------------------------------------------------------------------
// post, legacy 'monoloco_p' flavour: extract_outputs_mono (reference process.py:330-360).  raw columns: x, y, z,
// s = log(b/z) shares column 3 with nothing else -- zb = raw[:,2:4] -- then h, w, l, sin, cos.
//   bi = exp(raw3) * raw2 (unnormalize_bi on zb), d = ||xyz||_2, yaw = atan2(raw7, raw8),
//   yaw_ego = back_correct_angles(yaw, xyz).  Same packed layout as post_kernel (aux, conf, uc, vc = 0).
__global__ __launch_bounds__(256) void post_mono_p_kernel(const float* __restrict__ raw, int64_t m,
                                                          float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float* r = raw + i * 9;
    const float x = r[0], y = r[1], z = r[2];
    const float bi = __fmul_rn(expf(r[3]), z);                                            // process.py:131
    const float d = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));  // :351
    const float yaw = atan2f(r[7], r[8]);                                                 // :356
    float ego = __fadd_rn(yaw, atan2f(x, z));                                             // camera.py:203-204
    const float PI_F = 3.14159274101257324f, TWO_PI_F = 6.28318548202514648f;
    if (ego > PI_F) ego = __fsub_rn(ego, TWO_PI_F);
    if (ego < -PI_F) ego = __fadd_rn(ego, TWO_PI_F);
    float* o = out + i * 16;
    o[0] = x; o[1] = y; o[2] = z; o[3] = d;
    o[4] = bi; o[5] = yaw; o[6] = ego; o[7] = 0.f;
    o[8] = r[4]; o[9] = r[5]; o[10] = r[6]; o[11] = 0.f;
    o[12] = r[7]; o[13] = r[8]; o[14] = 0.f; o[15] = 0.f;
}

// ------------------------------------------------------------------------------------------
// dataset preparation rows (reference prep/preprocess_kitti.py:190-253): every matched annotation is normalised
// with the K of ITS image.  One thread per (row, joint); kinv_table holds the inverses of the distinct K,
// k_index the table entry of each row.  With right keypoints the row is the stereo training input
// [L (34), L - R (34)] (preprocess_kitti.py:242-247), else the 34 mono inputs.
__global__ __launch_bounds__(256) void prep_rows_kernel(const float* __restrict__ kps, const float* __restrict__ kps_r,
                                                        int64_t m, const Kinv* __restrict__ kinv_table,
                                                        const int32_t* __restrict__ k_index, float z_met,
                                                        float* __restrict__ x) {
#pragma clang fp contract(off)  // L - R must subtract the ROUNDED right value: no fma(-acc, z, lx)
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= m * NKP) return;
    const int64_t row = id / NKP;
    const int j = (int)(id - row * NKP);
    const float* kr = kinv_table[k_index[row]].k;
    const float* p = kps + row * KPS_ROW;
    const float lx = cam_row(p[j], p[NKP + j], kr + 0, z_met);
    const float ly = cam_row(p[j], p[NKP + j], kr + 3, z_met);
    const int width = kps_r ? 2 * NIN : NIN;
    float* o = x + row * width;
    o[2 * j] = lx;
    o[2 * j + 1] = ly;
    if (kps_r) {
        const float* q = kps_r + row * KPS_ROW;
        float rx = cam_row(q[j], q[NKP + j], kr + 0, z_met);
        float ry = cam_row(q[j], q[NKP + j], kr + 3, z_met);
        asm volatile("" : "+v"(rx), "+v"(ry));  // the rounded products, opaque to the contraction pass
        o[NIN + 2 * j] = lx - rx;
        o[NIN + 2 * j + 1] = ly - ry;
    }
}

}  // namespace mlk
