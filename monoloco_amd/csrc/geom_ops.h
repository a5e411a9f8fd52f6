// geom_ops.h -- stand-alone element-wise geometry behind the reference's utility API
// (monoloco/utils/camera.py), one thread per point/person.  Inside the fused pipeline the same
// math lives in prep_kernel / post_kernel; these exist so that the host mirror of
// pixel_to_camera / get_keypoints / xyz_from_distance / to_cartesian / back_correct_angles runs on
// the device too instead of falling back to host arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "geom_kernels.h"

namespace mlk {

// camera.py:10-29: out[i] = [u, v, 1] . Kinv^T * z_met, uv (n,2) -> out (n,3)
__global__ __launch_bounds__(256) void pix2cam_kernel(const float* __restrict__ uv, int64_t n, Kinv ki, float z,
                                                      float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float u = uv[i * 2], v = uv[i * 2 + 1];
    out[i * 3 + 0] = cam_row(u, v, ki.k + 0, z);
    out[i * 3 + 1] = cam_row(u, v, ki.k + 3, z);
    out[i * 3 + 2] = cam_row(u, v, ki.k + 6, z);
}

// camera.py:69-107: mode 0 center, 1 bottom, 2 head (0:5), 3 shoulder (5:7), 4 hip (11:13), 5 ankle (15:17)
__global__ __launch_bounds__(256) void keypoints_kernel(const float* __restrict__ kps, int64_t m, int mode,
                                                        float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float* u = kps + i * KPS_ROW;
    const float* v = u + NKP;
    float ou, ov;
    if (mode <= 1) {
        float umin = u[0], umax = u[0], vmin = v[0], vmax = v[0];
        for (int j = 1; j < NKP; ++j) {
            umin = __builtin_fminf(umin, u[j]);
            umax = __builtin_fmaxf(umax, u[j]);
            vmin = __builtin_fminf(vmin, v[j]);
            vmax = __builtin_fmaxf(vmax, v[j]);
        }
        ou = __fadd_rn(__fmul_rn(__fsub_rn(umax, umin), 0.5f), umin);
        ov = mode == 0 ? __fadd_rn(__fmul_rn(__fsub_rn(vmax, vmin), 0.5f), vmin) : vmax;
    } else {
        int a, b;
        if (mode == 2) { a = 0; b = 5; }
        else if (mode == 3) { a = 5; b = 7; }
        else if (mode == 4) { a = 11; b = 13; }
        else { a = 15; b = 17; }
        float su = 0.f, sv = 0.f;
        for (int j = a; j < b; ++j) {
            su = __fadd_rn(su, u[j]);
            sv = __fadd_rn(sv, v[j]);
        }
        ou = su / (float)(b - a);
        ov = sv / (float)(b - a);
    }
    out[i * 2] = ou;
    out[i * 2 + 1] = ov;
}

// camera.py:161-177: xyz = c * d / sqrt(1 + cx^2 + cy^2); d (m) broadcast if d_stride == 0
__global__ __launch_bounds__(256) void xyz_from_distance_kernel(const float* __restrict__ d, int d_stride,
                                                                const float* __restrict__ c, int64_t m,
                                                                float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float dd = d[i * d_stride];
    const float cx = c[i * 3], cy = c[i * 3 + 1], cz = c[i * 3 + 2];
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(1.0f, __fmul_rn(cx, cx)), __fmul_rn(cy, cy)));
    out[i * 3 + 0] = __fmul_rn(cx, dd) / nrm;
    out[i * 3 + 1] = __fmul_rn(cy, dd) / nrm;
    out[i * 3 + 2] = __fmul_rn(cz, dd) / nrm;
}

// camera.py:223-237 (tensor branch): rtp rows are (theta, psi, r) for mode 'x' (0) / 'y' (1) -> (m,1);
// mode 2: generic (r, theta, psi) rows -> xyz (m,3) as the function's fall-through branch does
__global__ __launch_bounds__(256) void to_cartesian_kernel(const float* __restrict__ rtp, int64_t m, int mode,
                                                           float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float a = rtp[i * 3], b = rtp[i * 3 + 1], c = rtp[i * 3 + 2];
    if (mode == 0) {
        out[i] = __fmul_rn(__fmul_rn(c, sinf(b)), cosf(a));
    } else if (mode == 1) {
        out[i] = __fmul_rn(c, cosf(b));
    } else {
        out[i * 3 + 0] = __fmul_rn(__fmul_rn(a, sinf(c)), cosf(b));
        out[i * 3 + 1] = __fmul_rn(a, cosf(c));
        out[i * 3 + 2] = __fmul_rn(__fmul_rn(a, sinf(c)), sinf(b));
    }
}

// camera.py:202-208: yaw + atan2(x, z), wrapped once into (-pi, pi]
__global__ __launch_bounds__(256) void back_correct_kernel(const float* __restrict__ yaw,
                                                           const float* __restrict__ xyz, int64_t m,
                                                           float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    float e = __fadd_rn(yaw[i], atan2f(xyz[i * 3], xyz[i * 3 + 2]));
    const float PI_F = 3.14159274101257324f, TWO_PI_F = 6.28318548202514648f;
    if (e > PI_F) e = __fsub_rn(e, TWO_PI_F);
    if (e < -PI_F) e = __fadd_rn(e, TWO_PI_F);
    out[i] = e;
}

// The whole per-person geometry of Loco.post_process (net.py:195-215) in one pass over the keypoints:
//   out[i] = { uv_shoulder(2), uv_head(2), uv_center(2), xy_center(3) = pixel_to_camera(uv_center, K, 1),
//              xyz_pred(3) = xyz_from_distance(d[i], xy_center) }      (12 floats)
// Same arithmetic, in the same order, as keypoints_kernel / pix2cam_kernel / xyz_from_distance_kernel.
__global__ __launch_bounds__(256) void post_geometry_kernel(const float* __restrict__ kps, int64_t m, Kinv ki,
                                                            const float* __restrict__ d, int64_t d_stride,
                                                            float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    post_geometry_person(kps, i, ki, d ? d[i * d_stride] : 0.0f, out);
}

// The end of a stereo frame in ONE launch (ml_loco_frame_stereo), one lane per left person: stereo_best_kernel's arg-max over the
// person's mr pair rows (first maximum; ties and NaN candidates counted), post_person of the winning row (post_kernel), the
// post_process geometry of the person (post_geometry_kernel) -- same arithmetic, same order -- then the frame's completion:
// every workgroup makes its stores visible system-wide and checks in; the last one writes the tie count behind the arg-max
// indices, resets both device words and releases the completion word (FrameDone, geom_kernels.h).  words_dst = int32 [ties, best[ml]].
__global__ __launch_bounds__(256) void stereo_tail_frame_kernel(const float* __restrict__ raw_all, int out_f, int64_t ml, int64_t mr,
                                                                const float* __restrict__ centre, Kinv ki, const float* __restrict__ kps,
                                                                float* __restrict__ out, float* __restrict__ xyzds,
                                                                float* __restrict__ geo, int32_t* __restrict__ row_index,
                                                                int32_t* __restrict__ ties_dev, int32_t* __restrict__ words_dst,
                                                                FrameDone done) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < ml) {
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
        words_dst[1 + i] = bj;
        row_index[i] = (int32_t)(i * mr + bj);
        if (cnt > 1 || has_nan) atomicAdd(ties_dev, 1);
        const float* r = raw_all + (i * mr + bj) * out_f;
        post_person(r, out_f, i, centre, ki, (const float*)nullptr, out, xyzds);
        post_geometry_person(kps, i, ki, r[2], geo);   // (r[2] = the distance post_person stores in column 3 of the packed row)
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(done.arrive, 1) == (int)gridDim.x - 1) {
            words_dst[0] = __hip_atomic_load(ties_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ties_dev, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done.arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            if (done.flag) __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace mlk
