// Ground-truth association of Loco.post_process (reference monoloco/network/net.py:170-190): the IoU of every
// detection box with every ground-truth box (monoloco/utils/iou.py:6-41) and the greedy pass by confidence
// (iou.py:44-64), for a whole image in one call instead of m*g scalar Python calls.
//
// All arithmetic is IEEE double in the order the reference's Python expressions evaluate (Python floats ARE doubles;
// -ffp-contract=off keeps the products and sums unfused), including Python's max/min (first argument wins unless the
// second compares strictly greater / smaller) and np.argmax (first maximum, first NaN if any) -- so the matches are
// identical to the reference's, ties included.  A zero union (Python: ZeroDivisionError) is reported, not hidden.
//   * device: iou_best_kernel (one wavefront per detection, lanes stride the ground-truth boxes, cross-lane arg-max)
//             and iou_matrix_kernel (one IoU per lane, coalesced row-major store) for images with many boxes;
//   * host:   the same expressions in plain loops for a handful of boxes, where a launch + copy would cost more than
//             the work (a 16 x 16 frame is 256 IoUs), and the O(m) greedy pass, which is sequential by definition.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/monoloco_hip.h"

namespace {

thread_local char g_merr[256] = "";

int mfail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_merr, sizeof(g_merr), fmt, ap);
    va_end(ap);
    return code;
}

// Python's max(a, b) / min(a, b) on floats: a unless b is strictly greater / smaller (NaN in `a` stays, NaN in `b` is dropped)
__host__ __device__ inline double py_max(double a, double b) { return b > a ? b : a; }
__host__ __device__ inline double py_min(double a, double b) { return b < a ? b : a; }

// calculate_iou(box1, box2), reference iou.py:6-28; *zero is set when the union is 0 (Python raises there)
__host__ __device__ inline double iou_pair(const double* b1, const double* b2, bool* zero) {
    const double xi1 = py_max(b1[0], b2[0]);
    const double yi1 = py_max(b1[1], b2[1]);
    const double xi2 = py_min(b1[2], b2[2]);
    const double yi2 = py_min(b1[3], b2[3]);
    const double inter = py_max(xi2 - xi1, 0.0) * py_max(yi2 - yi1, 0.0);
    const double a1 = (b1[2] - b1[0]) * (b1[3] - b1[1]);
    const double a2 = (b2[2] - b2[0]) * (b2[3] - b2[1]);
    const double uni = a1 + a2 - inter;
    if (uni == 0.0) *zero = true;
    return inter / uni;
}

// np.argmax order on (value, index) candidates: a NaN beats every number, the earlier index breaks ties
__host__ __device__ inline bool argmax_before(double v1, int j1, double v2, int j2) {
    const bool n1 = v1 != v1, n2 = v2 != v2;
    if (n1 || n2) return n1 && (!n2 || j1 < j2);
    return v1 > v2 || (v1 == v2 && j1 < j2);
}

// one wavefront per detection row: lane l looks at ground-truth boxes l, l+64, ...; then the 64 candidates are reduced
__global__ __launch_bounds__(256) void iou_best_kernel(const double* __restrict__ boxes, long m, long ldb,
                                                       const double* __restrict__ gt, long g, long ldg,
                                                       int* __restrict__ jmax, double* __restrict__ vmax, int* __restrict__ zero_div) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    double b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = boxes[row * ldb + c];
    double best = 0.0;
    int bj = 0x7fffffff;   // "no candidate yet": loses every comparison below
    bool zero = false;
    for (long j = lane; j < g; j += 64) {
        double q[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = gt[j * ldg + c];
        const double v = iou_pair(b, q, &zero);
        if (bj == 0x7fffffff || argmax_before(v, (int)j, best, bj)) {
            best = v;
            bj = (int)j;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best, off, 64);
        const int oj = __shfl_xor(bj, off, 64);
        if (oj != 0x7fffffff && (bj == 0x7fffffff || argmax_before(ov, oj, best, bj))) {
            best = ov;
            bj = oj;
        }
    }
    if (__any(zero) && lane == 0) atomicOr(zero_div, 1);
    if (lane == 0) {
        jmax[row] = bj;
        vmax[row] = best;
    }
}

// get_iou_matrix (reference iou.py:31-41): out (m, g) row-major, one IoU per lane
__global__ __launch_bounds__(256) void iou_matrix_kernel(const double* __restrict__ boxes, long m, long ldb,
                                                         const double* __restrict__ gt, long g, long ldg,
                                                         double* __restrict__ out, int* __restrict__ zero_div) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = blockIdx.y;
    bool zero = false;
    if (j < g) {
        double b[4], q[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            b[c] = boxes[row * ldb + c];
            q[c] = gt[j * ldg + c];
        }
        out[row * g + j] = iou_pair(b, q, &zero);
    }
    if (zero) atomicOr(zero_div, 1);
}

int check_boxes(const void* boxes, int64_t m, int64_t ldb, const void* gt, int64_t g, int64_t ldg) {
    if (m < 0 || g < 0 || ldb < 4 || ldg < 4) return mfail(ML_ERR_ARG, "iou: need m, g >= 0 and row strides >= 4 (got %lld %lld %lld %lld)",
                                                           (long long)m, (long long)g, (long long)ldb, (long long)ldg);
    if ((m > 0 && !boxes) || (g > 0 && !gt)) return mfail(ML_ERR_ARG, "iou: null box pointer");
    if (g > 0x7ffffff0LL || m > 0x7ffffff0LL) return mfail(ML_ERR_ARG, "iou: too many boxes");
    return ML_OK;
}

int launched(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mfail(ML_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return ML_OK;
}

}  // namespace

extern "C" {

const char* ml_matching_last_error(void) { return g_merr; }

int ml_iou_best(const double* boxes_dev, int64_t m, int64_t ldb, const double* gt_dev, int64_t g, int64_t ldg, int32_t* jmax_dev,
                double* vmax_dev, int32_t* zero_div_dev, void* stream) {
    if (int rc = check_boxes(boxes_dev, m, ldb, gt_dev, g, ldg)) return rc;
    if (!jmax_dev || !vmax_dev || !zero_div_dev) return mfail(ML_ERR_ARG, "ml_iou_best: null output pointer");
    if (g == 0) return mfail(ML_ERR_ARG, "ml_iou_best: no ground-truth boxes (np.argmax of an empty list raises)");
    if (m == 0) return ML_OK;
    iou_best_kernel<<<dim3((unsigned)((m + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(boxes_dev, m, ldb, gt_dev, g, ldg, jmax_dev, vmax_dev,
                                                                                          zero_div_dev);
    return launched("ml_iou_best");
}

int ml_iou_matrix(const double* boxes_dev, int64_t m, int64_t ldb, const double* gt_dev, int64_t g, int64_t ldg, double* out_dev,
                  int32_t* zero_div_dev, void* stream) {
    if (int rc = check_boxes(boxes_dev, m, ldb, gt_dev, g, ldg)) return rc;
    if (!zero_div_dev || (!out_dev && m * g > 0)) return mfail(ML_ERR_ARG, "ml_iou_matrix: null output pointer");
    if (m == 0 || g == 0) return ML_OK;
    if (m > 65535) return mfail(ML_ERR_ARG, "ml_iou_matrix: at most 65535 detection rows per call");
    iou_matrix_kernel<<<dim3((unsigned)((g + 255) / 256), (unsigned)m), dim3(256), 0, (hipStream_t)stream>>>(boxes_dev, m, ldb, gt_dev, g, ldg,
                                                                                                           out_dev, zero_div_dev);
    return launched("ml_iou_matrix");
}

int ml_iou_matrix_host(const double* boxes, int64_t m, int64_t ldb, const double* gt, int64_t g, int64_t ldg, double* out,
                       int32_t* zero_div) {
    if (int rc = check_boxes(boxes, m, ldb, gt, g, ldg)) return rc;
    if (!zero_div || (!out && m * g > 0)) return mfail(ML_ERR_ARG, "ml_iou_matrix_host: null output pointer");
    bool zero = false;
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < g; ++j) out[i * g + j] = iou_pair(boxes + i * ldb, gt + j * ldg, &zero);
    *zero_div = zero ? 1 : 0;
    return ML_OK;
}

int ml_iou_best_host(const double* boxes, int64_t m, int64_t ldb, const double* gt, int64_t g, int64_t ldg, int32_t* jmax, double* vmax,
                     int32_t* zero_div) {
    if (int rc = check_boxes(boxes, m, ldb, gt, g, ldg)) return rc;
    if (!zero_div || (m > 0 && (!jmax || !vmax))) return mfail(ML_ERR_ARG, "ml_iou_best_host: null output pointer");
    if (g == 0) return mfail(ML_ERR_ARG, "ml_iou_best_host: no ground-truth boxes (np.argmax of an empty list raises)");
    bool zero = false;
    for (int64_t i = 0; i < m; ++i) {
        double best = iou_pair(boxes + i * ldb, gt, &zero);
        int bj = 0;
        for (int64_t j = 1; j < g; ++j) {
            const double v = iou_pair(boxes + i * ldb, gt + j * ldg, &zero);
            if (argmax_before(v, (int)j, best, bj)) {
                best = v;
                bj = (int)j;
            }
        }
        jmax[i] = bj;
        vmax[i] = best;
    }
    *zero_div = zero ? 1 : 0;
    return ML_OK;
}

int ml_iou_greedy(const int64_t* order, int64_t n, const int32_t* jmax, const double* vmax, int64_t m, int64_t g, double iou_min,
                  const int64_t* order_left, int64_t* pairs, int64_t* n_pairs) {
    if (n < 0 || m < 0 || g < 0 || !n_pairs || (n > 0 && (!order || !jmax || !vmax || !pairs)))
        return mfail(ML_ERR_ARG, "ml_iou_greedy: bad argument");
    std::vector<unsigned char> used((size_t)g, 0);
    std::vector<int64_t> gt_of;   // detection -> its ground-truth box (-1: unmatched), only for the left-to-right order
    if (order_left) gt_of.assign((size_t)m, -1);
    int64_t k = 0;
    for (int64_t t = 0; t < n; ++t) {
        const int64_t idx = order[t];
        if (idx < 0 || idx >= m) return mfail(ML_ERR_ARG, "ml_iou_greedy: order[%lld] = %lld outside 0..%lld", (long long)t, (long long)idx, (long long)m);
        const int32_t j = jmax[idx];
        if (j < 0 || j >= g) return mfail(ML_ERR_ARG, "ml_iou_greedy: jmax[%lld] = %d outside 0..%lld", (long long)idx, j, (long long)g);
        if (vmax[idx] >= iou_min && !used[(size_t)j]) {   // (a NaN IoU compares false, like Python's >=)
            used[(size_t)j] = 1;
            if (order_left) {
                if (gt_of[(size_t)idx] < 0) gt_of[(size_t)idx] = j;   // (a detection listed twice in `order` keeps its first match)
            } else {
                pairs[2 * k] = idx;
                pairs[2 * k + 1] = j;
            }
            ++k;
        }
    }
    if (order_left) {
        // reorder_matches (reference iou.py:86-100): the matched detections in the order of np.argsort(left edges)
        k = 0;
        for (int64_t t = 0; t < m; ++t) {
            const int64_t idx = order_left[t];
            if (idx < 0 || idx >= m) return mfail(ML_ERR_ARG, "ml_iou_greedy: order_left[%lld] = %lld outside 0..%lld", (long long)t, (long long)idx, (long long)m);
            if (gt_of[(size_t)idx] >= 0) {
                pairs[2 * k] = idx;
                pairs[2 * k + 1] = gt_of[(size_t)idx];
                gt_of[(size_t)idx] = -1;
                ++k;
            }
        }
    }
    *n_pairs = k;
    return ML_OK;
}

int ml_iou_matches_host(const double* boxes, int64_t m, int64_t ldb, const double* gt, int64_t g, int64_t ldg, const int64_t* order,
                        double iou_min, const int64_t* order_left, int64_t* pairs, int64_t* n_pairs, int32_t* zero_div) {
    if (m < 0 || m > (1 << 20)) return mfail(ML_ERR_ARG, "ml_iou_matches_host: row count out of range");
    std::vector<int32_t> jmax((size_t)m);
    std::vector<double> vmax((size_t)m);
    if (int rc = ml_iou_best_host(boxes, m, ldb, gt, g, ldg, jmax.data(), vmax.data(), zero_div)) return rc;
    return ml_iou_greedy(order, m, jmax.data(), vmax.data(), m, g, iou_min, order_left, pairs, n_pairs);
}

int ml_xyz_from_distance_host(const float* d, int d_is_scalar, const float* centres, int64_t m, float* out) {
    // xyz_from_distance (reference utils/camera.py:161-177) on host arrays, for the few matched persons of a frame whose geometry
    // block is already on the host: centre * d / sqrt(1 + x^2 + y^2), fp32, one rounding per operation in the reference's order
    // (-ffp-contract=off) -- the same bits as ml_xyz_from_distance's kernel
    if (m < 0 || (m > 0 && (!d || !centres || !out))) return mfail(ML_ERR_ARG, "ml_xyz_from_distance_host: bad argument");
    for (int64_t i = 0; i < m; ++i) {
        const float* c = centres + 3 * i;
        const float dd = d[d_is_scalar ? 0 : i];
        const float xx = c[0] * c[0], yy = c[1] * c[1];
        const float s1 = 1.0f + xx;
        const float nrm = sqrtf(s1 + yy);
        for (int j = 0; j < 3; ++j) {
            const float num = c[j] * dd;
            out[3 * i + j] = num / nrm;
        }
    }
    return ML_OK;
}

}  // extern "C"
