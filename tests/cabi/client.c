/* A plain-C client of the C ABI (include/monoloco_hip.h): no Python, no torch types -- what a cgo / JNI / ctypes
 * binding on the reference side would do.  Loads a state_dict dump and one test case written by
 * tests/test_cabi_client.py, runs the fused mono pipeline on device 0 and compares the (x, y, z, d, sigma)
 * block with the expected values (computed by the CPU oracle).  Exit code 0 = within tolerance.
 *
 *   client <weights.bin> <case.bin> <tolerance>
 * weights.bin: int32 n_tensors, then per tensor: int32 key_len, key bytes, int64 numel, float32[numel]
 * case.bin:    int32 in_features, hidden, out_features, num_stage, int64 m, float32 kps[m*51], kinv[9],
 *              box_conf[m], expected[m*5]
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "monoloco_hip.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(2); } while (0)
#define ML(call) do { int rc_ = (call); if (rc_ != ML_OK) DIE("%s -> %d: %s", #call, rc_, ml_last_error()); } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) DIE("%s -> %s", #call, hipGetErrorString(e_)); } while (0)

static void rd(void* dst, size_t n, FILE* f) {
    if (fread(dst, 1, n, f) != n) DIE("short read");
}

int main(int argc, char** argv) {
    if (argc != 4) DIE("usage: client weights.bin case.bin tol");
    const double tol = atof(argv[3]);
    FILE* fc = fopen(argv[2], "rb");
    if (!fc) DIE("cannot open %s", argv[2]);
    int32_t dims[4];
    int64_t m;
    rd(dims, sizeof(dims), fc);
    rd(&m, sizeof(m), fc);
    float* kps = (float*)malloc((size_t)m * 51 * 4);
    float kinv[9];
    float* conf = (float*)malloc((size_t)m * 4);
    float* expect = (float*)malloc((size_t)m * 5 * 4);
    rd(kps, (size_t)m * 51 * 4, fc);
    rd(kinv, sizeof(kinv), fc);
    rd(conf, (size_t)m * 4, fc);
    rd(expect, (size_t)m * 5 * 4, fc);
    fclose(fc);

    if (ml_device_count() < 1) DIE("no HIP device: %s", ml_last_error());
    HIP(hipSetDevice(0));
    ml_loco* h = NULL;
    ML(ml_loco_create(dims[0], dims[1], dims[2], dims[3], &h));
    FILE* fw = fopen(argv[1], "rb");
    if (!fw) DIE("cannot open %s", argv[1]);
    int32_t nt;
    rd(&nt, 4, fw);
    for (int i = 0; i < nt; ++i) {
        int32_t kl;
        char key[256];
        int64_t numel;
        rd(&kl, 4, fw);
        if (kl <= 0 || kl >= (int)sizeof(key)) DIE("bad key length");
        rd(key, (size_t)kl, fw);
        key[kl] = 0;
        rd(&numel, 8, fw);
        float* data = (float*)malloc((size_t)numel * 4);
        rd(data, (size_t)numel * 4, fw);
        ML(ml_loco_set_tensor(h, key, data, numel));
        free(data);
    }
    fclose(fw);
    ML(ml_loco_finalize(h, ML_PREC_F16X2, ML_FLAG_MERGE_W2W3));
    ML(ml_loco_reserve(h, m));

    float *d_kps, *d_conf, *d_out, *d_xyzds;
    HIP(hipMalloc((void**)&d_kps, (size_t)m * 51 * 4));
    HIP(hipMalloc((void**)&d_conf, (size_t)m * 4));
    HIP(hipMalloc((void**)&d_out, (size_t)m * ML_OUT_STRIDE * 4));
    HIP(hipMalloc((void**)&d_xyzds, (size_t)m * ML_XYZDS_STRIDE * 4));
    HIP(hipMemcpy(d_kps, kps, (size_t)m * 51 * 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_conf, conf, (size_t)m * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    HIP(hipStreamCreate(&st));
    ML(ml_loco_forward_mono(h, d_kps, m, kinv, d_conf, NULL, d_out, d_xyzds, (void*)st));
    HIP(hipStreamSynchronize(st));
    float* got = (float*)malloc((size_t)m * 5 * 4);
    HIP(hipMemcpy(got, d_xyzds, (size_t)m * 5 * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int64_t i = 0; i < m * 5; ++i) {
        const double d = fabs((double)got[i] - (double)expect[i]);
        if (!(d <= worst)) worst = d;  /* NaN propagates */
    }
    printf("c-abi client: %lld persons, device bytes %lld, max |xyzds - expected| = %.3e (tolerance %.1e)\n", (long long)m,
           (long long)ml_loco_device_bytes(h), worst, tol);
    /* one image in one call (ml_loco_frame_mono): pinned, device-mapped host memory in and out, no device buffer of the caller is
     * touched for <= 128 persons; its (xyz_pred, d, bi) must be the rows the pipeline above produced */
    {
        const int64_t mf = m < 16 ? m : 16;
        float *p_kps, *p_out, *d_stage, *d_buf;
        HIP(hipHostMalloc((void**)&p_kps, (size_t)mf * 51 * 4, hipHostMallocDefault));
        HIP(hipHostMalloc((void**)&p_out, (size_t)mf * (ML_OUT_STRIDE + ML_POSTGEO_STRIDE) * 4, hipHostMallocDefault));
        HIP(hipMalloc((void**)&d_stage, (size_t)mf * 51 * 4));
        HIP(hipMalloc((void**)&d_buf, (size_t)mf * (ML_OUT_STRIDE + ML_POSTGEO_STRIDE) * 4));
        memcpy(p_kps, kps, (size_t)mf * 51 * 4);
        ML(ml_loco_frame_mono(h, p_kps, mf, kinv, d_stage, d_buf, NULL, p_out, (void*)st));
        double wf = 0.0;
        for (int64_t i = 0; i < mf; ++i) {
            const float* pk = p_out + i * ML_OUT_STRIDE;
            const float* geo = p_out + mf * ML_OUT_STRIDE + i * ML_POSTGEO_STRIDE;
            const float row[5] = {geo[9], geo[10], geo[11], pk[3], pk[4]};
            for (int c = 0; c < 5; ++c) {
                const double d = fabs((double)row[c] - (double)got[i * 5 + c]);
                if (!(d <= wf)) wf = d;
            }
        }
        /* the same call with ordinary (pageable) host memory: the library notices and stages the transfers instead of letting
         * kernels dereference the pointers */
        float* q_kps = (float*)malloc((size_t)mf * 51 * 4);
        float* q_out = (float*)malloc((size_t)mf * (ML_OUT_STRIDE + ML_POSTGEO_STRIDE) * 4);
        memcpy(q_kps, kps, (size_t)mf * 51 * 4);
        ML(ml_loco_frame_mono(h, q_kps, mf, kinv, d_stage, d_buf, NULL, q_out, (void*)st));
        if (memcmp(q_out, p_out, (size_t)mf * (ML_OUT_STRIDE + ML_POSTGEO_STRIDE) * 4) != 0) {
            wf = 1.0;
            fprintf(stderr, "frame entry: pageable and pinned buffers disagree\n");
        }
        free(q_kps); free(q_out);
        printf("c-abi client: frame entry, %lld persons, max |frame - pipeline| = %.3e\n", (long long)mf, wf);
        if (!(wf <= 5e-5)) worst = 1.0;   /* fails the run.  (16 persons take the small-row kernels, the batch above another dense
                                             kernel family: same operands, another fp32 summation order -- a few ulps at 20-60 m) */
        /* the pinned-buffer contract (include/monoloco_hip.h): the handle remembers p_kps / p_out as verified pinned ranges -- tell it to
         * forget them BEFORE they are freed (a later malloc may reuse the addresses as pageable memory) */
        ML(ml_loco_forget_pinned(h, p_kps));
        ML(ml_loco_forget_pinned(h, NULL));
        hipHostFree(p_kps); hipHostFree(p_out); hipFree(d_stage); hipFree(d_buf);
    }
    /* ground-truth association of post_process through the ABI (reference utils/iou.py:44-100): 4 detections x 3 ground-truth boxes,
     * the host entry and the device kernel + host greedy pass must give the pairs a reading of the reference gives: box 3 (highest
     * confidence) takes gt 0, box 0's best gt is then owned, box 1 takes gt 1; left to right: (3,0) comes after ... x1 = 1 > 0? --
     * box 1 starts at x = 20, box 3 at x = 1: order (3,0), (1,1) */
    {
        const double boxes[4 * 5] = {0, 0, 10, 10, 0.9, 20, 0, 30, 10, 0.5, 100, 100, 120, 130, 0.7, 1, 1, 11, 11, 0.95};
        const double gts[3 * 4] = {0, 0, 10, 10, 21, 0, 31, 10, 300, 300, 310, 310};
        const int64_t by_conf[4] = {3, 0, 2, 1};     /* reversed(np.argsort(conf)) */
        const int64_t by_left[4] = {0, 3, 1, 2};     /* np.argsort(x1) */
        int64_t pairs[8], n_pairs = 0;
        int32_t zero_div = 0;
        ML(ml_iou_matches_host(boxes, 4, 5, gts, 3, 4, by_conf, 0.3, by_left, pairs, &n_pairs, &zero_div));
        if (n_pairs != 2 || pairs[0] != 3 || pairs[1] != 0 || pairs[2] != 1 || pairs[3] != 1 || zero_div) DIE("host matching");
        double *d_b, *d_g, *d_v, vmax[4];
        int32_t *d_j, *d_z, jmax[4];
        HIP(hipMalloc((void**)&d_b, sizeof(boxes)));
        HIP(hipMalloc((void**)&d_g, sizeof(gts)));
        HIP(hipMalloc((void**)&d_v, 4 * 8));
        HIP(hipMalloc((void**)&d_j, 4 * 4));
        HIP(hipMalloc((void**)&d_z, 4));
        HIP(hipMemcpy(d_b, boxes, sizeof(boxes), hipMemcpyHostToDevice));
        HIP(hipMemcpy(d_g, gts, sizeof(gts), hipMemcpyHostToDevice));
        HIP(hipMemset(d_z, 0, 4));
        ML(ml_iou_best(d_b, 4, 5, d_g, 3, 4, d_j, d_v, d_z, (void*)st));
        HIP(hipStreamSynchronize(st));
        HIP(hipMemcpy(jmax, d_j, 16, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(vmax, d_v, 32, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(&zero_div, d_z, 4, hipMemcpyDeviceToHost));
        int64_t pairs_d[8], n_d = 0;
        ML(ml_iou_greedy(by_conf, 4, jmax, vmax, 4, 3, 0.3, NULL, pairs_d, &n_d));   /* visiting order: (3,0), (1,1) */
        if (zero_div || n_d != 2 || pairs_d[0] != 3 || pairs_d[1] != 0 || pairs_d[2] != 1 || pairs_d[3] != 1 || vmax[0] != 1.0)
            DIE("device matching");
        printf("c-abi client: ground-truth matching, host and device entries agree (2 pairs)\n");
        hipFree(d_b); hipFree(d_g); hipFree(d_v); hipFree(d_j); hipFree(d_z);
    }
    /* error behaviour: a hot call with a bad argument reports, it does not crash */
    if (ml_loco_forward_mono(h, NULL, m, kinv, NULL, NULL, d_out, d_xyzds, (void*)st) == ML_OK) DIE("null input accepted");
    ML(ml_loco_destroy(h));
    HIP(hipStreamDestroy(st));
    hipFree(d_kps); hipFree(d_conf); hipFree(d_out); hipFree(d_xyzds);
    return (worst <= tol) ? 0 : 1;
}
