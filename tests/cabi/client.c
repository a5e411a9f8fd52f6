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
        hipHostFree(p_kps); hipHostFree(p_out); hipFree(d_stage); hipFree(d_buf);
    }
    /* error behaviour: a hot call with a bad argument reports, it does not crash */
    if (ml_loco_forward_mono(h, NULL, m, kinv, NULL, NULL, d_out, d_xyzds, (void*)st) == ML_OK) DIE("null input accepted");
    ML(ml_loco_destroy(h));
    HIP(hipStreamDestroy(st));
    hipFree(d_kps); hipFree(d_conf); hipFree(d_out); hipFree(d_xyzds);
    return (worst <= tol) ? 0 : 1;
}
