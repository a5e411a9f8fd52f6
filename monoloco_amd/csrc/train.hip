// train.hip -- C ABI of the training step of LocoModel on gfx950 (BASELINE config 5).
// Stands in for one iteration of the reference's training loop (monoloco/train/trainer.py:150-161):
//   model.train() forward (batch-stat BatchNorm + running-stat update, dropout)  architectures.py:48-102
//   MultiTaskLoss                                                               losses.py:59-73, 112-131
//   backward, clip_grad_norm_(3), Adam, per-batch StepLR                        trainer.py:157-161, 128-131
// fp32 throughout, reductions in fp64.  GEMMs: exact-fp32 MFMA (sgemm_kernel); from g_train_fast_rows rows on, the forward
// and data-gradient GEMMs of the hidden x hidden layers run on the inference path's 3-product fp16 MFMA kernel
// (dense_kernel_w4<3, false, *, -2>, fp32 in / fp32 out, fp32-class accuracy).  Kernels: train_kernels.h.
#include "../../include/monoloco_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "train_kernels.h"
#include "dense_kernel_w4.h"
#include "train_mid.h"

namespace {

thread_local char t_err[512] = "";
int tfail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
    return code;
}
#define T_TRY(expr)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return tfail(ML_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct Slot {
    int64_t off = 0, numel = 0;
    bool is_param = true;  // false: BN running statistic
};

}  // namespace

extern "C" const char* ml_train_last_error(void) { return t_err; }

// fp64 reduction slots per step: a step takes 6 S + 8 of them (2 per BatchNorm forward / backward, the bias sums); sized per
// trainer from its num_stage, so a fresh pre-zeroed slot always exists (two reductions never share one)
inline int red_slots_for(int num_stage) { return 6 * num_stage + 16; }
constexpr int HL_GRID = 256;   // most workgroups heads_loss_kernel is launched with (partial sums the host adds)

struct ml_trainer {
    int in_f, H, C, S;
    float p_drop, lr0, gamma;
    int sched_step;
    uint32_t seed;
    int64_t step = 0;
    int64_t fwd_pending_rows = 0;   // rows of a forward whose backward has not run yet (0: none)
    const float* fwd_pending_x = nullptr;
    int64_t fwd_calls = 0;          // ml_trainer_forward_train calls: the dropout masks of a forward / backward pair outside ml_trainer_step
                                    // (no optimizer step moves `step` there) differ from call to call
    std::map<std::string, Slot> slots;
    int64_t n_param = 0, n_stat = 0;
    float *w = nullptr, *g = nullptr, *m1 = nullptr, *m2 = nullptr, *stat = nullptr;
    // activation workspace (rows cap)
    int64_t cap = 0;
    std::vector<float*> bufs;  // all (cap x H) fp32 buffers
    float *d_out = nullptr, *d_dout = nullptr, *d_y2aux = nullptr;
    float *bn_mean = nullptr, *bn_invstd = nullptr;  // (nbn x H)
    double* d_red = nullptr;                          // current slot of the fp64 reduction scratch (2*H + 32 doubles)
    double* d_red_base = nullptr;                     // red_slots slots, zeroed once per step (one memset instead of ~27)
    int red_slot = 0, red_slots = 0;
    float* d_splitk = nullptr;                        // split-K partials of the weight-gradient GEMMs
    size_t splitk_cap = 0;                            // floats
    int nbn = 0;
    // fast path (3-product fp16 MFMA GEMMs, see fast_linear_fwd / fast_linear_bwd_data): line-format copies of the
    // activations that feed an H x H Linear (backward: of dz), and per such Linear the device-packed weights
    std::vector<char*> lbufs;                         // 2S + 2 buffers of cap x H lines
    std::vector<char*> wl;                            // 2S + 2 packed weight images (H x H lines)
    std::vector<float*> wbs;                          // bias * 2^e
    std::vector<char*> wlT;                           // ... and of W^T (data gradient); all images of a step are packed up front
    mlt::WLayer* d_wdesc = nullptr;                   // device table of the 2S + 2 Linears for wmax_multi / wpack_multi
    mlt::WLayer* d_wdesc_m = nullptr;                 // ... with w2 and w3 replaced by their product (2S + 1 entries; the merged pair)
    bool packed_all = false;
    mlt::ColSumItems csf;                             // deferred fp64 column sums -> fp32 gradient vectors (flushed before the optimizer)
    bool csf_defer = false;                          // this step's images are already packed (pack_all_weights)
    float* wsc_base = nullptr;                        // per image 8 scale words (train_kernels.h, wmax_kernel)
    char *tl_dz = nullptr, *tl_x = nullptr;           // transposed lines [H][capT] of dz and of a layer input (dW = dz^T . x)
    int64_t capT = 0;
    int ks = 1;                                       // split of the batch reduction of the fast weight-gradient GEMM
    float* zero_bias = nullptr;                       // H zeros (the data-gradient GEMMs have no bias)
    // round 6, large-batch route: w2 -> w3 as ONE Linear (train_kernels.h "w2 -> w3 pair"): W3 W2 and W3 b2 + b3 (packed as Linear slot
    // 2S + 2), u = W2^T w_aux, c = w_aux . b2 + b_aux, v = daux^T a_S, and the H x H scratch the weight gradients go through
    int merge23 = 1;
    int head_in_bwd = 1;   // ... and b3's backward passes form dy3 = dout . w_fin themselves (0: skinny_out_kernel writes it; MONOLOCO_TRAIN_HEAD_IN_BWD)
    float *d_w32 = nullptr, *d_b32 = nullptr, *d_u23 = nullptr, *d_c23 = nullptr, *d_v23 = nullptr, *d_m23 = nullptr;
    int n_cu = 256;
    // AutoTuneMultiTaskLoss (reference losses.py:17-43, trainer.py:95-96): one learnable log_sigma per task, optimised by
    // the same Adam (same lr schedule, NOT clipped: clip_grad_norm_ sees model.parameters() only), not part of the
    // state_dict.  Eight scalars: kept and updated on the host, their task weights uploaded per step.
    bool auto_tune = false;
    bool weighted = false;   // lambdas other than all 1 (the reference's Trainer.lambdas, trainer.py:42)
    float lambdas[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    float log_sigma[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ls_m1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ls_m2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float h_tw[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    float* d_tw = nullptr;
    // route selection, per handle (ml_trainer_set_route): 0 auto, 1 exact-fp32 GEMMs only, 2 mid route whenever it can run,
    // 3 large-batch (fast) route whenever it can run.  auto: >= fast_rows rows -> fast, below -> mid, else exact.
    int route = 0;
    int64_t fast_rows = 4096;
    // mid route (train_mid.h): per-workgroup partial sums of squares of the weight-gradient GEMMs, the tables of the
    // H x H matrices / the narrow segments between them, a pinned landing zone for the loss values
    double* d_ssq = nullptr;
    double* d_gn = nullptr;          // GN_PARTS partial sums of the gradient norm
    // the weight gradients are off the critical path (only the optimizer needs them): they run on a side stream beside the
    // bwd_apply -> data-gradient chain.  ev_dz[k]: dz of Linear k is ready; ev_w[k]: its weight gradient has read it
    hipStream_t st2 = nullptr;
    std::vector<hipEvent_t> ev_dz, ev_w;
    int side_stream = 0;             // ml_trainer_set_tuning: 1 = weight gradients on the side stream (measured: pays from ~2000 rows)
    int ssq_per_mat = 0;
    int lines_chain = 1;             // large-batch route: dz / inner activations / the residual stream as lines ONLY (round 4); 0 keeps their
                                     // fp32 copies and the dz -> lines pass (ml_trainer_set_tuning dw_layout 2: the same-bits test's reference)
    int dw_trans = 1;                // weight-gradient GEMM operands: 1 = reduction-major lines as they lie (dense_kernel_w4<.., -3, true>),
                                     // 0 = transposed copies (tlines_kernel; the round-2 path, kept for the bit-identity test)
    double* d_colpart = nullptr;     // [cap / 128][2][H]: per 128-row block column sums / sums of squares the forward GEMM's epilogue leaves
    float* d_colmax = nullptr;       // per BatchNorm layer 2 H floats: column maxima of |dy|, |xhat| (bn_bwd_lines_kernel's scale bound)
    char* dzl = nullptr;             // [capT][H] lines: the scaled gradient dz of the layer being back-propagated
    int64_t pad_m = -1;              // rows [pad_m, round_up(pad_m, 64 ks)) of every line buffer are zero
    int apply_cols = 8;              // columns a workgroup of the column-owner kernels takes (4 | 8 | 16): ml_trainer_set_tuning
    int pair_gemm = 1;               // both gradients of a Linear in one launch (xgemm_pair_kernel); side_stream = 2 turns it off
    double* h_loss = nullptr;        // pinned: up to HL_GRID x LOSS_NV partial sums
    double* h_loss_dev = nullptr;    // the device address of h_loss (heads_loss_kernel writes there directly), null: copy from d_lpart
    double* d_lpart = nullptr;       // the same on the device (heads_loss_kernel)
    float* w_snap = nullptr;         // ml_trainer_snapshot: parameters + running statistics kept on the device (best epoch)
    double last_vals[10] = {0};      // plain task means + validation-type d (L1) / ori (angle, radians) of the last step's outputs
    std::vector<int64_t> mat_off;    // flat offsets of the H x H weight matrices by Linear slot
    mlt::AdamSegs segs;
    int last_route = -1;             // route the last step took (0 exact, 1 fast, 2 mid): ml_trainer_last_route
};


namespace {

int add_slot(ml_trainer* t, const std::string& key, int64_t numel, bool is_param) {
    Slot s;
    s.numel = numel;
    s.is_param = is_param;
    s.off = is_param ? t->n_param : t->n_stat;
    (is_param ? t->n_param : t->n_stat) += numel;
    t->slots[key] = s;
    return 0;
}
void add_linear(ml_trainer* t, const std::string& n, int out, int in) {
    add_slot(t, n + ".weight", (int64_t)out * in, true);
    add_slot(t, n + ".bias", out, true);
}
void add_bn(ml_trainer* t, const std::string& n, int h) {
    add_slot(t, n + ".weight", h, true);
    add_slot(t, n + ".bias", h, true);
    add_slot(t, n + ".running_mean", h, false);
    add_slot(t, n + ".running_var", h, false);
}
float* P(ml_trainer* t, const std::string& k) { return t->w + t->slots[k].off; }
float* G(ml_trainer* t, const std::string& k) { return t->g + t->slots[k].off; }
float* ST(ml_trainer* t, const std::string& k) { return t->stat + t->slots[k].off; }

// c (M x N, row stride ldc) (+)= a (M x K) . b (K x N) + bias on the exact-fp32 MFMA GEMM.  When the output has too
// few 128x128 tiles to fill the chip (a weight gradient: small output, K = batch; or any layer of a small batch) the
// reduction is split over blockIdx.z into a partial buffer and the partials are added in a fixed order
// (deterministic, unlike atomics).
int gemm(ml_trainer* t, hipStream_t st, const float* a, long sai, long sak, const float* b, long sbk, long sbj, const float* bias,
         float* c, int ldc, int M, int N, int K, int accumulate) {
    const int tiles = ((N + mlt::GBN - 1) / mlt::GBN) * ((M + mlt::GBM - 1) / mlt::GBM);
    int splits = 1;
    while (splits < 32 && tiles * splits < 512 && K / (splits * 2) >= 128) splits *= 2;
    if ((size_t)splits * M * ldc > t->splitk_cap) splits = 1;
    int kchunk = K;
    if (splits > 1) {
        kchunk = ((K + splits - 1) / splits + 15) / 16 * 16;
        splits = (K + kchunk - 1) / kchunk;  // the rounding may leave fewer, all non-empty, chunks
    }
    mlt::GemmParams p;
    p.a = a; p.b = b; p.bias = bias; p.c = c;
    p.M = M; p.N = N; p.K = K;
    p.sai = sai; p.sak = sak; p.sbk = sbk; p.sbj = sbj;
    p.ldc = ldc; p.accumulate = accumulate;
    p.kchunk = K;
    if (splits > 1) {
        p.c = t->d_splitk;
        p.bias = nullptr;
        p.accumulate = 0;
        p.kchunk = kchunk;
    }
    dim3 grid((N + mlt::GBN - 1) / mlt::GBN, (M + mlt::GBM - 1) / mlt::GBM, splits);
    hipLaunchKernelGGL(mlt::sgemm_kernel, grid, dim3(256), 0, st, p);
    if (splits > 1) {
        const int64_t n = (int64_t)M * N;
        hipLaunchKernelGGL(mlt::splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           (const float*)t->d_splitk, splits, M, N, ldc, bias, accumulate, c);
    }
    if (hipGetLastError() != hipSuccess) return tfail(ML_ERR_HIP, "sgemm launch failed");
    return 0;
}

// y (m x n) = x (m x k) . W^T + b           (nn.Linear forward)
int linear_fwd(ml_trainer* t, hipStream_t st, const float* x, int ldx, const float* W, const float* b, float* y, int ldy, int m, int n, int k) {
    return gemm(t, st, x, ldx, 1, W, 1, k, b, y, ldy, m, n, k, 0);
}
// dx (m x k) (+)= dy (m x n) . W
int linear_bwd_data(ml_trainer* t, hipStream_t st, const float* dy, int lddy, const float* W, float* dx, int lddx, int m, int n, int k, int acc) {
    return gemm(t, st, dy, lddy, 1, W, k, 1, nullptr, dx, lddx, m, k, n, acc);
}
// dW (n x k) = dy^T (n x m) . x (m x k)
int linear_bwd_weight(ml_trainer* t, hipStream_t st, const float* dy, int lddy, const float* x, int ldx, float* dW, int m, int n,
                      int k) {
    return gemm(t, st, dy, 1, lddy, x, ldx, 1, nullptr, dW, k, n, k, m, 0);
}

unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

// column sums (fp64) of z (m x n) and of z*w2 (or z*z) into t->d_red[0..n) and [n..2n)
int col_stats(ml_trainer* t, hipStream_t st, const float* z, const float* w2, int64_t m, int n) {
    // every reduction of a step gets a fresh, pre-zeroed slot (ml_trainer_step zeroes them all with one memset)
    if (t->red_slot + 1 >= t->red_slots) return tfail(ML_ERR_HIP, "out of reduction slots (internal)");
    t->d_red = t->d_red_base + (size_t)(++t->red_slot) * (2 * t->H + 32);
    int gy = (int)((m + 255) / 256);  // 16 rows per workgroup pass; enough workgroups to fill the chip, few atomics
    if (gy > 128) gy = 128;
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(mlt::col_stats_kernel, dim3((n + 63) / 64, gy), dim3(256), 0, st, z, w2, m, n, t->d_red, t->d_red + n);
    return 0;
}

// a vector of fp64 column sums (a reduction slot of this step) becomes an fp32 gradient vector: one launch each, or -- large-batch
// route -- noted and converted by ONE launch in front of the optimizer (the slots live until the end of the step)
void col_sum_to_float(ml_trainer* t, hipStream_t st, const double* src, int n, float* dst) {
    if (t->csf_defer && t->csf.count < 16) {
        t->csf.src[t->csf.count] = src;
        t->csf.dst[t->csf.count] = dst;
        t->csf.n[t->csf.count] = n;
        ++t->csf.count;
        return;
    }
    hipLaunchKernelGGL(mlt::col_sum_to_float_kernel, dim3(nblk(n)), dim3(256), 0, st, src, n, dst);
}
void flush_col_sums(ml_trainer* t, hipStream_t st) {
    if (t->csf.count > 0) {
        int nmax = 0;
        for (int i = 0; i < t->csf.count; ++i) nmax = t->csf.n[i] > nmax ? t->csf.n[i] : nmax;
        hipLaunchKernelGGL(mlt::col_sum_multi_kernel, dim3(nblk(nmax), t->csf.count), dim3(256), 0, st, t->csf);
    }
    t->csf.count = 0;
}

// ---- skinny products (input layer, output heads) on their own kernels (train_kernels.h) at every batch size; callers
// check skinny_ok (64-column workgroups)
bool skinny_ok(const ml_trainer* t, int nc) { return t->H % 64 == 0 && nc >= 1 && nc <= mlt::SK_NC; }

int skinny_out(ml_trainer* t, hipStream_t st, const float* s, int lds, int nc, const float* w, int64_t wsc, int64_t wsj,
               const float* bias, float* out, int64_t m, int accumulate) {
    int64_t gy = (m + 63) / 64;
    if (gy > 128) gy = 128;
    hipLaunchKernelGGL(mlt::skinny_out_kernel, dim3(t->H / 64, (unsigned)gy), dim3(256), 0, st, s, lds, nc, w, wsc, wsj, bias, out, m,
                       t->H, accumulate);
    if (hipGetLastError() != hipSuccess) return tfail(ML_ERR_HIP, "skinny product launch failed");
    return 0;
}

// dst (nc x H, or H x nc when transpose) = s^T (nc x m) . x (m x H)
int skinny_dw(ml_trainer* t, hipStream_t st, const float* s, int lds, int nc, const float* x, int64_t m, float* dst, int transpose) {
    int64_t gy = (m + 15) / 16;
    if (gy > 64) gy = 64;
    while (gy > 1 && (size_t)gy * nc * t->H > t->splitk_cap) gy /= 2;
    const dim3 block(256);
    if (nc == 1) hipLaunchKernelGGL(mlt::skinny_dw_kernel<1>, dim3(t->H / 64, (unsigned)gy, 1), block, 0, st, s, lds, nc, x, m, t->H, t->d_splitk);
    else if (nc <= 12) hipLaunchKernelGGL(mlt::skinny_dw_kernel<12>, dim3(t->H / 64, (unsigned)gy, 1), block, 0, st, s, lds, nc, x, m, t->H, t->d_splitk);
    else hipLaunchKernelGGL(mlt::skinny_dw_kernel<36>, dim3(t->H / 64, (unsigned)gy, (nc + 35) / 36), block, 0, st, s, lds, nc, x, m, t->H, t->d_splitk);
    hipLaunchKernelGGL(mlt::skinny_reduce_kernel, dim3(nblk((int64_t)nc * t->H)), block, 0, st, (const float*)t->d_splitk, (int)gy, nc,
                       t->H, dst, transpose);
    if (hipGetLastError() != hipSuccess) return tfail(ML_ERR_HIP, "skinny weight-gradient launch failed");
    return 0;
}

// out[i][c] = x[i] . w[c] + b[c], c < nc in {1, 8, 9}; false: no kernel for this shape (H % 4, nc * H <= 15360 floats of LDS): the caller
// takes the GEMM
bool skinny_heads(ml_trainer* t, hipStream_t st, const float* x, int64_t m, const float* w, const float* b, int nc, float* out, int ldo) {
    if (t->H % 4 != 0 || (int64_t)nc * t->H > 15360) return false;
    int64_t g = (m + 3) / 4;
    if (g > 2048) g = 2048;
    const dim3 grid((unsigned)g), block(256);
    const size_t lds = (size_t)nc * t->H * 4;   // (<= 61440: no attribute needed)
    if (nc == 1) hipLaunchKernelGGL(mlt::skinny_heads_kernel<1>, grid, block, lds, st, x, m, t->H, w, b, out, ldo);
    else if (nc == 8) hipLaunchKernelGGL(mlt::skinny_heads_kernel<8>, grid, block, lds, st, x, m, t->H, w, b, out, ldo);
    else if (nc == 9) hipLaunchKernelGGL(mlt::skinny_heads_kernel<9>, grid, block, lds, st, x, m, t->H, w, b, out, ldo);
    else return false;
    return hipGetLastError() == hipSuccess;   // (a failed launch sends the caller to the generic GEMM, which reports its own errors)
}

struct Block {  // Linear + BatchNorm + ReLU + Dropout
    std::string lin, bn;
    int bn_idx;
    int in_dim;
    const float* x = nullptr;  // input activations (m x in_dim)
    const char* x_lines = nullptr;   // ... and, on the large-batch route, the same activations as lines (the forward's GEMM operand)
    float* z = nullptr;        // pre-BN (m x H)
    float* y = nullptr;        // output (m x H) (residual already added if any)
    uint32_t site;
    // round 6, the block right under the w_fin head on the large-batch route: its incoming gradient dy = hs . hw is formed inside the two
    // backward passes (bwd_stats_kernel<NHG>) instead of being written by skinny_out_kernel and read back twice
    const float* hs = nullptr;
    int ldh = 0, nh = 0;
    const float* hw = nullptr;
};

// one launch of the inference path's dense kernel with the fp32 epilogue: out (m x H fp32) [+]= x_lines . w_lines^T + bias
int launch_fast_gemm(ml_trainer* t, hipStream_t st, const char* x_lines, const char* w_lines, const float* bias_scaled,
                     const float* descale_ptr, float* out, int64_t m, bool accumulate, bool col_stats_too = false) {
    const int H = t->H;
    mlk::DenseParams p;
    p.x = x_lines;
    p.w = w_lines;
    p.bias = nullptr;
    p.bias_scaled = bias_scaled;
    p.res = accumulate ? (const char*)out : nullptr;
    p.y = (char*)out;
    p.descale = 1.0f;
    p.descale_ptr = descale_ptr;
    p.M_pad = (int)((m + 255) / 256 * 256);
    p.N = H;
    p.K = H;
    p.relu = 0;
    p.debug = 0;
    p.trace = nullptr;
    p.head_w = nullptr;
    p.head_part = nullptr;
    if (col_stats_too && !accumulate) {   // BatchNorm batch statistics of the tile as it is stored (rows < m)
        p.colpart = t->d_colpart;
        p.m_valid = (int)m;
    }
    const int tiles = (p.M_pad / mlk::BM) * (p.N / mlk::BN);
    const dim3 grid(tiles < t->n_cu ? tiles : t->n_cu), block(mlk::W4_THREADS);
    if (accumulate) hipLaunchKernelGGL((mlk::dense_kernel_w4<3, false, true, -2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((mlk::dense_kernel_w4<3, false, false, -2>), grid, block, 0, st, p);
    if (hipGetLastError() != hipSuccess) return tfail(ML_ERR_HIP, "fast GEMM launch failed");
    return 0;
}

// z (m x H, fp32) = x . W^T + b for an H x H Linear whose input exists as lines: the weights are scaled / split / packed on
// the device (they change every step), then ONE launch of the inference path's dense kernel with the fp32 epilogue
// (dense_kernel_w4<3, false, false, -2>): 3 fp16 MFMAs per product, fp32 accumulate -- fp32-class accuracy (~2^-22
// relative per operand) at ~3x the rate of the exact-fp32 MFMA GEMM.  `slot` = which packed-weight image to use.
int fast_linear_fwd(ml_trainer* t, hipStream_t st, const char* x_lines, const std::string& lin, float* z, int64_t m, int slot,
                    bool col_stats_too = false) {
    const int H = t->H;
    float* sc = t->wsc_base + 8 * slot;   // (zeroed at the start of the step)
    if (!t->packed_all) {
        hipLaunchKernelGGL(mlt::wmax_kernel, dim3(64), dim3(256), 0, st, (const float*)P(t, lin + ".weight"), (int64_t)H * H, sc + 2);
        hipLaunchKernelGGL(mlt::wpack_kernel<false>, dim3(nblk((int64_t)H * H / 8)), dim3(256), 0, st, (const float*)P(t, lin + ".weight"),
                           (const float*)P(t, lin + ".bias"), H, H, H, sc, t->wl[slot], t->wbs[slot]);
    }
    return launch_fast_gemm(t, st, x_lines, t->wl[slot], t->wbs[slot], sc + 1, z, m, false, col_stats_too);
}

// dx (m x H, fp32) [+]= dz . W for the same Linear, dz given as lines: the image of W^T replaces the forward image (which
// the step no longer needs), its scale is the forward's.
int fast_linear_bwd_data(ml_trainer* t, hipStream_t st, const char* dz_lines, const std::string& lin, float* dx, int64_t m, int slot,
                         bool accumulate) {
    const int H = t->H;
    float* sc = t->wsc_base + 8 * slot;
    if (t->packed_all) return launch_fast_gemm(t, st, dz_lines, t->wlT[slot], t->zero_bias, sc + 4, dx, m, accumulate);
    hipLaunchKernelGGL(mlt::wpack_kernel<true>, dim3(nblk((int64_t)H * H / 8)), dim3(256), 0, st, (const float*)P(t, lin + ".weight"),
                       (const float*)nullptr, H, H, H, sc, t->wl[slot], (float*)nullptr);
    return launch_fast_gemm(t, st, dz_lines, t->wl[slot], t->zero_bias, sc + 4, dx, m, accumulate);
}

int ensure_tl(ml_trainer* t) {   // the comparison path's transposed operand copies
    if (t->tl_dz) return 0;
    T_TRY(hipMalloc((void**)&t->tl_dz, (size_t)t->capT * t->H * 4));
    T_TRY(hipMalloc((void**)&t->tl_x, (size_t)t->capT * t->H * 4));
    return 0;
}

// dz (m x H fp32; its max |.| already in the Linear's scale word 3) -> scaled lines (t->dzl: the operand of dx = dz . W and,
// read reduction-major, of dW = dz^T . x); publishes both descales.  Comparison path (dw_trans == 0): also the transposed lines.
int fast_grad_lines(ml_trainer* t, hipStream_t st, const float* dz, int64_t m, int slot) {
    if (t->dw_trans) {
        hipLaunchKernelGGL(mlt::grad_lines_kernel, dim3(nblk(m * t->H / 8)), dim3(256), 0, st, dz, m, t->H, t->wsc_base + 8 * slot, t->dzl);
        return 0;
    }
    int rc = ensure_tl(t);
    if (rc) return rc;
    const int64_t mT = (m + 64 * t->ks - 1) / (64 * t->ks) * (64 * t->ks);
    hipLaunchKernelGGL(mlt::tlines_kernel<true>, dim3(t->H / 64, (unsigned)(mT / 64)), dim3(256), 0, st, dz, m, t->H, mT,
                       t->wsc_base + 8 * slot, t->tl_dz, t->dzl);
    return 0;
}

// dW (H x H) = dz^T . x on the 3-product kernel: both operands as transposed lines (the reduction runs over the batch),
// the batch split over t->ks work items per output tile, partials added in a fixed order.
int fast_linear_bwd_weight(ml_trainer* t, hipStream_t st, const float* x, const char* x_lines, const std::string& lin, int64_t m,
                           int slot) {
    const int H = t->H;
    const int64_t mT = (m + 64 * t->ks - 1) / (64 * t->ks) * (64 * t->ks);
    const bool trans = t->dw_trans && x_lines;
    if (!trans) {
        int rc = ensure_tl(t);
        if (rc) return rc;
        hipLaunchKernelGGL(mlt::tlines_kernel<false>, dim3(H / 64, (unsigned)(mT / 64)), dim3(256), 0, st, x, m, H, mT, (float*)nullptr,
                           t->tl_x, (char*)nullptr);
    }
    mlk::DenseParams p;
    p.x = trans ? t->dzl : t->tl_dz;      // [mT][H] lines of dz (rows >= m zero) | [H][mT] transposed lines
    p.w = trans ? x_lines : t->tl_x;      // the forward's lines of the layer input | its transposed copy
    p.bias = nullptr;
    p.bias_scaled = t->zero_bias;
    p.res = nullptr;
    p.y = (char*)t->d_splitk;
    p.descale = 1.0f;
    p.descale_ptr = t->wsc_base + 8 * slot + 5;
    p.M_pad = H;
    p.N = H;
    p.K = (int)(mT / t->ks);
    p.ksplit = t->ks;
    p.relu = 0;
    p.debug = 0;
    p.trace = nullptr;
    p.head_w = nullptr;
    p.head_part = nullptr;
    const int items = (H / mlk::BM) * (H / mlk::BN) * t->ks;
    const dim3 wg_grid(items < t->n_cu ? items : t->n_cu), wg_block(mlk::W4_THREADS);
    if (trans) hipLaunchKernelGGL((mlk::dense_kernel_w4<3, false, false, -3, true>), wg_grid, wg_block, 0, st, p);
    else hipLaunchKernelGGL((mlk::dense_kernel_w4<3, false, false, -3>), wg_grid, wg_block, 0, st, p);
    hipLaunchKernelGGL(mlt::splitk_reduce_kernel, dim3(nblk((int64_t)H * H)), dim3(256), 0, st, (const float*)t->d_splitk, t->ks, H, H,
                       H, (const float*)nullptr, 0, G(t, lin + ".weight"));
    if (hipGetLastError() != hipSuccess) return tfail(ML_ERR_HIP, "fast weight-gradient GEMM launch failed");
    return 0;
}

bool fast_possible(const ml_trainer* t) { return t->H % 256 == 0 && !t->lbufs.empty(); }
constexpr int64_t MID_AUTO_MAX_ROWS = 16384;
// xgemm_tile keeps its per-lane operand offsets in 32 bits (train_mid.h): an m x H fp32 operand must stay below 4 GiB
bool mid_rows_ok(const ml_trainer* t, int64_t m) { return (m + 64) * (int64_t)t->H * 4 < (int64_t)1 << 32; }
bool mid_possible(const ml_trainer* t) {
    return t->H % 64 == 0 && t->d_ssq != nullptr && t->in_f <= mlt::SK_NC && (int64_t)t->C * t->H <= mlt::HL_MAXW;
}
// 0 exact, 1 fast (large batches: 256 x 256-tile 3-product GEMMs on line-format operands), 2 mid (train_mid.h)
int pick_route(const ml_trainer* t, int64_t m) {
    switch (t->route) {
        case 1: return 0;
        case 2: return mid_possible(t) ? 2 : 0;
        case 3: return fast_possible(t) ? 1 : 0;
        default: break;
    }
    if (t->fast_rows > 0 && m >= t->fast_rows && fast_possible(t)) return 1;
    // mid: built for the reference's batch sizes (its column-owner kernels walk all rows in H / apply_cols workgroups); above
    // MID_AUTO_MAX_ROWS a shape the fast route cannot take goes to the exact route's row-parallel kernels instead
    if (mid_possible(t) && m <= MID_AUTO_MAX_ROWS) return 2;
    return 0;
}

// x_lines: the block input as lines (or null: exact-fp32 MFMA GEMM on b.x); y_lines: where to put the block output as
// lines as well (or null)
// lines_only: the output is written as lines only (nothing reads it as fp32); res_lines: the residual comes from its lines
int block_fwd(ml_trainer* t, hipStream_t st, Block& b, int64_t m, const float* residual, const char* x_lines = nullptr,
              char* y_lines = nullptr, int slot = -1, bool lines_only = false, const char* res_lines = nullptr) {
    const int H = t->H;
    int rc;
    const bool gemm_stats = x_lines && slot >= 0;   // the batch statistics come out of the GEMM's own epilogue
    if (gemm_stats) rc = fast_linear_fwd(t, st, x_lines, b.lin, b.z, m, slot, true);
    else if (b.in_dim != H && skinny_ok(t, b.in_dim))   // the (narrow) input layer
        rc = skinny_out(t, st, b.x, b.in_dim, b.in_dim, P(t, b.lin + ".weight"), 1, b.in_dim, P(t, b.lin + ".bias"), b.z, m, 0);
    else rc = linear_fwd(t, st, b.x, b.in_dim, P(t, b.lin + ".weight"), P(t, b.lin + ".bias"), b.z, H, (int)m, H, b.in_dim);
    if (rc) return rc;
    float* mean = t->bn_mean + (size_t)b.bn_idx * H;
    float* inv = t->bn_invstd + (size_t)b.bn_idx * H;
    if (gemm_stats) {
        hipLaunchKernelGGL(mlt::bn_finalize_parts_kernel, dim3((H + 7) / 8), dim3(256), 0, st, (const double*)t->d_colpart,
                           (int)((m + 127) / 128), m, H, 1e-5f, 0.1f, mean, inv, ST(t, b.bn + ".running_mean"),
                           ST(t, b.bn + ".running_var"));
    } else {
        if ((rc = col_stats(t, st, b.z, nullptr, m, H))) return rc;
        hipLaunchKernelGGL(mlt::bn_finalize_kernel, dim3(nblk(H)), dim3(256), 0, st, (const double*)t->d_red,
                           (const double*)(t->d_red + H), m, H, 1e-5f, 0.1f, mean, inv, ST(t, b.bn + ".running_mean"),
                           ST(t, b.bn + ".running_var"));
    }
    if (y_lines || (x_lines && (H & 7) == 0))   // (also the large-batch route's last block, no lines: 4 columns per lane instead of 1)
        hipLaunchKernelGGL(mlt::bn_relu_drop_lines_kernel, dim3(nblk(m * H / 4)), dim3(256), 0, st, (const float*)b.z, m, H,
                           (const float*)mean, (const float*)inv, (const float*)P(t, b.bn + ".weight"),
                           (const float*)P(t, b.bn + ".bias"), t->p_drop, t->seed + (uint32_t)(t->step + t->fwd_calls) * 977u, b.site,
                           res_lines ? (const float*)nullptr : residual, lines_only ? (float*)nullptr : b.y, y_lines, res_lines);
    else
        hipLaunchKernelGGL(mlt::bn_relu_drop_kernel, dim3(nblk(m * H)), dim3(256), 0, st, (const float*)b.z, m, H,
                           (const float*)mean, (const float*)inv, (const float*)P(t, b.bn + ".weight"),
                           (const float*)P(t, b.bn + ".bias"), t->p_drop, t->seed + (uint32_t)(t->step + t->fwd_calls) * 977u, b.site, residual, b.y);
    return 0;
}

// dout: gradient wrt the block output (m x H), overwritten with dz; xhat: scratch (m x H).  Produces the
// parameter gradients of the block; the caller propagates dz through the Linear to the block input.
// fresh pre-zeroed fp64 reduction slot (2 * H + 32 doubles): see col_stats
int next_red_slot(ml_trainer* t, hipStream_t) {
    if (t->red_slot + 1 >= t->red_slots) return tfail(ML_ERR_HIP, "out of reduction slots (internal)");
    t->d_red = t->d_red_base + (size_t)(++t->red_slot) * (2 * t->H + 32);
    return 0;
}

// slot >= 0: the Linear is an H x H one on the fast path: dz also goes to lbufs[0] / tl_dz as scaled lines, dW runs there;
// slot == -2: the (narrow) input layer
// din: where the incoming gradient lies when it is not dout itself (fused chain only: dz is written to dout, din stays intact)
int block_bwd(ml_trainer* t, hipStream_t st, Block& b, int64_t m, float* dout, float* xhat, int slot = -1, const float* din = nullptr) {
    const int H = t->H;
    float* mean = t->bn_mean + (size_t)b.bn_idx * H;
    float* inv = t->bn_invstd + (size_t)b.bn_idx * H;
    const uint32_t seed = t->seed + (uint32_t)(t->step + t->fwd_calls) * 977u;
    int rc;
    if ((H & 3) == 0) {
        // fused chain (train_kernels.h): pass 1 = sum(dy), sum(dy * xhat) with dy / xhat recomputed from (dout, z); pass 2 =
        // dz over dout + column sums of dz.  4 reads + 1 write of an (m, H) matrix instead of 7 + 3; bit-identical results.
        int gy = (int)((m + 255) / 256);
        if (gy > 128) gy = 128;
        if (gy < 1) gy = 1;
        if ((rc = next_red_slot(t, st))) return rc;
        double* s_dy = t->d_red;
        const float* src = din ? din : dout;
        // large-batch route, H x H Linear below (slot >= 0): dz leaves as scaled lines only (bn_bwd_lines_kernel), scaled by a bound
        // the statistics pass collects the column maxima for
        const bool lines_only = slot >= 0 && t->dw_trans && t->lines_chain && b.x_lines;
        float* colmax = lines_only ? t->d_colmax + (size_t)slot * 2 * H : nullptr;
        const bool headg = lines_only && b.hs && (b.nh == 8 || b.nh == 9);
#define ML_BWD_STATS(NHG) hipLaunchKernelGGL(mlt::bwd_stats_kernel<NHG>, dim3((H + 63) / 64, gy), dim3(256), 0, st, src, (const float*)b.z, m, H, \
                           (const float*)mean, (const float*)inv, (const float*)P(t, b.bn + ".weight"),                                   \
                           (const float*)P(t, b.bn + ".bias"), t->p_drop, seed, b.site, s_dy, s_dy + H, colmax, b.hs, b.ldh, b.hw)
        if (headg && b.nh == 8) ML_BWD_STATS(8);
        else if (headg) ML_BWD_STATS(9);
        else ML_BWD_STATS(0);
#undef ML_BWD_STATS
        if ((rc = next_red_slot(t, st))) return rc;
        double* s_dz = t->d_red;
#define ML_BWD_LINES(NHG) hipLaunchKernelGGL(mlt::bn_bwd_lines_kernel<NHG>, dim3((H + 63) / 64, gy), dim3(256), 0, st, src, (const float*)b.z, m, H, \
                               (const float*)mean, (const float*)inv, (const float*)P(t, b.bn + ".weight"),                                  \
                               (const float*)P(t, b.bn + ".bias"), t->p_drop, seed, b.site, (const double*)s_dy,                            \
                               (const double*)(s_dy + H), G(t, b.bn + ".weight"), G(t, b.bn + ".bias"), s_dz, (const float*)colmax,        \
                               t->wsc_base + 8 * slot, t->dzl, b.hs, b.ldh, b.hw)
        if (lines_only && headg && b.nh == 8) ML_BWD_LINES(8);
        else if (lines_only && headg) ML_BWD_LINES(9);
        else if (lines_only) ML_BWD_LINES(0);
#undef ML_BWD_LINES
        else
            hipLaunchKernelGGL(mlt::bn_bwd_fused_kernel, dim3((H + 63) / 64, gy), dim3(256), 0, st, dout, src, (const float*)b.z, m, H,
                               (const float*)mean, (const float*)inv, (const float*)P(t, b.bn + ".weight"),
                               (const float*)P(t, b.bn + ".bias"), t->p_drop, seed, b.site, (const double*)s_dy,
                               (const double*)(s_dy + H), G(t, b.bn + ".weight"), G(t, b.bn + ".bias"), s_dz,
                               slot >= 0 ? t->wsc_base + 8 * slot + 3 : (float*)nullptr);
        col_sum_to_float(t, st, (const double*)s_dz, H, G(t, b.lin + ".bias"));
        if (slot >= 0) {
            if (!lines_only && (rc = fast_grad_lines(t, st, dout, m, slot))) return rc;
            return fast_linear_bwd_weight(t, st, b.x, b.x_lines, b.lin, m, slot);
        }
        if (slot == -2 && skinny_ok(t, b.in_dim))   // the input layer: dW1 (H x in) = dz^T . x
            return skinny_dw(t, st, b.x, b.in_dim, b.in_dim, dout, m, G(t, b.lin + ".weight"), 1);
        return linear_bwd_weight(t, st, dout, H, b.x, b.in_dim, G(t, b.lin + ".weight"), (int)m, H, b.in_dim);
    }
    if (din && din != dout) T_TRY(hipMemcpyAsync(dout, din, (size_t)m * H * 4, hipMemcpyDeviceToDevice, st));   // (unfused chain: in place)
    hipLaunchKernelGGL(mlt::relu_drop_bwd_kernel, dim3(nblk(m * H)), dim3(256), 0, st, dout, (const float*)b.z, m, H,
                       (const float*)mean, (const float*)inv, (const float*)P(t, b.bn + ".weight"),
                       (const float*)P(t, b.bn + ".bias"), t->p_drop, seed, b.site, xhat);
    rc = col_stats(t, st, dout, xhat, m, H);  // sum(dy), sum(dy*xhat)
    if (rc) return rc;
    hipLaunchKernelGGL(mlt::bn_bwd_kernel, dim3(nblk(m * H)), dim3(256), 0, st, dout, (const float*)xhat, m, H,
                       (const double*)t->d_red, (const double*)(t->d_red + H), (const float*)P(t, b.bn + ".weight"),
                       (const float*)inv, G(t, b.bn + ".weight"), G(t, b.bn + ".bias"));
    // Linear: db = sum(dz), dW = dz^T x
    if ((rc = col_stats(t, st, dout, nullptr, m, H))) return rc;
    hipLaunchKernelGGL(mlt::col_sum_to_float_kernel, dim3(nblk(H)), dim3(256), 0, st, (const double*)t->d_red, H,
                       G(t, b.lin + ".bias"));
    return linear_bwd_weight(t, st, dout, H, b.x, b.in_dim, G(t, b.lin + ".weight"), (int)m, H, b.in_dim);
}

int ensure_cap(ml_trainer* t, int64_t m) {
    if (m <= t->cap) return 0;
    m = (m + 255) / 256 * 256;   // whole 256-row panels: the fast forward GEMM writes full tiles
    T_TRY(hipDeviceSynchronize());
    for (float* p : t->bufs) (void)hipFree(p);
    t->bufs.clear();
    for (char* p : t->lbufs) (void)hipFree(p);
    t->lbufs.clear();
    if (t->H % 256 == 0) {
        // weight gradient: (H / 256)^2 output tiles, the batch reduction split so that the work items about fill the chip
        // (at most 32: the partial buffer); its operands are zero-padded to whole 64-row k-steps per split: the line buffers
        // (the forward's activations and the gradient dz, read reduction-major) are allocated and zeroed up to capT rows
        const int otiles = (t->H / 256) * (t->H / 256);
        t->ks = 1;
        while (t->ks < 32 && otiles * t->ks * 2 <= 256) t->ks *= 2;
        t->capT = (m + 64 * t->ks - 1) / (64 * t->ks) * (64 * t->ks);
        for (int i = 0; i < 2 * t->S + 2; ++i) {
            char* p = nullptr;
            T_TRY(hipMalloc((void**)&p, (size_t)t->capT * t->H * 4));
            T_TRY(hipMemset(p, 0, (size_t)t->capT * t->H * 4));
            t->lbufs.push_back(p);
        }
        if (t->dzl) (void)hipFree(t->dzl);
        if (t->d_colpart) (void)hipFree(t->d_colpart);
        t->dzl = nullptr;
        t->d_colpart = nullptr;
        T_TRY(hipMalloc((void**)&t->d_colpart, (size_t)(m / 128) * 2 * t->H * sizeof(double)));
        T_TRY(hipMalloc((void**)&t->dzl, (size_t)t->capT * t->H * 4));
        T_TRY(hipMemset(t->dzl, 0, (size_t)t->capT * t->H * 4));
        t->pad_m = -1;
        if (t->tl_dz) (void)hipFree(t->tl_dz);
        if (t->tl_x) (void)hipFree(t->tl_x);
        t->tl_dz = t->tl_x = nullptr;   // (the transposed copies of the comparison path: allocated on first use, ensure_tl)
        if (t->wl.empty()) {
            hipDeviceProp_t prop;
            int dev = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                t->n_cu = prop.multiProcessorCount;
            T_TRY(hipMalloc((void**)&t->wsc_base, (size_t)(2 * t->S + 3) * 32));
            T_TRY(hipMalloc((void**)&t->d_colmax, (size_t)(2 * t->S + 3) * 2 * t->H * sizeof(float)));
            T_TRY(hipMalloc((void**)&t->d_w32, (size_t)t->H * t->H * 4));
            T_TRY(hipMalloc((void**)&t->d_m23, (size_t)t->H * t->H * 4));
            T_TRY(hipMalloc((void**)&t->d_b32, (size_t)t->H * 4));
            T_TRY(hipMalloc((void**)&t->d_u23, (size_t)t->H * 4));
            T_TRY(hipMalloc((void**)&t->d_v23, (size_t)t->H * 4));
            T_TRY(hipMalloc((void**)&t->d_c23, 64));
            T_TRY(hipMalloc((void**)&t->zero_bias, (size_t)t->H * 4));
            T_TRY(hipMemset(t->zero_bias, 0, (size_t)t->H * 4));
            for (int i = 0; i < 2 * t->S + 3; ++i) {   // (slot 2S + 2: the merged w3 . w2)
                char* w = nullptr;
                float* b = nullptr;
                T_TRY(hipMalloc((void**)&w, (size_t)t->H * t->H * 4));
                T_TRY(hipMalloc((void**)&b, (size_t)t->H * 4));
                t->wl.push_back(w);
                t->wbs.push_back(b);
                char* wt = nullptr;
                T_TRY(hipMalloc((void**)&wt, (size_t)t->H * t->H * 4));
                t->wlT.push_back(wt);
            }
            {   // the table the multi-layer pack kernels index by Linear slot (2s, 2s + 1 = stage s w1 / w2, 2S = w2, 2S + 1 = w3)
                std::vector<mlt::WLayer> desc;
                auto add = [&](const std::string& lin, int slot) {
                    mlt::WLayer l;
                    l.w = P(t, lin + ".weight"); l.bias = P(t, lin + ".bias"); l.sc = t->wsc_base + 8 * slot;
                    l.lines = t->wl[slot]; l.linesT = t->wlT[slot]; l.bias_scaled = t->wbs[slot];
                    desc.push_back(l);
                };
                for (int s2 = 0; s2 < t->S; ++s2) {
                    add("linear_stages." + std::to_string(s2) + ".w1", 2 * s2);
                    add("linear_stages." + std::to_string(s2) + ".w2", 2 * s2 + 1);
                }
                add("w2", 2 * t->S);
                add("w3", 2 * t->S + 1);
                {
                    const int slot = 2 * t->S + 2;
                    mlt::WLayer l;
                    l.w = t->d_w32; l.bias = t->d_b32; l.sc = t->wsc_base + 8 * slot;
                    l.lines = t->wl[slot]; l.linesT = t->wlT[slot]; l.bias_scaled = t->wbs[slot];
                    desc.push_back(l);
                }
                T_TRY(hipMalloc((void**)&t->d_wdesc, desc.size() * sizeof(mlt::WLayer)));
                T_TRY(hipMemcpy(t->d_wdesc, desc.data(), desc.size() * sizeof(mlt::WLayer), hipMemcpyHostToDevice));
                desc.erase(desc.begin() + 2 * t->S, desc.begin() + 2 * t->S + 2);   // (w2, w3 themselves are not multiplied with anything batch-sized)
                T_TRY(hipMalloc((void**)&t->d_wdesc_m, desc.size() * sizeof(mlt::WLayer)));
                T_TRY(hipMemcpy(t->d_wdesc_m, desc.data(), desc.size() * sizeof(mlt::WLayer), hipMemcpyHostToDevice));
            }
        }
    }
    if (t->d_out) (void)hipFree(t->d_out);
    if (t->d_dout) (void)hipFree(t->d_dout);
    const int nb = 4 * t->S + 10;  // a_s (S+1), t_s (S), z (2S+2), y2, y3, xhat, 2 gradient buffers (+ 2: the mid route's rotating dz)
    for (int i = 0; i < nb; ++i) {
        float* p = nullptr;
        T_TRY(hipMalloc((void**)&p, (size_t)m * t->H * 4));
        t->bufs.push_back(p);
    }
    T_TRY(hipMalloc((void**)&t->d_out, (size_t)m * t->C * 4));
    T_TRY(hipMalloc((void**)&t->d_dout, (size_t)m * t->C * 4));
    t->cap = m;
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// helpers shared by the routes: task weights of the loss, the host side of a finished step
int upload_task_weights(ml_trainer* t, hipStream_t st, bool task_weights) {
    if (!task_weights) return 0;
    for (int i = 0; i < 8; ++i) {
        const float e = std::exp(t->log_sigma[i]);
        // lam * l (losses.py:66), or lam * l / (2.0 * (log_sigma.exp() ** 2)) (losses.py:34)
        t->h_tw[i] = t->auto_tune ? t->lambdas[i] / (2.0f * (e * e)) : t->lambdas[i];
    }
    if (!t->d_tw) T_TRY(hipMalloc((void**)&t->d_tw, 8 * sizeof(float)));
    T_TRY(hipMemcpyAsync(t->d_tw, t->h_tw, 8 * sizeof(float), hipMemcpyHostToDevice, st));
    return 0;
}

// lv: the 8 task means of this step (already on the host).  Fills losses_host, runs the log_sigma Adam of the auto-tuned loss.
void finish_step_host(ml_trainer* t, const double* lv, bool task_weights, int update, float lr, float bc1, float bc2,
                      double* losses_host) {
    const int nt = (t->C == 10) ? 8 : 7;
    for (int i = 0; i < mlt::LOSS_NV; ++i) t->last_vals[i] = lv[i];
    if (losses_host) {
        double tot = 0;
        for (int i = 0; i < 8; ++i) {
            // auto-tune: the training-phase values are the weighted ones, the total adds the log_sigmas (losses.py:34-39)
            const double v = (task_weights && i < nt) ? (double)((float)lv[i] * t->h_tw[i]) : lv[i];
            losses_host[1 + i] = v;
            if (i < nt) tot += v + (t->auto_tune ? (double)t->log_sigma[i] : 0.0);
        }
        losses_host[0] = tot;
    }
    if (t->auto_tune && update) {   // Adam on the log_sigmas with this step's task losses (fp32 like torch, no clipping)
        for (int i = 0; i < nt; ++i) {
            const float g = 1.0f - 2.0f * t->h_tw[i] * (float)lv[i];   // d/ds [ l / (2 exp(2 s)) + s ]
            t->ls_m1[i] = 0.9f * t->ls_m1[i] + 0.1f * g;
            t->ls_m2[i] = 0.999f * t->ls_m2[i] + 0.001f * g * g;
            const float denom = std::sqrt(t->ls_m2[i]) / std::sqrt(bc2) + 1e-8f;
            t->log_sigma[i] -= (lr / bc1) * (t->ls_m1[i] / denom);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The mid route (train_mid.h): one training step in ~57 launches, every GEMM on the exact fp32 matrix instruction.
int launch_xgemm(hipStream_t st, const float* a, long lda, int alay, const float* b, long ldb, int blay, float* c, long ldc, int M,
                 int N, int K, const float* bias, const float* res, double* sumsq) {
    mlt::XGemmParams p;
    p.a = a; p.b = b; p.c = c; p.res = res; p.bias = bias; p.sumsq = sumsq;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    const dim3 grid(N / mlt::XG_BN, (M + mlt::XG_BM - 1) / mlt::XG_BM), blk(256);
    if (alay == 0 && blay == 0) hipLaunchKernelGGL((mlt::xgemm_kernel<0, 0>), grid, blk, 0, st, p);
    else if (alay == 0 && blay == 1) hipLaunchKernelGGL((mlt::xgemm_kernel<0, 1>), grid, blk, 0, st, p);
    else if (alay == 1 && blay == 1) hipLaunchKernelGGL((mlt::xgemm_kernel<1, 1>), grid, blk, 0, st, p);
    else hipLaunchKernelGGL((mlt::xgemm_kernel<1, 0>), grid, blk, 0, st, p);
    if (hipGetLastError() != hipSuccess) return tfail(ML_ERR_HIP, "xgemm launch failed");
    return 0;
}

// dx = dz . W (+ res) and dW = dz^T . x in one launch (xgemm_pair_kernel)
int launch_xgemm_pair(hipStream_t st, const float* dz, const float* w, float* dx, const float* res, const float* x, float* dw,
                      double* sumsq, int m, int H) {
    mlt::XGemmParams pd, pw;
    pd.a = dz; pd.b = w; pd.c = dx; pd.res = res; pd.bias = nullptr; pd.sumsq = nullptr;
    pd.lda = H; pd.ldb = H; pd.ldc = H; pd.M = m; pd.N = H; pd.K = H;
    pw.a = dz; pw.b = x; pw.c = dw; pw.res = nullptr; pw.bias = nullptr; pw.sumsq = sumsq;
    pw.lda = H; pw.ldb = H; pw.ldc = H; pw.M = H; pw.N = H; pw.K = m;
    const int ndx = H / mlt::XG_BN, nd = ndx * ((m + mlt::XG_BM - 1) / mlt::XG_BM);
    const int nwx = H / mlt::XG_BN, nw = nwx * (H / mlt::XG_BM);
    hipLaunchKernelGGL(mlt::xgemm_pair_kernel, dim3(nd + nw), dim3(256), 0, st, pd, pw, nd, ndx, nwx);
    if (hipGetLastError() != hipSuccess) return tfail(ML_ERR_HIP, "xgemm pair launch failed");
    return 0;
}

template <typename P>
void launch_fwd_apply(const ml_trainer* t, hipStream_t st, const P& p) {
    const dim3 blk(256);
    if (t->apply_cols == 4) hipLaunchKernelGGL(mlt::fwd_apply_kernel<4>, dim3(t->H / 4), blk, 0, st, p);
    else if (t->apply_cols == 8) hipLaunchKernelGGL(mlt::fwd_apply_kernel<8>, dim3(t->H / 8), blk, 0, st, p);
    else hipLaunchKernelGGL(mlt::fwd_apply_kernel<16>, dim3(t->H / 16), blk, 0, st, p);
}
template <typename P>
void launch_bwd_apply(const ml_trainer* t, hipStream_t st, const P& p) {
    const dim3 blk(256);
    if (t->apply_cols == 4) hipLaunchKernelGGL(mlt::bwd_apply_kernel<4>, dim3(t->H / 4), blk, 0, st, p);
    else if (t->apply_cols == 8) hipLaunchKernelGGL(mlt::bwd_apply_kernel<8>, dim3(t->H / 8), blk, 0, st, p);
    else hipLaunchKernelGGL(mlt::bwd_apply_kernel<16>, dim3(t->H / 16), blk, 0, st, p);
}

void sum_loss_parts(const ml_trainer* t, int parts, double* lv) {   // the per-workgroup partial sums, in order
    for (int q = 0; q < mlt::LOSS_NV; ++q) {
        double a = 0.0;
        for (int b = 0; b < parts; ++b) a += t->h_loss[b * mlt::LOSS_NV + q];
        lv[q] = a;
    }
}

// The mid route's forward + heads + loss values (+ loss gradient when training) on the trainer's buffers.  eval: BatchNorm with the
// running statistics, no dropout, no gradient (the reference's model.eval() forward, trainer.py:167-178).
int mid_forward(ml_trainer* t, hipStream_t st, const float* x_dev, const float* labels_dev, int label_cols, int64_t m, uint32_t seed,
                bool eval, bool task_weights, int* hl_grid_out) {
    const int H = t->H, S = t->S, C = t->C;
    int rc;
    int bi = 0;
    auto nb = [&]() { return t->bufs[bi++]; };
    std::vector<float*> a(S + 1), tt(S), za(S), zb(S);
    for (auto& p : a) p = nb();
    for (auto& p : tt) p = nb();
    float* z0 = nb();
    for (int s = 0; s < S; ++s) { za[s] = nb(); zb[s] = nb(); }
    float* z3 = nb();
    float* y2 = nb();
    float* y3 = nb();
    auto mean_of = [&](int bn_idx) { return t->bn_mean + (size_t)bn_idx * H; };
    auto inv_of = [&](int bn_idx) { return t->bn_invstd + (size_t)bn_idx * H; };
    auto fwd_apply = [&](float* z, const std::string& bn, int bn_idx, uint32_t site, const float* residual, float* y, bool input_layer) {
        mlt::FwdApplyParams p;
        p.z = input_layer ? nullptr : z;
        p.x_in = input_layer ? x_dev : nullptr;
        p.w_in = input_layer ? P(t, "w1.weight") : nullptr;
        p.b_in = input_layer ? P(t, "w1.bias") : nullptr;
        p.z_out = input_layer ? z : nullptr;
        p.in_dim = t->in_f;
        p.m = (long)m; p.H = H;
        p.gamma = P(t, bn + ".weight"); p.beta = P(t, bn + ".bias");
        p.run_mean = ST(t, bn + ".running_mean"); p.run_var = ST(t, bn + ".running_var");
        p.mean_out = mean_of(bn_idx); p.invstd_out = inv_of(bn_idx);
        p.p_drop = t->p_drop; p.seed = seed; p.site = site;
        p.residual = residual; p.y = y;
        p.eval = eval ? 1 : 0;
        launch_fwd_apply(t, st, p);
    };
    // z (m x H) = x . W^T + b: both operands k-contiguous
    auto lin_fwd = [&](const float* x, const std::string& lin, float* z) {
        return launch_xgemm(st, x, H, 0, P(t, lin + ".weight"), H, 0, z, H, (int)m, H, H, P(t, lin + ".bias"), nullptr, nullptr);
    };
    fwd_apply(z0, "batch_norm1", 0, 0, nullptr, a[0], true);
    for (int s = 0; s < S; ++s) {
        const std::string p = "linear_stages." + std::to_string(s) + ".";
        if ((rc = lin_fwd(a[s], p + "w1", za[s]))) return rc;
        fwd_apply(za[s], p + "batch_norm1", 1 + 2 * s, 1 + 2 * s, nullptr, tt[s], false);
        if ((rc = lin_fwd(tt[s], p + "w2", zb[s]))) return rc;
        fwd_apply(zb[s], p + "batch_norm2", 2 + 2 * s, 2 + 2 * s, a[s], a[s + 1], false);   // a_{s+1} = a_s + block(t_s)
    }
    if ((rc = lin_fwd(a[S], "w2", y2))) return rc;
    if ((rc = lin_fwd(y2, "w3", z3))) return rc;
    fwd_apply(z3, "batch_norm3", 2 * S + 1, 2 * S + 1, nullptr, y3, false);
    // ---------------- both heads, the loss (and its gradient): one launch, per-workgroup partial sums the host adds
    int hl_grid = (int)((m + 3) / 4);
    if (hl_grid > HL_GRID) hl_grid = HL_GRID;
    {
        mlt::HeadsLossParams p;
        p.y3 = y3; p.y2 = y2;
        p.w_fin = P(t, "w_fin.weight"); p.b_fin = P(t, "w_fin.bias");
        p.w_aux = P(t, "w_aux.weight"); p.b_aux = P(t, "w_aux.bias");
        p.lab = labels_dev; p.L = label_cols; p.m = (long)m; p.H = H;
        p.out = t->d_out; p.dout = eval ? nullptr : t->d_dout;
        p.tw = task_weights ? t->d_tw : nullptr;
        // the per-workgroup partial sums go straight into PINNED host memory (a kernel may write device-mapped host memory; the
        // step's stream synchronisation makes them visible): no device-to-host copy operation behind the kernel
        p.part = t->h_loss_dev ? t->h_loss_dev : t->d_lpart;
        if (C == 10) hipLaunchKernelGGL(mlt::heads_loss_kernel<10>, dim3(hl_grid), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(mlt::heads_loss_kernel<9>, dim3(hl_grid), dim3(256), 0, st, p);
    }
    *hl_grid_out = hl_grid;
    return 0;
}

int step_mid(ml_trainer* t, const float* x_dev, const float* labels_dev, int label_cols, int64_t m, int update,
             double* losses_host, float* raw_out_dev, hipStream_t st) {
    const int H = t->H, S = t->S, C = t->C;
    int rc;
    // buffer plan (the exact route's)
    int bi = 0;
    auto nb = [&]() { return t->bufs[bi++]; };
    std::vector<float*> a(S + 1), tt(S), za(S), zb(S);
    for (auto& p : a) p = nb();
    for (auto& p : tt) p = nb();
    float* z0 = nb();
    for (int s = 0; s < S; ++s) { za[s] = nb(); zb[s] = nb(); }
    float* z3 = nb();
    float* y2 = nb();
    float* y3 = nb();
    float* gE = nb();   // (the exact route's xhat scratch)
    float* gA = nb();
    float* gBs[3] = {nb(), nb(), nb()};   // dz of consecutive Linears rotate through three buffers: the weight gradient of Linear k
                                          // (side stream) may still read its dz while the main stream writes the next two
    const bool side = t->side_stream && t->st2;
    hipStream_t sw = side ? t->st2 : st;
    int nlin = 0;                         // Linears whose backward has started (index into the events / the rotation)
    const uint32_t seed = t->seed + (uint32_t)(t->step + t->fwd_calls) * 977u;
    auto mean_of = [&](int bn_idx) { return t->bn_mean + (size_t)bn_idx * H; };
    auto inv_of = [&](int bn_idx) { return t->bn_invstd + (size_t)bn_idx * H; };

    // ---------------- forward (train mode), both heads, the loss and its gradient
    const bool task_weights = t->auto_tune || t->weighted;
    if ((rc = upload_task_weights(t, st, task_weights))) return rc;
    int hl_grid = 0;
    if ((rc = mid_forward(t, st, x_dev, labels_dev, label_cols, m, seed, false, task_weights, &hl_grid))) return rc;
    if (raw_out_dev) T_TRY(hipMemcpyAsync(raw_out_dev, t->d_out, (size_t)m * C * 4, hipMemcpyDeviceToDevice, st));
    if (!t->h_loss_dev)
        T_TRY(hipMemcpyAsync(t->h_loss, t->d_lpart, (size_t)hl_grid * mlt::LOSS_NV * sizeof(double), hipMemcpyDeviceToHost, st));   // pinned
    // ---------------- backward (every gradient tensor is written in full: no memset of g).  The narrow gradients (head weights
    // and biases, input-layer weights) ride in the column-owner kernels of the blocks whose data they read.
    // dz of one block from its incoming gradient
    // extra: 0 none, 1 = block 3 (w_fin weight gradient from y3 + both head biases), 2 = w2 (w_aux weight gradient from y2)
    auto bwd_apply = [&](const float* dy, bool from_heads, bool aux, const float* z, const std::string& bn, int bn_idx, uint32_t site,
                         const std::string& lin, float* dz, int extra) {
        mlt::BwdApplyParams p;
        p.dy = from_heads ? nullptr : dy;
        p.dout = from_heads ? t->d_dout : nullptr;
        p.dld = C; p.nc = C - 1;
        p.w_head = from_heads ? P(t, "w_fin.weight") : nullptr;
        p.aux_d = aux ? t->d_dout + (C - 1) : nullptr;
        p.aux_ld = C;
        p.w_aux = aux ? P(t, "w_aux.weight") : nullptr;
        p.z = z;
        p.mean = z ? mean_of(bn_idx) : nullptr;
        p.invstd = z ? inv_of(bn_idx) : nullptr;
        p.gamma = z ? P(t, bn + ".weight") : nullptr;
        p.beta = z ? P(t, bn + ".bias") : nullptr;
        p.p_drop = t->p_drop; p.seed = seed; p.site = site;
        p.m = (long)m; p.H = H;
        p.dz = dz;
        p.dgamma = z ? G(t, bn + ".weight") : nullptr;
        p.dbeta = z ? G(t, bn + ".bias") : nullptr;
        p.dbias = G(t, lin + ".bias");
        p.sc = nullptr; p.sld = C; p.ns = 0; p.ysrc = nullptr; p.dwh = nullptr;
        p.hb_src = nullptr; p.hb_ld = C; p.hb_n0 = C - 1; p.hb0 = nullptr; p.hb1 = nullptr;
        if (extra == 1) {
            p.sc = t->d_dout; p.ns = C - 1; p.ysrc = y3; p.dwh = G(t, "w_fin.weight");
            p.hb_src = t->d_dout; p.hb0 = G(t, "w_fin.bias"); p.hb1 = G(t, "w_aux.bias");
        } else if (extra == 2) {
            p.sc = t->d_dout + (C - 1); p.ns = 1; p.ysrc = y2; p.dwh = G(t, "w_aux.weight");
        }
        launch_bwd_apply(t, st, p);
    };
    // the dz buffer of the next Linear; before it is overwritten the weight gradient that read it three Linears ago must be done
    auto next_dz = [&]() -> float* {
        if (side && nlin >= 3) (void)hipStreamWaitEvent(st, t->ev_w[nlin - 3], 0);
        return gBs[nlin % 3];
    };
    // dW (H x H) = dz^T . x: both operands reduction-major (the batch is the reduction); leaves its sum of squares.  On the side
    // stream, behind the event that says dz is complete.
    auto wgrad = [&](const float* dz, const float* x, const std::string& lin, int slot) {
        if (side) {
            (void)hipEventRecord(t->ev_dz[nlin], st);
            (void)hipStreamWaitEvent(sw, t->ev_dz[nlin], 0);
        }
        const int r = launch_xgemm(sw, dz, H, 1, x, H, 1, G(t, lin + ".weight"), H, H, H, (int)m, nullptr, nullptr,
                                   t->d_ssq + (size_t)slot * t->ssq_per_mat);
        if (side) (void)hipEventRecord(t->ev_w[nlin], sw);
        ++nlin;
        return r;
    };
    // dx (m x H) = dz . W (+ res): dz k-contiguous, W reduction-major as it lies (W[n][k]: n is the reduction)
    auto dgrad = [&](const float* dz, const std::string& lin, float* dx, const float* res) {
        return launch_xgemm(st, dz, H, 0, P(t, lin + ".weight"), H, 1, dx, H, (int)m, H, H, nullptr, res, nullptr);
    };
    // both gradients of one Linear: one launch on the main stream (xgemm_pair_kernel), or the two of them with the weight
    // gradient on the side stream
    auto grads = [&](const float* dz, const float* x, const std::string& lin, int slot, float* dx, const float* res) {
        if (!side && t->pair_gemm) {
            ++nlin;
            return launch_xgemm_pair(st, dz, P(t, lin + ".weight"), dx, res, x, G(t, lin + ".weight"),
                                     t->d_ssq + (size_t)slot * t->ssq_per_mat, (int)m, H);
        }
        const int r = wgrad(dz, x, lin, slot);
        return r ? r : dgrad(dz, lin, dx, res);
    };
    // Linear `slot`s: 2s = stage s w1, 2s + 1 = stage s w2, 2S = w2, 2S + 1 = w3
    float* gB = next_dz();
    bwd_apply(nullptr, true, false, z3, "batch_norm3", 2 * S + 1, 2 * S + 1, "w3", gB, 1);                        // gB = dz3
    if ((rc = grads(gB, y2, "w3", 2 * S + 1, gA, nullptr))) return rc;                                           // gA = dy2 (w3 part)
    gB = next_dz();
    bwd_apply(gA, false, true, nullptr, "", 0, 0, "w2", gB, 2);                                                  // gB = dz2 = dy2 + daux (x) w_aux
    if ((rc = grads(gB, a[S], "w2", 2 * S, gA, nullptr))) return rc;                                             // gA = da_S
    for (int s = S - 1; s >= 0; --s) {   // a_{s+1} = a_s + B(A(a_s))
        const std::string p = "linear_stages." + std::to_string(s) + ".";
        gB = next_dz();
        bwd_apply(gA, false, false, zb[s], p + "batch_norm2", 2 + 2 * s, 2 + 2 * s, p + "w2", gB, 0);              // gB = dz_b
        if ((rc = grads(gB, tt[s], p + "w2", 2 * s + 1, gE, nullptr))) return rc;                                 // gE = d t_s
        gB = next_dz();
        bwd_apply(gE, false, false, za[s], p + "batch_norm1", 1 + 2 * s, 1 + 2 * s, p + "w1", gB, 0);              // gB = dz_a
        if ((rc = grads(gB, a[s], p + "w1", 2 * s, gA, gA))) return rc;                                           // gA = da_s = da_{s+1} + dz_a . W
    }
    gB = next_dz();
    bwd_apply(gA, false, false, z0, "batch_norm1", 0, 0, "w1", gB, 0);                                            // gB = dz0
    {   // the input layer's weight gradient, also beside the main stream (it is the last thing before the optimizer)
        if (side) {
            (void)hipEventRecord(t->ev_dz[nlin], st);
            (void)hipStreamWaitEvent(sw, t->ev_dz[nlin], 0);
        }
        if ((rc = skinny_dw(t, sw, x_dev, t->in_f, t->in_f, gB, m, G(t, "w1.weight"), 1))) return rc;
        if (side) {   // the optimizer needs every gradient: the main stream joins the side stream here
            (void)hipEventRecord(t->ev_w[nlin], sw);
            (void)hipStreamWaitEvent(st, t->ev_w[nlin], 0);
        }
        ++nlin;
    }
    // ---------------- clip (always) + Adam + StepLR (per batch, only when updating)
    const int64_t k = t->step + 1;
    const float lr = t->lr0 * std::pow(t->gamma, (float)(t->step / t->sched_step));
    const float bc1 = 1.f - std::pow(0.9f, (float)k), bc2 = 1.f - std::pow(0.999f, (float)k);
    {
        hipLaunchKernelGGL(mlt::gradnorm_kernel, dim3(mlt::GN_PARTS), dim3(256), 0, st, (const float*)t->g, t->segs,
                           (const double*)t->d_ssq, (int)(t->mat_off.size() * t->ssq_per_mat), t->d_gn);
        hipLaunchKernelGGL(mlt::clip_adam_parts_kernel, dim3(nblk(t->n_param)), dim3(256), 0, st, t->w, t->g, t->m1, t->m2, t->n_param,
                           (const double*)t->d_gn, 3.0f, lr, 0.9f, 0.999f, 1e-8f, bc1, bc2, update ? 1 : 0);
        if (hipGetLastError() != hipSuccess) return tfail(ML_ERR_HIP, "optimizer launch failed");
        if (update) t->step++;
    }
    T_TRY(hipStreamSynchronize(st));
    double lv[mlt::LOSS_NV];
    sum_loss_parts(t, hl_grid, lv);
    finish_step_host(t, lv, task_weights, update, lr, bc1, bc2, losses_host);
    return ML_OK;
}

}  // namespace

extern "C" {

int ml_trainer_create(int in_features, int hidden, int out_features, int num_stage, float p_dropout, float lr,
                      float sched_gamma, int sched_step, uint32_t seed, ml_trainer** out) {
    if (!out) return tfail(ML_ERR_ARG, "out is null");
    if (in_features <= 0 || hidden <= 0 || hidden > 8192 || (out_features != 9 && out_features != 10) || num_stage < 0 ||
        num_stage > 16 || !(p_dropout >= 0.f && p_dropout < 1.f) || sched_step <= 0)
        return tfail(ML_ERR_SHAPE, "unsupported trainer shape / hyper-parameters");
    ml_trainer* t = new (std::nothrow) ml_trainer();
    if (!t) return tfail(ML_ERR_HIP, "out of host memory");
    t->in_f = in_features; t->H = hidden; t->C = out_features; t->S = num_stage;
    t->p_drop = p_dropout; t->lr0 = lr; t->gamma = sched_gamma; t->sched_step = sched_step; t->seed = seed;
    if (const char* e = getenv("MONOLOCO_TRAIN_HEAD_IN_BWD")) t->head_in_bwd = atoi(e) ? 1 : 0;   // (the A/B switch of round 6's change)
    add_linear(t, "w1", hidden, in_features);
    add_bn(t, "batch_norm1", hidden);
    for (int s = 0; s < num_stage; ++s) {
        const std::string p = "linear_stages." + std::to_string(s) + ".";
        add_linear(t, p + "w1", hidden, hidden);
        add_bn(t, p + "batch_norm1", hidden);
        add_linear(t, p + "w2", hidden, hidden);
        add_bn(t, p + "batch_norm2", hidden);
    }
    add_linear(t, "w2", hidden, hidden);
    add_linear(t, "w3", hidden, hidden);
    add_bn(t, "batch_norm3", hidden);
    add_linear(t, "w_aux", 1, hidden);
    add_linear(t, "w_fin", out_features - 1, hidden);
    t->nbn = 2 * num_stage + 2;
    T_TRY(hipMalloc((void**)&t->w, (size_t)t->n_param * 4));
    T_TRY(hipMalloc((void**)&t->g, (size_t)t->n_param * 4));
    T_TRY(hipMalloc((void**)&t->m1, (size_t)t->n_param * 4));
    T_TRY(hipMalloc((void**)&t->m2, (size_t)t->n_param * 4));
    T_TRY(hipMalloc((void**)&t->stat, (size_t)t->n_stat * 4));
    T_TRY(hipMalloc((void**)&t->bn_mean, (size_t)t->nbn * hidden * 4));
    T_TRY(hipMalloc((void**)&t->bn_invstd, (size_t)t->nbn * hidden * 4));
    t->red_slots = red_slots_for(num_stage);
    T_TRY(hipMalloc((void**)&t->d_red_base, (size_t)t->red_slots * (2 * hidden + 32) * sizeof(double)));
    T_TRY(hipHostMalloc((void**)&t->h_loss, HL_GRID * mlt::LOSS_NV * sizeof(double), hipHostMallocDefault));
    if (hipHostGetDevicePointer((void**)&t->h_loss_dev, t->h_loss, 0) != hipSuccess) {
        (void)hipGetLastError();
        t->h_loss_dev = nullptr;
    }
    T_TRY(hipMalloc((void**)&t->d_lpart, HL_GRID * mlt::LOSS_NV * sizeof(double)));
    {   // flat offsets of the H x H matrices by Linear slot (2s, 2s + 1 = stage s w1 / w2, 2S = w2, 2S + 1 = w3) and the
        // segments between them (everything else), for the mid route's gradient norm
        std::vector<std::string> names;
        for (int s = 0; s < num_stage; ++s) {
            names.push_back("linear_stages." + std::to_string(s) + ".w1.weight");
            names.push_back("linear_stages." + std::to_string(s) + ".w2.weight");
        }
        names.push_back("w2.weight");
        names.push_back("w3.weight");
        std::vector<int64_t> offs;
        for (size_t i = 0; i < names.size(); ++i) {
            t->mat_off.push_back(t->slots[names[i]].off);
            offs.push_back(t->slots[names[i]].off);
        }
        std::sort(offs.begin(), offs.end());
        const int64_t hh = (int64_t)hidden * hidden;
        int64_t pos = 0, total = 0;
        t->segs.count = 0;
        auto add_seg = [&](int64_t from, int64_t to) {
            if (to <= from) return;
            t->segs.off[t->segs.count] = from;
            t->segs.start[t->segs.count] = total;
            total += to - from;
            ++t->segs.count;
        };
        for (int64_t o : offs) {
            add_seg(pos, o);
            pos = o + hh;
        }
        add_seg(pos, t->n_param);
        t->segs.start[t->segs.count] = total;
    }
    if (hidden % 64 == 0) {   // one partial sum of squares per workgroup of a weight-gradient GEMM
        t->ssq_per_mat = (hidden / mlt::XG_BN) * (hidden / mlt::XG_BM);
        T_TRY(hipMalloc((void**)&t->d_ssq, (size_t)t->mat_off.size() * t->ssq_per_mat * sizeof(double)));
        T_TRY(hipMalloc((void**)&t->d_gn, mlt::GN_PARTS * sizeof(double)));
        T_TRY(hipStreamCreateWithFlags(&t->st2, hipStreamNonBlocking));
        for (int i = 0; i < 2 * num_stage + 3; ++i) {
            hipEvent_t e0, e1;
            T_TRY(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
            T_TRY(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
            t->ev_dz.push_back(e0);
            t->ev_w.push_back(e1);
        }
    }
    t->splitk_cap = (size_t)32 * hidden * (hidden > in_features ? hidden : in_features);
    T_TRY(hipMalloc((void**)&t->d_splitk, t->splitk_cap * 4));
    T_TRY(hipMemset(t->w, 0, (size_t)t->n_param * 4));
    T_TRY(hipMemset(t->g, 0, (size_t)t->n_param * 4));
    T_TRY(hipMemset(t->m1, 0, (size_t)t->n_param * 4));
    T_TRY(hipMemset(t->m2, 0, (size_t)t->n_param * 4));
    T_TRY(hipMemset(t->stat, 0, (size_t)t->n_stat * 4));
    *out = t;
    return ML_OK;
}

int ml_trainer_destroy(ml_trainer* t) {
    if (!t) return ML_OK;
    (void)hipDeviceSynchronize();
    for (float* p : t->bufs) (void)hipFree(p);
    for (char* p : t->lbufs) (void)hipFree(p);
    for (char* p : t->wl) (void)hipFree(p);
    for (char* p : t->wlT) (void)hipFree(p);
    if (t->d_wdesc) (void)hipFree(t->d_wdesc);
    if (t->d_wdesc_m) (void)hipFree(t->d_wdesc_m);
    for (float* p : t->wbs) (void)hipFree(p);
    if (t->h_loss) (void)hipHostFree(t->h_loss);
    for (hipEvent_t e : t->ev_dz) (void)hipEventDestroy(e);
    for (hipEvent_t e : t->ev_w) (void)hipEventDestroy(e);
    if (t->st2) (void)hipStreamDestroy(t->st2);
    void* ptrs[] = {t->w_snap, t->d_lpart, t->d_ssq, t->d_gn, t->d_tw, t->tl_dz, t->tl_x, t->dzl, t->d_colpart, t->d_colmax, t->wsc_base, t->zero_bias, t->w, t->g, t->m1, t->m2, t->stat, t->d_out, t->d_dout, t->bn_mean, t->bn_invstd, t->d_red_base, t->d_splitk, t->d_w32, t->d_b32, t->d_u23, t->d_c23, t->d_v23, t->d_m23};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete t;
    return ML_OK;
}

static int xfer(ml_trainer* t, const char* key, float* host, const float* chost, int64_t numel, int what) {
    if (!t || !key) return tfail(ML_ERR_ARG, "null argument");
    const std::string k(key);
    const std::string nbt = "num_batches_tracked";
    if (k.size() >= nbt.size() && k.compare(k.size() - nbt.size(), nbt.size(), nbt) == 0) return ML_OK;
    auto it = t->slots.find(k);
    if (it == t->slots.end()) return tfail(ML_ERR_ARG, "unknown tensor '%s'", key);
    if (it->second.numel != numel) return tfail(ML_ERR_ARG, "tensor '%s' has %lld elements, expected %lld", key,
                                                (long long)numel, (long long)it->second.numel);
    float* base = what == 2 ? t->g : (it->second.is_param ? t->w : t->stat);
    if (what == 2 && !it->second.is_param) return tfail(ML_ERR_ARG, "'%s' is a buffer, it has no gradient", key);
    T_TRY(hipDeviceSynchronize());
    // (hipMemcpyDefault: the caller's pointer may be host OR device memory -- the autograd module hands its CUDA parameters and
    //  gradient buffers over without a host round trip)
    if (what == 0) {
        T_TRY(hipMemcpy(base + it->second.off, chost, (size_t)numel * 4, hipMemcpyDefault));
    } else T_TRY(hipMemcpy(host, base + it->second.off, (size_t)numel * 4, hipMemcpyDefault));
    return ML_OK;
}
int ml_trainer_set_tensor(ml_trainer* t, const char* key, const float* host_data, int64_t numel) {
    return xfer(t, key, nullptr, host_data, numel, 0);
}
int ml_trainer_get_tensor(ml_trainer* t, const char* key, float* host_data, int64_t numel) {
    return xfer(t, key, host_data, nullptr, numel, 1);
}
// The flat buffers behind the keyed access: parameters (and their gradients) lie in ONE allocation in slot order, the BatchNorm
// running statistics in another.  ml_trainer_flat_offset: where `key` starts in its buffer (elements; -1: unknown) and which buffer
// (*is_param); ml_trainer_flat_numel: the two sizes; ml_trainer_copy_flat: one stream-ordered copy of a whole buffer, device to device --
// what 0: parameters <- dev_ptr, 1: parameters -> dev_ptr, 2: gradients -> dev_ptr, 3: statistics <- dev_ptr, 4: statistics -> dev_ptr.
// The autograd-capable module moves its 62 tensors with two copies per iteration this way instead of 62 synchronising ones.
int64_t ml_trainer_flat_offset(const ml_trainer* t, const char* key, int* is_param) {
    if (!t || !key) return -1;
    auto it = t->slots.find(key);
    if (it == t->slots.end()) return -1;
    if (is_param) *is_param = it->second.is_param ? 1 : 0;
    return it->second.off;
}
int ml_trainer_flat_numel(const ml_trainer* t, int64_t* n_param, int64_t* n_stat) {
    if (!t) return tfail(ML_ERR_ARG, "null argument");
    if (n_param) *n_param = t->n_param;
    if (n_stat) *n_stat = t->n_stat;
    return ML_OK;
}
int ml_trainer_copy_flat(ml_trainer* t, int what, float* dev_ptr, int64_t numel, void* stream) {
    if (!t || !dev_ptr || what < 0 || what > 4) return tfail(ML_ERR_ARG, "bad argument");
    const bool stats = what >= 3;
    if (numel != (stats ? t->n_stat : t->n_param))
        return tfail(ML_ERR_ARG, "ml_trainer_copy_flat: %lld elements, the buffer has %lld", (long long)numel, (long long)(stats ? t->n_stat : t->n_param));
    float* mine = what == 2 ? t->g : (stats ? t->stat : t->w);
    const bool in = what == 0 || what == 3;
    T_TRY(hipMemcpyAsync(in ? mine : dev_ptr, in ? dev_ptr : mine, (size_t)numel * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return ML_OK;
}
int ml_trainer_get_grad(ml_trainer* t, const char* key, float* host_data, int64_t numel) {
    return xfer(t, key, host_data, nullptr, numel, 2);
}
int64_t ml_trainer_num_steps(const ml_trainer* t) { return t ? t->step : 0; }

int ml_trainer_set_auto_tune(ml_trainer* t, int enable) {
    if (!t) return tfail(ML_ERR_ARG, "null trainer");
    t->auto_tune = enable != 0;
    return ML_OK;
}

int ml_trainer_set_lambdas(ml_trainer* t, const float* host8) {
    if (!t || !host8) return tfail(ML_ERR_ARG, "null argument");
    t->weighted = false;
    for (int i = 0; i < 8; ++i) {
        t->lambdas[i] = host8[i];
        if (host8[i] != 1.0f) t->weighted = true;
    }
    return ML_OK;
}

int ml_trainer_get_log_sigmas(ml_trainer* t, float* host8) {
    if (!t || !host8) return tfail(ML_ERR_ARG, "null argument");
    for (int i = 0; i < 8; ++i) host8[i] = t->log_sigma[i];
    return ML_OK;
}

int ml_trainer_set_log_sigmas(ml_trainer* t, const float* host8) {
    if (!t || !host8) return tfail(ML_ERR_ARG, "null argument");
    for (int i = 0; i < 8; ++i) t->log_sigma[i] = host8[i];
    return ML_OK;
}

int ml_trainer_set_route(ml_trainer* t, int route, int64_t fast_rows) {
    if (!t || route < 0 || route > 3) return tfail(ML_ERR_ARG, "bad route");
    t->route = route;
    if (fast_rows >= 0) t->fast_rows = fast_rows;
    return ML_OK;
}

int ml_trainer_last_route(const ml_trainer* t) { return t ? t->last_route : -1; }

int ml_trainer_eval(ml_trainer* t, const float* x_dev, const float* labels_dev, int label_cols, int64_t m, double* vals_host,
                    float* raw_out_dev, void* stream) {
    if (!t || !x_dev || !labels_dev || m < 1 || label_cols < 10 || !vals_host) return tfail(ML_ERR_ARG, "bad argument");
    if (t->C == 10 && label_cols < 11) return tfail(ML_ERR_ARG, "stereo labels need 11 columns");
    if (!mid_possible(t)) return tfail(ML_ERR_SHAPE, "ml_trainer_eval needs hidden % 64 == 0 (and out_features * hidden <= 15360)");
    // a whole validation set arrives in one call (Trainer.evaluate): walked in chunks so that the buffers stay at batch size and
    // the column-owner kernels at the row counts they were built for; the chunk means are combined weighted by their rows
    const int64_t CH = 8192;
    hipStream_t st = (hipStream_t)stream;
    t->fwd_pending_rows = 0;   // (the evaluation runs through the workspace a pending ml_trainer_forward_train left its activations in)
    double acc[mlt::LOSS_NV];
    for (int q = 0; q < mlt::LOSS_NV; ++q) acc[q] = 0.0;
    int rc;
    for (int64_t off = 0; off < m; off += CH) {
        const int64_t mc = m - off < CH ? m - off : CH;
        if ((rc = ensure_cap(t, mc))) return rc;
        int parts = 0;
        if ((rc = mid_forward(t, st, x_dev + off * t->in_f, labels_dev + off * label_cols, label_cols, mc, 0u, true, false, &parts)))
            return rc;
        if (raw_out_dev)
            T_TRY(hipMemcpyAsync(raw_out_dev + off * t->C, t->d_out, (size_t)mc * t->C * 4, hipMemcpyDeviceToDevice, st));
        if (!t->h_loss_dev)
            T_TRY(hipMemcpyAsync(t->h_loss, t->d_lpart, (size_t)parts * mlt::LOSS_NV * sizeof(double), hipMemcpyDeviceToHost, st));
        T_TRY(hipStreamSynchronize(st));
        double v[mlt::LOSS_NV];
        sum_loss_parts(t, parts, v);
        if (mc == m) {
            for (int q = 0; q < mlt::LOSS_NV; ++q) acc[q] = v[q];
        } else {
            for (int q = 0; q < mlt::LOSS_NV; ++q) acc[q] += v[q] * ((double)mc / (double)m);
        }
    }
    for (int q = 0; q < mlt::LOSS_NV; ++q) vals_host[q] = acc[q];
    return ML_OK;
}

int ml_trainer_can_eval(const ml_trainer* t) { return (t && mid_possible(t)) ? 1 : 0; }

int ml_val_stats(const float* raw_dev, int out_features, const float* labels_dev, int label_cols, int64_t m, double* vals_host,
                 void* stream) {
    if (!raw_dev || !labels_dev || !vals_host || m < 1 || (out_features != 9 && out_features != 10) || label_cols < 10 ||
        (out_features == 10 && label_cols < 11))
        return tfail(ML_ERR_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    int grid = (int)((m + 255) / 256);
    if (grid > 256) grid = 256;
    double* part = nullptr;
    T_TRY(hipMallocAsync((void**)&part, (size_t)grid * mlt::VAL_NV * sizeof(double), st));
    hipLaunchKernelGGL(mlt::val_stats_kernel, dim3(grid), dim3(256), 0, st, raw_dev, out_features, labels_dev, label_cols, m, part);
    std::vector<double> h((size_t)grid * mlt::VAL_NV);
    hipError_t e = hipMemcpyAsync(h.data(), part, h.size() * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFreeAsync(part, st);
    if (e != hipSuccess) return tfail(ML_ERR_HIP, "ml_val_stats: %s", hipGetErrorString(e));
    double s[mlt::VAL_NV];
    for (int q = 0; q < mlt::VAL_NV; ++q) {
        double a = 0.0;
        for (int b = 0; b < grid; ++b) a += h[(size_t)b * mlt::VAL_NV + q];
        s[q] = a;
    }
    const double dm = (double)m;
    for (int q = 0; q < mlt::LOSS_NV; ++q) vals_host[q] = s[q] / dm;
    vals_host[10] = s[10] / dm;                                                    // mean bi
    vals_host[11] = s[11] / dm;                                                    // share of |mu - d| <= bi
    vals_host[12] = m > 1 ? std::sqrt(std::max(0.0, (s[13] - s[12] * s[12] / dm) / (dm - 1.0))) : NAN;   // unbiased std of |mu - d|
    vals_host[13] = out_features == 10 ? 1.0 - s[14] / dm : 0.0;                   // aux accuracy
    return ML_OK;
}

int ml_gather_rows(const float* src_dev, int width, const int64_t* idx_dev, int64_t n, float* dst_dev, void* stream) {
    if (n == 0) return ML_OK;
    if (!src_dev || !idx_dev || !dst_dev || n < 0 || width < 1) return tfail(ML_ERR_ARG, "bad argument");
    const int64_t e = n * width;
    hipLaunchKernelGGL(mlt::gather_rows_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src_dev, width,
                       idx_dev, n, dst_dev);
    T_TRY(hipGetLastError());
    return ML_OK;
}

int ml_trainer_snapshot(ml_trainer* t, void* stream) {
    if (!t) return tfail(ML_ERR_ARG, "null trainer");
    if (!t->w_snap) T_TRY(hipMalloc((void**)&t->w_snap, (size_t)(t->n_param + t->n_stat) * 4));
    T_TRY(hipMemcpyAsync(t->w_snap, t->w, (size_t)t->n_param * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    T_TRY(hipMemcpyAsync(t->w_snap + t->n_param, t->stat, (size_t)t->n_stat * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return ML_OK;
}

int ml_trainer_restore(ml_trainer* t, void* stream) {
    if (!t || !t->w_snap) return tfail(ML_ERR_ARG, "no snapshot");
    T_TRY(hipMemcpyAsync(t->w, t->w_snap, (size_t)t->n_param * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    T_TRY(hipMemcpyAsync(t->stat, t->w_snap + t->n_param, (size_t)t->n_stat * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    T_TRY(hipStreamSynchronize((hipStream_t)stream));
    return ML_OK;
}

int ml_trainer_last_val_values(const ml_trainer* t, double* host10) {
    if (!t || !host10) return tfail(ML_ERR_ARG, "null argument");
    for (int i = 0; i < mlt::LOSS_NV; ++i) host10[i] = t->last_vals[i];
    return ML_OK;
}

int ml_trainer_set_tuning(ml_trainer* t, int apply_cols, int side_stream, int dw_layout) {
    if (!t || (apply_cols != 0 && apply_cols != 4 && apply_cols != 8 && apply_cols != 16))
        return tfail(ML_ERR_ARG, "apply_cols must be 4, 8 or 16 (0: unchanged)");
    if (apply_cols) t->apply_cols = apply_cols;
    if (dw_layout >= 0) {   // large-batch route: 0 transposed operand copies (round 2); 1 reduction-major operands + lines-only chain
        t->dw_trans = dw_layout ? 1 : 0;   // (default); 2 reduction-major operands, fp32 chain as with 0
        t->lines_chain = (dw_layout == 1 || dw_layout == 3) ? 1 : 0;
        t->merge23 = dw_layout == 3 ? 0 : 1;   // 3: layout 1 with w2 and w3 as two Linears (rounds 4-5: the A/B reference of round 6's merged pair)
    }
    if (side_stream >= 0) {   // 0: both gradients of a Linear in one launch (default); 1: weight gradients on the side stream; 2: two launches
        t->side_stream = side_stream == 1 ? 1 : 0;
        t->pair_gemm = side_stream == 0 ? 1 : 0;
    }
    return ML_OK;
}

int ml_trainer_debug_read(ml_trainer* t, int which, float* host_data, int64_t numel) {
    if (!t || !host_data || numel <= 0) return tfail(ML_ERR_ARG, "bad argument");
    const float* src = nullptr;
    int64_t cap = 0;
    if (which >= 0 && which < (int)t->bufs.size()) { src = t->bufs[which]; cap = t->cap * t->H; }
    else if (which == 200) { src = t->d_out; cap = t->cap * t->C; }
    else if (which == 201) { src = t->d_dout; cap = t->cap * t->C; }
    if (!src || numel > cap) return tfail(ML_ERR_ARG, "no such buffer / too many elements");
    T_TRY(hipDeviceSynchronize());
    T_TRY(hipMemcpy(host_data, src, (size_t)numel * 4, hipMemcpyDeviceToHost));
    return ML_OK;
}

int ml_debug_xgemm(const float* a_dev, int64_t lda, int a_layout, const float* b_dev, int64_t ldb, int b_layout, float* c_dev, int M,
                   int N, int K, const float* bias_dev, const float* res_dev, double* sumsq_dev, void* stream) {
    if (!a_dev || !b_dev || !c_dev || M < 1 || N < 64 || N % 64 || K < 1) return tfail(ML_ERR_ARG, "bad xgemm shape");
    if ((a_layout == 0 && K % 32) || (a_layout == 1 && M % 32) || (b_layout == 0 && K % 32))
        return tfail(ML_ERR_ARG, "xgemm: a k-contiguous operand needs K % 32 == 0, a reduction-major A needs M % 32 == 0");
    return launch_xgemm((hipStream_t)stream, a_dev, (long)lda, a_layout, b_dev, (long)ldb, b_layout, c_dev, N, M, N, K, bias_dev, res_dev,
                        sumsq_dev);
}

}  // extern "C"

namespace {

// The exact / large-batch routes of one iteration of the reference's loop (trainer.py:150-161), in PHASES: PH_FWD = the train-mode
// forward (`outputs = self.model(inputs)`), PH_LOSS = MultiTaskLoss and its gradient, PH_BWD = `loss.backward()` (every parameter
// gradient, unclipped, into t->g), PH_OPT = clip_grad_norm_(3) + Adam + StepLR.  ml_trainer_step runs all four; ml_trainer_forward_train
// runs PH_FWD and ml_trainer_backward PH_BWD from a caller-supplied gradient of the outputs (the autograd-capable module of
// monoloco_amd/network/architectures.py: a third-party loop that calls the model, computes its own loss and steps its own optimizer).
// The buffer plan below is a pure function of (t, m, route): a backward-only call finds every activation where the forward left it.
enum { PH_FWD = 1, PH_LOSS = 2, PH_BWD = 4, PH_OPT = 8, PH_ALL = 15 };

int step_phases(ml_trainer* t, const float* x_dev, const float* labels_dev, int label_cols, int64_t m, int update, double* losses_host,
                float* raw_out_dev, hipStream_t st, int phases, int route, const float* dout_dev) {
    const int H = t->H, S = t->S, C = t->C;
    int rc;
    if (phases & PH_FWD) {
        T_TRY(hipMemsetAsync(t->d_red_base, 0, (size_t)t->red_slots * (2 * H + 32) * sizeof(double), st));
        t->red_slot = 0;
        t->d_red = t->d_red_base;
    }
    // buffer plan
    int bi = 0;
    auto nb = [&]() { return t->bufs[bi++]; };
    std::vector<float*> a(S + 1), tt(S), za(S), zb(S);
    for (auto& p : a) p = nb();
    for (auto& p : tt) p = nb();
    float* z0 = nb();
    for (int s = 0; s < S; ++s) { za[s] = nb(); zb[s] = nb(); }
    float* z3 = nb();
    float* y2 = nb();
    float* y3 = nb();
    float* xhat = nb();
    float* gA = nb();
    float* gB = nb();
    // ---------------- forward (train mode)
    std::vector<Block> blocks;
    Block b0;
    b0.lin = "w1"; b0.bn = "batch_norm1"; b0.bn_idx = 0; b0.in_dim = t->in_f; b0.x = x_dev; b0.z = z0; b0.y = a[0]; b0.site = 0;
    // fast forward path: the activations that feed an H x H Linear also exist as lines (la[s] = a_s, lt[s] = t_s, ly2 = y2);
    // Linear `slot`s: 2s = stage s w1, 2s + 1 = stage s w2, 2S = w2, 2S + 1 = w3
    const bool fast = route == 1;
    if (phases & PH_FWD) t->packed_all = false;
    // w2 -> w3 as one Linear (train_kernels.h): where every consumer of the residual stream reads lines and the heads run on the skinny kernels
    const bool merged = fast && t->merge23 && t->dw_trans && t->lines_chain && skinny_ok(t, C - 1) && H % 256 == 0 && H <= 4096 && t->d_w32;
    if (fast && (phases & PH_FWD)) {
        T_TRY(hipMemsetAsync(t->wsc_base, 0, (size_t)(2 * S + 3) * 32, st));
        T_TRY(hipMemsetAsync(t->d_colmax, 0, (size_t)(2 * S + 3) * 2 * H * sizeof(float), st));
        if (merged) {   // W3 W2 (exact-fp32 MFMA GEMM of the mid route), W3 b2 + b3, u = W2^T w_aux, c = w_aux . b2 + b_aux
            if ((rc = launch_xgemm(st, P(t, "w3.weight"), H, 0, P(t, "w2.weight"), H, 1, t->d_w32, H, H, H, H, nullptr, nullptr, nullptr)))
                return rc;
            hipLaunchKernelGGL(mlt::gemv_rows_kernel, dim3(H / 4), dim3(256), 0, st, (const float*)P(t, "w3.weight"), H, H, H,
                               (const float*)P(t, "w2.bias"), (const float*)P(t, "w3.bias"), (const float*)nullptr, t->d_b32);
            hipLaunchKernelGGL(mlt::gemv_cols_kernel, dim3(H / 16), dim3(256), 0, st, (const float*)P(t, "w2.weight"), H, H, H,
                               (const float*)P(t, "w_aux.weight"), (const float*)nullptr, (const float*)nullptr, t->d_u23);
            hipLaunchKernelGGL(mlt::gemv_rows_kernel, dim3(1), dim3(256), 0, st, (const float*)P(t, "w_aux.weight"), H, 1, H,
                               (const float*)P(t, "w2.bias"), (const float*)P(t, "w_aux.bias"), (const float*)nullptr, t->d_c23);
        }
        // every weight image of the step -- W for the forward GEMMs, W^T for the data-gradient GEMMs -- in two launches up front
        // (the weights only change in the optimizer at the end of a step)
        const int nslots = merged ? 2 * S + 1 : 2 * S + 2;
        const mlt::WLayer* wdesc = merged ? t->d_wdesc_m : t->d_wdesc;
        hipLaunchKernelGGL(mlt::wmax_multi_kernel, dim3(64, nslots), dim3(256), 0, st, wdesc, (int64_t)H * H);
        hipLaunchKernelGGL(mlt::wpack_multi_kernel, dim3(nblk((int64_t)H * H / 8), nslots, 2), dim3(256), 0, st, wdesc, H);
        t->packed_all = true;
        if (t->pad_m != m) {
            // the weight-gradient GEMMs reduce over whole 64-row k-steps per split: rows m .. mT of every line buffer must be
            // zero (a smaller batch than the last one leaves old rows there); the writers only touch rows < m
            const int64_t mT = (m + 64 * t->ks - 1) / (64 * t->ks) * (64 * t->ks);
            if (mT > m) {
                for (char* lb : t->lbufs) T_TRY(hipMemsetAsync(lb + (size_t)m * H * 4, 0, (size_t)(mT - m) * H * 4, st));
                T_TRY(hipMemsetAsync(t->dzl + (size_t)m * H * 4, 0, (size_t)(mT - m) * H * 4, st));
            }
            t->pad_m = m;
        }
    }
    auto la = [&](int s) { return fast ? t->lbufs[s] : (char*)nullptr; };
    auto lt = [&](int s) { return fast ? t->lbufs[S + 1 + s] : (char*)nullptr; };
    char* ly2 = fast ? t->lbufs[2 * S + 1] : nullptr;
    // (round 4: on the large-batch route the residual stream a_0 .. a_S exists as lines only -- every consumer, the next Linear's
    // GEMMs and the next stage's skip connection, reads the lines)
    const bool stream_lines = fast && t->dw_trans && t->lines_chain;
    const bool fwd = (phases & PH_FWD) != 0;
    if (fwd && (rc = block_fwd(t, st, b0, m, nullptr, nullptr, la(0), -1, stream_lines))) return rc;
    std::vector<Block> sa(S), sb(S);
    for (int s = 0; s < S; ++s) {
        const std::string p = "linear_stages." + std::to_string(s) + ".";
        sa[s].lin = p + "w1"; sa[s].bn = p + "batch_norm1"; sa[s].bn_idx = 1 + 2 * s; sa[s].in_dim = H;
        sa[s].x = a[s]; sa[s].x_lines = la(s); sa[s].z = za[s]; sa[s].y = tt[s]; sa[s].site = 1 + 2 * s;
        // (t_s is read by the next Linear's GEMMs only -- forward and, reduction-major, weight gradient --: lines, no fp32 copy)
        if (fwd && (rc = block_fwd(t, st, sa[s], m, nullptr, la(s), lt(s), fast ? 2 * s : -1, stream_lines))) return rc;
        sb[s].lin = p + "w2"; sb[s].bn = p + "batch_norm2"; sb[s].bn_idx = 2 + 2 * s; sb[s].in_dim = H;
        sb[s].x = tt[s]; sb[s].x_lines = lt(s); sb[s].z = zb[s]; sb[s].y = a[s + 1]; sb[s].site = 2 + 2 * s;
        if (fwd && (rc = block_fwd(t, st, sb[s], m, a[s], lt(s), la(s + 1), fast ? 2 * s + 1 : -1, stream_lines,
                                   stream_lines ? la(s) : nullptr)))
            return rc;  // a_{s+1} = a_s + block(t_s)
    }
    const bool skinny = skinny_ok(t, C - 1);
    Block b3;
    b3.lin = "w3"; b3.bn = "batch_norm3"; b3.bn_idx = 2 * S + 1; b3.in_dim = H; b3.x = y2; b3.x_lines = ly2; b3.z = z3; b3.y = y3;
    b3.site = 2 * S + 1;
    if (merged) {   // z3 = a_S (W3 W2)^T + (W3 b2 + b3): y2 is never formed
        b3.x = nullptr;
        b3.x_lines = la(S);
    }
    if (fwd && merged) {
        int64_t ga = (m + 7) / 8;
        if (ga > 2048) ga = 2048;
        hipLaunchKernelGGL(mlt::aux_lines_kernel, dim3((unsigned)ga), dim3(256), 0, st, (const char*)la(S), m, H, (const float*)t->d_u23,
                           (const float*)t->d_c23, t->d_out + (C - 1), C);
        if ((rc = block_fwd(t, st, b3, m, nullptr, la(S), nullptr, 2 * S + 2))) return rc;
        if (!skinny_heads(t, st, y3, m, P(t, "w_fin.weight"), P(t, "w_fin.bias"), C - 1, t->d_out, C))
            if ((rc = linear_fwd(t, st, y3, H, P(t, "w_fin.weight"), P(t, "w_fin.bias"), t->d_out, C, (int)m, C - 1, H))) return rc;
        if (raw_out_dev) T_TRY(hipMemcpyAsync(raw_out_dev, t->d_out, (size_t)m * C * 4, hipMemcpyDeviceToDevice, st));
    } else if (fwd) {
        if (fast) {
            if ((rc = fast_linear_fwd(t, st, la(S), "w2", y2, m, 2 * S))) return rc;
            hipLaunchKernelGGL(mlt::act_lines_kernel, dim3(nblk(m * H / 8)), dim3(256), 0, st, (const float*)y2, m, H, ly2);
        } else if ((rc = linear_fwd(t, st, a[S], H, P(t, "w2.weight"), P(t, "w2.bias"), y2, H, (int)m, H, H))) return rc;
        if (!(skinny && skinny_heads(t, st, y2, m, P(t, "w_aux.weight"), P(t, "w_aux.bias"), 1, t->d_out + (C - 1), C)))
            if ((rc = linear_fwd(t, st, y2, H, P(t, "w_aux.weight"), P(t, "w_aux.bias"), t->d_out + (C - 1), C, (int)m, 1, H))) return rc;
        if ((rc = block_fwd(t, st, b3, m, nullptr, ly2, nullptr, fast ? 2 * S + 1 : -1))) return rc;
        if (!(skinny && skinny_heads(t, st, y3, m, P(t, "w_fin.weight"), P(t, "w_fin.bias"), C - 1, t->d_out, C)))
            if ((rc = linear_fwd(t, st, y3, H, P(t, "w_fin.weight"), P(t, "w_fin.bias"), t->d_out, C, (int)m, C - 1, H))) return rc;
        if (raw_out_dev) T_TRY(hipMemcpyAsync(raw_out_dev, t->d_out, (size_t)m * C * 4, hipMemcpyDeviceToDevice, st));
    }
    // ---------------- loss and its gradient
    double lv[16] = {0};
    const bool task_weights = t->auto_tune || t->weighted;
    if (phases & PH_LOSS) {
        double* d_loss = t->d_red + 2 * H;  // the tail of the current (pre-zeroed) slot
        T_TRY(hipMemsetAsync(t->d_dout, 0, (size_t)m * C * 4, st));
        if ((rc = upload_task_weights(t, st, task_weights))) return rc;
        hipLaunchKernelGGL(mlt::loss_kernel, dim3(nblk(m)), dim3(256), 0, st, (const float*)t->d_out, C, labels_dev, label_cols, m,
                           t->d_dout, d_loss, (const float*)(task_weights ? t->d_tw : nullptr));
        T_TRY(hipMemcpyAsync(lv, d_loss, mlt::LOSS_NV * sizeof(double), hipMemcpyDeviceToHost, st));
    } else if ((phases & PH_BWD) && dout_dev) {   // the caller's own loss: its gradient with respect to the (m, C) outputs
        T_TRY(hipMemcpyAsync(t->d_dout, dout_dev, (size_t)m * C * 4, hipMemcpyDeviceToDevice, st));
    }
    if (!(phases & PH_BWD)) return ML_OK;
    // ---------------- backward
    // (large-batch route: every gradient tensor is written in full by its producer -- no 34 MB memset of the gradient buffer)
    if (!fast) T_TRY(hipMemsetAsync(t->g, 0, (size_t)t->n_param * 4, st));
    t->csf.count = 0;
    t->csf_defer = fast;
    // heads: column sums of dout give both biases
    if ((rc = col_stats(t, st, t->d_dout, nullptr, m, C))) return rc;
    col_sum_to_float(t, st, (const double*)t->d_red, C - 1, G(t, "w_fin.bias"));
    col_sum_to_float(t, st, (const double*)(t->d_red + (C - 1)), 1, G(t, "w_aux.bias"));
    // (round 6: with the merged pair b3's backward passes are lines-only -- they form dy3 = dout . w_fin themselves, Block::hs)
    const bool head_in_bwd = merged && (C - 1 == 8 || C - 1 == 9) && t->head_in_bwd;
    if (head_in_bwd) {
        b3.hs = t->d_dout;
        b3.ldh = C;
        b3.nh = C - 1;
        b3.hw = P(t, "w_fin.weight");
    }
    if (skinny) {
        if ((rc = skinny_dw(t, st, t->d_dout, C, C - 1, y3, m, G(t, "w_fin.weight"), 0))) return rc;
        if (!head_in_bwd) rc = skinny_out(t, st, t->d_dout, C, C - 1, P(t, "w_fin.weight"), H, 1, nullptr, gA, m, 0);
    } else {
        if ((rc = linear_bwd_weight(t, st, t->d_dout, C, y3, H, G(t, "w_fin.weight"), (int)m, C - 1, H))) return rc;
        rc = linear_bwd_data(t, st, t->d_dout, C, P(t, "w_fin.weight"), gA, H, (int)m, C - 1, H, 0);  // dy3
    }
    if (rc) return rc;
    // fast path: the H x H data and weight gradients run on the 3-product kernel as well; dz goes there as scaled lines
    // (t->dzl; the forward's line buffers stay: they are the weight-gradient GEMMs' second operand), see block_bwd
    char* dzl = fast ? t->dzl : nullptr;
    // what the merged pair still owes behind the deferred column sums (s3 = sum dz3 -> w3.bias, sum daux -> w_aux.bias): the weight
    // gradients from M = dz3^T a_S (left in w3.weight's gradient slot by block_bwd), and the vectors
    auto merged_tail = [&]() -> int {
        if (!merged) return 0;
        int r2;
        T_TRY(hipMemcpyAsync(t->d_m23, G(t, "w3.weight"), (size_t)H * H * 4, hipMemcpyDeviceToDevice, st));
        // dW3 = M W2^T + s3 (x) b2
        if ((r2 = launch_xgemm(st, t->d_m23, H, 0, P(t, "w2.weight"), H, 0, G(t, "w3.weight"), H, H, H, H, nullptr, nullptr, nullptr))) return r2;
        hipLaunchKernelGGL(mlt::rank1_add_kernel, dim3(nblk((int64_t)H * H)), dim3(256), 0, st, G(t, "w3.weight"), H, H, H,
                           (const float*)G(t, "w3.bias"), (const float*)P(t, "w2.bias"));
        // dW2 = W3^T M + w_aux (x) v
        if ((r2 = launch_xgemm(st, P(t, "w3.weight"), H, 1, t->d_m23, H, 1, G(t, "w2.weight"), H, H, H, H, nullptr, nullptr, nullptr))) return r2;
        hipLaunchKernelGGL(mlt::rank1_add_kernel, dim3(nblk((int64_t)H * H)), dim3(256), 0, st, G(t, "w2.weight"), H, H, H,
                           (const float*)P(t, "w_aux.weight"), (const float*)t->d_v23);
        // db2 = W3^T s3 + w_aux sum(daux);  dw_aux = W2 v + b2 sum(daux)
        hipLaunchKernelGGL(mlt::gemv_cols_kernel, dim3(H / 16), dim3(256), 0, st, (const float*)P(t, "w3.weight"), H, H, H,
                           (const float*)G(t, "w3.bias"), (const float*)P(t, "w_aux.weight"), (const float*)G(t, "w_aux.bias"),
                           G(t, "w2.bias"));
        hipLaunchKernelGGL(mlt::gemv_rows_kernel, dim3(H / 4), dim3(256), 0, st, (const float*)P(t, "w2.weight"), H, H, H,
                           (const float*)t->d_v23, (const float*)P(t, "w2.bias"), (const float*)G(t, "w_aux.bias"), G(t, "w_aux.weight"));
        if (hipGetLastError() != hipSuccess) return tfail(ML_ERR_HIP, "merged w2 / w3 gradient launches failed");
        return 0;
    };
    if (merged) {
        if ((rc = block_bwd(t, st, b3, m, gA, xhat, 2 * S + 2))) return rc;            // dz3 as lines; M = dz3^T a_S in w3.weight's gradient slot
        if ((rc = fast_linear_bwd_data(t, st, dzl, "w3", gA, m, 2 * S + 2, false))) return rc;   // gA = dz3 (W3 W2)
        {   // v = daux^T a_S, then gA += daux (x) u: gA = d a_S
            int64_t gy = (m + 15) / 16;
            if (gy > 64) gy = 64;
            while (gy > 1 && (size_t)gy * H > t->splitk_cap) gy /= 2;
            hipLaunchKernelGGL(mlt::dvec_lines_kernel, dim3(H / 128, (unsigned)gy), dim3(256), 0, st, (const float*)(t->d_dout + (C - 1)), C,
                               (const char*)la(S), m, H, t->d_splitk);
            hipLaunchKernelGGL(mlt::skinny_reduce_kernel, dim3(nblk(H)), dim3(256), 0, st, (const float*)t->d_splitk, (int)gy, 1, H, t->d_v23, 0);
            if ((rc = skinny_out(t, st, t->d_dout + (C - 1), C, 1, t->d_u23, H, 1, nullptr, gA, m, 1))) return rc;
        }
    } else {   // w2 and w3 as two Linears (every other route; dw_layout 3)
        if ((rc = block_bwd(t, st, b3, m, gA, xhat, fast ? 2 * S + 1 : -1))) return rc;                              // gA = dz3
        if (fast) rc = fast_linear_bwd_data(t, st, dzl, "w3", gB, m, 2 * S + 1, false);
        else rc = linear_bwd_data(t, st, gA, H, P(t, "w3.weight"), gB, H, (int)m, H, H, 0);                          // gB = dy2
        if (rc) return rc;
        if (skinny) {
            if ((rc = skinny_dw(t, st, t->d_dout + (C - 1), C, 1, y2, m, G(t, "w_aux.weight"), 0))) return rc;
            rc = skinny_out(t, st, t->d_dout + (C - 1), C, 1, P(t, "w_aux.weight"), H, 1, nullptr, gB, m, 1);
        } else {
            if ((rc = linear_bwd_weight(t, st, t->d_dout + (C - 1), C, y2, H, G(t, "w_aux.weight"), (int)m, 1, H))) return rc;
            rc = linear_bwd_data(t, st, t->d_dout + (C - 1), C, P(t, "w_aux.weight"), gB, H, (int)m, 1, H, 1);  // += daux (x) w_aux
        }
        if (rc) return rc;
        // y2 = w2 a_S + b2
        if ((rc = col_stats(t, st, gB, nullptr, m, H))) return rc;
        col_sum_to_float(t, st, (const double*)t->d_red, H, G(t, "w2.bias"));
        if (fast) {
            hipLaunchKernelGGL(mlt::wmax_kernel, dim3(1024), dim3(256), 0, st, (const float*)gB, m * H, t->wsc_base + 8 * (2 * S) + 3);
            if ((rc = fast_grad_lines(t, st, gB, m, 2 * S))) return rc;
            if ((rc = fast_linear_bwd_weight(t, st, a[S], la(S), "w2", m, 2 * S))) return rc;
            rc = fast_linear_bwd_data(t, st, dzl, "w2", gA, m, 2 * S, false);
        } else {
            if ((rc = linear_bwd_weight(t, st, gB, H, a[S], H, G(t, "w2.weight"), (int)m, H, H))) return rc;
            rc = linear_bwd_data(t, st, gB, H, P(t, "w2.weight"), gA, H, (int)m, H, H, 0);                            // gA = da_S
        }
        if (rc) return rc;
    }
    // residual stages, last to first:  a_{s+1} = a_s + B(A(a_s))
    for (int s = S - 1; s >= 0; --s) {
        // gB = dz_b from gA = d a_{s+1} (which stays: the skip connection adds to it below)
        if ((rc = block_bwd(t, st, sb[s], m, gB, xhat, fast ? 2 * s + 1 : -1, gA))) return rc;
        if (fast) {   // gB's fp32 dz_b has been consumed (dW) and its lines feed the GEMM: d t_s lands in gB directly
            if ((rc = fast_linear_bwd_data(t, st, dzl, sb[s].lin, gB, m, 2 * s + 1, false))) return rc;
        } else {
            float* gT = xhat;  // xhat is free again: reuse it for d t_s
            if ((rc = linear_bwd_data(t, st, gB, H, P(t, sb[s].lin + ".weight"), gT, H, (int)m, H, H, 0))) return rc;
            T_TRY(hipMemcpyAsync(gB, gT, (size_t)m * H * 4, hipMemcpyDeviceToDevice, st));                      // gB = d t_s
        }
        if ((rc = block_bwd(t, st, sa[s], m, gB, xhat, fast ? 2 * s : -1))) return rc;                           // gB = dz_a
        if (fast) rc = fast_linear_bwd_data(t, st, dzl, sa[s].lin, gA, m, 2 * s, true);
        else rc = linear_bwd_data(t, st, gB, H, P(t, sa[s].lin + ".weight"), gA, H, (int)m, H, H, 1);            // da_s += ...
        if (rc) return rc;
    }
    if ((rc = block_bwd(t, st, b0, m, gA, xhat, -2))) return rc;
    if (!(phases & PH_OPT)) {   // the caller clips and steps its own optimizer: the raw gradients stay in t->g
        flush_col_sums(t, st);
        t->csf_defer = false;
        if ((rc = merged_tail())) return rc;
        T_TRY(hipStreamSynchronize(st));
        return ML_OK;
    }
    // ---------------- clip (always) + Adam + StepLR (per batch, only when updating)
    const int64_t k = t->step + 1;
    const float lr = t->lr0 * std::pow(t->gamma, (float)(t->step / t->sched_step));
    const float bc1 = 1.f - std::pow(0.9f, (float)k), bc2 = 1.f - std::pow(0.999f, (float)k);
    {
        double* d_ss = t->d_red + 2 * H + 16;  // pre-zeroed, never shared with d_loss (other offset)
        flush_col_sums(t, st);
        t->csf_defer = false;
        if ((rc = merged_tail())) return rc;
        hipLaunchKernelGGL(mlt::sumsq_kernel, dim3(512), dim3(256), 0, st, (const float*)t->g, t->n_param, d_ss);
        hipLaunchKernelGGL(mlt::clip_adam_kernel, dim3(nblk(t->n_param)), dim3(256), 0, st, t->w, t->g, t->m1, t->m2, t->n_param,
                           (const double*)d_ss, 3.0f, lr, 0.9f, 0.999f, 1e-8f, bc1, bc2, update ? 1 : 0);
        if (update) t->step++;
    }
    T_TRY(hipStreamSynchronize(st));
    finish_step_host(t, lv, task_weights, update, lr, bc1, bc2, losses_host);
    return ML_OK;
}

}  // namespace

extern "C" {

int ml_trainer_step(ml_trainer* t, const float* x_dev, const float* labels_dev, int label_cols, int64_t m,
                    int update, double* losses_host, float* raw_out_dev, void* stream) {
    if (!t || !x_dev || !labels_dev || m <= 1 || label_cols < 10) return tfail(ML_ERR_ARG, "bad argument");
    if (t->C == 10 && label_cols < 11) return tfail(ML_ERR_ARG, "stereo labels need 11 columns");
    int rc = ensure_cap(t, m);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int route = pick_route(t, m);
    t->last_route = route;
    t->fwd_pending_rows = 0;
    if (route == 2) {
        if (!mid_rows_ok(t, m)) return tfail(ML_ERR_SHAPE, "the mid route takes operands below 4 GiB (rows x hidden x 4)");
        return step_mid(t, x_dev, labels_dev, label_cols, m, update, losses_host, raw_out_dev, st);
    }
    return step_phases(t, x_dev, labels_dev, label_cols, m, update, losses_host, raw_out_dev, st, PH_ALL, route, nullptr);
}

// The train-mode forward on its own (reference trainer.py:155 `outputs = self.model(inputs)` with the module in train mode,
// architectures.py:48-71: batch-statistics BatchNorm with the running-statistics update, dropout from the counter-based generator --
// fresh masks per call): raw_out_dev (m, C).  Everything the backward needs stays in the trainer's workspace; x_dev must stay valid and
// unchanged until ml_trainer_backward has run.  Exact-fp32 route below fast_rows rows, the 3-product large-batch route from there on
// (the fused small-batch route of ml_trainer_step has no seam between forward, loss and backward).
int ml_trainer_forward_train(ml_trainer* t, const float* x_dev, int64_t m, float* raw_out_dev, void* stream) {
    if (!t || !x_dev || !raw_out_dev || m <= 1) return tfail(ML_ERR_ARG, "bad argument");
    int rc = ensure_cap(t, m);
    if (rc) return rc;
    int route = pick_route(t, m);
    if (route == 2) route = 0;
    t->last_route = route;
    ++t->fwd_calls;
    t->fwd_pending_rows = 0;
    if ((rc = step_phases(t, x_dev, nullptr, 0, m, 0, nullptr, raw_out_dev, (hipStream_t)stream, PH_FWD, route, nullptr))) return rc;
    t->fwd_pending_rows = m;
    t->fwd_pending_x = x_dev;
    return ML_OK;
}

// `loss.backward()` for the outputs of the last ml_trainer_forward_train (trainer.py:158): dout_dev (m, C) = gradient of the caller's
// loss with respect to those outputs -> every parameter gradient, UNCLIPPED, readable through ml_trainer_get_grad (the caller clips and
// steps its own optimizer, trainer.py:159-160).  Synchronises the stream.  The gradient with respect to the inputs is not computed
// (the reference's inputs are data).
int ml_trainer_backward(ml_trainer* t, const float* dout_dev, int64_t m, void* stream) {
    if (!t || !dout_dev) return tfail(ML_ERR_ARG, "bad argument");
    if (t->fwd_pending_rows != m || m <= 1)
        return tfail(ML_ERR_STATE, "ml_trainer_backward: no ml_trainer_forward_train of %lld rows is pending", (long long)m);
    const float* x_dev = t->fwd_pending_x;
    t->fwd_pending_rows = 0;
    return step_phases(t, x_dev, nullptr, 0, m, 0, nullptr, nullptr, (hipStream_t)stream, PH_BWD, t->last_route, dout_dev);
}

}  // extern "C"
