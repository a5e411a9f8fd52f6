// monoloco_hip.hip -- C ABI (include/monoloco_hip.h) of the gfx950 monoloco hot path.
//
// Host side of the library: checkpoint intake keyed by the reference's state_dict names,
// BatchNorm folding / w3*w2 merging in fp64, fp16 hi|lo weight packing, workspace management and
// the launch sequences.  Device side: dense_kernel.h (MFMA dense layers) and geom_kernels.h.
// No torch types, no exceptions across the ABI, no allocation in hot calls once reserved.
#include "../../include/monoloco_hip.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "dense_kernel.h"
#include "dense_kernel_pp.h"
#include "dense_kernel_w4.h"
#include "dense_small.h"
#include "dense_mid.h"
#include "geom_kernels.h"
#include "geom_ops.h"
#include "mc_kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(ML_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

// ---- fp32 -> fp16 bits, round-to-nearest-even (host; the device uses v_cvt_f16_f32) ----
uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (x >= 0x477ff000u) {  // >= 65520 rounds to inf
        return (uint16_t)(sign | 0x7c00u);
    }
    if (x < 0x38800000u) {  // below the smallest normal half (2^-14): subnormal or zero
        if (x < 0x33000000u) return (uint16_t)sign;  // < 2^-25 -> 0
        const int e = (int)(x >> 23);                 // biased exponent
        const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;  // 14..24 : bits dropped so that the result counts 2^-24 units
        const uint32_t q = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1);
        uint32_t r = q;
        if (rem > half || (rem == half && (q & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    const uint32_t e = (x >> 23) - 112;  // re-bias 127 -> 15
    const uint32_t mant = x & 0x7fffffu;
    uint32_t h = (e << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu;
    const uint32_t m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else {
            float v = (float)m * 5.9604644775390625e-8f;  // m * 2^-24
            memcpy(&x, &v, 4);
            x |= sign;
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e + 112) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// fp32 -> bf16 bits, round-to-nearest-even (what v_cvt_pk_bf16_f32 does on the device)
uint16_t f32_to_bf16_bits(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u);  // quiet NaN
    x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}

void split_host(float v, uint16_t& hi, uint16_t& lo) {
    float c = v;
    if (c > 65504.0f) c = 65504.0f;
    if (c < -65504.0f) c = -65504.0f;
    hi = f32_to_f16_bits(c);
    lo = f32_to_f16_bits(c - f16_bits_to_f32(hi));
}

inline int round_up(int v, int q) { return (v + q - 1) / q * q; }
inline int64_t round_up64(int64_t v, int64_t q) { return (v + q - 1) / q * q; }

// One dense (MFMA) layer after folding: y = act(x . W^T + b) (+ residual)
struct DenseLayer {
    int n = 0, k = 0, kpad = 0;
    int relu = 0;
    int scale_pow2 = 0;
    std::vector<float> w, b;  // folded fp32 (n*k), (n)
    std::vector<uint16_t> packed;  // host image of d_w (kept only for host-only models)
    char* d_w = nullptr;      // device, line format (n, kpad), pre-scaled
    float* d_b = nullptr;     // device fp32 (n)
    float* d_bs = nullptr;    // device fp32 (n): bias * 2^scale_pow2
    int src = 0, dst = 0;     // activation buffer ids: 0 = input lines, 1 = A, 2 = B
    int res = -1;             // residual buffer id or -1
};

struct Head {
    int nh = 0;
    int src = 0;   // activation buffer id
    int col0 = 0;  // first raw column
    int after_layer = 0;  // run after dense layer index
    bool mc_site = false; // MC-dropout: the top-level dropout sits on this head's input (w_fin only)
    std::vector<float> w, b;
    float* d_w = nullptr;
    float* d_b = nullptr;
};

}  // namespace

// Path selection of ONE handle (ml_loco_set_tuning; nothing process-global: a handle per device / stream stays thread-compatible):
//   rows <= small_rows take the small-row dense kernels (no LDS staging) instead of the 256x256-tile persistent kernel; above
//   small32_rows those use 32x32 output tiles (16x16 below); chunk_rows > 0 walks the batch in row chunks through all layers
//   (Infinity-Cache residency experiment, off by default); tile_kernel 4 = dense_kernel_w4 for the long-K layers +
//   dense_kernel_pp for the input / fused-head layers, 2 = dense_kernel_pp everywhere; tile_all: dense_kernel_w4 wherever it runs;
//   small_rows < rows <= mid_rows take dense_mid_kernel (128 x 64 / 128 x 128 tiles: the 256x256 tiles are fewer than the CUs
//   there), mid_tile 64 | 128 forces its tile height (0 = 64 while the 128-row tiles are fewer than the CUs).
struct Tuning {
    int small_rows = 512, small32_rows = 128, chunk_rows = 0;
    int tile_kernel = 4, tile_all = 0;
    int mid_rows = 8192, mid_tile = 0;
    int half_from = 4096;   // mid window, rows above this: the long-K layers on dense_kernel_w4's half-size tile (mid_tile 256 forces it)
    int small_multi = 1;    // small-row window, > 64 rows: dense_small_multi_kernel (option "small_multi"; 0: one tile per workgroup, rounds 1-4)
    int mid_splitk = -1;    // dense_mid_kernel: k ranges per output tile (workgroups that share a tile, last arriver runs the epilogue):
                            // -1 = auto (as many as it takes to reach mid_wgs workgroups: 1 | 2 | 4), 0 / 1 = off, 2 / 4 = forced where K allows
    int mid_wgs = 256;      // ... auto: workgroups a dense_mid_kernel launch should reach: one per CU -- splitting only fills IDLE CUs (fewer
                            // tiles than CUs); two co-resident workgroups per CU lose 10-40 % (measured, profiles/r06_ablation.md)
    int mid_prep = 0;       // 1: the mono pipeline's input layer pre-processes its own persons (dense_mid_kernel<.., PREP>) instead of prep_kernel
                            // in front of it -- built, bit-identical, and NO gain (2048 rows 133.6 vs 133.5 us per forward, 8192 rows
                            // 361.8 vs 368.2: eight column tiles repeat the work in a serial prologue, and prep_kernel's 5.6 us already hid
                            // behind the previous call's tail; profiles/r06_ablation.md section 6): off, kept as the A/B switch
    int mid_dma = 1;        // dense_mid_kernel's loader: 1 = LDS-DMA into a three-stage ring (round 6), 0 = global -> VGPR -> ds_write (rounds 3-5)
    int half_heads = 1;     // the whole mid window: both heads ride in the dense epilogues (half-size w4 tile / dense_mid_kernel) + tail_mono_kernel
                            // (option "mid_heads"; 0: heads_pair_kernel behind the last layer, rounds 3-4)
};

struct ml_loco {
    Tuning tune;
    int in_f = 0, hidden = 0, out_f = 0, num_stage = 0;
    int hidden_real = 0;  // the checkpoint's linear_size; `hidden` is that rounded up to the 256-column tile
    int precision = ML_PREC_F16X2, flags = 0;
    bool finalized = false;
    bool host_only = false;
    bool legacy = false;  // MonolocoModel (w1, stages, w2 -> out_f) instead of LocoModel
    int device = -1;
    std::map<std::string, std::vector<float>> tensors;
    std::vector<DenseLayer> layers;
    std::vector<Head> heads;
    int k0pad = 0;
    // workspace
    int64_t cap_rows = 0;  // padded rows the buffers hold
    char* buf[3] = {nullptr, nullptr, nullptr};
    float* d_xf32 = nullptr;    // (cap_rows, in_f) fp32 staging (stereo pre-process)
    float* d_centre = nullptr;  // (cap_rows, 2)
    float* d_raw = nullptr;     // (cap_rows, out_f)
    float* d_xl = nullptr;      // stereo: (cap_side, 34) each
    float* d_xr = nullptr;
    float* d_cl = nullptr;      // stereo: left centres
    int32_t* d_rowidx = nullptr;
    float* d_part = nullptr;    // fused-head partial sums [2*hidden/256][cap_rows][16] ...
    float* d_part_aux = nullptr;  // ... and, behind them, the fused w_aux head's [2*hidden/256][cap_rows] (same allocation)
    int64_t part_cells = 0;     // (slice, row) cells per head d_part / d_part_aux hold (part_cells_for)
    float* d_kpart = nullptr;   // dense_mid_kernel<.., SPLITK>: fp32 partial tiles [tiles][ksplit][128 x TM] ...
    unsigned* d_kcount = nullptr;   // ... and the tiles' arrival counters (zero between launches)
    int64_t kpart_floats = 0;
    // ml_loco_frame_mono's completion word (pinned, coherent) + the arrival counter of the last launch's workgroups; frame_flag_req:
    // the frame entry asks run_network to arm the flag in the launch that ends a single image's forward
    int* h_done = nullptr;
    int* d_arrive = nullptr;
    int done_seq = 0;
    bool frame_flag_req = false, frame_flag_armed = false;
    struct PinnedSeen { const void* p = nullptr; size_t bytes = 0; } pinned_seen[8];   // ml_loco_frame_*: host ranges verified to be pinned
    int pinned_next = 0;
    int tune_version = 0;       // bumped by ml_loco_set_tuning / ml_loco_set_option: cached route plans of older versions are stale
    struct PlanCache* plans = nullptr;   // the route plans of the last few (rows, MC-dropout) calls (plan_for; freed in ml_loco_destroy)
    double* d_mc = nullptr;     // MC-dropout: running (sum, sum of squares) per person + per-(pass, person) partials: 4 doubles per row
    int64_t cap_side = 0;
    int64_t dev_bytes = 0;
    // optional per-launch timing of the dense kernel (ml_loco_profile_*): HIP events recorded on
    // the stream the kernel is launched on, one (start, stop) pair per launch
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;  // 2 per recorded launch
    std::vector<int> ev_layer;        // layer index of each recorded launch
    size_t ev_used = 0;               // pairs in use
};

namespace {

template <typename T>
int dev_alloc(ml_loco* h, T** p, int64_t bytes) {
    if (bytes <= 0) bytes = 16;
    HIP_TRY(hipMalloc((void**)p, (size_t)bytes));
    h->dev_bytes += bytes;
    return ML_OK;
}

void dev_free(void* p) {
    if (p) (void)hipFree(p);
}

const std::vector<float>* get_t(const ml_loco* h, const std::string& key, int64_t numel) {
    auto it = h->tensors.find(key);
    if (it == h->tensors.end()) {
        fail(ML_ERR_STATE, "state_dict tensor '%s' was never set", key.c_str());
        return nullptr;
    }
    if ((int64_t)it->second.size() != numel) {
        fail(ML_ERR_ARG, "tensor '%s' has %lld elements, expected %lld", key.c_str(),
             (long long)it->second.size(), (long long)numel);
        return nullptr;
    }
    return &it->second;
}

// Fold Linear `lin` (n x k) with eval-mode BatchNorm `bn` (or none) in fp64:
//   W' = W * g/sqrt(var+eps),  b' = (b - mean) * g/sqrt(var+eps) + beta
int fold(const ml_loco* h, const std::string& lin, const std::string& bn, int n, int k,
         std::vector<double>& W, std::vector<double>& B) {
    const auto* w = get_t(h, lin + ".weight", (int64_t)n * k);
    const auto* b = get_t(h, lin + ".bias", n);
    if (!w || !b) return ML_ERR_STATE;
    W.assign(w->begin(), w->end());
    B.assign(b->begin(), b->end());
    if (!bn.empty()) {
        const auto* g = get_t(h, bn + ".weight", n);
        const auto* be = get_t(h, bn + ".bias", n);
        const auto* mu = get_t(h, bn + ".running_mean", n);
        const auto* var = get_t(h, bn + ".running_var", n);
        if (!g || !be || !mu || !var) return ML_ERR_STATE;
        const double eps = 1e-5;  // nn.BatchNorm1d default, architectures.py:25,34
        for (int i = 0; i < n; ++i) {
            const double s = (double)(*g)[i] / std::sqrt((double)(*var)[i] + eps);
            for (int j = 0; j < k; ++j) W[(size_t)i * k + j] *= s;
            B[i] = (B[i] - (double)(*mu)[i]) * s + (double)(*be)[i];
        }
    }
    return ML_OK;
}

// W is (n x k) as folded; the layer is stored as (np x kp) with np >= n, kp >= k: any linear_size runs on the
// 256-column tiles -- padded output columns have zero weights and zero bias (relu(0) = 0, residual 0 + 0), padded
// input columns multiply activations that are exactly zero
void add_dense(ml_loco* h, const std::vector<double>& W, const std::vector<double>& B, int n, int k, int np, int kp,
               int relu, int src, int dst, int res) {
    DenseLayer L;
    L.n = np;
    L.k = kp;
    L.kpad = round_up(kp, 64);  // whole k64 double steps: the w4 kernel's ring holds two k32 steps
    L.relu = relu;
    L.src = src;
    L.dst = dst;
    L.res = res;
    L.w.assign((size_t)np * kp, 0.0f);
    L.b.assign(np, 0.0f);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < k; ++j) L.w[(size_t)i * kp + j] = (float)W[(size_t)i * k + j];
    for (int i = 0; i < n; ++i) L.b[i] = (float)B[i];
    h->layers.push_back(std::move(L));
}

// head weights (nh x k) -> (nh x kp), zero padded
std::vector<float> pad_head(const float* w, int nh, int k, int kp) {
    std::vector<float> out((size_t)nh * kp, 0.0f);
    for (int o = 0; o < nh; ++o)
        for (int j = 0; j < k; ++j) out[(size_t)o * kp + j] = w[(size_t)o * k + j];
    return out;
}

// C (n x k) = A (n x p) * B (p x k), fp64, i-k-j order (unit stride inner loop)
void matmul64(const std::vector<double>& A, const std::vector<double>& Bm, int n, int p, int k,
              std::vector<double>& C) {
    C.assign((size_t)n * k, 0.0);
    for (int i = 0; i < n; ++i) {
        double* c = &C[(size_t)i * k];
        for (int l = 0; l < p; ++l) {
            const double a = A[(size_t)i * p + l];
            const double* b = &Bm[(size_t)l * k];
            for (int j = 0; j < k; ++j) c[j] += a * b[j];
        }
    }
}

// Host half of the weight preparation: per-layer power-of-two scale (max |W| lands in
// [2^13, 2^14) so that the fp16 lo parts of all but negligible weights stay normal -- DESIGN.md
// "precision"), fp16 hi|lo split and the k32 line layout of dense_kernel.h.
void pack_layer(int precision, DenseLayer& L) {
    float mx = 0.f;
    for (float v : L.w) mx = std::fmax(mx, std::fabs(v));
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) e = (int)std::floor(std::log2(16384.0 / (double)mx));
    if (e > 40) e = 40;
    if (e < -40) e = -40;
    L.scale_pow2 = e;
    const float sc = std::ldexp(1.0f, e);
    L.packed.assign((size_t)L.n * L.kpad * 2, 0);
    for (int i = 0; i < L.n; ++i) {
        uint16_t* row = &L.packed[(size_t)i * L.kpad * 2];
        for (int j = 0; j < L.k; ++j) {
            uint16_t hi, lo;
            split_host(L.w[(size_t)i * L.k + j] * sc, hi, lo);
            const int b = j >> 5, o = j & 31;
            if (precision == ML_PREC_BF16) {  // one bf16 value in the hi slot (the scale is a power of two: exact)
                hi = f32_to_bf16_bits(L.w[(size_t)i * L.k + j] * sc);
                lo = 0;
            }
            row[b * 64 + o] = hi;
            row[b * 64 + 32 + o] = (precision == ML_PREC_F16X2) ? lo : 0;
        }
    }
}

int upload_layer(ml_loco* h, DenseLayer& L) {
    const size_t bytes = L.packed.size() * 2;
    int rc = dev_alloc(h, &L.d_w, (int64_t)bytes);
    if (rc) return rc;
    rc = dev_alloc(h, &L.d_b, (int64_t)L.n * 4);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(L.d_w, L.packed.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(L.d_b, L.b.data(), (size_t)L.n * 4, hipMemcpyHostToDevice));
    rc = dev_alloc(h, &L.d_bs, (int64_t)L.n * 4);
    if (rc) return rc;
    std::vector<float> bs(L.n);
    for (int i = 0; i < L.n; ++i) bs[i] = std::ldexp(L.b[i], L.scale_pow2);
    HIP_TRY(hipMemcpy(L.d_bs, bs.data(), (size_t)L.n * 4, hipMemcpyHostToDevice));
    std::vector<uint16_t>().swap(L.packed);
    return ML_OK;
}

int free_workspace(ml_loco* h) {
    for (auto& b : h->buf) {
        dev_free(b);
        b = nullptr;
    }
    dev_free(h->d_xf32);
    dev_free(h->d_centre);
    dev_free(h->d_raw);
    dev_free(h->d_rowidx);
    dev_free(h->d_part);
    dev_free(h->d_kpart);
    dev_free(h->d_kcount);
    h->d_kpart = nullptr;
    h->d_kcount = nullptr;
    h->kpart_floats = 0;
    dev_free(h->d_mc);
    h->d_mc = nullptr;
    h->d_xf32 = h->d_centre = h->d_raw = h->d_part = h->d_part_aux = nullptr;
    h->d_rowidx = nullptr;
    h->cap_rows = 0;
    return ML_OK;
}

// (slice, row) cells of fused-head partial sums a workspace of `need` rows holds: 128-column slices from the tile kernels at every
// size, dense_mid_kernel's finer 64-column slices for the rows the mid window can be tuned to (<= 16384).  ONE formula for the
// allocation (ensure_rows) and for the plan's prediction (part_fits): the printed plan is the plan that runs
static int64_t part_cells_for(int hidden, int64_t need) {
    const int64_t fine = (int64_t)(hidden / 64) * (need < 16384 ? need : 16384);
    const int64_t coarse = (int64_t)(2 * hidden / 256) * need;
    return fine > coarse ? fine : coarse;
}

constexpr int MID_KCOUNT = 8192;   // arrival counters of dense_mid_kernel<.., SPLITK>: one per output tile of a launch

int ensure_rows(ml_loco* h, int64_t rows) {
    const int64_t need = round_up64(rows > 0 ? rows : 1, 256);
    if (need <= h->cap_rows) return ML_OK;
    HIP_TRY(hipDeviceSynchronize());
    free_workspace(h);
    int rc;
    if ((rc = dev_alloc(h, &h->buf[0], need * (int64_t)h->k0pad * 4))) return rc;
    if ((rc = dev_alloc(h, &h->buf[1], need * (int64_t)h->hidden * 4))) return rc;
    if ((rc = dev_alloc(h, &h->buf[2], need * (int64_t)h->hidden * 4))) return rc;
    if ((rc = dev_alloc(h, &h->d_xf32, need * (int64_t)h->in_f * 4))) return rc;
    if ((rc = dev_alloc(h, &h->d_centre, need * 2 * 4))) return rc;
    if ((rc = dev_alloc(h, &h->d_raw, need * (int64_t)h->out_f * 4))) return rc;
    if ((rc = dev_alloc(h, &h->d_rowidx, need * 4))) return rc;
    // partial sums of the fused heads (16 floats per cell for w_fin, then 1 per cell for w_aux)
    h->part_cells = part_cells_for(h->hidden, need);
    if ((rc = dev_alloc(h, &h->d_part, h->part_cells * 17 * 4))) return rc;
    h->d_part_aux = h->d_part + h->part_cells * 16;
    if ((rc = dev_alloc(h, &h->d_mc, need * 4 * (int64_t)sizeof(double)))) return rc;
    // split-K partial tiles of the mid window (dense_mid_kernel<.., SPLITK>): two k ranges of up to 4096 rows (four of 2048, ...)
    h->kpart_floats = (need < 4096 ? need : 4096) * (int64_t)h->hidden * 2;
    if ((rc = dev_alloc(h, &h->d_kpart, h->kpart_floats * 4))) return rc;
    if ((rc = dev_alloc(h, &h->d_kcount, MID_KCOUNT * 4))) return rc;
    HIP_TRY(hipMemset(h->d_kcount, 0, MID_KCOUNT * 4));
    h->cap_rows = need;
    return ML_OK;
}

int ensure_side(ml_loco* h, int64_t persons) {
    if (persons <= h->cap_side) return ML_OK;
    HIP_TRY(hipDeviceSynchronize());
    dev_free(h->d_xl);
    dev_free(h->d_xr);
    dev_free(h->d_cl);
    const int64_t need = round_up64(persons, 256);
    int rc;
    if ((rc = dev_alloc(h, &h->d_xl, need * mlk::NIN * 4))) return rc;
    if ((rc = dev_alloc(h, &h->d_xr, need * mlk::NIN * 4))) return rc;
    if ((rc = dev_alloc(h, &h->d_cl, need * 2 * 4))) return rc;
    h->cap_side = need;
    return ML_OK;
}

mlk::Kinv make_kinv(const float* k) {
    mlk::Kinv ki;
    for (int i = 0; i < 9; ++i) ki.k[i] = k[i];
    return ki;
}

int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}


bool use_small_path(const Tuning& tu, int precision, int64_t rows) {
    // (the bf16 comparison mode exists on the tile path only)
    return precision != ML_PREC_BF16 && rows <= tu.small_rows;
}

bool use_mid_path(const Tuning& tu, int precision, int64_t rows) {
    return precision != ML_PREC_BF16 && rows > tu.small_rows && rows <= tu.mid_rows;
}

// does a small-row layer of `rows` rows and reduction length K run on dense_small_multi_kernel (a workgroup keeps its weight rows and
// walks several row tiles)?  From 65 rows on (up to 64 rows the 16 x 16 tiles are at most one per CU and a layer costs one latency)
bool small_multi_runs(const Tuning& tu, int64_t rows, int K) {
    // ... up to 128: a workgroup then has at most two row tiles, both requested with the weights (beyond, the 32 x 32 tiles win:
    // tools/sweep_small_rows.py)
    return tu.small_multi && rows > 64 && rows <= 128 && K / 32 <= mlk::SMALL_MULTI_MAX_LINES;
}

// does the tile path run dense_kernel_w4 for a layer with this K (and fused head width)?
bool w4_runs(const Tuning& tu, int K, int head_nh) {
    return tu.tile_kernel == 4 && K % 64 == 0 && (tu.tile_all || (K > 128 && head_nh <= 0));
}

// does a call of `rows` network rows (inside the mid window) put its long-K layers on dense_kernel_w4's half-size tile?  Decided on
// the whole call like the window itself: row chunks of one call must take the same kernels (bit-identical results)
bool use_half_tile(const Tuning& tu, int precision, int64_t rows) {
    return precision == ML_PREC_F16X2 && (tu.mid_tile == 256 || (tu.mid_tile == 0 && round_up64(rows, 256) > tu.half_from));
}

// tile height of a dense_mid_kernel launch whose 128-row tiling has `tiles128` tiles (ONE rule for launch_dense and for the route label)
int mid_tile_rows(const Tuning& tu, int precision, int tiles128) {
    if (tu.mid_tile == 64 || tu.mid_tile == 128) return tu.mid_tile;
    // (LDS-DMA loader: 128-row tiles as soon as the 64-row tiles no longer fit the CUs in ONE wave -- a forward costs 128-139 us with up to
    //  256 of the 64-row tiles, 197-203 us with 272 ... 384 of them and 183-190 us with the 136 ... 192 128-row tiles of the same rows:
    //  2176 ... 3072 rows, tools/ab_options.py, profiles/r06_ablation.md section 3)
    if (tu.mid_dma && precision == ML_PREC_F16X2) return 2 * tiles128 > num_cus() ? 128 : 64;
    return tiles128 >= num_cus() ? 128 : 64;
}

// k ranges per output tile of a dense_mid_kernel launch (round 6): the smallest of 1, 2, 4 that reaches tu.mid_wgs workgroups, limited
// by what the reduction length allows (>= 4 k32 steps per range, equal ranges), by `room` (ranges the partial-tile workspace holds) and by
// Tuning::mid_splitk (0 / 1 off, 2 / 4 forced); the 3-product mode only
int mid_ksplit(const Tuning& tu, int precision, int tiles, int nk, int room) {
    if (precision != ML_PREC_F16X2 || tu.mid_splitk == 0 || tu.mid_splitk == 1 || room < 2) return 1;
    // (auto: two ranges when the tiles cover at most half of mid_wgs, never four -- 640 rows: 110 us per forward with 2, 136 with 4, 124 with 1)
    int want = tu.mid_splitk > 1 ? tu.mid_splitk : (2 * tiles <= tu.mid_wgs ? 2 : 1);
    if (want > room) want = room >= 4 ? 4 : (room >= 2 ? 2 : 1);
    while (want > 1 && (nk % want != 0 || nk / want < 4)) want >>= 1;
    return want < 1 ? 1 : want;
}

// mid: 0 = no, 1 = the mid-size path (dense_mid_kernel), 2 = ... with dense_kernel_w4's half-size tile for the long-K layers
int launch_dense(const Tuning& tu, int precision, const mlk::DenseParams& p_in, hipStream_t st, int head_nh = 0, int64_t rows = -1,
                 int mid = 0) {
    mlk::DenseParams p = p_in;
    p.debug = 0;
    p.trace = nullptr;

    if (mid) {  // the caller chose the mid-size path (fused heads only on the half-size w4 tile)
        if (p.N % mlk::MID_TN != 0) return fail(ML_ERR_STATE, "dense_mid_kernel: N %% 128 == 0");
        // the upper part of the window: dense_kernel_w4 with its HALF-SIZE tile (256 n x 128 m, NJ = 2) for the long-K layers --
        // one wave per SIMD, AGPR accumulators, the LDS-DMA ring: half the LDS traffic per MFMA of dense_mid_kernel's 64 x 64 wave
        // tiles, and (rows / 128) * (N / 256) tiles where the full-size kernel has half as many (8192 rows: 256 instead of 128)
        const bool half = mid == 2 && precision == ML_PREC_F16X2 && p.K > 128 && p.K % 64 == 0 && p.N % 256 == 0 && p.M_pad % 128 == 0;
        if (half) {
            const int htiles = (p.M_pad / 128) * (p.N / mlk::BN);
            const dim3 hgrid((unsigned)(htiles < num_cus() ? htiles : num_cus()));
#define ML_HALF(RL, RS, HD) hipLaunchKernelGGL((mlk::dense_kernel_w4<3, RL, RS, HD, false, 2>), hgrid, dim3(mlk::W4_THREADS), 0, st, p)
            // round 5: the heads ride in the half-size tile's epilogues exactly as in the full-size tile's (HEAD = -1: w_aux in the
            // store epilogue of the layer that produces its input; HEAD = 8 | 9: w_fin's partial sums instead of the activation tile)
            if (head_nh == -1) {
                if (p.relu && p.res) ML_HALF(true, true, -1);
                else if (!p.relu && !p.res) ML_HALF(false, false, -1);
                else return fail(ML_ERR_STATE, "fused aux head: unsupported layer form");
            } else if (head_nh == 8) ML_HALF(true, false, 8);
            else if (head_nh == 9) ML_HALF(true, false, 9);
            else if (head_nh != 0) return fail(ML_ERR_STATE, "half-size tile: unsupported fused head %d", head_nh);
            else if (p.relu) {
                if (p.res) ML_HALF(true, true, 0);
                else ML_HALF(true, false, 0);
            } else {
                if (p.res) ML_HALF(false, true, 0);
                else ML_HALF(false, false, 0);
            }
#undef ML_HALF
            HIP_TRY(hipGetLastError());
            return ML_OK;
        }
        const int tiles128 = (p.M_pad / 128) * (p.N / mlk::MID_TN);
        // 128-row tiles once they cover the CUs -- with the LDS-DMA loader (one 96 KiB workgroup per CU) as soon as the 64-row tiles would
        // need a second wave (2304 rows: 183 vs 198 us per forward; 2048 rows, 256 64-row tiles: 139 vs 181)
        const int tm = mid_tile_rows(tu, precision, tiles128);
        const int tiles = (p.M_pad / tm) * (p.N / mlk::MID_TN);
        // round 6, split-K: k ranges per output tile so that the launch reaches tu.mid_wgs workgroups (two per CU) -- 1, 2 or 4, each at
        // least four k32 steps long, as many as the caller's workspace holds (p.ksplit on entry; no workspace: none) and the counters cover
        const int ksplit = mid_ksplit(tu, precision, tiles, p.K / 32, (p.kpart && p.kcount && tiles <= MID_KCOUNT) ? p.ksplit : 1);
        p.ksplit = ksplit;
        const dim3 grid((unsigned)(((tiles * ksplit + 7) / 8) * 8));
        const bool dma = tu.mid_dma && precision == ML_PREC_F16X2;
        if (p.prep_kps) {   // the mono pipeline's input layer with the pre-process inside (run_network only asks where mid_prep_runs says so)
            if (!(dma && head_nh == 0 && p.relu && !p.res && p.K == 64))
                return fail(ML_ERR_STATE, "dense_mid_kernel: the fused pre-process needs the plain K = 64 input layer on the LDS-DMA loader");
            p.ksplit = 1;
            const dim3 pgrid((unsigned)(((tiles + 7) / 8) * 8));
            if (tm == 128) hipLaunchKernelGGL((mlk::dense_mid_kernel<3, true, false, 128, 0, false, true, true>), pgrid, dim3(mlk::MID_THREADS), 0, st, p);
            else hipLaunchKernelGGL((mlk::dense_mid_kernel<3, true, false, 64, 0, false, true, true>), pgrid, dim3(mlk::MID_THREADS), 0, st, p);
            HIP_TRY(hipGetLastError());
            return ML_OK;
        }
#define ML_MID_L(NS, RL, RS, TMV, HD, SK, DM) \
    hipLaunchKernelGGL((mlk::dense_mid_kernel<NS, RL, RS, TMV, HD, SK, DM>), grid, dim3(mlk::MID_THREADS), 0, st, p)
#define ML_MID(NS, RL, RS, HD)                                                   \
    do {                                                                         \
        if (NS == 3 && dma) {                                                    \
            if (ksplit > 1) {                                                    \
                if (tm == 128) ML_MID_L(3, RL, RS, 128, HD, true, true);         \
                else ML_MID_L(3, RL, RS, 64, HD, true, true);                    \
            } else if (tm == 128) ML_MID_L(3, RL, RS, 128, HD, false, true);     \
            else ML_MID_L(3, RL, RS, 64, HD, false, true);                       \
        } else if (ksplit > 1 && NS == 3) {                                      \
            if (tm == 128) ML_MID_L(3, RL, RS, 128, HD, true, false);            \
            else ML_MID_L(3, RL, RS, 64, HD, true, false);                       \
        } else if (tm == 128) ML_MID_L(NS, RL, RS, 128, HD, false, false);       \
        else ML_MID_L(NS, RL, RS, 64, HD, false, false);                         \
    } while (0)
#define ML_MID_NS(NS)                                     \
    do {                                                  \
        if (p.relu) {                                     \
            if (p.res) ML_MID(NS, true, true, 0);         \
            else ML_MID(NS, true, false, 0);              \
        } else {                                          \
            if (p.res) ML_MID(NS, false, true, 0);        \
            else ML_MID(NS, false, false, 0);             \
        }                                                 \
    } while (0)
        if (head_nh != 0) {   // round 5: the heads in dense_mid_kernel's epilogues (3-product mode; 64-column slices of partial sums)
            if (precision != ML_PREC_F16X2) return fail(ML_ERR_STATE, "dense_mid_kernel: fused heads in the f16x2 mode only");
            if (head_nh == -1) {
                if (p.relu && p.res) ML_MID(3, true, true, -1);
                else if (!p.relu && !p.res) ML_MID(3, false, false, -1);
                else return fail(ML_ERR_STATE, "fused aux head: unsupported layer form");
            } else if (head_nh == 8 && p.relu && !p.res) ML_MID(3, true, false, 8);
            else if (head_nh == 9 && p.relu && !p.res) ML_MID(3, true, false, 9);
            else return fail(ML_ERR_STATE, "dense_mid_kernel: unsupported fused head %d", head_nh);
        } else if (precision == ML_PREC_F16X2) ML_MID_NS(3);
        else ML_MID_NS(1);
#undef ML_MID_NS
#undef ML_MID
#undef ML_MID_L
        HIP_TRY(hipGetLastError());
        return ML_OK;
    }

    if (rows >= 0 && head_nh == 0) {  // the caller chose the small-row path
        // 16x16 tiles while they are few (latency: more workgroups), 32x32 tiles (half the L2 traffic) once there
        // are at least ~256 of those
        if (rows == 0) return ML_OK;
        if (small_multi_runs(tu, rows, p.K)) {
            // round 5: more than 64 rows -- workgroups keep their 16 weight rows and walk several row tiles (dense_small_multi_kernel);
            // (N/16) * gy workgroups, at most one per CU
            const int nrt = (int)((rows + 15) / 16);
            int gy = num_cus() / (p.N / 16);
            if (gy < 1) gy = 1;
            if (gy > nrt) gy = nrt;
            const dim3 mgrid((unsigned)(p.N / 16), (unsigned)gy);
#define ML_SMM(NS, RL, RS) hipLaunchKernelGGL((mlk::dense_small_multi_kernel<NS, RL, RS>), mgrid, dim3(mlk::SMALL_THREADS), 0, st, p, nrt)
#define ML_SMM_NS(NS)                                  \
    do {                                               \
        if (p.relu) {                                  \
            if (p.res) ML_SMM(NS, true, true);         \
            else ML_SMM(NS, true, false);              \
        } else {                                       \
            if (p.res) ML_SMM(NS, false, true);        \
            else ML_SMM(NS, false, false);             \
        }                                              \
    } while (0)
            if (precision == ML_PREC_F16X2) ML_SMM_NS(3);
            else ML_SMM_NS(1);
#undef ML_SMM_NS
#undef ML_SMM
            HIP_TRY(hipGetLastError());
            return ML_OK;
        }
        const bool t32 = rows > tu.small32_rows;
        const int T = t32 ? 32 : 16;
        const dim3 grid((unsigned)(p.N / T), (unsigned)((rows + T - 1) / T));
#define ML_SM(NS, RL, RS)                                                                                              \
    do {                                                                                                               \
        if (t32) hipLaunchKernelGGL((mlk::dense_small32_kernel<NS, RL, RS>), grid, dim3(mlk::SMALL_THREADS), 0, st, p); \
        else hipLaunchKernelGGL((mlk::dense_small_kernel<NS, RL, RS>), grid, dim3(mlk::SMALL_THREADS), 0, st, p);      \
    } while (0)
#define ML_SM_NS(NS)                                  \
    do {                                              \
        if (p.relu) {                                 \
            if (p.res) ML_SM(NS, true, true);         \
            else ML_SM(NS, true, false);              \
        } else {                                      \
            if (p.res) ML_SM(NS, false, true);        \
            else ML_SM(NS, false, false);             \
        }                                             \
    } while (0)
        if (precision == ML_PREC_F16X2) ML_SM_NS(3);
        else ML_SM_NS(1);
#undef ML_SM_NS
#undef ML_SM
        HIP_TRY(hipGetLastError());
        return ML_OK;
    }

    const int tiles = (p.M_pad / mlk::BM) * (p.N / mlk::BN);
    if (head_nh == -1 && !w4_runs(tu, p.K, 0)) return fail(ML_ERR_STATE, "fused aux head needs dense_kernel_w4");
    int grid = tiles < num_cus() ? tiles : num_cus();
    // dense_kernel_w4 for the long-K layers; the short input layer (K <= 128: two or four k-steps per tile, all epilogue)
    // and the layer with the fused output head stay on dense_kernel_pp, whose two waves per SIMD overlap those
    // VALU-heavy epilogues (measured: 0.063 vs 0.077 ms and 0.351 vs 0.381 ms per layer at 65536 rows)
    if (w4_runs(tu, p.K, head_nh)) {
#define ML_W4(NS, RL, RS, HD) \
    hipLaunchKernelGGL((mlk::dense_kernel_w4<NS, RL, RS, HD>), dim3(grid), dim3(mlk::W4_THREADS), 0, st, p)
#define ML_W4_NT(RL, RS, HD) \
    hipLaunchKernelGGL((mlk::dense_kernel_w4<3, RL, RS, HD, false, 4, true>), dim3(grid), dim3(mlk::W4_THREADS), 0, st, p)
#define ML_W4_NS(NS)                                              \
    do {                                                          \
        if (head_nh == -1) {                                      \
            if (p.relu && p.res && res_nt && NS == 3) ML_W4_NT(true, true, -1); \
            else if (p.relu && p.res) ML_W4(NS, true, true, -1);  \
            else if (!p.relu && !p.res) ML_W4(NS, false, false, -1); \
            else return fail(ML_ERR_STATE, "fused aux head: unsupported layer form"); \
        } else if (head_nh == 8) ML_W4(NS, true, false, 8);       \
        else if (head_nh == 9) ML_W4(NS, true, false, 9);         \
        else if (p.relu) {                                        \
            if (p.res && res_nt && NS == 3) ML_W4_NT(true, true, 0); \
            else if (p.res) ML_W4(NS, true, true, 0);             \
            else ML_W4(NS, true, false, 0);                       \
        } else {                                                  \
            if (p.res) ML_W4(NS, false, true, 0);                 \
            else ML_W4(NS, false, false, 0);                      \
        }                                                         \
    } while (0)
        // a residual matrix larger than the Infinity Cache (65536 rows x 1024: 256 MiB) is read once, long after it was written: non-temporal
        const bool res_nt = p.res && (size_t)p.M_pad * (size_t)p.N * 4 > ((size_t)192 << 20);
        if (precision == ML_PREC_F16X2) ML_W4_NS(3);
        else if (precision == ML_PREC_F16) ML_W4_NS(1);
        else ML_W4_NS(0);
#undef ML_W4_NS
#undef ML_W4_NT
#undef ML_W4
        HIP_TRY(hipGetLastError());
        return ML_OK;
    }
#define ML_PP(NS, RL, RS, HD) \
    hipLaunchKernelGGL((mlk::dense_kernel_pp<NS, RL, RS, HD>), dim3(grid), dim3(mlk::DENSE_THREADS), 0, st, p)
#define ML_PP_NS(NS)                                              \
    do {                                                          \
        if (head_nh == 8) ML_PP(NS, true, false, 8);              \
        else if (head_nh == 9) ML_PP(NS, true, false, 9);         \
        else if (p.relu) {                                        \
            if (p.res) ML_PP(NS, true, true, 0);                  \
            else ML_PP(NS, true, false, 0);                       \
        } else {                                                  \
            if (p.res) ML_PP(NS, false, true, 0);                 \
            else ML_PP(NS, false, false, 0);                      \
        }                                                         \
    } while (0)
    if (precision == ML_PREC_F16X2) ML_PP_NS(3);
    else if (precision == ML_PREC_F16) ML_PP_NS(1);
    else ML_PP_NS(0);  // ML_PREC_BF16
#undef ML_PP_NS
#undef ML_PP
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

// The heads kernels keep their weights in dynamic LDS (nh x hidden floats; heads_pair_kernel: both heads): beyond 64 KiB a kernel
// needs its limit raised once -- done at ml_loco_finalize for the widths this model will launch, not on the hot calls.
int set_head_lds_limits(const ml_loco* h) {
    auto raise = [](const void* fn, size_t lds) -> hipError_t {
        return lds > 65536 ? hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) : hipSuccess;
    };
    int pair = 0;
    for (const Head& hd : h->heads) {
        const size_t lds = (size_t)hd.nh * h->hidden * 4;
        switch (hd.nh) {
            case 1: HIP_TRY(raise((const void*)mlk::heads_kernel<1>, lds)); break;
            case 2: HIP_TRY(raise((const void*)mlk::heads_kernel<2>, lds)); break;
            case 8: HIP_TRY(raise((const void*)mlk::heads_kernel<8>, lds)); pair = 8; break;
            case 9: HIP_TRY(raise((const void*)mlk::heads_kernel<9>, lds)); pair = 9; break;
            case 10: HIP_TRY(raise((const void*)mlk::heads_kernel<10>, lds)); break;
            default: break;
        }
    }
    if (pair == 8) HIP_TRY(raise((const void*)mlk::heads_pair_kernel<8>, (size_t)9 * h->hidden * 4));
    if (pair == 9) HIP_TRY(raise((const void*)mlk::heads_pair_kernel<9>, (size_t)10 * h->hidden * 4));
    return ML_OK;
}

int launch_heads(const Head& hd, const char* act, int H, float* raw, int raw_stride, int64_t m,
                 hipStream_t st, int bf16 = 0) {
    const size_t lds = (size_t)hd.nh * H * 4;
    int64_t nquads = (m + 3) / 4;
    int grid = (int)((nquads + 3) / 4);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
#define ML_HEADS(NH)                                                                                 \
    case NH:                                                                                         \
        /* (dynamic LDS above 64 KiB: raised once at ml_loco_finalize, set_head_lds_limits) */         \
        hipLaunchKernelGGL(mlk::heads_kernel<NH>, dim3(grid), dim3(256), lds, st, act, H, hd.d_w,    \
                           hd.d_b, raw, raw_stride, hd.col0, m, bf16);                               \
        break;
    switch (hd.nh) {
        ML_HEADS(1)
        ML_HEADS(2)
        ML_HEADS(8)
        ML_HEADS(9)
        ML_HEADS(10)
        default:
            return fail(ML_ERR_SHAPE, "unsupported head width %d", hd.nh);
    }
#undef ML_HEADS
    HIP_TRY(hipGetLastError());
    return ML_OK;
}


// Runs the dense chain + heads on `rows` network rows whose line-format input already sits in
// buf[0]; leaves raw (rows, out_f) fp32 in raw_out.
// MC-dropout: active when mc_p > 0 -- dropout after dense layer 0 and on the input of the w_fin head
struct McPass {
    float p = 0.f;
    uint32_t seed = 0;   // seed of the first batched pass; pass k uses seed + k
    int64_t m_per = 0;   // persons per pass (rows = passes * m_per)
};

// The caller's post-process, handed to run_network so that the mono tile path can end in ONE launch (tail_mono_kernel:
// both head reductions + post_person) instead of head_reduce + aux_reduce + post; done says whether that happened.
struct TailMono {
    const float* centre;
    mlk::Kinv ki;
    const float* box_conf;
    float* out;
    float* xyzds;
    float* raw;   // the caller's raw buffer, or null (then the raw rows never leave the registers)
    bool done = false;
    // optional: the post_process geometry block (ml_post_geometry) of every row from these keypoints into geo_out, in the same
    // launch as the post-process when a single image's forward ends in one (geo_done says whether that happened)
    const float* geo_kps = nullptr;
    float* geo_out = nullptr;
    bool geo_done = false;
    // optional (round 6): the raw keypoints (m, 3, 17) when the input layer is to pre-process its own persons (mid_prep_runs): the
    // caller then launches no prep_kernel, the layer writes the box centres `centre` points at
    const float* prep_kps = nullptr;
    float* prep_centre = nullptr;
};

// ---------------------------------------------------------------------------------------------------------------------
// The route plan of one forward: WHICH launches a call of `rows` network rows consists of on this handle (its precision and
// tuning), decided in one place by plain predicates -- run_network below only executes it, ml_loco_plan prints it (so tests can
// assert the fusion state per row count), ml_loco_route reports its family.
//   per dense layer: the kernel family and what rides in its epilogue (the w_fin head's partial sums, the w_aux head's, none);
//   after which layer a head still needs its own launch; where the top-level MC-dropout masks are applied;
//   how the call ends: everything inside the last dense launches + tail_mono_kernel (both reductions + post-process), the
//   mid-size pair kernel, a single image's one-launch heads, or plain reductions / heads launches.
enum DenseFamily { FAM_SMALL16, FAM_SMALL32, FAM_MID64, FAM_MID128, FAM_HALF, FAM_W4, FAM_PP };
enum HeadsEnd { END_SEPARATE, END_SMALL_ONE, END_PAIR };

struct LayerStep {
    int family = FAM_PP;
    const Head* fused_fin = nullptr;   // w_fin (8 | 9 outputs) as partial sums out of this layer's epilogue
    const Head* fused_aux = nullptr;   // w_aux (1 output) as partial sums out of this layer's store epilogue
    bool dropout_after = false;        // MC-dropout: masks on this layer's output
    std::vector<const Head*> heads_after;   // heads launched on their own behind this layer (END_SEPARATE only)
};

struct RoutePlan {
    int route = ML_ROUTE_TILE;   // family of the long-K layers (ml_loco_route)
    bool small = false, mid = false;
    int mid_mode = 0;            // launch_dense's `mid` argument: 0 | 1 dense_mid_kernel | 2 with the half-size w4 tile
    int heads_end = END_SEPARATE;
    int nparts = 0;              // slices of partial sums per fused head (N / 128 from the tile kernels, N / 64 from dense_mid_kernel)
    const Head *pair_fin = nullptr, *pair_aux = nullptr;   // END_PAIR
    std::vector<LayerStep> layers;
};

int route_family(const ml_loco* h, int64_t rows) {
    if (use_small_path(h->tune, h->precision, rows))
        return (rows > h->tune.small32_rows && !small_multi_runs(h->tune, rows, h->hidden)) ? ML_ROUTE_SMALL32 : ML_ROUTE_SMALL16;
    if (use_mid_path(h->tune, h->precision, rows)) {
        const int64_t m_pad = round_up64(rows, 256);
        if (use_half_tile(h->tune, h->precision, rows)) return ML_ROUTE_HALF;
        return mid_tile_rows(h->tune, h->precision, (int)((m_pad / 128) * (h->hidden / mlk::MID_TN))) == 128 ? ML_ROUTE_MID128 : ML_ROUTE_MID64;
    }
    return ML_ROUTE_TILE;
}

// do `slices` slices of partial sums per head fit the workspace this call will run in (the current one, or the one ensure_rows
// allocates when the call is larger than it)?
bool part_fits(const ml_loco* h, int64_t rows, int slices) {
    const int64_t m_pad = round_up64(rows > 0 ? rows : 1, 256);
    return (int64_t)slices * m_pad <= (m_pad > h->cap_rows ? part_cells_for(h->hidden, m_pad) : h->part_cells);
}

RoutePlan make_plan(const ml_loco* h, int64_t rows, bool mc_on) {
    RoutePlan pl;
    pl.route = route_family(h, rows);
    pl.small = use_small_path(h->tune, h->precision, rows);   // decided on the whole call, not per chunk
    pl.mid = use_mid_path(h->tune, h->precision, rows);       // (heads as their own launches, like the small path)
    pl.mid_mode = pl.mid ? (use_half_tile(h->tune, h->precision, rows) ? 2 : 1) : 0;
    const bool tile = !pl.small && !pl.mid;
    // may this layer carry a head in its epilogue?  The persistent tile kernels, and (round 5) dense_kernel_w4's half-size tile in
    // the upper mid window when Tuning::half_heads is on
    // (the mid window as a whole: Tuning::half_heads; the partial sums come in 128-column slices from the w4 tiles, in 64-column
    // slices from dense_mid_kernel -- both heads of a call must agree, and the workspace must hold that many slices)
    auto half_layer = [&](const DenseLayer& L) {
        return pl.mid_mode == 2 && h->precision == ML_PREC_F16X2 && L.kpad > 128 && L.kpad % 64 == 0 && L.n % 256 == 0;
    };
    auto slices_of = [&](const DenseLayer& L) { return (tile || half_layer(L)) ? L.n / 128 : L.n / 64; };
    auto fusable = [&](const DenseLayer& L) {
        return tile || (pl.mid && h->tune.half_heads && h->precision == ML_PREC_F16X2 && L.n % 128 == 0 && part_fits(h, rows, slices_of(L)));
    };
    pl.layers.resize(h->layers.size());
    int n_fused = 0;
    bool slices_differ = false;
    for (size_t li = 0; li < h->layers.size(); ++li) {
        const DenseLayer& L = h->layers[li];
        LayerStep& s = pl.layers[li];
        // the w_fin head rides in the epilogue of the layer that feeds it (persistent kernel, relu, no residual)
        for (const Head& hd : h->heads)
            if (hd.after_layer == (int)li && (hd.nh == 8 || hd.nh == 9) && L.relu && L.res < 0 && !mc_on && fusable(L)) s.fused_fin = &hd;
        // the one-output w_aux head rides in the store epilogue of the layer that produces its input (w4 kernel: residual+relu
        // layer when w3*w2 are merged, the plain w2 layer otherwise); its partials have their own region behind the w_fin head's
        if (!s.fused_fin && fusable(L) && !mc_on && (pl.mid || w4_runs(h->tune, L.kpad, 0)) && ((L.relu && L.res >= 0) || (!L.relu && L.res < 0)))
            for (const Head& hd : h->heads)
                if (hd.after_layer == (int)li && hd.nh == 1 && hd.src == L.dst) s.fused_aux = &hd;
        n_fused += (s.fused_fin ? 1 : 0) + (s.fused_aux ? 1 : 0);
        if (s.fused_fin || s.fused_aux) {
            if (pl.nparts && pl.nparts != slices_of(L)) slices_differ = true;
            pl.nparts = slices_of(L);
        }
    }
    // the mid window fuses every head or none: a single fused head would leave the pair kernel half a job
    if (pl.mid && (n_fused != (int)h->heads.size() || slices_differ)) {
        for (LayerStep& s : pl.layers) s.fused_fin = s.fused_aux = nullptr;
        n_fused = 0;
    }
    if (!pl.nparts || !n_fused) pl.nparts = 2 * h->hidden / 256;
    const bool mid_fused = pl.mid && n_fused == (int)h->heads.size() && n_fused > 0;
    // how the call ends
    if (pl.mid && !mid_fused && !mc_on && h->heads.size() == 2 && h->precision != ML_PREC_BF16) {
        // the mid-size path: both heads (+ the mono post-process) in one launch behind the last layer, when the model has the
        // LocoModel pair (w_fin: 8 | 9 outputs in columns 0.., w_aux: the last column)
        const Head *pf = nullptr, *pa = nullptr;
        for (const Head& hd : h->heads) {
            if ((hd.nh == 8 || hd.nh == 9) && hd.col0 == 0) pf = &hd;
            else if (hd.nh == 1) pa = &hd;
        }
        if (pf && pa && pa->col0 == pf->nh && h->out_f == pf->nh + 1) {
            pl.heads_end = END_PAIR;
            pl.pair_fin = pf;
            pl.pair_aux = pa;
        }
    }
    // a single image's worth of rows: all heads in ONE launch after the last dense layer (their source buffers are both still
    // intact there), one workgroup per row
    if (pl.heads_end == END_SEPARATE && pl.small && rows <= 128 && !mc_on && h->heads.size() <= 2) pl.heads_end = END_SMALL_ONE;
    for (size_t li = 0; li < h->layers.size(); ++li) {
        const DenseLayer& L = h->layers[li];
        LayerStep& s = pl.layers[li];
        // the kernel family launch_dense picks for this layer
        if (pl.mid) {
            const bool half = pl.mid_mode == 2 && h->precision == ML_PREC_F16X2 && L.kpad > 128 && L.kpad % 64 == 0 && L.n % 256 == 0;
            s.family = half ? FAM_HALF : (pl.route == ML_ROUTE_MID64 ? FAM_MID64 : (pl.route == ML_ROUTE_MID128 ? FAM_MID128 :
                       ((round_up64(rows, 256) / 128) * (L.n / mlk::MID_TN) >= num_cus() ? FAM_MID128 : FAM_MID64)));
        } else if (pl.small) {
            s.family = (rows > h->tune.small32_rows && !small_multi_runs(h->tune, rows, L.kpad)) ? FAM_SMALL32 : FAM_SMALL16;
        } else {
            s.family = w4_runs(h->tune, L.kpad, s.fused_fin ? s.fused_fin->nh : (s.fused_aux ? -1 : 0)) ? FAM_W4 : FAM_PP;
        }
        // top-level dropout sites only (reference net.py:141): after relu(bn1) = output of layer 0, and after relu(bn3) = the
        // input of the w_fin head
        if (mc_on) {
            s.dropout_after = (li == 0);
            for (const Head& hd : h->heads)
                if (hd.after_layer == (int)li && hd.mc_site) s.dropout_after = true;
        }
        if (pl.heads_end == END_SEPARATE)
            for (const Head& hd : h->heads)
                if (hd.after_layer == (int)li && &hd != s.fused_fin && &hd != s.fused_aux) s.heads_after.push_back(&hd);
    }
    return pl;
}

// "route=tile; L0 pp; L1 w4; ...; L6 w4+aux; L7 pp+fin8; end=tail_mono" -- the plan as text (ml_loco_plan)
std::string plan_text(const ml_loco* h, const RoutePlan& pl, bool with_tail) {
    static const char* fam[] = {"small16", "small32", "mid64", "mid128", "half", "w4", "pp"};
    static const char* rt[] = {"small16", "small32", "mid64", "mid128", "half", "tile"};
    std::string s = std::string("route=") + rt[pl.route];
    char tmp[64];
    const Head *fin = nullptr, *aux = nullptr;
    for (size_t li = 0; li < pl.layers.size(); ++li) {
        const LayerStep& st = pl.layers[li];
        snprintf(tmp, sizeof(tmp), "; L%zu %s", li, fam[st.family]);
        s += tmp;
        if (st.fused_fin) {
            snprintf(tmp, sizeof(tmp), "+fin%d", st.fused_fin->nh);
            s += tmp;
            fin = st.fused_fin;
        }
        if (st.fused_aux) {
            s += "+aux";
            aux = st.fused_aux;
        }
        if (st.dropout_after) s += "+dropout";
        for (const Head* hd : st.heads_after) {
            snprintf(tmp, sizeof(tmp), " heads%d", hd->nh);
            s += tmp;
        }
    }
    const bool fuse_tail = fin && aux && fin->col0 == 0 && aux->col0 == fin->nh && h->out_f == fin->nh + 1;
    if (pl.heads_end == END_PAIR) s += with_tail ? "; end=heads_pair+post" : "; end=heads_pair";
    else if (pl.heads_end == END_SMALL_ONE) s += with_tail ? "; end=heads_small+post" : "; end=heads_small";
    else if (fin || aux) s += (with_tail && fuse_tail) ? "; end=tail_mono" : "; end=reduce";
    else s += "; end=heads";
    return s;
}

}  // namespace

// Plans are built once per (rows, MC-dropout, tuning version, workspace) and kept: a stream of calls with recurring row counts
// (a video: a few distinct person counts; a batch job: one) walks a stored plan and builds nothing.
struct PlanCache {
    struct Entry {
        int64_t rows, cap_rows;
        int mc_on, version;
        int64_t part_cells;
        RoutePlan plan;
    };
    std::vector<Entry> entries;
    size_t next = 0;   // FIFO replacement once 16 plans are stored
};

namespace {

const RoutePlan& plan_for(ml_loco* h, int64_t rows, bool mc_on) {
    if (!h->plans) h->plans = new PlanCache();
    for (const PlanCache::Entry& e : h->plans->entries)
        if (e.rows == rows && e.mc_on == (int)mc_on && e.version == h->tune_version && e.cap_rows == h->cap_rows &&
            e.part_cells == h->part_cells)
            return e.plan;
    PlanCache::Entry e{rows, h->cap_rows, (int)mc_on, h->tune_version, h->part_cells, make_plan(h, rows, mc_on)};
    if (h->plans->entries.size() < 16) {
        h->plans->entries.push_back(std::move(e));
        return h->plans->entries.back().plan;
    }
    PlanCache::Entry& slot = h->plans->entries[h->plans->next];
    h->plans->next = (h->plans->next + 1) % 16;
    slot = std::move(e);
    return slot.plan;
}

int run_network(ml_loco* h, int64_t rows, float* raw_out, hipStream_t st, McPass mc = McPass(), TailMono* tail = nullptr) {
    const int64_t m_pad_all = round_up64(rows, 256);
    // (row chunking is an inference experiment knob; the batched MC-dropout passes index their masks by global row)
    const int64_t chunk_rows = h->tune.chunk_rows / 256 * 256;
    const int64_t chunk = (chunk_rows > 0 && mc.p <= 0.f) ? chunk_rows : m_pad_all;
    const RoutePlan& pl = plan_for(h, rows, mc.p > 0.f);
    const bool defer = tail && chunk == m_pad_all;   // head reductions wait for the end of the (single) chunk
    const Head *def_fin = nullptr, *def_aux = nullptr;
    const int nparts = pl.nparts;
    if (h->precision == ML_PREC_BF16) {
        // the pre-process kernels write fp16 hi|lo lines; the bf16 comparison mode re-rounds them once (hi + lo -> one
        // bf16 in the hi slot, 16 B per row chunk; 17 MB at 65536 rows -- < 0.5 % of a step, counted in its time)
        if (mc.p > 0.f) return fail(ML_ERR_ARG, "MC-dropout is not available in the bf16 comparison mode");
        const int64_t pairs = m_pad_all * (int64_t)(h->k0pad / 32) * 4;
        hipLaunchKernelGGL(mlk::lines_to_bf16_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, h->buf[0], pairs);
        HIP_TRY(hipGetLastError());
    }
    for (int64_t r0 = 0; r0 < m_pad_all; r0 += chunk) {
        const int64_t m_pad = (m_pad_all - r0 < chunk) ? (m_pad_all - r0) : chunk;
        const int64_t rows_here = (rows - r0 < m_pad) ? (rows - r0) : m_pad;
        auto at = [&](int b) { return h->buf[b] + r0 * (int64_t)(b == 0 ? h->k0pad : h->hidden) * 4; };
        for (size_t li = 0; li < h->layers.size(); ++li) {
            const DenseLayer& L = h->layers[li];
            const LayerStep& step = pl.layers[li];
            const bool last = li + 1 == h->layers.size();
            mlk::DenseParams p;
            p.x = at(L.src);
            p.w = L.d_w;
            p.bias = L.d_b;
            p.bias_scaled = L.d_bs;
            p.res = L.res >= 0 ? at(L.res) : nullptr;
            p.y = at(L.dst);
            p.descale = std::ldexp(1.0f, -L.scale_pow2);
            p.descale_ptr = nullptr;
            p.M_pad = (int)m_pad;
            p.N = L.n;
            p.K = L.kpad;
            p.relu = L.relu;
            p.debug = 0;
            p.trace = nullptr;
            p.head_w = nullptr;
            p.head_part = nullptr;
            p.kpart = h->d_kpart;          // (launch_dense decides whether this layer's reduction is split)
            p.kcount = h->d_kcount;
            p.ksplit = (int)(h->kpart_floats / ((int64_t)m_pad * L.n > 0 ? (int64_t)m_pad * L.n : 1));   // k ranges the workspace can hold
            if (li == 0 && tail && tail->prep_kps) {   // (forward_mono_impl checked mid_prep_runs: the whole call is one chunk on the mid path)
                p.prep_kps = tail->prep_kps;
                p.prep_centre = tail->prep_centre;
                for (int q = 0; q < 6; ++q) p.prep_kinv[q] = tail->ki.k[q];
                p.prep_z = 10.0f;
                p.prep_m = (int)rows;
            }
            if (step.fused_fin) {
                p.head_w = step.fused_fin->d_w;
                p.head_part = h->d_part + r0 * (int64_t)nparts * 16;
            } else if (step.fused_aux) {
                p.head_w = step.fused_aux->d_w;
                p.head_part = h->d_part_aux + r0 * (int64_t)nparts;
            }
            const bool timed = h->profiling && (h->ev_used + 1) * 2 <= h->ev_pool.size();
            if (timed) HIP_TRY(hipEventRecord(h->ev_pool[h->ev_used * 2], st));
            int rc = launch_dense(h->tune, h->precision, p, st, step.fused_fin ? step.fused_fin->nh : (step.fused_aux ? -1 : 0),
                                  pl.small ? rows_here : -1, pl.mid_mode);
            if (rc) return rc;
            if (timed) {
                HIP_TRY(hipEventRecord(h->ev_pool[h->ev_used * 2 + 1], st));
                h->ev_layer[h->ev_used] = (int)li;
                h->ev_used++;
            }
            if (rows_here <= 0) continue;
            // partial sums of a fused head: reduced at the end of the call (one chunk + a tail), or right here
            if (step.fused_fin && defer) def_fin = step.fused_fin;
            else if (step.fused_fin) {
                hipLaunchKernelGGL(mlk::head_reduce_kernel, dim3((unsigned)((rows_here * 16 + 255) / 256)), dim3(256), 0, st,
                                   (const float*)(h->d_part + r0 * (int64_t)nparts * 16), nparts, m_pad, rows_here, step.fused_fin->nh,
                                   (const float*)step.fused_fin->d_b, raw_out + r0 * h->out_f, h->out_f, step.fused_fin->col0);
                HIP_TRY(hipGetLastError());
            }
            if (step.fused_aux && defer) def_aux = step.fused_aux;
            else if (step.fused_aux) {
                hipLaunchKernelGGL(mlk::aux_reduce_kernel, dim3((unsigned)((rows_here + 255) / 256)), dim3(256), 0, st,
                                   (const float*)(h->d_part_aux + r0 * (int64_t)nparts), nparts, m_pad, rows_here,
                                   (const float*)step.fused_aux->d_b, raw_out + r0 * h->out_f, h->out_f, step.fused_aux->col0);
                HIP_TRY(hipGetLastError());
            }
            if (step.dropout_after) {
                const int64_t groups = m_pad * (L.n / 8);
                hipLaunchKernelGGL(mlk::dropout_lines_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, at(L.dst), m_pad,
                                   L.n, mc.p, mc.seed, (uint32_t)li, mc.m_per > 0 ? mc.m_per : m_pad_all);
                HIP_TRY(hipGetLastError());
            }
            if (pl.heads_end == END_PAIR && last) {
                const Head *pf = pl.pair_fin, *pa = pl.pair_aux;
                const size_t lds = (size_t)(pf->nh + 1) * h->hidden * 4;
                int grid = (int)((((rows_here + 1) / 2) + 3) / 4);   // 4 waves per workgroup, 2 rows per wave and pass (heads_pair_kernel's RW)
                if (grid > 2048) grid = 2048;
                const bool with_post = defer;
                float* raw_dst = with_post ? tail->raw : raw_out + r0 * h->out_f;
#define ML_PAIR(NH)                                                                                                          \
    hipLaunchKernelGGL(mlk::heads_pair_kernel<NH>, dim3(grid), dim3(256), lds, st, (const char*)at(pf->src),                 \
                       (const char*)at(pa->src), h->hidden, (const float*)pf->d_w, (const float*)pf->d_b,                    \
                       (const float*)pa->d_w, (const float*)pa->d_b, raw_dst, rows_here,                                     \
                       with_post ? tail->centre : (const float*)nullptr, with_post ? tail->ki : mlk::Kinv{},                 \
                       with_post ? tail->box_conf : (const float*)nullptr, with_post ? tail->out : (float*)nullptr,          \
                       with_post ? tail->xyzds : (float*)nullptr)
                if (pf->nh == 8) ML_PAIR(8);   // (dynamic LDS above 64 KiB: the attribute was set once, at ml_loco_finalize)
                else ML_PAIR(9);
#undef ML_PAIR
                HIP_TRY(hipGetLastError());
                if (with_post) tail->done = true;
            } else if (pl.heads_end == END_SMALL_ONE && last) {
                mlk::SmallHeads hp = {};
                int k = 0;
                for (const Head& hd : h->heads) {
                    hp.act[k] = at(hd.src);
                    hp.w[k] = hd.d_w;
                    hp.b[k] = hd.d_b;
                    hp.nh[k] = hd.nh;
                    hp.col0[k] = hd.col0;
                    ++k;
                }
                // (a mono forward of a single image ends here: the post-process rides in the same launch)
                const bool with_post = defer && h->out_f <= 16 && hp.col0[0] + hp.nh[0] <= 16 && hp.col0[1] + hp.nh[1] <= 16;
                if (with_post) {
                    mlk::FrameDone fd;
                    if (h->frame_flag_req && tail->geo_out && h->h_done && h->d_arrive) {   // this launch ends a frame: arm the host's flag
                        fd.arrive = h->d_arrive;
                        fd.flag = h->h_done;
                        fd.seq = ++h->done_seq;
                        h->frame_flag_armed = true;
                    }
                    hipLaunchKernelGGL(mlk::heads_small_kernel, dim3((unsigned)rows_here), dim3(256), 0, st, hp, h->hidden, tail->raw,
                                       h->out_f, rows_here, tail->centre, tail->ki, tail->box_conf, tail->out, tail->xyzds,
                                       tail->geo_kps, tail->geo_out, fd);
                    tail->done = true;
                    tail->geo_done = tail->geo_out != nullptr;
                } else {
                    hipLaunchKernelGGL(mlk::heads_small_kernel, dim3((unsigned)rows_here), dim3(256), 0, st, hp, h->hidden,
                                       raw_out + r0 * h->out_f, h->out_f, rows_here, (const float*)nullptr, mlk::Kinv{},
                                       (const float*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)nullptr, (float*)nullptr,
                                       mlk::FrameDone());
                }
                HIP_TRY(hipGetLastError());
            }
            for (const Head* hd : step.heads_after) {
                rc = launch_heads(*hd, at(hd->src), h->hidden, raw_out + r0 * h->out_f, h->out_f, rows_here, st, h->precision == ML_PREC_BF16);
                if (rc) return rc;
            }
        }
    }
    if (defer && rows > 0) {
        const bool fuse_tail = def_fin && def_aux && def_fin->col0 == 0 && def_aux->col0 == def_fin->nh && h->out_f == def_fin->nh + 1 &&
                               (def_fin->nh == 8 || def_fin->nh == 9);
        const dim3 grid((unsigned)((rows + 255) / 256)), block(256);
        if (fuse_tail) {
            if (def_fin->nh == 8)
                hipLaunchKernelGGL(mlk::tail_mono_kernel<8>, grid, block, 0, st, (const float*)h->d_part, (const float*)h->d_part_aux, nparts,
                                   m_pad_all, rows, (const float*)def_fin->d_b, (const float*)def_aux->d_b, tail->raw, tail->centre, tail->ki,
                                   tail->box_conf, tail->out, tail->xyzds);
            else
                hipLaunchKernelGGL(mlk::tail_mono_kernel<9>, grid, block, 0, st, (const float*)h->d_part, (const float*)h->d_part_aux, nparts,
                                   m_pad_all, rows, (const float*)def_fin->d_b, (const float*)def_aux->d_b, tail->raw, tail->centre, tail->ki,
                                   tail->box_conf, tail->out, tail->xyzds);
            tail->done = true;
        } else {   // only one of the heads was fused into a dense epilogue: the ordinary reductions, the caller post-processes
            if (def_fin)
                hipLaunchKernelGGL(mlk::head_reduce_kernel, dim3((unsigned)((rows * 16 + 255) / 256)), block, 0, st, (const float*)h->d_part,
                                   nparts, m_pad_all, rows, def_fin->nh, (const float*)def_fin->d_b, raw_out, h->out_f, def_fin->col0);
            if (def_aux)
                hipLaunchKernelGGL(mlk::aux_reduce_kernel, grid, block, 0, st, (const float*)h->d_part_aux, nparts, m_pad_all, rows,
                                   (const float*)def_aux->d_b, raw_out, h->out_f, def_aux->col0);
        }
        HIP_TRY(hipGetLastError());
    }
    return ML_OK;
}

int check_ready(const ml_loco* h) {
    if (!h) return fail(ML_ERR_ARG, "null model handle");
    if (!h->finalized) return fail(ML_ERR_STATE, "model not finalized");
    if (h->host_only) return fail(ML_ERR_STATE, "model was finalized host-only (ML_FLAG_HOST_ONLY): it cannot run");
    return ML_OK;
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

int ml_version(void) { return 100; }

const char* ml_last_error(void) { return g_err; }

int ml_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        fail(ML_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
        return -ML_ERR_HIP;
    }
    return n;
}

int ml_loco_create(int in_features, int hidden, int out_features, int num_stage, ml_loco** out) {
    if (!out) return fail(ML_ERR_ARG, "out is null");
    if (in_features <= 0 || in_features > 1024) return fail(ML_ERR_SHAPE, "in_features %d unsupported", in_features);
    if (hidden <= 0 || hidden > 4096) return fail(ML_ERR_SHAPE, "hidden size %d unsupported (1 .. 4096)", hidden);
    if (out_features < 2 || out_features > 11) return fail(ML_ERR_SHAPE, "out_features %d unsupported", out_features);
    if (num_stage < 0 || num_stage > 16) return fail(ML_ERR_SHAPE, "num_stage %d unsupported", num_stage);
    ml_loco* h = new (std::nothrow) ml_loco();
    if (!h) return fail(ML_ERR_HIP, "out of host memory");
    h->in_f = in_features;
    h->hidden_real = hidden;
    h->hidden = round_up(hidden, 256);  // any linear_size (reference architectures.py:14): zero-padded to the tile
    h->out_f = out_features;
    h->num_stage = num_stage;
    *out = h;
    return ML_OK;
}

int ml_loco_set_tensor(ml_loco* h, const char* key, const float* host_data, int64_t numel) {
    if (!h || !key) return fail(ML_ERR_ARG, "null argument");
    if (h->finalized) return fail(ML_ERR_STATE, "model already finalized");
    const std::string k(key);
    const std::string nbt = "num_batches_tracked";
    if (k.size() >= nbt.size() && k.compare(k.size() - nbt.size(), nbt.size(), nbt) == 0) return ML_OK;
    if (!host_data || numel <= 0) return fail(ML_ERR_ARG, "tensor '%s': null data or bad size", key);
    h->tensors[k].assign(host_data, host_data + numel);
    return ML_OK;
}

int ml_loco_finalize(ml_loco* h, int precision, int flags) {
    if (!h) return fail(ML_ERR_ARG, "null model handle");
    if (h->finalized) return fail(ML_ERR_STATE, "model already finalized");
    if (precision != ML_PREC_F16X2 && precision != ML_PREC_F16 && precision != ML_PREC_BF16)
        return fail(ML_ERR_ARG, "unknown precision %d", precision);
    h->precision = precision;
    h->flags = flags;
    h->host_only = (flags & ML_FLAG_HOST_ONLY) != 0;
    // where the small-row kernels hand over to dense_mid_kernel (round 6, tools/ab_small_mid.py): at the headline width the LDS-DMA loader
    // with two k ranges per tile takes 320-512 rows in 106 us per forward against 159-162 us on the 32 x 32 tiles (256 rows: 104 vs 105);
    // narrower models have too few 128-column tiles for that and keep the measured round-3 boundary
    if (precision == ML_PREC_F16X2 && h->hidden >= 1024) h->tune.small_rows = 256;
    if (!h->host_only) HIP_TRY(hipGetDevice(&h->device));
    const int H = h->hidden_real, HP = h->hidden, IN = h->in_f;
    const int NFIN = h->out_f - 1;  // w_fin rows; the aux head adds the last column (architectures.py:12,70)
    std::vector<double> W, B;
    int rc;
    // w1 + batch_norm1 + relu (architectures.py:50-53): input lines (buf0) -> A (buf1)
    if ((rc = fold(h, "w1", "batch_norm1", H, IN, W, B))) return rc;
    add_dense(h, W, B, H, IN, HP, IN, 1, 0, 1, -1);
    // residual stages (architectures.py:88-102): A -> B -> A (+A)
    for (int s = 0; s < h->num_stage; ++s) {
        const std::string p = "linear_stages." + std::to_string(s) + ".";
        if ((rc = fold(h, p + "w1", p + "batch_norm1", H, H, W, B))) return rc;
        add_dense(h, W, B, H, H, HP, HP, 1, 1, 2, -1);
        if ((rc = fold(h, p + "w2", p + "batch_norm2", H, H, W, B))) return rc;
        add_dense(h, W, B, H, H, HP, HP, 1, 2, 1, 1);
    }
    h->legacy = h->tensors.count("w_fin.weight") == 0 && h->tensors.count("w3.weight") == 0;
    if (h->legacy) {
        // MonolocoModel (architectures.py:105-145): the stages are followed by one Linear w2: hidden -> out_f,
        // a GEMV-shaped head on a3 (buffer A)
        const auto* w2 = get_t(h, "w2.weight", (int64_t)h->out_f * H);
        const auto* b2 = get_t(h, "w2.bias", h->out_f);
        if (!w2 || !b2) return ML_ERR_STATE;
        if (h->out_f != 2 && h->out_f != 9) return fail(ML_ERR_SHAPE, "legacy MonolocoModel: output size %d unsupported (2 or 9)", h->out_f);
        Head out;
        out.nh = h->out_f;
        out.col0 = 0;
        out.src = 1;
        out.after_layer = (int)h->layers.size() - 1;
        out.w = pad_head(w2->data(), h->out_f, H, HP);
        out.b.assign(b2->begin(), b2->end());
        h->heads.push_back(std::move(out));
    } else {
    std::vector<double> W2, B2, W3, B3, WA, BA;
    if ((rc = fold(h, "w2", "", H, H, W2, B2))) return rc;
    if ((rc = fold(h, "w3", "batch_norm3", H, H, W3, B3))) return rc;
    if ((rc = fold(h, "w_aux", "", 1, H, WA, BA))) return rc;
    const auto* wf = get_t(h, "w_fin.weight", (int64_t)NFIN * H);
    const auto* bf = get_t(h, "w_fin.bias", NFIN);
    if (!wf || !bf) return ML_ERR_STATE;
    Head aux, fin;
    aux.nh = 1;
    aux.col0 = NFIN;
    fin.nh = NFIN;
    fin.col0 = 0;
    fin.w = pad_head(wf->data(), NFIN, H, HP);
    fin.b.assign(bf->begin(), bf->end());
    if (flags & ML_FLAG_MERGE_W2W3) {
        // y3 = relu(bn3(w3(w2 a + b2) + b3)) = relu((W3' W2) a + (W3' b2 + b3'));  aux = (wa W2) a + (wa b2 + ba)
        std::vector<double> W32, B32(H), WAM, BAM(1);
        matmul64(W3, W2, H, H, H, W32);
        for (int i = 0; i < H; ++i) {
            double s = B3[i];
            for (int j = 0; j < H; ++j) s += W3[(size_t)i * H + j] * B2[j];
            B32[i] = s;
        }
        matmul64(WA, W2, 1, H, H, WAM);
        double s = BA[0];
        for (int j = 0; j < H; ++j) s += WA[j] * B2[j];
        BAM[0] = s;
        aux.w.assign(HP, 0.0f);
        for (int j = 0; j < H; ++j) aux.w[j] = (float)WAM[j];
        aux.b.assign(1, (float)BAM[0]);
        aux.src = 1;  // reads a3 (buffer A) once the last stage is done
        aux.after_layer = (int)h->layers.size() - 1;
        add_dense(h, W32, B32, H, H, HP, HP, 1, 1, 2, -1);  // A -> B
        fin.src = 2;
        fin.after_layer = (int)h->layers.size() - 1;
    } else {
        add_dense(h, W2, B2, H, H, HP, HP, 0, 1, 2, -1);  // y2 = w2 a      A -> B   (architectures.py:59)
        aux.w.assign(HP, 0.0f);
        for (int j = 0; j < H; ++j) aux.w[j] = (float)WA[j];
        aux.b.assign(1, (float)BA[0]);
        aux.src = 2;
        aux.after_layer = (int)h->layers.size() - 1;
        add_dense(h, W3, B3, H, H, HP, HP, 1, 2, 1, -1);  // y3 = relu(bn3(w3 y2))  B -> A
        fin.src = 1;
        fin.after_layer = (int)h->layers.size() - 1;
    }
    fin.mc_site = true;  // net.py:141 re-enables the top-level dropout, which sits in front of w_fin
    h->heads.push_back(std::move(aux));
    h->heads.push_back(std::move(fin));
    }
    h->k0pad = h->layers[0].kpad;
    for (auto& L : h->layers) pack_layer(h->precision, L);
    h->tensors.clear();
    if (h->host_only) {
        h->finalized = true;
        return ML_OK;
    }
    for (auto& L : h->layers)
        if ((rc = upload_layer(h, L))) return rc;
    for (auto& hd : h->heads) {
        if ((rc = dev_alloc(h, &hd.d_w, (int64_t)hd.w.size() * 4))) return rc;
        if ((rc = dev_alloc(h, &hd.d_b, (int64_t)hd.b.size() * 4))) return rc;
        HIP_TRY(hipMemcpy(hd.d_w, hd.w.data(), hd.w.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(hd.d_b, hd.b.data(), hd.b.size() * 4, hipMemcpyHostToDevice));
    }
    h->tensors.clear();
    if ((rc = set_head_lds_limits(h))) return rc;
    // ml_loco_frame_mono's completion word (pinned, coherent: the host polls it while the last launch of a frame writes it) and the
    // arrival counter of that launch's workgroups; without them a frame simply ends in hipStreamSynchronize
    if (hipHostMalloc((void**)&h->h_done, 64, hipHostMallocCoherent) == hipSuccess) {
        *h->h_done = 0;
        if (hipMalloc((void**)&h->d_arrive, 64) != hipSuccess || hipMemset(h->d_arrive, 0, 64) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostFree(h->h_done);
            h->h_done = nullptr;
            h->d_arrive = nullptr;
        }
    } else {
        (void)hipGetLastError();
        h->h_done = nullptr;
    }
    h->finalized = true;
    return ML_OK;
}

int ml_loco_reserve(ml_loco* h, int64_t max_rows) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (max_rows <= 0) return fail(ML_ERR_ARG, "max_rows must be positive");
    return ensure_rows(h, max_rows);
}

int ml_loco_profile_begin(ml_loco* h, int max_launches) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (max_launches <= 0 || max_launches > (1 << 20)) return fail(ML_ERR_ARG, "max_launches out of range");
    while (h->ev_pool.size() < (size_t)max_launches * 2) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        h->ev_pool.push_back(e);
    }
    h->ev_layer.assign(h->ev_pool.size() / 2, -1);
    h->ev_used = 0;
    h->profiling = true;
    return ML_OK;
}

int ml_loco_profile_end(ml_loco* h, int64_t* launches, double* total_ms, double* per_layer_ms, int64_t* per_layer_n,
                        int n_layers) {
    int rc = check_ready(h);
    if (rc) return rc;
    h->profiling = false;
    double tot = 0.0;
    for (int i = 0; i < n_layers; ++i) {
        if (per_layer_ms) per_layer_ms[i] = 0.0;
        if (per_layer_n) per_layer_n[i] = 0;
    }
    for (size_t i = 0; i < h->ev_used; ++i) {
        HIP_TRY(hipEventSynchronize(h->ev_pool[i * 2 + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev_pool[i * 2], h->ev_pool[i * 2 + 1]));
        tot += ms;
        const int li = h->ev_layer[i];
        if (li >= 0 && li < n_layers) {
            if (per_layer_ms) per_layer_ms[li] += ms;
            if (per_layer_n) per_layer_n[li] += 1;
        }
    }
    if (launches) *launches = (int64_t)h->ev_used;
    if (total_ms) *total_ms = tot;
    h->ev_used = 0;
    return ML_OK;
}

int ml_loco_destroy(ml_loco* h) {
    if (!h) return ML_OK;
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    free_workspace(h);
    dev_free(h->d_xl);
    dev_free(h->d_xr);
    dev_free(h->d_cl);
    for (auto& L : h->layers) {
        dev_free(L.d_w);
        dev_free(L.d_b);
        dev_free(L.d_bs);
    }
    for (auto& hd : h->heads) {
        dev_free(hd.d_w);
        dev_free(hd.d_b);
    }
    if (h->h_done) (void)hipHostFree(h->h_done);
    dev_free(h->d_arrive);
    delete h->plans;
    delete h;
    return ML_OK;
}

int64_t ml_loco_device_bytes(const ml_loco* h) { return h ? h->dev_bytes : 0; }

// prep_kernel with the workgroup size in persons chosen from the rows it has to cover (the persons, or the zero-filled
// line rows behind them): 32 per workgroup while that still leaves CUs idle, 256 for the big batches
static int launch_prep(hipStream_t st, const float* kps, int64_t m, const mlk::Kinv& ki, float z_met, float* x_f32, float* centre,
                char* lines, int kpad, int64_t fill_rows, int zero_center) {
    const int64_t cover = (lines && fill_rows > m) ? fill_rows : m;
    if (cover <= 0) return ML_OK;
    if (cover <= 8192)
        hipLaunchKernelGGL(mlk::prep_kernel<32>, dim3((unsigned)((cover + 31) / 32)), dim3(256), 0, st, kps, m, ki, z_met, x_f32, centre,
                           lines, kpad, fill_rows, zero_center);
    else   // 64 persons per workgroup (round 4; 256 before: one workgroup per CU at 65536 persons, its load / compute / store phases
           // in sequence with nothing to overlap them: 24.6 us = 1.25 TB/s; with 22 KiB of LDS several workgroups share a CU)
        hipLaunchKernelGGL(mlk::prep_kernel<64>, dim3((unsigned)((cover + 63) / 64)), dim3(256), 0, st, kps, m, ki, z_met, x_f32,
                           centre, lines, kpad, fill_rows, zero_center);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

// ---------------------------------------------------------------- stand-alone geometry
int ml_preprocess_mono(const float* kps_dev, int64_t m, const float* kinv_host, float z_met, int zero_center,
                       float* x_dev, float* centre_dev, void* stream) {
    if (m == 0) return ML_OK;
    if (m < 0 || !kinv_host || !kps_dev) return fail(ML_ERR_ARG, "bad argument");
    return launch_prep((hipStream_t)stream, kps_dev, m, make_kinv(kinv_host), z_met, x_dev, centre_dev, (char*)nullptr, 0, m, zero_center);
}

int ml_stereo_pairs(const float* xl_dev, int64_t ml, const float* xr_dev, int64_t mr, float* rows_dev,
                    void* stream) {
    if (ml == 0 || mr == 0) return ML_OK;
    if (ml < 0 || mr < 0 || !rows_dev || !xl_dev || !xr_dev) return fail(ML_ERR_ARG, "bad argument");
    const int64_t total = ml * mr * 2 * mlk::NIN;
    const int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(mlk::pairs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, xl_dev, ml, xr_dev, mr,
                       rows_dev, (char*)nullptr, 0, (int64_t)0);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

int ml_extract_outputs(const float* raw_dev, int out_features, const int32_t* row_index_dev, int64_t m,
                       const float* centre_dev, const float* kinv_host, const float* box_conf_dev,
                       float* out_dev, float* xyzds_dev, void* stream) {
    if (out_features != 9 && out_features != 10) return fail(ML_ERR_SHAPE, "out_features must be 9 or 10");
    if (m == 0) return ML_OK;
    if (m < 0 || !out_dev || !raw_dev) return fail(ML_ERR_ARG, "bad argument");
    if (centre_dev && !kinv_host) return fail(ML_ERR_ARG, "centre given without kinv");
    mlk::Kinv ki;
    for (int i = 0; i < 9; ++i) ki.k[i] = kinv_host ? kinv_host[i] : 0.f;
    const int grid = (int)((m + 255) / 256);
    hipLaunchKernelGGL(mlk::post_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, raw_dev, out_features,
                       row_index_dev, m, centre_dev, ki, box_conf_dev, out_dev, xyzds_dev);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

#define ML_GRID(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, (hipStream_t)stream

int ml_preprocess_rows(const float* kps_dev, const float* kps_r_dev, int64_t m, const float* kinv_table_host, int nk,
                       const int32_t* k_index_dev, float z_met, float* x_dev, void* stream) {
    if (m < 0 || nk <= 0 || !kinv_table_host || (m > 0 && (!kps_dev || !k_index_dev || !x_dev)))
        return fail(ML_ERR_ARG, "bad argument");
    if (m == 0) return ML_OK;
    hipStream_t st = (hipStream_t)stream;
    // The device copy of the table comes from a small per-DEVICE ring of (pinned staging, device table, event) slots -- this entry point
    // has no handle to hang a workspace on.  A slot is reused only after the event recorded behind the kernel that read it has
    // completed, so the call itself is asynchronous on `stream` like every other entry (no stream synchronisation), callers on other
    // devices never meet, and callers on other streams of one device hold the device's lock only while they enqueue.  The rings live
    // for the life of the process (a few KiB per device that ever prepared rows).
    struct Slot {
        mlk::Kinv* host = nullptr;
        mlk::Kinv* dev = nullptr;
        int cap = 0;
        hipEvent_t done = nullptr;
        bool in_flight = false;
    };
    struct DevRing {
        std::mutex mu;
        Slot slot[4];
        int next = 0;
    };
    static std::mutex map_mu;
    static std::map<int, DevRing*> rings;
    int dev_id = 0;
    HIP_TRY(hipGetDevice(&dev_id));
    DevRing* ring;
    {
        std::lock_guard<std::mutex> lock(map_mu);
        DevRing*& r = rings[dev_id];
        if (!r) r = new DevRing();
        ring = r;
    }
    std::lock_guard<std::mutex> lock(ring->mu);
    Slot& sl = ring->slot[ring->next];
    ring->next = (ring->next + 1) & 3;
    if (sl.in_flight) {
        HIP_TRY(hipEventSynchronize(sl.done));   // the launch that read this slot four calls ago
        sl.in_flight = false;
    }
    if (sl.cap < nk) {
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.host) (void)hipHostFree(sl.host);
        sl.dev = sl.host = nullptr;
        sl.cap = 0;
        const int cap = nk < 64 ? 64 : nk;
        HIP_TRY(hipMalloc((void**)&sl.dev, (size_t)cap * sizeof(mlk::Kinv)));
        HIP_TRY(hipHostMalloc((void**)&sl.host, (size_t)cap * sizeof(mlk::Kinv), hipHostMallocDefault));
        if (!sl.done) HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        sl.cap = cap;
    }
    for (int i = 0; i < nk; ++i) sl.host[i] = make_kinv(kinv_table_host + (size_t)i * 9);
    HIP_TRY(hipMemcpyAsync(sl.dev, sl.host, (size_t)nk * sizeof(mlk::Kinv), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(mlk::prep_rows_kernel, ML_GRID(m * mlk::NKP), kps_dev, kps_r_dev, m, (const mlk::Kinv*)sl.dev, k_index_dev, z_met,
                       x_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sl.done, st));
    sl.in_flight = true;
    return ML_OK;
}

int ml_post_geometry_strided(const float* kps_dev, int64_t m, const float* kinv_host, const float* d_dev, int64_t d_stride,
                             float* out_dev, void* stream) {
    if (m < 0 || !kinv_host || (m > 0 && (!kps_dev || !out_dev)) || d_stride < 1) return fail(ML_ERR_ARG, "bad argument");
    if (m == 0) return ML_OK;
    hipLaunchKernelGGL(mlk::post_geometry_kernel, ML_GRID(m), kps_dev, m, make_kinv(kinv_host), d_dev, d_stride, out_dev);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

int ml_post_geometry(const float* kps_dev, int64_t m, const float* kinv_host, const float* d_dev, float* out_dev,
                     void* stream) {
    return ml_post_geometry_strided(kps_dev, m, kinv_host, d_dev, 1, out_dev, stream);
}

int ml_extract_outputs_mono(const float* raw_dev, int64_t m, float* out_dev, void* stream) {
    if (m < 0 || (m > 0 && (!raw_dev || !out_dev))) return fail(ML_ERR_ARG, "bad argument");
    if (m == 0) return ML_OK;
    hipLaunchKernelGGL(mlk::post_mono_p_kernel, ML_GRID(m), raw_dev, m, out_dev);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

int ml_laplace_sampling(const float* mu_b_dev, int64_t m, int n_samples, uint32_t seed, float* out_dev, void* stream) {
    if (m < 0 || n_samples <= 0 || (m > 0 && (!mu_b_dev || !out_dev))) return fail(ML_ERR_ARG, "bad argument");
    if (m == 0) return ML_OK;
    hipLaunchKernelGGL(mlk::laplace_sample_kernel, ML_GRID(m * (int64_t)n_samples), mu_b_dev, m, n_samples, seed, out_dev);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

int ml_pixel_to_camera(const float* uv_dev, int64_t n, const float* kinv_host, float z_met, float* out_dev,
                       void* stream) {
    if (n < 0 || !kinv_host || (n > 0 && (!uv_dev || !out_dev))) return fail(ML_ERR_ARG, "bad argument");
    if (n == 0) return ML_OK;
    hipLaunchKernelGGL(mlk::pix2cam_kernel, ML_GRID(n), uv_dev, n, make_kinv(kinv_host), z_met, out_dev);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

int ml_get_keypoints(const float* kps_dev, int64_t m, int mode, float* out_dev, void* stream) {
    if (m < 0 || mode < 0 || mode > 5 || (m > 0 && (!kps_dev || !out_dev))) return fail(ML_ERR_ARG, "bad argument");
    if (m == 0) return ML_OK;
    hipLaunchKernelGGL(mlk::keypoints_kernel, ML_GRID(m), kps_dev, m, mode, out_dev);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

int ml_xyz_from_distance(const float* d_dev, int d_is_scalar, const float* centres_dev, int64_t m, float* out_dev,
                         void* stream) {
    if (m < 0 || (m > 0 && (!d_dev || !centres_dev || !out_dev))) return fail(ML_ERR_ARG, "bad argument");
    if (m == 0) return ML_OK;
    hipLaunchKernelGGL(mlk::xyz_from_distance_kernel, ML_GRID(m), d_dev, d_is_scalar ? 0 : 1, centres_dev, m, out_dev);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

int ml_to_cartesian(const float* rtp_dev, int64_t m, int mode, float* out_dev, void* stream) {
    if (m < 0 || mode < 0 || mode > 2 || (m > 0 && (!rtp_dev || !out_dev))) return fail(ML_ERR_ARG, "bad argument");
    if (m == 0) return ML_OK;
    hipLaunchKernelGGL(mlk::to_cartesian_kernel, ML_GRID(m), rtp_dev, m, mode, out_dev);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

int ml_back_correct_angles(const float* yaw_dev, const float* xyz_dev, int64_t m, float* out_dev, void* stream) {
    if (m < 0 || (m > 0 && (!yaw_dev || !xyz_dev || !out_dev))) return fail(ML_ERR_ARG, "bad argument");
    if (m == 0) return ML_OK;
    hipLaunchKernelGGL(mlk::back_correct_kernel, ML_GRID(m), yaw_dev, xyz_dev, m, out_dev);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}
#undef ML_GRID

// ---------------------------------------------------------------- the MLP
int ml_loco_forward_raw(ml_loco* h, const float* x_dev, int64_t m, float* raw_dev, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (m == 0) return ML_OK;
    if (m < 0 || !raw_dev || !x_dev) return fail(ML_ERR_ARG, "bad argument");
    if ((rc = ensure_rows(h, m))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t m_pad = round_up64(m, 256);
    const int64_t chunks = m_pad * (h->k0pad / 4);
    hipLaunchKernelGGL(mlk::f32_to_lines_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, x_dev, m,
                       h->in_f, h->buf[0], h->k0pad, m_pad);
    HIP_TRY(hipGetLastError());
    return run_network(h, m, raw_dev, st);
}

// ---------------------------------------------------------------- fused pipelines
// geo_out != null: also the post_process geometry block (from the same keypoints); *geo_done = it was written by the launch that
// ended the forward (a single image), otherwise the caller runs ml_post_geometry_strided
// does the mono pipeline's input layer pre-process its own persons for a call of `rows` rows (dense_mid_kernel<.., PREP>)?  Inside the
// mid window (its K = 64 layer runs dense_mid_kernel on both of the window's tile families), 3-product mode, LDS-DMA loader, the
// whole call one chunk, the plain first layer of a LocoModel (34 inputs padded to 64, ReLU, nothing riding in its epilogue)
static bool mid_prep_runs(const ml_loco* h, int64_t rows) {
    if (!h->tune.mid_prep || !h->tune.mid_dma || h->precision != ML_PREC_F16X2 || h->tune.chunk_rows > 0) return false;
    if (!use_mid_path(h->tune, h->precision, rows) || h->layers.empty()) return false;
    const DenseLayer& L = h->layers[0];
    if (!(L.kpad == 64 && L.relu && L.res < 0 && L.src == 0 && L.n % mlk::MID_TN == 0 && h->in_f == mlk::NIN && h->k0pad == 64)) return false;
    for (const Head& hd : h->heads)
        if (hd.after_layer == 0) return false;
    return rows <= 0x7fffffff;
}

static int forward_mono_impl(ml_loco* h, const float* kps_dev, int64_t m, const float* kinv_host, const float* box_conf_dev,
                             float* raw_dev, float* out_dev, float* xyzds_dev, void* stream, float* geo_out, bool* geo_done) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->in_f != mlk::NIN) return fail(ML_ERR_SHAPE, "mono pipeline needs a 34-input model, this one has %d", h->in_f);
    if (h->legacy || (h->out_f != 9 && h->out_f != 10))
        return fail(ML_ERR_SHAPE, "the fused mono pipeline needs a LocoModel with 9 or 10 outputs (legacy MonolocoModel: "
                                  "ml_preprocess_mono + ml_loco_forward_raw)");
    if (m == 0) return ML_OK;
    if (m < 0 || !kinv_host || !out_dev || !kps_dev) return fail(ML_ERR_ARG, "bad argument");
    if ((rc = ensure_rows(h, m))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t m_pad = round_up64(m, 256);
    const mlk::Kinv ki = make_kinv(kinv_host);
    // the small-row dense kernels read whole 32-row tiles only: no need to zero-fill up to the 256-row panel
    const bool fused_prep = mid_prep_runs(h, m);   // the input layer pre-processes its own persons (no prep_kernel, no X0 round trip)
    if (!fused_prep && (rc = launch_prep(st, kps_dev, m, ki, 10.0f, (float*)nullptr, h->d_centre, h->buf[0], h->k0pad,
                                         use_small_path(h->tune, h->precision, m) ? round_up64(m, 32) : m_pad, 0)))
        return rc;
    float* raw = raw_dev ? raw_dev : h->d_raw;
    TailMono tail{h->d_centre, ki, box_conf_dev, out_dev, xyzds_dev, raw_dev};
    if (fused_prep) {
        tail.prep_kps = kps_dev;
        tail.prep_centre = h->d_centre;
    }
    tail.geo_kps = kps_dev;
    tail.geo_out = geo_out;
    if ((rc = run_network(h, m, raw, st, McPass(), &tail))) return rc;
    if (geo_done) *geo_done = tail.geo_done;
    if (!tail.done)
        hipLaunchKernelGGL(mlk::post_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, raw, h->out_f,
                           (const int32_t*)nullptr, m, h->d_centre, ki, box_conf_dev, out_dev, xyzds_dev);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

int ml_loco_forward_mono(ml_loco* h, const float* kps_dev, int64_t m, const float* kinv_host,
                         const float* box_conf_dev, float* raw_dev, float* out_dev, float* xyzds_dev,
                         void* stream) {
    return forward_mono_impl(h, kps_dev, m, kinv_host, box_conf_dev, raw_dev, out_dev, xyzds_dev, stream, nullptr, nullptr);
}

// the stereo pipeline up to the raw rows of all ml x mr pairs: both pre-processes, the all-vs-all pairing, the network
static int stereo_front(ml_loco* h, const float* kps_l_dev, int64_t ml, const float* kps_r_dev, int64_t mr, const mlk::Kinv& ki,
                        float* raw, hipStream_t st, bool one_launch = false) {
    int rc;
    const int64_t rows = ml * mr;
    if ((rc = ensure_rows(h, rows))) return rc;
    if ((rc = ensure_side(h, ml > mr ? ml : mr))) return rc;
    const int64_t rows_pad = round_up64(rows, 256);
    const int64_t chunks = rows_pad * (h->k0pad / 4);
    if (one_launch && ml + mr <= mlk::STEREO_FRONT_MAX) {
        // a frame: both pre-processes and the pairing in ONE launch (stereo_front_kernel; same values as the three below)
        int grid = (int)((chunks + 255) / 256);
        if (grid > 64) grid = 64;
        hipLaunchKernelGGL(mlk::stereo_front_kernel, dim3((unsigned)grid), dim3(256), 0, st, kps_l_dev, (int)ml, kps_r_dev, (int)mr, ki, 10.0f,
                           h->d_cl, h->buf[0], h->k0pad, rows_pad);
    } else {
        if ((rc = launch_prep(st, kps_l_dev, ml, ki, 10.0f, h->d_xl, h->d_cl, (char*)nullptr, 0, ml, 0))) return rc;
        if ((rc = launch_prep(st, kps_r_dev, mr, ki, 10.0f, h->d_xr, (float*)nullptr, (char*)nullptr, 0, mr, 0))) return rc;
        hipLaunchKernelGGL(mlk::pairs_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, h->d_xl, ml, h->d_xr,
                           mr, (float*)nullptr, h->buf[0], h->k0pad, rows_pad);
    }
    HIP_TRY(hipGetLastError());
    return run_network(h, rows, raw ? raw : h->d_raw, st);
}

int ml_loco_forward_stereo(ml_loco* h, const float* kps_l_dev, int64_t ml, const float* kps_r_dev, int64_t mr,
                           const float* kinv_host, const float* box_conf_dev, float* raw_all_dev, float* out_dev,
                           float* xyzds_dev, int32_t* best_dev, int32_t* ties_dev, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->in_f != 2 * mlk::NIN || h->out_f != 10)
        return fail(ML_ERR_SHAPE, "stereo pipeline needs a 68-input / 10-output model");
    if (ml == 0) return ML_OK;
    if (ml < 0 || mr <= 0 || !kinv_host || !out_dev || !best_dev || !ties_dev || !kps_l_dev || !kps_r_dev)
        return fail(ML_ERR_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const mlk::Kinv ki = make_kinv(kinv_host);
    float* raw = raw_all_dev ? raw_all_dev : h->d_raw;
    if ((rc = stereo_front(h, kps_l_dev, ml, kps_r_dev, mr, ki, raw_all_dev, st))) return rc;
    raw = raw_all_dev ? raw_all_dev : h->d_raw;   // (the workspace may have been re-allocated by this call)
    HIP_TRY(hipMemsetAsync(ties_dev, 0, 4, st));
    hipLaunchKernelGGL(mlk::stereo_best_kernel, dim3((unsigned)((ml + 255) / 256)), dim3(256), 0, st, raw, h->out_f, ml,
                       mr, best_dev, h->d_rowidx, ties_dev);
    hipLaunchKernelGGL(mlk::post_kernel, dim3((unsigned)((ml + 255) / 256)), dim3(256), 0, st, raw, h->out_f,
                       (const int32_t*)h->d_rowidx, ml, h->d_cl, ki, box_conf_dev, out_dev, xyzds_dev);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

// ---------------------------------------------------------------- MC-dropout epistemic uncertainty
// kps_dev + kinv_host (pixel keypoints, pre-processed here) or x_dev (the (m, 34) network inputs the reference's method takes)
static int epistemic_impl(ml_loco* h, const float* kps_dev, const float* kinv_host, const float* x_dev, int64_t m, int n_dropout,
                          float p_dropout, int n_samples, uint32_t seed, float* epi_dev, float* raw_passes_dev, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->in_f != mlk::NIN) return fail(ML_ERR_SHAPE, "epistemic uncertainty is defined for the mono nets (net.py:139)");
    if (m == 0) return ML_OK;
    // p_dropout == 0 is legal (reference net.py:135-161 with a p = 0 model): the passes are then identical and the
    // spread is the purely aleatoric one of the Laplace samples
    if (m < 0 || !(x_dev || (kps_dev && kinv_host)) || !epi_dev || n_dropout <= 0 || n_samples <= 0 || !(p_dropout >= 0.f) ||
        !(p_dropout < 1.f))
        return fail(ML_ERR_ARG, "bad argument");
    if ((rc = ensure_rows(h, m))) return rc;
    hipStream_t st = (hipStream_t)stream;
    // The stochastic passes are independent network rows: batch as many as fit a ~128 k-row launch (a single image
    // with 50 passes is ONE 800-row forward instead of 50 latency-bound 16-row ones).
    int64_t per_chunk = 131072 / m;
    if (per_chunk < 1) per_chunk = 1;
    if (per_chunk > n_dropout) per_chunk = n_dropout;
    if ((rc = ensure_rows(h, per_chunk * m))) return rc;
    // acc: running (sum, sum of squares) per person, then per-(pass, person) partials of one chunk -- in the workspace
    // ml_loco_reserve / ensure_rows sized (4 doubles per row >= 2 m + 2 per_chunk m): no allocation on this call
    double* acc = h->d_mc;
    HIP_TRY(hipMemsetAsync(acc, 0, (size_t)m * 2 * sizeof(double), st));
    double* part = acc + m * 2;
    const mlk::Kinv ki = x_dev ? mlk::Kinv() : make_kinv(kinv_host);
    const unsigned grid_m = (unsigned)((m + 255) / 256);
    const int64_t line_bytes = (int64_t)m * h->k0pad * 4;  // the m input rows in line format
    for (int pass0 = 0; pass0 < n_dropout && !rc; pass0 += (int)per_chunk) {
        const int pc = (int)((n_dropout - pass0 < per_chunk) ? (n_dropout - pass0) : per_chunk);
        const int64_t rows = (int64_t)pc * m;
        const int64_t rows_pad = round_up64(rows, 256);
        // the stochastic forward consumes the input lines afresh every time (buffer A is overwritten); rows beyond
        // the batched passes are zero-filled by the first prep launch only up to round_up(m), so clear the tail
        // legacy 'monoloco' (2 outputs = d, s) is fed zero-centred inputs (net.py:96)
        if (x_dev) {
            const int64_t mp = round_up64(m, 256), chunks = mp * (h->k0pad / 4);
            hipLaunchKernelGGL(mlk::f32_to_lines_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, x_dev, m, h->in_f,
                               h->buf[0], h->k0pad, mp);
        } else if ((rc = launch_prep(st, kps_dev, m, ki, 10.0f, (float*)nullptr, (float*)nullptr, h->buf[0], h->k0pad,
                                     round_up64(m, 256), (h->legacy && h->out_f == 2) ? 1 : 0))) {
            return rc;
        }
        if (pc > 1) {
            const int64_t n16 = line_bytes / 16 * (pc - 1);
            hipLaunchKernelGGL(mlk::replicate_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, h->buf[0], line_bytes, pc);
        }
        auto hip_rc = [&](hipError_t e, const char* what) {
            return e == hipSuccess ? ML_OK : fail(ML_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
        };
        if (rows_pad > rows)  // pad rows of the last 256-row panel: finite values so that nothing propagates NaN patterns
            rc = hip_rc(hipMemsetAsync(h->buf[0] + rows * (int64_t)h->k0pad * 4, 0, (size_t)(rows_pad - rows) * h->k0pad * 4, st),
                        "hipMemsetAsync");
        if (rc) break;
        McPass mc;
        mc.p = p_dropout;
        mc.seed = seed * 7919u + (uint32_t)pass0 + 1u;
        mc.m_per = m;
        rc = run_network(h, rows, h->d_raw, st, mc);
        if (rc) break;
        if (raw_passes_dev)
            rc = hip_rc(hipMemcpyAsync(raw_passes_dev + (size_t)pass0 * m * h->out_f, h->d_raw, (size_t)rows * h->out_f * 4,
                                       hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
        if (rc) break;
        hipLaunchKernelGGL(mlk::mc_accumulate_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st,
                           (const float*)h->d_raw, h->out_f, (h->legacy && h->out_f == 2) ? 0 : 2, m, pc, n_samples, seed,
                           part);  // net.py:148-151
        hipLaunchKernelGGL(mlk::mc_reduce_kernel, dim3(grid_m), dim3(256), 0, st, (const double*)part, m, pc, acc, acc + m);
    }
    if (!rc) {
        hipLaunchKernelGGL(mlk::mc_finish_kernel, dim3(grid_m), dim3(256), 0, st, (const double*)acc, (const double*)(acc + m),
                           m, (double)n_dropout * (double)n_samples, epi_dev);
        if (hipGetLastError() != hipSuccess) rc = fail(ML_ERR_HIP, "mc kernel launch failed");
    }
    return rc;
}

int ml_loco_epistemic_mono(ml_loco* h, const float* kps_dev, int64_t m, const float* kinv_host, int n_dropout,
                           float p_dropout, int n_samples, uint32_t seed, float* epi_dev, float* raw_passes_dev,
                           void* stream) {
    if (m > 0 && (!kps_dev || !kinv_host)) return fail(ML_ERR_ARG, "bad argument");
    return epistemic_impl(h, kps_dev, kinv_host, nullptr, m, n_dropout, p_dropout, n_samples, seed, epi_dev, raw_passes_dev, stream);
}

int ml_loco_epistemic_inputs(ml_loco* h, const float* x_dev, int64_t m, int n_dropout, float p_dropout, int n_samples,
                             uint32_t seed, float* epi_dev, float* raw_passes_dev, void* stream) {
    if (m > 0 && !x_dev) return fail(ML_ERR_ARG, "bad argument");
    return epistemic_impl(h, nullptr, nullptr, x_dev, m, n_dropout, p_dropout, n_samples, seed, epi_dev, raw_passes_dev, stream);
}

// filter_outputs' mask (process.py:319-327) as a row list: for every left person, in order, every pair row whose aux logit is >=
// the maximum over its right candidates (a NaN among them empties the person, as `val >= nan` does).
int ml_stereo_tied_rows(const float* raw_all_dev, int out_features, int64_t ml, int64_t mr, int32_t* rows_dev, int32_t* count_dev,
                        void* stream) {
    if (!raw_all_dev || !rows_dev || !count_dev || ml < 0 || mr <= 0 || out_features < 1 || ml * mr > 0x7fffffffLL)
        return fail(ML_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(mlk::stereo_tied_rows_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, raw_all_dev, out_features, ml, mr,
                       rows_dev, count_dev);
    HIP_TRY(hipGetLastError());
    return ML_OK;
}

// ---------------------------------------------------------------- test hooks
int ml_debug_split_f16(const float* host_in, int64_t n, uint16_t* host_hi, uint16_t* host_lo) {
    if (!host_in || !host_hi || !host_lo || n < 0) return fail(ML_ERR_ARG, "bad argument");
    for (int64_t i = 0; i < n; ++i) split_host(host_in[i], host_hi[i], host_lo[i]);
    return ML_OK;
}

int ml_loco_route(const ml_loco* h, int64_t rows) {
    // which dense kernel family a forward of `rows` network rows takes on this handle (its precision and tuning); the mid window's
    // long-K layers: 64- / 128-row dense_mid_kernel tiles or dense_kernel_w4's half-size tile
    if (!h || rows < 0) return -1;
    return route_family(h, rows);
}

int ml_loco_plan(const ml_loco* h, int64_t rows, int mc_dropout, int with_post, char* text, int64_t cap) {
    // the launch plan run_network executes for such a call, as text (the same make_plan: nothing is decided anywhere else)
    if (!h || rows < 0 || !text || cap <= 0) return fail(ML_ERR_ARG, "ml_loco_plan: bad argument");
    if (!h->finalized) return fail(ML_ERR_STATE, "model not finalized");
    const std::string s = plan_text(h, make_plan(h, rows, mc_dropout != 0), with_post != 0);
    if ((int64_t)s.size() + 1 > cap) return fail(ML_ERR_ARG, "ml_loco_plan: the text needs %zu bytes", s.size() + 1);
    memcpy(text, s.c_str(), s.size() + 1);
    return ML_OK;
}

int ml_loco_set_tuning(ml_loco* h, int small_rows, int small32_rows, int chunk_rows, int tile_kernel, int mid_rows, int mid_tile) {
    // negative = keep; the defaults are 512 (256 from hidden 1024 on, set at finalize) / 128 / 0 / 4 / 8192 / 0 (measured crossovers,
    // profiles/r03_mid_sweep.txt, profiles/r06_ablation.md)
    if (mid_tile > 0 && mid_tile != 64 && mid_tile != 128 && mid_tile != 256)
        return fail(ML_ERR_ARG, "mid tile must be 0 (auto), 64 or 128 (dense_mid_kernel's tile height) or 256 (dense_kernel_w4's half-size tile)");
    if (!h) return fail(ML_ERR_ARG, "null handle");
    if (tile_kernel >= 0) {
        const int which = tile_kernel & 255;
        if (which != 2 && which != 4) return fail(ML_ERR_ARG, "tile kernel must be 2 (ping-pong) or 4 (one wave per SIMD)");
        h->tune.tile_kernel = which;
        h->tune.tile_all = (tile_kernel & 256) ? 1 : 0;
    }
    if (small_rows >= 0) h->tune.small_rows = small_rows;
    if (small32_rows >= 0) h->tune.small32_rows = small32_rows;
    if (chunk_rows >= 0) h->tune.chunk_rows = chunk_rows;
    if (mid_rows >= 0) h->tune.mid_rows = mid_rows;
    if (mid_tile >= 0) h->tune.mid_tile = mid_tile;
    ++h->tune_version;
    return ML_OK;
}

int ml_loco_set_option(ml_loco* h, const char* name, int value) {
    // named per-handle switches of the route plan (A/B runs and tests; results stay within the route-invariance bars)
    if (!h || !name) return fail(ML_ERR_ARG, "null argument");
    const std::string n(name);
    if (n == "mid_heads" || n == "half_heads") h->tune.half_heads = value ? 1 : 0;
    else if (n == "half_from") h->tune.half_from = value;
    else if (n == "small_multi") h->tune.small_multi = value ? 1 : 0;
    else if (n == "mid_splitk") h->tune.mid_splitk = value;
    else if (n == "mid_wgs") h->tune.mid_wgs = value;
    else if (n == "mid_dma") h->tune.mid_dma = value ? 1 : 0;
    else if (n == "mid_prep") h->tune.mid_prep = value ? 1 : 0;
    else return fail(ML_ERR_ARG, "unknown option '%s'", name);
    ++h->tune_version;
    return ML_OK;
}

static std::atomic<long long> g_frames_without_copies{0};   // (test hook: ml_debug_frames_without_copies)
static std::atomic<long long> g_frame_flag_timeouts{0};     // frames whose completion word did not arrive within 5 ms (expected: 0)
static std::atomic<int> g_frame_spin{1};                    // ml_debug_frame_spin: 0 = always hipStreamSynchronize (the A/B reference)

// Is [p, p + bytes) pinned (device-mapped) host memory a kernel may dereference?  Both ends of the range are asked of the runtime
// (hipPointerGetAttributes: a driver call each), and the verdict is remembered per handle for the last 8 (pointer, extent) pairs: a
// caller streams its frames through the same staging buffers.  A remembered range is trusted until ml_loco_forget_pinned -- the
// caller's side of the contract (include/monoloco_hip.h): memory that was handed to a frame entry stays pinned until it is forgotten
// or the handle is destroyed (hipHostUnregister / hipHostFree of a remembered buffer, then reuse of the address by malloc, would
// otherwise put pageable memory behind a remembered address: a GPU page fault, not an error code).
static bool frame_pinned(ml_loco* h, const void* p, size_t bytes) {
    if (!p || !bytes) return false;
    if (h)
        for (const auto& q : h->pinned_seen)
            if (q.p == p && bytes <= q.bytes) return true;
    const void* ends[2] = {p, (const char*)p + bytes - 1};
    for (const void* e : ends) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, e) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (a.type != hipMemoryTypeHost) return false;
    }
    if (h) {
        int slot = h->pinned_next;
        for (int i = 0; i < 8; ++i)
            if (h->pinned_seen[i].p == p) slot = i;       // the same buffer with a larger extent: replace its entry
        h->pinned_seen[slot].p = p;
        h->pinned_seen[slot].bytes = bytes;
        if (slot == h->pinned_next) h->pinned_next = (h->pinned_next + 1) % 8;
    }
    return true;
}

int ml_loco_forget_pinned(ml_loco* h, const void* host_ptr) {
    if (!h) return fail(ML_ERR_ARG, "bad argument");
    for (auto& q : h->pinned_seen)
        if (!host_ptr || q.p == host_ptr) {
            q.p = nullptr;
            q.bytes = 0;
        }
    return ML_OK;
}

// One image through the mono pipeline in ONE call: pinned host keypoints in, [packed (m, 16) | post-process geometry (m, 12)] in
// pinned host memory out, one stream synchronisation.  What Loco.forward does per frame (reference net.py:83-133 + the geometry
// of :195-215); as one entry the host pays one foreign call instead of five.
//   * up to 128 persons (one image; the forward ends in heads_small_kernel): NO copy operation at all -- prep_kernel reads the
//     pinned keypoints over the link, the last launch writes both result blocks straight into the pinned output (device
//     pointers of pinned host memory are valid kernel arguments): 10 launches and the synchronisation;
//   * more: keypoints to the device (async copy), pipeline, ml_post_geometry_strided, one copy back.
int ml_loco_frame_mono(ml_loco* h, const float* kps_host, int64_t m, const float* kinv_host, float* kps_dev, float* buf_dev,
                       float* xyzds_dev, float* out_host, void* stream) {
    if (m < 0 || !kinv_host || (m > 0 && (!kps_host || !kps_dev || !buf_dev || !out_host))) return fail(ML_ERR_ARG, "bad argument");
    if (m == 0) return ML_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    // kernels may only dereference PINNED (device-mapped) host memory: anything else takes the staged route, where the runtime's
    // copies accept pageable memory as well (a wrong guess here would be a GPU page fault, not an error code)
    // (frame_pinned: the verdict is remembered per handle for the last few (pointer, extent) pairs -- the contract of
    //  ml_loco_forget_pinned in include/monoloco_hip.h)
    if (m <= 128 && h && use_small_path(h->tune, h->precision, m) && frame_pinned(h, kps_host, (size_t)m * 3 * mlk::NKP * 4) && frame_pinned(h, out_host, (size_t)m * (ML_OUT_STRIDE + ML_POSTGEO_STRIDE) * 4)) {
        bool geo_done = false;
        h->frame_flag_req = g_frame_spin.load(std::memory_order_relaxed) != 0;
        h->frame_flag_armed = false;
        rc = forward_mono_impl(h, kps_host, m, kinv_host, nullptr, nullptr, out_host, xyzds_dev, stream,
                               out_host + (size_t)m * ML_OUT_STRIDE, &geo_done);
        h->frame_flag_req = false;
        if (rc) return rc;
        if (!geo_done &&   // (a model whose heads do not end in the one launch: the geometry as its own launch, still into host memory)
            (rc = ml_post_geometry_strided(kps_host, m, kinv_host, out_host + 3, ML_OUT_STRIDE, out_host + (size_t)m * ML_OUT_STRIDE, stream)))
            return rc;
        bool seen = false;
        if (h->frame_flag_armed && geo_done) {
            // the last launch releases done_seq into the pinned word behind every store of the frame: poll it (the launches above are
            // ~60 us of device time; a frame that has not reported after 5 ms falls back to the stream synchronisation)
            const int want = h->done_seq;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0;; ++spins) {
                if (__atomic_load_n(h->h_done, __ATOMIC_ACQUIRE) == want) {
                    seen = true;
                    break;
                }
                if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
                __builtin_ia32_pause();
            }
            if (!seen) g_frame_flag_timeouts.fetch_add(1, std::memory_order_relaxed);
        }
        if (!seen) HIP_TRY(hipStreamSynchronize(st));
        g_frames_without_copies.fetch_add(1, std::memory_order_relaxed);   // counted once the frame has gone through
        return ML_OK;
    }
    HIP_TRY(hipMemcpyAsync(kps_dev, kps_host, (size_t)m * 3 * mlk::NKP * 4, hipMemcpyHostToDevice, st));
    if ((rc = ml_loco_forward_mono(h, kps_dev, m, kinv_host, nullptr, nullptr, buf_dev, xyzds_dev, stream))) return rc;
    float* geo = buf_dev + (size_t)m * ML_OUT_STRIDE;
    if ((rc = ml_post_geometry_strided(kps_dev, m, kinv_host, buf_dev + 3, ML_OUT_STRIDE, geo, stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out_host, buf_dev, (size_t)m * (ML_OUT_STRIDE + ML_POSTGEO_STRIDE) * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return ML_OK;
}

// One STEREO image pair in one call (MonStereo's Loco.forward, reference net.py:112-122 + the geometry of :195-215): pinned host
// keypoints of both views in; [packed (ml, 16) | geometry (ml, 12) | int32 ties | int32 best (ml)] in pinned host memory out.  With
// pinned buffers no copy operation at all -- the pre-process kernels read the keypoints over the link, the post-process / arg-max /
// geometry kernels write the pinned block, the last launch releases the completion word the host polls; otherwise staged copies.
int ml_loco_frame_stereo(ml_loco* h, const float* kps_l_host, int64_t ml, const float* kps_r_host, int64_t mr, const float* kinv_host,
                         float* kps_dev, float* buf_dev, float* xyzds_dev, float* out_host, void* stream) {
    if (ml < 0 || mr <= 0 || !kinv_host || (ml > 0 && (!kps_l_host || !kps_r_host || !kps_dev || !buf_dev || !xyzds_dev || !out_host)))
        return fail(ML_ERR_ARG, "bad argument");
    if (ml == 0) return ML_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const size_t n_packed = (size_t)ml * ML_OUT_STRIDE, n_geo = (size_t)ml * ML_POSTGEO_STRIDE;
    const size_t words = n_packed + n_geo + 1 + (size_t)ml;
    const bool direct = h && h->d_arrive && frame_pinned(h, kps_l_host, (size_t)ml * 3 * mlk::NKP * 4) &&
                        frame_pinned(h, kps_r_host, (size_t)mr * 3 * mlk::NKP * 4) && frame_pinned(h, out_host, words * 4);
    const float* kl = kps_l_host;
    const float* kr = kps_r_host;
    float* blk = out_host;
    if (!direct) {
        HIP_TRY(hipMemcpyAsync(kps_dev, kps_l_host, (size_t)ml * 3 * mlk::NKP * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(kps_dev + (size_t)ml * 3 * mlk::NKP, kps_r_host, (size_t)mr * 3 * mlk::NKP * 4, hipMemcpyHostToDevice, st));
        kl = kps_dev;
        kr = kps_dev + (size_t)ml * 3 * mlk::NKP;
        blk = buf_dev;
    }
    const mlk::Kinv ki = make_kinv(kinv_host);
    mlk::FrameDone fd;
    const bool spin = direct && h->h_done && g_frame_spin.load(std::memory_order_relaxed) != 0;
    if (direct) {
        // the whole end of the frame -- per-left arg-max, post-process of the winners, geometry, tie count, completion word -- is ONE
        // launch (stereo_tail_frame_kernel); the tie counter is the second word of the handle's arrival block (0 between frames)
        if ((rc = check_ready(h))) return rc;
        if (h->in_f != 2 * mlk::NIN || h->out_f != 10) return fail(ML_ERR_SHAPE, "stereo pipeline needs a 68-input / 10-output model");
        if ((rc = stereo_front(h, kl, ml, kr, mr, ki, nullptr, st, true))) return rc;
        fd.arrive = h->d_arrive;
        if (spin) {
            fd.flag = h->h_done;
            fd.seq = ++h->done_seq;
        }
        hipLaunchKernelGGL(mlk::stereo_tail_frame_kernel, dim3((unsigned)((ml + 255) / 256)), dim3(256), 0, st, (const float*)h->d_raw,
                           h->out_f, ml, mr, (const float*)h->d_cl, ki, kl, blk, xyzds_dev, blk + n_packed, h->d_rowidx,
                           (int32_t*)h->d_arrive + 1, (int32_t*)(blk + n_packed + n_geo), fd);
        HIP_TRY(hipGetLastError());
    } else {
        // staged route: the public pipeline + the geometry launch into the device block, copied out below
        int32_t* ties = (int32_t*)(buf_dev + n_packed + n_geo);
        int32_t* best = ties + 1;
        if ((rc = ml_loco_forward_stereo(h, kl, ml, kr, mr, kinv_host, nullptr, nullptr, blk, xyzds_dev, best, ties, stream))) return rc;
        if ((rc = ml_post_geometry_strided(kl, ml, kinv_host, blk + 3, ML_OUT_STRIDE, blk + n_packed, stream))) return rc;
    }
    if (!direct) HIP_TRY(hipMemcpyAsync(out_host, buf_dev, words * 4, hipMemcpyDeviceToHost, st));
    bool seen = false;
    if (spin) {
        const int want = fd.seq;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(h->h_done, __ATOMIC_ACQUIRE) == want) {
                seen = true;
                break;
            }
            if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
            __builtin_ia32_pause();
        }
        if (!seen) g_frame_flag_timeouts.fetch_add(1, std::memory_order_relaxed);
    }
    if (!seen) HIP_TRY(hipStreamSynchronize(st));
    if (direct) g_frames_without_copies.fetch_add(1, std::memory_order_relaxed);
    return ML_OK;
}

long long ml_debug_frames_without_copies(void) { return g_frames_without_copies.load(std::memory_order_relaxed); }

long long ml_debug_frame_spin(int enable) {
    // enable 0 / 1 switches the completion-word polling of ml_loco_frame_mono off / on (process-wide; < 0 leaves it); returns the
    // number of frames whose word did not arrive in time so far
    if (enable >= 0) g_frame_spin.store(enable ? 1 : 0, std::memory_order_relaxed);
    return g_frame_flag_timeouts.load(std::memory_order_relaxed);
}

int ml_debug_num_layers(const ml_loco* h) { return h ? (int)h->layers.size() : 0; }

int ml_debug_get_layer(const ml_loco* h, int layer, float* w_host, float* b_host, int* n, int* k, int* scale_pow2) {
    if (!h || layer < 0 || layer >= (int)h->layers.size()) return fail(ML_ERR_ARG, "bad layer index");
    const DenseLayer& L = h->layers[layer];
    if (n) *n = L.n;
    if (k) *k = L.k;
    if (scale_pow2) *scale_pow2 = L.scale_pow2;
    if (w_host) memcpy(w_host, L.w.data(), L.w.size() * 4);
    if (b_host) memcpy(b_host, L.b.data(), L.b.size() * 4);
    return ML_OK;
}

int ml_debug_get_packed(const ml_loco* h, int layer, uint16_t* lines_host, int64_t capacity) {
    if (!h || layer < 0 || layer >= (int)h->layers.size()) return fail(ML_ERR_ARG, "bad layer index");
    const DenseLayer& L = h->layers[layer];
    if (L.packed.empty()) return fail(ML_ERR_STATE, "packed image only kept for ML_FLAG_HOST_ONLY models");
    if (!lines_host || capacity < (int64_t)L.packed.size()) return fail(ML_ERR_ARG, "buffer too small");
    memcpy(lines_host, L.packed.data(), L.packed.size() * 2);
    return ML_OK;
}

int ml_debug_get_head(const ml_loco* h, int head, float* w_host, float* b_host, int* nh, int* col0, int* src_buf,
                      int* after_layer) {
    if (!h || head < 0 || head >= (int)h->heads.size()) return fail(ML_ERR_ARG, "bad head index");
    const Head& hd = h->heads[head];
    if (nh) *nh = hd.nh;
    if (col0) *col0 = hd.col0;
    if (src_buf) *src_buf = hd.src;
    if (after_layer) *after_layer = hd.after_layer;
    if (w_host) memcpy(w_host, hd.w.data(), hd.w.size() * 4);
    if (b_host) memcpy(b_host, hd.b.data(), hd.b.size() * 4);
    return ML_OK;
}

int ml_debug_linear(const float* x_dev, int64_t m, int k, const float* w_host, const float* b_host, int n, int relu,
                    const float* res_dev, float* y_dev, int precision, void* stream) {
    if (!x_dev || !w_host || !b_host || !y_dev || m <= 0 || k <= 0 || n <= 0 || n % 256 != 0)
        return fail(ML_ERR_ARG, "bad argument (n must be a multiple of 256)");
    // the tile kernel unless ML_DEBUG_SMALL_PATH is or-ed into `precision` (then the small-row kernels, any m)
    const bool small_path = (precision & ML_DEBUG_SMALL_PATH) != 0;
    Tuning tu;   // which tile kernel: ML_DEBUG_TILE_PP = dense_kernel_pp, ML_DEBUG_TILE_W4 = dense_kernel_w4 wherever it runs
    if (precision & ML_DEBUG_TILE_PP) tu.tile_kernel = 2;
    if (precision & ML_DEBUG_TILE_W4) tu.tile_all = 1;
    const bool mid_path = (precision & (ML_DEBUG_MID_64 | ML_DEBUG_MID_128)) != 0;   // dense_mid_kernel with that tile height
    if (mid_path)   // (both bits: dense_kernel_w4's half-size tile for the long-K layers)
        tu.mid_tile = ((precision & ML_DEBUG_MID_64) && (precision & ML_DEBUG_MID_128)) ? 256 : ((precision & ML_DEBUG_MID_64) ? 64 : 128);
    // dense_mid_kernel's reduction in 2 / 4 k ranges per tile (split-K, last arriver runs the epilogue); without the bits: one range
    const int dbg_split = (precision & ML_DEBUG_MID_SPLIT4) ? 4 : ((precision & ML_DEBUG_MID_SPLIT2) ? 2 : 1);
    tu.mid_splitk = dbg_split;
    tu.mid_dma = (precision & ML_DEBUG_MID_NODMA) ? 0 : 1;
    precision &= ~(ML_DEBUG_SMALL_PATH | ML_DEBUG_TILE_PP | ML_DEBUG_TILE_W4 | ML_DEBUG_MID_64 | ML_DEBUG_MID_128 | ML_DEBUG_MID_SPLIT2 |
                   ML_DEBUG_MID_SPLIT4 | ML_DEBUG_MID_NODMA);
    if ((small_path || mid_path) && precision == ML_PREC_BF16) return fail(ML_ERR_ARG, "the bf16 mode runs on the 256x256-tile kernels only");
    hipStream_t st = (hipStream_t)stream;
    ml_loco tmp;
    tmp.precision = precision;
    DenseLayer L;
    L.n = n;
    L.k = k;
    L.kpad = round_up(k, 64);
    L.relu = relu;
    L.w.assign(w_host, w_host + (size_t)n * k);
    L.b.assign(b_host, b_host + n);
    pack_layer(precision, L);
    int rc = upload_layer(&tmp, L);
    const int64_t m_pad = round_up64(m, 256);
    char *xl = nullptr, *yl = nullptr, *rl = nullptr;
    float* kpart = nullptr;
    unsigned* kcount = nullptr;
    if (!rc && mid_path && dbg_split > 1) {
        rc = dev_alloc(&tmp, &kpart, m_pad * (int64_t)n * 4 * dbg_split);
        if (!rc) rc = dev_alloc(&tmp, &kcount, MID_KCOUNT * 4);
        if (!rc && hipMemset(kcount, 0, MID_KCOUNT * 4) != hipSuccess) rc = fail(ML_ERR_HIP, "debug_linear: memset failed");
    }
    if (!rc) rc = dev_alloc(&tmp, &xl, m_pad * (int64_t)L.kpad * 4);
    if (!rc) rc = dev_alloc(&tmp, &yl, m_pad * (int64_t)n * 4);
    if (!rc && res_dev) rc = dev_alloc(&tmp, &rl, m_pad * (int64_t)n * 4);
    if (!rc) {
        int64_t chunks = m_pad * (L.kpad / 4);
        hipLaunchKernelGGL(mlk::f32_to_lines_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, x_dev, m, k,
                           xl, L.kpad, m_pad);
        if (res_dev) {
            chunks = m_pad * (n / 4);
            hipLaunchKernelGGL(mlk::f32_to_lines_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, res_dev,
                               m, n, rl, n, m_pad);
        }
        mlk::DenseParams p;
        p.x = xl;
        p.w = L.d_w;
        p.bias = L.d_b;
        p.bias_scaled = L.d_bs;
        p.res = rl;
        p.y = res_dev ? rl : yl;  // exercises the in-place residual form the model uses
        p.descale = std::ldexp(1.0f, -L.scale_pow2);
        p.descale_ptr = nullptr;
        p.M_pad = (int)m_pad;
        p.N = n;
        p.K = L.kpad;
        p.relu = relu;
        p.debug = 0;
        p.trace = nullptr;
        p.head_w = nullptr;
        p.head_part = nullptr;
        p.kpart = kpart;
        p.kcount = kcount;
        p.ksplit = dbg_split;
        if (precision == ML_PREC_BF16) {
            int64_t pairs = m_pad * (int64_t)(L.kpad / 32) * 4;
            hipLaunchKernelGGL(mlk::lines_to_bf16_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, xl, pairs);
            if (res_dev) {
                pairs = m_pad * (int64_t)(n / 32) * 4;
                hipLaunchKernelGGL(mlk::lines_to_bf16_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, rl, pairs);
            }
        }
        rc = launch_dense(tu, precision, p, st, 0, small_path ? m : -1, mid_path ? (use_half_tile(tu, precision, m) ? 2 : 1) : 0);
        if (!rc) {
            const int64_t groups = m * (n / 8);
            hipLaunchKernelGGL(mlk::lines_to_f32_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, p.y, m,
                               n, y_dev, precision == ML_PREC_BF16 ? 1 : 0);
            if (hipGetLastError() != hipSuccess) rc = fail(ML_ERR_HIP, "lines_to_f32 launch failed");
        }
        if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = fail(ML_ERR_HIP, "debug_linear: stream sync failed");
    }
    dev_free(xl);
    dev_free(yl);
    dev_free(rl);
    dev_free(kpart);
    dev_free(kcount);
    dev_free(L.d_w);
    dev_free(L.d_b);
    dev_free(L.d_bs);
    return rc;
}

}  // extern "C"
