/*
 * monoloco_hip.h -- C ABI of the MI355X (gfx950) implementation of monoloco's
 * keypoint -> 3D inference hot path.
 *
 * The reference (vita-epfl/monoloco) has no FFI: its boundary is the Python API
 * of monoloco/network/{net,process}.py and monoloco/utils/camera.py.  Every entry
 * point below names the reference interface it stands in for (file:line in the
 * reference checkout).  Signatures carry only plain pointers and sizes -- no torch
 * types -- so any host language can bind them (INTEGRATION.md shows the ctypes
 * binding the Python host side uses).
 *
 * Conventions
 *   - every function returns 0 (ML_OK) or a positive ML_ERR_* code; no exceptions
 *     cross the ABI; ml_last_error() returns a thread-local message.
 *   - "dev" pointers are HIP device pointers on the device that was current when the
 *     model was finalized; "host" pointers are ordinary host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All hot
 *     calls are asynchronous on that stream and make no allocation once
 *     ml_loco_reserve() has been called with a large enough row count.
 *   - a model handle is thread-compatible: use one handle per stream/thread.
 *   - keypoints are the reference layout (m, 3, 17) fp32, rows u, v, confidence
 *     (monoloco/network/process.py:210-218).
 */
#ifndef MONOLOCO_HIP_H
#define MONOLOCO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ML_OK 0
#define ML_ERR_ARG 1    /* bad argument (null pointer, size mismatch, unknown key) */
#define ML_ERR_HIP 2    /* a HIP runtime call failed; see ml_last_error()           */
#define ML_ERR_STATE 3  /* call order violated (e.g. forward before finalize)       */
#define ML_ERR_SHAPE 4  /* unsupported model shape                                  */

/* Arithmetic of the dense layers (the MFMA kernels).
 *   ML_PREC_F16X2 : operands split into fp16 hi+lo, 3 MFMA products per term, fp32
 *                   accumulate -- fp32-class accuracy (the default; meets 1e-4 abs).
 *   ML_PREC_F16   : plain fp16 operands, 1 MFMA per term (fast, ~1e-2 abs deviation).
 *   ML_PREC_BF16  : plain bf16 operands (v_mfma_f32_32x32x16_bf16), 1 MFMA per term -- the mode
 *                   BASELINE configs[1] names; a COMPARISON mode (misses the 1e-4 bar by orders of
 *                   magnitude, bench.py reports its measured deviation); tile path and plain forward only
 *                   (no MC-dropout, no small-row kernels). */
#define ML_PREC_F16X2 0
#define ML_PREC_F16 1
#define ML_PREC_BF16 2

/* flags for ml_loco_finalize */
#define ML_FLAG_MERGE_W2W3 1 /* pre-multiply w3*w2 and w_aux*w2 on the host in fp64
                                (no non-linearity between them, architectures.py:59-63) */
#define ML_FLAG_HOST_ONLY 256 /* test hook: fold + pack on the host, upload nothing (no GPU
                                 needed); such a model cannot run, only be inspected */

/* layout of the packed per-person result written by ml_extract_outputs / ml_loco_forward_* */
#define ML_OUT_STRIDE 16
#define ML_OUT_X 0        /* xyzd[:,0]  d*sin(psi)*cos(theta)   process.py:263, camera.py:231-233 */
#define ML_OUT_Y 1        /* xyzd[:,1]  d*cos(psi)              camera.py:235-237                 */
#define ML_OUT_Z 2        /* xyzd[:,2]  sqrt(d^2-x^2-y^2)       process.py:265 (NaN if negative)  */
#define ML_OUT_D 3        /* d                                   process.py:264                    */
#define ML_OUT_BI 4       /* bi = exp(s)*d                       process.py:125-133                */
#define ML_OUT_YAW 5      /* atan2(ori0, ori1)                   process.py:272                    */
#define ML_OUT_YAW_EGO 6  /* yaw + atan2(x,z), wrapped once      camera.py:202-208                 */
#define ML_OUT_AUX 7      /* sigmoid(aux logit) (stereo) / raw aux-head value (mono) process.py:276 */
#define ML_OUT_H 8
#define ML_OUT_W 9
#define ML_OUT_L 10
#define ML_OUT_CONF 11    /* 0.035*box_conf/(bi/|xyz_pred|)      net.py:215 (0 if no box_conf)     */
#define ML_OUT_ORI0 12
#define ML_OUT_ORI1 13
#define ML_OUT_UC 14      /* centre pixel u,v  ((max-min)/2+min) camera.py:83-86                   */
#define ML_OUT_VC 15
/* the parity tensor (m,5): back-projected x,y,z (net.py:198,213), d, sigma=bi */
#define ML_XYZDS_STRIDE 5

typedef struct ml_loco ml_loco; /* opaque model + workspace handle */

/* ---- library ---------------------------------------------------------------------- */
int ml_version(void);
const char* ml_last_error(void);
/* Number of HIP devices visible, or a negative ML_ERR code. */
int ml_device_count(void);

/* ---- model lifetime: stands in for Loco.__init__ (monoloco/network/net.py:30-81) --- */
/* Shape of LocoModel (monoloco/network/architectures.py:8-46): in_features 34 (mono) or
 * 68 (stereo), hidden = any linear_size 1..4096 (run.py:101 default 1024; zero-padded internally to the
 * 256-column tile), out_features 9 or 10 (includes the auxiliary
 * head, architectures.py:70), num_stage residual stages.  The legacy MonolocoModel
 * (architectures.py:105-176: the same stages followed by one Linear w2 -> out_features, 2 or 9; no w3 /
 * w_aux / w_fin keys) is recognised at ml_loco_finalize from the tensors that were fed; it runs through
 * ml_loco_forward_raw and ml_loco_epistemic_mono (the fused mono pipeline refuses it). */
int ml_loco_create(int in_features, int hidden, int out_features, int num_stage, ml_loco** out);
/* Feed one tensor of the reference state_dict by its key (e.g. "linear_stages.0.w1.weight",
 * "batch_norm3.running_var"; net.py:77 loads exactly these).  `data` is host fp32, row-major,
 * `numel` must match.  "*.num_batches_tracked" keys are accepted and ignored. */
int ml_loco_set_tensor(ml_loco* h, const char* key, const float* host_data, int64_t numel);
/* Fold eval-mode BatchNorm (eps 1e-5) into the Linear layers, optionally merge w3*w2,
 * scale/split/pack the weights for the MFMA kernels and upload them to the current device. */
int ml_loco_finalize(ml_loco* h, int precision, int flags);
/* Pre-allocate activation workspace for up to max_rows network rows (persons, or
 * left*right pairs for stereo).  Hot calls with more rows grow it (a hidden allocation). */
int ml_loco_reserve(ml_loco* h, int64_t max_rows);
int ml_loco_destroy(ml_loco* h);
/* Introspection: bytes of device memory held, and the per-layer power-of-two weight scale. */
int64_t ml_loco_device_bytes(const ml_loco* h);

/* ---- stand-alone geometry kernels ------------------------------------------------- */
/* preprocess_monoloco (process.py:47-67) = pixel_to_camera(kps[:,0:2,:], K, z_met)
 * (camera.py:10-29) reshaped to (m,34) interleaved x0,y0,x1,y1,...  `kinv_host` is
 * inverse(K) row-major (9 floats; the host computes it as the reference does, with
 * torch.inverse).  zero_center != 0 subtracts the normalised box centre (legacy MonoLoco,
 * process.py:61-62).  x_dev (m,34) and/or centre_dev (m,2: get_keypoints(..,'center'),
 * camera.py:82-86) may be NULL. */
int ml_preprocess_mono(const float* kps_dev, int64_t m, const float* kinv_host, float z_met,
                       int zero_center, float* x_dev, float* centre_dev, void* stream);
/* preprocess_monstereo (process.py:25-44): all-vs-all rows [L_i, L_i - R_j], i-major;
 * xl_dev (ml,34), xr_dev (mr,34) -> rows_dev (ml*mr, 68). */
int ml_stereo_pairs(const float* xl_dev, int64_t ml, const float* xr_dev, int64_t mr,
                    float* rows_dev, void* stream);
/* extract_outputs (process.py:231-278) + the per-person geometry of Loco.post_process
 * (net.py:195-215).  raw_dev (m, out_features); row_index_dev (m) optional int32 gather
 * into raw (stereo: the selected pair row), NULL = identity; centre_dev (m,2) pixel centres
 * or NULL (then xyz_pred/conf/uc/vc are 0); box_conf_dev (m) or NULL.  Writes out_dev
 * (m, ML_OUT_STRIDE) and, if not NULL, xyzds_dev (m,5). */
int ml_extract_outputs(const float* raw_dev, int out_features, const int32_t* row_index_dev,
                       int64_t m, const float* centre_dev, const float* kinv_host,
                       const float* box_conf_dev, float* out_dev, float* xyzds_dev, void* stream);

/* pixel_to_camera (utils/camera.py:10-29) on n points: uv_dev (n,2) -> out_dev (n,3) = [u,v,1].Kinv^T*z_met. */
int ml_pixel_to_camera(const float* uv_dev, int64_t n, const float* kinv_host, float z_met, float* out_dev,
                       void* stream);
/* get_keypoints (utils/camera.py:69-107): kps_dev (m,3,17) -> out_dev (m,2).  mode: 0 center, 1 bottom,
 * 2 head, 3 shoulder, 4 hip, 5 ankle. */
int ml_get_keypoints(const float* kps_dev, int64_t m, int mode, float* out_dev, void* stream);
/* The per-person geometry of Loco.post_process (network/net.py:195-215) in one launch: kps_dev (m,3,17), d_dev (m)
 * predicted distances (NULL: xyz_pred = 0) -> out_dev (m, ML_POSTGEO_STRIDE = 12):
 * uv_shoulder(2), uv_head(2), uv_center(2) (get_keypoints, utils/camera.py:69-107), xy_center(3) =
 * pixel_to_camera(uv_center, K, 1) (:10-29), xyz_pred(3) = xyz_from_distance(d, xy_center) (:161-177). */
#define ML_POSTGEO_STRIDE 12
int ml_post_geometry(const float* kps_dev, int64_t m, const float* kinv_host, const float* d_dev, float* out_dev,
                     void* stream);
/* ... with the distances read at d_dev[i * d_stride] (a column of the packed (m, ML_OUT_STRIDE) result: Loco.forward
 * computes this block right behind the network, from the keypoints it already has on the device). */
int ml_post_geometry_strided(const float* kps_dev, int64_t m, const float* kinv_host, const float* d_dev, int64_t d_stride,
                             float* out_dev, void* stream);
/* Dataset-preparation rows in one launch (prep/preprocess_kitti.py:190-253 calls preprocess_monoloco once per
 * matched annotation, each with the K of its image): kps_dev (m,3,17); kinv_table_host (nk,9) = inverses of the
 * distinct intrinsic matrices; k_index_dev (m) int32 = table entry of each row (not range-checked).  kps_r_dev
 * NULL -> x_dev (m,34) mono inputs; else (m,3,17) right keypoints -> x_dev (m,68) stereo training rows
 * [L, L - R] (preprocess_kitti.py:242-247).  Bit-identical to the per-annotation calls.  Asynchronous on `stream` (the table is
 * staged through a per-device ring of pinned slots; kinv_table_host may be reused as soon as the call returns); callers on
 * different devices never serialise, callers on different streams of one device only while they enqueue. */
int ml_preprocess_rows(const float* kps_dev, const float* kps_r_dev, int64_t m, const float* kinv_table_host, int nk,
                       const int32_t* k_index_dev, float z_met, float* x_dev, void* stream);
/* extract_outputs_mono (process.py:330-360, legacy 'monoloco_p' outputs x, y, z, log(b/z), h, w, l, sin, cos):
 * raw_dev (m,9) -> out_dev (m, ML_OUT_STRIDE) with X,Y,Z = the raw xyz, D = ||xyz||, BI = exp(raw3)*raw2,
 * YAW, YAW_EGO, H, W, L, ORI0/1; AUX, CONF, UC, VC are 0. */
int ml_extract_outputs_mono(const float* raw_dev, int64_t m, float* out_dev, void* stream);
/* laplace_sampling (process.py:101-122): mu_b_dev (m,2) = (mu, b) -> out_dev (n_samples, m) draws of
 * Laplace(mu, |b|) from the library's counter-based generator (same draws as ml_loco_epistemic_mono for the same
 * seed; the reference seeds torch's generator with 1, so parity is statistical). */
int ml_laplace_sampling(const float* mu_b_dev, int64_t m, int n_samples, uint32_t seed, float* out_dev, void* stream);
/* xyz_from_distance (utils/camera.py:161-177): d_dev (m) (or one value if d_is_scalar), centres_dev (m,3)
 * -> out_dev (m,3). */
int ml_xyz_from_distance(const float* d_dev, int d_is_scalar, const float* centres_dev, int64_t m, float* out_dev,
                         void* stream);
/* to_cartesian, tensor branch (utils/camera.py:223-248): rtp_dev (m,3); mode 0 = 'x', 1 = 'y' with rows
 * (theta, psi, r) -> out (m); mode 2 = rows (r, theta, psi) -> out (m,3). */
int ml_to_cartesian(const float* rtp_dev, int64_t m, int mode, float* out_dev, void* stream);
/* back_correct_angles (utils/camera.py:202-208): yaw_dev (m), xyz_dev (m,3) -> out_dev (m). */
int ml_back_correct_angles(const float* yaw_dev, const float* xyz_dev, int64_t m, float* out_dev, void* stream);

/* ---- the MLP: stands in for LocoModel.forward (architectures.py:48-71) ------------- */
/* x_dev (m, in_features) fp32 -> raw_dev (m, out_features) fp32 (aux head last). */
int ml_loco_forward_raw(ml_loco* h, const float* x_dev, int64_t m, float* raw_dev, void* stream);

/* ---- fused device pipelines: stand in for Loco.forward + post_process geometry ----- */
/* mono (net.py:106-110,195-215): kps_dev (m,3,17) -> out_dev (m,16), xyzds_dev (m,5) or NULL,
 * raw_dev (m,out_features) or NULL.  box_conf_dev (m) or NULL. */
int ml_loco_forward_mono(ml_loco* h, const float* kps_dev, int64_t m, const float* kinv_host,
                         const float* box_conf_dev, float* raw_dev, float* out_dev,
                         float* xyzds_dev, void* stream);
/* One image in one call (what the reference's Loco.forward + the geometry of post_process do per frame, net.py:83-133,
 * 195-215): kps_host (m,3,17) and out_host (m * 28 floats = [packed (m,16) | post-process geometry (m,12)]) are PINNED,
 * device-mapped host memory (hipHostMalloc / torch pin_memory: the pointers are dereferenced by kernels); one stream
 * synchronisation inside.  Up to 128 persons no copy operation is issued at all (the first kernel reads the keypoints over the
 * link, the last one writes both blocks into out_host); beyond that kps_dev (m,3,17) and buf_dev (m * 28 floats) stage the
 * transfers.  xyzds_dev (m,5) or NULL. */
int ml_loco_frame_mono(ml_loco* h, const float* kps_host, int64_t m, const float* kinv_host, float* kps_dev, float* buf_dev,
                       float* xyzds_dev, float* out_host, void* stream);
/* One STEREO image pair in one call -- MonStereo's Loco.forward (monoloco/network/net.py:112-122: all-vs-all pairing, network,
 * cluster_outputs / filter_outputs, extract_outputs) + the geometry block of post_process (net.py:195-215).  kps_l_host (ml,3,17),
 * kps_r_host (mr,3,17) and out_host are HOST buffers; out_host receives ml * ML_OUT_STRIDE floats (packed rows of the per-left
 * winners), ml * ML_POSTGEO_STRIDE floats (geometry), one int32 = number of left persons whose best aux logit is tied (the reference
 * keeps every tied pair row: the caller re-does such a frame through ml_loco_forward_stereo + ml_stereo_tied_rows), ml int32 = the
 * winning right index per left person.  Pinned (device-mapped) host buffers: no copy operation, the host polls the frame's completion
 * word; pageable ones are staged through kps_dev ((ml + mr) * 51 floats) and buf_dev (same size as out_host).  xyzds_dev (ml, 5). */
int ml_loco_frame_stereo(ml_loco* h, const float* kps_l_host, int64_t ml, const float* kps_r_host, int64_t mr,
                         const float* kinv_host, float* kps_dev, float* buf_dev, float* xyzds_dev, float* out_host, void* stream);
/* The pinned-buffer contract of the two frame entries above.  Kernels dereference kps*_host / out_host directly, so before the
 * first use of a host range both of its ends are asked of the runtime (hipPointerGetAttributes must say "host"); anything else --
 * pageable memory, a failed query -- takes the staged route (copies through kps_dev / buf_dev), never an error.  The verdict is
 * remembered per handle for the last 8 (pointer, byte extent) pairs, because a caller streams its frames through the same staging
 * buffers.  In return the CALLER guarantees: a host buffer that was passed to ml_loco_frame_mono / _stereo stays pinned (no
 * hipHostUnregister / hipHostFree / torch un-pinning of it) until either ml_loco_forget_pinned(h, that pointer) -- or
 * ml_loco_forget_pinned(h, NULL): every remembered range -- has been called, or the handle is destroyed.  Freeing a remembered
 * buffer without forgetting it and letting malloc reuse the address would put pageable memory behind a trusted pointer: a GPU page
 * fault, not an error code.  out_host must be coherent pinned memory (hipHostMalloc default flags / torch pin_memory: NOT
 * hipHostMallocNonCoherent) for the completion word the host polls to become visible; a frame that has not reported after 5 ms
 * falls back to hipStreamSynchronize.  (The reference has no such call: it copies Python lists into fresh tensors per frame,
 * monoloco/network/net.py:92-93.) */
int ml_loco_forget_pinned(ml_loco* h, const void* host_ptr);
/* stereo (net.py:112-122, process.py:307-327): all left x right pairs, per-left arg-max of the
 * aux logit.  best_dev (ml) int32 receives the first arg-max right index; ties_dev (1) int32
 * receives the number of left persons with more than one maximal pair (the reference keeps
 * all tied rows; the host side re-does those rare cases from raw_all_dev).  raw_all_dev
 * (ml*mr, out_features) may be NULL. */
int ml_loco_forward_stereo(ml_loco* h, const float* kps_l_dev, int64_t ml, const float* kps_r_dev,
                           int64_t mr, const float* kinv_host, const float* box_conf_dev,
                           float* raw_all_dev, float* out_dev, float* xyzds_dev,
                           int32_t* best_dev, int32_t* ties_dev, void* stream);

/* ---- MC-dropout epistemic uncertainty: stands in for Loco.epistemic_uncertainty (net.py:135-161) ---- */
/* n_dropout stochastic forwards of the mono model with dropout (probability p_dropout) active at the two
 * top-level sites only (net.py:141, architectures.py:53,66); per pass n_samples (reference: 100) draws of
 * Laplace(d, |exp(s) d|) with the same seed every pass (process.py:103,119-120); epi_dev (m) receives the
 * unbiased standard deviation over all n_dropout*n_samples draws.  Counter-based RNG: statistically, not
 * bitwise, equal to the reference's torch stream.  raw_passes_dev (n_dropout, m, out_features) optionally
 * receives every pass' raw network output.  Allocates a 16*m byte scratch with hipMallocAsync. */
int ml_loco_epistemic_mono(ml_loco* h, const float* kps_dev, int64_t m, const float* kinv_host, int n_dropout,
                           float p_dropout, int n_samples, uint32_t seed, float* epi_dev, float* raw_passes_dev,
                           void* stream);
/* The same on PRE-PROCESSED network inputs x_dev (m, 34): the argument Loco.epistemic_uncertainty(inputs) takes (net.py:135);
 * same passes, masks and draws as ml_loco_epistemic_mono on the keypoints those inputs came from. */
int ml_loco_epistemic_inputs(ml_loco* h, const float* x_dev, int64_t m, int n_dropout, float p_dropout, int n_samples,
                             uint32_t seed, float* epi_dev, float* raw_passes_dev, void* stream);
/* filter_outputs' mask as a row list (process.py:319-327): raw_all_dev (ml*mr, out_features) pair rows, i-major; rows_dev
 * (capacity ml*mr) receives, left person by left person, every pair row whose aux logit (last column) is >= the maximum over that
 * person's right candidates -- exact ties keep several rows, a NaN among the candidates keeps none, as in the reference --;
 * count_dev (1) the number of rows written.  ml_loco_forward_stereo reports whether any person is tied; this lists them. */
int ml_stereo_tied_rows(const float* raw_all_dev, int out_features, int64_t ml, int64_t mr, int32_t* rows_dev, int32_t* count_dev,
                        void* stream);

/* ---- training step: stands in for one iteration of Trainer.train (monoloco/train/trainer.py:150-161) ---- */
typedef struct ml_trainer ml_trainer;
/* LocoModel(in_features, out_features, hidden, p_dropout, num_stage) in train mode (trainer.py:115-123) with
 * Adam(lr) + StepLR(step_size = sched_step BATCHES, gamma = sched_gamma) (trainer.py:128-131, :160-161) and
 * clip_grad_norm_(3) (:159).  Parameters start at zero: feed them with ml_trainer_set_tensor.  seed drives the
 * dropout masks (counter-based RNG; p_dropout = 0 gives the deterministic path used for parity). */
int ml_trainer_create(int in_features, int hidden, int out_features, int num_stage, float p_dropout, float lr,
                      float sched_gamma, int sched_step, uint32_t seed, ml_trainer** out);
/* state_dict access by the reference's keys (parameters and BatchNorm running statistics).  `host_data` may also be a DEVICE
 * pointer of the trainer's device (the copy is hipMemcpyDefault): the autograd-capable module hands its CUDA parameters in and takes
 * the gradients out without a host round trip.  Each call synchronises the device. */
int ml_trainer_set_tensor(ml_trainer* t, const char* key, const float* host_data, int64_t numel);
int ml_trainer_get_tensor(ml_trainer* t, const char* key, float* host_data, int64_t numel);
/* The flat buffers behind the keyed access (parameters and their gradients in slot order in one allocation, the BatchNorm running
 * statistics in another): where `key` starts (elements; -1: unknown key; *is_param says which buffer), the two sizes, and ONE
 * stream-ordered device-to-device copy of a whole buffer -- what 0: parameters <- dev_ptr, 1: parameters -> dev_ptr, 2: gradients ->
 * dev_ptr, 3: statistics <- dev_ptr, 4: statistics -> dev_ptr.  (No reference counterpart: torch modules own their tensors; the
 * autograd-capable LocoModel moves its 62 tensors with two copies per iteration this way.) */
int64_t ml_trainer_flat_offset(const ml_trainer* t, const char* key, int* is_param);
int ml_trainer_flat_numel(const ml_trainer* t, int64_t* n_param, int64_t* n_stat);
int ml_trainer_copy_flat(ml_trainer* t, int what, float* dev_ptr, int64_t numel, void* stream);
/* gradient of the last step (after clipping when the step updated), parameters only */
int ml_trainer_get_grad(ml_trainer* t, const char* key, float* host_data, int64_t numel);
/* One step on a batch resident on the device: x_dev (m, in_features), labels_dev (m, label_cols) with the
 * reference's label columns theta, psi, z, d, h, w, l, sin, cos, yaw(, aux) (process.py:293-301).  Forward in
 * train mode (batch-statistics BatchNorm, running-stat update with momentum 0.1, dropout), MultiTaskLoss
 * (losses.py:59-73: LaplacianLoss on d :112-131, L1 on x, y, h, w, l, ori, BCE-with-logits on aux for 10 outputs),
 * backward; if update != 0: clip, Adam, advance the per-batch StepLR.  losses_host[0] = total, [1..8] = the task
 * means d, x, y, h, w, l, ori, aux.  raw_out_dev (m, out_features) optionally receives the train-mode outputs.
 * Synchronises the stream before returning (the reference syncs with .item() too). */
int ml_trainer_step(ml_trainer* t, const float* x_dev, const float* labels_dev, int label_cols, int64_t m,
                    int update, double* losses_host, float* raw_out_dev, void* stream);
/* The same iteration with the seams a caller-owned loop needs (reference trainer.py:155-160: `outputs = self.model(inputs)` in train
 * mode, the caller's own criterion, `loss.backward()`, the caller's own clip_grad_norm_ and optimizer): ml_trainer_forward_train =
 * LocoModel.forward in train mode (architectures.py:48-71) -> raw_out_dev (m, out_features), running BatchNorm statistics updated,
 * fresh dropout masks per call; ml_trainer_backward = backward from dout_dev (m, out_features), the gradient of the caller's loss with
 * respect to those outputs -> every parameter gradient, UNCLIPPED, through ml_trainer_get_grad.  x_dev must stay valid and unchanged
 * between the two calls; exactly one backward per forward, and no ml_trainer_step / ml_trainer_eval in between -- they run through the
 * same workspace and cancel the pending forward (ML_ERR_STATE from the backward); the gradient with respect to x is not computed.
 * Exact-fp32 route below fast_rows rows, the large-batch route from there on.  ml_trainer_backward synchronises the stream.
 * monoloco_amd.network.architectures.LocoModel routes its train-mode forward through these (torch.autograd.Function). */
int ml_trainer_forward_train(ml_trainer* t, const float* x_dev, int64_t m, float* raw_out_dev, void* stream);
int ml_trainer_backward(ml_trainer* t, const float* dout_dev, int64_t m, void* stream);
/* AutoTuneMultiTaskLoss (train/losses.py:17-43, selected by `--auto_tune_mtl`, trainer.py:95-96): every task loss is divided by
 * 2 exp(log_sigma)^2 and the log_sigmas are added to the total; the eight log_sigmas (d, x, y, h, w, l, ori, aux; zero at
 * creation) are optimised by the same Adam and schedule, unclipped, and are not part of the state_dict.  With it on,
 * ml_trainer_step reports the weighted task values and the total including the log_sigmas. */
int ml_trainer_set_auto_tune(ml_trainer* t, int enable);
/* Task weights of the multi-task loss (the reference's `Trainer.lambdas`, train/trainer.py:42, all 1 by default; order d, x, y,
 * h, w, l, ori, aux): loss = sum lambda_i * l_i (losses.py:66); reported task values are the weighted ones. */
int ml_trainer_set_lambdas(ml_trainer* t, const float* host8);
int ml_trainer_get_log_sigmas(ml_trainer* t, float* host8);
int ml_trainer_set_log_sigmas(ml_trainer* t, const float* host8);
int64_t ml_trainer_num_steps(const ml_trainer* t);
/* Which GEMM route a step takes, per handle (thread-compatible: nothing process-global).  route 0 = automatic: batches of at
 * least fast_rows rows (default 4096; hidden % 256 == 0) run the hidden x hidden products on the large-batch 3-product fp16
 * MFMA kernel (256 x 256 tiles, line-format operands); smaller ones (hidden % 64 == 0) take the "mid" route -- exact-fp32
 * MFMA GEMMs on 32 x 64 tiles that read the row-major fp32 tensors as they lie (weight and data gradients included), BatchNorm
 * statistics per column owner, ~50 launches per step (monoloco_amd/csrc/train_mid.h): the reference's real batch sizes
 * (run.py:95 --bs 512); anything else the generic exact-fp32 GEMM.  route 1 = generic exact-fp32 only, 2 = mid whenever it can
 * run, 3 = large-batch route whenever it can run.  fast_rows < 0 leaves the threshold unchanged. */
int ml_trainer_set_route(ml_trainer* t, int route, int64_t fast_rows);
/* route of the last ml_trainer_step: 0 exact, 1 large-batch, 2 mid (-1: no step yet) */
int ml_trainer_last_route(const ml_trainer* t);
/* The unweighted values of the last step's (train-mode) outputs the reference logs for the training phase as well
 * (trainer.py:163-165, losses.py:85-96): host10[0..7] = the plain task means d (Laplace), x, y, h, w, l, ori (L1), aux (BCE) --
 * ml_trainer_step reports them weighted when task weights are set --, host10[8] = mean |mu - d| (the validation-type value of d),
 * host10[9] = mean |atan2(out7, out8) - atan2(sin, cos)| in radians (that of ori). */
int ml_trainer_last_val_values(const ml_trainer* t, double* host10);
/* The validation pass of the reference's loop (trainer.py:167-178: model.eval(), no_grad) on the trainer's OWN weights: eval-mode
 * forward (BatchNorm with the running statistics, no dropout; nothing is updated) of a batch resident on the device, on the mid
 * route's kernels (hidden % 64 == 0; ML_ERR_SHAPE otherwise: use ml_loco_* on the state_dict).  vals_host (10): the means of the
 * training-type task terms d (Laplace), x, y, h, w, l, ori, aux, then the validation-type d (L1) and ori (angle, radians).
 * raw_out_dev (m, out_features) optionally receives the outputs.  Synchronises the stream. */
int ml_trainer_eval(ml_trainer* t, const float* x_dev, const float* labels_dev, int label_cols, int64_t m, double* vals_host,
                    float* raw_out_dev, void* stream);
/* 1 if ml_trainer_eval takes this trainer's shape (the predicate its ML_ERR_SHAPE is raised from), 0 otherwise. */
int ml_trainer_can_eval(const ml_trainer* t);
/* The statistics the reference's Trainer computes with torch from raw outputs and labels (trainer.py:163-165 epoch values,
 * :213-232 evaluate; losses.py:85-96; process.py:125-133 unnormalize_bi; trainer.py:384-389 get_accuracy), in one launch on rows
 * that are on the device anyway: raw_dev (m, 9|10), labels_dev (m, label_cols >= 10|11) -> vals_host (14):
 * [0..7] means of the training-type task terms d (Laplace), x, y, h, w, l, ori, aux (BCE); [8] mean |mu - d|; [9] mean angle error
 * (radians); [10] mean bi = exp(s) d; [11] share of rows with |mu - d| <= bi; [12] unbiased std of |mu - d| (torch .std());
 * [13] aux accuracy 1 - mean |[sigmoid(a) >= 0.5] - label| (0 for 9 outputs).  fp64 sums, fixed order.  Synchronises the stream. */
int ml_val_stats(const float* raw_dev, int out_features, const float* labels_dev, int label_cols, int64_t m, double* vals_host,
                 void* stream);
/* dst_dev (n, width) = rows idx_dev[0..n) (int64) of src_dev (*, width): the collation of one batch of the epoch's row
 * permutation (the reference's DataLoader does it on the host, trainer.py:150-152). */
int ml_gather_rows(const float* src_dev, int width, const int64_t* idx_dev, int64_t n, float* dst_dev, void* stream);
/* The best-epoch bookkeeping of the reference's loop (trainer.py:173-177, 183: deepcopy of the state_dict / load_state_dict) without
 * leaving the device: snapshot = parameters + BatchNorm running statistics copied aside (device to device), restore = copied back
 * (optimizer state untouched, like load_state_dict). */
int ml_trainer_snapshot(ml_trainer* t, void* stream);
int ml_trainer_restore(ml_trainer* t, void* stream);
/* Per-handle tuning of the mid route, same results: apply_cols = columns per workgroup of the column-owner kernels (4, 8 or 16;
 * default 8; 0 leaves it); side_stream: 0 (default) = everything on the caller's stream, the data gradient and the weight
 * gradient of a Linear in ONE launch (xgemm_pair_kernel); 1 = the weight-gradient GEMMs, which only the optimizer needs, on an
 * internal side stream beside the data-gradient chain (events both ways; measured to pay from ~2000 rows); 2 = two launches per
 * Linear on the caller's stream (the A/B reference of 0); < 0 leaves it.  dw_layout (large-batch route): 1 (default) = the
 * weight-gradient GEMM dW = dz^T . x reads dz and x reduction-major, as the [batch][hidden] lines they already exist as
 * (gfx950's transposing LDS read feeds the MFMA), and dz / a stage's inner activation / the residual stream exist as lines only;
 * 0 = through transposed copies of both operands with the fp32 chain of rounds 2-3 (2.2 ms more per 65536-row step); 2 = the
 * reduction-major GEMM on that same fp32 chain (same bits as 0: the test's reference); < 0 leaves it.  Round 6: with layout 1 the pair
 * w2 -> w3 (reference architectures.py:60-66, nothing non-linear between them) runs as ONE Linear -- z3 = a (W3 W2)^T + (W3 b2 + b3),
 * aux = a (W2^T w_aux) + (w_aux . b2 + b_aux); dW3 = (dz3^T a) W2^T + s3 (x) b2, dW2 = W3^T (dz3^T a) + w_aux (x) (daux^T a),
 * da = dz3 (W3 W2) + daux (x) (W2^T w_aux): three batch-sized GEMMs instead of six, y2 and its gradient never exist; the H x H
 * products of weights on the exact-fp32 GEMM -- 3 = layout 1 with the two Linears apart (rounds 4-5; the A/B reference). */
int ml_trainer_set_tuning(ml_trainer* t, int apply_cols, int side_stream, int dw_layout);
int ml_trainer_destroy(ml_trainer* t);
const char* ml_train_last_error(void);

/* ---- on-disk formats either side of the path (host only; no GPU, no HIP calls) ---------- */
/* Counts the annotation objects of an OpenPifPaf `*.predictions.json` text (a top-level array). */
int ml_pifpaf_count(const char* json, int64_t len, int64_t* n_annotations);
/* json.load + preprocess_pifpaf (monoloco/network/process.py:155-207, prepare_pif_kps :210-218) in one pass
 * over the JSON text: per annotation 'keypoints' (51 numbers) -> [xs(17), ys(17), cs(17)]; 'bbox' grown by
 * h/10, w/5 (x,y,w,h boxes, when a 'score' key exists) or by 1/7, 1/3.5 of its height/width (corner boxes,
 * conf = numpy float64 mean of the 17 confidences), halved when enlarge_boxes == 0, clamped to
 * (im_w, im_h) when has_im_size != 0; annotations with conf < min_conf are dropped.  boxes (cap,5) =
 * x1,y1,x2,y2,conf and keypoints (cap,3,17) are doubles, bit-identical to the Python floats of the
 * reference; *m = rows written.  Errors: ML_ERR_ARG (syntax, missing key, "Bounding box <=0", cap too
 * small), ML_ERR_SHAPE (wrong keypoint / bbox count); text in ml_formats_last_error(). */
int ml_pifpaf_parse(const char* json, int64_t len, int has_im_size, double im_w, double im_h, int enlarge_boxes,
                    double min_conf, int64_t cap, double* boxes, double* keypoints, int64_t* m);
/* The lines save_txts writes for one image (monoloco/eval/generate_kitti.py:202-253):
 * "<Pedestrian|Cyclist> -1 -1 alpha x1 y1 x2 y2 h w l x y z ry conf bi epi \n", every number as "%f ".
 * boxes (m,5) as returned by preprocess_pifpaf; xyz (m,3); tt (3) is subtracted from xyz when not NULL;
 * zz_override (m) replaces z when not NULL ('geometric'); alpha / ry NULL -> -10; hwl (m,3) NULL -> 0;
 * cat (m): < 0.1 is a pedestrian; conf = conf_scale * box_conf / (bi / |xyz|).  With out == NULL only
 * the needed size is returned in *written. */
int ml_kitti_txt_format(int64_t m, const double* boxes, const double* xyz, const double* bi, const double* epi,
                        const double* alpha, const double* ry, const double* hwl, const double* zz_override,
                        const double* tt, const double* cat, double conf_scale, char* out, int64_t cap,
                        int64_t* written);
const char* ml_formats_last_error(void);

/* ---- ground-truth association of Loco.post_process (monoloco/network/net.py:170-190) ---- */
/* All of this section is IEEE double in the evaluation order of the reference's Python expressions (Python floats are
 * doubles), with Python's max / min and np.argmax tie rules, so the matches equal the reference's exactly.  Boxes are rows
 * of >= 4 doubles x1, y1, x2, y2 (detections carry their confidence in column 4; only columns 0..3 are read here);
 * ldb / ldg are the row strides in doubles.  *zero_div is OR-ed with 1 when some union area is 0 -- the reference's
 * calculate_iou raises ZeroDivisionError there (monoloco/utils/iou.py:25), and the host side re-raises it.
 * The device entry points are asynchronous on `stream` and allocate nothing; the *_host ones are plain host loops for
 * images with a handful of boxes (where the launch and the copies would cost more than the m*g IoUs).
 * Error text: ml_matching_last_error(). */
/* Per detection the first arg-max over all ground-truth boxes of calculate_iou (monoloco/utils/iou.py:6-28, the inner loop
 * and np.argmax of get_iou_matches, iou.py:55-60): jmax (m) int32, vmax (m) double.  g must be >= 1. */
int ml_iou_best(const double* boxes_dev, int64_t m, int64_t ldb, const double* gt_dev, int64_t g, int64_t ldg,
                int32_t* jmax_dev, double* vmax_dev, int32_t* zero_div_dev, void* stream);
int ml_iou_best_host(const double* boxes, int64_t m, int64_t ldb, const double* gt, int64_t g, int64_t ldg,
                     int32_t* jmax, double* vmax, int32_t* zero_div);
/* get_iou_matrix (monoloco/utils/iou.py:31-41): out (m, g) row-major doubles. */
int ml_iou_matrix(const double* boxes_dev, int64_t m, int64_t ldb, const double* gt_dev, int64_t g, int64_t ldg,
                  double* out_dev, int32_t* zero_div_dev, void* stream);
int ml_iou_matrix_host(const double* boxes, int64_t m, int64_t ldb, const double* gt, int64_t g, int64_t ldg, double* out,
                       int32_t* zero_div);
/* The greedy pass of get_iou_matches (monoloco/utils/iou.py:53-63) over host arrays: visit detections in `order` (n entries,
 * each in 0..m-1; the caller passes reversed(np.argsort(confidences)) so that ties keep numpy's order), match a detection to
 * its best ground-truth box jmax[idx] when vmax[idx] >= iou_min and that box is still free.  pairs (min(n,g), 2) int64 =
 * (idx, idx_gt), *n_pairs = matches found.  order_left == NULL: pairs in visiting order (get_iou_matches' result);
 * otherwise order_left (m) = np.argsort(left edges) and the pairs come out left to right -- reorder_matches
 * (monoloco/utils/iou.py:86-100) applied to that result, as Loco.post_process does (net.py:187-188). */
int ml_iou_greedy(const int64_t* order, int64_t n, const int32_t* jmax, const double* vmax, int64_t m, int64_t g,
                  double iou_min, const int64_t* order_left, int64_t* pairs, int64_t* n_pairs);
/* get_iou_matches (monoloco/utils/iou.py:44-64) [+ reorder_matches] in one host call: ml_iou_best_host + ml_iou_greedy. */
int ml_iou_matches_host(const double* boxes, int64_t m, int64_t ldb, const double* gt, int64_t g, int64_t ldg,
                        const int64_t* order, double iou_min, const int64_t* order_left, int64_t* pairs, int64_t* n_pairs,
                        int32_t* zero_div);
/* xyz_from_distance (utils/camera.py:161-177) on HOST arrays -- the ground-truth side of the matched persons of a frame
 * (net.py:242-247), whose normalised centres are already on the host inside the geometry block: d (m) (or one value), centres (m,3)
 * -> out (m,3); fp32, the reference's operation order, same bits as ml_xyz_from_distance. */
int ml_xyz_from_distance_host(const float* d, int d_is_scalar, const float* centres, int64_t m, float* out);
const char* ml_matching_last_error(void);

/* ---- measurement: per-launch timing of the dense (MFMA) kernel ----------------------- */
/* After ml_loco_profile_begin, every dense-kernel launch made through this handle is bracketed by
 * a pair of HIP events recorded on the launch stream (up to max_launches launches).
 * ml_loco_profile_end stops recording, waits for the events and returns the number of recorded
 * launches, their summed duration, and per-dense-layer sums/counts (arrays of n_layers, may be
 * NULL).  bench.py uses this for the live roofline figure. */
int ml_loco_profile_begin(ml_loco* h, int max_launches);
int ml_loco_profile_end(ml_loco* h, int64_t* launches, double* total_ms, double* per_layer_ms,
                        int64_t* per_layer_n, int n_layers);

/* ---- test hooks (exercise single kernels / host packing; used by tests/ only) ------ */
/* Dense layer on its own: y = [relu](x . W^T + b) [+ res]; x (m,k), w (n,k), b (n), res (m,n)
 * or NULL, y (m,n); all fp32 device pointers except w/b which are host.  k, n: n multiple of 256.
 * The 256x256-tile kernel runs it unless ML_DEBUG_SMALL_PATH is or-ed into `precision` (then the
 * small-row kernels the model path takes for <= 512 rows). */
#define ML_DEBUG_SMALL_PATH 256
#define ML_DEBUG_TILE_PP 512    /* ... the tile path on dense_kernel_pp */
#define ML_DEBUG_TILE_W4 1024   /* ... the tile path on dense_kernel_w4 wherever it runs (default: w4 for K > 128) */
#define ML_DEBUG_MID_64 2048    /* ... dense_mid_kernel with 128 x 64 tiles */
#define ML_DEBUG_MID_128 4096   /* ... dense_mid_kernel with 128 x 128 tiles */
#define ML_DEBUG_MID_SPLIT2 8192   /* ... its reduction cut into 2 k ranges per output tile (split-K, the last arriver runs the epilogue) */
#define ML_DEBUG_MID_SPLIT4 16384  /* ... into 4 */
#define ML_DEBUG_MID_NODMA 32768   /* ... dense_mid_kernel with the rounds-3-5 loader (global -> VGPR -> ds_write) instead of LDS-DMA */
int ml_debug_linear(const float* x_dev, int64_t m, int k, const float* w_host, const float* b_host,
                    int n, int relu, const float* res_dev, float* y_dev, int precision, void* stream);
/* Host fp32 -> fp16 hi/lo split used by the packer (round-to-nearest-even), for unit tests. */
int ml_debug_split_f16(const float* host_in, int64_t n, uint16_t* host_hi, uint16_t* host_lo);
/* Copy back the folded (and merged) fp32 weight/bias of dense layer `layer` after finalize:
 * w_host (n*k), b_host (n); returns n and k through the pointers; scale_pow2 = weight exponent. */
int ml_debug_get_layer(const ml_loco* h, int layer, float* w_host, float* b_host, int* n, int* k,
                       int* scale_pow2);
int ml_debug_num_layers(const ml_loco* h);
/* How many ml_loco_frame_mono calls of this process ran without any copy operation (pinned buffers, <= 128 persons). */
long long ml_debug_frames_without_copies(void);
/* ml_loco_frame_mono, single image in pinned buffers: the host learns of the frame's end from a word of pinned memory the last
 * launch releases (busy-polled for at most 5 ms, then hipStreamSynchronize) instead of sleeping in the stream synchronisation.
 * enable 0 / 1 turns that off / on for the process (< 0: unchanged); returns how many frames have timed out so far (expected 0). */
long long ml_debug_frame_spin(int enable);
/* Which dense kernel family a forward of `rows` network rows takes on this handle (its precision and tuning): one of the
 * ML_ROUTE_* codes below (-1: bad argument).  Reporting only (bench.py labels its per-batch-size lines with it). */
#define ML_ROUTE_SMALL16 0 /* dense_small_kernel, 16 x 16 output tiles (a single image) */
#define ML_ROUTE_SMALL32 1 /* dense_small32_kernel */
#define ML_ROUTE_MID64 2   /* dense_mid_kernel, 128 x 64 workgroup tiles */
#define ML_ROUTE_MID128 3  /* dense_mid_kernel, 128 x 128 */
#define ML_ROUTE_HALF 4    /* dense_kernel_w4 with its half-size 256 x 128 tile for the long-K layers (dense_mid_kernel for the input layer) */
#define ML_ROUTE_TILE 5    /* the persistent 256 x 256-tile kernels (dense_kernel_w4 / dense_kernel_pp) with fused heads */
int ml_loco_route(const ml_loco* h, int64_t rows);
/* The launch plan of a forward of `rows` network rows on this handle, as text -- the very plan the forward executes (one
 * make_plan in the library decides, run_network only walks it): "route=<family>; L<i> <kernel>[+fin8|+fin9][+aux][+dropout]
 * [ heads<n>] ...; end=<tail_mono | reduce | heads_pair[+post] | heads_small[+post] | heads>".  +fin / +aux: that head leaves the
 * layer's epilogue as partial sums (its activation is never re-read); heads<n>: a head launched on its own behind that layer.
 * mc_dropout != 0: the plan of a stochastic (MC-dropout) pass; with_post != 0: as called from ml_loco_forward_mono (the
 * post-process may ride in the last launch).  Tests assert the fusion state per row count with it. */
int ml_loco_plan(const ml_loco* h, int64_t rows, int mc_dropout, int with_post, char* text, int64_t cap);
/* Named switches of the route plan of ONE handle (A/B runs, tests): "mid_heads" (default 1; alias "half_heads"): in the mid window
 * both heads ride in the dense epilogues (dense_mid_kernel's, or the half-size dense_kernel_w4 tile's) and tail_mono_kernel ends the
 * call; 0 = heads_pair_kernel behind the last layer (rounds 3-4); "half_from" (default 4096): the half-size tile takes the long-K
 * layers of calls with more rows than this (inside the mid window); "small_multi" (default 1): small-row calls of more than 64 rows
 * run dense_small_multi_kernel (a workgroup keeps its 16 weight rows and walks several row tiles), 0 = one tile per workgroup. */
int ml_loco_set_option(ml_loco* h, const char* name, int value);
/* Path selection of ONE handle, for tests / A-B runs that compare the paths (negative = leave unchanged; defaults 512 / 128 /
 * 0 / 4): rows <= small_rows take the small-row dense kernels, above small32_rows those use 32x32 tiles; chunk_rows > 0 walks
 * the batch in row chunks of that size through all layers; tile_kernel: 4 = dense_kernel_w4 for the long-K layers,
 * dense_kernel_pp for the short input layer and the fused-head layer (default), 2 = dense_kernel_pp everywhere, 4 | 256 =
 * dense_kernel_w4 wherever it can run; small_rows < rows <= mid_rows (default 8192) take dense_mid_kernel, whose tile
 * height mid_tile is 0 (chosen from the row count, default), 64 or 128.  Nothing here is process-global: handles stay
 * thread-compatible. */
int ml_loco_set_tuning(ml_loco* h, int small_rows, int small32_rows, int chunk_rows, int tile_kernel, int mid_rows,
                       int mid_tile);
/* Training, bring-up: copy an internal fp32 buffer of the trainer to the host (after a device sync).  which: 0 .. 4S+7 the
 * (rows x hidden) activation / gradient buffers in allocation order (a_0..a_S, t_0.., z0, (za, zb)_s, z3, y2, y3, scratch,
 * gA, gB), 200 / 201 the raw outputs / their gradient. */
int ml_trainer_debug_read(ml_trainer* t, int which, float* host_data, int64_t numel);
/* The mid route's exact-fp32 MFMA GEMM on its own: c (M, N) = sum_k A(i, k) B(j, k) (+ bias (N)) (+ res (M, N)).
 * layout 0: the operand is k-contiguous (A(i, k) = a[i * lda + k]; K % 32 == 0), layout 1: reduction-major (A(i, k) =
 * a[k * lda + i]; for A: M % 32 == 0; rows K .. ceil32(K) of such an operand must be readable, they contribute 0).  N % 64 == 0.
 * sumsq_dev: optional (N / 64) * ceil(M / 32) doubles, the per-workgroup sums of squares of c.  All device pointers. */
int ml_debug_xgemm(const float* a_dev, int64_t lda, int a_layout, const float* b_dev, int64_t ldb, int b_layout, float* c_dev, int M,
                   int N, int K, const float* bias_dev, const float* res_dev, double* sumsq_dev, void* stream);
/* Packed fp16 hi|lo line image of a dense layer's weights (n * kpad * 2 uint16); only kept for
 * models finalized with ML_FLAG_HOST_ONLY. */
int ml_debug_get_packed(const ml_loco* h, int layer, uint16_t* lines_host, int64_t capacity);
/* Head `head` (0 = aux, 1 = w_fin): weights (nh*hidden), bias (nh), first raw column, the
 * activation buffer it reads and the dense layer it runs after. */
int ml_debug_get_head(const ml_loco* h, int head, float* w_host, float* b_host, int* nh, int* col0, int* src_buf,
                      int* after_layer);

#ifdef __cplusplus
}
#endif
#endif /* MONOLOCO_HIP_H */
