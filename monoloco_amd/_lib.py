"""ctypes binding of the C ABI declared in ``include/monoloco_hip.h``.

The shared library is built in-tree by ``__graft_entry__.build()`` (or ``make -C
monoloco_amd/csrc``) into ``monoloco_amd/lib/libmonoloco_hip.so``.  There is no
fallback: if the library is missing or a call fails, a ``MonolocoHipError`` is raised.
"""
import ctypes
import os

# PyTorch-ROCm bundles its own HIP/HSA runtime (torch/lib/libamdhip64.so).  It has to be in the process BEFORE
# this library is dlopen'ed: the library's NEEDED libamdhip64.so.7 then binds to that already-loaded runtime and
# both sides share one device context.  Loaded the other way round, /opt/rocm's runtime comes in first, torch
# later brings a second HSA runtime and HIP reports "no ROCm-capable device" (seen on the GPU box).
import torch  # noqa: F401  (import order matters, see above)
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint16, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MONOLOCO_HIP_LIB: another build of the same library (tools/ use it for the -DML_BRINGUP ablation build)
LIB_PATH = os.environ.get('MONOLOCO_HIP_LIB') or os.path.join(_HERE, 'lib', 'libmonoloco_hip.so')

ML_PREC_F16X2 = 0
ML_PREC_F16 = 1
ML_PREC_BF16 = 2
ML_DEBUG_SMALL_PATH = 256
ML_DEBUG_TILE_PP = 512
ML_DEBUG_TILE_W4 = 1024
ML_DEBUG_MID_64 = 2048
ML_DEBUG_MID_128 = 4096
ML_DEBUG_MID_SPLIT2 = 8192
ML_DEBUG_MID_SPLIT4 = 16384
ML_DEBUG_MID_NODMA = 32768
ML_FLAG_MERGE_W2W3 = 1
ML_FLAG_HOST_ONLY = 256
ML_OUT_STRIDE = 16
ML_XYZDS_STRIDE = 5
OUT_COLS = dict(x=0, y=1, z=2, d=3, bi=4, yaw=5, yaw_ego=6, aux=7, h=8, w=9, l=10, conf=11,
                ori0=12, ori1=13, uc=14, vc=15)


class MonolocoHipError(RuntimeError):
    """Raised when the HIP library is missing or one of its entry points reports an error."""


# name -> (restype, argtypes); must list every symbol of include/monoloco_hip.h
_P = c_void_p
SIGNATURES = {
    'ml_version': (c_int, []),
    'ml_last_error': (c_char_p, []),
    'ml_device_count': (c_int, []),
    'ml_loco_create': (c_int, [c_int, c_int, c_int, c_int, POINTER(_P)]),
    'ml_loco_set_tensor': (c_int, [_P, c_char_p, POINTER(c_float), c_int64]),
    'ml_loco_finalize': (c_int, [_P, c_int, c_int]),
    'ml_loco_reserve': (c_int, [_P, c_int64]),
    'ml_loco_destroy': (c_int, [_P]),
    'ml_loco_device_bytes': (c_int64, [_P]),
    'ml_preprocess_mono': (c_int, [_P, c_int64, POINTER(c_float), c_float, c_int, _P, _P, _P]),
    'ml_stereo_pairs': (c_int, [_P, c_int64, _P, c_int64, _P, _P]),
    'ml_extract_outputs': (c_int, [_P, c_int, _P, c_int64, _P, POINTER(c_float), _P, _P, _P, _P]),
    'ml_post_geometry': (c_int, [_P, c_int64, POINTER(c_float), _P, _P, _P]),
    'ml_post_geometry_strided': (c_int, [_P, c_int64, POINTER(c_float), _P, c_int64, _P, _P]),
    'ml_preprocess_rows': (c_int, [_P, _P, c_int64, POINTER(c_float), c_int, _P, c_float, _P, _P]),
    'ml_extract_outputs_mono': (c_int, [_P, c_int64, _P, _P]),
    'ml_laplace_sampling': (c_int, [_P, c_int64, c_int, c_uint32, _P, _P]),
    'ml_pixel_to_camera': (c_int, [_P, c_int64, POINTER(c_float), c_float, _P, _P]),
    'ml_get_keypoints': (c_int, [_P, c_int64, c_int, _P, _P]),
    'ml_xyz_from_distance': (c_int, [_P, c_int, _P, c_int64, _P, _P]),
    'ml_to_cartesian': (c_int, [_P, c_int64, c_int, _P, _P]),
    'ml_back_correct_angles': (c_int, [_P, _P, c_int64, _P, _P]),
    'ml_loco_forward_raw': (c_int, [_P, _P, c_int64, _P, _P]),
    'ml_loco_forward_mono': (c_int, [_P, _P, c_int64, POINTER(c_float), _P, _P, _P, _P, _P]),
    'ml_debug_frames_without_copies': (ctypes.c_longlong, []),
    'ml_debug_frame_spin': (ctypes.c_longlong, [c_int]),
    'ml_loco_frame_mono': (c_int, [_P, _P, c_int64, POINTER(c_float), _P, _P, _P, _P, _P]),
    'ml_loco_forget_pinned': (c_int, [_P, _P]),
    'ml_loco_frame_stereo': (c_int, [_P, _P, c_int64, _P, c_int64, POINTER(c_float), _P, _P, _P, _P, _P]),
    'ml_loco_forward_stereo': (c_int, [_P, _P, c_int64, _P, c_int64, POINTER(c_float), _P, _P, _P, _P,
                                       _P, _P, _P]),
    'ml_loco_epistemic_mono': (c_int, [_P, _P, c_int64, POINTER(c_float), c_int, c_float, c_int, c_uint32, _P, _P, _P]),
    'ml_loco_epistemic_inputs': (c_int, [_P, _P, c_int64, c_int, c_float, c_int, c_uint32, _P, _P, _P]),
    'ml_stereo_tied_rows': (c_int, [_P, c_int, c_int64, c_int64, _P, _P, _P]),
    'ml_trainer_can_eval': (c_int, [_P]),
    'ml_val_stats': (c_int, [_P, c_int, _P, c_int, c_int64, POINTER(c_double), _P]),
    'ml_gather_rows': (c_int, [_P, c_int, _P, c_int64, _P, _P]),
    'ml_trainer_create': (c_int, [c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int, c_uint32, POINTER(_P)]),
    'ml_trainer_set_tensor': (c_int, [_P, c_char_p, POINTER(c_float), c_int64]),
    'ml_trainer_get_tensor': (c_int, [_P, c_char_p, POINTER(c_float), c_int64]),
    'ml_trainer_get_grad': (c_int, [_P, c_char_p, POINTER(c_float), c_int64]),
    'ml_trainer_set_auto_tune': (c_int, [_P, c_int]),
    'ml_trainer_set_lambdas': (c_int, [_P, POINTER(c_float)]),
    'ml_trainer_get_log_sigmas': (c_int, [_P, POINTER(c_float)]),
    'ml_trainer_set_log_sigmas': (c_int, [_P, POINTER(c_float)]),
    'ml_trainer_step': (c_int, [_P, _P, _P, c_int, c_int64, c_int, POINTER(c_double), _P, _P]),
    'ml_trainer_flat_offset': (c_int64, [_P, c_char_p, POINTER(c_int)]),
    'ml_trainer_flat_numel': (c_int, [_P, POINTER(c_int64), POINTER(c_int64)]),
    'ml_trainer_copy_flat': (c_int, [_P, c_int, _P, c_int64, _P]),
    'ml_trainer_forward_train': (c_int, [_P, _P, c_int64, _P, _P]),
    'ml_trainer_backward': (c_int, [_P, _P, c_int64, _P]),
    'ml_trainer_num_steps': (c_int64, [_P]),
    'ml_trainer_set_route': (c_int, [_P, c_int, c_int64]),
    'ml_trainer_last_route': (c_int, [_P]),
    'ml_trainer_last_val_values': (c_int, [_P, POINTER(c_double)]),
    'ml_trainer_eval': (c_int, [_P, _P, _P, c_int, c_int64, POINTER(c_double), _P, _P]),
    'ml_trainer_snapshot': (c_int, [_P, _P]),
    'ml_trainer_restore': (c_int, [_P, _P]),
    'ml_trainer_debug_read': (c_int, [_P, c_int, POINTER(c_float), c_int64]),
    'ml_trainer_set_tuning': (c_int, [_P, c_int, c_int, c_int]),
    'ml_debug_xgemm': (c_int, [_P, c_int64, c_int, _P, c_int64, c_int, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    'ml_trainer_destroy': (c_int, [_P]),
    'ml_train_last_error': (c_char_p, []),
    'ml_pifpaf_count': (c_int, [c_char_p, c_int64, POINTER(c_int64)]),
    'ml_pifpaf_parse': (c_int, [c_char_p, c_int64, c_int, c_double, c_double, c_int, c_double, c_int64,
                                POINTER(c_double), POINTER(c_double), POINTER(c_int64)]),
    'ml_kitti_txt_format': (c_int, [c_int64] + [POINTER(c_double)] * 10 + [c_double, _P, c_int64, POINTER(c_int64)]),
    'ml_formats_last_error': (c_char_p, []),
    'ml_iou_best': (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, _P, _P, _P]),
    'ml_iou_best_host': (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, _P, _P]),
    'ml_iou_matrix': (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, _P, _P]),
    'ml_iou_matrix_host': (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, _P]),
    'ml_iou_greedy': (c_int, [_P, c_int64, _P, _P, c_int64, c_int64, c_double, _P, _P, _P]),
    'ml_iou_matches_host': (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_double, _P, _P, _P, _P]),
    'ml_xyz_from_distance_host': (c_int, [_P, c_int, _P, c_int64, _P]),
    'ml_matching_last_error': (c_char_p, []),
    'ml_loco_profile_begin': (c_int, [_P, c_int]),
    'ml_loco_profile_end': (c_int, [_P, POINTER(c_int64), POINTER(c_double), POINTER(c_double), POINTER(c_int64), c_int]),
    'ml_debug_linear': (c_int, [_P, c_int64, c_int, POINTER(c_float), POINTER(c_float), c_int, c_int, _P, _P,
                                c_int, _P]),
    'ml_debug_split_f16': (c_int, [POINTER(c_float), c_int64, POINTER(c_uint16), POINTER(c_uint16)]),
    'ml_debug_get_layer': (c_int, [_P, c_int, POINTER(c_float), POINTER(c_float), POINTER(c_int), POINTER(c_int),
                                   POINTER(c_int)]),
    'ml_debug_num_layers': (c_int, [_P]),
    'ml_loco_route': (c_int, [_P, c_int64]),
    'ml_loco_set_option': (c_int, [_P, c_char_p, c_int]),
    'ml_loco_plan': (c_int, [_P, c_int64, c_int, c_int, ctypes.c_char_p, c_int64]),
    'ml_loco_set_tuning': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int]),
    'ml_debug_get_packed': (c_int, [_P, c_int, POINTER(c_uint16), c_int64]),
    'ml_debug_get_head': (c_int, [_P, c_int, POINTER(c_float), POINTER(c_float), POINTER(c_int), POINTER(c_int),
                                  POINTER(c_int), POINTER(c_int)]),
}

_lib = None


def load():
    """Load the shared library once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MonolocoHipError(
            "HIP library not built: %s is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C monoloco_amd/csrc` (needs hipcc); there is no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # e.g. libamdhip64 not found
        raise MonolocoHipError("cannot load %s: %s" % (LIB_PATH, exc)) from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library ever diverge
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, train=False):
    if code != 0:
        msg = load().ml_train_last_error() if train else load().ml_last_error()
        raise MonolocoHipError("monoloco_hip error %d: %s" % (code, msg.decode() if msg else '?'))


def fptr(array):
    """float32 numpy array -> POINTER(c_float) (array must stay alive during the call)."""
    return array.ctypes.data_as(POINTER(c_float))
