"""ctypes binding of include/llava_reward_hip.h (no torch types cross this boundary)."""
from __future__ import annotations

import ctypes as C
import os

from ._build import LIB_PATH

LR_DT_BF16, LR_DT_F16, LR_DT_F32 = 0, 1, 2
LR_FWD_TRAINING_LAST_TOKEN = 1
LR_FWD_NO_FINAL_NORM = 2
LR_FWD_KEEP_HIDDEN_STATES = 4
LR_MAX_HALF_HEAD = 64
LR_MAX_PINPOINTS = 8
LR_MAX_FULLATT = 8
LR_BACKBONE_PHI3V, LR_BACKBONE_LLAVA_NEXT, LR_BACKBONE_QWEN2_5_VL = 0, 1, 2
LR_ABI_VERSION = 9
EPI_OUT_OP, EPI_OUT_F32, EPI_RESADD_F32, EPI_SWIGLU_OP, EPI_ROPE_OP = 0, 1, 2, 3, 4
ACT_NONE, ACT_QUICK_GELU, ACT_GELU_ERF = 0, 1, 2


class ModelDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("vocab_size", C.c_int32), ("hidden", C.c_int32), ("intermediate", C.c_int32),
        ("layers", C.c_int32), ("heads", C.c_int32),
        ("rms_eps", C.c_float),
        ("orig_max_pos", C.c_int32),
        ("rope_scaling", C.c_float),
        ("inv_freq_short", C.c_float * LR_MAX_HALF_HEAD),
        ("inv_freq_long", C.c_float * LR_MAX_HALF_HEAD),
        ("clip_hidden", C.c_int32), ("clip_heads", C.c_int32), ("clip_mlp", C.c_int32),
        ("clip_layers", C.c_int32), ("clip_image", C.c_int32), ("clip_patch", C.c_int32),
        ("clip_ln_eps", C.c_float),
        ("value_head_dim", C.c_int32), ("add_cross_attention", C.c_int32),
        ("ca_eps", C.c_float),
        ("max_batch", C.c_int32), ("max_seq", C.c_int32), ("max_crops", C.c_int32),
        ("operand_dtype", C.c_int32),
        ("backbone", C.c_int32), ("kv_heads", C.c_int32), ("head_dim", C.c_int32), ("image_token_id", C.c_int32),
        ("n_pinpoints", C.c_int32), ("pinpoints", C.c_int32 * (2 * LR_MAX_PINPOINTS)),
        ("vit_depth", C.c_int32), ("vit_hidden", C.c_int32), ("vit_heads", C.c_int32), ("vit_intermediate", C.c_int32),
        ("vit_patch", C.c_int32), ("vit_temporal_patch", C.c_int32), ("vit_merge", C.c_int32), ("vit_window", C.c_int32),
        ("vit_in_ch", C.c_int32),
        ("vit_n_fullatt", C.c_int32), ("vit_fullatt", C.c_int32 * LR_MAX_FULLATT),
        ("vit_rope_theta", C.c_float), ("vit_eps", C.c_float),
        ("mrope_section", C.c_int32 * 3),
        ("ca_token_id", C.c_int32), ("max_patches", C.c_int32),
        ("precise", C.c_int32),
        ("mean_hidden_state", C.c_int32),
        ("w8a8", C.c_int32),
        ("lora_rank", C.c_int32),
        ("rope_flash_convention", C.c_int32),
    ]


_lib = None

_SIGS = {
    "lr_abi_version": (C.c_int, []),
    "lr_create": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "lr_destroy": (C.c_int, [C.c_void_p]),
    "lr_last_error": (C.c_char_p, [C.c_void_p]),
    "lr_upload_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int]),
    "lr_synth_weights": (C.c_int, [C.c_void_p, C.c_uint64]),
    "lr_synth_weights_ex": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int]),
    "lr_num_weights": (C.c_int, [C.c_void_p]),
    "lr_weight_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "lr_finalize": (C.c_int, [C.c_void_p]),
    "lr_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "lr_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64),
                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "lr_forward_qwen": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64),
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "lr_last_hidden_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "lr_vision_embeds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.c_void_p]),
    "lr_read_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "lr_set_layer_limits": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "lr_set_precision_map": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "lr_set_precision_sites": (C.c_int, [C.c_void_p] + [C.c_int] * 5),
    "lr_set_attention_lazy_threshold": (C.c_int, [C.c_void_p, C.c_float, C.c_float]),
    "lr_weights_epoch": (C.c_uint64, [C.c_void_p]),
    "lr_set_gemm_tile": (C.c_int, [C.c_void_p, C.c_int]),
    "lr_op_gemm_bt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 10 + [C.c_void_p]),
    "lr_op_gemm_bt_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]),
    "lr_op_gemm_bt_ext": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 8 + [C.c_void_p]),
    "lr_op_gemm_rope": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_void_p]),
    "lr_op_attention": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 11 + [C.c_float, C.c_int, C.c_void_p]),
    "lr_op_attention_split": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 13 + [C.c_float, C.c_int, C.c_void_p]),
    "lr_op_attention_split_ex": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 13 + [C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "lr_op_attention_segments": (C.c_int, [C.c_void_p] * 4 + [C.POINTER(C.c_int32)] + [C.c_int] * 8 + [C.c_float, C.c_int, C.c_void_p]),
    "lr_op_quantize_rows_fp8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "lr_op_gemm_fp8": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 7 + [C.c_void_p]),
    "lr_op_gemm_bt_mixed": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_void_p]),
    "lr_op_lo8_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "lr_op_norm_rows": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "lr_op_synth_fill": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64, C.c_char_p, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "lr_hd_transform_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "lr_qwen_image_grid": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    "lr_qwen_image_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int64, C.c_int64]),
    "lr_qwen_image_transform": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_int64),
                                          C.c_void_p, C.c_size_t, C.c_void_p]),
    "lr_llava_image_geometry": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32)]),
    "lr_llava_image_workspace": (C.c_size_t, [C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int]),
    "lr_llava_image_transform": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_void_p,
                                           C.POINTER(C.c_int64), C.c_void_p, C.c_size_t, C.c_void_p]),
    "lr_hd_transform": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
}

EXPORTS = tuple(_SIGS)


def load(path: str = None):
    """Load the HIP library; fails loudly (no CPU fallback exists for the product path)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("LLAVA_REWARD_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build it with `python __graft_entry__.py build` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback for the scoring path")
    lib = C.CDLL(p)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)        # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


class HipError(RuntimeError):
    pass


def check(lib, rc: int, handle=None, what: str = ""):
    if rc != 0:
        msg = lib.lr_last_error(handle)
        raise HipError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
