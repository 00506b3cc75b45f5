"""Python handle on the HIP scoring engine.  PyTorch is used only for device memory and streams."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Iterable, Optional

import numpy as np
import torch

from . import _lib as L
from .synth import LlavaConfig, QwenConfig, RewardConfig

_DT = {"bf16": L.LR_DT_BF16, "f16": L.LR_DT_F16, "fp16": L.LR_DT_F16, "f16x2": L.LR_DT_F16, "bf16x2": L.LR_DT_BF16, "f16x2f8": L.LR_DT_F16,
       "fp8": L.LR_DT_F16}     # "fp8" = W8A8: f16 activations in HBM, e4m3 GEMM operands (lr_model_desc.w8a8)
# split-operand mode: activations as hi + lo (include/llava_reward_hip.h `precise`); "f16x2f8" = residual pass of the big GEMMs in e4m3
_PRECISE = {"f16x2": 1, "bf16x2": 1, "f16x2f8": 2}


def rope_inv_freq(factors, head_dim: int, theta: float) -> torch.Tensor:
    """1 / (ext_factors * base ** (arange(0, dim, 2) / dim)) with the reference's exact torch ops
    (modeling_phi3_v.py:454-455), so the device table matches the oracle bit for bit."""
    ext = torch.tensor(list(factors), dtype=torch.float32)
    inv_shape = torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim
    return 1.0 / (ext * theta ** inv_shape)


def make_desc(cfg, max_batch: int, max_seq: int, max_crops: int, operand_dtype: str, max_patches: int = 0,
              mean_hidden_state: bool = False) -> L.ModelDesc:
    d = L.ModelDesc()
    d.struct_size = C.sizeof(L.ModelDesc)
    d.vocab_size, d.hidden, d.intermediate = cfg.vocab_size, cfg.hidden, cfg.intermediate
    d.layers, d.heads = cfg.layers, cfg.heads
    d.rms_eps = cfg.rms_eps
    half = cfg.head_dim // 2
    if half > L.LR_MAX_HALF_HEAD:
        raise ValueError("head_dim too large")
    if isinstance(cfg, QwenConfig):
        v = cfg.vision
        d.backbone = L.LR_BACKBONE_QWEN2_5_VL
        d.kv_heads, d.head_dim, d.image_token_id = cfg.kv_heads, cfg.head_dim, cfg.image_token_id
        d.orig_max_pos, d.rope_scaling = 1 << 30, 1.0        # default rope_type (Qwen2_5_VLRotaryEmbedding)
        inv = rope_inv_freq([1.0] * half, cfg.head_dim, cfg.rope_theta)
        for i in range(half):
            d.inv_freq_short[i] = d.inv_freq_long[i] = float(inv[i])
        d.vit_depth, d.vit_hidden, d.vit_heads, d.vit_intermediate = v.depth, v.hidden, v.heads, v.intermediate
        d.vit_patch, d.vit_temporal_patch, d.vit_merge, d.vit_window, d.vit_in_ch = v.patch, v.temporal_patch, v.merge, v.window, v.in_ch
        if len(v.fullatt) > L.LR_MAX_FULLATT:
            raise ValueError("too many fullatt_block_indexes")
        d.vit_n_fullatt = len(v.fullatt)
        for i, b in enumerate(v.fullatt):
            d.vit_fullatt[i] = b
        d.vit_rope_theta, d.vit_eps = v.rope_theta, v.eps
        for i in range(3):
            d.mrope_section[i] = cfg.mrope_section[i]
        from .synth import QWEN_CA_TOKEN_ID
        d.ca_token_id = QWEN_CA_TOKEN_ID
        d.max_patches = max_patches
    elif isinstance(cfg, LlavaConfig):
        d.backbone = L.LR_BACKBONE_LLAVA_NEXT
        d.kv_heads, d.head_dim, d.image_token_id = cfg.kv_heads, cfg.head_dim, cfg.image_token_id
        d.orig_max_pos, d.rope_scaling = 1 << 30, 1.0        # plain RoPE (modeling_mistral.py MistralRotaryEmbedding)
        inv = rope_inv_freq([1.0] * half, cfg.head_dim, cfg.rope_theta)
        for i in range(half):
            d.inv_freq_short[i] = d.inv_freq_long[i] = float(inv[i])
        if len(cfg.pinpoints) > L.LR_MAX_PINPOINTS:
            raise ValueError("too many image_grid_pinpoints")
        d.n_pinpoints = len(cfg.pinpoints)
        for i, (ph, pw) in enumerate(cfg.pinpoints):
            d.pinpoints[2 * i], d.pinpoints[2 * i + 1] = ph, pw
    else:
        d.backbone = L.LR_BACKBONE_PHI3V
        d.kv_heads, d.head_dim, d.image_token_id = cfg.heads, cfg.head_dim, -1
        d.orig_max_pos = cfg.orig_max_pos
        scale = cfg.max_pos / cfg.orig_max_pos
        d.rope_scaling = 1.0 if scale <= 1.0 else math.sqrt(1 + math.log(scale) / math.log(cfg.orig_max_pos))
        for dst, fac in ((d.inv_freq_short, cfg.short_factor), (d.inv_freq_long, cfg.long_factor)):
            inv = rope_inv_freq(fac, cfg.head_dim, cfg.rope_theta)
            for i in range(half):
                dst[i] = float(inv[i])
    if not isinstance(cfg, QwenConfig):
        c = cfg.clip
        d.clip_hidden, d.clip_heads, d.clip_mlp, d.clip_layers = c.hidden, c.heads, c.mlp, c.layers_used
        d.clip_image, d.clip_patch, d.clip_ln_eps = c.image, c.patch, c.ln_eps
    d.value_head_dim = cfg.value_head_dim
    d.add_cross_attention = int(bool(cfg.add_cross_attention))
    d.ca_eps = getattr(cfg, "ca_eps", 1e-5)
    d.max_batch, d.max_seq, d.max_crops = max_batch, max_seq, max_crops
    d.operand_dtype = _DT[operand_dtype]
    d.precise = _PRECISE.get(operand_dtype, 0)
    d.mean_hidden_state = 1 if mean_hidden_state else 0
    d.w8a8 = 1 if operand_dtype == "fp8" else 0
    d.lora_rank = int(getattr(cfg, "lora_rank", 0))
    # Phi3FlashAttention2's su-RoPE switch point (modeling_phi3_v.py:793-794): what the reference's --flash_attn scripts run
    d.rope_flash_convention = 1 if getattr(cfg, "rope_flash_convention", False) else 0
    if d.lora_rank and d.w8a8:
        raise ValueError("operand_dtype='fp8' (W8A8) runs merged weights only: load the adapter with merge=True")
    return d


class RewardEngine:
    """Owns one lr_handle (one GPU).  Not thread-safe; forward() enqueues on the current torch stream."""

    def __init__(self, cfg, device: int = 0, max_batch: int = 32, max_seq: int = 2816,
                 max_crops: int = 17, operand_dtype: str = "f16x2f8", max_patches: int = 0, mean_hidden_state: bool = False):
        if not torch.cuda.is_available():
            raise RuntimeError("RewardEngine needs a HIP device (torch.cuda.is_available() is False); "
                               "the scoring path has no CPU fallback")
        self.lib = L.load()
        self.cfg = cfg
        self.device = int(device)
        self.operand_dtype = operand_dtype
        self.max_batch, self.max_seq, self.max_crops = max_batch, max_seq, max_crops
        torch.cuda.set_device(self.device)
        torch.zeros(1, device=f"cuda:{self.device}")       # make sure torch owns a context on this device
        if isinstance(cfg, QwenConfig) and max_patches <= 0:
            max_patches = max_batch * 1280 * cfg.vision.merge_unit      # the reference's max_pixels = 1280 * 28^2 (utils/utils.py:36)
        self.max_patches = max_patches
        self._desc = make_desc(cfg, max_batch, max_seq, max_crops, operand_dtype, max_patches, mean_hidden_state)
        h = C.c_void_p()
        L.check(self.lib, self.lib.lr_create(C.byref(self._desc), self.device, C.byref(h)), None, "lr_create")
        self.h = h
        self.finalized = False
        self.gemm_tile = -1          # lr_set_gemm_tile: -1 = the heuristic

    # ------------------------------------------------------------------ weights
    def weight_names(self):
        n = self.lib.lr_num_weights(self.h)
        return [self.lib.lr_weight_name(self.h, i).decode() for i in range(n)]

    def upload(self, name: str, t) -> None:
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t))
        t = t.detach()
        if t.dtype == torch.float32:
            dt = L.LR_DT_F32
        elif t.dtype == torch.bfloat16:
            dt = L.LR_DT_BF16
        elif t.dtype == torch.float16:
            dt = L.LR_DT_F16
        else:
            t, dt = t.float(), L.LR_DT_F32
        t = t.contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        L.check(self.lib, self.lib.lr_upload_weight(self.h, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim(), dt,
                                                    1 if t.is_cuda else 0), self.h, f"lr_upload_weight({name})")

    def load_state_dict(self, sd: Dict[str, "torch.Tensor"], strict: bool = True) -> None:
        names = set(self.weight_names())
        for k, v in sd.items():
            if k in names:
                self.upload(k, v)
                names.discard(k)
        if strict and names:
            raise KeyError(f"missing weights: {sorted(names)[:8]}{' ...' if len(names) > 8 else ''}")

    def synth_weights(self, seed: int, fp32_valued: bool = False, profile: int = 0) -> None:
        """`profile`: synth.PROFILE_* flags (outlier-bearing / e4m3-valued weights), the same bits lr_synth_weights_ex takes."""
        L.check(self.lib, self.lib.lr_synth_weights_ex(self.h, C.c_uint64(seed), (1 if fp32_valued else 0) | int(profile)), self.h,
                "lr_synth_weights")

    def finalize(self) -> None:
        L.check(self.lib, self.lib.lr_finalize(self.h), self.h, "lr_finalize")
        self.finalized = True

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, pixel_values: torch.Tensor,
                image_sizes, training: bool = False, out: Optional[torch.Tensor] = None, no_final_norm: bool = False,
                keep_hidden_states: bool = False) -> torch.Tensor:
        """keep_hidden_states: run the last decoder layer for every token (LR_FWD_KEEP_HIDDEN_STATES) -- needed before
        last_hidden_state() / read_tap("x"); by default that layer computes only the row each reward is read from."""
        dev = torch.device("cuda", self.device)
        ids = input_ids.to(dev, torch.int64).contiguous()
        mask = attention_mask.to(dev, torch.int64).contiguous()
        if pixel_values.dtype == torch.bfloat16:
            pdt = L.LR_DT_BF16
        else:
            pixel_values, pdt = pixel_values.to(torch.float32), L.LR_DT_F32
        pix = pixel_values.to(dev).contiguous()
        sizes = torch.as_tensor(image_sizes).to("cpu", torch.int64).contiguous()
        B, S = ids.shape
        if pix.dim() != 5 or pix.shape[0] != B:
            raise ValueError("pixel_values must be [B, crops, 3, H, W]")
        if out is None:
            out = torch.empty(B, self.cfg.value_head_dim, device=dev, dtype=torch.float32)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.lr_forward(self.h, C.c_void_p(ids.data_ptr()), C.c_void_p(mask.data_ptr()), C.c_void_p(pix.data_ptr()),
                                 pdt, C.cast(sizes.data_ptr(), C.POINTER(C.c_int64)), B, S, pix.shape[1],
                                 (L.LR_FWD_TRAINING_LAST_TOKEN if training else 0) | (L.LR_FWD_NO_FINAL_NORM if no_final_norm else 0)
                                 | (L.LR_FWD_KEEP_HIDDEN_STATES if keep_hidden_states else 0),
                                 C.c_void_p(out.data_ptr()), C.c_void_p(stream))
        L.check(self.lib, rc, self.h, "lr_forward")
        # keep inputs alive until the stream has consumed them
        for t in (ids, mask, pix):
            t.record_stream(torch.cuda.current_stream(dev))
        return out

    def forward_qwen(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, pixel_values: torch.Tensor,
                     image_grid_thw, training: bool = False, out: Optional[torch.Tensor] = None,
                     keep_hidden_states: bool = False) -> torch.Tensor:
        """The qwen branch's inputs_batch (rw_model_general_preference.py:354-357): pixel_values [sum t*h*w, 1176]."""
        dev = torch.device("cuda", self.device)
        ids = input_ids.to(dev, torch.int64).contiguous()
        mask = attention_mask.to(dev, torch.int64).contiguous()
        if pixel_values.dtype == torch.bfloat16:
            pdt = L.LR_DT_BF16
        else:
            pixel_values, pdt = pixel_values.to(torch.float32), L.LR_DT_F32
        pix = pixel_values.to(dev).contiguous()
        grid = torch.as_tensor(image_grid_thw).to("cpu", torch.int64).contiguous()
        B, S = ids.shape
        if grid.dim() != 2 or grid.shape[1] != 3:
            raise ValueError("image_grid_thw must be [n_images, 3]")
        if pix.dim() != 2 or pix.shape[1] != self.cfg.vision.patch_dim or pix.shape[0] != int(grid.prod(dim=1).sum()):
            raise ValueError("pixel_values must be [sum(t*h*w), in_ch*temporal_patch*patch^2] matching image_grid_thw")
        if out is None:
            out = torch.empty(B, self.cfg.value_head_dim, device=dev, dtype=torch.float32)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.lr_forward_qwen(self.h, C.c_void_p(ids.data_ptr()), C.c_void_p(mask.data_ptr()), C.c_void_p(pix.data_ptr()),
                                      pdt, C.cast(grid.data_ptr(), C.POINTER(C.c_int64)), grid.shape[0], B, S,
                                      (L.LR_FWD_TRAINING_LAST_TOKEN if training else 0) | (L.LR_FWD_KEEP_HIDDEN_STATES if keep_hidden_states else 0),
                                      C.c_void_p(out.data_ptr()), C.c_void_p(stream))
        L.check(self.lib, rc, self.h, "lr_forward_qwen")
        for t in (ids, mask, pix):
            t.record_stream(torch.cuda.current_stream(dev))
        return out

    def last_hidden_state(self, B: int, S: int, no_final_norm: bool = False) -> torch.Tensor:
        """[B, S, hidden] fp32 on the device: final-norm output of every token of the last forward (lr_last_hidden_state)."""
        dev = torch.device("cuda", self.device)
        out = torch.empty(B, S, self.cfg.hidden, device=dev, dtype=torch.float32)
        stream = torch.cuda.current_stream(dev).cuda_stream
        L.check(self.lib, self.lib.lr_last_hidden_state(self.h, C.c_void_p(out.data_ptr()), out.numel(), 1 if no_final_norm else 0,
                                                        C.c_void_p(stream)), self.h, "lr_last_hidden_state")
        return out

    def vision_embeds(self, B: int) -> torch.Tensor:
        """[B, V_max, hidden] fp32 on the device: the zero-padded projected image tokens of the last forward (lr_vision_embeds) -- the
        last entry of the reference backbone's `hidden_states` (modeling_phi3_v.py:242-245, :1505)."""
        dev = torch.device("cuda", self.device)
        stream = torch.cuda.current_stream(dev).cuda_stream
        vmax = C.c_int(0)
        L.check(self.lib, self.lib.lr_vision_embeds(self.h, C.c_void_p(0), 0, C.byref(vmax), C.c_void_p(stream)), self.h, "lr_vision_embeds")
        out = torch.empty(B, vmax.value, self.cfg.hidden, device=dev, dtype=torch.float32)
        L.check(self.lib, self.lib.lr_vision_embeds(self.h, C.c_void_p(out.data_ptr()), out.numel(), C.byref(vmax), C.c_void_p(stream)),
                self.h, "lr_vision_embeds")
        return out

    def read_tap(self, name: str, numel: int) -> np.ndarray:
        buf = np.empty(numel, dtype=np.float32)
        n = C.c_size_t(0)
        L.check(self.lib, self.lib.lr_read_tap(self.h, name.encode(), buf.ctypes.data_as(C.c_void_p), numel, C.byref(n)),
                self.h, "lr_read_tap")
        return buf[: n.value]

    def set_layer_limits(self, n_clip: int = -1, n_layers: int = -1) -> None:
        self.lib.lr_set_layer_limits(self.h, n_clip, n_layers)

    def set_precision_map(self, clip_form: int = -1, decoder_mid_form: int = -1, decoder_first: int = 0, decoder_last: int = 0) -> None:
        """Operand form per stage (lr_set_precision_map): -1 the descriptor's, 0 single pass, 1 split, 2 split + e4m3 residual passes."""
        L.check(self.lib, self.lib.lr_set_precision_map(self.h, clip_form, decoder_mid_form, decoder_first, decoder_last), self.h,
                "lr_set_precision_map")

    def set_precision_sites(self, qkv: int = -1, attention: int = -1, o_proj: int = -1, gate_up: int = -1, down: int = -1) -> None:
        """Operand form per site of the decoder layers the precision map covers (lr_set_precision_sites): -1 the layer's, 1 strict, 2 default."""
        L.check(self.lib, self.lib.lr_set_precision_sites(self.h, qkv, attention, o_proj, gate_up, down), self.h, "lr_set_precision_sites")

    def set_attention_lazy_threshold(self, default_stages: float = 8.0, strict_stages: float = 0.0) -> None:
        """Lazy softmax reference maximum of the attention launches (lr_set_attention_lazy_threshold; log2 units, 0 = exact): in stages
        that run the default operand form / in stages that run the strict form."""
        L.check(self.lib, self.lib.lr_set_attention_lazy_threshold(self.h, float(default_stages), float(strict_stages)), self.h,
                "lr_set_attention_lazy_threshold")

    def weights_epoch(self) -> int:
        """Counts the calls that changed this handle's weights (lr_weights_epoch): what was derived from them is stale once it moves."""
        return int(self.lib.lr_weights_epoch(self.h))

    def set_gemm_tile(self, tile: int) -> None:
        L.check(self.lib, self.lib.lr_set_gemm_tile(self.h, tile), self.h, "lr_set_gemm_tile")
        self.gemm_tile = int(tile)

    def workspace_bytes(self) -> int:
        return int(self.lib.lr_workspace_bytes(self.h))

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.lr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
