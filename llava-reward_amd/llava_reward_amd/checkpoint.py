"""Readers for the on-disk layout `load_reward_adaptor` consumes:

    <pretrain>/config.json + *.safetensors         HF base checkpoint (bf16)
    <pm_path>/reward_config.yaml                   4 keys (utils/deepspeed.py:402-404)
    <pm_path>/pytorch_model.bin                    value_head / W_q / W_k / W_v / ca_layernorm / img_projection
    <pm_path>/lora/adapter_config.json + adapter_model.{bin,safetensors}   PEFT LoRA adapter

The LoRA adapter of the decoder linears stays UN-MERGED, as the reference runs it (eval/reward_adaptor_loader.py:44-45):
A and (alpha/r) * B go to the engine as tensors of their own (lr_model_desc.lora_rank); adapters on other modules, and
everything with merge=True (debug switch; the W8A8 mode), are merged on the host in fp32, W' = W + (alpha/r) * B @ A.
"""
from __future__ import annotations

import glob
import json
import os
import re
from typing import Dict, Tuple

import torch

from .synth import ClipConfig, LlavaConfig, QwenConfig, QwenVisionConfig, RewardConfig


def config_from_hf(path: str, reward_cfg: dict) -> RewardConfig:
    """Phi3VConfig fields (configuration_phi3_v.py:119-145) + reward_config.yaml -> RewardConfig."""
    with open(os.path.join(path, "config.json")) as f:
        c = json.load(f)
    ip = c.get("img_processor") or {}
    if ip.get("name", "clip_vision_model") != "clip_vision_model":
        raise NotImplementedError(f"img_processor = {ip}, not implemented")       # modeling_phi3_v.py:151
    el = c.get("embd_layer") or {}
    if isinstance(el, dict) and el:
        if el.get("hd_transform_order", "sub_glb") != "sub_glb":
            raise AssertionError(f"hd_transform_order `{el.get('hd_transform_order')}` not implemented")   # :259
        if el.get("projection_cls", "mlp") != "mlp" or not el.get("use_hd_transform", True):
            raise NotImplementedError("only the HD-transform MLP projector is implemented")
    rs = c.get("rope_scaling")
    if not rs or rs.get("type", rs.get("rope_type")) not in ("su", "longrope"):
        raise ValueError("rope_scaling of type 'su' with short_factor/long_factor is required")
    clip = ClipConfig(hidden=ip.get("clip_hidden", 1024), heads=ip.get("clip_heads", 16), mlp=ip.get("clip_mlp", 4096),
                      layers_used=ip.get("clip_layers_used", 23))
    return RewardConfig(
        vocab_size=c["vocab_size"], hidden=c["hidden_size"], intermediate=c["intermediate_size"],
        layers=c["num_hidden_layers"], heads=c["num_attention_heads"], rms_eps=c.get("rms_norm_eps", 1e-5),
        rope_theta=c.get("rope_theta", 10000.0), max_pos=c.get("max_position_embeddings", 4096),
        orig_max_pos=c.get("original_max_position_embeddings", 4096),
        short_factor=tuple(rs["short_factor"]), long_factor=tuple(rs["long_factor"]), clip=clip,
        is_general_preference=bool(reward_cfg["is_general_preference"]),
        add_cross_attention=bool(reward_cfg["add_cross_attention"]),
        value_head_dim=int(reward_cfg["value_head_dim"]),
        general_preference_tau=float(reward_cfg["general_preference_tau"]),
        # the checkpoint's config.json names the attention class the reference instantiates (modeling_phi3_v.py:1123-1135); the
        # flash class switches the su-RoPE tables one token earlier (:793-794)
        rope_flash_convention=c.get("_attn_implementation") == "flash_attention_2")


def llava_config_from_hf(path: str, reward_cfg: dict) -> LlavaConfig:
    """llava-hf/llava-v1.6-*-hf config.json (LlavaNextConfig: text_config / vision_config) + reward_config.yaml."""
    with open(os.path.join(path, "config.json")) as f:
        c = json.load(f)
    t, v = c.get("text_config") or {}, c.get("vision_config") or {}
    if c.get("vision_feature_layer", -2) != -2 or c.get("vision_feature_select_strategy", "default") != "default":
        raise NotImplementedError("only vision_feature_layer=-2 with the 'default' (CLS dropped) strategy is implemented")
    if t.get("sliding_window") not in (None, 0):
        raise NotImplementedError("sliding-window attention is not implemented (Mistral v0.2 text towers have none)")
    if reward_cfg["add_cross_attention"]:
        raise AttributeError("'LlavaNextConfig' object has no attribute 'hidden_size'")      # rw_model:315 with llava
    heads = t.get("num_attention_heads", 32)
    hidden = t.get("hidden_size", 4096)
    layers_total = v.get("num_hidden_layers", 24)
    clip = ClipConfig(hidden=v.get("hidden_size", 1024), heads=v.get("num_attention_heads", 16),
                      mlp=v.get("intermediate_size", 4096), layers_used=layers_total - 1,
                      image=v.get("image_size", 336), patch=v.get("patch_size", 14), ln_eps=v.get("layer_norm_eps", 1e-5))
    return LlavaConfig(
        vocab_size=t.get("vocab_size", 32064), hidden=hidden, intermediate=t.get("intermediate_size", 14336),
        layers=t.get("num_hidden_layers", 32), heads=heads, kv_heads=t.get("num_key_value_heads", 8),
        head_dim=t.get("head_dim") or hidden // heads, rms_eps=t.get("rms_norm_eps", 1e-5),
        rope_theta=t.get("rope_theta", 1000000.0), clip=clip,
        image_token_id=c.get("image_token_index", c.get("image_token_id", 32000)),
        pad_token_id=t.get("pad_token_id") or c.get("pad_token_id") or 32001,
        pinpoints=tuple(tuple(p) for p in c.get("image_grid_pinpoints", [[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]])),
        is_general_preference=bool(reward_cfg["is_general_preference"]), add_cross_attention=False,
        value_head_dim=int(reward_cfg["value_head_dim"]), general_preference_tau=float(reward_cfg["general_preference_tau"]))


def qwen_config_from_hf(path: str, reward_cfg: dict) -> QwenConfig:
    """Qwen2.5-VL-*-Instruct config.json (flat text fields in the 4.50-era layout, `text_config` in the 5.x one;
    `vision_config`; rope_scaling.mrope_section) + reward_config.yaml -> QwenConfig."""
    with open(os.path.join(path, "config.json")) as f:
        c = json.load(f)
    t = dict(c)
    t.update(c.get("text_config") or {})
    v = c.get("vision_config") or {}
    if t.get("use_sliding_window", False):
        raise NotImplementedError("sliding-window attention is not implemented (Qwen2.5-VL checkpoints ship use_sliding_window=false)")
    rs = t.get("rope_scaling") or t.get("rope_parameters") or {}
    if "mrope_section" not in rs:
        raise ValueError("rope_scaling.mrope_section is required (Qwen2.5-VL multimodal RoPE)")
    if rs.get("rope_type", rs.get("type", "default")) not in ("default", "mrope"):
        raise NotImplementedError(f"rope type {rs.get('rope_type', rs.get('type'))!r} is not implemented")
    if v.get("hidden_act", "silu") != "silu" or t.get("hidden_act", "silu") != "silu":
        raise NotImplementedError("only SiLU MLPs are implemented")
    heads = t["num_attention_heads"]
    hidden = t["hidden_size"]
    if v.get("out_hidden_size", hidden) != hidden:
        raise ValueError("vision_config.out_hidden_size must equal the decoder hidden size")
    vis = QwenVisionConfig(depth=v.get("depth", 32), hidden=v.get("hidden_size", 1280), heads=v.get("num_heads", 16),
                           intermediate=v.get("intermediate_size", 3420), patch=v.get("patch_size", 14),
                           temporal_patch=v.get("temporal_patch_size", 2), merge=v.get("spatial_merge_size", 2),
                           window=v.get("window_size", 112), fullatt=tuple(v.get("fullatt_block_indexes", [7, 15, 23, 31])),
                           in_ch=v.get("in_channels", v.get("in_chans", 3)))
    return QwenConfig(
        vocab_size=t["vocab_size"], hidden=hidden, intermediate=t["intermediate_size"], layers=t["num_hidden_layers"],
        heads=heads, kv_heads=t.get("num_key_value_heads", heads), head_dim=hidden // heads,
        rms_eps=t.get("rms_norm_eps", 1e-6), rope_theta=rs.get("rope_theta", t.get("rope_theta", 1000000.0)),
        mrope_section=tuple(rs["mrope_section"]), vision=vis, image_token_id=c.get("image_token_id", 151655),
        pad_token_id=151643,
        is_general_preference=bool(reward_cfg["is_general_preference"]),
        add_cross_attention=bool(reward_cfg["add_cross_attention"]),
        value_head_dim=int(reward_cfg["value_head_dim"]), general_preference_tau=float(reward_cfg["general_preference_tau"]))


def canon_qwen_key(k: str) -> str:
    """Accept both checkpoint layouts: 4.50-era (visual.*, model.layers...) and the 5.x module tree
    (model.visual.*, model.language_model.layers...)."""
    if k.startswith("model.visual."):
        return k[len("model."):]
    if k.startswith("model.language_model."):
        return "model." + k[len("model.language_model."):]
    return k


def canon_llava_key(k: str) -> str:
    """Accept both checkpoint layouts: 4.50-era (language_model.model.layers..., vision_tower.vision_model...) and the
    5.x module tree (model.language_model.layers..., model.vision_tower...)."""
    n = k[len("model."):] if k.startswith("model.") else k
    n = n.replace("language_model.layers.", "language_model.model.layers.")
    n = n.replace("language_model.embed_tokens.", "language_model.model.embed_tokens.")
    n = n.replace("language_model.norm.", "language_model.model.norm.")
    if n.startswith("vision_tower.") and not n.startswith("vision_tower.vision_model."):
        n = n.replace("vision_tower.", "vision_tower.vision_model.", 1)
    return n


def read_base_weights(path: str, wanted, canon=None) -> Dict[str, torch.Tensor]:
    """Read the tensors named in `wanted` from *.safetensors (or pytorch_model*.bin) under `path`."""
    wanted = set(wanted)
    out: Dict[str, torch.Tensor] = {}
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if files:
        from safetensors import safe_open
        for fn in files:
            with safe_open(fn, framework="pt", device="cpu") as f:
                for k in f.keys():
                    ck = canon(k) if canon else k
                    if ck in wanted:
                        out[ck] = f.get_tensor(k)
    else:
        for fn in sorted(glob.glob(os.path.join(path, "pytorch_model*.bin"))):
            sd = torch.load(fn, map_location="cpu")
            out.update({(canon(k) if canon else k): v for k, v in sd.items() if (canon(k) if canon else k) in wanted})
    missing = wanted - set(out)
    if missing:
        raise FileNotFoundError(f"{path}: base checkpoint lacks {sorted(missing)[:4]} (+{max(0, len(missing) - 4)} more)")
    return out


_LORA_RE = re.compile(r"^(?:base_model\.model\.)?(.+)\.lora_([AB])(?:\.[^.]+)?\.weight$")
# adapter_config.json switches that change the arithmetic of peft's Linear.forward and are not implemented here
_LORA_UNSUPPORTED = ("use_rslora", "use_dora", "fan_in_fan_out", "rank_pattern", "alpha_pattern", "layer_replication", "megatron_config")


def read_lora(lora_dir: str) -> Tuple[Dict[str, Tuple[torch.Tensor, torch.Tensor]], float, int]:
    """-> ({module name: (A [r,in], B [out,r])}, lora_alpha / r, r).  Layout written by DeepspeedStrategy.save_model_lora
    (utils/deepspeed.py:386-398: LoraConfig.save_pretrained + get_peft_model_state_dict -> keys `<module>.lora_A.weight`, with or
    without the `base_model.model.` prefix and the adapter name)."""
    with open(os.path.join(lora_dir, "adapter_config.json")) as f:
        ac = json.load(f)
    for k in _LORA_UNSUPPORTED:
        if ac.get(k):
            raise NotImplementedError(f"adapter_config.json: {k}={ac[k]!r} is not implemented (plain LoRA, scaling = lora_alpha / r, only)")
    if ac.get("bias", "none") != "none" or ac.get("modules_to_save"):
        raise NotImplementedError("adapter_config.json: LoRA bias terms / modules_to_save are not implemented")
    r = int(ac["r"])
    scale = float(ac["lora_alpha"]) / float(r)
    st = os.path.join(lora_dir, "adapter_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(lora_dir, "adapter_model.bin"), map_location="cpu")
    pairs: Dict[str, dict] = {}
    stray = []
    for k, v in sd.items():
        m = _LORA_RE.match(k)
        if m:
            pairs.setdefault(m.group(1), {})[m.group(2)] = v
        else:
            stray.append(k)
    if stray:
        raise KeyError(f"LoRA adapter: tensors that are not lora_A / lora_B weights: {stray[:4]}")
    out = {}
    for mod, ab in pairs.items():
        if "A" not in ab or "B" not in ab:
            raise KeyError(f"LoRA adapter: incomplete pair for {mod}")
        if ab["A"].shape[0] != r or ab["B"].shape[1] != r:
            raise ValueError(f"LoRA adapter: {mod} has rank {ab['A'].shape[0]}, adapter_config.json says {r}")
        out[mod] = (ab["A"], ab["B"])
    if not out:
        raise KeyError(f"{lora_dir}: the adapter holds no lora_A / lora_B tensors")
    return out, scale, r


# Adapter modules that exist in a checkpoint but not on the scoring path: the CLIP layer patch_clip_for_lora deletes
# (utils/utils.py:277-281 keeps layers [0, layer_idx]) and heads the reward model never evaluates.
_OFF_PATH = re.compile(r"(?:^|\.)(?:lm_head|post_layernorm|encoder\.layers\.(\d+)\..*)$")


def _off_path(mod: str, clip_layers_used: int) -> bool:
    m = _OFF_PATH.search(mod)
    if not m:
        return False
    return m.group(1) is None or int(m.group(1)) >= clip_layers_used


def attach_lora(weights: Dict[str, torch.Tensor], lora: Dict[str, Tuple[torch.Tensor, torch.Tensor]], scale: float, rank: int,
                engine_names, canon=None, merge: bool = False, clip_layers_used: int = 1 << 30) -> Dict[str, int]:
    """Put a PEFT adapter into `weights` (in place).  Decoder linears the engine runs un-merged (names `<module>.lora_A.weight` /
    `.lora_B.weight` in `engine_names`, i.e. weight_specs of a config with lora_rank = rank) get A and the PRE-SCALED scale * B as
    tensors of their own, which is how the reference runs them (eval/reward_adaptor_loader.py:44-45) and keeps the bf16 base
    weights exact; un-targeted engine slots are zero-filled.  Other adapted modules on the path (vision tower, projector: only
    when the run did not freeze the vision model, utils/utils.py:203-214) and everything when merge=True are merged in fp32,
    W + scale * B @ A.  A module that resolves to nothing raises unless it is known to be off the scoring path.
    Returns {"unmerged": n, "merged": n, "skipped": n}."""
    engine_names = set(engine_names)
    n = {"unmerged": 0, "merged": 0, "skipped": 0}
    for mod, (A, B) in lora.items():
        cm = canon(mod + ".weight")[: -len(".weight")] if canon else mod
        if not merge and cm + ".lora_A.weight" in engine_names:
            weights[cm + ".lora_A.weight"] = A.float()
            weights[cm + ".lora_B.weight"] = scale * B.float()
            n["unmerged"] += 1
        elif cm + ".weight" in weights:
            weights[cm + ".weight"] = weights[cm + ".weight"].float() + scale * (B.float() @ A.float())
            n["merged"] += 1
        elif _off_path(cm, clip_layers_used):
            n["skipped"] += 1
        else:
            raise KeyError(f"LoRA adapter module {mod!r} matches no weight of the scoring path (canonical name {cm!r}); "
                           "refusing to load a model with part of its adapter silently dropped")
    if n["unmerged"] + n["merged"] == 0:
        raise KeyError("LoRA adapter: no module of the adapter applies to the scoring path")
    if not merge:
        for name in engine_names:                      # engine slots of linears this adapter does not target
            if name.endswith(".lora_A.weight") and name not in weights:
                base = weights[name[: -len(".lora_A.weight")] + ".weight"]
                weights[name] = torch.zeros(rank, base.shape[1])
                weights[name[: -len("A.weight")] + "B.weight"] = torch.zeros(base.shape[0], rank)
    return n


def read_heads(pm_path: str, cfg: RewardConfig, ft_projector: bool) -> Dict[str, torch.Tensor]:
    """pytorch_model.bin keys filtered by substring exactly as eval/reward_adaptor_loader.py:46-60."""
    sd = torch.load(os.path.join(pm_path, "pytorch_model.bin"), map_location="cpu")
    out = {}

    def pick(sub, dst_prefix, nparts=1):
        found = {".".join(k.split(".")[-nparts:]): v for k, v in sd.items() if sub in k}
        for tail, v in found.items():
            out[dst_prefix + tail] = v
        return found

    if not pick("value_head", "value_head."):
        raise KeyError("pytorch_model.bin has no value_head")
    if cfg.add_cross_attention:
        for nm in ("W_q", "W_k", "W_v", "ca_layernorm"):
            if not pick(nm, nm + "."):
                raise KeyError(f"pytorch_model.bin has no {nm}")
    if ft_projector and isinstance(cfg, QwenConfig):      # reward_adaptor_loader.py:92-104: keys containing 'merger'
        found = {".".join(k.split(".")[-2:]): v for k, v in sd.items() if "merger" in k}
        for src, dst in (("ln_q.weight", "ln_q.weight"), ("0.weight", "mlp.0.weight"), ("0.bias", "mlp.0.bias"),
                         ("2.weight", "mlp.2.weight"), ("2.bias", "mlp.2.bias")):
            out["visual.merger." + dst] = found[src]       # KeyError if the head file lacks it, as in the reference
    elif ft_projector:
        if isinstance(cfg, LlavaConfig):          # reward_adaptor_loader.py:137-145
            pick("multi_modal_projector", "multi_modal_projector.", nparts=2)
        else:
            pick("img_projection", "model.vision_embed_tokens.img_projection.", nparts=2)
    return out
