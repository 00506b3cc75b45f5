"""Drop-in counterparts of the reference's eval/reward_adaptor_loader.py callables:
load_reward_adaptor (:24-156), inference_process_phi3v (:158-173), preference_compute (:174-181).
Same names, argument order, `args` mutation and return conventions."""
from __future__ import annotations

import os

import numpy as np
import torch
import yaml

from . import checkpoint as ckpt
from .model import RewardModel
from .synth import llava_weight_specs, qwen_weight_specs, weight_specs


class UnknownModelType(UnboundLocalError, ValueError):
    """The reference falls off its if/elif chain and dies with UnboundLocalError (:156)."""


def _attach_adapter(args, cfg, weights, spec_fn, canon=None):
    """<pm_path>/lora -> `weights` (checkpoint.attach_lora).  The decoder adapter stays un-merged, as the reference runs it
    (:44-45); `args.merge_lora = True` (debug switch) or the W8A8 operand mode merges everything on the host instead."""
    lora, scale, rank = ckpt.read_lora(os.path.join(args.pm_path, "lora"))
    merge = bool(getattr(args, "merge_lora", False)) or getattr(args, "operand_dtype", "f16x2f8") == "fp8"
    if not merge:
        cfg.lora_rank = rank
    clip = getattr(cfg, "clip", None)
    stats = ckpt.attach_lora(weights, lora, scale, rank, [n for n, *_ in spec_fn(cfg)], canon=canon, merge=merge,
                             clip_layers_used=clip.layers_used if clip is not None else 1 << 30)
    args.lora_modules = stats             # {"unmerged", "merged", "skipped"}: how the adapter's modules were applied
    return stats


def _form_args(args):
    """Knobs the reference does not have: args.operand_form = "<name>" pins the operand form of the default parity
    mode across deployments (no probe); args.check_inputs = "deferred" drops the per-forward host check of device-resident inputs."""
    rd = getattr(args, "reward_dtype", None)
    if isinstance(rd, str):
        rd = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": None, "float32": None, "fp16": torch.float16}[rd]
    return dict(operand_form=getattr(args, "operand_form", None), check_inputs=getattr(args, "check_inputs", "eager"),
                parity_budget=getattr(args, "parity_budget", 1.5e-4), reward_dtype=rd,
                vision_layer_id=getattr(args, "vision_layer_id", -1))          # rw_model:296 (only -1, its default, is served: RewardModel refuses others)


def load_reward_adaptor(args, model_type, reward_config_path, load_tokenizer=False):
    with open(reward_config_path) as f:
        reward_cfg = yaml.safe_load(f)
    args.is_general_preference = reward_cfg["is_general_preference"]
    args.add_cross_attention = reward_cfg["add_cross_attention"]
    args.value_head_dim = reward_cfg["value_head_dim"]
    args.general_preference_tau = reward_cfg["general_preference_tau"]
    if model_type == "phi3v":
        if not os.path.isdir(args.pretrain):
            raise FileNotFoundError(f"args.pretrain={args.pretrain!r} must be a local checkpoint directory "
                                    "(config.json + *.safetensors); hub download is not available offline")
        cfg = ckpt.config_from_hf(args.pretrain, reward_cfg)
        # su-RoPE switch convention: the eval loader of the reference never reads args.flash_attn (:24-60) -- the attention class, hence
        # the convention, comes from the checkpoint's config.json alone, which config_from_hf has already read (_attn_implementation)
        names = [n for n, *_ in weight_specs(cfg)]
        head_names = {n for n in names if n.split(".")[0] in ("value_head", "W_q", "W_k", "W_v", "ca_layernorm")}
        weights = ckpt.read_base_weights(args.pretrain, [n for n in names if n not in head_names])
        _attach_adapter(args, cfg, weights, weight_specs)
        heads = ckpt.read_heads(args.pm_path, cfg, getattr(args, "ft_projector", False))
        weights.update(heads)
        model = RewardModel(cfg, weights=weights,
                            max_batch=getattr(args, "max_batch", 32), max_seq=getattr(args, "max_seq", 2816),
                            max_crops=getattr(args, "max_crops", 17),
                            operand_dtype=getattr(args, "operand_dtype", "f16x2f8"), calibrate=getattr(args, "calibrate", True), **_form_args(args))
        model.model_type = "phi3v"
        if load_tokenizer:
            from transformers import AutoProcessor
            processor = AutoProcessor.from_pretrained(args.pretrain, cache_dir=getattr(args, "cache_dir", None),
                                                      padding_side="left", trust_remote_code=True, num_crops=16,
                                                      model_max_length=131072)     # utils/utils.py:19-32
            tokenizer = processor.tokenizer
            tokenizer.padding_side = "left"
            if tokenizer.pad_token is None:
                tokenizer.pad_token = tokenizer.eos_token
                tokenizer.pad_token_id = tokenizer.eos_token_id
            tokenizer.truncation_side = "right"
    elif model_type == "llava":                      # reward_adaptor_loader.py:110-148
        if not os.path.isdir(args.pretrain):
            raise FileNotFoundError(f"args.pretrain={args.pretrain!r} must be a local checkpoint directory "
                                    "(config.json + *.safetensors); hub download is not available offline")
        cfg = ckpt.llava_config_from_hf(args.pretrain, reward_cfg)
        names = [n for n, *_ in llava_weight_specs(cfg)]
        weights = ckpt.read_base_weights(args.pretrain, [n for n in names if n != "value_head.weight"], canon=ckpt.canon_llava_key)
        _attach_adapter(args, cfg, weights, llava_weight_specs, canon=ckpt.canon_llava_key)
        weights.update(ckpt.read_heads(args.pm_path, cfg, getattr(args, "ft_projector", False)))
        model = RewardModel(cfg, weights=weights, max_batch=getattr(args, "max_batch", 32),
                            max_seq=getattr(args, "max_seq", 4096), max_crops=getattr(args, "max_crops", 5),
                            operand_dtype=getattr(args, "operand_dtype", "f16x2f8"), calibrate=getattr(args, "calibrate", True), **_form_args(args))
        if load_tokenizer:
            from transformers import LlavaNextProcessor
            processor = LlavaNextProcessor.from_pretrained(args.pretrain, cache_dir=getattr(args, "cache_dir", None))   # utils/utils.py:46-55
            tokenizer = processor.tokenizer
            tokenizer.padding_side = "left"
            if tokenizer.pad_token is None:
                tokenizer.pad_token = tokenizer.eos_token
                tokenizer.pad_token_id = tokenizer.eos_token_id
            tokenizer.truncation_side = "right"
    elif model_type == "qwen":                       # reward_adaptor_loader.py:64-109
        if not os.path.isdir(args.pretrain):
            raise FileNotFoundError(f"args.pretrain={args.pretrain!r} must be a local checkpoint directory "
                                    "(config.json + *.safetensors); hub download is not available offline")
        cfg = ckpt.qwen_config_from_hf(args.pretrain, reward_cfg)
        names = [n for n, *_ in qwen_weight_specs(cfg)]
        head_names = {n for n in names if n.split(".")[0] in ("value_head", "W_q", "W_k", "W_v", "ca_layernorm")}
        weights = ckpt.read_base_weights(args.pretrain, [n for n in names if n not in head_names], canon=ckpt.canon_qwen_key)
        _attach_adapter(args, cfg, weights, qwen_weight_specs, canon=ckpt.canon_qwen_key)
        weights.update(ckpt.read_heads(args.pm_path, cfg, getattr(args, "ft_projector", False)))
        model = RewardModel(cfg, weights=weights, max_batch=getattr(args, "max_batch", 32),
                            max_seq=getattr(args, "max_seq", 2048), max_patches=getattr(args, "max_patches", 0),
                            operand_dtype=getattr(args, "operand_dtype", "f16x2f8"), calibrate=getattr(args, "calibrate", True), **_form_args(args))
        if load_tokenizer:
            from transformers import AutoProcessor
            processor = AutoProcessor.from_pretrained(args.pretrain, cache_dir=getattr(args, "cache_dir", None),
                                                      min_pixels=256 * 28 * 28, max_pixels=1280 * 28 * 28)    # utils/utils.py:34-44
            tokenizer = processor.tokenizer
            tokenizer.padding_side = "left"
            if tokenizer.pad_token is None:
                tokenizer.pad_token = tokenizer.eos_token
                tokenizer.pad_token_id = tokenizer.eos_token_id
            tokenizer.truncation_side = "right"
    else:
        raise UnknownModelType(f"local variable 'model' referenced before assignment (model_type={model_type!r})")
    if load_tokenizer:
        return args, model, processor, tokenizer
    return args, model


def inference_process_phi3v(args, processor, tokenizer, img_dir_list, caption, device="cuda"):
    from PIL import Image
    img_list = [Image.open(d).convert("RGB") for d in img_dir_list]
    prompt_messages = {"role": "user", "content": f"<|image_1|>\n{caption}"}
    prompt = tokenizer.apply_chat_template([prompt_messages], tokenize=False, add_generation_prompt=True)[:-22] + tokenizer.eos_token
    img_inputs = []
    for img in img_list:
        img_input = processor(text=prompt, images=[img], return_tensors="pt", padding=True, truncation=True)
        for k in img_input:
            img_input[k] = img_input[k].to(device)
        img_inputs.append(img_input)
    return img_inputs


def preference_compute(args, chosen_rewards, reject_rewards):
    if args.is_general_preference and args.value_head_dim == 2:
        gpm_product = chosen_rewards[:, 0] * reject_rewards[:, 1] - chosen_rewards[:, 1] * reject_rewards[:, 0]
        prob = torch.sigmoid(gpm_product / args.general_preference_tau)
    else:
        prob = torch.sigmoid((chosen_rewards - reject_rewards) / args.general_preference_tau).squeeze(-1)
    return prob.float().cpu().numpy()
