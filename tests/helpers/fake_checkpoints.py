"""Fabricated HF checkpoint directories for the 'llava' and 'qwen' branches of load_reward_adaptor
(/root/reference eval/reward_adaptor_loader.py:64-148), in both tensor-name layouts the loader accepts: "4.50" (the pinned transformers:
language_model.model.layers.* / visual.*) and "5.x" (the module tree of current transformers: model.language_model.layers.* /
model.visual.* / model.vision_tower.*).  The PEFT adapter carries SEPARATE q / k / v / o / gate / up / down modules
(llava_reward/utils/utils.py:223-262), `base_model.model.` prefix and the `.default.` adapter-name infix."""
import json
import os

import torch
import yaml

from llava_reward_amd import synth


def _adapter(names_shapes, r, seed, layers_without=()):
    g = torch.Generator().manual_seed(seed)
    sd, eff = {}, {}
    for name, (n_out, n_in) in names_shapes:
        if name in layers_without:
            continue
        A = (torch.randn(r, n_in, generator=g) * 0.05).to(torch.bfloat16)
        B = (torch.randn(n_out, r, generator=g) * 0.05).to(torch.bfloat16)
        sd[f"base_model.model.{name}.lora_A.default.weight"] = A
        sd[f"base_model.model.{name}.lora_B.default.weight"] = B
        eff[name + ".lora_A.weight"], eff[name + ".lora_B.weight"] = A.float(), B.float()
    return sd, eff


def write_llava(tmp, cfg, seed, layout, r=8, alpha=12.0):
    """-> (pretrain dir, pm dir, oracle weights with the adapter UN-merged + 'lora_scaling', target modules count)."""
    from safetensors.torch import save_file
    pre, pm = os.path.join(tmp, f"pre_{layout}"), os.path.join(tmp, f"pm_{layout}")
    os.makedirs(pre); os.makedirs(os.path.join(pm, "lora"))
    c = cfg.clip
    json.dump({"image_token_index": cfg.image_token_id, "image_grid_pinpoints": [list(p) for p in cfg.pinpoints],
               "vision_feature_layer": -2, "vision_feature_select_strategy": "default",
               "text_config": {"vocab_size": cfg.vocab_size, "hidden_size": cfg.hidden, "intermediate_size": cfg.intermediate,
                               "num_hidden_layers": cfg.layers, "num_attention_heads": cfg.heads, "num_key_value_heads": cfg.kv_heads,
                               "head_dim": cfg.head_dim, "rms_norm_eps": cfg.rms_eps, "rope_theta": cfg.rope_theta, "sliding_window": None},
               "vision_config": {"hidden_size": c.hidden, "num_attention_heads": c.heads, "intermediate_size": c.mlp,
                                 "num_hidden_layers": c.layers_used + 1, "image_size": 336, "patch_size": 14}},
              open(os.path.join(pre, "config.json"), "w"))
    W = {k: torch.from_numpy(v) for k, v in synth.llava_make_weights(cfg, seed).items() if ".lora_" not in k}

    def name5(k):
        k = k.replace("language_model.model.", "language_model.").replace("vision_tower.vision_model.", "vision_tower.")
        return "model." + k
    base = {(name5(k) if layout == "5.x" else k): v.to(torch.bfloat16) for k, v in W.items() if k != "value_head.weight"}
    save_file(base, os.path.join(pre, "model.safetensors"))
    torch.save({"base_model.model.value_head.weight": W["value_head.weight"]}, os.path.join(pm, "pytorch_model.bin"))
    yaml.safe_dump({"is_general_preference": bool(cfg.is_general_preference), "add_cross_attention": False, "value_head_dim": int(cfg.value_head_dim),
                    "general_preference_tau": 0.1}, open(os.path.join(pm, "reward_config.yaml"), "w"))
    Hq, Hkv, D, I = cfg.heads * cfg.head_dim, cfg.kv_heads * cfg.head_dim, cfg.hidden, cfg.intermediate
    mods = []
    for l in range(cfg.layers):
        p = f"language_model.model.layers.{l}."
        mods += [(p + "self_attn.q_proj", (Hq, D)), (p + "self_attn.k_proj", (Hkv, D)), (p + "self_attn.v_proj", (Hkv, D)),
                 (p + "self_attn.o_proj", (D, Hq)), (p + "mlp.gate_proj", (I, D)), (p + "mlp.up_proj", (I, D)), (p + "mlp.down_proj", (D, I))]
    skip = (f"language_model.model.layers.{cfg.layers - 1}.mlp.up_proj",)          # one module without an adapter: zero-filled slot
    sd, eff = _adapter(mods, r, seed + 1, skip)
    if layout == "5.x":          # peft names follow the module tree of the transformers that trained the adapter
        sd = {k.replace("base_model.model.language_model.model.", "base_model.model.model.language_model."): v for k, v in sd.items()}
    save_file(sd, os.path.join(pm, "lora", "adapter_model.safetensors"))
    json.dump({"r": r, "lora_alpha": alpha, "peft_type": "LORA",
               "target_modules": ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]},
              open(os.path.join(pm, "lora", "adapter_config.json"), "w"))
    Wo = {k: (v.to(torch.bfloat16).float() if k != "value_head.weight" else v.float()) for k, v in W.items()}
    Wo.update(eff)
    Wo["lora_scaling"] = alpha / r
    return pre, pm, Wo, len(mods) - len(skip)


def write_qwen(tmp, cfg, seed, layout, r=8, alpha=12.0):
    from safetensors.torch import save_file
    v = cfg.vision
    pre, pm = os.path.join(tmp, f"pre_{layout}"), os.path.join(tmp, f"pm_{layout}")
    os.makedirs(pre); os.makedirs(os.path.join(pm, "lora"))
    text = {"vocab_size": cfg.vocab_size, "hidden_size": cfg.hidden, "intermediate_size": cfg.intermediate,
            "num_hidden_layers": cfg.layers, "num_attention_heads": cfg.heads, "num_key_value_heads": cfg.kv_heads,
            "rms_norm_eps": cfg.rms_eps, "rope_theta": cfg.rope_theta, "hidden_act": "silu", "use_sliding_window": False,
            "rope_scaling": {"type": "mrope", "mrope_section": list(cfg.mrope_section)}}
    vis = {"depth": v.depth, "hidden_size": v.hidden, "num_heads": v.heads, "intermediate_size": v.intermediate,
           "patch_size": 14, "temporal_patch_size": 2, "spatial_merge_size": 2, "window_size": 112,
           "fullatt_block_indexes": list(v.fullatt), "out_hidden_size": cfg.hidden, "hidden_act": "silu"}
    json.dump(dict(text, image_token_id=cfg.image_token_id, vision_config=vis), open(os.path.join(pre, "config.json"), "w"))     # 4.50-era flat config
    W = {k: torch.from_numpy(a) for k, a in synth.qwen_make_weights(cfg, seed).items() if ".lora_" not in k}
    heads = ("value_head", "W_q", "W_k", "W_v", "ca_layernorm")

    def name5(k):
        return "model." + k if k.startswith("visual.") else k.replace("model.", "model.language_model.", 1)
    save_file({(name5(k) if layout == "5.x" else k): t.to(torch.bfloat16) for k, t in W.items() if k.split(".")[0] not in heads},
              os.path.join(pre, "model.safetensors"))
    torch.save({f"base_model.model.{k}": t for k, t in W.items() if k.split(".")[0] in heads}, os.path.join(pm, "pytorch_model.bin"))
    yaml.safe_dump({"is_general_preference": bool(cfg.is_general_preference), "add_cross_attention": bool(cfg.add_cross_attention),
                    "value_head_dim": int(cfg.value_head_dim), "general_preference_tau": 0.1}, open(os.path.join(pm, "reward_config.yaml"), "w"))
    Hq, Hkv, D, I = cfg.heads * cfg.head_dim, cfg.kv_heads * cfg.head_dim, cfg.hidden, cfg.intermediate
    mods = []
    for l in range(cfg.layers):
        p = f"model.layers.{l}."
        mods += [(p + "self_attn.q_proj", (Hq, D)), (p + "self_attn.k_proj", (Hkv, D)), (p + "self_attn.v_proj", (Hkv, D)),
                 (p + "self_attn.o_proj", (D, Hq)), (p + "mlp.gate_proj", (I, D)), (p + "mlp.up_proj", (I, D)), (p + "mlp.down_proj", (D, I))]
    skip = ("model.layers.0.self_attn.k_proj",)
    sd, eff = _adapter(mods, r, seed + 1, skip)
    if layout == "5.x":
        sd = {k.replace("base_model.model.model.layers.", "base_model.model.model.language_model.layers."): t for k, t in sd.items()}
    save_file(sd, os.path.join(pm, "lora", "adapter_model.safetensors"))
    json.dump({"r": r, "lora_alpha": alpha, "peft_type": "LORA",
               "target_modules": ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]},
              open(os.path.join(pm, "lora", "adapter_config.json"), "w"))
    Wo = {k: (t.to(torch.bfloat16).float() if k.split(".")[0] not in heads else t.float()) for k, t in W.items()}
    Wo.update(eff)
    Wo["lora_scaling"] = alpha / r
    return pre, pm, Wo, len(mods) - len(skip)
