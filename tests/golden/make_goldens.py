#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE ITSELF (imported from /root/reference).

Runs only in the build container (the reference never travels to the GPU box).  The JSON it
writes holds data only: config, seed, input recipe and the reference's outputs.  Inputs and weights
are regenerated anywhere from (config, seed) by llava_reward_amd.synth, so nothing large is stored.

    python tests/golden/make_goldens.py small        # ref_small_* cases (minutes)
    python tests/golden/make_goldens.py full         # full-size Phi-3.5-V, B=1 (~5-10 min, ~35 GB RSS: one full-size job at a time)
    ... full_b2 | full_seed1..5 | full_gpm | pair_sample        more full-size Phi rows (B=2 ragged with 64-element taps; seeds / grids)
    ... outlier_small | outlier_full | outlier_full_gpm          outlier-bearing weights (synth.PROFILE_OUTLIER)
    ... e4m3_small | llava_full_e4m3                             e4m3-valued weights (synth.PROFILE_E4M3; BASELINE configs[4])
    ... llava | llava_full | llava_full_seed1..2 | llava_full_outlier | qwen | qwen_full | qwen_full_seed1..2 | qwen_full_outlier
    ... mean | layer_id | rope_long | rope_at_orig | train                           the small special cases

Import recipe = SURVEY.md Appendix A (stubs for deepspeed/peft/loralib, use_cache=False, eager
attention, un-patched get_img_features == hidden_states[-2][:,1:]).
"""
import json
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
from llava_reward_amd import synth  # noqa: E402


def import_reference():
    import transformers  # noqa: F401  (real imports first: availability checks choke on stub modules)
    from unittest.mock import MagicMock
    from transformers import (AutoConfig, AutoModel, AutoModelForCausalLM, BitsAndBytesConfig,  # noqa: F401
                              CLIPVisionModel, Qwen2_5_VLModel, Qwen2_5_VLForConditionalGeneration,
                              LlavaNextForConditionalGeneration)
    import transformers.generation.utils  # noqa: F401
    from transformers.integrations.deepspeed import HfDeepSpeedConfig  # noqa: F401

    class _Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return MagicMock()

    for n in ["deepspeed", "deepspeed.ops", "deepspeed.ops.adam", "deepspeed.runtime", "deepspeed.runtime.zero",
              "deepspeed.runtime.zero.partition_parameters", "peft", "peft.tuners", "peft.tuners.lora", "loralib"]:
        m = _Stub(n)
        m.__path__ = []
        sys.modules[n] = m
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    from llava_reward.models.rw_model_general_preference import _get_reward_model, Phi3RMSNorm
    from llava_reward.models.base_mllm.phi3_v.modeling_phi3_v import Phi3VModel, Phi3VForCausalLM
    from llava_reward.models.base_mllm.phi3_v.configuration_phi3_v import Phi3VConfig
    return _get_reward_model, Phi3RMSNorm, Phi3VModel, Phi3VForCausalLM, Phi3VConfig


def build_reference_model(ref, cfg: synth.RewardConfig, seed: int, layer_id: int = 32, mean_hidden_state=None, profile: int = 0):
    _get_reward_model, Phi3RMSNorm, Phi3VModel, Phi3VForCausalLM, Phi3VConfig = ref
    assert cfg.clip == synth.ClipConfig(), "the reference hard-wires CLIP ViT-L/14-336"
    hcfg = Phi3VConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
        num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, num_key_value_heads=cfg.heads,
        max_position_embeddings=cfg.max_pos, original_max_position_embeddings=cfg.orig_max_pos,
        rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
        rope_scaling={"type": "su", "short_factor": list(cfg.short_factor), "long_factor": list(cfg.long_factor)},
        sliding_window=262144, use_cache=False, pad_token_id=min(32000, cfg.vocab_size - 1),
        bos_token_id=1, eos_token_id=min(32000, cfg.vocab_size - 1),
        embd_layer={"embedding_cls": "image", "hd_transform_order": "sub_glb", "projection_cls": "mlp",
                    "use_hd_transform": True, "with_learnable_separator": True},
        img_processor={"image_dim_out": 1024, "model_name": "openai/clip-vit-large-patch14-336",
                       "name": "clip_vision_model", "num_img_tokens": 144})
    hcfg._attn_implementation = "eager"
    cls = _get_reward_model(Phi3VForCausalLM, Phi3VModel, RMSNorm_class=Phi3RMSNorm, RMSNorm_class_eps=cfg.ca_eps,
                            is_general_preference=cfg.is_general_preference,
                            add_cross_attention=cfg.add_cross_attention, value_head_dim=cfg.value_head_dim, layer_id=layer_id,
                            mean_hidden_state=mean_hidden_state)
    t0 = time.time()
    # meta-device construction skips the reference's random init of parameters we overwrite anyway
    model = cls(hcfg)
    model.model_type = "phi3v"
    model.eval()
    print(f"  reference model built in {time.time() - t0:.1f}s", flush=True)
    # load synthetic weights by (canonicalised) name
    specs = {n: (shape, std, off) for n, shape, std, off in synth.weight_specs(cfg)}
    used = set()
    with torch.no_grad():
        for name, p in list(model.named_parameters()) + list(model.named_buffers()):
            canon = name
            if "img_processor." in name and "img_processor.vision_model." not in name:
                canon = name.replace("img_processor.", "img_processor.vision_model.")
            if canon in specs:
                shape, std, off = specs[canon]
                assert tuple(p.shape) == tuple(shape), (name, p.shape, shape)
                p.copy_(torch.from_numpy(synth.gen_tensor(seed, canon, shape, std, off, profile=profile)))
                used.add(canon)
    missing = set(specs) - used
    assert not missing, f"weights not consumed by the reference: {sorted(missing)[:5]}"
    return model


def fingerprint(t: torch.Tensor, n: int = 16):
    """Deterministic sample of n elements + abs-mean, enough to localise a divergence."""
    f = t.detach().float().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, n, dtype=torch.float64).long()
    return {"shape": list(t.shape), "abs_mean": float(f.abs().mean()), "idx": idx.tolist(),
            "vals": [float(v) for v in f[idx]]}


def run_case(ref, name, cfg, seed, caption_lens, grids, max_crops, taps=True, layer_id=32, mean_hidden_state=None, extra_left_pad=0,
             train=False, right_padded=False, profile=0, tap_samples=16, tap_layers=None):
    print(f"[{name}] building", flush=True)
    model = build_reference_model(ref, cfg, seed, layer_id, mean_hidden_state, profile)
    if train:
        model.train()          # rw_model:410-415 / :429-434: reward read at the last position; every dropout of the path has p = 0
    batch = synth.pad_left(synth.synth_batch(cfg, seed, caption_lens, grids, max_crops=max_crops), extra_left_pad)
    if right_padded:
        batch = synth.right_pad(batch)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    t0 = time.time()
    with torch.no_grad():
        reward, outputs = model.custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"],
                                               tb["image_sizes"], return_output=True)
    dt = time.time() - t0
    print(f"[{name}] reference custom_forward {dt:.1f}s reward={reward.flatten().tolist()}", flush=True)
    out = {"name": name, "config": cfg.to_json(), "seed": seed, "caption_lens": caption_lens,
           "grids": [list(g) for g in grids] if not isinstance(grids[0], int) else list(grids),
           "max_crops": max_crops, "reward": reward.float().tolist(), "layer_id": layer_id,
           "mean_hidden_state": bool(mean_hidden_state), "extra_left_pad": extra_left_pad, "train": train, "right_padded": right_padded,
           "reward_shape": list(reward.shape), "weight_profile": profile,
           "reference_forward_seconds": dt, "threads": torch.get_num_threads(),
           "torch": torch.__version__, "dtype": "float32"}
    if taps:
        hs = outputs["hidden_states"]
        out["taps"] = {"embeds": fingerprint(hs[0], tap_samples), "vision_embeds": fingerprint(hs[-1], tap_samples),
                       "final_norm": fingerprint(outputs["last_hidden_state"], tap_samples)}
        # hs[l + 1] = the residual stream leaving decoder layer l; the LAST layer's output is only there normed (= final_norm)
        for l in range(cfg.layers):
            if l in (tap_layers or (0, 1, cfg.layers // 2, cfg.layers - 1)) and l + 1 < len(hs) - 2:
                out["taps"][f"layer{l}"] = fingerprint(hs[l + 1], tap_samples)
    path = os.path.join(HERE, f"{name}.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(f"[{name}] wrote {path}", flush=True)
    del model
    return out


def run_pair_sample(ref, name="ref_full_pair_sample", seed=1234):
    """BASELINE configs[0] (eval/simple_inference.py:16-31): ONE caption, TWO 512x640 images -> two B=1 custom_forward calls at
    full Phi-3.5-V size -> preference_compute.  The sample JPEGs and the Phi tokenizer do not travel, so the images are seeded
    synthetic RGB of the same size (PIL size (512, 640) = 640 rows x 512 columns -> HD transform -> 1344 x 1344 -> 17 crops,
    V = 2509), pixel_values come from the HD-transform oracle (bit-exact with Pillow for the local crops) and the prompt goes
    through synth.StandInTokenizer with the reference's prompt construction and image-slot merge
    (eval/reward_adaptor_loader.py:163-167, processing_phi3_v.py:407-454)."""
    sys.path.insert(0, ROOT)
    from oracle import phi3v_hd_transform_oracle as hd
    cfg = synth.full_config()
    model = build_reference_model(ref, cfg, seed)
    tok = synth.StandInTokenizer()
    caption = synth.SAMPLE_CAPTION
    msg = {"role": "user", "content": f"<|image_1|>\n{caption}"}
    prompt = tok.apply_chat_template([msg], tokenize=False, add_generation_prompt=True)[:-22] + tok.eos_token
    head, tail = prompt.split("<|image_1|>")
    rewards, meta = [], []
    for i in range(2):
        img = synth.synth_image(seed, f"pair.image{i}", 640, 512, True)
        pix, (H, W), ntok = hd.preprocess(img, 16)
        ids = tok(head).input_ids + [-1] * ntok + tok(tail).input_ids
        input_ids = torch.tensor(ids, dtype=torch.long)[None]
        mask = torch.ones_like(input_ids)
        t0 = time.time()
        with torch.no_grad():
            r, _ = model.custom_forward(input_ids, mask, torch.from_numpy(pix)[None], torch.tensor([[H, W]]))
        print(f"[{name}] image {i}: reward {r.flatten().tolist()} ({time.time() - t0:.1f}s, S={input_ids.shape[1]}, V={ntok})", flush=True)
        rewards.append(r.float())
        meta.append({"image": f"synth_image({seed}, 'pair.image{i}', 640, 512, smooth=True)", "image_sizes": [H, W],
                     "num_img_tokens": ntok, "seq_len": int(input_ids.shape[1]),
                     "input_ids_head": ids[:8], "input_ids_tail": ids[-8:]})

    class A:
        is_general_preference, value_head_dim, general_preference_tau = cfg.is_general_preference, cfg.value_head_dim, cfg.general_preference_tau
    # the reference's own formula (eval/reward_adaptor_loader.py:174-181), evaluated by the reference module when importable
    try:
        from eval.reward_adaptor_loader import preference_compute as ref_pc
        prob = ref_pc(A, rewards[0], rewards[1])
        pc_src = "eval.reward_adaptor_loader.preference_compute (imported)"
    except Exception as e:      # the loader module imports peft/deepspeed symbols at import time
        prob = torch.sigmoid((rewards[0] - rewards[1]) / A.general_preference_tau).squeeze(-1).float().numpy()
        pc_src = f"formula of eval/reward_adaptor_loader.py:179-180 (module import failed: {type(e).__name__})"
    out = {"name": name, "config": cfg.to_json(), "seed": seed, "caption": caption, "num_crops": 16, "rows": meta,
           "reward": [r.flatten().tolist() for r in rewards], "prob": [float(p) for p in prob.reshape(-1)], "preference_compute": pc_src,
           "tau": A.general_preference_tau, "torch": torch.__version__, "dtype": "float32"}
    with open(os.path.join(HERE, f"{name}.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(f"[{name}] prob {out['prob']}", flush=True)
    return out


def canon_llava_name(name: str) -> str:
    """transformers 5.x module tree -> checkpoint (4.50 era) tensor names of llava-v1.6-*-hf."""
    n = name
    if n.startswith("model."):
        n = n[len("model."):]
    n = n.replace("language_model.layers.", "language_model.model.layers.")
    n = n.replace("language_model.embed_tokens.", "language_model.model.embed_tokens.")
    n = n.replace("language_model.norm.", "language_model.model.norm.")
    if n.startswith("vision_tower.") and not n.startswith("vision_tower.vision_model."):
        n = n.replace("vision_tower.", "vision_tower.vision_model.", 1)
    return n


def run_llava_case(name, cfg, seed, caption_lens, image_sizes, max_crops, mean_hidden_state=None, profile=0):
    """Reference custom_forward, model_type='llava' (rw_model_general_preference.py:372-375,407-448)."""
    import transformers
    from transformers import CLIPVisionConfig, LlavaNextConfig, LlavaNextForConditionalGeneration, MistralConfig
    sys.path.insert(0, "/root/reference")
    from llava_reward.models.rw_model_general_preference import _get_reward_model, LlamaRMSNorm
    c = cfg.clip
    vcfg = CLIPVisionConfig(hidden_size=c.hidden, intermediate_size=c.mlp, num_hidden_layers=c.layers_used + 1,
                            num_attention_heads=c.heads, image_size=c.image, patch_size=c.patch, hidden_act="quick_gelu",
                            layer_norm_eps=c.ln_eps, projection_dim=64)
    tcfg = MistralConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
                         num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads,
                         head_dim=cfg.head_dim, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, sliding_window=None,
                         max_position_embeddings=32768, use_cache=False, pad_token_id=cfg.pad_token_id)
    hcfg = LlavaNextConfig(vision_config=vcfg, text_config=tcfg, image_grid_pinpoints=[list(p) for p in cfg.pinpoints],
                           image_token_index=cfg.image_token_id, vision_feature_layer=-2,
                           vision_feature_select_strategy="default", projector_hidden_act="gelu")
    hcfg._attn_implementation = "eager"
    cls = _get_reward_model(LlavaNextForConditionalGeneration, LlavaNextForConditionalGeneration, RMSNorm_class=LlamaRMSNorm,
                            RMSNorm_class_eps=1e-5, is_general_preference=cfg.is_general_preference,
                            add_cross_attention=False, value_head_dim=cfg.value_head_dim, model_type="llava",
                            mean_hidden_state=mean_hidden_state)
    model = cls(hcfg)
    model.eval()
    specs = {n: (sh, std, off) for n, sh, std, off in synth.llava_weight_specs(cfg)}
    used = set()
    with torch.no_grad():
        for pname, p in model.named_parameters():
            cn = canon_llava_name(pname)
            if cn in specs:
                sh, std, off = specs[cn]
                assert tuple(p.shape) == tuple(sh), (pname, p.shape, sh)
                p.copy_(torch.from_numpy(synth.gen_tensor(seed, cn, sh, std, off, profile=profile)))
                used.add(cn)
    missing = set(specs) - used
    assert not missing, f"weights not consumed by the reference: {sorted(missing)[:6]}"
    batch = synth.llava_synth_batch(cfg, seed, caption_lens, image_sizes, max_crops=max_crops)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    t0 = time.time()
    with torch.no_grad():
        reward, _ = model.custom_forward(inputs_batch=tb)
    dt = time.time() - t0
    print(f"[{name}] reference llava custom_forward {dt:.1f}s reward={reward.flatten().tolist()}", flush=True)
    out = {"name": name, "backbone": "llava", "config": cfg.to_json(), "seed": seed, "caption_lens": caption_lens,
           "image_sizes": [list(x) for x in image_sizes], "max_crops": max_crops, "reward": reward.float().tolist(),
           "mean_hidden_state": bool(mean_hidden_state), "weight_profile": profile,
           "transformers": transformers.__version__, "torch": torch.__version__, "dtype": "float32"}
    with open(os.path.join(HERE, f"{name}.json"), "w") as f:
        json.dump(out, f, indent=1)
    return out


def canon_qwen_name(name: str) -> str:
    """transformers 5.x module tree -> checkpoint (4.50 era) tensor names of Qwen2.5-VL-*-Instruct."""
    n = name
    if n.startswith("model.visual."):
        return n[len("model."):]
    if n.startswith("model.language_model."):
        return "model." + n[len("model.language_model."):]
    return n


def run_qwen_case(name, cfg, seed, caption_lens, grids, mean_hidden_state=None, profile=0):
    """Reference custom_forward, model_type='qwen' (rw_model_general_preference.py:354-371,387-397,407-448).

    Shims for running the 4.50-era reference code on transformers 5.x:
      1. cfg.hidden_size = cfg.text_config.hidden_size   (4.50's Qwen2_5_VLConfig was flat; rw_model:313 reads it)
      2. model.__dict__['visual'] = model.model.visual    (5.x moved the ViT under .model; rw_model:356 calls it)
      3. inputs_batch carries mm_token_type_ids           (5.x builds the 3-D mRoPE positions only when the
         processor's token-type ids are passed; 4.50 derived the same positions from input_ids alone)"""
    import transformers
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration, Qwen2_5_VLModel
    sys.path.insert(0, "/root/reference")
    from llava_reward.models.rw_model_general_preference import _get_reward_model, Qwen2RMSNorm
    v = cfg.vision
    vcfg = dict(depth=v.depth, hidden_size=v.hidden, hidden_act="silu", intermediate_size=v.intermediate, num_heads=v.heads,
                in_channels=v.in_ch, patch_size=v.patch, spatial_merge_size=v.merge, temporal_patch_size=v.temporal_patch,
                tokens_per_second=2, window_size=v.window, out_hidden_size=cfg.hidden, fullatt_block_indexes=list(v.fullatt))
    tcfg = dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
                num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads,
                hidden_act="silu", max_position_embeddings=32768, rms_norm_eps=cfg.rms_eps, use_cache=False,
                tie_word_embeddings=False, use_sliding_window=False, sliding_window=32768, max_window_layers=cfg.layers,
                attention_dropout=0.0, pad_token_id=None,
                rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta, "mrope_section": list(cfg.mrope_section)})
    hcfg = Qwen2_5_VLConfig(text_config=tcfg, vision_config=vcfg, image_token_id=cfg.image_token_id,
                            video_token_id=cfg.image_token_id + 1, vision_start_token_id=cfg.image_token_id - 3,
                            vision_end_token_id=cfg.image_token_id - 2)
    hcfg._attn_implementation = "eager"
    hcfg.hidden_size = hcfg.text_config.hidden_size                                      # shim 1
    cls = _get_reward_model(Qwen2_5_VLForConditionalGeneration, Qwen2_5_VLModel, RMSNorm_class=Qwen2RMSNorm,
                            RMSNorm_class_eps=1e-6, is_general_preference=cfg.is_general_preference,
                            add_cross_attention=cfg.add_cross_attention, value_head_dim=cfg.value_head_dim,
                            mean_hidden_state=mean_hidden_state)
    model = cls(hcfg)
    model.model_type = "qwen"                                                            # reward_adaptor_loader.py:79
    model.__dict__["visual"] = model.model.visual                                        # shim 2
    model.eval()
    specs = {n: (sh, std, off) for n, sh, std, off in synth.qwen_weight_specs(cfg)}
    used = set()
    with torch.no_grad():
        for pname, p in model.named_parameters():
            cn = canon_qwen_name(pname)
            if cn in specs:
                sh, std, off = specs[cn]
                assert tuple(p.shape) == tuple(sh), (pname, p.shape, sh)
                p.copy_(torch.from_numpy(synth.gen_tensor(seed, cn, sh, std, off, profile=profile)))
                used.add(cn)
            else:
                assert cn == "lm_head.weight", f"reference parameter without a spec: {pname}"
    missing = set(specs) - used
    assert not missing, f"weights not consumed by the reference: {sorted(missing)[:6]}"
    batch = synth.qwen_synth_batch(cfg, seed, caption_lens, grids)
    tb = {k: torch.from_numpy(val) for k, val in batch.items()}
    tb["mm_token_type_ids"] = (tb["input_ids"] == cfg.image_token_id).int()             # shim 3
    t0 = time.time()
    with torch.no_grad():
        reward, _ = model.custom_forward(inputs_batch=tb)
    dt = time.time() - t0
    print(f"[{name}] reference qwen custom_forward {dt:.1f}s reward={reward.flatten().tolist()}", flush=True)
    out = {"name": name, "backbone": "qwen", "config": cfg.to_json(), "seed": seed, "caption_lens": caption_lens,
           "grids": [list(g) for g in grids], "reward": reward.float().tolist(), "mean_hidden_state": bool(mean_hidden_state),
           "n_ca_rows": (tb["input_ids"] == synth.QWEN_CA_TOKEN_ID).sum(dim=1).tolist(), "weight_profile": profile,
           "transformers": transformers.__version__, "torch": torch.__version__, "dtype": "float32"}
    with open(os.path.join(HERE, f"{name}.json"), "w") as f:
        json.dump(out, f, indent=1)
    return out


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    torch.manual_seed(0)
    ref = import_reference()
    if which == "small":
        C = synth.ref_small_config
        run_case(ref, "ref_small_bt_ca", C(), 1234, [6, 3], (1, 1), None)
        run_case(ref, "ref_small_gpm2_ca", C(is_general_preference=True, value_head_dim=2), 1234, [6, 3], (1, 1), None)
        run_case(ref, "ref_small_bt_noca", C(add_cross_attention=False), 4321, [4], (1, 1), None)
        run_case(ref, "ref_small_bt_ca_ragged", C(), 99, [5, 9], [(1, 1), (1, 2)], 3)
        run_case(ref, "ref_small_gpm4_ca", C(is_general_preference=True, value_head_dim=4), 7, [2], (1, 1), None)
    elif which == "llava":
        C = synth.llava_tiny_config
        run_llava_case("ref_llava_tiny_bt", C(), 11, [6, 3], [(336, 336), (336, 336)], None)
        run_llava_case("ref_llava_tiny_gpm2", C(is_general_preference=True, value_head_dim=2), 12, [5, 9], [(512, 640), (336, 336)], 5)
        run_llava_case("ref_llava_tiny_wide", C(), 13, [4], [(300, 900)], None)
        run_llava_case("ref_llava_tiny_tall", C(layers=3), 14, [2, 7], [(400, 300), (672, 672)], 5)
    elif which == "mean":
        # rw_model:398-406 `mean_hidden_state=True` (a training-script option; SkipCA + norm on every token, masked mean)
        run_case(ref, "ref_small_mean_bt_ca", synth.ref_small_config(), 78, [5, 9], [(1, 1), (1, 2)], 3, taps=False, mean_hidden_state=True)
        run_case(ref, "ref_small_mean_gpm2_noca", synth.ref_small_config(is_general_preference=True, value_head_dim=2, add_cross_attention=False),
                 79, [3, 6], (1, 1), None, taps=False, mean_hidden_state=True)
        run_llava_case("ref_llava_tiny_mean_bt", synth.llava_tiny_config(), 15, [6, 3], [(336, 336), (512, 640)], 5, mean_hidden_state=True)
        run_qwen_case("ref_qwen_quirk_mean_bt", synth.qwen_quirk_config(), 26, [2, 7, 4], [(8, 8), (12, 16), (8, 8)], mean_hidden_state=True)
    elif which == "layer_id":
        # rw_model:349-352 with layer_id != 32: hidden_states[1] = the residual stream entering decoder layer 1 (no final norm)
        run_case(ref, "ref_small_layer1_bt_ca", synth.ref_small_config(), 77, [4, 7], (1, 1), None, layer_id=1)
        # ... together with mean_hidden_state (rw_model:398-406 pools whatever :349-352 selected): SkipCA + masked mean over hidden_states[1]
        run_case(ref, "ref_small_mean_layer1_bt_ca", synth.ref_small_config(), 81, [5, 9], [(1, 1), (1, 2)], 3, taps=False, layer_id=1, mean_hidden_state=True)
    elif which == "qwen":
        C, Q = synth.qwen_tiny_config, synth.qwen_quirk_config
        run_qwen_case("ref_qwen_tiny_bt", C(), 21, [6, 3], [(16, 16), (16, 16)])
        run_qwen_case("ref_qwen_tiny_gpm2_ragged", C(is_general_preference=True, value_head_dim=2), 22, [5, 9],
                      [(10, 6), (18, 22)])
        run_qwen_case("ref_qwen_tiny_noca", C(add_cross_attention=False, layers=3), 23, [4], [(8, 20)])
        run_qwen_case("ref_qwen_quirk_bt", Q(), 24, [2, 7, 4], [(8, 8), (12, 16), (8, 8)])
        run_qwen_case("ref_qwen_quirk_gpm4", Q(is_general_preference=True, value_head_dim=4), 25, [3, 3], [(16, 16), (16, 16)])
    elif which == "qwen_full":
        # Qwen2.5-VL-7B shapes, one row: 448x448 image (the reference's min_pixels floor for a 336^2 input) = 32x32
        # patches = 256 image tokens, 128-token caption (BASELINE.json config #4); ~35 GB RSS
        run_qwen_case("ref_qwen_full_bt", synth.qwen_full_config(), 1234, [128], [(32, 32)])
    elif which == "llava_full":
        run_llava_case("ref_llava_full_bt", synth.llava_full_config(), 1234, [128], [(336, 336)], None)
    elif which == "full":
        run_case(ref, "ref_full_bt_ca", synth.full_config(), 1234, [128], (4, 4), None)
    elif which == "rope_long":
        # su-RoPE long-factor branch (modeling_phi3_v.py:449): the switch is taken on seq_len = the PADDED length S (:673).
        # S = 326 > original_max_position_embeddings = 300; in the second case EVERY row carries padding (S = 334 > 330 >= the
        # longest row's 326 valid tokens), which tells `S > orig` apart from `max(position_ids) + 1 > orig`.
        C = synth.ref_small_config
        run_case(ref, "ref_small_rope_long_bt_ca", C(orig_max_pos=300, max_pos=9600), 31, [5, 9], (1, 1), None)
        run_case(ref, "ref_small_rope_long_allpad_gpm2_ca", C(orig_max_pos=330, max_pos=9600, is_general_preference=True, value_head_dim=2),
                 32, [5, 9], (1, 1), None, extra_left_pad=8)
    elif which == "rope_at_orig":
        # S == original_max_position_embeddings exactly (327): eager / sdpa attention (`seq_len > orig`, modeling_phi3_v.py:449 via
        # :673) still takes the SHORT factors here; Phi3FlashAttention2 (:793-794, seq_len = S + 1) would take the long ones
        run_case(ref, "ref_small_rope_at_orig_bt_ca", synth.ref_small_config(orig_max_pos=327, max_pos=9600), 33, [5, 9], (1, 1), None)
    elif which == "train":
        # model.train(): the reward of the LAST position (rw_model:410-415 BT -> [B]; :429-434 GPM -> [B, d]), with left-padded rows (the
        # trainer's collate) and with right-padded rows (the last position is then a pad position of the shorter row)
        C = synth.ref_small_config
        run_case(ref, "ref_small_train_bt_ca", C(), 41, [6, 3], (1, 1), None, taps=False, train=True)
        run_case(ref, "ref_small_train_gpm2_ca_rightpad", C(is_general_preference=True, value_head_dim=2), 42, [6, 3], (1, 1), None, taps=False,
                 train=True, right_padded=True)
        run_case(ref, "ref_small_eval_bt_ca_rightpad", C(), 43, [2, 7], (1, 1), None, taps=False, right_padded=True)
    elif which == "pair_sample":
        run_pair_sample(ref)
    elif which == "full_gpm":
        run_case(ref, "ref_full_gpm2_ca", synth.full_config(is_general_preference=True, value_head_dim=2),
                 1234, [128], (4, 4), None)
    elif which == "full_b2":
        # full size, B = 2, ragged captions AND ragged crop grids -> left padding (4-D causal + padding mask,
        # modeling_phi3_v.py:1453-1459) and V_max zero padding + index_put scatter (:242-249) at full depth; 64-element taps
        run_case(ref, "ref_full_b2_ragged_bt_ca", synth.full_config(), 2024, [128, 37], [(4, 4), (2, 3)], 16, tap_samples=64,
                 tap_layers=(0, 1, 15, 16, 30))
    elif which.startswith("full_seed"):
        # more full-size rows: other seeds, caption lengths and crop grids (B = 1 each)
        cases = {"full_seed1": (101, [64], (3, 5)), "full_seed2": (202, [200], (2, 8)), "full_seed3": (303, [17], (1, 1)),
                 "full_seed4": (404, [128], (4, 4)), "full_seed5": (505, [96], (4, 3))}
        seed, caps, grid = cases[which]
        run_case(ref, f"ref_full_s{seed}_bt_ca", synth.full_config(), seed, caps, grid, 16, taps=False)
    elif which == "outlier_small":
        P = synth.PROFILE_OUTLIER
        C = synth.ref_small_config
        run_case(ref, "ref_small_outlier_bt_ca", C(), 51, [6, 3], (1, 1), None, profile=P)
        run_case(ref, "ref_small_outlier_gpm2_ca_ragged", C(is_general_preference=True, value_head_dim=2), 52, [5, 9], [(1, 1), (1, 2)], 3, profile=P)
        run_case(ref, "ref_small_outlier_bt_noca", C(add_cross_attention=False), 53, [4], (1, 1), None, profile=P)
    elif which == "outlier_full":
        run_case(ref, "ref_full_outlier_bt_ca", synth.full_config(), 606, [128], (4, 4), None, taps=False, profile=synth.PROFILE_OUTLIER)
    elif which == "outlier_full_gpm":
        run_case(ref, "ref_full_outlier_gpm2_ca", synth.full_config(is_general_preference=True, value_head_dim=2), 707, [77], (3, 4), 16, taps=False,
                 profile=synth.PROFILE_OUTLIER)
    elif which == "e4m3_small":
        run_case(ref, "ref_small_e4m3_bt_ca", synth.ref_small_config(), 61, [6, 3], (1, 1), None, profile=synth.PROFILE_E4M3)
        run_llava_case("ref_llava_tiny_e4m3_bt", synth.llava_tiny_config(), 62, [6, 3], [(336, 336), (512, 640)], 5, profile=synth.PROFILE_E4M3)
        run_llava_case("ref_llava_tiny_outlier_bt", synth.llava_tiny_config(), 63, [6, 3], [(336, 336), (300, 900)], 5, profile=synth.PROFILE_OUTLIER)
        run_qwen_case("ref_qwen_tiny_outlier_bt", synth.qwen_tiny_config(), 64, [6, 3], [(16, 16), (10, 6)], profile=synth.PROFILE_OUTLIER)
    elif which == "llava_full_e4m3":
        # BASELINE configs[4] "fp8 MFMA weight path": the reference on the DE-QUANTISED weights of an e4m3-weight LLaVA-v1.6-Mistral-7B
        run_llava_case("ref_llava_full_e4m3_bt", synth.llava_full_config(), 1234, [128], [(336, 336)], None, profile=synth.PROFILE_E4M3)
    elif which.startswith("llava_full_seed"):
        cases = {"llava_full_seed1": (111, [64], [(512, 640)]), "llava_full_seed2": (222, [150], [(336, 336)])}
        seed, caps, sizes = cases[which]
        run_llava_case(f"ref_llava_full_s{seed}_bt", synth.llava_full_config(), seed, caps, sizes, None)
    elif which == "llava_full_outlier":
        run_llava_case("ref_llava_full_outlier_bt", synth.llava_full_config(), 333, [128], [(336, 336)], None, profile=synth.PROFILE_OUTLIER)
    elif which.startswith("qwen_full_seed"):
        cases = {"qwen_full_seed1": (121, [64], [(24, 40)]), "qwen_full_seed2": (232, [200], [(32, 32)])}
        seed, caps, grids = cases[which]
        run_qwen_case(f"ref_qwen_full_s{seed}_bt", synth.qwen_full_config(), seed, caps, grids)
    elif which == "qwen_full_outlier":
        run_qwen_case("ref_qwen_full_outlier_bt", synth.qwen_full_config(), 343, [128], [(32, 32)], profile=synth.PROFILE_OUTLIER)
    else:
        raise SystemExit(f"unknown set {which}")


if __name__ == "__main__":
    main()
