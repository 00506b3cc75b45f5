"""CPU-only checks of the host side: the C-ABI library exports what include/*.h declares, the
checkpoint importer (HF safetensors + PEFT LoRA merge + reward heads), the drop-in callables'
contracts, and the multi-process shard/all-gather path (gloo, world_size 2)."""
import json
import os
import re
import types

import numpy as np
import pytest
import torch
import yaml

from llava_reward_amd import _lib, checkpoint, synth
from llava_reward_amd.model import RewardModel
from llava_reward_amd.reward_adaptor_loader import UnknownModelType, load_reward_adaptor, preference_compute
from llava_reward_amd.scoring import gather_rewards, score_pairwise, score_single, shard_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "llava_reward_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t|uint64_t|const char\*)\s+(lr_\w+)\(", hdr, flags=re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()                      # resolves every symbol; no compute without a GPU
    assert lib.lr_abi_version() == _lib.LR_ABI_VERSION == 9
    assert _lib.ModelDesc.struct_size.offset == 0


def test_desc_validation_without_gpu():
    """lr_create must reject a bad descriptor with an error message instead of crashing."""
    import ctypes as C
    lib = _lib.load()
    d = _lib.ModelDesc()
    d.struct_size = 4
    h = C.c_void_p()
    assert lib.lr_create(C.byref(d), 0, C.byref(h)) != 0
    assert b"struct_size" in lib.lr_last_error(None)


def _write_fake_checkpoint(tmp, cfg, seed, reward_cfg, with_lora=True):
    from safetensors.torch import save_file
    pre = os.path.join(tmp, "pretrain")
    pm = os.path.join(tmp, "pm")
    os.makedirs(pre)
    os.makedirs(os.path.join(pm, "lora"))
    hf = {"vocab_size": cfg.vocab_size, "hidden_size": cfg.hidden, "intermediate_size": cfg.intermediate,
          "num_hidden_layers": cfg.layers, "num_attention_heads": cfg.heads, "rms_norm_eps": cfg.rms_eps,
          "rope_theta": cfg.rope_theta, "max_position_embeddings": cfg.max_pos,
          "original_max_position_embeddings": cfg.orig_max_pos,
          "rope_scaling": {"type": "su", "short_factor": list(cfg.short_factor), "long_factor": list(cfg.long_factor)},
          "embd_layer": {"embedding_cls": "image", "hd_transform_order": "sub_glb", "projection_cls": "mlp",
                         "use_hd_transform": True, "with_learnable_separator": True},
          "img_processor": {"name": "clip_vision_model", "clip_hidden": cfg.clip.hidden, "clip_heads": cfg.clip.heads,
                            "clip_mlp": cfg.clip.mlp, "clip_layers_used": cfg.clip.layers_used}}
    json.dump(hf, open(os.path.join(pre, "config.json"), "w"))
    W = {k: torch.from_numpy(v) for k, v in synth.make_weights(cfg, seed).items()}
    heads = {k: v for k, v in W.items() if k.split(".")[0] in ("value_head", "W_q", "W_k", "W_v", "ca_layernorm")}
    base = {k: v.to(torch.bfloat16) for k, v in W.items() if k not in heads}
    base["lm_head.weight"] = torch.zeros(4, 4, dtype=torch.bfloat16)          # ignored by the importer
    save_file(base, os.path.join(pre, "model-00001-of-00001.safetensors"))
    # heads as DeepspeedStrategy.save_model_lora writes them (utils/deepspeed.py:343-357)
    sd = {("base_model.model." + k): v for k, v in heads.items()}
    g = torch.Generator().manual_seed(1)
    proj = {f"base_model.model.model.vision_embed_tokens.img_projection.{i}.{p}":
            torch.randn(W[f"model.vision_embed_tokens.img_projection.{i}.{p}"].shape, generator=g) * 0.02
            for i in (0, 2) for p in ("weight", "bias")}
    sd.update(proj)
    torch.save(sd, os.path.join(pm, "pytorch_model.bin"))
    yaml.safe_dump(reward_cfg, open(os.path.join(pm, "reward_config.yaml"), "w"))
    r = 4
    lora = {}
    for mod in ("model.layers.0.self_attn.qkv_proj", "model.layers.1.mlp.down_proj"):
        out_f, in_f = W[mod + ".weight"].shape
        lora[f"base_model.model.{mod}.lora_A.weight"] = torch.randn(r, in_f, generator=g) * 0.1
        lora[f"base_model.model.{mod}.lora_B.weight"] = torch.randn(out_f, r, generator=g) * 0.1
    json.dump({"r": r, "lora_alpha": 8, "target_modules": ["qkv_proj", "down_proj"]}, open(os.path.join(pm, "lora", "adapter_config.json"), "w"))
    torch.save(lora, os.path.join(pm, "lora", "adapter_model.bin"))
    return pre, pm, W, lora, proj


def test_load_reward_adaptor_contract(tmp_path):
    cfg = synth.tiny_config(is_general_preference=True, value_head_dim=2)
    reward_cfg = {"is_general_preference": True, "add_cross_attention": True, "value_head_dim": 2, "general_preference_tau": 0.1}
    pre, pm, W, lora, proj = _write_fake_checkpoint(str(tmp_path), cfg, 3, reward_cfg)
    args = types.SimpleNamespace(pm_path=pm, pretrain=pre, cache_dir=None, ft_projector=True, disable_fast_tokenizer=False)
    ret = load_reward_adaptor(args, "phi3v", os.path.join(pm, "reward_config.yaml"))
    assert len(ret) == 2 and ret[0] is args                    # (args, model); args mutated in place (:27-30)
    assert args.is_general_preference is True and args.value_head_dim == 2 and args.general_preference_tau == 0.1
    model = ret[1]
    assert model.model_type == "phi3v" and model.device.type == "cpu"
    assert model.eval() is model and model.to("cpu") is model
    w = model._weights
    # the decoder adapter stays UN-MERGED, as the reference runs it (:44-45): A as is, B pre-scaled by alpha / r, base weights untouched
    mod = "model.layers.0.self_attn.qkv_proj"
    A, Bm = lora[f"base_model.model.{mod}.lora_A.weight"], lora[f"base_model.model.{mod}.lora_B.weight"]
    assert model.config.lora_rank == 4 and args.lora_modules == {"unmerged": 2, "merged": 0, "skipped": 0}
    assert torch.equal(w[mod + ".lora_A.weight"], A) and torch.equal(w[mod + ".lora_B.weight"], 2.0 * Bm)
    assert torch.equal(w[mod + ".weight"].float(), W[mod + ".weight"])
    assert torch.equal(w["model.layers.1.self_attn.qkv_proj.weight"].float(), W["model.layers.1.self_attn.qkv_proj.weight"])
    # linears the adapter does not target: zero adapters (the engine expects every slot)
    z = w["model.layers.1.self_attn.qkv_proj.lora_B.weight"]
    assert z.shape == (3 * cfg.hidden, 4) and not z.any() and not w["model.layers.0.mlp.gate_up_proj.lora_A.weight"].any()
    # debug switch: merged on the host in fp32, W + (alpha/r) B A, no adapter tensors
    args_m = types.SimpleNamespace(pm_path=pm, pretrain=pre, cache_dir=None, ft_projector=True, disable_fast_tokenizer=False, merge_lora=True)
    wm = load_reward_adaptor(args_m, "phi3v", os.path.join(pm, "reward_config.yaml"))[1]._weights
    assert torch.allclose(wm[mod + ".weight"], W[mod + ".weight"] + 2.0 * Bm @ A, atol=1e-6) and mod + ".lora_A.weight" not in wm
    assert args_m.lora_modules == {"unmerged": 0, "merged": 2, "skipped": 0}
    # ft_projector override and heads (substring-filtered, reward_adaptor_loader.py:46-60)
    assert torch.equal(w["model.vision_embed_tokens.img_projection.2.bias"], proj["base_model.model.model.vision_embed_tokens.img_projection.2.bias"])
    assert torch.equal(w["value_head.weight"], W["value_head.weight"]) and w["W_k.weight"].shape == (cfg.hidden, cfg.hidden)
    assert set(n for n, *_ in synth.weight_specs(model.config)) <= set(w)
    # an adapter module that resolves to nothing must not be dropped silently; off-path modules (deleted CLIP layer 24, lm_head) may
    lora_file = os.path.join(pm, "lora", "adapter_model.bin")
    good = torch.load(lora_file)
    torch.save(dict(good, **{"base_model.model.model.layers.0.self_attn.qkvproj.lora_A.weight": A,
                             "base_model.model.model.layers.0.self_attn.qkvproj.lora_B.weight": Bm}), lora_file)
    with pytest.raises(KeyError, match="matches no weight"):
        load_reward_adaptor(args, "phi3v", os.path.join(pm, "reward_config.yaml"))
    clip24 = "model.vision_embed_tokens.img_processor.vision_model.encoder.layers.%d.mlp.fc1" % cfg.clip.layers_used
    clip0 = "model.vision_embed_tokens.img_processor.vision_model.encoder.layers.0.mlp.fc1"
    extra = {}
    for m in (clip24, clip0):
        extra[f"base_model.model.{m}.lora_A.default.weight"] = torch.ones(4, cfg.clip.hidden) * 0.01      # adapter-name infix accepted
        extra[f"base_model.model.{m}.lora_B.default.weight"] = torch.ones(cfg.clip.mlp, 4) * 0.01
    torch.save(dict(good, **extra), lora_file)
    w2 = load_reward_adaptor(args, "phi3v", os.path.join(pm, "reward_config.yaml"))[1]._weights
    assert args.lora_modules == {"unmerged": 2, "merged": 1, "skipped": 1}        # vision-tower adapter: merged; deleted layer: skipped
    assert torch.allclose(w2[clip0 + ".weight"], W[clip0 + ".weight"] + 2.0 * 4 * 1e-4, atol=1e-6)
    torch.save(good, lora_file)
    for bad_cfg in ({"use_rslora": True}, {"rank_pattern": {"qkv_proj": 8}}, {"fan_in_fan_out": True}, {"alpha_pattern": {"x": 1}}):
        json.dump(dict({"r": 4, "lora_alpha": 8}, **bad_cfg), open(os.path.join(pm, "lora", "adapter_config.json"), "w"))
        with pytest.raises(NotImplementedError):
            load_reward_adaptor(args, "phi3v", os.path.join(pm, "reward_config.yaml"))
    json.dump({"r": 4, "lora_alpha": 8, "target_modules": ["qkv_proj", "down_proj"]}, open(os.path.join(pm, "lora", "adapter_config.json"), "w"))
    # fails loudly on CPU: no fallback path
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.custom_forward(torch.zeros(1, 4, dtype=torch.long), torch.ones(1, 4, dtype=torch.long), torch.zeros(1, 2, 3, 336, 336), torch.tensor([[336, 336]]))
    # error behaviour
    with pytest.raises(UnboundLocalError):
        load_reward_adaptor(args, "gemma", os.path.join(pm, "reward_config.yaml"))
    with pytest.raises((ValueError, KeyError)):              # a Phi-3-V checkpoint is not a Qwen2.5-VL one
        load_reward_adaptor(args, "qwen", os.path.join(pm, "reward_config.yaml"))
    with pytest.raises(FileNotFoundError):
        load_reward_adaptor(args, "phi3v", os.path.join(pm, "missing.yaml"))
    bad = os.path.join(str(tmp_path), "bad.yaml")
    yaml.safe_dump({"is_general_preference": False}, open(bad, "w"))
    with pytest.raises(KeyError):
        load_reward_adaptor(args, "phi3v", bad)
    assert issubclass(UnknownModelType, UnboundLocalError)


def test_preference_compute_matches_reference_formula():
    a = types.SimpleNamespace(is_general_preference=True, value_head_dim=2, general_preference_tau=0.1)
    c = torch.tensor([[0.3, -0.2], [1.0, 0.5]], dtype=torch.bfloat16)
    r = torch.tensor([[0.1, 0.4], [-0.5, 0.25]], dtype=torch.bfloat16)
    p = preference_compute(a, c, r)
    exp = torch.sigmoid((c[:, 0] * r[:, 1] - c[:, 1] * r[:, 0]) / 0.1).float().numpy()
    assert p.dtype == np.float32 and p.shape == (2,) and np.array_equal(p, exp)
    a = types.SimpleNamespace(is_general_preference=False, value_head_dim=1, general_preference_tau=0.5)
    p = preference_compute(a, torch.tensor([[0.3], [0.0]]), torch.tensor([[0.1], [0.2]]))
    np.testing.assert_allclose(p, 1 / (1 + np.exp(-np.array([0.2, -0.2]) / 0.5)), rtol=1e-6)
    # GPM with d=4 falls to the BT-style branch exactly like the reference (:175-180)
    a = types.SimpleNamespace(is_general_preference=True, value_head_dim=4, general_preference_tau=1.0)
    assert preference_compute(a, torch.zeros(2, 4), torch.zeros(2, 4)).shape == (2, 4)


def test_shard_rows_partition():
    for n in (0, 1, 7, 32, 33):
        for ws in (1, 2, 3, 8):
            sl = [shard_rows(n, r, ws) for r in range(ws)]
            assert sl[0].start == 0 and sl[-1].stop == n
            assert all(a.stop == b.start for a, b in zip(sl, sl[1:]))
            sizes = [s.stop - s.start for s in sl]
            assert max(sizes) - min(sizes) <= 1


class _FakeModel:
    """Deterministic stand-in for the HIP model: reward = f(row content), so sharding must not change it."""
    device = torch.device("cpu")

    def __init__(self, d):
        self.d = self.value_head_dim = d

    def custom_forward(self, ids, mask, pix, sizes, return_output=False, inputs_batch=None):
        if ids.shape[0] < 1:          # the engine's contract: lr_forward rejects B < 1 (engine.hip "batch exceeds max_batch")
            raise RuntimeError("lr_forward failed (code 1): lr_forward: batch exceeds max_batch")
        base = (ids.float() * mask.float()).sum(dim=1, keepdim=True) * 1e-3 + pix.flatten(1).sum(dim=1, keepdim=True)
        return torch.cat([base * (k + 1) for k in range(self.d)], dim=1), None


def _batches(n_batches, n, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_batches):
        def one():
            return {"input_ids": torch.randint(0, 50, (n, 1, 6), generator=g), "attention_mask": torch.ones(n, 1, 6, dtype=torch.long),
                    "pixel_values": torch.randn(n, 1, 2, 3, 4, 4, generator=g), "image_sizes": torch.full((n, 1, 2), 336)}
        out.append((one(), one(), None, None))
    return out


def _worker(rank, ws, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    args = types.SimpleNamespace(is_general_preference=True, value_head_dim=2, general_preference_tau=0.1)
    # 5 rows over 2 ranks: ragged shards; then a last partial batch of ONE row (n < world_size, drop_last=False): rank 1 scores
    # nothing but must still enter the all-gather
    res = score_pairwise(_FakeModel(2), args, _batches(3, 5, 0) + _batches(1, 1, 7))
    local = torch.arange(4, dtype=torch.float32).reshape(2, 2) + 10 * rank
    g = gather_rewards(local)
    q.put((rank, res["probs"], g.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_shard_and_gather_equals_single_process():
    import torch.multiprocessing as mp
    args = types.SimpleNamespace(is_general_preference=True, value_head_dim=2, general_preference_tau=0.1)
    single = score_pairwise(_FakeModel(2), args, _batches(3, 5, 0) + _batches(1, 1, 7))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, probs, g in got:
        assert probs == single["probs"]                       # bit-identical on every rank
        assert g == [[0.0, 1.0], [2.0, 3.0], [10.0, 11.0], [12.0, 13.0]]
    assert 0.0 <= single["proportion"] <= 1.0 and len(single["probs"]) == 16


def test_score_single_metrics():
    args = types.SimpleNamespace(is_general_preference=False, value_head_dim=1, general_preference_tau=0.1)
    bs = [(b[0], torch.tensor([1, 0, 1, 1, 0])) for b in _batches(2, 5, 1)]
    res = score_single(_FakeModel(1), args, bs, cls_based=True)
    assert len(res["rewards"]) == 10 and 0 <= res["accuracy"] <= 1 and 0 <= res["f1"] <= 1
    args.is_general_preference = True
    with pytest.raises(ValueError):
        score_single(_FakeModel(1), args, bs)


def test_reward_model_requires_weights_or_seed():
    with pytest.raises(ValueError):
        RewardModel(synth.tiny_config())
    assert RewardModel(synth.tiny_config(), synth_seed=1, mean_hidden_state=True).mean_hidden_state is True      # rw_model:398-406
    assert RewardModel(synth.tiny_config(), synth_seed=1, layer_id=1).layer_id == 1       # hidden_states[1] (rw_model:351-352)
    with pytest.raises(IndexError):
        RewardModel(synth.tiny_config(), synth_seed=1, layer_id=7)


def test_load_reward_adaptor_llava(tmp_path):
    """model_type='llava' (eval/reward_adaptor_loader.py:110-148): LlavaNext checkpoint layout, LoRA merge, heads."""
    from safetensors.torch import save_file
    cfg = synth.llava_tiny_config()
    pre, pm = os.path.join(str(tmp_path), "pre"), os.path.join(str(tmp_path), "pm")
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
    W = {k: torch.from_numpy(v) for k, v in synth.llava_make_weights(cfg, 9).items()}
    # store the base in the 5.x module-tree naming to exercise the key canonicaliser
    def new_name(k):
        k = k.replace("language_model.model.", "language_model.")
        k = k.replace("vision_tower.vision_model.", "vision_tower.")
        return "model." + k
    save_file({new_name(k): v.to(torch.bfloat16) for k, v in W.items() if k != "value_head.weight"}, os.path.join(pre, "model.safetensors"))
    torch.save({"base_model.model.value_head.weight": W["value_head.weight"]}, os.path.join(pm, "pytorch_model.bin"))
    yaml.safe_dump({"is_general_preference": False, "add_cross_attention": False, "value_head_dim": 1, "general_preference_tau": 0.1},
                   open(os.path.join(pm, "reward_config.yaml"), "w"))
    g = torch.Generator().manual_seed(2)
    mod = "language_model.model.layers.1.self_attn.k_proj"
    A, Bm = torch.randn(4, cfg.hidden, generator=g) * 0.1, torch.randn(cfg.kv_heads * cfg.head_dim, 4, generator=g) * 0.1
    torch.save({f"base_model.model.{mod}.lora_A.weight": A, f"base_model.model.{mod}.lora_B.weight": Bm}, os.path.join(pm, "lora", "adapter_model.bin"))
    json.dump({"r": 4, "lora_alpha": 8}, open(os.path.join(pm, "lora", "adapter_config.json"), "w"))
    args = types.SimpleNamespace(pm_path=pm, pretrain=pre, cache_dir=None, ft_projector=False, disable_fast_tokenizer=False)
    args, model = load_reward_adaptor(args, "llava", os.path.join(pm, "reward_config.yaml"))
    assert model.model_type == "llava" and model.config.kv_heads == cfg.kv_heads and model.config.clip.layers_used == c.layers_used
    assert model.config.lora_rank == 4 and set(n for n, *_ in synth.llava_weight_specs(model.config)) == set(model._weights)
    assert torch.equal(model._weights[mod + ".lora_A.weight"], A) and torch.equal(model._weights[mod + ".lora_B.weight"], 2.0 * Bm)
    assert torch.equal(model._weights[mod + ".weight"].float(), W[mod + ".weight"])
    args.merge_lora = True
    merged = load_reward_adaptor(args, "llava", os.path.join(pm, "reward_config.yaml"))[1]
    assert merged.config.lora_rank == 0 and torch.allclose(merged._weights[mod + ".weight"], W[mod + ".weight"] + 2.0 * Bm @ A, atol=1e-6)
    yaml.safe_dump({"is_general_preference": False, "add_cross_attention": True, "value_head_dim": 1, "general_preference_tau": 0.1},
                   open(os.path.join(pm, "reward_config.yaml"), "w"))
    with pytest.raises(AttributeError):          # the reference dies the same way (rw_model:315)
        load_reward_adaptor(args, "llava", os.path.join(pm, "reward_config.yaml"))


def test_load_reward_adaptor_qwen(tmp_path):
    """model_type='qwen' (eval/reward_adaptor_loader.py:64-109): Qwen2.5-VL checkpoint layout (4.50-era flat config and
    the 5.x module-tree tensor names), LoRA merge, reward heads incl. SkipCA, ft_projector -> visual.merger."""
    from safetensors.torch import save_file
    cfg = synth.qwen_tiny_config()
    v = cfg.vision
    pre, pm = os.path.join(str(tmp_path), "pre"), os.path.join(str(tmp_path), "pm")
    os.makedirs(pre); os.makedirs(os.path.join(pm, "lora"))
    json.dump({"vocab_size": cfg.vocab_size, "hidden_size": cfg.hidden, "intermediate_size": cfg.intermediate,
               "num_hidden_layers": cfg.layers, "num_attention_heads": cfg.heads, "num_key_value_heads": cfg.kv_heads,
               "rms_norm_eps": cfg.rms_eps, "rope_theta": cfg.rope_theta, "hidden_act": "silu", "use_sliding_window": False,
               "rope_scaling": {"type": "mrope", "mrope_section": list(cfg.mrope_section)}, "image_token_id": cfg.image_token_id,
               "vision_config": {"depth": v.depth, "hidden_size": v.hidden, "num_heads": v.heads, "intermediate_size": v.intermediate,
                                 "patch_size": 14, "temporal_patch_size": 2, "spatial_merge_size": 2, "window_size": 112,
                                 "fullatt_block_indexes": list(v.fullatt), "out_hidden_size": cfg.hidden, "hidden_act": "silu"}},
              open(os.path.join(pre, "config.json"), "w"))
    W = {k: torch.from_numpy(a) for k, a in synth.qwen_make_weights(cfg, 9).items()}
    heads = ("value_head", "W_q", "W_k", "W_v", "ca_layernorm")

    def new_name(k):                                 # 5.x module tree: model.visual.*, model.language_model.*
        return "model." + k if k.startswith("visual.") else k.replace("model.", "model.language_model.", 1)
    save_file({new_name(k): t.to(torch.bfloat16) for k, t in W.items() if k.split(".")[0] not in heads},
              os.path.join(pre, "model.safetensors"))
    ft = {"base_model.model.visual.merger.ln_q.weight": W["visual.merger.ln_q.weight"] + 1.0,
          "base_model.model.visual.merger.mlp.0.weight": W["visual.merger.mlp.0.weight"], "base_model.model.visual.merger.mlp.0.bias": W["visual.merger.mlp.0.bias"],
          "base_model.model.visual.merger.mlp.2.weight": W["visual.merger.mlp.2.weight"], "base_model.model.visual.merger.mlp.2.bias": W["visual.merger.mlp.2.bias"]}
    sd = {f"base_model.model.{k}": t for k, t in W.items() if k.split(".")[0] in heads}
    sd.update(ft)
    torch.save(sd, os.path.join(pm, "pytorch_model.bin"))
    yaml.safe_dump({"is_general_preference": True, "add_cross_attention": True, "value_head_dim": 2, "general_preference_tau": 0.1},
                   open(os.path.join(pm, "reward_config.yaml"), "w"))
    cfg2 = synth.qwen_tiny_config(is_general_preference=True, value_head_dim=2)
    sd["base_model.model.value_head.weight"] = torch.from_numpy(synth.qwen_make_weights(cfg2, 9)["value_head.weight"])
    torch.save(sd, os.path.join(pm, "pytorch_model.bin"))
    g = torch.Generator().manual_seed(2)
    mod = "model.layers.1.self_attn.v_proj"
    A, Bm = torch.randn(4, cfg.hidden, generator=g) * 0.1, torch.randn(cfg.kv_heads * cfg.head_dim, 4, generator=g) * 0.1
    torch.save({f"base_model.model.{mod}.lora_A.weight": A, f"base_model.model.{mod}.lora_B.weight": Bm}, os.path.join(pm, "lora", "adapter_model.bin"))
    json.dump({"r": 4, "lora_alpha": 8}, open(os.path.join(pm, "lora", "adapter_config.json"), "w"))
    args = types.SimpleNamespace(pm_path=pm, pretrain=pre, cache_dir=None, ft_projector=True, disable_fast_tokenizer=False)
    args, model = load_reward_adaptor(args, "qwen", os.path.join(pm, "reward_config.yaml"))
    assert model.model_type == "qwen" and args.add_cross_attention is True and args.value_head_dim == 2
    assert model.config.vision.fullatt == v.fullatt and model.config.mrope_section == cfg.mrope_section and model.config.ca_eps == 1e-6
    assert model.config.lora_rank == 4 and set(n for n, *_ in synth.qwen_weight_specs(model.config)) == set(model._weights)
    assert torch.equal(model._weights[mod + ".lora_A.weight"], A) and torch.equal(model._weights[mod + ".lora_B.weight"], 2.0 * Bm)
    args.operand_dtype = "fp8"                       # W8A8 runs merged weights only
    merged = load_reward_adaptor(args, "qwen", os.path.join(pm, "reward_config.yaml"))[1]
    assert merged.config.lora_rank == 0 and torch.allclose(merged._weights[mod + ".weight"], W[mod + ".weight"] + 2.0 * Bm @ A, atol=1e-6)
    assert torch.equal(model._weights["visual.merger.ln_q.weight"], W["visual.merger.ln_q.weight"] + 1.0)      # ft_projector wins
    with pytest.raises(RuntimeError):                # no GPU here: the product path refuses, it does not fall back
        model.custom_forward(inputs_batch={})


def test_shard_qwen_batch_rows_and_images():
    """Row sharding of a Qwen2.5-VL BatchFeature (eval/batch_inference_rm_qwen.py:76-80): pixel_values / image_grid_thw are
    concatenated over the batch, rows may carry different grids (and, in general, several images)."""
    from llava_reward_amd.scoring import shard_qwen_batch, shard_rows
    cfg = synth.qwen_tiny_config()
    b = synth.qwen_synth_batch(cfg, 3, [4, 2, 6, 3, 5], [(8, 8), (4, 12), (6, 4), (8, 8), (10, 6)])
    tb = {k: torch.from_numpy(v) for k, v in b.items()}
    pieces = [shard_qwen_batch(tb, shard_rows(5, r, 3), cfg.image_token_id, cfg.vision.merge_unit) for r in range(3)]
    assert [p["input_ids"].shape[0] for p in pieces] == [2, 2, 1]
    assert torch.equal(torch.cat([p["pixel_values"] for p in pieces]), tb["pixel_values"])
    assert torch.equal(torch.cat([p["image_grid_thw"] for p in pieces]), tb["image_grid_thw"])
    for p in pieces:
        assert int(p["image_grid_thw"].prod(dim=1).sum()) == p["pixel_values"].shape[0]
        assert int((p["input_ids"] == cfg.image_token_id).sum()) * cfg.vision.merge_unit == p["pixel_values"].shape[0]
    bad = dict(tb)
    bad["image_grid_thw"] = tb["image_grid_thw"][:-1]
    with pytest.raises(ValueError, match="do not match"):
        shard_qwen_batch(bad, slice(0, 5), cfg.image_token_id, cfg.vision.merge_unit)


def test_zero_pad_sequences_and_collate_rows_follow_the_reference_rules():
    """datasets/utils.py:5-13 + reward_dataset.py:164-179: left padding with the pad id / 0, pixel tensors stacked; host tensors here,
    the same code runs on device tensors (tests/test_gpu_trainer_shim.py)."""
    import torch.nn.functional as F
    from llava_reward_amd import collate_rows, zero_pad_sequences
    seqs = [torch.arange(1, 4)[None], torch.arange(1, 7)[None], torch.arange(5, 6)[None]]
    for side in ("left", "right"):
        got = zero_pad_sequences(seqs, side, 9)
        exp = torch.stack([F.pad(q, (6 - q.size(-1), 0) if side == "left" else (0, 6 - q.size(-1)), value=9) for q in seqs])
        assert torch.equal(got, exp) and got.shape == (3, 1, 6)
    rows = [{"input_ids": q, "attention_mask": torch.ones_like(q), "pixel_values": torch.full((1, 2, 3, 4, 4), float(i)),
             "image_sizes": torch.tensor([[336, 336 * (i + 1)]])} for i, q in enumerate(seqs)]
    b = collate_rows(rows, pad_token_id=7)
    assert b["input_ids"].tolist() == [[7, 7, 7, 1, 2, 3], [1, 2, 3, 4, 5, 6], [7, 7, 7, 7, 7, 5]]
    assert b["attention_mask"].tolist() == [[0, 0, 0, 1, 1, 1], [1] * 6, [0, 0, 0, 0, 0, 1]]
    assert b["pixel_values"].shape == (3, 2, 3, 4, 4) and b["pixel_values"][2].eq(2.0).all() and b["image_sizes"].tolist() == [[336, 336], [336, 672], [336, 1008]]
    raw = collate_rows(rows, pad_token_id=7, squeeze=False)          # the reference's own layout: singleton dim kept (:82-90 squeeze it)
    assert raw["input_ids"].shape == (3, 1, 6) and raw["pixel_values"].shape == (3, 1, 2, 3, 4, 4)


def test_bench_final_line_is_compact_and_last(tmp_path, capsys):
    """The driver keeps an 8 KB tail of bench.py's stdout and parses its last line: round 4's single 22 KB line came back `parsed: null`.
    bench.emit() on the FULL result object of that very run (profiles/r4_bench.json, every leg present) must end with one '{' line of
    <= 4 KB that carries the metric, `roofline` and `cpu_baseline`; the legs go to '#leg' lines in front of it and to bench_legs.json."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    res = json.load(open(os.path.join(ROOT, "profiles", "r4_bench.json")))
    assert len(json.dumps(res)) > 20000                      # the object that broke the driver
    res["multi_gpu"] = {"ranks_seen": 8, "backend": "nccl", "devices": list(range(8)), "distinct_devices": 8,
                        "per_rank_ms": [1300.123] * 8, "collective_us": 55.5, "gathered_rows": 256}
    bench.emit(res, legs_dir=str(tmp_path))
    out = capsys.readouterr().out.splitlines()
    assert out[-1].startswith("{") and all(not l.startswith("{") for l in out[:-1])
    assert len(out[-1]) <= bench.LINE_BUDGET <= 4096
    line = json.loads(out[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == pytest.approx(res["value"], rel=1e-4) and line["vs_baseline"] is None
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-2)
    assert rf["kernel_ms"] > 0 and "traffic" in rf and rf["whole_pass_frac"] > 0
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "reward-pairs/sec"
    assert line["config"]["workload"].startswith("BASELINE configs[1]") and "model" not in line["config"]
    assert line["multi_gpu"]["ranks_seen"] == 8
    assert line["legs"]["llava"]["value"] > 0 and "qwen.lora_unmerged" in line["legs"]
    full = json.load(open(tmp_path / "bench_legs.json"))
    assert full["llava"]["parity_check"]["per_golden"]        # nothing was lost: the per-golden detail lives in the file
    # a result with many more legs still fits: the optional parts are dropped, never the headline
    for i in range(200):
        res[f"extra_leg_{i}"] = {"value": 1.0, "ms_per_step": 2.0, "operand_form": {"form": "strict-vision+decoder/2"}}
    assert len(bench.compact_line(res)) <= bench.LINE_BUDGET


def test_form_ladder_and_loader_knobs_without_a_gpu():
    """Host side of the operand-form machinery (round 5): the probe's candidate ladder (cheapest first, strict from the front of the
    model in eighths of the decoder), the pin-only single-pass tails, the constructor's validation, and the loader's pass-through of the
    knobs the reference does not have (args.operand_form / check_inputs / parity_budget / reward_dtype)."""
    from llava_reward_amd.reward_adaptor_loader import _form_args
    m = RewardModel(synth.full_config(), synth_seed=1)
    assert [n for n, _ in m._form_candidates()] == ["default", "strict-vision", "strict-vision+decoder/8", "strict-vision+decoder/4",
                                                    "strict-vision+decoder*3/8", "strict-vision+decoder/2", "strict"]
    maps = dict(m._form_candidates() + m._pinnable_forms())
    assert maps["default"] == (-1, -1, 0, 0) and maps["strict"] == (1, 1, 0, 0) and maps["strict-vision"] == (1, -1, 0, 0)
    assert maps["strict-vision+decoder/8"] == (1, 1, 0, 28) and maps["strict-vision+decoder*3/8"] == (1, 1, 0, 20)      # first 4 / 12 of 32 layers
    assert maps["default+single-tail/8"] == (-1, 0, 28, 0) and maps["default+single-tail/4"] == (-1, 0, 24, 0)          # last 4 / 8 single-pass
    tiny = RewardModel(synth.tiny_config(layers=3), synth_seed=1)
    names = [n for n, _ in tiny._form_candidates()]
    assert names[0] == "default" and names[-1] == "strict" and len(set(names)) == len(names) and tiny._pinnable_forms() == []
    with pytest.raises(ValueError):
        RewardModel(synth.tiny_config(), synth_seed=1, operand_form="default+single-tail/8")        # (2 layers: no such tail)
    with pytest.raises(ValueError):
        RewardModel(synth.tiny_config(), synth_seed=1, check_inputs="sometimes")
    a = types.SimpleNamespace(operand_form="strict", check_inputs="deferred", parity_budget=1e-4, reward_dtype="bf16")
    kw = _form_args(a)
    assert kw == dict(operand_form="strict", check_inputs="deferred", parity_budget=1e-4, reward_dtype=torch.bfloat16, vision_layer_id=-1)
    assert _form_args(types.SimpleNamespace()) == dict(operand_form=None, check_inputs="eager", parity_budget=1.5e-4, reward_dtype=None, vision_layer_id=-1)
    # rw_model_general_preference.py:296,353: only the reference's default SkipCA key / value source is served; others are refused, not ignored
    with pytest.raises(NotImplementedError, match="vision_layer_id"):
        RewardModel(synth.tiny_config(), synth_seed=1, **_form_args(types.SimpleNamespace(vision_layer_id=2)))
    mm = RewardModel(synth.tiny_config(), synth_seed=1, **kw)
    assert mm.pinned_form == "strict" and mm.check_inputs == "deferred" and mm.reward_dtype == torch.bfloat16
