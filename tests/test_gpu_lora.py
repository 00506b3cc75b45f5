"""Un-merged LoRA adapters on the HIP path (lr_model_desc.lora_rank) against the oracle evaluated UN-MERGED, the way the reference
runs a checkpoint: eval/reward_adaptor_loader.py:44-45 loads the adapter with model.load_adapter and peft (0.13.2,
tuners/lora/layer.py Linear.forward) computes y = W x + (lora_alpha / r) B (A x) on the targets of
llava_reward/utils/utils.py:194-262.  The engine computes t = x A^T with one small GEMM per linear and lets t B^T ride in the K loop
of the base GEMM, so base weights stay bf16-exact.

Tolerances as everywhere: strict parity mode f16x2 1e-4, default mode f16x2f8 3e-4, single-pass f16 1e-3."""
import json
import os
import types

import numpy as np
import pytest
import torch
import yaml

from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel
from llava_reward_amd.reward_adaptor_loader import load_reward_adaptor
from oracle import llava_next_reward_oracle as lorc
from oracle import phi3v_reward_oracle as orc
from oracle import qwen2_5_vl_reward_oracle as qorc

pytestmark = pytest.mark.gpu

TOL = {"f16x2": 1e-4, "f16x2f8": 3e-4, "f16": 1e-3}


def _case(backbone, rank, seed):
    if backbone == "phi3v":
        cfg = synth.tiny_config(lora_rank=rank, is_general_preference=True, value_head_dim=2)
        Wn = synth.make_weights(cfg, seed)
        batch = synth.synth_batch(cfg, seed, [7, 3, 5], [(1, 1), (1, 2), (2, 1)], max_crops=4)
        fwd = lambda W: orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
    elif backbone == "llava":
        cfg = synth.llava_tiny_config(lora_rank=rank)
        Wn = synth.llava_make_weights(cfg, seed)
        batch = synth.llava_synth_batch(cfg, seed, [6, 3], [(336, 336), (300, 500)])
        fwd = lambda W: lorc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
    else:
        cfg = synth.qwen_tiny_config(lora_rank=rank)
        Wn = synth.qwen_make_weights(cfg, seed)
        batch = synth.qwen_synth_batch(cfg, seed, [7, 3, 5], [(16, 16), (10, 6), (18, 22)])
        fwd = lambda W: qorc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_grid_thw"])
    return cfg, Wn, batch, fwd


def _hip(cfg, batch, dtype, weights=None, seed=None):
    m = RewardModel(cfg, weights=weights, synth_seed=seed, max_batch=4, max_seq=4096 if isinstance(cfg, synth.LlavaConfig) else 1024,
                    max_crops=5, max_patches=4096, operand_dtype=dtype).to("cuda").eval()
    tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    if m.model_type == "phi3v":
        r, _ = m.custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"])
    else:
        r, _ = m.custom_forward(inputs_batch=tb)
    torch.cuda.synchronize()
    return r.cpu(), m


@pytest.mark.parametrize("dtype", ["f16x2", "f16x2f8", "f16"])
@pytest.mark.parametrize("backbone,rank", [("phi3v", 16), ("phi3v", 128), ("llava", 64), ("qwen", 8)])
def test_unmerged_adapter_vs_oracle(backbone, rank, dtype):
    """Synthetic adapters of every decoder linear (ranks below, at and above one K-tile: padded to 64 columns), bf16-valued like a
    bf16-trained checkpoint.  The adapter moves the reward by far more than the tolerance (asserted), so a dropped or mis-wired
    adapter cannot pass; device-side synthetic weights equal the uploaded ones bit for bit."""
    seed = 31
    cfg, Wn, batch, fwd = _case(backbone, rank, seed)
    W = orc.weights_to_torch(Wn)
    assert any(k.endswith("lora_A.weight") for k in W)
    ref = fwd(W)
    base = fwd({k: v for k, v in W.items() if ".lora_" not in k})
    got, m = _hip(cfg, batch, dtype, seed=seed)
    err = (got - ref).abs().max().item()
    moved = (ref - base).abs().max().item()
    print(f"[lora {backbone} r={rank} {dtype}] max |reward err| = {err:.2e}; adapter moves the reward by {moved:.2e}")
    assert moved > 30 * TOL[dtype] or moved > 0.05
    assert err < TOL[dtype]
    if dtype == "f16x2":
        up, _ = _hip(cfg, batch, dtype, weights={k: torch.from_numpy(v) for k, v in Wn.items()})
        assert torch.equal(up, got)
        # a row's reward does not depend on its batch (the t GEMM and the K-extension keep the fixed reduction order).  Row 2 carries
        # the batch's V_max image tokens; rows with fewer see V_max through the un-masked zero-padded SkipCA rows (rw_model:381-385),
        # in the reference as here, so only their V_max-preserving regroupings are bit-stable.
        tb = {k: torch.from_numpy(v) for k, v in batch.items()}
        if backbone == "phi3v":
            one, _ = m.custom_forward(tb["input_ids"][2:3].cuda(), tb["attention_mask"][2:3].cuda(), tb["pixel_values"][2:3].cuda(), tb["image_sizes"][2:3])
            assert torch.equal(one.cpu()[0], got[2])
            two, _ = m.custom_forward(tb["input_ids"][1:3].cuda(), tb["attention_mask"][1:3].cuda(), tb["pixel_values"][1:3].cuda(), tb["image_sizes"][1:3])
            assert torch.equal(two.cpu(), got[1:3])


@pytest.mark.parametrize("dtype", ["f16x2", "f16x2f8"])
@pytest.mark.parametrize("backbone", ["phi3v", "qwen"])
def test_unmerged_adapter_fp32_valued(backbone, dtype):
    """Adapters that are NOT exact in f16 (an fp32-saved adapter, or alpha / r that is not a power of two): A takes the 16-bit
    third segment x_hi x A_lo beside its e4m3 residual pass, B the segment t_hi x B_lo; parity holds, and the un-merged form
    agrees with the merged form of the same adapter run in the strict mode."""
    seed = 37
    cfg, Wn, batch, fwd = _case(backbone, 32, seed)
    g = torch.Generator().manual_seed(3)
    W = orc.weights_to_torch(Wn)
    for k in list(W):
        if ".lora_" in k:
            W[k] = (W[k] * (1.0 + 2.0 ** -9 * (torch.rand(W[k].shape, generator=g) - 0.5)) * 1.37).float()
    ref = fwd(W)
    got, _ = _hip(cfg, batch, dtype, weights=W)
    err = (got - ref).abs().max().item()
    print(f"[lora fp32-valued {backbone} {dtype}] max |reward err| = {err:.2e}")
    assert err < TOL[dtype]
    if dtype == "f16x2":
        import dataclasses
        merged = {k: v.clone() for k, v in W.items() if ".lora_" not in k}
        for k in W:
            if k.endswith(".lora_A.weight"):
                mod = k[: -len(".lora_A.weight")]
                merged[mod + ".weight"] = merged[mod + ".weight"] + W[mod + ".lora_B.weight"] @ W[k]
        gm, _ = _hip(dataclasses.replace(cfg, lora_rank=0), batch, dtype, weights=merged)
        assert (gm - got).abs().max().item() < 1e-4


def test_loader_to_engine_phi3v(tmp_path):
    """load_reward_adaptor on a fabricated checkpoint directory -- bf16 safetensors base, pytorch_model.bin heads + ft_projector,
    PEFT adapter with `base_model.model.` prefix and the `.default.` adapter-name infix in adapter_model.safetensors --
    -> .to('cuda') -> custom_forward, against the oracle on the same effective weights with the adapter evaluated un-merged
    (scaling = lora_alpha / r applied by the oracle, folded into B by the loader)."""
    from safetensors.torch import save_file
    from test_host_logic import _write_fake_checkpoint
    cfg = synth.tiny_config(is_general_preference=True, value_head_dim=2)
    reward_cfg = {"is_general_preference": True, "add_cross_attention": True, "value_head_dim": 2, "general_preference_tau": 0.1}
    pre, pm, W, lora, proj = _write_fake_checkpoint(str(tmp_path), cfg, 3, reward_cfg)
    os.remove(os.path.join(pm, "lora", "adapter_model.bin"))
    g = torch.Generator().manual_seed(8)
    r, alpha = 8, 12.0                                     # scaling 1.5: B * 1.5 is not bf16-valued
    sd, eff = {}, {}
    for l in range(cfg.layers):
        for mod, (n_out, n_in) in {"self_attn.qkv_proj": (3 * cfg.hidden, cfg.hidden), "self_attn.o_proj": (cfg.hidden, cfg.hidden),
                                   "mlp.gate_up_proj": (2 * cfg.intermediate, cfg.hidden), "mlp.down_proj": (cfg.hidden, cfg.intermediate)}.items():
            if l == 1 and mod == "self_attn.o_proj":
                continue                                   # one linear without an adapter: zero-filled slot
            A = (torch.randn(r, n_in, generator=g) * 0.05).to(torch.bfloat16)
            Bm = (torch.randn(n_out, r, generator=g) * 0.05).to(torch.bfloat16)
            name = f"model.layers.{l}.{mod}"
            sd[f"base_model.model.{name}.lora_A.default.weight"] = A
            sd[f"base_model.model.{name}.lora_B.default.weight"] = Bm
            eff[name + ".lora_A.weight"], eff[name + ".lora_B.weight"] = A.float(), Bm.float()
    save_file(sd, os.path.join(pm, "lora", "adapter_model.safetensors"))
    json.dump({"r": r, "lora_alpha": alpha, "target_modules": ["qkv_proj", "o_proj", "gate_up_proj", "down_proj"], "peft_type": "LORA"},
              open(os.path.join(pm, "lora", "adapter_config.json"), "w"))
    args = types.SimpleNamespace(pm_path=pm, pretrain=pre, cache_dir=None, ft_projector=True, disable_fast_tokenizer=False,
                                 max_batch=4, max_seq=1024, max_crops=5, operand_dtype="f16x2")
    args, model = load_reward_adaptor(args, "phi3v", os.path.join(pm, "reward_config.yaml"))
    assert args.lora_modules["unmerged"] == 4 * cfg.layers - 1 and model.config.lora_rank == r
    model.to("cuda")
    model.eval()
    batch = synth.synth_batch(cfg, 3, [7, 3, 5], [(1, 1), (1, 2), (2, 1)], max_crops=4)
    tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    got, _ = model.custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"])
    # the oracle's weights: what the checkpoint files hold (bf16 base, fp32 heads, the fine-tuned projector), adapter un-merged
    Wo = {k: (v.to(torch.bfloat16).float() if k.split(".")[0] not in ("value_head", "W_q", "W_k", "W_v", "ca_layernorm") else v.float())
          for k, v in W.items()}
    for k, v in proj.items():
        Wo[k[len("base_model.model."):]] = v.float()
    Wo.update(eff)
    Wo["lora_scaling"] = alpha / r
    ref = orc.custom_forward(Wo, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
    base = orc.custom_forward({k: v for k, v in Wo.items() if "lora" not in k}, cfg, batch["input_ids"], batch["attention_mask"],
                              batch["pixel_values"], batch["image_sizes"])
    err = (got.cpu() - ref).abs().max().item()
    print(f"[loader -> engine] max |reward err| = {err:.2e}; adapter moves the reward by {(ref - base).abs().max().item():.2e}")
    assert err < 1e-4 and (ref - base).abs().max().item() > 1e-2
    # the merge=True debug switch computes the same function
    args2 = types.SimpleNamespace(**{**vars(args), "merge_lora": True})
    args2, merged = load_reward_adaptor(args2, "phi3v", os.path.join(pm, "reward_config.yaml"))
    gm, _ = merged.to("cuda").eval().custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"])
    assert (gm.cpu() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("backbone,rank", [("phi3v", 128), ("qwen", 8)])
def test_unmerged_adapter_on_outlier_weights_locks_the_strict_form(backbone, rank):
    """Un-merged adapters on the outlier-bearing weight set.  The outlier adapters inject further massive channels (a 50-sigma element
    of B puts t straight into one output channel), so this model amplifies operand rounding beyond what the default form carries
    (5e-4 / 3e-3 measured with it pinned): the bare sequence -- .to('cuda') and nothing else -- finds that on its probe rows, locks the
    strict form (base GEMM, the adapter's t = x A^T GEMM and the K-extension all in 16-bit residual passes) and lands on the oracle
    (evaluated un-merged).  With the default form pinned (calibrate=False) the forward still runs, bit-stable, near the oracle."""
    seed = 41
    if backbone == "phi3v":
        cfg = synth.tiny_config(lora_rank=rank, hidden=1024, intermediate=2048, heads=16, layers=3)
        Wn = synth.make_weights(cfg, seed, synth.PROFILE_OUTLIER)
        batch = synth.synth_batch(cfg, seed, [7, 3, 5], (1, 1))
        ref = orc.custom_forward(orc.weights_to_torch(Wn), cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
    else:
        cfg = synth.qwen_tiny_config(lora_rank=rank, hidden=1024, intermediate=2048, heads=8, kv_heads=2)
        Wn = synth.qwen_make_weights(cfg, seed, synth.PROFILE_OUTLIER)
        batch = synth.qwen_synth_batch(cfg, seed, [7, 3, 5], [(16, 16), (10, 6), (18, 22)])
        ref = qorc.custom_forward(orc.weights_to_torch(Wn), cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_grid_thw"])
    W = {k: torch.from_numpy(v) for k, v in Wn.items()}
    tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    kw = tb if backbone == "phi3v" else {"inputs_batch": tb}
    pinned = RewardModel(cfg, weights=W, max_batch=4, max_seq=1024, max_crops=5, max_patches=4096, calibrate=False).to("cuda").eval()
    pinned.engine.set_gemm_tile(6)
    d = pinned.custom_forward(**kw)[0].cpu()
    assert pinned.operand_form == "default" and (d - ref).abs().max().item() < 5e-3 and torch.equal(pinned.custom_forward(**kw)[0].cpu(), d)
    m = RewardModel(cfg, weights=W, max_batch=4, max_seq=1024, max_crops=5, max_patches=4096).to("cuda").eval()
    got = m.custom_forward(**kw)[0].cpu()
    print(f"[lora + outlier weights, {backbone}] {m.form_info}; err pinned default form {(d - ref).abs().max().item():.2e}, "
          f"as locked {(got - ref).abs().max().item():.2e}")
    assert (got - ref).abs().max().item() < TOL["f16x2f8"]
    if m.operand_form == "strict":          # ... which is the f16x2 mode's arithmetic
        m2 = RewardModel(cfg, weights=W, max_batch=4, max_seq=1024, max_crops=5, max_patches=4096, operand_dtype="f16x2").to("cuda").eval()
        assert torch.equal(m2.custom_forward(**kw)[0].cpu(), got)


@pytest.mark.parametrize("layout", ["4.50", "5.x"])
@pytest.mark.parametrize("backbone", ["llava", "qwen"])
def test_loader_to_engine_llava_and_qwen(tmp_path, backbone, layout):
    """eval/reward_adaptor_loader.py:64-148 for the other two backbones, end to end on the GPU: a fabricated HF directory in the
    tensor-name layout of the pinned transformers 4.50 AND in the 5.x module tree (tests/helpers/fake_checkpoints.py: bf16 safetensors
    base, pytorch_model.bin heads, PEFT adapter with separate q / k / v / o / gate / up / down modules, `.default.` infix, scaling
    1.5, one module un-targeted) -> load_reward_adaptor -> .to('cuda') -> custom_forward, against the oracle on the same effective
    weights with the adapter evaluated UN-merged.  The fused linears of the engine (q|k|v, gate|up) stack the separate adapters'
    A matrices and hold B block-diagonally: a mis-routed part cannot pass (the adapter moves the reward far beyond the tolerance)."""
    from helpers import fake_checkpoints as fk
    seed = 43
    if backbone == "llava":
        cfg = synth.llava_tiny_config()
        pre, pm, Wo, n_mod = fk.write_llava(str(tmp_path), cfg, seed, layout)
        batch = synth.llava_synth_batch(cfg, seed, [6, 3], [(336, 336), (300, 500)])
        fwd = lambda W: lorc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
        extra = dict(max_seq=4096, max_crops=5)
    else:
        cfg = synth.qwen_tiny_config(is_general_preference=True, value_head_dim=2)
        pre, pm, Wo, n_mod = fk.write_qwen(str(tmp_path), cfg, seed, layout)
        batch = synth.qwen_synth_batch(cfg, seed, [7, 3, 5], [(16, 16), (10, 6), (18, 22)])
        fwd = lambda W: qorc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_grid_thw"])
        extra = dict(max_seq=1024, max_patches=4096)
    args = types.SimpleNamespace(pm_path=pm, pretrain=pre, cache_dir=None, ft_projector=False, disable_fast_tokenizer=False, max_batch=4, **extra)
    args, model = load_reward_adaptor(args, backbone, os.path.join(pm, "reward_config.yaml"))
    assert args.lora_modules == {"unmerged": n_mod, "merged": 0, "skipped": 0} and model.config.lora_rank == 8
    model.to("cuda")
    model.eval()
    got, _ = model.custom_forward(inputs_batch={k: torch.from_numpy(v).cuda() for k, v in batch.items()})
    ref = fwd(Wo)
    base = fwd({k: v for k, v in Wo.items() if "lora" not in k})
    err = (got.cpu() - ref).abs().max().item()
    moved = (ref - base).abs().max().item()
    print(f"[loader -> engine, {backbone}, {layout} layout] {model.form_info}; max |reward err| = {err:.2e}; adapter moves the reward by {moved:.2e}")
    assert moved > 1e-2 and err < (1e-4 if model.operand_form == "strict" else 3e-4)
    # the merge=True debug switch computes the same function (fp32-valued merged weights: the inexact-weight path of the default mode)
    args2 = types.SimpleNamespace(**{**vars(args), "merge_lora": True})
    args2, merged = load_reward_adaptor(args2, backbone, os.path.join(pm, "reward_config.yaml"))
    gm, _ = merged.to("cuda").eval().custom_forward(inputs_batch={k: torch.from_numpy(v).cuda() for k, v in batch.items()})
    print(f"[loader -> engine, {backbone}, {layout} layout] merged: {merged.form_info}; err {(gm.cpu() - ref).abs().max().item():.2e}")
    assert (gm.cpu() - ref).abs().max().item() < 3e-4
