#!/usr/bin/env python3
"""Full-size fixture for the W8A8 mode (operand_dtype="fp8", BASELINE configs[4] "fp8 MFMA weight path") of LLaVA-v1.6-Mistral-7B.

The reference has no fp8 path, so there is nothing of it to run here: this script runs the ORACLE (oracle/llava_next_reward_oracle.py,
itself pinned to the reference by tests/golden/ref_llava_*.json) with the engine's operand quantisation emulated (W8A8Round: f16
storage, every GEMM operand row to OCP e4m3 with one scale per row / output channel) on the row of ref_llava_full_e4m3_bt.json
(e4m3-VALUED weights: the reference's fp32 reward on the same weights is in that golden).  An e4m3 model is chaotic in its inputs,
so the fixture carries a TWIN emulation as well (the same quantisation without the f16 storage rounding): the distance between the
two is the bound the engine is held to (tests/test_gpu_llava.py::test_w8a8_full_size_batch64).  ~15 min, ~40 GB.

    python tests/golden/make_w8a8_emulation.py
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "llava-reward_amd"), ROOT]
from llava_reward_amd import synth  # noqa: E402
from oracle import llava_next_reward_oracle as lorc  # noqa: E402
from oracle import phi3v_reward_oracle as orc  # noqa: E402


def main():
    g = json.load(open(os.path.join(HERE, "ref_llava_full_e4m3_bt.json")))
    cfg = synth.LlavaConfig.from_json(g["config"])
    W = orc.weights_to_torch(synth.llava_make_weights(cfg, g["seed"], g["weight_profile"]))
    b = synth.llava_synth_batch(cfg, g["seed"], g["caption_lens"], [tuple(x) for x in g["image_sizes"]], max_crops=g["max_crops"])
    out = {"name": "ref_llava_full_w8a8_emulation", "source": "oracle/llava_next_reward_oracle.py with W8A8Round operand quantisation (NOT the reference: "
           "it has no fp8 path); row and weights of ref_llava_full_e4m3_bt.json", "row_golden": "ref_llava_full_e4m3_bt", "torch": torch.__version__}
    for key, opr in (("fp32", orc.Ident), ("w8a8", orc.W8A8Round(orc.f16_round)), ("w8a8_twin", orc.W8A8Round(orc.Ident))):
        t0 = time.time()
        r = lorc.custom_forward(W, cfg, b["input_ids"], b["attention_mask"], b["pixel_values"], b["image_sizes"], opr=opr)
        out[key] = r.flatten().tolist()
        print(f"[{key}] {out[key]} ({time.time() - t0:.0f}s)", flush=True)
    out["oracle_vs_reference_fp32"] = abs(out["fp32"][0] - g["reward"][0][0])
    assert out["oracle_vs_reference_fp32"] < 2e-5, out["oracle_vs_reference_fp32"]          # the oracle is pinned on this row too
    json.dump(out, open(os.path.join(HERE, "w8a8_llava_full_emulation.json"), "w"), indent=1)


def clip_stage():
    """Adds the stage-level fingerprint tests/test_gpu_llava.py::test_w8a8_full_size_batch64 holds the engine to (round 4): the CLIP tower's
    output for the golden row's crops (little has been amplified there yet), fp32 and with the W8A8 operand quantisation, as
    256 sampled elements each.  Only the tower runs (seconds): `python tests/golden/make_w8a8_emulation.py clip` merges into the fixture."""
    path = os.path.join(HERE, "w8a8_llava_full_emulation.json")
    out = json.load(open(path))
    g = json.load(open(os.path.join(HERE, "ref_llava_full_e4m3_bt.json")))
    cfg = synth.LlavaConfig.from_json(g["config"])
    names = [n for n, *_ in synth.llava_weight_specs(cfg) if n.startswith(lorc.CLIP_PREFIX)]
    specs = {n: (sh, std, off) for n, sh, std, off in synth.llava_weight_specs(cfg)}
    W = orc.weights_to_torch({n: synth.gen_tensor(g["seed"], n, specs[n][0], specs[n][1], specs[n][2], profile=g["weight_profile"]) for n in names})
    b = synth.llava_synth_batch(cfg, g["seed"], g["caption_lens"], [tuple(x) for x in g["image_sizes"]], max_crops=g["max_crops"])
    geo = synth.llava_geometry(*[int(v) for v in b["image_sizes"][0]], cfg.pinpoints, cfg.clip.image, cfg.clip.grid)
    crops = torch.from_numpy(b["pixel_values"][0, : 1 + geo[0] * geo[1]])
    gen = torch.Generator().manual_seed(99)
    for key, opr in (("fp32", orc.Ident), ("w8a8", orc.W8A8Round(orc.f16_round)), ("w8a8_twin", orc.W8A8Round(orc.Ident))):
        f = orc.clip_tower(W, crops, cfg.clip, opr, prefix=lorc.CLIP_PREFIX).reshape(-1)
        if key == "fp32":
            idx = torch.randperm(f.numel(), generator=gen)[:256].sort().values
            out["clip_out_shape"] = [crops.shape[0], cfg.clip.tokens - 1, cfg.clip.hidden]
            out["clip_out_idx"] = idx.tolist()
        out["clip_out_" + key] = f[idx].tolist()
    d = (torch.tensor(out["clip_out_fp32"]) - torch.tensor(out["clip_out_w8a8"])).abs().max().item()
    dt = (torch.tensor(out["clip_out_w8a8_twin"]) - torch.tensor(out["clip_out_w8a8"])).abs().max().item()
    print(f"CLIP tower, {crops.shape[0]} crops: over the 256 samples max |fp32 - w8a8 emulation| = {d:.3e}, max |twin emulation - w8a8 emulation| = {dt:.3e}")
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    clip_stage() if len(sys.argv) > 1 and sys.argv[1] == "clip" else main()
