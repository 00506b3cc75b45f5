#!/usr/bin/env python3
"""Reward of the full-size Qwen2.5-VL golden row under numerically equivalent kernel variants (GEMM tile shapes change
the summation order only) and under bf16 operands: separates rounding noise from a systematic error."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel

g = json.load(open(os.path.join(ROOT, "tests/golden/ref_qwen_full_bt.json")))
cfg = synth.QwenConfig.from_json(g["config"])
batch = synth.qwen_synth_batch(cfg, g["seed"], g["caption_lens"], [tuple(x) for x in g["grids"]])
tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
ref = g["reward"][0][0]
S = batch["input_ids"].shape[1]
for dt in ("f16", "bf16"):
    m = RewardModel(cfg, synth_seed=g["seed"], max_batch=1, max_seq=S, max_patches=1024, operand_dtype=dt).to("cuda").eval()
    for tile in (-1, 0, 1, 2, 6):
        m.engine.set_gemm_tile(tile)
        r, _ = m.custom_forward(inputs_batch=tb)
        torch.cuda.synchronize()
        print(f"{dt} tile {tile:2d}: reward {r.item():.6f}  err {r.item() - ref:+.2e}", flush=True)
    if len(sys.argv) > 1:       # layer-limit sweep: where does the deviation appear?  (compare with an oracle run offline)
        for nv, nl in ((0, 0), (8, 0), (32, 0), (32, 4), (32, 14)):
            m.engine.set_gemm_tile(-1); m.engine.set_layer_limits(nv, nl)
            r, _ = m.custom_forward(inputs_batch=tb); torch.cuda.synchronize()
            print(f"{dt} vit {nv} dec {nl}: reward {r.item():.6f}")
    del m
    torch.cuda.empty_cache()
