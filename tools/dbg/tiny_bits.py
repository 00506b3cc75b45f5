#!/usr/bin/env python3
"""The tiny Phi config of tests/test_gpu_forward.py through the library named by LLAVA_REWARD_HIP_LIB: rewards and per-layer hidden-state
checksums in every operand mode, one line each -- run under two builds and diff the output to find the first kernel whose bits moved."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
sys.path.insert(0, ROOT)
import torch
from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel

for variant, kw in (("bt_ca", {}), ("bt_noca", dict(add_cross_attention=False))):
    cfg = synth.tiny_config(**kw)
    batch = synth.synth_batch(cfg, 11, [7, 3, 5], [(1, 1), (1, 2), (2, 1)], max_crops=4)
    for dtype in ("f16", "bf16", "f16x2", "f16x2f8"):
        m = RewardModel(cfg, synth_seed=11, max_batch=4, max_seq=1024, max_crops=5, operand_dtype=dtype, layer_id=32, calibrate=False)
        m.keep_hidden_states = True
        m = m.to("cuda").eval()
        tb = {k: torch.from_numpy(v) for k, v in batch.items()}
        r, out = m.custom_forward(tb["input_ids"].cuda(), tb["attention_mask"].cuda(), tb["pixel_values"].cuda(), tb["image_sizes"].cuda(),
                                  output_hidden_states=True) if False else m.custom_forward(tb["input_ids"].cuda(), tb["attention_mask"].cuda(), tb["pixel_values"].cuda(), tb["image_sizes"].cuda())
        torch.cuda.synchronize()
        print(variant, dtype, "rewards", [float.hex(float(x)) for x in r.flatten().cpu()])
        hs = out["hidden_states"] if isinstance(out, dict) and "hidden_states" in out else None
        if hs is not None:
            for i in range(len(hs)):
                t = hs[i]
                print(variant, dtype, "hidden", i, float.hex(float(t.double().abs().sum().cpu())))
