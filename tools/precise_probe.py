#!/usr/bin/env python3
"""Reward error of the split-operand ("f16x2") mode vs the single-pass f16 mode against the CPU oracle (tiny configs) and the
reference's own goldens (small + full-size rows of the three backbones)."""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel

GOLD = os.path.join(ROOT, "tests", "golden")
which = sys.argv[1:] or ["small", "llava_tiny", "qwen", "full"]


def run(cfg, seed, batch, ref, name, qwen=False, **kw):
    out = []
    for dt in ("f16", "f16x2"):
        m = RewardModel(cfg, synth_seed=seed, operand_dtype=dt, **kw).to("cuda").eval()
        tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
        if cfg.__class__.__name__ == "RewardConfig":
            r, _ = m.custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"])
        else:
            r, _ = m.custom_forward(inputs_batch=tb)
        torch.cuda.synchronize()
        out.append((r.cpu().reshape(ref.shape) - ref).abs().max().item())
        del m
        torch.cuda.empty_cache()
    print(f"{name:28s} |r|max {ref.abs().max().item():5.2f}   f16 err {out[0]:.2e}   f16x2 err {out[1]:.2e}", flush=True)


for path in sorted(glob.glob(os.path.join(GOLD, "ref_*.json"))):
    g = json.load(open(path))
    name = g["name"]
    full = "_full_" in name
    if full and "full" not in which:
        continue
    bb = g.get("backbone", "phi3v")
    ref = torch.tensor(g["reward"], dtype=torch.float32)
    if bb == "qwen":
        if not full and "qwen" not in which: continue
        cfg = synth.QwenConfig.from_json(g["config"])
        batch = synth.qwen_synth_batch(cfg, g["seed"], g["caption_lens"], [tuple(x) for x in g["grids"]])
        S = batch["input_ids"].shape[1]
        run(cfg, g["seed"], batch, ref, name, max_batch=len(g["caption_lens"]), max_seq=S, max_patches=int(batch["pixel_values"].shape[0]))
    elif bb == "llava":
        if not full and "llava_tiny" not in which: continue
        cfg = synth.LlavaConfig.from_json(g["config"])
        batch = synth.llava_synth_batch(cfg, g["seed"], g["caption_lens"], [tuple(x) for x in g["image_sizes"]], max_crops=g["max_crops"])
        run(cfg, g["seed"], batch, ref, name, max_batch=len(g["caption_lens"]), max_seq=batch["input_ids"].shape[1], max_crops=5)
    else:
        if not full and "small" not in which: continue
        cfg = synth.RewardConfig.from_json(g["config"])
        grids = g["grids"]
        grids = tuple(grids) if isinstance(grids[0], int) else [tuple(x) for x in grids]
        batch = synth.synth_batch(cfg, g["seed"], g["caption_lens"], grids, max_crops=g["max_crops"])
        run(cfg, g["seed"], batch, ref, name, max_batch=len(g["caption_lens"]), max_seq=batch["input_ids"].shape[1], max_crops=17 if full else 5)
