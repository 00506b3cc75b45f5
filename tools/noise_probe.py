#!/usr/bin/env python3
"""Reward of the full-size golden row under numerically different but equivalent kernel variants."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import numpy as np, torch
from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel

for name in sys.argv[1:] or ["ref_full_bt_ca"]:
    g = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
    cfg = synth.RewardConfig.from_json(g["config"])
    grids = tuple(g["grids"]) if isinstance(g["grids"][0], int) else [tuple(x) for x in g["grids"]]
    batch = synth.synth_batch(cfg, g["seed"], g["caption_lens"], grids, max_crops=g["max_crops"])
    tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    ref = np.array(g["reward"]).reshape(-1)
    for dt in ("f16", "bf16"):
        m = RewardModel(cfg, synth_seed=g["seed"], max_batch=1, max_seq=batch["input_ids"].shape[1], max_crops=17, operand_dtype=dt).to("cuda").eval()
        errs = []
        for tile in (0, 2, 5, 6):
            m.engine.set_gemm_tile(tile)
            r, _ = m.custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"])
            e = r.cpu().numpy().reshape(-1) - ref
            errs.append(e)
            print(f"{name} {dt} tile={tile}: reward={r.flatten().tolist()} err={e.tolist()}", flush=True)
        errs = np.array(errs)
        print(f"  -> {dt}: mean err {errs.mean(0)}, std {errs.std(0)}, max |err| {np.abs(errs).max():.2e}")
        del m
        torch.cuda.empty_cache()
