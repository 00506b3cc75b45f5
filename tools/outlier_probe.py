#!/usr/bin/env python3
"""Default parity form on outlier-bearing weights, before / after lr_calibrate: |reward - reference| on the full-size outlier
golden(s) and the distance to the strict form over 8 more rows.   python tools/outlier_probe.py"""
import glob, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "llava-reward_amd"), ROOT]
import numpy as np, torch
from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel

for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_full_outlier_*.json"))):
    g = json.load(open(path))
    cfg = synth.RewardConfig.from_json(g["config"])
    grids = g["grids"]; grids = tuple(grids) if isinstance(grids[0], int) else [tuple(x) for x in grids]
    b = synth.synth_batch(cfg, g["seed"], g["caption_lens"], grids, max_crops=g["max_crops"])
    kw = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    ref = torch.tensor(g["reward"])
    m = RewardModel(cfg, synth_seed=g["seed"], max_batch=2, max_seq=b["input_ids"].shape[1], max_crops=17, synth_profile=g["weight_profile"]).to("cuda").eval()
    r0, _ = m.custom_forward(**kw)
    n = m.calibrate(kw)
    r1, _ = m.custom_forward(**kw)
    hot = {i: None for i in range(0)}
    print(f"[{g['name']}] default: err {float((r0.cpu() - ref).abs().max()):.2e}   calibrated ({n} operands with hot blocks): err {float((r1.cpu() - ref).abs().max()):.2e}")
    del m
    torch.cuda.empty_cache()

rows = 8
for profile in (2, 0):
    cfg = synth.full_config()
    b = synth.synth_batch(cfg, 77, [128, 64, 200, 17, 96, 128, 33, 150], (4, 4), with_pixels=False)
    ids, mask = torch.from_numpy(b["input_ids"]).cuda(), torch.from_numpy(b["attention_mask"]).cuda()
    pix = torch.randn(rows, 17, 3, 336, 336, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    m = RewardModel(cfg, synth_seed=77, max_batch=rows, max_seq=ids.shape[1], max_crops=17, synth_profile=profile).to("cuda").eval()
    def run():
        r = m.engine.forward(ids, mask, pix, b["image_sizes"]).clone(); torch.cuda.synchronize()
        t0 = time.perf_counter(); m.engine.forward(ids, mask, pix, b["image_sizes"]); torch.cuda.synchronize()
        return r.cpu().double(), (time.perf_counter() - t0) * 1e3
    m.engine.set_precision_map(1, 1, 0, 0); ref, _ = run(); m.engine.set_precision_map(-1, -1, 0, 0)
    r0, t0 = run()
    n = m.calibrate(dict(input_ids=ids[:4], attention_mask=mask[:4], pixel_values=pix[:4], image_sizes=torch.from_numpy(b["image_sizes"][:4])))
    r1, t1 = run()
    one = m.engine.forward(ids[5:6], mask[5:6], pix[5:6], b["image_sizes"][5:6]).cpu().double()
    d0, d1 = (r0 - ref).abs(), (r1 - ref).abs()
    print(f"profile {profile}: default rms {d0.pow(2).mean().sqrt():.2e} max {d0.max():.2e} ({t0:.1f} ms)   calibrated on rows 0-3, {n} operands hot: "
          f"rms {d1.pow(2).mean().sqrt():.2e} max {d1.max():.2e} ({t1:.1f} ms)   row 5 alone bit-identical: {bool(torch.equal(one[0], r1[5]))}")
    del m
    torch.cuda.empty_cache()
