#!/usr/bin/env python3
"""How far numerically EQUIVALENT builds land from the reference on the outlier-bearing full-size goldens (round 5): the same strict
split-operand forward with other GEMM tile shapes (= other fp32 summation orders, lr_set_gemm_tile) and, given other libraries on
the command line (e.g. an attention build with -DLR_ATT_LAZY=0), with those.  The spread is the floor any error bound on these rows has
to sit above: the reference's own fp32 arithmetic is one such draw.
    python tools/outlier_noise_probe.py [golden name] [other .so ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "llava-reward_amd"), ROOT]
import torch
from llava_reward_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "ref_full_outlier_gpm2_ca"
libs = [None] + sys.argv[2:]
g = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
cfg = synth.RewardConfig.from_json(g["config"])
grids = g["grids"]
grids = tuple(grids) if isinstance(grids[0], int) else [tuple(x) for x in grids]
batch = synth.synth_batch(cfg, g["seed"], g["caption_lens"], grids, max_crops=g["max_crops"])
ref = torch.tensor(g["reward"], dtype=torch.float32)
tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
for lib in libs:
    if lib:
        os.environ["LLAVA_REWARD_HIP_LIB"] = os.path.abspath(lib)
    for mod in [m for m in list(sys.modules) if m.startswith("llava_reward_amd")]:
        del sys.modules[mod]
    from llava_reward_amd.model import RewardModel
    for dtype in ("f16x2", "f16x2f8"):
        m = RewardModel(cfg, synth_seed=g["seed"], max_batch=2, max_seq=batch["input_ids"].shape[1], max_crops=17, operand_dtype=dtype,
                        synth_profile=g.get("weight_profile", 0), operand_form="strict" if dtype == "f16x2f8" else None).to("cuda").eval()
        out = []
        for tile in ((-1, 0, 1, 2) if dtype == "f16x2" else (-1,)):
            m.engine.set_gemm_tile(tile)
            r = m.custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"])[0]
            torch.cuda.synchronize()
            out.append((tile, (r.cpu().reshape(ref.shape) - ref).abs().max().item()))
        print(f"{os.path.basename(lib) if lib else 'product':22s} {dtype:8s} " + "  ".join(f"tile {t:2d}: {e:.2e}" for t, e in out), flush=True)
        m.engine.close()
