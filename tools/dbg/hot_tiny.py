import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "llava-reward_amd"), ROOT]
import torch, numpy as np
from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel
from oracle import phi3v_reward_oracle as orc
cfg = synth.tiny_config(hidden=1024, intermediate=2048, heads=16, layers=4)
seed = 17
batch = synth.synth_batch(cfg, seed, [7, 3, 5], [(1, 1), (1, 2), (2, 1)], max_crops=4)
for prof in (2,):
    W = orc.weights_to_torch(synth.make_weights(cfg, seed, prof))
    ref = orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
    kw = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    for dt in ("f16x2", "f16x2f8"):
        m = RewardModel(cfg, synth_seed=seed, max_batch=4, max_seq=1024, max_crops=5, operand_dtype=dt, synth_profile=prof).to("cuda").eval()
        m.engine.set_gemm_tile(6)
        r0 = m.custom_forward(**kw)[0].cpu()
        print(prof, dt, "err", (r0 - ref).abs().max().item())
        if dt == "f16x2f8":
            for which in ("all",):
                n = m.calibrate(kw)
                r1 = m.custom_forward(**kw)[0].cpu()
                print(prof, dt, "calibrated", n, "err", (r1 - ref).abs().max().item())
