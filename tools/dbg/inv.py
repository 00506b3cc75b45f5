import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd")); sys.path.insert(0, ROOT)
import torch, numpy as np
from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel
from oracle import phi3v_reward_oracle as orc
for rank in (0, 16):
    for dtype in ("f16x2", "f16x2f8", "f16"):
        cfg = synth.tiny_config(lora_rank=rank, is_general_preference=True, value_head_dim=2)
        seed = 31
        batch = synth.synth_batch(cfg, seed, [7, 3, 5], [(1, 1), (1, 2), (2, 1)], max_crops=4)
        W = orc.weights_to_torch(synth.make_weights(cfg, seed))
        m = RewardModel(cfg, synth_seed=seed, max_batch=4, max_seq=1024, max_crops=5, operand_dtype=dtype).to("cuda").eval()
        tb = {k: torch.from_numpy(v) for k, v in batch.items()}
        full, _ = m.custom_forward(tb["input_ids"].cuda(), tb["attention_mask"].cuda(), tb["pixel_values"].cuda(), tb["image_sizes"])
        ref = orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
        print(rank, dtype, "batch err", (full.cpu() - ref).abs().max().item())
        for i in range(3):
            sl = slice(i, i + 1)
            one, _ = m.custom_forward(tb["input_ids"][sl].cuda(), tb["attention_mask"][sl].cuda(), tb["pixel_values"][sl].cuda(), tb["image_sizes"][sl])
            r1 = orc.custom_forward(W, cfg, batch["input_ids"][sl], batch["attention_mask"][sl], batch["pixel_values"][sl], batch["image_sizes"][sl])
            print("   row", i, "hip one vs hip batch", (one.cpu()[0] - full.cpu()[i]).abs().max().item(), " oracle one vs oracle batch", (r1[0] - ref[i]).abs().max().item(),
                  " hip one vs oracle one", (one.cpu()[0] - r1[0]).abs().max().item())
