"""Default-mode reward error against the strict split-operand mode (f16x2) on several full-size rows: how much noise the one-byte
residual form adds.  LLAVA_REWARD_HIP_LIB selects the build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import numpy as np, torch
from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel
cfg = synth.full_config()
seed = 7
B = 8
batch = synth.synth_batch(cfg, seed, [128] * B, (4, 4))
tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
res = {}
for dt in ("f16x2", "f16x2f8"):
    m = RewardModel(cfg, synth_seed=seed, max_batch=B, max_seq=batch["input_ids"].shape[1], max_crops=17, operand_dtype=dt).to("cuda").eval()
    r, _ = m.custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"])
    res[dt] = r.cpu().numpy().reshape(-1).astype(np.float64)
    del m
    torch.cuda.empty_cache()
d = res["f16x2f8"] - res["f16x2"]
print("rewards", np.round(res["f16x2"], 4).tolist())
print("f16x2f8 - f16x2:", " ".join(f"{x:+.1e}" for x in d), "| rms %.2e max %.2e" % (np.sqrt((d ** 2).mean()), np.abs(d).max()))
