import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "llava-reward_amd"), ROOT]
import torch, numpy as np
from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel
cfg = synth.full_config()
rows = 2
b = synth.synth_batch(cfg, 77, [128, 64], (4, 4), with_pixels=False)
ids, mask = torch.from_numpy(b["input_ids"]).cuda(), torch.from_numpy(b["attention_mask"]).cuda()
pix = torch.randn(rows, 17, 3, 336, 336, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
m = RewardModel(cfg, synth_seed=77, max_batch=rows, max_seq=ids.shape[1], max_crops=17, synth_profile=2).to("cuda").eval()
e = m.engine
S, D = ids.shape[1], cfg.hidden
def stages(tag):
    out = {}
    for nl in (0, 1, 2, 8, 32):
        e.set_layer_limits(-1, nl if nl < 32 else -1)
        r = e.forward(ids, mask, pix, b["image_sizes"], keep_hidden_states=True, no_final_norm=nl < 32)
        torch.cuda.synchronize()
        out[f"x{nl}"] = e.read_tap("x", rows * S * D).copy()
        if nl == 0:
            out["clip"] = e.read_tap("clip_x", rows * 17 * 577 * 1024).copy()
            out["ev"] = e.read_tap("ev", rows * 2509 * D).copy()
        if nl == 32: out["r"] = r.cpu().numpy().copy()
    return out
e.set_precision_map(1, 1, 0, 0); ref = stages("strict"); e.set_precision_map(-1, -1, 0, 0)
d0 = stages("default")
n = m.calibrate(dict(input_ids=ids, attention_mask=mask, pixel_values=pix, image_sizes=torch.from_numpy(b["image_sizes"])))
d1 = stages("calibrated")
print("hot operands", n)
for k in ref:
    a = ref[k].astype(np.float64); s = np.sqrt((a ** 2).mean())
    print(f"{k:5s} rms {s:.3e}  default rel-rms err {np.sqrt(((d0[k]-a)**2).mean())/s:.2e}  max {np.abs(d0[k]-a).max():.2e} | calibrated {np.sqrt(((d1[k]-a)**2).mean())/s:.2e} max {np.abs(d1[k]-a).max():.2e}")
