#!/usr/bin/env python3
"""Per-stage operand forms (lr_set_precision_map): reward distance to the strict split-operand form (f16x2, itself <= 6e-6 from the
reference on every full-size golden) over N full-size Phi-3.5-V rows, for a list of maps, on two weight sets (default / outlier-bearing).
    python tools/prec_map_probe.py [rows] [profile]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "llava-reward_amd"), ROOT]
import numpy as np, torch
from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
profile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = synth.full_config(**({"lora_rank": int(os.environ["PROBE_LORA"])} if os.environ.get("PROBE_LORA") else {}))      # PROBE_LORA=128: un-merged rank-128 adapters
b = synth.synth_batch(cfg, 77, [128, 64, 200, 17, 96, 128, 33, 150][:rows] + [128] * max(0, rows - 8), (4, 4), with_pixels=False)
ids, mask = torch.from_numpy(b["input_ids"]).cuda(), torch.from_numpy(b["attention_mask"]).cuda()
pix = torch.randn(rows, 17, 3, 336, 336, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
m = RewardModel(cfg, synth_seed=77, max_batch=rows, max_seq=ids.shape[1], max_crops=17, synth_profile=profile, calibrate=False).to("cuda").eval()


def run(cm, mid, first, last):
    m.engine.set_precision_map(cm, mid, first, last)
    r = m.engine.forward(ids, mask, pix, b["image_sizes"]).clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        m.engine.forward(ids, mask, pix, b["image_sizes"])
    torch.cuda.synchronize()
    return r.cpu().double(), (time.perf_counter() - t0) / 2 * 1e3


ref, t_ref = run(1, 1, 0, 0)
print(f"profile {profile}: rewards (strict) {ref.flatten().tolist()}")
print(f"{'map':44s} {'rms':>9s} {'max':>9s} {'ms':>8s}")
L = cfg.layers
MAPS = [("strict everywhere (f16x2)", (1, 1, 0, 0)), ("default everywhere (f16x2f8)", (-1, -1, 0, 0)),
        ("CLIP single pass, decoder default", (0, -1, 0, 0)), ("CLIP strict, decoder default", (1, -1, 0, 0)),
        ("CLIP default, decoder single pass", (-1, 0, 0, 0)),
        ("decoder layers 4..27 single", (-1, 0, 4, 4)), ("decoder layers 8..23 single", (-1, 0, 8, 8)),
        ("decoder layers 12..19 single", (-1, 0, 12, 12)), ("decoder layers 0..15 single", (-1, 0, 0, 16)),
        ("decoder layers 16..31 single", (-1, 0, 16, 0)), ("everything single pass (f16)", (0, 0, 0, 0))]
if len(sys.argv) > 3 and sys.argv[3] == "strict-stages":       # which stages have to be strict on this weight set (round 4)
    MAPS = [("strict everywhere (f16x2)", (1, 1, 0, 0)), ("default everywhere (f16x2f8)", (-1, -1, 0, 0)),
            ("CLIP strict, decoder default", (1, -1, 0, 0)), ("CLIP default, decoder strict", (-1, 1, 0, 0)),
            ("CLIP default, decoder 0..7 strict", (-1, 1, 0, L - 8)), ("CLIP default, decoder 0..15 strict", (-1, 1, 0, L - 16)),
            ("CLIP default, decoder 16..31 strict", (-1, 1, 16, 0)), ("CLIP default, decoder 24..31 strict", (-1, 1, 24, 0)),
            ("CLIP strict, decoder 0..15 strict", (1, 1, 0, L - 16))]
if len(sys.argv) > 3 and sys.argv[3] == "ladder":       # round 5: the finer candidate ladder + cheaper-than-default tails
    MAPS = [("strict everywhere (f16x2)", (1, 1, 0, 0)), ("default everywhere (f16x2f8)", (-1, -1, 0, 0)),
            ("strict-vision", (1, -1, 0, 0))] + \
           [(f"strict-vision+decoder 0..{k - 1} strict", (1, 1, 0, L - k)) for k in (2, 4, 6, 8, 10, 12, 16)] + \
           [(f"default, decoder last {k} single pass", (-1, 0, L - k, 0)) for k in (1, 2, 4, 8)]
if len(sys.argv) > 3 and sys.argv[3] == "sites":        # round 6: which SITES of the strict decoder layers need their 16-bit residuals (lr_set_precision_sites)
    k = int(sys.argv[4]) if len(sys.argv) > 4 else L          # strict decoder layers 0..k-1 (vision tower strict), the rest default
    names = ("qkv", "attention", "o_proj", "gate_up", "down")
    SITES = [("all five sites strict", (1, 1, 1, 1, 1))] + [(f"{n} default, the others strict", tuple(2 if j == i else 1 for j in range(5))) for i, n in enumerate(names)] + \
            [("qkv + gate_up strict (the normed-stream GEMMs)", (1, 2, 2, 1, 2)), ("qkv + gate_up + attention strict", (1, 1, 2, 1, 2)),
             ("qkv + gate_up + down strict", (1, 2, 2, 1, 1)), ("qkv + gate_up + o_proj strict", (1, 2, 1, 1, 2)),
             ("o_proj + down strict (the residual-add GEMMs)", (2, 2, 1, 2, 1)), ("gate_up + down strict (the MLP)", (2, 2, 2, 1, 1)),
             ("qkv + attention + o_proj strict (the attention block)", (1, 1, 1, 2, 2)), ("all five sites default (= strict-vision)", (2, 2, 2, 2, 2))]
    print("the attention launches of DEFAULT-form stages with the lazy maximum (threshold 8) / with the exact one (lr_set_attention_lazy_threshold):")
    for nm, args in [("default everywhere", (-1, -1, 0, 0)), ("strict-vision", (1, -1, 0, 0))] + [(f"strict-vision+decoder 0..{kk - 1} strict", (1, 1, 0, L - kk)) for kk in (4, 8, 12, 16)]:
        out = []
        for thr in (8.0, 0.0):
            m.engine.set_attention_lazy_threshold(thr, 0.0)
            r, ms = run(*args)
            out.append(f"{(r - ref).abs().max().item():9.2e} {ms:8.1f} ms")
        print(f"{nm:56s} lazy {out[0]}   exact {out[1]}", flush=True)
    m.engine.set_attention_lazy_threshold(0.0, 0.0)
    print(f"strict-vision + decoder layers 0..{k - 1} strict, by site (exact maximum in every attention launch):")
    for name, st in SITES:
        m.engine.set_precision_sites(*st)
        r, ms = run(1, 1, 0, L - k)
        d = (r - ref).abs()
        print(f"{name:56s} {d.pow(2).mean().sqrt().item():9.2e} {d.max().item():9.2e} {ms:8.1f}", flush=True)
    m.engine.set_precision_sites()
    m.engine.set_attention_lazy_threshold(8.0, 0.0)
    sys.exit(0)
for name, args in MAPS:
    r, ms = run(*args)
    d = (r - ref).abs()
    print(f"{name:44s} {d.pow(2).mean().sqrt().item():9.2e} {d.max().item():9.2e} {ms:8.1f}")
