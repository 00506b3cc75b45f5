#!/usr/bin/env python3
"""What e4m3 K_lo / V_lo residuals in attention would cost in parity: rewards of N full-size Phi-3.5-V rows in
the default form, with and without LR_ATT_EMU_LO8=1 (the engine then rounds the K / V residuals its attention kernels read to e4m3
with one scale per (token, head): csrc/rowops.hip emulate_lo8_kernel), each against the strict form of the same process.  Two
processes (the switch is read once).   python3 tools/dbg/attn_lo8_probe.py [rows] [profile]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [os.path.join(ROOT, "llava-reward_amd"), ROOT]
    import torch
    from llava_reward_amd import synth
    from llava_reward_amd.model import RewardModel
    rows, profile = int(sys.argv[2]), int(sys.argv[3])
    cfg = synth.full_config()
    b = synth.synth_batch(cfg, 77, ([128, 64, 200, 17, 96, 128, 33, 150] * 4)[:rows], (4, 4), with_pixels=False)
    ids, mask = torch.from_numpy(b["input_ids"]).cuda(), torch.from_numpy(b["attention_mask"]).cuda()
    pix = torch.randn(rows, 17, 3, 336, 336, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    m = RewardModel(cfg, synth_seed=77, max_batch=rows, max_seq=ids.shape[1], max_crops=17, synth_profile=profile, calibrate=False).to("cuda").eval()
    out = {}
    for name, args in (("strict", (1, 1, 0, 0)), ("default", (-1, -1, 0, 0))):
        m.engine.set_precision_map(*args)
        out[name] = m.engine.forward(ids, mask, pix, b["image_sizes"]).clone().cpu().double()
    d = (out["default"] - out["strict"]).abs()
    print(f"RESULT {d.pow(2).mean().sqrt().item():.3e} {d.max().item():.3e}")
    sys.exit(0)
rows = sys.argv[1] if len(sys.argv) > 1 else "8"
for profile in ([sys.argv[2]] if len(sys.argv) > 2 else ["0", "2"]):
    for emu in ("0", "1"):
        env = dict(os.environ, LR_ATT_EMU_LO8=emu)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", rows, profile], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print(f"profile {profile}  K_lo / V_lo as e4m3: {'yes' if emu == '1' else 'no ':3s}  default form vs strict over {rows} rows: rms / max = "
              + (line[0][7:] if line else "FAILED\n" + r.stderr[-2000:]), flush=True)
