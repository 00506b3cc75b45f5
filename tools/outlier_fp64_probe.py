#!/usr/bin/env python3
"""Round 6: whose error is it on the outlier-bearing full-size goldens?  For every row of tests/golden/fp64_full_rows.json (the oracle in
DOUBLE precision, make_fp64_fixture.py) prints |hip - reference_fp32| and |hip - fp64| for the strict form (two GEMM tile shapes = two
fp32 summation orders), the form .to('cuda') locks, and the default form forced -- once per setting of the attention kernels' lazy
reference-maximum threshold (engine.h apply_prec: LR_ATT_LAZY_T for default-form stages, LR_ATT_LAZY_T_STRICT for strict stages; read once
per process, so every setting runs in a child process).
    python tools/outlier_fp64_probe.py [golden ...]            # parent: all settings
    python tools/outlier_fp64_probe.py --child golden ...      # one setting (the environment's)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "llava-reward_amd"), ROOT]
SETTINGS = [("0", "0"), ("2", "0"), ("4", "0"), ("8", "0"), ("8", "8")]          # (default-form stages, strict stages); product = ("8", "0")


def child(names):
    import torch
    from llava_reward_amd import synth
    from llava_reward_amd.model import RewardModel
    rows = json.load(open(os.path.join(ROOT, "tests", "golden", "fp64_full_rows.json")))
    for name in names:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
        cfg = synth.RewardConfig.from_json(g["config"])
        grids = g["grids"]
        grids = tuple(grids) if isinstance(grids[0], int) else [tuple(x) for x in grids]
        batch = synth.synth_batch(cfg, g["seed"], g["caption_lens"], grids, max_crops=g["max_crops"])
        ref = torch.tensor(g["reward"], dtype=torch.float64)
        f64 = torch.tensor(rows[name]["reward_fp64"], dtype=torch.float64)
        tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}

        def score(m):
            r = m.custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"])[0]
            torch.cuda.synchronize()
            r = r.double().cpu().reshape(ref.shape)
            return f"|hip-ref| {(r - ref).abs().max().item():.2e}  |hip-fp64| {(r - f64).abs().max().item():.2e}"
        kw = dict(synth_seed=g["seed"], max_batch=2, max_seq=batch["input_ids"].shape[1], max_crops=17, synth_profile=g.get("weight_profile", 0))
        m = RewardModel(cfg, operand_dtype="f16x2", **kw).to("cuda").eval()
        for tile in (-1, 0):
            m.engine.set_gemm_tile(tile)
            print(f"  {name:28s} strict f16x2 tile {tile:2d}            {score(m)}", flush=True)
        m.engine.close()
        m = RewardModel(cfg, operand_dtype="f16x2f8", **kw).to("cuda").eval()
        print(f"  {name:28s} locked: {m.operand_form:24s} {score(m)}", flush=True)
        for form in ("strict", "default"):
            m.operand_form = form
            m._apply_form()
            print(f"  {name:28s} f16x2f8 forced {form:16s} {score(m)}", flush=True)
        m.engine.close()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2:])
    names = sys.argv[1:] or ["ref_full_outlier_gpm2_ca", "ref_full_outlier_bt_ca", "ref_full_gpm2_ca"]
    rows = json.load(open(os.path.join(ROOT, "tests", "golden", "fp64_full_rows.json")))
    for n in names:
        r = rows[n]
        print(f"# {n}: |reference_fp32 - fp64| = {r['reference_minus_fp64']:.2e}   |oracle_fp32 - fp64| = {r.get('oracle_fp32_minus_fp64', float('nan')):.2e}", flush=True)
    for t_def, t_strict in SETTINGS:
        print(f"lazy threshold: default-form stages {t_def}, strict stages {t_strict}" + ("   <- product" if (t_def, t_strict) == ("8", "0") else ""), flush=True)
        env = dict(os.environ, LR_ATT_LAZY_T=t_def, LR_ATT_LAZY_T_STRICT=t_strict)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + names, env=env, check=True)


if __name__ == "__main__":
    main()
