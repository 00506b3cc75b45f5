#!/usr/bin/env python3
"""ORACLE fixture (not a reference output): the Phi-3.5-V reward path evaluated in DOUBLE precision on full-size golden rows.

Why: `ref_full_outlier_gpm2_ca` (outlier-bearing weights, rewards +-3.8) amplifies fp32-level rounding 15-25x, and the HIP strict form
sits 1e-4 .. 5e-4 from the reference's fp32 reward on it depending on fp32-level details (summation order, the softmax reference
maximum).  Whether that is the engine's error or the reference's own fp32 noise cannot be told from two fp32 numbers: this script runs
oracle/phi3v_reward_oracle.custom_forward with dtype=float64 (weights up-cast one tensor at a time, every activation, softmax, norm
and the RoPE table in double) on the same (config, seed) rows and stores

    reward_fp64          the double-precision value of the function the reference computes
    reward_oracle_fp32   the fp32 oracle on this machine (one more fp32 draw)
    reward_reference     copied from the committed reference golden (the reference itself, fp32, CPU)

so that the GPU tests can assert |hip - fp64| beside |hip - reference| (tests/test_gpu_forward.py).  Container only (~35 GB RSS,
10-25 min per row on 8 cores); the JSON it writes is data.

    python tests/golden/make_fp64_fixture.py [name ...]      default: the two outlier rows and the benign GPM row
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
sys.path.insert(0, ROOT)
from llava_reward_amd import synth  # noqa: E402
from oracle import phi3v_reward_oracle as orc  # noqa: E402

DEFAULT = ["ref_full_outlier_gpm2_ca", "ref_full_outlier_bt_ca", "ref_full_gpm2_ca"]


def run(name: str, with_fp32: bool = True) -> dict:
    g = json.load(open(os.path.join(HERE, f"{name}.json")))
    cfg = synth.RewardConfig.from_json(g["config"])
    seed, profile = g["seed"], g.get("weight_profile", 0)
    grids = [tuple(x) for x in g["grids"]] if isinstance(g["grids"][0], list) else tuple(g["grids"])
    batch = synth.pad_left(synth.synth_batch(cfg, seed, g["caption_lens"], grids, max_crops=g["max_crops"]), g.get("extra_left_pad", 0))
    t0 = time.time()
    W = orc.weights_to_torch(synth.make_weights(cfg, seed, profile))
    print(f"[{name}] weights in {time.time() - t0:.0f}s", flush=True)
    args = (cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
    out = {"name": name, "kind": "ORACLE fixture: oracle/phi3v_reward_oracle.py in float64 (not a reference output)",
           "seed": seed, "weight_profile": profile, "reward_reference": g["reward"], "torch": torch.__version__,
           "threads": torch.get_num_threads()}
    if with_fp32:
        t0 = time.time()
        r32 = orc.custom_forward(W, *args)
        out["reward_oracle_fp32"] = r32.double().tolist()
        print(f"[{name}] fp32 oracle {time.time() - t0:.0f}s {r32.flatten().tolist()}", flush=True)
    t0 = time.time()
    r64 = orc.custom_forward(orc.UpcastWeights(W, torch.float64), *args, dtype=torch.float64)
    out["reward_fp64"] = r64.tolist()
    out["fp64_seconds"] = time.time() - t0
    ref = np.asarray(g["reward"], dtype=np.float64)
    out["reference_minus_fp64"] = float(np.abs(ref - r64.numpy()).max())
    if with_fp32:
        out["oracle_fp32_minus_fp64"] = float(np.abs(np.asarray(out["reward_oracle_fp32"]) - r64.numpy()).max())
    print(f"[{name}] fp64 {out['fp64_seconds']:.0f}s {r64.flatten().tolist()}  |reference - fp64| = {out['reference_minus_fp64']:.3e}", flush=True)
    return out


def main():
    names = sys.argv[1:] or DEFAULT
    path = os.path.join(HERE, "fp64_full_rows.json")
    rows = json.load(open(path)) if os.path.exists(path) else {}
    for n in names:
        rows[n] = run(n)
        with open(path, "w") as f:
            json.dump(rows, f, indent=1)
    print(f"wrote {path}")


if __name__ == "__main__":
    main()
