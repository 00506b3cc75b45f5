#!/usr/bin/env python3
"""Full-size fixture for the W8A8 mode (operand_dtype="fp8", BASELINE configs[4] "fp8 MFMA weight path") of LLaVA-v1.6-Mistral-7B.

The reference has no fp8 path, so there is nothing of it to run here: this script runs the ORACLE (oracle/llava_next_reward_oracle.py,
itself pinned to the reference by tests/golden/ref_llava_*.json) with the engine's operand quantisation emulated (W8A8Round: f16
storage, every GEMM operand row to OCP e4m3 with one scale per row / output channel) on the row of ref_llava_full_e4m3_bt.json
(e4m3-VALUED weights: the reference's fp32 reward on the same weights is in that golden).  An e4m3 model is chaotic in its inputs,
so the fixture carries a TWIN emulation as well (the same quantisation without the f16 storage rounding): the distance between the
two is the bound the engine is held to (tests/test_gpu_llava.py::test_w8a8_full_size_batch64).  ~15 min, ~40 GB.

    python tests/golden/make_w8a8_emulation.py
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "llava-reward_amd"), ROOT]
from llava_reward_amd import synth  # noqa: E402
from oracle import llava_next_reward_oracle as lorc  # noqa: E402
from oracle import phi3v_reward_oracle as orc  # noqa: E402


def main():
    g = json.load(open(os.path.join(HERE, "ref_llava_full_e4m3_bt.json")))
    cfg = synth.LlavaConfig.from_json(g["config"])
    W = orc.weights_to_torch(synth.llava_make_weights(cfg, g["seed"], g["weight_profile"]))
    b = synth.llava_synth_batch(cfg, g["seed"], g["caption_lens"], [tuple(x) for x in g["image_sizes"]], max_crops=g["max_crops"])
    out = {"name": "ref_llava_full_w8a8_emulation", "source": "oracle/llava_next_reward_oracle.py with W8A8Round operand quantisation (NOT the reference: "
           "it has no fp8 path); row and weights of ref_llava_full_e4m3_bt.json", "row_golden": "ref_llava_full_e4m3_bt", "torch": torch.__version__}
    for key, opr in (("fp32", orc.Ident), ("w8a8", orc.W8A8Round(orc.f16_round)), ("w8a8_twin", orc.W8A8Round(orc.Ident))):
        t0 = time.time()
        r = lorc.custom_forward(W, cfg, b["input_ids"], b["attention_mask"], b["pixel_values"], b["image_sizes"], opr=opr)
        out[key] = r.flatten().tolist()
        print(f"[{key}] {out[key]} ({time.time() - t0:.0f}s)", flush=True)
    out["oracle_vs_reference_fp32"] = abs(out["fp32"][0] - g["reward"][0][0])
    assert out["oracle_vs_reference_fp32"] < 2e-5, out["oracle_vs_reference_fp32"]          # the oracle is pinned on this row too
    json.dump(out, open(os.path.join(HERE, "w8a8_llava_full_emulation.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
