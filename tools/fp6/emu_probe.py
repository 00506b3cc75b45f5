#!/usr/bin/env python3
"""Round 6, review item 1 (a): the NUMERICS of an FP6 residual pass, measured without its kernel.

tools/fp6/lib_emu_fp6.so (build_emu.sh, -DLR_EMU_FP6=1) snaps every one-byte residual of the default form -- A_lo at its producers and
the weights' twins -- to the OCP MX e2m3 grid with one scale per (row, 32 columns) before the product's e4m3 encoding, which holds the
snapped values exactly; the unchanged kernels then compute what v_mfma_scale_f32_16x16x128_f8f6f4 would with cbsz = blgp = 2.  For
every full-size reference golden of the three backbones this prints, for the product library and for the emulation:

    probe   max |default form - strict form| over the engine's 8 seeded probe rows on that golden's weights (what .to('cuda') measures;
            budget 1.5e-4) and the form it would lock
    golden  |reward - reference| with the DEFAULT form forced (calibrate=False)

    python tools/fp6/emu_probe.py [phi|llava|qwen ...]          # parent: both libraries, child processes
"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "llava-reward_amd"), ROOT]
GOLD = os.path.join(ROOT, "tests", "golden")


def cases(which):
    out = []
    if "phi" in which:
        out += [("phi", p) for p in sorted(glob.glob(os.path.join(GOLD, "ref_full_*.json"))) if "pair_sample" not in p]
    if "llava" in which:
        out += [("llava", p) for p in sorted(glob.glob(os.path.join(GOLD, "ref_llava_full_*.json")))]
    if "qwen" in which:
        out += [("qwen", p) for p in sorted(glob.glob(os.path.join(GOLD, "ref_qwen_full_*.json")))]
    return out


def child(which):
    import torch
    from llava_reward_amd import synth
    from llava_reward_amd.model import RewardModel
    tag = os.environ.get("FP6_TAG", "?")
    for kind, path in cases(which):
        g = json.load(open(path))
        prof = g.get("weight_profile", 0)
        if kind == "phi":
            cfg = synth.RewardConfig.from_json(g["config"])
            grids = g["grids"]
            grids = tuple(grids) if isinstance(grids[0], int) else [tuple(x) for x in grids]
            batch = synth.synth_batch(cfg, g["seed"], g["caption_lens"], grids, max_crops=g["max_crops"])
            B, S = batch["input_ids"].shape
            kw = dict(max_batch=2 * B, max_seq=S, max_crops=17)
        elif kind == "llava":
            cfg = synth.LlavaConfig.from_json(g["config"])
            batch = synth.llava_synth_batch(cfg, g["seed"], g["caption_lens"], [tuple(x) for x in g["image_sizes"]], max_crops=g["max_crops"])
            kw = dict(max_batch=3, max_seq=4096, max_crops=5)
        else:
            cfg = synth.QwenConfig.from_json(g["config"])
            batch = synth.qwen_synth_batch(cfg, g["seed"], g["caption_lens"], [tuple(x) for x in g["grids"]])
            kw = dict(max_batch=2, max_seq=batch["input_ids"].shape[1], max_patches=2 * int(batch["pixel_values"].shape[0]))
        ref = torch.tensor(g["reward"], dtype=torch.float32)
        m = RewardModel(cfg, synth_seed=g["seed"], operand_dtype="f16x2f8", synth_profile=prof, **kw).to("cuda").eval()
        info = dict(m.form_info or {})
        m.operand_form = "default"
        m._apply_form()
        tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
        if kind == "phi":
            r = m.custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"])[0]
        else:
            r = m.custom_forward(inputs_batch=tb)[0]
        torch.cuda.synchronize()
        err = (r.cpu().reshape(ref.shape) - ref).abs().max().item()
        print(json.dumps({"lib": tag, "golden": g["name"], "outlier": bool(prof & synth.PROFILE_OUTLIER), "probe_default_vs_strict": info.get("default_vs_strict"),
                          "locked": info.get("form"), "golden_err_default_form": err}), flush=True)
        m.engine.close()
        del m
        torch.cuda.empty_cache()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2:])
    which = sys.argv[1:] or ["phi", "llava", "qwen"]
    rows = {}
    for tag, lib in (("e4m3 (product)", None), ("e2m3/32 (emulated)", os.path.join(ROOT, "tools", "fp6", "lib_emu_fp6.so"))):
        env = dict(os.environ, FP6_TAG=tag)
        if lib:
            env["LLAVA_REWARD_HIP_LIB"] = lib
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + which, env=env, stdout=subprocess.PIPE, text=True, check=True)
        for line in p.stdout.splitlines():
            if line.startswith("{"):
                r = json.loads(line)
                rows.setdefault(r["golden"], {})[tag] = r
    print(f"{'golden':32s} {'probe e4m3':>11s} {'probe e2m3':>11s} {'locked e4m3':>26s} {'locked e2m3':>26s} {'golden e4m3':>12s} {'golden e2m3':>12s}")
    for gname, d in rows.items():
        a, b = d.get("e4m3 (product)", {}), d.get("e2m3/32 (emulated)", {})
        f = lambda x: f"{x:.2e}" if isinstance(x, float) else str(x)
        print(f"{gname:32s} {f(a.get('probe_default_vs_strict')):>11s} {f(b.get('probe_default_vs_strict')):>11s} {str(a.get('locked')):>26s} {str(b.get('locked')):>26s} "
              f"{f(a.get('golden_err_default_form')):>12s} {f(b.get('golden_err_default_form')):>12s}")


if __name__ == "__main__":
    main()
