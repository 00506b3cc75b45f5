#!/usr/bin/env python3
"""reward-pairs/sec of the HIP scoring path (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" = one custom_forward over one batch of 32 synthetic (caption, image) rows per GPU
(BASELINE.json configs[1]: Phi-3.5-V BT head + SkipCA, 336x336 image -> 17 crops -> 2509 image
tokens, 128-token caption, S = 2642) followed by the all-gather of the rewards (the only
collective of the path).  Weak scaling: rows per GPU fixed.
Inputs are resident in HBM before the timed region.  Weights: seeded synthetic (no checkpoint exists
offline); rank 0 at N=1 also times the CPU oracle on a bounded sample (cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "llava-reward_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOP_PER_PAIR = 26.86e12          # algorithmic 2*MAC per (caption, image) row, num_crops=16
PEAK_TFLOPS = 2500.0              # dense bf16/f16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"


def cpu_baseline(cfg_full):
    """Time the CPU oracle (oracle/phi3v_reward_oracle.py, 'port') on a bounded sample of the same
    workload (10-30 s of CPU work): ONE row at full shapes, run with 2 and with 6 layers of each tower; the
    per-layer cost is a quarter of the difference of the two runs and is scaled to the full depth (every layer of
    a tower has identical shapes)."""
    import dataclasses
    from llava_reward_amd import synth
    from oracle import phi3v_reward_oracle as orc
    g = torch.Generator().manual_seed(0)
    # pick the thread count that runs one decoder-layer-sized GEMM fastest (torch oversubscribes SMT boxes)
    ncpu = os.cpu_count() or 1
    probe_a, probe_w = torch.randn(2642, 3072, generator=g), torch.randn(16384, 3072, generator=g)
    best = (1e9, ncpu)
    for nt in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), min(ncpu, 32)}):
        torch.set_num_threads(nt)
        torch.nn.functional.linear(probe_a, probe_w)
        t0 = time.time()
        torch.nn.functional.linear(probe_a, probe_w)
        best = min(best, (time.time() - t0, nt))
    torch.set_num_threads(best[1])
    cores = torch.get_num_threads()
    del probe_a, probe_w

    def weights(cfg):
        W = {}
        for name, shape, std, off in synth.weight_specs(cfg):
            W[name] = torch.randn(shape, generator=g) * std + off
        return W

    batch = synth.synth_batch(cfg_full, 1234, [128], (4, 4), with_pixels=False)
    pix = torch.randn(1, 17, 3, 336, 336, generator=g)
    times = {}
    for nl in (2, 6):
        cfg = dataclasses.replace(cfg_full, layers=nl, clip=dataclasses.replace(cfg_full.clip, layers_used=nl))
        W = weights(cfg)
        t0 = time.time()
        feats = orc.clip_tower(W, pix.flatten(0, 1), cfg.clip)
        t1 = time.time()
        orc.custom_forward(W, dataclasses.replace(cfg, clip=dataclasses.replace(cfg.clip, layers_used=0)),
                           batch["input_ids"], batch["attention_mask"], pix, batch["image_sizes"])
        t2 = time.time()
        times[nl] = (t1 - t0, t2 - t1)
        del W, feats
    clip_layer = max((times[6][0] - times[2][0]) / 4, 1e-6)
    clip_base = max(times[2][0] - 2 * clip_layer, 0.0)
    dec_layer = max((times[6][1] - times[2][1]) / 4, 1e-6)
    dec_base = max(times[2][1] - 2 * dec_layer, 0.0)
    total = clip_base + cfg_full.clip.layers_used * clip_layer + dec_base + cfg_full.layers * dec_layer
    spent = sum(a + b for a, b in times.values())
    return {"value": 1.0 / total, "unit": "reward-pairs/sec", "cores": cores, "kind": "port",
            "sample": f"1 row at full shapes (17 crops, S={batch['input_ids'].shape[1]}); 2 + 6 of 23 CLIP and 2 + 6 of 32 decoder layers "
                      f"timed ({spent:.1f}s CPU), per-layer cost scaled to full depth -> {total:.1f}s per row, fp32 torch"}


def qwen_flop_per_row(cfg, grid, S):
    """Algorithmic FLOP of one Qwen2.5-VL row: ViT linears + window/full attention + merger, decoder
    linears + causal attention (lm_head excluded: the reference computes the logits and never reads them)."""
    v = cfg.vision
    N = grid[0] * grid[1]
    hd = v.head_dim
    lin = 2 * N * v.hidden * (3 * v.hidden + v.hidden + 3 * v.intermediate) * v.depth
    win = (v.window // v.patch) ** 2                                   # patches per full window
    att = 4 * N * hd * v.heads * (win * (v.depth - len(v.fullatt)) + N * len(v.fullatt))
    patch = 2 * N * v.patch_dim * v.hidden
    mh = v.hidden * v.merge_unit
    merger = 2 * (N // v.merge_unit) * (mh * mh + mh * cfg.hidden)
    D, I = cfg.hidden, cfg.intermediate
    dec = 2 * S * (D * (cfg.heads + 2 * cfg.kv_heads) * cfg.head_dim + cfg.heads * cfg.head_dim * D + 3 * D * I) * cfg.layers
    datt = 4 * (S * S // 2) * cfg.head_dim * cfg.heads * cfg.layers
    return float(lin + att + patch + merger + dec + datt)


def dominant_kernel_probe(dtype_code, tile, steps=5, split=False, lo8=False):
    """HIP-event timing of the dominant kernel (gemm_bt8 at the decoder gate_up shape, SwiGLU epilogue) on the launch stream.
    split: the split-operand form the parity mode runs (A = [A_hi | A_lo], output [hi | lo]): twice the MFMA work for the
    same algorithmic FLOPs."""
    import ctypes as C
    from llava_reward_amd import _lib as L
    lib = L.load()
    M, N, K = 32 * 2643, 16384, 3072
    tdt = torch.float16 if dtype_code == L.LR_DT_F16 else torch.bfloat16
    w = 2 if split else 1
    A = (torch.randn(M, K, device="cuda") * 1.0).to(tdt)
    if split:
        A = torch.cat([A, (torch.randn(M, K, device="cuda") * 2.0 ** -12).to(tdt)], dim=1).contiguous()     # [hi | lo]
    W = (torch.randn(N, K, device="cuda") * 0.02).to(tdt)
    out = torch.empty(M, w * N // 2, device="cuda", dtype=tdt)
    st = torch.cuda.current_stream()
    if lo8:      # split-operand form with the e4m3 residual pass: W8 twin prepared and the residual half encoded once, then timed
        W8 = torch.zeros_like(W)
        ae = torch.zeros(M, dtype=torch.int32, device="cuda")
        we = C.c_int(0)
        base = (C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(W8.data_ptr()), C.c_void_p(ae.data_ptr()), C.c_void_p(out.data_ptr()),
                C.c_void_p(0), M, N, K, L.EPI_SWIGLU_OP, 0, dtype_code)
        assert lib.lr_op_gemm_bt_mixed(*base, 7, C.byref(we), C.c_void_p(st.cuda_stream)) == 0
        fn, args = lib.lr_op_gemm_bt_mixed, base + (0, C.byref(we), C.c_void_p(st.cuda_stream))
    elif split:
        fn, args = lib.lr_op_gemm_bt_split, (C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(0),
                                             M, N, K, L.EPI_SWIGLU_OP, 0, dtype_code, tile, C.c_void_p(st.cuda_stream))
    else:
        fn, args = lib.lr_op_gemm_bt, (C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(0),
                                       M, N, K, K, K, N // 2, L.EPI_SWIGLU_OP, 0, dtype_code, tile, C.c_void_p(st.cuda_stream))
    assert fn(*args) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)                    # torch's current stream IS the stream the kernel is launched on
    for _ in range(steps):
        fn(*args)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    # PMC passes of these exact launches (profiles/r1_pmc_gemm_gate_up.md): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE
    pmc = ({"traffic": 23.9e9, "mfma_busy": 0.659, "clock_ghz": 1.75} if lo8 else
           {"traffic": 32.9e9, "mfma_busy": 0.652, "clock_ghz": 1.73} if split else {"traffic": 14.6e9, "mfma_busy": 0.626, "clock_ghz": 1.75})
    form = ", split-operand form, e4m3 residual pass" if lo8 else ", split-operand form" if split else ""
    return {"kernel": "gemm_bt8_kernel<SwiGLU> decoder gate_up" + form, "shape": [M, N, K],
            "avg_ms": ms, "tflops": 2.0 * M * N * K / (ms * 1e-3) / 1e12, "mfma_work_factor": 1.5 if lo8 else w,
            "algorithmic_bytes": 2.0 * ((1.5 if lo8 else w) * M * K + (1.5 if lo8 else 1) * N * K + w * M * N // 2), "traffic_bytes_pmc": pmc["traffic"],
            "mfma_busy_frac_pmc": pmc["mfma_busy"], "effective_clock_ghz_pmc": pmc["clock_ghz"]}


def golden_check(model, model_name):
    """Score the committed full-size golden row of this backbone (produced by the REFERENCE itself, tests/golden/make_goldens.py)
    with the engine that was just timed -- same synthetic weights (seed 1234) -- and report |reward - reference|."""
    name = {"phi3v": "ref_full_bt_ca", "llava": "ref_llava_full_bt", "qwen": "ref_qwen_full_bt"}[model_name]
    path = os.path.join(ROOT, "tests", "golden", name + ".json")
    if not os.path.exists(path):
        return None
    from llava_reward_amd import synth
    g = json.load(open(path))
    if g["seed"] != 1234:
        return None
    if model_name == "qwen":
        cfg = synth.QwenConfig.from_json(g["config"])
        b = synth.qwen_synth_batch(cfg, g["seed"], g["caption_lens"], [tuple(x) for x in g["grids"]])
    elif model_name == "llava":
        cfg = synth.LlavaConfig.from_json(g["config"])
        b = synth.llava_synth_batch(cfg, g["seed"], g["caption_lens"], [tuple(x) for x in g["image_sizes"]], max_crops=g["max_crops"])
    else:
        cfg = synth.RewardConfig.from_json(g["config"])
        b = synth.synth_batch(cfg, g["seed"], g["caption_lens"], tuple(g["grids"]), max_crops=g["max_crops"])
    tb = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    if model_name == "phi3v":
        r, _ = model.custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"])
    else:
        r, _ = model.custom_forward(inputs_batch=tb)
    torch.cuda.synchronize()
    ref = torch.tensor(g["reward"], dtype=torch.float32)
    return {"golden": name + ".json (reference fp32 CPU custom_forward)", "reference_reward": ref.flatten().tolist(),
            "reward": r.cpu().flatten().tolist(), "abs_err": (r.cpu().reshape(ref.shape) - ref).abs().max().item(), "tolerance": 1e-3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="rows per GPU")
    ap.add_argument("--dtype", default="f16x2f8", choices=["f16x2", "f16x2f8", "f16", "bf16", "bf16x2", "fp8"],
                    help="MFMA operands: f16x2f8 = split-operand parity mode with the residual pass of the big GEMMs in e4m3 (default, rewards "
                         "within 1e-3 of the fp32 reference with > 10x margin); f16x2 = the same with 16-bit residual passes (strict); "
                         "f16 / bf16 = single-pass fast modes (noise-limited, DESIGN.md §4)")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the secondary single-pass f16 measurement")
    ap.add_argument("--tile", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model", default="phi3v", choices=["phi3v", "llava", "qwen"],
                    help="phi3v = BASELINE metric (default); llava = LLaVA-v1.6-Mistral-7B shapes of configs[4] with 16-bit operands; "
                         "qwen = Qwen2.5-VL-7B shapes of configs[3]")
    ap.add_argument("--num-crops", type=int, default=16, choices=[16, 4],
                    help="phi3v: the processor's num_crops (16 = the reference's setting, utils/utils.py:24 -> 17 crops, V=2509; "
                         "4 -> 5 crops, V=757: the other reading of '336 px', SURVEY.md §8d)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU smoke tests)")
    ap.add_argument("--all-ranks-on-device", type=int, default=-1, help="smoke test: put every rank on this one GPU")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if a.all_ranks_on_device >= 0:
        local = a.all_ranks_on_device
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.backend)

    from llava_reward_amd import synth, _lib as L
    from llava_reward_amd.model import RewardModel
    from llava_reward_amd.scoring import gather_rewards

    B = a.batch
    rows = slice(rank * B, (rank + 1) * B)          # contiguous shard: gathered order == input order
    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    if a.model == "qwen":
        cfg = synth.qwen_full_config()              # Qwen2.5-VL-7B, BT head + the as-written SkipCA
        # 336^2 image -> 448^2 after the reference's min_pixels = 256*28^2 floor (utils/utils.py:35) -> 32x32 patches, 256 slots
        gb = synth.qwen_synth_batch(cfg, 1234, [128] * (B * world), [(32, 32)] * (B * world), with_pixels=False)
        ncrop, flop_per_pair = 0, qwen_flop_per_row(cfg, (32, 32), gb["input_ids"].shape[1])
        workload = "BASELINE configs[3] shapes: Qwen2.5-VL-7B, 448x448 px (32x32 patches, 256 image tokens), BT head + SkipCA"
    elif a.model == "llava":
        cfg = synth.llava_full_config()             # Mistral-7B decoder + CLIP-L, BT head, no SkipCA on this branch
        gb = synth.llava_synth_batch(cfg, 1234, [128] * (B * world), [(336, 336)] * (B * world), with_pixels=False)
        ncrop, flop_per_pair = 3, 19.9e12           # 3 crops -> 1176 image tokens; see DESIGN.md §8
        workload = "LLaVA-v1.6-Mistral-7B (BASELINE configs[4] shapes, 16-bit operands), 3 crops/img, V=1176"
    else:
        cfg = synth.full_config()                   # BT head (d=1) + SkipCA
        if a.num_crops == 16:
            gb = synth.synth_batch(cfg, 1234, [128] * (B * world), (4, 4), with_pixels=False)
            ncrop, flop_per_pair = 17, FLOP_PER_PAIR
            workload = "BASELINE configs[1]: Phi-3.5-V BT head + SkipCA, 17 crops/img, V=2509"
        else:
            gb = synth.synth_batch(cfg, 1234, [128] * (B * world), (2, 2), with_pixels=False)
            ncrop, flop_per_pair = 5, 8.48e12
            workload = "BASELINE configs[1] at num_crops=4: Phi-3.5-V BT head + SkipCA, 5 crops/img, V=757"
    S = gb["input_ids"].shape[1]
    ids = torch.from_numpy(gb["input_ids"][rows]).cuda()
    mask = torch.from_numpy(gb["attention_mask"][rows]).cuda()
    precise = "x2" in a.dtype
    if a.model == "qwen":
        sizes = torch.from_numpy(gb["image_grid_thw"][rows])
        pix = torch.randn(B * 32 * 32, cfg.vision.patch_dim, device="cuda", generator=gen)   # normalised pixel noise, fp32
    else:
        sizes = torch.from_numpy(gb["image_sizes"][rows])
        pix = torch.randn(B, ncrop, 3, 336, 336, device="cuda", generator=gen)      # CLIP-normalised pixel noise, fp32

    def build_model(dtype, fp32_valued=False):
        if a.model == "qwen":
            m = RewardModel(cfg, synth_seed=1234, max_batch=B, max_seq=S, max_patches=B * 32 * 32, operand_dtype=dtype)
        else:
            m = RewardModel(cfg, synth_seed=1234, max_batch=B, max_seq=S, max_crops=max(ncrop, 5 if a.model == 'llava' else 17), operand_dtype=dtype)
        m.synth_fp32_valued = fp32_valued
        m = m.to(f"cuda:{local}").eval()
        if a.tile >= 0:
            m.engine.set_gemm_tile(a.tile)
        return m

    def run_forward(m):
        return m.engine.forward_qwen(ids, mask, pix, sizes) if a.model == "qwen" else m.engine.forward(ids, mask, pix, sizes)

    def timed_steps(fn, warmup, steps):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    model = build_model(a.dtype)

    def step():
        r = run_forward(model)
        if world == 1:
            return r
        if a.backend != "nccl":               # gloo smoke path: collectives on host tensors
            return gather_rewards(r.cpu())
        return gather_rewards(r)

    for _ in range(a.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda" if a.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out).all(), "non-finite rewards"

    if rank == 0:
        value = world * B * a.steps / dt
        tf_per_gpu = value * flop_per_pair / world / 1e12
        res = {
            "metric": "reward-pairs/sec (336px img, 128-tok caption) " + {"phi3v": "Phi-3.5-V", "llava": "LLaVA-v1.6-Mistral-7B", "qwen": "Qwen2.5-VL-7B"}[a.model], "value": value, "unit": "reward-pairs/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": f"{a.dtype} MFMA operands (bf16-valued weights), f32 accumulate/residual/softmax",
            "data": "synthetic (seeded weights and inputs; no checkpoint offline)",
            "config": {"workload": workload + ", S=%d" % S,
                       "rows_per_gpu": B, "global_batch": B * world, "seq_len": S, "parallelism": f"dp{world}",
                       "collective": "all_gather rewards [B,1] fp32" if world > 1 else "none"},
            "roofline": {"bound": "mfma", "achieved": tf_per_gpu, "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf_per_gpu / PEAK_TFLOPS, "traffic": None,
                         "note": "whole pass: pairs/s x %.2f TFLOP ALGORITHMIC per pair, per GPU%s" % (
                             flop_per_pair / 1e12, ("; the split-operand mode executes 1.5x (linears: f16 pass + e4m3 residual pass at twice the rate) / 3x (attention) that MFMA time"
                              if a.dtype == "f16x2f8" else "; the split-operand mode executes 2x (linears) / 3x (attention) that MFMA work") if precise else "")},
        }
        if world == 1:
            full = a.model != "phi3v" or a.num_crops == 16      # the golden row is a 17-crop image: it does not fit the 5-crop engine
            res["parity_check"] = golden_check(model, a.model) if full else None
            if a.model == "phi3v" and a.num_crops == 16:
                dk = dominant_kernel_probe(L.LR_DT_F16 if a.dtype.startswith("f16") else L.LR_DT_BF16, a.tile, split=precise, lo8=a.dtype == "f16x2f8")
                # the roofline object proper: the dominant kernel, algorithmic FLOPs per launch / its live HIP-event duration
                res["roofline"].update({"achieved": dk["tflops"], "frac": dk["tflops"] / PEAK_TFLOPS, "traffic": dk["traffic_bytes_pmc"],
                                        "kernel": dk["kernel"], "dominant_kernel": dk,
                                        "whole_pass": {"achieved": tf_per_gpu, "frac": tf_per_gpu / PEAK_TFLOPS},
                                        "note": res["roofline"]["note"] + "; achieved/frac/traffic = the dominant kernel (algorithmic FLOPs per launch / HIP-event "
                                                "time; traffic = PMC bytes per launch, profiles/r1_pmc_gemm_gate_up.md); whole_pass = the same ratio for the step"})
            # the same step with the pixel hand-over included: fp32 pixels start in pinned host memory (what the processor
            # returns) and cross PCIe on the forward's stream every step.  Reported beside `value`, never as `value`.
            pix_host = pix.cpu().pin_memory()
            def step_h2d():
                pix.copy_(pix_host, non_blocking=True)
                return run_forward(model)
            ms_h2d = timed_steps(step_h2d, 1, a.steps)
            res["h2d_inclusive"] = {"value": B / (ms_h2d * 1e-3), "unit": "reward-pairs/sec", "ms_per_step": ms_h2d,
                                    "pixel_bytes_per_step": pix.numel() * 4, "note": "pinned host fp32 pixels copied in every step"}
            del model, pix_host
            torch.cuda.empty_cache()
            if a.model == "phi3v":
                # the image hand-over in front of the path: decoded uint8 336 px image -> pixel_values rows
                from llava_reward_amd import preprocess
                img = torch.from_numpy(synth.synth_image(1234, "bench.image", 336, 336)).cuda()
                preprocess.hd_transform_batch([img] * B, a.num_crops, out=pix)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    preprocess.hd_transform_batch([img] * B, a.num_crops, out=pix)
                torch.cuda.synchronize()
                us = 1e6 * (time.perf_counter() - t1) / (5 * B)
                res["input_handover"] = {"kernel": "lr_hd_transform (uint8 336x336 -> [%d,3,336,336] fp32, local crops bit-exact with Pillow)" % (ncrop),
                                         "us_per_image": us, "images_per_sec": 1e6 / us,
                                         "hbm_GBps_algorithmic": (ncrop * 3 * 336 * 336 * 4 + 336 * 336 * 3) / us / 1e3}
            if precise and not a.no_fast_mode:
                # secondary figure: the single-pass f16 mode of the same workload (noise-limited parity)
                fm = build_model("f16")
                ms = timed_steps(lambda: run_forward(fm), a.warmup, a.steps)
                fv = B * a.steps / (ms * 1e-3 * a.steps)
                res["fast_mode"] = {"dtype": "f16 single-pass MFMA operands", "value": fv, "unit": "reward-pairs/sec", "ms_per_step": ms,
                                    "roofline_frac_whole_pass": fv * flop_per_pair / 1e12 / PEAK_TFLOPS,
                                    "parity_check": golden_check(fm, a.model) if full else None}
                del fm
                torch.cuda.empty_cache()
            if precise and not a.no_fast_mode:
                # the same parity mode on weights that are NOT bf16-valued (fp32-valued synthetic weights): what a real LLaVA-Reward
                # checkpoint looks like once its all-linear LoRA adapter is merged (every GEMM carries the weights' residuals too)
                mm = build_model(a.dtype, fp32_valued=True)
                ms = timed_steps(lambda: run_forward(mm), a.warmup, a.steps)
                res["merged_lora_weights"] = {"dtype": a.dtype, "value": B / (ms * 1e-3), "unit": "reward-pairs/sec", "ms_per_step": ms,
                                              "note": "weights inexact in f16: third K segment per GEMM (e4m3 A_hi x e4m3 W_lo in f16x2f8; 16-bit in f16x2)"}
                del mm
                torch.cuda.empty_cache()
            if a.model == "phi3v" and a.num_crops == 16 and not a.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
