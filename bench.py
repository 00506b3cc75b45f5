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
import re
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
    """Time the CPU oracle (oracle/phi3v_reward_oracle.py, kind 'port') on the GPU box's host cores: ONE row of the headline
    workload at full shapes and FULL depth (17 crops x 23 CLIP layers, S = 2642 x 32 decoder layers, SkipCA, value head), fp32
    torch, nothing extrapolated.  The weights of one layer of each tower are random tensors shared by all of its
    layers: the time depends on shapes only, and 16.7 GB of distinct random fp32 weights would cost more to generate than the
    forward takes.  Threads: the count that runs a decoder-sized GEMM fastest (torch oversubscribes SMT boxes)."""
    from llava_reward_amd import synth
    from oracle import phi3v_reward_oracle as orc
    g = torch.Generator().manual_seed(0)
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
    threads = torch.get_num_threads()
    del probe_a, probe_w
    W, shared = {}, {}
    for name, shape, std, off in synth.weight_specs(cfg_full):
        key = re.sub(r"layers\.\d+\.", "layers.N.", name)      # one tensor per layer-local name
        if key not in shared:
            shared[key] = torch.randn(shape, generator=g) * std + off
        W[name] = shared[key]
    batch = synth.synth_batch(cfg_full, 1234, [128], (4, 4), with_pixels=False)
    pix = torch.randn(1, 17, 3, 336, 336, generator=g)
    t0 = time.time()
    r = orc.custom_forward(W, cfg_full, batch["input_ids"], batch["attention_mask"], pix, batch["image_sizes"])
    total = time.time() - t0
    assert torch.isfinite(r).all()
    # how it scales with rows and threads (SURVEY.md §8d asks for B = 1 and B = 2): the same pipeline at full SHAPES and reduced
    # DEPTH (2 of 23 CLIP layers, 2 of 32 decoder layers: a bounded sample), B = 1 and 2, at 32 / 64 / 128 threads
    import dataclasses
    cfg_s = dataclasses.replace(cfg_full, layers=2, clip=dataclasses.replace(cfg_full.clip, layers_used=2))
    Ws = {k: v for k, v in W.items() if not re.search(r"layers\.([2-9]|[1-9]\d)\.", k)}
    scaling = []
    for B in (1, 2):
        bb = synth.synth_batch(cfg_s, 1234, [128] * B, (4, 4), with_pixels=False)
        pp = torch.randn(B, 17, 3, 336, 336, generator=g)
        for nt in (32, 64, 128):
            if nt > ncpu:
                continue
            torch.set_num_threads(nt)
            t1 = time.time()
            orc.custom_forward(Ws, cfg_s, bb["input_ids"], bb["attention_mask"], pp, bb["image_sizes"])
            scaling.append({"rows": B, "threads": nt, "seconds": time.time() - t1})
    torch.set_num_threads(threads)
    return {"value": 1.0 / total, "unit": "reward-pairs/sec", "cores": threads, "host_cpus": ncpu, "kind": "port",
            "seconds_per_row": total,
            "cores_note": f"{threads} = the thread count that ran a decoder-sized GEMM fastest on this host (probed over {sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), min(ncpu, 32)})}), not the host's CPU count",
            "thread_scaling_sample": {"what": "same pipeline, full shapes, 2 CLIP + 2 decoder layers", "runs": scaling},
            "sample": f"1 row at full shapes and full depth (17 crops x {cfg_full.clip.layers_used} CLIP layers, S={batch['input_ids'].shape[1]} x "
                      f"{cfg_full.layers} decoder layers), measured end to end ({total:.1f}s), fp32 torch, {threads} threads of {ncpu} host CPUs"}


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


PMC_PROFILE = os.path.join(ROOT, "profiles", "r6_pmc_gemm_gate_up.json")
KERNEL_SOURCES = ("llava-reward_amd/csrc/gemm8.hip", "llava-reward_amd/csrc/common.h")


def kernel_source_sha16():
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def pmc_for(form):
    """PMC figures of the dominant kernel (HBM-side bytes per launch, MFMA-pipe busy fraction, effective clock) from the committed
    rocprofv3 summary profiles/r6_pmc_gemm_gate_up.json (tools/pmc_summary.py writes it from separate --pmc passes of
    tools/gemm_one.py).  They describe THIS build only if the kernel sources are the ones that were profiled: the file records
    their hash and the kernel's template signature; on any mismatch the fields are null (stale profile) instead of a stale number."""
    none = {"traffic": None, "mfma_busy": None, "clock_ghz": None, "pmc_source": None}
    try:
        prof = json.load(open(PMC_PROFILE))
    except Exception as e:
        return dict(none, pmc_note=f"no committed PMC profile ({type(e).__name__})")
    ent = prof.get("forms", {}).get(form)
    if not ent:
        return dict(none, pmc_note=f"profile has no entry for the '{form}' form")
    if prof.get("source_sha16") != kernel_source_sha16():
        return dict(none, pmc_note=f"stale PMC profile: kernel sources changed since {os.path.basename(PMC_PROFILE)} was collected")
    return {"traffic": ent["traffic_bytes_per_launch"], "mfma_busy": ent["mfma_busy_frac"], "clock_ghz": ent["effective_clock_ghz"],
            "pmc_source": os.path.relpath(PMC_PROFILE, ROOT), "pmc_kernel": ent["kernel_name"], "pmc_avg_ms": ent["avg_ms"]}


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
        # (as the engine launches it: the epilogue writes the output's residual half as block-scaled e4m3 too -- flag 32 -- because
        #  the down projection that reads it takes the same form)
        ae = torch.full((lib.lr_op_lo8_scratch_bytes(M, K) + lib.lr_op_lo8_scratch_bytes(M, N // 2),), 127, dtype=torch.uint8, device="cuda")
        we = C.c_int(0)
        base = (C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(W8.data_ptr()), C.c_void_p(ae.data_ptr()), C.c_void_p(out.data_ptr()),
                C.c_void_p(0), M, N, K, L.EPI_SWIGLU_OP, 0, dtype_code)
        assert lib.lr_op_gemm_bt_mixed(*base, 7, C.byref(we), C.c_void_p(st.cuda_stream)) == 0
        fn, args = lib.lr_op_gemm_bt_mixed, base + (32, C.byref(we), C.c_void_p(st.cuda_stream))
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
    # A yardstick for the nominal peak, NOT a product path: the vendor library's plain GEMM (torch.matmul -> hipBLASLt, f16 in, fp32
    # accumulate, no epilogue, one MFMA pass) on the same shape, timed the same way on the same box.
    del out
    Ah = A[:, :K].contiguous() if split else A
    torch.matmul(Ah, W.t())
    e0.record(st)
    for _ in range(steps):
        torch.matmul(Ah, W.t())
    e1.record(st)
    torch.cuda.synchronize()
    vms = e0.elapsed_time(e1) / steps
    vendor = {"what": "vendor library (hipBLASLt through torch.matmul), plain single-pass GEMM of the same shape, no epilogue: measurement only",
              "avg_ms": vms, "tflops": 2.0 * M * N * K / (vms * 1e-3) / 1e12, "frac_of_peak": 2.0 * M * N * K / (vms * 1e-3) / 1e12 / PEAK_TFLOPS}
    form = "mixed" if lo8 else "split" if split else "single"
    pmc = pmc_for(form)
    label = ", split-operand form, e4m3 residual pass" if lo8 else ", split-operand form" if split else ""
    return dict({"kernel": "gemm_bt8_kernel<SwiGLU> decoder gate_up" + label, "shape": [M, N, K],
                 "avg_ms": ms, "tflops": 2.0 * M * N * K / (ms * 1e-3) / 1e12, "mfma_work_factor": 1.5 if lo8 else w,
                 "algorithmic_bytes": 2.0 * ((1.5 if lo8 else w) * M * K + (1.5 if lo8 else 1) * N * K + (1.5 if lo8 else w) * M * N // 2),
                 "vendor_library_same_shape": vendor,
                 "mfma_rate_vs_vendor": (1.5 if lo8 else w) * vms / ms}, **pmc)


GOLDEN_GLOBS = {"phi3v": "ref_full_*.json", "llava": "ref_llava_full*.json", "qwen": "ref_qwen_full*.json"}


def _golden_batch(model_name, g):
    from llava_reward_amd import synth
    if model_name == "qwen":
        cfg = synth.QwenConfig.from_json(g["config"])
        return cfg, synth.qwen_synth_batch(cfg, g["seed"], g["caption_lens"], [tuple(x) for x in g["grids"]])
    if model_name == "llava":
        cfg = synth.LlavaConfig.from_json(g["config"])
        return cfg, synth.llava_synth_batch(cfg, g["seed"], g["caption_lens"], [tuple(x) for x in g["image_sizes"]], max_crops=g["max_crops"])
    cfg = synth.RewardConfig.from_json(g["config"])
    grids = g["grids"]
    grids = tuple(grids) if isinstance(grids[0], int) else [tuple(x) for x in grids]
    return cfg, synth.synth_batch(cfg, g["seed"], g["caption_lens"], grids, max_crops=g["max_crops"])


def _goldens(model_name):
    import glob
    out = []
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", GOLDEN_GLOBS[model_name]))):
        g = json.load(open(path))
        if "rows" in g:          # the configs[0] sample pair: its own test (tests/test_gpu_forward.py), inputs come from image files
            continue
        out.append(g)
    return out


def _golden_max_seq():
    from llava_reward_amd import synth
    out = {}
    for mn in GOLDEN_GLOBS:
        for g in _goldens(mn):
            n = max(g["caption_lens"])
            if mn == "phi3v":
                gr = g["grids"]
                gr = [tuple(gr)] if isinstance(gr[0], int) else [tuple(x) for x in gr]
                v = max(synth.num_img_tokens(336 * a, 336 * b) for a, b in gr)
            elif mn == "llava":
                v = max(synth.llava_geometry(int(h), int(w))[6] for h, w in g["image_sizes"])
            else:
                v = max(h * w // 4 for h, w in g["grids"])
            out[mn] = max(out.get(mn, 0), v + n + 8)
    return out


GOLDEN_MAX_SEQ = _golden_max_seq()


def golden_check(model, model_name, name=None, current=(1234, 0)):
    """Score EVERY committed full-size golden of this backbone whose reward head is the engine's (rows produced by the REFERENCE
    itself, tests/golden/make_goldens.py: several seeds, caption lengths and crop grids, a ragged B=2 batch, outlier-bearing and
    e4m3-valued weight sets) with the engine that was just timed -- its weights re-synthesised per (seed, weight profile), the timed
    set restored afterwards -- and report |reward - reference| per golden and the maximum.  name: one golden only."""
    from llava_reward_amd import synth
    per, worst, cur, forms, per_default = {}, None, current, {}, {}
    default_mode = model._opts["operand_dtype"] == "f16x2f8"
    head = (bool(model.is_general_preference), int(model.value_head_dim), bool(model.add_cross_attention))
    for g in _goldens(model_name):
        if name and g["name"] != name:
            continue
        cfg, b = _golden_batch(model_name, g)
        if (bool(cfg.is_general_preference), int(cfg.value_head_dim), bool(cfg.add_cross_attention)) != head:
            continue
        if g.get("mean_hidden_state") or g.get("layer_id", 32) != 32 or b["input_ids"].shape[0] > model.engine.max_batch \
                or b["input_ids"].shape[1] > model.engine.max_seq:
            continue
        want = (g["seed"], g.get("weight_profile", 0))
        tb = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
        if want != cur:
            # new weights behind the handle: the next custom_forward notices (lr_weights_epoch) and locks the operand form again on ITS
            # probe rows, exactly as .to('cuda') does for a freshly loaded checkpoint -- nothing is called here that a user would not call
            model.engine.synth_weights(want[0], getattr(model, "synth_fp32_valued", False), want[1])
            cur = want

        def score():
            if model_name == "phi3v":
                r, _ = model.custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"])
            else:
                r, _ = model.custom_forward(inputs_batch=tb)
            torch.cuda.synchronize()
            return r.cpu()
        ref = torch.tensor(g["reward"], dtype=torch.float32)
        per[g["name"]] = (score().reshape(ref.shape) - ref).abs().max().item()
        if default_mode:
            forms[g["name"]] = dict(model.form_info) if model.form_info else None
            if model.operand_form != "default":      # what the default form would have given on this weight set (reported, never used)
                model.engine.set_precision_map(-1, -1, 0, 0)
                try:
                    per_default[g["name"]] = (score().reshape(ref.shape) - ref).abs().max().item()
                finally:
                    model._apply_form()
            else:
                per_default[g["name"]] = per[g["name"]]
        if worst is None or per[g["name"]] > per[worst]:
            worst = g["name"]
    if cur != current:
        model.engine.synth_weights(current[0], getattr(model, "synth_fp32_valued", False), current[1])      # (the next forward re-locks the form)
    if not per:
        return None
    benign = {k: v for k, v in per.items() if "outlier" not in k}
    return {"golden": f"{len(per)} full-size goldens (reference fp32 CPU custom_forward; tests/golden/{GOLDEN_GLOBS[model_name]})",
            "abs_err": per[worst], "worst": worst, "per_golden": per, "max_abs_err_benign_weights": max(benign.values()) if benign else None,
            "operand_form": forms or None, "per_golden_if_default_form_were_forced": per_default or None, "tolerance": 1e-3,
            "note": "weights re-synthesised per golden through the bare drop-in sequence: the engine locks its operand form on each new weight "
                    "set by itself (seeded probe rows, default form vs strict form, no reference) and scores in that form; "
                    "per_golden_if_default_form_were_forced = the same rows with the default form pinned (reported only)"}


LINE_BUDGET = 4096                # bytes: the driver keeps an 8 KB stdout tail and parses the LAST line of it


def _r(x, nd=6):
    return None if x is None else (round(float(x), nd) if isinstance(x, (int, float)) and not isinstance(x, bool) else x)


def _form(fi):
    return None if not fi else fi.get("form")


def compact_line(res):
    """The ONE line the driver parses (rank 0's last stdout line): BASELINE.json's metric with `roofline` and `cpu_baseline`, every
    field a number or a short string, <= LINE_BUDGET bytes whatever legs ran.  Everything else the run measured -- the legs'
    full dicts, per-golden errors, notes -- goes to gpurun_out/bench_legs.json and to earlier '#leg' stdout lines (emit())."""
    rf, dk = res.get("roofline") or {}, (res.get("roofline") or {}).get("dominant_kernel") or {}
    wp = rf.get("whole_pass") or {}
    out = {k: res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data")}
    out["value"], out["ms_per_step"] = _r(out["value"], 4), _r(out["ms_per_step"], 3)
    fi = res.get("operand_form") or {}
    out["operand_form"] = {"mode": res.get("operand_mode"), "form": fi.get("form"), "default_vs_strict": _r(fi.get("default_vs_strict"), 8),
                           "budget": fi.get("budget"), "probe_rows": fi.get("rows"), "probe_seconds": _r(res.get("probe_seconds"), 3)}
    out["config"] = {k: v for k, v in (res.get("config") or {}).items() if k in ("workload", "rows_per_gpu", "global_batch", "seq_len",
                                                                                   "parallelism", "collective", "step", "input_check")}
    out["roofline"] = {"bound": rf.get("bound"), "achieved": _r(rf.get("achieved"), 2), "peak": rf.get("peak"), "unit": rf.get("unit"),
                       "frac": _r(rf.get("frac"), 4), "traffic": rf.get("traffic"), "kernel": rf.get("kernel"),
                       "kernel_ms": _r(dk.get("avg_ms"), 4), "algorithmic_flop_per_launch": (2.0 * dk["shape"][0] * dk["shape"][1] * dk["shape"][2]) if dk.get("shape") else None,
                       "mfma_busy": _r(dk.get("mfma_busy"), 4), "clock_ghz": _r(dk.get("clock_ghz"), 3), "pmc_source": dk.get("pmc_source") or dk.get("pmc_note"),
                       "whole_pass_frac": _r(wp.get("frac", rf.get("frac")), 4), "vendor_frac": _r(dk.get("mfma_rate_vs_vendor"), 4),
                       "vendor_plain_gemm_ms": _r((dk.get("vendor_library_same_shape") or {}).get("avg_ms"), 4)}
    cb = res.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _r(cb["value"], 6), "unit": cb["unit"], "cores": cb["cores"], "host_cpus": cb.get("host_cpus"),
                               "kind": cb["kind"], "seconds_per_row": _r(cb.get("seconds_per_row"), 2),
                               "sample": "1 row, full shapes, full depth, fp32 torch oracle"}
    pc = res.get("parity_check")
    if pc:
        out["parity_check"] = {"abs_err": _r(pc["abs_err"], 8), "worst": pc["worst"], "n_goldens": len(pc.get("per_golden") or {}),
                               "tolerance": pc["tolerance"], "against": "reference fp32 CPU custom_forward goldens"}
    if res.get("multi_gpu"):
        out["multi_gpu"] = res["multi_gpu"]
    legs = {}
    for name, leg in res.items():
        if not isinstance(leg, dict) or "value" not in leg or name in ("cpu_baseline",):
            continue
        e = {"value": _r(leg["value"], 3)}
        if leg.get("ms_per_step") is not None:
            e["ms"] = _r(leg["ms_per_step"], 2)
        if _form(leg.get("operand_form")):
            e["form"] = _form(leg["operand_form"])
        for k in ("vs_headline", "vs_plain", "vs_resident_inputs"):
            if leg.get(k) is not None:
                e[k] = _r(leg[k], 4)
        if isinstance(leg.get("parity_check"), dict):
            e["abs_err"] = _r(leg["parity_check"]["abs_err"], 8)
        if leg.get("parity") is not None:
            e["parity"] = leg["parity"]
        legs[name] = e
        for sub, sl in leg.items():
            if isinstance(sl, dict) and "value" in sl and sub != "parity_check":
                se = {"value": _r(sl["value"], 3)}
                if sl.get("vs_plain") is not None:
                    se["vs_plain"] = _r(sl["vs_plain"], 4)
                if _form(sl.get("operand_form")):
                    se["form"] = _form(sl["operand_form"])
                if sl.get("abs_err_vs_reference") is not None:
                    se["abs_err_vs_reference"] = _r(sl["abs_err_vs_reference"], 4)
                if sl.get("parity") is not None:
                    se["parity"] = sl["parity"]
                legs[f"{name}.{sub}"] = se
    if res.get("latency_b1"):
        legs["latency_b1"] = {"ms": _r(res["latency_b1"]["ms_per_forward"], 2), "ms_back_to_back": _r(res["latency_b1"]["ms_per_forward_enqueued_back_to_back"], 2)}
    if legs:
        out["legs"] = legs
    out["legs_file"] = res.get("legs_file")
    line = json.dumps(out, separators=(",", ":"))
    for k in ("legs", "multi_gpu"):        # never outgrow the driver: drop the optional parts (they are in legs_file)
        if len(line) > LINE_BUDGET and k in out:
            out[k] = {"dropped": "line budget; see legs_file"}
            line = json.dumps(out, separators=(",", ":"))
    assert len(line) <= LINE_BUDGET, len(line)
    return line


def emit(res, legs_dir=None):
    """stdout: one '#leg <name> <json>' line per secondary leg (greppable, each short enough to survive a tail), then -- LAST -- the
    compact headline line.  The full object goes to <legs_dir>/bench_legs.json (default gpurun_out/, which gpurun merges back)."""
    legs_dir = legs_dir or os.path.join(ROOT, "gpurun_out")
    path = None
    try:
        os.makedirs(legs_dir, exist_ok=True)
        path = os.path.join(legs_dir, "bench_legs.json")
        with open(path, "w") as f:
            json.dump(res, f, indent=1)
        res["legs_file"] = os.path.relpath(path, ROOT)
    except OSError:
        res["legs_file"] = None
    headline_keys = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"}
    for k, v in res.items():
        if k in headline_keys or not isinstance(v, dict):
            continue
        print("#leg " + k + " " + json.dumps(v, separators=(",", ":")), flush=True)
    print(compact_line(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="rows per GPU (default 32; 64 for --config gpm_pairwise)")
    ap.add_argument("--dtype", default="f16x2f8", choices=["f16x2", "f16x2f8", "f16", "bf16", "bf16x2", "fp8"],
                    help="MFMA operands: f16x2f8 = split-operand parity mode with the residual pass of the big GEMMs in e4m3 (default, rewards "
                         "within 1e-3 of the fp32 reference with > 10x margin); f16x2 = the same with 16-bit residual passes (strict); "
                         "f16 / bf16 = single-pass modes (NOT parity modes: noise-limited, DESIGN.md §4)")
    ap.add_argument("--config", default="bt", choices=["bt", "gpm_pairwise"],
                    help="bt = BASELINE configs[1] (BT head, B=32): the metric; gpm_pairwise = configs[2]: GPM head d=2 + SkipCA, a step = chosen and "
                         "rejected forward of B=64 rows each + preference_compute (reports preference-pairs/s beside reward-pairs/s)")
    ap.add_argument("--lora-rank", type=int, default=0, help="run the main workload with an un-merged rank-r adapter on the decoder linears")
    ap.add_argument("--quick", action="store_true", help="main line only: no secondary legs and no golden sweep -- only the timed workload's "
                                                          "launches (plus the dominant-kernel probe's) reach a profiler")
    ap.add_argument("--no-golden", action="store_true", help="skip the live golden sweep (parity_check)")
    ap.add_argument("--profile-run", action="store_true", help="for rocprofv3 runs (tools/profile_round.sh): --quick, and NOTHING but the timed workload's "
                    "launches reaches the profiler -- no operand-form probe at .to('cuda') (the default form is pinned: what the probe locks on these "
                    "weights), no dominant-kernel probe (roofline = the whole pass)")
    ap.add_argument("--profile-weights", default="", choices=["", "outlier", "e4m3"], help="synthetic weight profile of the main workload "
                    "(synth.PROFILE_*): outlier = massive channels / large norm gains, what trained checkpoints show")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the secondary single-pass f16 measurement")
    ap.add_argument("--no-other-backbones", action="store_true", help="skip the Qwen2.5-VL-7B / LLaVA-1.6-7B sub-lines of the default run")
    ap.add_argument("--check-inputs", default="eager", choices=["eager", "deferred"],
                    help="RewardModel(check_inputs=...): eager = the reference's exceptions raised by the forward itself (a stream drain per "
                         "forward when input_ids live on the device, as they do here); deferred = the engine marks such rows NaN")
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
    if a.profile_run:
        a.quick = True

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher -- one child process per GPU under torch.distributed.run, started before
        # anything here has touched the GPU; rank 0's JSON line passes through on stdout, any child failure is this process's exit code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

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

    import dataclasses
    from llava_reward_amd import synth, _lib as L
    from llava_reward_amd.model import RewardModel
    from llava_reward_amd.probe import PROBE_MIN_SEQ
    from llava_reward_amd.reward_adaptor_loader import preference_compute
    from llava_reward_amd.scoring import gather_rewards

    pairwise = a.config == "gpm_pairwise"
    if pairwise and a.model != "phi3v":
        raise SystemExit("--config gpm_pairwise is BASELINE configs[2]: Phi-3.5-V")

    def workload(model_name, B, num_crops=16, gpm=False, lora_rank=0):
        """Inputs (resident in HBM) and geometry of `B` rows per GPU of one backbone's headline workload."""
        rows = slice(rank * B, (rank + 1) * B)          # contiguous shard: gathered order == input order
        gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
        if model_name == "qwen":
            cfg = synth.qwen_full_config(lora_rank=lora_rank)              # Qwen2.5-VL-7B, BT head + the as-written SkipCA
            # 336^2 image -> 448^2 after the reference's min_pixels = 256*28^2 floor (utils/utils.py:35) -> 32x32 patches, 256 slots
            gb = synth.qwen_synth_batch(cfg, 1234, [128] * (B * world), [(32, 32)] * (B * world), with_pixels=False)
            ncrop, flop = 0, qwen_flop_per_row(cfg, (32, 32), gb["input_ids"].shape[1])
            name = "BASELINE configs[3] shapes: Qwen2.5-VL-7B, 448x448 px (32x32 patches, 256 image tokens), BT head + SkipCA"
            sizes = torch.from_numpy(gb["image_grid_thw"][rows])
            pix = torch.randn(B * 32 * 32, cfg.vision.patch_dim, device="cuda", generator=gen)   # normalised pixel noise, fp32
        elif model_name == "llava":
            cfg = synth.llava_full_config(lora_rank=lora_rank)             # Mistral-7B decoder + CLIP-L, BT head, no SkipCA on this branch
            gb = synth.llava_synth_batch(cfg, 1234, [128] * (B * world), [(336, 336)] * (B * world), with_pixels=False)
            ncrop, flop = 3, 19.9e12                    # 3 crops -> 1176 image tokens; see DESIGN.md §8
            name = "LLaVA-v1.6-Mistral-7B (BASELINE configs[4] shapes, 16-bit operands), 3 crops/img, V=1176"
            sizes = torch.from_numpy(gb["image_sizes"][rows])
            pix = torch.randn(B, ncrop, 3, 336, 336, device="cuda", generator=gen)
        else:
            cfg = synth.full_config(lora_rank=lora_rank, **(dict(is_general_preference=True, value_head_dim=2) if gpm else {}))
            head = "GPM head d=2 + SkipCA, pairwise" if gpm else "BT head + SkipCA"
            if num_crops == 16:
                gb = synth.synth_batch(cfg, 1234, [128] * (B * world), (4, 4), with_pixels=False)
                ncrop, flop = 17, FLOP_PER_PAIR
                name = f"BASELINE configs[{2 if gpm else 1}]: Phi-3.5-V {head}, 17 crops/img, V=2509"
            else:
                gb = synth.synth_batch(cfg, 1234, [128] * (B * world), (2, 2), with_pixels=False)
                ncrop, flop = 5, 8.48e12
                name = f"BASELINE configs[1] at num_crops=4: Phi-3.5-V {head}, 5 crops/img, V=757"
            sizes = torch.from_numpy(gb["image_sizes"][rows])
            pix = torch.randn(B, ncrop, 3, 336, 336, device="cuda", generator=gen)      # CLIP-normalised pixel noise, fp32
        ids = torch.from_numpy(gb["input_ids"][rows]).cuda()
        mask = torch.from_numpy(gb["attention_mask"][rows]).cuda()
        return dict(model=model_name, cfg=cfg, B=B, S=ids.shape[1], ids=ids, mask=mask, pix=pix, sizes=sizes, ncrop=ncrop, flop=flop, name=name)

    def build_model(w, dtype, fp32_valued=False, profile=0):
        cfg, B, S = w["cfg"], w["B"], w["S"]
        # check_inputs: "eager" (the library's and load_reward_adaptor's default, hence the headline's since round 6: the reference's
        # exceptions, at the price of a stream drain per forward because the ids are device-resident here) or "deferred" (no host check:
        # the engine's own slot check marks a mismatching row NaN; reported as the `deferred_input_check` leg)
        kw = dict(operand_dtype=dtype, synth_profile=profile, calibrate=not a.profile_run, check_inputs=a.check_inputs)
        if w["model"] == "qwen":
            m = RewardModel(cfg, synth_seed=1234, max_batch=B, max_seq=max(S, GOLDEN_MAX_SEQ.get("qwen", 0), PROBE_MIN_SEQ["qwen"]), max_patches=max(B * 32 * 32, 2048), **kw)
        else:
            # (max_seq admits every tier of the operand-form probe -- probe.py -- so the form this engine locks is the one the golden tests'
            #  engines and any full-size deployment of the same weights lock: capacity only sizes the workspace, it is not in the timed step)
            m = RewardModel(cfg, synth_seed=1234, max_batch=B, max_seq=max(S, GOLDEN_MAX_SEQ.get(w["model"], 0), PROBE_MIN_SEQ[w["model"]] if (w["model"] != "phi3v" or w["ncrop"] >= 17) else 0),
                            max_crops=max(w["ncrop"], 5 if w["model"] == "llava" else 17), **kw)
        m.synth_fp32_valued = fp32_valued
        torch.cuda.synchronize()
        t_to = time.perf_counter()
        m = m.to(f"cuda:{local}").eval()
        torch.cuda.synchronize()
        m.to_cuda_seconds = time.perf_counter() - t_to        # engine build + weight synthesis + the operand-form probe
        if a.tile >= 0:
            m.engine.set_gemm_tile(a.tile)
        return m

    def forward(m, w, pix=None):
        """One scoring pass through the DROP-IN API (model.custom_forward, rw_model_general_preference.py:334), wrapper checks included."""
        pix = w["pix"] if pix is None else pix
        if w["model"] == "phi3v":
            return m.custom_forward(w["ids"], w["mask"], pix, w["sizes"])[0]
        key = "image_grid_thw" if w["model"] == "qwen" else "image_sizes"
        return m.custom_forward(inputs_batch={"input_ids": w["ids"], "attention_mask": w["mask"], "pixel_values": pix, key: w["sizes"]})[0]

    def timed_steps(fn, warmup, steps):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    def release(m):
        m.engine.close()
        torch.cuda.empty_cache()

    B = a.batch if a.batch is not None else (64 if pairwise else 32)
    w = workload(a.model, B, a.num_crops, gpm=pairwise, lora_rank=a.lora_rank)
    cfg, S, flop_per_pair = w["cfg"], w["S"], w["flop"]
    precise = "x2" in a.dtype
    main_profile = synth.PROFILE_NAMES[a.profile_weights]
    model = build_model(w, a.dtype, profile=main_profile)
    if pairwise:       # the rejected image of every pair: other pixels, same caption
        pix_r = torch.randn(w["pix"].shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4321 + rank))
    pargs = type("A", (), dict(is_general_preference=cfg.is_general_preference, value_head_dim=cfg.value_head_dim,
                               general_preference_tau=cfg.general_preference_tau))

    def step():
        r = forward(model, w)
        if pairwise:       # eval/batch_inference_rm_phi.py:92-108: chosen forward, rejected forward, preference probability
            rr = forward(model, w, pix_r)
            if world > 1:
                r, rr = (gather_rewards(r), gather_rewards(rr)) if a.backend == "nccl" else (gather_rewards(r.cpu()), gather_rewards(rr.cpu()))
            return torch.from_numpy(preference_compute(pargs, r, rr))
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
    multi = None
    if world > 1:
        cdev = "cuda" if a.backend == "nccl" else "cpu"
        dt_own = dt
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # what lets a SCALE record confirm the collective saw N ranks on N devices: every rank's step time and device identity
        # all-gathered, and the reward all-gather itself timed (events on the stream it runs on; host clock under gloo)
        per = [torch.zeros(1, device=cdev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(per, torch.tensor([1e3 * dt_own / a.steps], device=cdev, dtype=torch.float64))
        prop = torch.cuda.get_device_properties(local)
        uid = str(getattr(prop, "uuid", "")) or f"{prop.name}:{local}"
        ident = [None] * world
        dist.all_gather_object(ident, (local, uid))
        probe_r = torch.zeros(B, cfg.value_head_dim, device="cuda")
        gather_rewards(probe_r if a.backend == "nccl" else probe_r.cpu())
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(20):
            gather_rewards(probe_r if a.backend == "nccl" else probe_r.cpu())
        torch.cuda.synchronize()
        multi = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "devices": [i[0] for i in ident],
                 "distinct_devices": len({i[1] + ":" + str(i[0]) for i in ident}) if a.all_ranks_on_device < 0 else 1,
                 "per_rank_ms": [round(float(p.item()), 3) for p in per],
                 "collective_us": round(1e6 * (time.perf_counter() - t1) / 20, 1), "gathered_rows": int(out.shape[0]) if out.dim() else None}
    assert torch.isfinite(out).all(), "non-finite rewards"

    if rank == 0:
        rows_per_step = world * B * (2 if pairwise else 1)
        value = rows_per_step * a.steps / dt
        tf_per_gpu = value * flop_per_pair / world / 1e12
        res = {
            "metric": "reward-pairs/sec (336px img, 128-tok caption) " + {"phi3v": "Phi-3.5-V", "llava": "LLaVA-v1.6-Mistral-7B", "qwen": "Qwen2.5-VL-7B"}[a.model], "value": value, "unit": "reward-pairs/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if a.dtype.startswith("f16") else "fp8" if a.dtype == "fp8" else "bf16", "operand_mode": a.dtype,
            "dtype_note": f"{a.dtype} MFMA operands (bf16-valued weights), f32 accumulate/residual/softmax",
            "operand_form": dict(model.form_info) if model.form_info else None,
            "probe_seconds": getattr(model, "to_cuda_seconds", None), "multi_gpu": multi,
            "data": "synthetic (seeded weights and inputs; no checkpoint offline)",
            "hbm_bytes": {"workspace": model.engine.workspace_bytes(), "note": "activation workspace sized for (rows_per_gpu, seq_len) at lr_finalize; weights "
                          "(operand copies + residual / e4m3 twins in the split-operand modes) are extra"},
            "config": {"workload": w["name"] + ", S=%d" % S + (", un-merged LoRA adapter r=%d on the decoder linears" % a.lora_rank if a.lora_rank else ""),
                       "rows_per_gpu": B, "global_batch": B * world, "seq_len": S, "parallelism": f"dp{world}",
                       "collective": "all_gather rewards [B,%d] fp32" % cfg.value_head_dim if world > 1 else "none",
                       "input_check": a.check_inputs},
            "roofline": {"bound": "mfma", "achieved": tf_per_gpu, "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf_per_gpu / PEAK_TFLOPS, "traffic": None,
                         "note": "whole pass: pairs/s x %.2f TFLOP ALGORITHMIC per pair, per GPU%s" % (
                             flop_per_pair / 1e12, ("; the split-operand mode executes 1.5x (linears: f16 pass + e4m3 residual pass at twice the rate) / 3x (attention) that MFMA time"
                              if a.dtype == "f16x2f8" else "; the split-operand mode executes 2x (linears) / 3x (attention) that MFMA work") if precise else "")},
        }
        if pairwise:
            res["preference_pairs_per_sec"] = world * B * a.steps / dt
            res["config"]["step"] = "chosen forward + rejected forward (B rows each) + preference_compute (GPM d=2 formula)"
        headline = a.model == "phi3v" and a.num_crops == 16 and not pairwise and not a.lora_rank
        if world == 1:
            full = (a.model != "phi3v" or a.num_crops == 16) and not a.lora_rank      # the golden rows: 17-crop images, no adapter
            res["parity_check"] = golden_check(model, a.model, current=(1234, main_profile)) if (full and not a.quick and not a.no_golden) else None
            if a.model == "phi3v" and a.num_crops == 16 and not a.profile_run:
                dk = dominant_kernel_probe(L.LR_DT_F16 if a.dtype.startswith("f16") else L.LR_DT_BF16, a.tile, split=precise, lo8=a.dtype == "f16x2f8")
                # the roofline object proper: the dominant kernel, algorithmic FLOPs per launch / its live HIP-event duration
                res["roofline"].update({"achieved": dk["tflops"], "frac": dk["tflops"] / PEAK_TFLOPS, "traffic": dk["traffic"],
                                        "kernel": dk["kernel"], "dominant_kernel": dk,
                                        "whole_pass": {"achieved": tf_per_gpu, "frac": tf_per_gpu / PEAK_TFLOPS},
                                        "note": res["roofline"]["note"] + "; achieved/frac = the dominant kernel (algorithmic FLOPs per launch / HIP-event "
                                                "time, measured by this run); traffic / mfma_busy / clock_ghz = PMC figures of the committed rocprofv3 summary, null "
                                                "unless its recorded kernel sources are this build's; whole_pass = the same ratio for the step"})
        if world == 1 and not a.quick:
            # the same step with the pixel hand-over included: fp32 pixels start in pinned host memory (what the processor
            # returns) and cross PCIe on the forward's stream every step.  Reported beside `value`, never as `value`.
            sub_steps = min(a.steps, 3)
            pix_host = w["pix"].cpu().pin_memory()
            def step_h2d():
                w["pix"].copy_(pix_host, non_blocking=True)
                return forward(model, w)
            ms_h2d = timed_steps(step_h2d, 1, sub_steps)
            res["h2d_inclusive"] = {"value": B / (ms_h2d * 1e-3), "unit": "reward-pairs/sec", "ms_per_step": ms_h2d,
                                    "pixel_bytes_per_step": w["pix"].numel() * 4, "note": "pinned host fp32 pixels copied in every step (one forward of B rows)"}
            del pix_host
            if a.check_inputs == "eager":
                # the other input-check mode, same engine: no per-forward stream drain (RewardModel(check_inputs="deferred"))
                model.check_inputs = "deferred"
                ms_def = timed_steps(lambda: forward(model, w), 1, sub_steps)
                model.check_inputs = "eager"
                res["deferred_input_check"] = {"value": B / (ms_def * 1e-3), "unit": "reward-pairs/sec", "ms_per_step": ms_def, "vs_headline": (B / (ms_def * 1e-3)) / value,
                                               "note": "check_inputs='deferred': no host-side slot check, no stream drain per forward (the headline runs the library default, 'eager')"}
            if a.model == "phi3v":
                # the image hand-over in front of the path: decoded uint8 336 px image -> pixel_values rows
                from llava_reward_amd import preprocess
                img = torch.from_numpy(synth.synth_image(1234, "bench.image", 336, 336)).cuda()
                preprocess.hd_transform_batch([img] * B, a.num_crops, out=w["pix"])
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    preprocess.hd_transform_batch([img] * B, a.num_crops, out=w["pix"])
                torch.cuda.synchronize()
                us = 1e6 * (time.perf_counter() - t1) / (5 * B)
                res["input_handover"] = {"kernel": "lr_hd_transform (uint8 336x336 -> [%d,3,336,336] fp32, local crops bit-exact with Pillow)" % (w["ncrop"]),
                                         "us_per_image": us, "images_per_sec": 1e6 / us,
                                         "hbm_GBps_algorithmic": (w["ncrop"] * 3 * 336 * 336 * 4 + 336 * 336 * 3) / us / 1e3}
                if headline:
                    # files on disk -> rewards (the reference's loop, eval/batch_inference_rm_phi.py:58-94): PNG files are
                    # decoded by a thread pool, uploaded as uint8 and HD-transformed on a side stream while the previous batch is scored
                    import tempfile
                    from PIL import Image
                    from llava_reward_amd.scoring import PrefetchingBatcher
                    with tempfile.TemporaryDirectory() as td:
                        files = []
                        for i in range(2 * B):
                            fp = os.path.join(td, f"{i}.png")
                            Image.fromarray(synth.synth_image(1234, f"bench.file{i}", 336, 336, i % 2 == 0)).save(fp)
                            files.append(fp)
                        tok = synth.StandInTokenizer()
                        cap = "x" * 110            # one token per character with the stand-in tokenizer (23 template characters): S = 2642, the workload's
                        nb = sub_steps + 1
                        items = [(files[i % len(files)], cap) for i in range(nb * B)]
                        torch.cuda.synchronize()
                        with PrefetchingBatcher(items, tok, batch_size=B, num_crops=a.num_crops, device=f"cuda:{local}", depth=2, workers=8) as pb:
                            it = iter(pb)
                            outs = [model.custom_forward(**next(it))[0]]          # warm-up batch (the producer is already filling the queue)
                            torch.cuda.synchronize()
                            t1 = time.perf_counter()
                            for bt in it:
                                outs.append(model.custom_forward(**bt)[0])
                            torch.cuda.synchronize()
                            ms_e2e = 1e3 * (time.perf_counter() - t1) / (nb - 1)
                    assert all(torch.isfinite(o).all() for o in outs)
                    res["end_to_end"] = {"value": B / (ms_e2e * 1e-3), "unit": "reward-pairs/sec", "ms_per_step": ms_e2e, "vs_resident_inputs": (B / (ms_e2e * 1e-3)) / value,
                                         "what": f"{B} PNG files (336x336) per step from disk -> decode (8 threads) -> uint8 H2D -> lr_hd_transform on a side stream, "
                                                 "prefetched 2 batches ahead -> custom_forward"}
            release(model)
            del model

            def leg(wl, dtype, steps=sub_steps, fp32_valued=False, golden=None, profile=0):
                m = build_model(wl, dtype, fp32_valued, profile)
                ms = timed_steps(lambda: forward(m, wl), 1, steps)
                out = {"dtype": dtype, "value": wl["B"] / (ms * 1e-3), "unit": "reward-pairs/sec", "ms_per_step": ms, "rows_per_step": wl["B"],
                       "workspace_bytes": m.engine.workspace_bytes(),
                       "operand_form": dict(m.form_info) if m.form_info else None,
                       "roofline_frac_whole_pass": wl["B"] / (ms * 1e-3) * wl["flop"] / 1e12 / PEAK_TFLOPS}
                if golden and not a.no_golden:
                    out["parity_check"] = golden_check(m, wl["model"], current=(1234, profile))
                release(m)
                return out

            def golden_check_one(wl, dtype, golden_name, profile):
                """One golden row on a small engine of its own (B = 2): weights of that golden, scored through the bare sequence."""
                if a.no_golden:
                    return None
                g1 = [g for g in _goldens(wl["model"]) if g["name"] == golden_name]
                if not g1:
                    return None
                wl1 = dict(wl, B=2)
                m = build_model(wl1, dtype, profile=profile)
                out = golden_check(m, wl["model"], name=golden_name, current=(1234, profile))
                release(m)
                return out

            if headline and a.dtype == "f16x2f8":
                # What the headline costs when the loaded weights do NOT let the default form through.  `value` above is the default
                # form, which .to('cuda') kept because the benign N(0, 0.02) weight set passes its probe; outlier-bearing weights
                # (massive channels, large norm gains: what trained checkpoints show) make the same call lock the strict form.
                res["strict_form"] = dict(leg(w, "f16x2"), vs_headline=None,
                                          note="the strict form (16-bit residual passes everywhere): what the engine runs when its probe rejects the default "
                                               "form on the loaded weights")
                res["strict_form"]["vs_headline"] = res["strict_form"]["value"] / value
                # ... and the likely shape of a real LLaVA-Reward checkpoint: outlier-bearing weights AND the un-merged rank-128 adapter of the
                # training recipes on every decoder linear, scored in whatever form the probe locks on them
                wr = workload(a.model, B, a.num_crops, lora_rank=128)
                rc = leg(wr, a.dtype, profile=synth.PROFILE_OUTLIER)
                res["real_checkpoint"] = dict(rc, rank=128, weight_profile="outlier", vs_headline=rc["value"] / value,
                                              note="synth.PROFILE_OUTLIER weights + un-merged LoRA r=128, through the bare sequence (.to('cuda') -> "
                                                   "custom_forward); operand_form = what the probe locked")
                # latency of ONE row (BASELINE configs[0] shape, eval/simple_inference.py: B = 1, 17 crops, S = 2642)
                w1 = workload(a.model, 1, a.num_crops)
                m1 = build_model(w1, a.dtype)
                ms_sync = timed_steps(lambda: (forward(m1, w1), torch.cuda.synchronize()), 2, 10)
                ms_b2b = timed_steps(lambda: forward(m1, w1), 2, 10)
                res["latency_b1"] = {"ms_per_forward": ms_sync, "ms_per_forward_enqueued_back_to_back": ms_b2b, "rows": 1, "seq_len": w1["S"],
                                     "operand_form": dict(m1.form_info) if m1.form_info else None,
                                     "note": "one (caption, image) row per custom_forward call, synchronised after each call"}
                release(m1)
                del m1
            if headline and precise:
                # what a real LLaVA-Reward checkpoint costs: every decoder linear carries the un-merged rank-128 adapter of the training
                # recipes (scripts/run_train_rm_single_lora_phi.sh: --lora_rank 128 --lora_alpha 256, vision tower frozen)
                wl = workload(a.model, B, a.num_crops, lora_rank=128)
                res["lora_unmerged"] = dict(leg(wl, a.dtype), rank=128,
                                            note="un-merged adapter: t = x A^T per linear + 128 extra K columns (x2: hi, lo) in the base GEMM; base weights stay bf16-exact")
                res["lora_unmerged"]["vs_headline"] = res["lora_unmerged"]["value"] / value
                # the fallback for adapters the engine does not run un-merged (vision-tower adapters, merge_lora=True): merged weights are
                # not bf16-valued any more, every GEMM carries a third K segment
                res["merged_lora_weights"] = dict(leg(w, a.dtype, steps=2, fp32_valued=True),
                                                  note="fallback: merged (fp32-valued) weights, third K segment per GEMM")
                # the other reading of "336 px" (SURVEY.md §8d): num_crops = 4 -> 5 crops, V = 757, 8.48 TFLOP per pair
                w4 = workload("phi3v", B, 4)
                res["num_crops4"] = dict(leg(w4, a.dtype), workload=w4["name"] + ", S=%d" % w4["S"])
                # BASELINE configs[2]: GPM d=2 + SkipCA, pairwise, B=64 rows per forward
                wg = workload("phi3v", 64, 16, gpm=True)
                mg = build_model(wg, a.dtype)
                pg = torch.randn(wg["pix"].shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4321))
                ga = type("A", (), dict(is_general_preference=True, value_head_dim=2, general_preference_tau=wg["cfg"].general_preference_tau))
                msg = timed_steps(lambda: preference_compute(ga, forward(mg, wg), forward(mg, wg, pg)), 1, 2)
                res["gpm_pairwise"] = {"workload": wg["name"] + ", B=64 rows per forward", "dtype": a.dtype, "value": 128 / (msg * 1e-3), "unit": "reward-pairs/sec",
                                       "workspace_bytes": mg.engine.workspace_bytes(),
                                       "preference_pairs_per_sec": 64 / (msg * 1e-3), "ms_per_step": msg,
                                       "step": "chosen forward + rejected forward + preference_compute", "parity_check": golden_check(mg, "phi3v")}
                release(mg)
                del mg, pg
            if headline and not a.no_other_backbones:
                for mname, bb in (("qwen", 32), ("llava", 64)):          # BASELINE configs[3] / [4] at their per-GPU batch
                    wl = workload(mname, bb)
                    res[mname] = dict(leg(wl, a.dtype, golden=True), workload=wl["name"] + ", S=%d, B=%d" % (wl["S"], bb))
                    if precise:      # what a real checkpoint of this backbone costs: un-merged rank-128 adapters on q/k/v/o/gate/up/down
                        ll = leg(workload(mname, bb, lora_rank=128), a.dtype)
                        res[mname]["lora_unmerged"] = dict(ll, rank=128, vs_plain=ll["value"] / res[mname]["value"])
                    if mname == "llava" and a.dtype == "f16x2f8":
                        # BASELINE configs[4] names an "fp8 MFMA weight path".  (1) its parity definition: the reference run on the
                        # DE-QUANTISED weights of an e4m3-weight checkpoint (synth.PROFILE_E4M3) -- the default form scores that model
                        # (f16 main pass, residual pass on the e4m3 twins) and is checked against the reference's golden row;
                        e4 = leg(wl, a.dtype, profile=synth.PROFILE_E4M3)
                        pc = golden_check_one(wl, a.dtype, "ref_llava_full_e4m3_bt", synth.PROFILE_E4M3)
                        res[mname]["e4m3_valued_weights"] = dict(e4, parity_check=pc, note="default parity form on e4m3-VALUED weights (parity holds: "
                                                                 "the golden row is the reference on the de-quantised weights)")
                        # (2) the literal all-fp8 mode: every GEMM on e4m3 operands.  NOT a parity mode -- labelled with its distance to
                        # the reference on the same golden row.
                        w8 = leg(wl, "fp8", profile=synth.PROFILE_E4M3)
                        pc8 = golden_check_one(wl, "fp8", "ref_llava_full_e4m3_bt", synth.PROFILE_E4M3)
                        res[mname]["w8a8"] = dict(w8, parity="NOT A PARITY MODE", abs_err_vs_reference=pc8["abs_err"] if pc8 else None,
                                                  note="operand_dtype='fp8': e4m3 x e4m3 on every GEMM with K % 128 == 0 (DESIGN.md §11); the reward moves "
                                                       "by abs_err_vs_reference on the reference's golden row -- quantisation noise of a random-weight model")
            if precise and not a.no_fast_mode:
                # secondary figure: the single-pass f16 mode of the same workload.  NOT a parity mode (noise-limited):
                # it counts as a product number only where its live golden check says PASS.
                fm = leg(w, "f16", golden=(a.model != "phi3v" or a.num_crops == 16) and not a.lora_rank)
                pc = fm.get("parity_check")
                fm["parity"] = None if not pc else ("PASS" if pc["abs_err"] <= pc["tolerance"] else "FAIL")      # over EVERY golden scored
                fm["note"] = "single-pass f16 operands: NOT a parity mode and not a product number (noise-limited at full depth, DESIGN.md §4)"
                res["fast_mode"] = fm
            if headline and not a.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(cfg)
        emit(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
