#!/usr/bin/env python3
"""Two builds of the attention kernel on the same inputs: outputs compared bit for bit, then timed interleaved.
    python3 tools/dbg/attn_equal.py tools/dbg/lib_prev.so
Shapes: the decoder's (hd 96, causal, left-padded rows), CLIP's (hd 64, 577 tokens), LLaVA's (hd 128, GQA), short / odd lengths."""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch
from llava_reward_amd import _lib as L

libs = [(os.path.basename(a), L.load(a)) for a in sys.argv[1:]] + [("product 3-D grid", L.load()), ("product no shift", L.load()), ("product plain loop", L.load()), ("product", L.load())]
ENV = {"product 3-D grid": ("LR_ATT_XCD_ORDER", "0"), "product no shift": ("LR_ATT_QSHIFT", "0"), "product plain loop": ("LR_ATT_PINGPONG", "0")}
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)


def case(name, B, S, H, hd, causal, Hkv=None, pads=None, reps=5):
    Hkv = Hkv or H
    W = (H + 2 * Hkv) * hd
    g = torch.Generator(device="cuda").manual_seed(S * 7 + hd)
    qkv = torch.cat([torch.randn(B * S, W, device="cuda", generator=g).half(), (torch.randn(B * S, W, device="cuda", generator=g) * 2.0 ** -12).half()], dim=1).contiguous()
    mask = kmin = None
    if causal:
        mask = torch.ones(B, S, dtype=torch.int64, device="cuda")
        kmin = torch.zeros(B, dtype=torch.int32, device="cuda")
        for b, p_ in enumerate(pads or []):
            mask[b, :p_] = 0
            kmin[b] = p_
    st = torch.cuda.current_stream()
    outs, t = [], [0.0] * len(libs)

    def args_for(out):
        return (P(qkv), P(qkv), P(qkv), P(out), P(mask), P(kmin), 2 * W, 2 * H * hd, 0, H * hd, (H + Hkv) * hd, W, H * hd, B, S, H, hd, int(causal), H // Hkv,
                1.0 / math.sqrt(hd), L.LR_DT_F16, C.c_void_p(st.cuda_stream))
    for n, lib in libs:
        out = torch.zeros(B * S, 2 * H * hd, device="cuda", dtype=torch.float16)
        if n in ENV:
            os.environ[ENV[n][0]] = ENV[n][1]
        assert lib.lr_op_attention_split(*args_for(out)) == 0
        for k, _ in ENV.values():
            os.environ.pop(k, None)
        torch.cuda.synchronize()
        outs.append(out)
    inner = max(1, int(20.0 / max(0.05, 1e-9 * B * H * S * S * hd / 50)))          # ~20 ms of launches per sample
    for r in range(reps):
        for i in ([*range(len(libs))] if r % 2 == 0 else [*range(len(libs))][::-1]):
            a = args_for(outs[i])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if libs[i][0] in ENV:
                os.environ[ENV[libs[i][0]][0]] = ENV[libs[i][0]][1]
            e0.record(st)
            for _ in range(inner):
                libs[i][1].lr_op_attention_split(*a)
            e1.record(st)
            for k, _ in ENV.values():
                os.environ.pop(k, None)
            torch.cuda.synchronize()
            t[i] += e0.elapsed_time(e1) / inner / reps
    valid = torch.ones(B * S, dtype=torch.bool, device="cuda")
    if causal and pads:
        for b, p_ in enumerate(pads):
            valid[b * S: b * S + p_] = False          # rows of padding positions are never read by the path
    eq = all(torch.equal(outs[0][valid], o[valid]) for o in outs[1:])
    tol = float(os.environ.get("ATTN_EQ_TOL", "0"))          # > 0: builds that change bits on purpose (e.g. -DLR_ATT_LAZY=1) -- max |diff| of the hi halves
    if not eq and tol > 0:
        Hh = H * hd
        md = max(float((outs[0][valid][:, :Hh].float() - o[valid][:, :Hh].float()).abs().max()) for o in outs[1:])
        print(f"  (not bit-identical: max |diff| of the hi halves = {md:.3e}, tolerance {tol:g})")
        eq = md <= tol
    print(f"{name:24s} B={B} S={S} H={H} hd={hd}: bit-identical={eq}  " + "  ".join(f"{n} {x:7.3f} ms ({(x / t[0] - 1) * 100:+.1f} %)" for (n, _), x in zip(libs, t)), flush=True)
    assert eq, name


case("decoder", 32, 2642, 32, 96, True)
case("decoder left-padded", 32, 2642, 32, 96, True, pads=[0, 13, 64, 100, 200, 777, 1, 63] * 4)
case("decoder S=2048", 16, 2048, 32, 96, True)
case("decoder S=2700", 32, 2700, 32, 96, True)
case("decoder S=1500", 32, 1500, 32, 96, True)
case("decoder S=1100", 32, 1100, 32, 96, True)
case("decoder S=700", 32, 700, 32, 96, True)
case("decoder S=600", 32, 600, 32, 96, True)
case("decoder S=800", 32, 800, 32, 96, True)
case("decoder S=960", 32, 960, 32, 96, True)
case("decoder S=1000", 32, 1000, 32, 96, True)
case("decoder S=890 no pads", 32, 890, 32, 96, True)
case("decoder S=1031", 8, 1031, 32, 96, True, pads=[5, 0, 300, 64, 65, 1000, 2, 511])
case("decoder short S=890", 32, 890, 32, 96, True, pads=[0, 7, 100, 333] * 8)
case("decoder tiny S=70", 4, 70, 32, 96, True, pads=[0, 3, 40, 69])
case("clip", 544, 577, 16, 64, False)
case("clip 257 tokens", 64, 257, 16, 64, False)
case("llava hd128 GQA", 64, 1313, 32, 128, True, Hkv=8, pads=[0, 100] * 32)
case("qwen-like hd128 S=700", 16, 700, 28, 128, True, Hkv=4, pads=[0, 17, 128, 300] * 4)
