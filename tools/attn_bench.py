#!/usr/bin/env python3
"""Microbenchmark of attn_kernel at the scoring path's shapes (B=32), random data, HIP-event timed."""
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch  # noqa: E402
from llava_reward_amd import _lib as L  # noqa: E402


def run(lib, name, B, S, H, hd, causal, reps=5, split=False, Hkv=None):
    if split:
        return run_split(lib, name, B, S, H, hd, causal, reps, Hkv or H)
    D = H * hd
    qkv = torch.randn(B * S, 3 * D, device="cuda").to(torch.float16)
    out = torch.zeros(B * S, D, device="cuda", dtype=torch.float16)
    mask = torch.ones(B, S, dtype=torch.int64, device="cuda") if causal else None
    kmin = torch.zeros(B, dtype=torch.int32, device="cuda") if causal else None
    st = torch.cuda.current_stream()
    P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
    args = (P(qkv), P(qkv), P(qkv), P(out), P(mask), P(kmin), 3 * D, D, 0, D, 2 * D, B, S, H, hd, int(causal), 1, 1.0 / math.sqrt(hd),
            L.LR_DT_F16, C.c_void_p(st.cuda_stream))
    assert lib.lr_op_attention(*args) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        lib.lr_op_attention(*args)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 4.0 * B * H * S * S * hd * (0.5 if causal else 1.0)
    print(f"{name:10s} B={B} S={S} H={H} hd={hd} causal={causal}: {ms:7.3f} ms  {flops / (ms * 1e-3) / 1e12:6.0f} TF/s (algorithmic)", flush=True)


def run_split(lib, name, B, S, H, hd, causal, reps, Hkv):
    """Split-operand form (the parity modes): rows [q k v | q_lo k_lo v_lo], 3 MFMA passes per contraction, output [O_hi | O_lo]."""
    W = (H + 2 * Hkv) * hd
    qkv = torch.cat([torch.randn(B * S, W, device="cuda").half(), (torch.randn(B * S, W, device="cuda") * 2.0 ** -12).half()], dim=1).contiguous()
    out = torch.zeros(B * S, 2 * H * hd, device="cuda", dtype=torch.float16)
    mask = torch.ones(B, S, dtype=torch.int64, device="cuda") if causal else None
    kmin = torch.zeros(B, dtype=torch.int32, device="cuda") if causal else None
    st = torch.cuda.current_stream()
    P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
    args = (P(qkv), P(qkv), P(qkv), P(out), P(mask), P(kmin), 2 * W, 2 * H * hd, 0, H * hd, (H + Hkv) * hd, W, H * hd, B, S, H, hd, int(causal), H // Hkv,
            1.0 / math.sqrt(hd), L.LR_DT_F16, C.c_void_p(st.cuda_stream))
    assert lib.lr_op_attention_split(*args) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        lib.lr_op_attention_split(*args)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 4.0 * B * H * S * S * hd * (0.5 if causal else 1.0)
    print(f"{name:10s} B={B} S={S} H={H} hd={hd} causal={causal} split: {ms:7.3f} ms  {flops / (ms * 1e-3) / 1e12:6.0f} TF/s (algorithmic; 3x executed)", flush=True)


if __name__ == "__main__":
    lib = L.load(sys.argv[1]) if len(sys.argv) > 1 else L.load()       # (a library path: A/B against another build on the same box)
    run(lib, "phi", 32, 2642, 32, 96, True)
    run(lib, "clip", 544, 577, 16, 64, False)
    run(lib, "phi", 32, 2642, 32, 96, True, split=True)
    run(lib, "clip", 544, 577, 16, 64, False, split=True)
    run(lib, "llava", 64, 1313, 32, 128, True, split=True, Hkv=8)
