#!/usr/bin/env python3
"""Kernel-level check of hot blocks: A with massive columns, default e4m3 residual form with / without hot blocks vs fp64."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "llava-reward_amd"), ROOT]
import torch
from llava_reward_amd import _lib as L
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
tdt = torch.float16
g = torch.Generator().manual_seed(3)
for (M, N, K, cols) in [(1000, 512, 1024, [5, 700]), (4100, 768, 3072, [279, 1500, 3000])]:
    A32 = (torch.randn(M, K, generator=g) * 0.03).cuda()
    for c in cols:
        A32[:, c] = (torch.randn(M, generator=g) * 30 + 900).cuda()
    W = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).to(tdt).cuda()
    hi = A32.to(tdt); lo = (A32 - hi.float()).to(tdt)
    ref = A32.double() @ W.double().t()
    # the small-signal part: without the massive columns
    mask = torch.ones(K, dtype=torch.bool); mask[cols] = False
    small = (A32[:, mask].double() @ W[:, mask].double().t()).abs().mean().item()
    A2 = torch.cat([hi, lo], dim=1).contiguous()
    res = {}
    for name, blocks in (("no hot blocks", []), ("hot blocks", sorted({c // 128 for c in cols}))):
        arr = (C.c_int * 4)(*(blocks + [0] * (4 - len(blocks))))
        assert lib.lr_op_set_hot_blocks(len(blocks), arr) == 0
        W8 = torch.zeros_like(W); scr = torch.full((lib.lr_op_lo8_scratch_bytes(M, K),), 127, dtype=torch.uint8, device="cuda")
        we = C.c_int(0); out = torch.empty(M, N, device="cuda", dtype=torch.float32); Aw = A2.clone()
        assert lib.lr_op_gemm_bt_mixed(P(Aw), P(W), P(W8), P(scr), P(out), None, M, N, K, L.EPI_OUT_F32, 0, L.LR_DT_F16, 3, C.byref(we), st) == 0
        torch.cuda.synchronize()
        res[name] = (out.double() - ref).abs().mean().item()
    lib.lr_op_set_hot_blocks(0, None)
    sp = torch.empty(M, N, device="cuda", dtype=torch.float32)
    assert lib.lr_op_gemm_bt_split(P(A2), P(W), P(sp), None, M, N, K, L.EPI_OUT_F32, 0, L.LR_DT_F16, 6, st) == 0
    torch.cuda.synchronize()
    res["strict split"] = (sp.double() - ref).abs().mean().item()
    print(f"M={M} N={N} K={K} massive cols {cols}: mean |small-signal part| {small:.3e}; mean abs err: " + ", ".join(f"{k} {v:.2e}" for k, v in res.items()))
