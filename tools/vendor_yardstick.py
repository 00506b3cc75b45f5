#!/usr/bin/env python3
"""A yardstick for the nominal MFMA peak, not a product path: the vendor library's plain f16 GEMM (torch.matmul -> hipBLASLt: one MFMA
pass, no epilogue, f16 out) on the path's big GEMM shapes, beside this library's kernel in its single-pass form (fused epilogue and all)
and in the default form (f16 pass + e4m3 residual pass = 1.5x the MFMA work), same box, HIP-event timed, random data.
    python3 tools/vendor_yardstick.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from llava_reward_amd import _lib as L
from narrow_bench import mixed_case, timed, P, st, lib

CASES = [("dec.gate_up", 84544, 16384, 3072, L.EPI_SWIGLU_OP), ("dec.qkv", 84544, 9216, 3072, L.EPI_OUT_OP), ("dec.o", 84544, 3072, 3072, L.EPI_RESADD_F32),
         ("dec.down", 84544, 3072, 8192, L.EPI_RESADD_F32), ("clip.qkv", 313888, 3072, 1024, L.EPI_OUT_OP), ("clip.fc1", 313888, 4096, 1024, L.EPI_OUT_OP),
         ("clip.fc2", 313888, 1024, 4096, L.EPI_RESADD_F32), ("llava.gate_up", 140800, 28672, 4096, L.EPI_SWIGLU_OP)]
print(f"{'GEMM':14s} {'vendor ms':>10s} {'PFLOP/s':>8s} | {'single-pass ms':>14s} {'PFLOP/s':>8s} {'of vendor':>9s} | {'default ms':>10s} {'x1.5 PFLOP/s':>12s} {'of vendor':>9s}")
for name, M, N, K, epi in CASES:
    A = torch.randn(M, K, device="cuda").to(torch.float16)
    W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16)
    tv = timed(lambda: torch.matmul(A, W.t())) / 1e3
    ldc = N // 2 if epi == L.EPI_SWIGLU_OP else N
    f32 = epi in (L.EPI_OUT_F32, L.EPI_RESADD_F32)
    out = torch.zeros(M, ldc, device="cuda", dtype=torch.float32 if f32 else torch.float16)
    ts = timed(lambda: lib.lr_op_gemm_bt(P(A), P(W), P(out), None, M, N, K, K, K, ldc, epi, 0, L.LR_DT_F16, 6, st())) / 1e3
    del out, A, W
    td = timed(mixed_case(M, N, K, epi)) / 1e3
    pf = lambda ms, f=1.0: f * 2.0 * M * N * K / ms / 1e12
    print(f"{name:14s} {tv:10.3f} {pf(tv):8.2f} | {ts:14.3f} {pf(ts):8.2f} {tv / ts:9.2f} | {td:10.3f} {pf(td, 1.5):12.2f} {1.5 * tv / td:9.2f}", flush=True)
