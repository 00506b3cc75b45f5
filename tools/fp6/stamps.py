#!/usr/bin/env python3
"""Round 6: where do the cycles of a RESIDUAL K-tile go?  In-kernel s_memtime stamps (gemm_bt8_kernel DBG == 3) of the split-operand
gate_up K loop, summed separately over the 16-bit K-tiles and the residual K-tiles, for the product's e4m3 residual form and for the FP6
form of the A/B build (tools/fp6/build_ab.sh).  Per wave group and super-phase: LOAD (fragment reads + LDS-DMA issue + counted wait),
BAR1, COMPUTE (B reads + MFMAs), BAR2; a K-tile = 2 super-phases.
    tools/fp6/build_ab.sh && python tools/fp6/stamps.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
os.environ.setdefault("LLAVA_REWARD_HIP_LIB", os.path.join(ROOT, "tools", "fp6", "lib_fp6ab.so"))
import torch
from llava_reward_amd import _lib as L

lib = L.load()
fp6 = lib.lr_op_gemm_bt_fp6ab
fp6.restype = C.c_int
fp6.argtypes = [C.c_void_p] * 7 + [C.c_int] * 6 + [C.c_void_p]
st = torch.cuda.current_stream()
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
S = C.c_void_p(st.cuda_stream)
M, N, K = 84544, 16384, 3072
A = torch.randn(M, 2 * K, device="cuda").to(torch.float16)
W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16)
W8 = torch.zeros(N, K, device="cuda", dtype=torch.float16)
out = torch.zeros(M, N, device="cuda", dtype=torch.float32)
ae = torch.full((lib.lr_op_lo8_scratch_bytes(M, K),), 127, dtype=torch.uint8, device="cuda")
we = C.c_int(0)
assert lib.lr_op_gemm_bt_mixed(P(A), P(W), P(W8), P(ae), P(out), None, M, N, K, L.EPI_OUT_F32, 0, L.LR_DT_F16, 7, C.byref(we), S) == 0
sa6 = torch.full(((K // 128) * ((M + 255) // 256) * 1024,), 127, dtype=torch.uint8, device="cuda")
sw6 = torch.full(((K // 128) * ((N + 255) // 256) * 1024,), 127, dtype=torch.uint8, device="cuda")
names = ["LOAD", "BAR1", "COMP", "BAR2"]
for tag, flag, sa, sw in (("e4m3 residual K-tiles (product form)", 128, ae, sw6), ("FP6 residual K-tiles (A/B form)", 64, sa6, sw6)):
    dbg = torch.zeros(8 * 8 * 16 + 8 * 8 * 4, device="cuda", dtype=torch.int32)
    for _ in range(2):
        dbg.zero_()
        assert fp6(P(A), P(W), P(W8), P(sa), P(sw), P(out), P(dbg), M, N, K, L.EPI_OUT_F32, 0, flag, S) == 0
    torch.cuda.synchronize()
    raw = dbg.cpu().numpy()
    d = raw[: 8 * 8 * 16].reshape(8, 8, 4, 4).astype(float)
    stamp = raw[8 * 8 * 16:].reshape(8, 8, 4).astype(float).mean(axis=(0, 1))[0] / (K // 64 + K // 128)
    print(f"{tag}   (cost of one stamp, inside every segment: {stamp:.0f} cycles)")
    for grp, waves in (("group 0 (waves 0-3)", [0, 1, 2, 3]), ("group 1 (waves 4-7)", [4, 5, 6, 7])):
        m = d[:, waves].mean(axis=(0, 1))
        for kind, rows, ntiles in (("16-bit K-tile", (0, 1), K // 64), ("residual K-tile", (2, 3), K // 128)):
            per = m[list(rows)] / ntiles
            print(f"   {grp}  {kind:16s} " + "   ".join(f"sp{i}: " + " ".join(f"{names[k]} {per[i, k]:5.0f}" for k in range(4)) for i in range(2)) + f"   = {per.sum():6.0f} cycles per K-tile")
