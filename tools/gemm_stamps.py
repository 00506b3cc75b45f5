#!/usr/bin/env python3
"""Where do the cycles of a K-tile phase go?  Runs the stamped diagnostic build of gemm_bt8 (variant 9)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch
from llava_reward_amd import _lib as L
lib = L.load()
VARIANT = int(sys.argv[1]) if len(sys.argv) > 1 else 13     # 13 = stamped build of the product (super-phase) schedule; 9 = 4-phase form
M, N, K = 84544, 16384, 3072
A = torch.randn(M, K, device="cuda").to(torch.float16)
W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16)
out = torch.zeros(M, N, device="cuda", dtype=torch.float32)
dbg = torch.zeros(8 * 8 * 16 + 8 * 8 * 4, device="cuda", dtype=torch.int32)
st = torch.cuda.current_stream()
for _ in range(2):
    lib.lr_op_gemm_bt(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(dbg.data_ptr()), M, N, K, K, K, N,
                      L.EPI_OUT_F32, 0, L.LR_DT_F16, VARIANT, C.c_void_p(st.cuda_stream))
torch.cuda.synchronize()
raw = dbg.cpu().numpy()
d = raw[: 8 * 8 * 16].reshape(8, 8, 4, 4).astype(float) / (K // 64)
l = raw[8 * 8 * 16:].reshape(8, 8, 4).astype(float) / (K // 64)
print('cost of one stamp (s_memtime + lgkmcnt(0)), included once in every segment below: %.0f cycles' % l.mean(axis=(0, 1))[0])
names = ["LOAD", "BAR1", "COMP", "BAR2"]
for grp, waves in (("group0 (waves 0-3)", [0, 1, 2, 3]), ("group1 (waves 4-7)", [4, 5, 6, 7])):
    m = d[:, waves].mean(axis=(0, 1))
    print(grp)
    for ph in range(2 if VARIANT == 13 else 4):
        print("   %s %d: " % ("super-phase" if VARIANT == 13 else "phase", ph) + "  ".join(f"{names[k]} {m[ph, k]:6.0f}" for k in range(4)) + f"   sum {m[ph].sum():6.0f}")
    print(f"   per K-tile: {m.sum():.0f} cycles (MFMA issue floor per wave: 4 x 256 = 1024)")
