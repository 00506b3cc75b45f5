#!/usr/bin/env python3
"""Launch the dominant GEMM (gate_up + SwiGLU, B=32 shapes) a few times; used under rocprofv3 --pmc."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch
from llava_reward_amd import _lib as L
lib = L.load()
M, N, K = 84544, 16384, 3072
A = torch.randn(M, K, device="cuda").to(torch.float16)
W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16)
out = torch.zeros(M, N // 2, device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    lib.lr_op_gemm_bt(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(0), M, N, K, K, K, N // 2,
                      L.EPI_SWIGLU_OP, 0, L.LR_DT_F16, 6, C.c_void_p(st.cuda_stream))
torch.cuda.synchronize()
print("algorithmic bytes per launch (A + W + out): %.3f GB" % ((M * K + N * K + M * N // 2) * 2 / 1e9))
