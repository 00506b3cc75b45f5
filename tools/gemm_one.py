#!/usr/bin/env python3
"""Launch the dominant GEMM (decoder gate_up + SwiGLU, B=32 shapes) a few times; used under rocprofv3 --pmc.
    python3 tools/gemm_one.py 3          single-pass operands (fast mode)
    python3 tools/gemm_one.py 3 split    split-operand form (strict parity mode): A = [A_hi | A_lo], output [hi | lo]
    python3 tools/gemm_one.py 3 mixed    the same with the e4m3 residual pass (default parity mode)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch
from llava_reward_amd import _lib as L
lib = L.load()
M, N, K = 84544, 16384, 3072
mixed = len(sys.argv) > 2 and sys.argv[2] == "mixed"
split = mixed or (len(sys.argv) > 2 and sys.argv[2] == "split")
w = 2 if split else 1
A = torch.randn(M, K, device="cuda").to(torch.float16)
if split:
    A = torch.cat([A, (torch.randn(M, K, device="cuda") * 2.0 ** -12).to(torch.float16)], dim=1).contiguous()
W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16)
out = torch.zeros(M, w * N // 2, device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream()
if mixed:
    W8 = torch.zeros(N, K, device="cuda", dtype=torch.float16); ae = torch.full((lib.lr_op_lo8_scratch_bytes(M, K) + lib.lr_op_lo8_scratch_bytes(M, N // 2),), 127, dtype=torch.uint8, device="cuda"); we = C.c_int(0)
    assert lib.lr_op_gemm_bt_mixed(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(W8.data_ptr()), C.c_void_p(ae.data_ptr()),
                                   C.c_void_p(out.data_ptr()), C.c_void_p(0), M, N, K, L.EPI_SWIGLU_OP, 0, L.LR_DT_F16, 7, C.byref(we), C.c_void_p(st.cuda_stream)) == 0
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    if mixed:
        lib.lr_op_gemm_bt_mixed(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(W8.data_ptr()), C.c_void_p(ae.data_ptr()),
                                C.c_void_p(out.data_ptr()), C.c_void_p(0), M, N, K, L.EPI_SWIGLU_OP, 0, L.LR_DT_F16, 32, C.byref(we), C.c_void_p(st.cuda_stream))      # 32: as the engine launches it
    elif split:
        lib.lr_op_gemm_bt_split(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(0), M, N, K,
                                L.EPI_SWIGLU_OP, 0, L.LR_DT_F16, 6, C.c_void_p(st.cuda_stream))
    else:
        lib.lr_op_gemm_bt(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(0), M, N, K, K, K, N // 2,
                          L.EPI_SWIGLU_OP, 0, L.LR_DT_F16, 6, C.c_void_p(st.cuda_stream))
torch.cuda.synchronize()
print("algorithmic bytes per launch (A + W + out): %.3f GB" % (((1.5 if mixed else w) * M * K + (1.5 if mixed else 1) * N * K + (1.5 if mixed else w) * M * N // 2) * 2 / 1e9))
