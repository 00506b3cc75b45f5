#!/usr/bin/env python3
"""W8A8 GEMM path: quantiser bytes against torch.float8_e4m3fn, GEMM against the dequantised fp32 product, rate against the f16 GEMM."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch
from llava_reward_amd import _lib as L
lib = L.load()
st = torch.cuda.current_stream()
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
S = C.c_void_p(st.cuda_stream)

def quant(x):
    M, K = x.shape
    q = torch.empty(M, K, dtype=torch.uint8, device="cuda"); s = torch.empty(M, dtype=torch.float32, device="cuda")
    assert lib.lr_op_quantize_rows_fp8(P(x), M, K, K, P(q), P(s), L.LR_DT_F16, S) == 0, lib.lr_last_error(None)
    return q, s

def check(M, N, K):
    x = (torch.randn(M, K, device="cuda") * torch.rand(M, 1, device="cuda") * 3).to(torch.float16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16)
    xq, xs = quant(x); wq, ws = quant(w)
    torch.cuda.synchronize()
    xc = x.float().cpu()                                                   # reference on the CPU: IEEE division
    amax = xc.abs().amax(1); sref = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    qref = (xc / sref[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    print(f"M={M} N={N} K={K}: scale equal {torch.equal(xs.cpu(), sref)}  bytes equal {torch.equal(xq.cpu(), qref)}  mismatches {(xq.cpu() != qref).sum().item()}")
    out = torch.zeros(M, N, device="cuda")
    rc = lib.lr_op_gemm_fp8(P(xq), P(xs), P(wq), P(ws), P(out), None, M, N, K, N, L.EPI_OUT_F32, 0, L.LR_DT_F16, S)
    assert rc == 0, lib.lr_last_error(None)
    torch.cuda.synchronize()
    ref = (xq.view(torch.float8_e4m3fn).float() @ wq.view(torch.float8_e4m3fn).float().T) * xs[:, None] * ws[None, :]
    full = x.float() @ w.float().T
    print(f"   vs dequantised product: max rel {((out - ref).abs().max() / ref.abs().max()).item():.2e};  vs f16 operands: rel rms {((out - full).norm() / full.norm()).item():.3e}")

check(300, 256, 128)
check(1000, 512, 1024)
check(4100, 768, 3072)

M, N, K = 84544, 16384, 3072
x = torch.randn(M, K, device="cuda").to(torch.float16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16)
xq, xs = quant(x); wq, ws = quant(w)
o16 = torch.zeros(M, N // 2, device="cuda", dtype=torch.float16)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
f16 = t(lambda: lib.lr_op_gemm_bt(P(x), P(w), P(o16), None, M, N, K, K, K, N // 2, L.EPI_SWIGLU_OP, 0, L.LR_DT_F16, 6, S))
f8 = t(lambda: lib.lr_op_gemm_fp8(P(xq), P(xs), P(wq), P(ws), P(o16), None, M, N, K, N // 2, L.EPI_SWIGLU_OP, 0, L.LR_DT_F16, S))
qt = t(lambda: lib.lr_op_quantize_rows_fp8(P(x), M, K, K, P(xq), P(xs), L.LR_DT_F16, S))
fl = 2.0 * M * N * K
print(f"gate_up SwiGLU {M}x{N}x{K}: f16 {f16:.3f} ms ({fl / f16 / 1e9:.0f} TF/s)  fp8 {f8:.3f} ms ({fl / f8 / 1e9:.0f} TF/s)  quantise A {qt:.3f} ms ({M * K * 3 / qt / 1e6:.0f} GB/s)")
