#!/usr/bin/env python3
"""Split-operand GEMM with the e4m3 residual pass against fp64 on un-rounded operands, and its rate against the 16-bit split form."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch
from llava_reward_amd import _lib as L
lib = L.load()
st = torch.cuda.current_stream()
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
S = C.c_void_p(st.cuda_stream)

def split(x):
    hi = x.to(torch.float16); lo = (x - hi.float()).to(torch.float16)
    return hi, lo

def check(M, N, K, scale_rows=False):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a32 = torch.randn(M, K, device="cuda", generator=g) * 0.7
    if scale_rows: a32 = a32 * torch.exp2(torch.randint(-6, 7, (M, 1), device="cuda", generator=g).float())
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.float16)
    hi, lo = split(a32)
    a2 = torch.cat([hi, lo], dim=1).contiguous()
    ref = (a32.double() @ w.double().T)
    o = torch.zeros(M, N, device="cuda")
    a_split = a2.clone()
    assert lib.lr_op_gemm_bt_split(P(a_split), P(w), P(o), None, M, N, K, L.EPI_OUT_F32, 0, L.LR_DT_F16, 6, S) == 0
    torch.cuda.synchronize()
    e_split = ((o.double() - ref).abs().max() / ref.abs().max()).item()
    w8 = torch.zeros(N, K, device="cuda", dtype=torch.float16); ae = torch.full((lib.lr_op_lo8_scratch_bytes(M, K),), 127, dtype=torch.uint8, device="cuda")
    o2 = torch.zeros(M, N, device="cuda")
    a_mixed = a2.clone()
    we = C.c_int(0)
    rc = lib.lr_op_gemm_bt_mixed(P(a_mixed), P(w), P(w8), P(ae), P(o2), None, M, N, K, L.EPI_OUT_F32, 0, L.LR_DT_F16, 3, C.byref(we), S)
    assert rc == 0, lib.lr_last_error(None)
    torch.cuda.synchronize()
    e_mixed = ((o2.double() - ref).abs().max() / ref.abs().max()).item()
    o1 = torch.zeros(M, N, device="cuda")
    assert lib.lr_op_gemm_bt(P(hi.contiguous()), P(w), P(o1), None, M, N, K, K, K, N, L.EPI_OUT_F32, 0, L.LR_DT_F16, 6, S) == 0
    torch.cuda.synchronize()
    e_single = ((o1.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"M={M} N={N} K={K} rows-scaled={scale_rows}: max err / max|ref|: single-pass {e_single:.2e}  f16x2 {e_split:.2e}  mixed {e_mixed:.2e}")

check(300, 256, 128)
check(1000, 512, 1024, True)
check(4100, 768, 3072, True)
check(8192, 8192, 512)

M, N, K = 84544, 16384, 3072
x = torch.randn(M, K, device="cuda"); hi, lo = split(x)
a2 = torch.cat([hi, lo], dim=1).contiguous()
w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16)
out = torch.zeros(M, N, device="cuda", dtype=torch.float16)       # [hi | lo] of N/2 columns each
w8 = torch.zeros(N, K, device="cuda", dtype=torch.float16); ae = torch.full((lib.lr_op_lo8_scratch_bytes(M, K),), 127, dtype=torch.uint8, device="cuda")
def t(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ts = t(lambda: lib.lr_op_gemm_bt_split(P(a2), P(w), P(out), None, M, N, K, L.EPI_SWIGLU_OP, 0, L.LR_DT_F16, 6, S))
we = C.c_int(0)
assert lib.lr_op_gemm_bt_mixed(P(a2), P(w), P(w8), P(ae), P(out), None, M, N, K, L.EPI_SWIGLU_OP, 0, L.LR_DT_F16, 7, C.byref(we), S) == 0    # prepare W8, encode A
tm = t(lambda: lib.lr_op_gemm_bt_mixed(P(a2), P(w), P(w8), P(ae), P(out), None, M, N, K, L.EPI_SWIGLU_OP, 0, L.LR_DT_F16, 0, C.byref(we), S))
tq = t(lambda: lib.lr_op_gemm_bt_mixed(P(a2), P(w), P(w8), P(ae), P(out), None, M, N, K, L.EPI_SWIGLU_OP, 0, L.LR_DT_F16, 6, C.byref(we), S))
print(f"gate_up split-operand SwiGLU {M}x{N}x{K}: 16-bit residual pass {ts:.3f} ms   e4m3 residual pass {tm:.3f} ms + {tq:.3f} ms in-place encoder   ratio {ts / (tm + tq):.2f}")
