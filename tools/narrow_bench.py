#!/usr/bin/env python3
"""Narrow GEMM tiles (csrc/gemm8.hip NW) against 256 x 256 ones on the shapes they exist for, default operand form (f16 pass + e4m3
residual pass), HIP-event timed, random data:  B = 1 decoder / CLIP GEMMs (M = 2642 / 9809), the gathered last layer (M = 32), the
adapters' t = x A^T (N = 128).  Also prints what the vendor library (torch.matmul -> hipBLASLt) reaches on the dominant GEMM's shape
in plain f16 -- a reference point for the K loop's wall, not a product path.
    python3 tools/narrow_bench.py [reps]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch
from llava_reward_amd import _lib as L

lib = L.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def mixed_case(M, N, K, epi):
    A = torch.cat([torch.randn(M, K, device="cuda").to(torch.float16), (torch.randn(M, K, device="cuda") * 2.0 ** -12).to(torch.float16)], dim=1).contiguous()
    W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16).to(torch.float16)
    W8 = torch.zeros_like(W)
    No = N // 2 if epi == L.EPI_SWIGLU_OP else N
    sc = torch.full((lib.lr_op_lo8_scratch_bytes(M, K) + lib.lr_op_lo8_scratch_bytes(M, max(No, 128)),), 127, dtype=torch.uint8, device="cuda")
    we = C.c_int(0)
    f32 = epi in (L.EPI_OUT_F32, L.EPI_RESADD_F32)
    out = torch.zeros(M, No if f32 else 2 * No, device="cuda", dtype=torch.float32 if f32 else torch.float16)
    fused = 32 if (not f32 and No % 128 == 0 and N > 128) else 0
    assert lib.lr_op_gemm_bt_mixed(P(A), P(W), P(W8), P(sc), P(out), None, M, N, K, epi, 0, L.LR_DT_F16, 7 | fused, C.byref(we), st()) == 0
    return lambda: lib.lr_op_gemm_bt_mixed(P(A), P(W), P(W8), P(sc), P(out), None, M, N, K, epi, 0, L.LR_DT_F16, fused, C.byref(we), st())


def main():
    CASES = [("decoder o_proj      B=1", 2642, 3072, 3072, L.EPI_RESADD_F32), ("decoder down        B=1", 2642, 3072, 8192, L.EPI_RESADD_F32),
             ("decoder qkv         B=1", 2642, 9216, 3072, L.EPI_OUT_OP), ("decoder gate_up     B=1", 2642, 16384, 3072, L.EPI_SWIGLU_OP),
             ("CLIP out-proj       B=1", 9809, 1024, 1024, L.EPI_RESADD_F32), ("CLIP fc2            B=1", 9809, 1024, 4096, L.EPI_RESADD_F32),
             ("CLIP qkv            B=1", 9809, 3072, 1024, L.EPI_OUT_OP), ("CLIP fc1            B=1", 9809, 4096, 1024, L.EPI_OUT_OP),
             ("decoder o_proj      B=2", 5284, 3072, 3072, L.EPI_RESADD_F32), ("decoder o_proj      B=4", 10568, 3072, 3072, L.EPI_RESADD_F32),
             ("gathered gate_up    M=32", 32, 16384, 3072, L.EPI_SWIGLU_OP), ("gathered down       M=32", 32, 3072, 8192, L.EPI_RESADD_F32),
             ("adapter t = x A^T   B=32", 84544, 128, 3072, L.EPI_OUT_OP), ("adapter t (down)    B=32", 84544, 128, 8192, L.EPI_OUT_OP),
             ("adapter t = x A^T   B=1", 2642, 128, 3072, L.EPI_OUT_OP)]
    print(f"{'GEMM':28s} {'M':>6s} {'N':>6s} {'K':>6s} {'256x256 us':>11s} {'128-row us':>11s} {'launcher us':>12s}")
    for name, M, N, K, epi in CASES:
        fn = mixed_case(M, N, K, epi)
        t = {}
        for env in ("0", "1", None):
            if env is None:
                os.environ.pop("LR_GEMM_NARROW", None)
            else:
                os.environ["LR_GEMM_NARROW"] = env
            t[env] = timed(fn)
        print(f"{name:28s} {M:6d} {N:6d} {K:6d} {t['0']:11.1f} {t['1']:11.1f} {t[None]:12.1f}")

    # vendor reference point: plain f16 GEMM of the dominant shape through torch.matmul (hipBLASLt), fp32 accumulate, f16 out
    M, N, K = 84544, 16384, 3072
    A = torch.randn(M, K, device="cuda").to(torch.float16)
    W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16)
    us = timed(lambda: torch.matmul(A, W.t()))
    print(f"vendor library, plain f16 {M} x {N} x {K}: {us / 1e3:.2f} ms = {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s")
    out = torch.zeros(M, N // 2, device="cuda", dtype=torch.float16)
    us2 = timed(lambda: lib.lr_op_gemm_bt(P(A), P(W), P(out), None, M, N, K, K, K, N // 2, L.EPI_SWIGLU_OP, 0, L.LR_DT_F16, 6, st()))
    print(f"this kernel, single-pass f16 + SwiGLU, same shape: {us2 / 1e3:.2f} ms = {2.0 * M * N * K / us2 / 1e6:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
