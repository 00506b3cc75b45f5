#!/usr/bin/env python3
"""What does a tile cost outside its K loop?  Times the default-mode (f16 pass + e4m3 residual pass) GEMM of the path's big shapes
with its epilogue and as the no-epilogue diagnostic of the same K loop (lr_op_gemm_bt_mixed flags & 16), random data, HIP events,
interleaved in one process.  product - no-epilogue = the epilogue's cost per launch."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch  # noqa: E402
from llava_reward_amd import _lib as L  # noqa: E402

SHAPES = [("clip.qkv", 313888, 3072, 1024, L.EPI_OUT_OP, 23), ("clip.out", 313888, 1024, 1024, L.EPI_RESADD_F32, 23),
          ("clip.fc1", 313888, 4096, 1024, L.EPI_OUT_OP, 23), ("clip.fc2", 313888, 1024, 4096, L.EPI_RESADD_F32, 23),
          ("dec.o", 84544, 3072, 3072, L.EPI_RESADD_F32, 32), ("dec.gate_up", 84544, 16384, 3072, L.EPI_SWIGLU_OP, 32),
          ("dec.down", 84544, 3072, 8192, L.EPI_RESADD_F32, 32)]
lib = L.load()
st = torch.cuda.current_stream()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
FLAGS = (0, 16) + tuple(int(x) << 8 for x in sys.argv[2].split(',')) if len(sys.argv) > 2 else (0, 16)
tot = [0.0, 0.0]
for name, M, N, K, epi, cnt in SHAPES:
    A = torch.cat([torch.randn(M, K, device="cuda").half(), (torch.randn(M, K, device="cuda") * 2.0 ** -12).half()], dim=1).contiguous()
    W = (torch.randn(N, K, device="cuda") * 0.02).half()
    W8 = torch.zeros_like(W)
    ae = torch.full((lib.lr_op_lo8_scratch_bytes(M, K) + lib.lr_op_lo8_scratch_bytes(M, N),), 127, dtype=torch.uint8, device="cuda")
    we = C.c_int(0)
    op = epi in (L.EPI_OUT_OP, L.EPI_SWIGLU_OP)
    nout = N // 2 if epi == L.EPI_SWIGLU_OP else N
    out = torch.zeros(M, 2 * nout if op else nout, device="cuda", dtype=torch.float16 if op else torch.float32)
    base = (C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(W8.data_ptr()), C.c_void_p(ae.data_ptr()), C.c_void_p(out.data_ptr()),
            C.c_void_p(0), M, N, K, epi, 0, L.LR_DT_F16)
    assert lib.lr_op_gemm_bt_mixed(*base, 7, C.byref(we), C.c_void_p(st.cuda_stream)) == 0
    res = {}
    for rnd in range(2):
        fused = 32 if (op and os.environ.get("LR_PROBE_FUSED", "1") != "0") else 0      # operand outputs as the engine writes them: one-byte residuals
        for flags in (FLAGS if epi == L.EPI_RESADD_F32 else (fused, 16)):
            args = base + (flags, C.byref(we), C.c_void_p(st.cuda_stream))
            assert lib.lr_op_gemm_bt_mixed(*args) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                lib.lr_op_gemm_bt_mixed(*args)
            e1.record(st)
            torch.cuda.synchronize()
            res[0 if flags == 32 else flags] = min(res.get(0 if flags == 32 else flags, 1e9), e0.elapsed_time(e1) / reps)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    per_cu = tiles / 256.0
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:4d} | product {res[0]:7.3f} ms | no epilogue {res[16]:7.3f} ms | epilogue {res[0] - res[16]:6.3f} ms "
          f"= {100 * (res[0] - res[16]) / res[0]:4.1f} % = {1e3 * (res[0] - res[16]) / per_cu:5.1f} us per tile; K loop + prologue {1e3 * res[16] / per_cu:6.1f} us per tile "
          f"({K // 64 + K // 128} K-tile units)" + "".join(f" | dbg{f >> 8}: {res[f]:7.3f}" for f in FLAGS if f > 16 and f in res), flush=True)
    tot[0] += res[0] * cnt
    tot[1] += res[16] * cnt
    del A, W, W8, out
print(f"per pass: product {tot[0]:.0f} ms, no epilogue {tot[1]:.0f} ms")
