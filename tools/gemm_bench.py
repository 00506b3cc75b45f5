#!/usr/bin/env python3
"""Microbenchmark of gemm_bt over the scoring path's real shapes (B=32), random data, HIP-event timed.
Interleaves tile variants in one process (cdna_hip_programming.md §5.4 rule 24)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch  # noqa: E402
from llava_reward_amd import _lib as L  # noqa: E402

SHAPES = [  # name, M, N, K, epi, count per pass
    ("clip.qkv", 313888, 3072, 1024, L.EPI_OUT_OP, 23),
    ("clip.out", 313888, 1024, 1024, L.EPI_RESADD_F32, 23),
    ("clip.fc1", 313888, 4096, 1024, L.EPI_OUT_OP, 23),
    ("clip.fc2", 313888, 1024, 4096, L.EPI_RESADD_F32, 23),
    ("clip.patch", 313344, 1024, 640, L.EPI_OUT_F32, 1),
    ("proj.0", 80288, 3072, 4096, L.EPI_OUT_OP, 1),
    ("dec.qkv", 84544, 9216, 3072, L.EPI_OUT_F32, 32),
    ("dec.o", 84544, 3072, 3072, L.EPI_RESADD_F32, 32),
    ("dec.gate_up", 84544, 16384, 3072, L.EPI_SWIGLU_OP, 32),
    ("dec.down", 84544, 3072, 8192, L.EPI_RESADD_F32, 32),
]


def main():
    tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3").split(",")]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dt = sys.argv[3] if len(sys.argv) > 3 else "f16"
    code, tdt = (L.LR_DT_F16, torch.float16) if dt == "f16" else (L.LR_DT_BF16, torch.bfloat16)
    lib = L.load()
    st = torch.cuda.current_stream()
    tot = {t: 0.0 for t in tiles}
    for name, M, N, K, epi, cnt in SHAPES:
        A = torch.randn(M, K, device="cuda").to(tdt)
        W = (torch.randn(N, K, device="cuda") * 0.02).to(tdt)
        ldc = N // 2 if epi == L.EPI_SWIGLU_OP else N
        Cb = torch.zeros(M, ldc, device="cuda", dtype=tdt if epi in (L.EPI_OUT_OP, L.EPI_SWIGLU_OP) else torch.float32)
        res = {}
        for rnd in range(2):
            for t in tiles:
                args = (C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(Cb.data_ptr()), C.c_void_p(0), M, N, K, K, K, ldc,
                        epi, 0, code, t, C.c_void_p(st.cuda_stream))
                assert lib.lr_op_gemm_bt(*args) == 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(reps):
                    lib.lr_op_gemm_bt(*args)
                e1.record(st)
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                res[t] = min(res.get(t, 1e9), ms)
        line = f"{name:12s} M={M:6d} N={N:5d} K={K:4d} "
        for t in tiles:
            tf = 2.0 * M * N * K / (res[t] * 1e-3) / 1e12
            tot[t] += res[t] * cnt
            line += f"| t{t}: {res[t]:7.3f} ms {tf:6.0f} TF "
        print(line, flush=True)
        del A, W, Cb
    print("per-pass GEMM total (ms): " + "  ".join(f"t{t}: {tot[t]:.0f}" for t in tiles))


if __name__ == "__main__":
    main()
