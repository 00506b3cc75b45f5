#!/usr/bin/env python3
"""A/B of the one-wave-per-SIMD GEMM main loop (tools/gemm4w/gemm4w.hip) against the product kernel (csrc/gemm8.hip) in ONE process on
one box (cdna_hip_programming.md §5.4 rule 24): correctness of the experiment against torch first, then HIP-event timings, random
f16 data, interleaved rounds, at the decoder's shapes (M = 330 row tiles: the experiment has no row clamps)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch  # noqa: E402
from llava_reward_amd import _lib as L  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libgemm4w.so")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "gemm4w.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
                           os.path.join(HERE, "gemm4w.hip"), "-o", SO])
g4 = C.CDLL(SO)
g4.g4_launch.restype = C.c_int
g4.g4_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib = L.load()
st = torch.cuda.current_stream()


def P(t):
    return C.c_void_p(t.data_ptr())


def check(M, N, K, sched):
    A = torch.randn(M, K, device="cuda").to(torch.float16)
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.float16)
    out = torch.zeros(M, N, device="cuda")
    assert g4.g4_launch(P(A), P(W), P(out), M, N, K, 0, sched, C.c_void_p(st.cuda_stream), C.c_void_p(0)) == 0
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t()
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    print(f"gemm4w check sched={sched} M={M} N={N} K={K}: max rel err {err:.2e}", flush=True)
    assert err < 2e-5, err
    # asymmetric operands, A = I-like block: catches swapped row / column maps
    A2 = torch.zeros(M, K, device="cuda", dtype=torch.float16)
    idx = torch.arange(min(M, K), device="cuda")
    A2[idx, idx] = 1.0
    W2 = (torch.arange(N, device="cuda").float()[:, None] * 0.001 + torch.arange(K, device="cuda").float()[None, :] * 0.01).to(torch.float16)
    out.zero_()
    assert g4.g4_launch(P(A2), P(W2), P(out), M, N, K, 0, sched, C.c_void_p(st.cuda_stream), C.c_void_p(0)) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, A2.float() @ W2.float().t())


def timeit(fn, reps):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    for sched in (0, 1, 2):
        for shp in ((256, 256, 128), (512, 768, 192), (1024, 512, 3072), (2304, 1280, 1024), (4096, 4096, 4096)):
            check(*shp, sched)
    reps = 5
    for name, M, N, K in (("dec.gate_up", 84480, 16384, 3072), ("dec.qkv", 84480, 9216, 3072), ("dec.down", 84480, 3072, 8192),
                          ("clip.fc1", 313856, 4096, 1024), ("sq8k", 8192, 8192, 8192)):
        A = torch.randn(M, K, device="cuda").to(torch.float16)
        W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16)
        out = torch.zeros(M, N, device="cuda")
        s = C.c_void_p(st.cuda_stream)
        variants = {
            "product, fp32-out epilogue": lambda: lib.lr_op_gemm_bt(P(A), P(W), P(out), C.c_void_p(0), M, N, K, K, K, N, L.EPI_OUT_F32, 0, L.LR_DT_F16, 6, s),
            "product, no epilogue": lambda: lib.lr_op_gemm_bt(P(A), P(W), P(out), C.c_void_p(0), M, N, K, K, K, N, L.EPI_OUT_F32, 0, L.LR_DT_F16, 12, s),
            "4w s0 no-epi": lambda: g4.g4_launch(P(A), P(W), P(out), M, N, K, 1, 0, s, C.c_void_p(0)),
            "4w s1 no-epi": lambda: g4.g4_launch(P(A), P(W), P(out), M, N, K, 1, 1, s, C.c_void_p(0)),
            "4w s2 (lean DMA) no-epi": lambda: g4.g4_launch(P(A), P(W), P(out), M, N, K, 1, 2, s, C.c_void_p(0)),
            "4w s0 no-epi resident": lambda: g4.g4_launch(P(A), P(W), P(out), M, N, K, 3, 0, s, C.c_void_p(0)),
            "4w s1 no-epi resident": lambda: g4.g4_launch(P(A), P(W), P(out), M, N, K, 3, 1, s, C.c_void_p(0)),
            "4w s1 direct stores": lambda: g4.g4_launch(P(A), P(W), P(out), M, N, K, 0, 1, s, C.c_void_p(0)),
        }
        best = {k: 1e9 for k in variants}
        for _ in range(3):
            for k, fn in variants.items():
                best[k] = min(best[k], timeit(fn, reps))
        line = f"{name:12s} M={M} N={N} K={K}: "
        for k, ms in best.items():
            line += f"| {k}: {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.0f} TF "
        print(line, flush=True)
        # in-kernel clock and MFMA issue share of the K loop (a diagnostic launch after >= 1 s of back-to-back launches on this data)
        for sched in (0, 1, 2):
            stamps = torch.zeros(3 * 256, dtype=torch.int64, device="cuda")
            for _ in range(max(8, int(1000 / best["4w s1 no-epi"]))):
                g4.g4_launch(P(A), P(W), P(out), M, N, K, 1, sched, s, C.c_void_p(0))
            g4.g4_launch(P(A), P(W), P(out), M, N, K, 1, sched, s, P(stamps))
            torch.cuda.synchronize()
            t = stamps.view(256, 3).double().cpu()
            t = t[t[:, 2] > 0]
            ghz = (t[:, 0] / t[:, 1] * 0.1).median().item()
            cyc_per_kt = (t[:, 0] / t[:, 2]).median().item()
            print(f"    4w s{sched}: in-kernel clock {ghz:.2f} GHz, {cyc_per_kt:.0f} shader cycles per K-tile in the K loop (128 MFMAs x 16 = 2048): MFMA issue {2048 / cyc_per_kt:.1%} of the time", flush=True)
        del A, W, out


if __name__ == "__main__":
    main()
