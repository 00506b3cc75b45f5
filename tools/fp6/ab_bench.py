#!/usr/bin/env python3
"""Round 6, review item 1 (b): the SPEED of an FP6 residual pass, measured on a real K loop.

tools/fp6/lib_fp6ab.so (build_ab.sh, -DLR_FP6_AB=1) carries kernel form F8 == 3 of gemm_bt8_kernel: the product's split-operand GEMM
whose residual K-tiles are OCP MX FP6 (e2m3, 96-byte K-tile rows, one E8M0 scale per (row, 32 elements) for A_lo and for the weight
twin, v_mfma_scale_f32_16x16x128_f8f6f4 with cbsz = blgp = 2) instead of e4m3.  This script

  1. checks that K loop on REAL FP6 data against fp64 (random e2m3 codes and scales packed on the host in the kernel's layout), and
  2. times it interleaved with the product's e4m3 form (lr_op_gemm_bt_mixed of the same library) on the step's big shapes, with and
     without the epilogue.

    tools/fp6/build_ab.sh && python tools/fp6/ab_bench.py
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
os.environ.setdefault("LLAVA_REWARD_HIP_LIB", os.path.join(ROOT, "tools", "fp6", "lib_fp6ab.so"))
import numpy as np
import torch
from llava_reward_amd import _lib as L

lib = L.load()
fp6 = lib.lr_op_gemm_bt_fp6ab
fp6.restype = C.c_int
fp6.argtypes = [C.c_void_p] * 7 + [C.c_int] * 6 + [C.c_void_p]
st = torch.cuda.current_stream()
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
S = C.c_void_p(st.cuda_stream)


def e2m3_value(code):
    """OCP MX e2m3: 1 sign, 2 exponent (bias 1), 3 mantissa bits; sub-normals at exponent 0."""
    code = code.astype(np.int64)
    s, e, m = code >> 5, (code >> 3) & 3, code & 7
    v = np.where(e == 0, m / 8.0, np.exp2(e - 1.0) * (1.0 + m / 8.0))
    return np.where(s == 1, -v, v)


def pack_rows(codes):
    """codes [R, K] (6-bit) -> bytes [R, K / 128, 96]: per K-tile the four 32-element blocks q as 24 little-endian bytes each (element e
    at bits 6e .. 6e + 5), stored as [16-byte parts of q = 0..3 | 8-byte parts of q = 0..3] (gemm8.hip, F8 == 3)."""
    R, K = codes.shape
    c = codes.reshape(R, K // 128, 4, 32).astype(np.uint64)
    out = np.zeros((R, K // 128, 4, 24), dtype=np.uint8)
    for w in range(3):                                   # 64 bits at a time: elements 32w/3 ...  (192 bits = 3 x 64)
        word = np.zeros((R, K // 128, 4), dtype=np.uint64)
        for e in range(32):
            lo = 6 * e - 64 * w
            if lo <= -6 or lo >= 64:
                continue
            word |= (c[..., e] << np.uint64(lo)) if lo >= 0 else (c[..., e] >> np.uint64(-lo))
        out[..., 8 * w: 8 * w + 8] = word[..., None].view(np.uint8).reshape(R, K // 128, 4, 8)
    return np.concatenate([out[..., :16].reshape(R, K // 128, 64), out[..., 16:].reshape(R, K // 128, 32)], axis=-1)


def scale_slices_A(sa, M):
    """sa [M, K / 32] uint8 -> [K-tile][256-row tile] x 1 KB: byte (wr * 64 + lane) * 8 + half * 4 + i = scale of row half * 128 + wr * 64 + i * 16 + l15, block l4."""
    nk, Mt = sa.shape[1] // 4, (M + 255) // 256
    pad = np.full((Mt * 256, sa.shape[1]), 127, dtype=np.uint8)
    pad[:M] = sa
    x = pad.reshape(Mt, 2, 2, 4, 16, nk, 4)              # tile, half, wr, i, l15, ktile, q
    x = x.transpose(5, 0, 2, 6, 4, 1, 3)                 # ktile, tile, wr, q(l4), l15, half, i   (lane = l4 * 16 + l15)
    return np.ascontiguousarray(x).reshape(-1)


def scale_slices_W(sw, N):
    """sw [N, K / 32] -> [K-tile][256-column tile] x 1 KB: byte (wc * 64 + lane) * 4 + hb * 2 + j = scale of column wc * 64 + hb * 32 + j * 16 + l15, block l4."""
    nk, Nt = sw.shape[1] // 4, (N + 255) // 256
    pad = np.full((Nt * 256, sw.shape[1]), 127, dtype=np.uint8)
    pad[:N] = sw
    x = pad.reshape(Nt, 4, 2, 2, 16, nk, 4)              # tile, wc, hb, j, l15, ktile, q
    x = x.transpose(5, 0, 1, 6, 4, 2, 3)                 # ktile, tile, wc, q, l15, hb, j
    return np.ascontiguousarray(x).reshape(-1)


def check(M, N, K, seed):
    rng = np.random.default_rng(seed)
    a_hi = (rng.standard_normal((M, K)) * 0.5).astype(np.float16)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float16)
    ca, cw = rng.integers(0, 64, (M, K)), rng.integers(0, 64, (N, K))
    sa, sw = rng.integers(125, 130, (M, K // 32)).astype(np.uint8), rng.integers(125, 130, (N, K // 32)).astype(np.uint8)
    A = np.zeros((M, 2 * K), dtype=np.float16)
    A[:, :K] = a_hi
    A.view(np.uint8).reshape(M, 4 * K)[:, 2 * K: 2 * K + 96 * (K // 128)] = pack_rows(ca).reshape(M, -1)
    W6 = np.zeros((N, K), dtype=np.float16)
    W6.view(np.uint8).reshape(N, 2 * K)[:, : 96 * (K // 128)] = pack_rows(cw).reshape(N, -1)
    va = e2m3_value(ca) * np.exp2(np.repeat(sa.astype(np.float64), 32, axis=1) - 127)
    vw = e2m3_value(cw) * np.exp2(np.repeat(sw.astype(np.float64), 32, axis=1) - 127)
    ref = a_hi.astype(np.float64) @ w.astype(np.float64).T + va @ vw.T
    tA, tW, tW6 = torch.from_numpy(A).cuda(), torch.from_numpy(w).cuda(), torch.from_numpy(W6).cuda()
    tsa, tsw = torch.from_numpy(scale_slices_A(sa, M)).cuda(), torch.from_numpy(scale_slices_W(sw, N)).cuda()
    out = torch.zeros(M, N, device="cuda")
    rc = fp6(P(tA), P(tW), P(tW6), P(tsa), P(tsw), P(out), None, M, N, K, L.EPI_OUT_F32, 0, 0, S)
    assert rc == 0, lib.lr_last_error(None)
    torch.cuda.synchronize()
    err = np.abs(out.cpu().numpy().astype(np.float64) - ref).max() / np.abs(ref).max()
    lo_part = np.abs(va @ vw.T).max() / np.abs(ref).max()
    print(f"[fp6 K loop vs fp64] M={M} N={N} K={K}: max err / max|ref| = {err:.2e}   (the FP6 pass is {lo_part:.2f} of the result's scale)", flush=True)
    assert err < 2e-6, err


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def ab(name, M, N, K, epi, act=0, bias=False, reps=4, rounds=3):
    A = torch.randn(M, 2 * K, device="cuda").to(torch.float16)
    A[:, K:] *= 2.0 ** -11
    W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16)
    W8 = torch.zeros(N, K, device="cuda", dtype=torch.float16)
    nout = N // 2 if epi == L.EPI_SWIGLU_OP else N
    op_out = epi in (L.EPI_OUT_OP, L.EPI_SWIGLU_OP)
    out = torch.zeros(M, 2 * nout if op_out else nout, device="cuda", dtype=torch.float16 if op_out else torch.float32)
    b = torch.zeros(N, device="cuda") if bias else None
    ae = torch.full((lib.lr_op_lo8_scratch_bytes(M, K),), 127, dtype=torch.uint8, device="cuda")
    we = C.c_int(0)
    assert lib.lr_op_gemm_bt_mixed(P(A), P(W), P(W8), P(ae), P(out), P(b), M, N, K, epi, act, L.LR_DT_F16, 7, C.byref(we), S) == 0      # twin + in-place encode
    nkt = K // 128
    sa6 = torch.full((nkt * ((M + 255) // 256) * 1024,), 127, dtype=torch.uint8, device="cuda")
    sw6 = torch.full((nkt * ((N + 255) // 256) * 1024,), 127, dtype=torch.uint8, device="cuda")
    res = {}
    for noepi in (0, 1):
        f8 = lambda: lib.lr_op_gemm_bt_mixed(P(A), P(W), P(W8), P(ae), P(out), P(b), M, N, K, epi, act, L.LR_DT_F16, 16 if noepi else 0, C.byref(we), S)
        f6 = lambda: fp6(P(A), P(W), P(W8), P(sa6), P(sw6), P(out), P(b), M, N, K, epi, act, 16 if noepi else 0, S)
        t8, t6 = [], []
        for _ in range(rounds):
            t8.append(timed(f8, reps))
            t6.append(timed(f6, reps))
        res[noepi] = (min(t8), min(t6))
    units8, units6 = K // 64 + K // 128, K // 64 + K // 256
    print(f"{name:34s} {M}x{N}x{K}:  with epilogue  e4m3 {res[0][0]:7.3f} ms  fp6 {res[0][1]:7.3f} ms ({100 * (res[0][1] / res[0][0] - 1):+5.1f} %)   "
          f"K loop alone  e4m3 {res[1][0]:7.3f}  fp6 {res[1][1]:7.3f} ({100 * (res[1][1] / res[1][0] - 1):+5.1f} %)   [MFMA time units per tile: {units8} -> {units6}: {100 * (units6 / units8 - 1):+.1f} %]",
          flush=True)


if __name__ == "__main__":
    check(300, 256, 128, 1)
    check(1000, 512, 1024, 2)
    check(2100, 768, 3072, 3)
    ab("decoder gate_up + SwiGLU", 84544, 16384, 3072, L.EPI_SWIGLU_OP)
    ab("decoder down + residual add", 84544, 3072, 8192, L.EPI_RESADD_F32)
    ab("decoder o_proj + residual add", 84544, 3072, 3072, L.EPI_RESADD_F32)
    ab("CLIP fc1 + bias + quick_gelu", 313888, 4096, 1024, L.EPI_OUT_OP, act=L.ACT_QUICK_GELU, bias=True)
