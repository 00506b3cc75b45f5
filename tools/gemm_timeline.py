#!/usr/bin/env python3
"""Timeline of the persistent GEMM's tiles (diagnostic build DBG = 9 of the default-mode kernel): per workgroup and tile the 100 MHz
stamps of tile start, K-loop start, epilogue start and epilogue end.  Prints how long the phases take and how the epilogues of the
256 workgroups are spread in time.  usage: gemm_timeline.py [shape ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch  # noqa: E402
from llava_reward_amd import _lib as L  # noqa: E402

SHAPES = {"clip.qkv": (313888, 3072, 1024, L.EPI_OUT_OP), "clip.out": (313888, 1024, 1024, L.EPI_RESADD_F32), "dec.o": (84544, 3072, 3072, L.EPI_RESADD_F32),
          "dec.down": (84544, 3072, 8192, L.EPI_RESADD_F32), "clip.fc1": (313888, 4096, 1024, L.EPI_OUT_OP)}
lib = L.load()
st = torch.cuda.current_stream()
for name in (sys.argv[1:] or list(SHAPES)):
    M, N, K, epi = SHAPES[name]
    A = torch.cat([torch.randn(M, K, device="cuda").half(), (torch.randn(M, K, device="cuda") * 2.0 ** -12).half()], dim=1).contiguous()
    W = (torch.randn(N, K, device="cuda") * 0.02).half()
    W8 = torch.zeros_like(W)
    ae = torch.full((lib.lr_op_lo8_scratch_bytes(M, K),), 127, dtype=torch.uint8, device="cuda")
    we = C.c_int(0)
    op = epi == L.EPI_OUT_OP
    out = torch.zeros(M, 2 * N if op else N, device="cuda", dtype=torch.float16 if op else torch.float32)
    tl = torch.zeros(256 * 128 * 16, dtype=torch.int64, device="cuda")
    base = (C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(W8.data_ptr()), C.c_void_p(ae.data_ptr()), C.c_void_p(out.data_ptr()))
    assert lib.lr_op_gemm_bt_mixed(*base, C.c_void_p(0), M, N, K, epi, 0, L.LR_DT_F16, 7, C.byref(we), C.c_void_p(st.cuda_stream)) == 0
    for _ in range(2):
        tl.zero_()
        assert lib.lr_op_gemm_bt_mixed(*base, C.c_void_p(tl.data_ptr()), M, N, K, epi, 0, L.LR_DT_F16, 9 << 8, C.byref(we), C.c_void_p(st.cuda_stream)) == 0
    torch.cuda.synchronize()
    t = tl.cpu().numpy().reshape(256, 128, 16).astype(np.float64)
    ntile = (t[:, :, 0] > 0).sum(axis=1)
    n = int(ntile.min())
    t0 = t[:, 0, 0].min()
    t = (t[:, :n] - t0) / 100.0            # us
    pro, kl, epi_t = t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2]
    gap = t[:, 1:, 0] - t[:, :-1, 3]
    inner = [t[:, 1:, b] - t[:, 1:, a] for a, b in ((2, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 3))]
    print("   inside the epilogue (us): " + "  ".join(f"{nm} {v.mean():.2f}" for nm, v in zip(
        ("start->sync", "stage+sync", "read+store", "start->sync", "stage+sync", "read+store", "tail"), inner)))
    print(f"{name}: {n} full rounds; per tile (us, mean over workgroups and tiles 1..{n - 1}): prologue {pro[:, 1:].mean():.1f}  K loop {kl[:, 1:].mean():.1f}  "
          f"epilogue {epi_t[:, 1:].mean():.1f}  gap {gap.mean():.2f};  first tile: prologue {pro[:, 0].mean():.1f}")
    for r in (1, n // 2, n - 1):
        es, ee = t[:, r, 2], t[:, r, 3]
        print(f"   round {r}: epilogue starts spread over {es.max() - es.min():.1f} us (std {es.std():.1f}), ends over {ee.max() - ee.min():.1f} us; "
              f"by XCD start mean: " + " ".join(f"{es[x::8].mean() - es.mean():+.1f}" for x in range(8)))
        x0 = es[0::8]                      # the 32 workgroups of XCD 0, in slot order
        print(f"      XCD 0: starts spread {x0.max() - x0.min():.1f} us (std {x0.std():.1f}); groups of 8 slots: " + " ".join(f"{x0[g * 8:(g + 1) * 8].mean() - x0.mean():+.1f}" for g in range(4))
              + f"; epilogue duration in XCD 0: {epi_t[0::8, r].mean():.1f} us (min {epi_t[0::8, r].min():.1f}, max {epi_t[0::8, r].max():.1f})")
