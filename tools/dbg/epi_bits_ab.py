#!/usr/bin/env python3
"""Two builds of the library, every GEMM epilogue of the single-pass and split-operand entry points, outputs compared bit for bit
(edge tiles and inside tiles, with and without a bias).   python tools/dbg/epi_bits_ab.py <old .so> <new .so>"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch
from llava_reward_amd import _lib as L


def _load(path):
    lib = C.CDLL(os.path.abspath(path))
    for name in ("lr_op_gemm_bt", "lr_op_gemm_bt_split", "lr_op_gemm_rope"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = L._SIGS[name]
    return lib


libs = [_load(p) for p in sys.argv[1:3]]
st = torch.cuda.current_stream()
S = C.c_void_p(st.cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
bad = 0
ncmp = 0
for (M, N, K) in ((100, 96, 64), (300, 192, 128), (700, 768, 256), (1100, 3072, 192), (1500, 1152, 384), (2100, 1152, 384), (640, 1024, 384), (1500, 384, 512), (2600, 1024, 384)):
    for tile in (-1, 2):
        for epi, name in ((L.EPI_OUT_F32, "out_f32"), (L.EPI_RESADD_F32, "resadd"), (L.EPI_OUT_OP, "out_op"), (L.EPI_SWIGLU_OP, "swiglu")):
            for use_bias in (False, True):
                for split in (0, 1):
                    torch.manual_seed(M + N + K)
                    A = torch.randn(M, (2 if split else 1) * K, device="cuda").half()
                    W = (torch.randn(N, K, device="cuda") * 0.05).half()
                    bias = torch.randn(N, device="cuda") if use_bias else None
                    nout = N // 2 if epi == L.EPI_SWIGLU_OP else N
                    op = epi in (L.EPI_OUT_OP, L.EPI_SWIGLU_OP)
                    outs = []
                    for lib in libs:
                        o = torch.randn(M, (2 if (op and split) else 1) * nout, device="cuda")
                        o = o.half() if op else o
                        torch.manual_seed(1)
                        if epi == L.EPI_RESADD_F32:
                            o = torch.randn(M, nout, device="cuda")
                        if split:
                            rc = lib.lr_op_gemm_bt_split(P(A), P(W), P(o), P(bias), M, N, K, epi, L.ACT_QUICK_GELU if epi == L.EPI_OUT_OP else 0, L.LR_DT_F16, tile, S)
                        else:
                            rc = lib.lr_op_gemm_bt(P(A), P(W), P(o), P(bias), M, N, K, K, K, o.shape[1], epi, L.ACT_QUICK_GELU if epi == L.EPI_OUT_OP else 0, L.LR_DT_F16, tile, S)
                        torch.cuda.synchronize()
                        outs.append(o if rc == 0 else None)
                    if outs[0] is None or outs[1] is None:
                        continue          # (a combination this entry point refuses)
                    same = torch.equal(outs[0], outs[1])
                    bad += not same
                    ncmp += 1
                    if not same:
                        d = (outs[0].float() - outs[1].float()).abs()
                        print(f"DIFF {name} M={M} N={N} K={K} tile={tile} bias={use_bias} split={split}: {int((d > 0).sum())} elements, max {d.max().item():.3e}")
        # RoPE epilogue (single pass)
        if N % 96 == 0:
            for use_bias in (False, True):
                torch.manual_seed(M + N)
                A = torch.randn(M, K, device="cuda").half()
                W = (torch.randn(N, K, device="cuda") * 0.05).half()
                bias = torch.randn(N, device="cuda") if use_bias else None
                cs = torch.randn(M, 48, 2, device="cuda")
                outs = []
                for lib in libs:
                    o = torch.zeros(M, N, device="cuda").half()
                    rc = lib.lr_op_gemm_rope(P(A), P(W), P(o), P(bias), P(cs), M, N, K, (768 if N == 1152 else (N // 96) * 96 * 2 // 3 // 96 * 96), 96, L.LR_DT_F16, tile, S)
                    torch.cuda.synchronize()
                    outs.append(o if rc == 0 else None)
                if outs[0] is None or outs[1] is None:
                    continue
                same = torch.equal(outs[0], outs[1])
                bad += not same
                ncmp += 1
                if not same:
                    d = (outs[0].float() - outs[1].float()).abs()
                    print(f"DIFF rope M={M} N={N} K={K} tile={tile} bias={use_bias}: {int((d > 0).sum())} elements, max {d.max().item():.3e}")
print("differences:", bad, " compared:", ncmp)
