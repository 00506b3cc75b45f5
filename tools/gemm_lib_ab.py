#!/usr/bin/env python3
"""Two builds of the library on the step's big GEMMs in the default operand form (f16 pass + e4m3 residual pass), timed interleaved in
one process on one box, outputs compared bit for bit.  Per-pass totals weight every shape by its launches per Phi-3.5-V forward.
    python tools/gemm_lib_ab.py <old .so> <new .so> [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch
from llava_reward_amd import _lib as L

def _load(path):          # only the two entry points this script calls: older builds lack later ABI symbols
    lib = C.CDLL(os.path.abspath(path))
    for name in ("lr_op_gemm_bt_mixed", "lr_op_lo8_scratch_bytes"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = L._SIGS[name]
    return lib


libs = [(os.path.basename(p), _load(p)) for p in sys.argv[1:3]]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
st = torch.cuda.current_stream()
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
S = C.c_void_p(st.cuda_stream)
CASES = [("dec.gate_up", 84544, 16384, 3072, L.EPI_SWIGLU_OP, 0, 32, 32), ("dec.qkv(out_op)", 84544, 9216, 3072, L.EPI_OUT_OP, 0, 0, 32),
         ("dec.o", 84544, 3072, 3072, L.EPI_RESADD_F32, 0, 0, 32), ("dec.down", 84544, 3072, 8192, L.EPI_RESADD_F32, 0, 0, 32),
         ("clip.qkv", 313888, 3072, 1024, L.EPI_OUT_OP, 0, 0, 23), ("clip.out", 313888, 1024, 1024, L.EPI_RESADD_F32, 0, 0, 23),
         ("clip.fc1", 313888, 4096, 1024, L.EPI_OUT_OP, L.ACT_QUICK_GELU, 32, 23), ("clip.fc2", 313888, 1024, 4096, L.EPI_RESADD_F32, 0, 0, 23),
         ("llava.gate_up", 64 * 2200, 28672, 4096, L.EPI_SWIGLU_OP, 0, 32, 0), ("llava.down", 64 * 2200, 4096, 14336, L.EPI_RESADD_F32, 0, 0, 0),
         # the CLIP linears as the step runs them: with a bias vector, M not a multiple of the tile height (counted above: weight 0)
         ("clip.out+bias", 313888 - 100, 1024, 1024, L.EPI_RESADD_F32, 0, 0, 0, True), ("clip.fc2+bias", 313888 - 100, 1024, 4096, L.EPI_RESADD_F32, 0, 0, 0, True),
         ("clip.fc1+bias", 313888 - 100, 4096, 1024, L.EPI_OUT_OP, L.ACT_QUICK_GELU, 32, 0, True)]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot = [0.0, 0.0]
print(f"{'GEMM':18s} {libs[0][0]:>24s} {libs[1][0]:>24s}   delta   same bits")
for name, M, N, K, epi, act, oflag, cnt, *rest in CASES:
    torch.manual_seed(M + N + K)
    bias = torch.randn(N, device="cuda") if rest and rest[0] else None
    A = torch.randn(M, 2 * K, device="cuda").to(torch.float16)
    A[:, K:] *= 2.0 ** -11
    W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.float16)
    W8 = torch.zeros(N, K, device="cuda", dtype=torch.float16)
    nout = N // 2 if epi == L.EPI_SWIGLU_OP else N
    op_out = epi in (L.EPI_OUT_OP, L.EPI_SWIGLU_OP)
    outs = [torch.zeros(M, 2 * nout if op_out else nout, device="cuda", dtype=torch.float16 if op_out else torch.float32) for _ in libs]
    ae = torch.full((libs[0][1].lr_op_lo8_scratch_bytes(M, K) + libs[0][1].lr_op_lo8_scratch_bytes(M, nout),), 127, dtype=torch.uint8, device="cuda")
    we = C.c_int(0)
    assert libs[0][1].lr_op_gemm_bt_mixed(P(A), P(W), P(W8), P(ae), P(outs[0]), P(bias), M, N, K, epi, act, L.LR_DT_F16, 7, C.byref(we), S) == 0
    fns = [(lambda lib=lib, o=o: lib.lr_op_gemm_bt_mixed(P(A), P(W), P(W8), P(ae), P(o), P(bias), M, N, K, epi, act, L.LR_DT_F16, oflag, C.byref(we), S)) for (_, lib), o in zip(libs, outs)]
    for o, f in zip(outs, fns):            # one launch each from zeroed outputs: the bits
        o.zero_()
        assert f() == 0
    torch.cuda.synchronize()
    same = torch.equal(outs[0], outs[1])
    t = [1e9, 1e9]
    for rnd in range(3):
        for i in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            t[i] = min(t[i], timed(fns[i]))
    for i in range(2):
        tot[i] += t[i] * cnt
    print(f"{name:18s} {t[0]:21.3f} ms {t[1]:21.3f} ms  {100 * (t[1] / t[0] - 1):+5.1f} %   {same}", flush=True)
    del A, W, W8, outs, ae
    torch.cuda.empty_cache()
print(f"{'per Phi forward':18s} {tot[0]:21.1f} ms {tot[1]:21.1f} ms  {100 * (tot[1] / tot[0] - 1):+5.1f} %")
