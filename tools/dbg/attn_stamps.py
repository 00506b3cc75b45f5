#!/usr/bin/env python3
"""Where a tile period of the ping-pong attention kernel goes: s_memtime stamps around the vector segment, the two workgroup barriers
and the matrix segment, summed over the key tiles of the heaviest workgroup (diagnostic build of attention.hip, -DLR_ATT_DIAG=16, or
16 | 1 = no softmax arithmetic, 16 | 2 = no LDS fragment reads ...; results of the variants with other bits are invalid).
    python3 tools/dbg/attn_stamps.py tools/dbg/lib_att_stamps.so [more diagnostic builds]
s_memtime counts shader cycles; MFMA issue of one matrix segment: 72 x 32 = 2304 cycles."""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch
from llava_reward_amd import _lib as L

P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
B, S, H, hd = 32, 2642, 32, 96
W = 3 * H * hd
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.cat([torch.randn(B * S, W, device="cuda", generator=g).half(), (torch.randn(B * S, W, device="cuda", generator=g) * 2.0 ** -12).half()], dim=1).contiguous()
mask = torch.ones(B, S, dtype=torch.int64, device="cuda")
kmin = torch.zeros(B, dtype=torch.int32, device="cuda")
for path in sys.argv[1:]:
    lib = L.load(path)
    out = torch.zeros(B * S * 2 * H * hd + 4096, device="cuda", dtype=torch.float16)
    st = torch.cuda.current_stream()
    args = (P(qkv), P(qkv), P(qkv), P(out), P(mask), P(kmin), 2 * W, 2 * H * hd, 0, H * hd, 2 * H * hd, W, H * hd, B, S, H, hd, 1, 1, 1.0 / math.sqrt(hd), L.LR_DT_F16,
            C.c_void_p(st.cuda_stream))
    for _ in range(3):
        assert lib.lr_op_attention_split(*args) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5):
        lib.lr_op_attention_split(*args)
    e1.record(st)
    torch.cuda.synchronize()
    dbg = out[B * S * 2 * H * hd:].view(torch.int64)[:64].cpu().view(8, 8)
    print(f"{os.path.basename(path)}: {e0.elapsed_time(e1) / 5:.3f} ms per launch")
    print("  wave  tiles     vector   barrier1     matrix   barrier2   (shader cycles per tile)")
    for w in range(8):
        n = max(1, int(dbg[w, 4]))
        v = [float(dbg[w, i]) / n for i in range(4)]
        print(f"  {w:4d} {n:6d} {v[0]:10.0f} {v[1]:10.0f} {v[2]:10.0f} {v[3]:10.0f}   period {sum(v):.0f}")
