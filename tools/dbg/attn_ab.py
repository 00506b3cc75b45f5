import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from llava_reward_amd import _lib as L
libs = {"new": L.load()}
# LR_ATT_DIAG bits: 1 no softmax arithmetic, 2 no LDS fragment reads, 4 no barriers in the loop, 8 no DMA in the loop
for v, k in ((1, "no_softmax"), (2, "no_lds_reads"), (3, "skeleton(1+2)"), (7, "skeleton_no_barriers(1+2+4)"), (11, "skeleton_no_dma(1+2+8)"),
             (15, "mfma_loop_only(1+2+4+8)")):
    f = f"lib_attd{v}.so"
    if os.path.exists(os.path.join(ROOT, "tools", "dbg", f)):
        libs[k] = L.load(os.path.join(ROOT, "tools", "dbg", f))
def bench(lib, B, S, H, hd, causal, Hkv, reps=5):
    W = (H + 2 * Hkv) * hd
    torch.manual_seed(0)
    qkv = torch.cat([torch.randn(B * S, W, device="cuda").half(), (torch.randn(B * S, W, device="cuda") * 2.0 ** -12).half()], dim=1).contiguous()
    out = torch.zeros(B * S, 2 * H * hd, device="cuda", dtype=torch.float16)
    mask = torch.ones(B, S, dtype=torch.int64, device="cuda") if causal else None
    kmin = torch.zeros(B, dtype=torch.int32, device="cuda") if causal else None
    st = torch.cuda.current_stream()
    P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
    args = (P(qkv), P(qkv), P(qkv), P(out), P(mask), P(kmin), 2 * W, 2 * H * hd, 0, H * hd, (H + Hkv) * hd, W, H * hd, B, S, H, hd, int(causal), H // Hkv,
            1.0 / math.sqrt(hd), L.LR_DT_F16, C.c_void_p(st.cuda_stream))
    assert lib.lr_op_attention_split(*args) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        lib.lr_op_attention_split(*args)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out.clone()
for name, cfgs in (("phi", (32, 2642, 32, 96, True, 32)),):
    res = {}
    for rnd in range(3):
        for k, lib in libs.items():
            ms, o = bench(lib, *cfgs)
            res.setdefault(k, []).append(ms); res[k + "_o"] = o
    print(name, {k: round(min(v), 3) for k, v in res.items() if not k.endswith("_o")})
