import ctypes as C, math, os, sys
sys.path.insert(0, "llava-reward_amd")
import torch
from llava_reward_amd import _lib as L
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
def case(name, B, S, H, hd, causal):
    W = 3 * H * hd
    g = torch.Generator(device="cuda").manual_seed(S)
    qkv = torch.cat([torch.randn(B * S, W, device="cuda", generator=g).half(), (torch.randn(B * S, W, device="cuda", generator=g) * 2.0 ** -12).half()], dim=1).contiguous()
    mask = torch.ones(B, S, dtype=torch.int64, device="cuda") if causal else None
    kmin = torch.zeros(B, dtype=torch.int32, device="cuda") if causal else None
    st = torch.cuda.current_stream()
    res = {}
    outs = {}
    for pm in ("1024", "256"):
        os.environ["LR_ATT_PP_MIN_S"] = pm
        out = torch.zeros(B * S, 2 * H * hd, device="cuda", dtype=torch.float16)
        a = (P(qkv), P(qkv), P(qkv), P(out), P(mask), P(kmin), 2 * W, 2 * H * hd, 0, H * hd, 2 * H * hd, W, H * hd, B, S, H, hd, int(causal), 1, 1.0 / math.sqrt(hd), L.LR_DT_F16, C.c_void_p(st.cuda_stream))
        for _ in range(3): assert lib.lr_op_attention_split(*a) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10): lib.lr_op_attention_split(*a)
        e1.record(st); torch.cuda.synchronize()
        res[pm] = e0.elapsed_time(e1) / 10; outs[pm] = out
    print(f"{name}: plain {res['1024']:.3f} ms, ping-pong {res['256']:.3f} ms ({(res['256']/res['1024']-1)*100:+.1f} %), same bits: {torch.equal(outs['1024'], outs['256'])}")
case("clip 577", 544, 577, 16, 64, False)
case("decoder 890", 32, 890, 32, 96, True)
case("decoder 700", 32, 700, 32, 96, True)
case("decoder 600", 32, 600, 32, 96, True)
