"""Per-kernel parity on a real MI355X: every HIP kernel, called through the C ABI, against a plain
torch fp32 computation of the same op on the SAME (already rounded) operands.

Tolerances: GEMM/attention accumulate in fp32 like the checker, so fp32-output results must agree
to 2e-5 relative to the output scale (summation order only); operand-dtype outputs additionally
carry one rounding of the output (bf16: 2^-8 relative, f16: 2^-11)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from llava_reward_amd import _lib as L
from llava_reward_amd import synth

pytestmark = pytest.mark.gpu

DTS = [("bf16", L.LR_DT_BF16, torch.bfloat16, 2.0 ** -8), ("f16", L.LR_DT_F16, torch.float16, 2.0 ** -11)]


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return L.load()


def P(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rnd(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).cuda()


def test_synth_fill_bit_exact(lib):
    for name, n, std, off in (("model.layers.0.mlp.down_proj.weight", 1 << 20, 0.02, 0.0), ("x.norm", 3072, 0.05, 1.0)):
        out = torch.empty(n, device="cuda", dtype=torch.float32)
        assert lib.lr_op_synth_fill(P(out), n, 1234, name.encode(), std, off, 1, stream()) == 0
        ref = synth.gen_tensor(1234, name, (n,), std, off)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("dt", DTS, ids=[d[0] for d in DTS])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("shape", [(300, 256, 128), (1000, 512, 640), (257, 1024, 1024), (4100, 768, 3072)])
def test_gemm_f32_out_and_resadd(lib, dt, tile, shape):
    _, code, tdt, _ = dt
    M, N, K = shape
    A = rnd((M, K), 1).to(tdt)
    W = rnd((N, K), 2, 0.05).to(tdt)
    bias = rnd((N,), 3)
    ref = A.float() @ W.float().t() + bias
    Cb = torch.empty(M, N, device="cuda", dtype=torch.float32)
    assert lib.lr_op_gemm_bt(P(A), P(W), P(Cb), P(bias), M, N, K, K, K, N, L.EPI_OUT_F32, 0, code, tile, stream()) == 0
    scale = ref.abs().max().item()
    assert (Cb - ref).abs().max().item() < 2e-5 * scale * math.sqrt(K / 64)
    # residual add in place, no bias
    res = rnd((M, N), 4)
    Cr = res.clone()
    assert lib.lr_op_gemm_bt(P(A), P(W), P(Cr), P(None), M, N, K, K, K, N, L.EPI_RESADD_F32, 0, code, tile, stream()) == 0
    ref2 = res + A.float() @ W.float().t()
    assert (Cr - ref2).abs().max().item() < 2e-5 * scale * math.sqrt(K / 64)


@pytest.mark.parametrize("dt", DTS, ids=[d[0] for d in DTS])
@pytest.mark.parametrize("tile", [0, 2, 3, 6])
@pytest.mark.parametrize("act", [L.ACT_NONE, L.ACT_QUICK_GELU, L.ACT_GELU_ERF])
def test_gemm_operand_out_activations(lib, dt, tile, act):
    _, code, tdt, ulp = dt
    M, N, K = 777, 512, 256
    A = rnd((M, K), 5).to(tdt)
    W = rnd((N, K), 6, 0.1).to(tdt)
    bias = rnd((N,), 7)
    y = A.float() @ W.float().t() + bias
    if act == L.ACT_QUICK_GELU:
        y = y * torch.sigmoid(1.702 * y)
    elif act == L.ACT_GELU_ERF:
        y = torch.nn.functional.gelu(y)
    out = torch.zeros(M, N, device="cuda", dtype=tdt)
    assert lib.lr_op_gemm_bt(P(A), P(W), P(out), P(bias), M, N, K, K, K, N, L.EPI_OUT_OP, act, code, tile, stream()) == 0
    err = (out.float() - y).abs()
    assert (err <= ulp * y.abs() + 1e-4).all(), err.max().item()


@pytest.mark.parametrize("dt", DTS, ids=[d[0] for d in DTS])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 5, 6])
def test_gemm_swiglu(lib, dt, tile):
    _, code, tdt, ulp = dt
    M, I, K = 515, 512, 384
    A = rnd((M, K), 8).to(tdt)
    W = rnd((2 * I, K), 9, 0.1).to(tdt)
    gu = A.float() @ W.float().t()
    ref = gu[:, I:] * torch.nn.functional.silu(gu[:, :I])
    # pack rows as the engine does: [g0..31, u0..31, g32..63, u32..63, ...]
    g = torch.arange(I)
    perm = torch.empty(2 * I, dtype=torch.long)
    perm[(g // 32) * 64 + g % 32] = g
    perm[(g // 32) * 64 + 32 + g % 32] = I + g
    Wp = W[perm.cuda()].contiguous()
    out = torch.zeros(M, I, device="cuda", dtype=tdt)
    assert lib.lr_op_gemm_bt(P(A), P(Wp), P(out), P(None), M, 2 * I, K, K, K, I, L.EPI_SWIGLU_OP, 0, code, tile, stream()) == 0
    err = (out.float() - ref).abs()
    assert (err <= ulp * ref.abs() + 1e-4).all(), err.max().item()
    # with gate / up biases (Qwen2.5-VL ViT MLP, modeling_qwen2_5_vl.py Qwen2_5_VLMLP(bias=True)), packed like the rows
    bias = rnd((2 * I,), 10, 0.5)
    gub = gu + bias
    refb = gub[:, I:] * torch.nn.functional.silu(gub[:, :I])
    bp = bias[perm.cuda()].contiguous()
    assert lib.lr_op_gemm_bt(P(A), P(Wp), P(out), P(bp), M, 2 * I, K, K, K, I, L.EPI_SWIGLU_OP, 0, code, tile, stream()) == 0
    err = (out.float() - refb).abs()
    assert (err <= ulp * refb.abs() + 1e-4).all(), err.max().item()


@pytest.mark.parametrize("dt", DTS, ids=[d[0] for d in DTS])
def test_gemm_fused_rope_epilogue(lib, dt):
    """QKV projection + RoPE in the epilogue; q/k head dims pair-interleaved (reference dims i, i+hd/2 at 2i, 2i+1)."""
    _, code, tdt, ulp = dt
    M, D, hd, K = 700, 384, 96, 256          # rope_cols = 2D = 768: multiple of the 256-wide tile and of the head
    N, half = 3 * D, hd // 2
    A = rnd((M, K), 31).to(tdt)
    W = rnd((N, K), 32, 0.1).to(tdt)
    ang = rnd((M, half), 33, 3.0)
    cs = torch.stack([torch.cos(ang) * 1.19, torch.sin(ang) * 1.19], dim=-1).contiguous()      # [M, half, 2]
    y = (A.float() @ W.float().t())
    ref = y.clone()
    qk = y[:, :2 * D].view(M, 2 * D // hd, half, 2)
    c, s_ = cs[:, None, :, 0], cs[:, None, :, 1]
    ref[:, :2 * D] = torch.stack([qk[..., 0] * c - qk[..., 1] * s_, qk[..., 1] * c + qk[..., 0] * s_], dim=-1).reshape(M, 2 * D)
    out = torch.zeros(M, N, device="cuda", dtype=tdt)
    assert lib.lr_op_gemm_rope(P(A), P(W), P(out), P(None), P(cs), M, N, K, 2 * D, hd, code, 5, stream()) == 0
    err = (out.float() - ref).abs()
    assert (err <= ulp * ref.abs() + 2e-4).all(), err.max().item()
    # q/k/v bias added before the rotation (Qwen2.5-VL: Qwen2_5_VLAttention q_proj/k_proj/v_proj bias=True)
    bias = rnd((N,), 34, 0.5)
    yb = y + bias
    refb = yb.clone()
    qk = yb[:, :2 * D].view(M, 2 * D // hd, half, 2)
    refb[:, :2 * D] = torch.stack([qk[..., 0] * c - qk[..., 1] * s_, qk[..., 1] * c + qk[..., 0] * s_], dim=-1).reshape(M, 2 * D)
    assert lib.lr_op_gemm_rope(P(A), P(W), P(out), P(bias), P(cs), M, N, K, 2 * D, hd, code, 5, stream()) == 0
    err = (out.float() - refb).abs()
    assert (err <= ulp * refb.abs() + 2e-4).all(), err.max().item()


def test_fused_rope_is_the_reference_arithmetic_bit_for_bit(lib):
    """The rotation of the fused epilogue is q * cos + rotate_half(q) * sin as the reference evaluates it in fp32
    (modeling_phi3_v.py:521-553): two ROUNDED products and one rounded sum per output, no fused multiply-add (common.h rope_pair; left
    to the compiler, which product is fused differed between instantiations of one kernel -- round 6).  Inputs whose GEMM sums are exact
    in fp32 (small integers) make the check bit for bit: the kernel's f16 output must equal the f16 rounding of torch's own mul / mul /
    add on the same fp32 values, on inside tiles, edge tiles and the narrow-tile instantiation alike."""
    for M, tile in ((700, 5), (1500, -1), (2100, 6), (256, -1)):          # (-1 / 6: the product kernel, narrow tiles chosen by M)
        D, hd, K = 384, 96, 128
        N, half = 3 * D, hd // 2
        g = torch.Generator().manual_seed(M)
        A = torch.randint(-4, 5, (M, K), generator=g).float().cuda().half()
        W = torch.randint(-1, 2, (N, K), generator=g).float().cuda().half()
        ang = rnd((M, half), 35, 3.0)
        cs = torch.stack([torch.cos(ang) * 1.19, torch.sin(ang) * 1.19], dim=-1).contiguous()
        y = A.float() @ W.float().t()                      # |sums| <= 512: exact in fp32 in any order
        assert (y == y.round()).all()
        qk = y[:, :2 * D].view(M, 2 * D // hd, half, 2)
        c, s_ = cs[:, None, :, 0], cs[:, None, :, 1]
        ref = y.clone()
        x0, x1 = qk[..., 0], qk[..., 1]
        ref[:, :2 * D] = torch.stack([torch.sub(torch.mul(x0, c), torch.mul(x1, s_)), torch.add(torch.mul(x1, c), torch.mul(x0, s_))], dim=-1).reshape(M, 2 * D)
        out = torch.zeros(M, N, device="cuda", dtype=torch.float16)
        assert lib.lr_op_gemm_rope(P(A), P(W), P(out), P(None), P(cs), M, N, K, 2 * D, hd, L.LR_DT_F16, tile, stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(out, ref.half()), (M, tile, (out.float() - ref.half().float()).abs().max().item())


def _attn_ref(q, k, v, mask, causal, scale):
    # q,k,v [B,H,S,hd] fp32; mask [B,S] or None
    B, H, S, _ = q.shape
    s = torch.matmul(q, k.transpose(2, 3)) * scale
    ok = torch.ones(B, 1, S, S, dtype=torch.bool, device=q.device)
    if causal:
        ok = ok & torch.tril(torch.ones(S, S, dtype=torch.bool, device=q.device))[None, None]
    if mask is not None:
        ok = ok & mask.bool()[:, None, None, :]
    s = s.masked_fill(~ok, float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)
    return torch.matmul(p, v)


@pytest.mark.parametrize("dt", DTS, ids=[d[0] for d in DTS])
@pytest.mark.parametrize("case", [(2, 3, 300, 96, True), (3, 2, 577, 64, False), (1, 4, 1000, 96, True), (2, 2, 130, 64, True),
                                  (2, 3, 1300, 96, True), (2, 2, 1100, 64, False), (1, 2, 1281, 64, True), (1, 2, 1024, 96, False)])   # >= 1024: ping-pong schedule
def test_attention(lib, dt, case):
    _, code, tdt, ulp = dt
    B, H, S, hd, causal = case
    D = H * hd
    qkv = rnd((B * S, 3 * D), 11).to(tdt)
    mask = None
    kmin = None
    if causal:
        mask = torch.ones(B, S, dtype=torch.int64)
        for b in range(B):
            mask[b, : 37 * b] = 0          # left padding, different per row
        kmin = torch.tensor([37 * b for b in range(B)], dtype=torch.int32).cuda()
        mask = mask.cuda()
    out = torch.zeros(B * S, D, device="cuda", dtype=tdt)
    scale = 1.0 / math.sqrt(hd)
    rc = lib.lr_op_attention(P(qkv), P(qkv), P(qkv), P(out), P(mask), P(kmin), 3 * D, D, 0, D, 2 * D, B, S, H, hd, int(causal), 1,
                             scale, code, stream())
    assert rc == 0
    f = qkv.float().view(B, S, 3, H, hd)
    q, k, v = (f[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ref = _attn_ref(q, k, v, mask, causal, scale).permute(0, 2, 1, 3).reshape(B * S, D)
    got = out.float()
    valid = torch.ones(B * S, dtype=torch.bool, device="cuda") if mask is None else mask.bool().reshape(-1)
    err = (got - ref).abs()[valid]
    # P is rounded to the operand dtype before the PV product: error ~ ulp * |v|max
    assert err.max().item() < 2.5 * ulp * v.abs().max().item(), err.max().item()
    # fully masked (pad) query rows come out as exact zeros
    if mask is not None:
        assert (got[~valid] == 0).all()


@pytest.mark.parametrize("dt", DTS, ids=[d[0] for d in DTS])
@pytest.mark.parametrize("H", [128, 384, 1024, 3072])
def test_norm_rows(lib, dt, H):
    _, code, tdt, ulp = dt
    rows = 1001
    x = rnd((rows, H), 21, 3.0) + 0.5
    w = rnd((H,), 22) * 0.1 + 1
    b = rnd((H,), 23) * 0.1
    y = torch.zeros(rows, H, device="cuda", dtype=tdt)
    assert lib.lr_op_norm_rows(P(x), P(w), P(b), P(y), rows, H, 1e-5, code, stream()) == 0
    ref = torch.nn.functional.layer_norm(x, (H,), w, b, 1e-5)
    assert ((y.float() - ref).abs() <= ulp * ref.abs() + 1e-5).all()
    assert lib.lr_op_norm_rows(P(x), P(w), P(None), P(y), rows, H, 1e-5, code, stream()) == 0
    ref = w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5))
    assert ((y.float() - ref).abs() <= ulp * ref.abs() + 1e-5).all()


@pytest.mark.parametrize("dt", DTS, ids=[d[0] for d in DTS])
@pytest.mark.parametrize("case", [(2, 8, 2, 333), (1, 4, 4, 700), (2, 4, 1, 130), (2, 4, 2, 1200)])
def test_attention_gqa_head_dim_128(lib, dt, case):
    """Mistral-style attention of the LLaVA branch: head_dim 128, several query heads per K/V head, causal + left padding."""
    _, code, tdt, ulp = dt
    B, H, KV, S = case
    hd = 128
    Hq, Hkv = H * hd, KV * hd
    ld = Hq + 2 * Hkv
    qkv = rnd((B * S, ld), 41).to(tdt)
    mask = torch.ones(B, S, dtype=torch.int64)
    for b in range(B):
        mask[b, : 29 * b] = 0
    kmin = torch.tensor([29 * b for b in range(B)], dtype=torch.int32).cuda()
    mask = mask.cuda()
    out = torch.zeros(B * S, Hq, device="cuda", dtype=tdt)
    scale = 1.0 / math.sqrt(hd)
    rc = lib.lr_op_attention(P(qkv), P(qkv), P(qkv), P(out), P(mask), P(kmin), ld, Hq, 0, Hq, Hq + Hkv, B, S, H, hd, 1, H // KV,
                             scale, code, stream())
    assert rc == 0
    f = qkv.float()
    q = f[:, :Hq].view(B, S, H, hd).permute(0, 2, 1, 3)
    k = f[:, Hq:Hq + Hkv].view(B, S, KV, hd).permute(0, 2, 1, 3).repeat_interleave(H // KV, dim=1)
    v = f[:, Hq + Hkv:].view(B, S, KV, hd).permute(0, 2, 1, 3).repeat_interleave(H // KV, dim=1)
    ref = _attn_ref(q, k, v, mask, True, scale).permute(0, 2, 1, 3).reshape(B * S, Hq)
    valid = mask.bool().reshape(-1)
    err = (out.float() - ref).abs()[valid]
    assert err.max().item() < 2.5 * ulp * v.abs().max().item(), err.max().item()
    assert (out.float()[~valid] == 0).all()


@pytest.mark.parametrize("dt", DTS, ids=[d[0] for d in DTS])
@pytest.mark.parametrize("hd", [96, 64])
def test_attention_ragged_segments(lib, dt, hd):
    """Block-diagonal dense attention over cu_seqlens (Qwen2_5_VLVisionAttention: windows of <= 64 patches, or one
    segment per image), including empty, 1-row, exactly-128 and multi-tile segments."""
    _, code, tdt, ulp = dt
    H = 3
    lens = [64, 48, 1, 0, 128, 130, 700, 64, 12]
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    N = int(cu[-1])
    ld = 3 * H * hd
    qkv = rnd((N, ld), 51, 0.7).to(tdt)
    out = torch.zeros(N, H * hd, device="cuda", dtype=tdt)
    scale = 80 ** -0.5                         # the ViT scales by the TRUE head dim (80), not the stored width
    assert lib.lr_op_attention_segments(P(qkv), P(qkv), P(qkv), P(out), cu.ctypes.data_as(C.POINTER(C.c_int32)), len(lens),
                                        ld, H * hd, 0, H * hd, 2 * H * hd, H, hd, scale, code, stream()) == 0
    torch.cuda.synchronize()
    f = qkv.float().view(N, 3, H, hd)
    ref = torch.zeros(N, H, hd, device="cuda")
    for a, b in zip(cu[:-1], cu[1:]):
        if b > a:
            q, k, v = (f[a:b, i].transpose(0, 1) for i in range(3))
            ref[a:b] = (torch.softmax(q @ k.transpose(1, 2) * scale, dim=-1) @ v).transpose(0, 1)
    err = (out.float().view(N, H, hd) - ref).abs()
    # P is rounded to the operand dtype before the PV product, the output once more: error ~ ulp * |v|max
    assert err.max().item() < 3 * ulp * f[:, 2].abs().max().item(), err.max().item()


def _split(x, tdt):
    """hi = round(x), lo = round(x - hi) in the operand type (csrc/common.h split2)."""
    hi = x.to(tdt)
    lo = (x - hi.float()).to(tdt)
    return hi, lo


@pytest.mark.parametrize("tile", [0, 5, 6])
def test_gemm_split_operand_matches_fp32(lib, tile):
    """Split-operand GEMM (parity mode): A = A_hi + A_lo (f16), W bf16-valued hence exact in f16 -> the fp32 product to ~2^-22;
    operand-typed outputs come back as [hi | lo] whose sum again carries ~22 bits."""
    code, tdt = L.LR_DT_F16, torch.float16
    M, N, K = 700, 512, 384
    A32 = rnd((M, K), 61)
    W = rnd((N, K), 62, 0.05).to(torch.bfloat16).to(tdt)            # bf16-valued weights, stored as f16 (exact)
    bias = rnd((N,), 63)
    hi, lo = _split(A32, tdt)
    A2 = torch.cat([hi, lo], dim=1).contiguous()
    ref = (A32.double() @ W.double().t() + bias.double()).float()
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    assert lib.lr_op_gemm_bt_split(P(A2), P(W), P(out), P(bias), M, N, K, L.EPI_OUT_F32, 0, code, tile, stream()) == 0
    scale = ref.abs().max().item()
    assert (out - ref).abs().max().item() < 2e-6 * scale
    single = torch.empty(M, N, device="cuda", dtype=torch.float32)      # same problem, single-pass operands: ~2^-11
    assert lib.lr_op_gemm_bt(P(hi), P(W), P(single), P(bias), M, N, K, K, K, N, L.EPI_OUT_F32, 0, code, tile, stream()) == 0
    assert (single - ref).abs().max().item() > 20 * (out - ref).abs().max().item()
    # operand-typed output with GELU: [hi | lo]
    o2 = torch.zeros(M, 2 * N, device="cuda", dtype=tdt)
    assert lib.lr_op_gemm_bt_split(P(A2), P(W), P(o2), P(bias), M, N, K, L.EPI_OUT_OP, L.ACT_GELU_ERF, code, tile, stream()) == 0
    g = torch.nn.functional.gelu(ref)
    got = o2[:, :N].float() + o2[:, N:].float()
    assert (got - g).abs().max().item() < 4e-6 * g.abs().max().item()
    assert torch.equal(o2[:, :N], got.to(tdt)) or (o2[:, :N].float() - g).abs().max().item() < 2.0 ** -10 * g.abs().max().item()
    # SwiGLU
    I = N // 2
    gidx = torch.arange(I)
    perm = torch.empty(N, dtype=torch.long)
    perm[(gidx // 32) * 64 + gidx % 32] = gidx
    perm[(gidx // 32) * 64 + 32 + gidx % 32] = I + gidx
    Wp = W[perm.cuda()].contiguous()
    gu = (A32.double() @ W.double().t()).float()
    sref = gu[:, I:] * torch.nn.functional.silu(gu[:, :I])
    o3 = torch.zeros(M, 2 * I, device="cuda", dtype=tdt)
    assert lib.lr_op_gemm_bt_split(P(A2), P(Wp), P(o3), P(None), M, N, K, L.EPI_SWIGLU_OP, 0, code, tile, stream()) == 0
    got = o3[:, :I].float() + o3[:, I:].float()
    assert (got - sref).abs().max().item() < 4e-6 * sref.abs().max().item() + 1e-6


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("k2,b_exact", [(64, True), (128, True), (128, False)])
def test_gemm_k_extension_adapter(lib, split, k2, b_exact):
    """K-extension of the deep-pipelined GEMM (un-merged LoRA adapter, lr_model_desc.lora_rank): C = A W^T + T B^T with T, B
    walked as extra K segments from their own buffers (t_hi x B, t_lo x B, t_hi x B_lo).  Against fp64 on the un-rounded
    operands; shapes with more tiles than one workgroup walks and K segments of 1-2 K-tiles between long ones."""
    code, tdt = L.LR_DT_F16, torch.float16
    for (M, N, K) in [(700, 512, 384), (4100, 768, 1024)]:
        A32, T32 = rnd((M, K), 71), rnd((M, k2), 72, 0.7)
        W = rnd((N, K), 73, 0.05).to(torch.bfloat16).to(tdt)
        B32 = rnd((N, k2), 74, 0.05)
        if b_exact:
            B32 = B32.to(torch.bfloat16).float()
        Bh, Bl = _split(B32, tdt)
        bias = rnd((N,), 75)
        if split:
            ah, al = _split(A32, tdt)
            th, tl = _split(T32, tdt)
            A, T = torch.cat([ah, al], dim=1).contiguous(), torch.cat([th, tl], dim=1).contiguous()
            Aeff, Teff = A32.double(), T32.double()
        else:
            A, T = A32.to(tdt), T32.to(tdt)
            Aeff, Teff = A.double(), T.double()
        Beff = B32.double() if (split and not b_exact) else Bh.double()
        ref = (Aeff @ W.double().t() + Teff @ Beff.t() + bias.double()).float()
        out = torch.empty(M, N, device="cuda", dtype=torch.float32)
        Blo = P(Bl) if (split and not b_exact) else P(None)
        assert lib.lr_op_gemm_bt_ext(P(A), P(W), P(T), P(Bh.contiguous()), Blo, P(out), P(bias), M, N, K, k2, split, L.EPI_OUT_F32, 0, code, stream()) == 0
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        print(f"[k-extension split={split} k2={k2} exact={b_exact} {M}x{N}x{K}] rel err {err:.2e}")
        assert err < (3e-6 if split else 2e-5)          # single-pass: operands rounded identically, only the summation order differs
        # the extension is really in: without it the result is far away
        base = (Aeff @ W.double().t() + bias.double()).float()
        assert (out - base).abs().max().item() > 100 * (out - ref).abs().max().item()
        if split:                                        # operand-typed output [hi | lo] through the same K loop
            o2 = torch.zeros(M, 2 * N, device="cuda", dtype=tdt)
            assert lib.lr_op_gemm_bt_ext(P(A), P(W), P(T), P(Bh.contiguous()), Blo, P(o2), P(bias), M, N, K, k2, 1, L.EPI_OUT_OP, 0, code, stream()) == 0
            got = o2[:, :N].float() + o2[:, N:].float()
            assert (got - ref).abs().max().item() < 4e-6 * ref.abs().max().item()


@pytest.mark.parametrize("S", [333, 1300])                # >= 1024: ping-pong schedule (head_dim 64 / 96)
@pytest.mark.parametrize("hd,causal,group", [(96, True, 1), (64, False, 1), (128, True, 4), (96, False, 1)])
def test_attention_split_operand_matches_fp32(lib, hd, causal, group, S):
    """3-pass attention (hi.hi + hi.lo + lo.hi for QK^T and PV) against fp64 softmax attention on the un-rounded q, k, v."""
    code, tdt = L.LR_DT_F16, torch.float16
    B, H = 2, 4
    Hkv = H // group
    wq, wkv = H * hd, Hkv * hd
    W = wq + 2 * wkv
    qkv32 = rnd((B * S, W), 71, 0.8)
    hi, lo = _split(qkv32, tdt)
    qkv2 = torch.cat([hi, lo], dim=1).contiguous()                    # [B*S, 2W]
    mask = torch.ones(B, S, dtype=torch.int64, device="cuda")
    mask[1, :37] = 0                                                  # left padding
    kmin = torch.tensor([0, 37], dtype=torch.int32, device="cuda")
    out = torch.zeros(B * S, 2 * wq, device="cuda", dtype=tdt)
    scale = hd ** -0.5
    rc = lib.lr_op_attention_split(P(qkv2), P(qkv2), P(qkv2), P(out), P(mask if causal else None), P(kmin if causal else None),
                                   2 * W, 2 * wq, 0, wq, wq + wkv, W, wq, B, S, H, hd, 1 if causal else 0, group, scale, code, stream())
    assert rc == 0
    torch.cuda.synchronize()
    f = qkv32.double().view(B, S, W)
    q = f[..., :wq].view(B, S, H, hd).transpose(1, 2)
    k = f[..., wq:wq + wkv].view(B, S, Hkv, hd).transpose(1, 2).repeat_interleave(group, dim=1)
    v = f[..., wq + wkv:].view(B, S, Hkv, hd).transpose(1, 2).repeat_interleave(group, dim=1)
    s = q @ k.transpose(2, 3) * scale
    if causal:
        ok = torch.tril(torch.ones(S, S, dtype=torch.bool, device="cuda"))[None, None] & (mask[:, None, None, :] != 0)
        s = s.masked_fill(~ok, float("-inf"))
    ref = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B * S, wq).float()
    got = out[:, :wq].float() + out[:, wq:].float()
    valid = (mask.reshape(-1) != 0) if causal else torch.ones(B * S, dtype=torch.bool, device="cuda")
    err = (got - ref).abs()[valid].max().item()
    assert err < 5e-6 * ref[valid].abs().max().item(), err


@pytest.mark.parametrize("pingpong", [1, 0], ids=["pingpong", "plain-loop"])
@pytest.mark.parametrize("hd,causal,group", [(96, True, 1), (64, False, 1), (128, True, 4)])
def test_attention_lazy_maximum_on_ramped_scores(lib, monkeypatch, hd, causal, group, pingpong):
    """The lazy reference maximum's OWN branch (attention.hip: a row's reference moves only when the new maximum exceeds it by more than
    AttnParams::lazy_t; alpha != 1 rescales, softmax weights up to 2^lazy_t).  N(0, 0.8) data never reaches it after the first key
    tile, so here the scores RAMP along the key axis: one head-dim coordinate carries +3 log2 units per 64-key tile and one jump of +20
    at key 700, i.e. the running maximum moves in every tile, sometimes by less than the threshold (reference kept, weights > 1) and
    sometimes by more (reference moved, outputs rescaled), on top of N(0, 0.8) noise.  Thresholds 0 (exact maximum: what the engine
    runs in strict stages), 3 and 8 (default stages) against fp64 softmax attention (the reference's softmax, modeling_phi3_v.py:685-701):
    the lazy forms within 2e-6 of the exact form's own error (whose floor here is the fp32 rounding of scores this large); threshold 0 must give lr_op_attention_split_ex's bits whatever the
    schedule (ping-pong / plain per-tile loop), and out-of-range thresholds are refused."""
    monkeypatch.setenv("LR_ATT_PINGPONG", str(pingpong))
    code, tdt = L.LR_DT_F16, torch.float16
    B, H, S = 2, 4, 1300
    Hkv = H // group
    wq, wkv = H * hd, Hkv * hd
    W = wq + 2 * wkv
    scale = hd ** -0.5
    qkv32 = rnd((B * S, W), 91, 0.8)
    j = torch.arange(S, device="cuda", dtype=torch.float32)
    ramp = 3.0 * torch.floor(j / 64) + 20.0 * (j >= 700).float()            # log2 units
    q0 = 4.0
    v = qkv32.view(B, S, W)
    for h in range(H):
        v[:, :, h * hd] = q0                                                  # the carrier coordinate of every query head
    for h in range(Hkv):
        v[:, :, wq + h * hd] = (ramp / (q0 * scale * 1.4426950408889634))[None, :]
    hi, lo = _split(qkv32, tdt)
    qkv2 = torch.cat([hi, lo], dim=1).contiguous()
    mask = torch.ones(B, S, dtype=torch.int64, device="cuda")
    mask[1, :37] = 0
    kmin = torch.tensor([0, 37], dtype=torch.int32, device="cuda")
    f = (hi.double() + lo.double()).view(B, S, W)                             # what the kernel is given: the 22-bit operands
    q = f[..., :wq].view(B, S, H, hd).transpose(1, 2)
    k = f[..., wq:wq + wkv].view(B, S, Hkv, hd).transpose(1, 2).repeat_interleave(group, dim=1)
    vv = f[..., wq + wkv:].view(B, S, Hkv, hd).transpose(1, 2).repeat_interleave(group, dim=1)
    sc = q @ k.transpose(2, 3) * scale
    if causal:
        ok = torch.tril(torch.ones(S, S, dtype=torch.bool, device="cuda"))[None, None] & (mask[:, None, None, :] != 0)
        sc = sc.masked_fill(~ok, float("-inf"))
    ref = (torch.softmax(sc, dim=-1) @ vv).transpose(1, 2).reshape(B * S, wq).float()
    valid = (mask.reshape(-1) != 0) if causal else torch.ones(B * S, dtype=torch.bool, device="cuda")
    outs, errs = {}, {}
    for thr in (0.0, 3.0, 8.0):
        out = torch.zeros(B * S, 2 * wq, device="cuda", dtype=tdt)
        rc = lib.lr_op_attention_split_ex(P(qkv2), P(qkv2), P(qkv2), P(out), P(mask if causal else None), P(kmin if causal else None),
                                          2 * W, 2 * wq, 0, wq, wq + wkv, W, wq, B, S, H, hd, 1 if causal else 0, group, scale, thr, code, stream())
        assert rc == 0
        torch.cuda.synchronize()
        got = out[:, :wq].float() + out[:, wq:].float()
        err = (got - ref).abs()[valid].max().item() / ref[valid].abs().max().item()
        print(f"[lazy maximum hd={hd} causal={causal} pingpong={pingpong} threshold {thr}] rel err {err:.2e}")
        errs[thr] = err
        outs[thr] = out
    # The exact form's own floor on THIS data is the fp32 rounding of the scores themselves: they reach 80 log2 units (|q . k| ~ 540),
    # so one ulp of a score is 80 x 2^-23 ~ 1e-5 log2 units -> ~7e-6 on a softmax weight (measured 3 .. 7e-6; on N(0, 0.8) data,
    # scores <= ~8, the same kernels sit below 5e-6: test_attention_split_operand_matches_fp32).  What is asserted of the lazy
    # thresholds is that they add nothing visible on top of it.
    assert errs[0.0] < 1.2e-5, errs
    assert errs[3.0] < errs[0.0] + 2e-6 and errs[8.0] < errs[0.0] + 2e-6, errs
    # the thresholds really differ in their arithmetic on this data (the branch under test is alive) ...
    assert not torch.equal(outs[0.0], outs[8.0])
    # ... lr_op_attention_split is the threshold-8 form, and the exact form does not depend on the schedule
    o8 = torch.zeros(B * S, 2 * wq, device="cuda", dtype=tdt)
    assert lib.lr_op_attention_split(P(qkv2), P(qkv2), P(qkv2), P(o8), P(mask if causal else None), P(kmin if causal else None),
                                     2 * W, 2 * wq, 0, wq, wq + wkv, W, wq, B, S, H, hd, 1 if causal else 0, group, scale, code, stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(o8[valid], outs[8.0][valid])
    monkeypatch.setenv("LR_ATT_PINGPONG", str(1 - pingpong))
    o0 = torch.zeros(B * S, 2 * wq, device="cuda", dtype=tdt)
    assert lib.lr_op_attention_split_ex(P(qkv2), P(qkv2), P(qkv2), P(o0), P(mask if causal else None), P(kmin if causal else None),
                                        2 * W, 2 * wq, 0, wq, wq + wkv, W, wq, B, S, H, hd, 1 if causal else 0, group, scale, 0.0, code, stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(o0[valid], outs[0.0][valid])
    for bad in (-1.0, 16.0, float("nan")):
        assert lib.lr_op_attention_split_ex(P(qkv2), P(qkv2), P(qkv2), P(o0), P(None), P(None), 2 * W, 2 * wq, 0, wq, wq + wkv, W, wq, B, S, H, hd,
                                            0, group, scale, bad, code, stream()) != 0


@pytest.mark.parametrize("tile", [4, 6])
@pytest.mark.parametrize("shape", [(8192, 8192, 512), (5284, 9216, 3072), (8192, 8192, 128)])
def test_gemm_persistent_walk_many_tiles(lib, tile, shape):
    """More output tiles than CUs: every workgroup of the persistent kernel walks several tiles, and the ring, the fragment
    prefetch of variant 6 and the staging epilogue are re-entered back to back (a race here shows up as scattered garbage,
    not as a small error).  Checked against torch's fp32 matmul of the same f16 operands."""
    M, N, K = shape
    A = rnd((M, K), 81).to(torch.float16)
    W = rnd((N, K), 82, 0.05).to(torch.float16)
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    assert lib.lr_op_gemm_bt(P(A), P(W), P(out), P(None), M, N, K, K, K, N, L.EPI_OUT_F32, 0, L.LR_DT_F16, tile, stream()) == 0
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t()
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < 2e-5 * ref.abs().max().item() * max(1.0, (K / 512) ** 0.5)


def test_gemm_dynamic_tile_claims_on_two_streams(lib):
    """The persistent kernel claims its tiles from per-XCD counters that live with the STREAM (one set per stream, left zero by the
    last workgroup of every launch).  Two streams launching back to back, many times, different shapes: every result bit-equal to the
    same launch made alone (a stale or shared counter would drop or repeat tiles)."""
    M, K = 5284, 1024
    A = rnd((M, K), 211).to(torch.float16)
    Ws = [rnd((n, K), 212 + i, 0.05).to(torch.float16) for i, n in enumerate((2048, 3072))]
    ref = []
    for W in Ws:
        out = torch.zeros(M, W.shape[0], device="cuda", dtype=torch.float32)
        assert lib.lr_op_gemm_bt(P(A), P(W), P(out), P(None), M, W.shape[0], K, K, K, W.shape[0], L.EPI_OUT_F32, 0, L.LR_DT_F16, 6, stream()) == 0
        torch.cuda.synchronize()
        ref.append(out.clone())
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[torch.zeros_like(ref[i]) for _ in range(6)] for i in range(2)]
    for rep in range(6):
        for i, st in enumerate(streams):
            W = Ws[i]
            assert lib.lr_op_gemm_bt(P(A), P(W), P(outs[i][rep]), P(None), M, W.shape[0], K, K, K, W.shape[0], L.EPI_OUT_F32, 0, L.LR_DT_F16, 6,
                                     C.c_void_p(st.cuda_stream)) == 0
    torch.cuda.synchronize()
    for i in range(2):
        for rep in range(6):
            assert torch.equal(outs[i][rep], ref[i]), (i, rep)


def test_gemm_e4m3_residual_pass_matches_fp64(lib):
    """Split-operand GEMM with the residual pass in e4m3 (precise == 2): A_hi W^T on the f16 instruction + A_lo8 W8^T on the scaled
    e4m3 instruction in the same accumulators.  ~8e-6 of the output scale against fp64 on the un-rounded A (single pass: 2e-4)."""
    code, tdt = L.LR_DT_F16, torch.float16
    for (M, N, K, rows_scaled) in [(700, 512, 384, False), (4100, 768, 3072, True), (8192, 8192, 512, True)]:     # last: 1024 tiles > 256 CUs
        A32 = rnd((M, K), 161, 0.7)
        if rows_scaled:
            A32 = A32 * torch.exp2(torch.randint(-6, 7, (M, 1), generator=torch.Generator().manual_seed(5)).float()).cuda()
        W = rnd((N, K), 162, 0.05).to(torch.bfloat16).to(tdt)
        bias = rnd((N,), 163)
        hi, lo = _split(A32, tdt)
        A2 = torch.cat([hi, lo], dim=1).contiguous()
        ref = (A32.double() @ W.double().t() + bias.double()).float()
        W8 = torch.zeros(N, K, device="cuda", dtype=tdt)
        aexp = torch.full((lib.lr_op_lo8_scratch_bytes(M, K),), 127, dtype=torch.uint8, device="cuda")
        wexp = C.c_int(0)
        out = torch.empty(M, N, device="cuda", dtype=torch.float32)
        A_work = A2.clone()
        assert lib.lr_op_gemm_bt_mixed(P(A_work), P(W), P(W8), P(aexp), P(out), P(bias), M, N, K, L.EPI_OUT_F32, 0, code, 3, C.byref(wexp), stream()) == 0
        torch.cuda.synchronize()
        scale = ref.abs().max().item()
        err = (out - ref).abs().max().item()
        single = torch.empty(M, N, device="cuda", dtype=torch.float32)
        assert lib.lr_op_gemm_bt(P(hi), P(W), P(single), P(bias), M, N, K, K, K, N, L.EPI_OUT_F32, 0, code, 6, stream()) == 0
        torch.cuda.synchronize()
        e1 = (single - ref).abs().max().item()
        assert err < 3e-5 * scale and err < e1 / 8, (M, N, K, err / scale, e1 / scale)
        # the residual half now holds e4m3 bytes with one power-of-two scale per (row, 128-column block), the scales grouped by
        # 4 K-tiles and 256-row tile in the consuming kernel's lane order (csrc/common.h lo8_scale_at): decode, compare with the f16
        # residuals -- every element within 2^-4 of its BLOCK's maximum
        lo8 = A_work[:, K:].contiguous().view(torch.uint8)[:, :K].contiguous().view(torch.float8_e4m3fn).float()
        Mb, nkb = (M + 255) // 256, K // 128
        planes = aexp[: Mb * 1024 * ((nkb + 3) // 4)].view((nkb + 3) // 4, Mb, 4, 256).permute(0, 2, 1, 3).reshape(-1, Mb, 256)[:nkb]   # [kblock, Mb, 256]
        r = torch.arange(256, device="cuda")
        sidx = ((((r >> 6) & 1) * 16 + (r & 15)) * 2 + (r >> 7)) * 4 + ((r >> 4) & 3)
        E = planes[:, :, sidx].reshape(nkb, Mb * 256)[:, :M].t().float()                  # [M, K / 128]
        dec = (lo8.view(M, K // 128, 128) * torch.exp2(E - 127)[:, :, None]).view(M, K)
        ref_lo = lo.float()
        bmax = ref_lo.view(M, K // 128, 128).abs().amax(dim=2, keepdim=True).expand(M, K // 128, 128).reshape(M, K)
        assert ((dec - ref_lo).abs() <= 2.0 ** -4 * bmax + 1e-30).all()
        print(f"[e4m3 residual pass, block scales] M={M} N={N} K={K}: err {err / scale:.2e} of the output scale (single pass {e1 / scale:.2e})")
        assert torch.equal(A_work[:, :K], hi)                                  # the hi half is untouched
        # operand-typed output with GELU: [hi | lo]
        o2 = torch.zeros(M, 2 * N, device="cuda", dtype=tdt)
        A_work = A2.clone()
        assert lib.lr_op_gemm_bt_mixed(P(A_work), P(W), P(W8), P(aexp), P(o2), P(bias), M, N, K, L.EPI_OUT_OP, L.ACT_GELU_ERF, code, 2, C.byref(wexp), stream()) == 0
        g = torch.nn.functional.gelu(ref)
        got = o2[:, :N].float() + o2[:, N:].float()
        assert (got - g).abs().max().item() < 4e-5 * g.abs().max().item()


def test_gemm_e4m3_residual_pass_more_scale_groups_than_lds_slots(lib):
    """K = 16640: 130 residual K-tiles = 33 scale groups, one more than the 32 slots the kernel keeps behind its LDS ring, so the last
    group is fetched late into the slot of the first (the Qwen2.5-VL down projection, K = 18944, is the path's only such GEMM)."""
    code, tdt = L.LR_DT_F16, torch.float16
    M, N, K = 300, 256, 16640
    A32 = rnd((M, K), 201, 0.7)
    W = rnd((N, K), 202, 0.02).to(torch.bfloat16).to(tdt)
    hi, lo = _split(A32, tdt)
    A2 = torch.cat([hi, lo], dim=1).contiguous()
    ref = (A32.double() @ W.double().t()).float()
    scr = torch.full((lib.lr_op_lo8_scratch_bytes(M, K),), 127, dtype=torch.uint8, device="cuda")
    W8 = torch.zeros_like(W)
    we = C.c_int(0)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    assert lib.lr_op_gemm_bt_mixed(P(A2), P(W), P(W8), P(scr), P(out), None, M, N, K, L.EPI_OUT_F32, 0, code, 3, C.byref(we), stream()) == 0
    single = torch.empty(M, N, device="cuda", dtype=torch.float32)
    assert lib.lr_op_gemm_bt(P(hi), P(W), P(single), None, M, N, K, K, K, N, L.EPI_OUT_F32, 0, code, 6, stream()) == 0
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    err, e1 = (out - ref).abs().max().item(), (single - ref).abs().max().item()
    print(f"[33 scale groups] err {err / scale:.2e} (single pass {e1 / scale:.2e})")
    assert err < 3e-5 * scale and err < e1 / 6


def test_gemm_e4m3_residuals_written_by_the_producing_epilogue(lib):
    """Two chained default-mode GEMMs, h = gelu(x W1^T + b) (or silu(gate) * up) then y = h W2^T: the first one's epilogue writes h's
    residual half in the one-byte block-scaled form itself (flag 32), the second reads it as it is (no in-place pass).  The chain
    against fp64 on un-rounded operands, as close as with the in-place encoder in between; h's bytes decode to within 2^-4 of their
    block maximum of the exact residuals; ragged M (rows beyond the last full 256-row tile)."""
    code, tdt = L.LR_DT_F16, torch.float16
    for (M, K, N1, N2, epi) in [(700, 384, 512, 256, L.EPI_OUT_OP), (4100, 1024, 2048, 768, L.EPI_SWIGLU_OP), (8190, 512, 4096, 512, L.EPI_OUT_OP)]:
        x32 = rnd((M, K), 191, 0.7) * torch.exp2(torch.randint(-4, 5, (M, 1), generator=torch.Generator().manual_seed(7)).float()).cuda()
        W1 = rnd((N1, K), 192, 0.05).to(torch.bfloat16).to(tdt)
        b1 = rnd((N1,), 193) if epi == L.EPI_OUT_OP else None
        Kh = N1 if epi == L.EPI_OUT_OP else N1 // 2
        W2 = rnd((N2, Kh), 194, 0.05).to(torch.bfloat16).to(tdt)
        hi, lo = _split(x32, tdt)
        pre = x32.double() @ W1.double().t()
        if epi == L.EPI_OUT_OP:
            h_ref = torch.nn.functional.gelu(pre + b1.double())
        else:      # weight rows interleaved in blocks of 32: [gate 0..31 | up 0..31 | gate 32..63 | ...]
            pr = pre.view(M, N1 // 64, 2, 32)
            h_ref = (torch.nn.functional.silu(pr[:, :, 0]) * pr[:, :, 1]).reshape(M, Kh)
        ref = (h_ref @ W2.double().t()).float()
        scale = ref.abs().max().item()
        outs = {}
        for fused in (True, False):
            A = torch.cat([hi, lo], dim=1).contiguous()
            s1 = torch.full((lib.lr_op_lo8_scratch_bytes(M, K) + lib.lr_op_lo8_scratch_bytes(M, Kh),), 127, dtype=torch.uint8, device="cuda")
            W81, W82 = torch.zeros_like(W1), torch.zeros_like(W2)
            we1, we2 = C.c_int(0), C.c_int(0)
            h = torch.zeros(M, 2 * Kh, device="cuda", dtype=tdt)
            y = torch.empty(M, N2, device="cuda", dtype=torch.float32)
            act = L.ACT_GELU_ERF if epi == L.EPI_OUT_OP else 0
            assert lib.lr_op_gemm_bt_mixed(P(A), P(W1), P(W81), P(s1), P(h), P(b1), M, N1, K, epi, act, code, 3 | (32 if fused else 0), C.byref(we1), stream()) == 0
            s2 = s1[lib.lr_op_lo8_scratch_bytes(M, K):]
            if fused:
                torch.cuda.synchronize()
                h_bytes, h_scales = h.clone(), s2.clone()
            assert lib.lr_op_gemm_bt_mixed(P(h), P(W2), P(W82), P(s2), P(y), None, M, N2, Kh, L.EPI_OUT_F32, 0, code, 1 if fused else 3, C.byref(we2), stream()) == 0
            torch.cuda.synchronize()
            outs[fused] = (y - ref).abs().max().item() / scale
        print(f"[fused residual encode] M={M} K={K} N1={N1} N2={N2} epi={epi}: chain err fused {outs[True]:.2e}, in-place {outs[False]:.2e}")
        assert outs[True] < 4e-5 and outs[True] < 1.5 * outs[False] + 2e-6
        # decode h's residual half
        hh = h_bytes[:, :Kh].float()
        lo8 = h_bytes[:, Kh:].contiguous().view(torch.uint8)[:, :Kh].contiguous().view(torch.float8_e4m3fn).float()
        Mb, nkb = (M + 255) // 256, Kh // 128
        planes = h_scales[: Mb * 1024 * ((nkb + 3) // 4)].view((nkb + 3) // 4, Mb, 4, 256).permute(0, 2, 1, 3).reshape(-1, Mb, 256)[:nkb]
        r = torch.arange(256, device="cuda")
        sidx = ((((r >> 6) & 1) * 16 + (r & 15)) * 2 + (r >> 7)) * 4 + ((r >> 4) & 3)
        E = planes[:, :, sidx].reshape(nkb, Mb * 256)[:, :M].t().float()
        dec = (lo8.view(M, nkb, 128) * torch.exp2(E - 127)[:, :, None]).view(M, Kh)
        # the epilogue's own fp32 h is not visible; hi + decoded lo must sit closer to the fp64 h than hi alone by ~2^-4 of the f16 step
        e_hi = (hh.double() - h_ref).abs()
        e_full = (hh.double() + dec.double() - h_ref).abs()
        assert e_full.max().item() < 2e-5 * h_ref.abs().max().item() and e_full.mean().item() < e_hi.mean().item() / 6


def test_gemm_w8a8_e4m3(lib):
    """W8A8 building block (BASELINE configs[4]): the row quantiser against torch.float8_e4m3fn on the CPU (bytes and scales equal),
    the e4m3 GEMM against the dequantised fp32 product."""
    code, tdt = L.LR_DT_F16, torch.float16
    for (M, N, K) in [(300, 256, 128), (4100, 768, 3072)]:
        x = (rnd((M, K), 171) * (torch.rand(M, 1, generator=torch.Generator().manual_seed(3)).cuda() * 3)).to(tdt)
        w = rnd((N, K), 172, 0.02).to(tdt)
        qs = []
        for t in (x, w):
            q = torch.empty(t.shape, dtype=torch.uint8, device="cuda")
            s = torch.empty(t.shape[0], dtype=torch.float32, device="cuda")
            assert lib.lr_op_quantize_rows_fp8(P(t), t.shape[0], K, K, P(q), P(s), code, stream()) == 0
            torch.cuda.synchronize()
            tc = t.float().cpu()
            amax = tc.abs().amax(1)
            sref = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
            qref = (tc / sref[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
            assert torch.equal(s.cpu(), sref) and torch.equal(q.cpu(), qref)
            qs.append((q, s))
        (xq, xs), (wq, ws) = qs
        bias = rnd((N,), 173)
        out = torch.zeros(M, N, device="cuda")
        assert lib.lr_op_gemm_fp8(P(xq), P(xs), P(wq), P(ws), P(out), P(bias), M, N, K, N, L.EPI_OUT_F32, 0, code, stream()) == 0
        torch.cuda.synchronize()
        ref = (xq.view(torch.float8_e4m3fn).float() @ wq.view(torch.float8_e4m3fn).float().T) * xs[:, None] * ws[None, :] + bias
        assert (out - ref).abs().max().item() < 1e-4 * ref.abs().max().item()
    assert lib.lr_op_gemm_fp8(P(xq), P(xs), P(wq), P(ws), P(out), None, M, N, 100, N, L.EPI_OUT_F32, 0, code, stream()) != 0     # K % 128


def test_gemm_e4m3_residual_pass_with_inexact_weights(lib):
    """Weights that are not exact in f16 (fp32-valued: a merged LoRA adapter): W = W_hi + W_lo, and the e4m3 form adds a third
    segment A_hi(e4m3) x e4m3(W_lo)^T.  Against fp64 on the un-rounded A and W."""
    code, tdt = L.LR_DT_F16, torch.float16
    for (M, N, K) in [(700, 512, 384), (4100, 768, 3072)]:
        A32 = rnd((M, K), 181, 0.7)
        W32 = rnd((N, K), 182, 0.05)
        Whi = W32.to(tdt)
        Wlo = (W32 - Whi.float()).to(tdt)
        hi, lo = _split(A32, tdt)
        A2 = torch.cat([hi, lo], dim=1).contiguous()
        ref = (A32.double() @ W32.double().t()).float()
        scale = ref.abs().max().item()
        twin = Wlo.clone().contiguous()
        aexp = torch.full((lib.lr_op_lo8_scratch_bytes(M, K),), 127, dtype=torch.uint8, device="cuda")
        wexp = (C.c_int * 2)(0, 0)
        out = torch.empty(M, N, device="cuda", dtype=torch.float32)
        assert lib.lr_op_gemm_bt_mixed(P(A2), P(Whi), P(twin), P(aexp), P(out), None, M, N, K, L.EPI_OUT_F32, 0, code, 11, wexp, stream()) == 0
        torch.cuda.synchronize()
        err = (out - ref).abs().max().item()
        exact_w = torch.empty(M, N, device="cuda", dtype=torch.float32)           # the same launch without the third segment: W_lo ignored
        A3 = torch.cat([hi, lo], dim=1).contiguous()
        twin2 = torch.zeros_like(Whi)
        assert lib.lr_op_gemm_bt_mixed(P(A3), P(Whi), P(twin2), P(aexp), P(exact_w), None, M, N, K, L.EPI_OUT_F32, 0, code, 3, wexp, stream()) == 0
        torch.cuda.synchronize()
        e_nolo = (exact_w - ref).abs().max().item()
        assert err < 3e-5 * scale and err < e_nolo / 4, (M, N, K, err / scale, e_nolo / scale)


def test_gemm_narrow_tiles_give_the_full_tile_bits(lib, monkeypatch):
    """Round 4: problems that do not fill the chip with 256 x 256 tiles run 128 x 256 ones, the adapters' t = x A^T (N = rank <= 128)
    256 x 128 ones (csrc/gemm8.hip NW).  An element's K order, matrix instructions and scales are the full tile's, so the bits are:
    every epilogue, 16-bit / split / e4m3-residual forms, ragged M, against LR_GEMM_NARROW=0 (256 x 256 only)."""
    code, tdt = L.LR_DT_F16, torch.float16

    def both(fn):
        outs = []
        for env in ("0", "1"):
            monkeypatch.setenv("LR_GEMM_NARROW", env)
            outs.append(fn())
            torch.cuda.synchronize()
        monkeypatch.delenv("LR_GEMM_NARROW")
        outs.append(fn())                       # the launcher's own choice
        torch.cuda.synchronize()
        return outs

    def same(outs, what):
        for o in outs[1:]:
            for a, b in zip(outs[0], o):
                assert torch.equal(a, b), what

    for (M, N, K) in [(300, 512, 384), (2642, 3072, 1024), (130, 256, 256), (1, 768, 128)]:
        A32 = rnd((M, K), 301, 0.7) * torch.exp2(torch.randint(-4, 5, (M, 1), generator=torch.Generator().manual_seed(9)).float()).cuda()
        W = rnd((N, K), 302, 0.05).to(torch.bfloat16).to(tdt)
        bias = rnd((N,), 303)
        hi, lo = _split(A32, tdt)
        A2 = torch.cat([hi, lo], dim=1).contiguous()
        res = rnd((M, N), 304)

        def plain():
            o1 = torch.empty(M, N, device="cuda", dtype=torch.float32)
            assert lib.lr_op_gemm_bt(P(hi), P(W), P(o1), P(bias), M, N, K, K, K, N, L.EPI_OUT_F32, 0, code, 6, stream()) == 0
            o2 = res.clone()
            assert lib.lr_op_gemm_bt(P(hi), P(W), P(o2), P(None), M, N, K, K, K, N, L.EPI_RESADD_F32, 0, code, 6, stream()) == 0
            o3 = torch.zeros(M, N, device="cuda", dtype=tdt)
            assert lib.lr_op_gemm_bt(P(hi), P(W), P(o3), P(bias), M, N, K, K, K, N, L.EPI_OUT_OP, L.ACT_QUICK_GELU, code, 6, stream()) == 0
            o4 = torch.zeros(M, N // 2, device="cuda", dtype=tdt)
            assert lib.lr_op_gemm_bt(P(hi), P(W), P(o4), P(None), M, N, K, K, K, N // 2, L.EPI_SWIGLU_OP, 0, code, 6, stream()) == 0
            return o1, o2, o3, o4
        same(both(plain), ("16-bit", M, N, K))

        def split():
            o1 = torch.empty(M, N, device="cuda", dtype=torch.float32)
            assert lib.lr_op_gemm_bt_split(P(A2), P(W), P(o1), P(bias), M, N, K, L.EPI_OUT_F32, 0, code, 6, stream()) == 0
            o2 = torch.zeros(M, 2 * N, device="cuda", dtype=tdt)
            assert lib.lr_op_gemm_bt_split(P(A2), P(W), P(o2), P(bias), M, N, K, L.EPI_OUT_OP, L.ACT_GELU_ERF, code, 6, stream()) == 0
            return o1, o2
        same(both(split), ("split", M, N, K))

        if K % 128 == 0:
            def mixed():
                outs = []
                for epi, act, fl in ((L.EPI_OUT_F32, 0, 3), (L.EPI_RESADD_F32, 0, 3), (L.EPI_OUT_OP, L.ACT_GELU_ERF, 3 | 32), (L.EPI_SWIGLU_OP, 0, 3 | 32)):
                    if (epi == L.EPI_SWIGLU_OP and N % 256) or (fl & 32 and (N // (2 if epi == L.EPI_SWIGLU_OP else 1)) % 128):
                        continue
                    Aw = A2.clone()
                    No = N // 2 if epi == L.EPI_SWIGLU_OP else N
                    sc = torch.full((lib.lr_op_lo8_scratch_bytes(M, K) + lib.lr_op_lo8_scratch_bytes(M, No),), 127, dtype=torch.uint8, device="cuda")
                    W8 = torch.zeros_like(W)
                    we = C.c_int(0)
                    o = res.clone() if epi in (L.EPI_OUT_F32, L.EPI_RESADD_F32) else torch.zeros(M, 2 * No, device="cuda", dtype=tdt)
                    assert lib.lr_op_gemm_bt_mixed(P(Aw), P(W), P(W8), P(sc), P(o), P(bias) if epi != L.EPI_SWIGLU_OP else None, M, N, K, epi, act, code, fl,
                                                   C.byref(we), stream()) == 0
                    outs += [o, sc]
                return outs
            same(both(mixed), ("e4m3 residual pass", M, N, K))

    # fused RoPE epilogue
    M, D, hd, K = 333, 256, 64, 256
    N = 3 * D
    A = rnd((M, K), 311).to(tdt)
    W = rnd((N, K), 312, 0.05).to(tdt)
    ang = rnd((M, hd // 2), 313, 3.0)
    cs = torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous()

    def rope():
        o = torch.zeros(M, N, device="cuda", dtype=tdt)
        assert lib.lr_op_gemm_rope(P(A), P(W), P(o), P(None), P(cs), M, N, K, 2 * D, hd, code, 5, stream()) == 0
        return (o,)
    same(both(rope), "rope")

    # the adapters' skinny GEMM in the default form: N = rank (256 x 128 tiles unless switched off), many row tiles, ragged M
    for (M, N, K) in [(5000, 128, 1024), (70000, 128, 3072), (700, 64, 384)]:
        A32 = rnd((M, K), 321, 0.7)
        W = rnd((N, K), 322, 0.05).to(torch.bfloat16).to(tdt)
        hi, lo = _split(A32, tdt)
        A2 = torch.cat([hi, lo], dim=1).contiguous()

        def skinny():
            Aw = A2.clone()
            sc = torch.full((lib.lr_op_lo8_scratch_bytes(M, K) + lib.lr_op_lo8_scratch_bytes(M, 128),), 127, dtype=torch.uint8, device="cuda")
            W8 = torch.zeros_like(W)
            we = C.c_int(0)
            o = torch.zeros(M, 2 * N, device="cuda", dtype=tdt)
            assert lib.lr_op_gemm_bt_mixed(P(Aw), P(W), P(W8), P(sc), P(o), None, M, N, K, L.EPI_OUT_OP, 0, code, 3, C.byref(we), stream()) == 0
            return (o,)
        outs = both(skinny)
        same(outs, ("skinny", M, N, K))
        ref = (A32.double() @ W.double().t()).float()
        got = outs[0][0][:, :N].float() + outs[0][0][:, N:].float()
        assert (got - ref).abs().max().item() < 4e-5 * ref.abs().max().item()


def test_attention_workgroup_order_and_query_tiling_are_bit_neutral(lib, monkeypatch):
    """Round 4: dense attention launches walk their workgroups in an XCD-aware order (all query tiles of a (sequence, kv head) on one
    XCD, heaviest first) and the long-sequence causal kernel shifts its query tiles towards the end of the sequence.  Neither may
    change a bit: a query's arithmetic does not depend on the workgroup, wave or lane that holds it.  Round 5 adds the kernel FORM: the
    ping-pong schedule and the plain per-tile loop give the same bits (so a row scored in a long batch and alone in a short one differ
    by nothing the attention kernels do)."""
    def run(B, S, H, hd, causal, Hkv, pads):
        Wd = (H + 2 * Hkv) * hd
        g = torch.Generator(device="cuda").manual_seed(S + hd)
        qkv = torch.cat([torch.randn(B * S, Wd, device="cuda", generator=g).half(),
                         (torch.randn(B * S, Wd, device="cuda", generator=g) * 2.0 ** -12).half()], dim=1).contiguous()
        mask = kmin = None
        if causal:
            mask = torch.ones(B, S, dtype=torch.int64, device="cuda")
            kmin = torch.zeros(B, dtype=torch.int32, device="cuda")
            for b, p_ in enumerate(pads):
                mask[b, :p_] = 0
                kmin[b] = p_
        outs = []
        # (round 5: LR_ATT_PINGPONG = 0 runs long sequences on the plain per-tile loop instead of the ping-pong schedule -- another
        #  wave-to-query assignment, tile walk and barrier structure, the same per-query arithmetic, lazy reference maximum included)
        for env in ({}, {"LR_ATT_XCD_ORDER": "0"}, {"LR_ATT_QSHIFT": "0"}, {"LR_ATT_XCD_ORDER": "0", "LR_ATT_QSHIFT": "0"}, {"LR_ATT_PINGPONG": "0"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            out = torch.zeros(B * S, 2 * H * hd, device="cuda", dtype=torch.float16)
            assert lib.lr_op_attention_split(P(qkv), P(qkv), P(qkv), P(out), P(mask), P(kmin), 2 * Wd, 2 * H * hd, 0, H * hd, (H + Hkv) * hd, Wd, H * hd,
                                             B, S, H, hd, int(causal), H // Hkv, 1.0 / math.sqrt(hd), L.LR_DT_F16, stream()) == 0
            torch.cuda.synchronize()
            for k in env:
                monkeypatch.delenv(k)
            valid = torch.ones(B * S, dtype=torch.bool, device="cuda")
            for b, p_ in enumerate(pads if causal else []):
                valid[b * S: b * S + p_] = False
            outs.append(out[valid])
        assert all(torch.equal(outs[0], o) for o in outs[1:]), (B, S, H, hd, causal)
        assert outs[0].float().abs().sum().item() > 0

    run(3, 1100, 8, 96, True, 8, [0, 77, 640])          # ping-pong kernel, shifted query tiles, 3 x 8 pairs = 3 groups of 8
    run(5, 1031, 4, 96, True, 4, [5, 0, 300, 64, 1000])  # 20 pairs: the grid is padded to 24
    run(2, 700, 8, 96, True, 8, [0, 33])
    run(9, 577, 16, 64, False, 16, [])                   # CLIP's shape
    run(3, 1313, 8, 128, True, 2, [0, 100, 1])           # GQA: 4 query heads per kv head on one XCD
    run(2, 70, 28, 128, True, 4, [0, 3])


def test_gemm_tile_walk_order_is_bit_neutral(lib, monkeypatch):
    """Round 4: the persistent kernel walks its tiles in bands of 4 tile rows, the XCDs' chunks cut at band boundaries (so that W and the
    A bands in flight fit the Infinity Cache, csrc/gemm8.hip launch8).  A tile's arithmetic does not depend on when or where it runs:
    band heights 8 / 4 / 1, equal chunks, the static walk -- the same bits, on a problem large enough for the band-aligned chunks
    (157 tile rows x 8 tile columns), ragged M, every operand form."""
    code, tdt = L.LR_DT_F16, torch.float16
    M, N, K = 40100, 2048, 512
    A32 = rnd((M, K), 401, 0.7)
    W = rnd((N, K), 402, 0.05).to(torch.bfloat16).to(tdt)
    bias = rnd((N,), 403)
    hi, lo = _split(A32, tdt)
    A2 = torch.cat([hi, lo], dim=1).contiguous()
    res = rnd((M, N), 404)

    def run():
        o1 = res.clone()
        assert lib.lr_op_gemm_bt(P(hi), P(W), P(o1), P(None), M, N, K, K, K, N, L.EPI_RESADD_F32, 0, code, 6, stream()) == 0
        o2 = torch.zeros(M, 2 * N, device="cuda", dtype=tdt)
        assert lib.lr_op_gemm_bt_split(P(A2), P(W), P(o2), P(bias), M, N, K, L.EPI_OUT_OP, L.ACT_QUICK_GELU, code, 6, stream()) == 0
        Aw = A2.clone()
        sc = torch.full((lib.lr_op_lo8_scratch_bytes(M, K) + lib.lr_op_lo8_scratch_bytes(M, N // 2),), 127, dtype=torch.uint8, device="cuda")
        W8 = torch.zeros_like(W)
        we = C.c_int(0)
        o3 = torch.zeros(M, N, device="cuda", dtype=tdt)
        assert lib.lr_op_gemm_bt_mixed(P(Aw), P(W), P(W8), P(sc), P(o3), None, M, N, K, L.EPI_SWIGLU_OP, 0, code, 3 | 32, C.byref(we), stream()) == 0
        torch.cuda.synchronize()
        return o1, o2, o3, sc

    base = run()
    ref = (A32.double() @ W.double().t()).float() + res
    assert (base[0] - ref).abs().max().item() < 2e-3 * ref.abs().max().item()         # (single-pass operands: sanity of the walk itself)
    for env in ({"LR_GEMM_GM": "8"}, {"LR_GEMM_GM": "1"}, {"LR_GEMM_BANDCHUNK": "0"}, {"LR_GEMM_GM": "8", "LR_GEMM_BANDCHUNK": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = run()
        for k in env:
            monkeypatch.delenv(k)
        for a, b in zip(base, got):
            assert torch.equal(a, b), env
