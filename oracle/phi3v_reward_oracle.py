"""CPU oracle for the Phi-3.5-V reward-scoring path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain torch-fp32 restatement of the reference's `custom_forward` for model_type == 'phi3v'.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (llava-reward_amd/) never does and fails loudly without its HIP library.

Parity pin: the reference holds no tests or golden vectors for this path, so
this restatement is pinned against outputs of the reference ITSELF, produced in the build
container by tests/golden/make_goldens.py (which imports /root/reference) and committed as
tests/golden/*.json.  tests/test_oracle_vs_golden.py checks it (fp32, tolerance 2e-5).

Every function cites the reference lines it restates (paths relative to /root/reference):
  RW  = llava_reward/models/rw_model_general_preference.py
  PHI = llava_reward/models/base_mllm/phi3_v/modeling_phi3_v.py
  UT  = llava_reward/utils/utils.py
  CLIP= transformers/models/clip/modeling_clip.py (pinned 4.50.0 in requirements.txt:9; third
        party, not vendored: embeddings + pre-LN encoder layer with quick_gelu MLP)
Weights are a dict name -> fp32 torch tensor keyed by the reference's state_dict names.
`opr` (operand rounding) is identity for the oracle proper; passing e.g. a bf16 round-trip
emulates a kernel that rounds GEMM/attention operands and keeps everything else in fp32.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

CLIP_PREFIX = "model.vision_embed_tokens.img_processor.vision_model."
EMB_PREFIX = "model.vision_embed_tokens."

Ident = lambda x: x  # noqa: E731


def _lin(x, w, b=None, opr=Ident):
    lq = getattr(opr, "lin", None)           # W8A8 emulation: an extra per-row quantisation of both GEMM operands
    if lq is not None and x.shape[-1] % 128 == 0:
        return F.linear(lq(opr(x)), lq(opr(w)), b)
    return F.linear(opr(x), opr(w), b)


def proj(W, name: str, x, b=None, opr=Ident):
    """Linear layer `name`, plus its LoRA adapter when W carries one, evaluated UN-MERGED as the reference runs it
    (eval/reward_adaptor_loader.py:44-45 model.load_adapter; third party peft==0.13.2, tuners/lora/layer.py Linear.forward:
    result = base_layer(x) + lora_B(lora_A(dropout(x))) * scaling, dropout inactive in eval, scaling = lora_alpha / r).
    W["lora_scaling"] holds the scaling (absent = 1.0: lora_B already multiplied by it, as the engine takes it)."""
    y = _lin(x, W[name + ".weight"], b, opr)
    a = W.get(name + ".lora_A.weight")
    if a is not None:
        y = y + F.linear(F.linear(x, a), W[name + ".lora_B.weight"]) * float(W.get("lora_scaling", 1.0))
    return y


class W8A8Round:
    """Operand rounding of the W8A8 mode (lr_model_desc.w8a8, BASELINE configs[4]): activations are stored in f16 (`act`), and in
    front of every GEMM with K % 128 == 0 the rows of both operands go to OCP e4m3 with one fp32 scale per row, max|x| / 448
    (csrc/rowops.hip quantize_rows_fp8_kernel).  Attention contractions see the f16 rounding only."""

    def __init__(self, act):
        self.act = act

    def __call__(self, t):
        return self.act(t)

    @staticmethod
    def lin(t):
        amax = t.abs().amax(dim=-1, keepdim=True)
        s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
        return (t / s).to(torch.float8_e4m3fn).float() * s


# ---------------------------------------------------------------------------------- CLIP tower
def clip_tower(W: Dict[str, torch.Tensor], pixels: torch.Tensor, ccfg, opr=Ident, prefix: str = CLIP_PREFIX) -> torch.Tensor:
    """CLIP-ViT patch features of `pixels [N,3,336,336]` -> [N, 576, hidden].

    Restates UT:266-273 (patched get_img_features == hidden_states[-2][:,1:] of PHI:208-219):
    embeddings -> pre_layrnorm -> first `layers_used` pre-LN encoder layers, CLS dropped.
    CLIP embeddings: Conv2d(3,H,k=14,s=14,bias=False) patches, class token first, learned
    position embedding added; encoder layer: x += out_proj(MHA(LN1 x)); x += fc2(quick_gelu(fc1(LN2 x)))
    with attention scale head_dim**-0.5 and quick_gelu(x) = x*sigmoid(1.702x) (PHI:68-83 config)."""
    p = prefix
    N = pixels.shape[0]
    H, nh, hd = ccfg.hidden, ccfg.heads, ccfg.head_dim
    wpe = W[p + "embeddings.patch_embedding.weight"]
    if getattr(opr, "lin", None) is not None:     # W8A8 emulation: the conv as the GEMM the engine runs (rows zero-padded to 640)
        pt = F.unfold(opr(pixels), kernel_size=ccfg.patch, stride=ccfg.patch).transpose(1, 2)      # [N,576,588], (c, ky, kx) order
        kp = (pt.shape[-1] + 127) // 128 * 128
        x = _lin(F.pad(pt, (0, kp - pt.shape[-1])), F.pad(wpe.reshape(H, -1), (0, kp - pt.shape[-1])), None, opr)
    else:
        x = F.conv2d(opr(pixels), opr(wpe), stride=ccfg.patch)                  # [N,H,24,24]
        x = x.flatten(2).transpose(1, 2)                                         # [N,576,H]
    cls = W[p + "embeddings.class_embedding"].expand(N, 1, H)
    x = torch.cat([cls, x], dim=1) + W[p + "embeddings.position_embedding.weight"]
    x = F.layer_norm(x, (H,), W[p + "pre_layrnorm.weight"], W[p + "pre_layrnorm.bias"], ccfg.ln_eps)
    T = x.shape[1]
    scale = hd ** -0.5
    for l in range(ccfg.layers_used):
        q_ = f"{p}encoder.layers.{l}."
        h = F.layer_norm(x, (H,), W[q_ + "layer_norm1.weight"], W[q_ + "layer_norm1.bias"], ccfg.ln_eps)
        q = _lin(h, W[q_ + "self_attn.q_proj.weight"], W[q_ + "self_attn.q_proj.bias"], opr)
        k = _lin(h, W[q_ + "self_attn.k_proj.weight"], W[q_ + "self_attn.k_proj.bias"], opr)
        v = _lin(h, W[q_ + "self_attn.v_proj.weight"], W[q_ + "self_attn.v_proj.bias"], opr)
        q = q.view(N, T, nh, hd).transpose(1, 2)
        k = k.view(N, T, nh, hd).transpose(1, 2)
        v = v.view(N, T, nh, hd).transpose(1, 2)
        s = torch.matmul(opr(q), opr(k).transpose(2, 3)) * scale
        a = torch.softmax(s, dim=-1)
        o = torch.matmul(opr(a), opr(v)).transpose(1, 2).reshape(N, T, H)
        x = x + _lin(o, W[q_ + "self_attn.out_proj.weight"], W[q_ + "self_attn.out_proj.bias"], opr)
        h = F.layer_norm(x, (H,), W[q_ + "layer_norm2.weight"], W[q_ + "layer_norm2.bias"], ccfg.ln_eps)
        h = _lin(h, W[q_ + "mlp.fc1.weight"], W[q_ + "mlp.fc1.bias"], opr)
        h = h * torch.sigmoid(1.702 * h)
        x = x + _lin(h, W[q_ + "mlp.fc2.weight"], W[q_ + "mlp.fc2.bias"], opr)
    return x[:, 1:]


# ------------------------------------------------------------------------- HD transform + projector
def merge_2x2(feat: torch.Tensor, h_crop: int, w_crop: int) -> torch.Tensor:
    """PHI:305-326 reshape_hd_patches_2x2merge: [h_crop*w_crop, 576, C] -> [h_crop*12, w_crop*12, 4C];
    channel block (di*2+dj) of merged token (i,j) is CLIP patch (2i+di, 2j+dj)."""
    N, L, C = feat.shape
    g = int(round(math.sqrt(L)))
    x = feat.reshape(N, g // 2, 2, g // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(N, g // 2, g // 2, 4 * C)
    x = x.reshape(h_crop, w_crop, g // 2, g // 2, 4 * C).permute(0, 2, 1, 3, 4)
    return x.reshape(h_crop * (g // 2), w_crop * (g // 2), 4 * C)


def add_newline(x: torch.Tensor, sub_gn: torch.Tensor) -> torch.Tensor:
    """PHI:351-362 add_image_newline: append the learned sub_GN vector to every row -> [(h)*(w+1), 4C]."""
    h, w, C = x.shape
    return torch.cat([x, sub_gn.reshape(1, 1, C).expand(h, 1, C)], dim=1).reshape(h * (w + 1), C)


def hd_project(W, feats: torch.Tensor, image_sizes, cfg, opr=Ident):
    """PHI:254-303 hd_feature_transform (order 'sub_glb': [sub crops, glb_GN, global crop]) followed
    by img_projection = Linear(4C,D) -> GELU(erf) -> Linear(D,D) (PHI:172-179).
    feats [B, crops, 576, C] (crop 0 = global).  Returns (proj [sum V, D], per-sample token counts)."""
    e = EMB_PREFIX
    sub_gn, glb_gn = W[e + "sub_GN"], W[e + "glb_GN"]
    rows, counts = [], []
    for i in range(feats.shape[0]):
        h, w = int(image_sizes[i][0]), int(image_sizes[i][1])
        hc, wc = h // 336, w // 336
        sub = add_newline(merge_2x2(feats[i, 1:1 + hc * wc], hc, wc), sub_gn)
        glb = add_newline(merge_2x2(feats[i, :1], 1, 1), sub_gn)
        rows += [sub, glb_gn.reshape(1, -1), glb]
        counts.append(sub.shape[0] + 1 + glb.shape[0])
    x = torch.cat(rows, dim=0)
    x = _lin(x, W[e + "img_projection.0.weight"], W[e + "img_projection.0.bias"], opr)
    x = F.gelu(x)
    x = _lin(x, W[e + "img_projection.2.weight"], W[e + "img_projection.2.bias"], opr)
    return x, counts


# --------------------------------------------------------------------------------------- decoder
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """PHI:377-391 / RW:19-33 Phi3RMSNorm."""
    var = x.to(torch.promote_types(x.dtype, torch.float32)).pow(2).mean(-1, keepdim=True)      # (.float() for fp32 / bf16 inputs)
    return w * (x * torch.rsqrt(var + eps))


def su_rope_cos_sin(position_ids: torch.Tensor, cfg, seq_len: Optional[int] = None, dtype=torch.float32):
    """PHI:446-476 Phi3SuScaledRotaryEmbedding.forward: short factors unless seq_len > original max; the attention layers pass
    seq_len = kv_seq_len = the padded length S (PHI:673, :1081; use_cache=False), so `max(position_ids)+1` (PHI:448) is only the
    fallback for callers that pass none; cos/sin of cat(freqs,freqs) scaled by sqrt(1 + ln(max_pos/orig)/ln(orig))."""
    hd = cfg.head_dim
    seq_len = seq_len or int(position_ids.max()) + 1
    fac = cfg.long_factor if seq_len > cfg.orig_max_pos else cfg.short_factor
    ext = torch.tensor(fac, dtype=dtype)
    inv_shape = torch.arange(0, hd, 2, dtype=torch.int64).to(dtype) / hd
    inv_freq = 1.0 / (ext * cfg.rope_theta ** inv_shape)
    freqs = position_ids[:, :, None].to(dtype) * inv_freq[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    scale = cfg.max_pos / cfg.orig_max_pos
    sf = 1.0 if scale <= 1.0 else math.sqrt(1 + math.log(scale) / math.log(cfg.orig_max_pos))
    return emb.cos() * sf, emb.sin() * sf


def rotate_half(x):
    """PHI:521-525."""
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def causal_padding_mask(attention_mask: torch.Tensor) -> torch.Tensor:
    """PHI:1453-1459 (_prepare_4d_causal_attention_mask; sliding window larger than S): additive
    [B,1,S,S] mask, 0 where key j <= query i and mask[j] == 1, finfo.min elsewhere."""
    B, S = attention_mask.shape
    neg = torch.finfo(torch.float32).min
    causal = torch.tril(torch.ones(S, S, dtype=torch.bool))
    ok = causal[None] & attention_mask.bool()[:, None, :]
    m = torch.zeros(B, S, S, dtype=torch.float32).masked_fill(~ok, neg)
    return m[:, None]


def decoder_layer(W, l: int, x, mask4d, cos, sin, cfg, opr=Ident):
    """PHI:1144-1205 Phi3DecoderLayer with PHI:641-720 eager Phi3Attention and PHI:566-572 Phi3MLP."""
    p = f"model.layers.{l}."
    B, S, D = x.shape
    nh, hd = cfg.heads, cfg.head_dim
    h = rms_norm(x, W[p + "input_layernorm.weight"], cfg.rms_eps)
    qkv = proj(W, p + "self_attn.qkv_proj", h, None, opr)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    q = q.view(B, S, nh, hd).transpose(1, 2)
    k = k.view(B, S, nh, hd).transpose(1, 2)
    v = v.view(B, S, nh, hd).transpose(1, 2)
    c, s = cos[:, None], sin[:, None]
    q = q * c + rotate_half(q) * s                       # PHI:529-553
    k = k * c + rotate_half(k) * s
    att = torch.matmul(opr(q), opr(k).transpose(2, 3)) / math.sqrt(hd) + mask4d
    att = torch.softmax(att, dim=-1, dtype=torch.promote_types(att.dtype, torch.float32))
    o = torch.matmul(opr(att), opr(v)).transpose(1, 2).reshape(B, S, D)
    x = x + proj(W, p + "self_attn.o_proj", o, None, opr)
    h = rms_norm(x, W[p + "post_attention_layernorm.weight"], cfg.rms_eps)
    gu = proj(W, p + "mlp.gate_up_proj", h, None, opr)
    gate, up = gu.chunk(2, dim=-1)
    x = x + proj(W, p + "mlp.down_proj", up * F.silu(gate), None, opr)
    return x


# ------------------------------------------------------------------------------------ full path
@torch.no_grad()
def custom_forward(W: Dict[str, torch.Tensor], cfg, input_ids, attention_mask, pixel_values, image_sizes,
                   training: bool = False, opr: Callable = Ident, taps: Optional[dict] = None, layer_id: int = 32,
                   mean_hidden_state: bool = False, dtype=torch.float32) -> torch.Tensor:
    """RW:334-448 CustomRewardModel.custom_forward, phi3v branch; layer_id == 32 -> last_hidden_state, else
    hidden_states[layer_id] (RW:349-352; PHI:1467-1505: entry k < L is the input of decoder layer k, entry L the final-norm
    output); mean_hidden_state unset.  Returns reward [B,1] (BT) or [B,d] (GPM), fp32.
    `taps`, if given, receives intermediate tensors keyed by stage name.
    `dtype=torch.float64` (with W = UpcastWeights(...)) evaluates the same function in double precision: the yardstick that says how
    far the reference's OWN fp32 arithmetic sits from the exact result on a row (tests/golden/make_fp64_fixture.py)."""
    input_ids = torch.as_tensor(input_ids)
    attention_mask = torch.as_tensor(attention_mask)
    pixel_values = torch.as_tensor(pixel_values, dtype=torch.float32).to(dtype)
    B, S = input_ids.shape
    D = cfg.hidden
    # RW:344-345
    position_ids = attention_mask.long().cumsum(-1) - 1
    position_ids = position_ids.masked_fill(attention_mask == 0, 1)
    # PHI:221-252 Phi3ImageEmbedding.forward
    neg = input_ids < 0
    ids = input_ids.clamp_min(0).clamp_max(cfg.vocab_size)
    x = W["model.embed_tokens.weight"][ids]
    nimg, ncrop = pixel_values.shape[:2]
    feats = clip_tower(W, pixel_values.flatten(0, 1), cfg.clip, opr).reshape(nimg, ncrop, -1, cfg.clip.hidden)
    proj, _ = hd_project(W, feats, image_sizes, cfg, opr)
    counts = neg.sum(dim=1).tolist()                      # PHI:242 bincount(positions[0])
    assert sum(counts) == proj.shape[0], "image-slot count != projected image tokens (PHI:247 index_put)"
    vmax = max(counts)
    ev = torch.zeros(B, vmax, D, dtype=dtype)
    off = 0
    for b, n in enumerate(counts):                        # PHI:243-245 split + zero-pad
        ev[b, :n] = proj[off:off + n]
        off += n
    x = x.clone()
    x[neg] = proj                                         # PHI:247-249 index_put (row-major order)
    if taps is not None:
        taps["clip_out"], taps["proj"], taps["embeds"] = feats, proj, x.clone()
    # PHI:1468-1500 decoder stack + final norm
    mask4d = causal_padding_mask(attention_mask)
    cos, sin = su_rope_cos_sin(position_ids, cfg, seq_len=S, dtype=dtype)
    states = []
    for l in range(cfg.layers):
        states.append(x)
        x = decoder_layer(W, l, x, mask4d, cos, sin, cfg, opr)
        if taps is not None:
            taps[f"layer{l}"] = x.clone()
    h = rms_norm(x, W["model.norm.weight"], cfg.rms_eps)
    states.append(h)
    if layer_id != 32:
        h = states[layer_id]
    # RW:376-386 SkipCA (zero-padded vision rows take part un-masked)
    if cfg.add_cross_attention:
        Q = F.linear(h, W["W_q.weight"])
        K = F.linear(ev, W["W_k.weight"])
        Vv = F.linear(ev, W["W_v.weight"])
        sc = torch.bmm(Q, K.transpose(1, 2)) / math.sqrt(D)
        h = rms_norm(h + torch.bmm(torch.softmax(sc, dim=-1), Vv), W["ca_layernorm.weight"], cfg.ca_eps)
    if taps is not None:
        taps["final_hidden"] = h.clone()
    if mean_hidden_state:
        return F.linear(mean_pool(h, attention_mask), W["value_head.weight"])
    values = F.linear(h, W["value_head.weight"])          # RW:408 / :427  [B,S,d]
    if training:                                          # RW:410-415 / :429-434
        return values[:, -1, :]
    eos = S - 1 - attention_mask.long().fliplr().argmax(dim=1)      # RW:420 / :439
    return values[torch.arange(B), eos, :]


def mean_pool(h: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    """RW:398-406 (`mean_hidden_state`): mask-weighted mean over tokens; reward = value_head(pooled) in train and eval."""
    m = attention_mask.to(h.dtype).unsqueeze(-1)
    return (h * m).sum(dim=1) / m.sum(dim=1).clamp(min=1e-8)


def preference_compute(cfg, chosen: torch.Tensor, reject: torch.Tensor):
    """eval/reward_adaptor_loader.py:174-181."""
    if cfg.is_general_preference and cfg.value_head_dim == 2:
        prod = chosen[:, 0] * reject[:, 1] - chosen[:, 1] * reject[:, 0]
        prob = torch.sigmoid(prod / cfg.general_preference_tau)
    else:
        prob = torch.sigmoid((chosen - reject) / cfg.general_preference_tau).squeeze(-1)
    return prob.float().cpu().numpy()


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def f16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float16).to(torch.float32)


def weights_to_torch(W_np) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(v) for k, v in W_np.items()}


class UpcastWeights(dict):
    """fp32 weights handed out in `dtype` one tensor at a time (a full-size model is 16 GB in fp32: the fp64 run never holds more than
    one up-cast tensor).  For custom_forward(..., dtype=torch.float64)."""

    def __init__(self, W, dtype=torch.float64):
        super().__init__(W)
        self._dtype = dtype

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return v.to(self._dtype) if torch.is_tensor(v) and v.is_floating_point() else v

    def get(self, k, default=None):
        return self[k] if k in self else default
