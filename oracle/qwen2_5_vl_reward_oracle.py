"""CPU oracle for the Qwen2.5-VL reward-scoring path -- TEST INFRASTRUCTURE ONLY.

Restates, in plain torch fp32, what the reference's `custom_forward` does for model_type == 'qwen'
(llava_reward/models/rw_model_general_preference.py:354-371, :387-397, then :407-448):

  * `self.visual(pixel_values, grid_thw)` (:356) -- result unused, not restated;
  * `self.forward(**inputs_batch, output_hidden_states=True)` (:357): ViT -> merger -> image rows scattered
    into the <|image_pad|> slots -> mRoPE decoder; `hidden_states[-1]` is the final-norm output, the lm_head
    logits are computed and never read (not restated);
  * SkipCA AS WRITTEN (:358-371, :387-395): `image_mask = input_ids == 151643` selects the PAD tokens
    (<|endoftext|>), not the image slots, and `vision_src = hidden_states[0]` is the embedding-layer output,
    so every un-masked K/V row is the same vector `wte[151643]`; the softmax over identical scores (masked
    columns at -1e4, whose K/V rows are zero anyway) is uniform and `attn_o = [v_len > 0] * W_v wte[151643]`
    for every query of a row, whatever W_q and W_k hold.  The oracle still goes through scores and softmax
    literally so that this claim is itself checked against the reference.

The backbone is THIRD PARTY, not vendored in /root/reference: transformers (pinned 4.50.0 in
requirements.txt:9; the container has 5.15.0) -- modeling_qwen2_5_vl.py (`Qwen2_5_VisionTransformerPretrainedModel`:
patch-embed conv3d == linear over 1176, window re-ordering, 2-D rotary, per-window / per-image attention,
RMSNorm + biased SwiGLU MLP, 2x2 patch merger; `Qwen2_5_VLModel.get_rope_index`; `Qwen2_5_VLTextModel`:
pre-norm decoder, biased q/k/v, GQA, multimodal RoPE with `mrope_section`, SwiGLU).
Parity pin: tests/golden/ref_qwen_*.json, produced by tests/golden/make_goldens.py from the reference's own
custom_forward running on the container's transformers (three shims, listed there and in DESIGN.md §9).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

from . import phi3v_reward_oracle as po

Ident = po.Ident


def vision_tower(W, vc, pixel_values: torch.Tensor, grid_thw, opr=Ident, taps=None) -> torch.Tensor:
    """Qwen2_5_VisionTransformerPretrainedModel.forward -> pooler_output: [sum t*h*w/4, out_hidden], one row
    per merged token, in the processor's (un-windowed) order."""
    from llava_reward_amd.synth import qwen_patch_positions, qwen_window_index
    N = pixel_values.shape[0]
    unit, hd = vc.merge_unit, vc.head_dim
    x = po._lin(pixel_values, W["visual.patch_embed.proj.weight"].reshape(vc.hidden, -1), None, opr)
    widx, cu_win = qwen_window_index(grid_thw, vc)
    widx = torch.from_numpy(widx)
    x = x.reshape(N // unit, unit, -1)[widx].reshape(N, -1)
    pos = torch.from_numpy(qwen_patch_positions(grid_thw, vc))                                  # [N, 2] (h, w)
    inv = 1.0 / (vc.rope_theta ** (torch.arange(0, hd // 2, 2, dtype=torch.float32) / (hd // 2)))
    fr = (pos[:, :, None].float() * inv).flatten(1)                                             # [N, hd/2]
    fr = fr.reshape(N // unit, unit, -1)[widx].reshape(N, -1)
    emb = torch.cat((fr, fr), dim=-1)
    cos, sin = emb.cos()[:, None, :], emb.sin()[:, None, :]
    # get_vision_cu_seqlens: one segment per (image, frame); still images have t == 1
    cu_full = [0]
    for t, h, w in grid_thw:
        for _ in range(t):
            cu_full.append(cu_full[-1] + h * w)
    for l in range(vc.depth):
        p = f"visual.blocks.{l}."
        cu = cu_full if l in vc.fullatt else cu_win.tolist()
        h = po.rms_norm(x, W[p + "norm1.weight"], vc.eps)
        qkv = po._lin(h, W[p + "attn.qkv.weight"], W[p + "attn.qkv.bias"], opr).reshape(N, 3, vc.heads, hd)
        q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
        q = q * cos + po.rotate_half(q) * sin
        k = k * cos + po.rotate_half(k) * sin
        outs = []
        for a, b in zip(cu[:-1], cu[1:]):
            qs, ks, vs = (t_[a:b].transpose(0, 1) for t_ in (q, k, v))                          # [heads, n, hd]
            att = torch.matmul(opr(qs), opr(ks).transpose(1, 2)) * hd ** -0.5
            att = torch.softmax(att, dim=-1, dtype=torch.float32)
            outs.append(torch.matmul(opr(att), opr(vs)).transpose(0, 1).reshape(b - a, -1))
        o = torch.cat(outs, dim=0)
        x = x + po._lin(o, W[p + "attn.proj.weight"], W[p + "attn.proj.bias"], opr)
        h = po.rms_norm(x, W[p + "norm2.weight"], vc.eps)
        g = po._lin(h, W[p + "mlp.gate_proj.weight"], W[p + "mlp.gate_proj.bias"], opr)
        u = po._lin(h, W[p + "mlp.up_proj.weight"], W[p + "mlp.up_proj.bias"], opr)
        x = x + po._lin(F.silu(g) * u, W[p + "mlp.down_proj.weight"], W[p + "mlp.down_proj.bias"], opr)
        if taps is not None:
            taps[f"vit{l}"] = x.clone()
    h = po.rms_norm(x, W["visual.merger.ln_q.weight"], 1e-6).reshape(N // unit, -1)
    h = F.gelu(po._lin(h, W["visual.merger.mlp.0.weight"], W["visual.merger.mlp.0.bias"], opr))
    h = po._lin(h, W["visual.merger.mlp.2.weight"], W["visual.merger.mlp.2.bias"], opr)
    return h[torch.argsort(widx)]


def mrope_cos_sin(pos3: torch.Tensor, cfg):
    """Qwen2_5_VLRotaryEmbedding + the section selection of apply_multimodal_rotary_pos_emb: frequency i of the
    half head uses the temporal / height / width position according to mrope_section.  -> [B, S, hd] each."""
    hd = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    stream = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(cfg.mrope_section)])
    p = pos3.float().permute(1, 2, 0)                                                           # [B, S, 3]
    fr = p[:, :, stream] * inv[None, None, :]
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos(), emb.sin()


def decoder_layer(W, l, x, mask4d, cos, sin, cfg, opr=Ident):
    """Qwen2_5_VLDecoderLayer (eager attention, repeat_kv for GQA, q/k/v bias, no o bias)."""
    p = f"model.layers.{l}."
    B, S, D = x.shape
    H, KV, hd = cfg.heads, cfg.kv_heads, cfg.head_dim
    h = po.rms_norm(x, W[p + "input_layernorm.weight"], cfg.rms_eps)
    q = po.proj(W, p + "self_attn.q_proj", h, W[p + "self_attn.q_proj.bias"], opr).view(B, S, H, hd).transpose(1, 2)
    k = po.proj(W, p + "self_attn.k_proj", h, W[p + "self_attn.k_proj.bias"], opr).view(B, S, KV, hd).transpose(1, 2)
    v = po.proj(W, p + "self_attn.v_proj", h, W[p + "self_attn.v_proj.bias"], opr).view(B, S, KV, hd).transpose(1, 2)
    c, s = cos[:, None], sin[:, None]
    q = q * c + po.rotate_half(q) * s
    k = k * c + po.rotate_half(k) * s
    k = k.repeat_interleave(H // KV, dim=1)
    v = v.repeat_interleave(H // KV, dim=1)
    att = torch.matmul(opr(q), opr(k).transpose(2, 3)) * hd ** -0.5 + mask4d
    att = torch.softmax(att, dim=-1, dtype=torch.float32)
    o = torch.matmul(opr(att), opr(v)).transpose(1, 2).reshape(B, S, H * hd)
    x = x + po.proj(W, p + "self_attn.o_proj", o, None, opr)
    h = po.rms_norm(x, W[p + "post_attention_layernorm.weight"], cfg.rms_eps)
    gate = po.proj(W, p + "mlp.gate_proj", h, None, opr)
    up = po.proj(W, p + "mlp.up_proj", h, None, opr)
    return x + po.proj(W, p + "mlp.down_proj", F.silu(gate) * up, None, opr)


def skip_ca(W, cfg, last: torch.Tensor, embeds: torch.Tensor, input_ids: torch.Tensor) -> torch.Tensor:
    """rw_model_general_preference.py:358-371 + :387-395, literally."""
    from llava_reward_amd.synth import QWEN_CA_TOKEN_ID
    B, L, H = last.shape
    image_mask = input_ids == QWEN_CA_TOKEN_ID
    vis_lens = image_mask.sum(dim=1)
    max_v = int(vis_lens.max())
    vision_pad = last.new_zeros(B, max_v, H)
    pad_mask = torch.ones(B, max_v, dtype=torch.bool)
    for i in range(B):
        n = int(vis_lens[i])
        vision_pad[i, :n] = embeds[i, image_mask[i]]
        pad_mask[i, :n] = False
    Q_ = F.linear(last, W["W_q.weight"])
    K_ = F.linear(vision_pad, W["W_k.weight"])
    V_ = F.linear(vision_pad, W["W_v.weight"])
    scores = torch.bmm(Q_, K_.transpose(1, 2)) / math.sqrt(H)
    scores = scores.masked_fill(pad_mask.unsqueeze(1), -1e4)
    attn_o = torch.bmm(F.softmax(scores, dim=-1), V_)
    return po.rms_norm(last + attn_o, W["ca_layernorm.weight"], cfg.ca_eps)


@torch.no_grad()
def custom_forward(W: Dict[str, torch.Tensor], cfg, input_ids, attention_mask, pixel_values, image_grid_thw,
                   training: bool = False, opr: Callable = Ident, taps: Optional[dict] = None,
                   mean_hidden_state: bool = False) -> torch.Tensor:
    from llava_reward_amd.synth import qwen_rope_index
    input_ids = torch.as_tensor(input_ids)
    attention_mask = torch.as_tensor(attention_mask)
    pixel_values = torch.as_tensor(pixel_values, dtype=torch.float32)
    grid = [tuple(int(v) for v in g) for g in torch.as_tensor(image_grid_thw).tolist()]
    B, S = input_ids.shape
    x = W["model.embed_tokens.weight"][input_ids].clone()
    rows = vision_tower(W, cfg.vision, pixel_values, grid, opr, taps)
    slot = input_ids == cfg.image_token_id
    assert int(slot.sum()) == rows.shape[0], "Image features and image tokens do not match"
    x[slot] = rows                                           # masked_scatter, row-major order
    embeds = x.clone()                                       # hidden_states[0]
    if taps is not None:
        taps["image_rows"], taps["embeds"] = rows, embeds
    pos3 = torch.from_numpy(qwen_rope_index(input_ids.numpy(), attention_mask.numpy(), grid, cfg))
    cos, sin = mrope_cos_sin(pos3, cfg)
    mask4d = po.causal_padding_mask(attention_mask)
    for l in range(cfg.layers):
        x = decoder_layer(W, l, x, mask4d, cos, sin, cfg, opr)
        if taps is not None:
            taps[f"layer{l}"] = x.clone()
    h = po.rms_norm(x, W["model.norm.weight"], cfg.rms_eps)                                    # hidden_states[-1]
    if cfg.add_cross_attention:
        h = skip_ca(W, cfg, h, embeds, input_ids)
    if mean_hidden_state:                                    # rw_model:398-406
        return F.linear(po.mean_pool(h, attention_mask), W["value_head.weight"])
    values = F.linear(h, W["value_head.weight"])
    if training:
        return values[:, -1, :]
    eos = S - 1 - attention_mask.long().fliplr().argmax(dim=1)
    return values[torch.arange(B), eos, :]
